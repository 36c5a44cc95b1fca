#!/bin/bash
# Build libebcsim from a SNAPSHOT of the sources: a compile reads its headers when it starts and writes its object
# minutes later, so editing a header meanwhile leaves an object that `make` believes is current (seen: a kernel missing
# from the device code of a "fresh" build).  usage: tools/build_snapshot.sh [make targets...]   (default: all fault)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
snap=$(mktemp -d /tmp/ebc_build.XXXXXX)
mkdir -p $snap/eb-cadrl_amd $snap/include
cp -r $root/eb-cadrl_amd/csrc $snap/eb-cadrl_amd/csrc
cp $root/include/ebcsim.h $snap/include/
targets=${@:-all fault}
for t in $targets; do
  make -C $snap/eb-cadrl_amd/csrc $t
done
mkdir -p $root/eb-cadrl_amd/lib
cp $snap/eb-cadrl_amd/lib/*.so $root/eb-cadrl_amd/lib/
rm -rf $snap
echo "snapshot build done: $targets"
