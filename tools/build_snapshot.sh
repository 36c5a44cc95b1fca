#!/bin/bash
# Build libebcsim from a SNAPSHOT of the sources: a compile reads its headers when it starts and writes its object
# minutes later, so editing a header meanwhile leaves an object that `make` believes is current (seen: a kernel missing
# from the device code of a "fresh" build).  Objects are kept between builds (eb-cadrl_amd/build) and carried into the
# snapshot with their time stamps; what a build makes is stamped with the SNAPSHOT's time before it comes back, so an
# edit made while it ran is newer than the object and rebuilds it next time.
# usage: tools/build_snapshot.sh [make targets...]   (default: all fault; FRESH=1: no kept objects)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
snap=$(mktemp -d /tmp/ebc_build.XXXXXX)
stamp=$snap/.stamp
touch $stamp
mkdir -p $snap/eb-cadrl_amd $snap/include
cp -rp $root/eb-cadrl_amd/csrc $snap/eb-cadrl_amd/csrc
cp -p $root/include/ebcsim.h $snap/include/
if [ -z "$FRESH" ] && [ -d $root/eb-cadrl_amd/build ]; then cp -rp $root/eb-cadrl_amd/build $snap/eb-cadrl_amd/build; fi
targets=${@:-all fault}
for t in $targets; do
  make -C $snap/eb-cadrl_amd/csrc $t
done
mkdir -p $root/eb-cadrl_amd/lib $root/eb-cadrl_amd/build
find $snap/eb-cadrl_amd/build -newer $stamp -type f -exec touch -r $stamp {} +
cp -p $snap/eb-cadrl_amd/build/* $root/eb-cadrl_amd/build/
cp $snap/eb-cadrl_amd/lib/*.so $root/eb-cadrl_amd/lib/
rm -rf $snap
echo "snapshot build done: $targets"
