#!/bin/bash
# A/B library for the streamed attention block: ebcsim_vn_stream.hip compiled with extra flags, linked with the kept
# objects of the main build (tools/build_snapshot.sh all first).  usage: tools/build_vns_variant.sh <name> <flags...>
#   -> eb-cadrl_amd/lib/libebcsim_<name>.so   (select with EBCSIM_LIB=...)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
snap=$(mktemp -d /tmp/ebc_vns.XXXXXX)
mkdir -p $snap/eb-cadrl_amd $snap/include
cp -r $root/eb-cadrl_amd/csrc $snap/eb-cadrl_amd/csrc
cp $root/include/ebcsim.h $snap/include/
cd $snap/eb-cadrl_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero \
  -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=14 "$@" -c -o $snap/vns.o ebcsim_vn_stream.hip
/opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o $root/eb-cadrl_amd/lib/libebcsim_$name.so \
  $root/eb-cadrl_amd/build/main_sim.o $root/eb-cadrl_amd/build/main_vn.o $snap/vns.o
rm -rf $snap
echo "built libebcsim_$name.so"
