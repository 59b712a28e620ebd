#!/bin/bash
# builds variants of the kernels translation unit for A/B timing on the GPU box: sz3_amd/lab/libsz3hip_<name>.so
# usage: tools/build_lab.sh name "-DFOO=1 -DBAR" [name2 "flags2" ...]   (the other objects come from sz3_amd/build)
set -e
cd "$(dirname "$0")/.."
mkdir -p sz3_amd/lab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c sz3_amd/csrc/sz3hip_kernels.hip -o sz3_amd/lab/k_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC sz3_amd/lab/k_$name.o sz3_amd/build/sz3hip_interp.hip.o sz3_amd/build/sz3hip_regress.hip.o \
      sz3_amd/build/sz3hip_api.cpp.o sz3_amd/build/sz3hip_host.cpp.o sz3_amd/build/sz3hip_comm.cpp.o -o sz3_amd/lab/libsz3hip_$name.so -ldl -lpthread &&
    rm -f sz3_amd/lab/k_$name.o && echo built $name ) &
done
wait
