#!/bin/bash
# builds variants of one kernel translation unit for A/B timing on the GPU box: sz3_amd/lab/libsz3hip_<name>.so
# usage: tools/build_lab.sh [-s source.hip] name "-DFOO=1 -DBAR" [name2 "flags2" ...]   (the other objects come from sz3_amd/build)
set -e
cd "$(dirname "$0")/.."
mkdir -p sz3_amd/lab
SRC=sz3hip_kernels.hip
if [ "$1" = "-s" ]; then SRC=$2; shift 2; fi
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
OBJS=""
for f in sz3hip_kernels.hip sz3hip_interp.hip sz3hip_regress.hip sz3hip_stock.hip sz3hip_sortlists.hip sz3hip_api.cpp sz3hip_host.cpp sz3hip_stock_host.cpp sz3hip_comm.cpp sz3hip_h5z.cpp; do
  [ "$f" = "$SRC" ] || OBJS="$OBJS sz3_amd/build/$f.o"
done
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c sz3_amd/csrc/$SRC -o sz3_amd/lab/k_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC sz3_amd/lab/k_$name.o $OBJS -o sz3_amd/lab/libsz3hip_$name.so -ldl -lpthread &&
    rm -f sz3_amd/lab/k_$name.o && echo built $name ) &
done
wait
