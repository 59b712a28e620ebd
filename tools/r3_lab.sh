#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/lab
for v in ${VARIANTS:-nopf m1 m1pf m2 ty8 ty2 tz32}; do SZ3HIP_LIB=$R/sz3_amd/lab/libsz3hip_$v.so timeout 300 python tools/k1_lab.py 0 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/lab/k1_variants2.txt
