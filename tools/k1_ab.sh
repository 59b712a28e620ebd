#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/lab
{ timeout 300 python tools/k1_lab.py 0 2>&1 | grep -v amdgpu.ids
for v in ${VARIANTS}; do SZ3HIP_LIB=$R/sz3_amd/lab/libsz3hip_$v.so timeout 300 python tools/k1_lab.py 0 2>&1 | grep -v amdgpu.ids; done
timeout 300 python tools/k1_lab.py 0 2>&1 | grep -v amdgpu.ids; } | tee gpurun_out/lab/k1_ab.txt
