#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
(python tools/r6/cb_phases.py c1 2>&1 | grep -v "$F"; echo "== 128^3"
SZ3_LAB_FLAGS=65536 LAB_SIZE=128 python tools/cb_lab.py 2>&1 | grep -v "$F"
SZ3_LAB_FLAGS=$((65536+262144)) LAB_SIZE=128 python tools/cb_lab.py 2>&1 | grep -v "$F") | tee gpurun_out/r6/run19.log
