#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
echo default; timeout 300 python tools/host_lab2.py 2>&1 | grep -v amdgpu.ids | grep -v "piece [1-6]" | tail -16
echo zstd16; SZ3HIP_ZSTD_THREADS=16 timeout 300 python tools/host_lab2.py 2>&1 | grep "iter" | tail -12
echo p0; SZ3HIP_PIECES=0 timeout 300 python tools/host_lab2.py 2>&1 | grep "iter" | tail -12
timeout 900 python -m pytest tests/test_gpu_pieces.py tests/test_gpu_multislab.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5
