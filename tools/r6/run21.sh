#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
(python tools/r6/c1_tl.py default 2>&1 | grep -v "$F"; python tools/r6/c1_tl.py 2>&1 | grep -v "$F") | tee gpurun_out/r6/run21.log
(timeout 2400 python -m pytest tests/test_gpu_tuner.py tests/test_gpu_regression_lowdim.py tests/test_gpu_small.py tests/test_gpu_sweeps.py -x -q 2>&1 | grep -v "$F" | tail -6) | tee gpurun_out/r6/run21_tests.log
