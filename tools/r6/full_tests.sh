#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
(timeout 3000 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_sampled.py --deselect tests/test_gpu_packb.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -60) > gpurun_out/r6/full_tests.log
tail -5 gpurun_out/r6/full_tests.log
