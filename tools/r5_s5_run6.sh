#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_interp.py tests/test_gpu_tuner.py tests/test_gpu_parity.py tests/test_gpu_sweeps.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
NTH=9 LASTK=k_publish bash tools/tl_case.sh --algo interp --eb 1e-4 2>&1 | tail -12
