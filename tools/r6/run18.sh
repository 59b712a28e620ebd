#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
(python tools/r6/cb_phases.py c1 2>&1 | grep -v "$F"
echo "== 128^3 Lorenzo 1e-3 (129 symbols), wave32 merge then one-wave rounds"
LAB_SIZE=128 python tools/cb_lab.py 2>&1 | grep -v "$F"
LAB_SIZE=128 SZ3_LAB_FLAGS=262144 python tools/cb_lab.py 2>&1 | grep -v "$F"
python tools/r6/c1_tl.py 2>&1 | grep -v "$F") | tee gpurun_out/r6/run18.log
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -8) | tee gpurun_out/r6/run18_tests.log
