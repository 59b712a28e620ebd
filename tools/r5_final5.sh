#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3) > gpurun_out/r05e_gputests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) > gpurun_out/r05e_smoke.log
cat gpurun_out/r05e_gputests.log gpurun_out/r05e_smoke.log
