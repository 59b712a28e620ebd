#!/bin/bash
# tests/checks/wild_data_sweep.py under a few seeds (data classes the other sweeps do not draw)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|OpenMP enabled'
for sd in ${SEEDS:-1 2 3}; do
  SEED=$sd N=${NCASES:-150} timeout 1200 python tests/checks/wild_data_sweep.py > /tmp/w.log 2>&1; rc=$?
  echo "seed $sd: rc $rc | $(grep -v "$F" /tmp/w.log | tail -1)"
  grep -v "$F" /tmp/w.log | grep "FAIL\|EXC\|MISMATCH\|skipped\|Traceback\|Error" | head -40
done 2>&1 | tee gpurun_out/r6/wild.log
