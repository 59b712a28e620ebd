#!/bin/bash
# the randomised checks of tests/checks with seeds the suite does not use (exit codes = mismatches)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
for sd in ${SEEDS:-101 202 303}; do
  for t in block_sweep lorenzo_sweep interp_sweep host_sweep stock_bytes_sweep default_algo_recon_sweep; do
    SEED=$sd N=${NCASES:-40} timeout 1500 python tests/checks/$t.py > /tmp/sw.log 2>&1; rc=$?
    echo "seed $sd $t: rc $rc | $(grep -v "$F" /tmp/sw.log | tail -1)"
    if [ $rc -ne 0 ]; then grep -v "$F" /tmp/sw.log | grep -i "mismatch\|error\|fail" | head -5; fi
  done
done 2>&1 | tee gpurun_out/r6/extra_sweeps.log
