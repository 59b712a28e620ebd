#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
for c in c1 c3 2d; do python tools/r6/cb_phases.py $c 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; done | tee gpurun_out/r6/run10_cb.log
