#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
(timeout 1500 python -m pytest tests/test_gpu_sampled.py -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4) > gpurun_out/r6/run9_tests.log
cat gpurun_out/r6/run9_tests.log
for m in "det 0" "det 0" "spec 65536"; do timeout 300 python tools/r6/lab_c2.py $m 2>&1 | tail -2; done | tee gpurun_out/r6/run9_lab.log
