#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
(timeout 1500 python -m pytest tests/test_gpu_sampled.py tests/test_gpu_packb.py tests/test_gpu_q16.py -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12) > gpurun_out/r6/run8_tests.log
cat gpurun_out/r6/run8_tests.log
for m in "det 0" "det 8"; do timeout 300 python tools/r6/lab_c2.py $m 2>&1 | tail -2; done | tee gpurun_out/r6/run8_lab.log
