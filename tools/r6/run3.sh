#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
(timeout 1200 python -m pytest tests/test_gpu_sampled.py tests/test_gpu_packb.py -q 2>&1 | tail -25) > gpurun_out/r6/run3_tests.log
cat gpurun_out/r6/run3_tests.log
for m in "spec 0" "det 0" "spec 65536"; do timeout 300 python tools/r6/lab_c2.py $m 2>&1 | tail -2; done | tee gpurun_out/r6/run3_lab.log
bash tools/r6/timeline.sh det 0 run3_timeline_sampled.txt
bash tools/r6/timeline.sh spec 65536 run3_timeline_spec.txt
