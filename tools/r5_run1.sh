#!/bin/bash
# round 5, first batch: the new tests, the decoder retry, a timeline of the C2 step
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5
(timeout 300 python -m pytest tests/test_gpu_q16.py -x -q 2>&1 | tail -15) > gpurun_out/r5/q16_test.log
(timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -k stock 2>&1 | tail -30) > gpurun_out/r5/fuzz_stock.log
(timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -k own 2>&1 | tail -30) > gpurun_out/r5/fuzz_own.log
(timeout 900 python -m pytest tests/test_gpu_regression.py tests/test_gpu_regression_lowdim.py tests/test_gpu_stock.py -q -x -k "decoders_agree or stock" 2>&1 | tail -15) > gpurun_out/r5/dec_retry.log
timeout 600 bash tools/prof_c2.sh > gpurun_out/r5/timeline.txt 2>&1
cp gpurun_out/prof_c2/kernel_stats.csv gpurun_out/r5/kernel_stats_c2.csv 2>/dev/null
tail -5 gpurun_out/prof_c2/bench.log > gpurun_out/r5/bench_under_prof.log
for f in q16_test fuzz_stock fuzz_own dec_retry; do echo "== $f"; cat gpurun_out/r5/$f.log; done; echo "== timeline"; tail -12 gpurun_out/r5/timeline.txt
