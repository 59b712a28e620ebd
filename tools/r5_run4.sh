#!/bin/bash
# round 5: the wide code book's code words over the whole chip (k_cb_assign) — affected suites, then the C3 kernel statistics
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_interp.py tests/test_gpu_parity.py tests/test_gpu_tuner.py tests/test_gpu_multislab.py -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r5_run4_tests.log
cat gpurun_out/r5_run4_tests.log
timeout 300 bash tools/prof_c3.sh > gpurun_out/r5_run4_c3.txt 2>&1; head -14 gpurun_out/r5_run4_c3.txt; tail -c 700 gpurun_out/r5_run4_c3.txt
