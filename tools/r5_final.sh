#!/bin/bash
# round 5: the whole GPU suite, smoke, the bench line as the driver runs it, rocprofv3 kernel statistics + PMC passes
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r05_gputests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/r05_smoke.log
timeout 1500 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
timeout 1500 bash tools/pmc.sh > gpurun_out/r05_pmc.log 2>&1
cp gpurun_out/pmc_summary.txt gpurun_out/r05_pmc_summary.txt; cp gpurun_out/kernel_stats.csv gpurun_out/r05_kernel_stats.csv; cp gpurun_out/kernel_stats_c3.csv gpurun_out/r05_kernel_stats_c3.csv
timeout 600 bash tools/prof_c2.sh > gpurun_out/r05_timeline_c2.txt 2>&1
cat gpurun_out/r05_gputests.log gpurun_out/r05_smoke.log; tail -c 600 gpurun_out/r05_bench_default.err; head -c 400 gpurun_out/r05_bench_default.json; echo; tail -9 gpurun_out/r05_timeline_c2.txt
