#!/bin/bash
# round 5, second half: the whole GPU suite, smoke, the bench line as the driver runs it, rocprofv3 kernel statistics + PMC passes (C2 and C3), timelines
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6) > gpurun_out/r05b_gputests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r05b_smoke.log
timeout 1500 python bench.py > gpurun_out/r05b_bench_default.json 2> gpurun_out/r05b_bench_default.err
timeout 1500 bash tools/pmc.sh > gpurun_out/r05b_pmc.log 2>&1
timeout 900 bash tools/pmc_c3.sh > gpurun_out/r05b_pmc_c3.log 2>&1
cp gpurun_out/pmc_summary.txt gpurun_out/r05b_pmc_summary.txt; cp gpurun_out/pmc_summary_c3.txt gpurun_out/r05b_pmc_summary_c3.txt
cp gpurun_out/kernel_stats.csv gpurun_out/r05b_kernel_stats.csv; cp gpurun_out/kernel_stats_c3.csv gpurun_out/r05b_kernel_stats_c3.csv
NTH=12 LASTK=k_publish bash tools/tl_case.sh > gpurun_out/r05b_timeline_c2.txt 2>&1
NTH=9 LASTK=k_publish bash tools/tl_case.sh --algo interp --eb 1e-4 > gpurun_out/r05b_timeline_c3.txt 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_summary.txt > gpurun_out/r05b_pmc_traffic.log 2>&1
cat gpurun_out/r05b_gputests.log gpurun_out/r05b_smoke.log; tail -c 400 gpurun_out/r05b_bench_default.err; head -c 600 gpurun_out/r05b_bench_default.json; echo; tail -8 gpurun_out/r05b_timeline_c2.txt
