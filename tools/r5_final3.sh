#!/bin/bash
# round 5, last state: suite, smoke, the bench line, C3's kernel statistics and timeline (the dense hand-over between the two finest levels came after r05b)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3) > gpurun_out/r05c_gputests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) > gpurun_out/r05c_smoke.log
timeout 1500 python bench.py > gpurun_out/r05c_bench_default.json 2> gpurun_out/r05c_bench_default.err
timeout 600 bash tools/prof_c3.sh > gpurun_out/r05c_prof_c3.txt 2>&1; cp gpurun_out/prof_c3/kernel_stats.csv gpurun_out/r05c_kernel_stats_c3.csv
NTH=9 LASTK=k_publish bash tools/tl_case.sh --algo interp --eb 1e-4 > gpurun_out/r05c_timeline_c3.txt 2>&1
timeout 900 bash tools/pmc_c3.sh > gpurun_out/r05c_pmc_c3.log 2>&1; cp gpurun_out/pmc_summary_c3.txt gpurun_out/r05c_pmc_summary_c3.txt
cat gpurun_out/r05c_gputests.log gpurun_out/r05c_smoke.log; head -c 400 gpurun_out/r05c_bench_default.json; echo; tail -16 gpurun_out/r05c_timeline_c3.txt | cut -c1-200
