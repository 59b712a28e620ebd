#!/bin/bash
# end-of-round GPU visit: GPU suite, smoke, the default bench line (driver's command), kernel stats + timeline C2, C3, C4 composed (both
# fields), decoder kernels, PMC passes (C2 traffic -> profiles/pmc_traffic.json inputs; the selection pass) -> gpurun_out/r3final/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3final; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc " $O/pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
timeout 600 bash tools/prof_c2.sh > $O/timeline_c2.txt 2>&1; cp gpurun_out/prof_c2/kernel_stats.csv $O/kernel_stats_c2.csv
timeout 600 bash tools/prof_c3.sh > $O/prof_c3.txt 2>&1; cp gpurun_out/prof_c3/kernel_stats.csv $O/kernel_stats_c3.csv
timeout 600 bash tools/c4_run.sh default > $O/c4_default.txt 2>&1; cp gpurun_out/prof_blk/kernel_stats.csv $O/kernel_stats_c4_composed.csv; cp gpurun_out/prof_blk/bench_noprof.json $O/bench_c4_composed.json
timeout 600 bash tools/c4_run.sh c4a > $O/c4_c4a.txt 2>&1; cp gpurun_out/prof_blk/kernel_stats.csv $O/kernel_stats_c4a_composed.csv; cp gpurun_out/prof_blk/bench_noprof.json $O/bench_c4a_composed.json
VARIANTS=default timeout 600 bash tools/prof_dec.sh > $O/decoder_kernels.txt 2>&1
timeout 900 bash tools/pmc.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_summary.txt $O/pmc_summary.txt
for d in pmc_fetch pmc_write; do cp gpurun_out/$d/*counter_collection.csv $O/${d}_counters.csv 2>/dev/null; done
timeout 600 bash tools/pmc_blk.sh > $O/pmc_blk_select.txt 2>&1
tail -5 $O/c4_default.txt; tail -3 $O/c4_c4a.txt
timeout 900 bash tools/blkn_run.sh > $O/blkn.txt 2>&1; cp gpurun_out/blkn/stats_1d.txt $O/blkn_kernels_1d.txt; cp gpurun_out/blkn/stats_2d.txt $O/blkn_kernels_2d.txt; cp gpurun_out/blkn/times.txt $O/blkn_times.txt
cat $O/blkn_times.txt
