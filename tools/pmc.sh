#!/bin/bash
# rocprofv3 passes for profiles/: (1) kernel stats of the bench command, (2)+(3) HBM traffic counters, each in its own
# pass (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 — MI355X_MICROARCH.md "rocprofv3 PMC slots"), (4) SQ counters.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-e2e --no-extra --no-cold --no-live-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r -- $B > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_sq -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_lds -o p -- $B > /dev/null 2>&1
grep metric $R/gpurun_out/prof_stats.log | cut -c1-300
# summaries for profiles/ (copied there by hand after a look)
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_fetch/*counter_collection.csv $R/gpurun_out/pmc_write/*counter_collection.csv $R/gpurun_out/pmc_sq/*counter_collection.csv $R/gpurun_out/pmc_lds/*counter_collection.csv > $R/gpurun_out/pmc_summary.txt 2>&1
cp $R/gpurun_out/prof_stats/*kernel_stats.csv $R/gpurun_out/kernel_stats.csv 2>/dev/null
# C3 (ALGO_INTERP_LORENZO) kernel stats
B3="python $R/bench.py --algo interp --eb 1e-4 --steps 10 --warmup 2 --no-cpu-baseline --no-host-e2e --no-cold --no-live-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_c3 -o r -- $B3 > $R/gpurun_out/prof_stats_c3.log 2>&1
cp $R/gpurun_out/prof_stats_c3/*kernel_stats.csv $R/gpurun_out/kernel_stats_c3.csv 2>/dev/null
