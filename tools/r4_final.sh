#!/bin/bash
# round 4: the bench line as the driver runs it + rocprofv3 kernel statistics + PMC passes (default two-pass form, and the fused form via SZ3HIP_FUSED=1)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
bash tools/pmc.sh > gpurun_out/r04_pmc.log 2>&1
cp gpurun_out/pmc_summary.txt gpurun_out/r04_pmc_summary.txt; cp gpurun_out/kernel_stats.csv gpurun_out/r04_kernel_stats.csv; cp gpurun_out/kernel_stats_c3.csv gpurun_out/r04_kernel_stats_c3.csv
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-e2e --no-extra --no-cold --no-live-traffic"
export SZ3HIP_FUSED=1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fused -o r -- $B > $R/gpurun_out/r04_bench_fused.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_f -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_f -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_sq_f -o p -- $B > /dev/null 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_fetch_f/*counter_collection.csv $R/gpurun_out/pmc_write_f/*counter_collection.csv $R/gpurun_out/pmc_sq_f/*counter_collection.csv > $R/gpurun_out/r04_pmc_summary_fused.txt 2>&1
cp $R/gpurun_out/prof_fused/*kernel_stats.csv $R/gpurun_out/r04_kernel_stats_fused.csv
