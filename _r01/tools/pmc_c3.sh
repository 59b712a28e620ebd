#!/bin/bash
# PMC passes for the interpolation path (C3: ALGO_INTERP_LORENZO at 1e-4), one counter group per pass like tools/pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B3="python $R/bench.py --algo interp --eb 1e-4 --steps 10 --warmup 2 --no-cpu-baseline --no-host-e2e"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc3_fetch -o p -- $B3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc3_write -o p -- $B3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc3_sq -o p -- $B3 > /dev/null 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc3_fetch/*counter_collection.csv $R/gpurun_out/pmc3_write/*counter_collection.csv $R/gpurun_out/pmc3_sq/*counter_collection.csv > $R/gpurun_out/pmc_summary_c3.txt 2>&1
head -50 $R/gpurun_out/pmc_summary_c3.txt
