#!/bin/bash
# rocprofv3 kernel stats of the block-composed path at C4's slab size
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_blk; mkdir -p $R/gpurun_out/prof_blk
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_blk/raw -o r -- python $R/bench.py --algo composed --field ${1:-default} --dtype f64 --shape 128,1024,1024 --eb 1e-6 --steps 5 --warmup 2 --no-cpu-baseline --no-host-e2e > $R/gpurun_out/prof_blk/bench.log 2>&1
cp $R/gpurun_out/prof_blk/raw/*kernel_stats.csv $R/gpurun_out/prof_blk/kernel_stats.csv
python - <<PY
import csv
for r in list(csv.DictReader(open("$R/gpurun_out/prof_blk/kernel_stats.csv")))[:14]:
    print("%-70s calls %4s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1000))
PY
rm -rf $R/gpurun_out/prof_blk/raw
