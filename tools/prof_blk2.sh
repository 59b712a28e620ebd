#!/bin/bash
# 2-D block stream, 8192^2 f32: decompress kernel by kernel (gpurun_out/blk2_stats.txt); args: eb
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rm -rf /tmp/pb2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb2 -o p -- python $R/tools/blkn_bench.py 8192,8192 ${1:-0.15} > $O/blk2.log 2>&1
f=$(find /tmp/pb2 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/blk2_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-64s calls %5s avg %9.1f us total %8.2f ms %5s %%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
tail -1 $O/blk2.log
