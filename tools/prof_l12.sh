#!/bin/bash
# 1-D, the tuner's set [Lorenzo-1, Lorenzo-2] in blocks of 128 at 2^27 values: kernel trace (gpurun_out/l12_stats.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rm -rf /tmp/pl12
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl12 -o p -- python $R/tools/blkn_bench.py ${1:-134217728} ${2:-1e-3} ${3:-f32} ${4:-l12} > $O/l12.log 2>&1
f=$(find /tmp/pl12 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/l12_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    print("%-70s calls %5s avg %9.1f us total %8.2f ms %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
tail -1 $O/l12.log
