#!/bin/bash
# C1 (1-D 2^20 f32, ALGO_LORENZO_REG defaults): kernel trace of a compress + decompress loop (gpurun_out/c1_stats.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rm -rf /tmp/pc1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc1 -o p -- python $R/tools/blkn_bench.py 1048576 1e-3 f32 ${1:-} > $O/c1.log 2>&1
f=$(find /tmp/pc1 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/c1_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = 0
for r in rows[:40]:
    print("%-62s calls %5s avg %8.1f us total %7.2f ms" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
grep -v rocprof $O/c1.log | tail -1
