#!/bin/bash
# block-composed predictor, 1-D / 2-D: timings + kernel trace (gpurun_out/blkn/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/blkn; mkdir -p $O
for cfg in "1048576 1e-3" "134217728 1e-3" "8192,8192 1e-3" "8192,8192 0.15" "8192,8192 1e-3 f32 plain"; do
  timeout 300 python $R/tools/blkn_bench.py $cfg 2>&1 | tail -1
done | tee $O/times.txt
for tag in 1d 2d; do
  cfg="134217728 1e-3"; [ $tag = 2d ] && cfg="8192,8192 0.15"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o p -- python $R/tools/blkn_bench.py $cfg > $O/prof_$tag.log 2>&1
  f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee $O/stats_$tag.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-60s calls %5s avg %9.1f us total %8.2f ms %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
  rm -rf $O/prof_$tag
done
