#!/bin/bash
# kernel statistics of one bench configuration under the round-1 build (_r01/) and the current one
cd /tmp && export TMPDIR=/tmp
for d in $GRAFT_REPO_ROOT/_r01 $GRAFT_REPO_ROOT; do
rm -rf /tmp/pab; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pab -o r -- python $d/bench.py "$@" --steps 8 --warmup 3 --no-cpu-baseline --no-host-e2e > /tmp/pab.log 2>&1
echo "== $d"; python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/pab/r_kernel_stats.csv")))
for r in rows[:16]:
    if "at::native" in r["Name"]: continue
    print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1000))
PY
done
