#!/bin/bash
# kernel statistics of one ab_cases.py case: AB_ONLY="3d 500^3 lorenzo" bash tools/prof_case.sh
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pcase; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pcase -o r -- python $GRAFT_REPO_ROOT/tools/ab_cases.py > /tmp/pcase.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/pcase/r_kernel_stats.csv")))
for r in rows[:18]:
    if "at::native" in r["Name"]: continue
    print("%-100s calls %4s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1000))
PY
grep compress /tmp/pcase.log
