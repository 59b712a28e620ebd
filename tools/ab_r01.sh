cd /tmp && export TMPDIR=/tmp
for d in $GRAFT_REPO_ROOT/_r01 $GRAFT_REPO_ROOT; do
rm -rf /tmp/pab; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pab -o r -- python $d/bench.py --algo lorenzo --dtype f64 --shape 128,1024,1024 --eb 1e-6 --steps 10 --warmup 3 --no-cpu-baseline --no-host-e2e > /tmp/pab.log 2>&1
echo "== $d"; python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/pab/r_kernel_stats.csv")))
for r in rows[:12]:
    if "at::native" in r["Name"]: continue
    print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1000))
PY
grep -o "\"stage_ms\": {[^}]*}" /tmp/pab.log | head -1
done
