#!/bin/bash
# k_decode's duration at an eighth of C2, at C2, and C3's decompress time
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for sh in 2,512,512 64,512,512 512,512,512; do
  rm -rf /tmp/dl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dl -o r -- python $R/tools/r6/dec_lab.py $sh > /tmp/dl.log 2>&1
  grep "^decompress" /tmp/dl.log
  python - <<PY
import csv,glob
f=glob.glob("/tmp/dl/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_decode" in r["Name"] and int(r["Calls"])>4 and float(r["AverageNs"])>10000: print("   %s: %.1f us (calls %s)" % (r["Name"][:50], float(r["AverageNs"])/1000, r["Calls"]))
PY
done
cd $R; python bench.py --algo interp --eb 1e-4 --steps 5 --warmup 2 --no-cpu-baseline --no-host-e2e --no-extra --no-cold --no-live-traffic 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): o=json.loads(l); print('C3 compress %.4f decompress %.4f ms' % (o['ms_per_step'], o['decompress_device']['ms']))"
