#!/bin/bash
# durations of the 3-D group fronts (k_blk_decode_g) by front size: C4a slab, one decompress
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pbd; rocprofv3 --kernel-trace --output-format csv -d /tmp/pbd -o r -- python $R/bench.py --algo composed --field c4a --dtype f64 --shape 128,1024,1024 --eb 1e-6 --steps 2 --warmup 1 --no-cpu-baseline --no-host-e2e --no-cold > /dev/null 2>&1
python - <<PY
import csv,glob
rows=[r for r in csv.DictReader(open(glob.glob("/tmp/pbd/*kernel_trace.csv")[0])) if "k_blk_decode_g" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-181:]
t0=int(rows[0]["Start_Timestamp"])
for i,r in enumerate(rows):
    if i%10==0 or i>170:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        print("front %3d grid %5s start %8.1f dur %6.1f us" % (i, r.get("Grid_Size_X", r.get("Grid_Size","?")), (s-t0)/1000, (e-s)/1000))
print("span %.1f us" % ((int(rows[-1]["End_Timestamp"])-t0)/1000))
PY
