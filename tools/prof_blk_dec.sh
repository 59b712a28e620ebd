#!/bin/bash
# timeline of one block-stream decompress (C4a slab): which kernels overlap
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pbd; rocprofv3 --kernel-trace --output-format csv -d /tmp/pbd -o r -- python $R/bench.py --algo composed --field c4a --dtype f64 --shape 128,1024,1024 --eb 1e-6 --steps 2 --warmup 1 --no-cpu-baseline --no-host-e2e --no-cold > /dev/null 2>&1
python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("/tmp/pbd/*kernel_trace.csv")[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last decompress: from the last k_dec_tables on
idx=[i for i,r in enumerate(rows) if "k_dec_tables" in r["Kernel_Name"]]
a=idx[-1]
# include side kernels launched just before
a=max(0,a-8)
t0=int(rows[a]["Start_Timestamp"])
shown=0
for r in rows[a:]:
    n=r["Kernel_Name"]
    if "k_blk_decode_g" in n and shown>14: continue
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-46s start %9.1f  end %9.1f  dur %8.1f us  queue %s" % (n.split("(")[0][-46:], (s-t0)/1000, (e-t0)/1000, (e-s)/1000, r.get("Queue_Id","?")))
    shown+=1
    if shown>40: break
PY
