#!/bin/bash
# kernel timeline of one warm C2 step: tools/r6/timeline.sh <mode> <flags> <out>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o r -- python $R/tools/r6/lab_c2.py $1 $2 12 > /tmp/tl.log 2>&1
python - <<PY | tee $R/gpurun_out/r6/$3
import csv,glob
f=glob.glob("/tmp/tl/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_publish" in r["Kernel_Name"]]
for back in (6,5):
    a,b=idx[-back-1],idx[-back]
    t0=int(rows[a+1]["Start_Timestamp"]); end=0
    for r in rows[a+1:b+1]:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        print("%-58s start %8.1f us  dur %7.1f us  gap %6.1f" % (r["Kernel_Name"][:58], (s-t0)/1000, (e-s)/1000, (s-end)/1000 if end else 0))
        end=max(end,e)
    print("step span %.1f us; gap to the next step's first kernel %.1f us" % ((end-t0)/1000, (int(rows[b+1]["Start_Timestamp"])-end)/1000))
PY
tail -1 /tmp/tl.log
