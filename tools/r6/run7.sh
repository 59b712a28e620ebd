#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
for v in composed lorenzo; do
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o r -- python $R/tools/r6/c1_tl.py $v > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log
python - <<PY | tee $R/gpurun_out/r6/c1_timeline_$v.txt
import csv,glob
f=glob.glob("/tmp/tl/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_publish" in r["Kernel_Name"]]
a,b=idx[-4],idx[-3]
t0=int(rows[a+1]["Start_Timestamp"]); end=0
for r in rows[a+1:b+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-58s start %8.1f us  dur %7.1f us  gap %6.1f" % (r["Kernel_Name"][:58], (s-t0)/1000, (e-s)/1000, (s-end)/1000 if end else 0))
    end=max(end,e)
print("step span %.1f us; gap to next %.1f" % ((end-t0)/1000, (int(rows[b+1]["Start_Timestamp"])-end)/1000))
PY
done
