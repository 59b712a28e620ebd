#!/bin/bash
# kernels of one forward filter call on a 4 MB chunk (the 10th of 12): sum of durations and span
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for args in "1d default" "3d default" "1d lorenzo_reg"; do
rm -rf /tmp/h5; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/h5 -o r -- python $R/tools/r6/h5z_chunk.py $args > /tmp/h5.log 2>&1
grep "h5z chunk" /tmp/h5.log
python - <<PY
import csv,glob
f=glob.glob("/tmp/h5/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
# calls are separated by long idle gaps (host work between): cut where the gap exceeds 300 us... take the third group from the end
groups=[[rows[0]]]
for p,r in zip(rows,rows[1:]):
    if int(r["Start_Timestamp"])-int(p["End_Timestamp"])>250000: groups.append([])
    groups[-1].append(r)
g=groups[-4] if len(groups)>4 else groups[-1]
t0=int(g[0]["Start_Timestamp"]); end=max(int(r["End_Timestamp"]) for r in g)
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in g)
print("  kernels of one call: %d launches, busy %.1f us, span %.1f us (groups found: %d)" % (len(g), busy/1000, (end-t0)/1000, len(groups)))
for r in g: print("    %-60s %7.1f" % (r["Kernel_Name"].replace("(anonymous namespace)::","")[:60], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000))
PY
done 2>&1 | tee $R/gpurun_out/r6/h5z_tl.log
