#!/bin/bash
# kernel timeline of the C1 step and of its decoder (tools/c1_step.py under rocprofv3 --kernel-trace): the kernels of the 15th step of the timed loop
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/c1tl; mkdir -p /tmp/c1tl
python $R/tools/c1_step.py
rocprofv3 --kernel-trace --output-format csv -d /tmp/c1tl/raw -o r -- python $R/tools/c1_step.py > /tmp/c1tl/log 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/c1tl/raw/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
# compression steps: cut at the step's first kernel (the one that follows the longest recurring period): print steps by the period of the name sequence
def show(lo, hi, title):
    print(title)
    t0=int(rows[lo]["Start_Timestamp"]); end=0
    for r in rows[lo:hi]:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        print("  %-60s start %7.1f us  dur %6.1f us  gap %5.1f" % (r["Kernel_Name"].replace("(anonymous namespace)::","")[:60], (s-t0)/1000, (e-s)/1000, (s-end)/1000 if end else 0))
        end=max(end,e)
    print("  span %.1f us, %d kernels, busy %.1f us" % ((end-t0)/1000, hi-lo, sum((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000 for r in rows[lo:hi])))
# the decoder's steps are the trailing ones: find the period at the end of the trace
def period(seq):
    for p in range(1, 80):
        if len(seq) >= 3*p and seq[-p:] == seq[-2*p:-p] == seq[-3*p:-2*p]: return p
    return None
pd=period(names)
if pd: show(len(rows)-2*pd, len(rows)-pd, "decoder step (%d kernels)" % pd)
# compression: strip the decoder's 23 steps, then the period again
if pd:
    cut=len(rows)
    while cut-pd>=0 and names[cut-pd:cut]==names[len(rows)-pd:]: cut-=pd
    idx=[i for i in range(cut) if "k_publish" in names[i]]  # (the compression step's last kernel)
    if len(idx) > 6: show(idx[-6]+1, idx[-5]+1, "compression step (%d kernels)" % (idx[-5]-idx[-6]))
PY
cat /tmp/c1tl/log | grep "C1 step"
