#!/bin/bash
# one step's kernel timeline (start, duration, gap to the previous end, queue) of any bench configuration: bench args in "$@";
# the step is cut between two launches of the kernel named in $LASTK (default k_publish)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-e2e --no-extra --no-cold --no-live-traffic $@"
rm -rf /tmp/tlc; mkdir -p /tmp/tlc
rocprofv3 --kernel-trace --output-format csv -d /tmp/tlc/raw -o r -- $B > /tmp/tlc/bench.log 2>&1
python - <<PY
import csv,glob,os
f=glob.glob("/tmp/tlc/raw/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
lastk=os.environ.get("LASTK","k_publish")
idx=[i for i,r in enumerate(rows) if lastk in r["Kernel_Name"]]
back=int(os.environ.get("BACK","4"))
a,b=idx[-back-1],idx[-back]
if os.environ.get("NTH"):  # the n-th step from the start instead (warm-up + timed loop come first: no profiling events in those)
    k=int(os.environ["NTH"]); a,b=idx[k-1],idx[k]
t0=int(rows[a+1]["Start_Timestamp"])
end=0
for r in rows[a+1:b+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-end)/1000 if end else 0
    print("%-56s q%-3s start %8.1f us  dur %7.1f us  gap %6.1f" % (r["Kernel_Name"].replace("(anonymous namespace)::","")[:56], r.get("Queue_Id","?"), (s-t0)/1000, (e-s)/1000, gap))
    end=max(end,e)
print("step span %.1f us" % ((end-t0)/1000))
PY
grep -m1 '^{"metric"' /tmp/tlc/bench.log | head -c 400; echo
