#!/bin/bash
# rocprofv3 kernel trace of the default bench (C2) -> gpurun_out/prof_c2/kernel_stats.csv (+ the bench line under the profiler)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-e2e --no-extra --no-cold"
rm -rf $R/gpurun_out/prof_c2; mkdir -p $R/gpurun_out/prof_c2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2/raw -o r -- $B > $R/gpurun_out/prof_c2/bench.log 2>&1
cp $R/gpurun_out/prof_c2/raw/*kernel_stats.csv $R/gpurun_out/prof_c2/kernel_stats.csv 2>/dev/null
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/prof_c2/raw/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# one step of the timed loop: from the packer's launch of one step (the step's last kernel) to the next one
idx=[i for i,r in enumerate(rows) if "k_pack<" in r["Kernel_Name"]]
a,b=idx[-6],idx[-5]
t0=int(rows[a+1]["Start_Timestamp"])
prev_end=None
for r in rows[a+1:b+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev_end)/1000 if prev_end else 0
    print("%-60s start %8.1f us  dur %7.1f us  gap %5.1f" % (r["Kernel_Name"][:60], (s-t0)/1000, (e-s)/1000, gap))
    prev_end=e
print("step span %.1f us" % ((prev_end-t0)/1000))
PY
rm -rf $R/gpurun_out/prof_c2/raw
