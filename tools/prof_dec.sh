#!/bin/bash
# rocprofv3 kernel trace of tools/dec_lab.py (device-resident decompression) -> per-kernel durations of the last decode call
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_dec; mkdir -p $R/gpurun_out/prof_dec
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec/raw -o r -- python $R/tools/dec_lab.py > $R/gpurun_out/prof_dec/lab.log 2>&1
cp $R/gpurun_out/prof_dec/raw/*kernel_stats.csv $R/gpurun_out/prof_dec/kernel_stats.csv 2>/dev/null
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/prof_dec/raw/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("k_dec_tables")]
a=idx[-3]; b=idx[-2]
t0=int(rows[a]["Start_Timestamp"]); prev=None
for r in rows[a:b]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-70s start %8.1f dur %7.1f gap %5.1f" % (r["Kernel_Name"][:70], (s-t0)/1000, (e-s)/1000, (s-prev)/1000 if prev else 0))
    prev=e
PY
rm -rf $R/gpurun_out/prof_dec/raw
