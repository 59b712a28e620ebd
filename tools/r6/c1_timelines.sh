#!/bin/bash
# C1 timelines: composed (bench C1) and the default algorithm
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r6
for v in composed default; do
  rm -rf /tmp/tl1; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl1 -o r -- python $R/tools/r6/c1_tl.py $v > /tmp/tl1.log 2>&1
  echo "== $v"; grep "C1 " /tmp/tl1.log
  python - <<PY
import csv,glob
f=glob.glob("/tmp/tl1/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_publish" in r["Kernel_Name"]]
a,b=idx[-3],idx[-2]
t0=int(rows[a+1]["Start_Timestamp"]); end=0
for r in rows[a+1:b+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-60s start %7.1f dur %6.1f gap %5.1f" % (r["Kernel_Name"].replace("(anonymous namespace)::","")[:60],(s-t0)/1000,(e-s)/1000,(s-end)/1000 if end else 0)); end=max(end,e)
print("span %.1f us" % ((end-t0)/1000))
PY
done 2>&1 | tee $R/gpurun_out/r6/run16.log
cd $R
python -m pytest tests/test_gpu_small.py tests/test_gpu_regression_lowdim.py -x -q -rs 2>&1 | tail -6
