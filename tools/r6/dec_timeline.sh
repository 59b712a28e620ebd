#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r6
for sh in 64,512,512 512,512,512; do
rm -rf /tmp/dl; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/dl -o r -- python $R/tools/r6/dec_lab.py $sh > /tmp/dl.log 2>&1
grep "^decompress" /tmp/dl.log
python - <<PY
import csv,glob
rows=[]
for f in glob.glob("/tmp/dl/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("(anonymous namespace)::","")[:60]))
for f in glob.glob("/tmp/dl/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")+" "+r.get("Name","")))
rows.sort()
idx=[i for i,r in enumerate(rows) if "k_dec_tables" in r[2]]
a,b=idx[-3],idx[-2]
t0=rows[a][0]; end=0
for s,e,nm in rows[a:b]:
    print("  %-62s start %8.1f dur %7.1f gap %6.1f" % (nm,(s-t0)/1000,(e-s)/1000,(s-end)/1000 if end else 0)); end=max(end,e)
print("  call to call %.1f us" % ((rows[b][0]-t0)/1000))
PY
done 2>&1 | tee $R/gpurun_out/r6/run25.log
