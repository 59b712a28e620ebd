#!/bin/bash
# kernel timeline of one late step of a python script that ends every step with k_publish: tools/r6/tl_py.sh <script> [args]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tlp; rocprofv3 --kernel-trace --output-format csv -d /tmp/tlp -o r -- python "$@" > /tmp/tlp.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" /tmp/tlp.log | tail -2
python - <<PY
import csv,glob
f=glob.glob("/tmp/tlp/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_publish" in r["Kernel_Name"]]
a,b=idx[-4],idx[-3]
t0=int(rows[a+1]["Start_Timestamp"]); end=0
for r in rows[a+1:b+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-58s q%-2s start %8.1f dur %7.1f gap %6.1f" % (r["Kernel_Name"].replace("(anonymous namespace)::","")[:58], r.get("Queue_Id","?"),(s-t0)/1000,(e-s)/1000,(s-end)/1000 if end else 0)); end=max(end,e)
print("span %.1f us" % ((end-t0)/1000))
PY
