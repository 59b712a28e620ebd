#!/bin/bash
# rocprofv3 kernel trace of the C3 bench (interpolation, abs 1e-4) -> gpurun_out/prof_c3/kernel_stats.csv; extra bench args: "$@"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --algo interp --eb 1e-4 --steps 10 --warmup 3 --no-cpu-baseline --no-host-e2e --no-extra $@"
rm -rf $R/gpurun_out/prof_c3; mkdir -p $R/gpurun_out/prof_c3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3/raw -o r -- $B > $R/gpurun_out/prof_c3/bench.log 2>&1
cp $R/gpurun_out/prof_c3/raw/*kernel_stats.csv $R/gpurun_out/prof_c3/kernel_stats.csv 2>/dev/null
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_c3/kernel_stats.csv")))
for r in rows[:16]:
    print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1000))
import glob
f=glob.glob("$R/gpurun_out/prof_c3/raw/*kernel_trace.csv")[0]
tr=list(csv.DictReader(open(f)))
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
lv=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000 for r in tr if "k_interp_level<float, false>" in r["Kernel_Name"] or "k_interp_level<double, false>" in r["Kernel_Name"]]
print("level launches of the last step (us):", lv[-5:] if len(lv)>=5 else lv)
lv=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000 for r in tr if "k_interp_level<float, true>" in r["Kernel_Name"]]
print("decoder level launches (us):", lv[-5:] if len(lv)>=5 else lv)
PY
grep -m1 '^{"metric"' $R/gpurun_out/prof_c3/bench.log | head -c 700
rm -rf $R/gpurun_out/prof_c3/raw
