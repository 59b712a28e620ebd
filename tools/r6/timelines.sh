#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
LASTK=k_publish BACK=3 bash tools/tl_case.sh --algo interp --eb 1e-4 > gpurun_out/r6/tl_c3.txt 2>&1
LASTK=k_publish BACK=2 bash tools/tl_case.sh --algo composed --dtype f64 --shape 128,1024,1024 --eb 1e-6 --field c4a > gpurun_out/r6/tl_c4a.txt 2>&1
# decoder timelines: kernel stats of the decode loops
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-e2e --no-extra --no-cold --no-live-traffic > /tmp/pd.log 2>&1
python - <<PY > $R/gpurun_out/r6/dec_c2_kernels.txt
import csv,glob
f=glob.glob("/tmp/pd/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:25]:
    print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1000))
PY
cat $R/gpurun_out/r6/tl_c3.txt | tail -40
