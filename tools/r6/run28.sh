#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r6
for sh in 2,512,512 64,512,512 512,512,512; do
  rm -rf /tmp/dl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dl -o r -- python $R/tools/r6/dec_lab.py $sh > /tmp/dl.log 2>&1
  grep "^decompress" /tmp/dl.log
  python - <<PY
import csv,glob
f=glob.glob("/tmp/dl/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_decode" in r["Name"] and int(r["Calls"])>4 and float(r["AverageNs"])>10000: print("   %s: %.1f us (calls %s)" % (r["Name"][:50], float(r["AverageNs"])/1000, r["Calls"]))
PY
done 2>&1 | tee $R/gpurun_out/r6/run28.log
cd $R; python tools/r6/c3_lab.py 0 5 2>&1 | tail -1
(timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_crafted_payloads.py tests/test_gpu_random_shapes.py tests/test_gpu_sampled.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6) | tee gpurun_out/r6/run28_tests.log
