#!/bin/bash
# the multi-symbol decoder table (lab build, flag 2) where k_decode is bound by the lane's chain: small arrays
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r6
export SZ3HIP_LIB=$R/sz3_amd/libsz3hip_lab.so
for sh in 4,512,512 8,512,512; do for fl in 0 2; do
  rm -rf /tmp/dl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dl -o r -- python $R/tools/r6/dec_lab.py $sh $fl > /tmp/dl.log 2>&1
  grep "^decompress" /tmp/dl.log
  python - <<PY
import csv,glob
f=glob.glob("/tmp/dl/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_decode<4, true" in r["Name"]: print("   %s: %.1f us (calls %s)" % (r["Name"][:40], float(r["AverageNs"])/1000, r["Calls"]))
PY
done; done 2>&1 | tee $R/gpurun_out/r6/run27.log
