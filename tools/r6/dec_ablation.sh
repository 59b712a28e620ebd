#!/bin/bash
# (the variants first, here: bash tools/build_lab.sh abl1 "-DLAB_DEC_ABL=1" abl2 "-DLAB_DEC_ABL=2" abl3 "-DLAB_DEC_ABL=3")
# k_decode alone at a size where a SIMD holds half a wave (64 planes) and at full size, with pieces switched off (lab variants: wrong results)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r6
for sh in 64,512,512 512,512,512; do for v in "" abl1 abl2 abl3; do
  if [ -n "$v" ]; then export SZ3HIP_LIB=$R/sz3_amd/lab/libsz3hip_$v.so; else unset SZ3HIP_LIB; fi
  rm -rf /tmp/dl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dl -o r -- python $R/tools/r6/dec_lab.py $sh > /tmp/dl.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/dl/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_decode<4, true" in r["Name"]: print("$sh variant '$v': k_decode %.1f us (calls %s)" % (float(r["AverageNs"])/1000, r["Calls"]))
PY
done; done 2>&1 | tee $R/gpurun_out/r6/run26.log
