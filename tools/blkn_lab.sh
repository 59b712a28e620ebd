#!/bin/bash
# k_blkn_decode2g across lab builds (what the group kernel's time is made of)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-g0 g1 g2 g3 g4}; do
  rm -rf /tmp/pb; SZ3HIP_LIB=$R/sz3_amd/lab/libsz3hip_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o r -- python $R/tools/blkn_bench.py 8192,8192 0.15 > /tmp/pb.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/pb/*kernel_stats.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "decode2" in r["Name"]]
print("$v", " | ".join("%s %s calls %.1f us" % (r["Name"].split("::")[-1][:24], r["Calls"], float(r["AverageNs"])/1000) for r in rows))
PY
done
