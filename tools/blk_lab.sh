#!/bin/bash
# kernel times of the block-composed path (C4 slab) across lab builds of sz3hip_regress.hip
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-b0 b1 b2 b3 b4}; do
  rm -rf /tmp/pb; SZ3HIP_LIB=$R/sz3_amd/lab/libsz3hip_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o r -- python $R/bench.py --algo composed --dtype f64 --shape 128,1024,1024 --eb 1e-6 --steps 5 --warmup 2 --no-cpu-baseline --no-host-e2e --no-cold > /tmp/pb.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/pb/*kernel_stats.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "k_blk_fit" in r["Name"] or "k_blk_lorenzo" in r["Name"]]
print("$v", " | ".join("%s %s calls %.0f us" % (r["Name"].split("::")[-1][:34], r["Calls"], float(r["AverageNs"])/1000) for r in rows))
PY
done
