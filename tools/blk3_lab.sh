#!/bin/bash
# k_blk_decode_g across lab builds (what a 3-D group front's time is made of); C4a slab
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-h0 h1 h2 h3}; do
  rm -rf /tmp/pb; SZ3HIP_LIB=$R/sz3_amd/lab/libsz3hip_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o r -- python $R/bench.py --algo composed --field c4a --dtype f64 --shape 128,1024,1024 --eb 1e-6 --steps 2 --warmup 1 --no-cpu-baseline --no-host-e2e --no-cold > /tmp/pb.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/pb/*kernel_stats.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "decode_g" in r["Name"] or "coef_parse" in r["Name"] or "coef_scan" in r["Name"] or "k_blk_pre3" in r["Name"]]
print("$v", " | ".join("%s %s calls %.1f us" % (r["Name"].split("::")[-1][:20], r["Calls"], float(r["AverageNs"])/1000) for r in rows))
PY
done
