#!/bin/bash
# the block-composed ENCODER at the C4a slab (128 x 1024^2 f64, 1e-6, Lorenzo + regression) across lab builds of sz3hip_regress.hip:
# the bench line's ms_per_step without the profiler, then the kernels' averages under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/lab
B="python $R/bench.py --algo composed --field c4a --dtype f64 --shape 128,1024,1024 --eb 1e-6 --steps 10 --warmup 3 --no-cpu-baseline --no-host-e2e --no-cold --no-extra"
for v in ${VARIANTS:-default}; do
  L=$R/sz3_amd/lab/libsz3hip_$v.so; [ "$v" = default ] && L=$R/sz3_amd/libsz3hip.so
  SZ3HIP_LIB=$L $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', 'ms_per_step', d['ms_per_step'], 'ratio', d.get('ratio'), 'decompress', (d.get('decompress_device') or {}).get('ms'))"
  rm -rf /tmp/pb; SZ3HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o r -- $B > /tmp/pb.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/pb/*kernel_stats.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if float(r["AverageNs"]) > 20000 and "at::native" not in r["Name"]]
for r in rows[:int("${TOPN:-12}")]: print("   %-70s calls %4s avg %8.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:70], r["Calls"], float(r["AverageNs"])/1000))
PY
done 2>&1 | tee $R/gpurun_out/lab/c4a_enc_lab.txt
