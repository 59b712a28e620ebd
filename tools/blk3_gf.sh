#!/bin/bash
# the 3-D block decoder: tests of the block path, then the C4a slab's decompress kernel by kernel (gpurun_out/blk3_gf.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
if [ -z "$NOTEST" ]; then timeout 900 python -m pytest tests/test_gpu_regression.py tests/test_gpu_crafted_payloads.py -x -q -m gpu > $O/blk3_gf_tests.txt 2>&1; tail -3 $O/blk3_gf_tests.txt; fi
cd /tmp
for lib in ${VARIANTS:-product}; do
  rm -rf /tmp/pb
  if [ "$lib" = product ]; then unset SZ3HIP_LIB; else export SZ3HIP_LIB=$R/sz3_amd/lab/libsz3hip_$lib.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o r -- python $R/bench.py --algo composed --field c4a --dtype f64 --shape 128,1024,1024 --eb 1e-6 --steps 3 --warmup 1 --no-cpu-baseline --no-host-e2e --no-cold --no-live-traffic > /tmp/pb.log 2>&1
  echo "== $lib"
  tail -1 /tmp/pb.log | python3 -c "
import sys, json
try:
    j = json.loads(sys.stdin.read()); print('compress ms', j.get('ms_per_step'), 'decompress', j.get('decompress_device'))
except Exception as e: print('no line', e)
"
  python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pb/*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows = [r for r in rows if any(k in r["Name"] for k in ("k_blk_decode", "k_blk_wave3", "k_blk_local3", "k_blk_pre3", "k_blk_final", "k_decode", "expand", "coef_parse", "k_blk_patch"))]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows: print("%-60s calls %5s avg %8.1f us total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
t = glob.glob("/tmp/pb/*kernel_trace.csv")
if t:
    rows = [r for r in csv.DictReader(open(t[0])) if "k_blk_decode_g" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    calls = len([r for r in csv.DictReader(open(t[0])) if "k_blk_local3" in r["Kernel_Name"] or "k_blk_pre3" in r["Kernel_Name"]]) or 1
    if rows:
        n = len(rows) // calls
        last = rows[-n:]
        print("fronts per call %d, from the first front's start to the last one's end (last call) %.1f us" % (n, (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3))
        for i in range(0, n, max(1, n // 6)):
            r = last[i]; nxt = last[min(i + 1, n - 1)]
            print("  front %3d grid %6s dur %6.1f us, to the next one's start %6.1f us" % (i, r.get("Grid_Size_X", r.get("Grid_Size", "?")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(nxt["Start_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
done 2>&1 | tee $O/blk3_gf.txt
