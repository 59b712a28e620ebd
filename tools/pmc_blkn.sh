#!/bin/bash
# PMC counters of the 1-D block path's kernels (2^27 f32 values, Lorenzo + regression): HBM traffic and instruction counts per launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/tools/blkn_bench.py 134217728 1e-3"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  rm -rf /tmp/pb; rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pb -o p -- $B > /dev/null 2>&1
  python - <<PY
import csv,glob,collections,re
f=glob.glob("/tmp/pb/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m=re.search(r"k_blkn_\w+|k_blk_coef_parse|k_blk_final", r["Kernel_Name"])
    if m: acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc: print(k, {c: "%.3g" % (sum(v)/len(v)) for c,v in acc[k].items()})
PY
done
