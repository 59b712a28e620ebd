#!/bin/bash
# kernel durations of the fused step under rocprofv3 for lab builds of the kernels (tools/build_lab.sh): VARIANTS="m1 m2" tools/merge_lab.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default ${VARIANTS}; do
  [ "$v" = default ] && unset SZ3HIP_LIB || export SZ3HIP_LIB=$R/sz3_amd/lab/libsz3hip_$v.so
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o r -- python $R/${LAB_SCRIPT:-tools/k1_lab.py} 0 > /dev/null 2>&1
  echo "== $v"
  python3 - /tmp/prof_$v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "").split("(")[0]
        if n.startswith("k_"): print("  %-46s calls %4s  avg %8.1f us" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
