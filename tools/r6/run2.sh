#!/bin/bash
# round 6: the sampled book — tests, then the C2 step in every mode
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
(timeout 1200 python -m pytest tests/test_gpu_sampled.py tests/test_gpu_packb.py -x -q 2>&1 | tail -25) > gpurun_out/r6/run2_tests.log
cat gpurun_out/r6/run2_tests.log
for m in "spec 0" "det 0" "nospec 0" "spec 65536" "det 65536"; do timeout 300 python tools/r6/lab_c2.py $m 2>&1 | tail -1; done | tee gpurun_out/r6/run2_lab.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o r -- python $R/tools/r6/lab_c2.py det 0 10 > /tmp/p1.log 2>&1
python - <<PY | tee $R/gpurun_out/r6/run2_kernels.log
import csv,glob
f=glob.glob("/tmp/p1/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1000))
PY
