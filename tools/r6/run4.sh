#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
for m in "det 0" "det 8388608" "det 0" "det 8388608"; do timeout 300 python tools/r6/lab_c2.py $m 2>&1 | tail -1; done | tee gpurun_out/r6/run4_lab.log
bash tools/r6/timeline.sh det 8388608 run4_timeline_plainstores.txt | head -7
