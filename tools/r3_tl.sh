#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/lab
timeout 600 bash tools/prof_c2.sh > gpurun_out/lab/timeline.txt 2>&1; tail -25 gpurun_out/lab/timeline.txt
