#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
(for f in 0 8192; do echo "== flags $f"; bash $R/tools/r6/tl_py.sh $R/tools/r6/c3_lab.py $f | grep -v "k_interp_pass\|k_profile\|k_gather\|rocclr\|k_code_cost\|k_interp_trials\|k_interp_anchors"; done) 2>&1 | tee $R/gpurun_out/r6/run23.log
