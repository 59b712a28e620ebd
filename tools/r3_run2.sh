#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${OUT:-r3c}; mkdir -p $O
timeout 300 python tools/k1_lab.py ${LABFLAGS:-0} 2>&1 | grep -v amdgpu.ids | tee $O/lab.txt
if [ -z "$NOTEST" ]; then timeout 1800 python -m pytest tests -m gpu -x -q ${PYK:+-k "$PYK"} > $O/pytest.log 2>&1; grep -E "passed|failed|rror|assert" $O/pytest.log | tail -12; fi
timeout 600 bash tools/prof_c2.sh > $O/timeline.txt 2>&1; tail -14 $O/timeline.txt
