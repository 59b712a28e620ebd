#!/bin/bash
# round 3, first GPU visit: the GPU suite, the default bench line, and the kernel timeline of one C2 step
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3a
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3a/pytest.log
tail -5 gpurun_out/r3a/pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; tail -c 3000 gpurun_out/r3a/bench.json
timeout 600 bash tools/prof_c2.sh > gpurun_out/r3a/timeline.txt 2>&1; cat gpurun_out/r3a/timeline.txt | tail -30
