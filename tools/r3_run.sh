#!/bin/bash
# GPU visit: the GPU suite (optionally a subset: $1 = pytest -k expression), the default bench line, the kernel timeline of one C2 step
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUT:-r3b}
mkdir -p $O
cd $R
if [ -n "$1" ]; then K=(-k "$1"); else K=(); fi
timeout 1800 python -m pytest tests -m gpu -x -q "${K[@]}" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|error|Error|assert" $O/pytest.log | tail -15
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 4500 $O/bench.json; tail -3 $O/bench.err
timeout 600 bash tools/prof_c2.sh > $O/timeline.txt 2>&1; tail -30 $O/timeline.txt
cp $R/gpurun_out/prof_c2/kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
