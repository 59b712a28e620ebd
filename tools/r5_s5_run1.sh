#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
LASTK=k_publish bash tools/tl_case.sh --algo interp --eb 1e-4 > gpurun_out/tl_c3.txt 2>&1
timeout 300 python tools/host_lab.py > gpurun_out/host_lab.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-live-traffic > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
cat gpurun_out/tl_c3.txt; tail -30 gpurun_out/host_lab.txt; head -c 300 gpurun_out/bench_quick.json
