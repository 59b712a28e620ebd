#!/bin/bash
# C4 slab with its specified predictor set (Lorenzo + regression): kernel stats + the bench line without the profiler
R=$GRAFT_REPO_ROOT; cd $R
bash tools/prof_blk.sh ${1:-default} | head -12
python bench.py --algo composed --field ${1:-default} --dtype f64 --shape 128,1024,1024 --eb 1e-6 --steps 10 --warmup 2 --no-cpu-baseline --no-host-e2e --no-cold 2>&1 | grep metric > gpurun_out/prof_blk/bench_noprof.json
python - <<PY
import json
d=json.load(open("gpurun_out/prof_blk/bench_noprof.json"))
print({k:d[k] for k in ("value","ms_per_step","ratio","decompress_device","stage_ms","block_selection")})
print(d["roofline"])
PY
