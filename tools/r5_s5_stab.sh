#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for i in 1 2; do (timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2); done
timeout 900 python bench.py > gpurun_out/r05c_bench_default.json 2> gpurun_out/r05c_bench_default.err; head -c 330 gpurun_out/r05c_bench_default.json; echo
python - <<'PY'
import json
for l in open("gpurun_out/r05c_bench_default.json"):
    if l.startswith("{"):
        d=json.loads(l); print("host", d["host_e2e"]["compress_gbps"], d["host_e2e"]["decompress_gbps"], "C3", d["extra_configs"]["C3"]["ms_per_step"], d["extra_configs"]["C3"]["roofline"]["traffic"], "C4a", d["extra_configs"]["C4a_slab"]["ms_per_step"])
PY
