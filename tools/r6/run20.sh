#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
(timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -8) | tee gpurun_out/r6/run20_tests.log
timeout 1500 python bench.py --no-host-e2e --no-cpu-baseline > gpurun_out/r6/run20_bench.json 2> gpurun_out/r6/run20_bench.err
python - <<PY
import json
o=json.loads([l for l in open("gpurun_out/r6/run20_bench.json") if l.startswith("{")][-1])
print("value", o["value"], "ms", o["ms_per_step"], "ratio", o["ratio"], "dec", o["decompress_device"]["ms"], "det", o.get("ms_per_step_deterministic"), "cold", o.get("ms_per_step_cold"))
for k,v in o.get("extra_configs",{}).items(): print(k, v.get("ms_per_step"), v.get("ratio"), v.get("decompress_device"), v.get("error"))
PY
