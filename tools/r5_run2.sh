#!/bin/bash
# round 5: the published state (k_publish + host poll) — stage tests, parity subset, K1 lab wall, bench
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5
(timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_q16.py -x -q 2>&1 | tail -8) > gpurun_out/r5/stages.log
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_interp.py tests/test_gpu_multislab.py -x -q 2>&1 | tail -8) > gpurun_out/r5/parity.log
(timeout 200 python tools/k1_lab.py 0 0 2>&1 | grep -v amdgpu.ids) > gpurun_out/r5/k1_wall.log
(timeout 600 python bench.py --no-cpu-baseline --no-host-e2e 2>&1 | tail -1) > gpurun_out/r5/bench_quick.json
for f in stages parity k1_wall; do echo "== $f"; cat gpurun_out/r5/$f.log; done
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_quick.json').read())
print({k:d[k] for k in ('value','ms_per_step')}, d.get('cold'), d.get('roofline'))
for k,v in (d.get('extra_configs') or {}).items(): print(k, v.get('ms_per_step'), v.get('decompress_device'))
print(d.get('decompress_device'))
PY
