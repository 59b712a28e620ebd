#!/bin/bash
# round 5, last state: suite, smoke, the bench line
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3) > gpurun_out/r05d_gputests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) > gpurun_out/r05d_smoke.log
timeout 1500 python bench.py > gpurun_out/r05d_bench_default.json 2> gpurun_out/r05d_bench_default.err
cat gpurun_out/r05d_gputests.log gpurun_out/r05d_smoke.log; head -c 330 gpurun_out/r05d_bench_default.json; echo
python - <<'PY'
import json
for l in open("gpurun_out/r05d_bench_default.json"):
    if l.startswith("{"):
        d=json.loads(l); e=d["extra_configs"]
        print("C2", d["ms_per_step"], d["ratio"], d["decompress_device"], "host", d["host_e2e"]["compress_gbps"], d["host_e2e"]["decompress_gbps"], "cold", d["cold"])
        print("C3", e["C3"]["ms_per_step"], e["C3"]["ratio"], e["C3"]["decompress_device"], e["C3"]["roofline"]["traffic"])
        print("C4a", e["C4a_slab"]["ms_per_step"], e["C4a_slab"]["decompress_device"]); print("C1", e["C1"]["ms_per_step"], e["C1"]["decompress_device"], e["C1_default_algorithm"]["ms_per_step"])
PY
