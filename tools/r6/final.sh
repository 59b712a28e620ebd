#!/bin/bash
# round 6, closing run: the whole GPU suite, smoke(), the driver's bench command, its rocprofv3 kernel stats + PMC passes, a step's timeline
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
(timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8) > gpurun_out/r6/final_gputests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) > gpurun_out/r6/final_smoke.log
cat gpurun_out/r6/final_gputests.log gpurun_out/r6/final_smoke.log
timeout 1500 python bench.py > gpurun_out/r6/final_bench.json 2> gpurun_out/r6/final_bench.err
bash tools/pmc.sh > gpurun_out/r6/final_pmc.log 2>&1
cp gpurun_out/pmc_summary.txt gpurun_out/r6/final_pmc_summary.txt 2>/dev/null
cp gpurun_out/kernel_stats.csv gpurun_out/r6/final_kernel_stats.csv 2>/dev/null
cp gpurun_out/kernel_stats_c3.csv gpurun_out/r6/final_kernel_stats_c3.csv 2>/dev/null
bash tools/r6/timeline.sh spec 0 final_timeline_c2.txt > /dev/null 2>&1
LASTK=k_publish NTH=8 bash tools/tl_case.sh --algo interp --eb 1e-4 > gpurun_out/r6/final_timeline_c3.txt 2>&1
python - <<PY
import json
o=json.loads([l for l in open("gpurun_out/r6/final_bench.json") if l.startswith("{")][-1])
print("value", o["value"], "ms", o["ms_per_step"], "median", o["ms_per_step_median"]["median_ms_host_clock"], "ratio", o["ratio"], "dec", o["decompress_device"]["ms"])
print("roofline", {k:o["roofline"][k] for k in ("achieved","frac","frac_kernel_compulsory","kernel_ms","traffic")}, "path", o["roofline_path"]["frac"], o["roofline_path"]["traffic_over_algorithmic"], "read frac", o["frac_read_peak_all_kernels"])
print("det", o.get("ms_per_step_deterministic"), "cold", o.get("ms_per_step_cold"), "two", o["two_contexts_in_flight"].get("ms_per_call"))
for k,v in o.get("extra_configs",{}).items(): print(k, v.get("ms_per_step"), v.get("ratio"), v.get("decompress_device"), v.get("error"))
print("host", {k:o["host_e2e"][k] for k in ("compress_gbps","decompress_gbps")}, "cpu", o["cpu_baseline"]["value"], o["cpu_baseline"]["all_cores"]["value"])
PY
