#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
timeout 1500 python bench.py > gpurun_out/r6/bench_default.json 2> gpurun_out/r6/bench_default.err
tail -3 gpurun_out/r6/bench_default.err
python - <<PY
import json
o=json.loads([l for l in open("gpurun_out/r6/bench_default.json") if l.startswith("{")][-1])
print("value", o["value"], "ms", o["ms_per_step"], "median", o.get("ms_per_step_median"), "ratio", o["ratio"], "dec", o["decompress_device"])
print("roofline", {k:o["roofline"][k] for k in ("achieved","frac","frac_kernel_compulsory","kernel_ms","traffic")})
print("path", o["roofline_path"]["frac"], "read frac", o["frac_read_peak_all_kernels"], "det", o.get("ms_per_step_deterministic"), "cold", o.get("ms_per_step_cold"))
print("cold", o.get("cold")); print("identical", o.get("identical_input")); print("two", o.get("two_contexts_in_flight"))
for k,v in o.get("extra_configs",{}).items(): print(k, v.get("ms_per_step"), v.get("ratio"), v.get("decompress_device"), v.get("error"))
print("host", o.get("host_e2e")); print("cpu", o.get("cpu_baseline"))
PY
