#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
(timeout 1500 python -m pytest tests/test_gpu_regression.py tests/test_gpu_regression_lowdim.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6) > gpurun_out/r6/run5_tests.log
cat gpurun_out/r6/run5_tests.log
timeout 600 python bench.py --algo composed --dtype f64 --shape 128,1024,1024 --eb 1e-6 --field c4a --steps 8 --warmup 3 --no-cpu-baseline --no-host-e2e --no-extra --no-cold --no-live-traffic 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C4a', o['ms_per_step'], o['ratio'], o['err_bound_ok'], o['decompress_device'], o['ms_per_step_median'])"
LASTK=k_publish BACK=2 bash tools/tl_case.sh --algo composed --dtype f64 --shape 128,1024,1024 --eb 1e-6 --field c4a 2>&1 | tail -24 | head -12
