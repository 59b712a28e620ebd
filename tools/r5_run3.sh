#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5
(timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -k "multi_symbol" 2>&1 | tail -15) > gpurun_out/r5/ms.log
(timeout 1200 python -m pytest tests/test_gpu_stages.py tests/test_gpu_parity.py tests/test_gpu_random_shapes.py tests/test_gpu_sweeps.py -x -q 2>&1 | tail -8) > gpurun_out/r5/ms_all.log
VARIANTS="pm" bash tools/k1_ab.sh > /dev/null 2>&1
(timeout 300 python tools/dec_lab.py 0 2 0 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/r5/dec_lab.log
for f in ms ms_all dec_lab; do echo "== $f"; cat gpurun_out/r5/$f.log; done; echo == pack map; cat gpurun_out/lab/k1_ab.txt
