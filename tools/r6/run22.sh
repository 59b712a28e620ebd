#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
bash tools/r6/timeline.sh spec 0 run22_timeline_c2.txt > /dev/null 2>&1; tail -8 gpurun_out/r6/run22_timeline_c2.txt | cut -c1-130
python tools/r6/c1_tl.py 2>&1 | grep -v "$F"
python tools/r6/lab_c2.py det 0 2>&1 | grep -v "$F" | tail -2
(timeout 3000 python -m pytest tests/test_gpu_sampled.py tests/test_gpu_packb.py tests/test_gpu_stages.py tests/test_gpu_parity.py -x -q 2>&1 | grep -v "$F" | tail -8) | tee gpurun_out/r6/run22_tests.log
