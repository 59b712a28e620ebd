#!/bin/bash
# decoder variants on the GPU box: VARIANTS="dB dC" bash tools/dec_lab.sh   (builds from tools/build_lab.sh; "default" = the shipped library)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/lab
{
timeout 300 python tools/dec_lab.py 2>&1 | grep -v amdgpu.ids
for v in ${VARIANTS:-dB dC dD}; do SZ3HIP_LIB=$R/sz3_amd/lab/libsz3hip_$v.so timeout 300 python tools/dec_lab.py 2>&1 | grep -v amdgpu.ids; done
} | tee gpurun_out/lab/dec_variants.txt
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest tests -m gpu -x -q $TESTS 2>&1 | tail -15 | tee gpurun_out/lab/dec_tests.txt; fi
