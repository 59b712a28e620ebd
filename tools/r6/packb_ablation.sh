#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
export SZ3HIP_LIB=$R/sz3_amd/libsz3hip_lab.so
for f in 0 16 512 528; do echo "== lab flags $f"; bash tools/r6/timeline.sh det $f run6_tl_$f.txt | grep "k_pack_b\|march3q" | head -2; done
