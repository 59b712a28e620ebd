#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
NTH=9 LASTK=k_publish bash tools/tl_case.sh --algo interp --eb 1e-4 > gpurun_out/tl_c3_timed.txt 2>&1
NTH=12 LASTK=k_publish bash tools/tl_case.sh > gpurun_out/tl_c2_timed.txt 2>&1
cat gpurun_out/tl_c3_timed.txt gpurun_out/tl_c2_timed.txt
