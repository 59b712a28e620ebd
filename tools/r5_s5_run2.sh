#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pieces.py -x -q 2>&1 | tail -15 > gpurun_out/pieces_tests.log
timeout 300 python tools/host_lab.py > gpurun_out/host_lab2.txt 2>&1
SZ3HIP_PIECES=4 timeout 300 python tools/host_lab.py > gpurun_out/host_lab2_p4.txt 2>&1
SZ3HIP_PIECES=0 timeout 300 python tools/host_lab.py > gpurun_out/host_lab2_p0.txt 2>&1
cat gpurun_out/pieces_tests.log; grep iter gpurun_out/host_lab2.txt; echo p4; grep iter gpurun_out/host_lab2_p4.txt; echo p0; grep iter gpurun_out/host_lab2_p0.txt
