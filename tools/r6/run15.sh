#!/bin/bash
# C4a slab timeline: the product library and variants with fewer persistent workgroups (SZ3HIP_LIB)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
for v in "" g1024 g512; do
  if [ -n "$v" ]; then export SZ3HIP_LIB=$R/sz3_amd/libsz3hip_$v.so; fi
  echo "== variant '$v'"
  LASTK=k_publish NTH=6 bash $R/tools/tl_case.sh --algo composed --dtype f64 --shape 128,1024,1024 --eb 1e-6 --field c4a 2>&1 | cut -c1-150
done 2>&1 | tee $R/gpurun_out/r6/run15.log
