#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
(echo "== 3-D small composed f64 c4a 80^3"; LASTK=k_publish NTH=6 bash $R/tools/tl_case.sh --algo composed --dtype f64 --shape 80,80,80 --eb 1e-6 --field c4a 2>&1 | cut -c1-150
echo "== C4a slab"; LASTK=k_publish NTH=6 bash $R/tools/tl_case.sh --algo composed --dtype f64 --shape 128,1024,1024 --eb 1e-6 --field c4a 2>&1 | cut -c1-150 | tail -4
echo "== 2-D 1024^2 composed f32"; LASTK=k_publish NTH=6 bash $R/tools/tl_case.sh --algo composed --shape 1024,1024 --eb 1e-3 2>&1 | cut -c1-150) | tee $R/gpurun_out/r6/run17.log
