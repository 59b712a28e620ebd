#!/bin/bash
# C4a slab, block stream: decompress time with the three 3-D decoders (debug flags 0 / 65536 / 8388608), same box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for f in ${FLAGS:-0 65536 0 65536}; do
  F=$f timeout 600 python - > /tmp/ab.log 2>&1 <<PY
import os, sys, runpy
sys.path.insert(0, "$R")
import sz3_amd
sz3_amd.lib().sz3hip_debug_flags(int(os.environ["F"]))
sys.argv = ["bench.py"] + "--algo composed --field c4a --dtype f64 --shape 128,1024,1024 --eb 1e-6 --steps 10 --warmup 2 --no-cpu-baseline --no-host-e2e --no-cold --no-live-traffic".split()
runpy.run_path("$R/bench.py", run_name="__main__")
PY
  grep '^{' /tmp/ab.log | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); print('flags $f: compress ms', j.get('ms_per_step'), 'decompress', j.get('decompress_device'))
" || tail -5 /tmp/ab.log
done 2>&1 | tee $O/blk3_ab.txt
