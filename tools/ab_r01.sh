#!/bin/bash
# A/B of the round-1 build (unpacked by hand into _r01/, not tracked) against the current one: bench lines of a few configs
cd /tmp && export TMPDIR=/tmp
run() { for d in $GRAFT_REPO_ROOT/_r01 $GRAFT_REPO_ROOT; do echo -n "$(basename $d) $* : "; python $d/bench.py "$@" --steps 8 --warmup 3 --no-cpu-baseline --no-host-e2e 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ratio'), d.get('decompress_device',{}).get('ms'), d.get('stage_ms'))"; done; }
run --algo interp --dtype f64 --shape 128,1024,1024 --eb 1e-6
run --algo lorenzo --eb 1e-6
run --algo interp --eb 1e-3
run --algo interp-notune --eb 1e-5
run --algo lorenzo --eb 1e-2
run --algo lorenzo --shape 96,500,500 --eb 1e-3
