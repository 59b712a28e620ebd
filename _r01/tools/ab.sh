#!/bin/bash
# A/B of two builds on the same box: ab/lib_base.so (SZ3HIP_LIB) against the in-tree library, alternating
ARGS="$@"
for i in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export SZ3HIP_LIB=$PWD/ab/lib_base.so; else unset SZ3HIP_LIB; fi
    python bench.py --no-cpu-baseline --no-host-e2e $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['stage_ms'])"
  done
done
