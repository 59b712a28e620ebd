#!/bin/bash
# A/B of device decompression (ab/lib_base.so against the in-tree library), alternating
ARGS="$@"
for i in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export SZ3HIP_LIB=$PWD/ab/lib_base.so; else unset SZ3HIP_LIB; fi
    python bench.py --no-cpu-baseline --no-host-e2e $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['decompress_device'], d['err_bound_ok'])"
  done
done
