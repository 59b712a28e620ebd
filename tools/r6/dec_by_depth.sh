#!/bin/bash
# does the decoder run faster per element when its int16 intermediate fits the memory-side cache? decompress time by slab depth
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6
for sh in 64,512,512 128,512,512 256,512,512 384,512,512 512,512,512; do
  python bench.py --shape $sh --steps 10 --warmup 3 --no-cpu-baseline --no-host-e2e --no-extra --no-cold --no-live-traffic 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); n=1
        for v in '$sh'.split(','): n*=int(v)
        print('$sh: compress %.4f ms, decompress %.4f ms = %.3f ns/Melem... per 2^27 elements: %.4f ms' % (o['ms_per_step'], o['decompress_device']['ms'], 0, o['decompress_device']['ms']*134217728/n))
"
done 2>&1 | tee gpurun_out/r6/run24.log
