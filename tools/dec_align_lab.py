#!/usr/bin/env python3
"""Does the decoder's time depend on where the output lies relative to the library's scratch? C2 decompress with the output
buffer shifted by a range of byte offsets (the z pass reads int16 planes 512 KB apart and writes float planes 1 MB apart)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
S = 512
a = field3d((S, S, S)); dev = torch.device("cuda:0")
d_in = torch.from_numpy(a).to(dev)
stream = torch.cuda.current_stream().cuda_stream
conf = sz3_amd.Config(S, S, S); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.regression = 0; conf.absErrorBound = 1e-3
dc = sz3_amd.DeviceCompressor(a.size, np.float32)
cap = dc.payload_bound(a.size); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
size = dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, stream)
big = torch.empty(a.nbytes + (8 << 20), dtype=torch.uint8, device=dev)
base = (big.data_ptr() + (4 << 20) - 1) & ~((4 << 20) - 1)   # 4 MB aligned
info = dc.debug_scratch_ptr() if hasattr(dc, "debug_scratch_ptr") else 0
print("output base %x  scratch %x" % (base, info))
offs = [0] + [1 << k for k in range(8, 22)] + [3 << 10, 5 << 12, 3 << 16, 5 << 17, 7 << 18]
for off in offs:
    p = base + off
    for _ in range(2): dc.decompress(pl.data_ptr(), size, p, stream)
    dc.set_profiling(True); rec = []
    for _ in range(6):
        dc.decompress(pl.data_ptr(), size, p, stream); torch.cuda.synchronize(); rec.append(dc.stage_times().get("huffman_decode", 0))
    dc.set_profiling(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dc.decompress(pl.data_ptr(), size, p, stream)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
    print("offset %8d (%7.1f KB): decode + scans %.1f us  wall %.1f us" % (off, off / 1024, 1e3 * np.median(rec), 1e3 * wall), flush=True)
