#!/usr/bin/env python3
"""device-resident decompression timing (per stage, HIP events)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd, time
from fields import field3d
S = int(os.environ.get("LAB_SIZE", "512")); eb = float(os.environ.get("LAB_EB", "1e-3"))
algo = {"interp": sz3_amd.ALGO_INTERP, "lorenzo": sz3_amd.ALGO_LORENZO_REG}[os.environ.get("LAB_ALGO", "lorenzo")]
a = field3d((S, S, S)); dev = torch.device("cuda:0"); d_in = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(S, S, S); conf.cmprAlgo = algo; conf.absErrorBound = eb
dc = sz3_amd.DeviceCompressor(a.size, np.float32); cap = dc.payload_bound(a.size)
pl = torch.empty(cap, dtype=torch.uint8, device=dev); out = torch.empty_like(d_in)
n = dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, 0)
for _ in range(3): dc.decompress(pl.data_ptr(), n, out.data_ptr(), 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
R = 10
for _ in range(R): dc.decompress(pl.data_ptr(), n, out.data_ptr(), 0)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / R
print("decompress %.3f ms -> %.1f GB/s; max err %.3g" % (t * 1e3, a.nbytes / t / 1e9, float((out - d_in).abs().max())))
dc.set_profiling(True); dc.decompress(pl.data_ptr(), n, out.data_ptr(), 0); torch.cuda.synchronize()
print({k: round(v, 4) for k, v in dc.stage_times().items() if k in ("huffman_decode", "reconstruct")})
