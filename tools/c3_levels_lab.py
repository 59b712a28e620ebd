#!/usr/bin/env python3
"""C3 (512^3 f32, default algorithm, abs 1e-4): the coarse levels through the level kernel (debug flag 4194304: a level per launch whatever
its block count) against the pass kernels (a launch per level and direction) — compress and decompress times, same payload"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
S = int(os.environ.get("LAB_SIZE", "512"))
a = field3d((S, S, S)); dev = torch.device("cuda:0"); d_in = torch.from_numpy(a).to(dev)
st = torch.cuda.current_stream().cuda_stream
sizes = {}
for flag in (0, 4194304, 0, 4194304):
    sz3_amd.lib().sz3hip_debug_flags(flag)
    conf = sz3_amd.Config(S, S, S); conf.absErrorBound = 1e-4
    dc = sz3_amd.DeviceCompressor(a.size, np.float32); cap = dc.payload_bound(a.size)
    pl = torch.empty(cap, dtype=torch.uint8, device=dev); out = torch.empty_like(d_in)
    for _ in range(4): size = dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): size = dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, st)
    torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / 20
    for _ in range(3): dc.decompress(pl.data_ptr(), size, out.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dc.decompress(pl.data_ptr(), size, out.data_ptr(), st)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 20
    h = hash(pl[:size].cpu().numpy().tobytes())
    print("flag %d: compress %.4f ms, decompress %.4f ms, size %d, max err %.3g, payload hash %x" % (flag, tc * 1e3, td * 1e3, size, float((out - d_in).abs().max()), h & 0xffffffff), flush=True)
sz3_amd.lib().sz3hip_debug_flags(0)
