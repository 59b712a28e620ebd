#!/usr/bin/env python3
"""round trip of one very large array on one GPU (beyond 2^32 elements): the field is generated and the error bound is
checked slab by slab on the device, so that the working set stays at input + output + the compressor's own buffers"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, sz3_amd
shape = tuple(int(v) for v in os.environ.get("LAB_SHAPE", "1100,2000,2000").split(","))
eb = float(os.environ.get("LAB_EB", "1e-3"))
algo = {"interp": sz3_amd.ALGO_INTERP, "lorenzo": sz3_amd.ALGO_LORENZO_REG, "default": sz3_amd.ALGO_INTERP_LORENZO}[os.environ.get("LAB_ALGO", "lorenzo")]
dev = torch.device("cuda:0")
Z, Y, X = shape
f = torch.empty(shape, dtype=torch.float32, device=dev)
g = torch.Generator(device=dev).manual_seed(7)
yy = torch.arange(Y, device=dev, dtype=torch.float32)[None, :, None]
xx = torch.arange(X, device=dev, dtype=torch.float32)[None, None, :]
step = max(1, (1 << 28) // (Y * X))
for z0 in range(0, Z, step):
    z1 = min(Z, z0 + step)
    zz = torch.arange(z0, z1, device=dev, dtype=torch.float32)[:, None, None]
    f[z0:z1] = torch.sin(2 * np.pi * zz / 29.0) + torch.sin(2 * np.pi * yy / 46.0) + torch.sin(2 * np.pi * xx / 63.0) + \
               2e-3 * torch.randn((z1 - z0, Y, X), device=dev, generator=g)
n = f.numel()
print("elements", n, "> 2^32" if n > 2 ** 32 else "", flush=True)
conf = sz3_amd.Config(*shape); conf.cmprAlgo = algo; conf.absErrorBound = eb
dc = sz3_amd.DeviceCompressor(n, np.float32); cap = dc.payload_bound(n)
pl = torch.empty(cap, dtype=torch.uint8, device=dev); out = torch.empty_like(f)
torch.cuda.synchronize(); t0 = time.perf_counter()
sz = dc.compress(conf, f.data_ptr(), pl.data_ptr(), cap, 0)
torch.cuda.synchronize(); tc = time.perf_counter() - t0; t0 = time.perf_counter()
dc.decompress(pl.data_ptr(), sz, out.data_ptr(), 0)
torch.cuda.synchronize(); td = time.perf_counter() - t0
worst = 0.0
for z0 in range(0, Z, step):
    z1 = min(Z, z0 + step)
    worst = max(worst, float((out[z0:z1].double() - f[z0:z1].double()).abs().max()))
raw = n * 4
print(shape, os.environ.get("LAB_ALGO", "lorenzo"), "eb", eb, "ratio %.2f" % (raw / sz), "compress %.1f ms %.0f GB/s" % (tc * 1e3, raw / tc / 1e9),
      "decompress %.1f ms %.0f GB/s" % (td * 1e3, raw / td / 1e9), "max err %.6g" % worst, "OK" if worst <= eb else "BOUND VIOLATED", dc.stats())
