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
tdt, ndt = (torch.float64, np.float64) if os.environ.get("LAB_DTYPE") == "f64" else (torch.float32, np.float32)
sigma = float(os.environ.get("LAB_SIGMA", "2e-3"))
dev = torch.device("cuda:0")
Z = shape[0]
f = torch.empty(shape, dtype=tdt, device=dev)
g = torch.Generator(device=dev).manual_seed(7)
inner = 1
for d in shape[1:]: inner *= d
rest = None  # the part of the field that does not depend on the slowest coordinate
for i, d in enumerate(shape[1:]):
    view = [1] * len(shape)
    view[i + 1] = d
    c = torch.sin(2 * np.pi * torch.arange(d, device=dev, dtype=tdt) / (46.0 + 17 * i)).view(view)
    rest = c if rest is None else rest + c
step = max(1, (1 << 28) // inner)
for z0 in range(0, Z, step):
    z1 = min(Z, z0 + step)
    zz = torch.arange(z0, z1, device=dev, dtype=tdt).view([-1] + [1] * (len(shape) - 1))
    f[z0:z1] = torch.sin(2 * np.pi * zz / 29.0) + rest + sigma * torch.randn((z1 - z0,) + shape[1:], device=dev, generator=g, dtype=tdt)
n = f.numel()
print("elements", n, "> 2^32" if n > 2 ** 32 else "", flush=True)
dc = sz3_amd.DeviceCompressor(n, ndt); cap = dc.payload_bound(n)
rel = os.environ.get("LAB_REL")
if rel:  # EB_REL the way the dispatcher converts it (utils/Statistic.hpp:12-56: bound = ratio x (max - min), the range found on the device: k_minmax)
    mn, mx = dc.minmax(f.data_ptr(), n, 0)
    ref_mn, ref_mx = float(f.min()), float(f.max())
    assert (mn, mx) == (ref_mn, ref_mx), (mn, mx, ref_mn, ref_mx)
    eb = float(rel) * (mx - mn)
    print("REL %s x range %.6g = abs %.6g" % (rel, mx - mn, eb), flush=True)
conf = sz3_amd.Config(*shape); conf.cmprAlgo = algo; conf.absErrorBound = eb
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
raw = n * f.element_size()
print(shape, os.environ.get("LAB_ALGO", "lorenzo"), "eb", eb, "ratio %.2f" % (raw / sz), "compress %.1f ms %.0f GB/s" % (tc * 1e3, raw / tc / 1e9),
      "decompress %.1f ms %.0f GB/s" % (td * 1e3, raw / td / 1e9), "max err %.6g" % worst, "OK" if worst <= eb else "BOUND VIOLATED", dc.stats())
