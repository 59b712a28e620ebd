#!/usr/bin/env python3
"""device-resident compress/decompress timing for an arbitrary shape/dtype/algorithm (looking for performance cliffs)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
shape = tuple(int(v) for v in os.environ.get("LAB_SHAPE", "12,256,256,256").split(","))
dt = np.float64 if os.environ.get("LAB_DTYPE") == "f64" else np.float32
eb = float(os.environ.get("LAB_EB", "3e-3"))
algo = {"interp": sz3_amd.ALGO_INTERP, "lorenzo": sz3_amd.ALGO_LORENZO_REG, "default": sz3_amd.ALGO_INTERP_LORENZO}[os.environ.get("LAB_ALGO", "lorenzo")]
dev = torch.device("cuda:0")
if os.environ.get("LAB_DBG"): sz3_amd.lib().sz3hip_debug_flags(int(os.environ["LAB_DBG"]))  # e.g. 128: one-point-per-thread kernels
g = torch.Generator(device=dev).manual_seed(5)
grids = torch.meshgrid(*[torch.arange(s, device=dev, dtype=torch.float32) for s in shape], indexing="ij")
f = sum(torch.sin(2 * np.pi * gr / (29.0 + 17 * i)) for i, gr in enumerate(grids))
del grids
f = (f + float(os.environ.get("LAB_SIGMA", "2e-3")) * torch.randn(shape, device=dev, generator=g)).to(torch.float64 if dt == np.float64 else torch.float32)
if os.environ.get("LAB_NAN"):
    f[torch.rand(shape, device=dev, generator=g) < float(os.environ["LAB_NAN"])] = float("nan")
n = f.numel()
conf = sz3_amd.Config(*shape); conf.cmprAlgo = algo; conf.absErrorBound = eb
dc = sz3_amd.DeviceCompressor(n, dt); cap = dc.payload_bound(n)
pl = torch.empty(cap, dtype=torch.uint8, device=dev); out = torch.empty_like(f)
def comp():
    try:
        return dc.compress(conf, f.data_ptr(), pl.data_ptr(), cap, 0)
    except sz3_amd.SZ3HipError as e:
        return -1
for _ in range(2): sz = comp()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): sz = comp()
torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / 5
if sz < 0:
    print(shape, 'compress refused (outlier capacity): %.2f ms per attempt' % (tc * 1e3)); sys.exit(0)
dc.decompress(pl.data_ptr(), sz, out.data_ptr(), 0); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): dc.decompress(pl.data_ptr(), sz, out.data_ptr(), 0)
torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 5
dc.set_profiling(True); comp(); torch.cuda.synchronize()
raw = n * f.element_size()
print(shape, dt.__name__, os.environ.get("LAB_ALGO", "lorenzo"), "eb", eb, "ratio %.2f" % (raw / sz), "compress %.2f ms %.0f GB/s" % (tc * 1e3, raw / tc / 1e9),
      "decompress %.2f ms %.0f GB/s" % (td * 1e3, raw / td / 1e9), {k: round(v, 3) for k, v in dc.stage_times().items()}, dc.stats()["n_symbols"] if "n_symbols" in dc.stats() else "",
      "max err %.3g" % float(torch.nan_to_num((out.double() - f.double()).abs(), nan=0.0).max()), dc.stats()["n_value_outliers"])
