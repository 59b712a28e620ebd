#!/usr/bin/env python3
"""random sweep: GPU interpolation codes / reconstruction vs the oracle (bit-exact) over shapes, directions, anchors, alpha/beta,
quantisation radii, linear/cubic, f32/f64 — a development check beyond tests/test_gpu_interp.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd, math
from oracle_binding import ALGO_INTERP, make_config, oracle_interp_codes
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
if os.environ.get("DBG_FLAGS"): sz3_amd.lib().sz3hip_debug_flags(int(os.environ["DBG_FLAGS"]))  # e.g. 4194304: level kernels whatever the size
pool = [5, 8, 9, 16, 17, 24, 31, 32, 33, 40, 48, 63, 64, 65, 72, 100]
bad = 0
for k in range(int(os.environ.get("N", "40"))):
    nd = int(rng.integers(1, 5))
    shape = tuple(int(rng.choice(pool)) for _ in range(nd))
    while np.prod(shape) > 1_500_000: shape = tuple(max(5, s // 2) for s in shape)
    if os.environ.get("LARGE"):  # 3-D arrays of the level kernels' natural size (>= 256 blocks of 32^3 at the finest level)
        nd = 3
        shape = tuple(int(rng.integers(193, 262)) for _ in range(3))
    dt = np.float32 if rng.random() < 0.7 else np.float64
    grids = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
    a = (sum(np.sin(2 * np.pi * g / (9.0 + 4 * i)) for i, g in enumerate(grids)) + 0.01 * rng.standard_normal(shape)).astype(dt)
    if k % 5 == 0: a.reshape(-1)[a.size // 2] = np.nan
    kw = dict(interpAlgo=int(rng.integers(0, 2)), interpDirection=int(rng.integers(0, math.factorial(nd))),
              interpAnchorStride=int(rng.choice([0, 4, 8, 16, 32, 64, 128])), interpAlpha=float(rng.choice([-1.0, 1.0, 1.25, 1.5, 2.0])),
              interpBeta=float(rng.choice([1.0, 2.0, 2.5, 3.0, 4.0])), quantbinCnt=int(rng.choice([256, 1024, 65536])))
    eb = float(10.0 ** rng.integers(-4, -1))
    dev = torch.device("cuda:0"); t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype); cap = dc.payload_bound(a.size); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_INTERP; conf.absErrorBound = eb
    for kk, v in kw.items(): setattr(conf, kk, v)
    try:
        size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
    except sz3_amd.SZ3HipError as e:
        print(k, shape, dt.__name__, kw, "compress refused:", str(e)[:60]); continue
    codes = dc.debug_codes(a.size); out = torch.empty_like(t); dc.decompress(pl.data_ptr(), size, out.data_ptr(), 0); torch.cuda.synchronize()
    okw = dict(abs_eb=eb, interp_algo=kw["interpAlgo"], interpDirection=kw["interpDirection"], interpAnchorStride=kw["interpAnchorStride"],
               interpAlpha=kw["interpAlpha"], interpBeta=kw["interpBeta"], quantbinCnt=kw["quantbinCnt"])
    oc = make_config(shape, algo=ALGO_INTERP, **okw)
    ocodes, order, recon, nun = oracle_interp_codes(a, oc)
    nat = np.zeros(a.size, dtype=np.int64); nat[order.astype(np.int64)] = ocodes
    same = np.array_equal(codes.astype(np.int64), nat) and np.array_equal(out.cpu().numpy(), recon.reshape(shape), equal_nan=True)
    if not same: bad += 1
    print(k, shape, dt.__name__, kw, "eb", eb, "OK" if same else "MISMATCH")
print("mismatches:", bad)
