#!/usr/bin/env python3
"""random sweep of the Lorenzo path: GPU codes vs the numpy model of K1 (tests/szh_ref.py), payload decode vs the model,
error bound; shapes, dtypes, quantisation radii (small radii put code 0 inside the histogram windows), bounds, NaNs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd, szh_ref
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
if os.environ.get("DBG_FLAGS"): sz3_amd.lib().sz3hip_debug_flags(int(os.environ["DBG_FLAGS"]))  # e.g. 4194304: level kernels whatever the size
pool = [1, 3, 8, 9, 16, 17, 31, 32, 33, 40, 64, 65, 100, 128, 129, 256, 260, 300, 500, 504, 512]
bad = 0
for k in range(int(os.environ.get("N", "40"))):
    nd = int(rng.integers(1, 5))
    shape = tuple(int(rng.choice(pool)) for _ in range(nd))
    while np.prod(shape) > 3_000_000: shape = tuple(max(1, s // 2) for s in shape)
    if np.prod(shape) < 16: shape = shape + (64,) if nd < 4 else (2, 3, 4, 64)
    dt = np.float32 if rng.random() < 0.7 else np.float64
    grids = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
    sig = float(rng.choice([1e-4, 2e-3, 5e-2]))
    a = (sum(np.sin(2 * np.pi * g / (11.0 + 5 * i)) for i, g in enumerate(grids)) + sig * rng.standard_normal(shape)).astype(dt)
    if k % 4 == 0: a.reshape(-1)[rng.integers(0, a.size, size=max(1, a.size // 500))] = np.nan
    if k % 7 == 3: a.reshape(-1)[rng.integers(0, a.size, size=max(1, a.size // 60))] = np.nan  # (enough to fill the per-wave queues of unpredictable values several times)
    qb = int(rng.choice([64, 256, 1024, 4096, 65536])); eb = float(10.0 ** rng.integers(-4, -1))
    dev = torch.device("cuda:0"); t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype); cap = dc.payload_bound(a.size, worst_case=True); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.regression = 0; conf.absErrorBound = eb; conf.quantbinCnt = qb
    try:
        size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
    except sz3_amd.SZ3HipError as e:
        print(k, shape, dt.__name__, "qb", qb, "eb", eb, "refused:", str(e)[:50]); continue
    st = dc.stats(); codes = dc.debug_codes(a.size)
    out = torch.empty_like(t); dc.decompress(pl.data_ptr(), size, out.data_ptr(), 0); torch.cuda.synchronize(); dec = out.cpu().numpy()
    a4 = a.reshape((1,) * (4 - a.ndim) + a.shape) if False else a
    q, d, exp_codes, badm, dout = szh_ref.dualquant(a, eb, radius=qb // 2, narrow=bool(st["narrow_codes"]))
    ok = np.array_equal(codes, exp_codes.reshape(-1)) and st["n_value_outliers"] == int(badm.sum()) and st["n_delta_outliers"] == int(dout.sum())
    h, o, sec = szh_ref.parse(pl[:size].cpu().numpy().tobytes())
    model = szh_ref.reconstruct(h, sec, exp_codes.reshape(-1)).reshape(a.shape)
    ok = ok and np.array_equal(dec, model, equal_nan=True)
    fin = np.isfinite(a)
    ok = ok and (not fin.any() or np.max(np.abs(dec[fin].astype(np.float64) - a[fin].astype(np.float64))) <= eb) and np.array_equal(np.isnan(dec), np.isnan(a))
    if not ok:
        bad += 1
        print("   codes", np.array_equal(codes, exp_codes.reshape(-1)), "vout", st["n_value_outliers"], int(badm.sum()), "dout", st["n_delta_outliers"], int(dout.sum()),
              "model", np.array_equal(dec, model, equal_nan=True), "max err", float(np.max(np.abs(dec[fin].astype(np.float64) - a[fin].astype(np.float64)))) if fin.any() else None,
              "nan", np.array_equal(np.isnan(dec), np.isnan(a)))
    print(k, shape, dt.__name__, "qb", qb, "eb", eb, "narrow", st["narrow_codes"], "vout", st["n_value_outliers"], "dout", st["n_delta_outliers"], "OK" if ok else "MISMATCH")
print("mismatches:", bad)
