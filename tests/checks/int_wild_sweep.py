#!/usr/bin/env python3
"""randomised sweep over INTEGER arrays the other sweeps do not draw (tests/checks/host_sweep.py: 1000 x a smooth field): the full range of the type,
constants, steps, alternating extremes, magnitudes beyond 2^53 (int64: kept lossless, the f64 pipeline is not exact there), all eight integer
types of the HDF5 face (int8 .. uint64 ride the int32 / int64 forms), bounds from 0.4 (-> 0: lossless) to 1e6, every algorithm.
Checked: dtype and shape kept, |x - x^| <= floor(eb) as integers. SEED, N from the environment; exit code = failures."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, sz3_amd
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
TYPES = [np.int32, np.int64, np.int32, np.int64, np.int8, np.uint8, np.int16, np.uint16, np.uint32, np.uint64]
bad = 0
for k in range(int(os.environ.get("N", "60"))):
    dt = TYPES[int(rng.integers(0, len(TYPES)))]
    nd = int(rng.integers(1, 5))
    shape = tuple(int(rng.integers(1, [0, 300000, 600, 70, 24][nd])) for _ in range(nd))
    n = int(np.prod(shape))
    info = np.iinfo(dt)
    kind = str(rng.choice(["full", "const", "steps", "extremes", "smooth", "big", "small"]))
    if kind == "full": a = rng.integers(info.min, info.max, size=shape, dtype=dt, endpoint=True)
    elif kind == "const": a = np.full(shape, int(rng.choice([info.min, info.max, 0, 7])), dtype=dt)
    elif kind == "steps":
        lv = np.zeros(n, dtype=np.float64)
        for c in np.sort(rng.integers(0, n, size=int(rng.integers(1, 9)))): lv[c:] += float(rng.choice([-90.0, 33.0, 100.0]))
        a = np.clip(lv, float(info.min), float(info.max)).astype(dt).reshape(shape)
    elif kind == "extremes": a = np.where(rng.random(shape) < 0.5, info.min, info.max).astype(dt)
    elif kind == "smooth":
        g = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
        amp = min(float(info.max) * 0.9, float(rng.choice([50.0, 3000.0, 1e6, 1e9])))
        a = (amp * 0.5 * (1 + np.prod([np.sin(2 * np.pi * x / (17.0 + 4 * i)) for i, x in enumerate(g)], axis=0))).astype(dt)
    elif kind == "big":  # beyond 2^53 where the type reaches that far
        a = (rng.integers(-1000, 1000, size=shape).astype(np.float64) + float(info.max) * 0.75).clip(float(info.min), float(info.max) * 0.999).astype(dt)
    else: a = rng.integers(max(info.min, -3), min(info.max, 3), size=shape, dtype=dt, endpoint=True)
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = int(rng.choice([sz3_amd.ALGO_LORENZO_REG, sz3_amd.ALGO_INTERP_LORENZO, sz3_amd.ALGO_INTERP, sz3_amd.ALGO_NOPRED]))
    eb = float(rng.choice([0.4, 1.0, 1.5, 3.0, 10.0, 1000.0, 1e6]))
    conf.absErrorBound = eb
    tag = "case %d %s %s %s algo %d eb %g" % (k, kind, shape, np.dtype(dt).name, conf.cmprAlgo, eb)
    try:
        blob, ratio = sz3_amd.compress(a, conf)
        dec, c2 = sz3_amd.decompress(blob, a.dtype, a.shape)
    except Exception as e:
        bad += 1; print("EXC", tag, str(e)[:120], flush=True); continue
    ok = dec.dtype == a.dtype and dec.shape == a.shape
    err = None
    if ok and n:  # exact integer differences (Python ints where a difference of 64-bit values could overflow)
        err = int(np.abs(a.astype(object) - dec.astype(object)).max()) if np.dtype(dt).itemsize == 8 else int(np.abs(a.astype(np.int64) - dec.astype(np.int64)).max())
        ok = err <= int(np.floor(eb))
    if not ok:
        bad += 1; print("FAIL", tag, "->", c2.cmprAlgo, "err", err, flush=True)
print("cases %s, failures: %d" % (os.environ.get("N", "60"), bad))
sys.exit(1 if bad else 0)
