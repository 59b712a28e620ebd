#!/usr/bin/env python3
"""random sweep of the block-composed path (predictor sets with Lorenzo-2 / regression; 3-D with block edges 4..8, or NDIM=1 / 2 / 4:
blocks of 4..200 values, edges 4..32, edges 4..6 — the 4-D sets without the second-order member): strict bound,
non-finite values bit for bit, and the numpy model of the block decoder (tests/szh_ref.py) reproduces the GPU's reconstruction
from the stream bit for bit — a development check beyond tests/test_gpu_regression.py.  SEED=.. N=.. python tests/checks/block_sweep.py"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, sz3_amd, szh_ref
from oracle_binding import oracle
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
if os.environ.get("DBG_FLAGS"): sz3_amd.lib().sz3hip_debug_flags(int(os.environ["DBG_FLAGS"]))

def payload_of(stream):
    b = stream.tobytes()
    plen, = struct.unpack_from("<Q", b, 8)
    blob = np.frombuffer(b[16:16 + plen], dtype=np.uint8).copy()
    rawlen, = struct.unpack_from("<Q", blob.tobytes(), 0)
    out = np.empty(rawlen, dtype=np.uint8)
    assert oracle().szo_zstd_decompress(blob.ctypes.data, blob.size, out.ctypes.data, rawlen) == rawlen
    return out.tobytes()

MASKS = [(0, 0, 1), (0, 1, 0), (1, 0, 1), (1, 1, 0), (1, 1, 1), (0, 1, 1)]
pool = [5, 6, 7, 8, 11, 12, 13, 17, 18, 24, 25, 30, 31, 36, 48, 49, 64]
bad = 0
ND = int(os.environ.get("NDIM", "3"))
for k in range(int(os.environ.get("N", "40"))):
    if ND == 3: shape = tuple(int(rng.choice(pool)) for _ in range(3))
    elif ND == 1: shape = (int(rng.integers(1, 60000)),)
    elif ND == 2: shape = tuple(int(rng.integers(1, 300)) for _ in range(2))
    else: shape = tuple(int(rng.choice([1, 4, 5, 6, 7, 9, 12, 13, 19, 24])) for _ in range(4))
    dt = np.float32 if rng.random() < 0.6 else np.float64
    g3 = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in ((1,) * (3 - min(ND, 3)) + shape[-3:])], indexing="ij")
    z, y, x = [g.reshape(shape[-3:] if ND >= 3 else shape) for g in g3]
    amp = float(10.0 ** rng.integers(-3, 2))
    a3 = amp * (np.sin(2 * np.pi * x / 23) * np.cos(2 * np.pi * y / 17) + 0.3 * np.sin(2 * np.pi * (x + y + 2 * z) / 11))
    a = (np.broadcast_to(a3, shape) * (1 + 0.02 * np.arange(shape[0]).reshape((-1,) + (1,) * (ND - 1))) if ND == 4 else a3) + amp * float(rng.choice([0.0, 0.01, 0.1])) * rng.standard_normal(shape)
    a = np.ascontiguousarray(a).astype(dt)
    if k % 4 == 0: a.reshape(-1)[rng.integers(0, a.size, size=3)] = np.nan
    if k % 7 == 0: a[shape[0] // 2:] += 300 * amp
    eb = amp * float(10.0 ** rng.integers(-4, -1))
    mask = MASKS[int(rng.integers(0, len(MASKS)))]
    if ND == 4: mask = [(0, 0, 1), (1, 0, 1)][int(rng.integers(0, 2))]
    block = int(rng.choice([4, 5, 6, 6, 6, 7, 8])) if ND == 3 else int(rng.choice([4, 5, 8, 16, 100, 128, 200])) if ND == 1 else int(rng.choice([4, 7, 8, 16, 16, 32])) if ND == 2 else int(rng.choice([4, 5, 6, 6]))
    conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.lorenzo, conf.lorenzo2, conf.regression = mask
    conf.absErrorBound = eb; conf.blockSize = block; conf.quantbinCnt = int(rng.choice([256, 1024, 65536]))
    try:
        blob, ratio = sz3_amd.compress(a, conf)
        dec, c2 = sz3_amd.decompress(blob, dt, shape)
    except sz3_amd.SZ3HipError as e:
        print(k, shape, dt.__name__, mask, "B", block, "refused:", str(e)[:70]); continue
    fin = np.isfinite(a)
    ok = np.array_equal(np.isnan(dec), np.isnan(a)) and (not fin.any() or float(np.max(np.abs(dec[fin].astype(np.float64) - a[fin].astype(np.float64)))) <= eb)
    what = "lossless" if c2.cmprAlgo == sz3_amd.ALGO_LOSSLESS else "?"
    if c2.cmprAlgo != sz3_amd.ALGO_LOSSLESS:
        h, o, sec = szh_ref.parse(payload_of(blob))
        what = "pred %d" % h["predictor"]
        if h["predictor"] == 2:
            if a.size > 30000 and ND == 4:  # (the 4-D numpy model walks element by element)
                print(k, shape, "model skipped"); continue
            model, sel = szh_ref.reconstruct_blocks(h, sec, szh_ref.huffman_decode(h, sec))
            u = np.uint32 if dt == np.float32 else np.uint64
            ok = ok and np.array_equal(model.reshape(-1).view(u), dec.reshape(-1).view(u))
            what += " reg %.0f%%" % (100.0 * np.mean(sel == 2))
    if not ok: bad += 1
    print(k, shape, dt.__name__, mask, "B", block, "qb", conf.quantbinCnt, "eb %.1e" % eb, what, "ratio %.2f" % ratio, "OK" if ok else "MISMATCH")
print("mismatches:", bad)
