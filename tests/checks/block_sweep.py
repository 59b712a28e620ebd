#!/usr/bin/env python3
"""random sweep of the block-composed path (predictor sets with Lorenzo-2 / regression, 3-D, block edges 4..8): strict bound,
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
for k in range(int(os.environ.get("N", "40"))):
    shape = tuple(int(rng.choice(pool)) for _ in range(3))
    dt = np.float32 if rng.random() < 0.6 else np.float64
    z, y, x = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
    amp = float(10.0 ** rng.integers(-3, 2))
    a = amp * (np.sin(2 * np.pi * x / 23) * np.cos(2 * np.pi * y / 17) + 0.3 * np.sin(2 * np.pi * (x + y + 2 * z) / 11) + float(rng.choice([0.0, 0.01, 0.1])) * rng.standard_normal(shape))
    a = a.astype(dt)
    if k % 4 == 0: a.reshape(-1)[rng.integers(0, a.size, size=3)] = np.nan
    if k % 7 == 0: a[shape[0] // 2:, :, :] += 300 * amp
    eb = amp * float(10.0 ** rng.integers(-4, -1))
    mask = MASKS[int(rng.integers(0, len(MASKS)))]
    block = int(rng.choice([4, 5, 6, 6, 6, 7, 8]))
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
            model, sel = szh_ref.reconstruct_blocks(h, sec, szh_ref.huffman_decode(h, sec))
            u = np.uint32 if dt == np.float32 else np.uint64
            ok = ok and np.array_equal(model.reshape(-1).view(u), dec.reshape(-1).view(u))
            what += " reg %.0f%%" % (100.0 * np.mean(sel == 2))
    if not ok: bad += 1
    print(k, shape, dt.__name__, mask, "B", block, "qb", conf.quantbinCnt, "eb %.1e" % eb, what, "ratio %.2f" % ratio, "OK" if ok else "MISMATCH")
print("mismatches:", bad)
