#!/usr/bin/env python3
"""1-D regression-only stock containers of arrays with NaN / Inf: where this library's stored (unpredictable) coefficients differ from the
reference's in the sign of a NaN, and what the block held. tools/r6/nan_sign_1d.py [seed]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SZ3HIP_STOCK_ONE_FRAME"] = "1"
import numpy as np, sz3_amd
from oracle_binding import ALGO_LORENZO_REG, make_config, ref_compress
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n, B = 128 * 840, 128
a = np.sin(np.arange(n) / 17.0).astype(np.float32)
kinds = {}
for b in range(0, 840, 30):
    k = ["nan", "inf", "-inf", "nan+inf", "inf+-inf", "nan first then inf", "inf first then nan"][(b // 30) % 7]
    kinds[b] = k
    p = b * B + int(rng.integers(5, 60))
    if k == "nan": a[p] = np.nan
    elif k == "inf": a[p] = np.inf
    elif k == "-inf": a[p] = -np.inf
    elif k == "nan+inf": a[p] = np.nan; a[p + 30] = np.inf
    elif k == "inf+-inf": a[p] = np.inf; a[p + 30] = -np.inf
    elif k == "nan first then inf": a[b * B] = np.nan; a[p + 30] = np.inf
    else: a[b * B] = np.inf; a[p + 30] = np.nan
conf = sz3_amd.Config(n); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.lorenzo, conf.lorenzo2, conf.regression = 0, 0, 1; conf.absErrorBound = 1e-3
L = sz3_amd.lib(); L.sz3hip_set_stock_format(1)
mine, _ = sz3_amd.compress(a, conf); mine = mine.copy()
L.sz3hip_set_stock_format(0)
ref = ref_compress(a, make_config(a.shape, algo=ALGO_LORENZO_REG, abs_eb=1e-3, lorenzo=False, lorenzo2=False, regression=True))
Z = C.CDLL("libzstd.so.1"); Z.ZSTD_decompress.restype = C.c_size_t
raws = []
for bl in (mine, ref):
    ln = int(np.frombuffer(bl[16:24].tobytes(), dtype=np.uint64)[0]); body = bl[24:].tobytes()
    buf = C.create_string_buffer(ln); Z.ZSTD_decompress(buf, C.c_size_t(ln), body, C.c_size_t(len(body)))
    raws.append(np.frombuffer(buf.raw[:ln], dtype=np.uint8))
print("containers equal:", mine.tobytes() == ref.tobytes(), "streams", raws[0].size, raws[1].size)
# the two quantisers' lists sit at the front: [u64 coefficient count][quantizer_independent: u8 uid, f64 eb, i32 radius, u64 count, values][quantizer_liner: same]
def lists(r):
    o = 8; out = []
    for _ in range(2):
        o += 1 + 8 + 4
        c = int(np.frombuffer(r[o:o + 8].tobytes(), dtype=np.uint64)[0]); o += 8
        out.append(np.frombuffer(r[o:o + 4 * c].tobytes(), dtype=np.uint32)); o += 4 * c
    return out
lm, lr = lists(raws[0]), lists(raws[1])
for name, x, y in (("independent", lm[0], lr[0]), ("linear", lm[1], lr[1])):
    print(name, "unpredictable coefficients:", x.size, y.size)
    for i in range(min(x.size, y.size)):
        print("   %2d  ours %08x  reference %08x  %s   block kind: %s" % (i, x[i], y[i], "" if x[i] == y[i] else "<-- differs", kinds.get(30 * i, "?")))
