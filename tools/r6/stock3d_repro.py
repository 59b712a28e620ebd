#!/usr/bin/env python3
"""one stock ALGO_LORENZO_REG container against the oracle's, with the blocks' choices compared: tools/r6/stock3d_repro.py"""
import os, sys, struct
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SZ3HIP_STOCK_ONE_FRAME"] = "1"; os.environ["SZ3HIP_STOCK_SELECT_TRACE"] = "1"
import numpy as np, sz3_amd
from fields import field3d
from oracle_binding import ALGO_LORENZO_REG, EB_REL, make_config, oracle_compress, oracle_decompress, oracle
shape = (108, 20, 94); rel = 0.018932569718267355
a = field3d(shape, np.float64)
conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.lorenzo, conf.lorenzo2, conf.regression = 1, 1, 1
conf.errorBoundMode = sz3_amd.EB_REL; conf.relErrorBound = rel
L = sz3_amd.lib(); L.sz3hip_set_stock_format(1)
blob, _ = sz3_amd.compress(a, conf); L.sz3hip_set_stock_format(0)
ob = oracle_compress(a, make_config(a.shape, eb_mode=EB_REL, rel_eb=rel, algo=ALGO_LORENZO_REG, lorenzo=True, lorenzo2=True, regression=True))
def raw(b):
    b = b.tobytes(); plen, = struct.unpack_from("<Q", b, 8); pay = np.frombuffer(b[16:16 + plen], dtype=np.uint8)
    rl, = struct.unpack_from("<Q", pay.tobytes(), 0); r = np.empty(rl, dtype=np.uint8)
    assert oracle().szo_zstd_decompress(pay.ctypes.data, pay.size, r.ctypes.data, rl) == rl
    return r
r1, r2 = raw(blob), raw(ob)
print("sizes", blob.size, ob.size, "raw", r1.size, r2.size)
m = min(r1.size, r2.size); d = np.nonzero(r1[:m] != r2[:m])[0]
print("first raw diffs at", d[:8], "count", d.size)
d1, _ = oracle_decompress(blob, a.dtype, a.shape); d2, _ = oracle_decompress(ob, a.dtype, a.shape)
dd = np.argwhere(d1 != d2)
print("decoded differ at", dd[:4].tolist(), "count", len(dd), "-> first block", (dd[0] // 6).tolist() if len(dd) else None)
