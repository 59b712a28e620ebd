import os, sys, struct
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["SZ3HIP_STOCK_ONE_FRAME"] = "1"
import numpy as np, sz3_amd
from fields import field1d
from oracle_binding import ALGO_LORENZO_REG, make_config, oracle_compress, oracle_decompress, oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 368898
eb = float(sys.argv[2]) if len(sys.argv) > 2 else 0.00014524286814550258
a = field1d(n, np.float32)
conf = sz3_amd.Config(n); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.lorenzo, conf.lorenzo2, conf.regression = 0, 0, 1; conf.absErrorBound = eb
L = sz3_amd.lib(); L.sz3hip_set_stock_format(1)
blob, _ = sz3_amd.compress(a, conf); L.sz3hip_set_stock_format(0)
ob = oracle_compress(a, make_config(a.shape, abs_eb=eb, algo=ALGO_LORENZO_REG, lorenzo=False, lorenzo2=False, regression=True))
def raw(b):
    b = b.tobytes(); plen, = struct.unpack_from("<Q", b, 8); pay = np.frombuffer(b[16:16 + plen], dtype=np.uint8)
    rl, = struct.unpack_from("<Q", pay.tobytes(), 0); r = np.empty(rl, dtype=np.uint8)
    assert oracle().szo_zstd_decompress(pay.ctypes.data, pay.size, r.ctypes.data, rl) == rl
    return r
r1, r2 = raw(blob), raw(ob)
print("sizes", blob.size, ob.size, "raw", r1.size, r2.size)
m = min(r1.size, r2.size); d = np.nonzero(r1[:m] != r2[:m])[0]
print("first diffs at", d[:10], "count", d.size)
d1, _ = oracle_decompress(blob, a.dtype, a.shape); d2, _ = oracle_decompress(ob, a.dtype, a.shape)
dd = np.nonzero(d1 != d2)[0]
print("decoded differ at", dd[:10], "count", dd.size, "max err ours", float(np.abs(d1.astype(np.float64) - a).max()), "ref", float(np.abs(d2.astype(np.float64) - a).max()))
