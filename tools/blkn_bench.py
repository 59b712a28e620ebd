"""Timing of the block-composed predictor on 1-D / 2-D arrays, device-resident (development tool; the numbers in DESIGN.md).
usage: python tools/blkn_bench.py n | dy,dx [eb] [f32|f64] [plain | l12 | default]   (l12: Lorenzo-1 + Lorenzo-2; default: the default algorithm, tuner included)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sz3_amd
from fields import field1d, field2d

shape = tuple(int(v) for v in sys.argv[1].split(","))
eb = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
dt = np.float64 if len(sys.argv) > 3 and sys.argv[3] == "f64" else np.float32
plain = "plain" in sys.argv
n = int(np.prod(shape))
if len(shape) == 1:
    base = field1d(min(n, 1 << 24), dt)
    a = np.tile(base, -(-n // base.size))[:n].copy()
else:
    a = field2d(shape, dt)
dev = torch.device("cuda:0")
d_in = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(*shape)
conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, int(not plain)
if "l12" in sys.argv:
    conf.lorenzo, conf.lorenzo2, conf.regression = 1, 1, 0
if "default" in sys.argv:
    conf = sz3_amd.Config(*shape)
conf.absErrorBound = eb
dc = sz3_amd.DeviceCompressor(n, dt, device=0)
cap = max(dc.payload_bound(n), dc.payload_bound_conf(conf))
d_pl = torch.empty(cap, dtype=torch.uint8, device=dev)
d_out = torch.empty(n, dtype=torch.float32 if dt == np.float32 else torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream

def comp():
    dc.stage1(conf, d_in.data_ptr(), st)
    dc.stage2(d_pl.data_ptr(), cap, st)
    return dc.finish(st)

for _ in range(3):
    size = comp()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    size = comp()
torch.cuda.synchronize()
tc = (time.perf_counter() - t0) / K
for _ in range(2):
    dc.decompress(d_pl.data_ptr(), size, d_out.data_ptr(), st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    dc.decompress(d_pl.data_ptr(), size, d_out.data_ptr(), st)
torch.cuda.synchronize()
td = (time.perf_counter() - t0) / K
err = float((d_out.double() - d_in.reshape(-1).double()).abs().max())
print("%s %s eb %g %s: payload ratio %.2f; compress %.3f ms (%.1f GB/s), decompress %.3f ms (%.1f GB/s); max err %.3g (ok %s)"
      % (shape, dt.__name__, eb, "Lorenzo-1 (plain stream)" if plain else "Lorenzo-1 + Lorenzo-2" if "l12" in sys.argv else "default algorithm" if "default" in sys.argv else "Lorenzo + regression", a.nbytes / size, tc * 1e3, a.nbytes / tc / 1e9,
         td * 1e3, a.nbytes / td / 1e9, err, err <= eb))
