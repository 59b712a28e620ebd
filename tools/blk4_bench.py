"""timing of the block-composed predictor on a 4-D array, device-resident (development tool)
usage: python tools/blk4_bench.py nt,nz,ny,nx [eb] [plain]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, sz3_amd
from fields import field4d
shape = tuple(int(v) for v in sys.argv[1].split(","))
eb = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2
a = field4d(shape)
n = a.size
dev = torch.device("cuda:0")
d_in = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.absErrorBound = eb
conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, int("plain" not in sys.argv)
dc = sz3_amd.DeviceCompressor(n, np.float32)
cap = max(dc.payload_bound(n), dc.payload_bound_conf(conf))
d_pl = torch.empty(cap, dtype=torch.uint8, device=dev); d_out = torch.empty(n, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def comp():
    dc.stage1(conf, d_in.data_ptr(), st); dc.stage2(d_pl.data_ptr(), cap, st); return dc.finish(st)
for _ in range(3): size = comp()
torch.cuda.synchronize(); K = 10; t0 = time.perf_counter()
for _ in range(K): size = comp()
torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / K
for _ in range(2): dc.decompress(d_pl.data_ptr(), size, d_out.data_ptr(), st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): dc.decompress(d_pl.data_ptr(), size, d_out.data_ptr(), st)
torch.cuda.synchronize(); td = (time.perf_counter() - t0) / K
err = float((d_out.double() - d_in.reshape(-1).double()).abs().max())
print("%s eb %g %s: payload ratio %.2f; compress %.3f ms (%.1f GB/s), decompress %.3f ms (%.1f GB/s); max err %.3g (ok %s)" % (
    shape, eb, "Lorenzo-1" if "plain" in sys.argv else "Lorenzo + regression", a.nbytes / size, tc * 1e3, a.nbytes / tc / 1e9, td * 1e3, a.nbytes / td / 1e9, err, err <= eb))
