"""C2 step under chosen context settings: ms per step, ratio. Usage: lab_c2.py [spec|nospec|det] [flags] [steps]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
import sz3_amd
from fields import field3d

mode = sys.argv[1] if len(sys.argv) > 1 else "spec"
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
S = int(os.environ.get("SIZE", "512"))
shape = (S, S, S)
dev = torch.device("cuda:0")
a = [torch.from_numpy(field3d(shape, seed=20260928)).to(dev), torch.from_numpy(field3d(shape, seed=20261928)).to(dev)]
n = a[0].numel()
conf = sz3_amd.Config(*shape)
conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, 0
conf.errorBoundMode = sz3_amd.EB_ABS
conf.absErrorBound = 1e-3
dc = sz3_amd.DeviceCompressor(n, np.float32)
if mode == "nospec":
    dc.set_speculation(False)
elif mode == "det":
    dc.set_deterministic(True)
cap = dc.payload_bound(n)
pl = torch.empty(cap, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
sz3_amd.lib().sz3hip_debug_flags(flags)
k = 0
def step():
    global k
    k ^= 1
    dc.stage1(conf, a[k].data_ptr(), st)
    dc.stage2(pl.data_ptr(), cap, st)
    return dc.finish(st)
for _ in range(4):
    ps = step()
torch.cuda.synchronize()
tt = []
t0 = time.perf_counter()
for _ in range(steps):
    t1 = time.perf_counter()
    ps = step()
    tt.append(time.perf_counter() - t1)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / steps
out = torch.empty_like(a[k])
dc.decompress(pl.data_ptr(), ps, out.data_ptr(), st)
torch.cuda.synchronize()
err = float((out.double() - a[k].double()).abs().max())
import ctypes as C
L = sz3_amd.lib(); L.sz3hip_debug_codebook_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
o = (C.c_uint64 * 16)(); L.sz3hip_debug_codebook_info(dc._h, o)
ts = [int(o[4 + i]) for i in range(12)]
print("book: n_symbols %d max_len %d | us since the sampling began: book start %.1f, sorted %.1f, merged %.1f, depths %.1f, lengths %.1f, ready %.1f" % (
    o[0], o[1], (ts[0] - ts[9]) / 100, (ts[3] - ts[9]) / 100, (ts[4] - ts[9]) / 100, (ts[5] - ts[9]) / 100, (ts[8] - ts[9]) / 100, (ts[10] - ts[9]) / 100))
print("mode %s flags %d: %.4f ms/step (median %.4f)  ratio %.4f  err %.3g  spec %s q16 %s" % (mode, flags, el * 1e3, sorted(tt)[len(tt) // 2] * 1e3, n * 4 / ps, err, dc.spec_stats(), dc.q16))
