import os, sys, ctypes as C
R = os.environ.get("AB_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, torch, sz3_amd
from fields import field3d
dev = torch.device("cuda:0")
shape = tuple(int(v) for v in os.environ.get("SHAPE", "80,520,500").split(",")); eb = 1e-3
a = field3d(shape); a[shape[0] // 2:] += float(os.environ.get("STEP", "2000"))
t = torch.from_numpy(a).to(dev)
L = sz3_amd.lib()
HAS = hasattr(L, "sz3hip_debug_decode_info")
if HAS: L.sz3hip_debug_decode_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
info = (C.c_uint32 * 4)()
conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.regression = 0; conf.absErrorBound = eb
for flag in (2097152, 0, 0, 0):
    L.sz3hip_debug_flags(flag)
    dc = sz3_amd.DeviceCompressor(a.size, np.float32) if flag or 'dc' not in dir() else dc
    cap = dc.payload_bound(a.size, worst_case=True); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    n = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
    out = torch.empty_like(t); dc.decompress(pl.data_ptr(), n, out.data_ptr(), 0); torch.cuda.synchronize()
    if HAS: L.sz3hip_debug_decode_info(dc._h, info)
    st = dc.stats(); print("   stats", {k: st[k] for k in ("n_value_outliers", "n_delta_outliers", "narrow_codes")})
    err = (out.double() - t.double()).abs()
    print("flag", flag, "modes", list(info), "max err", float(err.max()), "n bad", int((err > eb).sum()), "first bad", (err > eb).nonzero()[:3].tolist())
L.sz3hip_debug_flags(0)
