#!/usr/bin/env python3
"""a 4 MB chunk through the HDF5 filter function (include/sz3hip_h5z.h), forward: wall time per call; under rocprofv3 (tools/r6/h5z_tl.sh) the
kernels of one call. tools/r6/h5z_chunk.py [1d|3d] [algo: default|lorenzo_reg]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, sz3_amd
from fields import field3d, field1d
from test_capi_cpu import _H5ZClass2, _cd_values
kind = sys.argv[1] if len(sys.argv) > 1 else "3d"
algo = sys.argv[2] if len(sys.argv) > 2 else "default"
a = field1d(1 << 20) if kind == "1d" else field3d((64, 128, 128))
L = sz3_amd.lib()
L.H5PLget_plugin_info.restype = C.POINTER(_H5ZClass2)
rec = L.H5PLget_plugin_info().contents
filt = C.CFUNCTYPE(C.c_size_t, C.c_uint, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p))(rec.filter)
libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.free.argtypes = [C.c_void_p]
conf = sz3_amd.Config(*a.shape); conf.absErrorBound = 1e-3; conf.dataType = 0
if algo == "lorenzo_reg": conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
cdv, words = _cd_values(conf._c)
ts = []
for it in range(12):
    buf = C.c_void_p(libc.malloc(a.nbytes)); C.memmove(buf, a.ctypes.data, a.nbytes); size = C.c_size_t(a.nbytes)
    t0 = time.perf_counter(); n = filt(0, words, cdv, a.nbytes, C.byref(size), C.byref(buf)); ts.append(time.perf_counter() - t0)
    assert 0 < n < a.nbytes
    if it == 11:  # the read direction, a few times over copies of the stream (the first call of a process pays its allocations)
        stream = C.string_at(buf, n); tds = []
        for _ in range(6):
            libc.free(buf); buf = C.c_void_p(libc.malloc(n)); C.memmove(buf, stream, n); size = C.c_size_t(n)
            t0 = time.perf_counter(); m = filt(0x0100, words, cdv, n, C.byref(size), C.byref(buf)); tds.append(time.perf_counter() - t0)
            assert m == a.nbytes
        td = sorted(tds[1:])[len(tds[1:]) // 2]
    libc.free(buf)
ts = sorted(ts[2:])
print("h5z chunk %s %s: %d -> %d bytes (ratio %.2f); filter call %.3f ms median (min %.3f); read direction %.3f ms" % (kind, algo, a.nbytes, n, a.nbytes / n, ts[len(ts) // 2] * 1e3, ts[0] * 1e3, td * 1e3))
