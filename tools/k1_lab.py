#!/usr/bin/env python3
"""A/B timings of stage 1 (the predictor kernel) across builds of the kernels: SZ3HIP_LIB=<variant .so> python tools/k1_lab.py [flags...]
(flags = sz3hip_debug_flags values; with a LAB_ABLATE build 1 = no histogram atomics, 2 = no code stores: results are wrong)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
S = int(os.environ.get("LAB_SIZE", "512"))
a = field3d((S, S, S)); dev = torch.device("cuda:0")
d_in = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(S, S, S); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.regression = 0; conf.absErrorBound = 1e-3
dc = sz3_amd.DeviceCompressor(a.size, np.float32)
if os.environ.get('LAB_FUSED') == '1': dc.set_fused(True)   # (the fused stage 1, k_lorenzo_quant_march3f + k_merge)
cap = dc.payload_bound(a.size); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
L = sz3_amd.lib()
for _ in range(3): dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, stream)   # the context now holds its hints and a code book
flags = [int(x) for x in sys.argv[1:]] or [0]
for f in flags:
    L.sz3hip_debug_flags(f)
    dc.set_profiling(True)
    k1 = []; s1 = []; span = []; enc = []
    for _ in range(12):
        dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, stream)
        t = dc.stage_times(); k1.append(t.get("k1_kernel", 0)); s1.append(t.get("lorenzo_quant_hist", 0)); span.append(t.get("step_span", 0)); enc.append(t.get("encode", 0))
    dc.set_profiling(False)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(20): dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, stream)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e3
    print("%-24s flags %9d: k1 %.1f us  stage1 %.1f us  encode %.1f us  span %.1f us  wall %.1f us  spec %s" % (
        os.path.basename(os.environ.get("SZ3HIP_LIB", "default")), f, 1e3 * np.median(k1), 1e3 * np.median(s1), 1e3 * np.median(enc), 1e3 * np.median(span), 1e3 * wall, dc.spec_stats()))
L.sz3hip_debug_flags(0)
