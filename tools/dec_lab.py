#!/usr/bin/env python3
"""A/B timings of the decoder across builds of the kernels: SZ3HIP_LIB=<variant .so> python tools/dec_lab.py
C2 (Lorenzo, 1e-3: fused x prefix sum, half-width chain) and C3 (interpolation, 1e-4: plain code output), 512^3 f32."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
S = int(os.environ.get("LAB_SIZE", "512"))
a = field3d((S, S, S)); dev = torch.device("cuda:0")
d_in = torch.from_numpy(a).to(dev)
stream = torch.cuda.current_stream().cuda_stream
name = os.path.basename(os.environ.get("SZ3HIP_LIB", "default"))
FLAGS = [int(x) for x in sys.argv[1:]] or [0]   # sz3hip_debug_flags values to time (2: the one-symbol table instead of the multi-symbol one)
for flag, (label, algo, eb) in [(f, c) for f in FLAGS for c in (("C2", sz3_amd.ALGO_LORENZO_REG, 1e-3), ("C3", sz3_amd.ALGO_INTERP_LORENZO, 1e-4))]:
    sz3_amd.lib().sz3hip_debug_flags(0)
    conf = sz3_amd.Config(S, S, S); conf.cmprAlgo = algo; conf.regression = 0; conf.absErrorBound = eb
    dc = sz3_amd.DeviceCompressor(a.size, np.float32)
    cap = dc.payload_bound(a.size); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    out = torch.empty_like(d_in)
    size = dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, stream)
    sz3_amd.lib().sz3hip_debug_flags(flag)
    for _ in range(3): dc.decompress(pl.data_ptr(), size, out.data_ptr(), stream)
    torch.cuda.synchronize()
    err = float((out - d_in).abs().max())
    dc.set_profiling(True)
    huff = []; rec = []
    for _ in range(8):
        dc.decompress(pl.data_ptr(), size, out.data_ptr(), stream); torch.cuda.synchronize()
        t = dc.stage_times(); huff.append(t.get("huffman_decode", 0)); rec.append(t.get("reconstruct", 0))
    dc.set_profiling(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dc.decompress(pl.data_ptr(), size, out.data_ptr(), stream)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
    print("%-22s flags %d %s: decode stage %.1f us  reconstruct %.1f us  wall %.1f us  ratio %.3f  max err %.3g (eb %g)" % (
        name, flag, label, 1e3 * np.median(huff), 1e3 * np.median(rec), 1e3 * wall, a.nbytes / size, err, eb), flush=True)
sz3_amd.lib().sz3hip_debug_flags(0)
