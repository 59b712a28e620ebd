#!/usr/bin/env python3
"""host API, the reference's default algorithm (ALGO_INTERP_LORENZO) at 512^3 f32 1e-4 and 256^3 1e-3: a call with the tuner's trials priced the
reference's way (the host API's default since the end of round 5) against the device-side estimate (SZ3HIP_TUNER_EXACT=0)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, sz3_amd
from fields import field3d
for shape, eb in (((512, 512, 512), 1e-4), ((256, 256, 256), 1e-3), ((512, 512, 512), -1e-4)):
    a = field3d(shape)
    conf = sz3_amd.Config(*shape)
    if eb < 0: conf.errorBoundMode = sz3_amd.EB_REL; conf.relErrorBound = -eb   # (negative: relative to the range)
    else: conf.absErrorBound = eb
    out = np.empty(sz3_amd.compress_bound(conf, a.dtype), dtype=np.uint8)
    for mode in ("0", "1", None):
        if mode is None: os.environ.pop("SZ3HIP_TUNER_EXACT", None)
        else: os.environ["SZ3HIP_TUNER_EXACT"] = mode
        ts = []
        for k in range(8):
            t0 = time.perf_counter(); blob, ratio = sz3_amd.compress(a, conf, out=out); ts.append(time.perf_counter() - t0)
        print("%s eb %g  SZ3HIP_TUNER_EXACT=%s: %.2f ms per call (best of 6 after 2), ratio %.3f" % ("x".join(map(str, shape)), eb, mode, min(ts[2:]) * 1e3, ratio), flush=True)
