#!/usr/bin/env python3
"""reading stock ALGO_LORENZO_REG streams: time of sz3_amd.decompress against the oracle's (= the reference's CPU path)
LAB_SIZE^3 f32 (default 256), Lorenzo + regression at 1e-2 / Lorenzo alone at 1e-3; a 1-D series of 2^22 values with the tuner's set"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, sz3_amd
from fields import field3d, field1d
from oracle_binding import make_config, oracle_compress, oracle_decompress
S = int(os.environ.get("LAB_SIZE", "256"))
cases = [("3-D %d^3 Lorenzo + regression 1e-2" % S, field3d((S, S, S)), dict(abs_eb=1e-2, lorenzo=True, regression=True)),
         ("3-D %d^3 Lorenzo 1e-3" % S, field3d((S, S, S)), dict(abs_eb=1e-3, lorenzo=True, regression=False)),
         ("1-D 2^22 Lorenzo-1 + Lorenzo-2 1e-3", field1d(1 << 22), dict(abs_eb=1e-3, lorenzo=True, lorenzo2=True, regression=False))]
for name, a, kw in cases:
    blob = oracle_compress(a, make_config(a.shape, **kw))
    t0 = time.perf_counter(); want, _ = oracle_decompress(blob, a.dtype, a.shape); tc = time.perf_counter() - t0
    for _ in range(2):
        t0 = time.perf_counter(); got, _ = sz3_amd.decompress(blob, a.dtype, a.shape); tg = time.perf_counter() - t0
    print("%s: ratio %.2f; CPU %.0f ms (%.2f GB/s), here %.1f ms (%.2f GB/s), identical %s" % (
        name, a.nbytes / len(blob), 1e3 * tc, a.nbytes / tc / 1e9, 1e3 * tg, a.nbytes / tg / 1e9, bool(np.array_equal(got, want))), flush=True)
