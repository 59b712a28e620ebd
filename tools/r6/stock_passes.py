#!/usr/bin/env python3
"""how many selection passes a larger stock ALGO_LORENZO_REG container takes, and how long the call is: tools/r6/stock_passes.py [rel_eb]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SZ3HIP_STOCK_SELECT_TRACE"] = "1"
import numpy as np, sz3_amd
from fields import field3d
rel = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-2
a = field3d((256, 256, 256))
conf = sz3_amd.Config(*a.shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, 1
conf.errorBoundMode = sz3_amd.EB_REL; conf.relErrorBound = rel
L = sz3_amd.lib(); L.sz3hip_set_stock_format(1)
blob, _ = sz3_amd.compress(a, conf)
t0 = time.perf_counter(); blob, r = sz3_amd.compress(a, conf); dt = time.perf_counter() - t0
L.sz3hip_set_stock_format(0)
print("256^3 f32, Lorenzo + regression, REL %g, stock container: %.1f ms, ratio %.2f" % (rel, dt * 1e3, r))
