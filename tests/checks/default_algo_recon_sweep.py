#!/usr/bin/env python3
"""randomised sweep: the reference's default algorithm (ALGO_INTERP_LORENZO) through the host API in this library's OWN stream format — the
array it decompresses to must be, bit for bit, the array the reference's stream decompresses to (the tuner's decisions are the reference's by
default at this boundary, and interpolation reconstructs the reference's values). 2-D .. 4-D (a 1-D array's tuner takes Lorenzo, whose own
stream here reconstructs on the 2 eb lattice). SEED, N from the environment; exit code = mismatches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, sz3_amd
from fields import field2d, field3d, field4d
from oracle_binding import ALGO_INTERP_LORENZO, EB_REL, make_config, oracle_compress, oracle_decompress
os.environ.pop("SZ3HIP_TUNER_EXACT", None)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
bad = n_cases = 0
for k in range(int(os.environ.get("N", "20"))):
    nd = int(rng.choice([2, 3, 3, 4]))
    dtype = np.float64 if rng.random() < 0.25 else np.float32
    if nd == 2: shape = tuple(int(rng.integers(200, 1500)) for _ in range(2))
    elif nd == 3: shape = tuple(int(rng.integers(40, 180)) for _ in range(3))
    else: shape = (int(rng.integers(8, 16)),) + tuple(int(rng.integers(24, 50)) for _ in range(3))
    sigma = float(rng.choice([0.0, 0.0, 1e-3]))
    a = {2: lambda: field2d(shape, dtype), 3: lambda: field3d(shape, dtype, sigma=sigma) if sigma else field3d(shape, dtype), 4: lambda: field4d(shape, dtype)}[nd]()
    conf = sz3_amd.Config(*a.shape)
    kw = {}
    ebv = float(10.0 ** rng.uniform(-4.5, -1))
    if rng.random() < 0.3:
        conf.errorBoundMode = sz3_amd.EB_REL; conf.relErrorBound = ebv; kw.update(eb_mode=EB_REL, rel_eb=ebv)
    else:
        conf.absErrorBound = ebv; kw.update(abs_eb=ebv)
    blob, _ = sz3_amd.compress(a, conf)
    dec, _ = sz3_amd.decompress(blob, a.dtype, a.shape)
    ob = oracle_compress(a, make_config(a.shape, algo=ALGO_INTERP_LORENZO, regression=True, **kw))
    odec, oc = oracle_decompress(ob, a.dtype, a.shape)
    n_cases += 1
    if not np.array_equal(dec, odec):
        bad += 1
        print("MISMATCH case %d %s %s %s sigma %g (the reference's stream: cmprAlgo %d)" % (k, shape, dtype.__name__, kw, sigma, oc.cmprAlgo), flush=True)
print("arrays %d, mismatches %d" % (n_cases, bad))
sys.exit(1 if bad else 0)
