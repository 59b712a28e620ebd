#!/usr/bin/env python3
"""randomised sweep: containers written in stock format (sz3hip_set_stock_format, one zstd frame) against the oracle's — the reference's —
bytes: ALGO_INTERP with random parameters, the default algorithm (the host API prices its tuner the reference's way), ALGO_LORENZO_REG
with one-member and (round 6) mixed sets. SEED, N from the environment; exit code = mismatches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, sz3_amd
from fields import field1d, field2d, field3d, field4d
from oracle_binding import ALGO_INTERP, ALGO_INTERP_LORENZO, ALGO_LORENZO_REG, EB_REL, make_config, oracle_compress
os.environ["SZ3HIP_STOCK_ONE_FRAME"] = "1"
os.environ.pop("SZ3HIP_TUNER_EXACT", None)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
L = sz3_amd.lib()
bad = n_cases = 0
for k in range(int(os.environ.get("N", "30"))):
    nd = int(rng.choice([1, 2, 3, 3, 4]))
    dtype = np.float64 if rng.random() < 0.25 else np.float32
    if nd == 1: shape = (int(rng.integers(3000, 400000)),)
    elif nd == 2: shape = tuple(int(rng.integers(40, 700)) for _ in range(2))
    elif nd == 3: shape = tuple(int(rng.integers(12, 110)) for _ in range(3))
    else: shape = (int(rng.integers(5, 14)),) + tuple(int(rng.integers(10, 40)) for _ in range(3))
    a = {1: lambda: field1d(shape[0], dtype), 2: lambda: field2d(shape, dtype), 3: lambda: field3d(shape, dtype), 4: lambda: field4d(shape, dtype)}[nd]()
    kind = str(rng.choice(["interp", "interp", "default", "default", "lorenzo"]))
    conf = sz3_amd.Config(*a.shape)
    conf.regression = 0
    kw = {}
    rel = rng.random() < 0.3
    ebv = float(10.0 ** rng.uniform(-4, -1.5))
    if rel:
        conf.errorBoundMode = sz3_amd.EB_REL; conf.relErrorBound = ebv; kw.update(eb_mode=EB_REL, rel_eb=ebv)
    else:
        conf.absErrorBound = ebv; kw.update(abs_eb=ebv)
    if kind == "interp":
        fact = [1, 1, 2, 6, 24][nd]
        p = dict(interp_algo=int(rng.integers(0, 2)), interpDirection=int(rng.integers(0, fact)), interpAlpha=float(rng.choice([1.0, 1.25, 1.5, 2.0])),
                 interpBeta=float(rng.choice([1.0, 2.0, 2.5, 3.0])))
        conf.cmprAlgo = sz3_amd.ALGO_INTERP
        conf.interpAlgo, conf.interpDirection, conf.interpAlpha, conf.interpBeta = p["interp_algo"], p["interpDirection"], p["interpAlpha"], p["interpBeta"]
        kw.update(algo=ALGO_INTERP, **p)
    elif kind == "default":
        kw.update(algo=ALGO_INTERP_LORENZO)
    else:
        # (round 6: mixed sets in every dimension — the writer's repeated selection settles on the reference's choices; 4-D without the second-order
        # member, which the oracle does not restate there)
        sets = [(1, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 1), (1, 1, 1)] if nd != 4 else [(1, 0, 0), (1, 0, 1)]
        if nd == 1: sets += [(0, 0, 1)]
        l1, l2, rg = sets[int(rng.integers(0, len(sets)))]
        conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
        conf.lorenzo, conf.lorenzo2, conf.regression = l1, l2, rg
        kw.update(algo=ALGO_LORENZO_REG, lorenzo=bool(l1), lorenzo2=bool(l2), regression=bool(rg))
    L.sz3hip_set_stock_format(1)
    try:
        blob, _ = sz3_amd.compress(a, conf)
    except sz3_amd.SZ3HipError as e:
        print("case %d %s %s: %s (skipped)" % (k, kind, shape, str(e)[:80])); continue
    finally:
        L.sz3hip_set_stock_format(0)
    ob = oracle_compress(a, make_config(a.shape, **kw))
    n_cases += 1
    if blob.tobytes() != ob.tobytes():
        bad += 1
        print("MISMATCH case %d %s %s %s %s: %d vs %d bytes" % (k, kind, shape, dtype.__name__, kw, blob.size, ob.size), flush=True)
print("containers %d, mismatches %d" % (n_cases, bad))
sys.exit(1 if bad else 0)
