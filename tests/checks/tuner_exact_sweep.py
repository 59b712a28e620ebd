#!/usr/bin/env python3
"""randomised sweep: the ALGO_INTERP_LORENZO tuner with its trials priced the reference's way (sz3hip_ctx_set_tuner_exact) against the oracle's
run of the reference's trials — every interpolation trial's compressed size byte for byte, every decision; random shapes (1-D .. 4-D),
bounds, element types, smooth / noisy fields. SEED, N from the environment; exit code = mismatches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field1d, field2d, field3d, field4d
from oracle_binding import ALGO_INTERP, ALGO_INTERP_LORENZO, make_config, oracle_tune
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
dev = torch.device("cuda:0")
bad = ran = 0
for k in range(int(os.environ.get("N", "30"))):
    nd = int(rng.choice([1, 2, 3, 3, 3, 4]))
    dtype = np.float64 if rng.random() < 0.25 else np.float32
    if nd == 1: shape = (int(rng.integers(200000, 3000000)),)
    elif nd == 2: shape = tuple(int(rng.integers(300, 1500)) for _ in range(2))
    elif nd == 3: shape = tuple(int(rng.integers(60, 200)) for _ in range(3))
    else: shape = (int(rng.integers(9, 20)),) + tuple(int(rng.integers(30, 56)) for _ in range(3))
    sigma = float(rng.choice([0.0, 0.0, 1e-3, 1e-2]))
    a = {1: lambda: field1d(shape[0], dtype), 2: lambda: field2d(shape, dtype), 3: lambda: field3d(shape, dtype, sigma=sigma) if sigma else field3d(shape, dtype),
         4: lambda: field4d(shape, dtype)}[nd]()
    eb = float(10.0 ** rng.uniform(-5, -1))
    oc, rep, oran = oracle_tune(a, make_config(a.shape, algo=ALGO_INTERP_LORENZO, abs_eb=eb, regression=True))
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype); dc.set_tuner_exact(True); dc.set_deterministic(True)
    cap = dc.payload_bound(a.size)
    pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    conf = sz3_amd.Config(*a.shape); conf.absErrorBound = eb
    try:
        size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, torch.cuda.current_stream().cuda_stream)
    except sz3_amd.SZ3HipError as e:
        print("case %d %s %s eb %.3g: %s (skipped)" % (k, shape, dtype.__name__, eb, str(e)[:60])); continue
    g = dc.tuner_report()
    msg = []
    if bool(g["ran"]) != oran: msg.append("ran %s vs %s" % (g["ran"], oran))
    if oran:
        ran += 1
        raw = rep.n_blocks * (rep.sample_block_size + 1) ** a.ndim * a.itemsize
        want = [int(round(raw / rep.ratios[i])) for i in range(6)]
        if [int(x) for x in g["est_bytes"][:6]] != want: msg.append("sizes %s vs %s" % ([int(x) for x in g["est_bytes"][:6]], want))
        if bool(g["use_interp"]) != (oc.cmprAlgo == ALGO_INTERP): msg.append("interp / Lorenzo")
        elif g["use_interp"] and (g["interpAlgo"], g["interpDirection"], g["interpAlpha"], g["interpBeta"]) != (oc.interpAlgo, oc.interpDirection, oc.interpAlpha, oc.interpBeta):
            msg.append("parameters")
        elif not g["use_interp"]:
            hdr = bytes(pl[:16].cpu().numpy())
            if int.from_bytes(hdr[12:16], "little") != oc.quantbinCnt // 2: msg.append("radius")
    if msg:
        bad += 1
        print("MISMATCH case %d %s %s eb %.3g sigma %g: %s" % (k, shape, dtype.__name__, eb, sigma, "; ".join(msg)), flush=True)
print("cases with trials %d, mismatches %d" % (ran, bad))
sys.exit(1 if bad else 0)
