#!/usr/bin/env python3
"""GPU lab: the ALGO_INTERP_LORENZO tuner with its device-side estimate and with the reference's own pricing
(sz3hip_ctx_set_tuner_exact) over the sweep of tests/checks/estimator_study.py — decisions equal to the oracle's out of how many —
and what the switch costs per call (C3: 512^3 f32 at 1e-4; a 256^3 and a 1024 x 1024 call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sz3_amd
from fields import field2d, field3d, field4d
from oracle_binding import ALGO_INTERP_LORENZO, make_config, oracle_tune

dev = torch.device("cuda:0")
cases = []
for S in (96, 128, 160, 200):
    for eb in (1e-1, 3e-2, 1e-2, 3e-3, 1e-3, 1e-4, 1e-5):
        cases.append(("3d-%d-%g" % (S, eb), lambda S=S: field3d((S, S, S)), eb))
for eb in (1e-2, 1e-3, 1e-4): cases.append(("2d-%g" % eb, lambda: field2d((600, 700)), eb))
for eb in (1e-2, 1e-3, 1e-4): cases.append(("4d-%g" % eb, lambda: field4d((12, 40, 40, 40)), eb))
for eb in (1e-5, 1e-6, 1e-7): cases.append(("f64-%g" % eb, lambda: field3d((80, 90, 100), np.float64, sigma=2e-6), eb))
for sg in (1e-3, 1e-2):
    for eb in (1e-2, 1e-3): cases.append(("noisy%g-%g" % (sg, eb), lambda sg=sg: field3d((128, 128, 128), sigma=sg), eb))

def tune(a, eb, exact):
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    dc.set_tuner_exact(exact)
    cap = dc.payload_bound(a.size)
    payload = torch.empty(cap, dtype=torch.uint8, device=dev)
    conf = sz3_amd.Config(*a.shape)
    conf.absErrorBound = eb
    size = dc.compress(conf, t.data_ptr(), payload.data_ptr(), cap, torch.cuda.current_stream().cuda_stream)
    g = dc.tuner_report()
    return (g["interpAlgo"], g["interpDirection"], g["interpAlpha"], g["interpBeta"]), size, g

if not os.environ.get("SKIP_SWEEP"):
    agree = {False: 0, True: 0}; same_bytes = 0; tot = 0; bigger = []
    for name, gen, eb in cases:
        a = gen()
        oc, rep, ran = oracle_tune(a, make_config(a.shape, algo=ALGO_INTERP_LORENZO, abs_eb=eb, regression=True))
        if not ran: continue
        tot += 1
        ref = (oc.interpAlgo, oc.interpDirection, oc.interpAlpha, oc.interpBeta)
        raw = rep.n_blocks * (rep.sample_block_size + 1) ** a.ndim * a.itemsize
        row = "%-16s ref %s" % (name, ref)
        sizes = {}
        for exact in (False, True):
            try:
                d, size, g = tune(a, eb, exact)
            except sz3_amd.SZ3HipError as e:  # (the device API's lists of unpredictable values: the host API repeats with larger ones)
                row += " | %s" % e
                sizes[exact] = 1
                continue
            sizes[exact] = size
            agree[exact] += d == ref
            row += " | %s %s" % ("exact" if exact else "estimate", "ok" if d == ref else str(d))
            if exact: same_bytes += [int(x) for x in g["est_bytes"][:6]] == [int(round(raw / rep.ratios[k])) for k in range(6)]
        row += " | payload %d vs %d (%+.2f %%)" % (sizes[False], sizes[True], 100.0 * (sizes[False] - sizes[True]) / sizes[True])
        print(row, flush=True)
    print("cases %d: decisions equal to the reference's — estimate %d, exact %d; exact sizes byte for byte in %d" % (tot, agree[False], agree[True], same_bytes))

def timed(a, eb, exact, reps=12):
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    dc.set_tuner_exact(exact)
    cap = dc.payload_bound(a.size)
    payload = torch.empty(cap, dtype=torch.uint8, device=dev)
    conf = sz3_amd.Config(*a.shape)
    conf.absErrorBound = eb
    s = torch.cuda.current_stream().cuda_stream
    ts = []
    for k in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        size = dc.compress(conf, t.data_ptr(), payload.data_ptr(), cap, s)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    g = dc.tuner_report()
    return min(ts[2:]) * 1e3, ts[0] * 1e3, a.nbytes / size, g

for name, gen, eb in (("C3 512^3 f32 1e-4", lambda: field3d((512, 512, 512)), 1e-4), ("256^3 f32 1e-3", lambda: field3d((256, 256, 256)), 1e-3),
                      ("2048^2 f32 1e-3", lambda: field2d((2048, 2048)), 1e-3)):
    a = gen()
    for exact in (False, True):
        ms, first, ratio, g = timed(a, eb, exact)
        print("%-20s %-8s %.3f ms per call (first %.2f), ratio %.3f, %d blocks of %d, outcome %s" % (name, "exact" if exact else "estimate", ms, first, ratio, g["n_blocks"],
              g["sample_block_size"] + 1, (g["interpAlgo"], g["interpDirection"], g["interpAlpha"], g["interpBeta"])), flush=True)
