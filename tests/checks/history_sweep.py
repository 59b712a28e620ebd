#!/usr/bin/env python3
"""randomised sweep: ONE device context through a random series of calls — predictor sets, bounds, shapes (1-D .. 4-D, up to 4 M elements: the
sampled book's path among them), a decompression now and then — against a FRESH context on every call: the payload is a function of the
input and the configuration, whatever the context did before (speculated books and decisions, hints, counters zeroed behind a call).
SEED, N from the environment; exit code = mismatches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field1d, field2d, field3d, field4d, field_c4a
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("N", "60"))
dev = torch.device("cuda:0")
MAXN = 1 << 22
bad = 0
for dtype in (np.float32, np.float64):
    shared = sz3_amd.DeviceCompressor(MAXN, dtype)
    cap = shared.payload_bound(MAXN, worst_case=True) + (1 << 20)
    pl = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2)]
    pool = {}
    def field(kind):
        if kind not in pool:
            if kind == "1d": a = field1d(int(rng.integers(2000, 1 << 20)), dtype)
            elif kind == "2d": a = field2d((int(rng.integers(40, 900)), int(rng.integers(40, 900))), dtype)
            elif kind == "3d": a = field3d(tuple(int(rng.integers(12, 120)) for _ in range(3)), dtype)
            elif kind == "3d-ramps": a = (field_c4a((40, 60, 90), seed=5) * (1.0 if dtype == np.float64 else 1000.0)).astype(dtype)
            elif kind == "big": a = field3d((64, 256, 256), dtype)   # 4 M elements, x a multiple of 256: the sampled book
            elif kind == "4d": a = field4d((int(rng.integers(4, 9)),) + tuple(int(rng.integers(10, 30)) for _ in range(3)), dtype)
            # WILD=1: what a context sees between smooth fields in practice (round 6, beside tests/checks/wild_data_sweep.py)
            elif kind == "w-noise": a = rng.standard_normal(tuple(int(rng.integers(40, 100)) for _ in range(3))).astype(dtype)
            elif kind == "w-const": a = np.full((int(rng.integers(100, 900)), int(rng.integers(100, 900))), -3.25, dtype)
            elif kind == "w-zeros": a = np.zeros((64, 256, 256), dtype)
            elif kind == "w-spikes":
                a = field3d(tuple(int(rng.integers(30, 120)) for _ in range(3)), dtype)
                a.reshape(-1)[rng.integers(0, a.size, size=a.size // 700)] = 1e30
            elif kind == "w-steps":
                a = field1d(int(rng.integers(50000, 1 << 20)), dtype)
                for c in rng.integers(0, a.size, size=7): a[c:] += dtype(float(rng.choice([-100.0, 7.5, 1000.0])))
            elif kind == "w-bignoise": a = (field3d((64, 256, 256), dtype) + 0.05 * rng.standard_normal((64, 256, 256))).astype(dtype)
            elif kind == "w-tight": a = (1000.0 * field1d(int(rng.integers(20000, 400000)), np.float64)).astype(dtype)
            else: raise ValueError(kind)
            pool[kind] = (a, torch.from_numpy(a).to(dev))
        return pool[kind]
    kinds = ["1d", "2d", "3d", "3d-ramps", "4d", "big"]
    if os.environ.get("WILD"): kinds += ["w-noise", "w-const", "w-zeros", "w-spikes", "w-steps", "w-bignoise", "w-tight"]
    out = torch.empty(MAXN, dtype=torch.float32 if dtype == np.float32 else torch.float64, device=dev)
    last = None
    for k in range(N // 2):
        kind = kinds[int(rng.integers(0, len(kinds)))] if rng.random() < 0.85 else "big"
        a, t = field(kind)
        conf = sz3_amd.Config(*a.shape)
        algo = str(rng.choice(["lorenzo", "composed", "l12", "interp", "default"]))
        if kind == "3d-ramps": ebv = float(10.0 ** rng.uniform(-6.5, -5.5)) if dtype == np.float64 else float(10.0 ** rng.uniform(-3.5, -2.5))
        else: ebv = float(10.0 ** rng.uniform(-4, -2))
        conf.absErrorBound = ebv
        if algo == "lorenzo": conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, 0
        elif algo == "composed": conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, 1
        elif algo == "l12": conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.lorenzo, conf.lorenzo2, conf.regression = 1, (0 if a.ndim == 4 else 1), 0
        elif algo == "interp": conf.cmprAlgo = sz3_amd.ALGO_INTERP
        else: conf.cmprAlgo = sz3_amd.ALGO_INTERP_LORENZO
        if rng.random() < 0.3: shared.set_speculation(bool(rng.integers(0, 2)))
        try:
            n1 = shared.compress(conf, t.data_ptr(), pl[0].data_ptr(), cap, 0)
        except sz3_amd.SZ3HipError as e:
            print("case %d %s %s %s: %s (skipped)" % (k, dtype.__name__, kind, algo, str(e)[:70])); continue
        fresh = sz3_amd.DeviceCompressor(a.size, dtype)
        n2 = fresh.compress(conf, t.data_ptr(), pl[1].data_ptr(), cap, 0)
        torch.cuda.synchronize()
        same = n1 == n2 and bool(torch.equal(pl[0][:n1], pl[1][:n2]))
        if not same:
            bad += 1
            print("MISMATCH case %d %s %s %s eb %.3g shape %s: %d vs %d bytes" % (k, dtype.__name__, kind, algo, ebv, a.shape, n1, n2), flush=True)
        if rng.random() < 0.35:
            shared.decompress(pl[0].data_ptr(), n1, out.data_ptr(), 0); torch.cuda.synchronize()
            err = float((out[:a.size].double() - t.reshape(-1).double()).abs().max())
            if not err <= ebv * (1 + 1e-6):
                bad += 1
                print("ERROR BOUND case %d %s %s %s: %.3g > %.3g" % (k, dtype.__name__, kind, algo, err, ebv), flush=True)
        del fresh
print("history sweep: calls %d, mismatches %d" % (N // 2 * 2, bad))
sys.exit(1 if bad else 0)
