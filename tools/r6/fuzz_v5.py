#!/usr/bin/env python3
"""damaged version-5 payloads (sampled book: code words up to 24 bits, an escape symbol) through the device decoder: header fields, the
lengths' table, chunk words, restart offsets, the bit stream — every call ends in an error or an array: tools/r6/fuzz_v5.py [cases] [seed]"""
import os, sys, struct
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd, szh_ref
from fields import field3d
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
shape = (64, 256, 256); a = field3d(shape); dev = torch.device("cuda:0"); t = torch.from_numpy(a).to(dev)
conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.regression = 0; conf.absErrorBound = 1e-3
dc = sz3_amd.DeviceCompressor(a.size, np.float32); dc.set_deterministic(True) if hasattr(dc, "set_deterministic") else None
cap = dc.payload_bound(a.size); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
for _ in range(2): n = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
good = pl[:n].cpu().numpy().copy()
h, o, _ = szh_ref.parse(good.tobytes())
assert h["version"] == 5, h["version"]
out = torch.empty_like(t)
regions = {"header": (0, 160), "lens": (o["lens"], o["lens"] + h["sym_count"]), "chunkwords": (o["chunkwords"], o["chunkwords"] + 2 * h["n_chunks"]),
           "subbits": (o["subbits"], o["subbits"] + 2 * h["n_chunks"]), "bits": (o["bitstream"], n)}
names = list(regions)
refused = ok = 0
for k in range(cases):
    bad = good.copy()
    name = names[k % len(names)]
    lo, hi = regions[name]
    for _ in range(int(rng.integers(1, 6))):
        at = int(rng.integers(lo, hi))
        bad[at] = int(rng.integers(0, 256)) if rng.random() < 0.7 else [0, 255, 24, 25, 31][int(rng.integers(0, 5))]
    d = torch.from_numpy(bad).to(dev)
    try:
        dc.decompress(d.data_ptr(), n, out.data_ptr(), 0); torch.cuda.synchronize(); ok += 1
    except sz3_amd.SZ3HipError:
        refused += 1
d = torch.from_numpy(good).to(dev); dc.decompress(d.data_ptr(), n, out.data_ptr(), 0); torch.cuda.synchronize()
assert float((out - t).abs().max()) <= 1e-3 * (1 + 1e-6), "the context no longer decodes the undamaged stream"
print("fuzz v5: %d damaged payloads: %d refused, %d decoded to something; the context still decodes the good stream" % (cases, refused, ok))
