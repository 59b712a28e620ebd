#!/usr/bin/env python3
"""share of regression blocks the composed predictor picks on the bench's C4 slab (128 x 1024 x 1024 f64, 1e-6)"""
import os, sys, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d, field_c4a
dev = torch.device("cuda:0")
shape = (128, 1024, 1024)
for name, a in (("default sigma 2e-6", field3d(shape, np.float64, seed=20260928, sigma=2e-6)), ("c4a", field_c4a(shape, seed=20260928).astype(np.float64))):
    d_in = torch.from_numpy(a).to(dev)
    conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, 1
    conf.absErrorBound = 1e-6
    dc = sz3_amd.DeviceCompressor(a.size, np.float64)
    cap = dc.payload_bound(a.size); pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    size = dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, 0)
    hdr = pl[:160].cpu().numpy().tobytes()
    n_chunks, = struct.unpack_from("<Q", hdr, 72); sym_count, = struct.unpack_from("<I", hdr, 84)
    n_vout, n_dout = struct.unpack_from("<QQ", hdr, 88); side_bytes, = struct.unpack_from("<Q", hdr, 120)
    a16 = lambda x: (x + 15) & ~15
    off = a16(160 + sym_count); off = a16(off + 2 * n_chunks); off = a16(off + 2 * n_chunks)
    off += 8 * n_vout; off = a16(off + 8 * n_vout); off += 8 * n_dout; off = a16(off + 8 * n_dout)
    side = pl[off:off + 24].cpu().numpy().tobytes()
    coding, sel_bits, nb, nr = struct.unpack("<IIQQ", side)
    print("%-20s blocks %d  regression blocks %d (%.5f)  ratio %.3f  side %d B" % (name, nb, nr, nr / max(nb, 1), a.nbytes / size, side_bytes), flush=True)
    del d_in, pl, dc
