#!/usr/bin/env python3
"""block stream with a WIDE alphabet (the 16384-bin histogram window forms of k_blk_fit / k_blk_rows, taken from a context's second
call on): C4-like field with the hand-over to the plain path switched off; round trip within the bound, payload stable from the
second call on, codes equal to the tile pass's"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, sz3_amd
from fields import field3d
dev = torch.device("cuda:0")
shape = (48, 256, 512); eb = 1e-6
a = field3d(shape, np.float64, sigma=2e-6)
t = torch.from_numpy(a).to(dev)
L = sz3_amd.lib()
res = {}
for name, fl in (("tiles", 1073741824 | 67108864), ("rows", 1073741824)):
    L.sz3hip_debug_flags(fl)
    dc = sz3_amd.DeviceCompressor(a.size, np.float64)
    conf = sz3_amd.Config(*shape); conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, 1; conf.absErrorBound = eb
    cap = dc.payload_bound_conf(conf, worst_case=True)
    outs = []
    for k in range(3):
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        n = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
        import szh_ref, struct
        hb = pl[:n].cpu().numpy().tobytes()
        h, o, sec = szh_ref.parse(hb)
        sd = sec["side"].tobytes()
        print(name, k, "predictor", h["predictor"], "side_bytes", h["side_bytes"], "side hdr", struct.unpack("<IIQQ", sd[:24]) if len(sd) >= 24 else None,
              "nblocks", ((shape[0]+5)//6)*((shape[1]+5)//6)*((shape[2]+5)//6), flush=True)
        dec = torch.empty_like(t); dc.decompress(pl.data_ptr(), n, dec.data_ptr(), 0); torch.cuda.synchronize()
        err = float((dec - t).abs().max())
        outs.append((n, pl[:n].cpu().numpy().tobytes(), err, int(pl[11].item())))
    res[name] = outs
    print(name, [(o[0], o[2] <= eb, o[3]) for o in outs], "payloads equal:", outs[0][1] == outs[1][1] == outs[2][1])
L.sz3hip_debug_flags(0)
print("rows == tiles:", res["rows"][2][1] == res["tiles"][2][1])
