#!/usr/bin/env python3
"""the codes of the block stream by rows of blocks (k_blk_rows) against the tile pass (k_blk_lorenzo, debug flag 67108864) on shapes with ragged blocks: must agree everywhere"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, sz3_amd, szh_ref, struct
from fields import field3d, field_c4a
from oracle_binding import oracle
def payload_of(stream):
    b = stream.tobytes(); plen, = struct.unpack_from("<Q", b, 8)
    blob = np.frombuffer(b[16:16 + plen], dtype=np.uint8).copy(); rawlen, = struct.unpack_from("<Q", blob.tobytes(), 0)
    out = np.empty(rawlen, dtype=np.uint8); assert oracle().szo_zstd_decompress(blob.ctypes.data, blob.size, out.ctypes.data, rawlen) == rawlen
    return out.tobytes()
L = sz3_amd.lib()
def run(a, eb, masks, tag):
    c = sz3_amd.Config(*a.shape); c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; c.lorenzo, c.lorenzo2, c.regression = masks; c.absErrorBound = eb
    res = {}
    for name, fl in (("new", 1073741824), ("old", 1073741824 | 67108864)):
        L.sz3hip_debug_flags(fl)
        blob, _ = sz3_amd.compress(a, c)
        h, o, sec = szh_ref.parse(payload_of(blob))
        res[name] = (szh_ref.huffman_decode(h, sec), szh_ref.parse_side(h, sec)[0])
    L.sz3hip_debug_flags(0)
    d = np.nonzero(res["new"][0] != res["old"][0])[0]
    sel = res["new"][1]
    print("%-28s codes differ at %5d of %d; regression blocks %d of %d; first %s" % (tag, len(d), a.size, int((sel == 2).sum()), sel.size, d[:6]))
run(field3d((20,31,45), np.float32, sigma=2e-3), 1e-2, (0,0,1), "R ragged 20x31x45")
run(field3d((18,30,48), np.float32, sigma=2e-3), 1e-2, (0,0,1), "R full 18x30x48")
run(field3d((18,31,48), np.float32, sigma=2e-3), 1e-2, (0,0,1), "R ey=1 18x31x48")
run(field3d((18,30,45), np.float32, sigma=2e-3), 1e-2, (0,0,1), "R ex=3 18x30x45")
run(field3d((20,30,48), np.float32, sigma=2e-3), 1e-2, (0,0,1), "R ez=2 20x30x48")
run(field_c4a((18,30,48), seed=5), 1e-6, (1,0,1), "L1+R c4a 18x30x48")
run(field_c4a((20,31,45), seed=5), 1e-6, (1,0,1), "L1+R c4a ragged")
run(field_c4a((24,36,300), seed=5), 1e-6, (1,0,1), "L1+R c4a 24x36x300")
