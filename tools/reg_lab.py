"""where the bytes of a block-predictor stream go, next to the oracle's stream of the same configuration"""
import sys, os, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sz3_amd, szh_ref
from fields import field3d, field_c4a
from oracle_binding import make_config, oracle, oracle_compress, oracle_codes

def payload_of(stream):
    b = stream.tobytes(); plen, = struct.unpack_from("<Q", b, 8)
    blob = np.frombuffer(b[16:16 + plen], dtype=np.uint8).copy()
    rawlen, = struct.unpack_from("<Q", blob.tobytes(), 0)
    out = np.empty(rawlen, dtype=np.uint8)
    assert oracle().szo_zstd_decompress(blob.ctypes.data, blob.size, out.ctypes.data, rawlen) == rawlen
    return out.tobytes(), plen

def ent(c):
    h = np.bincount(c.astype(np.int64) - c.min()); h = h[h > 0].astype(np.float64); p = h / h.sum()
    return float(-(p * np.log2(p)).sum())

cases = [("C2@5e-2 R", field3d((96, 96, 96), np.float32), 5e-2, (0, 0, 1)), ("C2@5e-2 L+R", field3d((96, 96, 96), np.float32), 5e-2, (1, 0, 1)),
         ("C4a L+R", field_c4a((160, 160, 160)), 1e-6, (1, 0, 1)), ("C4a R", field_c4a((160, 160, 160)), 1e-6, (0, 0, 1))]
for name, a, eb, (l1, l2, rg) in cases:
    c = sz3_amd.Config(*a.shape); c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG; c.lorenzo, c.lorenzo2, c.regression = l1, l2, rg; c.absErrorBound = eb
    blob, ratio = sz3_amd.compress(a, c)
    pay, zlen = payload_of(blob)
    h, o, sec = szh_ref.parse(pay)
    codes = szh_ref.huffman_decode(h, sec) if a.size < 2e6 else None
    oc = make_config(a.shape, abs_eb=eb, lorenzo=bool(l1), lorenzo2=bool(l2), regression=bool(rg))
    ob, st = oracle_compress(a, oc, stats=True)
    ocodes, _ = oracle_codes(a, oc)
    print(name, "gpu ratio %.2f oracle %.2f | gpu: payload %d zstd %d side %d bitstream %d lens %d vout %d dout %d code-entropy %s | oracle: stream %d raw %d huff %d unpred %d code-entropy %.3f reg-blocks %d/%d"
          % (ratio, a.nbytes / len(ob), len(pay), zlen, h["side_bytes"], 4 * h["bitstream_words"], h["sym_count"], h["n_vout"], h["n_dout"],
             ("%.3f" % ent(codes)) if codes is not None else "-", len(ob), st.raw_bytes, st.huff_bytes, st.n_unpred, ent(ocodes), st.n_regression_blocks, st.n_blocks), flush=True)
