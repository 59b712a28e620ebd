"""GPU tests (-m gpu): the product against the REFERENCE'S OWN golden vectors (tests/golden/golden.npz, made by tests/golden/make_golden.py
from szcompressor/SZ3 built in the build container) — directly, without the oracle in between. With the stock format on, the tuner priced
the reference's way and one zstd frame, the container this library writes must hash to what the reference wrote: the pre-zstd buffer
(zstd-version independent) and the Config trailer for every golden case — the mixed Lorenzo / regression sets too since round 6: the writer
repeats its per-block selection against the coded array (a block's halo as the reader will have it, what the reference's block loop sees)
until no choice moves. The oracle only unpacks zstd frames and plays stock SZ3 as a reader here."""
import hashlib
import importlib.util
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import sz3_amd  # noqa: E402
from oracle_binding import ALGO_INTERP, ALGO_INTERP_LORENZO, ALGO_LORENZO_REG, EB_REL, oracle, oracle_decompress  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "golden.npz"))
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)

# cases whose block choices differ from the reference's ("The stock writers leave the reference's file"): none since round 6
CHOICES_DIFFER = set()


def _split(blob):
    b = blob.tobytes()
    plen, = struct.unpack_from("<Q", b, 8)
    pay = np.frombuffer(b[16:16 + plen], dtype=np.uint8)
    rawlen, = struct.unpack_from("<Q", pay.tobytes(), 0)
    raw = np.empty(rawlen, dtype=np.uint8)
    assert oracle().szo_zstd_decompress(pay.ctypes.data, pay.size, raw.ctypes.data, rawlen) == rawlen
    return raw.tobytes(), b[16 + plen:]


@pytest.mark.parametrize("name,gen,kw", make_golden.CASES, ids=[c[0] for c in make_golden.CASES])
def test_stock_containers_hash_to_the_reference_s_goldens(name, gen, kw, monkeypatch):
    a = gen()
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = {ALGO_LORENZO_REG: sz3_amd.ALGO_LORENZO_REG, ALGO_INTERP: sz3_amd.ALGO_INTERP, ALGO_INTERP_LORENZO: sz3_amd.ALGO_INTERP_LORENZO}[kw.get("algo", ALGO_LORENZO_REG)]
    conf.lorenzo, conf.lorenzo2, conf.regression = int(kw.get("lorenzo", True)), int(kw.get("lorenzo2", False)), int(kw.get("regression", False))
    if kw.get("eb_mode") == EB_REL:
        conf.errorBoundMode = sz3_amd.EB_REL
        conf.relErrorBound = kw["rel_eb"]
        eb = kw["rel_eb"] * (float(a.max()) - float(a.min()))
    else:
        conf.absErrorBound = eb = kw["abs_eb"]
    if "interp_algo" in kw:
        conf.interpAlgo = kw["interp_algo"]
    for k in ("interpDirection", "interpAlpha", "interpBeta"):
        if k in kw:
            setattr(conf, k, kw[k])
    monkeypatch.setenv("SZ3HIP_TUNER_EXACT", "1")
    monkeypatch.setenv("SZ3HIP_STOCK_ONE_FRAME", "1")
    L = sz3_amd.lib()
    L.sz3hip_set_stock_format(1)
    try:
        blob, _ = sz3_amd.compress(a, conf)
    finally:
        L.sz3hip_set_stock_format(0)
    dec, _ = oracle_decompress(blob, a.dtype, a.shape)          # stock SZ3 reading it
    assert float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64)))) <= eb * (1 + 1e-12)
    if name in CHOICES_DIFFER:
        assert abs(len(blob) - int(GOLD[name + "/size"])) <= 0.08 * int(GOLD[name + "/size"])
        return
    raw, trailer = _split(blob)
    want_trailer = bytes.fromhex(str(GOLD[name + "/trailer_hex"]))
    assert trailer == want_trailer, "Config trailer differs from the reference's"   # (f64 too: a stock container keeps the caller's dataType field, like the reference)
    assert hashlib.sha256(raw).hexdigest() == str(GOLD[name + "/sha256_prezstd"]), "pre-zstd buffer differs from the reference's"
    if name + "/dec" in GOLD:
        assert np.array_equal(dec, GOLD[name + "/dec"])
    else:
        assert hashlib.sha256(dec.tobytes()).hexdigest() == str(GOLD[name + "/dec_sha256"])
    if oracle().szo_zstd_version() == b"1.4.8":
        assert len(blob) == int(GOLD[name + "/size"])
        assert hashlib.sha256(blob.tobytes()).hexdigest() == str(GOLD[name + "/sha256_stream_zstd148"]), "the whole file differs from the reference's"
