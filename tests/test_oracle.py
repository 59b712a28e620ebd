"""CPU tests (-m "not gpu") of the checker itself: the oracle (oracle/sz3_oracle.c, a C restatement of the reference
algorithm) against
  (1) golden vectors generated from the reference itself (tests/golden/golden.npz, made by tests/golden/make_golden.py),
  (2) a stand-in of the reference's data fixture tools/sz3/testfloat_8_8_128.dat (same shape and character) with the CI criterion of
      .github/workflows/cmake.yml:53-65 (ABS 1 => max error <= 1),
  (3) restatements of the reference's unit tests tools/test/modules/test_{encoder,quantizer,lossless}.cpp,
  (4) the reference binary oracle/_ref/libsz3ref.so where it exists (marker `ref`): byte-identical streams.
"""
import hashlib
import os
import struct

import numpy as np
import pytest

from fields import field1d, field2d, field3d, field4d
from oracle_binding import (ALGO_LORENZO_REG, EB_ABS, EB_REL, SzoConfig, have_ref, make_config, oracle, oracle_codes,
                            oracle_compress, oracle_decompress, ref_compress, ref_decompress)
import ctypes as C

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "golden.npz"))
import importlib.util
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)


def _prezstd_sha(blob):
    L = oracle()
    b = blob.tobytes()
    plen, = struct.unpack_from("<Q", b, 8)
    pay = np.frombuffer(b[16:16 + plen], dtype=np.uint8)
    rawlen, = struct.unpack_from("<Q", pay.tobytes(), 0)
    raw = np.empty(rawlen, dtype=np.uint8)
    assert L.szo_zstd_decompress(pay.ctypes.data, pay.size, raw.ctypes.data, rawlen) == rawlen
    return hashlib.sha256(raw.tobytes()).hexdigest(), b[16 + plen:].hex()


@pytest.mark.parametrize("name,gen,kw", make_golden.CASES, ids=[c[0] for c in make_golden.CASES])
def test_oracle_matches_reference_goldens(name, gen, kw):
    a = gen()
    conf = make_golden.case_config(a.shape, kw)
    blob = oracle_compress(a, conf)
    sha, trailer = _prezstd_sha(blob)
    assert sha == str(GOLD[name + "/sha256_prezstd"]), "pre-zstd buffer differs from the reference's"
    assert trailer == str(GOLD[name + "/trailer_hex"]), "Config trailer differs from the reference's"
    if oracle().szo_zstd_version() == b"1.4.8":
        assert len(blob) == int(GOLD[name + "/size"])
        assert hashlib.sha256(blob.tobytes()).hexdigest() == str(GOLD[name + "/sha256_stream_zstd148"])
    dec, conf2 = oracle_decompress(blob, a.dtype, a.shape)
    if name + "/dec" in GOLD:
        assert np.array_equal(dec, GOLD[name + "/dec"]), "decompressed output differs from the reference's"
    else:
        assert hashlib.sha256(dec.tobytes()).hexdigest() == str(GOLD[name + "/dec_sha256"])
    assert abs(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) - float(GOLD[name + "/max_err"])) == 0


def test_reference_ci_fixture():
    """.github/workflows/cmake.yml:53-65: sz3 -f -i testfloat_8_8_128.dat -3 128 8 8 -M ABS 1 -> max error <= 1"""
    from fields import testfloat_like
    a = testfloat_like()  # (an analytic stand-in of the same shape and character: the reference's file is not kept in this repository)
    from oracle_binding import ALGO_INTERP_LORENZO
    conf = make_config(a.shape, algo=ALGO_LORENZO_REG, abs_eb=1.0, regression=True)
    blob = oracle_compress(a, conf)
    dec, _ = oracle_decompress(blob, np.float32, a.shape)
    assert np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) <= 1.0
    assert len(blob) < a.nbytes / 10


def test_restated_encoder_unit_test():
    """tools/test/modules/test_encoder.cpp:11-41: 1000 ints i%100 round-trip through preprocess_encode/save/encode"""
    L = oracle()
    codes = (np.arange(1000) % 100).astype(np.int32)
    buf = np.zeros(1 << 16, dtype=np.uint8)
    n = L.szo_huffman_encode(codes.ctypes.data, codes.size, buf.ctypes.data, buf.size)
    assert 0 < n < 1000 * 4
    out = np.empty_like(codes)
    used = L.szo_huffman_decode(buf.ctypes.data, codes.size, out.ctypes.data)
    assert used == n and np.array_equal(out, codes)
    # single-symbol input: leaf root, zero-length code, empty bit stream (HuffmanEncoder.hpp:233-237)
    const = np.full(77, 5, dtype=np.int32)
    n = L.szo_huffman_encode(const.ctypes.data, const.size, buf.ctypes.data, buf.size)
    out = np.empty_like(const)
    assert L.szo_huffman_decode(buf.ctypes.data, const.size, out.ctypes.data) == n and np.array_equal(out, const)
    assert struct.unpack_from("<Q", buf.tobytes(), n - 8)[0] == 0


def test_restated_quantizer_unit_test():
    """tools/test/modules/test_quantizer.cpp:7-62: LinearQuantizer<float>(eb=12.1973): recover within eb"""
    L = oracle()
    eb, r = 12.1973, 32768
    rng = np.random.default_rng(1)
    for _ in range(200):
        data = np.float32(rng.uniform(-1e5, 1e5))
        pred = np.float32(data + rng.uniform(-2000, 2000))
        d = C.c_float(data)
        code = L.szo_quantize_f32(C.byref(d), float(pred), eb, r)
        assert code != 0
        assert abs(float(d.value) - float(data)) <= eb
        assert np.float32(L.szo_recover_f32(float(pred), code, eb, r)) == np.float32(d.value)
    d = C.c_float(float("nan"))
    assert L.szo_quantize_f32(C.byref(d), 0.0, eb, r) == 0          # NaN is unpredictable
    d = C.c_float(1e30)
    assert L.szo_quantize_f32(C.byref(d), 0.0, eb, r) == 0          # outside the 65536 bins


def test_restated_lossless_unit_test():
    """tools/test/modules/test_lossless.cpp:9-30: zstd round trip of 1000 random bytes"""
    L = oracle()
    src = np.random.default_rng(3).integers(0, 256, 1000, dtype=np.uint8)
    cap = L.szo_zstd_bound(src.size) + 8
    dst = np.empty(cap, dtype=np.uint8)
    n = L.szo_zstd_compress(src.ctypes.data, src.size, dst.ctypes.data, cap)
    assert n > 8 and struct.unpack_from("<Q", dst.tobytes(), 0)[0] == 1000
    back = np.empty(1000, dtype=np.uint8)
    assert L.szo_zstd_decompress(dst.ctypes.data, n, back.ctypes.data, 1000) == 1000 and np.array_equal(back, src)


def test_pysz_style_roundtrips():
    """tools/pysz/tests/test_pysz.py:24-72 (f32 100x100 eb 1e-2, f64 50x50 eb 1e-6, f32 20x30x40 REL 1e-3)"""
    rng = np.random.default_rng(0)
    for a, kw in [(rng.random((100, 100), dtype=np.float32), dict(abs_eb=1e-2)),
                  (rng.random((50, 50)), dict(abs_eb=1e-6)),
                  (rng.random((20, 30, 40), dtype=np.float32), dict(eb_mode=EB_REL, rel_eb=1e-3))]:
        conf = make_config(a.shape, algo=ALGO_LORENZO_REG, regression=True, **kw)
        dec, c2 = oracle_decompress(oracle_compress(a, conf), a.dtype, a.shape)
        bound = c2.absErrorBound
        if kw.get("eb_mode") == EB_REL:
            assert abs(bound - 1e-3 * float(a.max() - a.min())) < 1e-12
        assert np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) <= bound


def test_edge_cases():
    # eb == 0 -> lossless (SZDispatcher.hpp:19-21), NaN/Inf kept, tiny arrays, dims of 1 dropped (Config.hpp:164-168)
    a = field3d((9, 10, 11))
    dec, c2 = oracle_decompress(oracle_compress(a, make_config(a.shape, abs_eb=0.0)), np.float32, a.shape)
    assert c2.cmprAlgo == 4 and np.array_equal(dec, a)
    b = a.copy()
    b[1, 2, 3] = np.nan
    b[4, 5, 6] = np.inf
    dec, _ = oracle_decompress(oracle_compress(b, make_config(b.shape, abs_eb=1e-3)), np.float32, b.shape)
    assert np.isnan(dec[1, 2, 3]) and dec[4, 5, 6] == np.inf
    m = np.isfinite(b)
    assert np.max(np.abs(dec[m].astype(np.float64) - b[m].astype(np.float64))) <= 1e-3
    c = make_config((5, 1, 7))
    assert c.N == 2 and [c.dims[i] for i in range(2)] == [5, 7] and c.blockSize == 16
    one = np.array([1.5], dtype=np.float32)
    dec, _ = oracle_decompress(oracle_compress(one, make_config((1,), abs_eb=1e-3)), np.float32, (1,))
    assert abs(float(dec[0]) - 1.5) <= 1e-3


@pytest.mark.ref
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference: make -C oracle ref)")
@pytest.mark.parametrize("case", [
    ("3d", lambda: field3d((33, 47, 50)), dict(abs_eb=1e-2, regression=True)),
    ("3d-l2", lambda: field3d((20, 21, 22)), dict(abs_eb=1e-3, lorenzo2=True, regression=True)),
    ("3d-f64", lambda: field3d((20, 30, 37), np.float64, sigma=2e-6), dict(abs_eb=1e-4, regression=True)),
    ("1d", lambda: field1d(70000), dict(abs_eb=1e-3, regression=True)),
    ("2d", lambda: field2d((123, 257)), dict(abs_eb=1e-1, regression=True)),
    ("4d-rel", lambda: field4d((7, 11, 13, 17)), dict(eb_mode=EB_REL, rel_eb=1e-3, regression=True)),
    ("tiny-eb", lambda: field3d((12, 13, 14)), dict(abs_eb=1e-9)),
], ids=lambda c: c[0])
def test_oracle_byte_identical_to_reference_build(case):
    _, gen, kw = case
    a = gen()
    conf = make_config(a.shape, algo=ALGO_LORENZO_REG, **kw)
    r = ref_compress(a, conf)
    o = oracle_compress(a, conf)
    assert r.tobytes() == o.tobytes()
    assert np.array_equal(oracle_decompress(r, a.dtype, a.shape)[0], ref_decompress(r, a.dtype, a.shape), equal_nan=True)
