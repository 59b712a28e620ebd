"""CPU tests of stock-stream interoperability, the part that needs no device: the emission rank of every element
of an ALGO_INTERP stream (sz3_amd/csrc/sz3hip_stock_geom.h) against the order in which the oracle's restatement of
InterpolationDecomposition::compress (decomposition/InterpolationDecomposition.hpp:79-147) visits the points."""
import ctypes as C

import numpy as np
import pytest

import sz3_amd
import struct

from fields import field1d, field2d, field3d, field4d
from oracle_binding import ALGO_INTERP, make_config, oracle, oracle_compress, oracle_interp_codes

CASES = [
    # (shape, interp algo, direction, anchor stride)
    ((33, 47, 50), 1, 0, 32), ((33, 47, 50), 0, 0, 32), ((34, 66, 36), 0, 0, 32), ((70, 64, 65), 1, 5, 32), ((40, 33, 29), 1, 0, 8),
    ((20, 21, 22), 1, 3, 0), ((65, 31, 2), 1, 2, 16), ((7, 9, 130), 0, 4, 4), ((64, 64, 64), 1, 1, 32), ((5, 6, 7), 1, 0, 32),
    ((100,), 1, 0, 4096), ((5000,), 1, 0, 128), ((4097,), 0, 0, 1024), ((9,), 1, 0, 4), ((300, 500), 1, 0, 128), ((300, 500), 0, 1, 128),
    ((129, 131), 1, 1, 16), ((40, 40), 1, 0, 0), ((12, 20, 20, 20), 1, 0, 16), ((5, 17, 9, 33), 0, 23, 8), ((6, 33, 18, 35), 1, 7, 16),
    ((1, 1, 40, 50), 1, 0, 16),
]


@pytest.mark.parametrize("shape,algo,direction,anchor", CASES, ids=["%s-%s-d%d-a%d" % ("x".join(map(str, c[0])), "cubic" if c[1] else "linear", c[2], c[3]) for c in CASES])
def test_emission_rank_of_every_element(shape, algo, direction, anchor):
    L = sz3_amd.lib()
    L.sz3hip_debug_stock_ranks.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_uint64, C.c_void_p]
    dims = [d for d in shape if d > 1] or [1]  # (Config::setDims drops extents of 1)
    a = np.random.default_rng(1).standard_normal(dims).astype(np.float32)
    oconf = make_config(a.shape, algo=ALGO_INTERP, abs_eb=1e-2, interp_algo=algo, interpDirection=direction, interpAnchorStride=anchor)
    _, order, _, _ = oracle_interp_codes(a, oconf)   # element index of every code, in emission order
    ranks = np.empty(a.size, dtype=np.uint64)
    arr = (C.c_uint64 * len(dims))(*dims)
    rc = L.sz3hip_debug_stock_ranks(len(dims), arr, algo, direction, anchor, ranks.ctypes.data)
    assert rc == 0, rc
    want = np.empty(a.size, dtype=np.uint64)
    want[order.astype(np.int64)] = np.arange(a.size, dtype=np.uint64)
    bad = np.nonzero(ranks != want)[0]
    assert bad.size == 0, (bad[:10], ranks[bad[:10]], want[bad[:10]])


TRIALS = [
    # (sample block, dtype, interp algo, direction, anchor stride, alpha, beta, bound)
    ((33, 33, 33), np.float32, 1, 0, 32, 1.25, 2.0, 1e-3), ((33, 33, 33), np.float32, 0, 5, 32, 1.25, 2.0, 1e-2), ((33, 33, 33), np.float32, 1, 0, 32, 2.0, 3.0, 1e-1),
    ((17, 17, 17), np.float64, 1, 5, 32, 1.0, 1.0, 1e-3), ((129, 129), np.float32, 1, 1, 128, 1.5, 2.5, 1e-3), ((9, 9, 9, 9), np.float32, 0, 23, 16, 1.25, 2.0, 1e-3),
    ((2049,), np.float32, 1, 0, 4096, 1.25, 2.0, 1e-4), ((33, 33, 33), np.float32, 1, 0, 32, 1.25, 2.0, 10.0),
]


@pytest.mark.parametrize("shape,dtype,algo,direction,anchor,alpha,beta,eb", TRIALS, ids=["%s-%s-d%d-%g" % ("x".join(map(str, c[0])), "cubic" if c[2] else "linear", c[3], c[7]) for c in TRIALS])
def test_a_tuner_trial_s_buffer_is_the_reference_s(shape, dtype, algo, direction, anchor, alpha, beta, eb):
    """the exact pricing of the ALGO_INTERP_LORENZO tuner (sz3hip_ctx_set_tuner_exact) without a device: from the per-element codes of one
    sample block — here the oracle's — stock::trial_buffer must make, byte for byte, the buffer the reference hands to zstd for that block
    (interp_compress_test, api/impl/SZAlgoInterp.hpp:42-78 = an ALGO_INTERP stream's body over the block): emission order, the
    quantizer's list, the tree as the reference's own queue shapes it (ties!), the bits. The last case: one symbol, no bits."""
    L = sz3_amd.lib()
    L.sz3hip_debug_trial_buffer.restype = C.c_int64
    L.sz3hip_debug_trial_buffer.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    a = {1: lambda: field1d(shape[0], dtype), 2: lambda: field2d(shape, dtype), 3: lambda: field3d(shape, dtype), 4: lambda: field4d(shape, dtype)}[len(shape)]()
    oconf = make_config(a.shape, algo=ALGO_INTERP, abs_eb=eb, interp_algo=algo, interpDirection=direction, interpAnchorStride=anchor, interpAlpha=alpha,
                        interpBeta=beta)
    codes, order, _, _ = oracle_interp_codes(a, oconf)
    per_elem = np.empty(a.size, dtype=np.uint16)
    per_elem[order.astype(np.int64)] = codes.astype(np.uint16)
    b = oracle_compress(a, oconf).tobytes()
    plen, = struct.unpack_from("<Q", b, 8)
    pay = np.frombuffer(b[16:16 + plen], dtype=np.uint8)
    rawlen, = struct.unpack_from("<Q", pay.tobytes(), 0)
    want = np.empty(rawlen, dtype=np.uint8)
    assert oracle().szo_zstd_decompress(pay.ctypes.data, pay.size, want.ctypes.data, rawlen) == rawlen
    got = np.empty(rawlen + 4096, dtype=np.uint8)
    arr = (C.c_uint64 * len(shape))(*shape)
    n = L.sz3hip_debug_trial_buffer(len(shape), arr, algo, direction, anchor, alpha, beta, eb, 32768, 0 if dtype == np.float32 else 1, per_elem.ctypes.data,
                                    a.ctypes.data, 1, got.ctypes.data, got.size)
    assert n == rawlen, (n, rawlen)
    assert got[:n].tobytes() == want.tobytes()
