"""GPU tests (-m gpu): a seeded corruption sweep over the streams the host parsers take apart — stock
ALGO_INTERP and ALGO_LORENZO_REG containers (sz3hip_stock_host.cpp: decomposition header, quantizer lists, the reference's Huffman
tree container, HuffmanEncoder.hpp:108-125 / 601-628) and this library's own 4-D block streams. The damage is done INSIDE the
lossless block (the payload is unpacked with libzstd, damaged, packed again: a flipped byte of the zstd frame itself only tests
zstd): bytes flipped, the payload truncated, a stretch copied over another. Every case must end in SZ3HipError or an array of the
right shape — never a hang, a fault or an exception of another kind. Reference-side input checks: api/sz.hpp:122-135."""
import ctypes as C
import ctypes.util
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import sz3_amd  # noqa: E402
from fields import field1d, field2d, field3d, field4d  # noqa: E402
from oracle_binding import ALGO_INTERP, ALGO_LORENZO_REG, make_config, oracle_compress  # noqa: E402


def _zstd():
    for name in ("libzstd.so.1", ctypes.util.find_library("zstd") or "libzstd.so.1", "/usr/lib/x86_64-linux-gnu/libzstd.so.1", "/opt/conda/lib/libzstd.so.1"):
        try:
            z = C.CDLL(name)
            break
        except OSError:
            continue
    else:
        pytest.skip("no libzstd for the test's own unpacking")
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    z.ZSTD_decompress.restype = C.c_size_t
    z.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    z.ZSTD_isError.restype = C.c_uint
    z.ZSTD_isError.argtypes = [C.c_size_t]
    return z


def _split(blob):
    """container (api/sz.hpp:43-82): magic, version, payload length, payload = [raw length, zstd frames], Config trailer"""
    b = bytes(blob)
    payload = struct.unpack_from("<Q", b, 8)[0]
    body, trailer = b[16:16 + payload], b[16 + payload:]
    raw_len = struct.unpack_from("<Q", body, 0)[0]
    return b[:8], body, trailer, raw_len


def _unpack(z, blob):
    head, body, trailer, raw_len = _split(blob)
    raw = C.create_string_buffer(raw_len)
    src = C.create_string_buffer(body[8:], len(body) - 8)
    got = z.ZSTD_decompress(raw, raw_len, src, len(body) - 8)
    assert not z.ZSTD_isError(got) and got == raw_len
    return head, bytearray(raw.raw), trailer


def _pack(z, head, raw, trailer):
    raw = bytes(raw)
    cap = z.ZSTD_compressBound(len(raw))
    dst = C.create_string_buffer(cap)
    src = C.create_string_buffer(raw, len(raw))
    n = z.ZSTD_compress(dst, cap, src, len(raw), 3)
    assert not z.ZSTD_isError(n)
    body = struct.pack("<Q", len(raw)) + dst.raw[:n]
    return head + struct.pack("<Q", len(body)) + body + trailer


def _damage(raw, rng, kind):
    r = bytearray(raw)
    n = len(r)
    if kind == 0:      # flip one to four bytes, biased to the front (the headers, the tree) half of the time
        for _ in range(int(rng.integers(1, 5))):
            at = int(rng.integers(0, min(n, 512))) if rng.random() < 0.5 else int(rng.integers(0, n))
            r[at] ^= int(rng.integers(1, 256))
    elif kind == 1:    # truncate
        r = r[:int(rng.integers(1, n))]
    elif kind == 2:    # a stretch copied over another
        ln = int(rng.integers(1, max(2, n // 8)))
        a, b = int(rng.integers(0, n - ln + 1)), int(rng.integers(0, n - ln + 1))
        r[b:b + ln] = r[a:a + ln]
    else:              # a field set to an extreme value (lengths and counts live in 4- and 8-byte words)
        at = int(rng.integers(0, max(1, min(n, 256) - 8)))
        r[at:at + 8] = struct.pack("<Q", [0, 1, 0xFFFFFFFF, 0x7FFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF][int(rng.integers(0, 5))])
    return r


def _sweep(blob, dtype, shape, cases, seed):
    z = _zstd()
    head, raw, trailer = _unpack(z, blob)
    good, _ = sz3_amd.decompress(_pack(z, head, raw, trailer), dtype, shape)   # the repacked stream itself is fine
    rng = np.random.default_rng(seed)
    refused = 0
    for k in range(cases):
        bad = _pack(z, head, _damage(raw, rng, k % 4), trailer)
        try:
            out, _ = sz3_amd.decompress(bad, dtype, shape)
        except sz3_amd.SZ3HipError:
            refused += 1
            continue
        assert out.shape == good.shape and out.dtype == good.dtype
    # the library is still in working order afterwards
    again, _ = sz3_amd.decompress(bytes(blob), dtype, shape)
    assert np.array_equal(again, good, equal_nan=True)
    return refused


STOCK = [
    ("interp-3d", lambda: field3d((24, 30, 36)), ALGO_INTERP, {}),
    ("interp-1d", lambda: field1d(20000), ALGO_INTERP, {}),
    ("interp-4d", lambda: field4d((5, 9, 10, 12)), ALGO_INTERP, {}),
    ("lorenzo-reg-3d", lambda: field3d((20, 26, 30)), ALGO_LORENZO_REG, dict(lorenzo=True, lorenzo2=True, regression=True)),
    ("lorenzo-reg-2d", lambda: field2d((90, 130)), ALGO_LORENZO_REG, dict(lorenzo=True, regression=True)),
    ("lorenzo-reg-1d", lambda: field1d(9000), ALGO_LORENZO_REG, dict(lorenzo=True, lorenzo2=True)),
    ("regression-3d", lambda: field3d((18, 24, 30)), ALGO_LORENZO_REG, dict(lorenzo=False, regression=True)),
]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,gen,algo,kw", STOCK, ids=[c[0] for c in STOCK])
def test_damaged_stock_streams_end_in_an_error_or_an_array(name, gen, algo, kw):
    a = gen()
    blob = oracle_compress(a, make_config(a.shape, algo=algo, abs_eb=1e-2, **kw)).tobytes()
    refused = _sweep(blob, a.dtype, a.shape, 32, seed=zlib.crc32(name.encode()) & 0xFFFF)
    assert refused > 0   # (truncations at least are always caught)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("shape,mask", [((6, 12, 13, 14), (1, 0, 1)), ((5, 8, 8, 20), (1, 0, 0)), ((20, 26, 30), (1, 1, 1)), ((70, 90), (1, 0, 1)),
                                        ((64, 256, 256), (1, 0, 0))],  # (4 M elements: round 6's sampled book, payload version 5 — code words up to 24 bits, an escape symbol in the header)
                         ids=["4d-blocks", "4d-plain", "3d-blocks", "2d-blocks", "3d-sampled-book-v5"])
def test_damaged_own_streams_end_in_an_error_or_an_array(shape, mask):
    a = field4d(shape) if len(shape) == 4 else (field3d(shape) if len(shape) == 3 else field2d(shape))
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.lorenzo, conf.lorenzo2, conf.regression = mask
    conf.absErrorBound = 1e-3 if a.size >= (1 << 22) else 1e-2
    blob, _ = sz3_amd.compress(a, conf)
    _sweep(bytes(blob), a.dtype, a.shape, 24, seed=sum(shape))
