"""CPU tests (no GPU) of the multi-slab container (SZ_compress_OMP / SZ_decompress_OMP, api/impl/SZImplOMP.hpp:16-186) as
the LIBRARY writes and reads it. Without a device only ALGO_LOSSLESS slabs can be produced (the predictor path has no CPU
implementation) — which is exactly what makes the container format checkable here: byte layout against the python model
in sz3_amd.distributed, interchange with the oracle's restatement of the OMP path and (marker `ref`) with the reference
itself built with OpenMP, truncation / corruption handling."""
import os
import struct

import numpy as np
import pytest

import sz3_amd
from oracle_binding import ALGO_LOSSLESS, have_ref, make_config, oracle, oracle_compress, oracle_decompress, ref_decompress
from sz3_amd import distributed as D


def _data(shape, dtype=np.float32):
    # (a few hundred distinct values: zstd gains, so the reference's per-slab capacity ZSTD_compressBound(raw) — which
    # leaves no room for the 8-byte length in front of an incompressible slab, SZImplOMP.hpp:73 — is enough)
    return np.round(np.random.default_rng(5).normal(size=shape) * 40).astype(dtype)


def _lossless_conf(shape, slabs, monkeypatch):
    monkeypatch.setenv("SZ3HIP_SLABS", str(slabs))
    c = sz3_amd.Config(*shape)
    c.cmprAlgo = sz3_amd.ALGO_LOSSLESS
    c.openmp = 1
    return c


@pytest.mark.parametrize("shape,slabs", [((7, 6, 5), 3), ((5, 40), 5), ((64,), 4), ((3, 4, 5, 6), 2), ((2, 9, 9), 8)])
def test_lossless_container_layout_and_roundtrip(shape, slabs, monkeypatch):
    a = _data(shape)
    conf = _lossless_conf(shape, slabs, monkeypatch)
    blob, _ = sz3_amd.compress(a, conf)
    G = min(slabs, shape[0])
    outer, confs, blobs = D.split_container(blob.tobytes())
    assert len(blobs) == G
    oc = sz3_amd.Config.load(outer)
    assert oc.openmp == 1 and oc.dims == tuple(d for d in shape if d > 1) and oc.cmprAlgo == sz3_amd.ALGO_LOSSLESS
    for g in range(G):
        lo, hi = D.slab_bounds(shape[0], G, g)
        sc = sz3_amd.Config.load(confs[g])
        want = tuple(d for d in (hi - lo,) + tuple(shape[1:]) if d > 1) or (1,)
        assert sc.dims == want and sc.cmprAlgo == sz3_amd.ALGO_LOSSLESS  # setDims drops a slab thickness of 1 (Config.hpp:164-168)
        rawlen, = struct.unpack_from("<Q", blobs[g], 0)
        assert rawlen == (hi - lo) * int(np.prod(shape[1:], dtype=np.int64)) * 4
    dec, c2 = sz3_amd.decompress(blob, np.float32, shape)
    assert np.array_equal(dec, a) and c2.openmp == 1
    # the python model of the layout reproduces the library's bytes
    assert D.assemble_container(confs, blobs, outer) == blob.tobytes()


def test_the_oracle_omp_decoder_reads_our_container(monkeypatch):
    """the oracle's restatement of SZ_decompress_OMP reads the library's container. (The other direction cannot be made
    with lossless slabs: the reference gives every slab ZSTD_compressBound(raw) bytes, SZImplOMP.hpp:73, and its lossless
    stage wants 8 more, Lossless_zstd.hpp:31-34 — SZ_compress_OMP of an ALGO_LOSSLESS config always throws.)"""
    shape = (9, 10, 11)
    a = _data(shape, np.float64)
    blob, _ = sz3_amd.compress(a, _lossless_conf(shape, 3, monkeypatch))
    dec, c = oracle_decompress(blob, np.float64, shape)
    assert np.array_equal(dec, a) and c.openmp == 1
    oracle().szo_set_omp_slabs(4)
    with pytest.raises(RuntimeError, match="not large enough"):
        oracle_compress(a, make_config(shape, algo=ALGO_LOSSLESS, openmp=True))


@pytest.mark.ref
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_the_reference_omp_decoder_reads_our_container(monkeypatch):
    for shape, slabs in (((6, 20, 24), 3), ((5, 40), 5), ((4, 3, 8, 8), 2)):
        a = _data(shape)
        blob, _ = sz3_amd.compress(a, _lossless_conf(shape, slabs, monkeypatch))
        assert np.array_equal(ref_decompress(blob, np.float32, shape), a)  # the reference's own SZ_decompress_OMP


def test_truncated_and_corrupt_containers_are_refused(monkeypatch):
    shape = (8, 16, 16)
    a = _data(shape)
    blob = sz3_amd.compress(a, _lossless_conf(shape, 4, monkeypatch))[0].tobytes()
    body_len, = struct.unpack_from("<Q", blob, 8)
    trailer = blob[16 + body_len:]
    for cut in (3, 4 + 10, 4 + 4 * 40 + 8, body_len // 2):
        bad = blob[:8] + struct.pack("<Q", cut) + blob[16:16 + cut] + trailer
        with pytest.raises(sz3_amd.SZ3HipError):
            sz3_amd.decompress(bad, np.float32, shape)
    b = bytearray(blob)
    struct.pack_into("<i", b, 16, 100000)  # slab count
    with pytest.raises(sz3_amd.SZ3HipError):
        sz3_amd.decompress(bytes(b), np.float32, shape)
    with pytest.raises(sz3_amd.SZ3HipError):  # trailer cut short: Config::load must not run past the buffer
        sz3_amd.decompress(blob[:-5], np.float32, shape)
    with pytest.raises(sz3_amd.SZ3HipError):
        sz3_amd.decompress(blob[:16 + body_len], np.float32, shape)


def test_lossless_stream_of_another_element_type_is_not_refused_by_the_trailer():
    """the reference never sets Config::dataType (api/sz.hpp:43-82): an ALGO_LOSSLESS stream of int32 data written by stock
    SZ3 carries dataType = SZ_FLOAT and must still open as int32 (the length check is the only guard, SZDispatcher.hpp:81-88)"""
    a = np.arange(-500, 500, dtype=np.int32).reshape(10, 100)
    c = sz3_amd.Config(10, 100)
    c.cmprAlgo = sz3_amd.ALGO_LOSSLESS
    blob = bytearray(sz3_amd.compress(a, c)[0].tobytes())
    body_len, = struct.unpack_from("<Q", blob, 8)
    tr = 16 + body_len
    n = blob[tr]
    # the trailer ends ... [bools][dataType][i32 quantbinCnt][i32 blockSize][u8 predDim]
    assert blob[tr + n - 10] == 7
    blob[tr + n - 10] = 0
    dec, _ = sz3_amd.decompress(bytes(blob), np.int32, a.shape)
    assert np.array_equal(dec, a)


def test_bound_covers_the_container_overhead(monkeypatch):
    shape = (64, 3, 3)
    monkeypatch.setenv("SZ3HIP_SLABS", "64")
    c = sz3_amd.Config(*shape)
    c.cmprAlgo = sz3_amd.ALGO_LOSSLESS
    single = sz3_amd.compress_bound(c, np.float32)
    c.openmp = 1
    assert sz3_amd.compress_bound(c, np.float32) >= single + 64 * (8 + 30)
    a = _data(shape)
    blob, _ = sz3_amd.compress(a, c)  # incompressible noise, 64 slabs of 9 values: the worst case for the overhead
    assert np.array_equal(sz3_amd.decompress(blob, np.float32, shape)[0], a)
