"""GPU tests (-m gpu) of the PIPELINED host path (csrc/sz3hip_host.cpp: piece_count, compress_pieces, the one-GPU branch of
decompress_slabs): a large array of a plain call — not conf.openmp — is cut along dims[0] into independently coded pieces on one GPU
(copy in of piece k + 1 beside the kernels of piece k beside the copy out + zstd of piece k - 1), the container is the reference's
multi-slab one (SZ_compress_OMP's, api/impl/SZImplOMP.hpp:100-110, read back by SZ_decompress_OMP's layout, :120-186). What must
hold: every piece is exactly the stream a plain call writes for that slab, the bound, the bytes from call to call, refusals and
damage handled without a hang (the pieces' turns are passed on whatever happens to a piece)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import sz3_amd  # noqa: E402
from fields import field1d, field2d, field3d, field4d  # noqa: E402
from sz3_amd import distributed as D  # noqa: E402


def _conf(shape, eb, algo=None, **kw):
    c = sz3_amd.Config(*shape)
    c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG if algo is None else algo
    c.absErrorBound = eb
    for k, v in kw.items():
        setattr(c, k, v)
    return c


@pytest.mark.parametrize("pieces,shape,dtype,kw", [
    (3, (96, 40, 48), np.float32, dict(regression=0)),
    (2, (70, 33, 50), np.float64, dict(regression=0)),
    (3, (100, 24, 40), np.float32, dict()),                      # Lorenzo + regression per block: the block-composed stream
    (2, (64, 64), np.float32, dict(regression=0)),               # 2-D: rows are the planes
    (4, (1 << 18,), np.float32, dict(regression=0)),             # 1-D
    (2, (66, 10, 12, 14), np.float32, dict(regression=0)),       # 4-D
])
def test_pieces_are_the_plain_streams_of_their_slabs(pieces, shape, dtype, kw, monkeypatch):
    a = {1: lambda: field1d(shape[0], dtype), 2: lambda: field2d(shape, dtype), 3: lambda: field3d(shape, dtype), 4: lambda: field4d(shape, dtype)}[len(shape)]()
    eb = 1e-3
    monkeypatch.setenv("SZ3HIP_PIECES", str(pieces))
    conf = _conf(shape, eb, **kw)
    blob, ratio = sz3_amd.compress(a, conf)
    assert blob.size <= sz3_amd.compress_bound(conf, dtype)
    outer, confs, blobs = D.split_container(blob.tobytes())
    assert len(blobs) == pieces and sz3_amd.Config.load(outer).openmp == 1
    dec, c2 = sz3_amd.decompress(blob, dtype, shape)
    assert c2.openmp == 1 and c2.dims == tuple(shape)
    assert float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64)))) <= eb
    blob_again, _ = sz3_amd.compress(a, conf)
    assert np.array_equal(blob, blob_again), "the same call wrote other bytes"
    # piece by piece against a plain call on the slab (no pieces): the same payload, the same values
    monkeypatch.setenv("SZ3HIP_PIECES", "0")
    for g in range(pieces):
        lo, hi = D.slab_bounds(shape[0], pieces, g)
        sc = sz3_amd.Config.load(confs[g])
        assert sc.dims == (hi - lo,) + tuple(shape[1:]) and sc.openmp == 0
        one, _ = sz3_amd.compress(np.ascontiguousarray(a[lo:hi]), _conf((hi - lo,) + tuple(shape[1:]), eb, **kw))
        b = one.tobytes()
        plen = int.from_bytes(b[8:16], "little")
        assert b[16:16 + plen] == blobs[g], "piece %d is not the plain call's stream of its slab" % g
        d1, _ = sz3_amd.decompress(one, dtype, (hi - lo,) + tuple(shape[1:]))
        assert np.array_equal(d1, dec[lo:hi])


def test_calls_that_do_not_qualify_stay_whole(monkeypatch):
    shape = (96, 40, 48)
    a = field3d(shape)
    monkeypatch.setenv("SZ3HIP_PIECES", "3")
    c = sz3_amd.Config(*shape)
    c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    c.errorBoundMode = sz3_amd.EB_REL
    c.relErrorBound = 1e-3
    blob, _ = sz3_amd.compress(a, c)
    assert sz3_amd.decompress(blob, np.float32, shape)[1].openmp == 0   # the range of the whole array comes first
    blob, _ = sz3_amd.compress(a, _conf(shape, 1e-3, algo=sz3_amd.ALGO_INTERP_LORENZO))
    assert sz3_amd.decompress(blob, np.float32, shape)[1].openmp == 0   # the interpolation predictor spans the array
    monkeypatch.setenv("SZ3HIP_PIECES_ALL", "1")
    blob, _ = sz3_amd.compress(a, _conf(shape, 1e-3, algo=sz3_amd.ALGO_INTERP_LORENZO))
    dec, c2 = sz3_amd.decompress(blob, np.float32, shape)
    assert c2.openmp == 1 and float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64)))) <= 1e-3
    monkeypatch.delenv("SZ3HIP_PIECES_ALL")
    small = field3d((40, 40, 48))
    blob, _ = sz3_amd.compress(small, _conf(small.shape, 1e-3))          # fewer than two pieces of 32 planes
    assert sz3_amd.decompress(blob, np.float32, small.shape)[1].openmp == 0
    monkeypatch.delenv("SZ3HIP_PIECES")
    blob, _ = sz3_amd.compress(a, _conf(shape, 1e-3))                    # default policy: far below the smallest piece
    assert sz3_amd.decompress(blob, np.float32, shape)[1].openmp == 0


def test_the_pipelined_decoder_at_its_size_and_with_damaged_pieces(monkeypatch):
    """256^3 f32 = 64 MiB: decompress_slabs takes its one-GPU pipeline (a thread per piece, copies out in turn behind the population of
    the output's pages); integer elements ride the same path; a damaged piece is an error, never a hang, and the next call works"""
    shape = (256, 256, 256)
    a = field3d(shape)
    eb = 1e-3
    monkeypatch.setenv("SZ3HIP_PIECES", "4")
    conf = _conf(shape, eb, regression=0)
    blob, ratio = sz3_amd.compress(a, conf)
    outer, confs, blobs = D.split_container(blob.tobytes())
    assert len(blobs) == 4
    dec, c2 = sz3_amd.decompress(blob, np.float32, shape)                     # a fresh array
    assert float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64)))) <= eb
    keep = np.zeros(a.size, dtype=np.float32)
    dec2, _ = sz3_amd.decompress(blob, np.float32, shape, out=keep)           # one the caller keeps
    assert np.array_equal(dec2, dec)
    monkeypatch.setenv("SZ3HIP_PIECES", "0")                                   # the same container through the piece-after-piece reader
    dec3, _ = sz3_amd.decompress(blob, np.float32, shape)
    assert np.array_equal(dec3, dec)
    whole, ratio_whole = sz3_amd.compress(a, conf)
    assert ratio > 0.99 * ratio_whole, (ratio, ratio_whole)                   # four first planes predicted in 2-D: a fraction of a percent
    monkeypatch.setenv("SZ3HIP_PIECES", "4")
    # damage in piece 1's blob: its first zstd frame's magic number (a certain error), then bytes in the middle (an error or an array)
    b = bytearray(blob.tobytes())
    head = len(b) - len(outer) - sum(len(x) for x in blobs)
    at = head + len(blobs[0]) + 8
    for k in range(4):
        b[at + k] ^= 0xFF
    with pytest.raises(sz3_amd.SZ3HipError):
        sz3_amd.decompress(np.frombuffer(bytes(b), dtype=np.uint8), np.float32, shape)
    b = bytearray(blob.tobytes())
    at = head + len(blobs[0]) + len(blobs[1]) // 2
    for k in range(64):
        b[at + k] ^= 0xFF
    try:
        dd, _ = sz3_amd.decompress(np.frombuffer(bytes(b), dtype=np.uint8), np.float32, shape)
        assert dd.shape == shape
    except sz3_amd.SZ3HipError:
        pass
    dec4, _ = sz3_amd.decompress(blob, np.float32, shape)
    assert np.array_equal(dec4, dec)
    ai = np.rint(a * 1000).astype(np.int32)
    ci = _conf(shape, 2.0, regression=0)
    bi, _ = sz3_amd.compress(ai, ci)
    di, c5 = sz3_amd.decompress(bi, np.int32, shape)
    assert c5.openmp == 1 and int(np.max(np.abs(di.astype(np.int64) - ai.astype(np.int64)))) <= 2


def test_pipelined_calls_beside_small_calls_from_other_threads(monkeypatch):
    """a pipelined call takes the host API exclusively, small calls share it: four threads compressing and decompressing small arrays
    while the main thread sends a large one through the pieces twice — nobody waits forever, everybody gets the right values (the pool of
    host threads and the staging ring are shared by all of them)"""
    import threading
    monkeypatch.setenv("SZ3HIP_PIECES", "4")
    big = field3d((256, 256, 256))
    small = [field3d((40 + 8 * k, 48, 56), seed=100 + k) for k in range(4)]
    errs = []

    def worker(k):
        try:
            for _ in range(6):
                a = small[k]
                c = sz3_amd.Config(*a.shape)
                c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
                c.regression = 0
                c.absErrorBound = 1e-3
                b, _ = sz3_amd.compress(a, c)
                d, _ = sz3_amd.decompress(b, np.float32, a.shape)
                if float(np.max(np.abs(d.astype(np.float64) - a.astype(np.float64)))) > 1e-3:
                    errs.append("bound %d" % k)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    conf = _conf(big.shape, 1e-3, regression=0)
    for _ in range(2):
        blob, _ = sz3_amd.compress(big, conf)
        dec, c2 = sz3_amd.decompress(blob, np.float32, big.shape)
        assert c2.openmp == 1 and float(np.max(np.abs(dec.astype(np.float64) - big.astype(np.float64)))) <= 1e-3
    for t in th:
        t.join(120)
        assert not t.is_alive(), "a small call never came back"
    assert not errs, errs
