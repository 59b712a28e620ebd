"""GPU tests (-m gpu) of the slab-parallel path AS THE LIBRARY DOES IT (csrc/sz3hip_host.cpp, csrc/sz3hip_comm.cpp):
conf.openmp = 1 -> slabs along dims[0] (SZ_compress_OMP, api/impl/SZImplOMP.hpp:16-117), the code histograms of all slabs
summed — through a real RCCL communicator (ncclCommInitAll / ncclCommInitRank + ncclAllReduce; one rank on the one-GPU
test box, the 8-GPU run is the driver's) — one code book for every slab, the reference's multi-slab container, and the
library's own decoder (SZ_decompress_OMP, :120-186) reading it back through the Python, C and C++ faces."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import sz3_amd  # noqa: E402
import szh_ref  # noqa: E402
from fields import field3d, field4d  # noqa: E402
from oracle_binding import oracle  # noqa: E402
from sz3_amd import distributed as D  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _unzstd(blob):
    """[u64 rawLen][zstd frames] -> raw bytes (the oracle's libzstd binding; checker only)"""
    L = oracle()
    rawlen, = struct.unpack_from("<Q", blob, 0)
    src = np.frombuffer(blob, dtype=np.uint8).copy()
    out = np.empty(rawlen, dtype=np.uint8)
    assert L.szo_zstd_decompress(src.ctypes.data, src.size, out.ctypes.data, rawlen) == rawlen
    return out.tobytes()


def _payload(stream):
    """single-slab SZ3 stream -> its payload (the blob between the 16-byte header and the Config trailer)"""
    b = stream.tobytes()
    plen, = struct.unpack_from("<Q", b, 8)
    return b[16:16 + plen]


@pytest.fixture
def rccl(monkeypatch):
    monkeypatch.setenv("SZ3HIP_RCCL_SINGLE", "1")  # one GPU here: the sum still goes through a (one-rank) RCCL communicator


@pytest.mark.parametrize("algo", ["lorenzo", "interp", "default"])
@pytest.mark.parametrize("slabs,dtype", [(2, np.float32), (3, np.float64)])
def test_openmp_container_of_gpu_streams_round_trips_with_one_code_book(algo, slabs, dtype, rccl, monkeypatch):
    monkeypatch.setenv("SZ3HIP_SLABS", str(slabs))
    shape = (50, 48, 64)
    a = field3d(shape, dtype)
    # the slabs differ a lot (the second half is 30x rougher): their own code books would differ, the shared one cannot
    a[shape[0] // 2:] += np.random.default_rng(1).normal(0, 0.06, size=a[shape[0] // 2:].shape).astype(dtype)
    eb = 1e-3
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = {"lorenzo": sz3_amd.ALGO_LORENZO_REG, "interp": sz3_amd.ALGO_INTERP, "default": sz3_amd.ALGO_INTERP_LORENZO}[algo]
    conf.regression = 0
    conf.absErrorBound = eb
    conf.openmp = 1
    blob, ratio = sz3_amd.compress(a, conf)
    outer, confs, blobs = D.split_container(blob.tobytes())
    assert len(blobs) == slabs and sz3_amd.Config.load(outer).openmp == 1
    books = []
    for g in range(slabs):
        sc = sz3_amd.Config.load(confs[g])
        lo, hi = D.slab_bounds(shape[0], slabs, g)
        assert sc.dims == (hi - lo,) + shape[1:]
        assert sc.cmprAlgo == (sz3_amd.ALGO_HIP_LORENZO if algo == "lorenzo" else sz3_amd.ALGO_HIP_INTERP)
        h, o, sec = szh_ref.parse(_unzstd(blobs[g]))
        assert h["n"] == (hi - lo) * shape[1] * shape[2]
        books.append((h["sym_min"], h["sym_count"], sec["lens"].tobytes()))
    assert all(b == books[0] for b in books), "slabs were coded with different code books: the histogram exchange did not happen"
    dec, c2 = sz3_amd.decompress(blob, dtype, shape)
    assert c2.openmp == 1
    assert float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64)))) <= eb
    # against single-slab streams of the same slabs: same reconstruction (the code book changes the bits, not the codes)
    conf.openmp = 0
    for g in range(slabs):
        lo, hi = D.slab_bounds(shape[0], slabs, g)
        sconf = sz3_amd.Config(hi - lo, *shape[1:])
        for k in ("cmprAlgo", "regression", "absErrorBound"):
            setattr(sconf, k, getattr(conf, k))
        if algo != "default":  # (the tuner decides per slab either way, but from the same samples)
            sb, _ = sz3_amd.compress(np.ascontiguousarray(a[lo:hi]), sconf)
            assert np.array_equal(sz3_amd.decompress(sb, dtype, (hi - lo,) + shape[1:])[0], dec[lo:hi])


@pytest.mark.parametrize("shape,eb", [((1 << 18,), 1e-3), ((600, 400), 0.15)], ids=["1d", "2d"])
def test_low_dimensional_lorenzo_plus_regression_through_the_slab_path(shape, eb, rccl, monkeypatch):
    """C1's predictor set (and its 2-D form) split into three slabs along dims[0]: every slab a block-composed stream of its own
    dimension count with the SAME code book, the container within the bound, every slab equal to a single-slab call on it"""
    from fields import field1d, field2d
    monkeypatch.setenv("SZ3HIP_SLABS", "3")
    a = field1d(shape[0], np.float32) if len(shape) == 1 else field2d(shape, np.float32)
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG  # (defaults: lorenzo + regression, blockSize 128 / 16)
    conf.absErrorBound = eb
    conf.openmp = 1
    blob, ratio = sz3_amd.compress(a, conf)
    outer, confs, blobs = D.split_container(blob.tobytes())
    assert len(blobs) == 3
    books, reg_blocks = [], 0
    for g in range(3):
        sc = sz3_amd.Config.load(confs[g])
        assert (sc.lorenzo, sc.lorenzo2, sc.regression) == (1, 0, 1)
        h, o, sec = szh_ref.parse(_unzstd(blobs[g]))
        assert h["predictor"] == 2 and h["ndim"] == len(shape) and h["blk_edge"] == (128 if len(shape) == 1 else 16)
        reg_blocks += int((np.asarray(szh_ref.parse_side(h, sec)[0]) == 2).sum())
        books.append((h["sym_min"], h["sym_count"], sec["lens"].tobytes()))
    assert all(b == books[0] for b in books) and reg_blocks > 0
    dec, c2 = sz3_amd.decompress(blob, np.float32, shape)
    assert c2.openmp == 1 and float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64)))) <= eb
    conf.openmp = 0
    for g in range(3):
        lo, hi = D.slab_bounds(shape[0], 3, g)
        sconf = sz3_amd.Config(hi - lo, *shape[1:])
        for k in ("cmprAlgo", "lorenzo", "lorenzo2", "regression", "absErrorBound"):
            setattr(sconf, k, getattr(conf, k))
        sb, _ = sz3_amd.compress(np.ascontiguousarray(a[lo:hi]), sconf)
        assert np.array_equal(sz3_amd.decompress(sb, np.float32, (hi - lo,) + shape[1:])[0], dec[lo:hi])


@pytest.mark.parametrize("slabs", [2, 3])
def test_c4_lorenzo_plus_regression_through_the_slab_path(slabs, rccl, monkeypatch):
    """BASELINE config C4 in small: float64, Lorenzo + regression chosen per block (the reference's default predictor set for
    ALGO_LORENZO_REG), abs 1e-6, split into slabs along dims[0] with the histogram summed over the slabs (one-rank RCCL here).
    Every slab is a block-composed stream (SZH1 predictor 2) coded with the SAME code book; the container decodes within the
    bound; every slab's reconstruction equals what a single-slab call on that slab gives; ratio against the oracle's
    OpenMP-container ratio for the same split."""
    from fields import field_c4a
    from oracle_binding import make_config, oracle_compress
    monkeypatch.setenv("SZ3HIP_SLABS", str(slabs))
    shape = (60, 48, 54)
    a = field_c4a(shape)
    eb = 1e-6
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, 1
    conf.absErrorBound = eb
    conf.openmp = 1
    blob, ratio = sz3_amd.compress(a, conf)
    outer, confs, blobs = D.split_container(blob.tobytes())
    assert len(blobs) == slabs
    books, reg_blocks = [], 0
    for g in range(slabs):
        sc = sz3_amd.Config.load(confs[g])
        assert sc.cmprAlgo == sz3_amd.ALGO_HIP_LORENZO and (sc.lorenzo, sc.lorenzo2, sc.regression) == (1, 0, 1)
        h, o, sec = szh_ref.parse(_unzstd(blobs[g]))
        assert h["predictor"] == 2 and h["blk_edge"] == 6 and h["blk_mask"] == 5
        sel, _ = szh_ref.parse_side(h, sec)
        reg_blocks += int((sel == 2).sum())
        books.append((h["sym_min"], h["sym_count"], sec["lens"].tobytes()))
    assert all(b == books[0] for b in books), "slabs were coded with different code books"
    assert reg_blocks > 0, "no block took regression: the field does not exercise the composed predictor"
    dec, c2 = sz3_amd.decompress(blob, np.float64, shape)
    assert c2.openmp == 1 and float(np.max(np.abs(dec - a))) <= eb
    conf.openmp = 0
    for g in range(slabs):
        lo, hi = D.slab_bounds(shape[0], slabs, g)
        sconf = sz3_amd.Config(hi - lo, *shape[1:])
        for k in ("cmprAlgo", "lorenzo", "lorenzo2", "regression", "absErrorBound"):
            setattr(sconf, k, getattr(conf, k))
        sb, _ = sz3_amd.compress(np.ascontiguousarray(a[lo:hi]), sconf)
        assert np.array_equal(sz3_amd.decompress(sb, np.float64, (hi - lo,) + shape[1:])[0], dec[lo:hi])
    oconf = make_config(shape, abs_eb=eb, lorenzo=True, regression=True, openmp=True)
    o_ratio = a.nbytes / len(oracle_compress(a, oconf))
    print("C4 in small, %d slabs: ratio %.2f (oracle, OpenMP container: %.2f), regression blocks %d" % (slabs, ratio, o_ratio, reg_blocks))
    assert ratio >= 0.9 * o_ratio


def test_shared_code_book_really_is_the_global_one(rccl, monkeypatch):
    """the lens table of a 2-slab container equals the table of ONE stream over the whole array only if the histograms
    were summed: here the two slabs have disjoint alphabets"""
    monkeypatch.setenv("SZ3HIP_SLABS", "2")
    shape = (32, 32, 64)
    a = np.zeros(shape, np.float32)
    rng = np.random.default_rng(3)
    a[:16] = rng.integers(0, 4, size=a[:16].shape) * 0.002   # tiny deltas
    a[16:] = rng.integers(0, 40, size=a[16:].shape) * 0.02    # large ones
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.regression = 0
    conf.absErrorBound = 1e-3
    conf.openmp = 1
    blob, _ = sz3_amd.compress(a, conf)
    _, _, blobs = D.split_container(blob.tobytes())
    h0, _, s0 = szh_ref.parse(_unzstd(blobs[0]))
    h1, _, s1 = szh_ref.parse(_unzstd(blobs[1]))
    assert (h0["sym_min"], h0["sym_count"]) == (h1["sym_min"], h1["sym_count"]) and np.array_equal(s0["lens"], s1["lens"])
    conf.openmp = 0
    own0 = szh_ref.parse(_unzstd(_payload(sz3_amd.compress(np.ascontiguousarray(a[:16]), _like(conf, (16, 32, 64)))[0])))[0]
    assert own0["sym_count"] < h0["sym_count"], "slab 0 by itself needs a much smaller alphabet than the shared book covers"
    dec, _ = sz3_amd.decompress(blob, np.float32, shape)
    assert float(np.max(np.abs(dec.astype(np.float64) - a))) <= 1e-3


def _like(conf, shape):
    c = sz3_amd.Config(*shape)
    for k in ("cmprAlgo", "regression", "absErrorBound", "errorBoundMode", "relErrorBound"):
        setattr(c, k, getattr(conf, k))
    return c


def test_range_based_bound_uses_the_global_range(rccl, monkeypatch):
    """SZImplOMP.hpp:57-69: REL bounds come from the range of the WHOLE array; C5's configuration in small"""
    monkeypatch.setenv("SZ3HIP_SLABS", "4")
    shape = (10, 20, 24, 28)
    a = field4d(shape)
    a[:3] *= 0.01  # a slab of its own range 100x smaller: a per-slab bound would be 100x tighter there
    conf = sz3_amd.Config(*shape)
    conf.errorBoundMode = sz3_amd.EB_REL
    conf.relErrorBound = 1e-3
    conf.openmp = 1
    blob, _ = sz3_amd.compress(a, conf)
    rng = float(a.max()) - float(a.min())
    outer, confs, _ = D.split_container(blob.tobytes())
    ebs = [sz3_amd.Config.load(c).absErrorBound for c in confs]
    assert all(abs(e - 1e-3 * rng) <= 1e-6 * rng for e in ebs), (ebs, rng)
    oc = sz3_amd.Config.load(outer)
    assert oc.errorBoundMode == sz3_amd.EB_ABS and abs(oc.absErrorBound - ebs[0]) == 0  # calAbsErrorBound rewrites conf (:64)
    dec, _ = sz3_amd.decompress(blob, np.float32, shape)
    assert float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64)))) <= ebs[0]


def test_slabs_that_fall_back_to_lossless_and_integer_input(rccl, monkeypatch):
    monkeypatch.setenv("SZ3HIP_SLABS", "3")
    shape = (12, 96, 128)
    a = field3d(shape)
    mask = np.random.default_rng(0).random(a[4:8].shape) < 0.3
    a[4:8][mask] = np.nan  # 30 % unpredictable values: beyond the n/8 the outlier lists grow to -> this slab goes lossless
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.absErrorBound = 1e-3
    conf.openmp = 1
    blob, _ = sz3_amd.compress(a, conf)
    _, confs, _ = D.split_container(blob.tobytes())
    algos = [sz3_amd.Config.load(c).cmprAlgo for c in confs]
    assert algos[1] == sz3_amd.ALGO_LOSSLESS and algos[0] == algos[2] == sz3_amd.ALGO_HIP_LORENZO, algos
    dec, _ = sz3_amd.decompress(blob, np.float32, shape)
    ok = ~np.isnan(a)
    assert np.array_equal(np.isnan(dec), np.isnan(a))
    assert float(np.max(np.abs(dec[ok].astype(np.float64) - a[ok].astype(np.float64)))) <= 1e-3
    ai = (field3d(shape) * 1000).astype(np.int32)
    ci = sz3_amd.Config(*shape)
    ci.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    ci.absErrorBound = 2.5
    ci.openmp = 1
    bi, _ = sz3_amd.compress(ai, ci)
    di, _ = sz3_amd.decompress(bi, np.int32, shape)
    assert int(np.max(np.abs(di.astype(np.int64) - ai))) <= 2


def test_rank_mode_compress_and_container_assembly():
    """one process per GPU, world size 1 on this box: ncclGetUniqueId -> ncclCommInitRank -> the collectives of
    sz3hip_compress_rank -> sz3hip_assemble_container -> the ordinary decoder"""
    comm = sz3_amd.Comm.rank(1, 0, 0, sz3_amd.Comm.unique_id())
    try:
        assert comm.size == 1 and comm.my_rank == 0 and comm.local_size == 1 and comm.device() == 0
        shape = (24, 40, 64)
        a = field3d(shape)
        conf = sz3_amd.Config(*shape)
        conf.errorBoundMode = sz3_amd.EB_REL
        conf.relErrorBound = 1e-3
        blob, sconf = sz3_amd.compress_rank(comm, a, conf)
        assert sconf.errorBoundMode == sz3_amd.EB_ABS and sconf.cmprAlgo in (sz3_amd.ALGO_HIP_INTERP, sz3_amd.ALGO_HIP_LORENZO)
        whole = sz3_amd.assemble_container(conf, a.dtype, [sconf], [blob])
        dec, c2 = sz3_amd.decompress(whole, np.float32, shape)
        assert c2.openmp == 1
        assert float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64)))) <= sconf.absErrorBound
        # the device-resident form bench.py times: stage 1 -> RCCL all-reduce inside the library -> stage 2
        import torch
        t = torch.from_numpy(a).cuda()
        dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
        cap = dc.payload_bound(a.size)
        pl = torch.empty(cap, dtype=torch.uint8, device="cuda")
        c3 = sz3_amd.Config(*shape)
        c3.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
        c3.absErrorBound = 1e-3
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            size = D.SlabCompressor(None, dc, comm=comm).compress(c3, t.data_ptr(), pl.data_ptr(), cap, s.cuda_stream)
            out = torch.empty_like(t)
            dc.decompress(pl.data_ptr(), size, out.data_ptr(), s.cuda_stream)
        s.synchronize()
        assert float((out.double() - t.double()).abs().max().item()) <= 1e-3
        # a rank's later calls: stage 2 packs with the previous call's (global) book while this call's is built from the summed
        # histogram — confirmed on the same array, replaced on another; every payload is what a context without a history and
        # without the exchange produces
        c3.regression = 0
        dc.set_speculation(True, backoff=False)
        dc.set_deterministic(True)  # (the payloads are compared with a fresh context's)
        b = field3d(shape, seed=9)
        tb = torch.from_numpy(b).cuda()
        sc = D.SlabCompressor(None, dc, comm=comm)
        for tt, want in ((t, None), (t, (1, 0)), (tb, (0, 1)), (tb, (1, 0))):
            h0, m0 = dc.spec_stats()
            with torch.cuda.stream(s):
                size = sc.compress(c3, tt.data_ptr(), pl.data_ptr(), cap, s.cuda_stream)
            s.synchronize()
            h1, m1 = dc.spec_stats()
            ref_dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
            pl2 = torch.empty(cap, dtype=torch.uint8, device="cuda")
            n2 = ref_dc.compress(c3, tt.data_ptr(), pl2.data_ptr(), cap, 0)
            torch.cuda.synchronize()
            assert size == n2 and torch.equal(pl[:size], pl2[:n2])
            if want is not None:
                assert (h1 - h0, m1 - m0) == want
    finally:
        comm.close()


@pytest.mark.parametrize("args", [("40", "48", "64", "1e-3", "0", "f"), ("30", "32", "40", "1e-4", "1", "d")])
def test_cxx_face_smoke_test_with_openmp(args, tmp_path, rccl, monkeypatch):
    """tools/sz3/sz3_smoke_test.cpp:10-52 restated against include/SZ3/api/sz.hpp: conf.openmp = true"""
    monkeypatch.setenv("SZ3HIP_SLABS", "2")
    exe = str(tmp_path / "multislab_roundtrip")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(HERE, "cxx", "multislab_roundtrip.cpp"),
                           "-o", exe, "-L" + os.path.join(ROOT, "sz3_amd"), "-lsz3hip", "-Wl,-rpath," + os.path.join(ROOT, "sz3_amd")])
    out = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("ok"), (out.stdout, out.stderr)  # (RCCL prints a banner)


def test_the_callers_device_is_left_alone():
    import torch
    if torch.cuda.device_count() < 1:
        pytest.skip("no device")
    a = field3d((16, 32, 64))
    conf = sz3_amd.Config(*a.shape)
    conf.absErrorBound = 1e-3
    before = torch.cuda.current_device()
    blob, _ = sz3_amd.compress(a, conf)
    sz3_amd.decompress(blob, np.float32, a.shape)
    assert torch.cuda.current_device() == before
