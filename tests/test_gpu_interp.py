"""GPU tests (-m gpu) of the interpolation predictor: BIT-EXACT parity with the oracle's restatement of
InterpolationDecomposition (which is byte-identical to the reference build): same quantisation codes at every
element, same unpredictable set, same reconstructed array; and the complete stream round trip."""
import numpy as np
import pytest

import sz3_amd
from fields import field1d, field2d, field3d, field4d
from oracle_binding import ALGO_INTERP, make_config, oracle_compress, oracle_decompress, oracle_interp_codes

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

CASES = [
    ("3d-cubic", lambda: field3d((33, 47, 50)), 1e-3, dict(interpAlgo=1)),
    ("3d-linear", lambda: field3d((33, 47, 50)), 1e-3, dict(interpAlgo=0)),
    ("3d-linear-even-lines", lambda: field3d((34, 66, 36)), 1e-2, dict(interpAlgo=0)),
    ("3d-cubic-dir5-a1b1", lambda: field3d((70, 64, 65)), 1e-4, dict(interpAlgo=1, interpDirection=5, interpAlpha=1.0, interpBeta=1.0)),
    ("3d-cubic-anchor8", lambda: field3d((40, 33, 29)), 1e-2, dict(interpAlgo=1, interpAnchorStride=8, interpAlpha=1.5, interpBeta=3.0)),
    ("3d-noanchor-dir3", lambda: field3d((20, 21, 22)), 1e-3, dict(interpAlgo=1, interpAnchorStride=0, interpDirection=3, interpAlpha=-1.0)),
    ("3d-f64", lambda: field3d((20, 30, 37), np.float64, sigma=2e-6), 1e-6, dict(interpAlgo=1)),
    ("1d-cubic", lambda: field1d(70001), 1e-3, dict(interpAlgo=1)),
    ("1d-linear", lambda: field1d(9000), 1e-2, dict(interpAlgo=0)),
    ("2d-cubic-dir1", lambda: field2d((123, 257)), 1e-3, dict(interpAlgo=1, interpDirection=1)),
    ("2d-linear", lambda: field2d((130, 66)), 1e-3, dict(interpAlgo=0)),
    ("4d-cubic", lambda: field4d((7, 11, 13, 17)), 1e-2, dict(interpAlgo=1)),
    ("4d-linear-dir23", lambda: field4d((5, 20, 33, 40)), 1e-3, dict(interpAlgo=0, interpDirection=23)),
    ("3d-nan", None, 1e-3, dict(interpAlgo=1)),
    # row lengths that are multiples of 8: the 8-wide level-1 kernels (every direction order puts x at a different place)
    ("3d-vec-cubic", lambda: field3d((37, 41, 64)), 1e-3, dict(interpAlgo=1)),
    ("3d-vec-cubic-dir5", lambda: field3d((33, 40, 48)), 1e-4, dict(interpAlgo=1, interpDirection=5)),
    ("3d-vec-cubic-dir2", lambda: field3d((70, 35, 104)), 1e-3, dict(interpAlgo=1, interpDirection=2, interpAlpha=1.5, interpBeta=3.0)),
    ("3d-vec-f64", lambda: field3d((20, 30, 40), np.float64, sigma=2e-6), 1e-6, dict(interpAlgo=1)),
    ("4d-vec-cubic", lambda: field4d((5, 9, 16, 24)), 1e-2, dict(interpAlgo=1)),
    ("4d-vec-cubic-dir17", lambda: field4d((6, 7, 34, 16)), 1e-3, dict(interpAlgo=1, interpDirection=17)),
    ("3d-vec-nan", "nan64", 1e-3, dict(interpAlgo=1)),
    # rows of a multiple of 4 but not of 8 elements: the 8-wide level-1 kernels end every row in a half group
    ("3d-vec-half-100", lambda: field3d((33, 40, 100)), 1e-3, dict(interpAlgo=1)),
    ("3d-vec-half-20-dir5", lambda: field3d((40, 36, 20)), 1e-4, dict(interpAlgo=1, interpDirection=5)),
    ("3d-vec-half-f64-dir2", lambda: field3d((18, 35, 68), np.float64, sigma=2e-6), 1e-6, dict(interpAlgo=1, interpDirection=2)),
    ("4d-vec-half-36", lambda: field4d((5, 9, 17, 36)), 1e-3, dict(interpAlgo=1, interpDirection=11)),
    ("3d-vec-half-500", lambda: field3d((9, 20, 500)), 1e-3, dict(interpAlgo=1, interpAlpha=1.0, interpBeta=1.0)),
    # 1-D / 2-D fields: the 8-wide level-1 kernels with the 1-D / 2-D interface's boundary rules (tails of 2..4 points too)
    ("2d-vec-cubic", lambda: field2d((123, 256)), 1e-3, dict(interpAlgo=1)),
    ("2d-vec-cubic-dir1-half", lambda: field2d((130, 260)), 1e-4, dict(interpAlgo=1, interpDirection=1)),
    ("2d-vec-tails-34x36", lambda: field2d((34, 36)), 1e-3, dict(interpAlgo=1)),
    ("2d-vec-tails-35x68-dir1", lambda: field2d((35, 68)), 1e-3, dict(interpAlgo=1, interpDirection=1)),
    ("2d-vec-tails-37x100", lambda: field2d((37, 100)), 1e-3, dict(interpAlgo=1, interpAlpha=1.5, interpBeta=2.0)),
    ("2d-vec-f64", lambda: field2d((96, 72), np.float64, sigma=2e-6), 1e-6, dict(interpAlgo=1)),
    ("1d-vec-cubic", lambda: field1d(70004), 1e-3, dict(interpAlgo=1)),
    ("1d-vec-cubic-tail3", lambda: field1d(32 * 100 + 3 + 1), 1e-3, dict(interpAlgo=1)),
    # small quantisers: code 0 (unpredictable) lies inside / at the edge of the histogram window around the radius
    ("3d-qbin1024", lambda: field3d((33, 40, 48)), 1e-3, dict(interpAlgo=1, quantbinCnt=1024)),
    ("3d-qbin256", lambda: field3d((33, 40, 48)), 1e-2, dict(interpAlgo=1, quantbinCnt=256)),
    ("3d-qbin2048-linear", lambda: field3d((20, 33, 64)), 1e-3, dict(interpAlgo=0, quantbinCnt=2048)),
    ("4d-qbin1024-noanchor", lambda: field4d((8, 9, 16, 9)), 1e-3, dict(interpAlgo=1, quantbinCnt=1024, interpAnchorStride=0, interpAlpha=-1.0, interpBeta=4.0)),
    ("1d-qbin1024", lambda: field1d(20001), 1e-3, dict(interpAlgo=1, quantbinCnt=1024, interpAnchorStride=0, interpBeta=4.0)),
]


def _device_roundtrip(a, eb, kw):
    dev = torch.device("cuda:0")
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    cap = dc.payload_bound(a.size, worst_case=True)  # (small quantisers leave many points unpredictable)
    payload = torch.empty(cap, dtype=torch.uint8, device=dev)
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_INTERP
    conf.absErrorBound = eb
    for k, v in kw.items():
        setattr(conf, k, v)
    s = torch.cuda.current_stream().cuda_stream
    size = dc.compress(conf, t.data_ptr(), payload.data_ptr(), cap, s)
    codes = dc.debug_codes(a.size)
    out = torch.empty_like(t)
    dc.decompress(payload.data_ptr(), size, out.data_ptr(), s)
    torch.cuda.synchronize()
    return codes, out.cpu().numpy(), size, dc.stats()


@pytest.mark.parametrize("name,gen,eb,kw", CASES, ids=[c[0] for c in CASES])
def test_interp_bit_exact_with_oracle(name, gen, eb, kw):
    if gen is None or gen == "nan64":
        a = field3d((24, 31, 40) if gen is None else (24, 31, 64))
        a[3, 4, 5] = np.nan
        a[10, 2, 7] = np.inf
        a[20, 20, 20] = 1e30
    else:
        a = gen()
    codes, dec, size, st = _device_roundtrip(a, eb, kw)
    okw = dict(abs_eb=eb, interp_algo=kw.get("interpAlgo", 1))
    okw.update({k: v for k, v in kw.items() if k != "interpAlgo"})
    oconf = make_config(a.shape, algo=ALGO_INTERP, **okw)
    ocodes, order, recon, nun = oracle_interp_codes(a, oconf)
    nat = np.zeros(a.size, dtype=np.int64)
    nat[order.astype(np.int64)] = ocodes
    assert np.array_equal(codes.astype(np.int64), nat), "quantisation codes differ from the reference algorithm"
    assert st["n_value_outliers"] == nun
    assert np.array_equal(dec, recon.reshape(a.shape), equal_nan=True), "reconstruction differs from the reference algorithm"
    odec, _ = oracle_decompress(oracle_compress(a, oconf), a.dtype, a.shape)
    assert np.array_equal(dec, odec, equal_nan=True)
    m = np.isfinite(a) & (np.abs(a) < 1e20)
    assert np.max(np.abs(dec[m].astype(np.float64) - a[m].astype(np.float64))) <= eb


LEVEL_CASES = [c for c in CASES if c[0].startswith("3d")] + [
    # shapes chosen for the level kernels: several blocks per axis with ragged last blocks (line lengths 33, 32, even, 2, 1),
    # every direction order, linear mode's extrapolated line ends, anchors on and off, f64
    ("lv-cubic-97x66x130", lambda: field3d((97, 66, 130)), 1e-3, dict(interpAlgo=1)),
    ("lv-cubic-dir5-65x64x96", lambda: field3d((65, 64, 96)), 1e-4, dict(interpAlgo=1, interpDirection=5)),
    ("lv-cubic-dir1-70x99x34", lambda: field3d((70, 99, 34)), 1e-3, dict(interpAlgo=1, interpDirection=1)),
    ("lv-cubic-dir2-34x67x100", lambda: field3d((34, 67, 100)), 1e-3, dict(interpAlgo=1, interpDirection=2, interpAlpha=1.5, interpBeta=3.0)),
    ("lv-cubic-dir3-66x35x68", lambda: field3d((66, 35, 68)), 1e-3, dict(interpAlgo=1, interpDirection=3)),
    ("lv-cubic-dir4-36x98x67", lambda: field3d((36, 98, 67)), 1e-3, dict(interpAlgo=1, interpDirection=4)),
    ("lv-linear-68x66x100", lambda: field3d((68, 66, 100)), 1e-3, dict(interpAlgo=0)),
    ("lv-linear-dir5-36x70x98", lambda: field3d((36, 70, 98)), 1e-2, dict(interpAlgo=0, interpDirection=5)),
    ("lv-linear-dir2-100x34x36", lambda: field3d((100, 34, 36)), 1e-3, dict(interpAlgo=0, interpDirection=2)),
    ("lv-linear-dir1-noanchor-38x40x70", lambda: field3d((38, 40, 70)), 1e-3, dict(interpAlgo=0, interpDirection=1, interpAnchorStride=0)),
    ("lv-noanchor-67x65x66", lambda: field3d((67, 65, 66)), 1e-3, dict(interpAlgo=1, interpAnchorStride=0, interpAlpha=-1.0)),
    ("lv-anchor8-40x73x61", lambda: field3d((40, 73, 61)), 1e-2, dict(interpAlgo=1, interpAnchorStride=8)),
    ("lv-f64-dir5-66x40x97", lambda: field3d((66, 40, 97), np.float64, sigma=2e-6), 1e-6, dict(interpAlgo=1, interpDirection=5)),
    ("lv-f64-linear-35x66x70", lambda: field3d((35, 66, 70), np.float64, sigma=2e-6), 1e-6, dict(interpAlgo=0)),
    ("lv-thin-3x130x131", lambda: field3d((3, 130, 131)), 1e-3, dict(interpAlgo=1)),
    ("lv-thin-dir5-130x2x131", lambda: field3d((130, 2, 131)), 1e-3, dict(interpAlgo=1, interpDirection=5)),
    ("lv-qbin256-66x40x70", lambda: field3d((66, 40, 70)), 1e-2, dict(interpAlgo=1, quantbinCnt=256)),
]


@pytest.mark.parametrize("name,gen,eb,kw", LEVEL_CASES, ids=[c[0] for c in LEVEL_CASES])
def test_level_kernels_bit_exact_with_oracle(name, gen, eb, kw):
    """debug flag 4194304 sends every level of every 3-D array through the level kernels (one launch per level, a block's three
    passes in LDS; normally taken from 256 blocks up): codes, unpredictable set and reconstruction against the oracle."""
    try:
        sz3_amd.lib().sz3hip_debug_flags(4194304)
        test_interp_bit_exact_with_oracle(name, gen, eb, kw)
    finally:
        sz3_amd.lib().sz3hip_debug_flags(0)


@pytest.mark.parametrize("kw", [dict(interpAlgo=1), dict(interpAlgo=1, interpDirection=5), dict(interpAlgo=0)], ids=["cubic", "cubic-dir5", "linear"])
def test_level_kernels_at_their_natural_size(kw):
    """231 x 200 x 193: 8 x 7 x 7 blocks at the finest level (the level kernels' own routing: fine levels in the level kernel, coarse
    ones pass by pass on the same work array), against the oracle; and the same bytes as the per-pass path (flag 128)"""
    a = field3d((231, 200, 193))
    a[100, 50, 60] = np.nan
    a[7, 199, 192] = 1e30
    test_interp_bit_exact_with_oracle("3d-natural", lambda: a.copy(), 1e-3, kw)
    dev = torch.device("cuda:0")
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    cap = dc.payload_bound(a.size)
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_INTERP
    conf.absErrorBound = 1e-3
    for k, v in kw.items():
        setattr(conf, k, v)
    res = []
    try:
        for flag in (0, 128):
            sz3_amd.lib().sz3hip_debug_flags(flag)
            pl = torch.empty(cap, dtype=torch.uint8, device=dev)
            n = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
            torch.cuda.synchronize()
            res.append(pl[:n].clone())
    finally:
        sz3_amd.lib().sz3hip_debug_flags(0)
    assert res[0].numel() == res[1].numel() and torch.equal(res[0], res[1])


def test_interp_host_api_and_ratio():
    a = field3d((96, 96, 96))
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_INTERP
    conf.absErrorBound = 1e-4
    blob, ratio = sz3_amd.compress(a, conf)
    dec, c2 = sz3_amd.decompress(blob, np.float32, a.shape)
    assert c2.cmprAlgo == sz3_amd.ALGO_HIP_INTERP
    oconf = make_config(a.shape, algo=ALGO_INTERP, abs_eb=1e-4)
    oblob = oracle_compress(a, oconf)
    odec, _ = oracle_decompress(oblob, np.float32, a.shape)
    assert np.array_equal(dec, odec)                      # bit-identical decompressed field
    assert ratio >= 0.97 * a.nbytes / len(oblob)          # same codes; only the entropy-stage container differs
    # default algorithm of a fresh Config (ALGO_INTERP_LORENZO) takes the interpolation path with its default parameters
    conf = sz3_amd.Config(*a.shape)
    conf.absErrorBound = 1e-3
    blob, ratio = sz3_amd.compress(a, conf)
    dec, c2 = sz3_amd.decompress(blob, np.float32, a.shape)
    assert c2.cmprAlgo == sz3_amd.ALGO_HIP_INTERP and np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) <= 1e-3


@pytest.mark.parametrize("VEC_SHAPE", [(48, 56, 128), (40, 33, 100), (17, 24, 36), (65, 33, 96), (34, 66, 132), (70, 40, 20), (200, 136), (67, 36), (40004,)])
def test_vector_and_scalar_level1_kernels_agree(VEC_SHAPE):
    """debug flag 128 forces the one-point-per-thread kernels: same payload, byte for byte"""
    a = {1: field1d, 2: field2d, 3: field3d}[len(VEC_SHAPE)](VEC_SHAPE if len(VEC_SHAPE) > 1 else VEC_SHAPE[0])
    a.flat[a.size // 3 + 7] = np.nan
    dev = torch.device("cuda:0")
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    cap = dc.payload_bound(a.size)
    pl = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2)]
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_INTERP
    conf.absErrorBound = 1e-3
    sizes, outs = [], []
    try:
        for k, flag in enumerate((0, 128)):
            sz3_amd.lib().sz3hip_debug_flags(flag)
            sizes.append(dc.compress(conf, t.data_ptr(), pl[k].data_ptr(), cap, 0))
            o = torch.empty_like(t)
            dc.decompress(pl[k].data_ptr(), sizes[-1], o.data_ptr(), 0)
            torch.cuda.synchronize()
            outs.append(o)
    finally:
        sz3_amd.lib().sz3hip_debug_flags(0)
    for k in (1,):
        assert sizes[0] == sizes[k] and torch.equal(pl[0][:sizes[0]], pl[k][:sizes[k]])
        assert bool(((outs[0] == outs[k]) | (outs[0].isnan() & outs[k].isnan())).all())


@pytest.mark.parametrize("dtype,eb", [(np.float32, 1e-6), (np.float64, 1e-6)])
def test_histogram_tail_passes_change_nothing(dtype, eb):
    """tight bound: the interpolation codes spread over the whole alphabet. Debug flag 8192 forces the histogram pass with
    the 16384-bin tier plus the three windowed tail passes (normally a per-context choice from the previous call's counts):
    the payload must be the same bytes as with the plain pass. (Noise of 5000 quantisation steps: a tenth of the codes lie
    beyond +-8192, none beyond the quantiser's range - lists of more than 32768 unpredictable values are not sorted.)"""
    a = field3d((96, 80, 128), dtype, sigma=5e-3)
    dev = torch.device("cuda:0")
    t = torch.from_numpy(a).to(dev)
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_INTERP
    conf.absErrorBound = eb
    res = []
    try:
        for flag in (0, 8192):
            sz3_amd.lib().sz3hip_debug_flags(flag)
            dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
            cap = dc.payload_bound(a.size, worst_case=True)
            pl = torch.empty(cap, dtype=torch.uint8, device=dev)
            size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
            torch.cuda.synchronize()
            res.append(pl[:size].clone())
    finally:
        sz3_amd.lib().sz3hip_debug_flags(0)
    assert res[0].numel() == res[1].numel() and torch.equal(res[0], res[1])
    st = dc.stats()
    assert st["n_value_outliers"] < 32768 and st["max_code_len"] > 12  # (the spread is real: thousands of distinct symbols)


@pytest.mark.parametrize("shape,dtype,algo", [((70, 81, 93), np.float32, 1), ((97, 64, 130), np.float32, 0), ((66, 67, 65), np.float64, 1), ((33, 130, 34), np.float32, 1)])
def test_dense_hand_over_between_the_two_finest_levels_changes_nothing(shape, dtype, algo):
    """round 5: when the levels of stride 2 and 1 both run as level launches, the coarser one leaves its grid in a dense array and the
    finest reads its coarse points from there (no partial-line stores, no strided gather). Same payload, byte for byte, as the
    hand-over in place (debug flag 536870912); level launches forced on these small arrays (flag 4194304); every direction order that
    puts another axis last, linear and cubic, ragged extents, f64"""
    import torch
    dev = torch.device("cuda:0")
    a = field3d(shape, dtype) if dtype == np.float32 else field3d(shape, dtype, sigma=2e-6)
    eb = 1e-3 if dtype == np.float32 else 1e-6
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.from_numpy(a).to(dev)
    for direction in (0, 3, 5):
        conf = sz3_amd.Config(*shape)
        conf.cmprAlgo = sz3_amd.ALGO_INTERP
        conf.absErrorBound = eb
        conf.interpAlgo = algo
        conf.interpDirection = direction
        outs = {}
        for flag in (4194304 | 536870912, 4194304):
            dc = sz3_amd.DeviceCompressor(a.size, a.dtype, device=0)
            dc.set_deterministic(True)
            cap = dc.payload_bound(a.size, worst_case=True)
            d_pl = torch.empty(cap, dtype=torch.uint8, device=dev)
            sz3_amd.lib().sz3hip_debug_flags(flag)
            try:
                size = dc.compress(conf, d_in.data_ptr(), d_pl.data_ptr(), cap, st)
            finally:
                sz3_amd.lib().sz3hip_debug_flags(0)
            torch.cuda.synchronize()
            outs[flag] = d_pl[:size].cpu().numpy().copy()
        assert np.array_equal(outs[4194304], outs[4194304 | 536870912]), "the dense hand-over changed the payload (direction %d)" % direction


def test_a_one_dimensional_array_takes_any_interp_direction():
    """Round 6 (tests/checks/wild_data_sweep.py, leg 3): a 1-D array has one order of dimensions whatever interpDirection says — the reference's OpenMP
    path hands one-row slabs of a 2-D array (a dimension dropped by Config::setDims) to the 1-D interpolation with the caller's value; this
    library used to refuse the value when writing and such a slab as corrupt when reading"""
    a = np.sin(np.arange(50000) / 37.0).astype(np.float32)
    blobs = []
    for d in (0, 1):
        conf = sz3_amd.Config(a.size)
        conf.cmprAlgo = sz3_amd.ALGO_INTERP
        conf.absErrorBound = 1e-3
        conf.interpDirection = d
        blob, _ = sz3_amd.compress(a, conf)
        out, _ = sz3_amd.decompress(blob, a.dtype, a.shape)
        assert float(np.max(np.abs(out.astype(np.float64) - a.astype(np.float64)))) <= 1e-3
        blobs.append(out)
    assert np.array_equal(blobs[0], blobs[1])
