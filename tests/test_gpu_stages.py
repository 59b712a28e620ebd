"""GPU stage-level tests (-m gpu): every stage of the HIP path against the numpy model of the SZH1 format
(tests/szh_ref.py), through the C ABI with device pointers."""
import numpy as np
import pytest

import sz3_amd
from fields import field1d, field2d, field3d, field4d
import szh_ref

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _conf(shape, eb):
    c = sz3_amd.Config(*shape)
    c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG   # the Lorenzo path (the default ALGO_INTERP_LORENZO takes interpolation)
    c.regression = 0  # Lorenzo-1 alone: the plain stream (with regression the block-composed predictor takes over)
    c.errorBoundMode = sz3_amd.EB_ABS
    c.absErrorBound = eb
    return c


def _roundtrip_device(a, eb):
    dev = torch.device("cuda:0")
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    cap = dc.payload_bound(a.size)
    payload = torch.empty(cap, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    size = dc.compress(_conf(a.shape, eb), t.data_ptr(), payload.data_ptr(), cap, stream)
    codes = dc.debug_codes(a.size)
    pl = payload[:size].cpu().numpy()
    out = torch.empty_like(t)
    dc.decompress(payload.data_ptr(), size, out.data_ptr(), stream)
    torch.cuda.synchronize()
    return codes, pl, out.cpu().numpy(), dc.stats()


CASES = [
    ("1d", lambda: field1d(70001), 1e-3),
    ("1d-long", lambda: field1d(300000), 1e-4),
    ("2d", lambda: field2d((123, 257)), 1e-3),
    ("3d", lambda: field3d((33, 47, 50)), 1e-3),
    ("3d-f64", lambda: field3d((20, 30, 37), np.float64, sigma=2e-6), 1e-6),
    ("4d", lambda: field4d((7, 11, 13, 17)), 1e-2),
    ("3d-fast", lambda: field3d((33, 47, 52)), 1e-3),
    ("3d-fast-exact-tiles", lambda: field3d((16, 24, 128)), 1e-4),
    ("3d-fast-f64", lambda: field3d((21, 30, 36), np.float64, sigma=2e-6), 1e-6),
    ("4d-fast", lambda: field4d((5, 11, 13, 16)), 1e-2),
    ("4d-fast-f64", lambda: field4d((3, 6, 10, 68), np.float64), 1e-3),
    ("3d-march", lambda: field3d((19, 13, 132)), 1e-3),
    ("3d-march-wide", lambda: field3d((35, 9, 520)), 1e-4),
    ("3d-march-f64", lambda: field3d((18, 7, 260), np.float64, sigma=2e-6), 1e-6),
    ("4d-march", lambda: field4d((3, 5, 7, 128)), 1e-2),
    ("4d-march-f64", lambda: field4d((2, 18, 5, 264), np.float64), 1e-3),
    ("3d-smooth", lambda: field3d((40, 40, 40), sigma=0.0), 1e-1),
    ("3d-const", lambda: np.full((17, 19, 23), 3.25, np.float32), 1e-3),
    # wide alphabets: ~1.5k symbols (LDS code book, 24-bit limit), ~6k (round-parallel merge, LDS queues),
    # ~40k symbols with deltas beyond the radius (global-memory queues, delta outliers)
    ("3d-rough-1k", lambda: _rough((40, 64, 64), 0.12), 1e-3),
    ("3d-rough-6k", lambda: _rough((48, 64, 64), 0.6), 1e-3),
    ("3d-rough-40k", lambda: _rough((48, 64, 128), 6.0), 1e-3),
]


def _rough(shape, sigma):
    rng = np.random.default_rng(7)
    return (field3d(shape) + rng.normal(0.0, sigma, shape)).astype(np.float32)


@pytest.mark.parametrize("name,gen,eb", CASES, ids=[c[0] for c in CASES])
def test_stages(name, gen, eb):
    a = gen()
    codes, pl, dec, st = _roundtrip_device(a, eb)
    q, d, exp_codes, bad, dout = szh_ref.dualquant(a, eb, narrow=bool(st["narrow_codes"]))
    # K1: codes bit-exact against the numpy model
    assert np.array_equal(codes, exp_codes.reshape(-1)), "quantisation codes differ from the model"
    h, o, sec = szh_ref.parse(pl)
    assert h["magic"] == szh_ref.MAGIC and h["n"] == a.size and h["payload_bytes"] == len(pl)
    assert h["n_vout"] == int(bad.sum()) and h["n_dout"] == int(dout.sum())
    assert st["n_value_outliers"] == h["n_vout"] and st["payload_bytes"] == len(pl)
    # K5: a complete prefix code over exactly the symbols that occur — small alphabets: plus margins of 8 symbols either side,
    # every empty bin of the widened range counted once (szh_ref.book_symbols; sz3hip_kernels.hip, cb_margins)
    present = np.unique(exp_codes)
    lens = sec["lens"]
    book, filled = szh_ref.book_symbols(present)
    assert set(h["sym_min"] + np.nonzero(lens)[0]) == (book if len(present) > 1 else set())
    if len(present) > 1:
        limit = szh_ref.SHORT_LEN if len(book) <= szh_ref.SHORT_SYMS else szh_ref.MAX_LEN
        assert abs(szh_ref.kraft(lens) - 1.0) < 1e-9 and lens.max() == h["max_len"] <= limit
        # optimality: total bits equal to an independent (unlimited) Huffman construction when that code respects the
        # length limit (16 bits up to 512 symbols, else 24), otherwise within 0.2 % of it (length-limited code)
        freq = np.bincount(exp_codes.reshape(-1), minlength=65536)[h["sym_min"]:h["sym_min"] + h["sym_count"]]
        if filled:
            assert h["sym_min"] == min(book) and h["sym_count"] == len(book)
            freq = np.maximum(freq, 1)
        import heapq
        heap = [(int(f), i, 0) for i, f in enumerate(freq) if f]   # (freq, tiebreak, height)
        heapq.heapify(heap)
        cost = 0
        cnt = len(freq)
        while len(heap) > 1:
            f1, _, h1 = heapq.heappop(heap)
            f2, _, h2 = heapq.heappop(heap)
            cost += f1 + f2
            cnt += 1
            heapq.heappush(heap, (f1 + f2, cnt, max(h1, h2) + 1))
        gpu_cost = int((freq * lens.astype(np.int64)).sum())
        if heap[0][2] <= limit:
            assert gpu_cost == cost, "code is not optimal"
        else:
            assert cost <= gpu_cost <= cost * 1.002, "length-limited code too far from optimal"
    # K6: python decoder reads the bit-stream back to the same codes
    if a.size <= 120000:
        assert np.array_equal(szh_ref.huffman_decode(h, sec), exp_codes.reshape(-1))
    # K8: GPU decode == model reconstruction, and the error bound holds strictly (in float64)
    model = szh_ref.reconstruct(h, sec, exp_codes.reshape(-1)).reshape(a.shape)
    assert np.array_equal(dec, model)
    assert np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) <= eb


def test_outliers_and_nonfinite():
    a = field3d((24, 31, 40))
    a[3, 4, 5] = np.nan
    a[10, 2, 7] = np.inf
    a[20, 20, 20] = 1e30
    a[1, 1, 1] = -3e9
    a[5, 6, 7] = 3000.0   # representable on the lattice (|x/2eb| < 2^23) but a huge Lorenzo delta -> delta outlier
    eb = 1e-3
    codes, pl, dec, st = _roundtrip_device(a, eb)
    q, d, exp_codes, bad, dout = szh_ref.dualquant(a, eb, narrow=bool(st["narrow_codes"]))
    assert np.array_equal(codes, exp_codes.reshape(-1))
    assert st["n_value_outliers"] == int(bad.sum()) >= 4 and st["n_delta_outliers"] == int(dout.sum()) > 0
    fin = np.isfinite(a)
    assert np.array_equal(np.isnan(dec), np.isnan(a)) and np.array_equal(dec[~fin & ~np.isnan(a)], a[~fin & ~np.isnan(a)])
    assert np.max(np.abs(dec[fin].astype(np.float64) - a[fin].astype(np.float64))) <= eb


@pytest.mark.parametrize("shape,dtype", [((19, 26, 68), np.float32), ((9, 17, 64), np.float64), ((3, 9, 11, 72), np.float32),
                                         ((33, 10, 260), np.float32), ((2, 17, 6, 132), np.float64)])
def test_fast_kernel_equals_generic(shape, dtype):
    """the tuned stage-1 kernel and the any-shape kernel must emit identical codes, outliers and payloads"""
    a = (field3d(shape, dtype) if len(shape) == 3 else field4d(shape, dtype))
    a[tuple(s // 2 for s in shape)] = np.nan
    a[tuple(s // 3 for s in shape)] = 4e4
    try:
        sz3_amd.lib().sz3hip_debug_flags(64)          # two-byte codes on both sides (the narrow mode has its own test)
        sz3_amd.lib().sz3hip_debug_force_generic(1)
        c0, p0, d0, s0 = _roundtrip_device(a, 1e-3)
        sz3_amd.lib().sz3hip_debug_force_generic(0)
        c1, p1, d1, s1 = _roundtrip_device(a, 1e-3)
    finally:
        sz3_amd.lib().sz3hip_debug_force_generic(0)
        sz3_amd.lib().sz3hip_debug_flags(0)
    assert np.array_equal(c0, c1)
    assert s0["n_value_outliers"] == s1["n_value_outliers"] and s0["n_delta_outliers"] == s1["n_delta_outliers"]
    h0, _, sec0 = szh_ref.parse(p0)
    h1, _, sec1 = szh_ref.parse(p1)
    assert np.array_equal(sec0["bitstream"], sec1["bitstream"]) and np.array_equal(sec0["lens"], sec1["lens"])
    assert np.array_equal(d0, d1, equal_nan=True)


def test_narrow_and_wide_code_paths_agree():
    """one-byte intermediate codes (probe says the deltas are small) vs two-byte codes: same payload"""
    a = field3d((24, 20, 256))
    c1, p1, d1, s1 = _roundtrip_device(a, 1e-3)
    assert s1["narrow_codes"] == 1
    try:
        sz3_amd.lib().sz3hip_debug_flags(64)   # forbid the narrow mode
        c0, p0, d0, s0 = _roundtrip_device(a, 1e-3)
    finally:
        sz3_amd.lib().sz3hip_debug_flags(0)
    assert s0["narrow_codes"] == 0
    assert np.array_equal(c0, c1) and np.array_equal(p0, p1) and np.array_equal(d0, d1)
    # rough data: the probe must refuse the narrow mode (deltas of a few hundred lattice steps)
    b = field3d((24, 20, 256), sigma=0.2)
    c2, p2, d2, s2 = _roundtrip_device(b, 1e-3)
    assert s2["narrow_codes"] == 0
    assert np.max(np.abs(d2.astype(np.float64) - b.astype(np.float64))) <= 1e-3
    # moderately rough: narrow with a few delta outliers
    e = field3d((24, 20, 256), sigma=0.012)
    c3, p3, d3, s3 = _roundtrip_device(e, 1e-3)
    q, dd, exp_codes, bad, dout = szh_ref.dualquant(e, 1e-3, narrow=bool(s3["narrow_codes"]))
    assert np.array_equal(c3, exp_codes.reshape(-1)) and s3["n_delta_outliers"] == int(dout.sum())
    assert np.max(np.abs(d3.astype(np.float64) - e.astype(np.float64))) <= 1e-3


@pytest.mark.parametrize("algo", ["lorenzo", "interp"])
def test_many_unpredictables_grow_the_lists(algo):
    """More unpredictable values than the default lists hold (n / 32): with a default-sized buffer the device call
    reports SZ3HIP_EOUTLIERS; with sz3hip_payload_bound_max it grows its lists and succeeds, and the host API keeps
    the GPU stream instead of falling back to lossless (the reference keeps any number of unpredictables)."""
    rng = np.random.default_rng(11)
    shape = (64, 64, 64)
    a = field3d(shape)
    mask = rng.random(shape) < 0.01          # 1 % spikes far outside a 64-bin quantiser: each one spoils its neighbours' predictions
    a[mask] += rng.choice([-1.0, 1.0], size=int(mask.sum())).astype(np.float32) * 50.0
    eb = 1e-3
    dev = torch.device("cuda:0")
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG if algo == "lorenzo" else sz3_amd.ALGO_INTERP
    conf.regression = 0
    conf.absErrorBound = eb
    conf.quantbinCnt = 64
    cap = dc.payload_bound(a.size)
    pl = torch.empty(dc.payload_bound(a.size, worst_case=True), dtype=torch.uint8, device=dev)
    with pytest.raises(sz3_amd.SZ3HipError):
        dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
    size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), pl.numel(), 0)
    st = dc.stats()
    assert max(st["n_value_outliers"], st["n_delta_outliers"]) > a.size // 32
    out = torch.empty_like(t)
    dc.decompress(pl.data_ptr(), size, out.data_ptr(), 0)
    torch.cuda.synchronize()
    dec = out.cpu().numpy()
    assert np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) <= eb
    # an easy input afterwards still fits the default bound (the grown lists do not change sz3hip_payload_bound)
    b = field3d(shape)
    tb = torch.from_numpy(b).to(dev)
    conf.quantbinCnt = 65536
    size_b = dc.compress(conf, tb.data_ptr(), pl.data_ptr(), cap, 0)
    assert 0 < size_b < b.nbytes // 4
    # host API: GPU stream, not the lossless fallback
    conf.quantbinCnt = 64
    blob, ratio = sz3_amd.compress(a, conf)
    back, got_conf = sz3_amd.decompress(blob, a.dtype, shape)
    assert np.max(np.abs(back.astype(np.float64) - a.astype(np.float64))) <= eb
    assert ratio > 2 and got_conf.cmprAlgo in (16, 17)


@pytest.mark.parametrize("shape,dtype,qb,eb,sigma,nan", [
    ((17, 260), np.float32, 256, 1e-3, 2e-3, False),       # two-byte codes, radius 128: code 0 lies inside the LDS window
    ((64, 65, 128), np.float64, 256, 1e-2, 5e-2, True),    # one-byte codes, radius 128: delta outliers must reach the histogram
    ((40, 128), np.float32, 256, 1e-3, 2e-3, True),
    ((128, 32, 128), np.float64, 1024, 1e-3, 2e-3, False),
    ((8, 3, 100, 260), np.float64, 4096, 1e-3, 5e-2, True),
    ((256, 256), np.float32, 64, 1e-2, 1e-4, False),
])
def test_small_quantiser_lorenzo(shape, dtype, qb, eb, sigma, nan):
    """quantbinCnt far below the default: the code range is narrower than the kernels' LDS histogram windows, so code 0
    (delta outlier) falls inside / next to them. Found by tests/checks/lorenzo_sweep.py; checked against the numpy model of K1."""
    rng = np.random.default_rng(2)
    grids = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
    a = (sum(np.sin(2 * np.pi * g / (11.0 + 5 * i)) for i, g in enumerate(grids)) + sigma * rng.standard_normal(shape)).astype(dtype)
    if nan:
        a.reshape(-1)[rng.integers(0, a.size, size=max(1, a.size // 500))] = np.nan
    dev = torch.device("cuda:0")
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    cap = dc.payload_bound(a.size, worst_case=True)
    pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.regression = 0  # Lorenzo-1 alone: the plain stream (with regression the block-composed predictor takes over)
    conf.absErrorBound = eb
    conf.quantbinCnt = qb
    size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
    st = dc.stats()
    codes = dc.debug_codes(a.size)
    out = torch.empty_like(t)
    dc.decompress(pl.data_ptr(), size, out.data_ptr(), 0)
    torch.cuda.synchronize()
    dec = out.cpu().numpy()
    q, d, exp_codes, bad, dout = szh_ref.dualquant(a, eb, radius=qb // 2, narrow=bool(st["narrow_codes"]))
    assert np.array_equal(codes, exp_codes.reshape(-1))
    assert (st["n_value_outliers"], st["n_delta_outliers"]) == (int(bad.sum()), int(dout.sum()))
    h, o, sec = szh_ref.parse(pl[:size].cpu().numpy().tobytes())
    model = szh_ref.reconstruct(h, sec, exp_codes.reshape(-1)).reshape(a.shape)
    assert np.array_equal(dec, model, equal_nan=True)
    fin = np.isfinite(a)
    assert np.max(np.abs(dec[fin].astype(np.float64) - a[fin].astype(np.float64))) <= eb
    assert np.array_equal(np.isnan(dec), np.isnan(a))


def test_two_class_code_book_and_its_fallback():
    """wide alphabet (> 4096 symbols): the two-class code book (frequent symbols one by one + one rare class) and, forced by
    debug flag 1024, the one-class construction it falls back to both give a decodable stream within the bound; the class
    form may cost at most 0.5 % of the size"""
    a = field3d((128, 160, 192))
    dev = torch.device("cuda:0")
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    cap = dc.payload_bound(a.size)
    out = torch.empty_like(t)
    sizes, lens = [], []
    try:
        for flag in (0, 1024):
            sz3_amd.lib().sz3hip_debug_flags(flag)
            pl = torch.empty(cap, dtype=torch.uint8, device=dev)
            conf = sz3_amd.Config(*a.shape)
            conf.cmprAlgo = sz3_amd.ALGO_INTERP
            conf.absErrorBound = 1e-5
            n = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
            dc.decompress(pl.data_ptr(), n, out.data_ptr(), 0)
            torch.cuda.synchronize()
            assert float((out.double() - t.double()).abs().max()) <= 1e-5
            h, o, sec = szh_ref.parse(pl[:n].cpu().numpy().tobytes())
            assert h["sym_count"] > 4096 and szh_ref.kraft(sec["lens"]) <= 1.0
            sizes.append(n)
            lens.append(sec["lens"])
    finally:
        sz3_amd.lib().sz3hip_debug_flags(0)
    assert not np.array_equal(lens[0], lens[1])          # the two constructions really differ ...
    assert sizes[1] <= sizes[0] <= 1.005 * sizes[1]       # ... and the class form costs next to nothing


def test_payload_does_not_depend_on_the_context_history():
    """window sizes of stage 1 / the packer are per-context choices taken from the previous call's probe: a context that
    has just seen a rough field (wide windows selected) must produce the same bytes as a fresh one"""
    shape = (96, 128, 256)
    rng = np.random.default_rng(4)
    rough = (field3d(shape).astype(np.float64) + 0.05 * rng.standard_normal(shape)).astype(np.float32)
    smooth = field3d(shape)
    dev = torch.device("cuda:0")

    def run(dc, a, eb):
        t = torch.from_numpy(a).to(dev)
        cap = dc.payload_bound(a.size, worst_case=True)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        conf = sz3_amd.Config(*a.shape)
        conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
        conf.regression = 0  # Lorenzo-1 alone: the plain stream (with regression the block-composed predictor takes over)
        conf.absErrorBound = eb
        n = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
        out = torch.empty_like(t)
        dc.decompress(pl.data_ptr(), n, out.data_ptr(), 0)
        torch.cuda.synchronize()
        assert float((out.double() - t.double()).abs().max()) <= eb
        return pl[:n].cpu().numpy().tobytes()

    used = sz3_amd.DeviceCompressor(smooth.size, smooth.dtype)
    p_rough_1 = run(used, rough, 1e-5)       # deltas of thousands of lattice steps: the wide windows get selected ...
    p_rough_2 = run(used, rough, 1e-5)       # ... and used
    p_smooth_used = run(used, smooth, 1e-3)  # first smooth call still runs with the wide windows
    p_smooth_used2 = run(used, smooth, 1e-3)
    fresh = sz3_amd.DeviceCompressor(smooth.size, smooth.dtype)
    p_smooth_fresh = run(fresh, smooth, 1e-3)
    p_rough_fresh = run(sz3_amd.DeviceCompressor(smooth.size, smooth.dtype), rough, 1e-5)
    assert p_rough_1 == p_rough_2 == p_rough_fresh
    assert p_smooth_used == p_smooth_used2 == p_smooth_fresh


def test_stage_calls_out_of_order_are_refused():
    """stage2 needs a stage1, and only one stage2 per stage1 (the code book's range words are accumulated by atomics)"""
    a = field3d((32, 32, 64))
    dev = torch.device("cuda:0")
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    cap = dc.payload_bound(a.size)
    pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.regression = 0  # Lorenzo-1 alone: the plain stream (with regression the block-composed predictor takes over)
    conf.absErrorBound = 1e-3
    with pytest.raises(sz3_amd.SZ3HipError):
        dc.stage2(pl.data_ptr(), cap, 0)
    dc.stage1(conf, t.data_ptr(), 0)
    dc.stage2(pl.data_ptr(), cap, 0)
    with pytest.raises(sz3_amd.SZ3HipError):
        dc.stage2(pl.data_ptr(), cap, 0)
    size = dc.finish(0)
    ref = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)  # the whole call again: same payload size
    assert size == ref



def test_context_hints_survive_a_change_of_regime():
    """A context launches ONE form of the stage-1 kernel and ONE form of the code-book kernel, chosen from what its previous
    call found (code width, alphabet size). Data that changes character between calls must still come out right: the
    one-byte form of stage 1 falls back to two-byte codes inside its own LDS budget when this call's probe says so, and a
    code-book form that meets the other form's alphabet makes `finish` repeat stage 2. Alternating a smooth field at a loose
    bound (128 symbols, one-byte codes) with the same field at a tight bound (thousands of symbols, two-byte codes), every
    call against a fresh context's result."""
    import torch
    dev = torch.device("cuda:0")
    shape = (48, 64, 256)
    a = field3d(shape)
    t = torch.from_numpy(a).to(dev)
    n = a.size
    shared = sz3_amd.DeviceCompressor(n, np.float32)
    shared.set_deterministic(True)  # (payloads are compared with a fresh context's: the previous book stands only when it IS this call's)
    cap = shared.payload_bound(n, worst_case=True)
    for eb in (1e-3, 1e-6, 1e-3, 1e-3, 2e-6, 1e-6, 1e-2):
        conf = _conf(shape, eb)
        out = {}
        for name, dc in (("shared", shared), ("fresh", sz3_amd.DeviceCompressor(n, np.float32))):
            pl = torch.empty(cap, dtype=torch.uint8, device=dev)
            size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
            dec = torch.empty_like(t)
            dc.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
            torch.cuda.synchronize()
            out[name] = (pl[:size].cpu().numpy().tobytes(), dec.cpu().numpy())
        assert out["shared"][0] == out["fresh"][0], "payload depends on the context's history (eb %g)" % eb
        assert float(np.max(np.abs(out["shared"][1].astype(np.float64) - a.astype(np.float64)))) <= eb


@pytest.mark.parametrize("shape", [(40, 64, 500), (24, 70, 504), (3, 40, 64, 500)], ids=["500", "504", "4d-500"])
def test_unpredictable_values_in_tiles_that_end_beyond_the_row(shape):
    """Rows of 500: the second 256-wide tile of the marching kernel runs with its last lanes off. 2 % NaN fill its per-wave
    queue of unpredictable values several times per tile, so it is flushed by the active lanes only — every record must
    still reach the list (striding the copy by 64 lost the records that fell on an inactive lane: those points came back as
    lattice values instead of their raw ones)."""
    gen = field3d if len(shape) == 3 else field4d
    a = gen(shape)
    flat = a.reshape(-1)
    rng = np.random.default_rng(7)
    idx = rng.choice(flat.size, size=flat.size // 50, replace=False)
    flat[idx] = np.nan
    flat[idx[:100]] = 1e30
    dev = torch.device("cuda:0")
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, np.float32)
    cap = dc.payload_bound(a.size, worst_case=True)
    pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.regression = 0
    conf.absErrorBound = 1e-3
    n = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
    assert idx.size <= dc.stats()["n_value_outliers"] <= idx.size + 200  # (+ the odd finite value whose lattice point misses the bound in f32)
    out = torch.empty_like(t)
    dc.decompress(pl.data_ptr(), n, out.data_ptr(), 0)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    m = np.isnan(a)
    assert np.array_equal(np.isnan(o), m)
    big = a == 1e30
    assert np.array_equal(o[big], a[big])
    ok = ~m & ~big
    assert np.max(np.abs(o[ok].astype(np.float64) - a[ok].astype(np.float64))) <= 1e-3


@pytest.mark.parametrize("shape", [(64,), (700,), (1, 1024)], ids=["1d-64", "1d-700", "2d-one-row"])  # (at most 1024 elements: the lists' floor)
def test_every_element_a_listed_delta(shape):
    """quantbinCnt 64 on a field that jumps by hundreds of lattice steps from element to element: every code is 0, the
    alphabet has one symbol (zero-length code words, empty bit stream) and the decoder's values all come from the delta list —
    on either chain (the half-width one keeps int16 values)."""
    n = int(np.prod(shape))
    a = (((-1.0) ** np.arange(n)) * 0.1 * (1 + np.arange(n) % 3)).astype(np.float32).reshape(shape)
    dev = torch.device("cuda:0")
    t = torch.from_numpy(a).to(dev)
    res = []
    try:
        for flag in (0, 2097152):
            sz3_amd.lib().sz3hip_debug_flags(flag)
            dc = sz3_amd.DeviceCompressor(a.size, np.float32)
            cap = dc.payload_bound(a.size, worst_case=True)
            pl = torch.empty(cap, dtype=torch.uint8, device=dev)
            conf = sz3_amd.Config(*shape)
            conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
            conf.regression = 0
            conf.absErrorBound = 1e-4
            conf.quantbinCnt = 64
            size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
            st = dc.stats()
            assert st["n_delta_outliers"] == n and st["max_code_len"] == 0
            out = torch.empty_like(t)
            dc.decompress(pl.data_ptr(), size, out.data_ptr(), 0)
            torch.cuda.synchronize()
            res.append(out.cpu().numpy())
    finally:
        sz3_amd.lib().sz3hip_debug_flags(0)
    assert np.array_equal(res[0], res[1])
    assert np.max(np.abs(res[0].astype(np.float64) - a.astype(np.float64))) <= 1e-4


@pytest.mark.parametrize("shape,dtype", [((200, 208, 224), np.float32), ((16, 96, 104, 112), np.float32),
                                         ((136, 144, 160), np.float64), ((20, 40, 44, 48), np.float32)],
                         ids=["3d-level-kernels", "4d-pass-kernels", "3d-f64-global-trials", "4d-global-trials"])
def test_speculative_stage1_follows_the_tuner(shape, dtype):
    """ALGO_INTERP_LORENZO on 3-D arrays of the level kernels: a context that holds a previous tuner outcome starts stage 1 with it
    beside the tuner and enqueues it again when the tuner decides otherwise. Fields whose outcomes differ (cubic / linear, both
    direction orders, three (alpha, beta) pairs) alternate on one context: every payload equals a fresh context's, the tuner's
    report is the same, and both a confirmed and a refuted speculation occur. The f64 and the second 4-D case sample blocks
    too large for the LDS trial kernel (33^3 x 8 bytes, 17^4 x 4 bytes): their trials keep codes in global memory — an array of
    the tuner's own, not the one the speculative stage 1 is filling meanwhile."""
    z, y, x = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape[-3:]], indexing="ij")
    rng = np.random.default_rng(3)
    fields = {
        "c2": field3d(shape[-3:]),
        "noisy": (np.sin(2 * np.pi * x / 40) + 0.3 * rng.standard_normal(shape[-3:])).astype(np.float32),
        "aniso-x": (np.sin(2 * np.pi * x / 7) + 0.05 * np.sin(2 * np.pi * z / 90)).astype(np.float32),
        "smooth": (np.sin(2 * np.pi * x / 150) * np.cos(2 * np.pi * y / 170) * np.sin(2 * np.pi * z / 130)).astype(np.float32),
    }
    del x, y, z
    if len(shape) == 4:  # (a slowly drifting copy per time step)
        fields = {k: np.stack([v * (1 + 0.01 * t) for t in range(shape[0])]) for k, v in fields.items()}
    fields = {k: v.astype(dtype) for k, v in fields.items()}
    dev = torch.device("cuda:0")
    n = int(np.prod(shape))
    shared = sz3_amd.DeviceCompressor(n, dtype)
    shared.set_deterministic(True)
    cap = shared.payload_bound(n, worst_case=True)
    conf = sz3_amd.Config(*shape)
    conf.absErrorBound = 1e-2
    seen, outcomes = [], set()
    for name in ["c2", "c2", "noisy", "noisy", "aniso-x", "c2", "smooth", "c2"]:
        t = torch.from_numpy(fields[name]).to(dev)
        pls = []
        reps = []
        for dc in (shared, sz3_amd.DeviceCompressor(n, dtype)):
            pl = torch.empty(cap, dtype=torch.uint8, device=dev)
            size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
            torch.cuda.synchronize()
            pls.append(pl[:size].clone())
            reps.append(dc.tuner_report())
            if dc is shared:
                seen.append(dc.speculated)
            else:
                assert dc.speculated == 0
        assert reps[0] == reps[1] and reps[0]["ran"] == 1 and reps[0]["use_interp"] == 1
        assert pls[0].numel() == pls[1].numel() and torch.equal(pls[0], pls[1]), "payload depends on the context's history (%s)" % name
        outcomes.add((reps[0]["interpAlgo"], reps[0]["interpDirection"], reps[0]["interpAlpha"], reps[0]["interpBeta"]))
        out = torch.empty_like(t)
        shared.decompress(pls[0].data_ptr(), pls[0].numel(), out.data_ptr(), 0)
        torch.cuda.synchronize()
        assert float((out.double() - t.double()).abs().max()) <= 1e-2, (name, seen)
    assert len(outcomes) >= 2, outcomes
    assert seen[0] == 0 and seen[1] == 1 and seen[3] == 1 and 2 in seen, seen


def test_speculative_code_book_is_confirmed_or_replaced():
    """Stage 2 of a context whose previous call left a small code book (same predictor, same radius, short outlier lists) packs
    with THAT book while this call's is built from this call's histogram by a workgroup of the packer's launch; `finish` reads
    the verdict and repeats the encoder when the books differ. Whatever happens the payload is the one a fresh context
    produces: same array again (hit), another realisation of the field (miss), another bound (miss), the interpolation
    predictor and wide alphabets (their books need a compute unit's whole LDS: built on a stream of their own beside the
    encoder, joined in front of a verdict — hit, miss, first call of a predictor: not speculated), deltas that need two-byte
    codes after a one-byte call (the one-launch form of stage 1 assumed one byte: the whole call is repeated, counted as a
    miss), speculation switched off."""
    dev = torch.device("cuda:0")
    shape = (40, 64, 256)
    a = field3d(shape)
    b = field3d(shape, seed=77)
    n = a.size
    shared = sz3_amd.DeviceCompressor(n, np.float32)
    shared.set_speculation(True, backoff=False)  # (the product sits out 1, 2, 4, 8 calls after a miss: the outcomes below are per call)
    shared.set_deterministic(True)  # (the verdict of this test: same book or the encoder once more; the tolerant verdict has a test of its own below)
    cap = shared.payload_bound(n, worst_case=True)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)

    def run(dc, t, conf):
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
        dec = torch.empty_like(t)
        dc.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
        torch.cuda.synchronize()
        assert float((dec.double() - t.double()).abs().max()) <= conf.absErrorBound
        return pl[:size].cpu().numpy().tobytes()

    interp = sz3_amd.Config(*shape)
    interp.cmprAlgo = sz3_amd.ALGO_INTERP
    interp.absErrorBound = 1e-3
    # (array, config, expected outcome: +1 hit, -1 miss, 0 not speculated)
    steps = [(ta, _conf(shape, 1e-3), 0), (ta, _conf(shape, 1e-3), +1), (tb, _conf(shape, 1e-3), -1), (tb, _conf(shape, 1e-3), +1),
             (ta, _conf(shape, 2e-3), -1), (ta, interp, 0), (ta, interp, +1), (tb, interp, -1), (ta, _conf(shape, 1e-6), -1),
             (ta, _conf(shape, 1e-6), 0), (ta, _conf(shape, 1e-3), 0), (tb, _conf(shape, 1e-6), -1)]  # (1e-6 on f32: long outlier lists, no speculation)
    for k, (t, conf, want) in enumerate(steps):
        h0, m0 = shared.spec_stats()
        got = run(shared, t, conf)
        h1, m1 = shared.spec_stats()
        assert (h1 - h0, m1 - m0) == {0: (0, 0), 1: (1, 0), -1: (0, 1)}[want], "step %d: hits %d misses %d" % (k, h1 - h0, m1 - m0)
        assert got == run(sz3_amd.DeviceCompressor(n, np.float32), t, conf), "step %d: payload depends on the context's history" % k
    shared.set_speculation(False)
    h0, m0 = shared.spec_stats()
    assert run(shared, ta, _conf(shape, 1e-3)) == run(sz3_amd.DeviceCompressor(n, np.float32), ta, _conf(shape, 1e-3))
    assert shared.spec_stats() == (h0, m0)
    shared.set_speculation(True, backoff=False)
    shared.forget()
    assert run(shared, ta, _conf(shape, 1e-3)) == run(sz3_amd.DeviceCompressor(n, np.float32), ta, _conf(shape, 1e-3))
    assert shared.spec_stats() == (h0, m0)  # a context that forgot its book does not speculate


def test_speculative_wide_code_book_on_its_own_stream():
    """Alphabets of thousands of symbols (C4: f64, deltas of thousands of lattice steps): the previous call's book packs while this
    call's is built by k_codebook<1> on a stream of its own, a verdict joins them. Same array again: hit; another realisation:
    miss, the encoder once more with the fresh book; a smooth field after a wide one (the wide form meets a small alphabet):
    miss, stage 2 once more. Every payload is a fresh context's."""
    dev = torch.device("cuda:0")
    shape = (24, 96, 256)
    a = field3d(shape, np.float64, sigma=2e-6)
    b = field3d(shape, np.float64, seed=5, sigma=2e-6)
    n = a.size
    shared = sz3_amd.DeviceCompressor(n, np.float64)
    shared.set_speculation(True, backoff=False)
    shared.set_deterministic(True)
    cap = shared.payload_bound(n, worst_case=True)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)

    def run(dc, t, conf):
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
        dec = torch.empty_like(t)
        dc.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
        torch.cuda.synchronize()
        assert float((dec - t).abs().max()) <= conf.absErrorBound
        return pl[:size].cpu().numpy().tobytes(), dc.stats()["n_symbols"] if "n_symbols" in dc.stats() else 0

    steps = [(ta, 1e-6, 0), (ta, 1e-6, +1), (tb, 1e-6, -1), (tb, 1e-6, +1), (ta, 1e-2, -1), (ta, 1e-2, +1), (ta, 1e-6, -1)]
    for k, (t, eb, want) in enumerate(steps):
        conf = _conf(shape, eb)
        h0, m0 = shared.spec_stats()
        got, _ = run(shared, t, conf)
        h1, m1 = shared.spec_stats()
        ref, _ = run(sz3_amd.DeviceCompressor(n, np.float64), t, conf)
        assert got == ref, "step %d: payload depends on the context's history" % k
        assert (h1 - h0, m1 - m0) == {0: (0, 0), 1: (1, 0), -1: (0, 1)}[want], "step %d: hits %d misses %d" % (k, h1 - h0, m1 - m0)
    import szh_ref
    h, _, _ = szh_ref.parse(run(sz3_amd.DeviceCompressor(n, np.float64), ta, _conf(shape, 1e-6))[0])
    assert h["sym_count"] > 2000  # (the alphabet the first steps are about)


@pytest.mark.parametrize("sigma,eb", [(2e-3, 1e-3), (8e-3, 1e-3), (3e-2, 1e-3)], ids=["smooth", "noisy", "rough"])
def test_repeated_calls_take_every_shortcut_and_change_nothing(sigma, eb):
    """A context's later calls take shortcuts from what the previous call found — the one-launch form of stage 1 (assumes the code
    width, runs the probe itself, sums the code bits per 256-element segment with the previous book's lengths), the encoder
    with the previous book while this call's is built by a workgroup of the packer's launch, the histogram fold in the scan's
    launch, the zeroing of the counters behind the previous call. Two realisations of one field alternate on one context;
    rows of 256 and of 512 (the segment sums need rows cut into whole 256-element segments), 3-D and 4-D: every payload equals a
    fresh context's, and the repeated arrays are confirmed."""
    dev = torch.device("cuda:0")
    for shape in ((24, 36, 256), (9, 20, 512), (3, 5, 7, 256)):
        rng = np.random.default_rng(11)
        base = field3d(shape[-3:], sigma=0.0)
        if len(shape) == 4:
            base = np.stack([base * (1 + 0.01 * t) for t in range(shape[0])])
        a = (base + rng.normal(0, sigma, shape)).astype(np.float32)
        b = (base + rng.normal(0, sigma, shape)).astype(np.float32)
        n = a.size
        shared = sz3_amd.DeviceCompressor(n, np.float32)
        shared.set_deterministic(True)
        cap = shared.payload_bound(n, worst_case=True)
        conf = _conf(shape, eb)

        def run(dc, arr):
            t = torch.from_numpy(arr).to(dev)
            pl = torch.empty(cap, dtype=torch.uint8, device=dev)
            size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
            dec = torch.empty_like(t)
            dc.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
            torch.cuda.synchronize()
            assert float((dec.double() - t.double()).abs().max()) <= eb
            return pl[:size].cpu().numpy().tobytes()

        want = {id(a): run(sz3_amd.DeviceCompressor(n, np.float32), a), id(b): run(sz3_amd.DeviceCompressor(n, np.float32), b)}
        for k, arr in enumerate([a, a, a, a, b, b, a, b, b, b, b]):
            assert run(shared, arr) == want[id(arr)], "call %d (%s): payload depends on the context's history" % (k, shape)
        hits, misses = shared.spec_stats()
        if sigma < 1e-2:  # (the rough field needs two-byte codes and a wide code book: no shortcuts to confirm)
            assert hits >= 3, (hits, misses)


@pytest.mark.parametrize("dtype,shape,eb,sigma", [(np.float32, (40, 64, 512), 1e-3, 2e-3), (np.float32, (24, 36, 256), 1e-3, 8e-3)],
                         ids=["small-book-512", "small-book-256"])
def test_a_previous_book_that_is_nearly_this_calls_book_stands(dtype, shape, eb, sigma):
    """The device API's default verdict (round 4): the previous call's code book stands when it is a complete code over this call's
    alphabet and codes it within 1/1024 of this call's own book's size. Realisations of one field — no two histograms equal — alternate
    on one context: (next to) every call after the first is a hit, every payload decodes within the bound with the decoder unaware,
    its size within 0.15 % of a fresh context's; the same context in deterministic mode gives the fresh context's bytes; a field of
    another character (another bound) is still a miss, and a book that lacks one of this call's symbols never stands. (Small
    alphabets: their books carry margins, cb_margins. A wide alphabet — C4's thousands of symbols — ends in hundreds of symbols that
    occur once: the next array always has some its predecessor's book lacks, and the verdict is a miss as before.)"""
    dev = torch.device("cuda:0")
    fields = [field3d(shape, dtype, seed=100 + k, sigma=sigma) for k in range(4)]
    n = fields[0].size
    shared = sz3_amd.DeviceCompressor(n, dtype)
    shared.set_speculation(True, backoff=False)
    cap = shared.payload_bound(n, worst_case=True)
    conf = _conf(shape, eb)

    def run(dc, arr, c=conf):
        t = torch.from_numpy(arr).to(dev)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        size = dc.compress(c, t.data_ptr(), pl.data_ptr(), cap, 0)
        dec = torch.empty_like(t)
        dc.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
        torch.cuda.synchronize()
        assert float((dec.double() - t.double()).abs().max()) <= c.absErrorBound
        return pl[:size].cpu().numpy().tobytes()

    fresh = [run(sz3_amd.DeviceCompressor(n, dtype), f) for f in fields]
    order = [0, 1, 2, 3, 0, 2, 1, 3, 3, 0]
    for k, i in enumerate(order):
        got = run(shared, fields[i])
        assert abs(len(got) - len(fresh[i])) <= 1.5e-3 * len(fresh[i]), (k, len(got), len(fresh[i]))
    hits, misses = shared.spec_stats()
    assert hits >= len(order) - 2 and misses <= 1, (hits, misses)
    # deterministic mode: the fresh context's bytes, and distinct realisations are misses again
    shared.set_deterministic(True)
    h0, m0 = shared.spec_stats()
    for i in (1, 2):
        assert run(shared, fields[i]) == fresh[i]
    h1, m1 = shared.spec_stats()
    assert (h1 - h0, m1 - m0) == (0, 2)
    shared.set_deterministic(False)
    # another bound = another alphabet: the wider book of the loose bound's successor lacks nothing, but costs more than 1/1024 -> miss;
    # the tighter bound's alphabet has symbols the book lacks -> miss; both payloads decode (run() checks) and are a fresh context's
    for factor in (4.0, 0.25):
        c2 = _conf(shape, eb * factor)
        h0, m0 = shared.spec_stats()
        got = run(shared, fields[0], c2)
        h1, m1 = shared.spec_stats()
        assert (h1 - h0, m1 - m0) == (0, 1), (factor, h1 - h0, m1 - m0)
        assert got == run(sz3_amd.DeviceCompressor(n, dtype), fields[0], c2)


def test_outlier_lists_beyond_the_sort_workgroups_reach_are_sorted_by_finish():
    """more than 32768 records in a list (a bound far below the noise): the code book's sort workgroups leave such a list in arrival
    order and finish() sorts the payload's list sections with a device-wide radix sort (sz3hip_sortlists.hip) — index order, the
    same payload from call to call, and the decoder's values within the bound"""
    import torch
    import szh_ref
    rng = np.random.default_rng(9)
    a = (np.sin(np.arange(1 << 21) / 5000.0) + rng.normal(0, 2e-4, 1 << 21)).astype(np.float64)
    a[rng.choice(a.size, 20000, replace=False)] += 5.0  # far deltas on both sides of every spike
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(a).to(dev)
    conf = sz3_amd.Config(a.size)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, 0
    conf.absErrorBound = 1e-6
    conf.quantbinCnt = 4096
    dc = sz3_amd.DeviceCompressor(a.size, np.float64)
    dc.set_deterministic(True)
    cap = dc.payload_bound_max(a.size) if hasattr(dc, "payload_bound_max") else 4 * dc.payload_bound(a.size)
    pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    out = torch.empty_like(d_in)
    st = torch.cuda.current_stream().cuda_stream
    blobs = []
    for _ in range(3):
        size = dc.compress(conf, d_in.data_ptr(), pl.data_ptr(), cap, st)
        blobs.append(pl[:size].cpu().numpy().tobytes())
    h, o, sec = szh_ref.parse(blobs[0])
    assert h["n_dout"] > 32768, h["n_dout"]
    assert (np.diff(sec["dout_idx"].astype(np.int64)) > 0).all() and (np.diff(sec["vout_idx"].astype(np.int64)) > 0).all()
    assert blobs[0] == blobs[1] == blobs[2]
    dc.decompress(pl.data_ptr(), size, out.data_ptr(), st)
    torch.cuda.synchronize()
    assert float((out - d_in).abs().max()) <= 1e-6


def test_wide_code_book_with_and_without_the_compaction_launch():
    """Round 5: the wide code book's keys are compacted by k_cb_compact over the whole chip in front of the book's launch; debug flag 1
    keeps the compaction inside the book's workgroup. Same book, same payload — on an interpolation stream (thousands of symbols, the
    two-class construction), on a rough Lorenzo stream, and across the repeat of stage 2 that a mispredicted book form causes."""
    dev = torch.device("cuda:0")
    L = sz3_amd.lib()
    a = field3d((96, 96, 96), seed=4)
    rough = (a + np.random.default_rng(2).normal(0, 0.2, a.shape)).astype(np.float32)
    cases = [(a, sz3_amd.ALGO_INTERP, 1e-5), (rough, sz3_amd.ALGO_LORENZO_REG, 1e-3), (a, sz3_amd.ALGO_INTERP, 1e-2)]
    outs = {}
    for flag in (0, 1):
        dc = sz3_amd.DeviceCompressor(a.size, np.float32)   # one context through all cases: small and wide books alternate (mispredicted forms)
        dc.set_deterministic(True)
        cap = dc.payload_bound(a.size, worst_case=True)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        L.sz3hip_debug_flags(flag)
        try:
            for k, (arr, algo, eb) in enumerate(cases + cases[:1]):
                conf = sz3_amd.Config(*arr.shape)
                conf.cmprAlgo = algo
                conf.regression = 0
                conf.absErrorBound = eb
                t = torch.from_numpy(arr).to(dev)
                size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
                outs[(flag, k)] = pl[:size].cpu().numpy().tobytes()
                o = torch.empty_like(t)
                dc.decompress(pl.data_ptr(), size, o.data_ptr(), 0)
                torch.cuda.synchronize()
                assert float((o.double() - t.double()).abs().max()) <= eb
        finally:
            L.sz3hip_debug_flags(0)
    for k in range(4):
        assert outs[(0, k)] == outs[(1, k)], "case %d: payloads differ" % k
