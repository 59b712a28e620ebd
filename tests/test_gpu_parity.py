"""GPU parity tests (-m gpu): the HIP path through the C ABI against the oracle (CPU restatement of the reference
algorithm, itself pinned to the reference build) and against the committed reference goldens.

Parity bar for this floating-point path (BASELINE.json north_star): the decompressed output stays within the user's
absolute error bound of the input on identical inputs — checked STRICTLY (<= eb, evaluated in float64), the same
criterion the reference's own tests use (tools/sz3/sz3_smoke_test.cpp:43-49, tools/pysz/tests/test_pysz.py:49,61).
Consequences also asserted: |x_gpu - x_reference| <= 2 eb, compression ratio within a stated band of the oracle's.
"""
import ctypes as C
import os

import numpy as np
import pytest

import sz3_amd
from fields import field1d, field2d, field3d, field4d
from oracle_binding import ALGO_LORENZO_REG, EB_ABS, EB_REL, make_config, oracle_compress, oracle_decompress

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "golden.npz"))


def _gpu_roundtrip(a, **kw):
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.regression = 0  # Lorenzo-1 alone = the plain stream; tests/test_gpu_regression.py covers the block-composed predictor
    for k, v in kw.items():
        setattr(conf, k, v)
    blob, ratio = sz3_amd.compress(a, conf)
    dec, c2 = sz3_amd.decompress(blob, a.dtype, a.shape)
    return blob, ratio, dec, c2


# BASELINE.json configs at sizes the oracle finishes in seconds: (name, generator, GPU config, oracle config, ratio band)
CASES = [
    ("C1-1d-2^20-lorenzo_reg-abs1e-3", lambda: field1d(1 << 20), dict(absErrorBound=1e-3, regression=1), dict(abs_eb=1e-3, regression=True), 0.97),  # (its specified set on both sides: 4.74 vs 4.75)
    ("C2-3d-128c-lorenzo-abs1e-3", lambda: field3d((128, 128, 128)), dict(absErrorBound=1e-3), dict(abs_eb=1e-3), 0.97),
    ("C3-3d-96c-abs1e-4", lambda: field3d((96, 96, 96)), dict(absErrorBound=1e-4), dict(abs_eb=1e-4), 0.97),
    ("C4-3d-f64-96c-lorenzo_reg-abs1e-6", lambda: field3d((96, 96, 96), np.float64, sigma=2e-6), dict(absErrorBound=1e-6), dict(abs_eb=1e-6, regression=True), 0.97),
    ("C5-4d-12x40x40x40-rel1e-3", lambda: field4d((12, 40, 40, 40)), dict(errorBoundMode=sz3_amd.EB_REL, relErrorBound=1e-3), dict(eb_mode=EB_REL, rel_eb=1e-3, regression=True), 0.90),
    ("2d-300x500-abs1e-2", lambda: field2d((300, 500)), dict(absErrorBound=1e-2), dict(abs_eb=1e-2), 0.95),
    ("ragged-3d-37x41x43", lambda: field3d((37, 41, 43)), dict(absErrorBound=1e-3), dict(abs_eb=1e-3), 0.95),
]


@pytest.mark.parametrize("name,gen,gkw,okw,band", CASES, ids=[c[0] for c in CASES])
def test_parity_with_oracle(name, gen, gkw, okw, band):
    a = gen()
    blob, ratio, dec, c2 = _gpu_roundtrip(a, **gkw)
    oconf = make_config(a.shape, algo=ALGO_LORENZO_REG, **okw)
    oblob = oracle_compress(a, oconf)
    odec, oc2 = oracle_decompress(oblob, a.dtype, a.shape)
    eb = oc2.absErrorBound
    assert c2.cmprAlgo == sz3_amd.ALGO_HIP_LORENZO and c2.errorBoundMode == sz3_amd.EB_ABS
    assert c2.absErrorBound == pytest.approx(eb, rel=1e-12), "error-bound conversion differs from the reference"
    a64, d64, o64 = a.astype(np.float64), dec.astype(np.float64), odec.astype(np.float64)
    assert np.max(np.abs(d64 - a64)) <= eb          # tolerance = the user's bound, strict
    assert np.max(np.abs(o64 - a64)) <= eb          # the reference algorithm keeps the same bound
    assert np.max(np.abs(d64 - o64)) <= 2 * eb
    oratio = a.nbytes / len(oblob)
    assert ratio >= band * oratio, "ratio %.3f vs reference %.3f" % (ratio, oratio)


def test_against_reference_goldens():
    """committed outputs of the reference itself (tests/golden/make_golden.py)"""
    for name, gen, gkw in [("f32_64c_lorenzo_1e-3", lambda: field3d((64, 64, 64)), dict(absErrorBound=1e-3)),
                           ("f32_8x8x128_lorenzo_1e-3", lambda: field3d((8, 8, 128)), dict(absErrorBound=1e-3)),
                           ("f32_2d_100x100_lorenzo_1e-2", lambda: field2d((100, 100)), dict(absErrorBound=1e-2)),
                           ("f32_1d_65536_lorenzo_reg_1e-3", lambda: field1d(65536), dict(absErrorBound=1e-3))]:
        a = gen()
        blob, ratio, dec, c2 = _gpu_roundtrip(a, **gkw)
        eb = gkw["absErrorBound"]
        assert np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) <= eb
        assert float(GOLD[name + "/max_err"]) <= eb
        if name + "/dec" in GOLD:
            assert np.max(np.abs(dec.astype(np.float64) - GOLD[name + "/dec"].astype(np.float64))) <= 2 * eb
        assert len(blob) <= 1.12 * int(GOLD[name + "/size"])


def test_reference_ci_fixture():
    """the reference's CI criterion on its 8 x 8 x 128 fixture (.github/workflows/cmake.yml:53-65): ABS 1 -> err <= 1"""
    from fields import testfloat_like
    a = testfloat_like()  # (an analytic stand-in of the same shape and character: the reference's file is not kept in this repository)
    blob, ratio, dec, c2 = _gpu_roundtrip(a, absErrorBound=1.0)
    assert np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) <= 1.0 and ratio > 10


def test_pysz_style_roundtrips():
    """tools/pysz/tests/test_pysz.py:24-72 on the GPU path"""
    rng = np.random.default_rng(0)
    a = rng.random((100, 100), dtype=np.float32)
    _, _, dec, _ = _gpu_roundtrip(a, absErrorBound=1e-2)
    assert np.max(np.abs(dec.astype(np.float64) - a)) <= 1e-2
    b = rng.random((50, 50))
    _, _, dec, _ = _gpu_roundtrip(b, absErrorBound=1e-6)
    assert np.max(np.abs(dec - b)) <= 1e-6 and dec.dtype == np.float64
    c = rng.random((20, 30, 40), dtype=np.float32)
    _, _, dec, c2 = _gpu_roundtrip(c, errorBoundMode=sz3_amd.EB_REL, relErrorBound=1e-3)
    assert np.max(np.abs(dec.astype(np.float64) - c)) <= 1e-3 * float(c.max() - c.min())
    md, psnr, nrmse = sz3_amd.verify(c, dec)
    assert md <= c2.absErrorBound and psnr > 50


def test_dispatcher_policies_and_edge_cases():
    a = field3d((24, 31, 40))
    # eb == 0 -> lossless stream, bit exact (api/impl/SZDispatcher.hpp:19-21)
    blob, ratio, dec, c2 = _gpu_roundtrip(a, absErrorBound=0.0)
    assert c2.cmprAlgo == sz3_amd.ALGO_LOSSLESS and np.array_equal(dec, a)
    # bound far below the noise: ratio < 3 -> zstd-only comparison keeps the smaller (SZDispatcher.hpp:62-74), or the
    # outlier lists overflow -> lossless fallback (SZDispatcher.hpp:44-59); either way the bound holds
    blob, ratio, dec, c2 = _gpu_roundtrip(a, absErrorBound=1e-9)
    assert np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) <= 1e-9
    # NaN / Inf / huge values survive bit-exactly, everything else keeps the bound
    b = a.copy()
    b[3, 4, 5] = np.nan
    b[10, 2, 7] = -np.inf
    b[20, 20, 20] = 1e30
    blob, ratio, dec, c2 = _gpu_roundtrip(b, absErrorBound=1e-3)
    m = np.isfinite(b) & (np.abs(b) < 1e20)
    assert np.isnan(dec[3, 4, 5]) and dec[10, 2, 7] == -np.inf and dec[20, 20, 20] == np.float32(1e30)
    assert np.max(np.abs(dec[m].astype(np.float64) - b[m].astype(np.float64))) <= 1e-3
    # constant field: single-symbol alphabet, zero-length code, empty bit stream
    c = np.full((40, 40, 40), 2.5, np.float32)
    blob, ratio, dec, c2 = _gpu_roundtrip(c, absErrorBound=1e-3)
    assert np.max(np.abs(dec - c)) <= 1e-3 and ratio > 500
    # tiny and degenerate shapes (dims of 1 dropped like Config.hpp:164-168)
    for shape in [(1,), (2,), (1, 1, 5), (3, 1, 4)]:
        d = np.arange(int(np.prod(shape)), dtype=np.float32).reshape(shape) * 0.37
        blob, ratio, dec, c2 = _gpu_roundtrip(d, absErrorBound=1e-2)
        assert np.max(np.abs(dec.reshape(-1) - d.reshape(-1))) <= 1e-2
    # buffer too small -> error, not a truncated stream (api/sz.hpp:47-49)
    conf = sz3_amd.Config(*a.shape)
    out = np.empty(100, dtype=np.uint8)
    assert sz3_amd.lib().sz3hip_compress(C.byref(conf._c), 0, a.ctypes.data, out.ctypes.data, out.size) == 0
    assert b"not large enough" in sz3_amd.lib().sz3hip_last_error()
    # a stream of the CPU reference (ALGO_LORENZO_REG; round 4) is read — the reference's own values, bit for bit (tests/test_gpu_stock.py) —;
    # a 4-D one since round 5; what is not built (4-D blocks beyond 6^4) is refused, not mis-decoded
    from oracle_binding import oracle_decompress
    oblob = oracle_compress(a, make_config(a.shape, abs_eb=1e-3))
    got, _ = sz3_amd.decompress(oblob, np.float32, a.shape)
    assert np.array_equal(got, oracle_decompress(oblob, np.float32, a.shape)[0])
    a4 = np.arange(5 * 6 * 7 * 8, dtype=np.float32).reshape(5, 6, 7, 8) * 0.01
    b4 = oracle_compress(a4, make_config(a4.shape, abs_eb=1e-3))
    got4, _ = sz3_amd.decompress(b4, np.float32, a4.shape)
    assert np.array_equal(got4, oracle_decompress(b4, np.float32, a4.shape)[0])
    with pytest.raises(sz3_amd.SZ3HipError):
        sz3_amd.decompress(oracle_compress(a4, make_config(a4.shape, abs_eb=1e-3, block_size=7)), np.float32, a4.shape)


def test_sz3c_abi_roundtrip():
    """tools/sz3c/include/sz3c.h:52-59 semantics through libsz3hip.so: r1 fastest, malloc'ed results, free_buf"""
    L = sz3_amd.lib()
    a = field3d((20, 30, 40))
    n = C.c_size_t(0)
    p = L.SZ_compress_args(0, a.ctypes.data, C.byref(n), 0, 1e-3, 0.0, 0.0, 0, 0, 20, 30, 40)
    assert p and 0 < n.value < a.nbytes
    q = L.SZ_decompress(0, p, n.value, 0, 0, 20, 30, 40)
    dec = np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_float)), shape=(a.size,)).copy().reshape(a.shape)
    L.free_buf(p)
    L.free_buf(q)
    assert np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) <= 1e-3
    b = field2d((64, 80), np.float64)
    p = L.SZ_compress_args(1, b.ctypes.data, C.byref(n), 1, 0.0, 1e-4, 0.0, 0, 0, 0, 64, 80)   # REL
    q = L.SZ_decompress(1, p, n.value, 0, 0, 0, 64, 80)
    dec = np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_double)), shape=(b.size,)).copy().reshape(b.shape)
    L.free_buf(p)
    L.free_buf(q)
    assert np.max(np.abs(dec - b)) <= 1e-4 * (b.max() - b.min())


@pytest.mark.parametrize("shape,dtype,eb", [((512, 512, 512), np.float32, 1e-3), ((256, 256, 256), np.float64, 1e-6),
                                            ((128, 1024, 1024), np.float64, 1e-6)],
                         ids=["C2-512c-f32", "C4-256c-f64", "C4-slab-128x1024x1024-f64"])
def test_full_size_properties(shape, dtype, eb):
    """size-independent properties at the benchmark size (no oracle: it would take minutes): strict bound after a
    device round trip, idempotence (re-compressing the decompressed field reproduces it bit for bit — the lattice is a
    fixed point), and determinism of the payload."""
    dev = torch.device("cuda:0")
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    g = torch.Generator(device=dev).manual_seed(1234)
    z, y, x = torch.meshgrid(*[torch.arange(s, device=dev, dtype=torch.float64) for s in shape], indexing="ij")
    f = torch.sin(2 * np.pi * x / 64) * torch.cos(2 * np.pi * y / 96) * torch.sin(2 * np.pi * z / 128) + \
        0.25 * torch.sin(2 * np.pi * (x + 2 * y + 3 * z) / 37)
    del x, y, z
    f = (f + (2e-3 if dtype == np.float32 else 2e-6) * torch.randn(shape, device=dev, dtype=torch.float64, generator=g)).to(tdt)
    n = f.numel()
    dc = sz3_amd.DeviceCompressor(n, dtype)
    cap = dc.payload_bound(n)
    pl1 = torch.empty(cap, dtype=torch.uint8, device=dev)
    pl2 = torch.empty(cap, dtype=torch.uint8, device=dev)
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.regression = 0  # (idempotence is a property of the lattice: regression blocks reconstruct off it)
    conf.absErrorBound = eb
    s = torch.cuda.current_stream().cuda_stream
    sz1 = dc.compress(conf, f.data_ptr(), pl1.data_ptr(), cap, s)
    out = torch.empty_like(f)
    dc.decompress(pl1.data_ptr(), sz1, out.data_ptr(), s)
    torch.cuda.synchronize()
    assert float((out.double() - f.double()).abs().max()) <= eb
    assert f.element_size() * n / sz1 > 3
    sz2 = dc.compress(conf, out.data_ptr(), pl2.data_ptr(), cap, s)
    out2 = torch.empty_like(f)
    dc.decompress(pl2.data_ptr(), sz2, out2.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(out2, out), "decompress(compress(x^)) != x^"
    sz3 = dc.compress(conf, f.data_ptr(), pl2.data_ptr(), cap, s)
    torch.cuda.synchronize()
    st = dc.stats()
    if max(st["n_value_outliers"], st["n_delta_outliers"]) <= 65536:   # longer outlier lists stay in arrival order
        assert sz3 == sz1 and torch.equal(pl1[:sz1], pl2[:sz1]), "payload is not deterministic"


def test_c4_slab_with_its_specified_predictor_set():
    """BASELINE config C4's per-GPU slab at full size — float64 128 x 1024 x 1024, abs 1e-6, Lorenzo + regression chosen per
    block (the block-composed stream) — through size-independent properties: strict bound after a device round trip,
    deterministic payload, the selection vector in the stream names only the enabled predictors and both occur on the C4a
    field, device payload (before zstd) within 6 % of the plain Lorenzo stream's on the same slab (measured 4 %:
    the side section — selection bits and coefficients of 14 % of the blocks — and the block-major code order; with zstd and on
    the oracle's side the two are level, tests/test_gpu_regression.py)."""
    dev = torch.device("cuda:0")
    shape, eb = (128, 1024, 1024), 1e-6
    g = torch.Generator(device=dev).manual_seed(99)
    z, y, x = torch.meshgrid(*[torch.arange(s, device=dev, dtype=torch.float64) for s in shape], indexing="ij")
    f = torch.sin(2 * np.pi * x / 64) * torch.cos(2 * np.pi * y / 96) * torch.sin(2 * np.pi * z / 128) + \
        0.25 * torch.sin(2 * np.pi * (x + 2 * y + 3 * z) / 37)
    del x, y, z
    f = 3.3e-5 * (f + 2e-3 * torch.randn(shape, device=dev, dtype=torch.float64, generator=g))  # C4a: both predictors are chosen
    n = f.numel()
    dc = sz3_amd.DeviceCompressor(n, np.float64)
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, 1
    conf.absErrorBound = eb
    cap = dc.payload_bound_conf(conf)
    pl1 = torch.empty(cap, dtype=torch.uint8, device=dev)
    pl2 = torch.empty(cap, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    sz1 = dc.compress(conf, f.data_ptr(), pl1.data_ptr(), cap, s)
    out = torch.empty_like(f)
    dc.decompress(pl1.data_ptr(), sz1, out.data_ptr(), s)
    torch.cuda.synchronize()
    assert float((out - f).abs().max()) <= eb
    sz2 = dc.compress(conf, f.data_ptr(), pl2.data_ptr(), cap, s)
    torch.cuda.synchronize()
    assert sz2 == sz1 and torch.equal(pl1[:sz1], pl2[:sz1]), "payload is not deterministic"
    import szh_ref
    h, o, sec = szh_ref.parse(pl1[:sz1].cpu().numpy().tobytes())
    assert h["predictor"] == 2 and h["blk_edge"] == 6 and h["blk_mask"] == 5 and h["n"] == n
    sel, _ = szh_ref.parse_side(h, sec)
    share = float((sel == 2).mean())
    assert set(np.unique(sel)) <= {0, 2} and 0.02 < share < 0.5, share
    conf.regression = 0
    szl = dc.compress(conf, f.data_ptr(), pl2.data_ptr(), cap, s)
    torch.cuda.synchronize()
    print("C4 slab, composed: ratio %.2f, regression share %.3f; plain Lorenzo ratio %.2f" % (8.0 * n / sz1, share, 8.0 * n / szl))
    assert sz1 <= 1.06 * szl


def test_histogram_split_path_equals_single_call():
    """stage1 / (all-reduce placeholder) / stage2 with a caller-owned histogram == the fused call"""
    dev = torch.device("cuda:0")
    a = field3d((64, 64, 128))
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, np.float32)
    cap = dc.payload_bound(a.size)
    p1 = torch.empty(cap, dtype=torch.uint8, device=dev)
    p2 = torch.empty(cap, dtype=torch.uint8, device=dev)
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.regression = 0
    conf.absErrorBound = 1e-3
    s = torch.cuda.current_stream().cuda_stream
    n1 = dc.compress(conf, t.data_ptr(), p1.data_ptr(), cap, s)
    narrow = bool(dc.stats()["narrow_codes"])
    hist = torch.zeros(65536, dtype=torch.int64, device=dev)
    dc.set_histogram(hist.data_ptr())
    dc.stage1(conf, t.data_ptr(), s)
    torch.cuda.synchronize()
    h = hist.cpu().numpy()
    assert h.sum() == a.size
    import szh_ref
    assert np.array_equal(h, np.bincount(szh_ref.dualquant(a, 1e-3, narrow=narrow)[2].reshape(-1), minlength=65536))
    dc.stage2(p2.data_ptr(), cap, s)
    n2 = dc.finish(s)
    assert n1 == n2 and torch.equal(p1[:n1], p2[:n2])
    # a summed histogram of 2 identical slabs gives the same code lengths -> same payload (what RCCL would deliver)
    dc.stage1(conf, t.data_ptr(), s)
    hist *= 2
    dc.stage2(p2.data_ptr(), cap, s)
    n3 = dc.finish(s)
    out = torch.empty_like(t)
    dc.decompress(p2.data_ptr(), n3, out.data_ptr(), s)
    torch.cuda.synchronize()
    assert float((out.double() - t.double()).abs().max()) <= 1e-3


def test_unmodified_reference_cli_runs_on_the_gpu_path(tmp_path):
    """oracle/_ref/sz3_hip: the reference's CLI source (tools/sz3/sz3.cpp, unmodified, compiled where it lies) built
    against include/SZ3/api/sz.hpp + libsz3hip.so. Its own round-trip report must show the bound, and the stream
    it wrote must decode through the Python binding too (tools/sz3/sz3.cpp:130-190)."""
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "sz3_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/sz3_hip not built (needs /root/reference at build time)")
    a = field3d((40, 50, 60))
    src, cmp_, dec = tmp_path / "a.f32", tmp_path / "a.sz", tmp_path / "a.out"
    a.tofile(src)
    for algo_ini, want in (("ALGO_LORENZO_REG", sz3_amd.ALGO_HIP_LORENZO), ("ALGO_INTERP", sz3_amd.ALGO_HIP_INTERP)):
        ini = tmp_path / "c.ini"
        ini.write_text("[GlobalSettings]\nCmprAlgo = %s\n" % algo_ini)
        r = subprocess.run([exe, "-f", "-i", str(src), "-z", str(cmp_), "-o", str(dec), "-3", "60", "50", "40", "-c", str(ini),
                            "-M", "ABS", "1e-3", "-a"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        m = re.search(r"Max absolute error = ([0-9.eE+-]+)", r.stdout)
        assert m and float(m.group(1)) <= 1e-3, r.stdout
        out = np.fromfile(dec, dtype=np.float32).reshape(a.shape)
        assert np.max(np.abs(out.astype(np.float64) - a.astype(np.float64))) <= 1e-3
        blob = np.fromfile(cmp_, dtype=np.uint8)
        d2, c2 = sz3_amd.decompress(blob, np.float32, a.shape)
        assert c2.cmprAlgo == want and np.array_equal(d2, out)
    # integer input through the CLI's -I 32 (SZ_compress<int32_t>)
    ai = (1000 * a).astype(np.int32)
    isrc, icmp, idec = tmp_path / "a.i32", tmp_path / "ai.sz", tmp_path / "ai.out"
    ai.tofile(isrc)
    r = subprocess.run([exe, "-I", "32", "-i", str(isrc), "-z", str(icmp), "-o", str(idec), "-3", "60", "50", "40", "-M", "ABS", "3", "-a"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    outi = np.fromfile(idec, dtype=np.int32).reshape(a.shape)
    assert np.max(np.abs(outi.astype(np.int64) - ai.astype(np.int64))) <= 3


def test_reference_tree_with_the_hip_algorithm_added(tmp_path):
    """oracle/_ref/sz3_algohip: the reference's CLI over the reference's OWN headers with the GPU path added as one more
    ALGO by the reference's documented recipe (include/SZ3/api/impl/SZAlgoHip.hpp + three edits, oracle/algohip_patch.py).
    Round trip through the CLI, and the file opens through this repository's faces too (same container, same ids)."""
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "sz3_algohip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/sz3_algohip not built (needs /root/reference at build time)")
    a = field3d((40, 50, 60))
    src, cmp_, dec = tmp_path / "a.f32", tmp_path / "a.sz", tmp_path / "a.out"
    a.tofile(src)
    for algo_ini, want in (("ALGO_HIP_LORENZO", sz3_amd.ALGO_HIP_LORENZO), ("ALGO_HIP_INTERP", sz3_amd.ALGO_HIP_INTERP)):
        ini = tmp_path / "c.ini"
        ini.write_text("[GlobalSettings]\nCmprAlgo = %s\n" % algo_ini)
        r = subprocess.run([exe, "-f", "-i", str(src), "-z", str(cmp_), "-o", str(dec), "-3", "60", "50", "40", "-c", str(ini),
                            "-M", "ABS", "1e-3", "-a"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (r.stdout, r.stderr)
        m = re.search(r"Max absolute error = ([0-9.eE+-]+)", r.stdout)
        assert m and float(m.group(1)) <= 1e-3, r.stdout
        out = np.fromfile(dec, dtype=np.float32).reshape(a.shape)
        assert np.max(np.abs(out.astype(np.float64) - a.astype(np.float64))) <= 1e-3
        d2, c2 = sz3_amd.decompress(np.fromfile(cmp_, dtype=np.uint8), np.float32, a.shape)
        assert c2.cmprAlgo == want and np.array_equal(d2, out)
    # integer data through the same recipe (the reference's CLI: -I 32): the blob records int32, SZ_compress_Hip hands that back
    # into the Config the trailer is written from, and the decoder takes the caller's type
    ai = np.round(a * 1000).astype(np.int32)
    isrc, idec = tmp_path / "a.i32", tmp_path / "a.i32.out"
    ai.tofile(isrc)
    ini.write_text("[GlobalSettings]\nCmprAlgo = ALGO_HIP_LORENZO\n")
    r = subprocess.run([exe, "-I", "32", "-i", str(isrc), "-z", str(cmp_), "-o", str(idec), "-3", "60", "50", "40", "-c", str(ini),
                        "-M", "ABS", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    iout = np.fromfile(idec, dtype=np.int32).reshape(a.shape)
    assert np.max(np.abs(iout.astype(np.int64) - ai.astype(np.int64))) <= 2
    # an algorithm of the reference itself still runs on its CPU path in the same binary (the recipe adds, it does not replace)
    ini.write_text("[GlobalSettings]\nCmprAlgo = ALGO_LORENZO_REG\n")
    r = subprocess.run([exe, "-f", "-i", str(src), "-z", str(cmp_), "-o", str(dec), "-3", "60", "50", "40", "-c", str(ini), "-M", "ABS", "1e-3", "-a"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and float(re.search(r"Max absolute error = ([0-9.eE+-]+)", r.stdout).group(1)) <= 1e-3


def test_c1_through_the_cli_boundary(tmp_path):
    """BASELINE.json configs[0] (C1): 1-D float32, 2^20 values (the first 2^20 of the C2 field), ALGO_LORENZO_REG, abs 1e-3,
    "via tools/sz3 CLI". The configuration is about the CLI boundary, so it goes through it three ways:
      sz3_hip       the unmodified CLI source over THIS repository's headers          -> GPU Lorenzo stream
      sz3_algohip   the CLI over the reference's tree + SZAlgoHip.hpp, ALGO_HIP_LORENZO -> the same GPU path
      sz3_algohip   ... with the reference's own ALGO_LORENZO_REG                      -> the reference's CPU path (stock stream)
    Every reconstruction within the bound, GPU and CPU reconstructions within 2 eb of each other, ratios comparable. (Stock
    CPU streams are opened by stock SZ3 — or by this very binary; the GPU library has no CPU decoder, by design.)"""
    import re
    import subprocess
    from fields import field1d
    ref_dir = os.path.join(os.path.dirname(HERE), "oracle", "_ref")
    exes = {k: os.path.join(ref_dir, k) for k in ("sz3_hip", "sz3_algohip")}
    if not all(os.path.exists(e) for e in exes.values()):
        pytest.skip("oracle/_ref CLIs not built (need /root/reference at build time)")
    a = field1d(1 << 20)
    src = tmp_path / "c1.f32"
    a.tofile(src)
    runs = {}
    for name, exe, algo in (("ours", "sz3_hip", "ALGO_LORENZO_REG"), ("recipe-gpu", "sz3_algohip", "ALGO_HIP_LORENZO"), ("recipe-cpu", "sz3_algohip", "ALGO_LORENZO_REG")):
        ini = tmp_path / (name + ".ini")
        ini.write_text("[GlobalSettings]\nCmprAlgo = %s\n" % algo)
        cmp_, dec = tmp_path / (name + ".sz"), tmp_path / (name + ".out")
        r = subprocess.run([exes[exe], "-f", "-i", str(src), "-z", str(cmp_), "-o", str(dec), "-1", str(1 << 20), "-c", str(ini), "-M", "ABS", "1e-3", "-a"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (name, r.stdout, r.stderr)
        assert float(re.search(r"Max absolute error = ([0-9.eE+-]+)", r.stdout).group(1)) <= 1e-3
        out = np.fromfile(dec, dtype=np.float32)
        assert np.max(np.abs(out.astype(np.float64) - a.astype(np.float64))) <= 1e-3
        runs[name] = (out, a.nbytes / os.path.getsize(cmp_))
    assert np.array_equal(runs["ours"][0], runs["recipe-gpu"][0])  # the same library under both faces
    assert np.max(np.abs(runs["ours"][0].astype(np.float64) - runs["recipe-cpu"][0].astype(np.float64))) <= 2e-3
    assert runs["ours"][1] >= 0.9 * runs["recipe-cpu"][1], {k: v[1] for k, v in runs.items()}
    # the file the stock code path wrote (ALGO_LORENZO_REG, 1-D: Lorenzo + regression in blocks of 128) is read by this library since
    # round 4, and to the values the stock CLI itself decoded
    got, _ = sz3_amd.decompress(np.fromfile(tmp_path / "recipe-cpu.sz", dtype=np.uint8), np.float32, a.shape)
    assert np.array_equal(got, runs["recipe-cpu"][0])


@pytest.mark.parametrize("shape,carry", [((96, 384, 512), 0), ((80, 520, 500), 1), ((48, 130, 768), 1), ((40, 36, 1024), 2), ((160, 200, 256), 0)],
                         ids=["rows-512", "rows-500-carry-pass", "rows-768-carry-pass", "rows-1024-carried-in-scan", "rows-256"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32-int16", "f64-int32"])
def test_half_width_decoder_intermediates_and_their_overflow_path(shape, carry, dtype):
    """The Lorenzo decoder keeps its x-scanned values and the y-prefixed ones as int16 (f64 data: int32) when they fit (smooth fields) and
    repeats the chain at full width behind a device-side gate when one does not. A step of 2000 between two planes makes
    D_z q = 10^6 lattice steps: the first call overflows and takes the gated chain, the following ones go to full width
    directly (the context fetched the flag with the next header); a smooth field stays on the half-width chain. All exact
    against each other and within the bound. A lane of the decoder sums a unit of 512 symbols: rows of 256 / 512 need nothing
    more, rows of 1024 get the sum of their first unit from the first strided scan as it reads (mode 2), rows of 500 / 768 from
    a pass of their own (mode 1) — on either chain; the pass and the in-scan form agree bit for bit."""
    import ctypes as C
    dev = torch.device("cuda:0")
    eb = 1e-3
    smooth = field3d(shape, dtype)
    step = smooth.copy()
    step[shape[0] // 2:] += 2000.0 if dtype == np.float32 else 5e6  # (f64: int32 intermediates, D_z q = 2.5e9 lattice steps)
    info = (C.c_uint32 * 4)()
    L = sz3_amd.lib()
    L.sz3hip_debug_decode_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    for a, overflows in ((smooth, False), (step, True)):
        t = torch.from_numpy(a).to(dev)
        dc = sz3_amd.DeviceCompressor(a.size, dtype)
        cap = dc.payload_bound(a.size, worst_case=True)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        conf = sz3_amd.Config(*shape)
        conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
        conf.regression = 0
        conf.absErrorBound = eb
        n = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
        outs, modes = [], []
        for _ in range(3):
            out = torch.empty_like(t)
            dc.decompress(pl.data_ptr(), n, out.data_ptr(), 0)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy())
            L.sz3hip_debug_decode_info(dc._h, info)
            modes.append((info[0], info[1]))
        assert modes[0] == (1, carry), "the first call tries the half-width chain"
        # (the overflow flag of a call is fetched with the header of the next one: the second call still tries, the third does not)
        assert modes[2] == ((0 if overflows else 1), carry)
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])
        assert float(np.max(np.abs(outs[0].astype(np.float64) - a.astype(np.float64)))) <= eb
        for flags, want in ((2097152, (0, carry)), (536870912, None), (2097152 | 536870912, (0, 1 if carry else 0))):
            L.sz3hip_debug_flags(flags)  # 2097152: full-width chain only; 536870912: the carries by their own pass
            try:
                ref = torch.empty_like(t)
                dc.decompress(pl.data_ptr(), n, ref.data_ptr(), 0)
                torch.cuda.synchronize()
                L.sz3hip_debug_decode_info(dc._h, info)
                if want is not None:
                    assert (info[0], info[1]) == want
            finally:
                L.sz3hip_debug_flags(0)
            assert np.array_equal(ref.cpu().numpy(), outs[0])


def test_full_size_interpolation_properties():
    """C3 at its full size (512^3 f32, ALGO_INTERP_LORENZO = tuner + interpolation, abs 1e-4): strict bound after a device
    round trip, payload determinism, and the tuner's report is the same on every run."""
    dev = torch.device("cuda:0")
    shape = (512, 512, 512)
    g = torch.Generator(device=dev).manual_seed(99)
    z, y, x = torch.meshgrid(*[torch.arange(s, device=dev, dtype=torch.float32) for s in shape], indexing="ij")
    f = torch.sin(2 * np.pi * x / 64) * torch.cos(2 * np.pi * y / 96) * torch.sin(2 * np.pi * z / 128) + \
        0.25 * torch.sin(2 * np.pi * (x + 2 * y + 3 * z) / 37)
    del x, y, z
    f = f + 2e-3 * torch.randn(shape, device=dev, dtype=torch.float32, generator=g)
    n = f.numel()
    dc = sz3_amd.DeviceCompressor(n, np.float32)
    cap = dc.payload_bound(n)
    pl1 = torch.empty(cap, dtype=torch.uint8, device=dev)
    pl2 = torch.empty(cap, dtype=torch.uint8, device=dev)
    conf = sz3_amd.Config(*shape)
    conf.absErrorBound = 1e-4
    s = torch.cuda.current_stream().cuda_stream
    sz1 = dc.compress(conf, f.data_ptr(), pl1.data_ptr(), cap, s)
    rep1 = dc.tuner_report()
    assert rep1["ran"] == 1 and rep1["use_interp"] == 1 and rep1["sample_block_size"] == 32 and rep1["n_blocks"] == 17
    out = torch.empty_like(f)
    dc.decompress(pl1.data_ptr(), sz1, out.data_ptr(), s)
    torch.cuda.synchronize()
    assert float((out.double() - f.double()).abs().max()) <= 1e-4
    assert 4.0 * n / sz1 > 3
    sz2 = dc.compress(conf, f.data_ptr(), pl2.data_ptr(), cap, s)
    torch.cuda.synchronize()
    assert dc.tuner_report() == rep1
    assert sz2 == sz1 and torch.equal(pl1[:sz1], pl2[:sz1]), "payload is not deterministic"


@pytest.mark.parametrize("algo", ["lorenzo", "default"])
def test_c5_shaped_4d_rel_roundtrip(algo):
    """C5's shape family (time x 3-D volume, REL 1e-3) through the host API at 12 x 128^3 (one rank's slab thickness)"""
    a = field4d((12, 128, 128, 128))
    conf = sz3_amd.Config(*a.shape)
    if algo == "lorenzo":
        conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.errorBoundMode = sz3_amd.EB_REL
    conf.relErrorBound = 1e-3
    blob, ratio = sz3_amd.compress(a, conf)
    dec, c2 = sz3_amd.decompress(blob, np.float32, a.shape)
    eb = 1e-3 * (float(a.max()) - float(a.min()))
    assert c2.absErrorBound == pytest.approx(eb, rel=1e-6)
    assert np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) <= c2.absErrorBound and ratio > 4


def test_integer_inputs_ride_the_f64_pipeline():
    """int32 / int64 arrays (tools/sz3/sz3.cpp:458-461 instantiates SZ_compress<int32_t/int64_t>): |x - x^| <= floor(eb)
    between integers; a bound below 1 and int64 magnitudes beyond 2^53 fall back to the lossless stream"""
    rng = np.random.default_rng(3)
    base = (1000 * field3d((40, 48, 56))).astype(np.int64) + rng.integers(-3, 4, (40, 48, 56))
    for dt in (np.int32, np.int64):
        a = base.astype(dt)
        for algo in (sz3_amd.ALGO_LORENZO_REG, sz3_amd.ALGO_INTERP):
            conf = sz3_amd.Config(*a.shape)
            conf.cmprAlgo = algo
            conf.absErrorBound = 4.7
            blob, ratio = sz3_amd.compress(a, conf)
            dec, c2 = sz3_amd.decompress(blob, dt, a.shape)
            assert dec.dtype == dt and c2.absErrorBound == 4.0 and c2.cmprAlgo in (sz3_amd.ALGO_HIP_LORENZO, sz3_amd.ALGO_HIP_INTERP)
            assert np.max(np.abs(dec.astype(np.int64) - a.astype(np.int64))) <= 4 and ratio > 2
        conf = sz3_amd.Config(*a.shape)
        conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
        conf.absErrorBound = 0.5                      # floor -> 0 -> lossless (SZDispatcher.hpp:19-21)
        blob, _ = sz3_amd.compress(a, conf)
        dec, c2 = sz3_amd.decompress(blob, dt, a.shape)
        assert c2.cmprAlgo == sz3_amd.ALGO_LOSSLESS and np.array_equal(dec, a)
        with pytest.raises(sz3_amd.SZ3HipError):
            sz3_amd.decompress(sz3_amd.compress(a, sz3_amd.Config(*a.shape))[0], np.float64, a.shape)
    big = base.copy()
    big[1, 2, 3] = (1 << 60) + 12345
    conf = sz3_amd.Config(*big.shape)
    conf.absErrorBound = 2.0
    blob, _ = sz3_amd.compress(big, conf)
    dec, c2 = sz3_amd.decompress(blob, np.int64, big.shape)
    assert c2.cmprAlgo == sz3_amd.ALGO_LOSSLESS and np.array_equal(dec, big)


def test_torch_can_start_after_the_library():
    """One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64, so a process that used this library BEFORE its
    first torch CUDA call used to end with "No HIP GPUs are available" (two runtimes). The binding loads the wheel's copy first
    when there is one (sz3_amd._share_torch_hip_runtime); a fresh process, library first, torch second."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, sz3_amd; a = np.random.rand(40, 40, 40).astype(np.float32); "
            "c = sz3_amd.Config(40, 40, 40); c.absErrorBound = 1e-3; b, r = sz3_amd.compress(a, c); d, _ = sz3_amd.decompress(b, np.float32, a.shape); "
            "assert abs(d - a).max() <= 1e-3; import torch; t = torch.ones(4, device='cuda:0'); print('both ok', float(t.sum()))") % os.path.dirname(HERE)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "both ok 4.0" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


def test_host_api_from_several_threads_at_once():
    """the host-buffer API leases a slot (context, staging buffers, stream) per call instead of taking a process-wide lock:
    four threads compress and decompress different arrays concurrently (the shape of an HDF5 chunk pipeline with worker
    threads); every result equals what the same call gives alone"""
    import threading
    arrays = [field3d((40 + 3 * k, 48, 64), np.float32 if k % 2 == 0 else np.float64, seed=100 + k) for k in range(4)]
    ebs = [1e-3, 1e-6, 1e-2, 1e-4]

    def conf_for(a, eb, algo):
        c = sz3_amd.Config(*a.shape)
        c.cmprAlgo = algo
        c.regression = 0
        c.absErrorBound = eb
        return c
    algos = [sz3_amd.ALGO_LORENZO_REG, sz3_amd.ALGO_INTERP_LORENZO, sz3_amd.ALGO_INTERP, sz3_amd.ALGO_LORENZO_REG]
    alone = [sz3_amd.compress(a, conf_for(a, eb, al))[0].tobytes() for a, eb, al in zip(arrays, ebs, algos)]
    out, errs = [None] * 4, []

    def worker(k):
        try:
            for _ in range(6):
                blob, _ = sz3_amd.compress(arrays[k], conf_for(arrays[k], ebs[k], algos[k]))
                dec, _ = sz3_amd.decompress(blob, arrays[k].dtype, arrays[k].shape)
                assert float(np.max(np.abs(dec.astype(np.float64) - arrays[k].astype(np.float64)))) <= ebs[k]
                out[k] = blob.tobytes()
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert out == alone


def test_more_host_threads_than_slots_wait_for_a_lease():
    """the slot pool is capped (SZ3HIP_HOST_SLOTS, default 4 per device and element type: a slot keeps its context and buffers for the
    life of the process): ten threads of f32 callers share the four slots — every call completes, every result is the lone call's"""
    import threading
    a = field3d((32, 48, 64), np.float32, seed=77)
    c = sz3_amd.Config(*a.shape)
    c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    c.regression = 0
    c.absErrorBound = 1e-3
    alone = sz3_amd.compress(a, c)[0].tobytes()
    errs, outs = [], [None] * 10

    def worker(k):
        try:
            cc = sz3_amd.Config(*a.shape)
            cc.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
            cc.regression = 0
            cc.absErrorBound = 1e-3
            for _ in range(5):
                blob, _ = sz3_amd.compress(a, cc)
                dec, _ = sz3_amd.decompress(blob, np.float32, a.shape)
                assert float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64)))) <= 1e-3
                outs[k] = blob.tobytes()
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(10)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th), "a caller never got a lease"
    assert not errs, errs
    assert all(o == alone for o in outs)
