"""GPU tests (-m gpu) of stock-stream interoperability: stock SZ3 streams of the interpolation compressor
(cmprAlgo ALGO_INTERP — what the reference's default ALGO_INTERP_LORENZO writes) READ by this library, and streams WRITTEN by this
library (sz3hip_set_stock_format) read by the reference. The stock side is played by the oracle — byte-identical to the reference
built in this image (tests/test_oracle.py) — and, where oracle/_ref is present, by the reference library itself. Reconstruction is
the reference's bit for bit in both directions (prediction, quantisation and reconstruction are the same arithmetic).
Since the end of round 5 the writers build their Huffman trees with the reference's own queue (stock::build_tree, ref_heap): wherever the
codes are the reference's — ALGO_INTERP, ALGO_NOPRED, the default algorithm with SZ3HIP_TUNER_EXACT=1, ALGO_LORENZO_REG (round 6: every predictor set,
the block choices repeated against the coded array until they stand) — a written container IS the reference's file, byte for byte (one zstd frame: buffers up to 1 MB, or
SZ3HIP_STOCK_ONE_FRAME=1)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import sz3_amd  # noqa: E402
from fields import field1d, field2d, field3d, field4d  # noqa: E402
from oracle_binding import (ALGO_INTERP, ALGO_INTERP_LORENZO, EB_REL, have_ref, make_config, oracle_compress, oracle_decompress,  # noqa: E402
                            ref_compress, ref_decompress)

CASES = [
    ("3d-cubic", lambda: field3d((64, 80, 96)), 1e-3, dict(interp_algo=1)),
    ("3d-linear-dir3", lambda: field3d((40, 66, 36)), 1e-2, dict(interp_algo=0, interpDirection=3)),
    ("3d-ragged-anchor8", lambda: field3d((33, 47, 50)), 1e-3, dict(interp_algo=1, interpAnchorStride=8, interpAlpha=1.5, interpBeta=3.0)),
    ("3d-noanchor", lambda: field3d((20, 21, 22)), 1e-3, dict(interp_algo=1, interpAnchorStride=0, interpDirection=5)),
    ("3d-f64", lambda: field3d((40, 50, 37), np.float64, sigma=2e-6), 1e-6, dict(interp_algo=1)),
    ("3d-nan-inf", lambda: _with_holes(field3d((30, 40, 50))), 1e-3, dict(interp_algo=1)),
    ("1d", lambda: field1d(70001), 1e-3, dict(interp_algo=1)),
    ("2d-linear", lambda: field2d((300, 500)), 1e-2, dict(interp_algo=0, interpDirection=1)),
    ("4d", lambda: field4d((7, 20, 24, 28)), 1e-2, dict(interp_algo=1, interpDirection=11)),
]


def _with_holes(a):
    a = a.copy()
    f = a.reshape(-1)
    f[np.random.default_rng(5).choice(f.size, 500, replace=False)] = np.nan
    f[7] = np.inf
    return a


def _trailer_algo(blob):
    """cmprAlgo in the Config trailer (utils/Config.hpp:312-354) via the library's own peek"""
    import ctypes as C
    c = sz3_amd.Config(1)
    b = np.ascontiguousarray(np.frombuffer(bytes(blob), dtype=np.uint8))
    assert sz3_amd.lib().sz3hip_peek_config(C.byref(c._c), b.ctypes.data, b.size) == 0
    return c.cmprAlgo


@pytest.mark.parametrize("name,gen,eb,kw", CASES, ids=[c[0] for c in CASES])
def test_stock_interp_streams_are_read_bit_for_bit(name, gen, eb, kw):
    a = gen()
    oconf = make_config(a.shape, algo=ALGO_INTERP, abs_eb=eb, **kw)
    blob = oracle_compress(a, oconf)                       # = the reference's bytes (tests/test_oracle.py)
    want, _ = oracle_decompress(blob, a.dtype, a.shape)
    assert _trailer_algo(blob) == sz3_amd.ALGO_INTERP
    got, c2 = sz3_amd.decompress(blob, a.dtype, a.shape)
    assert c2.cmprAlgo == sz3_amd.ALGO_INTERP
    assert np.array_equal(got, want, equal_nan=True), "reconstruction differs from stock SZ3's"
    if have_ref():
        rblob = ref_compress(a, oconf)
        got2, _ = sz3_amd.decompress(rblob, a.dtype, a.shape)
        assert np.array_equal(got2, ref_decompress(rblob, a.dtype, a.shape), equal_nan=True)


@pytest.mark.parametrize("name,gen,eb,kw", CASES, ids=[c[0] for c in CASES])
def test_streams_written_in_stock_format_are_read_by_stock_sz3(name, gen, eb, kw):
    a = gen()
    L = sz3_amd.lib()
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_INTERP
    conf.absErrorBound = eb
    conf.regression = 0                                     # (make_config's: the Config trailer holds the flags whatever the algorithm)
    conf.interpAlgo = kw.get("interp_algo", 1)
    for k in ("interpDirection", "interpAnchorStride", "interpAlpha", "interpBeta"):
        if k in kw:
            setattr(conf, k, kw[k])
    L.sz3hip_set_stock_format(1)
    try:
        blob, ratio = sz3_amd.compress(a, conf)
    finally:
        L.sz3hip_set_stock_format(0)
    assert _trailer_algo(blob) == sz3_amd.ALGO_INTERP      # a stock id, not 17
    oconf = make_config(a.shape, algo=ALGO_INTERP, abs_eb=eb, **kw)
    oblob = oracle_compress(a, oconf)
    want, _ = oracle_decompress(oblob, a.dtype, a.shape)   # what stock SZ3 reconstructs from its own stream
    got, _ = oracle_decompress(blob, a.dtype, a.shape)     # stock SZ3 reading OUR stream
    assert np.array_equal(got, want, equal_nan=True)
    if have_ref():
        assert np.array_equal(ref_decompress(blob, a.dtype, a.shape), want, equal_nan=True)
    mine, _ = sz3_amd.decompress(blob, a.dtype, a.shape)   # and this library reading it back
    assert np.array_equal(mine, want, equal_nan=True)
    # the same codes in the same order, the tree from the reference's own queue, one zstd frame: the reference's file
    assert blob.tobytes() == oblob.tobytes(), (len(blob), len(oblob))


def test_default_algorithm_in_stock_format_and_ids_without_the_switch():
    """the reference's default (ALGO_INTERP_LORENZO: tuner, then interpolation) written in stock format: the tuner runs on the GPU, its
    outcome goes into the stream's decomposition header, stock SZ3 reads it; REL bound through the range scan. Without the switch the
    same call writes this library's id 17."""
    a = field3d((72, 80, 88))
    L = sz3_amd.lib()
    conf = sz3_amd.Config(*a.shape)
    conf.errorBoundMode = sz3_amd.EB_REL
    conf.relErrorBound = 1e-3
    assert conf.cmprAlgo == sz3_amd.ALGO_INTERP_LORENZO
    own, _ = sz3_amd.compress(a, conf)
    assert _trailer_algo(own) == sz3_amd.ALGO_HIP_INTERP
    L.sz3hip_set_stock_format(1)
    try:
        blob, ratio = sz3_amd.compress(a, conf)
    finally:
        L.sz3hip_set_stock_format(0)
    assert _trailer_algo(blob) == sz3_amd.ALGO_INTERP
    dec, oc = oracle_decompress(blob, a.dtype, a.shape)
    eb = 1e-3 * (float(a.max()) - float(a.min()))
    assert float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64)))) <= eb * (1 + 1e-12)
    mine, _ = sz3_amd.decompress(blob, a.dtype, a.shape)
    ref_own, _ = sz3_amd.decompress(own, a.dtype, a.shape)
    assert np.array_equal(mine, dec) and np.array_equal(mine, ref_own)  # one reconstruction, three containers
    oblob = oracle_compress(a, make_config(a.shape, algo=ALGO_INTERP_LORENZO, eb_mode=EB_REL, rel_eb=1e-3))
    assert len(blob) <= 1.05 * len(oblob)  # (the GPU tuner may choose a neighbouring (alpha, beta): DESIGN.md section 2)


@pytest.mark.parametrize("name,gen,kwc", [("abs-3d", lambda: field3d((72, 80, 88)), dict(abs_eb=3e-2)), ("rel-3d", lambda: field3d((72, 80, 88)), dict(eb_mode=EB_REL, rel_eb=1e-3)),
                                          ("abs-2d", lambda: field2d((600, 700)), dict(abs_eb=1e-3)), ("abs-4d", lambda: field4d((12, 40, 40, 40)), dict(abs_eb=1e-2)),
                                          ("abs-1d", lambda: field1d(1 << 18), dict(abs_eb=1e-3))],  # (1-D: the tuner takes Lorenzo-1 + Lorenzo-2 in blocks of 128)
                         ids=["abs-3d", "rel-3d", "abs-2d", "abs-4d", "abs-1d"])
def test_default_algorithm_in_stock_format_with_exact_pricing_is_the_reference_s_file(name, gen, kwc, monkeypatch):
    """the reference's default algorithm end to end: its tuner's decisions (SZ3HIP_TUNER_EXACT=1: the trials priced the reference's way),
    its stream layout, its tree order, one zstd frame — the container equals the reference's byte for byte"""
    a = gen()
    conf = sz3_amd.Config(*a.shape)
    conf.regression = 0
    if "rel_eb" in kwc:
        conf.errorBoundMode = sz3_amd.EB_REL
        conf.relErrorBound = kwc["rel_eb"]
    else:
        conf.absErrorBound = kwc["abs_eb"]
    monkeypatch.setenv("SZ3HIP_TUNER_EXACT", "1")
    L = sz3_amd.lib()
    L.sz3hip_set_stock_format(1)
    try:
        blob, _ = sz3_amd.compress(a, conf)
    finally:
        L.sz3hip_set_stock_format(0)
    oblob = oracle_compress(a, make_config(a.shape, algo=ALGO_INTERP_LORENZO, **kwc))
    assert blob.tobytes() == oblob.tobytes(), (len(blob), len(oblob))


def test_corrupt_stock_streams_are_refused():
    a = field3d((24, 30, 36))
    oconf = make_config(a.shape, algo=ALGO_INTERP, abs_eb=1e-3)
    blob = bytearray(oracle_compress(a, oconf).tobytes())
    good, _ = sz3_amd.decompress(bytes(blob), a.dtype, a.shape)
    # flip a byte inside the zstd frame: the lossless stage or the container's checks must catch it, never a crash
    for at in (40, len(blob) // 2, len(blob) - 60):
        b = bytearray(blob)
        b[at] ^= 0x5A
        try:
            out, _ = sz3_amd.decompress(bytes(b), a.dtype, a.shape)
        except sz3_amd.SZ3HipError:
            continue
        assert out.shape == good.shape  # (a flipped bit that still parses decodes to SOMETHING of the right shape)


def test_files_interchange_between_the_two_clis(tmp_path):
    """A user with .sz archives: a file written by the stock CLI (oracle/_ref/sz3: the reference's tools/sz3/sz3.cpp over the
    reference's own headers, default algorithm) is decompressed by the CLI over THIS library (oracle/_ref/sz3_hip: the same source over
    include/SZ3/api/sz.hpp), and a file written by that one with SZ3HIP_STOCK_FORMAT=1 is decompressed by the stock CLI — to the same
    values, which are the values the stock CLI gets back from its own file."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    ref_dir = os.path.join(os.path.dirname(here), "oracle", "_ref")
    stock, ours = os.path.join(ref_dir, "sz3"), os.path.join(ref_dir, "sz3_hip")
    if not (os.path.exists(stock) and os.path.exists(ours)):
        pytest.skip("oracle/_ref CLIs not built (need /root/reference at build time)")
    a = field3d((60, 72, 84))
    src = tmp_path / "a.f32"
    a.tofile(src)
    dims = ["-3", "84", "72", "60"]

    def run(exe, args, env=None):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=dict(os.environ, **(env or {})))
        assert r.returncode == 0, (exe, r.stdout[-800:], r.stderr[-800:])

    # stock writes, both read
    run(stock, ["-f", "-i", str(src), "-z", str(tmp_path / "stock.sz")] + dims + ["-M", "ABS", "1e-3"])
    run(stock, ["-f", "-z", str(tmp_path / "stock.sz"), "-o", str(tmp_path / "stock.by_stock")] + dims)
    run(ours, ["-f", "-z", str(tmp_path / "stock.sz"), "-o", str(tmp_path / "stock.by_ours")] + dims)
    want = np.fromfile(tmp_path / "stock.by_stock", dtype=np.float32)
    assert np.array_equal(np.fromfile(tmp_path / "stock.by_ours", dtype=np.float32), want)
    assert float(np.max(np.abs(want.astype(np.float64) - a.reshape(-1).astype(np.float64)))) <= 1e-3
    # this library writes in stock format, both read
    run(ours, ["-f", "-i", str(src), "-z", str(tmp_path / "ours.sz")] + dims + ["-M", "ABS", "1e-3"], env={"SZ3HIP_STOCK_FORMAT": "1"})
    run(stock, ["-f", "-z", str(tmp_path / "ours.sz"), "-o", str(tmp_path / "ours.by_stock")] + dims)
    run(ours, ["-f", "-z", str(tmp_path / "ours.sz"), "-o", str(tmp_path / "ours.by_ours")] + dims)
    got = np.fromfile(tmp_path / "ours.by_stock", dtype=np.float32)
    assert np.array_equal(np.fromfile(tmp_path / "ours.by_ours", dtype=np.float32), got)
    assert float(np.max(np.abs(got.astype(np.float64) - a.reshape(-1).astype(np.float64)))) <= 1e-3
    # ... and the two .sz files are one and the same (end of round 5): the default algorithm's tuner priced the reference's way (the host API's
    # default), the reference's stream layout and tree order, one zstd frame — also under a relative bound and for a double-precision array
    assert (tmp_path / "ours.sz").read_bytes() == (tmp_path / "stock.sz").read_bytes(), "the CLI over this library writes another file than the stock CLI"
    assert np.array_equal(got, want)
    run(stock, ["-f", "-i", str(src), "-z", str(tmp_path / "stock_rel.sz")] + dims + ["-M", "REL", "1e-4"])
    run(ours, ["-f", "-i", str(src), "-z", str(tmp_path / "ours_rel.sz")] + dims + ["-M", "REL", "1e-4"], env={"SZ3HIP_STOCK_FORMAT": "1"})
    assert (tmp_path / "ours_rel.sz").read_bytes() == (tmp_path / "stock_rel.sz").read_bytes()
    d = field3d((40, 50, 60), np.float64)
    d.tofile(tmp_path / "d.f64")
    dd = ["-3", "60", "50", "40"]
    run(stock, ["-d", "-i", str(tmp_path / "d.f64"), "-z", str(tmp_path / "stock_d.sz")] + dd + ["-M", "ABS", "1e-4"])
    run(ours, ["-d", "-i", str(tmp_path / "d.f64"), "-z", str(tmp_path / "ours_d.sz")] + dd + ["-M", "ABS", "1e-4"], env={"SZ3HIP_STOCK_FORMAT": "1"})
    assert (tmp_path / "ours_d.sz").read_bytes() == (tmp_path / "stock_d.sz").read_bytes()


@pytest.mark.parametrize("what", ["noise", "periodic-below-spacing", "noise-tiny-bound"])
def test_the_two_clis_write_one_file_where_lossy_streams_are_not_worth_much(what, tmp_path):
    """Round 6 (tests/checks/wild_data_sweep.py): when a lossy stream gives way to the lossless one is the reference's rule alone — ZSTD_compressBound
    of the stream against the room the caller's buffer leaves (lossless/Lossless_zstd.hpp:29-37, SZDispatcher.hpp:44-74; the CLI allocates twice the
    array, tools/sz3/sz3.cpp:132). White noise (ratio < 3: the comparison with zstd alone), a periodic 1-D array at a bound below its values'
    spacing (nothing but unpredictable values — which zstd folds to a few hundred bytes: the writers used to hand such arrays to the lossless
    stream), noise at a bound that makes the Huffman tree outgrow the array (the reference falls back, this library used to fail)."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    ref_dir = os.path.join(os.path.dirname(here), "oracle", "_ref")
    stock, ours = os.path.join(ref_dir, "sz3"), os.path.join(ref_dir, "sz3_hip")
    if not (os.path.exists(stock) and os.path.exists(ours)):
        pytest.skip("oracle/_ref CLIs not built (need /root/reference at build time)")
    rng = np.random.default_rng(7)
    if what == "noise":
        a, dims, eb = rng.standard_normal((46, 44, 50)).astype(np.float32), ["-3", "50", "44", "46"], "1e-3"
    elif what == "periodic-below-spacing":
        a, dims, eb = (1000.0 * np.sin(2 * np.pi * np.arange(59359) / 13.0)).astype(np.float32), ["-1", "59359"], "6e-9"
    else:
        a, dims, eb = rng.standard_normal((24270,)).astype(np.float32), ["-1", "24270"], "1e-7"
    src = tmp_path / "a.f32"
    a.tofile(src)

    def run(exe, args, env=None):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=dict(os.environ, **(env or {})))
        assert r.returncode == 0, (exe, r.stdout[-800:], r.stderr[-800:])

    for algo in ("ALGO_INTERP_LORENZO", "ALGO_LORENZO_REG"):
        cfg = tmp_path / (algo + ".cfg")
        cfg.write_text("[GlobalSettings]\nCmprAlgo = %s\n" % algo)
        run(stock, ["-f", "-i", str(src), "-z", str(tmp_path / "stock.sz"), "-c", str(cfg)] + dims + ["-M", "ABS", eb])
        run(ours, ["-f", "-i", str(src), "-z", str(tmp_path / "ours.sz"), "-c", str(cfg)] + dims + ["-M", "ABS", eb], env={"SZ3HIP_STOCK_FORMAT": "1", "SZ3HIP_STOCK_ONE_FRAME": "1"})
        assert (tmp_path / "ours.sz").read_bytes() == (tmp_path / "stock.sz").read_bytes(), (what, algo)
        run(stock, ["-f", "-z", str(tmp_path / "ours.sz"), "-o", str(tmp_path / "back")] + dims)
        back = np.fromfile(tmp_path / "back", dtype=np.float32)
        assert float(np.max(np.abs(back.astype(np.float64) - a.reshape(-1).astype(np.float64)))) <= float(eb)


@pytest.mark.parametrize("eb", [1e-2, 1e-4], ids=["short-codes", "long-codes"])
def test_device_and_host_huffman_stages_agree(eb, monkeypatch):
    """The stock container's Huffman stage runs on the device (tiles coded in LDS and shifted into place; the decoder re-synchronises
    subsequence by subsequence, sz3hip_stock.hip) — its host twin (SZ3HIP_STOCK_HOST_HUFFMAN=1) must write the same bytes from the same
    codes and read the same codes from the same bytes. A field whose code words run beyond the decoder's 12-bit table at the tight bound."""
    a = field3d((96, 100, 104))
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_INTERP
    conf.absErrorBound = eb
    L = sz3_amd.lib()
    blobs, decs = {}, {}
    L.sz3hip_set_stock_format(1)
    try:
        for mode in ("0", "1"):
            monkeypatch.setenv("SZ3HIP_STOCK_HOST_HUFFMAN", mode)
            blobs[mode], _ = sz3_amd.compress(a, conf)
    finally:
        L.sz3hip_set_stock_format(0)
    assert blobs["0"].tobytes() == blobs["1"].tobytes()
    for mode in ("0", "1"):
        monkeypatch.setenv("SZ3HIP_STOCK_HOST_HUFFMAN", mode)
        decs[mode], _ = sz3_amd.decompress(blobs["0"], a.dtype, a.shape)
    assert np.array_equal(decs["0"], decs["1"])
    want, _ = oracle_decompress(blobs["0"], a.dtype, a.shape)
    assert np.array_equal(decs["0"], want)


# ---- stock ALGO_LORENZO_REG streams, read side (round 4: sz3hip_stock.hip k_slr_*) -------------------------------------------------
LR_CASES = [
    ("3d-defaults", lambda: field3d((40, 50, 66)), 1e-2, dict(lorenzo=True, regression=True)),
    ("3d-coarse-regression", lambda: field3d((37, 48, 50)), 1e-1, dict(lorenzo=True, regression=True)),
    ("3d-lorenzo-only", lambda: field3d((33, 47, 50)), 1e-3, dict(lorenzo=True, regression=False)),
    ("3d-all-three", lambda: field3d((30, 31, 44)), 2e-2, dict(lorenzo=True, lorenzo2=True, regression=True)),
    ("3d-second-order-only", lambda: field3d((20, 21, 22)), 1e-3, dict(lorenzo=False, lorenzo2=True, regression=False)),
    ("3d-regression-only-block5", lambda: field3d((23, 32, 15)), 5e-2, dict(lorenzo=False, regression=True, block_size=5)),
    ("3d-f64", lambda: field3d((40, 50, 37), np.float64, sigma=2e-6), 1e-6, dict(lorenzo=True, regression=True)),
    ("3d-nan-inf", lambda: _with_holes(field3d((30, 40, 50))), 1e-3, dict(lorenzo=True, regression=True)),
    ("2d-defaults", lambda: field2d((300, 500)), 1e-2, dict(lorenzo=True, regression=True)),
    ("2d-second-order", lambda: field2d((131, 77)), 1e-4, dict(lorenzo=True, lorenzo2=True, regression=False)),
    ("1d-tuner-set", lambda: field1d(70001), 1e-3, dict(lorenzo=True, lorenzo2=True, regression=False)),
    ("1d-defaults", lambda: field1d(30011), 1e-3, dict(lorenzo=True, regression=True)),
    ("1d-thin-last-block", lambda: field1d(128 * 300 + 1), 1e-2, dict(lorenzo=False, regression=True)),
    ("4d-defaults", lambda: field4d((7, 20, 24, 28)), 1e-2, dict(lorenzo=True, regression=True)),
    ("4d-lorenzo-only-ragged", lambda: field4d((5, 13, 14, 19)), 1e-3, dict(lorenzo=True, regression=False)),
    ("4d-coarse-regression-f64", lambda: field4d((6, 12, 18, 25), np.float64), 1e-1, dict(lorenzo=True, regression=True)),
]


@pytest.mark.parametrize("name,gen,eb,kw", LR_CASES, ids=[c[0] for c in LR_CASES])
def test_stock_lorenzo_reg_streams_are_read_bit_for_bit(name, gen, eb, kw):
    """the reference's arithmetic on the device: predictions from reconstructed values in T, in the reference's order of terms, and
    LinearQuantizer::recover — the values are the ones stock SZ3 decodes, bit for bit, for every predictor set"""
    a = gen()
    oconf = make_config(a.shape, abs_eb=eb, **kw)          # (cmprAlgo ALGO_LORENZO_REG)
    blob = oracle_compress(a, oconf)                       # = the reference's bytes (tests/test_oracle.py)
    want, _ = oracle_decompress(blob, a.dtype, a.shape)
    assert _trailer_algo(blob) == sz3_amd.ALGO_LORENZO_REG
    got, c2 = sz3_amd.decompress(blob, a.dtype, a.shape)
    assert c2.cmprAlgo == sz3_amd.ALGO_LORENZO_REG
    assert np.array_equal(got, want, equal_nan=True), "reconstruction differs from stock SZ3's"
    if have_ref():
        rblob = ref_compress(a, oconf)
        got2, _ = sz3_amd.decompress(rblob, a.dtype, a.shape)
        assert np.array_equal(got2, ref_decompress(rblob, a.dtype, a.shape), equal_nan=True)


def test_stock_default_algorithm_on_a_1d_array_is_read():
    """what stock SZ3 writes for a 1-D array with its default Config: the tuner's Lorenzo set (SZAlgoInterp.hpp:232-282)"""
    a = field1d(1 << 18)
    oconf = make_config(a.shape, algo=ALGO_INTERP_LORENZO, abs_eb=1e-3, regression=True)
    blob = oracle_compress(a, oconf)
    assert _trailer_algo(blob) == sz3_amd.ALGO_LORENZO_REG
    want, _ = oracle_decompress(blob, a.dtype, a.shape)
    got, c2 = sz3_amd.decompress(blob, a.dtype, a.shape)
    assert (c2.lorenzo, c2.lorenzo2, c2.regression) == (1, 1, 0) and np.array_equal(got, want)


def test_corrupt_stock_lorenzo_reg_streams_are_refused():
    a = field3d((24, 30, 36))
    blob = bytearray(oracle_compress(a, make_config(a.shape, abs_eb=1e-2, lorenzo=True, regression=True)).tobytes())
    for cut in (40, len(blob) // 2):
        with pytest.raises(sz3_amd.SZ3HipError):
            sz3_amd.decompress(bytes(blob[:cut]) + bytes(blob[-200:]), np.float32, a.shape)
    # a regression-only set whose last blocks are one element thin: the reference's fallback predictor then reads what lies in front of
    # the element in memory (no padding) — refused, not imitated
    at = field3d((23, 31, 16))
    bt = oracle_compress(at, make_config(at.shape, abs_eb=5e-2, lorenzo=False, regression=True, block_size=5))
    with pytest.raises(sz3_amd.SZ3HipError):
        sz3_amd.decompress(bt, np.float32, at.shape)
    a4 = field4d((5, 12, 14, 16))
    b4 = oracle_compress(a4, make_config(a4.shape, abs_eb=1e-2, lorenzo=True, regression=True, block_size=7))
    with pytest.raises(sz3_amd.SZ3HipError, match="block sizes up to 6"):
        sz3_amd.decompress(b4, np.float32, a4.shape)


# ---- stock ALGO_LORENZO_REG streams, WRITE side (round 5: sz3hip_stock.hip k_slw_*) ------------------------------------------------
LR_WRITE_CASES = LR_CASES + [  # (4-D arrays since round 5's second half: k_slw_select4 / k_slw_front4)
    ("4d-all-three-block4", lambda: field4d((9, 10, 11, 13)), 2e-2, dict(lorenzo=True, lorenzo2=True, regression=True, block_size=4)),
    ("4d-regression-only", lambda: field4d((6, 12, 18, 24)), 5e-2, dict(lorenzo=False, regression=True)),
    ("3d-block8", lambda: field3d((33, 40, 41)), 2e-2, dict(lorenzo=True, regression=True, block_size=8)),
    ("2d-f64-all-three", lambda: field2d((97, 130), np.float64), 1e-3, dict(lorenzo=True, lorenzo2=True, regression=True)),
    ("3d-512cube-slice", lambda: field3d((64, 256, 256)), 1e-3, dict(lorenzo=True, regression=True)),
    # (round 6, from the byte sweep: a block whose fit differs in the last place when index * value is a product in double instead of in T)
    ("1d-regression-only-t-products", lambda: field1d(368898), 0.00014524286814550258, dict(lorenzo=False, lorenzo2=False, regression=True)),
]


SEVERAL_FRAMES = {"3d-512cube-slice"}  # (a buffer beyond 1 MB leaves in several zstd frames by default: the one-frame test below has this array)


@pytest.mark.parametrize("name,gen,eb,kw", LR_WRITE_CASES, ids=[c[0] for c in LR_WRITE_CASES])
def test_streams_written_as_stock_lorenzo_reg_are_read_by_stock_sz3(name, gen, eb, kw):
    """sz3hip_set_stock_format(1) + cmprAlgo ALGO_LORENZO_REG: the reference's own Lorenzo / regression container
    (api/impl/SZAlgoLorenzoReg.hpp:67-84, decomposition/BlockwiseDecomposition.hpp:28-46, predictor/RegressionPredictor.hpp:94-149,
    quantizer/LinearQuantizer.hpp:43-71). Stock SZ3 (the oracle, the reference library) decodes it within the bound; this library's
    reader decodes it to the same values bit for bit; the size is near what stock SZ3 writes for the same Config (the writer chooses
    the blocks' predictors from original neighbours, the reference from reconstructed ones)."""
    a = gen()
    L = sz3_amd.lib()
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.absErrorBound = eb
    conf.lorenzo, conf.lorenzo2, conf.regression = int(kw.get("lorenzo", True)), int(kw.get("lorenzo2", False)), int(kw.get("regression", False))
    if "block_size" in kw:
        conf.blockSize = kw["block_size"]
    L.sz3hip_set_stock_format(1)
    try:
        blob, ratio = sz3_amd.compress(a, conf)
    finally:
        L.sz3hip_set_stock_format(0)
    assert _trailer_algo(blob) == sz3_amd.ALGO_LORENZO_REG, "not a stock stream"
    l2_in_4d = a.ndim == 4 and kw.get("lorenzo2", False)     # (the oracle restates no second-order member for N = 4; the reference's predicts 0)
    if l2_in_4d and not have_ref():
        pytest.skip("oracle/_ref/libsz3ref.so not built")
    got = ref_decompress(blob, a.dtype, a.shape) if l2_in_4d else oracle_decompress(blob, a.dtype, a.shape)[0]   # stock SZ3 reading OUR stream
    fin = np.isfinite(a)
    assert float(np.max(np.abs(got[fin].astype(np.float64) - a[fin].astype(np.float64)))) <= eb
    assert np.array_equal(got[~fin], a[~fin], equal_nan=True)
    if have_ref():
        assert np.array_equal(ref_decompress(blob, a.dtype, a.shape), got, equal_nan=True)
    mine, c2 = sz3_amd.decompress(blob, a.dtype, a.shape)    # and this library reading it back
    assert c2.cmprAlgo == sz3_amd.ALGO_LORENZO_REG
    assert np.array_equal(mine, got, equal_nan=True)
    if l2_in_4d:
        return
    oblob = oracle_compress(a, make_config(a.shape, abs_eb=eb, **kw))
    # (1-D: the chain is walked on the host in the reference's own order — the same choices, the same codes)
    assert len(blob) <= (1.01 if a.ndim == 1 else 1.08) * len(oblob) + 256, (len(blob), len(oblob))
    # The reference's file byte for byte wherever the codes are the reference's: sets of one member, 1-D arrays, and the arrays where the
    # writer's choices (from original neighbours) coincide with the reference's (from reconstructed ones) — tools/stock_bytes_lab.py
    # (round 6: every case — the selection is repeated against the coded array until it stands: the reference's own choices)
    if name not in SEVERAL_FRAMES:
        assert blob.tobytes() == oblob.tobytes(), (name, len(blob), len(oblob))
    if a.ndim == 1:
        want, _ = oracle_decompress(oblob, a.dtype, a.shape)
        assert np.array_equal(got, want), "a 1-D stream decodes to other values than stock SZ3's own stream"


def test_one_frame_switch_makes_a_large_stock_container_the_reference_s_file(monkeypatch):
    """a buffer beyond 1 MB leaves in several zstd frames by default (the pool's threads; stock SZ3 reads them) — SZ3HIP_STOCK_ONE_FRAME=1
    writes the one frame ZSTD_compress writes: the reference's file (64 x 256^2, Lorenzo + regression: 2.15 MB of buffer, the blocks'
    choices are the reference's)"""
    a = field3d((64, 256, 256))
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.absErrorBound = 1e-3
    L = sz3_amd.lib()
    oblob = oracle_compress(a, make_config(a.shape, abs_eb=1e-3, lorenzo=True, regression=True))
    blobs = {}
    for one in ("0", "1"):
        monkeypatch.setenv("SZ3HIP_STOCK_ONE_FRAME", one)
        L.sz3hip_set_stock_format(1)
        try:
            blobs[one], _ = sz3_amd.compress(a, conf)
        finally:
            L.sz3hip_set_stock_format(0)
    assert blobs["1"].tobytes() == oblob.tobytes()
    assert blobs["0"].tobytes() != oblob.tobytes() and abs(len(blobs["0"]) - len(oblob)) < 256
    assert np.array_equal(oracle_decompress(blobs["0"], a.dtype, a.shape)[0], oracle_decompress(oblob, a.dtype, a.shape)[0])


def test_stock_lorenzo_reg_writer_declines_what_it_does_not_take():
    """4-D blocks beyond 6^4 and a regression-only set with a one-element-thin block: this library's own stream instead (ids 16), never a wrong one"""
    L = sz3_amd.lib()
    for a, kw in ((field4d((5, 12, 14, 16)), dict(lorenzo=1, lorenzo2=0, regression=0, blockSize=7)), (field3d((23, 31, 16)), dict(lorenzo=0, lorenzo2=0, regression=1, blockSize=5)),
                  (field4d((6, 13, 12, 12)), dict(lorenzo=0, lorenzo2=0, regression=1))):
        conf = sz3_amd.Config(*a.shape)
        conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
        conf.absErrorBound = 1e-2
        for k, v in kw.items():
            setattr(conf, k, v)
        L.sz3hip_set_stock_format(1)
        try:
            blob, _ = sz3_amd.compress(a, conf)
        finally:
            L.sz3hip_set_stock_format(0)
        assert _trailer_algo(blob) == sz3_amd.ALGO_HIP_LORENZO
        dec, _ = sz3_amd.decompress(blob, a.dtype, a.shape)
        assert float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64)))) <= 1e-2


# ---- stock ALGO_NOPRED streams (round 5), both directions; the stock side is the reference library (the oracle restates no NOPRED) ----
@pytest.mark.parametrize("gen,eb", [(lambda: field3d((30, 41, 52)), 1e-2), (lambda: field1d(50001), 1e-3), (lambda: _with_holes(field3d((20, 30, 40))), 1e-2),
                                    (lambda: field2d((90, 130), np.float64), 1e-3)], ids=["3d", "1d", "3d-nan-inf", "2d-f64"])
def test_stock_nopred_streams_both_ways(gen, eb):
    """SZDispatcher.hpp:34-35 / 92-93 -> api/impl/SZAlgoNopred.hpp, decomposition/NoPredictionDecomposition.hpp:17-33: every value
    quantized against 0. Read: bit for bit what the reference decodes from its own stream. Written (sz3hip_set_stock_format +
    cmprAlgo ALGO_NOPRED): the reference decodes our stream to the values this library decodes, within the bound."""
    from oracle_binding import ALGO_NOPRED
    if not have_ref():
        pytest.skip("oracle/_ref/libsz3ref.so not built")
    a = gen()
    oconf = make_config(a.shape, algo=ALGO_NOPRED, abs_eb=eb)
    rblob = ref_compress(a, oconf)
    assert _trailer_algo(rblob) == sz3_amd.ALGO_NOPRED
    want = ref_decompress(rblob, a.dtype, a.shape)
    got, c2 = sz3_amd.decompress(rblob, a.dtype, a.shape)
    assert c2.cmprAlgo == sz3_amd.ALGO_NOPRED and np.array_equal(got, want, equal_nan=True)
    L = sz3_amd.lib()
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_NOPRED
    conf.absErrorBound = eb
    conf.regression = 0
    L.sz3hip_set_stock_format(1)
    try:
        blob, _ = sz3_amd.compress(a, conf)
    finally:
        L.sz3hip_set_stock_format(0)
    assert _trailer_algo(blob) == sz3_amd.ALGO_NOPRED
    back = ref_decompress(blob, a.dtype, a.shape)
    assert np.array_equal(back, want, equal_nan=True)      # the same quantizer on the same values: the same reconstruction
    mine, _ = sz3_amd.decompress(blob, a.dtype, a.shape)
    assert np.array_equal(mine, back, equal_nan=True)
    assert blob.tobytes() == rblob.tobytes(), "the same codes, the reference's tree order, one zstd frame: the reference's file"


def test_stock_4d_stream_with_the_second_order_member_is_read():
    """the reference's LorenzoPredictor<T, 4, 2>::predict returns 0 (predictor/LorenzoPredictor.hpp:92-94: no 4-D second-order stencil, the
    static_assert holds) and its estimate carries no noise term — a set that names the member still compresses and decompresses in stock
    SZ3; the reader follows (the oracle declines this set: the stock side is the reference library)"""
    if not have_ref():
        pytest.skip("oracle/_ref/libsz3ref.so not built")
    a = field4d((5, 12, 12, 14))
    oconf = make_config(a.shape, abs_eb=1e-2, lorenzo=True, lorenzo2=True, regression=True)
    rblob = ref_compress(a, oconf)
    want = ref_decompress(rblob, a.dtype, a.shape)
    got, c2 = sz3_amd.decompress(rblob, a.dtype, a.shape)
    assert c2.cmprAlgo == sz3_amd.ALGO_LORENZO_REG and np.array_equal(got, want)


def test_stock_lorenzo_reg_at_512_cubed_both_ways():
    """BASELINE's volume through the stock container: the reference's own ALGO_LORENZO_REG stream of a 512^3 f32 field read bit for
    bit, and this library's stock-format stream of the same field read by the reference within the bound, to the values this
    library reads from it (the stock figures came from a lab script, not from a test)."""
    if not have_ref():
        pytest.skip("oracle/_ref/libsz3ref.so not built")
    import time
    a = field3d((512, 512, 512))
    oconf = make_config(a.shape, abs_eb=1e-3, lorenzo=True, regression=True)
    rblob = ref_compress(a, oconf)
    want = ref_decompress(rblob, a.dtype, a.shape)
    t0 = time.perf_counter()
    got, _ = sz3_amd.decompress(rblob, a.dtype, a.shape)
    t_read = time.perf_counter() - t0
    assert np.array_equal(got, want)
    del got, want
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.absErrorBound = 1e-3
    L = sz3_amd.lib()
    L.sz3hip_set_stock_format(1)
    try:
        t0 = time.perf_counter()
        blob, _ = sz3_amd.compress(a, conf)
        t_write = time.perf_counter() - t0
    finally:
        L.sz3hip_set_stock_format(0)
    assert _trailer_algo(blob) == sz3_amd.ALGO_LORENZO_REG
    back = ref_decompress(blob, a.dtype, a.shape)
    assert float(np.max(np.abs(back.astype(np.float64) - a.astype(np.float64)))) <= 1e-3
    mine, _ = sz3_amd.decompress(blob, a.dtype, a.shape)
    assert np.array_equal(mine, back)
    assert len(blob) <= 1.08 * len(rblob)
    print("512^3 stock ALGO_LORENZO_REG: read %.0f ms, write %.0f ms, %d vs %d bytes" % (1e3 * t_read, 1e3 * t_write, len(blob), len(rblob)))
