"""The superseded forms kept in the LAB build of the library (python -m sz3_amd.build --lab -> sz3_amd/libsz3hip_lab.so, -DSZ3HIP_LAB):
the fused stage 1 with its merging encoder (round 4) and the decoder's multi-symbol table (round 5) — both correct, both slower than
the product's forms on this chip, both left out of libsz3hip.so. These are their tests, moved here from tests/test_gpu_stages.py; they
run in a process of their own with SZ3HIP_LIB pointing at the lab library (tests/test_gpu_lab.py starts it)."""
import numpy as np
import pytest

import sz3_amd
from fields import field1d, field2d, field3d, field4d
import szh_ref

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

if not sz3_amd.lib().sz3hip_lab_build():
    pytest.skip("not the lab build of the library", allow_module_level=True)


def _conf(shape, eb):
    c = sz3_amd.Config(*shape)
    c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    c.regression = 0
    c.errorBoundMode = sz3_amd.EB_ABS
    c.absErrorBound = eb
    return c


FUSE_SHAPES = [((40, 64, 512), np.float32), ((24, 36, 256), np.float32), ((9, 20, 512), np.float32), ((17, 7, 768), np.float32),
               ((33, 10, 1024), np.float32), ((24, 40, 512), np.float64), ((300, 512), np.float32)]


@pytest.mark.parametrize("shape,dtype", FUSE_SHAPES, ids=["x".join(map(str, s)) + ("-f64" if d is np.float64 else "") for s, d in FUSE_SHAPES])
def test_fused_stage1_writes_the_unfused_encoders_bytes(shape, dtype):
    """Round 4: a context whose previous call left a small code book codes with it INSIDE stage 1 (k_lorenzo_quant_march3f: the rows'
    bit strings leave the kernel, k_merge moves them to their places). Same book, same symbols: the payload must be the one the
    unfused speculative encoder (one byte per element, then k_pack) writes from the same context state — byte for byte — on rows of
    256 ... 1024, ragged y / z extents (tasks that end beyond the array), f64, a 2-D array; and every payload decodes within the bound."""
    dev = torch.device("cuda:0")
    gen = (lambda seed: field3d(shape, dtype, seed=seed)) if len(shape) == 3 else (lambda seed: field3d((1,) + shape, dtype, seed=seed).reshape(shape))
    a, b, c3 = gen(1), gen(2), gen(3)
    n = a.size
    eb = 1e-3
    conf = _conf(shape, eb)
    L = sz3_amd.lib()
    ctxs = [sz3_amd.DeviceCompressor(n, dtype), sz3_amd.DeviceCompressor(n, dtype)]
    for d in ctxs:
        d.set_fused(True)  # (opt-in: the default is the two-pass form)
    cap = ctxs[0].payload_bound(n, worst_case=True)

    def run(dc, arr, flags=0):
        t = torch.from_numpy(arr).to(dev)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        L.sz3hip_debug_flags(flags)
        try:
            size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
        finally:
            L.sz3hip_debug_flags(0)
        fused = dc.fused
        dec = torch.empty_like(t)
        dc.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
        torch.cuda.synchronize()
        assert float((dec.double() - t.double()).abs().max()) <= eb
        return pl[:size].cpu().numpy().tobytes(), fused

    for k, arr in enumerate([a, a, b, c3, b]):
        got, fused = run(ctxs[0], arr)
        ref, fused_ref = run(ctxs[1], arr, flags=2048)  # (2048: no fused stage 1)
        assert not fused_ref
        assert fused == (k > 0), (k, fused)
        assert got == ref, "call %d: the fused stage 1 and the unfused encoder disagree" % k
    hits, misses = ctxs[0].spec_stats()
    assert misses == 0 and hits == 4, (hits, misses)


def test_fused_stage1_misses_repeat_the_call():
    """What voids a fused stage 1's output, each followed by the whole call once more in the two-pass form and a payload that is a fresh
    context's: a book the verdict rejects (another bound: another alphabet), a symbol the book has no code word for (a step of
    thousands of lattice units: listed deltas, symbol 0), outlier lists too long for the sort roles (NaN-laden field)."""
    dev = torch.device("cuda:0")
    shape = (24, 40, 512)
    a = field3d(shape, seed=5)
    n = a.size
    stepped = a.copy()
    stepped[:, :, 300:] += 7.0   # deltas of thousands of lattice steps along one plane: beyond one-byte codes' range -> listed
    holes = a.copy()
    holes.reshape(-1)[np.random.default_rng(3).choice(n, size=6000, replace=False)] = np.nan
    dc = sz3_amd.DeviceCompressor(n, np.float32)
    dc.set_speculation(True, backoff=False)
    dc.set_fused(True)
    cap = dc.payload_bound(n, worst_case=True)

    def run(d, arr, eb):
        conf = _conf(shape, eb)
        t = torch.from_numpy(arr).to(dev)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        size = d.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
        fused = d.fused
        dec = torch.empty_like(t)
        d.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
        torch.cuda.synchronize()
        o, x = dec.cpu().numpy(), arr
        m = np.isnan(x)
        assert np.array_equal(np.isnan(o), m)
        assert float(np.max(np.abs(o[~m].astype(np.float64) - x[~m].astype(np.float64)))) <= eb
        return pl[:size].cpu().numpy().tobytes(), fused

    steps = [(a, 1e-3, None), (a, 1e-3, True), (a, 4e-3, False), (a, 4e-3, True), (stepped, 4e-3, False), (a, 4e-3, None),
             (a, 4e-3, True), (holes, 4e-3, False)]
    for k, (arr, eb, want_fused) in enumerate(steps):
        h0, m0 = dc.spec_stats()
        got, fused = run(dc, arr, eb)
        h1, m1 = dc.spec_stats()
        if want_fused is not None:
            assert fused == want_fused, (k, fused)
            assert (h1 - h0, m1 - m0) == ((1, 0) if want_fused else (0, 1)), (k, h1 - h0, m1 - m0)
        if not fused:
            assert got == run(sz3_amd.DeviceCompressor(n, np.float32), arr, eb)[0], "call %d: a repeated call's payload is not a fresh context's" % k


@pytest.mark.parametrize("shape,eb", [((40, 52, 512), 1e-3), ((17, 33, 768), 1e-3), ((19, 13, 132), 1e-3), ((24, 40, 300), 2e-3), ((64, 1024), 1e-3),
                                      ((70000,), 1e-3), ((33, 47, 50), 1e-2), ((9, 11, 2), 1e-3)],
                         ids=["40x52x512", "17x33x768", "19x13x132", "24x40x300", "64x1024", "70000", "33x47x50", "9x11x2"])
def test_multi_symbol_decoder_matches_the_one_symbol_decoder(shape, eb):
    """Round 5: small code books of f32 Lorenzo streams are decoded through a table of up to three code words per 12-bit window
    (k_decode's MS form, encoder/HuffmanEncoder.hpp:225-255 is the walk it replaces) behind debug flag 2. Same
    array bit for bit — rows that divide the unit, rows that do not (carries), rows shorter than a lookup, listed deltas and values."""
    dev = torch.device("cuda:0")
    if len(shape) == 3:
        a = field3d(shape, seed=3)
    elif len(shape) == 2:
        a = field2d(shape, seed=3)
    else:
        a = field1d(shape[0])
    rng = np.random.default_rng(5)
    f = a.reshape(-1)
    idx = rng.choice(f.size, size=max(4, f.size // 4000), replace=False)
    f[idx[::2]] += 3.0          # steps of thousands of lattice units: listed deltas (symbol 0)
    f[idx[1::4]] = np.nan       # and values stored raw
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, np.float32)
    cap = dc.payload_bound(a.size, worst_case=True)
    pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    size = dc.compress(_conf(shape, eb), t.data_ptr(), pl.data_ptr(), cap, 0)
    outs = []
    L = sz3_amd.lib()
    for flag in (0, 2, 2097152):   # one-symbol table, multi-symbol table (opt-in), no half-width chain at all
        L.sz3hip_debug_flags(flag)
        try:
            o = torch.empty_like(t)
            dc.decompress(pl.data_ptr(), size, o.data_ptr(), 0)
            torch.cuda.synchronize()
        finally:
            L.sz3hip_debug_flags(0)
        outs.append(o.cpu().numpy())
    assert np.array_equal(outs[0], outs[1], equal_nan=True) and np.array_equal(outs[0], outs[2], equal_nan=True)
    fin = np.isfinite(a)
    assert float(np.abs(outs[0][fin].astype(np.float64) - a[fin].astype(np.float64)).max()) <= eb


