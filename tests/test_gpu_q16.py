"""GPU tests (-m gpu) of the 16-bit form of the one-byte stage-1 kernel (round 5, k_lorenzo_quant_march3q): a lane's four lattice
values as two registers of packed 16-bit halves, packed f32 arithmetic, no per-element range test. It is licensed by the previous
call's probe (every sampled lattice value within +-2047 steps), writes THE SAME BYTES as the one-byte kernel (narrow_task), and
voids itself — the call is repeated with the one-byte kernel — when it meets a lattice value beyond +-4095 or a value that is not
finite. Reference kernels: BlockwiseDecomposition.hpp:28-46 + LorenzoPredictor.hpp:60-74 + LinearQuantizer.hpp:43-71 (the lattice
form of DESIGN.md section 2)."""
import numpy as np
import pytest

import sz3_amd
from fields import field1d, field2d, field3d
import szh_ref

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _conf(shape, eb):
    c = sz3_amd.Config(*shape)
    c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    c.regression = 0
    c.errorBoundMode = sz3_amd.EB_ABS
    c.absErrorBound = eb
    return c


def _gen(shape):
    if len(shape) == 3:
        return lambda seed: field3d(shape, seed=seed)
    if len(shape) == 2:
        return lambda seed: field2d(shape, seed=seed)
    return lambda seed: field3d((1, 1) + shape, seed=seed).reshape(shape)


SHAPES = [(40, 52, 512), (17, 33, 768), (19, 13, 132), (5, 7, 260), (24, 40, 300), (3, 2, 1024), (64, 1024), (37, 388), (8192,), (70000,)]


@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
def test_q16_stage1_writes_the_one_byte_kernels_bytes(shape):
    """Two contexts see the same series of arrays; one may take the 16-bit form (from its second call on), the other is kept to the
    one-byte kernel (debug flag 8). Payloads byte for byte, the one-byte codes element for element, and the bound — on rows of whole
    and broken 256-element segments, ragged y / z extents, 2-D and 1-D arrays."""
    dev = torch.device("cuda:0")
    gen = _gen(shape)
    arrs = [gen(1), gen(1), gen(2), gen(3), gen(2)]
    n = arrs[0].size
    eb = 1e-3
    conf = _conf(shape, eb)
    L = sz3_amd.lib()
    ctxs = [sz3_amd.DeviceCompressor(n, np.float32), sz3_amd.DeviceCompressor(n, np.float32)]
    for d in ctxs:
        d.set_deterministic(True)
    cap = ctxs[0].payload_bound(n, worst_case=True)

    def run(dc, arr, flags):
        t = torch.from_numpy(arr).to(dev)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        L.sz3hip_debug_flags(flags)
        try:
            size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
            codes = dc.debug_codes(n)
        finally:
            L.sz3hip_debug_flags(0)
        q16 = dc.q16
        dec = torch.empty_like(t)
        dc.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
        torch.cuda.synchronize()
        assert float((dec.double() - t.double()).abs().max()) <= eb
        return pl[:size].cpu().numpy().tobytes(), codes, q16

    for k, arr in enumerate(arrs):
        got, codes, q16 = run(ctxs[0], arr, 0)
        ref, codes_ref, q16_ref = run(ctxs[1], arr, 8)
        assert not q16_ref
        assert q16 == (k > 0), (k, q16)
        assert np.array_equal(codes, codes_ref), "call %d: codes differ at %s" % (k, np.flatnonzero(codes != codes_ref)[:8])
        assert got == ref, "call %d: the 16-bit form and the one-byte kernel disagree" % k
    # and the numpy model of stage 1 agrees with both
    q, d, exp_codes, bad, dout = szh_ref.dualquant(arrs[-1], eb, narrow=True)
    assert np.array_equal(codes.reshape(-1), exp_codes.reshape(-1))


def test_q16_stage1_with_listed_deltas_and_values():
    """Outliers inside the form's range: steps of a few hundred lattice units (deltas beyond one byte: listed), values the lattice
    reconstruction misses by rounding (none here) — the lists and the codes are the one-byte kernel's."""
    dev = torch.device("cuda:0")
    shape = (24, 40, 512)
    a = field3d(shape, seed=5)
    rng = np.random.default_rng(11)
    # (kept out of the probe's runs — 64 of every 32768 elements — with their seven upper neighbours: the call keeps one-byte codes)
    cand = rng.choice(a.size, size=3000, replace=False)
    offs = np.array([dz * shape[1] * shape[2] + dy * shape[2] + dx for dz in (0, 1) for dy in (0, 1) for dx in (0, 1)])
    keep = np.all(((cand[:, None] + offs[None, :]) % 32768) >= 64, axis=1)
    idx = cand[keep][:300]
    assert idx.size == 300
    a.reshape(-1)[idx] += rng.uniform(-1.0, 1.0, size=300).astype(np.float32)  # |q| stays below 2047: steps of up to 500 units
    n = a.size
    eb = 1e-3
    conf = _conf(shape, eb)
    L = sz3_amd.lib()
    out = []
    for flags in (0, 8):
        dc = sz3_amd.DeviceCompressor(n, np.float32)
        dc.set_deterministic(True)
        cap = dc.payload_bound(n, worst_case=True)
        t = torch.from_numpy(a).to(dev)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        L.sz3hip_debug_flags(flags)
        try:
            for _ in range(2):
                size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
        finally:
            L.sz3hip_debug_flags(0)
        assert dc.q16 == (flags == 0)
        st = dc.stats()
        assert st["n_delta_outliers"] > 300
        dec = torch.empty_like(t)
        dc.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
        torch.cuda.synchronize()
        assert float((dec.double() - t.double()).abs().max()) <= eb
        out.append(pl[:size].cpu().numpy().tobytes())
    assert out[0] == out[1]


@pytest.mark.parametrize("what", ["nan", "inf", "big", "fill", "edge"])
def test_q16_stage1_voids_itself_on_values_it_does_not_take(what):
    """A context licensed by a clean call meets an array with ONE value the form has no test for (NaN, Inf, a lattice value of 6000,
    a fill value of 1e35; 'edge': 4094 and 4096 steps, the form's last value and the first one beyond). The call is repeated with
    the one-byte kernel: the payload is a fresh context's, the context keeps to the one-byte kernel afterwards."""
    dev = torch.device("cuda:0")
    shape = (20, 24, 512)
    a = field3d(shape, seed=7)
    n = a.size
    eb = 1e-3
    b = a.copy()
    pos = (11, 13, 300)   # (the probe's runs are 64 of every 32768 elements: this one is not sampled)
    assert (np.ravel_multi_index(pos, shape) % 32768) >= 64
    expect_void = True
    if what == "nan":
        b[pos] = np.nan
    elif what == "inf":
        b[pos] = -np.inf
    elif what == "big":
        b[pos] = 6000 * 2e-3
    elif what == "fill":
        b[pos] = 1e35
    else:
        b[pos] = 4094 * 2e-3
        expect_void = False
    conf = _conf(shape, eb)

    def run(dc, arr):
        cap = dc.payload_bound(n, worst_case=True)
        t = torch.from_numpy(arr).to(dev)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
        dec = torch.empty_like(t)
        dc.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
        torch.cuda.synchronize()
        d, o = dec.cpu().numpy(), arr
        fin = np.isfinite(o)
        assert float(np.abs(d[fin].astype(np.float64) - o[fin].astype(np.float64)).max()) <= eb
        assert np.array_equal(d[~fin], o[~fin], equal_nan=True)
        return pl[:size].cpu().numpy().tobytes()

    dc = sz3_amd.DeviceCompressor(n, np.float32)
    dc.set_deterministic(True)
    run(dc, a)
    run(dc, a)
    assert dc.q16
    got = run(dc, b)
    assert dc.q16 == (not expect_void)
    fresh = sz3_amd.DeviceCompressor(n, np.float32)
    fresh.set_deterministic(True)
    assert got == run(fresh, b)
    if what == "edge":
        c = a.copy()
        c[pos] = 4100 * 2e-3   # beyond +-4095
        got = run(dc, c)
        assert not dc.q16
        fresh = sz3_amd.DeviceCompressor(n, np.float32)
        fresh.set_deterministic(True)
        assert got == run(fresh, c)
    run(dc, a)
    assert not dc.q16   # sits out the next calls


def test_q16_is_not_taken_for_wide_lattices_and_f64():
    """Lattice values beyond +-2047 in the probe's sample (a tight bound), or f64 data: the one-byte kernel stays."""
    dev = torch.device("cuda:0")
    shape = (16, 24, 512)
    for dtype, eb in ((np.float32, 1e-4), (np.float64, 1e-3)):
        a = field3d(shape, dtype, sigma=2e-5 if dtype is np.float32 else 2e-3, seed=3)
        dc = sz3_amd.DeviceCompressor(a.size, dtype)
        cap = dc.payload_bound(a.size, worst_case=True)
        t = torch.from_numpy(a).to(dev)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        for _ in range(3):
            size = dc.compress(_conf(shape, eb), t.data_ptr(), pl.data_ptr(), cap, 0)
            assert not dc.q16
        dec = torch.empty_like(t)
        dc.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
        torch.cuda.synchronize()
        assert float((dec.double() - t.double()).abs().max()) <= eb
