"""GPU tests (-m gpu) of round 6's small-array forms: the side section of a block-composed stream built by ONE workgroup (k_blk_side_small, up to
16384 blocks) against the eight launches it replaces (sz3hip_debug_flags(2048)), and the code book of a 257 .. 640-symbol alphabet built by
k_codebook<0> alone against the wide form (k_cb_compact + k_codebook<1> + k_cb_assign, which a first call launches beside it: part_hint -1) —
byte for byte, on 1-D, 2-D, 3-D and 4-D arrays, with and without regression blocks."""
import numpy as np
import pytest

import sz3_amd
from fields import field1d, field2d, field3d, field4d

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

EIGHT_LAUNCHES = 2048  # sz3hip_debug_flags: the side section's kernels one by one whatever the block count
BOTH_BOOKS = 131072    # ... both code book launches every call (the one whose alphabet it is builds the book)


def _payloads(a, conf, flags, calls=3):
    dev = torch.device("cuda:0")
    L = sz3_amd.lib()
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    cap = max(dc.payload_bound(a.size, worst_case=True), dc.payload_bound_conf(conf))
    pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    outs = []
    L.sz3hip_debug_flags(flags)
    try:
        for _ in range(calls):
            size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
            outs.append(pl[:size].cpu().numpy().tobytes())
    finally:
        L.sz3hip_debug_flags(0)
    dec = torch.empty_like(t)
    dc.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
    torch.cuda.synchronize()
    err = float((dec.double() - t.double()).abs().max())
    return outs, err, dc.stats()


def _composed(shape, eb, **kw):
    c = sz3_amd.Config(*shape)
    c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    c.errorBoundMode = sz3_amd.EB_ABS
    c.absErrorBound = eb
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _ramps(n, amp):
    """a line per block of 128 values, slopes and offsets drawn per block: regression wins in every block"""
    rng = np.random.default_rng(3)
    nb = n // 128
    slope, off = rng.uniform(-amp, amp, nb), rng.uniform(-amp, amp, nb)
    return (off[:, None] + slope[:, None] * np.arange(128)[None, :]).reshape(-1)


SIDE_CASES = [
    ("c1", lambda: field1d(1 << 20), 1e-3, {}),                                  # 8192 blocks of 128
    ("1d-ragged", lambda: field1d((1 << 19) + 77), 1e-3, {}),
    ("1d-l12", lambda: field1d(1 << 20), 1e-3, {"lorenzo2": 1, "regression": 0}),  # (no regression block: an empty coefficient chain)
    ("2d", lambda: field2d((700, 900)), 1e-3, {}),                                # 44 x 57 blocks of 16 x 16
    ("3d", lambda: field3d((64, 96, 128)), 1e-3, {}),                             # 11 x 16 x 22 blocks of 6^3
    ("3d-f64", lambda: (3.3e-5 * field3d((48, 80, 100), np.float64)), 1e-6, {}),
    ("3d-l2", lambda: field3d((40, 64, 96)), 1e-3, {"lorenzo2": 1}),
    ("4d", lambda: field4d((6, 20, 30, 40)), 1e-3, {}),                           # five coefficients
    # more than 8192 regression blocks (the loop form behind the register form's limit): a smooth series under a loose bound
    ("1d-many-regression-blocks", lambda: _ramps(1 << 21, 1.0), 1e-3, {}),
    # coefficient differences beyond 32 bits (the register form steps aside): slopes of 1e3 on a lattice of 4e-9
    ("1d-wide-coefficients", lambda: _ramps(1 << 18, 1e3), 1e-6, {}),
]


@pytest.mark.parametrize("name,gen,eb,kw", SIDE_CASES, ids=[c[0] for c in SIDE_CASES])
def test_side_section_by_one_workgroup_is_the_eight_launches_bytes(name, gen, eb, kw):
    a = gen()
    conf = _composed(a.shape, eb, **kw)
    one, err, st = _payloads(a, conf, 0)
    eight, err8, _ = _payloads(a, conf, EIGHT_LAUNCHES)
    assert err <= eb * (1 + 1e-6) and err8 <= eb * (1 + 1e-6)
    for k in range(3):
        assert one[k] == eight[k], "call %d of %s: the two forms of the side section disagree" % (k, name)
    assert one[0] == one[1] == one[2]


BOOK_CASES = [
    ("c1-398", lambda: field1d(1 << 20), 1e-3, {}),
    ("1d-lorenzo", lambda: field1d(1 << 20), 5e-4, {"regression": 0}),
    ("2d", lambda: field2d((1024, 1024)), 2.5e-4, {"regression": 0}),
    ("3d", lambda: field3d((64, 128, 256)), 2.2e-4, {"regression": 0}),
]


@pytest.mark.parametrize("name,gen,eb,kw", BOOK_CASES, ids=[c[0] for c in BOOK_CASES])
def test_code_book_launch_choice_does_not_change_the_bytes(name, gen, eb, kw):
    """an alphabet of 257 .. 640 symbols in a narrow range: k_codebook<0> alone (the context's hint from the second call on) against both
    launches every call; the first call of either context launches both (hint unknown)"""
    a = gen()
    conf = _composed(a.shape, eb, **kw)
    hinted, err, st = _payloads(a, conf, 0)
    if not 256 < st["n_symbols"] <= 640:
        pytest.skip("%d symbols: not the alphabet this test is about" % st["n_symbols"])
    both, _, _ = _payloads(a, conf, BOTH_BOOKS)
    assert err <= eb * (1 + 1e-6)
    for k in range(3):
        assert hinted[k] == both[k], "call %d of %s: hinted and unhinted code book launches disagree" % (k, name)
    assert hinted[0] == hinted[1] == hinted[2]


def test_block_streams_on_a_reused_context_are_a_function_of_the_input():
    """One context, one configuration (Lorenzo + regression), fields alternating between one the selection hands to the plain Lorenzo path and
    one that keeps its block stream, a decompression thrown in: every payload is the one a fresh context gives. (The context assumes the
    previous call's hand-over decision and zeroes the block predictor's counters behind a call — both must not show.)"""
    from fields import field_c4a
    dev = torch.device("cuda:0")
    shape = (60, 96, 132)
    smooth = field3d(shape, np.float64)                       # noise above the bound: regression never wins
    ramps = 1000.0 * field_c4a(shape, seed=5)                 # regression wins in a share of the blocks
    eb = 1e-3
    conf = _composed(shape, eb)
    ts = {"smooth": torch.from_numpy(smooth).to(dev), "ramps": torch.from_numpy(ramps).to(dev)}

    def fresh(name):
        dc = sz3_amd.DeviceCompressor(smooth.size, np.float64)
        cap = dc.payload_bound_conf(conf)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        n = dc.compress(conf, ts[name].data_ptr(), pl.data_ptr(), cap, 0)
        return pl[:n].cpu().numpy().tobytes()

    want = {k: fresh(k) for k in ts}
    assert want["smooth"][11] == 0 and want["ramps"][11] == 2, "the two fields were to take the plain and the block stream"
    dc = sz3_amd.DeviceCompressor(smooth.size, np.float64)
    cap = dc.payload_bound_conf(conf)
    pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    out = torch.empty_like(ts["smooth"])
    seq = ["smooth", "smooth", "ramps", "ramps", "ramps", "smooth", "ramps", "DEC", "ramps", "smooth", "DEC", "smooth", "ramps"]
    last = None
    for i, name in enumerate(seq):
        if name == "DEC":
            dc.decompress(pl.data_ptr(), len(want[last]), out.data_ptr(), 0)
            torch.cuda.synchronize()
            assert float((out - ts[last]).abs().max()) <= eb * (1 + 1e-9)
            continue
        n = dc.compress(conf, ts[name].data_ptr(), pl.data_ptr(), cap, 0)
        got = pl[:n].cpu().numpy().tobytes()
        assert got == want[name], "call %d (%s): a reused context's payload differs from a fresh one's" % (i, name)
        last = name
