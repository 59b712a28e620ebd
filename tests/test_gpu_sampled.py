"""GPU tests (-m gpu) of round 6's sampled code book (sz3hip_kernels.h, szk_samp): arrays of at least 2^22 elements in rows of whole
256-element segments whose Lorenzo stream has one-byte codes are coded with a book built from a sample of the array — inside stage 1's
launch once the context knows the stream's form (k_lorenzo_quant_march3q<., true>), by a launch of its own behind the other forms
(k_sample). The book is a function of the input: every form must write the same payload, whatever the context coded before."""
import numpy as np
import pytest

import sz3_amd
from fields import field3d
import szh_ref

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

NO_SAMPLE = 65536      # sz3hip_debug_flags: no sampled book (the exact histogram's book, as before round 6)
NO_Q16 = 8             # ... the one-byte kernel instead of its 16-bit form
SAMPLE_APART = 4194304  # ... the 16-bit form without the sampling workgroups: the sample as a launch of its own


def _conf(shape, eb):
    c = sz3_amd.Config(*shape)
    c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    c.regression = 0
    c.errorBoundMode = sz3_amd.EB_ABS
    c.absErrorBound = eb
    return c


def _run(dc, t, conf, cap, pl, flags=0):
    L = sz3_amd.lib()
    L.sz3hip_debug_flags(flags)
    try:
        size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
    finally:
        L.sz3hip_debug_flags(0)
    return pl[:size].cpu().numpy().tobytes()


def _decode_ok(dc, blob, t, eb):
    dev = t.device
    pl = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(dev)
    out = torch.empty_like(t)
    dc.decompress(pl.data_ptr(), len(blob), out.data_ptr(), 0)
    torch.cuda.synchronize()
    return float((out.double() - t.double()).abs().max()) <= eb


def _spiky(shape, count, seed=11):
    a = field3d(shape).copy()
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, a.size, count)
    a.reshape(-1)[idx] += rng.choice([-1.0, 1.0], idx.size).astype(np.float32) * rng.uniform(0.3, 2.0, idx.size).astype(np.float32)
    return a


CASES = [
    ("f32-3d", lambda: field3d((64, 256, 256)), 1e-3),
    ("f32-3d-wide-rows", lambda: field3d((20, 210, 1024)), 1e-3),      # ragged y: bricks that end beyond the array
    ("f32-2d", lambda: field3d((1, 2048, 2048)).reshape(2048, 2048), 1e-3),
    ("f32-listed-few", lambda: _spiky((64, 256, 256), 30), 1e-3),       # listed deltas: the escape symbol, the packer's sort roles
    ("f32-listed-many", lambda: _spiky((256, 256, 256), 330), 1e-3),    # ... more than the roles sort (2048): the classic stage 2 with the sampled book
    ("f64-3d", lambda: field3d((40, 256, 512), np.float64, sigma=2e-6), 1e-6),
]


@pytest.mark.parametrize("name,gen,eb", CASES, ids=[c[0] for c in CASES])
def test_every_form_of_stage_1_writes_the_same_payload(name, gen, eb):
    dev = torch.device("cuda:0")
    a = gen()
    n = a.size
    conf = _conf(a.shape, eb)
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(n, a.dtype)
    cap = dc.payload_bound(n, worst_case=True)
    pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    first = _run(dc, t, conf, cap, pl)          # a context's first call: probe, two-launch form, k_sample, classic stage 2
    if not dc.stats()["narrow_codes"]:
        pytest.skip("the field took two-byte codes: no sampled book")
    h, _, _ = szh_ref.parse(np.frombuffer(first, dtype=np.uint8))
    if name == "f32-listed-few":
        assert 0 < h["n_dout"] <= 2048, h["n_dout"]
    if name == "f32-listed-many":
        assert h["n_dout"] > 2048, h["n_dout"]
    assert h["esc_sym"] == h["radius"] + 128 and h["sym_min"] == h["radius"] - 127 and h["sym_count"] == 256, (h["esc_sym"], h["sym_min"], h["sym_count"])
    assert h["version"] == 5  # (version 4 + the escape symbol)
    assert _decode_ok(dc, first, t, eb)
    warm = [_run(dc, t, conf, cap, pl) for _ in range(3)]  # the one-launch forms (f32: the 16-bit form with the sampling workgroups inside)
    for k, w in enumerate(warm):
        assert w == first, "warm call %d differs from the context's first call" % k
    assert _run(dc, t, conf, cap, pl, NO_Q16) == first, "the one-byte kernel + k_sample differs"
    assert _run(dc, t, conf, cap, pl, SAMPLE_APART) == first, "the 16-bit form + k_sample differs"
    # a context with another history: another array first
    other = torch.from_numpy(field3d(a.shape if a.ndim == 3 else (1,) + a.shape, a.dtype, seed=5, sigma=2e-3 if a.dtype == np.float32 else 2e-6).reshape(a.shape)).to(dev)
    dc2 = sz3_amd.DeviceCompressor(n, a.dtype)
    for _ in range(2):
        _run(dc2, other, conf, cap, pl)
    assert _run(dc2, t, conf, cap, pl) == first, "the payload depends on what the context coded before"
    hits, misses = dc.spec_stats()
    assert misses == 0, (hits, misses)
    # against the exact histogram's book: the sampled book costs next to nothing
    exact = _run(sz3_amd.DeviceCompressor(n, a.dtype), t, conf, cap, pl, NO_SAMPLE)
    assert len(first) <= len(exact) * 1.002, (len(first), len(exact))
    h2, _, _ = szh_ref.parse(np.frombuffer(exact, dtype=np.uint8))
    assert h2["esc_sym"] == 0 and h2["version"] == 4


def test_sampled_payload_through_the_format_model():
    """the numpy model of the format (tests/szh_ref.py) decodes a sampled-book payload — escape symbol and all — to the device decoder's array"""
    dev = torch.device("cuda:0")
    a = _spiky((16, 512, 512), 30)
    eb = 1e-3
    conf = _conf(a.shape, eb)
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    cap = dc.payload_bound(a.size, worst_case=True)
    pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    _run(dc, t, conf, cap, pl)
    blob = _run(dc, t, conf, cap, pl)
    h, _, sec = szh_ref.parse(np.frombuffer(blob, dtype=np.uint8))
    assert h["esc_sym"] == h["radius"] + 128 and h["n_dout"] > 0
    # (the bit-serial model decoder takes a minute per 4 M symbols: the first 64 chunks and the listed deltas' places)
    h_small = dict(h, n=64 * 1024, n_chunks=64)
    codes = szh_ref.huffman_decode(h_small, sec)
    q, d, ref_codes, bad, far = szh_ref.dualquant(a, eb, narrow=True)
    assert np.array_equal(codes, ref_codes.reshape(-1)[:64 * 1024])
    out = torch.empty_like(t)
    dc.decompress(pl.data_ptr(), len(blob), out.data_ptr(), 0)
    torch.cuda.synchronize()
    assert float((out.double() - t.double()).abs().max()) <= eb


def test_a_value_beyond_the_16_bit_form_repeats_the_call_with_the_same_book():
    dev = torch.device("cuda:0")
    a = field3d((64, 256, 256))
    eb = 1e-3
    conf = _conf(a.shape, eb)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    cap = dc.payload_bound(a.size, worst_case=True)
    pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    t = torch.from_numpy(a).to(dev)
    for _ in range(2):
        _run(dc, t, conf, cap, pl)
    assert dc.q16
    b = a.copy()
    b[10, 20, 30] = 13.0  # 6500 lattice steps: beyond the 16-bit form (a value outlier? no: within the lattice — a listed delta)
    tb = torch.from_numpy(b).to(dev)
    got = _run(dc, tb, conf, cap, pl)
    assert not dc.q16
    assert _decode_ok(dc, got, tb, eb)
    assert got == _run(sz3_amd.DeviceCompressor(a.size, a.dtype), tb, conf, cap, pl)


def test_escape_symbol_field_of_the_header_is_checked():
    """the payload header's anchor_stride field of a Lorenzo stream names the symbol that stands for a listed delta: a value outside the
    lengths' table is refused, 0 (= symbol 0 itself) on a stream that was written with an escape symbol decodes to garbage-free refusal or
    to an array — never to a fault"""
    import struct
    dev = torch.device("cuda:0")
    a = _spiky((64, 256, 256), 30)
    eb = 1e-3
    conf = _conf(a.shape, eb)
    t = torch.from_numpy(a).to(dev)
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype)
    cap = dc.payload_bound(a.size, worst_case=True)
    pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    blob = bytearray(_run(dc, t, conf, cap, pl))
    h, _, _ = szh_ref.parse(np.frombuffer(bytes(blob), dtype=np.uint8))
    if not h["esc_sym"]:
        pytest.skip("the field took two-byte codes: no sampled book")
    for bad in (0, h["sym_min"] - 1, h["sym_min"] + h["sym_count"], 70000, 1 << 40):
        b2 = bytearray(blob)
        struct.pack_into("<Q", b2, 152, bad)
        d_pl = torch.from_numpy(np.frombuffer(bytes(b2), dtype=np.uint8).copy()).to(dev)
        out = torch.empty_like(t)
        with pytest.raises(sz3_amd.SZ3HipError):
            dc.decompress(d_pl.data_ptr(), len(b2), out.data_ptr(), 0)
    assert _decode_ok(dc, bytes(blob), t, eb)  # (the context is in working order afterwards)
