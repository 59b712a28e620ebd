"""GPU tests (-m gpu) of round 6's packer of one-byte codes (k_pack_b: pairs of codes looked up in a 128 x 128 LDS table) against
k_pack on the same context state: same book, same chunk table, so the payloads must agree byte for byte — on fields whose chunks
take the fast tier (C2-like), on fields whose octets exceed 64 bits or whose bytes leave the table's window (quads / pairs /
symbol by symbol), with listed deltas (byte 255), and with a ragged last chunk."""
import numpy as np
import pytest

import sz3_amd
from fields import field3d

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

OLD_PACKER = 32768  # sz3hip_debug_flags: k_pack for one-byte codes as before round 6
NO_SAMPLE = 65536   # ... the exact histogram's book (code words up to 16 bits, which both packers take; a call with a sampled book is k_pack_b's alone)


def _conf(shape, eb):
    c = sz3_amd.Config(*shape)
    c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    c.regression = 0
    c.errorBoundMode = sz3_amd.EB_ABS
    c.absErrorBound = eb
    return c


def _rough(shape, sigma, seed=7):
    rng = np.random.default_rng(seed)
    return (field3d(shape) + rng.normal(0.0, sigma, shape)).astype(np.float32)


def _spiky(shape):
    a = field3d(shape).copy()
    rng = np.random.default_rng(11)
    idx = rng.integers(0, a.size, 4000)
    a.reshape(-1)[idx] += rng.choice([-1.0, 1.0], idx.size) * rng.uniform(0.3, 2.0, idx.size).astype(np.float32)  # steps of hundreds of lattice units: listed deltas
    return a


CASES = [
    ("c2like", lambda: field3d((64, 256, 512)), 1e-3),
    ("c2like-1024", lambda: field3d((40, 128, 1024)), 1e-3),
    ("ragged-rows", lambda: field3d((61, 203, 516)), 1e-3),       # rows that are not whole segments: the bits pass, a ragged last chunk
    ("wide", lambda: _rough((64, 256, 512), 0.02), 1e-3),        # deltas of tens of lattice units: octets beyond 64 bits, bytes outside the window
    ("wider", lambda: _rough((48, 256, 512), 0.03), 1e-3),
    ("small-groups", lambda: field3d((32, 128, 1024)), 1e-3),    # 4096 chunks: eight work items per offset group
    ("spiky", lambda: _spiky((64, 256, 512)), 1e-3),
    ("smooth", lambda: field3d((64, 256, 512), sigma=0.0), 1e-2),  # one-bit codes
]


@pytest.mark.parametrize("name,gen,eb", CASES, ids=[c[0] for c in CASES])
def test_pair_table_packer_writes_the_old_packers_bytes(name, gen, eb):
    dev = torch.device("cuda:0")
    a = gen()
    n = a.size
    conf = _conf(a.shape, eb)
    L = sz3_amd.lib()
    t = torch.from_numpy(a).to(dev)

    def run(flags, spec):
        dc = sz3_amd.DeviceCompressor(n, a.dtype)
        dc.set_speculation(spec)
        cap = dc.payload_bound(n, worst_case=True)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        outs = []
        L.sz3hip_debug_flags(flags)
        try:
            for _ in range(3):  # (the first call of a context waits for the probe: the one-launch form — and with it k_pack_b — from the second call on)
                size = dc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
                outs.append(pl[:size].cpu().numpy().tobytes())
        finally:
            L.sz3hip_debug_flags(0)
        dec = torch.empty_like(t)
        dc.decompress(pl.data_ptr(), size, dec.data_ptr(), 0)
        torch.cuda.synchronize()
        assert float((dec.double() - t.double()).abs().max()) <= eb
        st = dc.stats()
        return outs, st

    new, st = run(NO_SAMPLE, False)
    old, _ = run(NO_SAMPLE | OLD_PACKER, False)
    if not st["narrow_codes"]:
        pytest.skip("the field took two-byte codes: not this packer's case")
    for k in range(3):
        assert new[k] == old[k], "call %d: k_pack_b and k_pack disagree (%s)" % (k, name)
    assert new[0] == new[1] == new[2]
