"""GPU test (-m gpu): seeded sweep over shapes the hand-tuned kernels special-case differently (row lengths that are /
are not multiples of 4, 8, 128; dims of 1; tiny arrays; 1-D..4-D; f32 / f64; all three algorithm selections). Every case
must respect the bound strictly after a device round trip and decode deterministically."""
import numpy as np
import pytest

import sz3_amd

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _cases():
    rng = np.random.default_rng(20260928)
    out = []
    pool = [1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 40, 63, 64, 65, 96, 127, 128, 129, 136, 200, 256, 260]
    for k in range(48):
        nd = int(rng.integers(1, 5))
        shape = tuple(int(rng.choice(pool)) for _ in range(nd))
        while int(np.prod(shape)) > 3_000_000:
            shape = tuple(max(1, s // 2) for s in shape)
        dtype = np.float32 if rng.random() < 0.7 else np.float64
        eb = float(10.0 ** rng.integers(-5, -1))
        algo = ["lorenzo", "interp", "default"][k % 3]
        out.append((k, shape, dtype, eb, algo))
    return out


@pytest.mark.parametrize("k,shape,dtype,eb,algo", _cases(), ids=lambda v: str(v) if not isinstance(v, type) else v.__name__)
def test_random_shape_roundtrip(k, shape, dtype, eb, algo):
    rng = np.random.default_rng(1000 + k)
    grids = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
    a = sum(np.sin(2 * np.pi * g / (7.0 + 3 * i)) for i, g in enumerate(grids)) + 0.01 * rng.standard_normal(shape)
    a = np.ascontiguousarray(a.astype(dtype))
    if k % 7 == 0 and a.size > 10:
        a.reshape(-1)[a.size // 3] = np.nan
        a.reshape(-1)[a.size // 2] = 1e30
    conf = sz3_amd.Config(*shape)
    conf.cmprAlgo = {"lorenzo": sz3_amd.ALGO_LORENZO_REG, "interp": sz3_amd.ALGO_INTERP, "default": sz3_amd.ALGO_INTERP_LORENZO}[algo]
    conf.absErrorBound = eb
    blob, ratio = sz3_amd.compress(a, conf)
    blob2, _ = sz3_amd.compress(a, conf)
    assert np.array_equal(blob, blob2), "stream is not deterministic"
    dec, c2 = sz3_amd.decompress(blob, dtype, shape)
    fin = np.isfinite(a) & (np.abs(a) < 1e20)
    assert np.array_equal(np.isnan(dec), np.isnan(a))
    assert np.array_equal(dec[~fin & ~np.isnan(a)], a[~fin & ~np.isnan(a)])
    if fin.any():
        assert np.max(np.abs(dec[fin].astype(np.float64) - a[fin].astype(np.float64))) <= eb
