"""Seeded synthetic fields of SURVEY.md section 8(d) / Appendix C (numpy, default_rng(20260928), 'ij', last axis fastest)."""
import numpy as np

SEED = 20260928


def field3d(shape, dtype=np.float32, sigma=2e-3, seed=SEED, scale=1.0):
    """C2/C3/C4 field: sin(2pi x/64) cos(2pi y/96) sin(2pi z/128) + 0.25 sin(2pi (x+2y+3z)/37) + N(0, sigma); x fastest."""
    nz, ny, nx = shape
    z, y, x = np.meshgrid(np.arange(nz, dtype=np.float64), np.arange(ny, dtype=np.float64),
                          np.arange(nx, dtype=np.float64), indexing="ij")
    f = np.sin(2 * np.pi * x / 64) * np.cos(2 * np.pi * y / 96) * np.sin(2 * np.pi * z / 128)
    f += 0.25 * np.sin(2 * np.pi * (x + 2 * y + 3 * z) / 37)
    f += np.random.default_rng(seed).normal(0.0, sigma, size=f.shape)
    return (scale * f).astype(dtype)


def field4d(shape, dtype=np.float32, sigma=2e-3, seed=SEED):
    """C5 field: sin(2pi (x+0.5t)/32) cos(2pi y/48) sin(2pi z/64) (1+0.01 t) + N(0, sigma); x fastest, t slowest."""
    nt, nz, ny, nx = shape
    t, z, y, x = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in (nt, nz, ny, nx)], indexing="ij")
    g = np.sin(2 * np.pi * (x + 0.5 * t) / 32) * np.cos(2 * np.pi * y / 48) * np.sin(2 * np.pi * z / 64) * (1 + 0.01 * t)
    g += np.random.default_rng(seed).normal(0.0, sigma, size=g.shape)
    return g.astype(dtype)


def field1d(n, dtype=np.float32):
    """C1: the first n values (x fastest) of the C2 field; generated from a (ceil(n/512^2),512,512) sub-volume."""
    nx = ny = 512
    nz = max(1, -(-n // (nx * ny)))
    return field3d((nz, ny, nx), dtype).reshape(-1)[:n].copy()


def field2d(shape, dtype=np.float32, sigma=2e-3, seed=SEED):
    ny, nx = shape
    y, x = np.meshgrid(np.arange(ny, dtype=np.float64), np.arange(nx, dtype=np.float64), indexing="ij")
    f = np.sin(2 * np.pi * x / 64) * np.cos(2 * np.pi * y / 96) + 0.25 * np.sin(2 * np.pi * (x + 2 * y) / 37)
    f += np.random.default_rng(seed).normal(0.0, sigma, size=f.shape)
    return f.astype(dtype)


def field_c4a(shape, seed=SEED):
    """C4a: 3.3e-5 x (the C2 formula evaluated in f64, noise sigma = 2e-3 added before scaling) — at abs 1e-6 the
    bound is ~3 % of the amplitude and the composed predictor picks regression for ~14 % of the blocks"""
    return field3d(shape, np.float64, sigma=2e-3, seed=seed, scale=3.3e-5)


def testfloat_like():
    """A stand-in for the reference's CI fixture tools/sz3/testfloat_8_8_128.dat (8 x 8 x 128 f32, x fastest; the reference's file is not
    kept here): an analytic field of the same shape and character — smooth along x, positive, range ~[0.2, 5], mean ~1."""
    z, y, x = np.meshgrid(np.arange(8, dtype=np.float64), np.arange(8, dtype=np.float64), np.arange(128, dtype=np.float64), indexing="ij")
    f = np.exp(0.9 * np.sin(2 * np.pi * x / 128 * 1.5 + 0.3 * y) + 0.45 * np.cos(2 * np.pi * y / 8 + 0.2 * z) + 0.3 * np.sin(2 * np.pi * z / 8))
    f = f / f.mean()
    f += np.random.default_rng(SEED).normal(0.0, 2e-5, size=f.shape)
    return f.astype(np.float32)
