"""GPU tests (-m gpu): the randomised sweeps of tests/checks/ as part of the suite (fixed seeds, a few seconds each).
tests/checks/lorenzo_sweep.py  - K1 codes / outlier counts / payload decode against the numpy model of the format (tests/szh_ref.py)
tests/checks/interp_sweep.py   - interpolation codes and reconstruction against the oracle, bit for bit
tests/checks/host_sweep.py     - the host API over dtypes, error-bound modes and algorithms: the user-visible guarantee of each mode
tests/checks/block_sweep.py    - the block-composed path (predictor sets, block edges, radii, NaNs, steps) against the numpy block decoder
They found the small-quantbinCnt bugs fixed in round 1 (code 0 inside the kernels' LDS histogram windows)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, seed, n, flags=None):
    env = dict(os.environ, SEED=str(seed), N=str(n))
    if flags is not None:
        env["DBG_FLAGS"] = str(flags)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", tool)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("seed", [11, 12])
def test_lorenzo_sweep(seed):
    out = _run("lorenzo_sweep.py", seed, 30)
    assert "mismatches: 0" in out, out[-3000:]


@pytest.mark.parametrize("seed", [11, 12])
def test_interp_sweep(seed):
    out = _run("interp_sweep.py", seed, 25)
    assert "mismatches: 0" in out, out[-3000:]


def test_interp_sweep_through_the_level_kernels():
    """debug flag 4194304: every level of every 3-D case runs in the level kernels (the sweep's arrays are below their size)"""
    out = _run("interp_sweep.py", 13, 40, flags=4194304)
    assert "mismatches: 0" in out, out[-3000:]


@pytest.mark.parametrize("seed", [1, 2])
def test_block_path_sweep(seed):
    out = _run("block_sweep.py", seed, 30)
    assert "mismatches: 0" in out, out[-3000:]


@pytest.mark.parametrize("ndim,n", [(1, 30), (2, 30), (4, 16)])
def test_block_path_sweep_in_other_dimensions(ndim, n, monkeypatch):
    """the same sweep over 1-D, 2-D (every set, second-order Lorenzo included: round 4) and 4-D arrays (Lorenzo-1 / regression)"""
    monkeypatch.setenv("NDIM", str(ndim))
    out = _run("block_sweep.py", 5, n)
    assert "mismatches: 0" in out, out[-3000:]


def test_host_api_sweep():
    out = _run("host_sweep.py", 11, 40)
    assert "failures: 0" in out, out[-3000:]


BIG = [  # (id, algorithm, shape, dtype, abs bound, noise, GiB of free HBM needed)
    ("4.4e9-f32-lorenzo", "lorenzo", "1100,2000,2000", "f32", "1e-3", "2e-3", 120),
    ("4.4e9-f32-interp", "interp", "1100,2000,2000", "f32", "1e-3", "2e-3", 120),
    ("C5-whole-lorenzo", "lorenzo", "100,500,500,500", "f32", "2.4e-3", "2e-3", 225),
    ("C5-whole-default-REL1e-3", "default", "100,500,500,500", "f32", "rel:1e-3", "2e-3", 225),  # (BASELINE.json configs[4]: REL errBound through the device range scan)
    ("C4-whole-lorenzo", "lorenzo", "1024,1024,1024", "f64", "1e-6", "2e-6", 60),
    ("C4-whole-default", "default", "1024,1024,1024", "f64", "1e-6", "2e-6", 60),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", BIG, ids=[c[0] for c in BIG])
def test_round_trip_at_full_benchmark_sizes_and_beyond_2_pow_32_elements(case):
    """BASELINE.json's largest configurations whole on ONE GPU - C5 100 x 500^3 f32 (1.25e10 elements, 50 GB) and C4 1024^3 f64
    (8 GiB) - and 1100 x 2000 x 2000 f32 (4.4e9 elements): every index, chunk and list position beyond 32 bits. The field is
    generated and the error bound checked slab by slab on the device (tests/checks/big_roundtrip.py, a process of its own)"""
    import torch
    _, algo, shape, dtype, eb, sigma, need = case
    import time
    for _ in range(30):  # (the previous case's process has exited, the driver may still be handing its memory back)
        free, _ = torch.cuda.mem_get_info(0)
        if free >= need * 2 ** 30:
            break
        time.sleep(1.0)
    if free < need * 2 ** 30:
        pytest.skip("needs %d GiB of free HBM, %.0f free" % (need, free / 2 ** 30))
    n = 1
    for d in shape.split(","):
        n *= int(d)
    env = dict(os.environ, LAB_ALGO=algo, LAB_SHAPE=shape, LAB_EB=eb, LAB_DTYPE=dtype, LAB_SIGMA=sigma)
    if eb.startswith("rel:"):
        env.update(LAB_REL=eb[4:], LAB_EB="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "big_roundtrip.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert " OK " in last and str(n) in r.stdout, last


@pytest.mark.parametrize("seed", [11, 12])
def test_exact_tuner_sweep(seed):
    """tests/checks/tuner_exact_sweep.py: 25 random arrays (1-D .. 4-D, f32 / f64, smooth / noisy, bounds over four decades) — with the
    trials priced the reference's way every interpolation trial's compressed size equals the reference's byte for byte and every decision
    (linear / cubic, order, (alpha, beta), interpolation or Lorenzo, the Lorenzo quantizer) is the reference's"""
    out = _run("tuner_exact_sweep.py", seed, 25)
    assert "mismatches 0" in out


@pytest.mark.parametrize("seed", [21, 22])
def test_stock_containers_sweep(seed):
    """tests/checks/stock_bytes_sweep.py: 40 random calls in stock format (ALGO_INTERP with random parameters, the default algorithm, Lorenzo sets
    of one member, mixed sets in every dimension since round 6; 1-D .. 4-D, f32 / f64, absolute and relative bounds) — every container is the reference's, byte for byte"""
    out = _run("stock_bytes_sweep.py", seed, 40)
    assert "mismatches 0" in out


@pytest.mark.parametrize("seed", [31, 32])
def test_default_algorithm_reconstruction_sweep(seed):
    """tests/checks/default_algo_recon_sweep.py: 25 random 2-D .. 4-D arrays under the reference's default algorithm through the host API, this
    library's own stream format — decompressed bit for bit to what the reference's stream decompresses to"""
    out = _run("default_algo_recon_sweep.py", seed, 25)
    assert "mismatches 0" in out



@pytest.mark.parametrize("seed", [41, 42])
def test_context_history_sweep(seed):
    """tests/checks/history_sweep.py (round 6): one device context through 60 random calls (predictor sets, bounds, shapes up to 4 M elements
    — the sampled book's path —, speculation switched, decompressions in between) against a fresh context on every call: the payload is a
    function of the input and the configuration, whatever the context did before"""
    out = _run("history_sweep.py", seed, 60)
    assert "mismatches 0" in out


@pytest.mark.parametrize("seed", [51, 52])
def test_wild_data_sweep(seed):
    """tests/checks/wild_data_sweep.py (round 6): 60 arrays of the kinds the other sweeps do not draw — white noise, constants, zeros, steps, spikes
    of 1e30, bounds below the values' spacing, magnitudes of 1e30 and f32 denormals, integer-valued floats, negative zeros, NaN / Inf — under a
    random algorithm: this library's payload within the bound; in stock format the container's bytes are the reference's (given the reference's
    capacity: its rule for giving a lossy stream up is about the caller's buffer), and the reference's container reads back to the reference's own
    values bit for bit. (Not counted: non-finite values under a set with the regression member — NaN coefficients' signs, containers the reference
    cannot read back itself.)"""
    out = _run("wild_data_sweep.py", seed, 60)
    assert "failures 0" in out


def test_integer_wild_sweep():
    """tests/checks/int_wild_sweep.py (round 6): 60 integer arrays over all eight integer types — the full range of the type, constants, steps,
    alternating extremes, magnitudes beyond 2^53, bounds from 0.4 (lossless) to 1e6, every algorithm: dtype and shape kept, |x - x^| <= floor(eb)"""
    out = _run("int_wild_sweep.py", 61, 60)
    assert "failures: 0" in out


def test_context_history_sweep_with_wild_arrays(monkeypatch):
    """the same with noise, constants, zeros, spikes of 1e30, steps and bounds near the values' spacing among the calls (WILD=1): calls the lists
    cannot hold end in SZ3HIP_EOUTLIERS and the context goes on — the next payloads are a fresh context's"""
    monkeypatch.setenv("WILD", "1")
    out = _run("history_sweep.py", 43, 120)
    assert "mismatches 0" in out
