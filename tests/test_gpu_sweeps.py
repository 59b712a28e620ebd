"""GPU tests (-m gpu): the randomised sweeps of tests/checks/ as part of the suite (fixed seeds, a few seconds each).
tests/checks/lorenzo_sweep.py  - K1 codes / outlier counts / payload decode against the numpy model of the format (tests/szh_ref.py)
tests/checks/interp_sweep.py   - interpolation codes and reconstruction against the oracle, bit for bit
tests/checks/host_sweep.py     - the host API over dtypes, error-bound modes and algorithms: the user-visible guarantee of each mode
They found the small-quantbinCnt bugs fixed in round 1 (code 0 inside the kernels' LDS histogram windows)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, seed, n):
    env = dict(os.environ, SEED=str(seed), N=str(n))
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


def test_host_api_sweep():
    out = _run("host_sweep.py", 11, 40)
    assert "failures: 0" in out, out[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["lorenzo", "interp"])
def test_round_trip_beyond_2_pow_32_elements(algo):
    """1100 x 2000 x 2000 f32 = 4.4e9 elements (17.6 GB): every index, chunk and list position beyond 32 bits; the field is
    generated and the bound checked slab by slab on the device (tests/checks/big_roundtrip.py, own process: ~70 GB of HBM for seconds)"""
    import torch
    free, _ = torch.cuda.mem_get_info(0)
    if free < 120 * 2 ** 30:
        pytest.skip("needs ~70 GB of free HBM")
    env = dict(os.environ, LAB_ALGO=algo, LAB_SHAPE="1100,2000,2000", LAB_EB="1e-3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "big_roundtrip.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert " OK " in last and "4400000000" in r.stdout, last
