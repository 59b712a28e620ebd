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
