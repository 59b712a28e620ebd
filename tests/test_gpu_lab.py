"""GPU test (-m gpu): the lab build's superseded forms (fused stage 1, multi-symbol decoder table) agree with the product's forms —
tests/lab_forms.py under the lab library, in a process of its own (the product library is this process's)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LAB = os.path.join(ROOT, "sz3_amd", "libsz3hip_lab.so")


def test_product_build_leaves_the_superseded_forms_out():
    import sz3_amd
    assert sz3_amd.lib().sz3hip_lab_build() == 0


def test_lab_forms_agree_with_the_products():
    if not os.path.exists(LAB):
        pytest.skip("no lab library (python -m sz3_amd.build --lab)")
    env = dict(os.environ, SZ3HIP_LIB=LAB)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "lab_forms.py"), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-500:]
