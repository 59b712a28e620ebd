import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "plain_exit: block-predictor tests that leave the hand-over to the plain Lorenzo path switched on")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the reference itself, built only where /root/reference exists)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """oracle C restatement is (re)built from source on demand (gcc, <1 s); the HIP library too when hipcc is there."""
    so = os.path.join(ROOT, "oracle", "libsz3oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"])
    lib = os.path.join(ROOT, "sz3_amd", "libsz3hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        from sz3_amd.build import build
        build(verbose=False)
    yield


try:  # torch before anything of libsz3hip touches the device (sz3_amd/__init__.py, _share_torch_hip_runtime: the other order can stall)
    import torch  # noqa: F401
except Exception:  # noqa: BLE001
    pass


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
