"""Builds sz3_amd/libsz3hip.so (hipcc, gfx950 only) in-tree.  `python -m sz3_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsz3hip.so")
SOURCES = ["sz3hip_kernels.hip", "sz3hip_interp.hip", "sz3hip_api.cpp"]
HEADERS = ["sz3hip_kernels.h", "sz3hip_format.h", "../../include/sz3hip.h", "../../include/sz3c.h"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           *[os.path.join(CSRC, f) for f in SOURCES], "-o", LIB, "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
