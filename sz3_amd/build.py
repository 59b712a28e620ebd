"""Builds sz3_amd/libsz3hip.so (hipcc, gfx950 only) in-tree.  `python -m sz3_amd.build [--force] [--lab]`.

Every source is compiled to its own object (in parallel, only when it or a header changed), then linked.
--lab: the lab build, sz3_amd/libsz3hip_lab.so (-DSZ3HIP_LAB): the product plus the superseded forms kept for reference — the fused
stage 1 with its merging encoder, the decoder's multi-symbol table. tests/test_gpu_lab.py checks them against the product's forms."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libsz3hip.so")
SOURCES = ["sz3hip_kernels.hip", "sz3hip_interp.hip", "sz3hip_regress.hip", "sz3hip_stock.hip", "sz3hip_sortlists.hip", "sz3hip_api.cpp", "sz3hip_host.cpp", "sz3hip_stock_host.cpp", "sz3hip_comm.cpp", "sz3hip_h5z.cpp"]
HEADERS = ["sz3hip_kernels.h", "sz3hip_format.h", "sz3hip_internal.h", "sz3hip_devutil.h", "sz3hip_stock_geom.h", "sz3hip_stock_host.h", "../../include/sz3hip.h", "../../include/sz3c.h", "../../include/sz3hip_h5z.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]


def _sources():
    return [f for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


LAB_SOURCES = ["sz3hip_kernels.hip", "sz3hip_api.cpp"]  # (the sources SZ3HIP_LAB changes: the others' objects are the product's)


def build(force=False, verbose=True, lab=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    if lab:
        build(force=force, verbose=verbose)  # (the shared objects)
    lib = os.path.join(HERE, "libsz3hip_lab.so") if lab else LIB
    th = _newest_header()
    jobs = []
    objs = []
    for f in _sources():
        src = os.path.join(CSRC, f)
        is_lab = lab and f in LAB_SOURCES
        obj = os.path.join(OBJ, f + (".lab.o" if is_lab else ".o"))
        objs.append(obj)
        if (lab and not is_lab):
            continue
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), th):
            jobs.append([hipcc, *FLAGS, *(["-DSZ3HIP_LAB"] if is_lab else []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib, "-ldl", "-lpthread"])
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, lab="--lab" in sys.argv)
