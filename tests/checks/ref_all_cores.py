"""bench.py's all-cores CPU baseline leg (checker code, never the product): the REFERENCE's own OpenMP slab path
(SZ_compress_OMP, api/impl/SZImplOMP.hpp:16-117, through oracle/_ref/libsz3ref.so) on the bench field, run in a process of
its own so that OMP_NUM_THREADS — set by the caller — is what libgomp starts with. Prints one JSON line.
usage: ref_all_cores.py z,y,x f32|f64 lorenzo|interp|interp-notune eb"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from fields import field3d  # noqa: E402
from oracle_binding import ALGO_INTERP, ALGO_INTERP_LORENZO, make_config, ref_compress  # noqa: E402

shape = tuple(int(v) for v in sys.argv[1].split(","))
dtype, algo, eb = sys.argv[2], sys.argv[3], float(sys.argv[4])
a = field3d(shape, np.float32) if dtype == "f32" else field3d(shape, np.float64, sigma=2e-6)
conf = (make_config(shape, abs_eb=eb, lorenzo=True, regression=algo == "composed", openmp=True) if algo in ("lorenzo", "composed") else
        make_config(shape, algo=ALGO_INTERP_LORENZO if algo == "interp" else ALGO_INTERP, abs_eb=eb, regression=True, openmp=True))
best = 1e30
for _ in range(2):
    blob, sec = ref_compress(a, conf, timing=True)
    best = min(best, sec)
print(json.dumps({"sec": best, "bytes": int(len(blob)), "threads": os.environ.get("OMP_NUM_THREADS")}))
