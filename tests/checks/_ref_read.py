#!/usr/bin/env python3
"""helper of wild_data_sweep.py: the reference's (or the oracle's) reading of a container, in a process of its own — the reference asserts
(aborts) on some containers it wrote itself (seen: a 1-D regression-only stream at a bound below the values' spacing).
argv: container file, dtype name, element count, output file, 1 = the reference build / 0 = the oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_binding import oracle_decompress, ref_decompress
blob = np.fromfile(sys.argv[1], dtype=np.uint8)
dt, n = np.dtype(sys.argv[2]), int(sys.argv[3])
out = ref_decompress(blob, dt, (n,)) if sys.argv[5] == "1" else oracle_decompress(blob, dt, (n,))[0]
np.ascontiguousarray(out).tofile(sys.argv[4])
