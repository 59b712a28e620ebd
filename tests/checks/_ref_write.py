#!/usr/bin/env python3
"""helper of wild_data_sweep.py: the reference's SZ_compress in a process of its own (its OpenMP path divides by zero or aborts on some shapes).
argv: array file, dtype name, shape "a,b,c", make_config's keywords as JSON, output file"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_binding import make_config, ref_compress
shape = tuple(int(x) for x in sys.argv[3].split(","))
a = np.fromfile(sys.argv[1], dtype=np.dtype(sys.argv[2])).reshape(shape)
ref_compress(a, make_config(shape, **json.loads(sys.argv[4]))).tofile(sys.argv[5])
