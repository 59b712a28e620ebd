#!/usr/bin/env python3
"""bench.py with a process-wide debug flag mask set first: python tools/flag_bench.py <flags> <bench args...> (same-box A/B of two equivalent paths)"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (before the library, as bench.py does)
import sz3_amd
sz3_amd.lib().sz3hip_debug_flags(int(sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
