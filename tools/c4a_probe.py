"""C4a field: GPU Lorenzo / interpolation ratio against the oracle's Lorenzo-only and Lorenzo+regression ratios."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import sz3_amd
from fields import field3d
from oracle_binding import make_config, oracle_compress

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for name, a, eb in (("C4a", field3d((n, n, n), np.float64, sigma=2e-3, scale=3.3e-5), 1e-6),
                    ("C2@1e-2", field3d((n, n, n), np.float32), 1e-2), ("C2@5e-2", field3d((n, n, n), np.float32), 5e-2)):
    out = {}
    for lab, kw in (("L", dict(lorenzo=True, regression=False)), ("L+R", dict(lorenzo=True, regression=True)), ("R", dict(lorenzo=False, regression=True)),
                    ("L+L2+R", dict(lorenzo=True, lorenzo2=True, regression=True))):
        b, st = oracle_compress(a, make_config(a.shape, abs_eb=eb, **kw), stats=True)
        out["oracle " + lab] = round(a.nbytes / len(b), 3)
    for lab, algo in (("gpu L", sz3_amd.ALGO_LORENZO_REG), ("gpu interp", sz3_amd.ALGO_INTERP_LORENZO)):
        c = sz3_amd.Config(*a.shape)
        c.cmprAlgo = algo
        c.regression = 0
        c.absErrorBound = eb
        blob, r = sz3_amd.compress(a, c)
        out[lab] = round(r, 3)
    print(name, n, out, flush=True)
