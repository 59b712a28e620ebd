"""the C4a slab through the 3-D block decoders (debug flags 0 / 65536): the same array bit for bit; ctl[1] of k_blk_wave3 is not raised
(the output would differ); a few repetitions, since the exchange between groups is a matter of timing"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sz3_amd
from tests.fields import field_c4a  # noqa

shape = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "128,1024,1024").split(","))
dtype = np.float64 if (len(sys.argv) <= 2 or sys.argv[2] == "f64") else np.float32
a = field_c4a(shape, seed=5).astype(dtype)
conf = sz3_amd.Config(*shape)
conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
conf.errorBoundMode = sz3_amd.EB_ABS
conf.absErrorBound = 1e-6 if dtype == np.float64 else 1e-4
conf.lorenzo, conf.lorenzo2, conf.regression = 1, int(os.environ.get("L2", "0")), 1
blob, _ = sz3_amd.compress(a, conf)
outs = []
for rep in range(4):
    for flag in (0, 65536):
        sz3_amd.lib().sz3hip_debug_flags(flag)
        dec, _ = sz3_amd.decompress(blob, dtype, shape)
        outs.append(dec)
sz3_amd.lib().sz3hip_debug_flags(0)
same = all(np.array_equal(outs[0], o) for o in outs[1:])
err = float(np.max(np.abs(outs[0].astype(np.float64) - a.astype(np.float64))))
print("shape", shape, dtype.__name__, "ratio %.2f" % (a.nbytes / len(blob)), "identical", same, "max err", err)
sys.exit(0 if same and err <= conf.absErrorBound else 1)
