"""oracle/algohip_patch.py — TEST INFRASTRUCTURE. Applies the three edits include/SZ3/api/impl/SZAlgoHip.hpp documents (the
reference's own extension recipe, tools/sz3/sz3_customized_demo.cpp:8-14) to scratch copies of the reference's
utils/Config.hpp and api/impl/SZDispatcher.hpp. usage: algohip_patch.py <reference include dir> <output dir>"""
import os
import sys

ref, out = sys.argv[1], sys.argv[2]


def edit(rel, pairs):
    s = open(os.path.join(ref, rel)).read()
    for old, new in pairs:
        if s.count(old) != 1:
            raise SystemExit("algohip_patch: expected exactly one %r in %s" % (old, rel))
        s = s.replace(old, new)
    dst = os.path.join(out, rel)
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "w").write(s)


edit("SZ3/utils/Config.hpp", [
    ("ALGO_BIOMD, ALGO_BIOMDXTC };", "ALGO_BIOMD, ALGO_BIOMDXTC, ALGO_HIP_LORENZO = 16, ALGO_HIP_INTERP = 17 };"),
    ('{"ALGO_BIOMDXTC", ALGO_BIOMDXTC},', '{"ALGO_BIOMDXTC", ALGO_BIOMDXTC}, {"ALGO_HIP_LORENZO", ALGO_HIP_LORENZO}, {"ALGO_HIP_INTERP", ALGO_HIP_INTERP},'),
])
edit("SZ3/api/impl/SZDispatcher.hpp", [
    ('#include "SZ3/api/impl/SZAlgoBioMD.hpp"', '#include "SZ3/api/impl/SZAlgoBioMD.hpp"\n#include "SZ3/api/impl/SZAlgoHip.hpp"'),
    ("            } else if (conf.cmprAlgo == ALGO_BIOMD) {\n                return SZ_compress_bioMD",
     "            } else if (conf.cmprAlgo == ALGO_HIP_LORENZO || conf.cmprAlgo == ALGO_HIP_INTERP) {\n"
     "                cmpSize = SZ_compress_Hip<T, N>(conf, dataCopy.data(), cmpData, cmpCap);\n"
     "            } else if (conf.cmprAlgo == ALGO_BIOMD) {\n                return SZ_compress_bioMD"),
    ("    } else if (conf.cmprAlgo == ALGO_BIOMD) {\n        SZ_decompress_bioMD",
     "    } else if (conf.cmprAlgo == ALGO_HIP_LORENZO || conf.cmprAlgo == ALGO_HIP_INTERP) {\n"
     "        SZ_decompress_Hip<T, N>(conf, cmpData, cmpSize, decData);\n"
     "    } else if (conf.cmprAlgo == ALGO_BIOMD) {\n        SZ_decompress_bioMD"),
])
