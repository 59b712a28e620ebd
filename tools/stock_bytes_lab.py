import sys, struct; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, sz3_amd
import test_gpu_stock as T
from oracle_binding import *
L=sz3_amd.lib()
def body(blob):
    b=blob.tobytes(); plen,=struct.unpack_from("<Q",b,8); pay=np.frombuffer(b[16:16+plen],dtype=np.uint8); rawlen,=struct.unpack_from("<Q",pay.tobytes(),0)
    raw=np.empty(rawlen,dtype=np.uint8); n=oracle().szo_zstd_decompress(pay.ctypes.data,pay.size,raw.ctypes.data,rawlen); return raw.tobytes() if n==rawlen else None, b[:16], b[16+plen:]
def cmp(name, blob, oblob):
    r1,h1,t1=body(blob); r2,h2,t2=body(oblob)
    same=blob.tobytes()==oblob.tobytes()
    d=-1
    if r1 is not None and r2 is not None and r1!=r2:
        d=next((i for i in range(min(len(r1),len(r2))) if r1[i]!=r2[i]), min(len(r1),len(r2)))
    print("%-28s container %s | body %s (%d vs %d, first diff %d) | head %s | trailer %s | sizes %d %d" % (name, same, r1==r2, len(r1 or b''), len(r2 or b''), d, h1==h2, t1==t2, blob.size, oblob.size), flush=True)
for name, gen, eb, kw in T.CASES:
    a=gen(); conf=sz3_amd.Config(*a.shape); conf.regression=0; conf.cmprAlgo=sz3_amd.ALGO_INTERP; conf.absErrorBound=eb; conf.interpAlgo=kw.get("interp_algo",1)
    for k in ("interpDirection","interpAnchorStride","interpAlpha","interpBeta"):
        if k in kw: setattr(conf,k,kw[k])
    L.sz3hip_set_stock_format(1)
    try: blob,_=sz3_amd.compress(a,conf)
    finally: L.sz3hip_set_stock_format(0)
    cmp("interp "+name, blob, oracle_compress(a, make_config(a.shape, algo=ALGO_INTERP, abs_eb=eb, **kw)))

for name, gen, eb, kw in T.LR_WRITE_CASES:
    a=gen(); conf=sz3_amd.Config(*a.shape); conf.cmprAlgo=sz3_amd.ALGO_LORENZO_REG; conf.absErrorBound=eb
    conf.lorenzo, conf.lorenzo2, conf.regression = int(kw.get("lorenzo", True)), int(kw.get("lorenzo2", False)), int(kw.get("regression", False))
    if "block_size" in kw: conf.blockSize=kw["block_size"]
    if a.ndim==4 and kw.get("lorenzo2"): continue
    L.sz3hip_set_stock_format(1)
    try: blob,_=sz3_amd.compress(a,conf)
    finally: L.sz3hip_set_stock_format(0)
    cmp("lorenzo_reg "+name, blob, oracle_compress(a, make_config(a.shape, abs_eb=eb, **kw)))
from fields import field1d, field2d, field3d
for name, gen, eb in (("3d", lambda: field3d((30, 41, 52)), 1e-2), ("1d", lambda: field1d(50001), 1e-3), ("2d-f64", lambda: field2d((90, 130), np.float64), 1e-3)):
    a=gen(); conf=sz3_amd.Config(*a.shape); conf.cmprAlgo=sz3_amd.ALGO_NOPRED; conf.absErrorBound=eb; conf.regression=0
    L.sz3hip_set_stock_format(1)
    try: blob,_=sz3_amd.compress(a,conf)
    finally: L.sz3hip_set_stock_format(0)
    cmp("nopred "+name, blob, ref_compress(a, make_config(a.shape, algo=ALGO_NOPRED, abs_eb=eb)))
import os
os.environ["SZ3HIP_TUNER_EXACT"]="1"
for name, gen, kwc in (("default abs 3d", lambda: field3d((72, 80, 88)), dict(abs_eb=3e-2)), ("default rel 3d", lambda: field3d((72, 80, 88)), dict(eb_mode=EB_REL, rel_eb=1e-3)), ("default abs 2d", lambda: field2d((600,700)), dict(abs_eb=1e-3))):
    a=gen(); conf=sz3_amd.Config(*a.shape); conf.regression=0
    if "rel_eb" in kwc: conf.errorBoundMode=sz3_amd.EB_REL; conf.relErrorBound=kwc["rel_eb"]
    else: conf.absErrorBound=kwc["abs_eb"]
    L.sz3hip_set_stock_format(1)
    try: blob,_=sz3_amd.compress(a,conf)
    finally: L.sz3hip_set_stock_format(0)
    ob=oracle_compress(a, make_config(a.shape, algo=ALGO_INTERP_LORENZO, **kwc))
    cmp(name, blob, ob)
    _,_,t1=body(blob); _,_,t2=body(ob)
    if t1!=t2: print(t1.hex()); print(t2.hex())
