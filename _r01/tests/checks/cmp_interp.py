import sys, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,R); sys.path.insert(0,R+"/tests")
import numpy as np, torch, sz3_amd
from fields import field3d
from oracle_binding import *
S=256
a=field3d((S,S,S)); eb=1e-4
oconf=make_config(a.shape, algo=ALGO_INTERP, abs_eb=eb)
ob, st = oracle_compress(a, oconf, stats=True)
print("oracle: stream", len(ob), "raw", st.raw_bytes, "huff", st.huff_bytes, "unpred", st.n_unpred, "nodes", st.huff_node_count)
codes,_,_,_ = oracle_interp_codes(a, oconf)
h=np.bincount(codes, minlength=65536).astype(np.float64); p=h[h>0]/h.sum()
print("entropy bits/elem", -(p*np.log2(p)).sum(), "symbols", (h>0).sum())
dev=torch.device("cuda:0"); t=torch.from_numpy(a).to(dev)
dc=sz3_amd.DeviceCompressor(a.size, np.float32); cap=dc.payload_bound(a.size); pl=torch.empty(cap,dtype=torch.uint8,device=dev)
conf=sz3_amd.Config(*a.shape); conf.cmprAlgo=sz3_amd.ALGO_INTERP; conf.absErrorBound=eb
n=dc.compress(conf,t.data_ptr(),pl.data_ptr(),cap,0); print("gpu payload", n, dc.stats())
import szh_ref
hh,o,sec=szh_ref.parse(pl[:n].cpu().numpy())
lens=sec["lens"].astype(np.int64); f=h[hh["sym_min"]:hh["sym_min"]+hh["sym_count"]]
print("gpu code cost bits/elem", (f*lens).sum()/a.size, "max_len", hh["max_len"], "zero-len symbols with freq", int(((lens==0)&(f>0)).sum()))
print("chunkwords sum*4", int(sec["chunkwords"].astype(np.int64).sum())*4)
