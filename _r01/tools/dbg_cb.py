import sys, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,R); sys.path.insert(0,R+"/tests")
import numpy as np, torch, sz3_amd, szh_ref
from fields import field1d, field3d
def run(a, eb, algo):
    dev=torch.device("cuda:0"); t=torch.from_numpy(a).to(dev)
    dc=sz3_amd.DeviceCompressor(a.size,a.dtype); cap=dc.payload_bound(a.size); pl=torch.empty(cap,dtype=torch.uint8,device=dev)
    conf=sz3_amd.Config(*a.shape); conf.cmprAlgo=algo; conf.absErrorBound=eb
    n=dc.compress(conf,t.data_ptr(),pl.data_ptr(),cap,0)
    codes=dc.debug_codes(a.size)
    h,o,sec=szh_ref.parse(pl[:n].cpu().numpy())
    lens=sec["lens"]; print("n",a.size,"payload",n,"max_len",h["max_len"],"sym_min",h["sym_min"],"sym_count",h["sym_count"],"kraft",szh_ref.kraft(lens),"nsyms",(lens>0).sum(), "present", len(np.unique(codes)))
    f=np.bincount(codes,minlength=65536)[h["sym_min"]:h["sym_min"]+h["sym_count"]]
    print(" zero-len with freq", int(((lens==0)&(f>0)).sum()), " len with zero freq", int(((lens>0)&(f==0)).sum()))
    if a.size<=200000:
        d=szh_ref.huffman_decode(h,sec); print(" python decode ok:", np.array_equal(d,codes))
        if not np.array_equal(d,codes):
            bad=np.nonzero(d!=codes)[0]; print(" first bad", bad[:10], d[bad[:5]], codes[bad[:5]])
    out=torch.empty_like(t); dc.decompress(pl.data_ptr(),n,out.data_ptr(),0); torch.cuda.synchronize()
    print(" max err", float((out.double()-t.double()).abs().max()))
run(field1d(70001),1e-3,sz3_amd.ALGO_INTERP)
run(field1d(9000),1e-2,sz3_amd.ALGO_INTERP)
run(field3d((33,47,50)),1e-3,sz3_amd.ALGO_INTERP)
