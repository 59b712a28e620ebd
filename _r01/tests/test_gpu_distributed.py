"""GPU test (-m gpu) of the slab-parallel flow on REAL kernels with world_size 2: both ranks share the one GPU of the test
box and talk through gloo (the 8-GPU RCCL run is the driver's; the flow — stage 1, histogram all-reduce on the device
tensor, stage 2 with the SAME code book on every rank, per-slab payloads, container — is identical).
Checks: identical code-length tables on both ranks (= one global code book), the reduced histogram equals the histogram
of all slabs' codes, every slab decodes within the bound, the assembled container splits back into the slabs."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q, algo_name):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sz3_amd
        import szh_ref
        from fields import field3d
        from sz3_amd import distributed as D
        a = field3d((40, 48, 64))
        eb = 1e-3
        lo, hi = D.slab_bounds(a.shape[0], world, rank)
        slab = np.ascontiguousarray(a[lo:hi])
        dev = torch.device("cuda:0")
        t = torch.from_numpy(slab).to(dev)
        dc = sz3_amd.DeviceCompressor(slab.size, slab.dtype)
        hist = torch.zeros(65536, dtype=torch.int64, device=dev)
        sc = D.SlabCompressor(dist, dc, hist)
        cap = dc.payload_bound(slab.size)
        pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        conf = sz3_amd.Config(*slab.shape)
        conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG if algo_name == "lorenzo" else sz3_amd.ALGO_INTERP
        conf.absErrorBound = eb
        size = sc.compress(conf, t.data_ptr(), pl.data_ptr(), cap, 0)
        codes = dc.debug_codes(slab.size).astype(np.int64)
        payload = pl[:size].cpu().numpy().tobytes()
        h, o, sec = szh_ref.parse(payload)
        # every rank decodes its own slab
        out = torch.empty_like(t)
        dc.decompress(pl.data_ptr(), size, out.data_ptr(), 0)
        torch.cuda.synchronize()
        err = float(np.max(np.abs(out.cpu().numpy().astype(np.float64) - slab.astype(np.float64))))
        assert err <= eb
        gathered = [None] * world
        dist.all_gather_object(gathered, (h["sym_min"], h["sym_count"], sec["lens"].tobytes(), np.bincount(codes, minlength=65536), payload))
        # one code book for all ranks: same alphabet range and the same code lengths
        assert all(g[0] == gathered[0][0] and g[1] == gathered[0][1] and g[2] == gathered[0][2] for g in gathered)
        # the device histogram after the all-reduce is the histogram of ALL slabs' codes
        total = sum(g[3] for g in gathered)
        assert np.array_equal(hist.cpu().numpy(), total)
        if rank == 0:
            confs = []
            for r in range(world):
                l, hh = D.slab_bounds(a.shape[0], world, r)
                c = sz3_amd.Config(hh - l, *a.shape[1:])
                c.absErrorBound = eb
                confs.append(c.save())
            outer = sz3_amd.Config(*a.shape)
            outer.absErrorBound = eb
            outer.openmp = 1
            whole = D.assemble_container(confs, [g[4] for g in gathered], outer.save())
            o2, c2, blobs = D.split_container(whole)
            assert len(blobs) == world and all(blobs[r] == gathered[r][4] for r in range(world)) and o2 == outer.save()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["lorenzo", "interp"])
def test_two_ranks_share_one_code_book(algo):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + (7 if algo == "interp" else 0)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, algo)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res
