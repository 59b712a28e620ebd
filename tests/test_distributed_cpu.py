"""world_size-2 gloo tests (CPU) of the slab-parallel driver logic (sz3_amd/distributed.py): slab bounds, the global
value range for REL bounds, the histogram all-reduce, and the multi-slab container — which the oracle's restatement of
SZ_decompress_OMP (api/impl/SZImplOMP.hpp:120-186) must decode."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        from fields import field3d
        from oracle_binding import ALGO_LORENZO_REG, EB_REL, make_config, oracle, oracle_codes, oracle_compress, oracle_decompress
        from sz3_amd import distributed as D
        a = field3d((21, 24, 28))
        lo, hi = D.slab_bounds(a.shape[0], world, rank)
        slab = a[lo:hi]
        # (1) global range through two scalar all-reduces == range of the whole array
        mn, mx = D.allreduce_range(float(slab.min()), float(slab.max()), dist)
        assert mn == float(a.min()) and mx == float(a.max())
        eb = D.abs_bound_from_range(EB_REL, 0.0, 1e-3, float(np.float32(mx) - np.float32(mn)))
        # (2) histogram all-reduce: sum of per-slab code histograms (each slab predicted with zero halo)
        conf = make_config(slab.shape, abs_eb=eb)
        codes, _ = oracle_codes(slab, conf)
        hist = torch.from_numpy(np.bincount(codes, minlength=65536).astype(np.int64))
        D.allreduce_histogram(hist, dist)
        allc = []
        for r in range(world):
            l, h = D.slab_bounds(a.shape[0], world, r)
            allc.append(oracle_codes(a[l:h], make_config(a[l:h].shape, abs_eb=eb))[0])
        assert np.array_equal(hist.numpy(), np.bincount(np.concatenate(allc), minlength=65536))
        # (3) container: every rank compresses its slab (here with the oracle), rank 0 assembles, the oracle's
        #     SZ_decompress_OMP restatement decodes the whole volume
        blob = oracle_compress(slab, conf)
        plen = int(np.frombuffer(blob[8:16].tobytes(), dtype=np.uint64)[0])
        payload, trailer = blob[16:16 + plen].tobytes(), blob[16 + plen:].tobytes()
        gathered = [None] * world
        dist.all_gather_object(gathered, (trailer, payload))
        if rank == 0:
            outer = make_config(a.shape, eb_mode=0, abs_eb=eb, openmp=True)
            buf = (C.c_ubyte * 256)()
            n = oracle().szo_config_save(C.byref(outer), buf)
            whole = D.assemble_container([g[0] for g in gathered], [g[1] for g in gathered], bytes(buf[:n]))
            o2, confs, blobs = D.split_container(whole)
            assert len(confs) == world and blobs[1] == gathered[1][1] and o2 == bytes(buf[:n])
            dec, c2 = oracle_decompress(np.frombuffer(whole, dtype=np.uint8), np.float32, a.shape)
            assert c2.openmp == 1
            assert np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) <= eb
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_slab_logic():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res


def test_slab_bounds_match_reference_formula():
    from sz3_amd import distributed as D
    for dim0, world in [(512, 8), (100, 8), (7, 8), (13, 4), (1, 3)]:
        w = D.effective_world(dim0, world)
        b = [D.slab_bounds(dim0, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == dim0 and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert all(h > l for l, h in b)
    assert [D.slab_bounds(100, 8, r)[1] - D.slab_bounds(100, 8, r)[0] for r in range(8)] == [12, 13, 12, 13, 12, 13, 12, 13]


def _worker8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        from fields import field4d
        from oracle_binding import EB_REL, make_config, oracle, oracle_compress, oracle_decompress
        from sz3_amd import distributed as D
        a = field4d((100, 6, 6, 8))  # C5's 100 time steps of a small volume
        lo, hi = D.slab_bounds(a.shape[0], world, rank)
        slab = np.ascontiguousarray(a[lo:hi])
        mn, mx = D.allreduce_range(float(slab.min()), float(slab.max()), dist)
        eb = D.abs_bound_from_range(EB_REL, 0.0, 1e-3, float(np.float32(mx) - np.float32(mn)))
        conf = make_config(slab.shape, abs_eb=eb)
        blob = oracle_compress(slab, conf)
        plen = int(np.frombuffer(blob[8:16].tobytes(), dtype=np.uint64)[0])
        payload, trailer = blob[16:16 + plen].tobytes(), blob[16 + plen:].tobytes()
        gathered = [None] * world
        dist.all_gather_object(gathered, (hi - lo, trailer, payload))
        if rank == 0:
            assert [g[0] for g in gathered] == [12, 13, 12, 13, 12, 13, 12, 13]  # api/impl/SZImplOMP.hpp:48-50 on dims[0] = 100, 8 slabs
            outer = make_config(a.shape, eb_mode=0, abs_eb=eb, openmp=True)
            buf = (C.c_ubyte * 256)()
            n = oracle().szo_config_save(C.byref(outer), buf)
            whole = D.assemble_container([g[1] for g in gathered], [g[2] for g in gathered], bytes(buf[:n]))
            o2, confs, blobs = D.split_container(whole)
            assert len(confs) == 8 and len(blobs) == 8 and o2 == bytes(buf[:n])  # eight Config blocks, eight blobs
            dec, c2 = oracle_decompress(np.frombuffer(whole, dtype=np.uint8), np.float32, a.shape)  # SZ_decompress_OMP restated (SZImplOMP.hpp:120-186)
            assert c2.openmp == 1
            assert np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))) <= eb
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_eight_rank_c5_split_and_container():
    """world_size 8 — C5's 100 time steps are cut 12, 13, 12, 13, 12, 13, 12, 13, the REL bound comes from the
    all-reduced range, the container carries eight Config blocks, and the oracle's SZ_decompress_OMP reads it within the bound."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res
