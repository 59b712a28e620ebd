"""GPU test (-m gpu) of the one-process-per-GPU flow on REAL kernels with world_size 2: both ranks share the one GPU of the
test box, so the histogram travels through gloo here (RCCL needs a GPU per rank; the library's own RCCL exchange, the
container and its decoder are exercised on this box by tests/test_gpu_multislab.py, the 8-GPU run is the driver's).
Checks: identical code-length tables on both ranks (= one global code book), the reduced histogram equals the histogram
of all slabs' codes, every slab decodes within the bound."""
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


def _bench(args, env_extra, timeout=900):
    import json
    import subprocess
    env = dict(os.environ, **env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(lines[-1])


def test_bench_relaunches_itself_as_two_ranks_and_prints_one_line():
    """The driver's multi-GPU run is the first time bench.py --gpus N > 1 executes on real hardware. Its plumbing — the relaunch as N
    ranks under torch.distributed.run on 127.0.0.1, the barrier + max-over-ranks timing, the summed payload, ONE line from rank 0 — is
    run here with both ranks on the one GPU (SZ3_BENCH_ONE_GPU=1: the histogram then travels through gloo; RCCL wants a GPU per rank)."""
    d = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "128", "--no-extra", "--no-cold", "--no-host-e2e", "--no-cpu-baseline",
                "--no-live-traffic"], {"SZ3_BENCH_ONE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["unit"] == "GB/s"
    assert "gloo" in d["config"]["exchange"] and d["config"]["parallelism"] == "slab2"
    assert d["value"] > 0 and d["err_bound_ok"] and d["ratio"] > 2
    assert "roofline" in d and "cpu_baseline" not in d


def test_bench_two_ranks_print_the_multi_gpu_configs():
    """with N > 1 the line also carries BASELINE.json's configs[3] (1024^3 f64, Lorenzo + regression, 8 slabs: both
    fields) and configs[4] (100 x 500^3 f32, REL 1e-3 of the all-reduced range, 8 slabs of time steps) — here at reduced sizes
    (SZ3_BENCH_EXTRA_SCALE=small), two ranks on the one GPU. api/impl/SZImplOMP.hpp:48-107 is the split they mirror."""
    d = _bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--size", "128", "--no-cold", "--no-host-e2e", "--no-cpu-baseline", "--no-live-traffic"],
               {"SZ3_BENCH_ONE_GPU": "1", "SZ3_BENCH_EXTRA_SCALE": "small"})
    assert d["n_gpus"] == 2 and d["err_bound_ok"] and "roofline" in d
    ex = d["extra_configs"]
    c4 = ex["C4_8slab"]["fields"]
    for f in ("C4a", "C4b"):
        assert "error" not in c4[f], c4[f]
        assert c4[f]["err_bound_ok"] and c4[f]["ratio"] > 2 and c4[f]["value"] > 0 and c4[f]["decompress_device"]["ms"] > 0
    c5 = ex["C5_8slab"]
    assert "error" not in c5, c5
    assert c5["err_bound_ok"] and c5["ratio"] > 2 and c5["slab_rank0"] == [12, 16, 16, 16] and c5["slab_steps_by_rank"] == [12, 13]
    assert abs(c5["abs_bound_from_range"] - 1e-3 * (c5["value_range"][1] - c5["value_range"][0])) < 1e-12


def test_bench_eight_ranks_on_the_one_gpu():
    """the command the driver will run on an 8-GPU node — bench.py --gpus 8 — with the eight ranks on this box's one
    GPU (SZ3_BENCH_ONE_GPU=1: gloo carries the histogram) and the multi-GPU legs at reduced sizes: eight ranks rendezvous, the line says
    n_gpus 8 / slab8, C4's fields are coded as eight slabs, C5's 100 time steps are cut 12, 13, 12, 13, ... as the reference cuts them
    (api/impl/SZImplOMP.hpp:48-50) and every leg decodes within its bound."""
    d = _bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--size", "96", "--no-cold", "--no-host-e2e", "--no-cpu-baseline", "--no-live-traffic"],
               {"SZ3_BENCH_ONE_GPU": "1", "SZ3_BENCH_EXTRA_SCALE": "small"}, timeout=1500)
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "slab8" and d["scaling"] == "weak" and d["err_bound_ok"]
    ex = d["extra_configs"]
    for f in ("C4a", "C4b"):
        assert "error" not in ex["C4_8slab"]["fields"][f], ex["C4_8slab"]["fields"][f]
        assert ex["C4_8slab"]["fields"][f]["err_bound_ok"]
    c5 = ex["C5_8slab"]
    assert "error" not in c5, c5
    assert c5["slab_steps_by_rank"] == [12, 13, 12, 13, 12, 13, 12, 13] and c5["err_bound_ok"]


def test_bench_on_every_visible_gpu_over_rccl():
    """with more than one GPU on the box: the real thing at N = 2 (RCCL all-reduce of the histogram inside the library, one rank per GPU)"""
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    d = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "256", "--no-extra", "--no-cold", "--no-host-e2e", "--no-cpu-baseline",
                "--no-live-traffic"], {})
    assert d["n_gpus"] == 2 and "RCCL" in d["config"]["exchange"] and d["err_bound_ok"]


def test_one_process_slabs_over_every_visible_gpu():
    """conf.openmp on a box with several GPUs: compress_slabs' per-GPU host threads and ncclAllReduce between DIFFERENT devices
    (sz3hip_host.cpp, sz3hip_comm.cpp) — every slab coded with one book, the container read back by the library"""
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    sys.path.insert(0, HERE)
    import sz3_amd
    import szh_ref
    from fields import field3d
    import struct
    g = torch.cuda.device_count()
    a = field3d((8 * g, 96, 128))
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.regression = 0
    conf.absErrorBound = 1e-3
    conf.openmp = 1
    blob, ratio = sz3_amd.compress(a, conf)
    dec, c2 = sz3_amd.decompress(blob, np.float32, a.shape)
    assert c2.openmp == 1 and ratio > 3
    assert float(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64)))) <= 1e-3
    b = blob.tobytes()
    nslab, = struct.unpack_from("<i", b, 16)
    assert nslab == g
