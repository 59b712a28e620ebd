#!/usr/bin/env python3
"""bench.py — SZ3 hot path on MI355X: compression throughput (GB/s) + ratio at a fixed absolute error bound.

Workload at N=1 (BASELINE.json configs[1], "C2"): 3-D float32 512x512x512 synthetic field, Lorenzo predictor,
abs errBound 1e-3.  A "step" is one pass of the hot path over that volume with the input already resident in HBM:
    stage1 (prequantise + integer Lorenzo + code emission + outlier capture + histogram)
    [N>1: RCCL all-reduce(sum) of the 65536 x u64 code histogram, issued by the library itself
          (sz3hip_comm_allreduce_histogram, csrc/sz3hip_comm.cpp) on the compress stream — the path's only exchange]
    stage2 (canonical Huffman codebook, chunked bit-pack, payload assembly)  -> payload resident in HBM
    finish (stream sync + payload size to the host)
N>1 = weak scaling: every rank compresses its own 512^3 slab of an (N*512) x 512 x 512 volume (slabs are independent
like the reference's SZ_compress_OMP slabs, api/impl/SZImplOMP.hpp:48-55), value = total bytes of all ranks / time.
One process per GPU: under a launcher (torchrun sets WORLD_SIZE) this process is one rank; started plainly with
--gpus N > 1 it re-launches itself as N ranks (python -m torch.distributed.run, 127.0.0.1). The ranks' RCCL
communicator lives in the library (ncclCommInitRank with rank 0's id shipped over the launcher's process group) and the
bench asserts that RCCL reports N ranks.

The timed loop ALTERNATES two realisations of the field (same analytic field, two noise seeds): no two consecutive calls see the
same array, so what a context carries over from its previous call (kernel forms, the code book it speculates with) is what a
series of similar arrays gives it, not what a repeated array gives it. `identical_input` reports the repeated-array figure beside it.

Printed JSON line (rank 0): the driver contract + "roofline" (the path's algorithmic bytes per launch — input +
payload — over the dominant kernel's duration, live HIP-event timing; `frac_kernel_compulsory` prices the kernel's own bytes;
"roofline_path" = the same bytes over the whole step; "roofline_decompress" the decode) + "ms_per_step_median" (every step timed on
its own) + "value_deterministic" / "value_cold" + "cpu_baseline" (the
reference itself from oracle/_ref when present, else the oracle port; one thread AND all host cores through the
reference's OpenMP slab path; rank 0, N=1 only) + "extra_configs" (C3 = ALGO_INTERP_LORENZO at 1e-4, same protocol,
fewer steps) + informational extras (ratio, per-stage ms, host end-to-end incl. PCIe and zstd — never `value`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512, help="edge of the cubic volume per GPU (512 = the metric's config)")
    ap.add_argument("--eb", type=float, default=1e-3)
    ap.add_argument("--algo", choices=["lorenzo", "interp", "interp-notune", "composed"], default="lorenzo",
                    help="lorenzo = the metric's config C2 (default); interp = C3 (ALGO_INTERP_LORENZO: sampling auto-tuner + "
                         "interpolation; use --eb 1e-4); interp-notune = ALGO_INTERP with the default cubic parameters; composed = "
                         "ALGO_LORENZO_REG with Lorenzo + regression chosen per block (C4's predictor set: --dtype f64 --shape 128,1024,1024 --eb 1e-6)")
    ap.add_argument("--field", choices=["default", "c4a"], default="default", help="c4a: SURVEY.md 8(d)'s C4a field (f64, scale 3.3e-5) where both predictors are chosen")
    ap.add_argument("--shape", default=None, help="z,y,x of the per-GPU volume instead of --size^3 (e.g. 128,1024,1024 = one C4 slab)")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32", help="f64 + --shape 128,1024,1024 --eb 1e-6 = C4's per-GPU slab")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-e2e", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs (C3) leg")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold / hint-miss timings")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 --pmc child passes that measure roofline.traffic in this run")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)  # (the child of those passes: warm-up + steps, nothing else, no line)
    return ap.parse_args()


def relaunch_as_ranks(n):
    """no launcher around us and --gpus N > 1: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


class Workload:
    """one configuration, device resident: input tensor, context, payload buffer; step() = one pass of the hot path"""

    def __init__(self, torch, sz3_amd, dev, local_rank, rank, shape, dtype, algo, eb, comm=None, dist=None, field="default"):
        from fields import field3d, field_c4a
        self.torch, self.sz = torch, sz3_amd
        self.shape, self.algo, self.eb, self.dtype = shape, algo, eb, dtype
        self.npdt = np.float32 if dtype == "f32" else np.float64
        self.esz = 4 if dtype == "f32" else 8
        self.n = int(np.prod(shape))
        # every rank: its own slab of the same analytic field family (different noise seed per rank)
        if field == "c4a":
            self.a = field_c4a(shape, seed=20260928 + rank).astype(self.npdt)
        else:
            self.a = field3d(shape, self.npdt, seed=20260928 + rank) if dtype == "f32" else field3d(shape, self.npdt, seed=20260928 + rank, sigma=2e-6)
        self.d_in = torch.from_numpy(self.a).to(dev)
        # the second realisation the timed loop alternates with (same field, another noise seed)
        if field == "c4a":
            other = field_c4a(shape, seed=20261928 + rank).astype(self.npdt)
        else:
            other = field3d(shape, self.npdt, seed=20261928 + rank) if dtype == "f32" else field3d(shape, self.npdt, seed=20261928 + rank, sigma=2e-6)
        self.pair = [self.d_in, torch.from_numpy(other).to(dev)]
        del other
        self.alternate = True
        self.flip = 0
        conf = sz3_amd.Config(*shape)
        conf.cmprAlgo = {"lorenzo": sz3_amd.ALGO_LORENZO_REG, "interp": sz3_amd.ALGO_INTERP_LORENZO, "interp-notune": sz3_amd.ALGO_INTERP,
                         "composed": sz3_amd.ALGO_LORENZO_REG}[algo]
        conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, int(algo == "composed")
        conf.errorBoundMode = sz3_amd.EB_ABS
        conf.absErrorBound = eb
        self.conf = conf
        self.dc = sz3_amd.DeviceCompressor(self.n, self.npdt, device=local_rank)
        self.cap = self.dc.payload_bound(self.n)
        self.d_payload = torch.empty(self.cap, dtype=torch.uint8, device=dev)
        self.comm, self.dist = comm, dist
        self.hist = None
        if comm is None and dist is not None:  # ranks sharing one GPU (SZ3_BENCH_ONE_GPU): gloo reduces a caller-owned histogram
            self.hist = torch.zeros(65536, dtype=torch.int64, device=dev)
            self.dc.set_histogram(self.hist.data_ptr())
        self.stream = torch.cuda.current_stream().cuda_stream

    def step(self):
        dc = self.dc
        if self.alternate:
            self.flip ^= 1
            self.d_in = self.pair[self.flip]
        dc.stage1(self.conf, self.d_in.data_ptr(), self.stream)
        if self.comm is not None:
            self.comm.allreduce_histogram([dc], [self.stream])
        elif self.hist is not None:
            self.dist.all_reduce(self.hist, op=self.dist.ReduceOp.SUM)
        dc.stage2(self.d_payload.data_ptr(), self.cap, self.stream)
        return dc.finish(self.stream)

    def cold_numbers(self, steps, barrier):
        """What `value` leaves out: the context remembers the previous call (kernel forms, the tuner's outcome, the code book). Still
        on alternating realisations: (a) every call a context's first call (sz3hip_ctx_forget before it, allocations excluded);
        (b) code-book speculation switched off; (c) deterministic payloads (the previous book must BE this call's: on distinct
        arrays every attempt fails)."""
        torch = self.torch
        out = {}

        def run(prep, n):
            for _ in range(2):
                prep()
                self.step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                prep()
                self.step()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / n
        out["first_call_ms"] = round(run(self.dc.forget, steps), 4)
        if self.algo == "interp":
            self.dc.tuner_report()
            out["alternating_fields_tuner_speculated"] = self.dc.speculated  # 1: stage 1 started with the previous outcome and was confirmed
        self.dc.set_speculation(False)
        out["no_codebook_speculation_ms"] = round(run(lambda: None, steps), 4)
        self.dc.set_speculation(True)
        # deterministic mode (sz3hip_ctx_set_deterministic: the previous book stands only when it IS this call's book — what the host
        # API sets): on alternating realisations every attempt fails and is repeated; the back-off keeps it near the plain path
        self.dc.set_deterministic(True)
        h0, m0 = self.dc.spec_stats()
        out["deterministic_payloads_ms"] = round(run(lambda: None, steps), 4)
        h1, m1 = self.dc.spec_stats()
        out["deterministic_payloads_codebook_speculation"] = {"hits": h1 - h0, "misses": m1 - m0}
        self.dc.set_deterministic(False)
        self.step()
        self.step()
        return out

    def identical_input(self, steps, barrier):
        """the same array every call (rounds 1-3's timed loop): every shortcut a context has is confirmed"""
        torch = self.torch
        self.alternate = False
        self.d_in = self.pair[0]
        for _ in range(3):
            self.step()
        barrier()
        h0, m0 = self.dc.spec_stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        h1, m1 = self.dc.spec_stats()
        self.alternate = True
        self.step()
        self.step()
        return {"ms_per_step": round(ms, 4), "gbps": round(self.n * self.esz / (ms * 1e-3) / 1e9, 2),
                "codebook_speculation": {"hits": h1 - h0, "misses": m1 - m0},
                "note": "the same array every call; informational — `value` is measured on alternating realisations"}

    def per_step(self, steps):
        """SURVEY.md 8(d)'s protocol beside the loop mean: every step timed on its own (a step ends in finish(): a completed call) — on the
        host clock and between two HIP events on the launch stream — and the medians of >= 20 of them"""
        torch = self.torch
        wall, dev = [], []
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for i in range(steps):
            t0 = time.perf_counter()
            evs[i][0].record()
            self.step()
            evs[i][1].record()
            wall.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        dev = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        wall.sort()
        return {"steps": steps, "median_ms_host_clock": round(1e3 * wall[len(wall) // 2], 4), "median_ms_hip_events": round(dev[len(dev) // 2], 4),
                "min_ms_host_clock": round(1e3 * wall[0], 4),
                "note": "each step on its own, outside the timed loop (two event records per step cost a few microseconds of host time)"}

    def stage_profile(self, reps=10):
        """per-stage kernel time, HIP events on the launch stream (outside the timed loop)"""
        self.dc.set_profiling(True)
        acc = {}
        for _ in range(reps):
            self.step()
            for k, v in self.dc.stage_times().items():
                acc[k] = acc.get(k, 0.0) + v / reps
        self.dc.set_profiling(False)
        return acc

    def stream_predictor(self):
        """predictor id in the payload's header (sz3hip_format.h): 0 plain Lorenzo, 1 interpolation, 2 block-composed"""
        return int(self.d_payload[11].item())

    def verify_and_time_decode(self, psize):
        torch = self.torch
        psize = self.step()  # (the payload buffer holds the LAST step's stream — the profiled steps ran since `psize` was returned, on the other realisation)
        d_out = torch.empty_like(self.d_in)
        self.dc.decompress(self.d_payload.data_ptr(), psize, d_out.data_ptr(), self.stream)
        torch.cuda.synchronize()
        max_err = float((d_out.double() - self.d_in.double()).abs().max().item())
        for _ in range(3):  # (the error check above left the allocator and the caches in another state)
            self.dc.decompress(self.d_payload.data_ptr(), psize, d_out.data_ptr(), self.stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            self.dc.decompress(self.d_payload.data_ptr(), psize, d_out.data_ptr(), self.stream)
        torch.cuda.synchronize()
        return max_err, (time.perf_counter() - t0) / 20 * 1e3


def live_traffic(args, kernel_prefixes, child_steps=6, child_warmup=4):
    """HBM traffic of the dominant kernel and of the whole step, measured in THIS run: two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE
    need separate passes: /opt/skills/guides/MI355X_MICROARCH.md, "rocprofv3 PMC slots") over a child that runs the same workload's
    warm-up + steps and nothing else. Per launch: mean over the child's dispatches of the kernel; FETCH_SIZE doubled (the guide's
    gfx950 correction for wide coalesced reads), WRITE_SIZE as reported, counters in KB. None when rocprofv3 is not on PATH or a pass fails."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    base = [sys.executable, os.path.abspath(__file__), "--traffic-child", "--steps", str(child_steps), "--warmup", str(child_warmup),
            "--algo", args.algo, "--eb", repr(args.eb), "--dtype", args.dtype, "--field", args.field, "--size", str(args.size)]
    if args.shape:
        base += ["--shape", args.shape]
    tmp = tempfile.mkdtemp(prefix="sz3_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    per = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            r = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--"] + base,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            rows = []
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") != ctr:
                    continue
                rows.append((int(row.get("Dispatch_Id", 0)), row["Kernel_Name"].replace("void ", "").split("(")[0], float(row["Counter_Value"])))
            rows.sort()
            per[ctr] = rows
        def dominant(rows):
            return [i for i, (_, name, _) in enumerate(rows) if name.startswith(tuple(kernel_prefixes))]
        out = {}
        doms = {c: dominant(per[c]) for c in per}
        if any(len(v) < child_steps for v in doms.values()):
            return None
        vals = {}
        for c, rows in per.items():
            idx = doms[c][-child_steps:]                 # the last `child_steps` launches of the dominant kernel = the timed-like steps
            vals[c + "_kernel"] = sum(rows[i][2] for i in idx) / len(idx)
            vals[c + "_step"] = sum(v for (_, _, v) in rows[idx[0]:]) / len(idx)   # everything from the first of them on, per step
            vals["kernel_name"] = rows[idx[-1]][1]
        out["kernel"] = vals["kernel_name"]
        out["per_launch_bytes"] = int((2 * vals["FETCH_SIZE_kernel"] + vals["WRITE_SIZE_kernel"]) * 1024)
        out["per_step_bytes"] = int((2 * vals["FETCH_SIZE_step"] + vals["WRITE_SIZE_step"]) * 1024)
        out["raw_KB"] = {k: round(v, 1) for k, v in vals.items() if isinstance(v, float)}
        return out
    except Exception:  # noqa: BLE001 - a failing profiler pass must not lose the bench line
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def rooflines(w, acc, psize, ms_per_step, traffic_key, live=None):
    """the dominant kernel against ITS compulsory bytes; the whole path's algorithmic bytes over the whole step"""
    raw = w.n * w.esz
    out = {}
    k1_ms = acc.get("k1_kernel", acc.get("lorenzo_quant_hist", float("nan")))
    # device time of one step: from the first launch of stage 1 to the end of the last kernel of stage 2, HIP events on the caller's
    # stream (a wide alphabet's code book is built beside the encoder on a stream of its own: the stages' own times overlap and do not add up)
    kernels_ms = acc.get("step_span", sum(acc.get(k, 0.0) for k in ("tuner", "lorenzo_quant_hist", "codebook", "encode", "assemble")))
    # (the span is measured in separate, profiled steps — every stage bracketed by HIP events, which costs a few microseconds per
    # record; the device time of a step cannot exceed the timed loop's wall time per step, which contains it)
    span_profiled = kernels_ms
    kernels_ms = min(kernels_ms, ms_per_step)
    traffic = None
    traffic_source = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if live:
        traffic = live["per_step_bytes"] if w.algo != "lorenzo" else live["per_launch_bytes"]
        traffic_source = "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run (2 x FETCH + WRITE, KB x 1024), kernel " + live["kernel"]
    elif traffic_key and os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(traffic_key)
            traffic_source = "profiles/pmc_traffic.json (committed; not measured in this run: rocprofv3 not on PATH, a pass failed, or --no-live-traffic)"
        except Exception:
            traffic = None
    if w.algo == "composed" and w.stream_predictor() == 0:
        # the selection pass found (next to) no block for regression: the array went to the plain Lorenzo kernel
        k_bytes = w.n * (2 * w.esz + 2)  # input read by the selection pass and again by the predictor kernel, one 2-byte code per element
        kname = "stage 1 = k_blk_select (choice of every block, a block per lane) + k_lorenzo_quant_march: the selection handed the array to the plain Lorenzo path"
        note = "read sizeof(T) twice, write 2 B of codes per element"
        k1_ms = acc.get("tuner", 0.0) + acc.get("lorenzo_quant_hist", float("nan"))
    elif w.algo == "composed":
        k_bytes = w.n * (2 * w.esz + 2)  # input in (selection pass, coding pass), one 2-byte code per element (the regression blocks' lattice values are a fraction)
        kname = "stage 1 = k_blk_select + k_blk_fit (regression blocks) + k_blk_rows (Lorenzo elements) + side information (block-composed predictor, sz3hip_regress.hip)"
        note = "read sizeof(T) twice, write 2 B of codes per element"
        k1_ms = acc.get("tuner", 0.0) + acc.get("lorenzo_quant_hist", float("nan"))
    elif w.algo == "lorenzo":
        stats = w.dc.stats()
        code_bytes = 1 if stats.get("narrow_codes") else 2
        k_bytes = w.n * (w.esz + code_bytes)  # the predictor kernel reads the array once and writes one code per element
        kname = ("k_lorenzo_quant_march (HIP events around its launch)" if "k1_kernel" in acc else "k_lorenzo_quant (stage lorenzo_quant_hist)")
        note = "read sizeof(T) + write %d B of codes per element" % code_bytes
    else:
        k_bytes = w.n * (2 * w.esz + 2)  # working copy in + reconstruction out + one 2-byte code per point, each once
        kname = "stage 1 = copy + interpolation passes + code histogram (a multi-kernel stage: see profiles/)"
        note = "read sizeof(T) + write sizeof(T) of reconstruction + 2 B of codes per element, each once"
    algo_bytes = raw + psize  # read sizeof(T) + write sizeof(T)/ratio per element
    # `achieved` = the PATH's algorithmic bytes per launch (one launch = the whole volume) over the dominant kernel's
    # duration; the kernel's own compulsory bytes (what it must read and write itself) are priced beside it
    ach = algo_bytes / (k1_ms * 1e-3) / 1e9
    kach = k_bytes / (k1_ms * 1e-3) / 1e9
    out["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                       "algorithmic_bytes_per_launch": int(algo_bytes),
                       "bytes_note": "SURVEY.md 8(d): sizeof(T) read + sizeof(T) / ratio written per element (input + payload), one launch = the whole volume",
                       "kernel_ms": round(k1_ms, 4),
                       "frac_kernel_compulsory": round(kach / HBM_PEAK_GBS, 4), "compulsory_bytes_per_launch": int(k_bytes),
                       "compulsory_note": note}
    pach = algo_bytes / (ms_per_step * 1e-3) / 1e9
    out["roofline_path"] = {"bound": "hbm", "what": "whole step (all kernels + launch gaps + the final sync), algorithmic bytes = input + payload",
                            "achieved": round(pach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(pach / HBM_PEAK_GBS, 4),
                            "algorithmic_bytes_per_step": int(algo_bytes),
                            "traffic": live["per_step_bytes"] if live else None,
                            "traffic_over_algorithmic": round(live["per_step_bytes"] / float(algo_bytes), 3) if live else None}
    out["kernels_ms"] = round(kernels_ms, 4)
    out["kernels_ms_note"] = ("min(device span of a profiled step = %.4f ms, wall time per step of the timed loop = %.4f ms)" % (span_profiled, ms_per_step))
    out["frac_read_peak_all_kernels"] = round(raw / (kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return out


def cpu_baseline(w):
    """the reference's CPU path on this box's host cores (checker code, never the product): one thread, and all cores
    through its OpenMP slab path (SZ_compress_OMP, api/impl/SZImplOMP.hpp:16-117) in a child process so that
    OMP_NUM_THREADS takes effect"""
    from oracle_binding import have_ref, make_config, oracle_compress, ref_compress
    from oracle_binding import ALGO_INTERP as O_INTERP
    from oracle_binding import ALGO_INTERP_LORENZO as O_TUNED
    shape, eb, algo = w.shape, w.eb, w.algo
    raw = w.n * w.esz

    def oconf(openmp):
        if algo in ("lorenzo", "composed"):
            return make_config(shape, abs_eb=eb, lorenzo=True, regression=algo == "composed", openmp=openmp)
        return make_config(shape, algo=O_TUNED if algo == "interp" else O_INTERP, abs_eb=eb, regression=True, openmp=openmp)
    if have_ref():
        blob, sec = ref_compress(w.a, oconf(False), timing=True)
        kind = "reference"
    else:
        t0 = time.perf_counter()
        blob = oracle_compress(w.a, oconf(False))
        sec = time.perf_counter() - t0
        kind = "port"
    what = {"lorenzo": "ALGO_LORENZO_REG (Lorenzo only)", "interp": "ALGO_INTERP_LORENZO (default)", "interp-notune": "ALGO_INTERP (cubic)",
            "composed": "ALGO_LORENZO_REG (Lorenzo + regression)"}[algo]
    out = {"value": round(raw / sec / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": kind,
           "sample": "whole %dx%dx%d volume, SZ_compress<T> %s abs %g, single thread, %.2f s" % (shape[0], shape[1], shape[2], what, eb, sec),
           "ratio": round(raw / float(len(blob)), 4), "host_cpus": os.cpu_count()}
    if have_ref():
        try:
            import psutil
            phys = psutil.cpu_count(logical=False) or os.cpu_count()
        except Exception:
            phys = os.cpu_count()
        threads = int(min(phys, shape[0]))
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="spread")
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "ref_all_cores.py"), ",".join(str(v) for v in shape),
                                w.dtype, algo, repr(eb)], env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            m = json.loads(line)
            out["all_cores"] = {"value": round(raw / m["sec"] / 1e9, 4), "unit": "GB/s", "cores": threads, "kind": "reference",
                                "sample": "same volume, conf.openmp = true (SZ_compress_OMP: %d slabs, one per thread), OMP_NUM_THREADS=%d "
                                          "(physical cores, capped at dims[0]), best of 2, %.3f s" % (threads, threads, m["sec"]),
                                "ratio": round(raw / float(m["bytes"]), 4)}
        except Exception as e:  # noqa: BLE001 - a missing all-cores leg must not lose the bench line
            out["all_cores"] = {"error": repr(e)[:200]}
    return out


def c1_numbers(torch, sz3_amd, dev, local_rank, steps=20, default_algo=False):
    """C1 on the device path: the first 2^20 values of the C2 field as a 1-D array, Lorenzo + regression per block of 128 values
    (sz3hip_regress.hip, k_blkn_*), abs 1e-3; device-resident in -> device payload. 4 MB: the step is bound by its launches.
    default_algo: the same array under the reference's default algorithm — in 1-D its tuner takes the set [Lorenzo-1, Lorenzo-2]
    in blocks of 128 (SZAlgoInterp.hpp:232-282; round 4)."""
    from fields import field1d
    n = 1 << 20
    a = field1d(n, np.float32)
    d_in = torch.from_numpy(a).to(dev)
    conf = sz3_amd.Config(n)
    if not default_algo:
        conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG  # (defaults: lorenzo = regression = 1, blockSize 128)
    conf.errorBoundMode = sz3_amd.EB_ABS
    conf.absErrorBound = 1e-3
    dc = sz3_amd.DeviceCompressor(n, np.float32, device=local_rank)
    cap = max(dc.payload_bound(n), dc.payload_bound_conf(conf))
    d_pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def comp():
        dc.stage1(conf, d_in.data_ptr(), st)
        dc.stage2(d_pl.data_ptr(), cap, st)
        return dc.finish(st)

    for _ in range(3):
        size = comp()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        size = comp()
    torch.cuda.synchronize()
    tc = (time.perf_counter() - t0) / steps
    for _ in range(2):
        dc.decompress(d_pl.data_ptr(), size, d_out.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        dc.decompress(d_pl.data_ptr(), size, d_out.data_ptr(), st)
    torch.cuda.synchronize()
    td = (time.perf_counter() - t0) / steps
    err = float((d_out.double() - d_in.double()).abs().max())
    hdr = bytes(d_pl[:160].cpu().numpy())
    if default_algo:
        return {"config": "C1's array under the default algorithm (ALGO_INTERP_LORENZO: in 1-D the tuner's Lorenzo-1 + Lorenzo-2 set, blocks of 128), abs errBound=1e-3, 1 GPU",
                "value": round(a.nbytes / tc / 1e9, 3), "unit": "GB/s", "steps": steps, "ms_per_step": round(tc * 1e3, 4),
                "ratio": round(a.nbytes / float(size), 4), "max_abs_err": err, "err_bound_ok": bool(err <= 1e-3),
                "stream_predictor": int(hdr[11]), "block_predictor_set": int(hdr[148]),  # (1 Lorenzo-1 | 2 Lorenzo-2 | 4 regression)
                "decompress_device": {"ms": round(td * 1e3, 4), "gbps": round(a.nbytes / td / 1e9, 2)},
                "tuner": dc.tuner_report(),
                "note": "oracle (= the reference) on this array: ratio 7.29 after zstd, 5.61 with Lorenzo-1 alone; the step includes the tuner"}
    return {"config": "C1: 1D float32 2^20 values, ALGO_LORENZO_REG defaults (Lorenzo + regression per block of 128), abs errBound=1e-3, 1 GPU",
            "value": round(a.nbytes / tc / 1e9, 3), "unit": "GB/s", "steps": steps, "ms_per_step": round(tc * 1e3, 4),
            "ratio": round(a.nbytes / float(size), 4), "max_abs_err": err, "err_bound_ok": bool(err <= 1e-3),
            "stream_predictor": int(hdr[11]),  # (SZH1 header: 2 = block-composed)
            "decompress_device": {"ms": round(td * 1e3, 4), "gbps": round(a.nbytes / td / 1e9, 2)},
            "note": "4 MB per call: launch-bound (eight kernels a step since round 6, twenty before: DESIGN.md, Round 6: small arrays)"}


def device_field4d(torch, dev, t_lo, nt, edge, seed):
    """C5's field (tests/fields.py field4d's formula) made ON the device, one time step at a time: a rank's 12 or 13 steps of 500^3
    are 1.6e9 values — not something to build with numpy meshgrids on the host. The noise comes from torch's generator (seeded)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty((nt, edge, edge, edge), dtype=torch.float32, device=dev)
    ar = torch.arange(edge, dtype=torch.float64, device=dev)
    cy = torch.cos(2 * np.pi * ar / 48)[None, :, None]
    sz = torch.sin(2 * np.pi * ar / 64)[:, None, None]
    for i in range(nt):
        t = float(t_lo + i)
        sx = torch.sin(2 * np.pi * (ar + 0.5 * t) / 32)[None, None, :]
        plane = (sz * cy) * sx * (1 + 0.01 * t)
        plane += torch.randn(plane.shape, generator=g, device=dev, dtype=torch.float64) * 2e-3
        out[i] = plane.to(torch.float32)
        del plane
    return out


def multi_gpu_extras(torch, sz3_amd, dev, local_rank, rank, world, comm, dist, one_gpu, timed, steps):
    """BASELINE.json configs[3] and [4] as the ranks of THIS run see them (every rank calls this; rank 0 keeps the numbers).
    C4: 1024^3 f64, Lorenzo + regression, abs 1e-6, 8 slabs of 128 planes (api/impl/SZImplOMP.hpp:48-50) — rank r codes slab r, the
    code histogram all-reduced between the stages like the headline's; both fields of SURVEY.md 8(d): C4a (regression chosen in a share
    of the blocks: the block stream) and C4b (noise above the bound: the selection hands the array to the plain stream).
    C5: 100 x 500^3 f32, REL 1e-3, 8 slabs of 12 / 13 time steps: the value range all-reduced first (SZImplOMP.hpp:57-69), then the 4-D
    Lorenzo stream with the histogram all-reduce, decompressed and checked against the bound.
    With N < 8 ranks the first N slabs are coded (weak scaling, like the headline); at N = 8 the legs ARE configs[3] / [4]."""
    from sz3_amd import distributed as D
    small = os.environ.get("SZ3_BENCH_EXTRA_SCALE") == "small"   # (tests: the same legs at sizes a shared GPU takes in seconds)
    cpu = torch.device("cpu")
    red_dev = cpu if one_gpu else dev

    def red(x, op):
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=op)
        return float(t.item())

    res = {}
    c4_shape = (12, 96, 120) if small else (128, 1024, 1024)
    def agreed(ok):
        # every rank reaches this point whatever happened to it before: a leg whose set-up failed anywhere is skipped everywhere
        # (the timed part is full of collectives — a rank that is not there would leave the others waiting)
        return red(1.0 if ok else 0.0, dist.ReduceOp.MIN) >= 1.0

    for name, field in (("C4a", "c4a"), ("C4b", "default")):
        w, why = None, ""
        try:
            w = Workload(torch, sz3_amd, dev, local_rank, rank, c4_shape, "f64", "composed", 1e-6, comm=comm, dist=dist if comm is None else None, field=field)
        except Exception as e:  # noqa: BLE001
            why = repr(e)[:300]
        if not agreed(w is not None):
            res[name] = {"error": "set-up failed on a rank: " + (why or "another rank's")}
            del w
            torch.cuda.empty_cache()
            continue
        try:
            el, ps = timed(w, steps, 2)
            err, dec_ms = w.verify_and_time_decode(ps)
            el = red(el, dist.ReduceOp.MAX)
            total = red(float(ps), dist.ReduceOp.SUM)
            err = red(err, dist.ReduceOp.MAX)
            dec_ms = red(dec_ms, dist.ReduceOp.MAX)
            raw = w.n * 8
            res[name] = {"ms_per_step": round(1e3 * el / steps, 4), "value": round(world * raw / (el / steps) / 1e9, 3), "unit": "GB/s",
                         "ratio": round(world * raw / total, 4), "max_abs_err": err, "err_bound_ok": bool(err <= 1e-6),
                         "decompress_device": {"ms": round(dec_ms, 4), "gbps": round(world * raw / (dec_ms * 1e-3) / 1e9, 2)},
                         "stream_predictor_rank0": w.stream_predictor(), "slab": list(c4_shape), "steps": steps}
            del w
        except Exception as e:  # noqa: BLE001 - (a rank that fails here leaves the others in a collective: the legs are kept simple)
            res[name] = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
    out = {"C4_8slab": {"config": "configs[3]: 3D float64 1024^3 as 8 slabs of 128x1024x1024 (rank r = slab r; %d of 8 coded here), ALGO_LORENZO_REG "
                                  "Lorenzo + regression per 6^3 block, abs errBound=1e-6, histogram all-reduce between the stages" % min(world, 8),
                        "fields": res, "scaling": "weak"}}
    # ---- C5 ----
    c5 = None
    why5 = ""
    try:
        nt_total, edge = (100, 16) if small else (100, 500)  # (small: the same 100 time steps — the same 12 / 13 split — of a 16^3 volume)
        lo, hi = D.slab_bounds(nt_total, 8, rank % 8)
        nt = hi - lo
        d_in = device_field4d(torch, dev, lo, nt, edge, 20260928 + rank)
        n = d_in.numel()
        dc = sz3_amd.DeviceCompressor(n, np.float32, device=local_rank)
        stream = torch.cuda.current_stream().cuda_stream
        hist = None
        if comm is None:
            hist = torch.zeros(65536, dtype=torch.int64, device=dev)
            dc.set_histogram(hist.data_ptr())
        mn, mx = dc.minmax(d_in.data_ptr(), n, stream)
        cap = dc.payload_bound(n)
        d_pl = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_out = torch.empty_like(d_in)
        c5 = True
    except Exception as e:  # noqa: BLE001
        why5 = repr(e)[:300]
    if not agreed(c5 is not None):
        out["C5_8slab"] = {"error": "set-up failed on a rank: " + (why5 or "another rank's")}
        torch.cuda.empty_cache()
        return out
    try:
        if comm is not None:
            mns, mxs = comm.allreduce_minmax([mn], [mx], [stream])
            gmn, gmx = mns[0], mxs[0]
        else:
            gmn, gmx = D.allreduce_range(mn, mx, dist, device=cpu)
        eb = 1e-3 * (gmx - gmn)
        conf = sz3_amd.Config(nt, edge, edge, edge)
        conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
        conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, 0
        conf.errorBoundMode = sz3_amd.EB_ABS   # (REL resolved with the GLOBAL range, as SZ_compress_OMP does before it splits)
        conf.absErrorBound = eb

        def step():
            dc.stage1(conf, d_in.data_ptr(), stream)
            if comm is not None:
                comm.allreduce_histogram([dc], [stream])
            else:
                dist.all_reduce(hist, op=dist.ReduceOp.SUM)
            dc.stage2(d_pl.data_ptr(), cap, stream)
            return dc.finish(stream)

        for _ in range(2):
            ps = step()
        dist.barrier()
        torch.cuda.synchronize()
        k = max(2, steps // 2)
        t0 = time.perf_counter()
        for _ in range(k):
            ps = step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        dc.decompress(d_pl.data_ptr(), ps, d_out.data_ptr(), stream)
        torch.cuda.synchronize()
        err = 0.0
        for i in range(nt):   # (step by step: a float64 copy of the slab would be 13 GB)
            err = max(err, float((d_out[i].double() - d_in[i].double()).abs().max().item()))
        t0 = time.perf_counter()
        for _ in range(3):
            dc.decompress(d_pl.data_ptr(), ps, d_out.data_ptr(), stream)
        torch.cuda.synchronize()
        dec_ms = (time.perf_counter() - t0) / 3 * 1e3
        el = red(el, dist.ReduceOp.MAX)
        total = red(float(ps), dist.ReduceOp.SUM)
        raw_all = red(float(n * 4), dist.ReduceOp.SUM)
        steps_of = torch.zeros(world, dtype=torch.float64, device=red_dev)  # every rank's slab thickness (SZImplOMP.hpp:48-50: 12, 13, 12, 13, ...)
        steps_of[rank] = nt
        dist.all_reduce(steps_of, op=dist.ReduceOp.SUM)
        err = red(err, dist.ReduceOp.MAX)
        dec_ms = red(dec_ms, dist.ReduceOp.MAX)
        out["C5_8slab"] = {"config": "configs[4]: 4D float32 100x500x500x500 as 8 slabs of 12 / 13 time steps (rank r = slab r; %d of 8 coded here), "
                                     "Lorenzo, REL errBound=1e-3 of the all-reduced value range, histogram all-reduce between the stages" % min(world, 8),
                           "ms_per_step": round(1e3 * el / k, 4), "value": round(raw_all / (el / k) / 1e9, 3), "unit": "GB/s", "steps": k,
                           "ratio": round(raw_all / total, 4), "abs_bound_from_range": eb, "value_range": [gmn, gmx],
                           "max_abs_err": err, "err_bound_ok": bool(err <= eb),
                           "decompress_device": {"ms": round(dec_ms, 4), "gbps": round(raw_all / (dec_ms * 1e-3) / 1e9, 2)},
                           "slab_rank0": [nt, edge, edge, edge], "slab_steps_by_rank": [int(v) for v in steps_of.tolist()], "scaling": "weak",
                           "data": "synthetic: fields.field4d's formula evaluated on the device, noise from torch's generator"}
        del d_in, d_out, d_pl, dc
    except Exception as e:  # noqa: BLE001
        out["C5_8slab"] = {"error": repr(e)[:300]}
    torch.cuda.empty_cache()
    return out


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch_as_ranks(args.gpus)

    import torch
    import sz3_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU implementation)")
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE); refusing to print a line for "
                         "another GPU count than asked" % (args.gpus, world))
    # SZ3_BENCH_ONE_GPU=1 (testing the multi-rank code path on a 1-GPU box): every rank uses cuda:0; RCCL cannot put two
    # ranks on one device, so the histogram goes through gloo there (the line says so)
    one_gpu = os.environ.get("SZ3_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    elif world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d ranks but %d visible GPUs" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    comm = None
    exchange = "none (one rank)"
    force_comm = os.environ.get("SZ3_BENCH_FORCE_COMM") == "1" and "RANK" in os.environ  # (one rank under a launcher: the RCCL path end to end)
    if world > 1 or force_comm:
        import torch.distributed as dist
        from sz3_amd import distributed as D
        if one_gpu:
            dist.init_process_group(backend="gloo")
            exchange = "gloo all_reduce (SZ3_BENCH_ONE_GPU: ranks share one GPU, RCCL needs one GPU per rank)"
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # the launcher's own group: barrier + max-over-ranks timing
            comm = D.init_comm(dist, local_rank)
            seen = comm.size
            if seen != world:
                raise SystemExit("bench.py: RCCL reports %d ranks, expected %d" % (seen, world))
            exchange = "RCCL ncclAllReduce(u64[65536], sum) inside libsz3hip.so, %d ranks" % seen

    S = args.size
    shape = tuple(int(v) for v in args.shape.split(",")) if args.shape else (S, S, S)
    w = Workload(torch, sz3_amd, dev, local_rank, rank, shape, args.dtype, args.algo, args.eb, comm=comm, dist=dist if (world > 1 and comm is None) else None,
                 field=args.field)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(wl, steps, warmup):
        for _ in range(warmup):
            ps = wl.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            ps = wl.step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return el, ps

    elapsed, psize = timed(w, args.steps, args.warmup)
    if args.traffic_child:  # (a rocprofv3 --pmc pass of live_traffic: the counters were what this process was for)
        torch.cuda.synchronize()
        return
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev if not one_gpu else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
        ps = torch.tensor([psize], dtype=torch.int64, device=dev if not one_gpu else "cpu")
        dist.all_reduce(ps, op=dist.ReduceOp.SUM)
        total_payload = int(ps.item())
    else:
        total_payload = psize
    barrier()

    raw_bytes = w.n * w.esz
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * raw_bytes / (elapsed / args.steps) / 1e9
    ratio = world * raw_bytes / float(total_payload)
    h_spec, m_spec = w.dc.spec_stats()
    per_step = w.per_step(max(20, args.steps))  # (every rank: a step of a multi-rank run holds a collective)
    acc = w.stage_profile()
    stats = w.dc.stats()
    max_err, dec_ms = w.verify_and_time_decode(psize)

    out = None
    if rank == 0:
        is_c2 = args.algo == "lorenzo" and args.dtype == "f32" and tuple(shape) == (512, 512, 512) and args.eb == 1e-3
        is_c3 = args.algo == "interp" and args.dtype == "f32" and tuple(shape) == (512, 512, 512) and args.eb == 1e-4
        # (the PMC passes of tools/pmc_blk.sh were taken on the benchmark field, where the selection hands the array to the plain path)
        is_c4c = (args.algo == "composed" and args.dtype == "f64" and tuple(shape) == (128, 1024, 1024) and args.eb == 1e-6 and args.field == "default"
                  and w.stream_predictor() == 0)
        is_c4a = (args.algo == "composed" and args.dtype == "f64" and tuple(shape) == (128, 1024, 1024) and args.eb == 1e-6 and args.field == "c4a"
                  and w.stream_predictor() == 2)  # (tools/pmc_blk.sh with FIELD=c4a: profiles/r03_pmc_blk_c4a.txt)
        out = {
            "metric": "compression throughput GB/s + ratio at fixed abs errBound, 512^3 f32",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s: 3D %s %dx%dx%d synthetic field per GPU, %s predictor, abs errBound=%g, "
                                   "device-resident in -> device-resident Huffman payload"
                                   % ("C2" if is_c2 else "C3" if is_c3 else "custom",
                                      "float32" if args.dtype == "f32" else "float64", shape[0], shape[1], shape[2],
                                      {"lorenzo": "Lorenzo", "interp": "ALGO_INTERP_LORENZO (auto-tuned interpolation)",
                                       "interp-notune": "interpolation (ALGO_INTERP)", "composed": "Lorenzo + regression per block"}[args.algo], args.eb),
                       "parallelism": "slab%d" % world, "algo": "HIP_LORENZO(16)" if args.algo in ("lorenzo", "composed") else "HIP_INTERP(17)", "eb": args.eb,
                       "exchange": exchange},
            "ratio": round(ratio, 4), "max_abs_err": max_err, "err_bound_ok": bool(max_err <= args.eb),
            "payload_bytes_rank0": int(psize),
            "ms_per_step_median": per_step,
            "decompress_device": {"ms": round(dec_ms, 4), "gbps": round(raw_bytes / (dec_ms * 1e-3) / 1e9, 2)},
            # SURVEY.md 8(d), decompression: read sizeof(T) / ratio, write sizeof(T) per element — the whole decode (all its kernels) against the peak
            "roofline_decompress": {"bound": "hbm", "what": "whole decode (all kernels), algorithmic bytes = payload read + array written",
                                    "achieved": round((raw_bytes + psize) / (dec_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round((raw_bytes + psize) / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "algorithmic_bytes": int(raw_bytes + psize), "ms": round(dec_ms, 4)},
            "outliers": {"value": stats["n_value_outliers"], "delta": stats["n_delta_outliers"]},
            "narrow_codes": stats.get("narrow_codes"),
            "stage_ms": {k: round(v, 4) for k, v in acc.items()},
            "tuner": w.dc.tuner_report() if args.algo == "interp" else None,
            "codebook_speculation": {"hits": h_spec, "misses": m_spec,
                                     "note": "round 6: a one-byte Lorenzo stream of >= 2^22 elements (C2) is coded with a book built from a SAMPLE of the array "
                                             "inside stage 1's own launch (a function of the input alone): nothing is speculated, nothing verified, hits = misses = 0, "
                                             "and `value`, `value_deterministic` and a context's second call are the same path. Other streams (wide alphabets: C3, the C4 "
                                             "slab) still pack with the previous call's book while this call's is built beside the encoder; a miss repeats the encoder"},
        }
        if args.algo == "composed":
            pid = w.stream_predictor()
            out["block_selection"] = {"stream_predictor": pid,
                                      "note": ("the selection pass found fewer than 1/4096 of the blocks choosing another predictor than Lorenzo-1: "
                                               "plain Lorenzo stream" if pid == 0 else "block-composed stream (selection bits + regression coefficients)")}
        live = None
        if world == 1 and not args.no_live_traffic:
            prefixes = {"lorenzo": ["k_lorenzo_quant_march", "k_lorenzo_quant"], "composed": ["k_blk_select"],
                        "interp": ["k_interp_level", "k_interp_pass", "k_interp_vec"], "interp-notune": ["k_interp_level", "k_interp_pass", "k_interp_vec"]}[args.algo]
            if args.algo == "lorenzo":  # (one stage-1 launch per step; the other algorithms' stages are multi-kernel: committed profiles)
                live = live_traffic(args, prefixes)
        out.update(rooflines(w, acc, psize, ms_per_step,
                             "lorenzo_quant_hist_hbm_bytes_per_launch" if is_c2 else "c3_stage1_hbm_bytes_per_step" if is_c3
                             else "c4_composed_stage1_hbm_bytes_per_step" if is_c4c
                             else "c4a_composed_stage1_hbm_bytes_per_step" if is_c4a else None, live=live))
        out["input"] = ("two realisations of the field (noise seeds 20260928 + rank, 20261928 + rank) alternate call by call in the timed loop "
                        "and in every other leg except `identical_input`")

    if rank == 0 and world == 1 and not args.no_cold:
        out["identical_input"] = w.identical_input(args.steps, barrier)
        out["cold"] = w.cold_numbers(max(5, args.steps // 2), barrier)
        # what a caller of the reference's boundary gets (sz3hip_ctx_set_deterministic: SZ_compress<T>, sz3c, the CLI, the HDF5 filter) and
        # what a context's first call gets, as values of the line's own metric
        out["value_deterministic"] = round(raw_bytes / (out["cold"]["deterministic_payloads_ms"] * 1e-3) / 1e9, 3)
        out["ms_per_step_deterministic"] = out["cold"]["deterministic_payloads_ms"]
        out["value_cold"] = round(raw_bytes / (out["cold"]["first_call_ms"] * 1e-3) / 1e9, 3)
        out["ms_per_step_cold"] = out["cold"]["first_call_ms"]
        # What `value` also leaves out, in the other direction: a producer with a SERIES of arrays keeps two contexts in flight on two
        # streams — stage 1 of one call (bound by memory) runs beside stage 2 of the other (bound by instruction issue). Not `value`:
        # a step of `value` is one call, finished before the next begins.
        try:
            w2 = Workload(torch, sz3_amd, dev, local_rank, rank + 1, shape, args.dtype, args.algo, args.eb, field=args.field)
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            pair = [w, w2]

            def two(n):
                pending = [False, False]
                for i in range(n):
                    k = i & 1
                    st = streams[k].cuda_stream
                    if pending[k]:
                        pair[k].dc.finish(st)
                    src = pair[k].pair[(i >> 1) & 1]  # (each context alternates its two realisations too)
                    pair[k].dc.stage1(pair[k].conf, src.data_ptr(), st)
                    pair[k].dc.stage2(pair[k].d_payload.data_ptr(), pair[k].cap, st)
                    pending[k] = True
                for k in range(2):
                    if pending[k]:
                        pair[k].dc.finish(streams[k].cuda_stream)

            two(8)
            barrier()
            t0 = time.perf_counter()
            two(4 * args.steps)
            torch.cuda.synchronize()
            per = (time.perf_counter() - t0) / (4 * args.steps)
            out["two_contexts_in_flight"] = {"ms_per_call": round(1e3 * per, 4), "gbps": round(raw_bytes / per / 1e9, 2),
                                             "note": "two contexts alternating on two streams, the same workload (another realisation of the field on "
                                                     "the second): one call's stage 1 beside the other's stage 2; informational, not `value`"}
            del w2
        except Exception as e:  # (memory for a second set of buffers, mostly)
            out["two_contexts_in_flight"] = {"error": str(e)[:200]}
    # ---- host end-to-end (PCIe + zstd inclusive; informational) ----
    if rank == 0 and world == 1 and not args.no_host_e2e:
        best_c = best_d = 0.0
        keep = []  # (the previous call's arrays stay alive: unmapping 0.5 GB is not part of the next call)
        for _ in range(4):  # the first call creates the host API's cached contexts, pinned staging buffers and host threads
            t0 = time.perf_counter()
            res_c = sz3_amd.compress(w.a, w.conf)
            t1 = time.perf_counter()
            res_d = sz3_amd.decompress(res_c[0], w.npdt, shape)
            t2 = time.perf_counter()
            blob, hratio = res_c
            dec = res_d[0]
            keep.append((res_c, res_d))
            if len(keep) > 2:
                keep.pop(0)
            best_c = max(best_c, raw_bytes / (t1 - t0) / 1e9)
            best_d = max(best_d, raw_bytes / (t2 - t1) / 1e9)
        del keep
        # the same two calls into buffers the caller keeps from call to call (the reference's pre-allocated overloads, api/sz.hpp:43-62,
        # 84-110): a fresh 537 MB array is 131 072 pages touched for the first time inside the device->host copy
        reuse_c = reuse_d = 0.0
        cbuf = np.empty(sz3_amd.compress_bound(w.conf, w.npdt), dtype=np.uint8)
        dbuf = np.empty(w.a.size, dtype=w.npdt)
        cbuf[:] = 0
        dbuf[:] = 0
        for _ in range(3):
            t0 = time.perf_counter()
            blob2, _ = sz3_amd.compress(w.a, w.conf, out=cbuf)
            t1 = time.perf_counter()
            dec2, _ = sz3_amd.decompress(blob2, w.npdt, shape, out=dbuf)
            t2 = time.perf_counter()
            reuse_c = max(reuse_c, raw_bytes / (t1 - t0) / 1e9)
            reuse_d = max(reuse_d, raw_bytes / (t2 - t1) / 1e9)
        out["host_e2e"] = {"compress_gbps": round(best_c, 3), "ratio": round(hratio, 4),
                           "decompress_gbps": round(best_d, 3),
                           "compress_gbps_buffers_reused": round(reuse_c, 3), "decompress_gbps_buffers_reused": round(reuse_d, 3),
                           "max_abs_err": float(np.max(np.abs(dec.astype(np.float64) - w.a.astype(np.float64)))),
                           "identical_with_reused_buffers": bool(np.array_equal(dec2.reshape(-1), dec.reshape(-1)) and np.array_equal(blob2, blob)),
                           "note": "host buffer in -> host SZ3 container out, fresh output arrays every call: a pipeline of pieces on the one GPU (round 5: copy in of "
                                   "piece k + 1 beside the kernels of k beside the copy out + zstd of k - 1; the reference's multi-slab container), the decoded array "
                                   "through a pinned staging ring whose chunks host threads copy on and fault in; best of 4 calls. "
                                   "*_buffers_reused: output buffers the caller allocated once"}
        del cbuf, dbuf

    # ---- the other single-GPU configuration of BASELINE.json, same protocol, as an extra object of the same line ----
    if rank == 0 and world == 1 and not args.no_extra and args.algo == "lorenzo" and args.dtype == "f32" and tuple(shape) == (512, 512, 512):
        try:
            w3 = Workload(torch, sz3_amd, dev, local_rank, rank, shape, "f32", "interp", 1e-4)
            st3 = max(5, args.steps // 2)
            el3, ps3 = timed(w3, st3, 2)
            acc3 = w3.stage_profile(5)
            err3, dec3 = w3.verify_and_time_decode(ps3)
            ms3 = 1e3 * el3 / st3
            c3 = {"config": "C3: 3D float32 512x512x512, ALGO_INTERP_LORENZO (sampling tuner + interpolation), abs errBound=1e-4, 1 GPU",
                  "value": round(raw_bytes / (el3 / st3) / 1e9, 3), "unit": "GB/s", "steps": st3, "ms_per_step": round(ms3, 4),
                  "ratio": round(raw_bytes / float(ps3), 4), "max_abs_err": err3, "err_bound_ok": bool(err3 <= 1e-4),
                  "decompress_device": {"ms": round(dec3, 4), "gbps": round(raw_bytes / (dec3 * 1e-3) / 1e9, 2)},
                  "stage_ms": {k: round(v, 4) for k, v in acc3.items()}, "tuner": w3.dc.tuner_report()}
            c3.update(rooflines(w3, acc3, ps3, ms3, "c3_stage1_hbm_bytes_per_step"))
            if not args.no_cold:
                c3["cold"] = w3.cold_numbers(5, barrier)
            try:  # the same step with the tuner's trials priced the reference's way (sz3hip_ctx_set_tuner_exact: tree + bits + zstd per trial on host threads)
                w3.dc.set_tuner_exact(True)
                elx, psx = timed(w3, 5, 2)
                c3["reference_priced_tuner"] = {"ms_per_step": round(1e3 * elx / 5, 4), "ratio": round(raw_bytes / float(psx), 4), "tuner": w3.dc.tuner_report(),
                                                "note": "est_bytes are then the reference's own compressed trial sizes byte for byte and the decisions the reference's "
                                                        "(tests/test_gpu_tuner.py); the host API's default, a device context's option"}
                w3.dc.set_tuner_exact(False)
            except Exception as e:  # noqa: BLE001
                c3["reference_priced_tuner"] = {"error": repr(e)[:200]}
            out["extra_configs"] = {"C3": c3}
            del w3
        except Exception as e:  # noqa: BLE001 - the headline line must survive a failing extra
            out["extra_configs"] = {"C3": {"error": repr(e)[:300]}}

    # ---- configs[0] of BASELINE.json (C1: 1-D, 2^20 values, ALGO_LORENZO_REG defaults = Lorenzo + regression in blocks of 128) ----
    if rank == 0 and world == 1 and not args.no_extra and args.algo == "lorenzo" and args.dtype == "f32" and tuple(shape) == (512, 512, 512):
        try:
            out.setdefault("extra_configs", {})["C1"] = c1_numbers(torch, sz3_amd, dev, local_rank)
        except Exception as e:  # noqa: BLE001
            out.setdefault("extra_configs", {})["C1"] = {"error": repr(e)[:300]}
        try:
            out["extra_configs"]["C1_default_algorithm"] = c1_numbers(torch, sz3_amd, dev, local_rank, default_algo=True)
        except Exception as e:  # noqa: BLE001
            out["extra_configs"]["C1_default_algorithm"] = {"error": repr(e)[:300]}

    # ---- one C4 slab of a field where the block stream is kept (C4a: regression wins in a share of the blocks): the block-composed
    # predictor both ways — its decoder is the chain of block fronts in one launch (round 4) ----
    if rank == 0 and world == 1 and not args.no_extra and args.algo == "lorenzo" and args.dtype == "f32" and tuple(shape) == (512, 512, 512):
        try:
            w4 = Workload(torch, sz3_amd, dev, local_rank, rank, (128, 1024, 1024), "f64", "composed", 1e-6, field="c4a")
            st4 = max(3, args.steps // 4)
            el4, ps4 = timed(w4, st4, 2)
            err4, dec4 = w4.verify_and_time_decode(ps4)
            raw4 = w4.n * 8
            out.setdefault("extra_configs", {})["C4a_slab"] = {
                "config": "one C4 slab, 128x1024x1024 float64, field C4a (SURVEY 8d), ALGO_LORENZO_REG Lorenzo + regression per 6^3 block, abs errBound=1e-6, 1 GPU",
                "value": round(raw4 / (el4 / st4) / 1e9, 3), "unit": "GB/s", "steps": st4, "ms_per_step": round(1e3 * el4 / st4, 4),
                "ratio": round(raw4 / float(ps4), 4), "max_abs_err": err4, "err_bound_ok": bool(err4 <= 1e-6),
                "stream_predictor": w4.stream_predictor(),
                "decompress_device": {"ms": round(dec4, 4), "gbps": round(raw4 / (dec4 * 1e-3) / 1e9, 2)},
                "note": "round 3: decompress 5.2 ms (181 launches for the chain of block fronts + a final pass); round 4: k_blk_local3v + k_blk_wave3, one launch"}
            del w4
        except Exception as e:  # noqa: BLE001
            out.setdefault("extra_configs", {})["C4a_slab"] = {"error": repr(e)[:300]}

    # ---- N > 1: BASELINE.json's multi-GPU configurations (configs[3], configs[4]) as extra objects of the same line ----
    if world > 1 and not args.no_extra and args.algo == "lorenzo" and args.dtype == "f32":
        extras = multi_gpu_extras(torch, sz3_amd, dev, local_rank, rank, world, comm, dist, one_gpu, timed, max(3, args.steps // 4))
        if rank == 0:
            out.setdefault("extra_configs", {}).update(extras)
    # ---- CPU baseline on this box's host cores (rank 0, N=1 only; checker code, never the product) ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
