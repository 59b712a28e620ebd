#!/usr/bin/env python3
"""randomised sweep over DATA the other sweeps do not draw (they use smooth synthetic fields): white noise, constants, zeros, steps, spikes of
1e30, bounds far below the values' spacing, magnitudes of 1e30 and 1e-38 (f32 denormals), integer-valued floats, negative zeros, NaN / Inf.
Per case, with ABS or REL bounds and a random algorithm:
  (1) this library's own payload: decompress(compress(a)) within the bound, non-finite values where they were;
  (2) stock format (one zstd frame): the container's bytes against the REFERENCE's (oracle/_ref/libsz3ref.so; the oracle's restatement when that
      build is absent), and this library's reading of the reference's container against the reference's own, bit for bit.
SEED, N from the environment; exit code = failures."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np, sz3_amd
from oracle_binding import (ALGO_INTERP, ALGO_INTERP_LORENZO, ALGO_LORENZO_REG, ALGO_NOPRED, EB_REL, EB_PSNR, EB_L2NORM, EB_ABS_AND_REL, EB_ABS_OR_REL, make_config, oracle_compress,
                            have_ref, ref_compress, ref, oracle, _dtype_id)
os.environ["SZ3HIP_STOCK_ONE_FRAME"] = "1"
os.environ.pop("SZ3HIP_TUNER_EXACT", None)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
L = sz3_amd.lib()
USE_REF = have_ref() and not os.environ.get("NO_REF")
KINDS = ["noise", "const", "constrel", "zeros", "steps", "spikes", "tight", "huge", "denormal", "integers", "negzero", "nonfinite", "ramp", "sparse"]
ONLY = os.environ.get("KINDS")
SEL = set(int(x) for x in os.environ["CASES"].split(",")) if os.environ.get("CASES") else None
VERBOSE = bool(os.environ.get("VERBOSE"))
BIG = bool(os.environ.get("BIG"))
OMP = bool(os.environ.get("OMP"))
if ONLY: KINDS = ONLY.split(",")


def smooth(shape):
    grids = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
    return sum(np.sin(2 * np.pi * g / (13.0 + 7 * i)) for i, g in enumerate(grids))


def draw(kind, shape, dtype):
    """-> (array, absolute bound, may_rel)"""
    n = int(np.prod(shape))
    if kind == "noise":
        return rng.standard_normal(shape).astype(dtype), float(10.0 ** rng.uniform(-5, -1)), True
    if kind == "const":
        return np.full(shape, float(rng.choice([1.0, -3.25, 1e10, 1e-10])), dtype), float(10.0 ** rng.uniform(-4, -1)), False
    if kind == "constrel":  # a constant array under a range-based bound: range 0 -> bound 0 -> the lossless stream (SZDispatcher.hpp:19-21)
        return np.full(shape, float(rng.choice([1.0, -3.25, 0.0])), dtype), float(10.0 ** rng.uniform(-4, -1)), True
    if kind == "zeros":
        return np.zeros(shape, dtype), float(10.0 ** rng.uniform(-4, -1)), False
    if kind == "steps":
        a = smooth(shape) * 0.01
        cuts = np.sort(rng.integers(0, n, size=int(rng.integers(1, 12))))
        lv = np.zeros(n)
        for c in cuts: lv[c:] += float(rng.choice([-100.0, 7.5, 1000.0, -0.5]))
        return (a + lv.reshape(shape)).astype(dtype), float(10.0 ** rng.uniform(-4, -2)), True
    if kind == "spikes":
        a = smooth(shape)
        k = max(1, n // int(rng.choice([50, 500, 5000])))
        a.reshape(-1)[rng.integers(0, n, size=k)] = rng.choice([1e30, -1e30, 1e15, 65536.0], size=k)
        return a.astype(dtype), float(10.0 ** rng.uniform(-4, -2)), False
    if kind == "tight":  # the bound below the spacing of the values: every point beyond the quantiser's range or exactly on the lattice
        a = (smooth(shape) * 1000.0).astype(dtype)
        return a, float(10.0 ** rng.uniform(-9, -6)) if dtype == np.float32 else float(10.0 ** rng.uniform(-15, -12)), False
    if kind == "huge":
        return (smooth(shape) * 1e30).astype(dtype), float(10.0 ** rng.uniform(25, 28)), True
    if kind == "denormal":
        return (smooth(shape) * 1e-39).astype(dtype), float(10.0 ** rng.uniform(-43, -41)), True
    if kind == "integers":
        return np.rint(smooth(shape) * float(rng.choice([3.0, 100.0, 30000.0]))).astype(dtype), float(rng.choice([0.4, 0.5, 1.0, 2.5])), True
    if kind == "negzero":
        a = smooth(shape).astype(dtype)
        a.reshape(-1)[rng.integers(0, n, size=max(1, n // 20))] = -0.0
        a[np.abs(a) < 0.3] = -0.0
        return a, float(10.0 ** rng.uniform(-4, -2)), True
    if kind == "nonfinite":
        a = smooth(shape).astype(dtype)
        k = max(1, n // int(rng.choice([20, 300, 4000])))
        a.reshape(-1)[rng.integers(0, n, size=k)] = rng.choice([np.nan, np.inf, -np.inf], size=k)
        return a, float(10.0 ** rng.uniform(-4, -2)), False
    if kind == "ramp":  # exactly linear: every predictor is exact, regression's coefficients sit on their lattice
        grids = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
        return sum((i + 1) * 0.125 * g for i, g in enumerate(grids)).astype(dtype), float(10.0 ** rng.uniform(-4, -1)), True
    if kind == "sparse":  # zeros with a few islands
        a = np.zeros(n)
        for _ in range(int(rng.integers(1, 6))):
            c, w = int(rng.integers(0, n)), int(rng.integers(1, max(2, n // 50)))
            a[c:c + w] = rng.standard_normal(min(w, n - c)) * float(rng.choice([1.0, 1e4]))
        return a.reshape(shape).astype(dtype), float(10.0 ** rng.uniform(-4, -2)), True
    raise ValueError(kind)


def ref_read(blob, a):
    """the reference's (the oracle's) reading of a container, in a process of its own (it aborts on some of its own containers:
    tests/checks/_ref_read.py) -> (array or None, the last line of its stderr)"""
    with tempfile.TemporaryDirectory() as td:
        np.ascontiguousarray(blob).tofile(os.path.join(td, "c.sz"))
        pr = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "_ref_read.py"), os.path.join(td, "c.sz"), np.dtype(a.dtype).name, str(a.size),
                             os.path.join(td, "d.bin"), "1" if USE_REF else "0"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        if pr.returncode == 0:
            return np.fromfile(os.path.join(td, "d.bin"), dtype=a.dtype).reshape(a.shape), ""
        err = [l for l in pr.stderr.decode(errors="replace").strip().splitlines() if "OpenMP enabled" not in l]
        return None, (err[-1][:160] if err else "rc %d" % pr.returncode)


def same_bits(x, y):
    return x.shape == y.shape and x.dtype == y.dtype and x.tobytes() == y.tobytes()


bad = n_cases = n_stock = n_unreadable = n_known = n_cross = n_omp = n_omp_dead = 0
for k in range(int(os.environ.get("N", "40"))):
    nd = int(rng.choice([1, 2, 3, 3, 4]))
    dtype = np.float64 if rng.random() < 0.3 else np.float32
    if BIG:  # 4 M .. 20 M elements: the forms large arrays take (the sampled code book, the one-byte packer, the fused decoders)
        if nd == 1: shape = (int(rng.integers(1 << 22, 20_000_000)),)
        elif nd == 2: shape = tuple(int(rng.integers(2048, 4400)) for _ in range(2))
        elif nd == 3: shape = tuple(int(rng.integers(160, 270)) for _ in range(3))
        else: shape = (int(rng.integers(6, 20)),) + tuple(int(rng.integers(64, 100)) for _ in range(3))
    elif nd == 1: shape = (int(rng.integers(1, 200000)),)
    elif nd == 2: shape = tuple(int(rng.integers(1, 500)) for _ in range(2))
    elif nd == 3: shape = tuple(int(rng.integers(1, 80)) for _ in range(3))
    else: shape = (int(rng.integers(1, 10)),) + tuple(int(rng.integers(1, 30)) for _ in range(3))
    do_stock = (not BIG) or rng.random() < 0.2  # (the reference codes ~10 M elements per second)
    omp_leg = rng.random() < 0.3
    kind = str(rng.choice(KINDS))
    a, ebv, may_rel = draw(kind, shape, dtype)
    algo = str(rng.choice(["interp", "default", "lorenzo", "lorenzo", "nopred"]))
    ndim = max(1, sum(1 for d in a.shape if d > 1))  # (SZ3::Config drops extents of one)
    conf = sz3_amd.Config(*a.shape)
    conf.regression = 0
    kw = {}
    fin = np.isfinite(a)
    af = a.astype(np.float64)
    vr = float(af[fin].max() - af[fin].min()) if fin.any() else 0.0
    # the bound's mode (the range-based ones where the range is a finite positive number) and the quantiser's bin count
    mode = "abs"
    if may_rel and (vr > 0 or kind == "constrel") and rng.random() < (0.4 if vr > 0 else 1.0):
        mode = str(rng.choice(["rel", "rel", "abs_and_rel", "abs_or_rel", "psnr", "l2norm"] if vr > 0 else ["rel", "abs_and_rel", "abs_or_rel"]))
    vr_t = float(np.float32(af[fin].max()) - np.float32(af[fin].min())) if (dtype == np.float32 and fin.any()) else vr
    relv = float(10.0 ** rng.uniform(-5, -1.5))
    conf.absErrorBound = ebv; kw.update(abs_eb=ebv)
    bound = ebv
    if mode == "rel":
        conf.errorBoundMode = sz3_amd.EB_REL; conf.relErrorBound = relv; kw.update(eb_mode=EB_REL, rel_eb=relv)
        bound = relv * vr_t
    elif mode in ("abs_and_rel", "abs_or_rel"):
        both = mode == "abs_and_rel"
        conf.errorBoundMode = sz3_amd.EB_ABS_AND_REL if both else sz3_amd.EB_ABS_OR_REL; conf.relErrorBound = relv
        kw.update(eb_mode=EB_ABS_AND_REL if both else EB_ABS_OR_REL, rel_eb=relv)
        bound = min(ebv, relv * vr_t) if both else max(ebv, relv * vr_t)
    elif mode == "psnr":
        ps = float(rng.choice([30.0, 50.0, 70.0, 90.0]))
        conf.errorBoundMode = sz3_amd.EB_PSNR; conf.psnrErrorBound = ps; kw.update(eb_mode=EB_PSNR, psnrErrorBound=ps)
        bound = None  # (what the mode promises is checked by tests/checks/host_sweep.py; here: the container's bytes)
    elif mode == "l2norm":
        l2 = ebv * float(np.sqrt(a.size))
        conf.errorBoundMode = sz3_amd.EB_L2NORM; conf.l2normErrorBound = l2; kw.update(eb_mode=EB_L2NORM, l2normErrorBound=l2)
        bound = None
    if rng.random() < 0.3:
        qb = int(rng.choice([16, 64, 256, 1024, 4096, 32768]))
        conf.quantbinCnt = qb; kw.update(quantbinCnt=qb)
    if algo == "interp":
        fact = [1, 1, 2, 6, 24][ndim]
        p = dict(interp_algo=int(rng.integers(0, 2)), interpDirection=int(rng.integers(0, fact)), interpAlpha=float(rng.choice([1.0, 1.25, 1.5, 2.0])),
                 interpBeta=float(rng.choice([1.0, 2.0, 2.5, 3.0])))
        conf.cmprAlgo = sz3_amd.ALGO_INTERP
        conf.interpAlgo, conf.interpDirection, conf.interpAlpha, conf.interpBeta = p["interp_algo"], p["interpDirection"], p["interpAlpha"], p["interpBeta"]
        kw.update(algo=ALGO_INTERP, **p)
        if rng.random() < 0.4:  # the anchor grid's stride (0: no anchors, the top level starts from one point)
            st = int(rng.choice([0, 4, 8, 16, 64, 512]))
            conf.interpAnchorStride = st; kw.update(interpAnchorStride=st)
    elif algo == "default":
        kw.update(algo=ALGO_INTERP_LORENZO)
    elif algo == "nopred":
        conf.cmprAlgo = sz3_amd.ALGO_NOPRED
        kw.update(algo=ALGO_NOPRED)
    else:
        sets = [(1, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 1), (1, 1, 1)] if ndim != 4 else [(1, 0, 0), (1, 0, 1)]
        if ndim == 1: sets += [(0, 0, 1)]
        l1, l2, rg = sets[int(rng.integers(0, len(sets)))]
        conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
        conf.lorenzo, conf.lorenzo2, conf.regression = l1, l2, rg
        kw.update(algo=ALGO_LORENZO_REG, lorenzo=bool(l1), lorenzo2=bool(l2), regression=bool(rg))
        if rng.random() < 0.4:  # the block's edge (defaults: 128, 16, 6, 6), within what the block kernels are built for (4 .. 65535 / 32 / 8 / 6:
            # beyond it a set with Lorenzo-1 is coded by that member alone and one without it is refused, SZ3HIP_EUNSUPPORTED)
            bs = int(rng.choice({1: [16, 64, 100, 256], 2: [4, 8, 12, 32], 3: [4, 5, 7, 8], 4: [4, 5, 6]}[ndim]))
            conf.blockSize = bs; kw.update(block_size=bs)
    slabs = int(rng.integers(2, 6))
    # (OMP=1: this library's own container leaves as slabs — SZ_compress_OMP's layout, several on the one GPU. Slabs of at least two rows: one of a
    # single row loses a dimension (Config::setDims drops extents of one), and an interpDirection of the full rank is then out of range — an error
    # here, an index past the reference's table of orders)
    omp = OMP and 2 * slabs <= a.shape[0]
    tag = "case %d %s %s %s %s %s%s" % (k, kind, algo, a.shape, dtype.__name__, kw, " slabs %d" % slabs if omp else "")
    if SEL and k not in SEL: continue  # (CASES=3,17: only those — every draw is made above, the sequence is the seed's)
    if VERBOSE: print(tag, flush=True)
    n_cases += 1
    # (1) this library's payload
    try:
        if omp:
            conf.openmp = 1; os.environ["SZ3HIP_SLABS"] = str(slabs)
        try:
            blob, ratio = sz3_amd.compress(a, conf)
        finally:
            conf.openmp = 0; os.environ.pop("SZ3HIP_SLABS", None)
        dec, _ = sz3_amd.decompress(blob, a.dtype, a.shape)
        df = dec.astype(np.float64)
        err = float(np.abs(df[fin] - af[fin]).max()) if fin.any() else 0.0
        ok = dec.shape == a.shape and np.array_equal(np.isnan(df), np.isnan(af)) and np.array_equal(df[~fin & ~np.isnan(af)], af[~fin & ~np.isnan(af)])
        # (f32: the decoder's value is the f32 nearest to the f64 reconstruction — the reference's guarantee too)
        if bound is not None:
            slack = bound * 1e-6 + (float(np.spacing(np.float32(np.abs(af[fin]).max()))) if dtype == np.float32 and fin.any() else 0.0) * 0.5
            ok = ok and err <= bound + slack
        if not ok:
            bad += 1
            print("NATIVE FAIL %s: err %.6g bound %s" % (tag, err, bound), flush=True)
    except Exception as e:
        bad += 1
        print("NATIVE EXC %s: %s" % (tag, str(e)[:120]), flush=True)
    if not do_stock: continue
    # (2) stock format against the reference
    # (the caller's capacity decides when the reference gives a lossy stream up, lossless/Lossless_zstd.hpp:29-37 + SZDispatcher.hpp:44-59: the same
    # capacity on both sides — what tests/oracle_binding.py hands the reference, the CLI's 2 x the array and more)
    oc = make_config(a.shape, **kw)
    if USE_REF:
        shp = [int(oc.dims[i]) for i in range(oc.N)]
        cap = int(ref().ref_compress_bound(_dtype_id(a), len(shp), (C.c_size_t * len(shp))(*shp)))
    else:
        cap = int(oracle().szo_compress_bound(C.byref(oc), _dtype_id(a))) + 2 * a.nbytes
    L.sz3hip_set_stock_format(1)
    try:
        sblob, _ = sz3_amd.compress(a, conf, out=np.empty(max(cap, sz3_amd.compress_bound(conf, a.dtype)), dtype=np.uint8))
        sblob = sblob.copy()
    except sz3_amd.SZ3HipError as e:
        print("%s: stock writer: %s (skipped)" % (tag, str(e)[:100]), flush=True)
        sblob = None
    finally:
        L.sz3hip_set_stock_format(0)
    try:
        ob = ref_compress(a, oc) if USE_REF else oracle_compress(a, oc)
    except Exception as e:
        print("%s: reference compress: %s (skipped)" % (tag, str(e)[:100]), flush=True)
        continue
    # Non-finite values under a set with the regression member, reported but not counted: the reference's writer and reader lose step there —
    # a block with an extent of 1 whose Lorenzo estimate is +Inf "selects" the invalid regression member (DBL_MAX < Inf), is coded by the
    # fallback predictor and leaves NO selection entry (ComposedPredictor.hpp:25-39, BlockwiseDecomposition.hpp:35-37), while the reader takes
    # one per block (ComposedPredictor.hpp:47-50): its own file decodes to other values or past the end of the list (the crashes below). This
    # library writes the entry — a file the reference reads correctly — and refuses the reference's as corrupt when the count is short.
    # (A NaN coefficient's sign — stored as it is — follows x86's rules since round 6: sz3hip_stock.hip, x86_op.)
    known = (not fin.all()) and bool(kw.get("regression", False))
    if sblob is not None and ("block_size" in kw or "interpAnchorStride" in kw):
        own = int(sz3_amd.decompress(sblob, a.dtype, a.shape)[1].cmprAlgo)
        if own >= 16:  # (block edges beyond the stock writer's: 4-D > 6, 3-D > 8, 2-D > 32)
            print("%s: no stock form of this call here (the container keeps this library's id %d) (skipped)" % (tag, own), flush=True)
            sblob = None
    if sblob is not None:
        n_stock += 1
        if sblob.tobytes() != ob.tobytes():
            if known: n_known += 1
            else: bad += 1
            print("STOCK BYTES MISMATCH%s %s: %d vs %d bytes" % (" (non-finite values + regression: not counted)" if known else "", tag, sblob.size, ob.size), flush=True)
            # other bytes, then at least a container the reference reads to what this library reads from it
            xr, why = ref_read(sblob, a)
            if xr is None:
                print("   (the reference cannot read this library's container either: %s)" % why, flush=True)
            else:
                mine, _ = sz3_amd.decompress(sblob, a.dtype, a.shape)
                if not same_bits(np.ascontiguousarray(mine), np.ascontiguousarray(xr)):
                    bad += 1
                    print("   CROSS READ MISMATCH: the reference reads this library's container to other values", flush=True)
                else:
                    n_cross += 1
            if VERBOSE:
                m = min(sblob.size, ob.size)
                d = np.flatnonzero(sblob[:m] != ob[:m])
                print("   first difference at byte %s; heads %s | %s" % (d[0] if d.size else m, sblob[:48].tobytes().hex(), ob[:48].tobytes().hex()))
                try:
                    print("   algorithm in the container: ours %d, the reference's %d" % (sz3_amd.decompress(sblob, a.dtype, a.shape)[1].cmprAlgo,
                                                                                       sz3_amd.decompress(ob, a.dtype, a.shape)[1].cmprAlgo))
                except Exception as e:
                    print("   (reading them back: %s)" % str(e)[:100])
                try:  # the streams in front of zstd
                    import ctypes as C
                    Z = C.CDLL("libzstd.so.1"); Z.ZSTD_decompress.restype = C.c_size_t
                    raws = []
                    for bl in (sblob, ob):
                        ln = int(np.frombuffer(bl[16:24].tobytes(), dtype=np.uint64)[0]); body = bl[24:].tobytes()
                        buf = C.create_string_buffer(ln); got = Z.ZSTD_decompress(buf, C.c_size_t(ln), body, C.c_size_t(len(body)))
                        raws.append(np.frombuffer(buf.raw[:ln], dtype=np.uint8))
                    m = min(raws[0].size, raws[1].size)
                    d = np.flatnonzero(raws[0][:m] != raws[1][:m])
                    print("   streams in front of zstd: %d vs %d bytes, %d differing bytes, first at %s, last at %s" % (raws[0].size, raws[1].size, d.size, d[0] if d.size else None, d[-1] if d.size else None))
                    if d.size:
                        o = max(0, int(d[0]) - 16)
                        print("   ours      @%d: %s\n   reference @%d: %s" % (o, raws[0][o:o + 64].tobytes().hex(), o, raws[1][o:o + 64].tobytes().hex()))
                    if os.environ.get("DUMP"):
                        raws[0].tofile(os.environ["DUMP"] + ".ours"); raws[1].tofile(os.environ["DUMP"] + ".ref"); a.tofile(os.environ["DUMP"] + ".in")
                except Exception as e:
                    print("   (streams in front of zstd: %s)" % str(e)[:100])
    rd, why = ref_read(ob, a)
    if rd is None:
        n_unreadable += 1
        try: sz3_amd.decompress(ob, a.dtype, a.shape)  # (this library's reading of it: an array or an error, nothing to compare with)
        except Exception: pass
        print("%s: the reference cannot read its own container (%s) (skipped)" % (tag, why), flush=True)
        continue
    try:
        md, _ = sz3_amd.decompress(ob, a.dtype, a.shape)
        if not same_bits(np.ascontiguousarray(md), np.ascontiguousarray(rd)):
            if not known: bad += 1
            neq = int(np.sum(md.reshape(-1).view("u%d" % md.itemsize) != rd.reshape(-1).view("u%d" % rd.itemsize)))
            print("STOCK READ MISMATCH %s: %d of %d values differ" % (tag, neq, a.size), flush=True)
    except Exception as e:
        if not known: bad += 1
        print("STOCK READ EXC%s %s: %s" % (" (non-finite values + regression: not counted)" if known else "", tag, str(e)[:120]), flush=True)
    # (3) a container the reference wrote with its OpenMP path (SZ_compress_OMP: a slab per thread along the slowest extent), read here
    if USE_REF and omp_leg and a.shape[0] >= 4 and not known:
        kw2 = dict(kw); kw2["openmp"] = True
        with tempfile.TemporaryDirectory() as td:  # (in a process of its own: the reference's OpenMP path aborts or divides by zero on some shapes)
            a.tofile(os.path.join(td, "a.bin"))
            pr = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "_ref_write.py"), os.path.join(td, "a.bin"), np.dtype(a.dtype).name,
                                 ",".join(str(d) for d in a.shape), json.dumps(kw2), os.path.join(td, "c.sz")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            ob2 = np.fromfile(os.path.join(td, "c.sz"), dtype=np.uint8) if pr.returncode == 0 else None
        if ob2 is None:
            n_omp_dead += 1
            continue
        rd2, why2 = ref_read(ob2, a)
        if rd2 is None:
            print("%s: the reference cannot read its own OpenMP container (%s) (skipped)" % (tag, why2), flush=True)
            continue
        n_omp += 1
        try:
            md2, _ = sz3_amd.decompress(ob2, a.dtype, a.shape)
            if not same_bits(np.ascontiguousarray(md2), np.ascontiguousarray(rd2)):
                bad += 1
                print("OMP CONTAINER READ MISMATCH %s" % tag, flush=True)
        except Exception as e:
            bad += 1
            print("OMP CONTAINER READ EXC %s: %s" % (tag, str(e)[:120]), flush=True)
print("the reference's OpenMP containers read: %d (its OpenMP writer died on %d more)" % (n_omp, n_omp_dead))
print("cases %d (stock containers %d, reference = %s; %d the reference could not read back; %d other bytes with non-finite values under regression, "
      "%d of them read by the reference to this library's values), failures %d" % (n_cases, n_stock, "the reference build" if USE_REF else "the oracle", n_unreadable, n_known, n_cross, bad))
sys.exit(1 if bad else 0)
