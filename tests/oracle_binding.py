"""ctypes bindings for the TEST-ONLY checkers:

  * oracle/libsz3oracle.so      — our plain-C restatement of the reference algorithm (oracle/sz3_oracle.c)
  * oracle/_ref/libsz3ref.so    — the reference itself (szcompressor/SZ3 v3.3.2) built by `make -C oracle ref`,
                                  present only where that build was run (this container; travels to the GPU box
                                  as a binary, never as source)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libsz3oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libsz3ref.so")

EB_ABS, EB_REL, EB_PSNR, EB_L2NORM, EB_ABS_AND_REL, EB_ABS_OR_REL = range(6)
ALGO_LORENZO_REG, ALGO_INTERP_LORENZO, ALGO_INTERP, ALGO_NOPRED, ALGO_LOSSLESS = range(5)
INTERP_LINEAR, INTERP_CUBIC = 0, 1


class SzoConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("dims", C.c_uint64 * 4), ("num", C.c_uint64),
        ("cmprAlgo", C.c_uint8), ("errorBoundMode", C.c_uint8),
        ("absErrorBound", C.c_double), ("relErrorBound", C.c_double),
        ("psnrErrorBound", C.c_double), ("l2normErrorBound", C.c_double),
        ("openmp", C.c_uint8), ("quantbinCnt", C.c_int32), ("blockSize", C.c_int32),
        ("predDim", C.c_uint8), ("dataType", C.c_uint8),
        ("lorenzo", C.c_uint8), ("lorenzo2", C.c_uint8), ("regression", C.c_uint8), ("regression2", C.c_uint8),
        ("interpAlgo", C.c_uint8), ("interpDirection", C.c_uint8),
        ("interpAnchorStride", C.c_int32), ("interpAlpha", C.c_double), ("interpBeta", C.c_double),
    ]


class SzoStats(C.Structure):
    _fields_ = [
        ("n_unpred", C.c_uint64), ("raw_bytes", C.c_uint64), ("huff_bytes", C.c_uint64),
        ("n_regression_blocks", C.c_uint64), ("n_blocks", C.c_uint64), ("huff_node_count", C.c_uint32),
        ("t_decomp", C.c_double), ("t_hist_tree", C.c_double), ("t_encode", C.c_double), ("t_zstd", C.c_double),
    ]


class SzoTunerReport(C.Structure):
    _fields_ = [("sample_block_size", C.c_uint64), ("n_filtered", C.c_uint64), ("n_blocks", C.c_uint64),
                ("profiling", C.c_int32), ("reserved", C.c_int32), ("ratios", C.c_double * 8),
                ("best_interp", C.c_double), ("best_lorenzo", C.c_double), ("raw_bytes", C.c_uint64 * 8), ("huff_bytes", C.c_uint64 * 8),
                ("node_count", C.c_uint64 * 8), ("n_unpred", C.c_uint64 * 8), ("entropy_bits", C.c_double * 8)]


def _dtype_id(a):
    if a.dtype == np.float32:
        return 0
    if a.dtype == np.float64:
        return 1
    raise TypeError(a.dtype)


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError("oracle/libsz3oracle.so missing: run `make -C oracle port` (or __graft_entry__.build())")
        L = C.CDLL(ORACLE_SO)
        L.szo_config_init.argtypes = [C.POINTER(SzoConfig), C.c_int, C.POINTER(C.c_uint64)]
        L.szo_config_save.restype = C.c_size_t
        L.szo_config_save.argtypes = [C.POINTER(SzoConfig), C.c_void_p]
        L.szo_config_load.restype = C.c_size_t
        L.szo_config_load.argtypes = [C.POINTER(SzoConfig), C.c_void_p]
        L.szo_compress_bound.restype = C.c_size_t
        L.szo_compress_bound.argtypes = [C.POINTER(SzoConfig), C.c_int]
        L.szo_compress.restype = C.c_size_t
        L.szo_compress.argtypes = [C.POINTER(SzoConfig), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(SzoStats)]
        L.szo_decompress.restype = C.c_size_t
        L.szo_decompress.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(SzoConfig)]
        L.szo_last_error.restype = C.c_char_p
        L.szo_quantize_f32.restype = C.c_int32
        L.szo_quantize_f32.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_double, C.c_int32]
        L.szo_quantize_f64.restype = C.c_int32
        L.szo_quantize_f64.argtypes = [C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int32]
        L.szo_recover_f32.restype = C.c_float
        L.szo_recover_f32.argtypes = [C.c_float, C.c_int32, C.c_double, C.c_int32]
        L.szo_recover_f64.restype = C.c_double
        L.szo_recover_f64.argtypes = [C.c_double, C.c_int32, C.c_double, C.c_int32]
        L.szo_huffman_encode.restype = C.c_size_t
        L.szo_huffman_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.szo_huffman_decode.restype = C.c_size_t
        L.szo_huffman_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.szo_zstd_compress.restype = C.c_size_t
        L.szo_zstd_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.szo_zstd_decompress.restype = C.c_size_t
        L.szo_zstd_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.szo_zstd_bound.restype = C.c_size_t
        L.szo_zstd_bound.argtypes = [C.c_size_t]
        L.szo_zstd_version.restype = C.c_char_p
        L.szo_decomposition_codes.restype = C.c_size_t
        L.szo_decomposition_codes.argtypes = [C.POINTER(SzoConfig), C.c_int, C.c_void_p, C.c_void_p]
        L.szo_set_omp_slabs.argtypes = [C.c_int]
        L.szo_interp_codes.restype = C.c_size_t
        L.szo_interp_codes.argtypes = [C.POINTER(SzoConfig), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.szo_tune_interp_lorenzo.restype = C.c_int
        L.szo_tune_interp_lorenzo.argtypes = [C.POINTER(SzoConfig), C.c_int, C.c_void_p, C.POINTER(SzoTunerReport)]
        _oracle = L
    return _oracle


def make_config(shape, algo=ALGO_LORENZO_REG, eb_mode=EB_ABS, abs_eb=1e-3, rel_eb=0.0, lorenzo=True, lorenzo2=False,
                regression=False, openmp=False, interp_algo=None, block_size=None, **extra):
    """SZ3::Config(dims...) + field assignments, dims slowest first (= numpy shape)."""
    L = oracle()
    c = SzoConfig()
    d = (C.c_uint64 * len(shape))(*shape)
    L.szo_config_init(C.byref(c), len(shape), d)
    c.cmprAlgo = algo
    c.errorBoundMode = eb_mode
    c.absErrorBound = abs_eb
    c.relErrorBound = rel_eb
    c.lorenzo, c.lorenzo2, c.regression = int(lorenzo), int(lorenzo2), int(regression)
    c.openmp = int(openmp)
    if interp_algo is not None:
        c.interpAlgo = interp_algo
    if block_size:
        c.blockSize = block_size
    for k, v in extra.items():
        setattr(c, k, v)
    return c


def oracle_selection(a, conf):
    """the predictor the oracle's composed predictor chooses for every block of `a` (0 Lorenzo-1, 1 Lorenzo-2, 2 regression),
    in block raster order"""
    L = oracle()
    bs = int(conf.blockSize)
    nb = int(np.prod([(d + bs - 1) // bs for d in a.shape]))
    sel = np.full(nb, -1, dtype=np.int8)
    L.szo_debug_selection_sink.argtypes = [C.c_void_p, C.c_size_t]
    L.szo_debug_selection_sink.restype = None
    L.szo_debug_selection_sink(sel.ctypes.data, nb)
    try:
        oracle_compress(a, conf)
    finally:
        L.szo_debug_selection_sink(None, 0)
    return sel


def oracle_compress(a, conf, stats=False):
    L = oracle()
    a = np.ascontiguousarray(a)
    cap = L.szo_compress_bound(C.byref(conf), _dtype_id(a)) + 2 * a.nbytes
    out = np.empty(cap, dtype=np.uint8)
    st = SzoStats()
    n = L.szo_compress(C.byref(conf), _dtype_id(a), a.ctypes.data, out.ctypes.data, cap, C.byref(st))
    if n == 0:
        raise RuntimeError("oracle compress: " + L.szo_last_error().decode())
    blob = out[:n].copy()
    return (blob, st) if stats else blob


def oracle_decompress(blob, dtype, shape):
    L = oracle()
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    dec = np.empty(int(np.prod(shape)), dtype=dtype)
    conf = SzoConfig()
    n = L.szo_decompress(_dtype_id(dec), blob.ctypes.data, blob.size, dec.ctypes.data, C.byref(conf))
    if n == 0:
        raise RuntimeError("oracle decompress: " + L.szo_last_error().decode())
    return dec.reshape(shape), conf


def oracle_codes(a, conf):
    L = oracle()
    a = np.ascontiguousarray(a)
    codes = np.empty(a.size, dtype=np.int32)
    n = L.szo_decomposition_codes(C.byref(conf), _dtype_id(a), a.ctypes.data, codes.ctypes.data)
    if n == 2 ** 64 - 1:
        raise RuntimeError("oracle codes: unsupported config")
    return codes, n


def oracle_interp_codes(a, conf):
    """(codes in emission order, element index of every code, reconstructed array, #unpredictable)"""
    L = oracle()
    a = np.ascontiguousarray(a)
    codes = np.empty(a.size, dtype=np.int32)
    order = np.empty(a.size, dtype=np.uint64)
    recon = np.empty_like(a)
    n = L.szo_interp_codes(C.byref(conf), _dtype_id(a), a.ctypes.data, codes.ctypes.data, order.ctypes.data, recon.ctypes.data)
    return codes, order, recon, n


def oracle_tune(a, conf):
    """SZ_compress_Interp_lorenzo's decisions (api/impl/SZAlgoInterp.hpp:122-262) on array a: returns (tuned config copy,
    report, ran) — tuned.cmprAlgo is ALGO_INTERP (interpAlgo / interpDirection / interpAlpha / interpBeta chosen) or
    ALGO_LORENZO_REG (1-D only)."""
    L = oracle()
    a = np.ascontiguousarray(a)
    c = SzoConfig.from_buffer_copy(conf)
    c.cmprAlgo = ALGO_INTERP_LORENZO
    rep = SzoTunerReport()
    ran = L.szo_tune_interp_lorenzo(C.byref(c), _dtype_id(a), a.ctypes.data, C.byref(rep))
    if ran < 0:
        raise RuntimeError(L.szo_last_error().decode())
    return c, rep, bool(ran)


# ---------------------------------------------------------------------------------------------------------
_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_compress.restype = C.c_size_t
        L.ref_compress.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_double,
                                   C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_size_t, C.POINTER(C.c_double)]
        L.ref_compress_ex.restype = C.c_size_t
        L.ref_compress_ex.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_double,
                                      C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_double, C.c_double, C.c_void_p, C.c_size_t, C.POINTER(C.c_double)]
        L.ref_compress_bound.restype = C.c_size_t
        L.ref_compress_bound.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_size_t)]
        if hasattr(L, "ref_set_extra"):  # (quantbinCnt, psnrErrorBound, l2normErrorBound: fields the entry points' lists do not carry)
            L.ref_set_extra.restype = None
            L.ref_set_extra.argtypes = [C.c_int, C.c_double, C.c_double]
        L.ref_decompress.restype = C.c_size_t
        L.ref_decompress.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_double)]
        _ref = L
    return _ref


def ref_compress(a, conf, timing=False):
    """Drive the real reference SZ_compress<T> with the fields of an SzoConfig."""
    L = ref()
    a = np.ascontiguousarray(a)
    shape = [int(conf.dims[i]) for i in range(conf.N)]
    d = (C.c_size_t * len(shape))(*shape)
    cap = L.ref_compress_bound(_dtype_id(a), len(shape), d)
    out = np.empty(cap, dtype=np.uint8)
    sec = C.c_double(0)
    extra = hasattr(L, "ref_set_extra")
    if extra:
        L.ref_set_extra(int(conf.quantbinCnt), float(conf.psnrErrorBound), float(conf.l2normErrorBound))
    elif conf.quantbinCnt != 65536 or conf.errorBoundMode in (EB_PSNR, EB_L2NORM):
        raise RuntimeError("this build of oracle/_ref/libsz3ref.so has no ref_set_extra (make -C oracle ref)")
    try:
        n = L.ref_compress_ex(_dtype_id(a), a.ctypes.data, len(shape), d, conf.cmprAlgo, conf.errorBoundMode,
                              conf.absErrorBound, conf.relErrorBound, conf.lorenzo, conf.lorenzo2, conf.regression,
                              conf.openmp, conf.interpAlgo, conf.blockSize, conf.interpDirection, conf.interpAnchorStride,
                              conf.interpAlpha, conf.interpBeta, out.ctypes.data, cap, C.byref(sec))
    finally:
        if extra: L.ref_set_extra(0, 0.0, 0.0)
    if n == 0:
        raise RuntimeError("reference compress failed")
    blob = out[:n].copy()
    return (blob, sec.value) if timing else blob


def ref_decompress(blob, dtype, shape, timing=False):
    L = ref()
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    dec = np.empty(int(np.prod(shape)), dtype=dtype)
    sec = C.c_double(0)
    n = L.ref_decompress(_dtype_id(dec), blob.ctypes.data, blob.size, dec.ctypes.data, C.byref(sec))
    if n == 0:
        raise RuntimeError("reference decompress failed")
    dec = dec.reshape(shape)
    return (dec, sec.value) if timing else dec
