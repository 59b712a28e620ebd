"""sz3_amd — host-side mirror of the SZ3 API for the MI355X (gfx950) hot path.

The product is the C-ABI shared library ``sz3_amd/libsz3hip.so`` (include/sz3hip.h, include/sz3c.h): predictor ->
quantizer -> Huffman as hand-written HIP kernels.  This module is a thin ctypes binding that mirrors the reference's
Python face ``pysz`` (tools/pysz/src/pysz/sz.pyx:185-405: ``sz.compress(ndarray, szConfig) -> (uint8 ndarray, ratio)``,
``sz.decompress(bytes, dtype, shape) -> (ndarray, szConfig)``, ``sz.verify -> (max_diff, psnr, nrmse)``) plus the
device-resident entry points used by bench.py and the multi-GPU driver.

There is no CPU implementation here: importing works anywhere, but every call needs the built library and a HIP
device and raises otherwise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SZ3HIP_LIB") or os.path.join(_HERE, "libsz3hip.so")  # (SZ3HIP_LIB: A/B runs of two builds)

# include/SZ3/utils/Config.hpp:66,80,93 (enum EB / ALGO / INTERP_ALGO) + the GPU stream id (include/sz3hip.h)
EB_ABS, EB_REL, EB_PSNR, EB_L2NORM, EB_ABS_AND_REL, EB_ABS_OR_REL = range(6)
ALGO_LORENZO_REG, ALGO_INTERP_LORENZO, ALGO_INTERP, ALGO_NOPRED, ALGO_LOSSLESS = range(5)
ALGO_HIP_LORENZO = 16
ALGO_HIP_INTERP = 17
INTERP_ALGO_LINEAR, INTERP_ALGO_CUBIC = 0, 1
SZ_FLOAT, SZ_DOUBLE = 0, 1


class SZ3HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sz3hip error %d: %s" % (code, msg))
        self.code = code


class _CConfig(C.Structure):
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


class _CTunerReport(C.Structure):
    _fields_ = [("ran", C.c_int32), ("use_interp", C.c_int32), ("sample_block_size", C.c_uint64), ("n_filtered", C.c_uint64),
                ("n_blocks", C.c_uint64), ("profiling", C.c_int32), ("interpAlgo", C.c_int32), ("interpDirection", C.c_int32),
                ("speculated", C.c_int32), ("interpAlpha", C.c_double), ("interpBeta", C.c_double), ("est_bytes", C.c_double * 8)]


class _CStats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("n_value_outliers", C.c_uint64), ("n_delta_outliers", C.c_uint64),
                ("n_chunks", C.c_uint64), ("bitstream_bytes", C.c_uint64), ("payload_bytes", C.c_uint64),
                ("n_symbols", C.c_uint32), ("max_code_len", C.c_uint32),
                ("narrow_codes", C.c_uint32), ("reserved", C.c_uint32)]


_lib = None


def _share_torch_hip_runtime():
    """A process must hold ONE HIP runtime. PyTorch-ROCm wheels bundle their own libamdhip64 (found by rpath): when this library
    is loaded first it brings the system's copy in, and a later `import torch` + first CUDA call then finds "No HIP GPUs" — two
    runtimes cannot share the device. If a torch with a bundled runtime is installed, that copy is loaded first (by path, without
    importing torch), so that whoever comes second binds to it; torch-first already worked that way (same soname)."""
    try:
        import importlib.util
        import sys
        if "torch" in sys.modules:
            return
        # An `import torch` that comes AFTER this library has been at work on the device (contexts, streams, kernels launched) stalled
        # for good in about one fresh process out of five on the MI355X boxes — inside torch/__init__'s load of its C extension, i.e.
        # while torch's bundled libraries register their code objects with a HIP runtime that is already busy (caught with pytest's
        # faulthandler_timeout, round 4). The cure is the ORDER — torch first — and it is the caller's to keep: programs that use
        # torch beside this package import it before the first call (tests/conftest.py, bench.py, sz3_amd.distributed do), or set
        # SZ3HIP_TORCH_PRELOAD=1 and have it imported here. By default only the runtime library is preloaded: a host that never
        # touches torch (the CLI, the HDF5 filter under h5py, CPU-side tools) does not pay torch's import or its runtime initialisation.
        if os.environ.get("SZ3HIP_TORCH_PRELOAD", "0") == "1":
            try:
                import torch  # noqa: F401
                return
            except Exception:  # noqa: BLE001 - no usable torch: fall through to the runtime library alone
                pass
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        _warn_on_late_torch_import(sys)
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libamdhip64.so",):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001 - best effort: without it the library still works, only torch-after-us does not
        pass


class _LateTorchImport:
    """sys.meta_path entry that finds nothing: it only notices the first `import torch` that comes after this library has been loaded
    (the hazard described in _share_torch_hip_runtime) and says so once — the caller can then fix the order instead of meeting an
    intermittent hang."""
    warned = False

    def find_spec(self, name, path=None, target=None):
        if name == "torch" and not _LateTorchImport.warned and _lib is not None:
            _LateTorchImport.warned = True
            import warnings
            warnings.warn("`import torch` after sz3_amd has loaded libsz3hip.so: on MI355X boxes this order stalled inside torch's import in "
                          "about one fresh process out of five once the device had been used. Import torch before the first sz3_amd call, "
                          "or set SZ3HIP_TORCH_PRELOAD=1.", RuntimeWarning, stacklevel=2)
        return None


def _warn_on_late_torch_import(sys):
    if not any(isinstance(f, _LateTorchImport) for f in sys.meta_path):
        sys.meta_path.insert(0, _LateTorchImport())


def lib():
    """Loads libsz3hip.so; fails loudly when it has not been built (python -m sz3_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("sz3_amd/libsz3hip.so is missing — build it with `python -m sz3_amd.build` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.sz3hip_last_error.restype = C.c_char_p
    L.sz3hip_version.restype = C.c_char_p
    L.sz3hip_config_init.argtypes = [P(_CConfig), C.c_int, P(C.c_uint64)]
    L.sz3hip_config_save.restype = C.c_size_t
    L.sz3hip_config_save.argtypes = [P(_CConfig), C.c_void_p]
    L.sz3hip_config_load.restype = C.c_size_t
    L.sz3hip_config_load.argtypes = [P(_CConfig), C.c_void_p]
    L.sz3hip_compress_bound.restype = C.c_size_t
    L.sz3hip_compress_bound.argtypes = [P(_CConfig), C.c_int]
    L.sz3hip_compress.restype = C.c_size_t
    L.sz3hip_compress.argtypes = [P(_CConfig), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.sz3hip_decompress.restype = C.c_int
    L.sz3hip_decompress.argtypes = [P(_CConfig), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.sz3hip_peek_config.restype = C.c_int
    L.sz3hip_peek_config.argtypes = [P(_CConfig), C.c_void_p, C.c_size_t]
    L.sz3hip_ctx_create.restype = C.c_void_p
    L.sz3hip_ctx_create.argtypes = [C.c_int, C.c_uint64, C.c_int]
    L.sz3hip_ctx_destroy.argtypes = [C.c_void_p]
    L.sz3hip_payload_bound.restype = C.c_size_t
    L.sz3hip_payload_bound.argtypes = [C.c_void_p, C.c_uint64]
    L.sz3hip_payload_bound_max.restype = C.c_size_t
    L.sz3hip_payload_bound_max.argtypes = [C.c_void_p, C.c_uint64]
    L.sz3hip_minmax_device.restype = C.c_int
    L.sz3hip_minmax_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_double), P(C.c_double), C.c_void_p]
    L.sz3hip_compress_stage1.restype = C.c_int
    L.sz3hip_compress_stage1.argtypes = [C.c_void_p, P(_CConfig), C.c_void_p, C.c_void_p]
    L.sz3hip_histogram_ptr.restype = C.c_void_p
    L.sz3hip_histogram_ptr.argtypes = [C.c_void_p]
    L.sz3hip_histogram_len.restype = C.c_size_t
    L.sz3hip_histogram_len.argtypes = [C.c_void_p]
    L.sz3hip_ctx_set_histogram.restype = C.c_int
    L.sz3hip_ctx_set_histogram.argtypes = [C.c_void_p, C.c_void_p]
    L.sz3hip_compress_stage2.restype = C.c_int
    L.sz3hip_compress_stage2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.sz3hip_compress_finish.restype = C.c_int
    L.sz3hip_compress_finish.argtypes = [C.c_void_p, P(C.c_size_t), C.c_void_p]
    L.sz3hip_compress_device.restype = C.c_int
    L.sz3hip_compress_device.argtypes = [C.c_void_p, P(_CConfig), C.c_void_p, C.c_void_p, C.c_size_t, P(C.c_size_t), C.c_void_p]
    L.sz3hip_decompress_device.restype = C.c_int
    L.sz3hip_decompress_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.sz3hip_get_stats.restype = C.c_int
    L.sz3hip_get_stats.argtypes = [C.c_void_p, P(_CStats)]
    L.sz3hip_get_tuner_report.restype = C.c_int
    L.sz3hip_get_tuner_report.argtypes = [C.c_void_p, P(_CTunerReport)]
    L.sz3hip_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.sz3hip_ctx_forget.argtypes = [C.c_void_p]
    L.sz3hip_ctx_forget.restype = None
    L.sz3hip_ctx_set_speculation.argtypes = [C.c_void_p, C.c_int]
    L.sz3hip_ctx_set_speculation.restype = None
    L.sz3hip_ctx_set_deterministic.argtypes = [C.c_void_p, C.c_int]
    L.sz3hip_ctx_set_deterministic.restype = None
    L.sz3hip_ctx_set_tuner_exact.argtypes = [C.c_void_p, C.c_int]
    L.sz3hip_ctx_set_tuner_exact.restype = None
    L.sz3hip_set_stock_format.argtypes = [C.c_int]
    L.sz3hip_set_stock_format.restype = None
    L.sz3hip_last_call_fused.argtypes = [C.c_void_p]
    L.sz3hip_last_call_fused.restype = C.c_int
    L.sz3hip_last_call_q16.argtypes = [C.c_void_p]
    L.sz3hip_last_call_q16.restype = C.c_int
    L.sz3hip_ctx_set_fused.argtypes = [C.c_void_p, C.c_int]
    L.sz3hip_ctx_set_fused.restype = None
    L.sz3hip_get_spec_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.sz3hip_get_spec_stats.restype = None
    L.sz3hip_payload_bound_conf.restype = C.c_size_t
    L.sz3hip_payload_bound_conf.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.sz3hip_get_stage_times.restype = C.c_int
    L.sz3hip_get_stage_times.argtypes = [C.c_void_p, P(C.c_char_p), P(C.c_float), C.c_int]
    L.sz3hip_debug_copy_codes.restype = C.c_int
    L.sz3hip_debug_copy_codes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.sz3hip_debug_force_generic.argtypes = [C.c_int]
    L.sz3hip_config_load_n.restype = C.c_size_t
    L.sz3hip_config_load_n.argtypes = [P(_CConfig), C.c_void_p, C.c_size_t]
    L.sz3hip_comm_create_local.restype = C.c_void_p
    L.sz3hip_comm_create_local.argtypes = [C.c_int, P(C.c_int)]
    L.sz3hip_comm_unique_id.restype = C.c_int
    L.sz3hip_comm_unique_id.argtypes = [C.c_void_p]
    L.sz3hip_comm_create_rank.restype = C.c_void_p
    L.sz3hip_comm_create_rank.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.sz3hip_comm_destroy.argtypes = [C.c_void_p]
    for f in ("size", "rank", "local_size"):
        getattr(L, "sz3hip_comm_" + f).restype = C.c_int
        getattr(L, "sz3hip_comm_" + f).argtypes = [C.c_void_p]
    L.sz3hip_comm_device.restype = C.c_int
    L.sz3hip_comm_device.argtypes = [C.c_void_p, C.c_int]
    L.sz3hip_comm_allreduce_histogram.restype = C.c_int
    L.sz3hip_comm_allreduce_histogram.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_void_p)]
    L.sz3hip_comm_allreduce_u64.restype = C.c_int
    L.sz3hip_comm_allreduce_u64.argtypes = [C.c_void_p, P(C.c_void_p), C.c_size_t, P(C.c_void_p)]
    L.sz3hip_comm_allreduce_minmax.restype = C.c_int
    L.sz3hip_comm_allreduce_minmax.argtypes = [C.c_void_p, P(C.c_double), P(C.c_double), P(C.c_void_p)]
    L.sz3hip_compress_rank.restype = C.c_size_t
    L.sz3hip_compress_rank.argtypes = [C.c_void_p, P(_CConfig), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, P(_CConfig)]
    L.sz3hip_assemble_container.restype = C.c_size_t
    L.sz3hip_assemble_container.argtypes = [P(_CConfig), C.c_int, C.c_int, P(_CConfig), P(C.c_void_p), P(C.c_size_t), C.c_void_p, C.c_size_t]
    L.SZ_compress_args.restype = C.c_void_p
    L.SZ_compress_args.argtypes = [C.c_int, C.c_void_p, P(C.c_size_t), C.c_int, C.c_double, C.c_double, C.c_double,
                                   C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
    L.SZ_decompress.restype = C.c_void_p
    L.SZ_decompress.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
    L.free_buf.argtypes = [C.c_void_p]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise SZ3HipError(rc, lib().sz3hip_last_error().decode())


_SZ_TYPES = {"float32": 0, "float64": 1, "uint8": 2, "int8": 3, "uint16": 4, "int16": 5, "uint32": 6, "int32": 7, "uint64": 8, "int64": 9}


def _dtype_id(dt):
    """SZ_FLOAT .. SZ_INT64 (include/SZ3/def.hpp:27-36); integers: host-buffer API only (compress / decompress), they ride the f64 pipeline"""
    dt = np.dtype(dt)
    if dt.name in _SZ_TYPES:
        return _SZ_TYPES[dt.name]
    raise TypeError("sz3_amd supports float32 / float64 and 8 ... 64-bit integers (got %s)" % dt)


class Config:
    """Mirror of SZ3::Config (include/SZ3/utils/Config.hpp:138-479): ``Config(d0, d1, ...)`` with dims slowest first,
    size-1 dims dropped; public fields with the reference's names and defaults."""

    _FIELDS = [f[0] for f in _CConfig._fields_ if f[0] not in ("dims",)]

    def __init__(self, *dims):
        if len(dims) == 1 and isinstance(dims[0], (tuple, list)):
            dims = tuple(dims[0])
        if not dims:
            dims = (1,)
        self._c = _CConfig()
        self.setDims(dims)

    def setDims(self, dims):
        dims = [int(d) for d in dims]
        saved = {k: getattr(self._c, k) for k in self._FIELDS} if getattr(self._c, "num", 0) else None
        arr = (C.c_uint64 * len(dims))(*dims)
        lib().sz3hip_config_init(C.byref(self._c), len(dims), arr)
        if saved:  # setDims only touches dims/N/num/predDim/blockSize (Config.hpp:161-177)
            for k, v in saved.items():
                if k not in ("N", "num", "predDim", "blockSize"):
                    setattr(self._c, k, v)
        return int(self._c.num)

    @property
    def dims(self):
        return tuple(int(self._c.dims[i]) for i in range(self._c.N))

    def __getattr__(self, k):
        if k != "_c" and k in Config._FIELDS:
            return getattr(self._c, k)
        raise AttributeError(k)

    def __setattr__(self, k, v):
        if k != "_c" and k in Config._FIELDS:
            setattr(self._c, k, v)
        else:
            object.__setattr__(self, k, v)

    def save(self):
        buf = (C.c_ubyte * 256)()
        n = lib().sz3hip_config_save(C.byref(self._c), buf)
        return bytes(buf[:n])

    @classmethod
    def load(cls, raw):
        c = cls(1)
        b = (C.c_ubyte * len(raw)).from_buffer_copy(raw)
        if lib().sz3hip_config_load_n(C.byref(c._c), b, len(raw)) == 0:
            raise ValueError("truncated serialised Config")
        return c

    def copy(self):
        c = Config(1)
        C.memmove(C.byref(c._c), C.byref(self._c), C.sizeof(_CConfig))
        return c


# ---- pysz-like host API (tools/pysz/src/pysz/sz.pyx) -----------------------------------------------------------
def set_stock_format(on=True):
    """on: compress() writes streams stock SZ3 reads wherever this library has a stock form (ALGO_INTERP, which is what the default
    ALGO_INTERP_LORENZO resolves to on most data); reading stock ALGO_INTERP / ALGO_LOSSLESS streams needs no switch"""
    lib().sz3hip_set_stock_format(int(on))


def compress_bound(conf, dtype):
    return int(lib().sz3hip_compress_bound(C.byref(conf._c), _dtype_id(dtype)))


def compress(data, conf, out=None):
    """sz.compress (sz.pyx:185-272): returns (uint8 ndarray, ratio). `out`: a caller's uint8 buffer of at least
    compress_bound(conf, dtype) bytes — the pre-allocated form, SZ_compress(conf, data, cmpData, cmpCap) (api/sz.hpp:43-62); a buffer
    used before costs no first-touch page faults."""
    a = np.ascontiguousarray(data)
    dt = _dtype_id(a.dtype)
    if int(conf.num) != a.size:
        raise ValueError("config dims do not match the array")
    cap = compress_bound(conf, a.dtype)
    if out is None:
        out = np.empty(cap, dtype=np.uint8)
    elif out.dtype != np.uint8 or not out.flags.c_contiguous or out.size < cap:
        raise ValueError("out must be a contiguous uint8 array of at least compress_bound(conf, dtype) = %d bytes" % cap)
    else:
        cap = out.size
    n = lib().sz3hip_compress(C.byref(conf._c), dt, a.ctypes.data, out.ctypes.data, cap)
    if n == 0:
        raise SZ3HipError(-1, lib().sz3hip_last_error().decode())
    blob = out[:n]  # a view, like pysz (sz.pyx:230-272): the untouched rest of the bound-sized buffer is never resident
    return blob, a.nbytes / float(n)


def decompress(blob, dtype, shape=None, out=None):
    """sz.decompress (sz.pyx:276-365): returns (ndarray, Config). `out`: a caller's array of the stream's element count and `dtype`
    — the pre-allocated form, SZ_decompress(conf, cmpData, cmpSize, decData) (api/sz.hpp:84-110)."""
    blob = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8) if isinstance(blob, (bytes, bytearray)) else blob)
    conf = Config(1)
    _check(lib().sz3hip_peek_config(C.byref(conf._c), blob.ctypes.data, blob.size))
    if out is None:
        dec = np.empty(int(conf.num), dtype=dtype)
    else:
        if out.dtype != np.dtype(dtype) or not out.flags.c_contiguous or out.size != int(conf.num):
            raise ValueError("out must be a contiguous %s array of %d elements" % (np.dtype(dtype), int(conf.num)))
        dec = out.reshape(-1)
    _check(lib().sz3hip_decompress(C.byref(conf._c), _dtype_id(dtype), blob.ctypes.data, blob.size, dec.ctypes.data))
    if shape is None:
        shape = conf.dims
    return dec.reshape(shape), conf


def verify(ori, dec):
    """sz.verify (sz.pyx:368-405, utils/Statistic.hpp:80-137): (max_diff, psnr, nrmse) in float64."""
    o = np.asarray(ori, dtype=np.float64).ravel()
    d = np.asarray(dec, dtype=np.float64).ravel()
    err = np.abs(d - o)
    rng = o.max() - o.min()
    mse = float(np.mean(err * err))
    psnr = 20 * np.log10(rng) - 10 * np.log10(mse) if mse > 0 and rng > 0 else float("inf")
    nrmse = np.sqrt(mse) / rng if rng > 0 else 0.0
    return float(err.max()), float(psnr), float(nrmse)


# ---- device-resident API ---------------------------------------------------------------------------------------
class DeviceCompressor:
    """Workspace + entry points for arrays that already live in HBM (pointers are plain ints, e.g. tensor.data_ptr())."""

    def __init__(self, max_elems, dtype, device=0):
        self.dtype = np.dtype(dtype)
        self._h = lib().sz3hip_ctx_create(int(device), int(max_elems), _dtype_id(dtype))
        if not self._h:
            raise SZ3HipError(-4, lib().sz3hip_last_error().decode())
        self.max_elems = int(max_elems)

    def close(self):
        if self._h:
            lib().sz3hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def payload_bound(self, n, worst_case=False):
        """device payload bound; worst_case=True leaves room for outlier lists of n / 8 entries (compress then grows
        its lists on demand instead of raising SZ3HIP_EOUTLIERS)"""
        if worst_case:
            return int(lib().sz3hip_payload_bound_max(self._h, int(n)))
        return int(lib().sz3hip_payload_bound(self._h, int(n)))

    def minmax(self, d_in, n, stream=0):
        mn, mx = C.c_double(), C.c_double()
        _check(lib().sz3hip_minmax_device(self._h, d_in, int(n), C.byref(mn), C.byref(mx), stream))
        return mn.value, mx.value

    def stage1(self, conf, d_in, stream=0):
        _check(lib().sz3hip_compress_stage1(self._h, C.byref(conf._c), d_in, stream))

    def histogram_ptr(self):
        return int(lib().sz3hip_histogram_ptr(self._h)), int(lib().sz3hip_histogram_len(self._h))

    def set_histogram(self, d_hist):
        _check(lib().sz3hip_ctx_set_histogram(self._h, d_hist))

    def stage2(self, d_payload, cap, stream=0):
        _check(lib().sz3hip_compress_stage2(self._h, d_payload, int(cap), stream))

    def finish(self, stream=0):
        sz = C.c_size_t()
        _check(lib().sz3hip_compress_finish(self._h, C.byref(sz), stream))
        return int(sz.value)

    def compress(self, conf, d_in, d_payload, cap, stream=0):
        sz = C.c_size_t()
        _check(lib().sz3hip_compress_device(self._h, C.byref(conf._c), d_in, d_payload, int(cap), C.byref(sz), stream))
        return int(sz.value)

    def decompress(self, d_payload, size, d_out, stream=0):
        _check(lib().sz3hip_decompress_device(self._h, d_payload, int(size), d_out, stream))

    def stats(self):
        st = _CStats()
        lib().sz3hip_get_stats(self._h, C.byref(st))
        return {k: int(getattr(st, k)) for k, _ in _CStats._fields_}

    def tuner_report(self):
        """what the ALGO_INTERP_LORENZO auto-tuner decided in the last stage1 / compress call (sz3hip_get_tuner_report)"""
        r = _CTunerReport()
        lib().sz3hip_get_tuner_report(self._h, C.byref(r))
        out = {k: getattr(r, k) for k, _ in _CTunerReport._fields_ if k not in ("est_bytes", "speculated")}
        out["est_bytes"] = [float(x) for x in r.est_bytes]
        self.speculated = int(r.speculated)  # (how stage 1 related to the tuner; not part of the decision the report describes)
        return out

    def set_profiling(self, on=True):
        lib().sz3hip_set_profiling(self._h, int(on))

    def forget(self):
        """drop what earlier calls left in the context (kernel forms, windows, tuner outcome, code book): the next call is a first call"""
        lib().sz3hip_ctx_forget(self._h)

    def set_speculation(self, on=True, backoff=True):
        """on=False: every stage 2 builds its code book first; backoff=False: a miss does not make the next calls sit out (tests)"""
        lib().sz3hip_ctx_set_speculation(self._h, (0 if backoff else 2) if on else 1)

    def set_deterministic(self, on=True):
        """on: the previous call's code book stands only when it IS this call's book — the payload is a pure function of the input
        (off, the device API's default: also when it is complete over this call's alphabet and within 1/1024 of its own book's size)"""
        lib().sz3hip_ctx_set_deterministic(self._h, int(on))

    def set_tuner_exact(self, on=True):
        """on: the ALGO_INTERP_LORENZO tuner prices its trials the reference's way (Huffman tree + bits + zstd on the host: the reference's
        own compressed sizes and decisions, milliseconds per tuning); off: the device-side estimate (include/sz3hip.h)"""
        lib().sz3hip_ctx_set_tuner_exact(self._h, int(on))

    def set_fused(self, on=True):
        """opt in to the fused stage 1 (the previous call's book codes inside the predictor kernel; off by default)"""
        lib().sz3hip_ctx_set_fused(self._h, int(on))

    @property
    def fused(self):
        """the last finished compression ran the fused stage 1 (coded with the previous call's book inside the predictor kernel)"""
        return bool(lib().sz3hip_last_call_fused(self._h))

    @property
    def q16(self):
        """the last finished compression ran the 16-bit form of the one-byte stage-1 kernel (f32, lattice values within +-4095)"""
        return bool(lib().sz3hip_last_call_q16(self._h))

    def spec_stats(self):
        h, m = C.c_uint32(), C.c_uint32()
        lib().sz3hip_get_spec_stats(self._h, C.byref(h), C.byref(m))
        return int(h.value), int(m.value)

    def payload_bound_conf(self, conf, worst_case=False):
        return int(lib().sz3hip_payload_bound_conf(self._h, C.byref(conf._c), int(bool(worst_case))))

    def stage_times(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        k = lib().sz3hip_get_stage_times(self._h, names, ms, 16)
        return {names[i].decode(): float(ms[i]) for i in range(k)}

    def debug_codes(self, n):
        out = np.empty(int(n), dtype=np.uint16)
        _check(lib().sz3hip_debug_copy_codes(self._h, out.ctypes.data, int(n)))
        return out


# ---- multi-GPU exchange (RCCL over xGMI, in the library) --------------------------------------------------------
COMM_ID_BYTES = 128


class Comm:
    """sz3hip_comm: the RCCL communicator of the slab-parallel path. ``Comm.local(ndev)`` = one process drives ndev GPUs;
    ``Comm.rank(nranks, rank, device, id)`` = one process per GPU, ``id`` being ``Comm.unique_id()`` of rank 0 shipped by
    the launcher (sz3_amd.distributed.init_comm does that over torch.distributed)."""

    def __init__(self, handle):
        if not handle:
            raise SZ3HipError(lib().sz3hip_last_error_code(), lib().sz3hip_last_error().decode())
        self._h = handle

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * COMM_ID_BYTES)()
        _check(lib().sz3hip_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def local(cls, ndev=0, devices=None):
        arr = (C.c_int * len(devices))(*devices) if devices else None
        return cls(lib().sz3hip_comm_create_local(int(ndev if not devices else len(devices)), arr))

    @classmethod
    def rank(cls, nranks, rank, device, id_bytes):
        b = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        return cls(lib().sz3hip_comm_create_rank(int(nranks), int(rank), int(device), b))

    def close(self):
        if self._h:
            lib().sz3hip_comm_destroy(self._h)
            self._h = None

    @property
    def size(self):
        return int(lib().sz3hip_comm_size(self._h))

    @property
    def my_rank(self):
        return int(lib().sz3hip_comm_rank(self._h))

    @property
    def local_size(self):
        return int(lib().sz3hip_comm_local_size(self._h))

    def device(self, member=0):
        return int(lib().sz3hip_comm_device(self._h, member))

    def allreduce_histogram(self, compressors, streams):
        """in-place sum of the code histograms of the local members' DeviceCompressors, enqueued on their streams"""
        m = len(compressors)
        ctxs = (C.c_void_p * m)(*[dc._h for dc in compressors])
        st = (C.c_void_p * m)(*[int(x) if x else None for x in streams])
        _check(lib().sz3hip_comm_allreduce_histogram(self._h, ctxs, st))

    def allreduce_u64(self, ptrs, count, streams):
        m = len(ptrs)
        bufs = (C.c_void_p * m)(*[int(p) for p in ptrs])
        st = (C.c_void_p * m)(*[int(x) if x else None for x in streams])
        _check(lib().sz3hip_comm_allreduce_u64(self._h, bufs, int(count), st))

    def allreduce_minmax(self, mins, maxs, streams):
        m = len(mins)
        a = (C.c_double * m)(*mins)
        b = (C.c_double * m)(*maxs)
        st = (C.c_void_p * m)(*[int(x) if x else None for x in streams])
        _check(lib().sz3hip_comm_allreduce_minmax(self._h, a, b, st))
        return list(a), list(b)


def compress_rank(comm, slab, global_conf):
    """this rank's slab of a slab-parallel compress (sz3hip_compress_rank; collective): returns (blob, slab Config)"""
    a = np.ascontiguousarray(slab)
    sc = Config(1)
    cap = int(lib().sz3hip_compress_bound(C.byref(global_conf._c), _dtype_id(a.dtype)))  # (>= the slab's own bound)
    out = np.empty(cap, dtype=np.uint8)
    n = lib().sz3hip_compress_rank(comm._h, C.byref(global_conf._c), _dtype_id(a.dtype), a.ctypes.data, out.ctypes.data, cap, C.byref(sc._c))
    if n == 0:
        raise SZ3HipError(lib().sz3hip_last_error_code(), lib().sz3hip_last_error().decode())
    return out[:n].copy(), sc


def assemble_container(global_conf, dtype, slab_confs, blobs):
    """sz3hip_assemble_container: the multi-slab SZ3 stream from the ranks' blobs and Configs (rank order)"""
    G = len(blobs)
    arrs = [np.ascontiguousarray(np.frombuffer(b, dtype=np.uint8) if isinstance(b, (bytes, bytearray)) else b) for b in blobs]
    confs = (_CConfig * G)()
    for i, c in enumerate(slab_confs):
        C.memmove(C.byref(confs[i]), C.byref(c._c), C.sizeof(_CConfig))
    ptrs = (C.c_void_p * G)(*[a.ctypes.data for a in arrs])
    sizes = (C.c_size_t * G)(*[a.size for a in arrs])
    cap = 16 + 4 + G * (8 + 160) + 160 + sum(a.size for a in arrs)
    out = np.empty(cap, dtype=np.uint8)
    n = lib().sz3hip_assemble_container(C.byref(global_conf._c), _dtype_id(dtype), G, confs, ptrs, sizes, out.ctypes.data, cap)
    if n == 0:
        raise SZ3HipError(lib().sz3hip_last_error_code(), lib().sz3hip_last_error().decode())
    return out[:n]
