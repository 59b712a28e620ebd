"""GPU test (-m gpu) of the HDF5 filter face (include/sz3hip_h5z.h; tools/H5Z-SZ3/src/H5Z_SZ3.cpp:154-227): a chunk through the
filter function HDF5 would call — cd_values = Config::save bytes, buffers malloc'ed and swapped in place — forward and in the
read direction (H5Z_FLAG_REVERSE), for the element types the library has a path for; garbage in the read direction fails."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import sz3_amd  # noqa: E402
from fields import field3d  # noqa: E402
from test_capi_cpu import _H5ZClass2, _cd_values  # noqa: E402

SZ_TYPES = {np.float32: 0, np.float64: 1, np.int32: 7, np.int64: 9}


@pytest.mark.parametrize("dtype,eb", [(np.float32, 1e-3), (np.float64, 1e-6), (np.int32, 2.0), (np.int64, 3.0)])
def test_a_chunk_through_the_filter_function(dtype, eb):
    L = sz3_amd.lib()
    L.H5PLget_plugin_info.restype = C.POINTER(_H5ZClass2)
    rec = L.H5PLget_plugin_info().contents
    filt = C.CFUNCTYPE(C.c_size_t, C.c_uint, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p))(rec.filter)
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.free.argtypes = [C.c_void_p]
    shape = (40, 48, 56)
    a = field3d(shape, np.float64)
    a = (a * 1000).astype(dtype) if np.issubdtype(dtype, np.integer) else a.astype(dtype)
    conf = sz3_amd.Config(*shape)
    conf.absErrorBound = eb
    conf.dataType = SZ_TYPES[dtype]
    cdv, words = _cd_values(conf._c)
    buf = C.c_void_p(libc.malloc(a.nbytes))
    C.memmove(buf, a.ctypes.data, a.nbytes)
    size = C.c_size_t(a.nbytes)
    n = filt(0, words, cdv, a.nbytes, C.byref(size), C.byref(buf))
    assert 0 < n < a.nbytes and size.value == n
    stream = C.string_at(buf, n)
    ref, _ = sz3_amd.decompress(np.frombuffer(stream, dtype=np.uint8), dtype, shape)  # the filter's output IS an SZ3 container of this library
    m = filt(0x0100, words, cdv, n, C.byref(size), C.byref(buf))
    assert m == a.nbytes and size.value == a.nbytes
    out = np.frombuffer(C.string_at(buf, m), dtype=dtype).reshape(shape)
    assert np.array_equal(out, ref)
    assert float(np.max(np.abs(out.astype(np.float64) - a.astype(np.float64)))) <= eb
    # read direction on bytes that are no stream: failure (0), the buffer untouched and still ours to free
    junk = C.c_void_p(libc.malloc(4096))
    C.memset(junk, 0x5A, 4096)
    js = C.c_size_t(4096)
    assert filt(0x0100, words, cdv, 4096, C.byref(js), C.byref(junk)) == 0
    libc.free(junk)
    libc.free(buf)
