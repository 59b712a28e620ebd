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


ALL_TYPES = [(np.float32, (1, 4, 1), 1e-3), (np.float64, (1, 8, 1), 1e-6), (np.uint8, (0, 1, 0), 2.0), (np.int8, (0, 1, 1), 2.0),
             (np.uint16, (0, 2, 0), 3.0), (np.int16, (0, 2, 1), 3.0), (np.uint32, (0, 4, 0), 4.0), (np.int32, (0, 4, 1), 4.0),
             (np.uint64, (0, 8, 0), 5.0), (np.int64, (0, 8, 1), 5.0)]


@pytest.mark.parametrize("dtype,h5type,eb", ALL_TYPES, ids=[np.dtype(t[0]).name for t in ALL_TYPES])
def test_a_dataset_the_way_hdf5_drives_the_plugin(dtype, h5type, eb):
    """What HDF5 does with a filter plugin when a chunked dataset is created with filter 32024 and written (H5Z_SZ3.cpp:74-227;
    tools/test/integration/test_h5_filter.py:19-35 through h5py): the user's cd_values (a Config that knows neither the dataset's type nor
    its chunk shape) sit on the creation property list; the record's set_local is called with the list, the element type and the
    chunk's dataspace; every chunk then goes through the record's filter function with the list's cd_values — forward on write, with
    H5Z_FLAG_REVERSE on read. HDF5 itself is played by tests/h5stub (not in this image). All ten element types of the reference's
    filter; integers come back within floor(eb), exactly representable values included at the type's limits."""
    from h5stub_loader import load_stub
    h5 = load_stub()
    if h5 is None:
        pytest.skip("no gcc")
    L = sz3_amd.lib()
    L.H5PLget_plugin_info.restype = C.POINTER(_H5ZClass2)
    rec = L.H5PLget_plugin_info().contents
    set_local = C.CFUNCTYPE(C.c_int, C.c_int64, C.c_int64, C.c_int64)(rec.set_local)
    filt = C.CFUNCTYPE(C.c_size_t, C.c_uint, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p))(rec.filter)
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.free.argtypes = [C.c_void_p]
    chunk = (1, 36, 44, 52)  # (HDF5 hands the chunk's full rank over: the extent of 1 is dropped by set_local like Config::setDims does)
    shape = chunk[1:]
    info = np.iinfo(dtype) if np.issubdtype(dtype, np.integer) else None
    if info is not None:
        f = field3d(shape, np.float64, sigma=0.0)
        span = float(info.max) - float(info.min)
        a01 = (f - f.min()) / (f.max() - f.min())                    # 0 .. 1, smooth
        scale = min(span, 2.0 ** 16)                                 # (wide types: a smooth field of 2^16 levels somewhere inside the range)
        a = np.floor(a01 * scale).astype(dtype)
        # the type's limits (64-bit: +-2^40 — beyond 2^53 an array is kept lossless, which is not what this test is about)
        lo, hi = (int(info.min), int(info.max)) if a.itemsize < 8 else (0 if info.min == 0 else -2 ** 40, 2 ** 40)
        a.reshape(-1)[:4] = [lo, hi, 0, 1]
    else:
        a = field3d(shape, np.float64, sigma=2e-6 if dtype is np.float64 else 2e-3).astype(dtype)
    user = sz3_amd.Config(5)            # what an application passes as compression_opts: made for no dataset in particular
    user.absErrorBound = eb
    user.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    cdv, words = _cd_values(user._c)
    dcpl = h5.h5stub_plist_new()
    assert h5.H5Pset_filter(dcpl, 32024, 0, words, cdv) == 0
    arr = (C.c_ulonglong * 4)(*chunk)
    assert set_local(dcpl, h5.h5stub_type_new(*h5type), h5.h5stub_space_new(4, arr)) > 0
    cd = (C.c_uint * 64)()
    ncd = C.c_size_t(64)
    fl, fc = C.c_uint(0), C.c_uint(0)
    assert h5.H5Pget_filter_by_id2(dcpl, 32024, C.byref(fl), C.byref(ncd), cd, 0, None, C.byref(fc)) == 0
    buf = C.c_void_p(libc.malloc(a.nbytes))
    C.memmove(buf, a.ctypes.data, a.nbytes)
    size = C.c_size_t(a.nbytes)
    n = filt(0, ncd.value, cd, a.nbytes, C.byref(size), C.byref(buf))       # write
    assert 0 < n and size.value == n
    if a.itemsize >= 4:
        assert n < a.nbytes
    m = filt(0x0100, ncd.value, cd, n, C.byref(size), C.byref(buf))          # read
    assert m == a.nbytes and size.value == a.nbytes
    out = np.frombuffer(C.string_at(buf, m), dtype=dtype).reshape(shape)
    libc.free(buf)
    if info is not None:
        err = np.max(np.abs(out.astype(np.float64) - a.astype(np.float64)))
        assert err <= np.floor(eb), (err, eb)
    else:
        assert float(np.max(np.abs(out.astype(np.float64) - a.astype(np.float64)))) <= eb
