"""CPU tests of the HDF5 plugin's "set local" callback and the two conf helpers (tools/H5Z-SZ3/src/H5Z_SZ3.cpp:26-150) through a small
stand-in for libhdf5 (tests/h5stub/h5stub.c: property lists with a filter pipeline, datatypes, dataspaces — HDF5 itself is not in
this image). The plugin links against no HDF5: it finds the functions in the process that loaded it, which is what is exercised here —
with the stand-in loaded RTLD_LOCAL (an application's private copy, h5py's way: found by walking the loaded objects)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import sz3_amd
from test_capi_cpu import _H5ZClass2, _cd_values

HERE = os.path.dirname(os.path.abspath(__file__))
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
SGN_NONE, SGN_2 = 0, 1


@pytest.fixture()
def h5():
    from h5stub_loader import load_stub
    lib = load_stub()
    if lib is None:
        pytest.skip("no gcc")
    return lib


def _plugin():
    L = sz3_amd.lib()
    L.H5PLget_plugin_info.restype = C.POINTER(_H5ZClass2)
    rec = L.H5PLget_plugin_info().contents
    set_local = C.CFUNCTYPE(C.c_int, C.c_int64, C.c_int64, C.c_int64)(rec.set_local)
    L.sz3hip_h5z_conf_to_H5.argtypes = [C.c_int64, C.c_void_p]
    L.sz3hip_h5z_conf_from_H5.argtypes = [C.c_int64, C.c_void_p]
    return L, rec, set_local


def _space(h5, dims):
    arr = (C.c_ulonglong * len(dims))(*dims)
    return h5.h5stub_space_new(len(dims), arr)


def _conf_on(h5, plist):
    """the Config in the list's cd_values for filter 32024 (None: the filter is not on the list)"""
    cd = (C.c_uint * 64)()
    n = C.c_size_t(64)
    flags, fc = C.c_uint(0), C.c_uint(0)
    if h5.H5Pget_filter_by_id2(plist, 32024, C.byref(flags), C.byref(n), cd, 0, None, C.byref(fc)) < 0:
        return None, 0
    c = sz3_amd.Config(1)
    assert sz3_amd.lib().sz3hip_config_load_n(C.byref(c._c), cd, n.value * 4) > 0
    return c, n.value


CASES = [  # (class, size, sign) -> SZ data type (include/SZ3/def.hpp:27-36), as tools/H5Z-SZ3/src/H5Z_SZ3.cpp:99-137 maps them
    ((H5T_FLOAT, 4, SGN_2), 0), ((H5T_FLOAT, 8, SGN_2), 1), ((H5T_INTEGER, 1, SGN_NONE), 2), ((H5T_INTEGER, 1, SGN_2), 3),
    ((H5T_INTEGER, 2, SGN_NONE), 4), ((H5T_INTEGER, 2, SGN_2), 5), ((H5T_INTEGER, 4, SGN_NONE), 6), ((H5T_INTEGER, 4, SGN_2), 7),
    ((H5T_INTEGER, 8, SGN_NONE), 8), ((H5T_INTEGER, 8, SGN_2), 9)]


def test_set_local_fills_type_and_chunk_shape_into_the_users_config(h5):
    """H5Z_SZ3.cpp:74-150: the user's Config (bound, algorithm ...) arrives in cd_values with whatever dims and type it was made with;
    set_local replaces element type and extents by the dataset's (extents of 1 dropped, blockSize following N, Config.hpp:152-177) and
    leaves the rest; every one of the ten element types is mapped like the reference maps it."""
    L, rec, set_local = _plugin()
    assert rec.set_local  # (round 3: NULL)
    for (cls, size, sign), want in CASES:
        h5.h5stub_reset()
        user = sz3_amd.Config(7)           # made without knowing the dataset
        user.errorBoundMode = sz3_amd.EB_REL
        user.relErrorBound = 2.5e-4
        user.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
        user.quantbinCnt = 1024
        cdv, words = _cd_values(user._c)
        dcpl = h5.h5stub_plist_new()
        assert h5.H5Pset_filter(dcpl, 32024, 0, words, cdv) == 0
        rc = set_local(dcpl, h5.h5stub_type_new(cls, size, sign), _space(h5, [1, 30, 1, 40, 50]))
        assert rc > 0 and h5.H5Pget_nfilters(dcpl) == 1  # (modified in place: a second entry of the same id breaks decompression, :46-47)
        got, _ = _conf_on(h5, dcpl)
        assert got.dataType == want, (cls, size, sign, got.dataType)
        assert (got.N, list(got.dims[:3]), got.num, got.blockSize) == (3, [30, 40, 50], 60000, 6)
        assert (got.errorBoundMode, got.relErrorBound, got.cmprAlgo, got.quantbinCnt) == (sz3_amd.EB_REL, 2.5e-4, sz3_amd.ALGO_LORENZO_REG, 1024)
    assert C.c_int.in_dll(h5, "h5stub_calls").value > 0  # the plugin found the stand-in's functions although nobody exported them globally


def test_set_local_without_user_values_and_its_refusals(h5):
    L, rec, set_local = _plugin()
    h5.h5stub_reset()
    # h5py's compression=32024 with no compression_opts: cd_nelmts == 0 -> a default Config with the dataset's type and shape
    dcpl = h5.h5stub_plist_new()
    assert h5.H5Pset_filter(dcpl, 32024, 0, 0, None) == 0
    assert set_local(dcpl, h5.h5stub_type_new(H5T_FLOAT, 8, SGN_2), _space(h5, [64, 128])) > 0
    got, n = _conf_on(h5, dcpl)
    ref = sz3_amd.Config(64, 128)
    assert n > 0 and (got.N, list(got.dims[:2]), got.dataType, got.blockSize) == (2, [64, 128], 1, 16)
    assert (got.cmprAlgo, got.errorBoundMode, got.absErrorBound, got.quantbinCnt) == (ref.cmprAlgo, ref.errorBoundMode, ref.absErrorBound, ref.quantbinCnt)
    # five extents above 1: SZ_compress refuses N > 4 (api/sz.hpp:71); a string class: neither integer nor float (:139, returns 0); bad ids
    assert set_local(dcpl, h5.h5stub_type_new(H5T_FLOAT, 4, SGN_2), _space(h5, [2, 3, 4, 5, 6])) < 0
    assert set_local(dcpl, h5.h5stub_type_new(H5T_STRING, 4, SGN_2), _space(h5, [64, 128])) == 0
    assert set_local(dcpl, 4242, _space(h5, [64, 128])) < 0
    assert set_local(dcpl, h5.h5stub_type_new(H5T_FLOAT, 4, SGN_2), 4242) < 0


def test_conf_helpers_set_then_modify(h5):
    """set_SZ3_conf_to_H5 / get_SZ3_conf_from_H5 (H5Z_SZ3.cpp:26-72): the first call puts the filter on the list, later calls modify
    it (never a second entry); a list without the filter reads back as a default Config"""
    L, rec, set_local = _plugin()
    h5.h5stub_reset()
    dcpl = h5.h5stub_plist_new()
    blank = sz3_amd.Config(3)
    assert L.sz3hip_h5z_conf_from_H5(dcpl, C.byref(blank._c)) == 1 and (blank.N, blank.num) == (1, 1)
    a = sz3_amd.Config(20, 30)
    a.absErrorBound = 0.125
    assert L.sz3hip_h5z_conf_to_H5(dcpl, C.byref(a._c)) == 1 and h5.H5Pget_nfilters(dcpl) == 1
    a.absErrorBound = 0.5
    a.cmprAlgo = sz3_amd.ALGO_INTERP
    assert L.sz3hip_h5z_conf_to_H5(dcpl, C.byref(a._c)) == 1 and h5.H5Pget_nfilters(dcpl) == 1
    back = sz3_amd.Config(1)
    assert L.sz3hip_h5z_conf_from_H5(dcpl, C.byref(back._c)) == 1
    assert (back.N, list(back.dims[:2]), back.absErrorBound, back.cmprAlgo) == (2, [20, 30], 0.5, sz3_amd.ALGO_INTERP)


def test_reference_named_cxx_helpers_compile_and_link(tmp_path, h5):
    """include/H5Z_SZ3.hpp: set_SZ3_conf_to_H5 / get_SZ3_conf_from_H5 with the reference's names and SZ3::Config& signatures
    (tools/H5Z-SZ3/include/H5Z_SZ3.hpp:51,53), for applications that configure the filter the reference's way"""
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(HERE)
    src = tmp_path / "use.cpp"
    src.write_text('#include "H5Z_SZ3.hpp"\n'
                   'int main() { SZ3::Config c(20, 30); c.absErrorBound = 0.25; hid_t p = 77; herr_t (*f)(const hid_t, SZ3::Config &) = set_SZ3_conf_to_H5; '
                   'herr_t (*g)(const hid_t, SZ3::Config &) = get_SZ3_conf_from_H5; return (f && g && H5Z_FILTER_SZ3 == 32024 && p == 77) ? 0 : 1; }\n')
    exe = str(tmp_path / "use")
    subprocess.check_call([gxx, "-std=c++17", "-I" + os.path.join(root, "include"), str(src), "-o", exe, "-L" + os.path.join(root, "sz3_amd"), "-lsz3hip",
                           "-Wl,-rpath," + os.path.join(root, "sz3_amd")])
    assert subprocess.run([exe]).returncode == 0
