"""Builds (gcc) and loads tests/h5stub/h5stub.c — the stand-in for libhdf5 the HDF5-plugin tests drive set_local through — ONCE per
process: the plugin caches the HDF5 function pointers it resolves, so every test of a process must talk to the same copy. Loaded
RTLD_LOCAL (an application's private libhdf5, h5py's way): the plugin finds it by walking the loaded objects."""
import ctypes as C
import os
import shutil
import subprocess
import tempfile

_LIB = None


def load_stub():
    global _LIB
    if _LIB is not None:
        _LIB.h5stub_reset()
        return _LIB
    gcc = shutil.which("gcc")
    if not gcc:
        return None
    d = tempfile.mkdtemp(prefix="sz3hip_h5stub_")
    so = os.path.join(d, "libhdf5_stubfortests.so")
    subprocess.check_call([gcc, "-O1", "-shared", "-fPIC", os.path.join(os.path.dirname(os.path.abspath(__file__)), "h5stub", "h5stub.c"), "-o", so])
    lib = C.CDLL(so, mode=os.RTLD_LOCAL)
    for f in ("h5stub_plist_new", "h5stub_type_new", "h5stub_space_new"):
        getattr(lib, f).restype = C.c_int64
    lib.h5stub_type_new.argtypes = [C.c_int, C.c_size_t, C.c_int]
    lib.h5stub_space_new.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
    lib.H5Pset_filter.argtypes = [C.c_int64, C.c_int, C.c_uint, C.c_size_t, C.c_void_p]
    lib.H5Pget_filter_by_id2.argtypes = [C.c_int64, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint)]
    lib.H5Pget_nfilters.argtypes = [C.c_int64]
    _LIB = lib
    return lib
