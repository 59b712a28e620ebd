"""CPU tests (-m "not gpu") of the product's host side: libsz3hip.so loads, exports every symbol include/*.h declares,
Config semantics / serialisation are byte-identical to the reference's (through the oracle), and every compute entry
point FAILS LOUDLY without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import sz3_amd

HERE = os.path.dirname(os.path.abspath(__file__))
from oracle_binding import make_config, oracle, SzoConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sz3hip_[a-z0-9_]+|SZ_compress_args|SZ_decompress|free_buf)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = sz3_amd.lib()
    names = _declared_symbols("sz3hip.h") + _declared_symbols("sz3c.h")
    assert "sz3hip_compress_stage1" in names and "SZ_compress_args" in names and len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libsz3hip.so does not export " + n


@pytest.mark.parametrize("dims", [(100,), (1, 500), (3, 1, 7, 1), (512, 512, 512), (100, 500, 500, 500), (1,), (2 ** 33, 3)])
def test_config_matches_reference_semantics(dims):
    c = sz3_amd.Config(*dims)
    o = make_config(dims, algo=1, regression=True)  # defaults of SZ3::Config (Config.hpp:452-478)
    assert c.N == o.N and c.dims == tuple(int(o.dims[i]) for i in range(o.N)) and c.num == o.num
    assert (c.blockSize, c.quantbinCnt, c.cmprAlgo, c.lorenzo, c.regression, c.interpAnchorStride) == \
           (o.blockSize, o.quantbinCnt, o.cmprAlgo, o.lorenzo, o.regression, o.interpAnchorStride)
    buf = (C.c_ubyte * 256)()
    n = oracle().szo_config_save(C.byref(o), buf)
    assert c.save() == bytes(buf[:n]), "Config::save bytes differ from the reference layout"
    for mode, kw in [(sz3_amd.EB_REL, dict(relErrorBound=1e-3)), (sz3_amd.EB_ABS_AND_REL, dict(absErrorBound=0.5, relErrorBound=1e-4)),
                     (sz3_amd.EB_PSNR, dict(psnrErrorBound=80.0)), (sz3_amd.EB_L2NORM, dict(l2normErrorBound=2.5))]:
        c.errorBoundMode = o.errorBoundMode = mode
        for k, v in kw.items():
            setattr(c, k, v)
            setattr(o, k, v)
        n = oracle().szo_config_save(C.byref(o), buf)
        assert c.save() == bytes(buf[:n])
        back = sz3_amd.Config.load(c.save())
        assert back.dims == c.dims and back.errorBoundMode == mode and back.save() == c.save()


def test_observed_trailer_bytes():
    # 8x8x128 f32, Lorenzo only, eb 1e-3 -> 35-byte trailer "23 03 08 08 08 80 ..."
    c = sz3_amd.Config(8, 8, 128)
    c.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    c.regression = 0
    raw = c.save()
    assert len(raw) == 35 and raw[:6].hex() == "230308080880"


def test_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    a = np.zeros((8, 8, 8), np.float32)
    with pytest.raises(sz3_amd.SZ3HipError):
        sz3_amd.compress(a, sz3_amd.Config(8, 8, 8))
    with pytest.raises(sz3_amd.SZ3HipError):
        sz3_amd.DeviceCompressor(512, np.float32)
    with pytest.raises(sz3_amd.SZ3HipError):  # (round 4: the 8 ... 64-bit integer types are served — on a device)
        sz3_amd.compress(a.astype(np.uint8), sz3_amd.Config(8, 8, 8))
    with pytest.raises(TypeError):
        sz3_amd.compress(a.astype(np.complex64), sz3_amd.Config(8, 8, 8))


def test_peek_rejects_foreign_streams():
    L = sz3_amd.lib()
    junk = np.zeros(64, dtype=np.uint8)
    conf = sz3_amd.Config(1)
    assert L.sz3hip_peek_config(C.byref(conf._c), junk.ctypes.data, junk.size) == -3  # SZ3HIP_EFORMAT: bad magic
    assert b"magic number mismatch" in L.sz3hip_last_error()


def test_cxx_header_layer_compiles_and_mirrors_config(tmp_path):
    """include/SZ3/api/sz.hpp: a C++ program written against the reference's public header builds against ours and
    links libsz3hip.so; Config semantics (dims, INI dialect, save/load bytes) match the oracle's."""
    import shutil, subprocess, textwrap
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(HERE)
    src = tmp_path / "t.cpp"
    src.write_text(textwrap.dedent(r"""
        #include "SZ3/api/sz.hpp"
        #include "SZ3/utils/Config.hpp"
        int main() {
            SZ3::Config c(100, 1, 300);
            c.load_ini("[GlobalSettings]\nCmprAlgo = algo_interp\nErrorBoundMode=rel\nRelErrorBound = 1e-2\n# c\n"
                       "[AlgoSettings]\nInterpolationAlgo=INTERP_ALGO_LINEAR\nBlockSize = 8\nLorenzo2ndOrder = yes\n");
            unsigned char buf[256]; unsigned char *p = buf; size_t n = c.save(p);
            SZ3::Config d; const unsigned char *q = buf; d.load(q);
            printf("%d %zu %d %d %g %d %d %d %zu %zu %zu\n", d.N, d.num, d.cmprAlgo, d.errorBoundMode, d.relErrorBound,
                   d.interpAlgo, d.blockSize, (int)d.lorenzo2, n, d.dims[0], d.dims[1]);
            for (size_t i = 0; i < n; i++) printf("%02x", buf[i]);
            printf("\n");
            SZ3::Config e({4, 5, 6});
            printf("%d %zu %d\n", e.N, e.num, e.blockSize);
            std::vector<float> x(1000, 1.f);
            try { size_t s; SZ_compress(SZ3::Config(1000), x.data(), s); printf("compressed\n"); }
            catch (const std::exception &ex) { printf("exception\n"); }
            return 0;
        }"""))
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L" + os.path.join(root, "sz3_amd"), "-lsz3hip", "-Wl,-rpath," + os.path.join(root, "sz3_amd")])
    out = subprocess.check_output([str(exe)]).decode().splitlines()
    # interpAlgo is not part of Config::save (it travels with the decomposition, Config.hpp:472-478): back to the default 1
    assert out[0] == "2 30000 2 1 0.01 1 8 1 %d 100 300" % (len(out[1]) // 2)
    # same bytes as the oracle's Config::save for the same settings (itself checked against the reference build)
    o = make_config((100, 300), algo=2, eb_mode=1, rel_eb=1e-2, regression=True, interp_algo=0)
    o.blockSize = 8
    o.lorenzo2 = 1
    buf = (C.c_ubyte * 256)()
    n = oracle().szo_config_save(C.byref(o), buf)
    assert bytes.fromhex(out[1]) == bytes(buf[:n])
    assert out[2] == "3 120 6"
    assert out[3] in ("exception", "compressed")   # no GPU here -> the library refuses loudly; on a GPU box it compresses


def test_reference_cli_builds_against_our_headers():
    """oracle/_ref/sz3_hip = the unmodified reference CLI source compiled against include/SZ3 + libsz3hip.so
    (oracle/Makefile `hipcli`; built by __graft_entry__.build() where /root/reference exists)"""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "sz3_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/sz3_hip not built (needs /root/reference)")
    out = subprocess.run([exe, "-v"], capture_output=True, text=True).stdout
    assert "SZ3 Version: 3.3.2" in out


def test_extension_recipe_header_builds_inside_the_reference_tree(tmp_path):
    """include/SZ3/api/impl/SZAlgoHip.hpp is for the REFERENCE's header tree (its own recipe for a new ALGO,
    tools/sz3/sz3_customized_demo.cpp:8-14): `make -C oracle algohip` applies the three documented edits to scratch copies
    of the reference's Config.hpp / SZDispatcher.hpp and builds the reference CLI with them -> oracle/_ref/sz3_algohip.
    Here (no GPU): the binary exists, knows the new algorithm names, and the library under it refuses loudly."""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "sz3_algohip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/sz3_algohip not built (needs /root/reference)")
    assert "SZ3 Version: 3.3.2" in subprocess.run([exe, "-v"], capture_output=True, text=True).stdout
    a = (np.arange(32 * 32 * 64, dtype=np.float32) * 0.01).reshape(32, 32, 64)  # (the CLI's buffer is 2 x raw: enough only from ~10^4 values on)
    src = tmp_path / "a.f32"
    a.tofile(src)
    ini = tmp_path / "c.ini"
    ini.write_text("[GlobalSettings]\nCmprAlgo = ALGO_HIP_LORENZO\n")
    r = subprocess.run([exe, "-f", "-i", str(src), "-z", str(tmp_path / "a.sz"), "-3", "64", "32", "32", "-c", str(ini), "-M", "ABS", "1e-3"],
                       capture_output=True, text=True, timeout=120)
    from conftest import gpu_available
    if not gpu_available():
        assert r.returncode != 0 and "HIP" in (r.stderr + r.stdout), (r.stdout, r.stderr)  # reached SZ_compress_Hip -> libsz3hip -> no device


def test_c_headers_are_plain_c(tmp_path):
    """include/sz3hip.h and include/sz3c.h are the FFI boundary: they must compile as C99 (cgo / JNI / ctypes generators
    read them as C), not only as C++"""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "sz3hip.h"\n#include "sz3c.h"\n#include "sz3hip_h5z.h"\nint main(void) { return 0; }\n')
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_committed_traffic_file_has_the_key_the_bench_reads():
    """bench.py takes roofline.traffic from profiles/pmc_traffic.json (written by tools/pmc_traffic.py from the PMC passes);
    a renamed kernel must not silently turn it into null"""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    d = json.load(open(path))
    t = d.get("lorenzo_quant_hist_hbm_bytes_per_launch")
    assert isinstance(t, int) and 537_000_000 + 134_000_000 <= t < 2 * 671_000_000  # at least the compulsory read + write


class _H5ZClass2(C.Structure):  # include/sz3hip_h5z.h: HDF5's H5Z_class2_t
    _fields_ = [("version", C.c_int), ("id", C.c_int), ("encoder_present", C.c_uint), ("decoder_present", C.c_uint), ("name", C.c_char_p),
                ("can_apply", C.c_void_p), ("set_local", C.c_void_p), ("filter", C.c_void_p)]


def _cd_values(conf_c):
    buf = (C.c_ubyte * 512)()
    n = sz3_amd.lib().sz3hip_config_save(C.byref(conf_c), buf)
    words = (n + 3) // 4
    return (C.c_uint * words).from_buffer_copy(bytes(buf[:words * 4])), words


def test_hdf5_plugin_face_without_a_device():
    """tools/H5Z-SZ3/src/H5Z_SZ3.cpp:11-24, 179-193: the two symbols HDF5 looks up in a plugin, the class record (version 1, id
    32024, encoder + decoder, the filter function), the pass-through cases — and the filter FAILS (returns 0, HDF5's "filter
    failed") on a box without a HIP device instead of compressing on the CPU"""
    L = sz3_amd.lib()
    L.H5PLget_plugin_type.restype = C.c_int
    L.H5PLget_plugin_info.restype = C.POINTER(_H5ZClass2)
    L.sz3hip_h5z_filter.restype = C.c_size_t
    L.sz3hip_h5z_filter.argtypes = [C.c_uint, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
    assert L.H5PLget_plugin_type() == 0  # H5PL_TYPE_FILTER
    rec = L.H5PLget_plugin_info().contents
    assert (rec.version, rec.id, rec.encoder_present, rec.decoder_present) == (1, 32024, 1, 1) and b"SZ3" in rec.name
    assert rec.filter == C.cast(L.sz3hip_h5z_filter, C.c_void_p).value and rec.can_apply is None
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.free.argtypes = [C.c_void_p]
    data = np.arange(4096, dtype=np.float32)
    buf = C.c_void_p(libc.malloc(data.nbytes))
    C.memmove(buf, data.ctypes.data, data.nbytes)
    size = C.c_size_t(data.nbytes)
    assert L.sz3hip_h5z_filter(0, 0, None, data.nbytes, C.byref(size), C.byref(buf)) == data.nbytes  # cd_nelmts == 0: not values
    small = sz3_amd.Config(10)
    cdv, words = _cd_values(small._c)
    assert L.sz3hip_h5z_filter(0, words, cdv, 40, C.byref(size), C.byref(buf)) == 40  # conf.num < 20: passed through
    conf = sz3_amd.Config(16, 16, 16)
    conf.absErrorBound = 1e-3
    cdv, words = _cd_values(conf._c)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        assert L.sz3hip_h5z_filter(0, words, cdv, data.nbytes, C.byref(size), C.byref(buf)) == 0
        assert b"device" in L.sz3hip_last_error().lower() or b"hip" in L.sz3hip_last_error().lower()
    assert L.sz3hip_h5z_filter(0, 3, cdv, data.nbytes, C.byref(size), C.byref(buf)) == 0  # truncated cd_values: no Config in them
    libc.free(buf)


def test_loading_the_library_does_not_import_torch_unless_asked():
    """a host that never touches torch (CLI, HDF5 filter under h5py, CPU tools) must not pay torch's import through
    sz3_amd.lib(); SZ3HIP_TORCH_PRELOAD=1 asks for it (the order that matters on a GPU box: torch before the library's first call)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, %r); import sz3_amd; sz3_amd.lib(); print('torch' in sys.modules)" % root
    env = {k: v for k, v in os.environ.items() if k != "SZ3HIP_TORCH_PRELOAD"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == "False", (out.stdout, out.stderr)
    env["SZ3HIP_TORCH_PRELOAD"] = "1"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == "True", (out.stdout, out.stderr)
