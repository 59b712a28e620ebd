#!/usr/bin/env python3
"""Generates tests/golden/golden.npz from the REFERENCE ITSELF (oracle/_ref/libsz3ref.so = szcompressor/SZ3 v3.3.2
built from /root/reference by `make -C oracle ref`).  Run in the build container only; the GPU box never sees the
reference sources.  Stored per case: the generator spec (shape, dtype, field, config) and what the reference
produced — compressed size, sha256 of the pre-zstd payload and of the whole stream (libzstd 1.4.8), max abs error,
and (small cases) the reference's decompressed output itself.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from fields import field1d, field2d, field3d, field4d  # noqa: E402
from oracle_binding import (ALGO_INTERP, ALGO_INTERP_LORENZO, ALGO_LORENZO_REG, EB_ABS, EB_REL, make_config, oracle,  # noqa: E402
                            ref_compress, ref_decompress)
import ctypes as C

CASES = [
    # name, generator, kwargs for make_config
    ("f32_8x8x128_lorenzo_1e-3", lambda: field3d((8, 8, 128)), dict(abs_eb=1e-3)),
    ("f32_64c_lorenzo_1e-2", lambda: field3d((64, 64, 64)), dict(abs_eb=1e-2)),
    ("f32_64c_lorenzo_1e-3", lambda: field3d((64, 64, 64)), dict(abs_eb=1e-3)),
    ("f32_64c_lorenzo_1e-4", lambda: field3d((64, 64, 64)), dict(abs_eb=1e-4)),
    ("f32_64c_lorenzo_reg_1e-3", lambda: field3d((64, 64, 64)), dict(abs_eb=1e-3, regression=True)),
    ("f32_64c_lorenzo_reg_1e-1", lambda: field3d((64, 64, 64)), dict(abs_eb=1e-1, regression=True)),
    ("f32_64c_lorenzo2_reg_1e-2", lambda: field3d((64, 64, 64)), dict(abs_eb=1e-2, regression=True, lorenzo2=True)),
    ("f64_48x40x36_lorenzo_reg_1e-6", lambda: field3d((48, 40, 36), np.float64, sigma=2e-6), dict(abs_eb=1e-6, regression=True)),
    ("f32_4d_12x20x20x20_rel_1e-3", lambda: field4d((12, 20, 20, 20)), dict(eb_mode=EB_REL, rel_eb=1e-3, regression=True)),
    ("f32_1d_65536_lorenzo_reg_1e-3", lambda: field1d(65536), dict(abs_eb=1e-3, regression=True)),
    ("f32_2d_100x100_lorenzo_1e-2", lambda: field2d((100, 100)), dict(abs_eb=1e-2)),
    # interpolation predictor with explicit parameters (ALGO_INTERP) ...
    ("f32_3d_33x47x50_interp_cubic_1e-3", lambda: field3d((33, 47, 50)), dict(algo=ALGO_INTERP, abs_eb=1e-3, interp_algo=1)),
    ("f32_3d_34x66x36_interp_linear_dir5_1e-2", lambda: field3d((34, 66, 36)), dict(algo=ALGO_INTERP, abs_eb=1e-2, interp_algo=0, interpDirection=5)),
    ("f64_3d_20x30x37_interp_cubic_1e-6", lambda: field3d((20, 30, 37), np.float64, sigma=2e-6), dict(algo=ALGO_INTERP, abs_eb=1e-6, interp_algo=1)),
    ("f32_2d_123x257_interp_cubic_a1.5b3", lambda: field2d((123, 257)), dict(algo=ALGO_INTERP, abs_eb=1e-3, interp_algo=1, interpAlpha=1.5, interpBeta=3.0)),
    ("f32_1d_70001_interp_cubic_1e-3", lambda: field1d(70001), dict(algo=ALGO_INTERP, abs_eb=1e-3, interp_algo=1)),
    # ... and the default algorithm with its sampling auto-tuner (ALGO_INTERP_LORENZO)
    ("f32_3d_96c_tuned_1e-3", lambda: field3d((96, 96, 96)), dict(algo=ALGO_INTERP_LORENZO, abs_eb=1e-3, regression=True)),
    ("f32_3d_70x101x130_tuned_1e-2", lambda: field3d((70, 101, 130)), dict(algo=ALGO_INTERP_LORENZO, abs_eb=1e-2, regression=True)),
    ("f64_3d_80x90x100_tuned_1e-6", lambda: field3d((80, 90, 100), np.float64, sigma=2e-6), dict(algo=ALGO_INTERP_LORENZO, abs_eb=1e-6, regression=True)),
    ("f32_2d_600x700_tuned_1e-3", lambda: field2d((600, 700)), dict(algo=ALGO_INTERP_LORENZO, abs_eb=1e-3, regression=True)),
    ("f32_1d_2^20_tuned_1e-3", lambda: field1d(1 << 20), dict(algo=ALGO_INTERP_LORENZO, abs_eb=1e-3, regression=True)),
    ("f32_4d_12x40x40x40_tuned_1e-3", lambda: field4d((12, 40, 40, 40)), dict(algo=ALGO_INTERP_LORENZO, abs_eb=1e-3, regression=True)),
    ("f32_3d_20x21x22_tuner_skipped_1e-3", lambda: field3d((20, 21, 22)), dict(algo=ALGO_INTERP_LORENZO, abs_eb=1e-3, regression=True)),
]


def case_config(shape, kw):
    kw = dict(kw)
    return make_config(shape, algo=kw.pop("algo", ALGO_LORENZO_REG), **kw)


def payload_sha(blob):
    """sha256 of the pre-zstd buffer (zstd-version independent): header 16 B, [u64 rawLen][zstd frame], trailer."""
    L = oracle()
    import struct
    plen, = struct.unpack_from("<Q", blob.tobytes(), 8)
    pay = np.frombuffer(blob.tobytes()[16:16 + plen], dtype=np.uint8)
    rawlen, = struct.unpack_from("<Q", pay.tobytes(), 0)
    raw = np.empty(rawlen, dtype=np.uint8)
    got = L.szo_zstd_decompress(pay.ctypes.data, pay.size, raw.ctypes.data, rawlen)
    assert got == rawlen
    return hashlib.sha256(raw.tobytes()).hexdigest(), blob.tobytes()[16 + plen:].hex()


def main():
    out = {}
    for name, gen, kw in CASES:
        a = gen()
        conf = case_config(a.shape, kw)
        blob = ref_compress(a, conf)
        dec = ref_decompress(blob, a.dtype, a.shape)
        raw_sha, trailer_hex = payload_sha(blob)
        out[name + "/size"] = np.int64(len(blob))
        out[name + "/sha256_stream_zstd148"] = np.array(hashlib.sha256(blob.tobytes()).hexdigest())
        out[name + "/sha256_prezstd"] = np.array(raw_sha)
        out[name + "/trailer_hex"] = np.array(trailer_hex)
        out[name + "/max_err"] = np.float64(np.max(np.abs(dec.astype(np.float64) - a.astype(np.float64))))
        if a.size <= 70000:
            out[name + "/dec"] = dec
        else:
            out[name + "/dec_sha256"] = np.array(hashlib.sha256(dec.tobytes()).hexdigest())
        print(name, len(blob), "ratio %.3f" % (a.nbytes / len(blob)), "max_err %.3g" % out[name + "/max_err"])
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    print("wrote", os.path.join(HERE, "golden.npz"), os.path.getsize(os.path.join(HERE, "golden.npz")), "bytes")


if __name__ == "__main__":
    main()
