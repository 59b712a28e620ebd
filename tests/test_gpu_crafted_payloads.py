"""GPU tests (-m gpu): a device payload whose header was tampered with is refused by sz3hip_decompress_device (SZ3HIP_EFORMAT /
ECAPACITY), never decoded — extents that do not multiply to n (also through 64-bit wrap-around), block-predictor fields outside
what the kernels are built for, list lengths beyond the array. Header layout: sz3_amd/csrc/sz3hip_format.h (tests/szh_ref.parse)."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import sz3_amd  # noqa: E402
import szh_ref  # noqa: E402
from fields import field1d, field3d  # noqa: E402
from test_gpu_regression import NO_EXIT  # noqa: E402


def _device_payload(a, regression):
    import torch
    dev = torch.device("cuda:0")
    conf = sz3_amd.Config(*a.shape)
    conf.cmprAlgo = sz3_amd.ALGO_LORENZO_REG
    conf.lorenzo, conf.lorenzo2, conf.regression = 1, 0, int(regression)
    conf.absErrorBound = 1e-3
    dc = sz3_amd.DeviceCompressor(a.size, a.dtype.type, device=0)
    cap = max(dc.payload_bound(a.size), dc.payload_bound_conf(conf))
    d_in = torch.from_numpy(a).to(dev)
    d_pl = torch.empty(cap, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    dc.stage1(conf, d_in.data_ptr(), st)
    dc.stage2(d_pl.data_ptr(), cap, st)
    size = dc.finish(st)
    return dc, bytearray(d_pl[:size].cpu().numpy().tobytes()), d_in


def _try(dc, blob, n, dtype):
    import torch
    dev = torch.device("cuda:0")
    d_pl = torch.from_numpy(np.frombuffer(bytes(blob), dtype=np.uint8).copy()).to(dev)
    d_out = torch.zeros(n, dtype=torch.float32 if dtype == np.float32 else torch.float64, device=dev)
    dc.decompress(d_pl.data_ptr(), len(blob), d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


@pytest.mark.parametrize("kind", ["plain-3d", "block-1d", "block-3d", "block-4d"])
def test_tampered_headers_are_refused(kind):
    L = sz3_amd.lib()
    try:
        L.sz3hip_debug_flags(NO_EXIT)
        if kind == "block-1d":
            a = field1d(40000, np.float32)
        elif kind == "block-4d":
            from fields import field4d
            a = field4d((6, 14, 20, 22), np.float32)
        else:
            a = field3d((24, 40, 56), np.float32)
        dc, blob, d_in = _device_payload(a, regression=kind != "plain-3d")
    finally:
        L.sz3hip_debug_flags(0)
    n = a.size
    h, o, sec = szh_ref.parse(bytes(blob))
    assert h["predictor"] == (0 if kind == "plain-3d" else 2)
    good = _try(dc, blob, n, np.float32)
    assert float(np.max(np.abs(good.astype(np.float64) - a.reshape(-1).astype(np.float64)))) <= 1e-3
    dims = list(h["dims"])

    def with_(off, fmt, *vals):
        b = bytearray(blob)
        struct.pack_into(fmt, b, off, *vals)
        return b

    cases = {
        "extents do not multiply to n": with_(16, "<4Q", dims[0], dims[1], dims[2], dims[3] + 1),
        "extents wrap around 64 bits": with_(16, "<4Q", (1 << 63) + 1, 2, 1, n // 2),
        "a zero extent": with_(16, "<4Q", 0, dims[1], dims[2], dims[3]),
        "more delta outliers than elements": with_(96, "<Q", n + 1),
        "unknown predictor": with_(11, "<B", 9),
        "wrong format version": with_(4, "<I", 3),
    }
    if kind != "plain-3d":
        cases["block edge below 4"] = with_(144, "<I", 2)
        cases["empty predictor set"] = with_(148, "<I", 0)
        cases["side section shorter than its header"] = with_(120, "<Q", 8)
        # the Rice parameters behind the selection bits are shift counts in the side section's parser: > 63 is refused
        B = h["blk_edge"]
        nblocks = int(np.prod([(d + B - 1) // B for d in (h["dims"] if kind == "block-4d" else h["dims"][1:])]))
        sel_bytes = ((nblocks + 3) // 4 + 7) & ~7
        cases["Rice parameter beyond 63"] = with_(o["side"] + 24 + sel_bytes + 2, "<B", 200)
    if kind == "block-1d":
        cases["a predictor set beyond the three members"] = with_(148, "<I", 9)
        cases["1-D stream with a second extent"] = with_(16, "<4Q", 1, 1, 2, n // 2)
    if kind == "block-3d":
        cases["3-D block edge above 8"] = with_(144, "<I", 16)
    if kind == "block-4d":
        cases["4-D block edge above 6"] = with_(144, "<I", 8)
        cases["second-order Lorenzo in 4-D"] = with_(148, "<I", 7)
        cases["fifth Rice parameter beyond 63"] = with_(o["side"] + 24 + sel_bytes + 4, "<B", 99)
    for what, b in cases.items():
        with pytest.raises(sz3_amd.SZ3HipError):
            _try(dc, b, n, np.float32)
        print("refused:", what, "->", L.sz3hip_last_error().decode()[:90] if hasattr(L, "sz3hip_last_error") else "")
    # and the context still decodes the untouched payload afterwards
    again = _try(dc, blob, n, np.float32)
    assert np.array_equal(again, good)
