"""numpy model of the SZH1 device payload (sz3_amd/csrc/sz3hip_format.h) — TEST INFRASTRUCTURE.

Used by the GPU tests to localise faults stage by stage: expected dual-quantisation codes, payload parsing, a slow
pure-python canonical-Huffman decoder and the N-d prefix-sum reconstruction. Not used by the product.
"""
import struct
import numpy as np

MAGIC = 0x31485A53
CHUNK = 1024
MAX_LEN = 24        # format limit; code books of <= SHORT_SYMS symbols are limited to SHORT_LEN
SHORT_SYMS, SHORT_LEN = 512, 16


def dualquant(a, eb, radius=32768, narrow=False):
    """Expected lattice indices, codes and outliers for array `a` (any ndim <= 4), exactly as K1 computes them.
    narrow: stage 1 kept one-byte codes, i.e. deltas outside [-127, 127] became delta outliers."""
    a = np.ascontiguousarray(a)
    T = a.dtype
    # lattice arithmetic in the data type (sz3hip_kernels.hip, Lattice<T>): one rounding per multiply, no FMA
    if T == np.float32:
        qt = np.int32
        recip = np.float32(1.0 / (2.0 * eb))
        two_eb = np.float32(2.0 * eb)
        eb_lo = np.float32(eb)
        if float(eb_lo) > eb:
            eb_lo = np.nextafter(eb_lo, np.float32(0))
        lim = np.float32(8388608.0)
    else:
        qt = np.int64
        recip = 1.0 / (2.0 * eb)
        two_eb = 2.0 * eb
        eb_lo = eb
        lim = 4503599627370496.0
    with np.errstate(invalid="ignore", over="ignore"):
        s = a * recip
        ok = np.abs(s) < lim
        r = np.rint(np.where(ok, s, 0)).astype(T)
        q = r.astype(qt)
        dec = r * two_eb
        diff = np.abs(dec - a)
        bad = ~ok | ~(diff <= eb_lo)
    # N-d Lorenzo = successive first differences with zero halo, wrap-around integer arithmetic
    d = q.copy()
    for ax in range(a.ndim):
        if a.shape[ax] > 1 or True:
            pad = [(0, 0)] * a.ndim
            pad[ax] = (1, 0)
            p = np.pad(d, pad)
            sl_hi = [slice(None)] * a.ndim
            sl_lo = [slice(None)] * a.ndim
            sl_hi[ax] = slice(1, None)
            sl_lo[ax] = slice(0, -1)
            with np.errstate(over="ignore"):
                d = (p[tuple(sl_hi)] - p[tuple(sl_lo)]).astype(qt)
    inr = ((d >= -127) & (d <= 127)) if narrow else ((d > -radius) & (d < radius))
    codes = np.where(inr, d + radius, 0).astype(np.uint16)
    return q, d, codes, bad, ~inr


def parse(payload):
    b = bytes(payload)
    (magic, version, dtype, ndim, qbytes, predictor, radius) = struct.unpack_from("<IIBBBBI", b, 0)
    dims = struct.unpack_from("<4Q", b, 16)
    eb, n, chunk_syms, max_len, n_chunks, sym_min, sym_count, n_vout, n_dout, words, pbytes = struct.unpack_from(
        "<dQIIQIIQQQQ", b, 48)
    h = dict(magic=magic, version=version, dtype=dtype, ndim=ndim, qbytes=qbytes, radius=radius, dims=dims, eb=eb, n=n,
             chunk_syms=chunk_syms, max_len=max_len, n_chunks=n_chunks, sym_min=sym_min, sym_count=sym_count,
             n_vout=n_vout, n_dout=n_dout, bitstream_words=words, payload_bytes=pbytes, predictor=predictor)
    a16 = lambda x: (x + 15) & ~15
    tsz = 4 if dtype == 0 else 8
    off = 160
    o = {}
    o["lens"] = off
    off = a16(off + sym_count)
    o["chunkwords"] = off
    off = a16(off + 2 * n_chunks)
    o["vout_idx"] = off
    off += 8 * n_vout
    o["vout_val"] = off
    off = a16(off + tsz * n_vout)
    o["dout_idx"] = off
    off += 8 * n_dout
    o["dout_val"] = off
    off = a16(off + qbytes * n_dout)
    o["bitstream"] = off
    o["end"] = off + 4 * words
    T = np.float32 if dtype == 0 else np.float64
    Q = np.int32 if dtype == 0 else np.int64
    buf = np.frombuffer(b, dtype=np.uint8)
    sec = dict(
        lens=buf[o["lens"]:o["lens"] + sym_count].copy(),
        chunkwords=np.frombuffer(b, dtype=np.uint16, count=n_chunks, offset=o["chunkwords"]).copy(),
        vout_idx=np.frombuffer(b, dtype=np.uint64, count=n_vout, offset=o["vout_idx"]).copy(),
        vout_val=np.frombuffer(b, dtype=T, count=n_vout, offset=o["vout_val"]).copy(),
        dout_idx=np.frombuffer(b, dtype=np.uint64, count=n_dout, offset=o["dout_idx"]).copy(),
        dout_val=np.frombuffer(b, dtype=Q, count=n_dout, offset=o["dout_val"]).copy(),
        bitstream=np.frombuffer(b, dtype=np.uint32, count=words, offset=o["bitstream"]).copy(),
    )
    return h, o, sec


def canonical_codes(lens):
    """(code, len) per symbol index from code lengths; canonical order = (len, symbol)."""
    lens = np.asarray(lens, dtype=np.int64)
    cnt = np.bincount(lens, minlength=MAX_LEN + 2)
    cnt[0] = 0
    first = np.zeros(MAX_LEN + 2, dtype=np.int64)
    code = 0
    for l in range(1, MAX_LEN + 1):
        code = (code + (cnt[l - 1] if l > 1 else 0)) << (1 if l > 1 else 0)
        first[l] = code
    nxt = first.copy()
    codes = np.zeros(len(lens), dtype=np.int64)
    for i, l in enumerate(lens):
        if l:
            codes[i] = nxt[l]
            nxt[l] += 1
    return codes, first, cnt


def kraft(lens):
    lens = np.asarray(lens, dtype=np.int64)
    return float(np.sum(2.0 ** (-lens[lens > 0].astype(np.float64))))


def huffman_decode(h, sec):
    """slow reference decoder of the chunked bit-stream -> uint16 codes"""
    n = h["n"]
    out = np.zeros(n, dtype=np.uint16)
    lens = sec["lens"]
    if h["max_len"] == 0:
        out[:] = h["sym_min"]
        return out
    codes, _, _ = canonical_codes(lens)
    table = {}
    for i, l in enumerate(lens):
        if l:
            table[(int(l), int(codes[i]))] = h["sym_min"] + i
    offs = np.concatenate([[0], np.cumsum(sec["chunkwords"].astype(np.int64))])
    bs = sec["bitstream"]
    for c in range(h["n_chunks"]):
        words = bs[offs[c]:offs[c + 1]]
        bits = np.unpackbits(words.astype(">u4").view(np.uint8)) if len(words) else np.zeros(0, np.uint8)
        s0 = c * CHUNK
        ns = min(CHUNK, n - s0)
        pos = 0
        for i in range(ns):
            v = 0
            l = 0
            while True:
                v = (v << 1) | int(bits[pos])
                pos += 1
                l += 1
                sym = table.get((l, v))
                if sym is not None:
                    out[s0 + i] = sym
                    break
                if l > MAX_LEN:
                    raise ValueError("bad code in chunk %d" % c)
    return out


def reconstruct(h, sec, codes):
    T = np.float32 if h["dtype"] == 0 else np.float64
    Q = np.int32 if h["dtype"] == 0 else np.int64
    d = np.where(codes == 0, 0, codes.astype(np.int64) - h["radius"]).astype(Q)
    d[sec["dout_idx"].astype(np.int64)] = sec["dout_val"]
    q = d.reshape(h["dims"])
    for ax in range(4):
        with np.errstate(over="ignore"):
            q = np.cumsum(q, axis=ax, dtype=Q)
    x = (q.astype(T) * T(2.0 * h["eb"])).reshape(-1)
    x[sec["vout_idx"].astype(np.int64)] = sec["vout_val"]
    return x


def decode_payload(payload):
    h, o, sec = parse(payload)
    codes = huffman_decode(h, sec)
    return reconstruct(h, sec, codes), h
