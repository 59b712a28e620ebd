// sz3_amd/csrc/sz3hip_kernels.hip — hand-written CDNA4 (gfx950, wave64) kernels of the SZ3 hot path.
//
//   K0  k_minmax            global min/max for REL/PSNR bounds            (reference: utils/Statistic.hpp:12-21)
//   K1  k_lorenzo_quant     prequantise to the 2*eb lattice, integer N-d Lorenzo from an LDS-staged halo tile,
//                           code emission (u16), outlier capture, LDS-privatised histogram
//                           (reference loops: decomposition/BlockwiseDecomposition.hpp:33-44,
//                            predictor/LorenzoPredictor.hpp:60-95, quantizer/LinearQuantizer.hpp:43-71,
//                            encoder/HuffmanEncoder.hpp:520-524)
//   K5  k_codebook          canonical, length-limited Huffman code from the histogram, one workgroup
//                           (reference: encoder/HuffmanEncoder.hpp:516-561, 478-508)
//   K6  k_chunk_bits / k_scan_chunks / k_encode   two-pass chunked bit-pack  (reference: HuffmanEncoder.hpp:140-218)
//   K8  k_dec_tables / k_decode / k_expand_codes / k_scatter_dout / k_scan_x* / k_scan_strided /
//       k_dequant / k_patch_vout      chunk-parallel Huffman decode, Lorenzo inverse = N-d inclusive prefix sums
//                           (reference: HuffmanEncoder.hpp:225-255, BlockwiseDecomposition.hpp:48-67)
//
// The reference predicts from already *reconstructed* neighbours (a loop-carried dependency through the whole
// array).  Here every value is first snapped to the lattice q = rint(x / 2eb); the Lorenzo stencil then runs on
// exact integers, which makes compression embarrassingly parallel and decompression an N-dimensional prefix sum.
// The reconstruction (T)(q*2eb) is verified against the bound exactly like LinearQuantizer.hpp:57-66 and the raw
// value is kept losslessly when the check fails (NaN/Inf/huge magnitudes).
//
// No MFMA anywhere: the path is integer/byte work bounded by HBM bandwidth.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "sz3hip_format.h"
#include "sz3hip_kernels.h"

#define WAVE 64

// ------------------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

template <typename V>
__device__ __forceinline__ V wave_incl_scan(V v) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        V t = __shfl_up(v, d, WAVE);
        if (lane_id() >= d) v += t;
    }
    return v;
}
template <typename V>
__device__ __forceinline__ V wave_sum(V v) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

template <typename T> struct QTraits;
template <> struct QTraits<float> {
    using Q = int32_t;
    using UQ = uint32_t;
    static constexpr double QMAX = 1073741824.0;  // 2^30
};
template <> struct QTraits<double> {
    using Q = int64_t;
    using UQ = uint64_t;
    static constexpr double QMAX = 4611686018427387904.0;  // 2^62
};

// prequantisation: q = rint(x * 1/(2eb)) in double; reconstruct (T)(q * 2eb); keep the raw value when the
// reconstruction misses the bound (same acceptance test as quantizer/LinearQuantizer.hpp:57-60: |dec-data| in T,
// compared with eb in double). Non-finite or huge values take q = 0 so that neighbours still predict sanely.
template <typename T>
__device__ __forceinline__ typename QTraits<T>::Q prequant(T x, double recip, double two_eb, double eb, bool &bad) {
    using Q = typename QTraits<T>::Q;
    double s = (double)x * recip;
    Q q = 0;
    bad = true;
    if (fabs(s) < QTraits<T>::QMAX) {  // false for NaN
        double r = rint(s);
        q = (Q)r;
        T dec = (T)(r * two_eb);
        T diff = dec - x;
        diff = diff < 0 ? -diff : diff;
        bad = !((double)diff <= eb);
    }
    return q;
}

template <typename T>
__device__ __forceinline__ T dequant(typename QTraits<T>::Q q, double two_eb) {
    return (T)((double)q * two_eb);
}

// ------------------------------------------------------------------------------------------------------------
// K0: min / max
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_minmax(const T *__restrict__ in, uint64_t n, double *partial) {
    double mn = INFINITY, mx = -INFINITY;
    bool has_nan = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        double v = (double)in[i];
        // the reference's comparisons (Statistic.hpp:15-18: "if (max < d) max = d; if (min > d) min = d") skip NaN
        if (v < mn) mn = v;
        if (v > mx) mx = v;
        has_nan |= (v != v);
    }
    (void)has_nan;
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, d, WAVE));
        mx = fmax(mx, __shfl_xor(mx, d, WAVE));
    }
    __shared__ double smn[4], smx[4];
    int w = threadIdx.x / WAVE;
    if (lane_id() == 0) {
        smn[w] = mn;
        smx[w] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) {
            mn = fmin(mn, smn[i]);
            mx = fmax(mx, smx[i]);
        }
        partial[2 * blockIdx.x] = mn;
        partial[2 * blockIdx.x + 1] = mx;
    }
}
__global__ void k_minmax_final(const double *partial, int nblocks, double *out) {
    double mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nblocks; i += WAVE) {
        mn = fmin(mn, partial[2 * i]);
        mx = fmax(mx, partial[2 * i + 1]);
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, d, WAVE));
        mx = fmax(mx, __shfl_xor(mx, d, WAVE));
    }
    if (threadIdx.x == 0) {
        out[0] = mn;
        out[1] = mx;
    }
}

// ------------------------------------------------------------------------------------------------------------
// K1 (generic N = 1..4): LDS halo tile of prequantised integers, then the separable Lorenzo stencil.
// Tile = TW(=1) x TZ x TY x TX owned elements; LDS holds (NW) x (TZ+HZ) x (TY+HY) x (TX+1) lattice indices,
// NW = 2 for N = 4 (hyper-planes w and w-1).  Low-side halo outside the array is 0, exactly the zero padding of
// block_data (utils/BlockwiseIterator.hpp:200-220).
// ------------------------------------------------------------------------------------------------------------
#define HIST_WIN 1024  // LDS histogram window (bins) centred on the radius
#define HIST_COPIES 4

template <typename T, int NDIM, int TX, int TY, int TZ>
__global__ __launch_bounds__(256) void k_lorenzo_quant(const T *__restrict__ in, uint16_t *__restrict__ codes,
                                                       szk_k1_params p) {
    using Q = typename QTraits<T>::Q;
    constexpr int HY = NDIM >= 2 ? 1 : 0, HZ = NDIM >= 3 ? 1 : 0, NW = NDIM >= 4 ? 2 : 1;
    constexpr int PX = TX + 1, PY = TY + HY, PZ = TZ + HZ;
    constexpr int CELLS = NW * PZ * PY * PX;
    __shared__ Q lq[CELLS];
    __shared__ uint32_t lh[HIST_COPIES * HIST_WIN];

    const uint64_t d0 = p.d[3], d1 = p.d[2], d2 = p.d[1], d3 = p.d[0];  // x, y, z, w extents
    const uint32_t ntx = (uint32_t)((d0 + TX - 1) / TX), nty = (uint32_t)((d1 + TY - 1) / TY),
                   ntz = (uint32_t)((d2 + TZ - 1) / TZ);
    uint64_t b = blockIdx.x;
    const uint64_t tx = b % ntx;
    b /= ntx;
    const uint64_t ty = b % nty;
    b /= nty;
    const uint64_t tz = b % ntz;
    const uint64_t w = b / ntz;
    const int64_t x0 = (int64_t)(tx * TX), y0 = (int64_t)(ty * TY), z0 = (int64_t)(tz * TZ);

    for (int i = threadIdx.x; i < HIST_COPIES * HIST_WIN; i += 256) lh[i] = 0;

    // ---- load + prequantise (tile + low-side halo) ----
    for (int c = threadIdx.x; c < CELLS; c += 256) {
        int lx = c % PX;
        int r = c / PX;
        int ly = r % PY;
        r /= PY;
        int lz = r % PZ;
        int lw = r / PZ;
        int64_t gx = x0 + lx - 1, gy = y0 + ly - HY, gz = z0 + lz - HZ, gw = (int64_t)w - lw;
        Q q = 0;
        if (gx >= 0 && gy >= 0 && gz >= 0 && gw >= 0 && gx < (int64_t)d0 && gy < (int64_t)d1 && gz < (int64_t)d2) {
            uint64_t gi = (((uint64_t)gw * d2 + (uint64_t)gz) * d1 + (uint64_t)gy) * d0 + (uint64_t)gx;
            T x = in[gi];
            bool bad;
            q = prequant<T>(x, p.recip, p.two_eb, p.eb, bad);
            bool owned = (lx >= 1) && (ly >= HY) && (lz >= HZ) && (lw == 0);
            if (bad && owned) {
                unsigned long long pos = atomicAdd((unsigned long long *)p.n_vout, 1ull);
                if (pos < p.out_cap) {
                    p.vout_idx[pos] = gi;
                    ((T *)p.vout_val)[pos] = x;
                }
            }
        }
        lq[c] = q;
    }
    __syncthreads();

    // ---- integer Lorenzo + code emission ----
    const int radius = (int)p.radius;
    const int win_lo = radius - HIST_WIN / 2;
    uint32_t center_count = 0;
    uint32_t *myh = lh + (threadIdx.x & (HIST_COPIES - 1)) * HIST_WIN;
    constexpr int OWNED = TZ * TY * TX;
    for (int j = threadIdx.x; j < OWNED; j += 256) {
        int lx = j % TX;
        int r = j / TX;
        int ly = r % TY;
        int lz = r / TY;
        uint64_t gx = (uint64_t)x0 + lx, gy = (uint64_t)y0 + ly, gz = (uint64_t)z0 + lz;
        bool inb = gx < d0 && gy < d1 && gz < d2;
        int code = -1;
        if (inb) {
            using UQ = typename QTraits<T>::UQ;
            UQ delta = 0;
#pragma unroll
            for (int lw = 0; lw < NW; lw++) {
                const Q *base = lq + ((lw * PZ + (lz + HZ)) * PY + (ly + HY)) * PX + (lx + 1);
                UQ s = (UQ)base[0] - (UQ)base[-1];
                if (NDIM >= 2) s += (UQ)base[-PX - 1] - (UQ)base[-PX];
                if (NDIM >= 3) {
                    const Q *bz = base - PY * PX;
                    s += (UQ)bz[-1] - (UQ)bz[0];
                    s += (UQ)bz[-PX] - (UQ)bz[-PX - 1];
                }
                delta = lw == 0 ? s : (UQ)(delta - s);
            }
            Q sd = (Q)delta;
            uint64_t gi = ((w * d2 + gz) * d1 + gy) * d0 + gx;
            if (sd > -(Q)radius && sd < (Q)radius) {
                code = (int)sd + radius;
            } else {
                code = 0;
                unsigned long long pos = atomicAdd((unsigned long long *)p.n_dout, 1ull);
                if (pos < p.out_cap) {
                    p.dout_idx[pos] = gi;
                    ((Q *)p.dout_val)[pos] = sd;
                }
            }
            codes[gi] = (uint16_t)code;
        }
        // histogram: the centre bin is counted with one ballot per wave, the rest goes to the LDS window
        unsigned long long mc = __ballot(code == radius);
        if (lane_id() == 0) center_count += (uint32_t)__popcll(mc);
        if (code >= 0 && code != radius) {
            int bin = code - win_lo;
            if (bin >= 0 && bin < HIST_WIN) atomicAdd(&myh[bin], 1u);
            else atomicAdd((unsigned long long *)&p.hist[code], 1ull);
        }
    }
    if (lane_id() == 0 && center_count) atomicAdd(&lh[radius - win_lo], center_count);
    __syncthreads();
    for (int bnn = threadIdx.x; bnn < HIST_WIN; bnn += 256) {
        uint32_t s = lh[bnn] + lh[HIST_WIN + bnn] + lh[2 * HIST_WIN + bnn] + lh[3 * HIST_WIN + bnn];
        int sym = win_lo + bnn;
        if (s && sym >= 0 && sym < (int)SZH_HIST_BINS) atomicAdd((unsigned long long *)&p.hist[sym], (unsigned long long)s);
    }
}

// ------------------------------------------------------------------------------------------------------------
// K5: canonical length-limited Huffman codebook, one 1024-thread workgroup.
// ------------------------------------------------------------------------------------------------------------
#define CB_THREADS 1024
#define CB_LDS_SYMS 2048

__device__ void cb_bitonic_sort(uint64_t *keys, uint32_t npow2) {  // ascending; keys in LDS or global
    for (uint32_t k = 2; k <= npow2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < npow2; i += CB_THREADS) {
                uint32_t ixj = i ^ j;
                if (ixj > i) {
                    uint64_t a = keys[i], bb = keys[ixj];
                    bool up = (i & k) == 0;
                    if ((a > bb) == up) {
                        keys[i] = bb;
                        keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// keys[] (global scratch, >= 65536 + padding), work arrays in global scratch; small alphabets are staged in LDS.
__global__ __launch_bounds__(CB_THREADS) void k_codebook(const uint64_t *__restrict__ hist, szk_cb_params p) {
    __shared__ uint64_t s_keys[CB_LDS_SYMS];
    __shared__ uint64_t s_ifreq[CB_LDS_SYMS];
    __shared__ uint16_t s_pleaf[CB_LDS_SYMS], s_pint[CB_LDS_SYMS];
    __shared__ uint16_t s_depth[CB_LDS_SYMS];
    __shared__ uint32_t s_scan[CB_THREADS];
    __shared__ uint32_t s_m, s_symmin, s_symmax;
    __shared__ uint32_t s_nextcode[SZH_MAX_LEN + 2], s_cnt[SZH_MAX_LEN + 2];

    const uint32_t nbins = SZH_HIST_BINS;
    const uint32_t per = nbins / CB_THREADS;  // 64 consecutive bins per thread
    // clear encode table and lens
    for (uint32_t i = threadIdx.x; i < nbins; i += CB_THREADS) {
        p.enc[i] = 0;
        p.lens[i] = 0;
    }
    // 1. compaction of non-zero bins, in symbol order
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < per; i++) cnt += hist[threadIdx.x * per + i] != 0;
    s_scan[threadIdx.x] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < CB_THREADS; i++) {
            uint32_t c = s_scan[i];
            s_scan[i] = run;
            run += c;
        }
        s_m = run;
        s_symmin = 0xFFFFFFFFu;
        s_symmax = 0;
    }
    __syncthreads();
    const uint32_t m = s_m;
    {
        uint32_t pos = s_scan[threadIdx.x];
        for (uint32_t i = 0; i < per; i++) {
            uint32_t sym = threadIdx.x * per + i;
            uint64_t f = hist[sym];
            if (f) {
                p.keys[pos] = (f << 16) | sym;  // freq < 2^48
                p.syms[pos] = (uint16_t)sym;    // symbol order copy
                pos++;
                atomicMin(&s_symmin, sym);
                atomicMax(&s_symmax, sym);
            }
        }
    }
    __syncthreads();
    if (m == 0) {
        if (threadIdx.x == 0) {
            p.info->n_symbols = 0;
            p.info->max_len = 0;
            p.info->sym_min = 0;
            p.info->sym_count = 0;
        }
        return;
    }
    // 2. sort by (freq, sym)
    uint32_t npow2 = 1;
    while (npow2 < m) npow2 <<= 1;
    const bool small = m <= CB_LDS_SYMS;
    uint64_t *keys = small ? s_keys : p.keys;
    __threadfence_block();
    if (small) {
        for (uint32_t i = threadIdx.x; i < npow2; i += CB_THREADS) s_keys[i] = i < m ? p.keys[i] : ~0ull;
    } else {
        for (uint32_t i = m + threadIdx.x; i < npow2; i += CB_THREADS) p.keys[i] = ~0ull;
    }
    __syncthreads();
    cb_bitonic_sort(keys, npow2);

    // 3. two-queue Huffman merge (thread 0), lengths, limit, canonical codes
    uint64_t *ifreq = small ? s_ifreq : p.ifreq;
    uint16_t *pleaf = small ? s_pleaf : p.pleaf;
    uint16_t *pint = small ? s_pint : p.pint;
    uint16_t *depth = small ? s_depth : p.depth;
    if (threadIdx.x == 0) {
        uint32_t max_len = 0;
        if (m == 1) {
            // single symbol: zero-length code, empty bit-stream (as encoder/HuffmanEncoder.hpp:233-237)
            p.lens[(uint32_t)(keys[0] & 0xFFFF)] = 0;
        } else {
            uint32_t i = 0, j = 0, k = 0;  // leaf cursor, internal cursor, internal count
            for (; k < m - 1; k++) {
                uint64_t f = 0;
                for (int t = 0; t < 2; t++) {
                    bool take_leaf;
                    if (i >= m) take_leaf = false;
                    else if (j >= k) take_leaf = true;
                    else take_leaf = (keys[i] >> 16) <= ifreq[j];
                    if (take_leaf) {
                        f += keys[i] >> 16;
                        pleaf[i++] = (uint16_t)k;
                    } else {
                        f += ifreq[j];
                        pint[j++] = (uint16_t)k;
                    }
                }
                ifreq[k] = f;
            }
            // depths: root = internal m-2
            depth[m - 2] = 0;
            for (int32_t q = (int32_t)m - 3; q >= 0; q--) depth[q] = (uint16_t)(depth[pint[q]] + 1);  // depth <= m-2 < 65536
            for (uint32_t q = 0; q < SZH_MAX_LEN + 2; q++) s_cnt[q] = 0;
            bool over = false;
            for (uint32_t q = 0; q < m; q++) {
                uint32_t l = (uint32_t)depth[pleaf[q]] + 1;
                if (l > SZH_MAX_LEN) {
                    l = SZH_MAX_LEN;
                    over = true;
                }
                // reuse pleaf[] as the per-leaf length store (sorted position q)
                pleaf[q] = (uint16_t)l;
                s_cnt[l]++;
            }
            if (over) {
                // Kraft repair: sum 2^(MAX-len) must be <= 2^MAX.  Lengthen the longest codes shorter than MAX
                // (cheapest in expected bits: leaves are sorted by ascending frequency).
                uint64_t kraft = 0;
                for (uint32_t l = 1; l <= SZH_MAX_LEN; l++) kraft += (uint64_t)s_cnt[l] << (SZH_MAX_LEN - l);
                const uint64_t budget = 1ull << SZH_MAX_LEN;
                while (kraft > budget) {
                    // pick the least frequent leaf with len < MAX and the largest such len
                    int best = -1;
                    uint32_t bestl = 0;
                    for (uint32_t q = 0; q < m; q++) {
                        uint32_t l = pleaf[q];
                        if (l < SZH_MAX_LEN && l > bestl) {
                            bestl = l;
                            best = (int)q;
                        }
                    }
                    if (best < 0) break;
                    pleaf[best] = (uint16_t)(bestl + 1);
                    s_cnt[bestl]--;
                    s_cnt[bestl + 1]++;
                    kraft -= 1ull << (SZH_MAX_LEN - bestl - 1);
                }
            }
            for (uint32_t q = 0; q < m; q++) {
                uint32_t sym = (uint32_t)(keys[q] & 0xFFFF);
                uint32_t l = pleaf[q];
                p.lens[sym] = (uint8_t)l;
                if (l > max_len) max_len = l;
            }
            // canonical first codes
            uint32_t code = 0;
            for (uint32_t l = 1; l <= SZH_MAX_LEN; l++) {
                code = (code + (l > 1 ? s_cnt[l - 1] : 0)) << (l > 1 ? 1 : 0);
                s_nextcode[l] = code;
            }
            __threadfence_block();
            // assign in symbol order (syms[] is the compacted alphabet in increasing symbol order)
            for (uint32_t q = 0; q < m; q++) {
                uint32_t sym = p.syms[q];
                uint32_t l = p.lens[sym];
                uint32_t c = s_nextcode[l]++;
                p.enc[sym] = (c << 5) | l;
            }
        }
        p.info->n_symbols = m;
        p.info->max_len = max_len;
        p.info->sym_min = s_symmin;
        p.info->sym_count = s_symmax - s_symmin + 1;
    }
}

// ------------------------------------------------------------------------------------------------------------
// payload layout: header + section offsets, computed on the device once the outlier counts and the alphabet
// range are known (no host round trip between stage 1 and the encoder).
// ------------------------------------------------------------------------------------------------------------
__device__ __host__ inline void szh_compute_offsets(const szh_header &h, szh_offsets &o) {
    uint64_t tsz = h.dtype == 0 ? 4 : 8;
    uint64_t off = sizeof(szh_header);
    o.lens = off;
    off = szh_align16(off + h.sym_count);
    o.chunkwords = off;
    off = szh_align16(off + 2 * h.n_chunks);
    o.vout_idx = off;
    off += 8 * h.n_vout;
    o.vout_val = off;
    off = szh_align16(off + tsz * h.n_vout);
    o.dout_idx = off;
    off += 8 * h.n_dout;
    o.dout_val = off;
    off = szh_align16(off + (uint64_t)h.qbytes * h.n_dout);
    o.bitstream = off;
    o.end = off + 4 * h.bitstream_words;
}

__global__ void k_layout_pre(szk_layout_params p) {  // after K1 + K5, before the encoder
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    szh_header h = p.proto;  // dtype, ndim, dims, eb, radius, n, chunk geometry filled by the host
    uint64_t nv = *p.n_vout, nd = *p.n_dout;
    p.state->overflow = (nv > p.out_cap) || (nd > p.out_cap);
    if (nv > p.out_cap) nv = p.out_cap;
    if (nd > p.out_cap) nd = p.out_cap;
    h.n_vout = nv;
    h.n_dout = nd;
    h.sym_min = p.info->sym_min;
    h.sym_count = p.info->sym_count;
    h.max_len = p.info->max_len;
    h.bitstream_words = 0;
    szh_offsets o;
    szh_compute_offsets(h, o);
    p.state->hdr = h;
    p.state->off = o;
}

// ------------------------------------------------------------------------------------------------------------
// K6 pass 1: 32-bit words needed by every chunk of SZH_CHUNK_SYMS symbols (one wave per chunk, 16 symbols/lane)
// ------------------------------------------------------------------------------------------------------------
#define ENC_PER_LANE 16
#define ENC_WIN 4096  // LDS-cached slice of the encode table around the radius

__device__ __forceinline__ uint32_t enc_lookup(const uint32_t *s_enc, const uint32_t *__restrict__ g_enc, int win_lo,
                                               uint32_t sym) {
    int rel = (int)sym - win_lo;
    return (rel >= 0 && rel < ENC_WIN) ? s_enc[rel] : g_enc[sym];
}

__device__ __forceinline__ void load_codes16(const uint16_t *__restrict__ codes, uint64_t base, uint64_t n,
                                             uint16_t (&c)[ENC_PER_LANE]) {
    if (base + ENC_PER_LANE <= n) {
        const uint4 *v = reinterpret_cast<const uint4 *>(codes + base);  // base is a multiple of 16 -> 32-byte aligned
        uint4 a = v[0], b = v[1];
        uint32_t wds[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c[2 * i] = (uint16_t)(wds[i] & 0xFFFF);
            c[2 * i + 1] = (uint16_t)(wds[i] >> 16);
        }
    } else {
#pragma unroll
        for (int i = 0; i < ENC_PER_LANE; i++) c[i] = (base + i < n) ? codes[base + i] : (uint16_t)0xFFFF;
    }
}

__global__ __launch_bounds__(256) void k_chunk_bits(const uint16_t *__restrict__ codes, uint64_t n,
                                                    const uint32_t *__restrict__ g_enc, int radius,
                                                    uint16_t *__restrict__ chunk_words) {
    __shared__ uint32_t s_enc[ENC_WIN];
    const int win_lo = radius - ENC_WIN / 2;
    for (int i = threadIdx.x; i < ENC_WIN; i += 256) {
        int sym = win_lo + i;
        s_enc[i] = (sym >= 0 && sym < (int)SZH_HIST_BINS) ? g_enc[sym] : 0;
    }
    __syncthreads();
    const uint64_t chunk = (uint64_t)blockIdx.x * 4 + threadIdx.x / WAVE;
    const uint64_t base = chunk * SZH_CHUNK_SYMS + (uint64_t)lane_id() * ENC_PER_LANE;
    if (chunk * SZH_CHUNK_SYMS >= n) return;
    uint16_t c[ENC_PER_LANE];
    load_codes16(codes, base, n, c);
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < ENC_PER_LANE; i++)
        if (base + i < n) bits += enc_lookup(s_enc, g_enc, win_lo, c[i]) & 31u;
    bits = wave_sum(bits);
    if (lane_id() == 0) chunk_words[chunk] = (uint16_t)((bits + 31) >> 5);
}

// exclusive scan of the chunk word counts (one workgroup; n_chunks is n/1024)
__global__ __launch_bounds__(1024) void k_scan_chunks(const uint16_t *__restrict__ chunk_words, uint64_t n_chunks,
                                                      uint64_t *__restrict__ chunk_off, uint64_t *total_words) {
    __shared__ uint64_t s_part[1024];
    const uint64_t per = (n_chunks + 1023) / 1024;
    const uint64_t lo = (uint64_t)threadIdx.x * per, hi = (lo + per < n_chunks) ? lo + per : n_chunks;
    uint64_t s = 0;
    for (uint64_t i = lo; i < hi; i++) s += chunk_words[i];
    s_part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; i++) {
            uint64_t v = s_part[i];
            s_part[i] = run;
            run += v;
        }
        *total_words = run;
    }
    __syncthreads();
    uint64_t run = s_part[threadIdx.x];
    for (uint64_t i = lo; i < hi; i++) {
        chunk_off[i] = run;
        run += chunk_words[i];
    }
}

// K6 pass 2: bit-pack. One wave per chunk; every lane packs its 16 code words at its bit offset inside the
// chunk (wave prefix sum) into an LDS staging area with ds_or, then the wave streams the words out coalesced.
__global__ __launch_bounds__(256) void k_encode(const uint16_t *__restrict__ codes, uint64_t n,
                                                const uint32_t *__restrict__ g_enc, int radius,
                                                const uint64_t *__restrict__ chunk_off,
                                                const szk_state *__restrict__ state, uint8_t *__restrict__ payload) {
    constexpr int STAGE_WORDS = SZH_CHUNK_SYMS * SZH_MAX_LEN / 32;  // 768 words per chunk at most
    __shared__ uint32_t s_enc[ENC_WIN];
    __shared__ uint32_t s_stage[4][STAGE_WORDS];
    const int win_lo = radius - ENC_WIN / 2;
    for (int i = threadIdx.x; i < ENC_WIN; i += 256) {
        int sym = win_lo + i;
        s_enc[i] = (sym >= 0 && sym < (int)SZH_HIST_BINS) ? g_enc[sym] : 0;
    }
    const int wv = threadIdx.x / WAVE;
    uint32_t *stage = s_stage[wv];
    for (int i = lane_id(); i < STAGE_WORDS; i += WAVE) stage[i] = 0;
    __syncthreads();
    const uint64_t chunk = (uint64_t)blockIdx.x * 4 + wv;
    if (chunk * SZH_CHUNK_SYMS >= n) return;
    const uint64_t base = chunk * SZH_CHUNK_SYMS + (uint64_t)lane_id() * ENC_PER_LANE;
    uint16_t c[ENC_PER_LANE];
    load_codes16(codes, base, n, c);
    uint32_t e[ENC_PER_LANE];
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < ENC_PER_LANE; i++) {
        e[i] = (base + i < n) ? enc_lookup(s_enc, g_enc, win_lo, c[i]) : 0u;
        bits += e[i] & 31u;
    }
    uint32_t incl = wave_incl_scan(bits);
    uint32_t total_bits = __shfl(incl, WAVE - 1, WAVE);
    uint32_t pos = incl - bits;  // bit offset of this lane inside the chunk
    // pack: 64-bit accumulator, MSB-first within 32-bit words
    uint32_t word = pos >> 5;
    uint32_t fill = pos & 31;  // bits already occupied in the current word (by the previous lane)
    uint64_t acc = 0;          // pending bits, left-aligned at bit (63 - fill)
    uint32_t have = fill;      // number of valid bit positions consumed in acc (including the skipped prefix)
#pragma unroll
    for (int i = 0; i < ENC_PER_LANE; i++) {
        uint32_t len = e[i] & 31u;
        uint64_t cw = e[i] >> 5;
        if (len) {
            acc |= cw << (64 - have - len);
            have += len;
            if (have >= 32) {
                atomicOr(&stage[word], (uint32_t)(acc >> 32));
                word++;
                acc <<= 32;
                have -= 32;
            }
        }
    }
    if (have > 0 && (uint32_t)(acc >> 32) != 0) atomicOr(&stage[word], (uint32_t)(acc >> 32));
    // (a lane whose bits end exactly on a word boundary has have == 0; a lane with no bits writes nothing)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const uint32_t nwords = (total_bits + 31) >> 5;
    uint32_t *out = reinterpret_cast<uint32_t *>(payload + state->off.bitstream) + chunk_off[chunk];
    for (uint32_t i = lane_id(); i < nwords; i += WAVE) out[i] = stage[i];
}

// header + side sections (lens, chunk table, outliers) into the payload
__global__ __launch_bounds__(256) void k_assemble(szk_asm_params p) {
    const szh_header h0 = p.state->hdr;
    const szh_offsets o = p.state->off;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    if (tid == 0) {
        szh_header h = h0;
        h.bitstream_words = *p.total_words;
        szh_offsets oo;
        szh_compute_offsets(h, oo);
        h.payload_bytes = oo.end;
        *reinterpret_cast<szh_header *>(p.payload) = h;
        p.state->hdr = h;
        p.state->off = oo;
        p.state->cap_exceeded = oo.end > p.cap;
    }
    for (uint64_t i = tid; i < h0.sym_count; i += nth) p.payload[o.lens + i] = p.lens[h0.sym_min + i];
    uint16_t *cw = reinterpret_cast<uint16_t *>(p.payload + o.chunkwords);
    for (uint64_t i = tid; i < h0.n_chunks; i += nth) cw[i] = p.chunk_words[i];
    uint64_t *vi = reinterpret_cast<uint64_t *>(p.payload + o.vout_idx);
    uint64_t *di = reinterpret_cast<uint64_t *>(p.payload + o.dout_idx);
    for (uint64_t i = tid; i < h0.n_vout; i += nth) vi[i] = p.vout_idx[i];
    for (uint64_t i = tid; i < h0.n_dout; i += nth) di[i] = p.dout_idx[i];
    const uint64_t tsz = h0.dtype == 0 ? 4 : 8, qsz = h0.qbytes;
    for (uint64_t i = tid; i < h0.n_vout * tsz; i += nth) p.payload[o.vout_val + i] = ((const uint8_t *)p.vout_val)[i];
    for (uint64_t i = tid; i < h0.n_dout * qsz; i += nth) p.payload[o.dout_val + i] = ((const uint8_t *)p.dout_val)[i];
}

// ------------------------------------------------------------------------------------------------------------
// K8: decode side
// ------------------------------------------------------------------------------------------------------------
// canonical decode tables from the code lengths: first code / first rank per length and the symbols sorted by
// (len, sym).  One workgroup; the alphabet is at most 65536 symbols.
__global__ __launch_bounds__(1024) void k_dec_tables(const uint8_t *__restrict__ lens, uint32_t sym_min,
                                                     uint32_t sym_count, szk_dec_tables *t) {
    __shared__ uint32_t s_cnt[SZH_MAX_LEN + 2];
    __shared__ uint32_t s_first_rank[SZH_MAX_LEN + 2];
    if (threadIdx.x < SZH_MAX_LEN + 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < sym_count; i += 1024) {
        uint32_t l = lens[i];
        if (l) atomicAdd(&s_cnt[l], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t code = 0, rank = 0, maxl = 0, single = 0xFFFFFFFFu;
        for (uint32_t l = 1; l <= SZH_MAX_LEN; l++) {
            code = (code + (l > 1 ? s_cnt[l - 1] : 0)) << (l > 1 ? 1 : 0);
            t->first_code[l] = code;
            t->first_rank[l] = rank;
            t->count[l] = s_cnt[l];
            s_first_rank[l] = rank;
            rank += s_cnt[l];
            if (s_cnt[l]) maxl = l;
        }
        t->max_len = maxl;
        t->n_coded = rank;
        // serial placement keeps (len, sym) order; alphabets are small in practice
        for (uint32_t i = 0; i < sym_count; i++) {
            uint32_t l = lens[i];
            if (l) t->sorted_syms[s_first_rank[l]++] = (uint16_t)(sym_min + i);
        }
        (void)single;
    }
}

// one thread per chunk: canonical decode by length search on a 64-bit MSB-aligned window
__global__ __launch_bounds__(256) void k_decode(const uint8_t *__restrict__ payload, szk_dec_params p,
                                                uint16_t *__restrict__ codes) {
    __shared__ uint32_t s_first_code[SZH_MAX_LEN + 2], s_first_rank[SZH_MAX_LEN + 2], s_count[SZH_MAX_LEN + 2];
    if (threadIdx.x <= SZH_MAX_LEN) {
        s_first_code[threadIdx.x] = threadIdx.x ? p.tables->first_code[threadIdx.x] : 0;
        s_first_rank[threadIdx.x] = threadIdx.x ? p.tables->first_rank[threadIdx.x] : 0;
        s_count[threadIdx.x] = threadIdx.x ? p.tables->count[threadIdx.x] : 0;
    }
    __syncthreads();
    const uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= p.n_chunks) return;
    const uint64_t s0 = chunk * SZH_CHUNK_SYMS;
    const uint32_t nsym = (uint32_t)((p.n - s0 < SZH_CHUNK_SYMS) ? (p.n - s0) : SZH_CHUNK_SYMS);
    uint16_t *out = codes + s0;
    const uint32_t max_len = p.tables->max_len;
    if (max_len == 0) {  // single-symbol alphabet: zero-length code
        uint16_t sym = (uint16_t)p.single_sym;
        for (uint32_t i = 0; i < nsym; i++) out[i] = sym;
        return;
    }
    const uint32_t *in = reinterpret_cast<const uint32_t *>(payload + p.bitstream_off) + p.chunk_off[chunk];
    const uint16_t *sorted = p.tables->sorted_syms;
    uint64_t buf = 0;  // next bits at the MSB end
    int have = 0;
    uint32_t wi = 0;
    const uint32_t nwords = p.chunk_words[chunk];
    for (uint32_t i0 = 0; i0 < nsym; i0 += 16) {
        uint32_t packed[8];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint32_t sym = 0;
            if (i0 + k < nsym) {
                if (have <= 32) {
                    uint32_t wd = wi < nwords ? in[wi] : 0u;
                    wi++;
                    buf |= (uint64_t)wd << (32 - have);
                    have += 32;
                }
                uint32_t l;
                for (l = 1; l <= max_len; l++) {
                    uint32_t v = (uint32_t)(buf >> (64 - l));
                    uint32_t rel = v - s_first_code[l];
                    if (rel < s_count[l]) {  // unsigned compare also rejects v < first_code
                        sym = sorted[s_first_rank[l] + rel];
                        break;
                    }
                }
                buf <<= l;
                have -= (int)l;
            }
            if (k & 1) packed[k >> 1] |= sym << 16;
            else packed[k >> 1] = sym;
        }
        if (i0 + 16 <= nsym) {
            uint4 *o4 = reinterpret_cast<uint4 *>(out + i0);
            o4[0] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
            o4[1] = make_uint4(packed[4], packed[5], packed[6], packed[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (i0 + k < nsym) out[i0 + k] = (uint16_t)((k & 1) ? (packed[k >> 1] >> 16) : (packed[k >> 1] & 0xFFFF));
        }
    }
}

// codes -> integer deltas (code 0 -> 0, patched by k_scatter_dout)
template <typename Q>
__global__ __launch_bounds__(256) void k_expand_codes(const uint16_t *__restrict__ codes, uint64_t n, int radius,
                                                      Q *__restrict__ q) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        int c = codes[i];
        q[i] = c ? (Q)(c - radius) : (Q)0;
    }
}
template <typename Q>
__global__ __launch_bounds__(256) void k_scatter_dout(const uint8_t *__restrict__ payload, uint64_t idx_off,
                                                      uint64_t val_off, uint64_t cnt, uint64_t n, Q *__restrict__ q) {
    const uint64_t *idx = reinterpret_cast<const uint64_t *>(payload + idx_off);
    const Q *val = reinterpret_cast<const Q *>(payload + val_off);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t k = idx[i];
        if (k < n) q[k] = val[i];
    }
}

// inclusive scan along the contiguous axis. Rows of length L are cut into segments of SEG elements; one
// workgroup scans one segment (256 threads x 4 elements per step with a running carry) and writes the segment
// total; k_scan_x_fix then adds the exclusive prefix of the preceding segments of the same row.
#define SCANX_SEG 16384
template <typename Q>
__global__ __launch_bounds__(256) void k_scan_x(Q *__restrict__ q, uint64_t L, uint64_t nrows, uint64_t segs_per_row,
                                                Q *__restrict__ seg_tot) {
    using UQ = typename std::make_unsigned<Q>::type;
    __shared__ UQ s_w[4];
    __shared__ UQ s_carry;
    const uint64_t row = blockIdx.x / segs_per_row, seg = blockIdx.x % segs_per_row;
    const uint64_t lo = seg * SCANX_SEG, hi = (lo + SCANX_SEG < L) ? lo + SCANX_SEG : L;
    Q *r = q + row * L;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t b = lo; b < hi; b += 1024) {
        uint64_t i0 = b + (uint64_t)threadIdx.x * 4;
        UQ v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (i0 + k < hi) ? (UQ)r[i0 + k] : (UQ)0;
        v[1] += v[0];
        v[2] += v[1];
        v[3] += v[2];
        UQ incl = wave_incl_scan(v[3]);
        if (lane_id() == WAVE - 1) s_w[threadIdx.x / WAVE] = incl;
        __syncthreads();
        UQ off = s_carry + (incl - v[3]);
        for (int w = 0; w < (int)(threadIdx.x / WAVE); w++) off += s_w[w];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i0 + k < hi) r[i0 + k] = (Q)(v[k] + off);
        __syncthreads();
        if (threadIdx.x == 255) s_carry = off + v[3];
        __syncthreads();
    }
    if (threadIdx.x == 0 && seg_tot) seg_tot[blockIdx.x] = (Q)s_carry;
    (void)nrows;
}
// adds the inclusive-scanned total of the preceding segments of the same row (seg_incl is seg_tot after its own scan)
template <typename Q>
__global__ __launch_bounds__(256) void k_scan_x_fix(Q *__restrict__ q, uint64_t L, uint64_t segs_per_row,
                                                    const Q *__restrict__ seg_incl) {
    using UQ = typename std::make_unsigned<Q>::type;
    const uint64_t row = blockIdx.x / segs_per_row, seg = blockIdx.x % segs_per_row;
    if (seg == 0) return;
    const UQ off = (UQ)seg_incl[row * segs_per_row + seg - 1];
    const uint64_t lo = seg * SCANX_SEG, hi = (lo + SCANX_SEG < L) ? lo + SCANX_SEG : L;
    Q *r = q + row * L;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) r[i] = (Q)((UQ)r[i] + off);
}

// inclusive scan along a strided axis: one thread per line, lanes adjacent along the contiguous axis.
// element index = outer * (L * inner) + a * inner + in,  a = 0..L-1 ; inner = product of faster dims
template <typename Q>
__global__ __launch_bounds__(256) void k_scan_strided(Q *__restrict__ q, uint64_t L, uint64_t inner, uint64_t nlines) {
    using UQ = typename std::make_unsigned<Q>::type;
    const uint64_t line = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= nlines) return;
    const uint64_t outer = line / inner, in = line % inner;
    Q *pp = q + outer * L * inner + in;
    UQ run = 0;
    uint64_t a = 0;
    for (; a + 4 <= L; a += 4) {
        UQ v0 = (UQ)pp[(a + 0) * inner], v1 = (UQ)pp[(a + 1) * inner], v2 = (UQ)pp[(a + 2) * inner],
           v3 = (UQ)pp[(a + 3) * inner];
        v0 += run;
        v1 += v0;
        v2 += v1;
        v3 += v2;
        pp[(a + 0) * inner] = (Q)v0;
        pp[(a + 1) * inner] = (Q)v1;
        pp[(a + 2) * inner] = (Q)v2;
        pp[(a + 3) * inner] = (Q)v3;
        run = v3;
    }
    for (; a < L; a++) {
        run += (UQ)pp[a * inner];
        pp[a * inner] = (Q)run;
    }
}

// lattice index -> value, in place (Q and T have the same size)
template <typename T>
__global__ __launch_bounds__(256) void k_dequant(void *buf, uint64_t n, double two_eb) {
    using Q = typename QTraits<T>::Q;
    Q *q = reinterpret_cast<Q *>(buf);
    T *o = reinterpret_cast<T *>(buf);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Q v = q[i];
        o[i] = dequant<T>(v, two_eb);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_patch_vout(const uint8_t *__restrict__ payload, uint64_t idx_off,
                                                    uint64_t val_off, uint64_t cnt, uint64_t n, T *__restrict__ out) {
    const uint64_t *idx = reinterpret_cast<const uint64_t *>(payload + idx_off);
    const T *val = reinterpret_cast<const T *>(payload + val_off);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t k = idx[i];
        if (k < n) out[k] = val[i];
    }
}

// ------------------------------------------------------------------------------------------------------------
// launchers (host side, called from sz3hip_api.cpp)
// ------------------------------------------------------------------------------------------------------------
#define SZK_CHECK_LAUNCH()                      \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

static inline uint32_t grid_for(uint64_t n, uint32_t block, uint32_t cap) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

int szk_launch_minmax(int dtype, const void *d_in, uint64_t n, double *d_partial, double *d_out, hipStream_t s) {
    const int nb = 1024;
    if (dtype == 0) hipLaunchKernelGGL(k_minmax<float>, dim3(nb), dim3(256), 0, s, (const float *)d_in, n, d_partial);
    else hipLaunchKernelGGL(k_minmax<double>, dim3(nb), dim3(256), 0, s, (const double *)d_in, n, d_partial);
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(64), 0, s, d_partial, nb, d_out);
    SZK_CHECK_LAUNCH();
    return 0;
}

template <typename T>
static int launch_k1(int ndim, const void *d_in, uint16_t *codes, const szk_k1_params &p, hipStream_t s) {
    const uint64_t d0 = p.d[3], d1 = p.d[2], d2 = p.d[1], d3 = p.d[0];
    auto tiles = [&](uint64_t tx, uint64_t ty, uint64_t tz) {
        return ((d0 + tx - 1) / tx) * ((d1 + ty - 1) / ty) * ((d2 + tz - 1) / tz) * d3;
    };
    uint64_t nb;
    switch (ndim) {
        case 1:
            nb = tiles(4096, 1, 1);
            if (nb > 0x7FFFFFFFull) return -1;
            hipLaunchKernelGGL((k_lorenzo_quant<T, 1, 4096, 1, 1>), dim3((uint32_t)nb), dim3(256), 0, s, (const T *)d_in, codes, p);
            break;
        case 2:
            nb = tiles(128, 32, 1);
            if (nb > 0x7FFFFFFFull) return -1;
            hipLaunchKernelGGL((k_lorenzo_quant<T, 2, 128, 32, 1>), dim3((uint32_t)nb), dim3(256), 0, s, (const T *)d_in, codes, p);
            break;
        case 3:
            nb = tiles(64, 8, 8);
            if (nb > 0x7FFFFFFFull) return -1;
            hipLaunchKernelGGL((k_lorenzo_quant<T, 3, 64, 8, 8>), dim3((uint32_t)nb), dim3(256), 0, s, (const T *)d_in, codes, p);
            break;
        default:
            nb = tiles(64, 8, 4);
            if (nb > 0x7FFFFFFFull) return -1;
            hipLaunchKernelGGL((k_lorenzo_quant<T, 4, 64, 8, 4>), dim3((uint32_t)nb), dim3(256), 0, s, (const T *)d_in, codes, p);
            break;
    }
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_k1(int dtype, int ndim, const void *d_in, uint16_t *codes, const szk_k1_params *p, hipStream_t s) {
    return dtype == 0 ? launch_k1<float>(ndim, d_in, codes, *p, s) : launch_k1<double>(ndim, d_in, codes, *p, s);
}

int szk_launch_codebook(const uint64_t *d_hist, const szk_cb_params *p, hipStream_t s) {
    hipLaunchKernelGGL(k_codebook, dim3(1), dim3(CB_THREADS), 0, s, d_hist, *p);
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_layout_pre(const szk_layout_params *p, hipStream_t s) {
    hipLaunchKernelGGL(k_layout_pre, dim3(1), dim3(64), 0, s, *p);
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_encode(const uint16_t *codes, uint64_t n, const uint32_t *d_enc, int radius, uint16_t *chunk_words,
                      uint64_t *chunk_off, uint64_t *total_words, const szk_state *state, uint8_t *payload,
                      hipStream_t s) {
    const uint64_t n_chunks = (n + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS;
    const uint64_t nb = (n_chunks + 3) / 4;
    if (nb > 0x7FFFFFFFull) return -1;
    hipLaunchKernelGGL(k_chunk_bits, dim3((uint32_t)nb), dim3(256), 0, s, codes, n, d_enc, radius, chunk_words);
    hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, s, chunk_words, n_chunks, chunk_off, total_words);
    hipLaunchKernelGGL(k_encode, dim3((uint32_t)nb), dim3(256), 0, s, codes, n, d_enc, radius, chunk_off, state, payload);
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_assemble(const szk_asm_params *p, hipStream_t s) {
    hipLaunchKernelGGL(k_assemble, dim3(512), dim3(256), 0, s, *p);
    SZK_CHECK_LAUNCH();
    return 0;
}

int szk_launch_dec_tables(const uint8_t *d_lens, uint32_t sym_min, uint32_t sym_count, szk_dec_tables *t, hipStream_t s) {
    hipLaunchKernelGGL(k_dec_tables, dim3(1), dim3(1024), 0, s, d_lens, sym_min, sym_count, t);
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_decode(const uint8_t *payload, const szk_dec_params *p, uint16_t *codes, uint64_t *chunk_off,
                      uint64_t *total_words, hipStream_t s) {
    hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, s, p->chunk_words, p->n_chunks, chunk_off, total_words);
    const uint64_t nb = (p->n_chunks + 255) / 256;
    if (nb > 0x7FFFFFFFull) return -1;
    hipLaunchKernelGGL(k_decode, dim3((uint32_t)nb), dim3(256), 0, s, payload, *p, codes);
    SZK_CHECK_LAUNCH();
    return 0;
}

// inclusive scan of `nrows` contiguous rows of length L (recursive over segment totals for long rows)
template <typename Q>
static int scan_rows(Q *q, uint64_t L, uint64_t nrows, Q *scratch, hipStream_t s) {
    const uint64_t segs = (L + SCANX_SEG - 1) / SCANX_SEG;
    if (nrows * segs > 0x7FFFFFFFull) return -1;
    hipLaunchKernelGGL(k_scan_x<Q>, dim3((uint32_t)(nrows * segs)), dim3(256), 0, s, q, L, nrows, segs,
                       segs > 1 ? scratch : (Q *)nullptr);
    if (segs > 1) {
        int rc = scan_rows<Q>(scratch, segs, nrows, scratch + nrows * segs, s);  // totals -> inclusive per row
        if (rc) return rc;
        hipLaunchKernelGGL(k_scan_x_fix<Q>, dim3((uint32_t)(nrows * segs)), dim3(256), 0, s, q, L, segs, (const Q *)scratch);
    }
    return 0;
}

template <typename T>
static int launch_reconstruct(const uint8_t *payload, const szh_header &h, const szh_offsets &o, const uint16_t *codes,
                              void *d_out, void *d_segtot, hipStream_t s) {
    using Q = typename QTraits<T>::Q;
    Q *q = reinterpret_cast<Q *>(d_out);
    const uint64_t n = h.n;
    hipLaunchKernelGGL(k_expand_codes<Q>, dim3(grid_for(n, 256, 65536)), dim3(256), 0, s, codes, n, (int)h.radius, q);
    if (h.n_dout)
        hipLaunchKernelGGL(k_scatter_dout<Q>, dim3(grid_for(h.n_dout, 256, 4096)), dim3(256), 0, s, payload, o.dout_idx,
                           o.dout_val, h.n_dout, n, q);
    // axis x (contiguous)
    const uint64_t L = h.dims[3], nrows = n / L;
    {
        int rc = scan_rows<Q>(q, L, nrows, (Q *)d_segtot, s);
        if (rc) return rc;
    }
    // strided axes y, z, w
    uint64_t inner = L;
    for (int ax = 2; ax >= 0; ax--) {
        const uint64_t La = h.dims[ax];
        if (La > 1) {
            const uint64_t nlines = n / La;
            hipLaunchKernelGGL(k_scan_strided<Q>, dim3(grid_for(nlines, 256, 0x7FFFFFFF)), dim3(256), 0, s, q, La, inner, nlines);
        }
        inner *= La;
    }
    hipLaunchKernelGGL(k_dequant<T>, dim3(grid_for(n, 256, 65536)), dim3(256), 0, s, d_out, n, 2.0 * h.eb);
    if (h.n_vout)
        hipLaunchKernelGGL(k_patch_vout<T>, dim3(grid_for(h.n_vout, 256, 4096)), dim3(256), 0, s, payload, o.vout_idx,
                           o.vout_val, h.n_vout, n, (T *)d_out);
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_reconstruct(const uint8_t *payload, const szh_header *h, const szh_offsets *o, const uint16_t *codes,
                           void *d_out, void *d_segtot, hipStream_t s) {
    return h->dtype == 0 ? launch_reconstruct<float>(payload, *h, *o, codes, d_out, d_segtot, s)
                         : launch_reconstruct<double>(payload, *h, *o, codes, d_out, d_segtot, s);
}

void szk_host_offsets(const szh_header *h, szh_offsets *o) { szh_compute_offsets(*h, *o); }
