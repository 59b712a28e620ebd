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
//   K6  k_chunk_bits2 / k_scan_groups / k_pack    two-pass chunked bit-pack  (reference: HuffmanEncoder.hpp:140-218)
//   K8  k_dec_tables / k_decode / k_scan_x_wave (k_expand_codes, k_scatter_dout, k_scan_x*) / k_scan_strided /
//       k_scan_strided_dequant / k_patch_vout   chunk-parallel Huffman decode, Lorenzo inverse = N-d inclusive prefix sums
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
#include <cstddef>
#include <stdint.h>
#include <type_traits>
#include <mutex>
#include <utility>
#include <vector>
#include "sz3hip_format.h"
#include "sz3hip_kernels.h"

#include "sz3hip_devutil.h"

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

    const uint64_t d0 = p.d[3], d1 = p.d[2], d2 = p.d[1];  // x, y, z extents (w = p.d[0] multiplies the tile count)
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
    const Lattice<T> lat(p.lat);

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
            q = lat.quant(x, bad);
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
// K1 fast path (N = 3 or 4, x extent a multiple of 4, extents < 2^31): 64 x 8 x TZ tile; every thread owns TZ/2
// "row quads" (4 consecutive x at one (y,z)) that it loads with one 16-byte (f32) / two 16-byte (f64) global loads,
// prequantises, parks in LDS (row pitch 68: halo column at index 3, tile at 4..67 so that quads stay 16-byte
// aligned) and keeps in registers for the stencil; the three neighbour rows come back from LDS as one b128 + one
// b32 read each.  Codes leave as one 8-byte store per quad.  Index math is 32-bit inside the tile.
// ------------------------------------------------------------------------------------------------------------
template <typename T> struct V4 { T v[4]; };
__device__ __forceinline__ V4<float> ldg4(const float *p) {
    float4 t = *reinterpret_cast<const float4 *>(p);
    return {{t.x, t.y, t.z, t.w}};
}
__device__ __forceinline__ V4<double> ldg4(const double *p) {
    double2 a = reinterpret_cast<const double2 *>(p)[0], b = reinterpret_cast<const double2 *>(p)[1];
    return {{a.x, a.y, b.x, b.y}};
}
__device__ __forceinline__ void lds_st4(int32_t *p, const int32_t (&v)[4]) {
    *reinterpret_cast<int4 *>(p) = make_int4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void lds_st4(int64_t *p, const int64_t (&v)[4]) {
    reinterpret_cast<longlong2 *>(p)[0] = make_longlong2(v[0], v[1]);
    reinterpret_cast<longlong2 *>(p)[1] = make_longlong2(v[2], v[3]);
}
__device__ __forceinline__ void lds_ld4(const int32_t *p, int32_t (&v)[4]) {
    int4 t = *reinterpret_cast<const int4 *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void lds_ld4(const int64_t *p, int64_t (&v)[4]) {
    longlong2 a = reinterpret_cast<const longlong2 *>(p)[0], b = reinterpret_cast<const longlong2 *>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

// Persistent: the grid is a few workgroups per CU, each walks tiles blockIdx.x, +gridDim.x, ... so that the LDS
// histogram is cleared and flushed once per workgroup; the flush goes to a private row of `hist_partial`
// (no same-address global atomics: 32768 tiles hammering one 64-bit counter serialise at ~90 atomics/us) and
// k_hist_reduce folds the rows into the 65536-bin histogram.
// Software pipeline: the global loads of tile i+1 are issued into registers before the stencil of tile i runs, so
// every workgroup keeps ~20 KB of HBM reads in flight all the time (4 workgroups per CU -> ~80 KB per CU).
#define HIST_CSTRIDE (HIST_WIN + 8)  // copy k starts 8 banks further: one bin in different copies = different banks

__device__ __forceinline__ int32_t dpp_shr1(int32_t old, int32_t src) {  // lane l <- lane l-1 inside rows of 16 lanes
    return __builtin_amdgcn_update_dpp(old, src, 0x111, 0xf, 0xf, false);
}
__device__ __forceinline__ int64_t dpp_shr1(int64_t old, int64_t src) {
    int32_t lo = __builtin_amdgcn_update_dpp((int32_t)old, (int32_t)src, 0x111, 0xf, 0xf, false);
    int32_t hi = __builtin_amdgcn_update_dpp((int32_t)(old >> 32), (int32_t)(src >> 32), 0x111, 0xf, 0xf, false);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// four consecutive elements held in the native 16-byte vector registers a global load returns (no repacking, so
// the compiler has no reason to wait for the load before the first real use)
template <typename T> struct Quad;
template <> struct Quad<float> {
    float4 a;
#ifdef LAB_NT_LOAD  // (lab: streaming loads)
    __device__ __forceinline__ void load(const float *p) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
        a = make_float4(v.x, v.y, v.z, v.w);
    }
#else
    __device__ __forceinline__ void load(const float *p) { a = *reinterpret_cast<const float4 *>(p); }
#endif
    __device__ __forceinline__ float get(int i) const { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : a.w; }
};
template <> struct Quad<double> {
    double2 a, b;
    __device__ __forceinline__ void load(const double *p) {
        a = reinterpret_cast<const double2 *>(p)[0];
        b = reinterpret_cast<const double2 *>(p)[1];
    }
    __device__ __forceinline__ double get(int i) const { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? b.x : b.y; }
};

template <typename T, int NW, int RPT> struct K1Regs {
    Quad<T> own[NW][RPT];  // this thread's row quads
    Quad<T> h0[NW], h1[NW];  // halo-row quads (h1 only for threads 0..15)
    T col[NW];               // halo-column element (threads 0..PZ*PY-1)
    uint32_t valid;          // bit (lw*8 + k): own[lw][k]; bit (16 + lw*4 + {0,1,2}): h0, h1, col
};

template <typename T, int NDIM, int TZ>
__global__ __launch_bounds__(256, (NDIM == 3 ? 3 : 2)) void k_lorenzo_quant_v4(const T *__restrict__ in, uint16_t *__restrict__ codes,
                                                          szk_k1_params p, uint32_t ntiles) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    constexpr int TX = 64, TY = 8, NW = NDIM == 4 ? 2 : 1;
    constexpr int PX = TX + 4, PY = TY + 1, PZ = TZ + 1, SLAB = PZ * PY * PX;
    constexpr int RPT = TZ / 2;  // row quads per thread
    static_assert((PZ + TY) * 16 <= 256 + 16 && PZ * PY <= 256, "halo mapping assumes one/two quads per thread");
    __shared__ __attribute__((aligned(16))) Q lq[NW * SLAB];
    __shared__ uint32_t lh[HIST_COPIES * HIST_CSTRIDE + WAVE];  // + one private dummy bin per lane for the centre code

    const uint32_t d0 = (uint32_t)p.d[3], d1 = (uint32_t)p.d[2], d2 = (uint32_t)p.d[1];
    const uint32_t ntx = (d0 + TX - 1) / TX, nty = (d1 + TY - 1) / TY, ntz = (d2 + TZ - 1) / TZ;
    const uint32_t plane = d1 * d0;  // (TZ+1) planes < 2^31 elements guaranteed by the launcher
    const uint64_t vol = (uint64_t)plane * d2;
    const int t = threadIdx.x;
    const int lx4 = t & 15, ly = (t >> 4) & 7, lzb = t >> 7;
    const Lattice<T> lat(p.lat);
    const int radius = (int)p.radius;
    const int win_lo = radius - HIST_WIN / 2;
    uint32_t *myh = lh + (t & (HIST_COPIES - 1)) * HIST_CSTRIDE;
    const uint32_t dummy_bin = (uint32_t)(HIST_COPIES * HIST_CSTRIDE + lane_id()) - (uint32_t)((t & (HIST_COPIES - 1)) * HIST_CSTRIDE);
    uint32_t center_count = 0;
    // halo-row quads handled by this thread: index t (all threads) and 256 + t (threads 0..15 when needed)
    const int hr0 = t >> 4, hx0 = t & 15;
    const int h0_lz = hr0 < PZ ? hr0 - 1 : -1, h0_y = hr0 < PZ ? -1 : hr0 - PZ;
    constexpr bool HAS_H1 = (PZ + TY) * 16 > 256;
    const int hr1 = 16 + (t >> 4);  // only meaningful for t < 16
    const int h1_lz = hr1 < PZ ? hr1 - 1 : -1, h1_y = hr1 < PZ ? -1 : hr1 - PZ;
    const bool h0_on = hr0 < PZ + TY, h1_on = HAS_H1 && t < ((PZ + TY) * 16 - 256);
    const int c_lz = t / PY - 1, c_y = t % PY - 1;
    const bool col_on = t < PZ * PY;

    for (int i = t; i < HIST_COPIES * HIST_CSTRIDE + WAVE; i += 256) lh[i] = 0;

    auto decode = [&](uint32_t tile, int &x0, int &y0, int &z0, uint32_t &w) {
        uint32_t b = tile;
        x0 = (int)((b % ntx) * TX);
        b /= ntx;
        y0 = (int)((b % nty) * TY);
        b /= nty;
        z0 = (int)((b % ntz) * TZ);
        w = b / ntz;
    };
    // issue the global loads of one tile: always a load (out-of-array quads read element 0 and are masked later), no
    // branches, so that all of them go out back to back and stay in flight until the next iteration consumes them
    auto fetch = [&](uint32_t tile, K1Regs<T, NW, RPT> &R) {
        int x0, y0, z0;
        uint32_t w;
        decode(tile, x0, y0, z0, w);
        uint32_t valid = 0;
#pragma unroll
        for (int lw = 0; lw < NW; lw++) {
            const bool wok = (int)w - lw >= 0;
            const T *src = in + (uint64_t)(wok ? w - lw : 0) * vol + ((uint64_t)z0 * plane + (uint64_t)y0 * d0 + x0);
#pragma unroll
            for (int k = 0; k < RPT; k++) {
                const int lz = lzb + 2 * k;
                const uint32_t gz = z0 + lz, gy = y0 + ly, gx = x0 + 4 * lx4;
                const bool ok = wok && gz < d2 && gy < d1 && gx < d0;
                R.own[lw][k].load(ok ? src + ((uint32_t)lz * plane + (uint32_t)ly * d0 + 4u * lx4) : in);
                valid |= (uint32_t)ok << (lw * 8 + k);
            }
            {
                const int gz = z0 + h0_lz, gy = y0 + h0_y;
                const bool ok = h0_on && wok && gz >= 0 && gy >= 0 && (uint32_t)gz < d2 && (uint32_t)gy < d1 && (uint32_t)(x0 + 4 * hx0) < d0;
                R.h0[lw].load(ok ? src + ((int64_t)h0_lz * (int64_t)plane + (int64_t)h0_y * (int64_t)d0 + 4 * hx0) : in);
                valid |= (uint32_t)ok << (16 + lw * 4);
            }
            if (HAS_H1) {
                const int gz = z0 + h1_lz, gy = y0 + h1_y;
                const bool ok = h1_on && wok && gz >= 0 && gy >= 0 && (uint32_t)gz < d2 && (uint32_t)gy < d1 && (uint32_t)(x0 + 4 * hx0) < d0;
                R.h1[lw].load(ok ? src + ((int64_t)h1_lz * (int64_t)plane + (int64_t)h1_y * (int64_t)d0 + 4 * hx0) : in);
                valid |= (uint32_t)ok << (17 + lw * 4);
            }
            {
                const int gz = z0 + c_lz, gy = y0 + c_y;
                const bool ok = col_on && wok && x0 > 0 && gz >= 0 && gy >= 0 && (uint32_t)gz < d2 && (uint32_t)gy < d1;
                R.col[lw] = *(ok ? src + ((int64_t)c_lz * (int64_t)plane + (int64_t)c_y * (int64_t)d0 - 1) : in);
                valid |= (uint32_t)ok << (18 + lw * 4);
            }
        }
        R.valid = valid;
    };

    K1Regs<T, NW, RPT> R;
    uint32_t tile = blockIdx.x;
    if (tile < ntiles) fetch(tile, R);
    for (; tile < ntiles; tile += gridDim.x) {
        int x0, y0, z0;
        uint32_t w;
        decode(tile, x0, y0, z0, w);
        if ((p.dbg & 16) && tile != blockIdx.x) fetch(tile, R);
        __syncthreads();  // previous tile's stencil reads are done before the slab is overwritten

        // ---- prequantise the fetched registers into the LDS slab ----
        Q qreg[RPT][4];
        uint32_t badmask = 0;
        const uint32_t valid = R.valid;
#pragma unroll
        for (int lw = 0; lw < NW; lw++) {
            Q *slab = lq + lw * SLAB;
#pragma unroll
            for (int k = 0; k < RPT; k++) {
                const int lz = lzb + 2 * k;
                const bool ok = (valid >> (lw * 8 + k)) & 1u;
                Q q[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    bool bad;
                    const Q v = lat.quant(R.own[lw][k].get(i), bad);
                    q[i] = ok ? v : (Q)0;
                    if (lw == 0) badmask |= (uint32_t)(bad & ok) << (4 * k + i);
                }
                lds_st4(slab + ((lz + 1) * PY + (ly + 1)) * PX + 4 + 4 * lx4, q);
                if (lw == 0) {
#pragma unroll
                    for (int i = 0; i < 4; i++) qreg[k][i] = q[i];
                }
            }
            if (h0_on) {
                const bool ok = (valid >> (16 + lw * 4)) & 1u;
                Q q[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    bool bad;
                    const Q v = lat.quant(R.h0[lw].get(j), bad);
                    q[j] = ok ? v : (Q)0;
                }
                lds_st4(slab + ((h0_lz + 1) * PY + (h0_y + 1)) * PX + 4 + 4 * hx0, q);
            }
            if (HAS_H1 && h1_on) {
                const bool ok = (valid >> (17 + lw * 4)) & 1u;
                Q q[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    bool bad;
                    const Q v = lat.quant(R.h1[lw].get(j), bad);
                    q[j] = ok ? v : (Q)0;
                }
                lds_st4(slab + ((h1_lz + 1) * PY + (h1_y + 1)) * PX + 4 + 4 * hx0, q);
            }
            if (col_on) {
                const bool ok = (valid >> (18 + lw * 4)) & 1u;
                bool bad;
                const Q v = lat.quant(R.col[lw], bad);
                slab[((c_lz + 1) * PY + (c_y + 1)) * PX + 3] = ok ? v : (Q)0;
            }
        }
        // ---- next tile's loads go out now and stay in flight across the barrier and the stencil ----
        if (!(p.dbg & 16) && tile + gridDim.x < ntiles) fetch(tile + gridDim.x, R);
        __syncthreads();

        uint16_t *ctile = codes + (uint64_t)w * vol + ((uint64_t)z0 * plane + (uint64_t)y0 * d0 + x0);
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int lz = lzb + 2 * k;
            const uint32_t gz = z0 + lz, gy = y0 + ly, gx = x0 + 4 * lx4;
            if (!(gz < d2 && gy < d1 && gx < d0)) continue;
            const int c = ((lz + 1) * PY + (ly + 1)) * PX + 4 + 4 * lx4;
            UQ delta[4];
#pragma unroll
            for (int lw = 0; lw < NW; lw++) {
                const Q *L = lq + lw * SLAB;
                Q cur[4], ra[4], rb[4], rd[4];
                if (lw == 0) {
#pragma unroll
                    for (int i = 0; i < 4; i++) cur[i] = qreg[k][i];
                } else {
                    lds_ld4(L + c, cur);
                }
                if (p.dbg & 4) {
#pragma unroll
                    for (int i = 0; i < 4; i++) ra[i] = rb[i] = rd[i] = cur[i];
                } else {
                    lds_ld4(L + c - PX, ra);
                    lds_ld4(L + c - PY * PX, rb);
                    lds_ld4(L + c - PY * PX - PX, rd);
                }
                // x-1 neighbours: the left lane's 4th element (DPP row_shr:1); the 16-lane row leader reads the halo
                // column from LDS (the other lanes read one common word: broadcast, no bank conflict)
                const bool lead = lx4 == 0;
                const Q hc = L[lead ? c - 1 : 0], ha = L[lead ? c - PX - 1 : 0], hb = L[lead ? c - PY * PX - 1 : 0],
                        hd = L[lead ? c - PY * PX - PX - 1 : 0];
                UQ pc = (UQ)dpp_shr1(hc, cur[3]), pa = (UQ)dpp_shr1(ha, ra[3]), pb = (UQ)dpp_shr1(hb, rb[3]),
                   pd = (UQ)dpp_shr1(hd, rd[3]);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    UQ s = ((UQ)cur[i] - pc) - ((UQ)ra[i] - pa) - ((UQ)rb[i] - pb) + ((UQ)rd[i] - pd);
                    pc = (UQ)cur[i];
                    pa = (UQ)ra[i];
                    pb = (UQ)rb[i];
                    pd = (UQ)rd[i];
                    delta[i] = lw == 0 ? s : (UQ)(delta[i] - s);
                }
            }
            // branch-free hot path: code, histogram bin (centre code -> private dummy bin + register count)
            uint32_t code[4];
            bool slow = ((badmask >> (4 * k)) & 15u) != 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const UQ shifted = delta[i] + (UQ)radius;  // in (0, 2r) when |delta| < r
                const bool inr = (UQ)(shifted - 1) < (UQ)(2 * radius - 1);
                code[i] = inr ? (uint32_t)shifted : 0u;
                const uint32_t bin = code[i] - (uint32_t)win_lo;
                const bool inwin = bin < (uint32_t)HIST_WIN;
                const bool ctr = code[i] == (uint32_t)radius;
                center_count += ctr;
                slow |= !inr | !inwin;
#ifdef LAB_ABLATE  // (lab builds: bit 1 of the debug flags switches the histogram off; in the product library the bit belongs to the code book)
                if (!(p.dbg & 1))
#endif
                atomicAdd(&myh[(ctr | !inwin) ? dummy_bin : bin], 1u);
            }
            const uint32_t off = (uint32_t)lz * plane + (uint32_t)ly * d0 + 4u * lx4;
            uint2 pk;
            pk.x = code[0] | (code[1] << 16);
            pk.y = code[2] | (code[3] << 16);
            if (!(p.dbg & 2)) *reinterpret_cast<uint2 *>(ctile + off) = pk;
            if (slow) {  // rare: delta outliers, value outliers, codes outside the LDS histogram window
                const uint64_t gi = (uint64_t)w * vol + (uint64_t)gz * plane + (uint64_t)gy * d0 + gx;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (code[i] == 0) {
                        unsigned long long pos = atomicAdd((unsigned long long *)p.n_dout, 1ull);
                        if (pos < p.out_cap) {
                            p.dout_idx[pos] = gi + i;
                            ((Q *)p.dout_val)[pos] = (Q)delta[i];
                        }
                    }
                    if ((badmask >> (4 * k + i)) & 1u) {
                        unsigned long long pos = atomicAdd((unsigned long long *)p.n_vout, 1ull);
                        if (pos < p.out_cap) {
                            p.vout_idx[pos] = gi + i;
                            ((T *)p.vout_val)[pos] = in[gi + i];
                        }
                    }
                    if (code[i] - (uint32_t)win_lo >= (uint32_t)HIST_WIN)
                        atomicAdd((unsigned long long *)&p.hist[code[i]], 1ull);
                }
            }
        }
    }
    // centre counts: one LDS add per thread, then the private row of hist_partial
    atomicAdd(&lh[radius - win_lo], center_count);
    __syncthreads();
    uint32_t *row = p.hist_partial + (uint64_t)blockIdx.x * HIST_WIN;
    for (int bnn = t; bnn < HIST_WIN; bnn += 256)
        row[bnn] = lh[bnn] + lh[HIST_CSTRIDE + bnn] + lh[2 * HIST_CSTRIDE + bnn] + lh[3 * HIST_CSTRIDE + bnn];
}

// ------------------------------------------------------------------------------------------------------------
// "Narrow" intermediate codes.  When a sparse probe of the Lorenzo deltas (64 consecutive elements of every 32768) finds fewer than one
// in 4096 outside [-127, 127], stage 1 stores the codes as ONE byte (delta + 128, 0 = delta outlier) instead of two
// and the packers read one byte — 0.4 GB less HBM traffic per 512^3 volume.  Symbols, histogram, code book and
// payload are unchanged (symbol = delta + radius); the rare wider deltas join the delta-outlier list.  The decision
// is a pure function of the probe counter, recomputed by every kernel (no host round trip).
// ------------------------------------------------------------------------------------------------------------
#define MARCH_WIDE_WIN 8192  // LDS histogram bins of the march kernel when the codes are two bytes wide (x2 on request)
__device__ __forceinline__ bool szk_is_narrow(const szk_mode &m) {
    return m.allow && (unsigned long long)(*m.probe_big) * 4096ull <= m.n_samples;
}

// adds `cnt` to a bin of the global histogram; whoever finds the bin empty also enters it into the alphabet's range words
// (range[0] = max(65535 - bin), [1] = max bin, [2] = number of non-empty bins): the code book then needs no pass of its own
// over the 65536 bins (k_hist_range) as long as nobody else touched the histogram
__device__ __forceinline__ void hist_add_ranged(uint64_t *hist, uint32_t *range, uint32_t sym, unsigned long long cnt) {
    if (!range) {  // (no range words kept: the add needs no return value — a returning atomic per bin and workgroup made the wide
                   // window's flush cost 0.47 ms at C4's slab)
        atomicAdd((unsigned long long *)&hist[sym], cnt);
        return;
    }
    const unsigned long long old = atomicAdd((unsigned long long *)&hist[sym], cnt);
    if (old == 0) {
        atomicMax(&range[0], 0xFFFFu - sym);
        atomicMax(&range[1], sym);
        atomicAdd(&range[2], 1u);
    }
}

// (a device function: k_probe runs it as a launch of its own; the one-launch form of stage 1, which a context takes after a
// one-byte call, runs it as the first thing its own workgroups do — it ASSUMES one-byte codes and nothing in it waits for
// the probe; the encoder's kernels read the counters afterwards and the host repeats the call when the assumption was wrong)
template <typename T, int NDIM>
__device__ __forceinline__ void probe_body(const T *__restrict__ in, const szk_k1_params &p, uint64_t n, uint32_t *probe_big, uint32_t *s_p) {
    using UQ = typename QTraits<T>::UQ;
    const Lattice<T> lat(p.lat);
    const uint64_t d0 = p.d[3], d1 = p.d[2], d2 = p.d[1];
    // counts per thread over a grid-stride loop, then one atomic per counter and WORKGROUP (on rough data every wave finds
    // some: 4096 waves x same-address atomics at ~90/us were 45 us of a 56 us kernel)
    uint32_t n_big = 0, n_far = 0, n_pfar = 0, n_qbig = 0;  // n_qbig: stencil values beyond Q16_LIM / 2 lattice steps, or not finite (the 16-bit form's licence)
    const uint64_t n_runs = (n + SZK_PROBE_STRIDE - 1) / SZK_PROBE_STRIDE;
    for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; (s >> 6) < n_runs; s += (uint64_t)gridDim.x * 256) {
    const uint64_t i = (s >> 6) * SZK_PROBE_STRIDE + (s & 63);  // runs of 64 consecutive elements, one run per stride
    bool qbig = false;
    bool big = false, far = false, pfar = false;  // far: beyond the 8192-bin stage-1 window of the two-byte kernel, inside the 16384-bin one
                                                  // pfar: beyond the packers' 4096-entry table window, inside the doubled one
    if (i < n) {
        uint64_t r = i;
        const int64_t x = (int64_t)(r % d0);
        r /= d0;
        const int64_t y = (int64_t)(r % d1);
        r /= d1;
        const int64_t z = (int64_t)(r % d2);
        const int64_t w = (int64_t)(r / d2);
        UQ delta = 0;
#pragma unroll
        for (int c = 0; c < (1 << NDIM); c++) {
            const int64_t xx = x - (c & 1), yy = y - ((c >> 1) & 1), zz = z - ((c >> 2) & 1), ww = w - ((c >> 3) & 1);
            UQ q = 0;
            if (xx >= 0 && yy >= 0 && zz >= 0 && ww >= 0) {
                bool bad;
                const T v = in[(((uint64_t)ww * d2 + (uint64_t)zz) * d1 + (uint64_t)yy) * d0 + (uint64_t)xx];
                q = (UQ)lat.quant(v, bad);
                const T sv = v * lat.recip;
                qbig |= !((sv < (T)0 ? -sv : sv) <= (T)2047);  // (NaN: counted)
            }
            delta = (__popc(c) & 1) ? (UQ)(delta - q) : (UQ)(delta + q);
        }
        big = (UQ)(delta + 127) > (UQ)254;
        // codes the doubled window would catch and the plain one would not (what lies beyond both costs the same either way)
        far = (UQ)(delta + (UQ)(MARCH_WIDE_WIN / 2)) >= (UQ)MARCH_WIDE_WIN && (UQ)(delta + (UQ)MARCH_WIDE_WIN) < (UQ)(2 * MARCH_WIDE_WIN);
        pfar = (UQ)(delta + (UQ)2048) >= (UQ)4096 && (UQ)(delta + (UQ)4096) < (UQ)8192;
    }
    n_big += big;
    n_far += far;
    n_pfar += pfar;
    n_qbig += qbig;
    }
    if (threadIdx.x < 4) s_p[threadIdx.x] = 0;
    __syncthreads();
    n_big = wave_sum(n_big);
    n_far = wave_sum(n_far);
    n_pfar = wave_sum(n_pfar);
    n_qbig = wave_sum(n_qbig);
    if (lane_id() == 0) {
        if (n_qbig) atomicAdd(&s_p[3], n_qbig);
        if (n_big) atomicAdd(&s_p[0], n_big);
        if (n_far) atomicAdd(&s_p[1], n_far);
        if (n_pfar) atomicAdd(&s_p[2], n_pfar);
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_p[threadIdx.x]) atomicAdd(probe_big + threadIdx.x, s_p[threadIdx.x]);  // ([1], [2], [3]: read by the host after the call)
}
template <typename T, int NDIM>
__global__ __launch_bounds__(256) void k_probe(const T *__restrict__ in, szk_k1_params p, uint64_t n, uint32_t *probe_big) {
    __shared__ uint32_t s_p[4];
    probe_body<T, NDIM>(in, p, n, probe_big, s_p);
}

// ------------------------------------------------------------------------------------------------------------
// K1 "march" (N = 3 or 4, x extent a multiple of 4 and >= 128): register-only stencil, no LDS tile, no barriers.
// One wave owns a 256 (x) x TY (y) x TZ (z) brick: every lane holds 4 consecutive x, the wave walks the rows of a
// plane and the planes of the brick keeping
//     d1(x)   = q(x) - q(x-1)            left neighbour through DPP wave_shr:1 (lane 0: one extra 4-byte load)
//     d2(x,y) = d1(y) - d1(y-1)          previous row in registers
//     delta   = d2(z) - d2(z-1)          previous plane's TY rows in registers
// (the N-d Lorenzo stencil is the product of first differences).  Row loads are 1 KiB contiguous per wave
// (16 B per lane); a plane's TY+1 rows are requested together so ~5 KiB per wave are in flight, with 5-8 waves per
// SIMD.  The low-side halo row / plane of a brick is re-read and re-quantised (L2 hits), never exchanged.
// The histogram lives in LDS per workgroup ([bin][4 copies] interleaved), flushed to a private row of
// hist_partial (see k_hist_reduce).
// ------------------------------------------------------------------------------------------------------------
#define MARCH_TX 256
#ifndef MARCH_TZ
#define MARCH_TZ 16
#endif

__device__ __forceinline__ int32_t dpp_wave_shr1(int32_t old, int32_t src) {  // lane l <- lane l-1 across the wave
    return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int64_t dpp_wave_shr1(int64_t old, int64_t src) {
    int32_t lo = __builtin_amdgcn_update_dpp((int32_t)old, (int32_t)src, 0x138, 0xf, 0xf, false);
    int32_t hi = __builtin_amdgcn_update_dpp((int32_t)(old >> 32), (int32_t)(src >> 32), 0x138, 0xf, 0xf, false);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

#define MARCH_OQ 128  // records per wave in the LDS outlier staging queue (f32, two-byte kernel: 32 KB window + 6 KB queues = 4 workgroups per CU)
// MODE 0: the code width (one or two bytes, decided on the device by k_probe) is a run-time branch. MODE 1 / 2: the kernel
// is specialised for one-byte / two-byte codes and returns at once when the probe chose the other width; the host launches
// both. The one-byte specialisation needs a third of the LDS (16 KB histogram, 8 KB outlier queue): 4 waves per SIMD
// instead of 3.
// geometry of a form's LDS (shared by the kernel, which declares the arrays, and the body, which indexes them)
template <int MODE, bool WIN16> struct MarchLds {
    // MODE 1 (one-byte codes) / MODE 4 (two-byte fallback inside the one-byte launch): HIST_WIN x 4 words, short outlier queues
    static constexpr int WIDE_WIN = MODE == 4 ? HIST_WIN * 4 : (WIN16 ? 2 * MARCH_WIDE_WIN : MARCH_WIDE_WIN);
    static constexpr int LH_WORDS = (MODE == 1 || MODE == 4) ? HIST_WIN * 4 : WIDE_WIN;
    static constexpr int OQ = (MODE == 1 || MODE == 4) ? 128 : MARCH_OQ;
};
// MODE 0: code width decided at run time; 1 / 2: one-byte / two-byte specialisation of the two-launch form (returns at once
// when the probe chose the other width); 4: two-byte codes inside the one-byte form's LDS budget (see k_lorenzo_quant_march3)
template <typename T, int NDIM, int TY, int MODE, bool WIN16>
__device__ __forceinline__ void march_body(const T *__restrict__ in, uint16_t *__restrict__ codes, const szk_k1_params &p, uint32_t ntasks,
                                           uint32_t nrows, uint32_t *lh, uint64_t (*s_oq_idx)[MarchLds<MODE, WIN16>::OQ],
                                           typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type (*s_oq_val)[MarchLds<MODE, WIN16>::OQ]) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    constexpr int NW = NDIM == 4 ? 2 : 1;
    constexpr int WIDE_WIN = MarchLds<MODE, WIN16>::WIDE_WIN, LH_WORDS = MarchLds<MODE, WIN16>::LH_WORDS, OQ = MarchLds<MODE, WIN16>::OQ;
    if ((MODE == 1 || MODE == 2) && szk_is_narrow(p.mode) != (MODE == 1)) return;
    using OQV = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;  // raw bits of a value

    const uint32_t d0 = (uint32_t)p.d[3], d1 = (uint32_t)p.d[2], d2 = (uint32_t)p.d[1];
    const uint32_t ntx = (d0 + MARCH_TX - 1) / MARCH_TX, nty = (d1 + TY - 1) / TY, ntz = (d2 + MARCH_TZ - 1) / MARCH_TZ;
    const uint64_t plane = (uint64_t)d1 * d0, vol = plane * d2;
    const int lane = lane_id();
    const Lattice<T> lat(p.lat);
    const int radius = (int)p.radius;
    const uint32_t copy = (uint32_t)lane & 3u;
    const bool narrow = MODE == 1 ? true : ((MODE == 2 || MODE == 4) ? false : szk_is_narrow(p.mode));
    const uint32_t win_bins = narrow ? (uint32_t)HIST_WIN : (uint32_t)WIDE_WIN;
    const uint32_t win_lo = (uint32_t)radius - win_bins / 2;
    // in-range test of a delta, one form for both code widths: (delta + rng_lo) <= rng_span (unsigned)
    const UQ rng_lo = narrow ? (UQ)127 : (UQ)(radius - 1), rng_span = narrow ? (UQ)254 : (UQ)(2 * radius - 2);
    uint8_t *codes8 = reinterpret_cast<uint8_t *>(codes);

    for (int i = threadIdx.x; i < LH_WORDS + 4; i += 256) lh[i] = 0;
    __syncthreads();

    uint64_t *oq_idx = s_oq_idx[threadIdx.x / WAVE];
    OQV *oq_val = s_oq_val[threadIdx.x / WAVE];
    uint32_t oq_n = 0;  // fill level; lane 0 takes part in every update, the other lanes re-read its copy before use
    auto oq_flush = [&]() {
        // Called by whatever lanes are active: a tile that ends beyond the array's x extent flushes (when its queue fills) with
        // its last lanes off. The records are dealt over the ACTIVE lanes — striding by 64 dropped the records whose index fell
        // on an inactive lane (3 of 64 for rows of 500: a bound violation on fields with many unpredictable values).
        oq_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)oq_n);
        if (oq_n == 0) return;
        const unsigned long long act = __ballot(1);
        const uint32_t nact = (uint32_t)__popcll(act), rank = (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
        unsigned long long base = 0;
        if (rank == 0) base = atomicAdd((unsigned long long *)p.n_vout, (unsigned long long)oq_n);
        const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
        const unsigned long long b0 = ((unsigned long long)bhi << 32) | blo;
        for (uint32_t k = rank; k < oq_n; k += nact) {
            const unsigned long long pos = b0 + k;
            if (pos < p.out_cap) {
                p.vout_idx[pos] = oq_idx[k];
                if (sizeof(T) == 4) reinterpret_cast<uint32_t *>(p.vout_val)[pos] = (uint32_t)oq_val[k];
                else reinterpret_cast<uint64_t *>(p.vout_val)[pos] = oq_val[k];
            }
        }
        oq_n = 0;
    };
    // XCD-aware task order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the workgroups of one
    // XCD take a contiguous range of tasks: tiles that share halo rows / planes then share an L2 instead of each pulling
    // the shared lines from HBM
    // (grids that are not a multiple of 8 keep the plain order: the remapped sequence would have holes)
    const uint32_t per_xcd = gridDim.x / 8u;
    const uint32_t wg_seq = gridDim.x % 8u == 0 && !(p.dbg & 4096u) ? (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u : blockIdx.x;
    const uint32_t wave_gid = wg_seq * 4 + threadIdx.x / WAVE, nwaves = gridDim.x * 4u;
    for (uint32_t task = wave_gid; task < ntasks; task += nwaves) {
        uint32_t b = task;
        const uint32_t x0 = (b % ntx) * MARCH_TX;
        b /= ntx;
        const uint32_t y0 = (b % nty) * TY;
        b /= nty;
        const uint32_t z0 = (b % ntz) * MARCH_TZ;
        const uint32_t w = b / ntz;
        const uint32_t x = x0 + 4 * lane;
        const bool xok = x < d0;            // quad granular (d0 % 4 == 0)
        const bool has_left = x0 > 0;       // lane 0 needs element x0 - 1
        const uint32_t ny = (d1 - y0 < (uint32_t)TY) ? d1 - y0 : (uint32_t)TY;
        // per-lane element offsets inside a row, fixed for the task: own quad, and the element left of it (lane 0 of a tile
        // that has a left neighbour reads x0 - 1; the other lanes' values are never used)
        const uint32_t lane_off = xok ? x : 0u;

        UQ pp[NW][TY][4];  // d2 of the previous plane
#pragma unroll
        for (int lw = 0; lw < NW; lw++)
#pragma unroll
            for (int yy = 0; yy < TY; yy++)
#pragma unroll
                for (int i = 0; i < 4; i++) pp[lw][yy][i] = 0;

        const int zstart = z0 > 0 ? -1 : 0;
        for (int zz = zstart; zz < MARCH_TZ; zz++) {
            const uint32_t gz = z0 + zz;
            if (gz >= d2) break;
            // ---- request the plane's rows: halo row y0-1 (slot 0) and rows y0 .. y0+TY-1 (slots 1..TY) ----
            Quad<T> rq[NW][TY + 1];
            T rl[NW][TY + 1];
#pragma unroll
            for (int lw = 0; lw < NW; lw++) {
                const bool wok = (int)w - lw >= 0;
                const T *src = in + (uint64_t)(wok ? w - lw : 0) * vol + (uint64_t)gz * plane;  // wave-uniform
#pragma unroll
                for (int r = 0; r <= TY; r++) {
                    const int gy = (int)y0 + r - 1;
                    const bool rok = wok && gy >= 0 && (uint32_t)gy < d1;
                    // rows outside the array are read from row 0 of the plane (valid memory) and masked below:
                    // the row offset is wave-uniform (scalar unit), only the lane offset is per-lane
                    const T *row = src + (uint64_t)(rok ? (uint32_t)gy : 0u) * d0;
                    rq[lw][r].load(row + lane_off);
                    // wave-uniform address (element left of the tile). 3-D: tiles at x = 0 skip it; the 4-D kernel (two time
                    // slabs in flight) is 20 % slower with that branch in its row loop and keeps the unconditional form
                    rl[lw][r] = (NW == 2 || has_left) ? row[x0 > 0 ? x0 - 1 : 0] : (T)0;
                }
            }
            // ---- rows ----
            UQ pd1[NW][4];  // d1 of the previous row
            UQ delta[4];
#pragma unroll
            for (int r = 0; r <= TY; r++) {
                const int gy = (int)y0 + r - 1;
                uint32_t badmask = 0;
#pragma unroll
                for (int lw = 0; lw < NW; lw++) {
                    const bool wok = (int)w - lw >= 0;
                    const bool rok = wok && gy >= 0 && (uint32_t)gy < d1;
                    const bool ok = rok && xok;
                    Q q[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        bool bad;
                        const Q v = lat.quant(rq[lw][r].get(i), bad);
                        q[i] = ok ? v : (Q)0;
                        if (lw == 0) badmask |= (uint32_t)(bad & ok) << i;
                    }
                    Q left0 = (Q)0;  // only lane 0's value is used
                    if (NW == 2 || has_left) {  // wave-uniform (compile-time true for 4-D)
                        bool badl;
                        const Q ql = lat.quant(rl[lw][r], badl);
                        left0 = (rok && has_left) ? ql : (Q)0;
                    }
                    UQ pv = (UQ)dpp_wave_shr1(left0, q[3]);
                    UQ d1v[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        d1v[i] = (UQ)q[i] - pv;
                        pv = (UQ)q[i];
                    }
                    if (r > 0) {
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const UQ d2v = d1v[i] - pd1[lw][i];
                            const UQ s = d2v - pp[lw][r - 1][i];
                            pp[lw][r - 1][i] = d2v;
                            delta[i] = lw == 0 ? s : (UQ)(delta[i] - s);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) pd1[lw][i] = d1v[i];
                }
                if (r == 0 || zz < 0) continue;                       // halo row / halo plane: state only
                if ((uint32_t)(r - 1) >= ny || !xok) continue;         // beyond the array
                // ---- codes, histogram, store ----
                uint32_t code[4];
                bool rare = badmask != 0;
                const uint64_t gi = (uint64_t)w * vol + (uint64_t)gz * plane + (uint64_t)gy * d0 + x;
                if (narrow) {  // one byte per code: delta + 128, 0 = delta outlier; every in-range code lies inside the window
                    uint32_t pk8 = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        // t = delta + 127 as an unsigned number: in range <=> t <= 254; clamping it to 255 sends every
                        // out-of-range delta to the spare bin next to the window's last used one and to byte (255 + 1) & 255 = 0
                        const UQ tq = delta[i] + (UQ)127;
                        const bool inr = tq <= (UQ)254;
                        const uint32_t tc = inr ? (uint32_t)tq : 255u;
                        const uint32_t byte = tc;  // (the stored byte: delta + 127, 255 = delta outlier)
                        const uint32_t bin = tc + (uint32_t)(HIST_WIN / 2 - 127);  // delta + radius - win_lo; 255 + 385 = spare bin, skipped by the flush
                        rare |= !inr;
                        pk8 |= byte << (8 * i);
                        code[i] = inr ? (uint32_t)delta[i] + (uint32_t)radius : 0u;  // only read on the rare path
                        atomicAdd(&lh[bin * 4 + copy], 1u);
                    }
                    *reinterpret_cast<uint32_t *>(codes8 + gi) = pk8;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const UQ shifted = delta[i] + (UQ)radius;  // in (0, 2r) when |delta| < r
                        const bool inr = (UQ)(delta[i] + rng_lo) <= rng_span;
                        code[i] = inr ? (uint32_t)shifted : 0u;
                        uint32_t bin = code[i] - win_lo;           // wraps to a huge value below the window
                        // (a delta outlier is rare whatever its bin: with a small radius code 0 lies inside the window)
                        rare |= !inr || bin >= (uint32_t)WIDE_WIN;
                        bin = bin < (uint32_t)WIDE_WIN ? bin : (uint32_t)WIDE_WIN;
                        atomicAdd(&lh[bin], 1u);
                    }
                    uint2 pk;
                    pk.x = code[0] | (code[1] << 16);
                    pk.y = code[2] | (code[3] << 16);
                    *reinterpret_cast<uint2 *>(codes + gi) = pk;
                }
                if (__ballot(rare)) {  // some lane has outliers or codes outside the LDS histogram window
                    {   // the far deltas of the lane's four elements: one atomic per wave for all of them (an atomic per element position
                        // made rows with a far delta in most waves crawl: same-address atomics run at ~90 per us)
                        uint32_t nfar = 0;
#pragma unroll
                        for (int i = 0; i < 4; i++) nfar += rare && code[i] == 0;
                        unsigned long long pd = wave_append_run(nfar, p.n_dout);
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            if (rare && code[i] == 0) {
                                if (pd < p.out_cap) {
                                    p.dout_idx[pd] = gi + i;
                                    ((Q *)p.dout_val)[pd] = (Q)delta[i];
                                }
                                pd++;
                            }
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const bool is_vout = rare && ((badmask >> i) & 1u);
                        const unsigned long long vm = __ballot(is_vout);
                        if (vm) {
                            oq_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)oq_n);
                            if (oq_n + WAVE > (uint32_t)OQ) oq_flush();
                            if (is_vout) {
                                const uint32_t slot = oq_n + (uint32_t)__popcll(vm & ((1ull << lane) - 1ull));
                                oq_idx[slot] = gi + i;
                                const T raw = in[gi + i];
                                OQV bits = 0;
                                memcpy(&bits, &raw, sizeof(T));
                                oq_val[slot] = bits;
                            }
                            oq_n += (uint32_t)__popcll(vm);
                        }
                        // what the LDS window did not count goes to the global histogram: one-byte mode keeps its delta
                        // outliers (code 0) out of the window (overflow bin) even when a small radius puts bin 0 inside it
                        const bool in_lds = narrow ? code[i] != 0 : code[i] - win_lo < win_bins;
                        if (rare && !in_lds) {
                            // code 0 (delta outliers) is one address for the whole grid: one atomic per wave
                            const unsigned long long zm = __ballot(code[i] == 0);
                            // (the range words are kept by the one-launch kernel's two bodies only: in the two-byte
                            // specialisation the ranged add — even with a null range — cost 195 of 555 us at C4's slab)
                            constexpr bool RANGED = MODE == 1 || MODE == 4;
                            if (RANGED) {
                                if (code[i] != 0) hist_add_ranged(p.hist, p.range, code[i], 1ull);
                                else if (lane == __ffsll((long long)zm) - 1) hist_add_ranged(p.hist, p.range, 0u, (unsigned long long)__popcll(zm));
                            } else {
                                if (code[i] != 0) atomicAdd((unsigned long long *)&p.hist[code[i]], 1ull);
                                else if (lane == __ffsll((long long)zm) - 1) atomicAdd((unsigned long long *)&p.hist[0], (unsigned long long)__popcll(zm));
                            }
                        }
                    }
                }
            }
        }
    }
    oq_flush();
    __syncthreads();
    uint32_t *row = p.hist_partial + (uint64_t)blockIdx.x * HIST_WIN;
    if (narrow) {
        for (int bnn = threadIdx.x; bnn < HIST_WIN; bnn += 256)  // (bin 255 + 385 collected the out-of-range deltas: not a symbol)
            row[bnn] = bnn == 255 + HIST_WIN / 2 - 127 ? 0u : lh[bnn * 4] + lh[bnn * 4 + 1] + lh[bnn * 4 + 2] + lh[bnn * 4 + 3];
    } else {  // wide window: straight into the global histogram (the bins are spread, no hot address), empty rows for the fold
        // (nrows = rows the fold reads: the larger grid of the two specialisations)
        for (uint32_t r = blockIdx.x; r < nrows; r += gridDim.x)
            for (int bnn = threadIdx.x; bnn < HIST_WIN; bnn += 256) p.hist_partial[(uint64_t)r * HIST_WIN + bnn] = 0;
        for (int bnn = threadIdx.x; bnn < WIDE_WIN; bnn += 256) {
            const uint32_t v = lh[bnn];
            const uint32_t sym = win_lo + (uint32_t)bnn;
            if (v && sym < SZH_HIST_BINS) {
                if (MODE == 1 || MODE == 4) hist_add_ranged(p.hist, p.range, sym, (unsigned long long)v);
                else atomicAdd((unsigned long long *)&p.hist[sym], (unsigned long long)v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// The one-byte form of the marching kernel (round 3). Same bricks, same walk and the same results as march_body<.., 1, ..>
// (which it replaces), written for the vector ALU's issue rate, the kernel's bound at 27.6 instructions per element:
//   * every task coordinate is wave-uniform by construction (readfirstlane): row / plane addresses, validity of the halo
//     row, the halo plane and the left neighbour live in scalar registers and are decided by scalar branches;
//   * the lattice value is the bit pattern of fl(x / 2eb + M) (Lattice<T>::qbits): one multiply, one add, one integer
//     clamp per value; the stencil differences the patterns as they are; a neighbour outside the array is the constant C;
//   * a brick whose 256 x TY footprint lies inside the array (EDGE = false: all of them when the extents divide) carries
//     no per-lane validity selects;
//   * codes are stored as t = min(delta + 127, 255) — 255 = "outside [-127, 127]", a delta outlier — which is also the
//     bin of the LDS histogram (256 bins x 4 copies);
//   * the bound check's outcomes stay lane masks in scalar registers; one scalar test per row decides whether any lane
//     has something rare to do.
// With a code-length table (the context's previous code book: speculative stage 2, sz3hip_api.cpp) the kernel also sums
// the code lengths of every 256-element row segment it codes — the bits pass of the encoder, without its trip over the codes.
// ------------------------------------------------------------------------------------------------------------
#define NARROW_BINS 256  // t = min(delta + 127, 255); bin 255 collects the out-of-range deltas (not a symbol)
template <typename T> struct NarrowCtx {
    using B = typename Lattice<T>::B;
    using OQV = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;
    const T *in;
    uint8_t *codes8;
    const szk_k1_params *p;
    uint32_t d0, d1, d2;
    uint64_t plane, vol;
    uint32_t *lh;          // [NARROW_BINS * 4] histogram, [bin][4 copies]
    const uint8_t *s_len;  // [256] code length by t (LDS), nullptr: no bit accounting
    uint16_t *seg_bits;    // [n / 256] out: code bits of every 256-element row segment
    uint64_t *oq_idx;      // this wave's staging queue of value outliers
    OQV *oq_val;
    uint32_t oq_n;
    int lane;
    // fused form (round 4): the rows' bit strings, coded with the context's previous book, leave stage 1 instead of the codes
    const uint32_t *s_enc;  // [256] (code word << 5) | length by stored byte t (LDS)
    uint32_t *stage;        // this wave's LDS stage: TY rows x FUSE_ROW_WORDS, zero between planes
    uint32_t *slot;         // the task's slot of the scratch (the code array's memory)
    uint32_t slot_base;     // ... its first word's index in the scratch
    uint32_t slot_w;        // words of the slot written (wave-uniform, a multiple of 32: whole 128-byte lines)
    uint32_t st_cnt;        // words waiting at the stage's front for the line they belong to to fill up (< 32, wave-uniform)
    uint32_t *seg_base;     // [n / 256] out: index of the first word of every segment's bit string in the scratch
    // sampled book (round 6): the lengths' table is filled when the book arrives (samp_words[SZK_SAMP_READY]); until then a plane's segment sums wait
    const uint32_t *samp_words;
    uint32_t *s_have;       // the workgroup's LDS word: its length table is filled
    uint32_t have_len;      // (wave-uniform) the table is filled
};
// a 256-element row segment is at most 256 x 16 bits = 128 words (small books: code words <= 16 bits); the stage of a wave holds a
// plane's TY rows one behind the other (+ slack for the unconditional emission and the two-word sweep); a task's slot holds its
// segments at their worst, a plane's rows padded to an even number of words
#define FUSE_STAGE_WORDS(TY) ((TY) * 128 + 32 + 8)  // a plane's rows + the words carried over + slack, a multiple of 4
// (rows and planes a task can have: min(TY, d1) x min(MARCH_TZ, d2) — a 2-D array's tasks are one plane deep)
static inline uint32_t fuse_slot_words(uint32_t ty, uint64_t d1, uint64_t d2) {
    return (uint32_t)((d1 < ty ? d1 : ty) * 128u * (d2 < MARCH_TZ ? d2 : MARCH_TZ) + 32u);  // (whole lines; the last one may be padded)
}
template <typename T>
__device__ __forceinline__ void narrow_oq_flush(NarrowCtx<T> &c) {
    // (called by whatever lanes are active; the records are dealt over the ACTIVE lanes, see march_body)
    c.oq_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.oq_n);
    if (c.oq_n == 0) return;
    const szk_k1_params &p = *c.p;
    const unsigned long long act = __ballot(1);
    const uint32_t nact = (uint32_t)__popcll(act), rank = (uint32_t)__popcll(act & ((1ull << c.lane) - 1ull));
    unsigned long long base = 0;
    if (rank == 0) base = atomicAdd((unsigned long long *)p.n_vout, (unsigned long long)c.oq_n);
    const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
    const unsigned long long b0 = ((unsigned long long)bhi << 32) | blo;
    for (uint32_t k = rank; k < c.oq_n; k += nact) {
        const unsigned long long pos = b0 + k;
        if (pos < p.out_cap) {
            p.vout_idx[pos] = c.oq_idx[k];
            if (sizeof(T) == 4) reinterpret_cast<uint32_t *>(p.vout_val)[pos] = (uint32_t)c.oq_val[k];
            else reinterpret_cast<uint64_t *>(p.vout_val)[pos] = c.oq_val[k];
        }
    }
    c.oq_n = 0;
}
// what a row's rare elements need: delta outliers into their list, values that failed the bound check into the wave's
// staging queue, delta outliers counted as symbol 0 of the global histogram. Out of line: the hot loop only branches here.
template <typename T>
__device__ __forceinline__ void narrow_rare(NarrowCtx<T> &c, uint64_t gi, const typename QTraits<T>::UQ (&delta)[4], uint32_t tmask, uint32_t badmask) {
    using Q = typename QTraits<T>::Q;
    using OQV = typename NarrowCtx<T>::OQV;
    const szk_k1_params &p = *c.p;
    constexpr uint32_t OQ = MarchLds<1, false>::OQ;
    {   // (one atomic per wave for the four element positions: see the wide form)
        unsigned long long pd = wave_append_run((uint32_t)__popc(tmask & 15u), p.n_dout);
#pragma unroll
        for (int i = 0; i < 4; i++)
            if ((tmask >> i) & 1u) {
                if (pd < p.out_cap) {
                    p.dout_idx[pd] = gi + i;
                    ((Q *)p.dout_val)[pd] = (Q)delta[i];
                }
                pd++;
            }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const bool is_vout = (badmask >> i) & 1u;
        const unsigned long long vm = __ballot(is_vout);
        if (vm) {
            c.oq_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.oq_n);
            if (c.oq_n + WAVE > OQ) narrow_oq_flush(c);
            if (is_vout) {
                const uint32_t slot = c.oq_n + (uint32_t)__popcll(vm & ((1ull << c.lane) - 1ull));
                c.oq_idx[slot] = gi + i;
                const T raw = c.in[gi + i];
                OQV bits = 0;
                memcpy(&bits, &raw, sizeof(T));
                c.oq_val[slot] = bits;
            }
            c.oq_n += (uint32_t)__popcll(vm);
        }
    }
    {   // code 0 (delta outliers) is one address for the whole grid: one atomic per wave for the four element positions
        const uint32_t cnt = (uint32_t)__popc(tmask & 15u);
        const unsigned long long b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u), any = b0 | b1 | b2;
        if (any && c.lane == __ffsll((long long)any) - 1)
            hist_add_ranged(p.hist, p.range, 0u, (unsigned long long)(__popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2)));
    }
}
// (always inlined, the kernel's parameters handed over piecemeal: a call would put the caller's parameter block on a stack, and a kernel
// with a stack pays for it in every wave)
template <typename T> __device__ __forceinline__ bool samp_take(const T *__restrict__ in, const szk_lattice &latp, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t *words, uint32_t role, uint32_t *s_h);
__device__ __forceinline__ void samp_book(const szk_samp &sp, uint32_t radius, uint8_t *pool);
#define SAMP_POOL_BYTES (SZK_CB_SMALL_SYMS * 28 + 256)
// a worker's wave looks for the sampled book (spin: waits for it) and, once it is there, fills ITS view of the workgroup's length table
// (all four waves write the same values): `put(byte value, length)` stores one entry in the form's layout
template <typename T, typename PUT>
__device__ __forceinline__ bool samp_poll(NarrowCtx<T> &c, bool spin, PUT put) {
    // Wave 0 of the workgroup asks the device-wide word (a load past the L2s: all workers asking would be a billion requests a second to
    // one memory channel — the one the sampling workgroups' atomics go to), fills the workgroup's table and raises the workgroup's LDS
    // word; the other waves watch that. Wave 0 leaves a task only with the book in hand, so the LDS word is raised before it exits.
    volatile uint32_t *s_have = c.s_have;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    if (wv != 0) {
        uint32_t h = (uint32_t)__builtin_amdgcn_readfirstlane((int)*s_have);
        if (!h) {
            if (!spin) return false;
            do {
                __builtin_amdgcn_s_sleep(16);
                h = (uint32_t)__builtin_amdgcn_readfirstlane((int)*s_have);
            } while (!h);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        c.have_len = 1;
        return true;
    }
    uint32_t r = __hip_atomic_load(const_cast<uint32_t *>(c.samp_words) + SZK_SAMP_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
    if (!r) {
        if (!spin) return false;
        do {
            __builtin_amdgcn_s_sleep(32);
            r = __hip_atomic_load(const_cast<uint32_t *>(c.samp_words) + SZK_SAMP_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
        } while (!r);
    }
    const uint32_t w = __hip_atomic_load(const_cast<uint32_t *>(c.samp_words) + SZK_SAMP_LENS + c.lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int j = 0; j < 4; j++) put(4u * (uint32_t)c.lane + j, (w >> (8 * j)) & 0xFFu);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (c.lane == 0) *s_have = 1u;
    c.have_len = 1;
    return true;
}
// one plane's rows as they come back from memory: halo row y0 - 1 (slot 0) and rows y0 .. y0 + TY - 1 (slots 1 .. TY)
template <typename T, int NW, int TY> struct NarrowPlane {
    Quad<T> rq[NW][TY + 1];
    T rl[NW][TY + 1];      // the element left of the brick (one address for the wave; only lane 0's copy is used)
    bool rok[NW][TY + 1];  // the row exists (wave-uniform)
};
// the first nw words of the wave's stage (nw even) to the task's slot at slot_w, two per lane, the stage zeroed behind them. Two
// predicated rounds (256 words, twice a smooth field's plane) so that the number of stores in flight is a constant; more only
// behind a wave-uniform test.
template <typename T>
__device__ __forceinline__ void fuse_sweep(NarrowCtx<T> &c, uint32_t nw) {
    uint32_t *out = c.slot + c.slot_w;
    auto sweep = [&](uint32_t i) {
        const uint2 wv = *reinterpret_cast<const uint2 *>(&c.stage[i]);
        *reinterpret_cast<uint2 *>(&c.stage[i]) = make_uint2(0u, 0u);
#if defined(LAB_FUSE) && (LAB_FUSE & 32)  // (lab: plain stores)
        *reinterpret_cast<unsigned long long *>(&out[i]) = (unsigned long long)wv.x | ((unsigned long long)wv.y << 32);
#else
        __builtin_nontemporal_store((unsigned long long)wv.x | ((unsigned long long)wv.y << 32), reinterpret_cast<unsigned long long *>(&out[i]));
#endif
    };
    const uint32_t i0 = 2u * (uint32_t)c.lane;
    if (i0 < nw) sweep(i0);
    if (i0 + 2u * WAVE < nw) sweep(i0 + 2u * WAVE);
    if (nw > 4u * WAVE)
        for (uint32_t i = i0 + 4u * WAVE; i < nw; i += 2u * WAVE) sweep(i);
}
// Only WHOLE 128-byte lines leave the stage (the slot is line-aligned, slot_w a multiple of 32 words): a plane's ~130 words written
// where the previous plane's ended cut two lines each, and partial-line streaming stores cost the kernel 17 of its 199 us. The words
// behind the last whole line move to the stage's front and wait for the next plane's.
template <typename T>
__device__ __forceinline__ void fuse_flush_lines(NarrowCtx<T> &c) {
#if defined(LAB_FUSE) && (LAB_FUSE & 4)  // (lab: no copy-out)
    const uint32_t nfl = 0;
    c.st_cnt &= 31u;
#else
    const uint32_t nfl = c.st_cnt & ~31u;
#endif
    if (nfl == 0) return;
    fuse_sweep(c, nfl);
    const uint32_t left = c.st_cnt - nfl;
    if (left) {  // (left < 32 <= nfl: source and destination do not overlap; the sweep zeroed the destination)
        if (2u * (uint32_t)c.lane < left) {
            const uint2 wv = *reinterpret_cast<const uint2 *>(&c.stage[nfl + 2u * c.lane]);
            *reinterpret_cast<uint2 *>(&c.stage[nfl + 2u * c.lane]) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2 *>(&c.stage[2u * c.lane]) = wv;
        }
    }
    c.st_cnt = left;
    c.slot_w += nfl;
    __builtin_amdgcn_wave_barrier();
}
#ifndef NARROW_PF
#define NARROW_PF 0  // 1: the next plane's rows are requested before the current plane is worked (two sets of row registers; measured slower: 161 vs 149 us)
#endif
template <typename T, int NDIM, int TY, bool EDGE, bool FUSE = false, bool SAMP = false>
__device__ __forceinline__ void narrow_task(NarrowCtx<T> &c, const Lattice<T> &lat, uint32_t x0, uint32_t y0, uint32_t z0, uint32_t w) {
    static_assert(!SAMP || (!FUSE && NDIM == 3), "the sampled book is the plain one-byte form's, 1-D ... 3-D arrays");
    using B = typename Lattice<T>::B;
    using UQ = typename QTraits<T>::UQ;
    constexpr int NW = NDIM == 4 ? 2 : 1;
    constexpr B CB = Lattice<T>::C;
    const uint32_t d0 = c.d0, d1 = c.d1, d2 = c.d2;
    const int lane = c.lane;
    const uint32_t x = x0 + 4u * (uint32_t)lane;
    const bool xok = EDGE ? x < d0 : true;       // quad granular (d0 % 4 == 0)
    const uint32_t lane_off = xok ? x : 0u;      // (a lane beyond the row reads the row's start and is masked)
#if defined(LAB_MODE) && LAB_MODE == 2
    const bool has_left = false;  // (lab: no left-neighbour loads: wrong at tile seams)
#else
    const bool has_left = x0 > 0;                // wave-uniform, like everything below that is not named "lane"
#endif
    const uint32_t copy = (uint32_t)lane & 3u;

    UQ pp[NW][TY][4];  // d2 of the previous plane
#pragma unroll
    for (int lw = 0; lw < NW; lw++)
#pragma unroll
        for (int yy = 0; yy < TY; yy++)
#pragma unroll
            for (int i = 0; i < 4; i++) pp[lw][yy][i] = 0;

    auto fetch = [&](int zz, NarrowPlane<T, NW, TY> &R) {
        const uint32_t gz = z0 + (uint32_t)zz;
#pragma unroll
        for (int lw = 0; lw < NW; lw++) {
            const bool wok = w >= (uint32_t)lw;
            const T *src = c.in + (uint64_t)(wok ? w - lw : 0) * c.vol + (uint64_t)gz * c.plane;
#pragma unroll
            for (int r = 0; r <= TY; r++) {
                const uint32_t gy = y0 + (uint32_t)r - 1u;  // (r = 0 at y0 = 0 wraps: not below d1)
                R.rok[lw][r] = wok && gy < d1;
                if (R.rok[lw][r]) {
                    const T *row = src + (uint64_t)gy * d0;
                    R.rq[lw][r].load(row + lane_off);
                    if (has_left) R.rl[lw][r] = row[x0 - 1];
                }
            }
        }
    };
    // (fetch_next, fused form: the next plane is requested between this plane's rows and its emission — into R's own registers, free by then)
    NarrowPlane<T, NW, TY> pa;
    auto work = [&](int zz, NarrowPlane<T, NW, TY> &R, bool fetch_next = false) {
        const uint32_t gz = z0 + (uint32_t)zz;
        if constexpr (FUSE) {
            // The previous plane's words leave the stage HERE: behind the wait for this plane's rows (all of them: loads and stores share
            // one in-order counter, and the waits the compiler places for rows requested behind a run-time number of stores are waits
            // for everything — the stores' acknowledgements included, which cost this kernel 25 us when the sweep sat at the plane's
            // end, right in front of that wait) and a plane's worth of arithmetic ahead of the next one.
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            fuse_flush_lines(c);
        }
#if defined(LAB_MODE) && LAB_MODE == 1  // (lab: the kernel's loads and stores alone — what its access pattern gets from the memory system)
        if (zz >= 0) {
#pragma unroll
            for (int r = 1; r <= TY; r++) {
                if (!R.rok[0][r]) continue;
                uint32_t acc = 0;
#pragma unroll
                for (int lw = 0; lw < NW; lw++)
#pragma unroll
                    for (int i = 0; i < 4; i++) acc += (uint32_t)lat.qbits(R.rq[lw][r].get(i)) + (has_left ? (uint32_t)lat.qbits(R.rl[lw][r]) : 0u);
                if (r == 1 && R.rok[0][0]) acc += (uint32_t)lat.qbits(R.rq[0][0].get(0));
                const uint64_t grow = (uint64_t)w * c.vol + (uint64_t)gz * c.plane + (uint64_t)(y0 + r - 1) * d0;
                if (!EDGE || xok) *reinterpret_cast<uint32_t *>(c.codes8 + grow + x) = acc;
            }
        }
        return;
#endif
        UQ pd1[NW][4];  // d1 of the previous row
        UQ delta[4];
        uint32_t bits_rows[(TY + 1) / 2];  // code bits of the plane's rows, two rows per register (16 bits each)
#pragma unroll
        for (int k = 0; k < (TY + 1) / 2; k++) bits_rows[k] = 0;
        uint32_t fp01[TY], fp23[TY];  // FUSE: the lane's four code words of every row, joined in pairs ...
        uint32_t fl[TY];              // ... their total length (0: the row is not coded) | length of the second pair << 8
#pragma unroll
        for (int k = 0; k < TY; k++) {
            fp01[k] = fp23[k] = 0;
            fl[k] = 0;
        }
#pragma unroll
        for (int r = 0; r <= TY; r++) {
            bool bad[4] = {false, false, false, false};
#pragma unroll
            for (int lw = 0; lw < NW; lw++) {
                UQ d1v[4];
                if (R.rok[lw][r]) {
                    B q[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const T v = R.rq[lw][r].get(i);
                        q[i] = lat.qbits(v);
                        if (lw == 0 && r > 0 && zz >= 0) bad[i] = lat.bad(v, q[i]);
                        if (EDGE) q[i] = xok ? q[i] : CB;
                    }
                    const B left0 = has_left ? lat.qbits(R.rl[lw][r]) : CB;
                    UQ pv = (UQ)dpp_wave_shr1(left0, q[3]);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        d1v[i] = (UQ)q[i] - pv;
                        pv = (UQ)q[i];
                    }
                } else {  // a row outside the array: the constant C, whose differences vanish
#pragma unroll
                    for (int i = 0; i < 4; i++) d1v[i] = 0;
                }
                if (r > 0) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const UQ d2v = d1v[i] - pd1[lw][i];
                        const UQ s = d2v - pp[lw][r - 1][i];
                        pp[lw][r - 1][i] = d2v;
                        delta[i] = lw == 0 ? s : (UQ)(delta[i] - s);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; i++) pd1[lw][i] = d1v[i];
            }
            if (r == 0 || zz < 0) continue;   // halo row / halo plane: state only
            if (!R.rok[0][r]) continue;        // beyond the array (EDGE bricks only)
            // ---- codes, histogram, store ----
            const uint64_t grow = (uint64_t)w * c.vol + (uint64_t)gz * c.plane + (uint64_t)(y0 + r - 1) * d0;  // the row's first element
            uint32_t t[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const UQ tq = delta[i] + (UQ)127;
                t[i] = tq <= (UQ)254 ? (uint32_t)tq : 255u;
            }
            const uint32_t tmax = max(max(t[0], t[1]), max(t[2], t[3]));
            bool rare = bad[0] | bad[1] | bad[2] | bad[3] | (tmax == 255u);
            if (EDGE) rare &= xok;
            if (!EDGE || xok) {
#ifdef LAB_ABLATE
                if (!(c.p->dbg & 1u))
#endif
                if (!SAMP)  // (a sampled book needs no histogram)
#pragma unroll
                for (int i = 0; i < 4; i++) atomicAdd(&c.lh[t[i] * 4u + copy], 1u);
                if constexpr (FUSE) {
                    // the four code words of the lane, joined: two pairs in 32 bits (code words <= 16 bits), the pairs in 64
#if defined(LAB_FUSE) && (LAB_FUSE & 1)  // (lab: no table lookups)
                    const uint32_t e0 = (t[0] << 5) | 4u, e1 = (t[1] << 5) | 4u, e2 = (t[2] << 5) | 4u, e3 = (t[3] << 5) | 4u;
#else
                    const uint32_t e0 = c.s_enc[t[0]], e1 = c.s_enc[t[1]], e2 = c.s_enc[t[2]], e3 = c.s_enc[t[3]];
#endif
                    const uint32_t l0 = e0 & 31u, l1 = e1 & 31u, l2 = e2 & 31u, l3 = e3 & 31u;
                    fp01[r - 1] = ((e0 >> 5) << l1) | (e1 >> 5);
                    fp23[r - 1] = ((e2 >> 5) << l3) | (e3 >> 5);
                    const uint32_t l23 = l2 + l3;
                    fl[r - 1] = (l0 + l1 + l23) | (l23 << 8);
                    // (a symbol the book has no code word for has length 0 here and codes nothing: the verdict of the merge's launch —
                    // book_rejected walks this call's complete histogram — finds it missing from the book and voids the call)
                } else {
#ifdef LAB_ABLATE
                if (!(c.p->dbg & 2u))
#endif
                // (streaming store: the codes are next read by another launch; what stays dirty in the L2s is written back at the kernel's end)
#ifdef LAB_PLAIN_STORE
                *reinterpret_cast<uint32_t *>(c.codes8 + grow + x) = t[0] | (t[1] << 8) | (t[2] << 16) | (t[3] << 24);
#else
                __builtin_nontemporal_store(t[0] | (t[1] << 8) | (t[2] << 16) | (t[3] << 24), reinterpret_cast<uint32_t *>(c.codes8 + grow + x));
#endif
                }
            }
            if (SAMP ? c.have_len != 0u : c.s_len != nullptr) {
                uint32_t b4 = (uint32_t)c.s_len[t[0]] + c.s_len[t[1]] + c.s_len[t[2]] + c.s_len[t[3]];
                if (EDGE) b4 = xok ? b4 : 0u;
                bits_rows[(r - 1) >> 1] |= ((r - 1) & 1) ? b4 << 16 : b4;
            }
            if (__builtin_amdgcn_ballot_w64(rare)) {  // some lane has outliers: rare
                uint32_t tmask = 0, badmask = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    tmask |= (uint32_t)(rare && t[i] == 255u) << i;
                    badmask |= (uint32_t)(rare && bad[i]) << i;
                }
                narrow_rare<T>(c, grow + x, delta, tmask, badmask);
            }
        }
        if constexpr (FUSE) {
            // The next plane's rows are requested HERE, in front of this plane's stores: loads and stores return in order on one
            // counter, and a wait for rows requested behind the sweep's stores waits for the stores' acknowledgements too (measured:
            // 45 us of the kernel's 199). The registers the rows land in are free: the row loop above was their last reader.
            if (fetch_next) fetch(zz + 1, R);
            if (zz >= 0) {
                // Bit positions of the lanes' strings inside their rows: one wave scan per PAIR of rows (16 bits each: a row is at most
                // 64 lanes x 64 bits). The rows follow one another word-aligned in the wave's LDS stage (every lane ORs its string in:
                // two ds_or when no lane of the plane holds more than 32 bits — the usual case, decided for the wave — else three), and
                // the plane's words leave for the task's slot in one sweep, two per lane; a lane per row notes length and place.
                uint32_t excl[(TY + 1) / 2], row_bits[TY], row_base[TY + 1];
                uint32_t lmax = 0;
#pragma unroll
                for (int k = 0; k < TY; k += 2) {
                    const uint32_t la = fl[k] & 0xFFu, lb = k + 1 < TY ? fl[k + 1] & 0xFFu : 0u;
                    const uint32_t packed = la | (lb << 16);
                    lmax = max(lmax, max(la, lb));
#if defined(LAB_FUSE) && (LAB_FUSE & 8)  // (lab: no scans)
                    const uint32_t incl = packed * (uint32_t)(lane + 1);
#else
                    const uint32_t incl = wave_incl_scan(packed);
#endif
                    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
                    excl[k / 2] = incl - packed;
                    row_bits[k] = tot & 0xFFFFu;
                    if (k + 1 < TY) row_bits[k + 1] = tot >> 16;
                }
                // (the stage's first st_cnt words are the previous planes' words that did not fill a 128-byte line yet)
                row_base[0] = c.st_cnt;
#pragma unroll
                for (int k = 0; k < TY; k++) row_base[k + 1] = row_base[k] + ((row_bits[k] + 31u) >> 5);
#if defined(LAB_FUSE) && (LAB_FUSE & 2)  // (lab: no emission)
                if (lmax > 200u) {
#else
                if (!__builtin_amdgcn_ballot_w64(lmax > 32u)) {
#endif
#pragma unroll
                    for (int k = 0; k < TY; k++) {
                        const uint32_t len = fl[k] & 0xFFu, l23 = fl[k] >> 8;
                        const uint32_t pos = (k & 1) ? excl[k / 2] >> 16 : excl[k / 2] & 0xFFFFu;
                        const uint32_t q = (fp01[k] << l23) | fp23[k];
                        const uint32_t v = q << ((32u - len) & 31u);  // left-aligned (len 0: q is 0)
                        uint32_t *st = c.stage + row_base[k] + (pos >> 5);
                        const uint32_t sh = pos & 31u;
                        atomicOr(&st[0], v >> sh);
                        atomicOr(&st[1], (uint32_t)(((uint64_t)v << 32) >> sh));
                    }
                }
#if defined(LAB_FUSE) && (LAB_FUSE & 2)
                else if (lmax > 300u) {
#else
                else {
#endif
#pragma unroll
                    for (int k = 0; k < TY; k++) {
                        const uint32_t len = fl[k] & 0xFFu, l23 = fl[k] >> 8;
                        const uint32_t pos = (k & 1) ? excl[k / 2] >> 16 : excl[k / 2] & 0xFFFFu;
                        const uint64_t q = ((uint64_t)fp01[k] << l23) | fp23[k];
                        const uint64_t v = len ? q << (64u - len) : 0ull;
                        uint32_t *st = c.stage + row_base[k] + (pos >> 5);
                        const uint32_t sh = pos & 31u;
                        const uint64_t tv = v >> sh;
                        atomicOr(&st[0], (uint32_t)(tv >> 32));
                        atomicOr(&st[1], (uint32_t)tv);
                        atomicOr(&st[2], (uint32_t)(((uint64_t)(uint32_t)v << 32) >> sh));
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (lane < TY) {  // lane k notes row k (the rows' values are wave-uniform: picked by selects)
                    uint32_t rbits = row_bits[0], rbase = row_base[0];
                    bool rok = R.rok[0][1];
#pragma unroll
                    for (int k = 1; k < TY; k++) {
                        rbits = lane == k ? row_bits[k] : rbits;
                        rbase = lane == k ? row_base[k] : rbase;
                        rok = lane == k ? R.rok[0][k + 1] : rok;
                    }
                    if (rok) {
                        const uint64_t seg = ((uint64_t)w * c.vol + (uint64_t)gz * c.plane + (uint64_t)(y0 + (uint32_t)lane) * d0 + x0) >> 8;
                        c.seg_bits[seg] = (uint16_t)rbits;
                        c.seg_base[seg] = c.slot_base + c.slot_w + rbase;
                    }
                }
                // (the plane's words stay in the stage until the NEXT plane's rows have arrived: fuse_flush_lines, at the top of work)
                c.st_cnt = row_base[TY];
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (!FUSE && (SAMP ? c.have_len != 0u : c.s_len != nullptr) && zz >= 0) {  // the plane's segment sums: one wave reduction per pair of rows
#pragma unroll
            for (int k = 0; k < (TY + 1) / 2; k++) {
                const uint32_t tot = wave_sum(bits_rows[k]);
                const uint32_t ra = y0 + 2u * k, rb = ra + 1u;
                if (lane == 0) {
                    const uint64_t g0 = (uint64_t)w * c.vol + (uint64_t)gz * c.plane + x0;
                    if (ra < d1) c.seg_bits[(g0 + (uint64_t)ra * d0) >> 8] = (uint16_t)(tot & 0xFFFFu);
                    if (2 * k + 1 < TY && rb < d1) c.seg_bits[(g0 + (uint64_t)rb * d0) >> 8] = (uint16_t)(tot >> 16);
                }
            }
        }
    };
    int zz = z0 > 0 ? -1 : 0;
    const int zend = d2 - z0 < (uint32_t)MARCH_TZ ? (int)(d2 - z0) : MARCH_TZ;
    if constexpr (SAMP) {
        // the sampled book (see narrow16_task): the planes coded before it arrived get their segment sums from the codes they stored
        uint8_t *const lt = const_cast<uint8_t *>(c.s_len);
        auto put_len = [&](uint32_t b, uint32_t len) { lt[b] = (uint8_t)len; };
        auto catch_up = [&](int zhi) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            constexpr int PB = 4;
            for (int p0 = 0; p0 < zhi; p0 += PB) {
                uint32_t cw[PB][TY];
#pragma unroll
                for (int j = 0; j < PB; j++) {
                    const uint64_t gp = (uint64_t)(z0 + (uint32_t)(p0 + j)) * c.plane;
#pragma unroll
                    for (int r = 0; r < TY; r++) {
                        const uint32_t ry = y0 + (uint32_t)r;
                        cw[j][r] = 0;
                        if (p0 + j < zhi && ry < d1 && xok)
                            cw[j][r] = __hip_atomic_load(reinterpret_cast<uint32_t *>(c.codes8 + gp + (uint64_t)ry * d0 + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
#pragma unroll
                for (int j = 0; j < PB; j++) {
                    const uint64_t gp = (uint64_t)(z0 + (uint32_t)(p0 + j)) * c.plane;
                    if (p0 + j < zhi)
#pragma unroll
                    for (int k = 0; k < (TY + 1) / 2; k++) {
                        uint32_t packed = 0;
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const uint32_t ry = y0 + 2u * k + h;
                            if (2 * k + h < TY && ry < d1 && xok) {
                                const uint32_t wd = cw[j][(2 * k + h) < TY ? 2 * k + h : 0];
                                const uint32_t b4 = (uint32_t)c.s_len[wd & 0xFFu] + c.s_len[(wd >> 8) & 0xFFu] + c.s_len[(wd >> 16) & 0xFFu] + c.s_len[wd >> 24];
                                packed |= h ? b4 << 16 : b4;
                            }
                        }
                        const uint32_t tot = wave_sum(packed);
                        const uint32_t ra = y0 + 2u * k, rb = ra + 1u;
                        if (lane == 0) {
                            const uint64_t g0 = gp + x0;
                            if (ra < d1) c.seg_bits[(g0 + (uint64_t)ra * d0) >> 8] = (uint16_t)(tot & 0xFFFFu);
                            if (2 * k + 1 < TY && rb < d1) c.seg_bits[(g0 + (uint64_t)rb * d0) >> 8] = (uint16_t)(tot >> 16);
                        }
                    }
                }
            }
        };
        for (; zz < zend; zz++) {
            fetch(zz, pa);
            work(zz, pa);
            if (!c.have_len && zz >= 0 && samp_poll(c, false, put_len)) catch_up(zz + 1);
        }
        if (!c.have_len) {
            samp_poll(c, true, put_len);
            catch_up(zend);
        }
        return;
    }
    if (NARROW_PF) {
        NarrowPlane<T, NW, TY> pb;
        fetch(zz, pa);
        for (;;) {
            if (zz + 1 < zend) fetch(zz + 1, pb);
            work(zz, pa);
            if (++zz >= zend) break;
            if (zz + 1 < zend) fetch(zz + 1, pa);
            work(zz, pb);
            if (++zz >= zend) break;
        }
    } else if (FUSE) {
        fetch(zz, pa);
        for (; zz < zend; zz++) work(zz, pa, zz + 1 < zend);
    } else {
        for (; zz < zend; zz++) {
            fetch(zz, pa);
            work(zz, pa);
        }
    }
}
template <typename T, int NDIM, int TY, bool FUSE = false, bool SAMP = false>
__device__ __forceinline__ void march_narrow(const T *__restrict__ in, uint16_t *__restrict__ codes, const szk_k1_params &p, uint32_t ntasks,
                                             uint32_t *lh, uint8_t *s_len, uint64_t (*s_oq_idx)[MarchLds<1, false>::OQ],
                                             typename NarrowCtx<T>::OQV (*s_oq_val)[MarchLds<1, false>::OQ],
                                             uint32_t *s_fenc = nullptr, uint32_t *s_fstage = nullptr) {
    const Lattice<T> lat(p.lat);
    NarrowCtx<T> c;
    c.in = in;
    c.codes8 = reinterpret_cast<uint8_t *>(codes);
    c.p = &p;
    c.d0 = (uint32_t)p.d[3];
    c.d1 = (uint32_t)p.d[2];
    c.d2 = (uint32_t)p.d[1];
    c.plane = (uint64_t)c.d1 * c.d0;
    c.vol = c.plane * c.d2;
    c.lh = lh;
    c.lane = lane_id();
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    c.oq_idx = s_oq_idx[wv];
    c.oq_val = s_oq_val[wv];
    c.oq_n = 0;
    // bit accounting: the code-length table of the context's previous book, by stored byte; only for rows cut into whole
    // 256-element segments (x extent a multiple of 256: a segment then never straddles two chunks of the packer)
    const bool acct = SAMP || (!FUSE && p.spec_lens != nullptr && c.d0 % MARCH_TX == 0);  // (the sampled form is launched for rows of whole segments only)
    c.s_len = acct ? s_len : nullptr;
    c.seg_bits = p.seg_bits;
    c.s_enc = s_fenc;
    c.stage = nullptr;
    c.slot = nullptr;
    c.slot_w = c.slot_base = c.st_cnt = 0;
    c.seg_base = p.seg_base;
    c.samp_words = p.samp.words;
    c.s_have = lh + NARROW_BINS * 4;
    c.have_len = 0;
    // (sampled form: the launch's first SZK_SAMP_ROLES workgroups take the sample; the workers are numbered behind them)
    const uint32_t bid = SAMP ? blockIdx.x - SZK_SAMP_ROLES : blockIdx.x, grid = SAMP ? gridDim.x - SZK_SAMP_ROLES : gridDim.x;
    for (int i = threadIdx.x; i < NARROW_BINS * 4; i += 256) lh[i] = 0;
    if (threadIdx.x == 0) lh[NARROW_BINS * 4] = 0;
    if constexpr (FUSE) {
        // the previous call's book by stored byte (255 = a listed delta: symbol 0). A book this form cannot use (code words beyond 16
        // bits) becomes a table of zero lengths and raises the flag; a single-symbol book has zero-length code words: nothing is emitted,
        // which is that book's bit stream
        const uint32_t ml = p.fuse_info->max_len, blo = p.fuse_info->sym_min, bcnt = p.fuse_info->sym_count;
        const uint32_t b = threadIdx.x;
        const uint32_t sym = b == 255u ? 0u : b + p.radius - 127u;
        // (outside the book's range the table may hold an older book's entries: the slot is zeroed over the new range only)
        s_fenc[b] = (ml <= 16u && sym >= blo && sym - blo < bcnt) ? p.fuse_enc[sym] : 0u;
        if (ml > 16u && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(p.fuse_flag, 1u);  // (cannot happen: small books are limited to 16 bits)
        for (int i = threadIdx.x; i < 4 * FUSE_STAGE_WORDS(TY); i += 256) s_fstage[i] = 0u;
        c.stage = s_fstage + wv * FUSE_STAGE_WORDS(TY);
        if (blockIdx.x == 0 && threadIdx.x == 0) *p.seg_made = 1u;
    }
    if (acct) {
        const uint32_t b = threadIdx.x;  // 256 threads, 256 byte values; 255 = the delta outliers' symbol 0
        s_len[b] = SAMP ? (uint8_t)0 : p.spec_lens[b == 255u ? 0u : b + p.radius - 127u];
        if (bid == 0 && threadIdx.x == 0) *p.seg_made = 1u;  // (the host assumed this form would run: the packer's book role checks)
    }
    __syncthreads();

    const uint32_t ntx = (c.d0 + MARCH_TX - 1) / MARCH_TX, nty = (c.d1 + TY - 1) / TY, ntz = (c.d2 + MARCH_TZ - 1) / MARCH_TZ;
    // XCD-aware task order (see march_body)
    const uint32_t per_xcd = grid / 8u;
    const uint32_t wg_seq = grid % 8u == 0 && !(p.dbg & 4096u) ? (bid % 8u) * per_xcd + bid / 8u : bid;
    const uint32_t nwaves = grid * 4u;
    for (uint32_t task = wg_seq * 4 + wv; task < ntasks; task += nwaves) {
        uint32_t b = task;
        const uint32_t x0 = (b % ntx) * MARCH_TX;
        b /= ntx;
        const uint32_t y0 = (b % nty) * TY;
        b /= nty;
        const uint32_t z0 = (b % ntz) * MARCH_TZ;
        const uint32_t w = b / ntz;
        if constexpr (FUSE) {
            c.slot_base = task * p.fuse_geom[3];  // (the launcher takes this form only for scratches below 2^32 words)
            c.slot = p.fuse_slots + c.slot_base;
            c.slot_w = 0;
        }
        if (x0 + MARCH_TX <= c.d0 && y0 + TY <= c.d1) narrow_task<T, NDIM, TY, false, FUSE, SAMP>(c, lat, x0, y0, z0, w);
        else narrow_task<T, NDIM, TY, true, FUSE, SAMP>(c, lat, x0, y0, z0, w);
        if constexpr (FUSE) {  // the task's last words (less than a line)
            c.st_cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.st_cnt);
            if (c.st_cnt) {  // (the last plane's words, whole lines and the rest)
                fuse_sweep(c, (c.st_cnt + 1u) & ~1u);
                c.st_cnt = 0;
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    narrow_oq_flush(c);
    if (SAMP) return;  // (no histogram rows)
    __syncthreads();
    // the workgroup's counts go to its private row of hist_partial (k_hist_reduce folds the rows): bin t = symbol t + radius - 127
    uint32_t *row = p.hist_partial + (uint64_t)blockIdx.x * HIST_WIN;
    for (int bnn = threadIdx.x; bnn < HIST_WIN; bnn += 256) {
        const int t = bnn - (HIST_WIN / 2 - 127);
        row[bnn] = (t >= 0 && t < 255) ? lh[t * 4] + lh[t * 4 + 1] + lh[t * 4 + 2] + lh[t * 4 + 3] : 0u;
    }
}

// ------------------------------------------------------------------------------------------------------------
// The 16-bit form of the one-byte kernel (round 5, f32 data, 1-D ... 3-D arrays). Same bricks, same walk and THE SAME BYTES out
// as narrow_task — codes, histogram, segment bit sums, outlier lists — for every array whose lattice values all lie within
// +-Q16_LIM (C2: |q| <= ~700), at 13 instead of 21 vector instructions per element (the one-byte kernel's bound was its issue
// rate: 54 M wave instructions = ~100 us of its 147 at C2). What it does differently:
//   * a lane's four lattice values are kept as TWO registers of packed 16-bit halves, (q0, q2) and (q1, q3) — the low 16 bits
//     of the magic-number patterns, picked by one v_perm each. The x difference is two packed subtractions (the pair (q-1, q1)
//     is one v_alignbit of the pair (q1, q3) and its wave_shr:1 copy); the y and z differences, the +127 and the min(., 255)
//     are packed instructions over two elements each; the four code bytes are (t0, t2) | (t1, t3) << 8: one instruction.
//     The Lorenzo stencil over 8 values of at most 4095 cannot wrap 16 bits;
//   * multiply, add of the magic number, and the bound check's three operations run as packed-f32 instructions (two elements
//     each); the range test of a value (|x / 2eb| <= 2^22, else q = 0) is not made per element at all:
//   * the kernel ASSUMES |q| <= Q16_LIM and finite data. A running maximum of |rint(x / 2eb)| (two v_max3 per row) is tested
//     once per plane, non-finite values fail the bound check (NaN compares false) and are looked at in the rare branch: either
//     raises q16_flag and everything this launch wrote is void — the packer's launch reports it (miss_kind bit 128) and the
//     host repeats the call with narrow_task's kernel (and keeps to it for the following calls). A context takes this form
//     behind a call whose probe saw nothing beyond Q16_LIM / 2 (probe counter [3]);
//   * the code-length table of the speculative bit accounting is laid out like the histogram ([byte][4 copies]): a code's
//     histogram address is also the address of its length.
// ------------------------------------------------------------------------------------------------------------
#define Q16_LIM 4095.0f
typedef float v2f32 __attribute__((ext_vector_type(2)));
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (v2u16)(__builtin_bit_cast(v2u16, a) - __builtin_bit_cast(v2u16, b)));
}
__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (v2u16)(__builtin_bit_cast(v2u16, a) + __builtin_bit_cast(v2u16, b)));
}
__device__ __forceinline__ uint32_t pk_min16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(v2u16, a), __builtin_bit_cast(v2u16, b)));
}
__device__ __forceinline__ uint32_t pk_max16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2u16, a), __builtin_bit_cast(v2u16, b)));
}
// the low halves of two registers side by side: (lo16(a), lo16(b))
__device__ __forceinline__ uint32_t pk_lo16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
// half * 16 + add (the byte address of a code's histogram counter): one v_mad_u32_u16 either way
__device__ __forceinline__ uint32_t mad16_lo(uint32_t p, uint32_t add) {
    uint32_t d;
    asm("v_mad_u32_u16 %0, %1, 16, %2" : "=v"(d) : "v"(p), "v"(add));
    return d;
}
__device__ __forceinline__ uint32_t mad16_hi(uint32_t p, uint32_t add) {
    uint32_t d;
    asm("v_mad_u32_u16 %0, %1, 16, %2 op_sel:[1,0,0,0]" : "=v"(d) : "v"(p), "v"(add));
    return d;
}
#define Q16_LEN_OFF (NARROW_BINS * 16)  // byte offset of the length table behind the histogram: [byte value][4 copies], the length in the word's low byte
template <int TY> struct Q16Plane {
    float4 rq[TY + 1];
    float rl[TY + 1];
    bool rok[TY + 1];
};
template <int TY, bool EDGE, bool SAMP>
__device__ __forceinline__ void narrow16_task(NarrowCtx<float> &c, const Lattice<float> &lat, uint32_t *q16_flag, uint32_t x0, uint32_t y0, uint32_t z0) {
    const uint32_t d0 = c.d0, d1 = c.d1, d2 = c.d2;
    const int lane = c.lane;
    const uint32_t x = x0 + 4u * (uint32_t)lane;
    const bool xok = EDGE ? x < d0 : true;
    const uint32_t lane_off = xok ? x : 0u;
    const bool has_left = x0 > 0;
    const uint32_t c4 = ((uint32_t)lane & 3u) * 4u;
    uint8_t *const lds = reinterpret_cast<uint8_t *>(c.lh);
    const v2f32 recip2 = {lat.recip, lat.recip}, magic2 = {12582912.0f, 12582912.0f}, two_eb2 = {lat.two_eb, lat.two_eb};

    uint32_t ppA[TY], ppB[TY];  // d2 of the previous plane, (x0, x2) and (x1, x3)
#pragma unroll
    for (int yy = 0; yy < TY; yy++) ppA[yy] = ppB[yy] = 0;

    auto fetch = [&](int zz, Q16Plane<TY> &R) {
        const float *src = c.in + (uint64_t)(z0 + (uint32_t)zz) * c.plane;
#pragma unroll
        for (int r = 0; r <= TY; r++) {
            const uint32_t gy = y0 + (uint32_t)r - 1u;  // (r = 0 at y0 = 0 wraps: not below d1)
            R.rok[r] = gy < d1;
            if (R.rok[r]) {
                const float *row = src + (uint64_t)gy * d0;
#ifdef LAB_Q16_NTL  // (lab: streaming loads of the rows)
                {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    const f4v q = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(row + lane_off));
                    R.rq[r] = make_float4(q.x, q.y, q.z, q.w);
                }
#else
                R.rq[r] = *reinterpret_cast<const float4 *>(row + lane_off);
#endif
                if (has_left) R.rl[r] = row[x0 - 1];
            }
        }
    };
    Q16Plane<TY> pa;
    auto work = [&](int zz, Q16Plane<TY> &R) {
        const uint32_t gz = z0 + (uint32_t)zz;
        uint32_t pd1A = 0, pd1B = 0;
        uint32_t bits_rows[(TY + 1) / 2];
#pragma unroll
        for (int k = 0; k < (TY + 1) / 2; k++) bits_rows[k] = 0;
        float qmax = 0.0f;  // max |rint(x / 2eb)| over the plane's coded rows
#pragma unroll
        for (int r = 0; r <= TY; r++) {
            uint32_t d1A = 0, d1B = 0;
            v2f32 s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f}, t01 = magic2, t23 = magic2;
            if (R.rok[r]) {
                const v2f32 v01 = {R.rq[r].x, R.rq[r].y}, v23 = {R.rq[r].z, R.rq[r].w};
                s01 = v01 * recip2;
                s23 = v23 * recip2;
                t01 = s01 + magic2;
                t23 = s23 + magic2;
                uint32_t PA = pk_lo16(__float_as_uint(t01.x), __float_as_uint(t23.x));
                uint32_t PB = pk_lo16(__float_as_uint(t01.y), __float_as_uint(t23.y));
                if (EDGE) {
                    PA = xok ? PA : 0u;
                    PB = xok ? PB : 0u;
                }
                const uint32_t leftp = has_left ? __float_as_uint(R.rl[r] * lat.recip + 12582912.0f) << 16 : 0u;
                const uint32_t pv = (uint32_t)dpp_wave_shr1((int32_t)leftp, (int32_t)PB);
                const uint32_t XA = __builtin_amdgcn_alignbit(PB, pv, 16);  // (q(x0 - 1), q(x1))
                d1A = pk_sub16(PA, XA);
                d1B = pk_sub16(PB, PA);
            }
            uint32_t sA = 0, sB = 0;
            if (r > 0) {
                const uint32_t d2A = pk_sub16(d1A, pd1A), d2B = pk_sub16(d1B, pd1B);
                sA = pk_sub16(d2A, ppA[r - 1]);
                sB = pk_sub16(d2B, ppB[r - 1]);
                ppA[r - 1] = d2A;
                ppB[r - 1] = d2B;
            }
            pd1A = d1A;
            pd1B = d1B;
            if (r == 0 || zz < 0) continue;  // halo row / halo plane: state only
            if (!R.rok[r]) continue;         // beyond the array (EDGE bricks only)
            // ---- codes, histogram, store ----
            const uint64_t grow = (uint64_t)gz * c.plane + (uint64_t)(y0 + r - 1) * d0;
            const uint32_t tqA = pk_add16(sA, 0x007F007Fu), tqB = pk_add16(sB, 0x007F007Fu);
            const uint32_t tA = pk_min16(tqA, 0x00FF00FFu), tB = pk_min16(tqB, 0x00FF00FFu);  // (t0, t2), (t1, t3)
            const uint32_t tm = pk_max16(tA, tB);
            // the bound check, the reference's acceptance test on the lattice reconstruction: Lattice<float>::bad's expression
            const v2f32 r01 = t01 - magic2, r23 = t23 - magic2;  // rint(s) as floats
            const v2f32 v01 = {R.rq[r].x, R.rq[r].y}, v23 = {R.rq[r].z, R.rq[r].w};
            const v2f32 e01 = r01 * two_eb2 - v01, e23 = r23 * two_eb2 - v23;
            const bool b0 = !(fabsf(e01.x) <= lat.eb_lo), b1 = !(fabsf(e01.y) <= lat.eb_lo), b2 = !(fabsf(e23.x) <= lat.eb_lo), b3 = !(fabsf(e23.y) <= lat.eb_lo);
            qmax = fmaxf(fmaxf(qmax, fabsf(r01.x)), fabsf(r01.y));
            qmax = fmaxf(fmaxf(qmax, fabsf(r23.x)), fabsf(r23.y));
            bool rare = b0 | b1 | b2 | b3 | ((tm & 0xFFFFu) == 255u) | (tm >= 0x00FF0000u);
            if (EDGE) rare &= xok;
            const uint32_t a0 = mad16_lo(tA, c4), a1 = mad16_lo(tB, c4), a2 = mad16_hi(tA, c4), a3 = mad16_hi(tB, c4);
            if (!EDGE || xok) {
                if (!SAMP) {  // (a sampled book needs no histogram)
                    atomicAdd(reinterpret_cast<uint32_t *>(lds + a0), 1u);
                    atomicAdd(reinterpret_cast<uint32_t *>(lds + a1), 1u);
                    atomicAdd(reinterpret_cast<uint32_t *>(lds + a2), 1u);
                    atomicAdd(reinterpret_cast<uint32_t *>(lds + a3), 1u);
                }
#if defined(LAB_ST) && LAB_ST == 1  // (lab: write-through stores — nothing of the code array dirty in the L2s at the kernel's end)
                __hip_atomic_store(reinterpret_cast<uint32_t *>(c.codes8 + grow + x), tA | (tB << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#elif defined(LAB_ST) && LAB_ST == 2
                *reinterpret_cast<uint32_t *>(c.codes8 + grow + x) = tA | (tB << 8);
#elif defined(LAB_ST) && LAB_ST == 3
                __hip_atomic_store(reinterpret_cast<uint32_t *>(c.codes8 + grow + x), tA | (tB << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                if (c.p->dbg & 8388608u) *reinterpret_cast<uint32_t *>(c.codes8 + grow + x) = tA | (tB << 8);  // (lab: plain stores — the codes stay in the caches for the packer)
                else __builtin_nontemporal_store(tA | (tB << 8), reinterpret_cast<uint32_t *>(c.codes8 + grow + x));
#endif
            }
            if (SAMP ? c.have_len != 0u : c.s_len != nullptr) {
                uint32_t b4 = (uint32_t)lds[Q16_LEN_OFF + a0] + lds[Q16_LEN_OFF + a1] + lds[Q16_LEN_OFF + a2] + lds[Q16_LEN_OFF + a3];
                if (EDGE) b4 = xok ? b4 : 0u;
                bits_rows[(r - 1) >> 1] |= ((r - 1) & 1) ? b4 << 16 : b4;
            }
            if (__builtin_amdgcn_ballot_w64(rare)) {  // some lane has outliers (or values this form does not take): rare
                const bool bad[4] = {b0, b1, b2, b3};
                const float sv[4] = {s01.x, s01.y, s23.x, s23.y};
                const uint32_t t[4] = {tA & 0xFFFFu, tB & 0xFFFFu, tA >> 16, tB >> 16};
                const uint32_t delta[4] = {(uint32_t)(int32_t)(int16_t)(sA & 0xFFFFu), (uint32_t)(int32_t)(int16_t)(sB & 0xFFFFu),
                                           (uint32_t)(int32_t)(int16_t)(sA >> 16), (uint32_t)(int32_t)(int16_t)(sB >> 16)};
                uint32_t tmask = 0, badmask = 0;
                bool alien = false;  // beyond the lattice, Inf, NaN: the one-byte kernel gives them q = 0, this form has no such test
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    tmask |= (uint32_t)(rare && t[i] == 255u) << i;
                    badmask |= (uint32_t)(rare && bad[i]) << i;
                    alien |= rare && bad[i] && !(fabsf(sv[i]) <= 4194304.0f);
                }
                if (alien) atomicOr(q16_flag, 1u);
                narrow_rare<float>(c, grow + x, delta, tmask, badmask);
            }
        }
        if (zz >= 0) {
            bool big = !(qmax <= Q16_LIM);
            if (EDGE) big &= xok;
            if (__builtin_amdgcn_ballot_w64(big)) {
                if (big) atomicOr(q16_flag, 1u);
            }
        }
        if ((SAMP ? c.have_len != 0u : c.s_len != nullptr) && zz >= 0) {  // the plane's segment sums: one wave reduction per pair of rows
#pragma unroll
            for (int k = 0; k < (TY + 1) / 2; k++) {
                const uint32_t tot = wave_sum(bits_rows[k]);
                const uint32_t ra = y0 + 2u * k, rb = ra + 1u;
                if (lane == 0) {
                    const uint64_t g0 = (uint64_t)gz * c.plane + x0;
                    if (ra < d1) c.seg_bits[(g0 + (uint64_t)ra * d0) >> 8] = (uint16_t)(tot & 0xFFFFu);
                    if (2 * k + 1 < TY && rb < d1) c.seg_bits[(g0 + (uint64_t)rb * d0) >> 8] = (uint16_t)(tot >> 16);
                }
            }
        }
    };
    int zz = z0 > 0 ? -1 : 0;
    const int zend = d2 - z0 < (uint32_t)MARCH_TZ ? (int)(d2 - z0) : MARCH_TZ;
    // sampled book: the planes of this task coded before the book arrived get their segment sums from the codes they stored (read back
    // past the L1: the wave's own streaming stores, acknowledged by the L2 first)
    auto put_len = [&](uint32_t b, uint32_t len) {
        uint32_t *lt = c.lh + NARROW_BINS * 4 + b * 4u;
        lt[0] = lt[1] = lt[2] = lt[3] = len;
    };
    auto catch_up = [&](int zhi) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        constexpr int PB = 4;  // planes per round: their rows' code words are requested together (a load past the L1 — the wave's own stores are in the XCD's L2 — is a round trip to the L2)
        for (int p0 = 0; p0 < zhi; p0 += PB) {
            uint32_t cw[PB][TY];
#pragma unroll
            for (int j = 0; j < PB; j++) {
                const uint64_t gp = (uint64_t)(z0 + (uint32_t)(p0 + j)) * c.plane;
#pragma unroll
                for (int r = 0; r < TY; r++) {
                    const uint32_t ry = y0 + (uint32_t)r;
                    cw[j][r] = 0x80808080u;  // (a row that is not there: its bytes are not looked up)
                    if (p0 + j < zhi && ry < d1 && xok)
                        cw[j][r] = __hip_atomic_load(reinterpret_cast<uint32_t *>(c.codes8 + gp + (uint64_t)ry * d0 + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
#pragma unroll
            for (int j = 0; j < PB; j++) {
                const uint64_t gp = (uint64_t)(z0 + (uint32_t)(p0 + j)) * c.plane;
                if (p0 + j < zhi)
#pragma unroll
                for (int k = 0; k < (TY + 1) / 2; k++) {
                    uint32_t packed = 0;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t ry = y0 + 2u * k + h;
                        if (2 * k + h < TY && ry < d1 && xok) {
                            const uint32_t w = cw[j][(2 * k + h) < TY ? 2 * k + h : 0];
                            const uint32_t b4 = (uint32_t)lds[Q16_LEN_OFF + (w & 0xFFu) * 16u + c4] + lds[Q16_LEN_OFF + ((w >> 8) & 0xFFu) * 16u + c4] +
                                                lds[Q16_LEN_OFF + ((w >> 16) & 0xFFu) * 16u + c4] + lds[Q16_LEN_OFF + (w >> 24) * 16u + c4];
                            packed |= h ? b4 << 16 : b4;
                        }
                    }
                    const uint32_t tot = wave_sum(packed);
                    const uint32_t ra = y0 + 2u * k, rb = ra + 1u;
                    if (lane == 0) {
                        const uint64_t g0 = gp + x0;
                        if (ra < d1) c.seg_bits[(g0 + (uint64_t)ra * d0) >> 8] = (uint16_t)(tot & 0xFFFFu);
                        if (2 * k + 1 < TY && rb < d1) c.seg_bits[(g0 + (uint64_t)rb * d0) >> 8] = (uint16_t)(tot >> 16);
                    }
                }
            }
        }
    };
    if (SAMP) {
        // (the look for the book and the catching-up sit BEHIND a plane's work: the plane's row registers are free then)
        for (; zz < zend; zz++) {
            fetch(zz, pa);
            work(zz, pa);
            if (!c.have_len && zz >= 0 && samp_poll(c, false, put_len)) catch_up(zz + 1);
        }
        if (!c.have_len) {  // (the book took longer than this task: wait for it)
            samp_poll(c, true, put_len);
            catch_up(zend);
        }
        return;
    }
#if defined(LAB_Q16PF) && LAB_Q16PF == 1
    // (lab: the next plane's rows are requested before the current plane is worked — two sets of row registers)
    Q16Plane<TY> pb;
    fetch(zz, pa);
    for (;;) {
        if (zz + 1 < zend) fetch(zz + 1, pb);
        work(zz, pa);
        if (++zz >= zend) break;
        if (zz + 1 < zend) fetch(zz + 1, pa);
        work(zz, pb);
        if (++zz >= zend) break;
    }
#else
    for (; zz < zend; zz++) {
        fetch(zz, pa);
        work(zz, pa);
    }
#endif
}
template <int TY, bool SAMP>
__device__ __forceinline__ void march_narrow16(const float *__restrict__ in, uint16_t *__restrict__ codes, const szk_k1_params &p, uint32_t ntasks,
                                               uint32_t *lh, uint64_t (*s_oq_idx)[MarchLds<1, false>::OQ], uint32_t (*s_oq_val)[MarchLds<1, false>::OQ]) {
    const Lattice<float> lat(p.lat);
    NarrowCtx<float> c;
    c.in = in;
    c.codes8 = reinterpret_cast<uint8_t *>(codes);
    c.p = &p;
    c.d0 = (uint32_t)p.d[3];
    c.d1 = (uint32_t)p.d[2];
    c.d2 = (uint32_t)p.d[1];
    c.plane = (uint64_t)c.d1 * c.d0;
    c.vol = c.plane * c.d2;
    c.lh = lh;
    c.lane = lane_id();
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    c.oq_idx = s_oq_idx[wv];
    c.oq_val = s_oq_val[wv];
    c.oq_n = 0;
    const bool acct = SAMP || (p.spec_lens != nullptr && c.d0 % MARCH_TX == 0);  // (the sampled form is launched for rows of whole segments only)
    c.s_len = acct ? reinterpret_cast<const uint8_t *>(lh) + Q16_LEN_OFF : nullptr;
    c.seg_bits = p.seg_bits;
    c.s_enc = nullptr;
    c.stage = nullptr;
    c.slot = nullptr;
    c.slot_w = c.slot_base = c.st_cnt = 0;
    c.seg_base = p.seg_base;
    c.samp_words = p.samp.words;
    c.s_have = lh + NARROW_BINS * 8;
    c.have_len = 0;
    // (sampled form: the launch's first SZK_SAMP_ROLES workgroups take the sample; the workers are numbered behind them)
    const uint32_t bid = SAMP ? blockIdx.x - SZK_SAMP_ROLES : blockIdx.x, grid = SAMP ? gridDim.x - SZK_SAMP_ROLES : gridDim.x;
    for (int i = threadIdx.x; i < NARROW_BINS * 4; i += 256) lh[i] = 0;
    if (threadIdx.x == 0) lh[NARROW_BINS * 8] = 0;  // (sampled form: the workgroup's "the length table is filled" word)
    {
        const uint32_t b = threadIdx.x;  // 256 threads, 256 byte values; 255 = the delta outliers' symbol 0
        const uint32_t len = (acct && !SAMP) ? p.spec_lens[b == 255u ? 0u : b + p.radius - 127u] : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) lh[NARROW_BINS * 4 + b * 4 + k] = len;
        if (acct && bid == 0 && threadIdx.x == 0) *p.seg_made = 1u;
    }
    __syncthreads();
    const uint32_t ntx = (c.d0 + MARCH_TX - 1) / MARCH_TX, nty = (c.d1 + TY - 1) / TY;
    const uint32_t per_xcd = grid / 8u;
    const uint32_t wg_seq = grid % 8u == 0 && !(p.dbg & 4096u) ? (bid % 8u) * per_xcd + bid / 8u : bid;
    const uint32_t nwaves = grid * 4u;
    for (uint32_t task = wg_seq * 4 + wv; task < ntasks; task += nwaves) {
        uint32_t b = task;
        const uint32_t x0 = (b % ntx) * MARCH_TX;
        b /= ntx;
        const uint32_t y0 = (b % nty) * TY;
        const uint32_t z0 = (b / nty) * MARCH_TZ;
        if (x0 + MARCH_TX <= c.d0 && y0 + TY <= c.d1) narrow16_task<TY, false, SAMP>(c, lat, p.q16_flag, x0, y0, z0);
        else narrow16_task<TY, true, SAMP>(c, lat, p.q16_flag, x0, y0, z0);
    }
    narrow_oq_flush(c);
    if (SAMP) return;  // (no histogram rows)
    __syncthreads();
    uint32_t *row = p.hist_partial + (uint64_t)blockIdx.x * HIST_WIN;
    for (int bnn = threadIdx.x; bnn < HIST_WIN; bnn += 256) {
        const int t = bnn - (HIST_WIN / 2 - 127);
        row[bnn] = (t >= 0 && t < 255) ? lh[t * 4] + lh[t * 4 + 1] + lh[t * 4 + 2] + lh[t * 4 + 3] : 0u;
    }
}

// LDS histogram. One-byte codes (narrow deltas): 1024 bins x 4 copies around the radius; two-byte codes (deltas of
// hundreds or thousands of lattice steps, e.g. C4's 1e-6 on f64): MARCH_WIDE_WIN bins x 1 copy — with the narrow window
// nearly every element of such a field would fall through to a global atomic. Last word = overflow bin (never flushed).
// Per-wave staging of value outliers (NaN / Inf / fill values can be percents of a field): records collect in LDS and go
// to the global list in batches, one global atomic per batch instead of one per wave instruction.
template <typename T, int NDIM, int TY, int MODE = 0, bool WIN16 = false>
__global__ __launch_bounds__(256) void k_lorenzo_quant_march(const T *__restrict__ in, uint16_t *__restrict__ codes,
                                                             szk_k1_params p, uint32_t ntasks, uint32_t nrows) {
    using L = MarchLds<MODE, WIN16>;
    using OQV = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;
    __shared__ uint32_t lh[L::LH_WORDS + 4];
    __shared__ uint64_t s_oq_idx[4][L::OQ];
    __shared__ OQV s_oq_val[4][L::OQ];
    if constexpr (MODE == 1) {
        __shared__ uint8_t s_len[256];
        if (!szk_is_narrow(p.mode)) return;
        march_narrow<T, NDIM, TY>(in, codes, p, ntasks, lh, s_len, reinterpret_cast<uint64_t(*)[MarchLds<1, false>::OQ]>(s_oq_idx),
                                  reinterpret_cast<OQV(*)[MarchLds<1, false>::OQ]>(s_oq_val));
        return;
    } else {
        march_body<T, NDIM, TY, MODE, WIN16>(in, codes, p, ntasks, nrows, lh, s_oq_idx, s_oq_val);
    }
}
// Launched ALONE when the context's previous call chose one-byte codes: it ASSUMES them for this call too. The probe that decides
// the width (a pure function of the data) runs inside this launch, as the first thing every workgroup does; nobody here
// waits for its outcome — the encoder's kernels read it, and when it says two bytes after all everything this launch wrote is
// void: the packer's launch reports it (szk_state::miss_kind bit 32) and the host repeats the call in the two-launch form.
// (Round 2 kept a two-byte body in this kernel behind the probe's decision: 106 VGPRs instead of 88, and the probe's launch
// and its dependency in front of every call.)
#ifdef LAB_WAVES
#define MARCH3_ATTR __attribute__((amdgpu_waves_per_eu(LAB_WAVES, LAB_WAVES)))
#else
#define MARCH3_ATTR
#endif
template <typename T, int NDIM, int TY, bool SAMP = false>
__global__ __launch_bounds__(256) MARCH3_ATTR void k_lorenzo_quant_march3(const T *__restrict__ in, uint16_t *__restrict__ codes,
                                                              szk_k1_params p, uint32_t ntasks, uint32_t nrows) {
    using L = MarchLds<1, false>;
    using OQV = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;
    constexpr int LH_WORDS = SAMP && (SAMP_POOL_BYTES + 3) / 4 > NARROW_BINS * 4 + 4 ? (SAMP_POOL_BYTES + 3) / 4 : NARROW_BINS * 4 + 4;
    __shared__ __align__(16) uint32_t lh[LH_WORDS];
    __shared__ uint64_t s_oq_idx[4][L::OQ];
    __shared__ OQV s_oq_val[4][L::OQ];
    __shared__ uint8_t s_len[256];
    __shared__ uint32_t s_p[4];
    if constexpr (SAMP) {  // the sampled book inside the launch (see k_lorenzo_quant_march3q)
        static_assert(NDIM == 3, "1-D ... 3-D arrays");
        if (blockIdx.x < SZK_SAMP_ROLES) {
            if (blockIdx.x == 0 && threadIdx.x == 0) p.samp.info->ts[9] = wall_clock64();
            if (samp_take<T>(in, p.lat, (uint32_t)p.d[3], (uint32_t)p.d[2], (uint32_t)p.d[1], p.samp.words, blockIdx.x, lh)) samp_book(p.samp, p.radius, reinterpret_cast<uint8_t *>(lh));
            __syncthreads();
            probe_body<T, NDIM>(in, p, p.mode.n_total, p.mode.probe_big, s_p);
            return;
        }
    }
    probe_body<T, NDIM>(in, p, p.mode.n_total, p.mode.probe_big, s_p);
    march_narrow<T, NDIM, TY, false, SAMP>(in, codes, p, ntasks, lh, s_len, s_oq_idx, s_oq_val);
}
// The 16-bit form of the one-launch kernel (round 5, narrow16_task): f32 data whose lattice values the previous call's probe found
// within +-Q16_LIM / 2. It assumes one-byte codes like the form above AND lattice values within +-Q16_LIM; a value beyond that
// (or not finite) raises q16_flag, the packer's launch reports it (miss_kind bit 128) and the host repeats the call above.
template <int TY, bool SAMP>
__global__ __launch_bounds__(256) void k_lorenzo_quant_march3q(const float *__restrict__ in, uint16_t *__restrict__ codes,
                                                               szk_k1_params p, uint32_t ntasks, uint32_t nrows) {
    using L = MarchLds<1, false>;
    __shared__ uint32_t lh[NARROW_BINS * 8 + 4];  // histogram [byte][4 copies], behind it the code lengths in the same layout
    __shared__ uint64_t s_oq_idx[4][L::OQ];
    __shared__ uint32_t s_oq_val[4][L::OQ];
    __shared__ uint32_t s_p[4];
    if constexpr (SAMP) {
        // the sampled book: the launch's first workgroups take the sample (before anything else: every microsecond the book comes later is a
        // plane the workers code without it), the last of them to finish builds the book (the histogram's memory holds the sample's counts,
        // then the book's scratch); their share of the probe comes afterwards
        static_assert(sizeof(lh) >= SAMP_POOL_BYTES && sizeof(lh) >= 4096, "the book's scratch fits the histogram");
        if (blockIdx.x < SZK_SAMP_ROLES) {
            if (blockIdx.x == 0 && threadIdx.x == 0) p.samp.info->ts[9] = wall_clock64();  // (tools: when the sampling began)
            if (samp_take<float>(in, p.lat, (uint32_t)p.d[3], (uint32_t)p.d[2], (uint32_t)p.d[1], p.samp.words, blockIdx.x, lh)) samp_book(p.samp, p.radius, reinterpret_cast<uint8_t *>(lh));
            __syncthreads();
            probe_body<float, 3>(in, p, p.mode.n_total, p.mode.probe_big, s_p);
            return;
        }
    }
    probe_body<float, 3>(in, p, p.mode.n_total, p.mode.probe_big, s_p);
    march_narrow16<TY, SAMP>(in, codes, p, ntasks, lh, s_oq_idx, s_oq_val);
}
// The FUSED form of the one-launch kernel (round 4): a context whose previous call left a small code book codes with THAT book
// inside stage 1 — the rows' bit strings leave the kernel instead of one byte per element (4 + 0.5 B/elem instead of 4 + 1, and
// no second trip over the codes: the encoder that follows only moves the strings to their places, k_merge). A lane looks its
// four symbols of a row up in a 256-entry LDS table, joins them, the wave scans the lengths (two rows per scan), the strings are
// OR-ed into an LDS stage and the row's words are copied to the task's slot of the scratch (the code array's memory: a task's
// 64 segments, word-aligned, one behind the other; seg_bits / seg_base note length and place). The verdict on the book is the
// packer's as before (book_rejected: this call's book is built from this call's histogram beside the merge); a miss — or a
// symbol the book has no code word for, fuse_flag — repeats the whole call in the two-pass form.
template <typename T, int NDIM, int TY>
__global__ __launch_bounds__(256) void k_lorenzo_quant_march3f(const T *__restrict__ in, uint16_t *__restrict__ codes,
                                                               szk_k1_params p, uint32_t ntasks, uint32_t nrows) {
    using L = MarchLds<1, false>;
    using OQV = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;
    __shared__ uint32_t lh[NARROW_BINS * 4 + 4];
    __shared__ uint64_t s_oq_idx[4][L::OQ];
    __shared__ OQV s_oq_val[4][L::OQ];
    __shared__ uint32_t s_p[4];
    __shared__ uint32_t s_fenc[256];
    __shared__ __align__(16) uint32_t s_fstage[4 * FUSE_STAGE_WORDS(TY)];
    probe_body<T, NDIM>(in, p, p.mode.n_total, p.mode.probe_big, s_p);
    march_narrow<T, NDIM, TY, true>(in, codes, p, ntasks, lh, nullptr, s_oq_idx, s_oq_val, s_fenc, s_fstage);
}

// folds the per-workgroup histogram rows into hist[win_lo + bin]: block (bx, by) sums rows by, by + gridDim.y, ... of
// 256 bins and adds its partial sum with one 64-bit atomic per non-empty bin (at most gridDim.y atomics per address)
__global__ __launch_bounds__(256) void k_hist_reduce(const uint32_t *__restrict__ partial, uint32_t nrows, int win_lo,
                                                     uint64_t *__restrict__ hist, uint32_t *range) {
    const int bin = blockIdx.x * 256 + threadIdx.x;
    if (bin >= HIST_WIN) return;
    uint64_t s = 0;
    for (uint32_t r = blockIdx.y; r < nrows; r += gridDim.y) s += partial[(uint64_t)r * HIST_WIN + bin];
    const int sym = win_lo + bin;
    if (s && sym >= 0 && sym < (int)SZH_HIST_BINS) hist_add_ranged(hist, range, (uint32_t)sym, (unsigned long long)s);
}

// ------------------------------------------------------------------------------------------------------------
// K5: canonical length-limited Huffman code book. One launch of 3 workgroups x 1024 threads:
//   block 0   code book.  Alphabets up to CB_LDS_SYMS symbols (every smooth field) run on 256 threads entirely in
//             LDS with the two-queue merge executed by one wave out of registers; wider alphabets (tight bounds, the
//             interpolation predictor's coarse levels) use all 1024 threads: tile bitonic sort, a round-parallel
//             merge (every round pairs ALL items below the smallest possible new node), chunked code assignment.
//   block 1/2 deterministic order of the two outlier lists.
// Code lengths are limited to 16 bits for alphabets up to 512 symbols (the packer then joins four code words per
// 64-bit register; the loss is < 0.01 bit/symbol) and to SZH_MAX_LEN = 24 bits otherwise (two per register).
// ------------------------------------------------------------------------------------------------------------
#define CB_THREADS 256       // threads of the small-alphabet path
#define CB_LAUNCH 1024       // threads per workgroup of the launch
#define CB_LDS_SYMS 2048     // capacity of the small-alphabet path's LDS arrays
#define CB_SMALL_SYMS SZK_CB_SMALL_SYMS    // alphabets up to this size take the small path (serial wave merge: ~0.1 us per symbol);
                             // beyond it the round-parallel merge of the wide path wins (37 us for 2000 symbols)
#define CB_POOL_BYTES 131072 // LDS pool, carved per phase
#define CB_SHORT_SYMS 512    // alphabets up to this size are limited to 16-bit code words
#define ENC_WIN 4096         // symbols of the encode table the packers cache in LDS (window around the most frequent symbol)

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

// ---- record sort: keys (u64, ascending) + optional value, global memory, LDS tiles ----------------------------
// Bitonic network in its "flip" form: every compare-exchange is ascending (min to the lower index), so virtual
// padding with +infinity at positions >= n never moves and is never stored.
struct RecView {
    uint64_t *k;
    uint8_t *v;
    bool has_val, v32;
};
__device__ __forceinline__ uint64_t rec_ldv(const RecView &r, uint64_t i) {
    return r.v32 ? (uint64_t) reinterpret_cast<const uint32_t *>(r.v)[i] : reinterpret_cast<const uint64_t *>(r.v)[i];
}
__device__ __forceinline__ void rec_stv(const RecView &r, uint64_t i, uint64_t x) {
    if (r.v32) reinterpret_cast<uint32_t *>(r.v)[i] = (uint32_t)x;
    else reinterpret_cast<uint64_t *>(r.v)[i] = x;
}
// one global compare-exchange step over all pairs; flip: partner mirrors inside the block of size k, else i + j
__device__ void rec_global_step(const RecView &r, uint32_t n, uint32_t np2, uint32_t k, uint32_t j, bool flip) {
    const uint32_t half = flip ? k >> 1 : j;
    for (uint32_t pr = threadIdx.x; pr < np2 / 2; pr += blockDim.x) {
        const uint32_t off = pr & (half - 1u);  // half is a power of two
        const uint32_t i = ((pr - off) << 1) + off;
        const uint32_t x = flip ? i + 2 * half - 1 - 2 * off : i + half;
        if (x < n) {  // i < x; a partner in the padding never swaps
            const uint64_t a = r.k[i], b = r.k[x];
            if (a > b) {
                r.k[i] = b;
                r.k[x] = a;
                if (r.has_val) {
                    const uint64_t va = rec_ldv(r, i), vb = rec_ldv(r, x);
                    rec_stv(r, i, vb);
                    rec_stv(r, x, va);
                }
            }
        }
    }
    __syncthreads();
}
// stages k_lo .. k_hi inside the LDS tile [base, base + T); if clean_from != 0 only the half-cleaners
// j = clean_from .. 1 of one stage are run (the tail of a stage whose wide steps ran in global memory)
__device__ void rec_tile_pass(const RecView &r, uint32_t n, uint32_t base, uint32_t T, uint32_t k_lo, uint32_t k_hi,
                              uint32_t clean_from, uint64_t *sk, uint64_t *sv) {
    for (uint32_t i = threadIdx.x; i < T; i += blockDim.x) {
        const uint32_t g = base + i;
        sk[i] = g < n ? r.k[g] : ~0ull;
        if (r.has_val) sv[i] = g < n ? rec_ldv(r, g) : 0ull;
    }
    __syncthreads();
    auto step = [&](uint32_t half, bool flip) {
        for (uint32_t pr = threadIdx.x; pr < T / 2; pr += blockDim.x) {
            const uint32_t off = pr & (half - 1u);  // half is a power of two
            const uint32_t i = ((pr - off) << 1) + off;
            const uint32_t x = flip ? i + 2 * half - 1 - 2 * off : i + half;
            const uint64_t a = sk[i], b = sk[x];
            if (a > b) {
                sk[i] = b;
                sk[x] = a;
                if (r.has_val) {
                    const uint64_t va = sv[i];
                    sv[i] = sv[x];
                    sv[x] = va;
                }
            }
        }
        __syncthreads();
    };
    if (clean_from) {
        for (uint32_t j = clean_from; j > 0; j >>= 1) step(j, false);
    } else {
        for (uint32_t k = k_lo; k <= k_hi; k <<= 1) {
            step(k >> 1, true);
            for (uint32_t j = k >> 2; j > 0; j >>= 1) step(j, false);
        }
    }
    for (uint32_t i = threadIdx.x; i < T; i += blockDim.x) {
        const uint32_t g = base + i;
        if (g < n) {
            r.k[g] = sk[i];
            if (r.has_val) rec_stv(r, g, sv[i]);
        }
    }
    __syncthreads();
}
// Full ascending sort of one LDS tile, register-blocked: every thread holds 16 elements whose indices differ in a 4-bit
// window [lo, lo + 4) of the index, so up to four compare-exchange distances are done per LDS round trip (29 round trips for
// 16384 keys instead of 105 steps: 143 us -> ~25 us). Standard bitonic network (direction = bit `stage` of the index; the
// tile is padded with +infinity keys in LDS). Indices are XOR-swizzled (bits 4..7 into bits 0..3): conflict-free for every window.
__device__ __forceinline__ uint32_t tile_sw(uint32_t i) { return i ^ ((i >> 4) & 15u); }
template <bool HAS_VAL, uint32_t LOGE>  // (records with values: 8 per thread, to stay within 128 VGPRs at 1024 threads)
__device__ void rec_tile_sort_fast(const RecView &r, uint32_t n, uint32_t base, uint32_t T, uint64_t *sk, uint64_t *sv) {
    constexpr uint32_t E = 1u << LOGE;
    const uint32_t tid = threadIdx.x, NA = T >> LOGE, logT = 31u - (uint32_t)__clz((int)T);
    uint64_t kv[E], vv[HAS_VAL ? E : 1];
    // (clamped unconditional loads, all in flight at once: a load under a condition makes hipcc wait for each one)
#pragma unroll
    for (uint32_t c = 0; c < E; c++) {
        const uint32_t g = base + tid + c * blockDim.x, gc = g < n ? g : n - 1;
        kv[c] = r.k[gc];
        if (HAS_VAL) vv[c] = rec_ldv(r, gc);
    }
#pragma unroll
    for (uint32_t c = 0; c < E; c++) {
        const uint32_t i = tid + c * blockDim.x;
        if (i < T) {
            sk[tile_sw(i)] = base + i < n ? kv[c] : ~0ull;
            if (HAS_VAL) sv[tile_sw(i)] = vv[c];
        }
    }
    __syncthreads();
    // one round: window lo, stages s0..s1 (s1 > s0 only for the first round, lo = 0); bits hi-1 .. lo of stage s
    auto round = [&](uint32_t lo, uint32_t s0, uint32_t s1, uint32_t hi) {
        if (tid < NA) {
            const uint32_t b0 = ((tid >> lo) << (lo + LOGE)) | (tid & ((1u << lo) - 1u));
#pragma unroll
            for (uint32_t c = 0; c < E; c++) {
                kv[c] = sk[tile_sw(b0 | (c << lo))];
                if (HAS_VAL) vv[c] = sv[tile_sw(b0 | (c << lo))];
            }
            for (uint32_t st = s0; st <= s1; st++) {
                const uint32_t top = s0 == s1 ? hi : st;  // first round: stage st runs bits st-1 .. 0
                const bool up_t = st < LOGE || ((tid >> (st - LOGE)) & 1u) == 0;  // (used when st >= lo + 4)
#pragma unroll
                for (int d = (int)LOGE - 1; d >= 0; d--) {
                    if (lo + (uint32_t)d >= top) continue;
#pragma unroll
                    for (uint32_t c = 0; c < E; c++) {
                        if (c & (1u << d)) continue;
                        const uint32_t c2 = c | (1u << d);
                        const bool up = st >= lo + LOGE ? up_t : ((c >> (st - lo)) & 1u) == 0;
                        const uint64_t a = kv[c], b = kv[c2];
                        if ((a > b) == up) {
                            kv[c] = b;
                            kv[c2] = a;
                            if (HAS_VAL) {
                                const uint64_t va = vv[c];
                                vv[c] = vv[c2];
                                vv[c2] = va;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (uint32_t c = 0; c < E; c++) {
                sk[tile_sw(b0 | (c << lo))] = kv[c];
                if (HAS_VAL) sv[tile_sw(b0 | (c << lo))] = vv[c];
            }
        }
        __syncthreads();
    };
    round(0, 1, LOGE, 0);
    for (uint32_t st = LOGE + 1; st <= logT; st++) {
        uint32_t hi = st;
        while (hi > 0) {
            const uint32_t lo = hi > LOGE ? hi - LOGE : 0;
            round(lo, st, st, hi);
            hi = lo;
        }
    }
    for (uint32_t i = tid; i < T; i += blockDim.x) {
        const uint32_t g = base + i;
        if (g < n) {
            r.k[g] = sk[tile_sw(i)];
            if (HAS_VAL) rec_stv(r, g, sv[tile_sw(i)]);
        }
    }
    __syncthreads();
}
// (not inlined: the sort's 16 keys per thread are the kernel's register peak; inlined, the allocator spilled around it in every other phase —
// 264 bytes of scratch per thread, the compaction 51 -> 80 us)
__device__ __noinline__ void rec_sort(const RecView &r, uint32_t n, uint8_t *pool) {
    if (n < 2) return;
    uint32_t np2 = 2;
    while (np2 < n) np2 <<= 1;
    uint32_t T = CB_POOL_BYTES / (r.has_val ? 16 : 8);  // power of two
    if (T > np2) T = np2;
    uint64_t *sk = reinterpret_cast<uint64_t *>(pool), *sv = sk + T;
    for (uint32_t base = 0; base < n; base += T) {
        if (T >= 1024 && blockDim.x * (r.has_val ? 8 : 16) >= T) {  // (every caller runs CB_LAUNCH = 1024 threads)
            if (r.has_val) rec_tile_sort_fast<true, 3>(r, n, base, T, sk, sv);
            else rec_tile_sort_fast<false, 4>(r, n, base, T, sk, sv);
        } else {
            rec_tile_pass(r, n, base, T, 2, T, 0, sk, sv);
        }
    }
    for (uint32_t k = 2 * T; k <= np2; k <<= 1) {
        rec_global_step(r, n, np2, k, 0, true);
        uint32_t j = k >> 2;
        for (; j >= T; j >>= 1) rec_global_step(r, n, np2, k, j, false);
        for (uint32_t base = 0; base < n; base += T) rec_tile_pass(r, n, base, T, 0, 0, T >> 1, sk, sv);
    }
}

// outlier lists are appended with atomics in arrival order; sorting them by element index makes the payload a pure
// function of the input (the reference's CI compares stream digests across platforms, .github/workflows/cmake.yml:295-310).
// Lists beyond 32768 records (a sign that the bound is too tight for the data; sorting them here, by one workgroup through global
// memory, takes tens of milliseconds) go into the payload in arrival order and are sorted there by finish() (sz3hip_sortlists.hip).
// With a 512 KB scratch area the sort runs on keys alone — (index << 16) | arrival position — which fit one 16384-key
// LDS tile up to 16384 records (records with their values take 16 bytes: 8192 per tile); the values follow by position.
__device__ void sort_outlier_list(uint64_t *idx, void *val, uint64_t n, uint64_t cap, bool v32, uint8_t *pool, uint64_t *scratch) {
    if (n > cap) n = cap;
    if (n < 2 || n > 32768) return;
    if (scratch) {  // [32768] keys, then [32768] saved values (indices are element offsets: far below 2^48)
        uint64_t *sk = scratch, *sv = scratch + 32768;
        const uint32_t m = (uint32_t)n;
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
            sk[i] = (idx[i] << 16) | i;
            sv[i] = v32 ? (uint64_t) reinterpret_cast<const uint32_t *>(val)[i] : reinterpret_cast<const uint64_t *>(val)[i];
        }
        __syncthreads();
        RecView r{sk, nullptr, false, false};
        rec_sort(r, m, pool);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
            const uint64_t k = sk[i];
            idx[i] = k >> 16;
            const uint64_t v = sv[(uint32_t)(k & 0xFFFFu)];
            if (v32) reinterpret_cast<uint32_t *>(val)[i] = (uint32_t)v;
            else reinterpret_cast<uint64_t *>(val)[i] = v;
        }
        return;
    }
    RecView r{idx, reinterpret_cast<uint8_t *>(val), true, v32};
    rec_sort(r, (uint32_t)n, pool);
}

// ---- two-queue Huffman merge, serial forms ---------------------------------------------------------------------
// Leaves sorted by ascending frequency (keys = freq << 16 | sym); internal node k is created at step k, so the
// internal queue is ifreq[j .. k). pleaf[i] / pint[j] = parent (internal node index).
// Generic serial form (thread 0, 64-bit frequencies): heads and their successors are kept in registers.
__device__ void cb_merge(const uint64_t *keys, uint64_t *ifreq, uint16_t *pleaf, uint16_t *pint, uint32_t m) {
    const uint64_t INF = ~0ull;
    uint32_t i = 0, j = 0;
    uint64_t lf = keys[0] >> 16, lf_next = m > 1 ? keys[1] >> 16 : INF;  // leaf queue: head, head + 1
    uint64_t nf = INF, nf_next = INF;                                     // internal queue: head, head + 1
    for (uint32_t k = 0; k + 1 < m; k++) {
        uint64_t f = 0;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            if (lf <= nf) {  // ties prefer the leaf; INF marks an exhausted / empty queue
                f += lf;
                pleaf[i++] = (uint16_t)k;
                lf = lf_next;
                lf_next = i + 1 < m ? keys[i + 1] >> 16 : INF;
            } else {
                f += nf;
                pint[j++] = (uint16_t)k;
                nf = nf_next;
                nf_next = j + 1 < k ? ifreq[j + 1] : INF;  // nodes < k exist; node k is appended below
            }
        }
        ifreq[k] = f;
        if (j == k) nf = f;
        else if (j + 1 == k) nf_next = f;
    }
}
// Wave form (all 64 lanes of one wave, frequencies < 2^32 - 1): each queue has a 64-entry register window, one
// entry per lane, read with v_readlane at a scalar index; the parents of the window's entries are collected in a
// second register per queue and written to LDS when the window moves on. No LDS access on the critical path except
// for nodes created beyond the internal window (near-uniform distributions). nfq: 32-bit internal frequencies (LDS).
__device__ void cb_merge_wave32(const uint64_t *keys, uint32_t *nfq, uint16_t *pleaf, uint16_t *pint, uint32_t m) {
    const uint32_t lane = lane_id();
    const uint32_t INF = 0xFFFFFFFFu;
    uint32_t i = 0, j = 0, ibase = 0, jbase = 0;
    uint32_t lreg = lane < m ? (uint32_t)(keys[lane] >> 16) : INF;
    uint32_t nreg = INF;
    uint32_t lpar = 0, npar = 0;  // parents of the window entries
    // the two queue heads live in scalar registers: a pick compares them there and re-reads only the queue it consumed
    // (one v_readlane per pick instead of two, no vector compare on the critical path)
    uint32_t lf = (uint32_t)__builtin_amdgcn_readlane((int)lreg, 0), nf = INF;
    for (uint32_t k = 0; k + 1 < m; k++) {
        uint32_t f = 0;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            if (lf <= nf) {
                f += lf;
                lpar = lane == i - ibase ? k : lpar;
                i++;
                if (i - ibase == WAVE) {
                    pleaf[ibase + lane] = (uint16_t)lpar;
                    ibase = i;
                    lreg = ibase + lane < m ? (uint32_t)(keys[ibase + lane] >> 16) : INF;
                }
                lf = (uint32_t)__builtin_amdgcn_readlane((int)lreg, (int)(i - ibase));
            } else {
                f += nf;
                npar = lane == j - jbase ? k : npar;
                j++;
                if (j - jbase == WAVE) {
                    pint[jbase + lane] = (uint16_t)npar;
                    jbase = j;
                    nreg = jbase + lane < k ? nfq[jbase + lane] : INF;
                }
                nf = (uint32_t)__builtin_amdgcn_readlane((int)nreg, (int)(j - jbase));  // (INF when the queue ran empty: j == k)
            }
        }
        if (k - jbase < WAVE) nreg = (lane == k - jbase) ? f : nreg;
        else if (lane == 0) nfq[k] = f;
        if (j == k) nf = f;  // the queue was empty: the new node is its head
    }
    if (ibase + lane < i) pleaf[ibase + lane] = (uint16_t)lpar;
    if (jbase + lane < j) pint[jbase + lane] = (uint16_t)npar;
}

// ---- round-parallel merge (wide alphabets, whole workgroup) -----------------------------------------------------
// Invariant of the two-queue merge: every pending internal node is <= c, the sum of the two smallest pending items,
// and every node created from now on is >= c. So ALL pending items below c (set A) get paired among themselves, in
// merged sorted order (ties: leaf first), before any new node is touched: one round creates |A|/2 nodes at once —
// a co-rank search per pair. c at least doubles the smallest pending item per round: O(log(total)) rounds.
// INLDS: 32-bit frequencies in LDS (lf32 leaves, nf32 internals); else 64-bit in global memory (keys, ifreq).
// ONEWAVE (round 6; alphabets of a few hundred symbols): the same rounds by ONE wave — no workgroup barrier and no LDS counter between the
// probe and the pairing (a round is ~0.4 us instead of ~1: C1's 398 symbols 23 -> ~10 us); called by the threads of wave 0 only, NT = 64.
template <bool INLDS, bool ONEWAVE = false>
__device__ void cb_merge_rounds(const uint64_t *keys, uint64_t *ifreq, const uint32_t *lf32, uint32_t *nf32,
                                uint16_t *pleaf, uint16_t *pint, uint32_t m, uint32_t *s_red /* [4], zeroed */, uint32_t NT /* live threads */) {
    const uint32_t t = threadIdx.x, lane = lane_id();
    auto sync = [&]() {
        if (ONEWAVE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    };
    const uint64_t INF = ~0ull;
    auto LF = [&](uint32_t x) -> uint64_t { return INLDS ? (uint64_t)lf32[x] : keys[x] >> 16; };
    auto NF = [&](uint32_t x) -> uint64_t { return INLDS ? (uint64_t)nf32[x] : ifreq[x]; };
    uint32_t i = 0, j = 0, k = 0, round = 0;
    while (k + 1 < m) {
        uint32_t *red = s_red + 2 * (round & 1), *red_next = s_red + 2 * ((round + 1) & 1);
        const uint64_t l0 = i < m ? LF(i) : INF, l1 = i + 1 < m ? LF(i + 1) : INF;
        const uint64_t n0 = j < k ? NF(j) : INF, n1 = j + 1 < k ? NF(j + 1) : INF;
        uint64_t c;
        if (l0 <= n0) c = l0 + (l1 <= n0 ? l1 : n0);
        else c = n0 + (l0 <= n1 ? l0 : n1);
        // items below c in each queue: NT-ary probe (one barrier), then a 64-ary probe inside the hit segment
        const uint32_t RL = m - i, RN = k - j;
        const uint32_t SL = (RL + NT - 1) / NT, SN = (RN + NT - 1) / NT;  // <= 64 (m <= 65536, NT = 1024)
        bool pl = false, pn = false;
        if (RL && t * SL < RL) {
            const uint32_t e = (t + 1) * SL < RL ? (t + 1) * SL : RL;
            pl = LF(i + e - 1) < c;
        }
        if (RN && t * SN < RN) {
            const uint32_t e = (t + 1) * SN < RN ? (t + 1) * SN : RN;
            pn = NF(j + e - 1) < c;
        }
        const uint32_t cl = (uint32_t)__popcll(__ballot(pl)), cn = (uint32_t)__popcll(__ballot(pn));
        uint32_t nl, nn;  // full segments below c
        if (ONEWAVE) {
            nl = cl * SL;
            nn = cn * SN;
        } else {
            if (lane == 0) {
                if (cl) atomicAdd(&red[0], cl);
                if (cn) atomicAdd(&red[1], cn);
            }
            if (t == 0) red_next[0] = red_next[1] = 0;
            __syncthreads();
            nl = red[0] * SL;
            nn = red[1] * SN;
        }
        if (nl < RL) {
            const uint32_t x = nl + lane;
            const bool b = lane < SL && x < RL && LF(i + x) < c;
            nl += (uint32_t)__popcll(__ballot(b));
        } else nl = RL;
        if (nn < RN) {
            const uint32_t x = nn + lane;
            const bool b = lane < SN && x < RN && NF(j + x) < c;
            nn += (uint32_t)__popcll(__ballot(b));
        } else nn = RN;
        const uint32_t tot = nl + nn, P = tot >> 1;  // tot >= 2
        // the unpaired last item of an odd A stays pending: it is the larger of the two tails (tie: the internal)
        bool leaf_last = false;
        if (tot & 1) leaf_last = nl > 0 && (nn == 0 || LF(i + nl - 1) > NF(j + nn - 1));
        for (uint32_t q = t; q < P; q += NT) {
            const uint32_t d = 2 * q;
            uint32_t lo = d > nn ? d - nn : 0, hi = d < nl ? d : nl;
            while (lo < hi) {  // co-rank: smallest a such that leaf[a] does not precede internal[d - a - 1]
                const uint32_t a = (lo + hi) >> 1;
                if (LF(i + a) <= NF(j + d - a - 1)) lo = a + 1;
                else hi = a;
            }
            uint32_t a = lo, b = d - lo;
            uint64_t sum = 0;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const uint64_t fl = a < nl ? LF(i + a) : INF, fn = b < nn ? NF(j + b) : INF;
                if (fl <= fn) {
                    sum += fl;
                    pleaf[i + a] = (uint16_t)(k + q);
                    a++;
                } else {
                    sum += fn;
                    pint[j + b] = (uint16_t)(k + q);
                    b++;
                }
            }
            if (INLDS) nf32[k + q] = (uint32_t)sum;
            else ifreq[k + q] = sum;
        }
        i += nl - ((tot & 1) && leaf_last ? 1 : 0);
        j += nn - ((tot & 1) && !leaf_last ? 1 : 0);
        k += P;
        round++;
        sync();
    }
}

// Kraft repair after clamping code lengths to L. Leaves are sorted by ascending frequency and Huffman lengths never
// increase along that order, so the cheapest leaves to lengthen are the first ones below the limit. Closed form:
// the c clamped leaves come first; promoting leaf q >= c all the way to the limit frees 2^(L - len_q) - 1 units;
// take the shortest prefix of them that covers the excess E, then hand the surplus of the last one back by shortening
// the most frequent maximal-length codes by one bit each. NT live threads, contiguous runs of leaves per thread;
// s_cnt is rebuilt. s_ws: [NT / 64 + 2] u64 scratch in LDS.
__device__ void cb_kraft_repair(uint16_t *pleaf, uint32_t m, uint32_t *s_cnt, uint32_t L, uint32_t NT, uint64_t *s_ws) {
    const uint32_t t = threadIdx.x;
    uint64_t kraft = 0;
    for (uint32_t l = 1; l <= L; l++) kraft += (uint64_t)s_cnt[l] << (L - l);
    const uint64_t E = kraft - (1ull << L);  // > 0: the caller saw a clamp
    const uint32_t c = s_cnt[L];
    const uint32_t per = (m + NT - 1) / NT;
    const uint32_t q0 = t * per < m ? t * per : m, q1 = q0 + per < m ? q0 + per : m;
    uint64_t run = 0;
    for (uint32_t q = q0; q < q1; q++) run += q >= c ? ((1ull << (L - pleaf[q])) - 1) : 0ull;
    const uint64_t incl = wave_incl_scan(run);
    uint32_t *s_k = reinterpret_cast<uint32_t *>(s_ws + NT / WAVE);  // [0] = kend, [1] = slack
    if (lane_id() == WAVE - 1) s_ws[t / WAVE] = incl;
    if (t == 0) {
        s_k[0] = 0xFFFFFFFFu;
        s_k[1] = 0;
    }
    __syncthreads();
    uint64_t acc = incl - run;
    for (uint32_t wv = 0; wv < t / WAVE; wv++) acc += s_ws[wv];
    for (uint32_t q = q0; q < q1; q++) {
        const uint64_t lo = acc;
        acc += q >= c ? ((1ull << (L - pleaf[q])) - 1) : 0ull;
        if (lo < E && E <= acc) {  // exactly one q satisfies this
            s_k[0] = q;
            s_k[1] = (uint32_t)(acc - E < 0xFFFFFFFFull ? acc - E : 0xFFFFFFFFull);
        }
    }
    __syncthreads();
    const uint32_t kend = s_k[0];
    if (kend != 0xFFFFFFFFu) {
        const uint32_t nlim = kend + 1;  // leaves [0, kend] now sit at the limit
        const uint32_t back = s_k[1] < nlim ? s_k[1] : nlim;
        for (uint32_t q = t; q <= kend; q += NT) pleaf[q] = (uint16_t)((q + back > kend) ? L - 1 : L);
    }
    if (t < SZH_MAX_LEN + 2) s_cnt[t] = 0;
    __syncthreads();
    for (uint32_t q = t; q < m; q += NT) atomicAdd(&s_cnt[pleaf[q]], 1u);
    __syncthreads();
}

// range of the non-empty histogram bins, many workgroups: range[0] = max(65535 - bin), range[1] = max(bin),
// range[2] = count of non-empty bins (all three start at 0 and only grow)
__global__ __launch_bounds__(256) void k_hist_range(const uint64_t *__restrict__ hist, uint32_t *range) {
    hist += (size_t)blockIdx.y * SZH_HIST_BINS;  // grid.y = code book of a batch
    range += blockIdx.y * 4;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;  // 256 workgroups x 256 bins
    const bool nz = hist[i] != 0;
    const unsigned long long m = __ballot(nz);
    if (m && lane_id() == 0) {
        const uint32_t base = i;  // lane 0's bin
        const uint32_t lo = base + (uint32_t)__ffsll((long long)m) - 1, hi = base + 63u - (uint32_t)__clzll((long long)m);
        atomicMax(&range[0], 0xFFFFu - lo);
        atomicMax(&range[1], hi);
        atomicAdd(&range[2], (uint32_t)__popcll(m));
    }
}

// canonical codes in (length, symbol) order for the compacted alphabet (syms ascending, len_of[q] = length of the
// q-th symbol): every thread owns a contiguous run of symbols; per-(length, thread) counts are scanned along the
// threads by one wave per length. cnt_tbl: u16 [(SZH_MAX_LEN + 1) * NT] in LDS.
__device__ void cb_assign_codes(const uint16_t *len_of, const uint16_t *syms, uint32_t m, const uint32_t *s_first,
                                uint16_t *cnt_tbl, uint32_t *enc, uint32_t NT /* live threads */) {
    const uint32_t t = threadIdx.x, lane = lane_id();
    const uint32_t per = (m + NT - 1) / NT;
    const uint32_t q0 = t * per < m ? t * per : m, q1 = q0 + per < m ? q0 + per : m;
    for (uint32_t l = 0; l <= SZH_MAX_LEN; l++) cnt_tbl[l * NT + t] = 0;
    // (a thread's symbols in batches of eight, the batch's global loads in flight together: one symbol per step was a chain of
    // dependent L2 round trips — the two loops were 29 us of the wide code book's 174 at C3)
    for (uint32_t qb = q0; qb < q1; qb += 8) {
        uint16_t l8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) l8[k] = len_of[qb + k < q1 ? qb + k : q1 - 1];
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (qb + k < q1) cnt_tbl[(uint32_t)l8[k] * NT + t]++;
    }
    __syncthreads();
    for (uint32_t l = 1 + t / WAVE; l <= SZH_MAX_LEN; l += NT / WAVE) {
        uint32_t carry = 0;
        for (uint32_t c0 = 0; c0 < NT; c0 += WAVE) {
            const uint32_t v = cnt_tbl[l * NT + c0 + lane];
            const uint32_t incl = wave_incl_scan(v);
            cnt_tbl[l * NT + c0 + lane] = (uint16_t)(carry + incl - v);
            carry += __shfl(incl, WAVE - 1, WAVE);
        }
    }
    __syncthreads();
    for (uint32_t qb = q0; qb < q1; qb += 8) {
        uint16_t l8[8], s8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            l8[k] = len_of[qb + k < q1 ? qb + k : q1 - 1];
            s8[k] = syms[qb + k < q1 ? qb + k : q1 - 1];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (qb + k >= q1) break;
            const uint32_t l = l8[k];
            const uint32_t rank = cnt_tbl[l * NT + t]++;
            enc[s8[k]] = ((s_first[l] + rank) << 5) | l;
        }
    }
}

// wide alphabets (m > CB_LDS_SYMS): all CB_LAUNCH threads, scratch arrays in global memory (L2-resident), LDS pool
// for the sort tiles, the 32-bit frequency queues and the code assignment table
// Wide alphabets (m > CB_CLASS_MIN: the interpolation predictor's coarse levels, tight bounds on rough data) are coded in
// two classes: the most frequent symbols (at most CB_CLASS_KEEP, else CB_CLASS_KEEP2 of them) take part in the Huffman
// construction one by one, all rarer symbols together as ONE pseudo-symbol of their summed weight; a rare symbol's code is
// the pseudo-symbol's code word followed by a fixed-length index (length = len(pseudo) + ceil(log2(number of rare
// symbols))). The format only stores code lengths, so the decoder does not know about classes. The construction then
// runs on a few thousand keys inside LDS instead of on up to 65536 keys through global memory (C3's 16 365 symbols: 0.25 ->
// 0.20 ms; C4's 25 887: 0.63 -> 0.40 ms). A tier is taken only when its rare class would hold at most 1/64 of the
// occurrences: the coded size then grows by 0.05-0.3 %.
#define CB_CLASS_MIN 4096u
#define CB_CLASS_KEEP 3072u    // first choice: everything after it runs on one 4096-key LDS tile
#define CB_CLASS_KEEP2 12288u  // second choice when the first would put too many occurrences into the rare class
// pre (round 5): the compaction was made by k_cb_compact over the whole chip in front of this launch — keys[] / syms[] hold the
// non-empty bins in symbol order, ifreq[0 .. CBC_BLOCKS) the workgroups' sums of counts (one workgroup reading the 48 183-bin range of
// C3 twice was 46 of the wide book's 164 us: the rate ONE compute unit reads memory at)
#define CBC_BLOCKS 64u
#define CB_ASSIGN_PENDING 0x5A5Au  // szk_cb_info::reserved: the wide book left its code words to k_cb_assign
template <bool ALLOW_CLS>
__device__ void codebook_wide(const uint64_t *__restrict__ hist, const szk_cb_params &p, uint8_t *pool, uint32_t lo,
                              uint32_t range, uint32_t *s_cnt, uint32_t *s_first, uint32_t *s_misc, bool pre = false) {
    const uint32_t t = threadIdx.x, NT = blockDim.x;
    __shared__ uint32_t s_wt[CB_LAUNCH / WAVE];
    __shared__ unsigned long long s_total;
    // 1. compaction of the non-zero bins in symbol order + total count: every wave owns a contiguous slice of the
    //    range and walks it 64 bins at a time (coalesced); positions come from ballot prefix counts
    const uint32_t lane = lane_id(), wv_id = t / WAVE, n_wv = NT / WAVE;
    const uint32_t seg = ((range + n_wv - 1) / n_wv + WAVE - 1) / WAVE * WAVE;  // bins per wave, multiple of 64
    const uint32_t b0 = wv_id * seg < range ? wv_id * seg : range, b1 = b0 + seg < range ? b0 + seg : range;
    uint32_t cnt = 0;
    uint64_t fsum = 0;
    if (pre) {
        if (t < CBC_BLOCKS) fsum = p.ifreq[t];
        if (wv_id == 0 && lane == 0) cnt = (uint32_t)p.ifreq[CBC_BLOCKS];  // (the number of keys, as k_cb_compact counted them)
    }
    // (three 64-bin groups per step, their loads in flight together: one per step is a chain of dependent L2 round trips)
    for (uint32_t i0 = b0; i0 < b1 && !pre; i0 += 3 * WAVE) {
        const uint32_t ia = i0 + lane, ib = ia + WAVE, ic = ib + WAVE;
        uint64_t fa = hist[lo + (ia < b1 ? ia : b1 - 1)], fb = hist[lo + (ib < b1 ? ib : b1 - 1)], fc = hist[lo + (ic < b1 ? ic : b1 - 1)];
        fa = ia < b1 ? fa : 0ull;
        fb = ib < b1 ? fb : 0ull;
        fc = ic < b1 ? fc : 0ull;
        cnt += (uint32_t)__popcll(__ballot(fa != 0)) + (uint32_t)__popcll(__ballot(fb != 0)) + (uint32_t)__popcll(__ballot(fc != 0));
        fsum += fa + fb + fc;
    }
    if (t == 0) s_total = 0;
    if (t < 8) s_misc[t] = 0;
    if (lane == 0) s_wt[wv_id] = cnt;
    __syncthreads();
    fsum = wave_sum(fsum);
    if (lane == 0 && fsum) atomicAdd(&s_total, (unsigned long long)fsum);
    uint32_t pos = 0, m = 0;
    for (uint32_t wv = 0; wv < n_wv; wv++) {
        if (wv < wv_id) pos += s_wt[wv];
        m += s_wt[wv];
    }
    // (four groups' loads in flight, like the counting pass: one load per step was 47 dependent L2 round trips per wave at C3 —
    // 51 of the kernel's 174 us)
    for (uint32_t i0 = b0; i0 < b1 && !pre; i0 += 4 * WAVE) {
        uint64_t f4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {  // (clamped, never conditional: a load under a branch is waited for inside it)
            const uint32_t i = i0 + k * WAVE + lane;
            const uint64_t f = hist[lo + (i < b1 ? i : b1 - 1)];
            f4[k] = i < b1 ? f : 0ull;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = i0 + k * WAVE + lane;
            const uint64_t f = f4[k];
            const unsigned long long bal = __ballot(f != 0);
            if (f) {
                const uint32_t at = pos + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                p.keys[at] = (f << 16) | (lo + i);
                p.syms[at] = (uint16_t)(lo + i);
            }
            pos += (uint32_t)__popcll(bal);
        }
    }
    __syncthreads();
    uint64_t total = s_total;
    const uint32_t L = m <= CB_SHORT_SYMS ? 16u : SZH_MAX_LEN;
    // 1b. two-class mode: keep the frequent symbols, fold the rare ones into one pseudo-symbol
    const bool cls = ALLOW_CLS && m > CB_CLASS_MIN;
    uint32_t mk = m, n_rare = 0, rare_bits = 0, pseudo_sym = 0;
    uint64_t rare_max = 0;  // a symbol is rare when its count is <= rare_max
    if (cls) {
        __shared__ uint32_t s_cls[48];
        __shared__ unsigned long long s_fsum;
        if (t < 48) s_cls[t] = 0;
        if (t == 0) s_fsum = 0;
        __syncthreads();
        if (pre) {  // (k_cb_compact counted the keys per octave, slice by slice: 64 x 48 words — 13 us of LDS atomics on a dozen addresses at C3's 16 365 keys otherwise)
            for (uint32_t q = t; q < CBC_BLOCKS * 48u; q += NT) {
                const uint32_t c = (uint32_t)p.ifreq[CBC_BLOCKS + 1 + q];
                if (c) atomicAdd(&s_cls[q % 48u], c);
            }
        } else {
            for (uint32_t q = t; q < m; q += NT) atomicAdd(&s_cls[63 - __clzll((long long)(p.keys[q] >> 16))], 1u);
        }
        __syncthreads();
        if (t == 0) p.info->ts[9] = wall_clock64();
        if (t == 0) {
            // a flat code only suits a class that carries little: the rare class may hold at most 1/64 of the occurrences
            // (estimated from the counts per octave); a near-uniform spread over tens of thousands of bins (ratio ~2)
            // keeps the one-by-one construction
            uint32_t choice = 0xFFFFFFFFu;
            const uint32_t keep[2] = {CB_CLASS_KEEP, CB_CLASS_KEEP2};
            for (int tier = 0; tier < 2 && choice == 0xFFFFFFFFu; tier++) {
                if (m <= keep[tier] + keep[tier] / 4) continue;  // (nothing to gain)
                uint32_t above = 0;  // smallest k with (symbols of count >= 2^(k+1)) <= keep
                int k = 47;
                while (k >= 0 && above + s_cls[k] <= keep[tier]) above += s_cls[k--];
                unsigned long long mass = 0;
                for (int j = 0; j <= k; j++) mass += (unsigned long long)s_cls[j] * (3ull << j) / 2ull;
                if (k >= 0 && mass * 64ull <= total) choice = (uint32_t)k;
            }
            s_misc[5] = choice;
            s_misc[6] = 0xFFFFFFFFu;
        }
        __syncthreads();
        if (s_misc[5] == 0xFFFFFFFFu) {
            __syncthreads();
            if (ALLOW_CLS) codebook_wide<false>(hist, p, pool, lo, range, s_cnt, s_first, s_misc);
            return;
        }
        rare_max = (2ull << s_misc[5]) - 1ull;
        // compaction of the frequent keys (symbol order kept) into ifreq[], as in step 1
        const uint32_t segk = ((m + n_wv - 1) / n_wv + WAVE - 1) / WAVE * WAVE;
        const uint32_t k0 = wv_id * segk < m ? wv_id * segk : m, k1 = k0 + segk < m ? k0 + segk : m;
        uint32_t c2 = 0;
        for (uint32_t i0 = k0; i0 < k1; i0 += 4 * WAVE) {  // (four loads in flight per step, here and below)
            uint64_t f4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = i0 + k * WAVE + lane;
                const uint64_t f = p.keys[i < k1 ? i : k1 - 1] >> 16;
                f4[k] = i < k1 ? f : 0ull;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) c2 += (uint32_t)__popcll(__ballot(f4[k] > rare_max));
        }
        __syncthreads();  // (s_wt of step 1 no longer read)
        if (lane == 0) s_wt[wv_id] = c2;
        __syncthreads();
        uint32_t pos2 = 0, n_freq = 0;
        for (uint32_t wv = 0; wv < n_wv; wv++) {
            if (wv < wv_id) pos2 += s_wt[wv];
            n_freq += s_wt[wv];
        }
        uint64_t fs = 0;
        uint32_t rare_sym = 0xFFFFFFFFu;
        for (uint32_t i0 = k0; i0 < k1; i0 += 4 * WAVE) {
            uint64_t k4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = i0 + k * WAVE + lane;
                const uint64_t kk = p.keys[i < k1 ? i : k1 - 1];
                k4[k] = i < k1 ? kk : 0ull;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint64_t key = k4[k];
                const uint64_t f = key >> 16;
                const bool fr = f > rare_max;
                const unsigned long long bal = __ballot(fr);
                if (fr) {
                    p.ifreq[pos2 + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = key;
                    fs += f;
                } else if (f) {
                    const uint32_t sy = (uint32_t)(key & 0xFFFF);
                    rare_sym = sy < rare_sym ? sy : rare_sym;
                }
                pos2 += (uint32_t)__popcll(bal);
            }
        }
        fs = wave_sum(fs);
        if (lane == 0 && fs) atomicAdd(&s_fsum, (unsigned long long)fs);
        if (rare_sym != 0xFFFFFFFFu) atomicMin(&s_misc[6], rare_sym);
        __syncthreads();
        n_rare = m - n_freq;
        rare_bits = n_rare > 1 ? 32u - (uint32_t)__clz((int)(n_rare - 1)) : 0u;
        pseudo_sym = s_misc[6];  // a rare symbol's id stands for the class (no frequent key carries it)
        // weight of the class: the rare occurrences, but at least enough for a code word short enough to leave room for
        // the index below it (Huffman gives a weight w of W a length of about log2(W / w), +-1)
        uint64_t w_rare = total - s_fsum;
        const uint32_t room = L > rare_bits + 3 ? L - rare_bits - 3 : 1;
        const uint64_t w_min = (total >> room) + 1;
        if (w_rare < w_min) w_rare = w_min;
        total = s_fsum + w_rare;
        for (uint32_t q = t; q < n_freq; q += NT) p.keys[q] = p.ifreq[q];
        if (t == 0) p.keys[n_freq] = (w_rare << 16) | pseudo_sym;
        mk = n_freq + 1;
        __syncthreads();
    }
    if (t == 0) p.info->ts[2] = wall_clock64();
    // 2. sort by (freq, sym)
    RecView rv{p.keys, nullptr, false, false};
    rec_sort(rv, mk, pool);
    if (t == 0) p.info->ts[3] = wall_clock64();
    // 3. merge
    uint16_t *pleaf = p.pleaf, *pint = p.pint, *aux = p.depth, *aux2 = p.aux2, *pint2 = p.pint2;
    constexpr uint32_t LDSQ = CB_POOL_BYTES / 8;  // symbols whose two 32-bit queues fit the pool
    if (mk <= LDSQ && total < 0xFFFFFFFFull) {
        uint32_t *lf32 = reinterpret_cast<uint32_t *>(pool), *nf32 = lf32 + LDSQ;
        for (uint32_t q = t; q < mk; q += NT) lf32[q] = (uint32_t)(p.keys[q] >> 16);
        __syncthreads();
        cb_merge_rounds<true>(p.keys, p.ifreq, lf32, nf32, pleaf, pint, mk, s_misc, blockDim.x);
    } else {
        cb_merge_rounds<false>(p.keys, p.ifreq, nullptr, nullptr, pleaf, pint, mk, s_misc, blockDim.x);
    }
    if (t == 0) p.info->ts[4] = wall_clock64();
    // 4. depth of every internal node by pointer doubling (min(depth, 2^rounds) is all the clamp needs); the four
    //    u16 arrays ping-pong in the LDS pool when they fit (m <= 16384), else in global memory
    {
        const bool in_lds = mk <= CB_POOL_BYTES / 8;
        uint16_t *dA = in_lds ? reinterpret_cast<uint16_t *>(pool) : aux, *pA = in_lds ? dA + CB_POOL_BYTES / 8 : pint;
        uint16_t *dB = in_lds ? pA + CB_POOL_BYTES / 8 : aux2, *pB = in_lds ? dB + CB_POOL_BYTES / 8 : pint2;
        __syncthreads();  // (the merge's LDS queues are dead from here on)
        for (uint32_t q = t; q + 1 < mk; q += NT) {
            pA[q] = q == mk - 2 ? (uint16_t)q : pint[q];
            dA[q] = q == mk - 2 ? 0 : 1;
        }
        __syncthreads();
        for (uint32_t span = 1; span < mk && span < 2 * L; span <<= 1) {
            for (uint32_t q = t; q + 1 < mk; q += NT) {
                const uint16_t jn = pA[q];
                const uint32_t sum = (uint32_t)dA[q] + dA[jn];
                dB[q] = (uint16_t)(sum > 0xFFFFu ? 0xFFFFu : sum);
                pB[q] = pA[jn];
            }
            __syncthreads();
            uint16_t *sw = dA;
            dA = dB;
            dB = sw;
            sw = pA;
            pA = pB;
            pB = sw;
        }
        // leaf lengths in sorted order, clamped
        for (uint32_t q = t; q < mk; q += NT) {
            uint32_t l = (uint32_t)dA[pleaf[q]] + 1;
            if (l > L) {
                l = L;
                s_misc[4] = 1;
            }
            pleaf[q] = (uint16_t)l;
            atomicAdd(&s_cnt[l], 1u);
        }
        __syncthreads();
    }
    if (t == 0) p.info->ts[5] = wall_clock64();
    // 5. Kraft repair when a natural depth exceeded the limit
    if (s_misc[4]) cb_kraft_repair(pleaf, mk, s_cnt, L, NT, reinterpret_cast<uint64_t *>(pool));
    // 6. lengths to symbol order (through the serialised lens[] table), then canonical codes
    __syncthreads();
    for (uint32_t qb = t; qb < mk; qb += 4 * NT) {  // (four loads in flight per step, as in the compaction)
        uint32_t sy4[4];
        uint16_t l4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t q = qb + k * NT;
            sy4[k] = (uint32_t)(p.keys[q < mk ? q : mk - 1] & 0xFFFF);
            l4[k] = pleaf[q < mk ? q : mk - 1];
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (qb + k * NT < mk) p.lens[sy4[k]] = (uint8_t)l4[k];
    }
    __syncthreads();
    if (cls) {  // every rare symbol: the class's code word + a fixed-length index
        const uint32_t len_cls = p.lens[pseudo_sym], len_rare = len_cls + rare_bits;
        if (len_rare > L || (p.dbg & 1u)) {  // (the weight floor above makes this all but impossible) one-class construction instead
            __syncthreads();
            if (t < SZH_MAX_LEN + 2) s_cnt[t] = 0;
            for (uint32_t q = t; q < range; q += NT) p.lens[lo + q] = 0;
            __syncthreads();
            if (ALLOW_CLS) codebook_wide<false>(hist, p, pool, lo, range, s_cnt, s_first, s_misc);
            return;
        }
        __syncthreads();
        for (uint32_t qb = t; qb < m; qb += 4 * NT) {
            uint32_t sy4[4];
            uint64_t f4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) sy4[k] = (uint32_t)p.syms[qb + k * NT < m ? qb + k * NT : m - 1];
#pragma unroll
            for (int k = 0; k < 4; k++) f4[k] = hist[sy4[k]];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (qb + k * NT < m && f4[k] <= rare_max) p.lens[sy4[k]] = (uint8_t)len_rare;
        }
        if (t == 0) {
            s_cnt[len_cls] -= 1;
            s_cnt[len_rare] += n_rare;
        }
        __syncthreads();
    }
    if (t == 0) {
        uint32_t code = 0;
        for (uint32_t l = 1; l <= L; l++) {
            code = (code + (l > 1 ? s_cnt[l - 1] : 0)) << (l > 1 ? 1 : 0);
            s_first[l] = code;
        }
        p.info->ts[6] = wall_clock64();
    }
    for (uint32_t qb = t; qb < m; qb += 4 * NT) {
        uint32_t sy4[4];
        uint8_t l4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) sy4[k] = (uint32_t)p.syms[qb + k * NT < m ? qb + k * NT : m - 1];
#pragma unroll
        for (int k = 0; k < 4; k++) l4[k] = p.lens[sy4[k]];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (qb + k * NT < m) aux[qb + k * NT] = l4[k];
    }
    __syncthreads();
    if (t == 0) p.info->ts[7] = wall_clock64();
    // the code words: by k_cb_assign over the whole chip behind this launch (round 5; p.assign_later), else here
    const bool defer = p.assign_later != 0 && p.n_books <= 1;
    if (defer) {
        if (t <= SZH_MAX_LEN) p.info->first_code[t] = t >= 1 && t <= L ? s_first[t] : 0u;
    } else {
        cb_assign_codes(aux, p.syms, m, s_first, reinterpret_cast<uint16_t *>(pool), p.enc, NT);
    }
    uint32_t max_len = 0;
    for (uint32_t l = 1; l <= L; l++)
        if (s_cnt[l]) max_len = l;
    if (t == 0) {
        uint32_t peak = (uint32_t)(p.keys[mk - 1] & 0xFFFF);  // most frequent symbol: centre of the packers' LDS window
        if (cls && peak == pseudo_sym && mk > 1) peak = (uint32_t)(p.keys[mk - 2] & 0xFFFF);  // (not the rare class)
        uint32_t wl = peak > lo + ENC_WIN / 2 ? peak - ENC_WIN / 2 : lo;
        if (range > ENC_WIN && wl > lo + range - ENC_WIN) wl = lo + range - ENC_WIN;
        if (range <= ENC_WIN) wl = lo;
        p.info->ts[8] = wall_clock64();
        p.info->n_symbols = m;
        p.info->max_len = max_len;
        p.info->sym_min = lo;
        p.info->sym_count = range;
        p.info->win_lo = wl;
        p.info->esc_sym = 0;
        p.info->reserved = defer ? CB_ASSIGN_PENDING : 0u;
    }
}

// The small-alphabet construction (up to CAP <= CB_LDS_SYMS symbols, 256 threads, everything in LDS): compaction, rank sort,
// two-queue merge by one wave out of registers, depths by pointer doubling, length limit + Kraft repair, canonical codes.
// A device function so that two launches can run it: k_codebook<0> (block 0) and — speculative stage 2 — a workgroup of the
// packer's launch (k_pack, book role). The caller zeroes s_cnt / s_over / s_total and enc / lens over [lo, lo + range).
// Margins (round 4). A small alphabet's book also gives a code word to every symbol that did NOT occur between CB_MARGIN below the
// smallest and CB_MARGIN above the largest one that did (each counted once: the histogram the book is optimal for is
// max(hist, 1) over that range). The ends of a smooth field's alphabet are symbols that occur once or twice in 10^8: the next
// array of a series has its own, and a book without margins is incomplete for it half of the time (the verdict of a speculative
// stage 2, book_rejected). Cost: a dozen 16-bit code words' share of the Kraft sum (2e-4) and as many bytes of the lengths' table.
// Not when symbol 0 occurs (listed deltas / unpredictable points: the range then spans half the code space), not for a single
// symbol (zero-length code), not when the widened range leaves the small path's 256 symbols. Same rule wherever the small
// book is built (k_codebook<0>, the packer's book role): the book stays a function of the histogram.
#define CB_MARGIN 8u
__device__ __forceinline__ bool cb_margins(uint32_t &lo, uint32_t &range, uint32_t n_nonzero) {
    if (lo == 0 || n_nonzero < 2) return false;
    const uint32_t hi = lo + range - 1;
    const uint32_t lo2 = lo > CB_MARGIN ? lo - CB_MARGIN : 1u, hi2 = hi + CB_MARGIN < SZH_HIST_BINS ? hi + CB_MARGIN : SZH_HIST_BINS - 1;
    if (hi2 - lo2 + 1 > CB_SMALL_SYMS) return false;
    lo = lo2;
    range = hi2 - lo2 + 1;
    return true;
}
template <uint32_t CAP>
__device__ void cb_small(const uint64_t *__restrict__ hist, const szk_cb_params &p, uint8_t *s_pool, uint32_t lo, uint32_t range,
                         uint32_t *s_wtot, uint32_t &s_over, uint32_t *s_first, uint32_t *s_cnt, uint32_t *s_misc, unsigned long long &s_total,
                         bool fill = false /* cb_margins widened [lo, lo + range): its empty bins count once */,
                         const uint32_t *wsrc = nullptr /* the symbols' weights instead of hist: wsrc[i] for symbol lo + i (round 6, the sampled book) */,
                         uint32_t force_L = 0 /* != 0: the length limit (else 16 bits up to CB_SHORT_SYMS symbols, SZH_MAX_LEN beyond) */) {
    const uint32_t t = threadIdx.x;
    // ---------------- small alphabets: LDS-resident, 256 threads ----------------
    uint64_t *keys = reinterpret_cast<uint64_t *>(s_pool);                          // [CAP]
    uint64_t *ifreq = keys + CAP;                                                    // [CAP] (u32 view in the wave merge)
    uint16_t *pleaf = reinterpret_cast<uint16_t *>(ifreq + CAP);                     // 6 x u16 [CAP]
    uint16_t *pint = pleaf + CAP, *aux = pint + CAP, *syms = aux + CAP;
    uint16_t *aux2 = syms + CAP, *pint2 = aux2 + CAP;
    uint16_t *cnt_tbl = pint2 + CAP;  // (SZH_MAX_LEN + 1) * 256 u16 (code assignment beyond 512 symbols; the packer's book role: the Kraft repair's few words)
    const uint32_t per = (range + CB_THREADS - 1) / CB_THREADS;
    // 1. compaction of the non-zero bins in symbol order
    uint32_t cnt = 0;
    uint64_t fsum = 0;
    for (uint32_t i = t * per; i < range && i < (t + 1) * per; i++) {
        uint64_t f = wsrc ? (uint64_t)wsrc[i] : hist[lo + i];
        if (fill && f == 0) f = 1;
        cnt += f != 0;
        fsum += f;
    }
    uint32_t incl = wave_incl_scan(cnt);
    if (lane_id() == WAVE - 1) s_wtot[t / WAVE] = incl;
    fsum = wave_sum(fsum);
    if (lane_id() == 0 && fsum) atomicAdd(&s_total, (unsigned long long)fsum);
    __syncthreads();
    uint32_t pos = incl - cnt, m = 0;
    for (uint32_t wv = 0; wv < CB_THREADS / WAVE; wv++) {
        if (wv < t / WAVE) pos += s_wtot[wv];
        m += s_wtot[wv];
    }
    for (uint32_t i = t * per; i < range && i < (t + 1) * per; i++) {
        uint64_t f = wsrc ? (uint64_t)wsrc[i] : hist[lo + i];
        if (fill && f == 0) f = 1;
        if (f) {
            keys[pos] = (f << 16) | (lo + i);  // freq < 2^48
            syms[pos] = (uint16_t)(lo + i);
            pos++;
        }
    }
    uint32_t npow2 = 1;
    while (npow2 < m) npow2 <<= 1;
    __syncthreads();
    for (uint32_t i = m + t; i < npow2; i += CB_THREADS) keys[i] = ~0ull;
    __syncthreads();
    if (t == 0) p.info->ts[2] = wall_clock64();
    // 2. sort by (freq, sym): rank sort for small alphabets (every thread counts the keys below its own; LDS
    //    broadcast reads), bitonic otherwise
    if (m <= 512) {
        uint64_t mykey[2];
        uint32_t myrank[2] = {0, 0};
        for (int c = 0; c < 2; c++) mykey[c] = t + c * CB_THREADS < m ? keys[t + c * CB_THREADS] : ~0ull;
        if (m <= CB_THREADS) {
            if (t < m)
                for (uint32_t o = 0; o < m; o++) myrank[0] += keys[o] < mykey[0];  // keys are distinct (symbol in the low bits)
        } else {  // (one walk over the keys serves the thread's two)
            for (uint32_t o = 0; o < m; o++) {
                const uint64_t ko = keys[o];
                myrank[0] += ko < mykey[0];
                myrank[1] += ko < mykey[1];
            }
        }
        __syncthreads();
        for (int c = 0; c < 2; c++)
            if (t + c * CB_THREADS < m) keys[myrank[c]] = mykey[c];
        __syncthreads();
    } else {
        cb_bitonic_sort(keys, npow2);
    }

    if (t == 0) p.info->ts[3] = wall_clock64();
    const uint32_t L = force_L ? force_L : (m <= CB_SHORT_SYMS ? 16u : SZH_MAX_LEN);
    uint32_t max_len = 0;
    if (m == 1) {
        // single symbol: zero-length code, empty bit-stream (as encoder/HuffmanEncoder.hpp:233-237)
    } else {
        // 3. merge: one wave out of registers (32-bit counts), or thread 0 (64-bit counts)
        if (s_total < 0xFFFFFFFFull && !(p.dbg & 2u)) {  // (the rounds by one wave; debug flag 262144: the serial wave merge — C2-like 145 symbols 33 us, C1's 398 87 us)
            // round-parallel merge on the 256 live threads (every round pairs ALL pending items below the smallest possible
            // new node, cb_merge_rounds): ~a dozen rounds for a smooth field's 128 symbols instead of 127 dependent picks of
            // one wave (28 us of the kernel's 38 at C2)
            uint32_t *nf32 = reinterpret_cast<uint32_t *>(ifreq), *lf32 = nf32 + CAP;
            for (uint32_t q = t; q < m; q += CB_THREADS) lf32[q] = (uint32_t)(keys[q] >> 16);
            if (t < 4) s_misc[t] = 0;
            __syncthreads();
            if (t < WAVE) cb_merge_rounds<true, true>(keys, ifreq, lf32, nf32, pleaf, pint, m, s_misc, WAVE);
        } else if (s_total < 0xFFFFFFFFull) {
            if (t < WAVE) cb_merge_wave32(keys, reinterpret_cast<uint32_t *>(ifreq), pleaf, pint, m);
        } else if (t == 0) {
            cb_merge(keys, ifreq, pleaf, pint, m);
        }
        __syncthreads();
        if (t == 0) p.info->ts[4] = wall_clock64();
        // 4. ... depth of every internal node: distance to the root (node m-2) by pointer doubling
        //    aux[q] = distance so far, pint[q] = current ancestor pointer
        for (uint32_t q = t; q + 1 < m; q += CB_THREADS) {
            if (q == m - 2) pint[q] = (uint16_t)q;
            aux[q] = q == m - 2 ? 0 : 1;
        }
        __syncthreads();
        // after r rounds aux[q] = min(depth, 2^r): 2^r >= L is all the clamp below needs
        for (uint32_t span = 1; span < m && span < 2 * L; span <<= 1) {
            for (uint32_t q = t; q + 1 < m; q += CB_THREADS) {
                const uint16_t j = pint[q];
                const uint32_t sum = (uint32_t)aux[q] + aux[j];
                aux2[q] = (uint16_t)(sum > 0xFFFFu ? 0xFFFFu : sum);
                pint2[q] = pint[j];
            }
            __syncthreads();
            for (uint32_t q = t; q + 1 < m; q += CB_THREADS) {
                aux[q] = aux2[q];
                pint[q] = pint2[q];
            }
            __syncthreads();
        }
        if (t == 0) p.info->ts[5] = wall_clock64();
        // 5. leaf lengths (sorted position q), clamp to L, per-length counts
        for (uint32_t q = t; q < m; q += CB_THREADS) {
            uint32_t l = (uint32_t)aux[pleaf[q]] + 1;
            if (l > L) {
                l = L;
                s_over = 1;
            }
            pleaf[q] = (uint16_t)l;
            atomicAdd(&s_cnt[l], 1u);
        }
        __syncthreads();
        if (s_over) cb_kraft_repair(pleaf, m, s_cnt, L, CB_THREADS, reinterpret_cast<uint64_t *>(cnt_tbl));
        // 6. canonical first code per length
        if (t == 0) {
            uint32_t code = 0;
            for (uint32_t l = 1; l <= SZH_MAX_LEN; l++) {
                code = (code + (l > 1 ? s_cnt[l - 1] : 0)) << (l > 1 ? 1 : 0);
                s_first[l] = code;
            }
        }
        if (t == 0) p.info->ts[6] = wall_clock64();
        // 7. scatter the lengths back to symbol order: aux[idx] = length of the idx-th symbol (syms[] is ascending)
        __syncthreads();
        for (uint32_t q = t; q < m; q += CB_THREADS) {
            const uint32_t sym = (uint32_t)(keys[q] & 0xFFFF), l = pleaf[q];
            uint32_t a = 0, bnd = m;  // lower_bound
            while (a < bnd) {
                uint32_t mid = (a + bnd) >> 1;
                if (syms[mid] < sym) a = mid + 1;
                else bnd = mid;
            }
            aux[a] = (uint16_t)l;
            p.lens[sym] = (uint8_t)l;
        }
        __syncthreads();
        if (t == 0) p.info->ts[7] = wall_clock64();
        // 8. codes in (len, symbol) order: rank among the earlier symbols of the same length
        if (m <= 512) {
            for (uint32_t q = t; q < m; q += CB_THREADS) {
                const uint32_t l = aux[q];
                uint32_t rank = 0;
                for (uint32_t r = 0; r < q; r++) rank += aux[r] == l;
                p.enc[syms[q]] = ((s_first[l] + rank) << 5) | l;
            }
        } else {
            cb_assign_codes(aux, syms, m, s_first, cnt_tbl, p.enc, CB_THREADS);
        }
        for (uint32_t l = 1; l <= SZH_MAX_LEN; l++)
            if (s_cnt[l]) max_len = l;
    }
    if (t == 0) {
        const uint32_t peak = (uint32_t)(keys[m - 1] & 0xFFFF);
        uint32_t wl = peak > lo + ENC_WIN / 2 ? peak - ENC_WIN / 2 : lo;
        if (range > ENC_WIN && wl > lo + range - ENC_WIN) wl = lo + range - ENC_WIN;
        if (range <= ENC_WIN) wl = lo;
        p.info->ts[8] = wall_clock64();
        p.info->n_symbols = m;
        p.info->max_len = max_len;
        p.info->sym_min = lo;
        p.info->sym_count = range;
        p.info->win_lo = wl;
        p.info->reserved = 0;
        p.info->esc_sym = 0;
    }
}

// ------------------------------------------------------------------------------------------------------------
// The sampled book (round 6; sz3hip_kernels.h, szk_samp). SZK_SAMP_ROLES workgroups of 256 threads take the sample — in the one-launch
// forms of stage 1 they are the launch's first workgroups, behind the two-launch form a launch of their own (k_sample) —, the last
// one to finish builds the book and raises words[SZK_SAMP_READY]; stage 1's workers poll that word between two planes.
// A unit is a 256-element row segment; unit u lies in segment floor(u * nseg / U) + a hash of u inside that stride (nseg = segments
// of the array, U = the number of units): a function of the extents alone. Every element of a unit gets the byte stage 1 gives it:
// t = min(delta + 127, 255) with delta the Lorenzo stencil over the lattice values (neighbours outside the array: 0).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void samp_unit_place(uint32_t u, uint32_t n_units, uint64_t nseg, uint32_t ntx, uint32_t d1, uint32_t &x0, uint32_t &y, uint32_t &z) {
    const uint64_t stride = nseg / n_units;  // >= 16 (arrays of at least SZK_SAMP_MIN_ELEMS elements)
    const uint64_t seg = (uint64_t)u * nseg / n_units + (uint64_t)((u * 2654435761u) >> 8) % stride;
    x0 = (uint32_t)(seg % ntx) * MARCH_TX;
    const uint64_t row = seg / ntx;
    y = (uint32_t)(row % d1);
    z = (uint32_t)(row / d1);
}
// returns true in the workgroup that finished last (all threads); s_h: [1024] words of LDS
template <typename T>
__device__ __forceinline__ bool samp_take(const T *__restrict__ in, const szk_lattice &latp, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t *words, uint32_t role, uint32_t *s_h) {
    using B = typename Lattice<T>::B;
    using UQ = typename QTraits<T>::UQ;
    constexpr B CB = Lattice<T>::C;
    constexpr int NB = sizeof(T) == 4 ? 2 : 1;  // units in flight per wave (4 rows of 16 / 32 bytes per lane each): the one-launch forms carry this code, and a kernel's registers are its hungriest path's
    __shared__ uint32_t s_last;
    const Lattice<T> lat(latp);
    const uint64_t plane = (uint64_t)d1 * d0;
    const uint32_t ntx = d0 / MARCH_TX;
    const uint64_t nseg = (uint64_t)ntx * d1 * d2;
    const int lane = lane_id();
    const uint32_t gw = role * 4u + threadIdx.x / WAVE, nw = SZK_SAMP_ROLES * 4u;
    for (uint32_t i = threadIdx.x; i < 1024u; i += 256u) s_h[i] = 0;
    __syncthreads();
    for (uint32_t u0 = gw * NB; u0 < SZK_SAMP_UNITS; u0 += nw * NB) {
        Quad<T> rq[NB][4];
        T rl[NB][4];
        bool rok[NB][4], has_left[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            uint32_t x0, y, z;
            samp_unit_place(u0 + b, SZK_SAMP_UNITS, nseg, ntx, d1, x0, y, z);
            has_left[b] = x0 > 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {  // rows (y, z), (y - 1, z), (y, z - 1), (y - 1, z - 1)
                const bool ym = r & 1, zm = r >> 1;
                rok[b][r] = (!ym || y > 0) && (!zm || z > 0);
                if (rok[b][r]) {
                    const T *row = in + (uint64_t)(z - (zm ? 1u : 0u)) * plane + (uint64_t)(y - (ym ? 1u : 0u)) * d0;
                    rq[b][r].load(row + x0 + 4u * (uint32_t)lane);
                    if (has_left[b]) rl[b][r] = row[x0 - 1];
                }
            }
        }
#pragma unroll
        for (int b = 0; b < NB; b++) {
            UQ delta[4] = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (!rok[b][r]) continue;  // (wave-uniform) a row outside the array: the constant C, whose differences vanish
                B q[4];
#pragma unroll
                for (int i = 0; i < 4; i++) q[i] = lat.qbits(rq[b][r].get(i));
                const B left0 = has_left[b] ? lat.qbits(rl[b][r]) : CB;
                UQ pv = (UQ)dpp_wave_shr1(left0, q[3]);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const UQ d1v = (UQ)q[i] - pv;
                    pv = (UQ)q[i];
                    delta[i] = (r == 0 || r == 3) ? (UQ)(delta[i] + d1v) : (UQ)(delta[i] - d1v);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const UQ tq = delta[i] + (UQ)127;
                const uint32_t t = tq <= (UQ)254 ? (uint32_t)tq : 255u;
                atomicAdd(&s_h[t * 4u + ((uint32_t)lane & 3u)], 1u);
            }
        }
    }
    __syncthreads();
    {
        const uint32_t t = threadIdx.x;
        const uint32_t c = s_h[t * 4] + s_h[t * 4 + 1] + s_h[t * 4 + 2] + s_h[t * 4 + 3];
        if (c) atomicAdd(&words[t], c);
    }
    // (no fence: a release at device scope writes back the XCD's whole L2 — megabytes of the workers' freshly stored codes — and took
    // 25 us here. Everything the workgroups exchange inside the launch goes through device-scope atomics, which are performed at the
    // memory side; the wait below orders this workgroup's counts before its ticket)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&words[SZK_SAMP_TICKET], 1u) == SZK_SAMP_ROLES - 1u ? 1u : 0u;
    __syncthreads();
    return s_last != 0;
}
// The book from the sample's counts (the workgroup that finished last: 256 threads); pool: CB_SMALL_SYMS * 28 + 256 bytes of LDS.
// Two classes. The byte values the sample met are coded by a Huffman code over their counts, limited to SZK_SAMP_SEEN_LEN bits (two such
// code words fit an entry of the packer's pair table: a chunk leaves its fast tier only for a value the sample never met). The values
// it did not meet — each a quarter of an occurrence — enter that code as ONE pseudo-symbol; a value's code word is the pseudo-symbol's
// followed by its index among them in ceil(log2(their number)) bits. The format stores code lengths and assigns canonical code words
// by (length, symbol): the decoder is unaware.
__device__ __forceinline__ void samp_book(const szk_samp &sp, uint32_t radius, uint8_t *pool) {
    constexpr uint32_t CAP = CB_SMALL_SYMS, L = SZK_SAMP_SEEN_LEN;
    __shared__ uint8_t s_l[256];
    __shared__ uint32_t s_wc[4], s_wf[4], s_lc[4][SZH_MAX_LEN + 2];
    __shared__ uint32_t s_over;
    __shared__ uint32_t s_first[SZH_MAX_LEN + 2], s_cnt[SZH_MAX_LEN + 2];
    // (cb_small's carving of the pool: the merge and the Kraft repair are its pieces)
    uint64_t *keys = reinterpret_cast<uint64_t *>(pool);                             // [CAP] (weight << 16) | byte value, sorted
    uint64_t *ifreq = keys + CAP;                                                     // [CAP] (u32 view: the internal nodes' weights)
    uint16_t *pleaf = reinterpret_cast<uint16_t *>(ifreq + CAP);                      // 6 x u16 [CAP]
    uint16_t *pint = pleaf + CAP, *aux = pint + CAP, *aux2 = aux + 2 * CAP, *pint2 = aux2 + CAP;
    uint64_t *kraft_ws = reinterpret_cast<uint64_t *>(pint2 + CAP);
    const uint32_t t = threadIdx.x, wv = t / WAVE, ln = t & (WAVE - 1);
    const unsigned long long lower = (1ull << ln) - 1ull;
    const uint32_t lo = radius - 127u, esc = lo + 255u;  // byte t = symbol lo + t; byte 255 (a listed delta) = symbol radius + 128
    const uint32_t cnt = __hip_atomic_load(&sp.words[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the values the sample did not meet: their number, the first of them (it carries the class in the Huffman code)
    const unsigned long long um = __ballot(cnt == 0u);
    if (ln == 0) {
        s_wc[wv] = (uint32_t)__popcll(um);
        s_wf[wv] = um ? wv * WAVE + (uint32_t)__ffsll((long long)um) - 1u : 256u;
    }
    if (t < SZH_MAX_LEN + 2) s_cnt[t] = 0;
    if (t == 0) {
        s_over = 0;
        sp.info->ts[0] = wall_clock64();
    }
    __syncthreads();
    const uint32_t n_un = s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
    const uint32_t rep = min(min(s_wf[0], s_wf[1]), min(s_wf[2], s_wf[3]));
    const uint32_t w = cnt ? 4u * cnt : (t == rep ? n_un : 0u);  // a value not met counts a quarter of an occurrence; their class, all of them
    const unsigned long long pm = __ballot(w != 0u);
    __syncthreads();
    if (ln == 0) s_wc[wv] = (uint32_t)__popcll(pm);
    __syncthreads();
    uint32_t a = (uint32_t)__popcll(pm & lower);
    for (uint32_t k = 0; k < wv; k++) a += s_wc[k];
    const uint32_t m = s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];  // >= 2: a value met and the class, or all 256 values
    const uint64_t mykey = ((uint64_t)w << 16) | t;
    if (w) keys[a] = mykey;
    __syncthreads();
    uint32_t r = 0;
    if (w)
        for (uint32_t o = 0; o < m; o++) r += keys[o] < mykey;  // rank sort (the keys are distinct: the byte value in the low bits)
    __syncthreads();
    if (w) keys[r] = mykey;
    __syncthreads();
    if (t == 0) sp.info->ts[3] = wall_clock64();
    // (cb_small's wave merge: the queues' windows in registers, read with v_readlane. A branch-free form of it — the registers picked by
    // selects on scalar conditions, every pick compare / select / add on the scalar unit — was built in round 6 and is SLOWER: 20 against
    // 14 us for C2's 147 symbols; the scalar unit's hand-overs to and from the vector unit cost more than the branches they replace)
    if (t < WAVE) cb_merge_wave32(keys, reinterpret_cast<uint32_t *>(ifreq), pleaf, pint, m);  // (weights < 2^32: 2^20 sampled values)
    __syncthreads();
    if (t == 0) sp.info->ts[4] = wall_clock64();
    // depth of every internal node (distance to the root, node m - 2) by pointer doubling, as cb_small does
    for (uint32_t q = t; q + 1 < m; q += CB_THREADS) {
        if (q == m - 2) pint[q] = (uint16_t)q;
        aux[q] = q == m - 2 ? 0 : 1;
    }
    __syncthreads();
    for (uint32_t span = 1; span < m && span < 2 * L; span <<= 1) {
        for (uint32_t q = t; q + 1 < m; q += CB_THREADS) {
            const uint16_t j = pint[q];
            const uint32_t sum = (uint32_t)aux[q] + aux[j];
            aux2[q] = (uint16_t)(sum > 0xFFFFu ? 0xFFFFu : sum);
            pint2[q] = pint[j];
        }
        __syncthreads();
        for (uint32_t q = t; q + 1 < m; q += CB_THREADS) {
            aux[q] = aux2[q];
            pint[q] = pint2[q];
        }
        __syncthreads();
    }
    if (t == 0) sp.info->ts[5] = wall_clock64();
    for (uint32_t q = t; q < m; q += CB_THREADS) {  // leaf lengths by sorted position, clamped
        uint32_t l = (uint32_t)aux[pleaf[q]] + 1;
        if (l > L) {
            l = L;
            s_over = 1;
        }
        pleaf[q] = (uint16_t)l;
        atomicAdd(&s_cnt[l], 1u);
    }
    __syncthreads();
    if (s_over) cb_kraft_repair(pleaf, m, s_cnt, L, CB_THREADS, kraft_ws);
    s_l[t] = 0;
    __syncthreads();
    for (uint32_t q = t; q < m; q += CB_THREADS) s_l[(uint32_t)(keys[q] & 0xFFFFu)] = (uint8_t)pleaf[q];
    __syncthreads();
    if (t == 0) sp.info->ts[8] = wall_clock64();
    uint32_t ubits = 0;
    while ((1u << ubits) < n_un) ubits++;
    const uint32_t len = (n_un == 0 || cnt != 0) ? s_l[t] : (uint32_t)s_l[rep] + ubits;  // met by the sample / not met: the class's code word + the index
    __syncthreads();
    s_l[t] = (uint8_t)len;
    // canonical code words by (length, byte value): a ballot per length inside the wave + the earlier waves' counts
    uint32_t rank = 0;
    for (uint32_t l = 1; l <= SZH_MAX_LEN; l++) {
        const unsigned long long lm = __ballot(len == l);
        if (ln == 0) s_lc[wv][l] = (uint32_t)__popcll(lm);
        if (len == l) rank = (uint32_t)__popcll(lm & lower);
    }
    __syncthreads();
    if (t == 0) {
        uint32_t code = 0, maxl = 0, prev = 0;
        for (uint32_t l = 1; l <= SZH_MAX_LEN; l++) {
            const uint32_t c = s_lc[0][l] + s_lc[1][l] + s_lc[2][l] + s_lc[3][l];
            code = (code + prev) << (l > 1 ? 1 : 0);
            s_first[l] = code;
            prev = c;
            if (c) maxl = l;
        }
        sp.info->n_symbols = 256;
        sp.info->max_len = maxl;
        sp.info->sym_min = lo;
        sp.info->sym_count = 256;
        sp.info->win_lo = lo;
        sp.info->reserved = 0;
        sp.info->esc_sym = esc;
    }
    __syncthreads();
    for (uint32_t k = 0; k < wv; k++) rank += s_lc[k][len];
    const uint32_t e = ((s_first[len] + rank) << 5) | len;
    sp.enc[lo + t] = e;
    sp.lens[lo + t] = (uint8_t)len;
    if (t == 255) sp.enc[0] = e;  // the encoder's tables know a listed delta as symbol 0 (byte 255 -> symbol 0 wherever codes are looked up): an alias of the escape's entry
    if (t < 64) {
        const uint32_t w4 = (uint32_t)s_l[4 * t] | ((uint32_t)s_l[4 * t + 1] << 8) | ((uint32_t)s_l[4 * t + 2] << 16) | ((uint32_t)s_l[4 * t + 3] << 24);
        __hip_atomic_store(&sp.words[SZK_SAMP_LENS + t], w4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (the book's tables in the slot are the next launches' to read — a launch boundary away; the lengths the workers of THIS launch want
    // went out as device-scope stores above, and the wait orders them before the word that announces them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        __hip_atomic_store(&sp.words[SZK_SAMP_READY], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sp.info->ts[10] = wall_clock64();
    }
}

// the sample and its book as a launch of their own: behind the two-launch form of stage 1 (a context's first call) and behind the forms
// that do not carry the sampling workgroups themselves. The probe is complete by then: a two-byte stream has no sampled book.
template <typename T>
__global__ __launch_bounds__(256) void k_sample(const T *__restrict__ in, szk_k1_params p) {
    __shared__ __align__(16) uint32_t s_pool[(SAMP_POOL_BYTES + 3) / 4 > 1024 ? (SAMP_POOL_BYTES + 3) / 4 : 1024];
    if (!szk_is_narrow(p.mode)) return;
    if (samp_take<T>(in, p.lat, (uint32_t)p.d[3], (uint32_t)p.d[2], (uint32_t)p.d[1], p.samp.words, blockIdx.x, s_pool)) samp_book(p.samp, p.radius, reinterpret_cast<uint8_t *>(s_pool));
}

// The wide code book's compaction over the whole chip (round 5): workgroup w owns bins [1024 w, 1024 (w + 1)); its keys' place is the
// number of non-empty bins in front of its slice, which it counts itself (up to 63 coalesced loads per thread, all in flight: the
// histogram is 512 KB in the L2s) — no scan across workgroups, no second launch. keys[] = (count << 16) | symbol and syms[] in symbol
// order, ifreq[w] = the slice's sum of counts, ifreq[64] = the number of keys (the range words' count is not used for it: a repeated
// stage 2 runs k_hist_range over words that already hold the first run's), ifreq[65 + 48 w + o] = the slice's keys whose count lies in octave o (round 6).
__global__ __launch_bounds__(1024) void k_cb_compact(const uint64_t *__restrict__ hist, uint64_t *__restrict__ keys, uint16_t *__restrict__ syms,
                                                     uint64_t *__restrict__ part) {
    __shared__ uint32_t s_w[16], s_before, s_oct[48];
    __shared__ unsigned long long s_sum;
    const uint32_t t = threadIdx.x, lane = lane_id(), wv = t / WAVE, w = blockIdx.x;
    if (t == 0) {
        s_before = 0;
        s_sum = 0;
    }
    if (t < 48) s_oct[t] = 0;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t b = 0; b < w; b++) before += hist[b * 1024u + t] != 0 ? 1u : 0u;
    before = wave_sum(before);
    if (lane == 0 && before) atomicAdd(&s_before, before);
    const uint32_t bin = w * 1024u + t;
    const uint64_t f = hist[bin];
    const unsigned long long bal = __ballot(f != 0);
    if (lane == 0) s_w[wv] = (uint32_t)__popcll(bal);
    const uint64_t fs = wave_sum(f);
    if (lane == 0 && fs) atomicAdd(&s_sum, (unsigned long long)fs);
    if (f) atomicAdd(&s_oct[(63 - __clzll((long long)f)) < 47 ? 63 - __clzll((long long)f) : 47], 1u);  // keys per octave of their count (the two-class rule's input: part[65 + 48 w ..])
    __syncthreads();
    if (t < 48) part[gridDim.x + 1 + w * 48u + t] = s_oct[t];
    uint32_t pos = s_before;
    for (uint32_t k = 0; k < wv; k++) pos += s_w[k];
    if (f) {
        const uint32_t at = pos + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        keys[at] = (f << 16) | bin;
        syms[at] = (uint16_t)bin;
    }
    if (t == 0) {
        part[w] = s_sum;
        if (w == gridDim.x - 1) {  // the last slice's place + its own keys = all of them
            uint32_t m = s_before;
            for (uint32_t k = 0; k < 16; k++) m += s_w[k];
            part[gridDim.x] = m;
        }
    }
}
// The wide code book's code words over the whole chip (round 5; cb_assign_codes with the symbols cut into gridDim.x slices): canonical
// codes in (length, symbol) order — a symbol's code word is its length's first code word + the number of symbols of that length in
// front of it. A workgroup counts the lengths in front of its slice itself (len_of is at most 128 KB, in the L2s), then works its
// slice like cb_assign_codes: a contiguous run per thread, per-(length, thread) counts scanned along the threads by a wave per length.
__global__ __launch_bounds__(1024) void k_cb_assign(const uint16_t *__restrict__ len_of, const uint16_t *__restrict__ syms, const szk_cb_info *__restrict__ info,
                                                    uint32_t *__restrict__ enc) {
    constexpr uint32_t NT = 1024;
    __shared__ uint16_t cnt_tbl[(SZH_MAX_LEN + 1) * NT];
    __shared__ uint32_t s_base[SZH_MAX_LEN + 2], s_first[SZH_MAX_LEN + 2];
    if (info->reserved != CB_ASSIGN_PENDING) return;  // (the book was built by another form, or not at all)
    const uint32_t t = threadIdx.x, lane = lane_id(), wv = t / WAVE;
    const uint32_t m = info->n_symbols;
    const uint32_t per_wg = (m + gridDim.x - 1) / gridDim.x;
    const uint32_t q_lo = blockIdx.x * per_wg < m ? blockIdx.x * per_wg : m, q_hi = q_lo + per_wg < m ? q_lo + per_wg : m;
    if (t <= SZH_MAX_LEN) {
        s_first[t] = info->first_code[t];
        s_base[t] = 0;
    }
    for (uint32_t l = 0; l <= SZH_MAX_LEN; l++) cnt_tbl[l * NT + t] = 0;
    __syncthreads();
    // the lengths in front of the slice (a thread's counts are its own column: no conflicts; at most 64 per thread)
    for (uint32_t qb = t; qb < q_lo; qb += 8 * NT) {
        uint16_t l8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) l8[k] = len_of[qb + k * NT < q_lo ? qb + k * NT : q_lo - 1];
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (qb + k * NT < q_lo && l8[k] <= SZH_MAX_LEN) cnt_tbl[(uint32_t)l8[k] * NT + t]++;
    }
    __syncthreads();
    for (uint32_t l = 1 + wv; l <= SZH_MAX_LEN; l += NT / WAVE) {
        uint32_t sum = 0;
        for (uint32_t c0 = 0; c0 < NT; c0 += WAVE) sum += cnt_tbl[l * NT + c0 + lane];
        sum = wave_sum(sum);
        if (lane == 0) s_base[l] = sum;
    }
    __syncthreads();
    for (uint32_t l = 0; l <= SZH_MAX_LEN; l++) cnt_tbl[l * NT + t] = 0;
    __syncthreads();
    const uint32_t n_own = q_hi - q_lo, per = (n_own + NT - 1) / NT;
    const uint32_t r0 = q_lo + (t * per < n_own ? t * per : n_own), r1 = r0 + per < q_hi ? r0 + per : q_hi;
    for (uint32_t q = r0; q < r1; q++) {
        const uint32_t l = len_of[q];
        if (l <= SZH_MAX_LEN) cnt_tbl[l * NT + t]++;
    }
    __syncthreads();
    for (uint32_t l = 1 + wv; l <= SZH_MAX_LEN; l += NT / WAVE) {
        uint32_t carry = 0;
        for (uint32_t c0 = 0; c0 < NT; c0 += WAVE) {
            const uint32_t v = cnt_tbl[l * NT + c0 + lane];
            const uint32_t incl = wave_incl_scan(v);
            cnt_tbl[l * NT + c0 + lane] = (uint16_t)(carry + incl - v);
            carry += __shfl(incl, WAVE - 1, WAVE);
        }
    }
    __syncthreads();
    for (uint32_t q = r0; q < r1; q++) {
        const uint32_t l = len_of[q];
        if (l == 0 || l > SZH_MAX_LEN) continue;
        const uint32_t rank = s_base[l] + cnt_tbl[l * NT + t]++;
        enc[syms[q]] = ((s_first[l] + rank) << 5) | l;
    }
}

// PART 0: alphabets up to CB_SMALL_SYMS symbols + the outlier-list sorts; PART 1: wider alphabets. Which one applies is
// known only on the device (k_hist_range), so both are launched and the other returns at once: the small path keeps its
// own register allocation and instruction footprint (sharing one kernel with the wide path cost it 10 us of its 40).
// which of the two forms a histogram's range words call for (sz3hip_kernels.h, szk_cb_part)
__device__ __forceinline__ int cb_part_of(const uint32_t *range) {
    const uint32_t n = range[2], span = n ? range[1] - (0xFFFFu - range[0]) + 1u : 0u;
    return n <= CB_SMALL_SYMS || (n <= SZK_CB_PART0_SYMS && span <= SZK_CB_PART0_RANGE) ? 0 : 1;
}
template <int PART>
__global__ __launch_bounds__(CB_LAUNCH) void k_codebook(const uint64_t *__restrict__ hist, szk_cb_params p) {
    constexpr uint32_t CAP = CB_LDS_SYMS;  // symbols the small path's LDS arrays hold
    __shared__ __align__(16) uint8_t s_pool[CB_POOL_BYTES];
    __shared__ uint32_t s_wtot[CB_THREADS / WAVE];
    __shared__ uint32_t s_over;
    __shared__ uint32_t s_first[SZH_MAX_LEN + 2], s_cnt[SZH_MAX_LEN + 2];
    __shared__ uint32_t s_misc[8];
    __shared__ unsigned long long s_total;
    const uint32_t t = threadIdx.x;
    if (blockIdx.x >= p.n_books) {  // the two blocks after the code books: deterministic order of the two outlier lists
        if (p.skip_sort) return;
        // (in the launch whose code-book path is the active one, so that they run beside it)
        if (PART != cb_part_of(p.range) && p.part_hint < 0) return;  // (launched alone: sorts whatever the alphabet)
        const bool d = blockIdx.x == p.n_books + 1;
        // scratch: the key tables of the batch slots 1 and 2, idle when a single code book is built (n_books <= 1)
        uint64_t *scratch = p.n_books <= 1 ? p.keys + (size_t)(d ? 2 : 1) * SZH_HIST_BINS : nullptr;
        sort_outlier_list(d ? p.dout_idx : p.vout_idx, d ? p.dout_val : p.vout_val, d ? *p.n_dout : *p.n_vout, p.out_cap,
                          d ? p.q_is_32bit != 0 : p.t_is_32bit != 0, s_pool, scratch);
        return;
    }
    if (p.samp_words && p.n_books <= 1 && __hip_atomic_load(const_cast<uint32_t *>(p.samp_words) + SZK_SAMP_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        // this call's book was built from the sample (samp_book): it is in the slot already — the launch only sorts the lists
        if (PART == 1 && t == 0) p.info->reserved = 0;  // (nothing for k_cb_assign behind this launch)
        return;
    }
    {  // book b of a batch (the tuner's trials) uses the b-th slice of every table
        const size_t b = blockIdx.x;
        hist += b * SZH_HIST_BINS;
        p.enc += b * SZH_HIST_BINS;
        p.lens += b * SZH_HIST_BINS;
        p.keys += b * SZH_HIST_BINS;
        p.syms += b * SZH_HIST_BINS;
        p.ifreq += b * SZH_HIST_BINS;
        p.pleaf += b * SZH_HIST_BINS;
        p.pint += b * SZH_HIST_BINS;
        p.depth += b * SZH_HIST_BINS;
        p.aux2 += b * SZH_HIST_BINS;
        p.pint2 += b * SZH_HIST_BINS;
        p.range += b * 4;
        p.info += b;
    }
    // range and number of the non-empty bins: found by k_hist_range (256 workgroups) just before this launch
    const uint32_t n_nonzero = p.range[2];
    if (PART != cb_part_of(p.range)) {  // the other form's case
        if (p.part_hint >= 0 && p.mispredict && threadIdx.x == 0) *p.mispredict = 1u;  // launched alone: the host repeats stage 2 with both
        if (PART == 1 && threadIdx.x == 0) p.info->reserved = 0;  // (nothing for k_cb_assign behind this launch)
        return;
    }
    if (n_nonzero == 0) {
        if (t == 0) {
            p.info->n_symbols = 0;
            p.info->max_len = 0;
            p.info->sym_min = 0;
            p.info->sym_count = 0;
            p.info->win_lo = 0;
            p.info->reserved = 0;
        }
        return;
    }
    uint32_t lo = 0xFFFFu - p.range[0], range = p.range[1] - lo + 1;  // range[0] = max(65535 - bin), range[1] = max bin
    const bool small = PART == 0;
    const bool fill = PART == 0 && n_nonzero <= CB_SMALL_SYMS && cb_margins(lo, range, n_nonzero);
    if (small && t >= CB_THREADS) return;  // the small path runs on 4 waves (cheap barriers)
    if (t == 0) p.info->ts[0] = wall_clock64();
    if (t < SZH_MAX_LEN + 2) s_cnt[t] = 0;
    if (t == 0) {
        s_over = 0;
        s_total = 0;
    }
    for (uint32_t i = t; i < range; i += small ? CB_THREADS : CB_LAUNCH) {  // only [lo, hi] is ever looked up / serialised
        p.enc[lo + i] = 0;
        p.lens[lo + i] = 0;
    }
    __syncthreads();
    if constexpr (PART == 1) {
        if (t == 0) p.info->ts[1] = wall_clock64();
        codebook_wide<true>(hist, p, s_pool, lo, range, s_cnt, s_first, s_misc, p.keys_ready != 0 && p.n_books <= 1);
        return;
    }
    cb_small<CAP>(hist, p, s_pool, lo, range, s_wtot, s_over, s_first, s_cnt, s_misc, s_total, fill);
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
    o.subbits = off;
    off = szh_align16(off + 2 * (SZH_SUBS - 1) * h.n_chunks);
    o.vout_idx = off;
    off += 8 * h.n_vout;
    o.vout_val = off;
    off = szh_align16(off + tsz * h.n_vout);
    o.dout_idx = off;
    off += 8 * h.n_dout;
    o.dout_val = off;
    off = szh_align16(off + (uint64_t)h.qbytes * h.n_dout);
    o.side = off;
    off = szh_align16(off + (h.predictor == 2 ? h.side_bytes : 0));
    o.bitstream = off;
    o.end = off + 4 * h.bitstream_words;
}

__device__ void layout_pre(const szk_layout_params &p) {  // after K1 + K5, before the packer (one thread)
    szh_header h = p.proto;  // dtype, ndim, dims, eb, radius, n, chunk geometry filled by the host
    uint64_t nv = *p.n_vout, nd = *p.n_dout;
    p.state->overflow = (nv > p.out_cap) || (nd > p.out_cap);
    p.state->n_vout_raw = nv;
    p.state->n_dout_raw = nd;
    p.state->book_miss = p.state->miss_kind = 0;
    if (nv > p.out_cap) nv = p.out_cap;
    if (nd > p.out_cap) nd = p.out_cap;
    h.n_vout = nv;
    h.n_dout = nd;
    h.sym_min = p.info->sym_min;
    h.sym_count = p.info->sym_count;
    h.max_len = p.info->max_len;
    if (h.predictor == 0 && p.info->esc_sym) {  // (Lorenzo streams: the symbol that stands for a listed delta, 0 = symbol 0 itself; sz3hip_format.h)
        h.anchor_stride = p.info->esc_sym;
        h.version = SZH_VERSION_ESC;
    }
    h.side_bytes = p.side_bytes ? *p.side_bytes : 0;
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

__device__ __forceinline__ uint32_t enc_lookup(const uint32_t *s_enc, const uint32_t *__restrict__ g_enc, int win_lo,
                                               uint32_t sym) {
    int rel = (int)sym - win_lo;
    return (rel >= 0 && rel < ENC_WIN) ? s_enc[rel] : g_enc[sym];
}

__device__ __forceinline__ void load_codes16(const uint16_t *__restrict__ codes, uint64_t base, uint64_t n, bool narrow,
                                             uint32_t sym_add, uint16_t (&c)[ENC_PER_LANE]) {
    if (narrow) {  // one byte per code (delta + 128, 0 = outlier) -> symbol
        const uint8_t *c8 = reinterpret_cast<const uint8_t *>(codes);
        if (base + ENC_PER_LANE <= n) {
            const uint4 a = *reinterpret_cast<const uint4 *>(c8 + base);  // base is a multiple of 16
            const uint32_t wds[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t b = (wds[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                c[i] = (uint16_t)(b != 255u ? b + sym_add : 0u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < ENC_PER_LANE; i++) {
                const uint32_t b = (base + i < n) ? c8[base + i] : 255u;
                c[i] = (uint16_t)(b != 255u ? b + sym_add : 0u);
            }
        }
        return;
    }
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
        for (int i = 0; i < ENC_PER_LANE; i++) c[i] = (base + i < n) ? codes[base + i] : (uint16_t)0;
    }
}

// ------------------------------------------------------------------------------------------------------------
// K6: three plain streaming launches, no inter-workgroup dependency.
//   k_chunk_bits2   words per 1024-symbol chunk                      (reads the codes once)
//   k_scan_groups   exclusive word offset of every group of 32 chunks (one workgroup, 4 KiB of offsets per 4M symbols)
//   k_pack          bit-pack: one wave per chunk, 16 symbols per lane; with code words <= 16 bit four of them always
//                   fit one 64-bit register, so a lane emits 4 registers at its bit offset (wave prefix sum) with
//                   three ds_or each into a zeroed LDS stage; the wave then streams the words out coalesced.
// The code table is LDS-resident for alphabets up to ENC_WIN symbols (sym_min .. sym_min + sym_count), which is
// every realistic case; wider alphabets take the per-symbol global lookup.
// ------------------------------------------------------------------------------------------------------------
#define PACK_GROUP 32  // chunks per offset group

template <uint32_t WIN = ENC_WIN>
__device__ __forceinline__ uint32_t enc_lookup2(const uint32_t *s_enc, const uint32_t *__restrict__ g_enc,
                                                uint32_t win_lo, bool all_lds, uint32_t sym) {
    const uint32_t rel = sym - win_lo;
    if (all_lds) return s_enc[rel & (ENC_WIN - 1)];
    return rel < WIN ? s_enc[rel] : g_enc[sym];
}
__device__ __forceinline__ void enc_table_load(uint32_t *s_enc, const uint32_t *__restrict__ g_enc, uint32_t win_lo,
                                               uint32_t sym_count) {
    const uint32_t cnt = sym_count < ENC_WIN ? sym_count : ENC_WIN;
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) s_enc[i] = g_enc[win_lo + i];
}

// raw code registers of one lane (16 symbols: 32 bytes as two-byte codes, 16 bytes as one-byte codes) — loaded
// unconditionally one chunk ahead so that the HBM latency hides behind the current chunk's work
struct CodeRegs {
    uint4 a, b;
};
__device__ __forceinline__ void fetch_codes(const uint16_t *__restrict__ codes, uint64_t base, bool narrow, CodeRegs &r) {
    if (narrow) {
#ifdef LAB_PACK_NTL  // (lab: the codes are read once — streaming loads)
        typedef uint32_t u4v __attribute__((ext_vector_type(4)));
        const u4v q = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(reinterpret_cast<const uint8_t *>(codes) + base));
        r.a = make_uint4(q.x, q.y, q.z, q.w);
#else
        r.a = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(codes) + base);
#endif
    } else {
        const uint4 *v = reinterpret_cast<const uint4 *>(codes + base);
        r.a = v[0];
        r.b = v[1];
    }
}
__device__ __forceinline__ void unpack_codes(const CodeRegs &r, bool narrow, uint32_t sym_add, uint16_t (&c)[ENC_PER_LANE]) {
    if (narrow) {
        const uint32_t wds[4] = {r.a.x, r.a.y, r.a.z, r.a.w};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t b = (wds[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            c[i] = (uint16_t)(b != 255u ? b + sym_add : 0u);
        }
    } else {
        const uint32_t wds[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c[2 * i] = (uint16_t)(wds[i] & 0xFFFF);
            c[2 * i + 1] = (uint16_t)(wds[i] >> 16);
        }
    }
}

// persistent: every wave walks chunks wave_gid, +nwaves, ...; full chunks take the prefetching path, the ragged last
// chunk (if any) the bounds-checked loader
__global__ __launch_bounds__(256) void k_chunk_bits2(const uint16_t *__restrict__ codes, uint64_t n,
                                                     const uint32_t *__restrict__ g_enc,
                                                     const szk_cb_info *__restrict__ info, szk_mode mode, uint32_t sym_add,
                                                     uint16_t *__restrict__ chunk_words, uint16_t *__restrict__ sub_bits) {
    // code lengths only: one byte per symbol, so the LDS table covers 16384 symbols around the most frequent one (the
    // packers' 4-byte entries cover 4096): C4's deltas (std 2600 lattice steps) miss a 4096-symbol window 44 % of the time
    constexpr uint32_t LEN_WIN = 4 * ENC_WIN;
    __shared__ uint8_t s_lenw[LEN_WIN];
    __shared__ uint8_t s_len8[256];  // one-byte codes: code length by byte value (no symbol arithmetic per element)
    const bool narrow = szk_is_narrow(mode);
    const uint64_t n_full = n / SZH_CHUNK_SYMS, n_chunks = (n + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS;
    const uint64_t wave_gid = (uint64_t)blockIdx.x * 4 + threadIdx.x / WAVE, nwaves = (uint64_t)gridDim.x * 4;
    const uint64_t lane_off = (uint64_t)lane_id() * ENC_PER_LANE;
    CodeRegs cur, nxt;
    uint64_t chunk = wave_gid;
    if (chunk < n_full) fetch_codes(codes, chunk * SZH_CHUNK_SYMS + lane_off, narrow, cur);  // in flight during the table load
    // window start: the packers' window is centred on the most frequent symbol; widen it symmetrically, inside [0, 65536)
    const uint32_t centre = info->win_lo + ENC_WIN / 2;
    uint32_t lw_lo = centre > LEN_WIN / 2 ? centre - LEN_WIN / 2 : 0u;
    if (lw_lo > SZH_HIST_BINS - LEN_WIN) lw_lo = SZH_HIST_BINS - LEN_WIN;
    if (!narrow || n_full < n_chunks)  // (the ragged last chunk goes through the symbol table in both modes)
        for (uint32_t i = threadIdx.x; i < LEN_WIN; i += 256) s_lenw[i] = (uint8_t)(g_enc[lw_lo + i] & 31u);
    if (narrow) s_len8[threadIdx.x] = (uint8_t)(g_enc[threadIdx.x != 255u ? threadIdx.x + sym_add : 0u] & 31u);
    __syncthreads();
    auto len_of = [&](uint32_t sym) -> uint32_t {
        const uint32_t rel = sym - lw_lo;
        return rel < LEN_WIN ? (uint32_t)s_lenw[rel] : (g_enc[sym] & 31u);
    };
    // a chunk's word count and the bit offsets of its later units (the decoder's restart points, sz3hip_format.h: subbits): the
    // running sum over the lanes, read at the units' last lanes
    auto emit = [&](uint64_t ch, uint32_t bits) {
        const uint32_t incl = wave_incl_scan(bits);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
        uint32_t sub[SZH_SUBS - 1];
#pragma unroll
        for (uint32_t u = 1; u < SZH_SUBS; u++) sub[u - 1] = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(u * (WAVE / SZH_SUBS) - 1));
        if (lane_id() == 0) {
            chunk_words[ch] = (uint16_t)((tot + 31) >> 5);
            if (sub_bits)
#pragma unroll
                for (uint32_t u = 1; u < SZH_SUBS; u++) sub_bits[ch * (SZH_SUBS - 1) + u - 1] = (uint16_t)sub[u - 1];
        }
    };
    if (narrow) {
        for (; chunk < n_full; chunk += nwaves) {
            const uint64_t nc = chunk + nwaves;
            if (nc < n_full) fetch_codes(codes, nc * SZH_CHUNK_SYMS + lane_off, narrow, nxt);
            const uint32_t wds[4] = {cur.a.x, cur.a.y, cur.a.z, cur.a.w};
            uint32_t bits = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                bits += (uint32_t)s_len8[wds[k] & 0xFFu] + s_len8[(wds[k] >> 8) & 0xFFu] + s_len8[(wds[k] >> 16) & 0xFFu] + s_len8[wds[k] >> 24];
            emit(chunk, bits);
            cur = nxt;
        }
    } else {
        for (; chunk < n_full; chunk += nwaves) {
            const uint64_t nc = chunk + nwaves;
            if (nc < n_full) fetch_codes(codes, nc * SZH_CHUNK_SYMS + lane_off, narrow, nxt);
            uint16_t c[ENC_PER_LANE];
            unpack_codes(cur, narrow, sym_add, c);
            // (the packer's form of this lookup — a plain LDS read modulo the window, the rest behind one wave-uniform test — measured slower here: 52.7 against 46 us at C3)
            uint32_t bits = 0;
#pragma unroll
            for (int i = 0; i < ENC_PER_LANE; i++) bits += len_of(c[i]);
            emit(chunk, bits);
            cur = nxt;
        }
    }
    if (n_full < n_chunks && wave_gid == 0) {  // ragged tail
        const uint64_t base = n_full * SZH_CHUNK_SYMS + lane_off;
        uint16_t c[ENC_PER_LANE];
        load_codes16(codes, base, n, narrow, sym_add, c);
        uint32_t bits = 0;
#pragma unroll
        for (int i = 0; i < ENC_PER_LANE; i++) {
            const uint32_t e = len_of(c[i]);
            bits += (base + i < n) ? e : 0u;
        }
        emit(n_full, bits);
    }
}

// offsets of the 32-chunk groups (exclusive scan of the groups' word counts) and the total; one workgroup, four
// consecutive groups per thread and round. The encoder's call also lays the payload out (layout_pre: the
// sections' offsets depend on the outlier counts and the alphabet, known since the code book kernel) — one launch less.
#define SCAN_GPT 4
// SEG: `seg_bits` holds one u32 per group — its word count, summed by k_seg_chunks (which also wrote the chunks' counts).
template <bool SEG>
__device__ void scan_groups_body(uint16_t *__restrict__ chunk_words, uint64_t n_chunks, uint64_t *__restrict__ group_off,
                                 uint64_t *total_words, const uint16_t *__restrict__ seg_bits, uint64_t n_segs, const uint32_t *seg_made) {
    __shared__ uint64_t s_w[16];
    __shared__ uint64_t s_carry;
    const uint64_t n_groups = (n_chunks + PACK_GROUP - 1) / PACK_GROUP;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t g0 = 0; g0 < n_groups; g0 += 1024 * SCAN_GPT) {
        const uint64_t gt = g0 + (uint64_t)threadIdx.x * SCAN_GPT;
        uint64_t sum[SCAN_GPT];
#pragma unroll
        for (int j = 0; j < SCAN_GPT; j++) {
            const uint64_t g = gt + j;
            sum[j] = 0;
            if (SEG) {  // the groups' word counts were summed by k_seg_chunks
                sum[j] = g < n_groups ? reinterpret_cast<const uint32_t *>(seg_bits)[g] : 0u;
            } else if (g < n_groups) {
                const uint64_t c0 = g * PACK_GROUP;
                if (c0 + PACK_GROUP <= n_chunks) {  // 32 x u16 = four 16-byte loads
                    const uint4 *v = reinterpret_cast<const uint4 *>(chunk_words + c0);
                    uint4 q[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        sum[j] += (q[k].x & 0xFFFF) + (q[k].x >> 16) + (q[k].y & 0xFFFF) + (q[k].y >> 16) + (q[k].z & 0xFFFF) +
                                  (q[k].z >> 16) + (q[k].w & 0xFFFF) + (q[k].w >> 16);
                } else {
                    for (uint64_t c = c0; c < n_chunks; c++) sum[j] += chunk_words[c];
                }
            }
        }
        uint64_t mine = 0;
#pragma unroll
        for (int j = 0; j < SCAN_GPT; j++) mine += sum[j];
        const uint64_t incl = wave_incl_scan(mine);
        if (lane_id() == WAVE - 1) s_w[threadIdx.x / WAVE] = incl;
        __syncthreads();
        uint64_t run = s_carry + incl - mine, tot = 0;
        for (int w = 0; w < 16; w++) {
            if (w < (int)(threadIdx.x / WAVE)) run += s_w[w];
            tot += s_w[w];
        }
#pragma unroll
        for (int j = 0; j < SCAN_GPT; j++) {
            if (gt + j < n_groups) group_off[gt + j] = run;
            run += sum[j];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_words = s_carry;
}
// Stage 1 summed the code bits of every 256-element row segment (march_narrow); a chunk is four consecutive segments. This
// launch turns them into the chunks' word counts (what the encoder's bits pass produces) and the groups' sums for the
// offset scan, all over the chip: one 16-byte load = 8 segments = 2 chunks per lane (coalesced), 16 consecutive lanes = one
// group of 32 chunks, summed by a DPP reduction over the row of 16 lanes. *seg_made == 0 (stage 1 ran another form than the
// host assumed): all counts are zero — the packer's output is thrown away anyway, it only must stay inside the payload.
__global__ __launch_bounds__(256) void k_seg_chunks(const uint16_t *__restrict__ seg_bits, uint64_t n_segs, const uint32_t *seg_made,
                                                    uint16_t *__restrict__ chunk_words, uint64_t n_chunks, uint32_t *__restrict__ group_sums,
                                                    uint16_t *__restrict__ sub_bits) {
    const bool seg_ok = *seg_made != 0;
    const uint64_t n_vec = (n_segs + 7) / 8;
    const uint64_t n_vec16 = (n_vec + 15) / 16 * 16;  // whole groups
    for (uint64_t vi = (uint64_t)blockIdx.x * 256 + threadIdx.x; vi < n_vec16; vi += (uint64_t)gridDim.x * 256) {
        const uint4 q = vi < n_vec ? reinterpret_cast<const uint4 *>(seg_bits)[vi] : make_uint4(0, 0, 0, 0);
        const uint64_t sg0 = vi * 8;
        uint32_t hw[8] = {q.x & 0xFFFF, q.x >> 16, q.y & 0xFFFF, q.y >> 16, q.z & 0xFFFF, q.z >> 16, q.w & 0xFFFF, q.w >> 16};
#pragma unroll
        for (int k = 0; k < 8; k++) hw[k] = (seg_ok && sg0 + k < n_segs) ? hw[k] : 0u;
        const uint32_t cw0 = (hw[0] + hw[1] + hw[2] + hw[3] + 31u) >> 5, cw1 = (hw[4] + hw[5] + hw[6] + hw[7] + 31u) >> 5;
        const uint64_t c = vi * 2;
        if (c + 1 < n_chunks) reinterpret_cast<uint32_t *>(chunk_words)[vi] = cw0 | (cw1 << 16);
        else if (c < n_chunks) chunk_words[c] = (uint16_t)cw0;
        if (sub_bits) {  // the units' bit offsets inside their chunks: running sums of the chunk's segments (a unit = 4 / SZH_SUBS segments)
            constexpr uint32_t SPU = 4 / SZH_SUBS;
#pragma unroll
            for (uint32_t h = 0; h < 2; h++) {
                uint32_t run = 0;
#pragma unroll
                for (uint32_t u = 1; u < SZH_SUBS; u++) {
#pragma unroll
                    for (uint32_t q = 0; q < SPU; q++) run += hw[4 * h + (u - 1) * SPU + q];
                    if (c + h < n_chunks) sub_bits[(c + h) * (SZH_SUBS - 1) + u - 1] = (uint16_t)run;
                }
            }
        }
        uint32_t gs = (c < n_chunks ? cw0 : 0u) + (c + 1 < n_chunks ? cw1 : 0u);
        gs += dpp_mov0<0x111, 0xf>(gs);  // inclusive sum along the row of 16 lanes: its last lane holds the group's total
        gs += dpp_mov0<0x112, 0xf>(gs);
        gs += dpp_mov0<0x114, 0xf>(gs);
        gs += dpp_mov0<0x118, 0xf>(gs);
        if ((lane_id() & 15) == 15) group_sums[vi >> 4] = gs;
    }
}
struct szk_fold_params {  // the fold of stage 1's histogram rows, riding in the scan's launch (blocks 1 .. 64) when stage 1 left it out
    const uint32_t *partial;
    uint32_t nrows;
    int win_lo;
    uint64_t *hist;
    uint32_t *range;
};
__global__ __launch_bounds__(1024) void k_scan_groups(uint16_t *__restrict__ chunk_words, uint64_t n_chunks,
                                                      uint64_t *__restrict__ group_off, uint64_t *total_words,
                                                      szk_layout_params lp, int do_layout, const uint16_t *__restrict__ seg_bits, uint64_t n_segs,
                                                      const uint32_t *seg_made, szk_fold_params fp) {
    if (blockIdx.x > 0) {  // k_hist_reduce's work: block b sums rows b - 1, b - 1 + 64, ...; 1024 threads = the 1024 bins of a row
        const int bin = threadIdx.x;
        uint64_t sum = 0;
        for (uint32_t r = blockIdx.x - 1; r < fp.nrows; r += gridDim.x - 1) sum += fp.partial[(uint64_t)r * HIST_WIN + bin];
        const int sym = fp.win_lo + bin;
        if (sum && sym >= 0 && sym < (int)SZH_HIST_BINS) hist_add_ranged(fp.hist, fp.range, (uint32_t)sym, (unsigned long long)sum);
        return;
    }
    if (do_layout && threadIdx.x == 1023) layout_pre(lp);
    if (seg_bits) scan_groups_body<true>(chunk_words, n_chunks, group_off, total_words, seg_bits, n_segs, seg_made);
    else scan_groups_body<false>(chunk_words, n_chunks, group_off, total_words, nullptr, 0, nullptr);
}

// pack 16 symbols of a lane into the wave's LDS stage at the lane's bit offset; returns the chunk's word count.
// G code words are joined per 64-bit register: G = 4 when the code book's longest word is <= 16 bits, else G = 2
// (<= 24 bits each); every register is emitted left-aligned at its bit offset with three ds_or.
template <int G, bool BYTE = false, uint32_t WIN = ENC_WIN>  // BYTE: c[] are one-byte codes; s_enc[0..255] = code word, s_len8 = its length, by byte value
__device__ __forceinline__ uint32_t pack_chunk(const uint16_t (&c)[ENC_PER_LANE], uint64_t base, uint64_t n, bool check_n,
                                               const uint32_t *s_enc, const uint32_t *__restrict__ g_enc,
                                               uint32_t sym_min, bool all_lds, uint32_t *stage, const uint8_t *s_len8 = nullptr) {
    constexpr int NG = ENC_PER_LANE / G;
    uint64_t g[NG];
    uint32_t gl[NG];
    uint32_t bits = 0;
    // two-byte codes: all sixteen table entries first, straight from the LDS window (index taken modulo the window: a plain ds_read — the
    // select between an LDS and a global address per symbol had become a flat load behind two wave-uniform branches, thirty-two branches a
    // chunk); the symbols outside the window are fetched from the table in memory afterwards, behind ONE wave-uniform test per chunk
    uint32_t ew[BYTE ? 1 : ENC_PER_LANE];
    if (!BYTE) {
        bool outside = false;
#pragma unroll
        for (int i = 0; i < ENC_PER_LANE; i++) {
            const uint32_t rel = (uint32_t)c[i] - sym_min;
            outside |= rel >= WIN;
            ew[i] = s_enc[rel & (WIN - 1)];
        }
        if (!all_lds && __builtin_amdgcn_ballot_w64(outside)) {
#pragma unroll
            for (int i = 0; i < ENC_PER_LANE; i++) {
                const uint32_t rel = (uint32_t)c[i] - sym_min;
                if (rel >= WIN) ew[i] = g_enc[c[i]];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NG; k++) {
        // two code words are joined with 32-bit ops when they are <= 16 bits each, two pairs with one 64-bit shift
        uint64_t pc[G / 2];
        uint32_t pl[G / 2];
#pragma unroll
        for (int h = 0; h < G / 2; h++) {
            if (BYTE) {  // code word and length come from separate tables: no shift / mask per symbol
                const uint32_t b0 = c[G * k + 2 * h], b1 = c[G * k + 2 * h + 1];
#if defined(LAB_PACK) && (LAB_PACK & 1)  // (lab, wrong results: no table lookups)
                pc[h] = ((b0 & 15u) << 4) | (b1 & 15u);
                pl[h] = 8;
#else
                const uint32_t l1 = s_len8[b1];
                pc[h] = (s_enc[b0] << l1) | s_enc[b1];
                pl[h] = (uint32_t)s_len8[b0] + l1;
#endif
                continue;
            }
            uint32_t e0 = ew[G * k + 2 * h], e1 = ew[G * k + 2 * h + 1];
            if (check_n) {
                e0 = (base + G * k + 2 * h < n) ? e0 : 0u;
                e1 = (base + G * k + 2 * h + 1 < n) ? e1 : 0u;
            }
            const uint32_t l1 = e1 & 31u;
            if (G == 4) pc[h] = ((e0 >> 5) << l1) | (e1 >> 5);
            else pc[h] = ((uint64_t)(e0 >> 5) << l1) | (e1 >> 5);
            pl[h] = (e0 & 31u) + l1;
        }
        if (G == 4) {
            g[k] = (pc[0] << pl[G / 2 - 1]) | pc[G / 2 - 1];
            gl[k] = pl[0] + pl[G / 2 - 1];
        } else {
            g[k] = pc[0];
            gl[k] = pl[0];
        }
        bits += gl[k];
    }
    const uint32_t incl = wave_incl_scan(bits);
    const uint32_t total_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
    uint32_t pos = incl - bits;
    if (BYTE && G == 4) {
        // one-byte codes: a lane's sixteen symbols are ~66 bits on a smooth field — two registers of eight symbols instead of four
        // of four halve the emission work whenever every lane's pairs of registers fit 64 bits (a wave-uniform test)
        const bool fits = gl[0] + gl[1] <= 64u && gl[2] + gl[3] <= 64u;
        if (!__builtin_amdgcn_ballot_w64(!fits)) {
#pragma unroll
            for (int k = 0; k < NG; k += 2) {
                const uint32_t len = gl[k] + gl[k + 1];
                const uint64_t joined = (gl[k + 1] < 64u ? g[k] << gl[k + 1] : 0ull) | g[k + 1];
                const uint64_t v = len ? joined << (64 - len) : 0ull;  // left-aligned
                const uint32_t word = pos >> 5, sh = pos & 31;
                const uint64_t t = v >> sh;
#if defined(LAB_PACK) && (LAB_PACK & 2)  // (lab, wrong results: no emission into the stage)
                if (t == 0x123456789ull) stage[word] = 1;
#else
                atomicOr(&stage[word], (uint32_t)(t >> 32));
                atomicOr(&stage[word + 1], (uint32_t)t);
                atomicOr(&stage[word + 2], (uint32_t)(((uint64_t)(uint32_t)v << 32) >> sh));
#endif
                pos += len;
            }
            return (total_bits + 31) >> 5;
        }
    }
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const uint64_t v = gl[k] ? g[k] << (64 - gl[k]) : 0ull;  // left-aligned
        const uint32_t word = pos >> 5, sh = pos & 31;
        const uint64_t t = v >> sh;
        const uint32_t w2 = (uint32_t)(((uint64_t)(uint32_t)v << 32) >> sh);
        atomicOr(&stage[word], (uint32_t)(t >> 32));
        atomicOr(&stage[word + 1], (uint32_t)t);
        if (G == 4 || gl[k] + sh > 64) atomicOr(&stage[word + 2], w2);
        pos += gl[k];
    }
    return (total_bits + 31) >> 5;
}

// header + side sections (lens, chunk table, outliers) into the payload. Everything it reads is final before the packer
// starts (outlier counts and alphabet since the code book, chunk table and total since the offset scan), so it runs as a few
// extra workgroups of the packer's launch instead of a launch of its own.
__device__ void assemble_body(const szk_asm_params &p, uint64_t tid, uint64_t nth) {
    // (the header and the offsets live in LDS, not in private memory: as locals their arrays went to scratch, and a kernel that
    // uses scratch at all pays for it in every wave — the packer this body rides on ran 213 -> 288 us at C4's slab)
    __shared__ szh_header s_h0, s_h;
    __shared__ szh_offsets s_o, s_oo;
    if (threadIdx.x == 0) {
        s_h0 = p.state->hdr;
        s_o = p.state->off;
    }
    __syncthreads();
    const szh_header &h0 = s_h0;
    const szh_offsets &o = s_o;
    if (tid == 0) {
        szh_header &h = s_h;
        szh_offsets &oo = s_oo;
        h = s_h0;
        h.bitstream_words = *p.total_words;
        szh_compute_offsets(h, oo);
        h.payload_bytes = oo.end;
        *reinterpret_cast<szh_header *>(p.payload) = h;
        p.state->hdr = h;
        p.state->off = oo;
        p.state->cap_exceeded = oo.end > p.cap;
        for (int i = 0; i < 6; i++) p.state->probe[i] = reinterpret_cast<const uint32_t *>(p.n_vout + 4)[i];
        // stage 1 assumed one-byte codes (one-launch form) and this call's probe says two: everything since is void
        if (p.assumed_narrow && !szk_is_narrow(p.mode)) atomicOr(&p.state->miss_kind, 32u);
        if (p.q16_flag && *p.q16_flag) atomicOr(&p.state->miss_kind, 128u);  // the 16-bit stage 1 met a value it does not take: the call is repeated
        if (p.samp_words && p.samp_words[SZK_SAMP_READY]) {  // the call codes with its sampled book: nothing to verify, no form to mispredict
            p.state->mispredict = 0;
            p.state->book_miss = 0;
            p.state->n_symbols = 256;
        } else if (!p.lists_by_roles) {  // (role mode: the book role of the same launch writes these; miss_kind was zeroed by layout_pre)
            p.state->mispredict = (uint32_t)p.n_vout[7];                                  // (d_counters[7]: raised by a code-book form launched alone)
            p.state->book_miss = 0;
            p.state->n_symbols = reinterpret_cast<const uint32_t *>(p.n_vout + 8)[2];   // (the range words: number of non-empty bins)
        }
        // alignment gaps between the sections are part of the payload: zero them so that it is a pure function of the input
        const uint64_t tsz0 = h.dtype == 0 ? 4 : 8;
        for (uint64_t a = oo.lens + h.sym_count; a < oo.chunkwords; a++) p.payload[a] = 0;
        for (uint64_t a = oo.chunkwords + 2 * h.n_chunks; a < oo.subbits; a++) p.payload[a] = 0;
        for (uint64_t a = oo.subbits + 2 * (SZH_SUBS - 1) * h.n_chunks; a < oo.vout_idx; a++) p.payload[a] = 0;
        for (uint64_t a = oo.vout_val + tsz0 * h.n_vout; a < oo.dout_idx; a++) p.payload[a] = 0;
        for (uint64_t a = oo.dout_val + (uint64_t)h.qbytes * h.n_dout; a < oo.side; a++) p.payload[a] = 0;
        for (uint64_t a = oo.side + (h.predictor == 2 ? h.side_bytes : 0); a < oo.bitstream; a++) p.payload[a] = 0;
    }
    for (uint64_t i = tid; i < h0.sym_count; i += nth) p.payload[o.lens + i] = p.lens[h0.sym_min + i];
    uint16_t *cw = reinterpret_cast<uint16_t *>(p.payload + o.chunkwords);
    for (uint64_t i = tid; i < h0.n_chunks; i += nth) cw[i] = p.chunk_words[i];
    uint16_t *sb = reinterpret_cast<uint16_t *>(p.payload + o.subbits);
    for (uint64_t i = tid; i < h0.n_chunks * (SZH_SUBS - 1); i += nth) sb[i] = p.sub_bits[i];
}
// the sections whose size the input decides — the two outlier lists and the block path's side information: copied by every
// thread the launch has (a field with a fill-value mask lists a million points; 32 workgroups copying them byte by byte took
// 0.44 ms at 512^3 f32 / 1e-6). Element-wide copies: the sections start on 8-byte boundaries.
__device__ void assemble_lists(const szk_asm_params &p, uint64_t tid, uint64_t nth) {
    const uint64_t n_vout = p.state->hdr.n_vout, n_dout = p.state->hdr.n_dout;
    const uint32_t dtype = p.state->hdr.dtype, qbytes = p.state->hdr.qbytes, predictor = p.state->hdr.predictor;
    const uint64_t side_bytes = p.state->hdr.side_bytes;
    const uint64_t o_vi = p.state->off.vout_idx, o_vv = p.state->off.vout_val, o_di = p.state->off.dout_idx, o_dv = p.state->off.dout_val,
                   o_side = p.state->off.side;
    uint64_t *vi = reinterpret_cast<uint64_t *>(p.payload + o_vi);
    uint64_t *di = reinterpret_cast<uint64_t *>(p.payload + o_di);
    for (uint64_t i = tid; i < n_vout; i += nth) vi[i] = p.vout_idx[i];
    for (uint64_t i = tid; i < n_dout; i += nth) di[i] = p.dout_idx[i];
    if (dtype == 0) {
        uint32_t *d = reinterpret_cast<uint32_t *>(p.payload + o_vv);
        for (uint64_t i = tid; i < n_vout; i += nth) d[i] = reinterpret_cast<const uint32_t *>(p.vout_val)[i];
    } else {
        uint64_t *d = reinterpret_cast<uint64_t *>(p.payload + o_vv);
        for (uint64_t i = tid; i < n_vout; i += nth) d[i] = reinterpret_cast<const uint64_t *>(p.vout_val)[i];
    }
    if (qbytes == 4) {
        uint32_t *d = reinterpret_cast<uint32_t *>(p.payload + o_dv);
        for (uint64_t i = tid; i < n_dout; i += nth) d[i] = reinterpret_cast<const uint32_t *>(p.dout_val)[i];
    } else if (qbytes == 8) {
        uint64_t *d = reinterpret_cast<uint64_t *>(p.payload + o_dv);
        for (uint64_t i = tid; i < n_dout; i += nth) d[i] = reinterpret_cast<const uint64_t *>(p.dout_val)[i];
    } else {
        for (uint64_t i = tid; i < n_dout * qbytes; i += nth) p.payload[o_dv + i] = ((const uint8_t *)p.dout_val)[i];
    }
    if (predictor == 2 && p.side) {
        const uint64_t nw = side_bytes / 4;  // (the side buffer and its section are 16-byte aligned)
        uint32_t *d = reinterpret_cast<uint32_t *>(p.payload + o_side);
        for (uint64_t i = tid; i < nw; i += nth) d[i] = reinterpret_cast<const uint32_t *>(p.side)[i];
        for (uint64_t i = nw * 4 + tid; i < side_bytes; i += nth) p.payload[o_side + i] = p.side[i];
    }
}
__global__ __launch_bounds__(256) void k_assemble(szk_asm_params p) {
    assemble_body(p, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
    assemble_lists(p, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

// persistent like k_chunk_bits2: a wave owns a private LDS stage; per chunk it zeroes the words it will use, packs,
// and streams them out; the next chunk's codes, word count and group offset are already in flight
// WIN: symbols of the encode table cached in LDS around the most frequent one: ENC_WIN (30 KB of LDS, 5 workgroups per CU)
// or 2 * ENC_WIN (46 KB, 3 per CU) for alphabets that spread wider (chosen per context from the previous call's alphabet)
// Speculative stage 2, small alphabets (sz3hip_api.cpp): the packer runs with the previous call's code book, and THIS call's
// book is built meanwhile by one workgroup of the packer's own launch (the first one dispatched) — no second stream, no
// events: cross-stream dependencies cost more than the code book itself on this runtime. Two more workgroups sort the two
// outlier lists (short ones: <= ROLE_SORT_MAX records each) and copy them into the payload.
#define ROLE_BLOCKS 3u
#define ROLE_SORT_MAX 2048u
struct szk_role_params {
    uint32_t on;  // 0: no role blocks in this launch
    uint32_t no_book;  // the sort roles only
    const uint64_t *hist;
    szk_cb_params cb;                // this call's book (fresh slot), part_hint = 0
    const szk_cb_info *used;         // the book the packer runs with
    const uint8_t *used_lens;
    uint32_t *flags;                 // [0] raised by a list too long for the short sort, [1] stage 1 summed the segments' bits
    int need_seg;
    int exact;                       // the verdict accepts the used book only when it IS this call's book (book_rejected)
};
// The verdict on the book the encoder ran with (`used`, the context's previous book) against the book this call's histogram gives
// (`fresh`). exact: the two must be the same book (a function of the code lengths: same range, same lengths) — the payload is then
// a pure function of the input. Otherwise (round 4) the used book stands when it is a COMPLETE code over this call's alphabet —
// every symbol that occurs has a code word in it — and the size it codes this call's symbols to is within 1/1024 of the fresh
// book's: the format stores code lengths, so a decoder is unaware, and a series of similar arrays keeps its shortcuts although
// no two histograms are equal (the reference builds a tree per call, encoder/HuffmanEncoder.hpp:96-105: its ratio is matched,
// not its schedule). s_acc: three zeroed 64-bit words in LDS; every thread of the workgroup calls; returns true = rejected.
__device__ bool book_rejected(const uint64_t *__restrict__ hist, const szk_cb_info *fresh, const uint8_t *__restrict__ fresh_lens,
                              const szk_cb_info *used, const uint8_t *__restrict__ used_lens, bool exact, uint32_t t, uint32_t nt,
                              unsigned long long *s_acc) {
    const uint32_t lo = fresh->sym_min, cnt = fresh->sym_count;
    if (exact) {
        bool diff = used->sym_min != lo || used->sym_count != cnt || used->max_len != fresh->max_len || used->n_symbols != fresh->n_symbols;
        if (!diff)
            for (uint32_t i = t; i < cnt; i += nt) diff |= used_lens[lo + i] != fresh_lens[lo + i];
        if (diff) atomicOr(&s_acc[0], 1ull);
    } else {
        const uint32_t ulo = used->sym_min, ucnt = used->sym_count;
        unsigned long long cu = 0, cf = 0, missing = 0;
        for (uint32_t i = t; i < cnt; i += nt) {
            const unsigned long long h = hist[lo + i];
            if (!h) continue;
            const uint32_t sym = lo + i;
            const uint32_t lu = (sym >= ulo && sym - ulo < ucnt) ? used_lens[sym] : 0u;  // (outside its range the slot may hold an older book's lengths)
            missing |= lu == 0u;
            cu += h * lu;
            cf += h * fresh_lens[sym];
        }
        cu = wave_sum(cu);
        cf = wave_sum(cf);
        const bool miss_any = __ballot(missing != 0) != 0;
        if (lane_id() == 0) {
            if (miss_any) atomicOr(&s_acc[0], 1ull);
            atomicAdd(&s_acc[1], cu);
            atomicAdd(&s_acc[2], cf);
        }
    }
    __syncthreads();
    if (s_acc[0]) return true;
    return !exact && s_acc[1] * 1024ull > s_acc[2] * 1025ull;
}
// book role: the small-alphabet code book from this call's histogram, then the verdict on the book the packer is using
__device__ void role_book(const szk_role_params &rp, szk_state *state, uint8_t *pool) {
    __shared__ uint32_t s_wtot[CB_THREADS / WAVE];
    __shared__ uint32_t s_over, s_diff;
    __shared__ uint32_t s_first[SZH_MAX_LEN + 2], s_cnt[SZH_MAX_LEN + 2];
    __shared__ uint32_t s_misc[8];
    __shared__ unsigned long long s_total;
    const szk_cb_params &p = rp.cb;
    const uint32_t t = threadIdx.x;
    const uint32_t n_nonzero = p.range[2];
    bool built = false;
    if (n_nonzero > CB_SMALL_SYMS) {
        if (t == 0) *p.mispredict = 1u;  // the other form's alphabet: the host repeats stage 2 the classic way
    } else if (n_nonzero == 0) {
        if (t == 0) {
            p.info->n_symbols = 0;
            p.info->max_len = 0;
            p.info->sym_min = 0;
            p.info->sym_count = 0;
            p.info->win_lo = 0;
            p.info->reserved = 0;
        }
        built = true;
    } else {
        uint32_t lo = 0xFFFFu - p.range[0], range = p.range[1] - lo + 1;
        const bool fill = cb_margins(lo, range, n_nonzero);
        if (t < SZH_MAX_LEN + 2) s_cnt[t] = 0;
        if (t == 0) {
            s_over = 0;
            s_total = 0;
            p.info->ts[0] = wall_clock64();
        }
        for (uint32_t i = t; i < range; i += CB_THREADS) {
            p.enc[lo + i] = 0;
            p.lens[lo + i] = 0;
        }
        __syncthreads();
        cb_small<CB_SMALL_SYMS>(rp.hist, p, pool, lo, range, s_wtot, s_over, s_first, s_cnt, s_misc, s_total, fill);
        built = true;
    }
    __shared__ unsigned long long s_acc[3];
    if (t == 0) s_diff = 0;
    if (t < 3) s_acc[t] = 0;
    __syncthreads();  // (the block's own global writes of info / lens are visible to it after the barrier)
    if (built && book_rejected(rp.hist, p.info, p.lens, rp.used, rp.used_lens, rp.exact != 0, t, CB_THREADS, s_acc) && t == 0) s_diff = 1;
    __syncthreads();
    if (t == 0) {
        const uint32_t kind = (s_diff ? 1u : 0u) | (!built ? 2u : 0u) | (rp.need_seg && !rp.flags[1] ? 8u : 0u);
        // (kind 4, a list too long for the sort roles, is OR-ed in by those roles: atomics on the same word)
        atomicOr(&state->miss_kind, kind);
        state->mispredict = built ? 0u : 1u;
        state->n_symbols = n_nonzero;
    }
}
// sort role: one outlier list (short) into index order, then into its section of the payload
__device__ void role_sort(const szk_role_params &rp, const szk_asm_params &ap, bool d, uint8_t *pool) {
    const szk_cb_params &p = rp.cb;
    uint64_t *sk = reinterpret_cast<uint64_t *>(pool);  // keys alone: (index << 16) | arrival position — 2048 of them in the 16 KB
    uint64_t *idx = d ? p.dout_idx : p.vout_idx;
    void *val = d ? p.dout_val : p.vout_val;
    const bool v32 = d ? p.q_is_32bit != 0 : p.t_is_32bit != 0;
    uint64_t n64 = d ? *p.n_dout : *p.n_vout;
    if (n64 > p.out_cap) n64 = p.out_cap;
    if (n64 > ROLE_SORT_MAX) {
        if (threadIdx.x == 0) atomicOr(&ap.state->miss_kind, 4u);
        return;
    }
    const uint32_t n = (uint32_t)n64;
    if (n == 0) return;
    uint32_t np2 = 2;
    while (np2 < n) np2 <<= 1;
    for (uint32_t i = threadIdx.x; i < np2; i += 256) {
        sk[i] = i < n ? (idx[i] << 16) | i : ~0ull;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= np2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < np2; i += 256) {
                const uint32_t x = i ^ j;
                if (x > i) {
                    const uint64_t a = sk[i], b = sk[x];
                    if ((a > b) == ((i & k) == 0)) {
                        sk[i] = b;
                        sk[x] = a;
                    }
                }
            }
            __syncthreads();
        }
    // straight into the payload (layout_pre laid the sections out in the scan's launch) — and back into the list itself: should
    // the encoder be repeated with another book, its assemble workgroups copy the list as they find it
    const uint64_t o_i = d ? ap.state->off.dout_idx : ap.state->off.vout_idx, o_v = d ? ap.state->off.dout_val : ap.state->off.vout_val;
    uint64_t *pi = reinterpret_cast<uint64_t *>(ap.payload + o_i);
    // the values follow their keys: from the list (still in arrival order) into the payload, then — all reads done — back
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint64_t k = sk[i];
        const uint32_t from = (uint32_t)(k & 0xFFFFu);
        pi[i] = k >> 16;
        if (v32) reinterpret_cast<uint32_t *>(ap.payload + o_v)[i] = reinterpret_cast<const uint32_t *>(val)[from];
        else reinterpret_cast<uint64_t *>(ap.payload + o_v)[i] = reinterpret_cast<const uint64_t *>(val)[from];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        idx[i] = sk[i] >> 16;
        if (v32) reinterpret_cast<uint32_t *>(val)[i] = reinterpret_cast<const uint32_t *>(ap.payload + o_v)[i];
        else reinterpret_cast<uint64_t *>(val)[i] = reinterpret_cast<const uint64_t *>(ap.payload + o_v)[i];
    }
}
template <uint32_t WIN>
__global__ __launch_bounds__(256) void k_pack(const uint16_t *__restrict__ codes, uint64_t n,
                                              const uint32_t *__restrict__ g_enc, const szk_cb_info *__restrict__ info,
                                              const uint16_t *__restrict__ chunk_words,
                                              const uint64_t *__restrict__ group_off, szk_mode mode, uint32_t sym_add,
                                              const szk_state *__restrict__ state, uint8_t *__restrict__ payload, szk_asm_params ap,
                                              uint32_t pack_blocks, szk_role_params rp) {
    constexpr int STAGE_WORDS = SZH_CHUNK_SYMS * SZH_MAX_LEN / 32 + 4;  // + slack for the unconditional 3-word emit
    __shared__ __align__(16) uint32_t s_enc[WIN];
    __shared__ uint32_t s_enc8[256];  // one-byte codes: code word by byte value ...
    __shared__ uint8_t s_plen8[256];  // ... and its length
    __shared__ __align__(8) uint32_t s_stage[4][STAGE_WORDS];
    // (a call that turns out to code with its sampled book — known on the device only behind the two-launch form of stage 1 — is packed by
    // k_pack_b, launched beside this kernel: its code words may be longer than the one-byte path here takes)
    if (ap.samp_words && ap.samp_words[SZK_SAMP_READY] && szk_is_narrow(mode)) return;
    const uint32_t roles = rp.on ? ROLE_BLOCKS : 0u;
    if (blockIdx.x < roles) {  // (the first workgroups dispatched; s_enc's 16 KB serve as their scratch)
        static_assert(WIN * 4 >= ROLE_SORT_MAX * 8 && WIN * 4 >= CB_SMALL_SYMS * 28 + 256, "role scratch fits the table");
        if (blockIdx.x == 0) {
            if (!rp.no_book) role_book(rp, ap.state, reinterpret_cast<uint8_t *>(s_enc));
        } else role_sort(rp, ap, blockIdx.x == 2, reinterpret_cast<uint8_t *>(s_enc));
        return;
    }
    const uint32_t bid = blockIdx.x - roles;  // the packer's own numbering
    const uint32_t nblk = gridDim.x - roles;
    if (bid >= pack_blocks) {  // the launch's last workgroups assemble the payload's other sections meanwhile
        assemble_body(ap, (uint64_t)(bid - pack_blocks) * 256 + threadIdx.x, (uint64_t)(nblk - pack_blocks) * 256);
        if (!ap.lists_by_roles) assemble_lists(ap, (uint64_t)bid * 256 + threadIdx.x, (uint64_t)nblk * 256);
        return;
    }
    const bool narrow = szk_is_narrow(mode);
    if (ap.assumed_narrow && !narrow) return;  // stage 1 assumed one-byte codes, the probe says two: nothing to pack (the call is repeated)
    const uint64_t n_full = n / SZH_CHUNK_SYMS, n_chunks = (n + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS;
    const uint64_t wave_gid = (uint64_t)bid * 4 + threadIdx.x / WAVE, nwaves = (uint64_t)pack_blocks * 4;
    const uint64_t lane_off = (uint64_t)lane_id() * ENC_PER_LANE;
    const int lane = lane_id();
    uint32_t *stage = s_stage[threadIdx.x / WAVE];
    uint32_t *out_base = reinterpret_cast<uint32_t *>(payload + state->off.bitstream);

    // side loads of a chunk: words of the chunks before it inside its group (one per lane) and the group offset
    auto side = [&](uint64_t ch, uint32_t &before_part, uint64_t &goff) {
        const uint64_t grp = ch / PACK_GROUP;
        const uint32_t cin = (uint32_t)(ch % PACK_GROUP);
        before_part = chunk_words[grp * PACK_GROUP + ((uint32_t)lane < cin ? lane : 0)];
        before_part = (uint32_t)lane < cin ? before_part : 0u;
        goff = group_off[grp];
    };
    CodeRegs cur, nxt;
    uint32_t bp_cur = 0, bp_nxt = 0;
    uint64_t go_cur = 0, go_nxt = 0;
    // (a workgroup taking a contiguous range of chunks instead of every nwaves-th one was measured, LAB_PACKMAP: no difference)
    const uint64_t c_end = n_full, c_step = nwaves;
    uint64_t chunk = wave_gid;
    if (chunk < c_end) {
        fetch_codes(codes, chunk * SZH_CHUNK_SYMS + lane_off, narrow, cur);
        side(chunk, bp_cur, go_cur);
    }
    uint32_t sym_min = info->win_lo;  // start of the LDS window
    const uint32_t sym_count = info->sym_count;
    // (a sampled book's escape symbol: its range is 256 symbols, but a listed delta still travels through the encoder's tables as symbol 0)
    const bool all_lds = WIN == ENC_WIN && sym_count <= ENC_WIN && info->esc_sym == 0;
    const bool wide = info->max_len > 16;  // two instead of four code words per 64-bit register
    if (WIN == ENC_WIN) {
        enc_table_load(s_enc, g_enc, sym_min, sym_count);
    } else {  // the doubled window: same centre, inside [0, 65536)
        const uint32_t centre = sym_min + ENC_WIN / 2;
        sym_min = centre > WIN / 2 ? centre - WIN / 2 : 0u;
        if (sym_min > SZH_HIST_BINS - WIN) sym_min = SZH_HIST_BINS - WIN;
        for (uint32_t i = threadIdx.x; i < WIN; i += 256) s_enc[i] = g_enc[sym_min + i];
    }
    if (narrow) {
        const uint32_t e = g_enc[threadIdx.x != 255u ? threadIdx.x + sym_add : 0u];
        s_enc8[threadIdx.x] = e >> 5;
        s_plen8[threadIdx.x] = (uint8_t)(e & 31u);
    }
    for (int i = lane; i < STAGE_WORDS; i += WAVE) stage[i] = 0;
    __syncthreads();
    for (; chunk < c_end; chunk += c_step) {
        const uint64_t nc = chunk + c_step;
        if (nc < c_end) {
            fetch_codes(codes, nc * SZH_CHUNK_SYMS + lane_off, narrow, nxt);
            side(nc, bp_nxt, go_nxt);
        }
        uint16_t c[ENC_PER_LANE];
        uint32_t nwords;
        if (narrow) {  // alphabets of one-byte codes have at most 256 symbols: code words <= 16 bits (a sampled book's may be longer: such a call is k_pack_b's)
            const uint32_t wds[4] = {cur.a.x, cur.a.y, cur.a.z, cur.a.w};
#pragma unroll
            for (int i = 0; i < 16; i++) c[i] = (uint16_t)((wds[i >> 2] >> (8 * (i & 3))) & 0xFFu);
            nwords = pack_chunk<4, true>(c, 0, 0, false, s_enc8, g_enc, sym_min, all_lds, stage, s_plen8);
        } else {
            unpack_codes(cur, narrow, sym_add, c);
            nwords = wide ? pack_chunk<2, false, WIN>(c, 0, 0, false, s_enc, g_enc, sym_min, all_lds, stage)
                          : pack_chunk<4, false, WIN>(c, 0, 0, false, s_enc, g_enc, sym_min, all_lds, stage);
        }
        const uint32_t before = wave_sum(bp_cur);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        uint32_t *out = out_base + go_cur + before;
        for (uint32_t i = 2 * lane; i < nwords + 2; i += 2 * WAVE) {  // copy out and re-zero the stage for the next chunk, two words per lane
            const uint2 wv = *reinterpret_cast<const uint2 *>(&stage[i]);
            *reinterpret_cast<uint2 *>(&stage[i]) = make_uint2(0u, 0u);
#if defined(LAB_PACK) && (LAB_PACK & 4)  // (lab, wrong results: no stores of the bit stream)
            if (wv.x == 0x12345678u && wv.y == 0x9abcdef0u) out[i] = 1;
#else
#ifdef LAB_PACK_NT  // (lab: the bit stream with streaming stores)
            if (i < nwords) __builtin_nontemporal_store(__builtin_bswap32(wv.x), &out[i]);
            if (i + 1 < nwords) __builtin_nontemporal_store(__builtin_bswap32(wv.y), &out[i + 1]);
#else
            if (i < nwords) out[i] = __builtin_bswap32(wv.x);  // bytes in stream order (see sz3hip_format.h)
            if (i + 1 < nwords) out[i + 1] = __builtin_bswap32(wv.y);
#endif
#endif
        }
        __builtin_amdgcn_wave_barrier();
        cur = nxt;
        bp_cur = bp_nxt;
        go_cur = go_nxt;
    }
    if (n_full < n_chunks && wave_gid == 0) {  // ragged tail
        const uint64_t base = n_full * SZH_CHUNK_SYMS + lane_off;
        uint16_t c[ENC_PER_LANE];
        load_codes16(codes, base, n, narrow, sym_add, c);
        uint32_t bp;
        uint64_t go;
        side(n_full, bp, go);
        const uint32_t nwords = wide ? pack_chunk<2, false, WIN>(c, base, n, true, s_enc, g_enc, sym_min, all_lds, stage)
                                     : pack_chunk<4, false, WIN>(c, base, n, true, s_enc, g_enc, sym_min, all_lds, stage);
        const uint32_t before = wave_sum(bp);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        uint32_t *out = out_base + go + before;
        for (uint32_t i = lane; i < nwords; i += WAVE) out[i] = __builtin_bswap32(stage[i]);
    }
    if (nblk > pack_blocks && !ap.lists_by_roles) assemble_lists(ap, (uint64_t)bid * 256 + threadIdx.x, (uint64_t)nblk * 256);  // (its share of the lists)
}

// ------------------------------------------------------------------------------------------------------------
// The packer of ONE-BYTE codes (round 6). k_pack spends ~14 vector instructions per symbol (two LDS lookups, the joins of code words
// into pairs, quads and octets, the emission): it is bound by their issue, not by the 1.5 bytes per symbol it moves. Here a lane
// looks up PAIRS of codes: a 128 x 128 table over the byte values t in [64, 192) (deltas -63 .. +64 around the mode of a one-byte
// stream, byte 127; 64 KB of LDS: entry = (code words of t0 and t1 joined) << 5 | their length, 0 when one of them has no code word
// or the two take more than 27 bits) — one lookup and one shift per TWO symbols. A workgroup is 1024 threads (16 waves share the
// table: 64 KB + 16 stages of 3 KB, one workgroup per CU); three code registers per lane are in flight (the chunk being packed and
// the next two: 16 waves x 3 KB per CU cover the HBM latency at the 3 TB/s the launch reads with).
//   fast tier (wave-uniform test): every byte of the chunk inside the window, every pair in the table, every lane's two octets <= 64 bits:
//     8 lookups, 2 emissions of 3 ds_or each;
//   otherwise the chunk goes symbol by symbol through the 256-entry tables like k_pack — pairs (<= 48 bits: code words up to 24 bits,
//     the books of round 6's sampled form give symbols the sample did not meet such lengths) joined to quads and octets where EVERY
//     lane's fit 64 bits (wave-uniform), emitted as octets, quads or pairs.
// Same bit stream as k_pack (same book, same chunk table): tests/test_gpu_stages.py compares the two byte for byte.
// Role workgroups (the two list sorts; the book role is k_pack's: this kernel is launched where no book is built beside the packer)
// come first in the grid, the assembly's last (see the kernel).
// ------------------------------------------------------------------------------------------------------------
#define PB_THREADS 1024u
#define PB_WAVES (PB_THREADS / WAVE)
#define PB_ASM_BLOCKS 32u
// The stage is an array of 64-bit words whose HIGH half is the earlier 32-bit word of the stream: a left-aligned 64-bit string at bit
// position pos lands in two of them — two 64-bit LDS atomics instead of three 32-bit ones.
__device__ __forceinline__ void packb_emit(uint64_t *stage, uint64_t val, uint32_t len, uint32_t pos) {  // val: len bits, right-aligned, 1 <= len <= 64
    const uint64_t v = val << ((64u - len) & 63u);  // left-aligned
    const uint32_t q = pos >> 6, sh = pos & 63u;
    atomicOr(reinterpret_cast<unsigned long long *>(&stage[q]), (unsigned long long)(v >> sh));
    atomicOr(reinterpret_cast<unsigned long long *>(&stage[q + 1]), (unsigned long long)((v << 1) << (63u - sh)));  // (= v << (64 - sh), and 0 for sh = 0)
}
// one chunk: the lane's 16 code bytes (w[0] = symbols 0 .. 3, low byte first) into the wave's stage; returns the chunk's word count
__device__ __forceinline__ uint32_t packb_chunk(const uint4 &cw, const uint32_t *__restrict__ s_pair, const uint32_t *__restrict__ s_enc8,
                                                const uint8_t *__restrict__ s_len8, uint64_t *stage) {
    const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
    // ---- fast tier ----
    {
        constexpr uint32_t K = 0x40404040u;  // t + 64 has bit 7 set exactly for t in [64, 192) (a carry out of a byte >= 192 can only clear the next byte's bit, never set a wrong one: such a byte's own bit is clear)
        const uint32_t inw = (w[0] + K) & (w[1] + K) & (w[2] + K) & (w[3] + K) & 0x80808080u;
        uint32_t e[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // entry of the pair (t0, t1): index (t1 & 127) << 7 | ((t0 ^ t1) & 127) — the XOR spreads the lookups over the LDS banks (a bank
            // is the index's low five bits: with t0 alone a smooth field's symbols, a dozen values around the mode, meet in a dozen banks)
            const uint32_t m = w[k] & 0x7F7F7F7Fu, x = m ^ (m >> 8), m1 = m >> 16, x1 = x >> 16;
            const uint32_t a0 = ((x & 0x7Fu) << 2) | ((m & 0x7F00u) << 1), a1 = ((x1 & 0x7Fu) << 2) | ((m1 & 0x7F00u) << 1);
            e[2 * k] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(s_pair) + a0);
            e[2 * k + 1] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(s_pair) + a1);
        }
        const uint32_t emin = min(min(min(e[0], e[1]), min(e[2], e[3])), min(min(e[4], e[5]), min(e[6], e[7])));
        uint64_t O[2];
        uint32_t OL[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint32_t l0 = e[4 * i] & 31u, l1 = e[4 * i + 1] & 31u, l2 = e[4 * i + 2] & 31u, l3 = e[4 * i + 3] & 31u;
            const uint64_t qa = ((uint64_t)(e[4 * i] >> 5) << l1) | (e[4 * i + 1] >> 5);      // <= 52 bits
            const uint64_t qb = ((uint64_t)(e[4 * i + 2] >> 5) << l3) | (e[4 * i + 3] >> 5);
            const uint32_t lb = l2 + l3;
            OL[i] = l0 + l1 + lb;
            O[i] = (qa << (lb & 63u)) | qb;
        }
        const bool fast = inw == 0x80808080u && emin != 0u && OL[0] <= 64u && OL[1] <= 64u;
        if (!__builtin_amdgcn_ballot_w64(!fast)) {
            const uint32_t bits = OL[0] + OL[1];
            const uint32_t incl = wave_incl_scan(bits);
            const uint32_t total_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
            const uint32_t pos = incl - bits;
            packb_emit(stage, O[0], OL[0], pos);
            packb_emit(stage, O[1], OL[1], pos + OL[0]);
            return (total_bits + 31u) >> 5;
        }
    }
    // ---- symbol by symbol ----
    uint64_t P[8];
    uint32_t PL[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t pr = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        const uint32_t b0 = pr & 0xFFu, b1 = pr >> 8;
        const uint32_t l1 = s_len8[b1];
        P[k] = ((uint64_t)s_enc8[b0] << l1) | s_enc8[b1];  // <= 48 bits
        PL[k] = (uint32_t)s_len8[b0] + l1;
    }
    uint64_t Q[4];
    uint32_t QL[4];
    bool fit_q = true;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        QL[j] = PL[2 * j] + PL[2 * j + 1];
        Q[j] = (P[2 * j] << (PL[2 * j + 1] & 63u)) | P[2 * j + 1];
        fit_q &= QL[j] <= 64u;
    }
    const bool fit_o = QL[0] + QL[1] <= 64u && QL[2] + QL[3] <= 64u;
    const uint32_t bits = QL[0] + QL[1] + QL[2] + QL[3];
    const uint32_t incl = wave_incl_scan(bits);
    const uint32_t total_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
    uint32_t pos = incl - bits;
    if (!__builtin_amdgcn_ballot_w64(!fit_o)) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint32_t len = QL[2 * i] + QL[2 * i + 1];
            if (len) packb_emit(stage, (Q[2 * i] << (QL[2 * i + 1] & 63u)) | Q[2 * i + 1], len, pos);
            pos += len;
        }
    } else if (!__builtin_amdgcn_ballot_w64(!fit_q)) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (QL[j]) packb_emit(stage, Q[j], QL[j], pos);
            pos += QL[j];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (PL[k]) packb_emit(stage, P[k], PL[k], pos);
            pos += PL[k];
        }
    }
    return (total_bits + 31u) >> 5;
}
__global__ __launch_bounds__(PB_THREADS) void k_pack_b(const uint16_t *__restrict__ codes, uint64_t n, const uint32_t *__restrict__ g_enc,
                                                        const uint16_t *__restrict__ chunk_words, const uint64_t *__restrict__ group_off, szk_mode mode,
                                                        uint32_t sym_add, const szk_state *__restrict__ state, uint8_t *__restrict__ payload,
                                                        szk_asm_params ap, uint32_t pack_blocks, uint32_t split, szk_role_params rp, uint32_t only_sampled) {
    constexpr int STAGE_WORDS = SZH_CHUNK_SYMS * SZH_MAX_LEN / 32 + 4;  // + slack for the unconditional 3-word emit
    __shared__ __align__(16) uint32_t s_pair[128 * 128];
    __shared__ uint32_t s_enc8[256];  // code word by byte value ...
    __shared__ uint8_t s_plen8[256];  // ... and its length
    __shared__ __align__(8) uint64_t s_stage[PB_WAVES][STAGE_WORDS / 2];
    // (launched beside k_pack behind the two-launch form of stage 1 — only_sampled: this kernel then packs, sorts and assembles only
    // when the call codes with its sampled book, k_pack otherwise)
    if ((only_sampled & 1u) && !(ap.samp_words && ap.samp_words[SZK_SAMP_READY] && szk_is_narrow(mode))) return;
    const uint32_t roles = rp.on ? ROLE_BLOCKS : 0u;
    if (blockIdx.x < roles) {  // (the pair table's memory serves as their scratch; they are 256-thread bodies: the other waves leave)
        if (threadIdx.x >= 256u) return;
        if (blockIdx.x != 0) role_sort(rp, ap, blockIdx.x == 2, reinterpret_cast<uint8_t *>(s_pair));
        return;
    }
    // Order in the grid: the role workgroups, the packers, the assembly's. A workgroup holds a compute unit's LDS: roles + packers = the
    // number of compute units, all of them start at once, and the assembly's short workgroups follow on whichever unit is free first (a
    // sort role's, after ~20 us). With the packers one per unit AND the roles in front, a packer starts when a role ends — and its
    // static share of the chunks ends that much later than everybody else's (measured: 93 instead of 62 us).
    const uint32_t asm_blocks = gridDim.x - roles - pack_blocks;
    if (blockIdx.x >= roles + pack_blocks) {
        const uint32_t ab = blockIdx.x - roles - pack_blocks;
        assemble_body(ap, (uint64_t)ab * PB_THREADS + threadIdx.x, (uint64_t)asm_blocks * PB_THREADS);
        if (!ap.lists_by_roles) assemble_lists(ap, (uint64_t)ab * PB_THREADS + threadIdx.x, (uint64_t)asm_blocks * PB_THREADS);
        return;
    }
    const uint32_t bid = blockIdx.x - roles;
    if (!szk_is_narrow(mode)) return;  // stage 1 assumed one-byte codes, the probe says two: nothing to pack (the assembly reports it, the call is repeated)
    const uint64_t n_full = n / SZH_CHUNK_SYMS, n_chunks = (n + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS;
    (void)split;
    const uint32_t wv = threadIdx.x / WAVE;
    const uint64_t wave_gid = (uint64_t)bid * PB_WAVES + wv, nwaves = (uint64_t)pack_blocks * PB_WAVES;
    const int lane = lane_id();
    const uint8_t *c8 = reinterpret_cast<const uint8_t *>(codes) + (uint64_t)lane * ENC_PER_LANE;
    uint64_t *stage = s_stage[wv];
    uint32_t *out_base = reinterpret_cast<uint32_t *>(payload + state->off.bitstream);
    // A wave's work unit: PB_BATCH consecutive chunks (a unit never straddles an offset group: PACK_GROUP is a multiple of it) — their bit
    // strings are one contiguous run of the stream, the run's place = the group's offset + the words of the group's chunks in front of it
    // (one load of the group's counts). Units are dealt round-robin over the launch's waves, so that at any time the chip reads one
    // window of consecutive units (16 MB of codes) and writes one window of the stream: a wave that owned a whole group — 4096 separate
    // streams of 32 KB over the chip — was a third slower whenever the codes came from HBM rather than from the memory-side cache.
    // The loads of a unit are issued together, a unit ahead — on gfx9 a wave's loads and stores share one counter (vmcnt) and complete
    // out of order with respect to each other, so waiting for a load means waiting for every store issued before: once per unit here,
    // once per chunk in k_pack's loop (whose wave then idles through the store acknowledgement of its previous chunk).
    constexpr uint32_t PB_BATCH = 4;
    const uint64_t n_units = (n_full + PB_BATCH - 1) / PB_BATCH;
#ifdef SZ3HIP_LAB  // (lab build: what the launch's time is made of — only_sampled bits 8: no stores of the stream, 16: every unit reads the first unit's codes)
    const bool lab_nost = (only_sampled & 8u) != 0, lab_nold = (only_sampled & 16u) != 0;
#else
    constexpr bool lab_nost = false, lab_nold = false;
#endif
    auto fetch = [&](uint64_t ch) { return ch < n_full ? *reinterpret_cast<const uint4 *>(c8 + (lab_nold ? ch % (4 * nwaves) : ch) * SZH_CHUNK_SYMS) : make_uint4(0u, 0u, 0u, 0u); };
    auto front_of = [&](uint64_t unit) -> uint32_t {  // (one lane-parallel load: the counts of the group's chunks in front of the unit)
        const uint64_t c_lo = unit * PB_BATCH, grp = c_lo / PACK_GROUP;
        const uint32_t first = (uint32_t)(c_lo % PACK_GROUP);
        return ((uint32_t)lane < first && grp * PACK_GROUP + lane < n_chunks) ? chunk_words[grp * PACK_GROUP + lane] : 0u;
    };
    uint64_t unit = wave_gid;
    uint4 cur[PB_BATCH], nxt[PB_BATCH];
    uint32_t fr_cur = 0, fr_nxt = 0;
    uint64_t go_cur = 0, go_nxt = 0;
#pragma unroll
    for (uint32_t j = 0; j < PB_BATCH; j++) cur[j] = make_uint4(0u, 0u, 0u, 0u);
    if (unit < n_units) {
#pragma unroll
        for (uint32_t j = 0; j < PB_BATCH; j++) cur[j] = fetch(unit * PB_BATCH + j);
        fr_cur = front_of(unit);
        go_cur = group_off[unit * PB_BATCH / PACK_GROUP];
    }
    // the tables: single symbols by byte value, then the pairs of the window from them
    if (threadIdx.x < 256u) {
        const uint32_t e = g_enc[threadIdx.x != 255u ? threadIdx.x + sym_add : 0u];
        s_enc8[threadIdx.x] = e >> 5;
        s_plen8[threadIdx.x] = (uint8_t)(e & 31u);
    }
    for (int i = lane; i < STAGE_WORDS / 2; i += WAVE) stage[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 128u * 128u; i += PB_THREADS) {
        const uint32_t i1 = i >> 7, i0 = (i & 127u) ^ i1;        // window indices t & 127 of the first and the second symbol (the table is swizzled: packb_chunk)
        const uint32_t t0 = i0 < 64u ? i0 + 128u : i0, t1 = i1 < 64u ? i1 + 128u : i1;  // byte values in [64, 192)
        const uint32_t l0 = s_plen8[t0], l1 = s_plen8[t1];
        const uint32_t len = l0 + l1;
        s_pair[i] = (l0 && l1 && len <= 27u) ? ((((s_enc8[t0] << l1) | s_enc8[t1]) << 5) | len) : 0u;
    }
    __syncthreads();
    for (; unit < n_units; unit += nwaves) {
        const uint64_t nu = unit + nwaves;
        if (nu < n_units) {
#pragma unroll
            for (uint32_t j = 0; j < PB_BATCH; j++) nxt[j] = fetch(nu * PB_BATCH + j);
            fr_nxt = front_of(nu);
            go_nxt = group_off[nu * PB_BATCH / PACK_GROUP];
        }
        uint32_t *out = out_base + go_cur + wave_sum(fr_cur);
        const uint64_t c_lo = unit * PB_BATCH;
#pragma unroll
        for (uint32_t j = 0; j < PB_BATCH; j++) {
            if (c_lo + j < n_full) {  // (wave-uniform: whole chunks only — the ragged last one is packed apart)
                const uint32_t nwords = packb_chunk(cur[j], s_pair, s_enc8, s_plen8, stage);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = 2 * lane; i < nwords + 2; i += 2 * WAVE) {  // copy out and re-zero the stage for the next chunk, two words per lane
                    const uint64_t v = stage[i >> 1];
                    stage[i >> 1] = 0;
                    if (lab_nost && v != 0x123456789abcdefull) continue;
                    if (i < nwords) out[i] = __builtin_bswap32((uint32_t)(v >> 32));  // bytes in stream order (see sz3hip_format.h)
                    if (i + 1 < nwords) out[i + 1] = __builtin_bswap32((uint32_t)v);
                }
                __builtin_amdgcn_wave_barrier();
                out += nwords;
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < PB_BATCH; j++) cur[j] = nxt[j];
        fr_cur = fr_nxt;
        go_cur = go_nxt;
    }
    if (n_full < n_chunks && wave_gid == 0) {  // ragged tail: the missing symbols have no bits — packed symbol by symbol
        const uint64_t base = n_full * SZH_CHUNK_SYMS + (uint64_t)lane * ENC_PER_LANE;
        const uint8_t *cb = reinterpret_cast<const uint8_t *>(codes);
        uint32_t bits = 0;
        uint64_t P[8];
        uint32_t PL[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint64_t i0 = base + 2 * k, i1 = i0 + 1;
            const uint32_t b0 = i0 < n ? cb[i0] : 0u, b1 = i1 < n ? cb[i1] : 0u;
            const uint32_t l0 = i0 < n ? s_plen8[b0] : 0u, l1 = i1 < n ? s_plen8[b1] : 0u;
            P[k] = ((uint64_t)(i0 < n ? s_enc8[b0] : 0u) << l1) | (i1 < n ? s_enc8[b1] : 0u);
            PL[k] = l0 + l1;
            bits += PL[k];
        }
        const uint32_t incl = wave_incl_scan(bits);
        const uint32_t total_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
        uint32_t pos = incl - bits;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (PL[k]) packb_emit(stage, P[k], PL[k], pos);
            pos += PL[k];
        }
        const uint32_t nwords = (total_bits + 31u) >> 5;
        const uint64_t grp = n_full / PACK_GROUP;
        const uint32_t cin = (uint32_t)(n_full % PACK_GROUP);
        uint32_t bp = (uint32_t)lane < cin ? chunk_words[grp * PACK_GROUP + lane] : 0u;
        const uint32_t before = wave_sum(bp);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        uint32_t *out = out_base + group_off[grp] + before;
        for (uint32_t i = lane; i < nwords; i += WAVE) out[i] = __builtin_bswap32((uint32_t)(stage[i >> 1] >> ((i & 1u) ? 0 : 32)));
    }
}

// ------------------------------------------------------------------------------------------------------------
// The encoder behind a FUSED stage 1 (k_lorenzo_quant_march3f): the rows' bit strings exist, in the tasks' slots of the scratch; a
// chunk of the payload is four consecutive 256-element segments (x extents that are multiples of 256), its place known since the
// offset scan. A wave per chunk, a lane per output word: the word's first bit lies in segment k = the number of segment starts at
// or before it, its bits come from two consecutive words of that segment's string (a funnel shift) and — where the segment ends
// inside the word — from the first word of the next one (segments are at least 256 bits long: a word meets at most two).
// 3 instructions per symbol instead of the packer's 13, and no code array: the launch reads 0.5 B/elem and writes 0.5.
// The role workgroups (this call's code book + verdict, list sorts) and the assembly ride along as in k_pack.
// ------------------------------------------------------------------------------------------------------------
#ifdef SZ3HIP_LAB  // (the encoder behind the fused stage 1: lab build only)
struct szk_merge_params {
    const uint32_t *slots;       // the scratch stage 1 wrote
    const uint16_t *seg_bits;    // [n / 256] bits of every segment's string
    const uint32_t *seg_base;    // [n / 256] index of its first word in the scratch
    const uint32_t *fuse_flag;   // raised by stage 1: a book it could not code with
};
#define MERGE_BLOCK 16  // chunks a wave works at a time: their 64 segments' lengths and places are one load per lane
__global__ __launch_bounds__(256) void k_merge(szk_merge_params mp, uint64_t n, const uint16_t *__restrict__ chunk_words,
                                               const uint64_t *__restrict__ group_off, szk_mode mode, const szk_state *__restrict__ state,
                                               uint8_t *__restrict__ payload, szk_asm_params ap, uint32_t pack_blocks, szk_role_params rp) {
    __shared__ __align__(16) uint32_t s_pool[ENC_WIN];  // the role workgroups' scratch (see k_pack)
    const uint32_t roles = rp.on ? ROLE_BLOCKS : 0u;
    if (blockIdx.x < roles) {
        if (blockIdx.x == 0) {
#if defined(LAB_MERGE) && (LAB_MERGE & 1)  // (lab: no book role)
            if (threadIdx.x == 0) { ap.state->mispredict = 0; ap.state->n_symbols = rp.cb.range[2]; }
#else
            if (!rp.no_book) role_book(rp, ap.state, reinterpret_cast<uint8_t *>(s_pool));
#endif
            if (threadIdx.x == 0 && *mp.fuse_flag) atomicOr(&ap.state->miss_kind, 64u);
        } else role_sort(rp, ap, blockIdx.x == 2, reinterpret_cast<uint8_t *>(s_pool));
        return;
    }
    const uint32_t bid = blockIdx.x - roles;
    const uint32_t nblk = gridDim.x - roles;
    if (bid >= pack_blocks) {
        assemble_body(ap, (uint64_t)(bid - pack_blocks) * 256 + threadIdx.x, (uint64_t)(nblk - pack_blocks) * 256);
        if (!ap.lists_by_roles) assemble_lists(ap, (uint64_t)bid * 256 + threadIdx.x, (uint64_t)nblk * 256);
        return;
    }
    if (ap.assumed_narrow && !szk_is_narrow(mode)) return;  // the probe says two-byte codes: the call is repeated
    const uint64_t n_chunks = (n + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS, n_segs = n >> 8;  // (n is a multiple of 256; the last chunk may hold fewer than four segments)
    const uint64_t n_blocks = (n_chunks + MERGE_BLOCK - 1) / MERGE_BLOCK;
    const uint64_t wave_gid = (uint64_t)bid * 4 + threadIdx.x / WAVE, nwaves = (uint64_t)pack_blocks * 4;
    const int lane = lane_id();
    uint32_t *out_base = reinterpret_cast<uint32_t *>(payload + state->off.bitstream);
#if defined(LAB_MERGE) && (LAB_MERGE & 2)  // (lab: the roles and the assembly alone)
    if (n_blocks) return;
#endif
    for (uint64_t blk = wave_gid; blk < n_blocks; blk += nwaves) {
        const uint64_t chunk0 = blk * MERGE_BLOCK;  // (a block is half a group of the offset scan: PACK_GROUP = 2 * MERGE_BLOCK)
        // lane l: segment 4 * chunk0 + l (= segment l & 3 of chunk chunk0 + l / 4); lane l < 32: words of chunk l of the group
        const uint64_t seg = chunk0 * 4 + (uint64_t)lane;
        const bool seg_ok = seg < n_segs;
        const uint32_t sb = seg_ok ? mp.seg_bits[seg] : 0u;
        const uint32_t sa = seg_ok ? mp.seg_base[seg] : 0u;
        const uint64_t grp = chunk0 / PACK_GROUP;
        const uint64_t gc = grp * PACK_GROUP + (uint64_t)(lane & 31);
        const uint32_t cw = (lane < 32 && gc < n_chunks) ? chunk_words[gc] : 0u;
        const uint64_t goff = group_off[grp];
        const uint32_t cw_excl = wave_incl_scan(cw) - cw;  // (lanes 0 .. 31: words of the group's chunks before chunk l)
        const uint32_t c_in0 = (uint32_t)(chunk0 % PACK_GROUP);
#pragma unroll 2
        for (uint32_t ci = 0; ci < MERGE_BLOCK; ci++) {
            if (chunk0 + ci >= n_chunks) break;
            const uint32_t nwords = (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)(c_in0 + ci));
            const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)cw_excl, (int)(c_in0 + ci));
            uint32_t b[4], base[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                b[k] = (uint32_t)__builtin_amdgcn_readlane((int)sb, (int)(4 * ci + k));
                base[k] = (uint32_t)__builtin_amdgcn_readlane((int)sa, (int)(4 * ci + k));
            }
            const uint32_t B1 = b[0], B2 = B1 + b[1], B3 = B2 + b[2];
            uint32_t *out = out_base + goff + before;
            // three words per lane in flight (a chunk of a smooth field is ~130 words; up to 512: the loop goes on)
            for (uint32_t j0 = (uint32_t)lane; j0 < nwords; j0 += 3 * WAVE) {
                uint32_t w0[3], w1[3], wn[3], rem[3], shv[3];
                bool nxt[3];
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const uint32_t j = j0 + u * WAVE;
                    const uint32_t bit = (j < nwords ? j : 0u) << 5;
                    const uint32_t k = (bit >= B1) + (bit >= B2) + (bit >= B3);
                    const uint32_t Bk = k == 0 ? 0u : (k == 1 ? B1 : (k == 2 ? B2 : B3));
                    const uint32_t bk = k == 0 ? b[0] : (k == 1 ? b[1] : (k == 2 ? b[2] : b[3]));
                    const uint32_t sk = k == 0 ? base[0] : (k == 1 ? base[1] : (k == 2 ? base[2] : base[3]));
                    const uint32_t sn = k == 0 ? base[1] : (k == 1 ? base[2] : base[3]);  // the next segment's string (k = 3: none)
                    const uint32_t bn = k == 0 ? b[1] : (k == 1 ? b[2] : (k == 2 ? b[3] : 0u));  // (0: the array ends with segment k)
                    const uint32_t o = bit - Bk;
                    shv[u] = o & 31u;
                    rem[u] = bk - o;  // bits of segment k from here on (>= 1)
                    nxt[u] = rem[u] < 32u && bn != 0u;
                    const uint32_t *src = mp.slots + sk + (o >> 5);
                    w0[u] = src[0];
                    w1[u] = src[1];  // (the scratch has slack behind its last slot)
                    wn[u] = mp.slots[nxt[u] ? sn : sk];
                }
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const uint32_t j = j0 + u * WAVE;
                    uint32_t v = (uint32_t)((((uint64_t)w0[u] << 32) | w1[u]) >> (32u - shv[u]));  // funnel shift left by shv
                    if (rem[u] < 32u) {
                        v &= ~0u << (32u - rem[u]);
                        if (nxt[u]) v |= wn[u] >> rem[u];
                    }
                    if (j < nwords) out[j] = __builtin_bswap32(v);  // bytes in stream order (see sz3hip_format.h)
                }
            }
        }
    }
    if (nblk > pack_blocks && !ap.lists_by_roles) assemble_lists(ap, (uint64_t)bid * 256 + threadIdx.x, (uint64_t)nblk * 256);
}

#endif  // SZ3HIP_LAB
// ------------------------------------------------------------------------------------------------------------
// K8: decode side
// ------------------------------------------------------------------------------------------------------------
// canonical decode tables from the code lengths: first code / first rank per length, the symbols sorted by
// (len, sym), and a direct lookup table over the next DEC_LUT_BITS bits of the stream: (symbol << 8) | length for every
// code word of at most DEC_LUT_BITS bits (0 = longer code: length search). One workgroup; <= 65536 symbols.
__global__ __launch_bounds__(1024) void k_dec_tables(const uint8_t *__restrict__ lens, uint32_t sym_min,
                                                     uint32_t sym_count, szk_dec_tables *t, uint32_t radius, uint32_t esc_sym, uint32_t *zero_word,
                                                     const uint16_t *__restrict__ chunk_words, uint64_t n_chunks, uint64_t *group_off,
                                                     uint64_t *total_words) {
    if (blockIdx.x == 1) {  // the decoder's other preparation, beside the tables: word offsets of the chunk groups
        scan_groups_body<false>(const_cast<uint16_t *>(chunk_words), n_chunks, group_off, total_words, nullptr, 0, nullptr);
        return;
    }
    if (zero_word && threadIdx.x == 0) *zero_word = 0;  // (the decoder's overflow flag of a half-width chain)
    __shared__ uint32_t s_cnt[SZH_MAX_LEN + 2], s_first_code[SZH_MAX_LEN + 2], s_first_rank[SZH_MAX_LEN + 2];
    __shared__ uint16_t s_tbl[(SZH_MAX_LEN + 1) * 1024];  // per-(length, thread) counts -> exclusive ranks
    const uint32_t tid = threadIdx.x, lane = lane_id();
    if (tid < SZH_MAX_LEN + 2) s_cnt[tid] = 0;
    // every thread owns a contiguous run of symbols (keeps (len, sym) order without a sort)
    const uint32_t per = (sym_count + 1023) / 1024;
    const uint32_t q0 = tid * per < sym_count ? tid * per : sym_count, q1 = q0 + per < sym_count ? q0 + per : sym_count;
    for (uint32_t l = 0; l <= SZH_MAX_LEN; l++) s_tbl[l * 1024 + tid] = 0;
    __syncthreads();
    for (uint32_t i = q0; i < q1; i++) {
        const uint32_t l = lens[i];
        if (l && l <= SZH_MAX_LEN) s_tbl[l * 1024 + tid]++;
    }
    __syncthreads();
    for (uint32_t l = 1 + tid / WAVE; l <= SZH_MAX_LEN; l += 1024 / WAVE) {  // one wave per length: scan along the threads
        uint32_t carry = 0;
        for (uint32_t c0 = 0; c0 < 1024; c0 += WAVE) {
            const uint32_t v = s_tbl[l * 1024 + c0 + lane];
            const uint32_t incl = wave_incl_scan(v);
            s_tbl[l * 1024 + c0 + lane] = (uint16_t)(carry + incl - v);
            carry += __shfl(incl, WAVE - 1, WAVE);
        }
        if (lane == 0) s_cnt[l] = carry;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t code = 0, rank = 0, maxl = 0;
        for (uint32_t l = 1; l <= SZH_MAX_LEN; l++) {
            code = (code + (l > 1 ? s_cnt[l - 1] : 0)) << (l > 1 ? 1 : 0);
            t->first_code[l] = code;
            t->first_rank[l] = rank;
            t->count[l] = s_cnt[l];
            s_first_code[l] = code;
            s_first_rank[l] = rank;
            rank += s_cnt[l];
            if (s_cnt[l]) maxl = l;
        }
        t->max_len = maxl;
        t->n_coded = rank;
        t->lut_bits = maxl < DEC_LUT_BITS ? maxl : DEC_LUT_BITS;
    }
    __syncthreads();
    for (uint32_t i = q0; i < q1; i++) {
        const uint32_t l = lens[i];
        // (the canonical order is the stored symbols'; a stream's escape symbol — the stand-in for a listed delta — decodes as symbol 0)
        if (l && l <= SZH_MAX_LEN) t->sorted_syms[s_first_rank[l] + s_tbl[l * 1024 + tid]++] = (uint16_t)(esc_sym && sym_min + i == esc_sym ? 0u : sym_min + i);
    }
    __threadfence();
    __syncthreads();
    uint32_t maxl = 0;
    for (uint32_t l = 1; l <= SZH_MAX_LEN; l++)
        if (s_cnt[l]) maxl = l;
    const uint32_t K = maxl < DEC_LUT_BITS ? maxl : DEC_LUT_BITS;
    for (uint32_t e = tid; e < (1u << K); e += 1024) {
        uint32_t ent = 0;
        for (uint32_t l = 1; l <= K; l++) {
            const uint32_t rel = (e >> (K - l)) - s_first_code[l];
            if (rel < s_cnt[l]) {
                ent = ((uint32_t)t->sorted_syms[s_first_rank[l] + rel] << 8) | l;
                break;
            }
        }
        t->lut[e] = ent;
    }
    if (radius && maxl >= 1 && maxl <= 16) {
        // the multi-symbol table (szk_dec_tables::mlut): every 12-bit window decoded greedily, up to three code words that fit it
        for (uint32_t e = tid; e < (1u << DEC_LUT_BITS); e += 1024) {
            uint32_t used = 0, cnt = 0;
            int d[3] = {0, 0, 0};
            for (int j = 0; j < 3; j++) {
                const uint32_t rem = DEC_LUT_BITS - used;
                const uint32_t w = (e << used) & ((1u << DEC_LUT_BITS) - 1u);  // the bits left, at the window's top
                uint32_t l = 0, sym = 0;
                for (uint32_t q = 1; q <= rem && q <= maxl; q++) {
                    const uint32_t rel = (w >> (DEC_LUT_BITS - q)) - s_first_code[q];
                    if (rel < s_cnt[q]) {
                        l = q;
                        sym = t->sorted_syms[s_first_rank[q] + rel];
                        break;
                    }
                }
                const int delta = (int)sym - (int)radius;
                if (l == 0 || sym == 0 || delta < -127 || delta > 127) break;
                d[cnt++] = delta;
                used += l;
            }
            uint32_t ent = 0;
            if (cnt) {
                const int s2 = cnt >= 2 ? d[0] + d[1] : d[0], d3 = cnt == 3 ? d[2] : 0;
                ent = (used - 1u) | (cnt << 4) | (((uint32_t)d[0] & 0xFFu) << 6) | (((uint32_t)s2 & 0x1FFu) << 14) | (((uint32_t)d3 & 0xFFu) << 23);
            }
            t->mlut[e] = ent;
        }
    }
}

// one thread per UNIT of 256 symbols (a quarter chunk: the chunk's start or one of its three restart offsets): table lookup on
// the next lut_bits bits of a 64-bit MSB-aligned window. Code words longer than the table (interpolation streams at tight bounds
// keep ~3 % of their symbols there: every step of a wave meets one) are decoded without a loop or a global access: the length
// is K + 1 + the number of lengths l whose left-aligned upper code bound is <= the window (canonical codes grow with the
// length), and the symbols of the long codes sit in LDS.
// Why units: a lane's symbols are one dependent chain (window -> table -> length -> shift), ~700 cycles per symbol with the two
// waves per SIMD that 131 072 chunks of a 512^3 array give. Four lanes per chunk are four times the chains in flight
// (round 3; the stream-word ring of round 2 — lanes fetching each other's cache lines through LDS — gained nothing and is gone).
#ifndef DEC_SORTED_LDS
#define DEC_SORTED_LDS 1024u  // symbols of the codes longer than the table kept in LDS (further ranks: global)
#endif
#ifndef DEC_STAGE_BYTES
// bytes of a lane's output collected in LDS before they are stored. 128 (whole lines per lane) leaves room for two workgroups per
// CU, 64 for four: measured 294 -> 218 us (C2, fused) and 419 -> 294 us (C3, plain codes); 32 (five workgroups): 240 / 330
#define DEC_STAGE_BYTES 64u
#endif
// delta of a delta outlier (code 0) by binary search in the sorted index list; 0 when the element is not listed (a corrupt
// stream must not crash the decoder)
template <typename QO>
__device__ __forceinline__ QO dec_dout(const szk_dec_params &p, uint64_t elem) {
    uint64_t lo = 0, hi = p.n_dout;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (p.dout_idx[mid] < elem) lo = mid + 1;
        else hi = mid;
    }
    return lo < p.n_dout && p.dout_idx[lo] == elem ? reinterpret_cast<const QO *>(p.dout_val)[lo] : (QO)0;
}
// QB: 0 = u16 codes out; 4 / 8 = fused Lorenzo x-scan, int32 / int64 lattice values out (see szk_dec_params::scan_row)
// MS (round 5; QB = 4, HALF): the multi-symbol table form for small code books — one lookup on the next 12 bits yields up to three
// code words and, for the fused x prefix sum, their partial sums (szk_dec_tables::mlut): 12 instead of 27 vector instructions and a
// third of the dependent LDS round trips per symbol at C2's 4.1 bits per symbol. A lane's values go to its stage as 16-bit stores at
// the symbol's place (a lookup's three sums at k, k + 1, k + 2: with fewer than three code words the next lookup overwrites the
// repeats), a block of 32 symbols leaves through the same cooperative path as before. Code words beyond 12 bits, listed deltas
// (symbol 0), the last two symbols of a row or a unit: one at a time through the canonical-code arithmetic (ms_single).
template <int QB, bool HALF = false, bool MS = false>
__global__ __launch_bounds__(256) void k_decode(const uint8_t *__restrict__ payload, szk_dec_params p,
                                                uint16_t *__restrict__ codes) {
    static_assert(!MS || (QB == 4 && HALF), "the multi-symbol form decodes f32 Lorenzo streams into the half-width chain");
    using QO = typename std::conditional<QB == 8, int64_t, int32_t>::type;
    if (p.gate && *p.gate == 0) return;  // (the full-width chain behind a half-width one that did not overflow)
    constexpr uint32_t UNIT = SZH_UNIT_SYMS;
    constexpr uint32_t SORTED_LDS = DEC_SORTED_LDS;
    // Output through LDS: a lane's 16 values of a round are one 32- / 64- / 128-byte run of ITS unit, 0.5 - 2 KB from its
    // neighbour's — stored directly, every store instruction touches 64 cache lines with 16 bytes each (0.23 of the kernel's
    // 0.5 ms at C2, round 2: measured by switching the stores off). Staged in LDS and read back piece-major, NP * NR consecutive
    // lanes write one unit's run: whole lines per instruction.
    constexpr uint32_t ELT = QB ? (HALF ? QB / 2 : QB) : 2;       // bytes per output element
    constexpr uint32_t NP = 16 * ELT / 16;                         // 16-byte pieces per lane and round
    constexpr uint32_t NR = NP * 16 >= DEC_STAGE_BYTES ? 1 : DEC_STAGE_BYTES / (NP * 16);  // rounds collected before a store
    static_assert((UNIT / 16) % NR == 0, "a unit's rounds are a multiple of the collected rounds");
    __shared__ uint4 s_out[4][64 * (NP * NR + 1)];
    __shared__ uint32_t s_first_code[SZH_MAX_LEN + 2], s_first_rank[SZH_MAX_LEN + 2], s_upper[SZH_MAX_LEN + 2];
    __shared__ uint32_t s_lut[1u << DEC_LUT_BITS];
    __shared__ uint16_t s_sorted[SORTED_LDS];
#ifdef LAB_DEC_PAD  // (lab: fewer workgroups per CU — fewer stream lines live in the L2)
    __shared__ uint32_t s_pad[LAB_DEC_PAD / 4];
    if (p.n == 1) s_pad[threadIdx.x] = 1;
#endif
    const uint32_t max_len = p.tables->max_len, K = p.tables->lut_bits, n_coded = p.tables->n_coded;
    if (threadIdx.x <= SZH_MAX_LEN + 1) {
        const uint32_t l = threadIdx.x;
        const bool in = l >= 1 && l <= SZH_MAX_LEN;
        const uint32_t fc = in ? p.tables->first_code[l] : 0, cnt = in ? p.tables->count[l] : 0;
        s_first_code[l] = fc;
        s_first_rank[l] = in ? p.tables->first_rank[l] : 0;
        // exclusive upper bound of the length-l code words, left-aligned to 32 bits (lengths >= max_len never count)
        s_upper[l] = in && l < max_len ? (fc + cnt) << (32 - l) : 0xFFFFFFFFu;
    }
    if (MS) {
        for (uint32_t e = threadIdx.x; e < (1u << DEC_LUT_BITS); e += 256) s_lut[e] = p.tables->mlut[e];
    } else {
        for (uint32_t e = threadIdx.x; e < (1u << K); e += 256) s_lut[e] = p.tables->lut[e];
    }
    const uint16_t *sorted = p.tables->sorted_syms;
    const uint32_t base_rank = MS ? 0u : (K < max_len ? p.tables->first_rank[K + 1] : n_coded);  // (MS: the whole book, at most SORTED_LDS symbols)
    for (uint32_t e = threadIdx.x; e < SORTED_LDS && base_rank + e < n_coded; e += 256) s_sorted[e] = sorted[base_rank + e];
    __syncthreads();
    const uint64_t unit = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_units = (p.n + UNIT - 1) / UNIT;
    static_assert(64 / SZH_SUBS == PACK_GROUP, "a wave's units are one offset group");
    uint32_t woff_in, nwords;
    {
        const uint64_t ch = unit / SZH_SUBS;
        nwords = ch < p.n_chunks ? (uint32_t)p.chunk_words[ch] : 0u;
        const uint32_t mine = (unit % SZH_SUBS) == 0 ? nwords : 0u;  // (a chunk counts once: at its first unit)
        woff_in = wave_incl_scan(mine) - nwords;  // (inclusive: the sum reaches the lane's own chunk at either of its units)
    }
    if (unit >= n_units) return;
    const uint64_t chunk = unit / SZH_SUBS;
    const uint32_t sub = (uint32_t)(unit % SZH_SUBS);
    const uint64_t s0 = unit * UNIT;
    const uint32_t nsym = (uint32_t)((p.n - s0 < UNIT) ? (p.n - s0) : UNIT);
    uint16_t *out = codes + s0;
    QO *qout = QB ? reinterpret_cast<QO *>(p.q_out) + s0 : nullptr;
    // (wave-uniform: all 64 lanes decode full units — everywhere but in the array's last wave)
    const bool coop = !(p.reserved & 2u) && __ballot(nsym == UNIT) == ~0ull;
    uint4 *stage = s_out[threadIdx.x / WAVE];
    uint8_t *wave_out = QB ? reinterpret_cast<uint8_t *>(p.q_out) + (s0 - (uint64_t)lane_id() * UNIT) * ELT
                           : reinterpret_cast<uint8_t *>(codes + (s0 - (uint64_t)lane_id() * UNIT));
    uint32_t ovf_seen = 0;
    // The stores lag one round behind (round 4): a wave's vector memory operations retire in order, so the wait for the next round's
    // stream words also waits for every store issued before it — the values are read back from the stage into registers when a
    // piece is complete and stored at the top of the following round, behind that round's stream load: both have a whole round to land.
    static_assert(NP * NR == 4 || NP * NR == 8, "pieces of a lane's staged output");
    uint4 h0, h1, h2, h3, h4, h5, h6, h7;  // (named, not an array: an array written in one lambda and read in another went to scratch memory)
    h0 = h1 = h2 = h3 = h4 = h5 = h6 = h7 = make_uint4(0, 0, 0, 0);
    uint32_t held_i0 = 0;
    bool held_any = false;
    auto flush1 = [&](uint32_t it, const uint4 &h) {
        const uint32_t idx = it * 64 + (uint32_t)lane_id(), c = idx / (NP * NR), j = idx % (NP * NR);
#ifdef LAB_DEC_NT  // (lab: the decoder's values with streaming stores)
        typedef uint32_t u4v __attribute__((ext_vector_type(4)));
        const u4v hv = {h.x, h.y, h.z, h.w};
        __builtin_nontemporal_store(hv, reinterpret_cast<u4v *>(wave_out + ((uint64_t)c * UNIT + held_i0) * ELT + j * 16));
#else
        *reinterpret_cast<uint4 *>(wave_out + ((uint64_t)c * UNIT + held_i0) * ELT + j * 16) = h;
#endif
    };
    auto coop_flush = [&]() {
        if (!held_any) return;
        flush1(0, h0);
        flush1(1, h1);
        flush1(2, h2);
        flush1(3, h3);
        if constexpr (NP * NR > 4) {
            flush1(4, h4);
            flush1(5, h5);
            flush1(6, h6);
            flush1(7, h7);
        }
        held_any = false;
    };
    auto stash1 = [&](uint32_t it, uint4 &h) {
        const uint32_t idx = it * 64 + (uint32_t)lane_id(), c = idx / (NP * NR), j = idx % (NP * NR);
        h = stage[c * (NP * NR + 1) + j];
    };
    auto coop_store = [&](const uint4 (&pc)[NP], uint32_t rnd) {
        const uint32_t sb = rnd % NR;
#pragma unroll
        for (uint32_t j = 0; j < NP; j++) stage[(uint32_t)lane_id() * (NP * NR + 1) + sb * NP + j] = pc[j];
        if (sb != NR - 1) return;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        held_i0 = (rnd - sb) * 16;
        stash1(0, h0);
        stash1(1, h1);
        stash1(2, h2);
        stash1(3, h3);
        if constexpr (NP * NR > 4) {
            stash1(4, h4);
            stash1(5, h5);
            stash1(6, h6);
            stash1(7, h7);
        }
        held_any = true;
        __builtin_amdgcn_wave_barrier();
    };
    // symbols left in the current row (the unit may start inside a row); the running sum restarts at every row start
    uint32_t left = QB ? p.scan_row - (uint32_t)(s0 % p.scan_row) : 0u;
    QO acc = 0;
    if (max_len == 0) {  // single-symbol alphabet: zero-length code
        uint16_t sym = (uint16_t)p.single_sym;
        if (QB) {
            for (uint32_t i = 0; i < nsym; i++) {
                const QO d = sym ? (QO)((int)sym - (int)p.radius) : dec_dout<QO>(p, s0 + i);
                acc += d;
                if (HALF) {  // (the half-width chain's buffer holds int16 / int32: a stream whose every element is a listed delta has values to match)
                    using HS = typename std::conditional<QB == 8, int32_t, int16_t>::type;
                    reinterpret_cast<HS *>(p.q_out)[s0 + i] = (HS)acc;
                    ovf_seen |= (QO)(HS)acc != acc;
                } else {
                    qout[i] = acc;
                }
                if (--left == 0) {
                    acc = 0;
                    left = p.scan_row;
                }
            }
            if (p.carry) reinterpret_cast<QO *>(p.carry)[unit] = acc;
            if (HALF && ovf_seen) atomicOr(p.ovf, 1u);
        } else {
            for (uint32_t i = 0; i < nsym; i++) out[i] = sym;
        }
        return;
    }
    // word offset of the chunk: group offset + the chunks before it inside its group (woff_in: a wave's 64 units are the 32 chunks
    // of ONE group — one load per lane and a wave scan, before the lanes past the array's end left)
    const uint64_t grp = chunk / PACK_GROUP;
    const uint64_t woff = p.group_off[grp] + woff_in;
    const uint32_t *bs = reinterpret_cast<const uint32_t *>(payload + p.bitstream_off);
    const uint64_t wlast = p.total_words ? p.total_words - 1 : 0;  // loads are clamped to the section, never conditional
    // the unit's first bit inside the chunk (a corrupt offset decodes other bits of the section: wrong values, no wrong access)
    const uint32_t bit0 = sub ? (uint32_t)p.sub_bits[chunk * (SZH_SUBS - 1) + sub - 1] : 0u;
    uint64_t buf = 0;  // next bits at the MSB end
    int have = 0;
    uint32_t wi = bit0 >> 5;
    if (bit0 & 31u) {
        const uint64_t a = woff + wi;
        uint32_t wd = __builtin_bswap32(bs[a < wlast ? a : wlast]);
        wd = wi < nwords ? wd : 0u;
        buf = (uint64_t)wd << (32 + (bit0 & 31u));
        have = 32 - (int)(bit0 & 31u);
        wi++;
    }
    const uint32_t nrounds = (nsym + 15) / 16;
    // Stream words, one round ahead (round 4). A round of 16 symbols takes its words from `qw` = the lane's next four; it used to load
    // them at its top — an address known only when the previous round was decoded, so every round began with an exposed trip to
    // the L2 or beyond (and, the operations retiring in order, with the wait for the previous round's stores). Now the top of a
    // round issues the load of the four words BEHIND its own four (`far`), and the next round's four are picked out of the eight
    // the lane then holds by the number of words this round took (0 .. 4: a three-stage select); a round that took more (code
    // words beyond 8 bits on average) loads directly as before.
    typedef uint32_t U4 __attribute__((ext_vector_type(4), aligned(4)));
    auto load4 = [&](uint64_t a, uint32_t (&w)[4]) {
        // ONE 16-byte load at the lane's word position (4-byte aligned: the hardware takes it) instead of four 4-byte ones: a
        // wave's loads touch 64 different cache lines per instruction either way
        if (a + 3 <= wlast) {
            const U4 v = *reinterpret_cast<const U4 *>(bs + a);
            w[0] = v.x;
            w[1] = v.y;
            w[2] = v.z;
            w[3] = v.w;
        } else {  // (the section's last words: clamped, never out of bounds)
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = bs[a + k < wlast ? a + k : wlast];
        }
    };
    uint32_t qw[4], far[4];
    load4(woff + wi, qw);
#pragma unroll
    for (int k = 0; k < 4; k++) qw[k] = __builtin_bswap32(qw[k]);
    if constexpr (MS) {
        // stream words: qw = the lane's next four, far = the four behind them (requested when qw is taken over: four words of decoding
        // ahead of their first use)
        uint32_t qn = 0;
        load4(woff + wi + 4, far);
        auto refill = [&]() {
            if (have <= 32) {
                uint32_t wd = qn == 0 ? qw[0] : (qn == 1 ? qw[1] : (qn == 2 ? qw[2] : qw[3]));
                wd = wi < nwords ? wd : 0u;
                buf |= (uint64_t)wd << (32 - have);
                have += 32;
                wi++;
                qn++;
                if (qn == 4) {
#pragma unroll
                    for (int k = 0; k < 4; k++) qw[k] = __builtin_bswap32(far[k]);
                    qn = 0;
                    load4(woff + wi + 4, far);
                }
            }
        };
        // one code word by the canonical code's arithmetic (any length up to max_len <= 16): its symbol, the buffer moved on
        auto ms_single = [&]() -> uint32_t {
            const uint32_t v = (uint32_t)(buf >> 32);
            uint32_t l = 1;
#pragma unroll
            for (uint32_t q = 1; q < 16; q++) l += (q < max_len && v >= s_upper[q]) ? 1u : 0u;
            uint32_t rank = s_first_rank[l] + ((v >> (32 - l)) - s_first_code[l]);
            rank = rank < n_coded ? rank : n_coded - 1;  // (corrupt streams must not read out of bounds)
            const uint32_t sym = rank < SORTED_LDS ? (uint32_t)s_sorted[rank] : (uint32_t)sorted[rank];
            buf <<= l;
            have -= (int)l;
            return sym;
        };
        int32_t acc32 = 0;
        uint32_t k = 0;
        if (coop) {  // (every lane of the wave decodes a whole unit)
            uint16_t *mine = reinterpret_cast<uint16_t *>(&stage[(uint32_t)lane_id() * (NP * NR + 1)]);  // 80 bytes: a block's 32 values + the overshoot
            for (uint32_t blk = 0; blk < UNIT / 32; blk++) {
                coop_flush();  // (the previous block's pieces, behind the loads requested meanwhile)
                const uint32_t kend = (blk + 1) * 32;
                while (k < kend) {
                    refill();
                    const uint32_t ent = s_lut[(uint32_t)(buf >> 52)];
                    const uint32_t room = left < UNIT - k ? left : UNIT - k;
                    const uint32_t pos = k - blk * 32;
                    if (ent != 0 && room >= 3) {
                        const uint32_t l = (ent & 15u) + 1u, cnt = (ent >> 4) & 3u;
                        const int32_t d1 = (int32_t)(ent << 18) >> 24, s2 = (int32_t)(ent << 9) >> 23, d3 = (int32_t)(ent << 1) >> 24;
                        const int32_t v1 = acc32 + d1, v2 = acc32 + s2, v3 = v2 + d3;
                        acc32 = v3;
                        mine[pos] = (uint16_t)v1;
                        mine[pos + 1] = (uint16_t)v2;
                        mine[pos + 2] = (uint16_t)v3;
                        buf <<= l;
                        have -= (int)l;
                        k += cnt;
                        left -= cnt;
                    } else {
                        const uint32_t sym = ms_single();
                        acc32 += sym ? (int32_t)((int)sym - (int)p.radius) : dec_dout<int32_t>(p, s0 + k);
                        mine[pos] = (uint16_t)acc32;
                        k++;
                        left--;
                    }
                    // (every value of a step lies within 381 of the running sum: a sum within +-32000 keeps them all inside int16)
                    ovf_seen |= (uint32_t)((uint32_t)(acc32 + 32000) > 64000u);
                    if (left == 0) {
                        acc32 = 0;
                        left = p.scan_row;
                    }
                }
                // the block's 64 bytes: from the stage into registers, piece-major (stored at the top of the next block)
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                held_i0 = blk * 32;
                stash1(0, h0);
                stash1(1, h1);
                stash1(2, h2);
                stash1(3, h3);
                held_any = true;
                __builtin_amdgcn_wave_barrier();
                // a lookup that ran past the block's end left its last one or two values behind it: they open the next block
                *reinterpret_cast<uint32_t *>(mine) = *reinterpret_cast<const uint32_t *>(mine + 32);
                __builtin_amdgcn_wave_barrier();
            }
            coop_flush();
        } else {  // (the array's last wave: one symbol at a time, stored directly)
            int16_t *ho = reinterpret_cast<int16_t *>(p.q_out) + s0;
            for (; k < nsym; k++) {
                refill();
                const uint32_t sym = ms_single();
                acc32 += sym ? (int32_t)((int)sym - (int)p.radius) : dec_dout<int32_t>(p, s0 + k);
                ho[k] = (int16_t)acc32;
                ovf_seen |= (uint32_t)(acc32 != (int32_t)(int16_t)acc32);
                if (--left == 0) {
                    acc32 = 0;
                    left = p.scan_row;
                }
            }
        }
        if (p.carry) reinterpret_cast<int32_t *>(p.carry)[unit] = acc32;
        if (__ballot(ovf_seen != 0) && lane_id() == 0) atomicOr(p.ovf, 1u);
        return;
    }
    for (uint32_t rnd = 0; rnd < nrounds; rnd++) {
        const uint32_t i0 = rnd * 16;
#if defined(LAB_DEC_ABL) && (LAB_DEC_ABL & 2)  // (lab, wrong results: no stream loads)
        far[0] = far[1] = far[2] = far[3] = 0x9E3779B9u * (rnd + 1);
#else
        load4(woff + wi + 4, far);
#endif
#if defined(LAB_DEC_ABL) && (LAB_DEC_ABL & 1)  // (lab, wrong results: no stores)
        held_any = false;
#endif
        if (coop) coop_flush();
        uint32_t qn = 0;
        uint32_t syms[16];
        // (round 6) the refill without a branch. `if (have <= 32) { pick the word by qn; ... }` came out of the compiler as a per-lane
        // branch with three more inside (the picks) — and with 64 lanes refilling every ~8 symbols each, some lane takes it at every
        // test. Now every lane runs the same two dozen instructions (masks, not conditional expressions: see the window's hand-over
        // below); only a lane beyond its four words — code words above 8 bits on average — loads where it stands, behind a wave-uniform test.
        // (Tried on top: the test before every second symbol for every book — a table hit takes at most 12 of the 33 bits a refill leaves —
        // with a refill before and after a code word beyond the table, behind wave-uniform tests: 189 -> 228 us at 64 planes, the two
        // ballots per symbol cost more than the tests saved; with the two refills inside the long code's own branch instead — no ballot, only the
        // lanes in there top up —: 223 us as well, 280 against 249 at C2. The test before every symbol stays for books beyond 16 bits.)
        // code books of at most 16-bit words (every alphabet up to 512 symbols): two symbols never need more than the 32 bits
        // a refill guarantees, so the buffer is looked at before every second symbol only
        const bool short_words = max_len <= 16;
        auto refill = [&]() __attribute__((always_inline)) {
            const bool need = have <= 32;
            uint32_t wdx = 0;
            if (__builtin_amdgcn_ballot_w64(need && qn >= 4)) {
                if (need && qn >= 4) {
                    const uint64_t a = woff + wi;
                    wdx = __builtin_bswap32(bs[a < wlast ? a : wlast]);
                }
            }
            const uint32_t q0 = 0u - (uint32_t)(qn == 0), q1 = 0u - (uint32_t)(qn == 1), q2 = 0u - (uint32_t)(qn == 2), q3 = 0u - (uint32_t)(qn == 3);
            uint32_t wd = (qw[0] & q0) | (qw[1] & q1) | (qw[2] & q2) | (qw[3] & q3) | wdx;
            wd &= 0u - (uint32_t)(need && wi < nwords);  // (no refill, or past the unit's words: zero bits — the shift below may then be anything)
            buf |= (uint64_t)wd << ((32u - (uint32_t)have) & 63u);
            const uint32_t nd = need ? 1u : 0u;
            have += (int)(nd << 5);
            wi += nd;
            qn += nd;
        };
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint32_t sym = 0;
            // (symbols past the end of the last chunk decode zero padding: harmless, only the stores are guarded)
            if ((k & 1) == 0 || !short_words) refill();
#if defined(LAB_DEC_ABL) && (LAB_DEC_ABL & 4)  // (lab, wrong results: no table lookup — 4-bit code words)
            const uint32_t ent = ((((uint32_t)(buf >> 60)) + p.radius - 8u) << 8) | 4u;
#else
            const uint32_t ent = s_lut[(uint32_t)(buf >> 32) >> (32 - K)];
#endif
            uint32_t l = ent & 0xFFu;
            sym = ent >> 8;
            if (ent == 0) {  // longer than the table
                const uint32_t v = (uint32_t)(buf >> 32);
                l = K + 1;
#pragma unroll
                for (uint32_t q = DEC_LUT_BITS + 1; q < SZH_MAX_LEN; q++) l += (q > K && v >= s_upper[q]) ? 1u : 0u;
                if (K < DEC_LUT_BITS) {  // (short tables only exist when max_len <= K: never here; keeps the rule general)
                    for (uint32_t q = K + 1; q <= DEC_LUT_BITS && q < SZH_MAX_LEN; q++) l += v >= s_upper[q] ? 1u : 0u;
                }
                l = l > max_len ? max_len : l;
                uint32_t rank = s_first_rank[l] + ((v >> (32 - l)) - s_first_code[l]);
                rank = rank < n_coded ? rank : n_coded - 1;  // (corrupt streams must not read out of bounds)
                const uint32_t rr = rank - base_rank;
                sym = rr < SORTED_LDS ? (uint32_t)s_sorted[rr] : (uint32_t)sorted[rank];
            }
            buf <<= l;
            have -= (int)l;
            syms[k] = sym;
        }
        if (QB) {  // codes -> deltas (0 = delta outlier: looked up) -> running sum, restarted at every row start
            QO qv[16];
            uint32_t zero_any = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) zero_any |= syms[k] == 0 ? 1u : 0u;
            if (!zero_any && left >= 16 && i0 + 16 <= nsym) {
                // the usual round: no delta outlier among the 16 symbols, no row start inside them
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    acc += (QO)((int)syms[k] - (int)p.radius);
                    qv[k] = acc;
                }
                left -= 16;
                if (left == 0) {
                    acc = 0;
                    left = p.scan_row;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const uint32_t sym = syms[k];
                    QO d = (QO)((int)sym - (int)p.radius);
                    if (sym == 0) d = i0 + k < nsym ? dec_dout<QO>(p, s0 + i0 + k) : (QO)0;  // delta outlier (rare)
                    acc += d;
                    qv[k] = acc;
                    if (--left == 0) {
                        acc = 0;
                        left = p.scan_row;
                    }
                }
            }
            if (p.reserved & 1u) {  // (experiment: no output traffic; the sum keeps the work alive)
                QO sx = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) sx += qv[k];
                if (sx == (QO)0x7FFFFFF1) qout[i0] = sx;
            } else if (HALF && QB == 8) {
                // f64 data, int32 out: 64 bytes per lane and round; a value that does not fit raises the flag (the full-width chain follows)
                uint32_t w32[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int64_t a = (int64_t)qv[k];
                    ovf_seen |= (uint32_t)(a != (int64_t)(int32_t)a);
                    w32[k] = (uint32_t)a;
                }
                if (coop) {
                    uint4 pc[NP];
#pragma unroll
                    for (int k = 0; k < 4; k++) pc[k % (int)NP] = make_uint4(w32[4 * k], w32[4 * k + 1], w32[4 * k + 2], w32[4 * k + 3]);
                    coop_store(pc, rnd);
                } else {
                    int32_t *ho = reinterpret_cast<int32_t *>(p.q_out) + s0;
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        if (i0 + k < nsym) ho[i0 + k] = (int32_t)w32[k];
                }
            } else if (HALF) {
                // int16 out: 32 bytes per lane and round; a value that does not fit raises the flag (the full-width chain follows)
                uint32_t hw[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int32_t a = (int32_t)qv[2 * k], b = (int32_t)qv[2 * k + 1];
                    ovf_seen |= (uint32_t)(a != (int32_t)(int16_t)a) | (uint32_t)(b != (int32_t)(int16_t)b);
                    hw[k] = ((uint32_t)a & 0xFFFFu) | ((uint32_t)b << 16);
                }
                if (coop) {
                    uint4 pc[NP];
                    pc[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                    pc[NP > 1 ? 1 : 0] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
                    coop_store(pc, rnd);
                } else {
                    int16_t *ho = reinterpret_cast<int16_t *>(p.q_out) + s0;
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        if (i0 + k < nsym) ho[i0 + k] = (int16_t)qv[k];
                }
            } else if (coop) {
                uint4 pc[NP];
                if (QB == 4) {
#pragma unroll
                    for (int k = 0; k < 4; k++) pc[k] = make_uint4((uint32_t)qv[4 * k], (uint32_t)qv[4 * k + 1], (uint32_t)qv[4 * k + 2], (uint32_t)qv[4 * k + 3]);
                } else {
#pragma unroll
                    for (int k = 0; k < (int)NP; k++) {
                        const unsigned long long a = (unsigned long long)qv[(2 * k) & 15], b = (unsigned long long)qv[(2 * k + 1) & 15];
                        pc[k] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
                    }
                }
                coop_store(pc, rnd);
            } else if (i0 + 16 <= nsym) {
                if (QB == 4) {
                    uint4 *o4 = reinterpret_cast<uint4 *>(qout + i0);
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        o4[k] = make_uint4((uint32_t)qv[4 * k], (uint32_t)qv[4 * k + 1], (uint32_t)qv[4 * k + 2], (uint32_t)qv[4 * k + 3]);
                } else {
                    ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(qout + i0);
#pragma unroll
                    for (int k = 0; k < 8; k++) o2[k] = make_ulonglong2((unsigned long long)qv[2 * k], (unsigned long long)qv[2 * k + 1]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if (i0 + k < nsym) qout[i0 + k] = qv[k];
            }
        } else if (coop) {
            uint32_t packed[8];
#pragma unroll
            for (int k = 0; k < 8; k++) packed[k] = syms[2 * k] | (syms[2 * k + 1] << 16);
            uint4 pc[NP];
            pc[0] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
            pc[NP > 1 ? 1 : 0] = make_uint4(packed[4], packed[5], packed[6], packed[7]);
            coop_store(pc, rnd);
        } else if (i0 + 16 <= nsym) {
            uint32_t packed[8];
#pragma unroll
            for (int k = 0; k < 8; k++) packed[k] = syms[2 * k] | (syms[2 * k + 1] << 16);
            uint4 *o4 = reinterpret_cast<uint4 *>(out + i0);
            o4[0] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
            o4[1] = make_uint4(packed[4], packed[5], packed[6], packed[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (i0 + k < nsym) out[i0 + k] = (uint16_t)syms[k];
        }
        {   // the next round's four words: the eight held (qw, far) from word qn on
            uint32_t t[7], u[5];
#pragma unroll
            for (int k = 0; k < 4; k++) far[k] = __builtin_bswap32(far[k]);
            // (bit masks, not conditional expressions: hipcc turns `c ? t[k + 2] : t[k]` into an indexed load from a copy of t in scratch memory)
            const uint32_t m4 = 0u - ((qn >> 2) & 1u), m2 = 0u - ((qn >> 1) & 1u), m1 = 0u - (qn & 1u);
#pragma unroll
            for (int k = 0; k < 4; k++) t[k] = (far[k] & m4) | (qw[k] & ~m4);
#pragma unroll
            for (int k = 0; k < 3; k++) t[4 + k] = far[k];
#pragma unroll
            for (int k = 0; k < 5; k++) u[k] = (t[k + 2] & m2) | (t[k] & ~m2);
#pragma unroll
            for (int k = 0; k < 4; k++) qw[k] = (u[k + 1] & m1) | (u[k] & ~m1);
            if (qn > 4) {  // (more than four words in sixteen symbols: long code words — loaded where the lane now stands)
                load4(woff + wi, qw);
#pragma unroll
                for (int k = 0; k < 4; k++) qw[k] = __builtin_bswap32(qw[k]);
            }
        }
    }
    if (coop) coop_flush();
    if (QB && p.carry) reinterpret_cast<QO *>(p.carry)[unit] = acc;
    if (HALF && __ballot(ovf_seen != 0) && lane_id() == 0) atomicOr(p.ovf, 1u);
}

// The decoder's running sums restart in every unit: a unit that starts inside a row misses what the row's earlier units summed
// up — carry[] of the units from the one that holds the row's start to its predecessor (a unit's carry is its running sum at
// its end, i.e. since the last row start inside it or since its own start). One wave per unit; rows of at most one chunk.
template <typename QO>
__device__ __forceinline__ bool unit_carry_in(const QO *__restrict__ carry, uint64_t n, uint64_t u, uint32_t L, uint64_t &s0, uint64_t &seg, QO &cin) {
    s0 = u * SZH_UNIT_SYMS;
    if (u == 0 || s0 >= n) return false;
    const uint32_t x0 = (uint32_t)(s0 % L);
    if (x0 == 0) return false;
    seg = L - x0;
    if (seg > SZH_UNIT_SYMS) seg = SZH_UNIT_SYMS;
    if (seg > n - s0) seg = n - s0;
    const uint64_t first = (s0 - x0) / SZH_UNIT_SYMS;  // the unit that holds the row's start
    QO part = 0;
    for (uint64_t v = first + lane_id(); v < u; v += WAVE) part += carry[v];
    cin = wave_sum(part);
    return cin != 0;
}
template <typename QO>
__global__ __launch_bounds__(256) void k_scan_carry(QO *__restrict__ q, const QO *__restrict__ carry, uint64_t n, uint32_t L, const uint32_t *gate) {
    if (gate && *gate == 0) return;  // (the full-width chain behind a half-width one that did not overflow)
    uint64_t s0, seg;
    QO cin;
    if (!unit_carry_in(carry, n, (uint64_t)blockIdx.x * 4 + threadIdx.x / WAVE, L, s0, seg, cin)) return;
    for (uint32_t i = lane_id(); i < seg; i += WAVE) q[s0 + i] += cin;
}
// the same on the half-width chain's int16 values (the carries stay 32 bits wide); a sum that does not fit raises the flag
template <typename HS, typename QC>  // int16 values / int32 carries (f32 data), int32 / int64 (f64 data)
__global__ __launch_bounds__(256) void k_scan_carry_half(HS *__restrict__ q, const QC *__restrict__ carry, uint64_t n, uint32_t L, uint32_t *ovf) {
    uint64_t s0, seg;
    QC cin;
    if (!unit_carry_in(carry, n, (uint64_t)blockIdx.x * 4 + threadIdx.x / WAVE, L, s0, seg, cin)) return;
    bool bad = false;
    for (uint32_t i = lane_id(); i < seg; i += WAVE) {
        const QC v = (QC)q[s0 + i] + cin;
        bad |= v != (QC)(HS)v;
        q[s0 + i] = (HS)v;
    }
    if (__ballot(bad) && lane_id() == 0) atomicOr(ovf, 1u);
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

// inclusive scan along the contiguous axis for rows of up to SCANX_SEG elements: one wave per row, 4 elements per lane
// and step, no barrier; FROM_CODES reads the decoded u16 codes directly (code c -> delta c - radius, 0 -> 0) instead
// of a previously expanded delta array.
template <typename Q, bool FROM_CODES>
__global__ __launch_bounds__(256) void k_scan_x_wave(Q *__restrict__ q, const uint16_t *__restrict__ codes, int radius,
                                                     uint64_t L, uint64_t nrows) {
    using UQ = typename std::make_unsigned<Q>::type;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + threadIdx.x / WAVE, nwaves = (uint64_t)gridDim.x * 4;
    const int lane = lane_id();
    for (uint64_t row = wave0; row < nrows; row += nwaves) {
        Q *r = q + row * L;
        const uint16_t *c = codes + row * L;
        UQ carry = 0;
        for (uint64_t b = 0; b < L; b += 256) {
            const uint64_t i0 = b + (uint64_t)lane * 4;
            UQ v[4];
            if (FROM_CODES) {
                if (i0 + 4 <= L && (((row * L + i0) & 3) == 0)) {
                    const uint2 w = *reinterpret_cast<const uint2 *>(c + i0);
                    const uint32_t cc[4] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16};
#pragma unroll
                    for (int k = 0; k < 4; k++) v[k] = cc[k] ? (UQ)(Q)((int)cc[k] - radius) : (UQ)0;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int cc = (i0 + k < L) ? (int)c[i0 + k] : 0;
                        v[k] = cc ? (UQ)(Q)(cc - radius) : (UQ)0;
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = (i0 + k < L) ? (UQ)r[i0 + k] : (UQ)0;
            }
            v[1] += v[0];
            v[2] += v[1];
            v[3] += v[2];
            const UQ incl = wave_incl_scan(v[3]);
            const UQ off = carry + (incl - v[3]);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i0 + k < L) r[i0 + k] = (Q)(v[k] + off);
            carry += __shfl(incl, WAVE - 1, WAVE);
        }
    }
}

// inclusive scan along a strided axis: one thread per (line, segment), lanes adjacent along the contiguous axis.
// element index = outer * (L * inner) + a * inner + in,  a = 0..L-1 ; inner = product of faster dims.
// Arrays with few lines (2-D, or a short slow axis product) cut every line into S segments so that the launch still fills
// the chip: k_strided_totals sums the segments first (one extra read), the scan adds the totals of the preceding segments.
// The last strided scan of the reconstruction also turns the lattice index into the value (same buffer, Q and T have the
// same size).
// Rows that are multiples of the decoder's unit (512, 768, 1024): the x prefix sums the decoder wrote restart at every unit, and
// what the row's earlier units summed up (szk_dec_params::carry) is added by the FIRST strided pass as it reads — element e at
// column x of its row gets the carries of the row's units before its own (SZH_SUBS - 1 loads at most, none in a row's first
// unit) — instead of a pass of its own over the array (k_scan_carry: 0.12 ms at 512^3 with units of 256).
// A strided line keeps its column: j = units of the row before the line's own, ub = first unit of the row the line starts in,
// ustep = units between two steps along the line (inner / unit: inner is a multiple of the row).
struct RowCarry {
    uint32_t j;
    uint64_t ub, ustep;
    __device__ __forceinline__ RowCarry(bool on, uint64_t base, uint64_t inner, uint32_t row) {
        const uint32_t x = on ? (uint32_t)(base % row) : 0u;
        j = x / SZH_UNIT_SYMS;
        ub = (base - x) / SZH_UNIT_SYMS;
        ustep = inner / SZH_UNIT_SYMS;
    }
    template <typename QC>
    __device__ __forceinline__ QC at(const QC *__restrict__ carry, uint64_t a) const {  // step a of the line
        QC c = 0;
        const uint64_t u0 = ub + a * ustep;
#pragma unroll
        for (uint32_t i = 0; i < SZH_SUBS - 1; i++) c += i < j ? carry[u0 + i] : (QC)0;
        return c;
    }
};
template <typename Q>
__global__ __launch_bounds__(256) void k_strided_totals(const Q *__restrict__ q, uint64_t L, uint64_t inner, uint64_t nlines, uint32_t S,
                                                        uint64_t Lseg, Q *__restrict__ totals, const Q *__restrict__ carry, uint32_t row) {
    using UQ = typename std::make_unsigned<Q>::type;
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nlines * S) return;
    const uint64_t seg = id / nlines, line = id % nlines;
    const uint64_t outer = line / inner, in = line % inner;
    const uint64_t base = outer * L * inner + in;
    const Q *pp = q + base;
    const uint64_t a0 = seg * Lseg, a1 = (a0 + Lseg < L) ? a0 + Lseg : L;
    UQ run = 0;
    const RowCarry rc(carry != nullptr, base, inner, row);
    for (uint64_t a = a0; a < a1; a++) run += (UQ)pp[a * inner] + (rc.j ? (UQ)rc.at(carry, a) : (UQ)0);
    totals[id] = (Q)run;
}
template <typename Q, typename T, bool DEQUANT, bool CARRY>
__device__ __forceinline__ void scan_strided_body(void *buf, uint64_t L, uint64_t inner, uint64_t nlines, uint32_t S, uint64_t Lseg,
                                                  const Q *__restrict__ totals, szk_lattice l, const Q *__restrict__ carry, uint32_t row) {
    using UQ = typename std::make_unsigned<Q>::type;
    const Lattice<T> lat(l);
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nlines * S) return;
    const uint64_t seg = id / nlines, line = id % nlines;
    const uint64_t outer = line / inner, in = line % inner;
    const uint64_t base = outer * L * inner + in;
    Q *pp = reinterpret_cast<Q *>(buf) + base;
    T *po = reinterpret_cast<T *>(buf) + base;
    const uint64_t a1 = (seg * Lseg + Lseg < L) ? seg * Lseg + Lseg : L;
    UQ run = 0;
    for (uint32_t k = 0; k < seg; k++) run += (UQ)totals[(uint64_t)k * nlines + line];
    uint64_t a = seg * Lseg;
    const RowCarry rc(CARRY, base, inner, row);
    const bool carried = CARRY && rc.j != 0;
    for (; a + 4 <= a1; a += 4) {
        UQ v0 = (UQ)pp[(a + 0) * inner], v1 = (UQ)pp[(a + 1) * inner], v2 = (UQ)pp[(a + 2) * inner],
           v3 = (UQ)pp[(a + 3) * inner];
        if (carried) {
            v0 += (UQ)rc.at(carry, a + 0);
            v1 += (UQ)rc.at(carry, a + 1);
            v2 += (UQ)rc.at(carry, a + 2);
            v3 += (UQ)rc.at(carry, a + 3);
        }
        v0 += run;
        v1 += v0;
        v2 += v1;
        v3 += v2;
        if (DEQUANT) {
            __builtin_nontemporal_store(lat.dequant((Q)v0), &po[(a + 0) * inner]);  // (final values: streaming stores, see k_scan_strided_half)
            __builtin_nontemporal_store(lat.dequant((Q)v1), &po[(a + 1) * inner]);
            __builtin_nontemporal_store(lat.dequant((Q)v2), &po[(a + 2) * inner]);
            __builtin_nontemporal_store(lat.dequant((Q)v3), &po[(a + 3) * inner]);
        } else {
            pp[(a + 0) * inner] = (Q)v0;
            pp[(a + 1) * inner] = (Q)v1;
            pp[(a + 2) * inner] = (Q)v2;
            pp[(a + 3) * inner] = (Q)v3;
        }
        run = v3;
    }
    for (; a < a1; a++) {
        run += (UQ)pp[a * inner] + (carried ? (UQ)rc.at(carry, a) : (UQ)0);
        if (DEQUANT) __builtin_nontemporal_store(lat.dequant((Q)run), &po[a * inner]);
        else pp[a * inner] = (Q)run;
    }
}
template <typename Q>
__global__ __launch_bounds__(256) void k_scan_strided(Q *__restrict__ q, uint64_t L, uint64_t inner, uint64_t nlines, uint32_t S,
                                                      uint64_t Lseg, const Q *__restrict__ totals, const uint32_t *gate,
                                                      const Q *__restrict__ carry, uint32_t row) {
    using T = typename std::conditional<sizeof(Q) == 4, float, double>::type;
    if (gate && *gate == 0) return;
    if (carry) scan_strided_body<Q, T, false, true>(q, L, inner, nlines, S, Lseg, totals, szk_lattice{}, carry, row);
    else scan_strided_body<Q, T, false, false>(q, L, inner, nlines, S, Lseg, totals, szk_lattice{}, carry, row);
}
template <typename T>
__global__ __launch_bounds__(256) void k_scan_strided_dequant(void *buf, uint64_t L, uint64_t inner, uint64_t nlines, uint32_t S, uint64_t Lseg,
                                                              const typename QTraits<T>::Q *__restrict__ totals, szk_lattice l, const uint32_t *gate,
                                                              const typename QTraits<T>::Q *__restrict__ carry, uint32_t row) {
    if (gate && *gate == 0) return;
    if (carry) scan_strided_body<typename QTraits<T>::Q, T, true, true>(buf, L, inner, nlines, S, Lseg, totals, l, carry, row);
    else scan_strided_body<typename QTraits<T>::Q, T, true, false>(buf, L, inner, nlines, S, Lseg, totals, l, carry, row);
}
// Half-width strided scans (f32 data, int16 storage, see szk_dec_params::half): one thread per PAIR of adjacent lines (two
// neighbouring x), marching along the axis; sums in int32. DEQ = false: in place, a sum outside int16 raises the flag;
// DEQ = true (the last axis): int16 in, dequantised float out.
template <bool DEQ, bool CARRY>
__global__ __launch_bounds__(256) void k_scan_strided_half(const int16_t *__restrict__ in, void *__restrict__ outp, uint64_t L, uint64_t inner,
                                                           uint64_t nlines, szk_lattice l, uint32_t *ovf, const int32_t *__restrict__ carry, uint32_t row) {
    const Lattice<float> lat(l);
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // pair index
    if (id * 2 >= nlines) return;
    const uint64_t line = id * 2, outer = line / inner, inn = line % inner;
    const uint64_t base = outer * L * inner + inn;
    const RowCarry rc(CARRY, base, inner, row);  // (a pair lies in one unit)
    const bool carried = CARRY && rc.j != 0;
    const uint32_t *pin = reinterpret_cast<const uint32_t *>(in + base);  // (inner is even: a pair is one aligned word)
    const uint64_t step = inner / 2;                                       // words between consecutive elements of a line
    int32_t r0 = 0, r1 = 0;
    uint32_t bad = 0;
    constexpr int DEPTH = 16;  // loads in flight per thread (half as many threads as the full-width scans: twice their depth)
    for (uint64_t a = 0; a < L; a += DEPTH) {
        uint32_t w[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) w[k] = a + k < L ? pin[(a + k) * step] : 0u;
        int32_t cw[CARRY ? DEPTH : 1];
        if (CARRY && carried) {
#pragma unroll
            for (int k = 0; k < DEPTH; k++) cw[k] = a + k < L ? rc.at(carry, a + k) : 0;
        }
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            if (a + k >= L) break;
            const int32_t c = (CARRY && carried) ? cw[CARRY ? k : 0] : 0;
            r0 += (int32_t)(int16_t)(w[k] & 0xFFFFu) + c;
            r1 += (int32_t)(int16_t)(w[k] >> 16) + c;
            if (DEQ) {
                // (round 5: the final values leave with streaming stores — nothing on the device reads them again, and 537 MB of them
                // need not push the int16 values still to be read out of the caches: C2 decompress 574 -> 530 us)
                typedef float f2v __attribute__((ext_vector_type(2)));
                const f2v v = {lat.dequant(r0), lat.dequant(r1)};
                __builtin_nontemporal_store(v, reinterpret_cast<f2v *>(reinterpret_cast<float *>(outp) + base + (a + k) * inner));
            } else {
                bad |= (uint32_t)(r0 != (int32_t)(int16_t)r0) | (uint32_t)(r1 != (int32_t)(int16_t)r1);
                reinterpret_cast<uint32_t *>(reinterpret_cast<int16_t *>(outp) + base)[(a + k) * step] = ((uint32_t)r0 & 0xFFFFu) | ((uint32_t)r1 << 16);
            }
        }
    }
    if (!DEQ && __ballot(bad != 0) && lane_id() == 0) atomicOr(ovf, 1u);
}
// the same for f64 data: int32 storage, sums in int64, double out (a thread per pair of adjacent lines: one 8-byte word)
template <bool DEQ, bool CARRY>
__global__ __launch_bounds__(256) void k_scan_strided_half64(const int32_t *__restrict__ in, void *__restrict__ outp, uint64_t L, uint64_t inner,
                                                             uint64_t nlines, szk_lattice l, uint32_t *ovf, const int64_t *__restrict__ carry, uint32_t row) {
    const Lattice<double> lat(l);
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // pair index
    if (id * 2 >= nlines) return;
    const uint64_t line = id * 2, outer = line / inner, inn = line % inner;
    const uint64_t base = outer * L * inner + inn;
    const RowCarry rc(CARRY, base, inner, row);
    const bool carried = CARRY && rc.j != 0;
    const uint2 *pin = reinterpret_cast<const uint2 *>(in + base);  // (inner is even: a pair is one aligned 8-byte word)
    const uint64_t step = inner / 2;
    int64_t r0 = 0, r1 = 0;
    uint32_t bad = 0;
    constexpr int DEPTH = 16;
    for (uint64_t a = 0; a < L; a += DEPTH) {
        uint2 w[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) w[k] = pin[(a + k < L ? a + k : L - 1) * step];
        int64_t cw[CARRY ? DEPTH : 1];
        if (CARRY && carried) {
#pragma unroll
            for (int k = 0; k < DEPTH; k++) cw[k] = a + k < L ? rc.at(carry, a + k) : 0;
        }
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            if (a + k >= L) break;
            const int64_t c = (CARRY && carried) ? cw[CARRY ? k : 0] : 0;
            r0 += (int64_t)(int32_t)w[k].x + c;
            r1 += (int64_t)(int32_t)w[k].y + c;
            if (DEQ) {
                typedef double d2v __attribute__((ext_vector_type(2)));
                const d2v v = {lat.dequant(r0), lat.dequant(r1)};
                __builtin_nontemporal_store(v, reinterpret_cast<d2v *>(reinterpret_cast<double *>(outp) + base + (a + k) * inner));
            } else {
                bad |= (uint32_t)(r0 != (int64_t)(int32_t)r0) | (uint32_t)(r1 != (int64_t)(int32_t)r1);
                reinterpret_cast<uint2 *>(reinterpret_cast<int32_t *>(outp) + base)[(a + k) * step] = make_uint2((uint32_t)r0, (uint32_t)r1);
            }
        }
    }
    if (!DEQ && __ballot(bad != 0) && lane_id() == 0) atomicOr(ovf, 1u);
}
__global__ __launch_bounds__(256) void k_dequant_half64(const int32_t *__restrict__ in, double *__restrict__ out, uint64_t n, szk_lattice l) {
    const Lattice<double> lat(l);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = lat.dequant((int64_t)in[i]);
}
__global__ __launch_bounds__(256) void k_dequant_half(const int16_t *__restrict__ in, float *__restrict__ out, uint64_t n, szk_lattice l) {
    const Lattice<float> lat(l);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = lat.dequant((int32_t)in[i]);
}

// lattice index -> value, in place (Q and T have the same size)
template <typename T>
__global__ __launch_bounds__(256) void k_dequant(void *buf, uint64_t n, szk_lattice l, const uint32_t *gate) {
    using Q = typename QTraits<T>::Q;
    if (gate && *gate == 0) return;
    const Lattice<T> lat(l);
    Q *q = reinterpret_cast<Q *>(buf);
    T *o = reinterpret_cast<T *>(buf);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Q v = q[i];
        o[i] = lat.dequant(v);
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

// ---- integer inputs ride the f64 pipeline: every element type of the reference's HDF5 filter (tools/H5Z-SZ3/src/H5Z_SZ3.cpp:195-227 —
// 8 / 16 / 32 / 64-bit, signed and unsigned) widened on the device; exact up to 2^53 in magnitude (beyond: the array stays lossless)
template <typename I>
__global__ __launch_bounds__(256) void k_int_to_f64(const I *__restrict__ in, uint64_t n, double *__restrict__ out, uint32_t *too_big) {
    bool big = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const I v = in[i];
        if (sizeof(I) == 8) {
            if (std::is_signed<I>::value) {
                const long long a = (long long)v;
                big |= a > (1ll << 53) || a < -(1ll << 53);
            } else {
                big |= (unsigned long long)v > (1ull << 53);
            }
        }
        out[i] = (double)v;
    }
    if (__ballot(big) && (threadIdx.x & 63) == 0) atomicOr(too_big, 1u);
}
template <typename I>
__global__ __launch_bounds__(256) void k_f64_to_int(const double *__restrict__ in, uint64_t n, I *__restrict__ out) {
    // (the reconstruction of an in-range integer within a bound >= 1 may leave the type's range by the bound: clamped, like a cast would not)
    constexpr double lo = std::is_signed<I>::value ? -(double)(1ull << (8 * sizeof(I) - 1)) : 0.0;
    constexpr double hi = sizeof(I) == 8 ? (std::is_signed<I>::value ? 9223372036854774784.0 : 18446744073709549568.0)  // largest doubles below 2^63 / 2^64
                                         : (double)((1ull << (8 * sizeof(I) - (std::is_signed<I>::value ? 1 : 0))) - 1ull);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const double v = fmin(fmax(rint(in[i]), lo), hi);
        out[i] = std::is_signed<I>::value ? (I)(long long)v : (I)(unsigned long long)v;
    }
}
// sz_type: SZ_UINT8 = 2, SZ_INT8 = 3, SZ_UINT16 = 4, SZ_INT16 = 5, SZ_UINT32 = 6, SZ_INT32 = 7, SZ_UINT64 = 8, SZ_INT64 = 9 (include/SZ3/def.hpp:27-36)
int szk_launch_int_to_f64(int sz_type, const void *d_in, uint64_t n, double *d_out, uint32_t *d_flag, hipStream_t s) {
    const uint32_t g = grid_for(n, 256, 65536);
#define SZK_W(T) hipLaunchKernelGGL((k_int_to_f64<T>), dim3(g), dim3(256), 0, s, (const T *)d_in, n, d_out, d_flag)
    switch (sz_type) {
        case 2: SZK_W(uint8_t); break;
        case 3: SZK_W(int8_t); break;
        case 4: SZK_W(uint16_t); break;
        case 5: SZK_W(int16_t); break;
        case 6: SZK_W(uint32_t); break;
        case 7: SZK_W(int32_t); break;
        case 8: SZK_W(uint64_t); break;
        case 9: SZK_W(int64_t); break;
        default: return -1;
    }
#undef SZK_W
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_f64_to_int(int sz_type, const double *d_in, uint64_t n, void *d_out, hipStream_t s) {
    const uint32_t g = grid_for(n, 256, 65536);
#define SZK_N(T) hipLaunchKernelGGL((k_f64_to_int<T>), dim3(g), dim3(256), 0, s, d_in, n, (T *)d_out)
    switch (sz_type) {
        case 2: SZK_N(uint8_t); break;
        case 3: SZK_N(int8_t); break;
        case 4: SZK_N(uint16_t); break;
        case 5: SZK_N(int16_t); break;
        case 6: SZK_N(uint32_t); break;
        case 7: SZK_N(int32_t); break;
        case 8: SZK_N(uint64_t); break;
        case 9: SZK_N(int64_t); break;
        default: return -1;
    }
#undef SZK_N
    SZK_CHECK_LAUNCH();
    return 0;
}

// ---- several slabs on one GPU: their code histograms are summed before / instead of the RCCL exchange ------------------
__global__ __launch_bounds__(256) void k_hist_add(uint64_t *__restrict__ dst, const uint64_t *__restrict__ src, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
int szk_launch_hist_add(uint64_t *d_dst, const uint64_t *d_src, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_hist_add, dim3((n + 255) / 256), dim3(256), 0, s, d_dst, d_src, n);
    SZK_CHECK_LAUNCH();
    return 0;
}

int szk_launch_minmax(int dtype, const void *d_in, uint64_t n, double *d_partial, double *d_out, hipStream_t s) {
    const int nb = 1024;
    if (dtype == 0) hipLaunchKernelGGL(k_minmax<float>, dim3(nb), dim3(256), 0, s, (const float *)d_in, n, d_partial);
    else hipLaunchKernelGGL(k_minmax<double>, dim3(nb), dim3(256), 0, s, (const double *)d_in, n, d_partial);
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(64), 0, s, d_partial, nb, d_out);
    SZK_CHECK_LAUNCH();
    return 0;
}

int szk_force_generic = 0;  // test hook: route every shape through the generic kernel
int szk_dbg_flags = 0;     // ablation switches (tools/k1_lab.py)

// persistent grid = resident workgroups of the kernel on this device (occupancy API x CU count), capped by the
// number of hist_partial rows and by the tile count
static uint32_t k1_grid(const void *kernel, uint64_t ntiles) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    // (the occupancy query is a few microseconds of host time in front of stage 1's launch — the step's critical path: asked once per kernel)
    static std::mutex occ_mu;
    static std::vector<std::pair<const void *, int>> occ;
    int per_cu = 0;
    {
        std::lock_guard<std::mutex> lk(occ_mu);
        for (const auto &e : occ)
            if (e.first == kernel) per_cu = e.second;
        if (!per_cu) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
            occ.emplace_back(kernel, per_cu);
        }
    }
    uint64_t g = (uint64_t)per_cu * (uint64_t)n_cu;
    if (g > SZK_K1_GRID) g = SZK_K1_GRID;
    if (g > ntiles) g = ntiles;
    if (g == 0) g = 1;
    // The workgroups are persistent and take tiles by stride: a grid that needs 1.6 rounds (8192 brick tasks on 5 x 256
    // workgroups of 4 waves) leaves 40 % of the chip idle in its last round — 166 us against 147 for the same kernel on 4 x 256.
    // Take the smallest grid that needs the same number of rounds: every round is then (nearly) full.
    const uint64_t rounds = (ntiles + g - 1) / g;
    uint64_t gb = (ntiles + rounds - 1) / rounds;
    if (gb % 8 && gb + (8 - gb % 8) <= g) gb += 8 - gb % 8;  // (a multiple of 8 keeps the XCD-aware task order)
    return (uint32_t)gb;
}

// the marching kernel + histogram fold. When one-byte codes are possible both specialisations are launched with the same
// grid (= rows of the fold); the one the probe did not choose returns at once.
template <typename T, int NDIM, int TY, bool WIN16>
static void launch_march_w(const void *d_in, uint16_t *codes, szk_k1_params &p, uint64_t nb, hipStream_t s) {
    uint32_t grid = 0;
    p.seg_expected = 0;
    p.assumed_q16 = 0;
    if (p.prof_ev0) (void)hipEventRecord((hipEvent_t)p.prof_ev0, s);
    if (p.mode.allow && !(szk_dbg_flags & 256) && p.hint_narrow > 0 && !(szk_dbg_flags & 131072)) {
        // the context's previous call took one-byte codes: one launch of the form built around them (it decides the width from
        // THIS call's probe and handles either). After a two-byte call the two specialisations below are launched as on a first
        // call: the run-time-width form is a third slower on two-byte codes (574 vs 363 us at C4's slab) than the specialised one
        // plus the 4 us of its returning twin.
        p.seg_expected = p.spec_lens && p.d[3] % MARCH_TX == 0;
        // the fused form: NDIM 3 bodies (1-D ... 3-D arrays), rows cut into whole segments, and a scratch that holds a slot per task
        const uint32_t slot_words = fuse_slot_words(TY, p.d[2], p.d[1]);
        const bool fuse = NDIM == 3 && p.fuse && p.seg_expected && p.fuse_enc && p.fuse_info && p.fuse_slots && p.seg_base && p.fuse_flag &&
                          nb * (uint64_t)slot_words + 2 <= p.fuse_cap_words && nb * (uint64_t)slot_words < (1ull << 32) && !(szk_dbg_flags & 2048);
#ifndef SZ3HIP_LAB  // (the fused form — round 4, slower than two passes on this chip — is part of the lab build only: python -m sz3_amd.build --lab)
        const bool fuse_built = false;
#else
        const bool fuse_built = true;
#endif
        p.fused = fuse && fuse_built ? 1 : 0;
        if (p.fused) {
#ifdef SZ3HIP_LAB
            if constexpr (NDIM == 3) {
                grid = k1_grid((const void *)k_lorenzo_quant_march3f<T, 3, TY>, (nb + 3) / 4);
                p.fuse_geom[3] = slot_words;
                hipLaunchKernelGGL((k_lorenzo_quant_march3f<T, 3, TY>), dim3(grid), dim3(256), 0, s, (const T *)d_in, codes, p, (uint32_t)nb, grid);
            }
#endif
        } else if (NDIM == 3 && sizeof(T) == 4 && p.hint_q16 > 0 && p.q16_flag && p.d[0] == 1 && !(szk_dbg_flags & 8)) {
            // the 16-bit form: the previous call's probe saw lattice values within +-Q16_LIM / 2 only (debug flag 8 keeps the form below)
            if constexpr (NDIM == 3 && sizeof(T) == 4) {
                p.assumed_q16 = 1;
                if (p.samp.words && !(szk_dbg_flags & 4194304)) {
                    // the sampled book inside the launch: SZK_SAMP_ROLES workgroups in front of the workers take the sample and build the book,
                    // the workers sum the segments' bits with it (and keep no histogram: nothing to fold)
                    p.samp_in_launch = 1;
                    p.seg_expected = 1;
                    grid = k1_grid((const void *)k_lorenzo_quant_march3q<TY, true>, (nb + 3) / 4);
                    hipLaunchKernelGGL((k_lorenzo_quant_march3q<TY, true>), dim3(grid + SZK_SAMP_ROLES), dim3(256), 0, s, (const float *)d_in, codes, p, (uint32_t)nb, grid);
                    grid = 0;  // (no histogram rows)
                } else {
                    grid = k1_grid((const void *)k_lorenzo_quant_march3q<TY, false>, (nb + 3) / 4);
                    hipLaunchKernelGGL((k_lorenzo_quant_march3q<TY, false>), dim3(grid), dim3(256), 0, s, (const float *)d_in, codes, p, (uint32_t)nb, grid);
                }
            }
        } else if (NDIM == 3 && sizeof(T) == 4 && p.samp.words && p.d[0] == 1 && !(szk_dbg_flags & 4194304)) {
            // (f32 only: the f64 kernel with the sampling code inlined needs a stack — 20 bytes a lane, paid by every wave — and runs at
            // three waves per SIMD as it is; f64 streams get their sampled book from k_sample behind the launch)
            if constexpr (NDIM == 3 && sizeof(T) == 4) {  // the one-byte kernel with the sampling workgroups in front (as the 16-bit form above)
                p.samp_in_launch = 1;
                p.seg_expected = 1;
                grid = k1_grid((const void *)k_lorenzo_quant_march3<T, 3, TY, true>, (nb + 3) / 4);
                hipLaunchKernelGGL((k_lorenzo_quant_march3<T, 3, TY, true>), dim3(grid + SZK_SAMP_ROLES), dim3(256), 0, s, (const T *)d_in, codes, p, (uint32_t)nb, grid);
                grid = 0;  // (no histogram rows)
            }
        } else {
            grid = k1_grid((const void *)k_lorenzo_quant_march3<T, NDIM, TY>, (nb + 3) / 4);
            hipLaunchKernelGGL((k_lorenzo_quant_march3<T, NDIM, TY>), dim3(grid), dim3(256), 0, s, (const T *)d_in, codes, p, (uint32_t)nb, grid);
        }
    } else if (p.mode.allow && !(szk_dbg_flags & 256)) {
        // (first call of a context) each specialisation gets the grid its own occupancy allows (all workgroups resident: the tasks are dealt by stride;
        // the two-byte kernel holds 38-72 KB of LDS); the fold reads the larger number of rows, the two-byte kernel leaves
        // them all empty
        const uint32_t g1 = k1_grid((const void *)k_lorenzo_quant_march<T, NDIM, TY, 1, false>, (nb + 3) / 4);
        const uint32_t g2 = k1_grid((const void *)k_lorenzo_quant_march<T, NDIM, TY, 2, WIN16>, (nb + 3) / 4);
        grid = g1 > g2 ? g1 : g2;
        // (a context that may not take the one-launch form — its histogram is exchanged between the stages — but has the previous
        // call's code lengths: the one-byte specialisation sums the segments' bits all the same; should the probe choose two
        // bytes, the flag it leaves unset tells the packer's book role and stage 2 is repeated)
        p.seg_expected = p.spec_lens && p.d[3] % MARCH_TX == 0;
        hipLaunchKernelGGL((k_lorenzo_quant_march<T, NDIM, TY, 1, false>), dim3(g1), dim3(256), 0, s, (const T *)d_in, codes, p, (uint32_t)nb, grid);
        hipLaunchKernelGGL((k_lorenzo_quant_march<T, NDIM, TY, 2, WIN16>), dim3(g2), dim3(256), 0, s, (const T *)d_in, codes, p, (uint32_t)nb, grid);
    } else {
        grid = k1_grid((const void *)k_lorenzo_quant_march<T, NDIM, TY, 0, WIN16>, (nb + 3) / 4);
        hipLaunchKernelGGL((k_lorenzo_quant_march<T, NDIM, TY, 0, WIN16>), dim3(grid), dim3(256), 0, s, (const T *)d_in, codes, p, (uint32_t)nb, grid);
    }
    if (p.prof_ev1) (void)hipEventRecord((hipEvent_t)p.prof_ev1, s);
    p.fold_rows = grid;
    if (!p.defer_fold && grid) hipLaunchKernelGGL(k_hist_reduce, dim3(HIST_WIN / 256, 64), dim3(256), 0, s, p.hist_partial, grid, (int)p.radius - HIST_WIN / 2, p.hist, p.range);
    // the sampled book behind the forms that do not carry the sampling workgroups: a launch of its own (the probe is complete: a stream
    // of two-byte codes leaves the words untouched and the classic code book is built)
    if (p.samp.words && !p.samp_in_launch) {
        if constexpr (NDIM == 3) {
            if (p.mode.allow && !(szk_dbg_flags & 256)) hipLaunchKernelGGL((k_sample<T>), dim3(SZK_SAMP_ROLES), dim3(256), 0, s, (const T *)d_in, p);
            else p.samp.words = nullptr;
        } else {
            p.samp.words = nullptr;
        }
    }
}
// the marching kernel + histogram fold. When one-byte codes are possible both specialisations are launched (the one the
// probe did not choose returns at once); the two-byte one with the LDS window the context asks for (szk_k1_params::wide16)
template <typename T, int NDIM, int TY>
static void launch_march(const void *d_in, uint16_t *codes, szk_k1_params &p, uint64_t nb, hipStream_t s) {
    if (p.wide16) launch_march_w<T, NDIM, TY, true>(d_in, codes, p, nb, s);
    else launch_march_w<T, NDIM, TY, false>(d_in, codes, p, nb, s);
}

template <typename T>
static int launch_k1(int ndim, const void *d_in, uint16_t *codes, szk_k1_params &p, hipStream_t s) {
    const uint64_t d0 = p.d[3], d1 = p.d[2], d2 = p.d[1], d3 = p.d[0];
    auto tiles = [&](uint64_t tx, uint64_t ty, uint64_t tz) {
        return ((d0 + tx - 1) / tx) * ((d1 + ty - 1) / ty) * ((d2 + tz - 1) / tz) * d3;
    };
    uint64_t nb;
    constexpr int FTZ = sizeof(T) == 4 ? 8 : 4;  // fast-path tile depth (LDS budget)
    // tuned kernel: quads need x % 4 == 0; 32-bit in-tile offsets need (TZ+1) planes < 2^31 elements; tile count < 2^31
    const bool fast = !szk_force_generic && (d0 % 4 == 0) && d0 < (1ull << 31) && d1 < (1ull << 31) && d2 < (1ull << 31) &&
                      d0 * d1 < (1ull << 27) && tiles(64, 8, FTZ) < (1ull << 31);
#ifndef LAB_MTY
#define LAB_MTY 4
#endif
    constexpr int MTY = LAB_MTY;
    const bool march = fast && !(szk_dbg_flags & 32) && d0 >= 128 && tiles(MARCH_TX, MTY, MARCH_TZ) < (1ull << 31) && (ndim == 3 || ndim == 4);
    // 1-D and 2-D arrays are 3-D arrays with d2 = 1 (and d1 = 1): the register-marching kernel needs no plane-size limit
    const bool march12 = !szk_force_generic && !(szk_dbg_flags & 32) && (ndim == 1 || ndim == 2) && d0 % 4 == 0 && d0 >= 128 &&
                         d0 < (1ull << 31) && d1 < (1ull << 31) && tiles(MARCH_TX, ndim == 1 ? 1 : MTY, MARCH_TZ) < (1ull << 31);
    if (!march && !march12) p.mode.allow = 0;
    if ((!march && !march12) || ndim > 3 || !p.mode.allow) p.samp.words = nullptr;  // (the sampled book is the one-byte marching forms')
    p.samp_in_launch = 0;
    p.fold_rows = 0;
    p.seg_expected = 0;
    // (same condition as launch_march_w's first branch: the one-launch form runs the probe itself)
    const bool one_launch = (march || march12) && p.mode.allow && !(szk_dbg_flags & 256) && p.hint_narrow > 0 && !(szk_dbg_flags & 131072);
    p.assumed_narrow = one_launch ? 1 : 0;
    // (the range words are kept by the one-launch form only: launch_march_w's first branch, same condition)
    p.range_kept = (march || march12) && p.range && p.mode.allow && !(szk_dbg_flags & 256) && p.hint_narrow > 0 && !(szk_dbg_flags & 131072) ? 1 : 0;
    if (!p.range_kept) p.range = nullptr;
    switch (ndim) {
        case 1:
            if (march12) {
                if (p.mode.allow && !one_launch) {
                    const uint64_t nsamp_threads = ((p.mode.n_total + SZK_PROBE_STRIDE - 1) / SZK_PROBE_STRIDE) * 64;
                    hipLaunchKernelGGL((k_probe<T, 3>), dim3((uint32_t)std::min<uint64_t>((nsamp_threads + 255) / 256, 1024)), dim3(256), 0, s, (const T *)d_in, p, p.mode.n_total, p.mode.probe_big);
                }
                nb = tiles(MARCH_TX, 1, MARCH_TZ);
                launch_march<T, 3, 1>(d_in, codes, p, nb, s);
                break;
            }
            nb = tiles(4096, 1, 1);
            if (nb > 0x7FFFFFFFull) return -1;
            hipLaunchKernelGGL((k_lorenzo_quant<T, 1, 4096, 1, 1>), dim3((uint32_t)nb), dim3(256), 0, s, (const T *)d_in, codes, p);
            break;
        case 2:
            if (march12) {
                if (p.mode.allow && !one_launch) {
                    const uint64_t nsamp_threads = ((p.mode.n_total + SZK_PROBE_STRIDE - 1) / SZK_PROBE_STRIDE) * 64;
                    hipLaunchKernelGGL((k_probe<T, 3>), dim3((uint32_t)std::min<uint64_t>((nsamp_threads + 255) / 256, 1024)), dim3(256), 0, s, (const T *)d_in, p, p.mode.n_total, p.mode.probe_big);
                }
                nb = tiles(MARCH_TX, MTY, MARCH_TZ);
                launch_march<T, 3, MTY>(d_in, codes, p, nb, s);
                break;
            }
            nb = tiles(128, 32, 1);
            if (nb > 0x7FFFFFFFull) return -1;
            hipLaunchKernelGGL((k_lorenzo_quant<T, 2, 128, 32, 1>), dim3((uint32_t)nb), dim3(256), 0, s, (const T *)d_in, codes, p);
            break;
        case 3:
            if (march) {
                if (p.mode.allow && !one_launch) {
                    const uint64_t nsamp_threads = ((p.mode.n_total + SZK_PROBE_STRIDE - 1) / SZK_PROBE_STRIDE) * 64;
                    hipLaunchKernelGGL((k_probe<T, 3>), dim3((uint32_t)std::min<uint64_t>((nsamp_threads + 255) / 256, 1024)), dim3(256), 0, s, (const T *)d_in, p, p.mode.n_total, p.mode.probe_big);
                }
                nb = tiles(MARCH_TX, MTY, MARCH_TZ);  // wave tasks
                launch_march<T, 3, MTY>(d_in, codes, p, nb, s);
                break;
            }
            if (fast) {
                nb = tiles(64, 8, FTZ);
                const uint32_t grid = k1_grid((const void *)k_lorenzo_quant_v4<T, 3, FTZ>, nb);
                hipLaunchKernelGGL((k_lorenzo_quant_v4<T, 3, FTZ>), dim3(grid), dim3(256), 0, s, (const T *)d_in, codes, p, (uint32_t)nb);
                hipLaunchKernelGGL(k_hist_reduce, dim3(HIST_WIN / 256, 64), dim3(256), 0, s, p.hist_partial, grid, (int)p.radius - HIST_WIN / 2, p.hist, (uint32_t *)nullptr);
                break;
            }
            nb = tiles(64, 8, 8);
            if (nb > 0x7FFFFFFFull) return -1;
            hipLaunchKernelGGL((k_lorenzo_quant<T, 3, 64, 8, 8>), dim3((uint32_t)nb), dim3(256), 0, s, (const T *)d_in, codes, p);
            break;
        default:
            if (march) {
                if (p.mode.allow && !one_launch) {
                    const uint64_t nsamp_threads = ((p.mode.n_total + SZK_PROBE_STRIDE - 1) / SZK_PROBE_STRIDE) * 64;
                    hipLaunchKernelGGL((k_probe<T, 4>), dim3((uint32_t)std::min<uint64_t>((nsamp_threads + 255) / 256, 1024)), dim3(256), 0, s, (const T *)d_in, p, p.mode.n_total, p.mode.probe_big);
                }
                nb = tiles(MARCH_TX, MTY, MARCH_TZ);  // wave tasks
                launch_march<T, 4, MTY>(d_in, codes, p, nb, s);
                break;
            }
            if (fast) {
                nb = tiles(64, 8, FTZ);
                const uint32_t grid = k1_grid((const void *)k_lorenzo_quant_v4<T, 4, FTZ>, nb);
                hipLaunchKernelGGL((k_lorenzo_quant_v4<T, 4, FTZ>), dim3(grid), dim3(256), 0, s, (const T *)d_in, codes, p, (uint32_t)nb);
                hipLaunchKernelGGL(k_hist_reduce, dim3(HIST_WIN / 256, 64), dim3(256), 0, s, p.hist_partial, grid, (int)p.radius - HIST_WIN / 2, p.hist, (uint32_t *)nullptr);
                break;
            }
            nb = tiles(64, 8, 4);
            if (nb > 0x7FFFFFFFull) return -1;
            hipLaunchKernelGGL((k_lorenzo_quant<T, 4, 64, 8, 4>), dim3((uint32_t)nb), dim3(256), 0, s, (const T *)d_in, codes, p);
            break;
    }
    SZK_CHECK_LAUNCH();
    return 0;
}
uint64_t szk_fuse_scratch_words(int ndim, const uint64_t d[4]) {  // (the tasks of launch_k1's marching forms: 256 x TY x MARCH_TZ, TY = 1 for 1-D arrays)
    if (ndim < 1 || ndim > 3 || d[3] % MARCH_TX != 0 || d[3] >= (1ull << 31) || d[2] >= (1ull << 31) || d[1] >= (1ull << 31)) return 0;
    const uint32_t ty = ndim == 1 ? 1u : (uint32_t)LAB_MTY;
    const uint64_t nb = (d[3] / MARCH_TX) * ((d[2] + ty - 1) / ty) * ((d[1] + MARCH_TZ - 1) / MARCH_TZ) * d[0];
    return nb * fuse_slot_words(ty, d[2], d[1]) + 2;
}
int szk_launch_k1(int dtype, int ndim, const void *d_in, uint16_t *codes, szk_k1_params *p, hipStream_t s) {
    szk_k1_params &pp = *p;  // mode.allow is cleared when the chosen kernel has no one-byte store path
    pp.dbg = (uint32_t)szk_dbg_flags;
    if (szk_dbg_flags & 64) pp.mode.allow = 0;
    return dtype == 0 ? launch_k1<float>(ndim, d_in, codes, pp, s) : launch_k1<double>(ndim, d_in, codes, pp, s);
}

int szk_launch_codebook(const uint64_t *d_hist, const szk_cb_params *p, hipStream_t s) {
    const uint32_t nb = p->n_books ? p->n_books : 1;  // > 1: batch of independent code books (tuner trials), no outlier sort
    szk_cb_params q = *p;
    q.n_books = nb;
    q.dbg = ((szk_dbg_flags & 1024) ? 1u : 0u) | ((szk_dbg_flags & 262144) ? 2u : 0u);  // (262144: the small path with the round-parallel merge)
    if (nb > 1) {  // (a single book's range words are zeroed by the caller together with its counters)
        hipError_t e = hipMemsetAsync(q.range, 0, 16 * nb, s);
        if (e != hipSuccess) return (int)e;
    }
    if (!p->range_ready) hipLaunchKernelGGL(k_hist_range, dim3(SZH_HIST_BINS / 256, nb), dim3(256), 0, s, d_hist, q.range);
    // which of the two forms applies is known on the device only; a context that remembers the previous call's alphabet
    // launches that form alone (solo): the kernel raises `mispredict` when it is the wrong one and the host repeats stage 2
    // (a single book that may be a wide one: its compaction over the whole chip first, sz3hip_debug_flags(1): inside the book's workgroup as before)
    q.keys_ready = nb == 1 && p->part_hint != 0 && !(szk_dbg_flags & 1) ? 1 : 0;
    if (q.keys_ready) hipLaunchKernelGGL(k_cb_compact, dim3(CBC_BLOCKS), dim3(1024), 0, s, d_hist, q.keys, q.syms, q.ifreq);
    if (p->part_hint != 1) hipLaunchKernelGGL(k_codebook<0>, dim3(nb == 1 ? 3 : nb), dim3(CB_LAUNCH), 0, s, d_hist, q);
    q.assign_later = q.keys_ready;  // (the same cases: a single book that may be a wide one)
    if (p->part_hint != 0) hipLaunchKernelGGL(k_codebook<1>, dim3(nb == 1 ? 3 : nb), dim3(CB_LAUNCH), 0, s, d_hist, q);
    if (p->part_hint != 0 && q.assign_later) hipLaunchKernelGGL(k_cb_assign, dim3(16), dim3(1024), 0, s, (const uint16_t *)q.depth, (const uint16_t *)q.syms, (const szk_cb_info *)q.info, q.enc);
    SZK_CHECK_LAUNCH();
    return 0;
}
__global__ __launch_bounds__(1024) void k_book_verdict(const szk_cb_info *__restrict__ fresh, const uint8_t *__restrict__ fresh_lens,
                                                       const szk_cb_info *__restrict__ used, const uint8_t *__restrict__ used_lens,
                                                       const uint32_t *__restrict__ mispredict, const uint32_t *__restrict__ range, szk_state *state,
                                                       const uint64_t *__restrict__ hist, int exact) {
    __shared__ uint32_t s_diff;
    __shared__ unsigned long long s_acc[3];
    if (threadIdx.x == 0) s_diff = 0;
    if (threadIdx.x < 3) s_acc[threadIdx.x] = 0;
    __syncthreads();
    const bool built = *mispredict == 0;
    if (built && book_rejected(hist, fresh, fresh_lens, used, used_lens, exact != 0, threadIdx.x, 1024u, s_acc) && threadIdx.x == 0) s_diff = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicOr(&state->miss_kind, (s_diff ? 1u : 0u) | (!built ? 2u : 0u));
        state->mispredict = built ? 0u : 1u;
        state->n_symbols = range[2];
    }
}
int szk_launch_book_verdict(const szk_cb_info *fresh, const uint8_t *fresh_lens, const szk_cb_info *used, const uint8_t *used_lens,
                            const uint32_t *mispredict, const uint32_t *range, szk_state *state, const uint64_t *hist, int exact, hipStream_t s) {
    hipLaunchKernelGGL(k_book_verdict, dim3(1), dim3(1024), 0, s, fresh, fresh_lens, used, used_lens, mispredict, range, state, hist, exact);
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_hist_range(const uint64_t *d_hist, uint32_t *range, hipStream_t s) {
    hipLaunchKernelGGL(k_hist_range, dim3(SZH_HIST_BINS / 256, 1), dim3(256), 0, s, d_hist, range);
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_hist_fold(const uint32_t *partial, uint32_t nrows, int radius, uint64_t *hist, uint32_t *range, hipStream_t s) {
    hipLaunchKernelGGL(k_hist_reduce, dim3(HIST_WIN / 256, 64), dim3(256), 0, s, partial, nrows, radius - HIST_WIN / 2, hist, range);
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_encode(const uint16_t *codes, uint64_t n, const uint32_t *d_enc, const szk_cb_info *info, int radius,
                      szk_mode mode, uint16_t *chunk_words, uint64_t *group_off, uint64_t *total_words,
                      const szk_state *state, uint8_t *payload, const szk_layout_params *layout, const szk_asm_params *asmp, hipStream_t s,
                      const uint16_t *seg_bits, const uint32_t *seg_made, const szk_encode_roles *er, const szk_merge_args *mg) {
    const uint64_t n_chunks = (n + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS;
    const uint64_t nb = (n_chunks + 3) / 4;
    if (nb > 0x7FFFFFFFull) return -1;
    const uint32_t sym_add = (uint32_t)radius - 127u;  // one-byte codes: stored byte t = delta + 127 (255: delta outlier, symbol 0) -> symbol t + sym_add
    const uint32_t pgrid = (uint32_t)(nb < 2048 ? nb : 2048);  // persistent: 8 workgroups per CU
    // (seg_bits: stage 1 summed the code bits per 256-element segment with the book the encoder uses: no bits pass)
    uint16_t *sub_bits = asmp ? asmp->sub_bits : nullptr;  // (the units' bit offsets: made with the chunks' word counts, copied into the payload with them)
    if (!seg_bits) hipLaunchKernelGGL(k_chunk_bits2, dim3(pgrid), dim3(256), 0, s, codes, n, d_enc, info, mode, sym_add, chunk_words, sub_bits);
    szk_fold_params fp{};
    if (er && er->fold_rows) {  // stage 1 left the fold of its histogram rows out: 64 more workgroups of this launch do it
        fp.partial = er->fold_partial;
        fp.nrows = er->fold_rows;
        fp.win_lo = radius - HIST_WIN / 2;
        fp.hist = er->fold_hist;
        fp.range = er->fold_range;
    }
    const uint16_t *scan_in = nullptr;
    if (seg_bits) {  // segments -> chunk word counts + group sums, all over the chip; the scan then walks one u32 per group
        const uint64_t n_segs = (n + 255) / 256;
        uint32_t *gsums = reinterpret_cast<uint32_t *>(group_off + (n_chunks + PACK_GROUP - 1) / PACK_GROUP + 1);  // behind the offsets (same array)
        hipLaunchKernelGGL(k_seg_chunks, dim3((uint32_t)std::min<uint64_t>(((n_segs + 7) / 8 + 255) / 256, 1024)), dim3(256), 0, s, seg_bits, n_segs, seg_made,
                           chunk_words, n_chunks, gsums, sub_bits);
        scan_in = reinterpret_cast<const uint16_t *>(gsums);
    }
    hipLaunchKernelGGL(k_scan_groups, dim3(fp.nrows ? 65 : 1), dim3(1024), 0, s, chunk_words, n_chunks, group_off, total_words, *layout, 1, scan_in,
                       (uint64_t)0, seg_made, fp);
    szk_role_params rp{};
    szk_asm_params apv = asmp ? *asmp : szk_asm_params{};
    if (er && er->roles && asmp) {
        rp.on = 1;
        rp.no_book = er->no_book ? 1u : 0u;
        rp.hist = er->hist;
        rp.cb = *er->cb;
        rp.used = info;
        rp.used_lens = er->used_lens;
        rp.flags = er->flags;
        rp.need_seg = seg_bits != nullptr;
        rp.exact = er->exact;
        apv.lists_by_roles = 1;
    }
    const uint32_t rb = rp.on ? ROLE_BLOCKS : 0u;
    // the packer's workgroups are persistent and split the chunks statically: all of them, the role and the assemble workgroups
    // must be resident together (5 per compute unit at 30 KB of LDS each), or the late ones double the launch's duration
    const uint32_t extra = rb + (asmp ? 32u : 0u);
    constexpr uint32_t ASM_BLOCKS = 32;
#ifdef SZ3HIP_LAB
    if (mg) {  // stage 1 was the fused form: the rows' bit strings only have to be moved to their places
        szk_merge_params mp;
        mp.slots = mg->slots;
        mp.seg_bits = seg_bits;
        mp.seg_base = mg->seg_base;
        mp.fuse_flag = mg->fuse_flag;
        const uint32_t pb = pgrid < 2048 - extra ? pgrid : 2048 - extra;
        hipLaunchKernelGGL(k_merge, dim3(rb + pb + (asmp ? ASM_BLOCKS : 0)), dim3(256), 0, s, mp, n, chunk_words, group_off, mode, state, payload, apv, pb, rp);
    } else
#else
    if (mg) return -2;  // (no fused stage 1 in this build: nothing hands a merge over)
#endif
    if (asmp && !(rp.on && !rp.no_book) && n_chunks >= 4096 && ((asmp->assumed_narrow && !(szk_dbg_flags & 32768)) || asmp->samp_words)) {
        // one-byte codes (stage 1's one-launch form assumed them; a probe that says otherwise voids the call) and no book built beside the
        // packer: the pair-table packer, one 1024-thread workgroup per CU (sz3hip_debug_flags(32768): k_pack as before)
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
            if (n_cu <= 0) n_cu = 256;
        }
        const uint32_t pb = (uint32_t)n_cu > rb + 1 ? (uint32_t)n_cu - rb : 1u;  // (the role workgroups take a compute unit each)
        const uint32_t split = 1;
        // (a call that may code with its sampled book is this kernel's whatever the debug flag says: k_pack's one-byte path takes code words
        // up to 16 bits. Behind the two-launch form of stage 1 — a context's first call — whether it does is known on the device only:
        // both packers are launched and the one whose case it is not returns at once)
        uint32_t beside = asmp->samp_words && !asmp->assumed_narrow ? 1u : 0u;
#ifdef SZ3HIP_LAB
        beside |= (szk_dbg_flags & 16) ? 8u : 0u;   // (lab switches of k_pack_b: see the kernel)
        beside |= (szk_dbg_flags & 512) ? 16u : 0u;
#endif
        hipLaunchKernelGGL(k_pack_b, dim3(rb + PB_ASM_BLOCKS + pb), dim3(PB_THREADS), 0, s, codes, n, d_enc, chunk_words, group_off, mode, sym_add, state,
                           payload, apv, pb, split, rp, beside);
        if (beside & 1u) {
            const uint32_t pb2 = pgrid < 1280 - extra ? pgrid : 1280 - extra;
            hipLaunchKernelGGL((k_pack<ENC_WIN>), dim3(rb + pb2 + (asmp ? ASM_BLOCKS : 0)), dim3(256), 0, s, codes, n, d_enc, info, chunk_words, group_off, mode,
                               sym_add, state, payload, apv, pb2, rp);
        }
    } else if (mode.pack_wide) {
        const uint32_t pb = pgrid < 768 - extra ? pgrid : 768 - extra;
        hipLaunchKernelGGL((k_pack<2 * ENC_WIN>), dim3(rb + pb + (asmp ? ASM_BLOCKS : 0)), dim3(256), 0, s, codes, n, d_enc, info, chunk_words, group_off, mode,
                           sym_add, state, payload, apv, pb, rp);
    } else {
        const uint32_t pb = pgrid < 1280 - extra ? pgrid : 1280 - extra;
        hipLaunchKernelGGL((k_pack<ENC_WIN>), dim3(rb + pb + (asmp ? ASM_BLOCKS : 0)), dim3(256), 0, s, codes, n, d_enc, info, chunk_words, group_off, mode,
                           sym_add, state, payload, apv, pb, rp);
    }
    SZK_CHECK_LAUNCH();
    return 0;
}
// Behind stage 2's last launch (round 5): the state block goes to the host's pinned copy and a sequence word after it — written by
// the device itself, system scope, instead of a device-to-host copy + event behind the launches (another engine, its own signals: ~20 us
// between the packer's end and the host's wake-up). finish() polls the word. The same launch zeroes the next call's histogram and
// counters when this call needs no repeat (state: no miss, no mispredicted book form) — the host draws the same conclusion from the
// same state — so that nothing is enqueued between a call's end and the next call's stage 1.
__global__ __launch_bounds__(256) void k_publish(const szk_state *__restrict__ state, uint32_t *host_state, uint32_t *host_seq, uint32_t seq,
                                                 uint4 *zero, uint32_t zero_vec16, const uint64_t *blk_others, uint4 *zero_blk0, uint4 *zero_blk1) {
    if (blockIdx.x == 0) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(state);
        constexpr uint32_t OW = offsetof(szk_state, blk_others) / 4;
        for (uint32_t i = threadIdx.x; i < sizeof(szk_state) / 4; i += 256) {
            uint32_t v = src[i];
            if (blk_others && (i == OW || i == OW + 1)) v = reinterpret_cast<const uint32_t *>(blk_others)[i - OW];
            __hip_atomic_store(&host_state[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // (round 6) the block predictor's counter blocks — 64 bytes of counters, 64 of Rice statistics + the selection's count, which the copy
        // above has just read — with the histogram: three fill launches off the front of a block-composed call (12 us of C1's 190)
        if (zero && zero_blk0 && state->miss_kind == 0 && state->mispredict == 0) {
            if (threadIdx.x < 4) zero_blk0[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
            else if (threadIdx.x < 9) zero_blk1[threadIdx.x - 4] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    if (zero && state->miss_kind == 0 && state->mispredict == 0)
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < zero_vec16; i += gridDim.x * 256) zero[i] = make_uint4(0u, 0u, 0u, 0u);
}
int szk_launch_publish(const szk_state *d_state, void *h_state, uint32_t *h_seq, uint32_t seq, void *d_zero, uint64_t zero_bytes, hipStream_t s,
                       const uint64_t *d_blk_others, void *d_zero_blk0, void *d_zero_blk1) {
    static_assert(sizeof(szk_state) % 4 == 0, "the state block is copied word by word");
    hipLaunchKernelGGL(k_publish, dim3(d_zero ? 128 : 1), dim3(256), 0, s, d_state, (uint32_t *)h_state, h_seq, seq, (uint4 *)d_zero, (uint32_t)(zero_bytes / 16),
                       d_blk_others, (uint4 *)d_zero_blk0, (uint4 *)d_zero_blk1);
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_assemble(const szk_asm_params *p, hipStream_t s) {
    hipLaunchKernelGGL(k_assemble, dim3(512), dim3(256), 0, s, *p);
    SZK_CHECK_LAUNCH();
    return 0;
}

int szk_launch_dec_tables(const uint8_t *d_lens, uint32_t sym_min, uint32_t sym_count, szk_dec_tables *t, uint32_t radius, uint32_t esc_sym, uint32_t *zero_word,
                          const uint16_t *chunk_words, uint64_t n_chunks, uint64_t *group_off, uint64_t *total_words, hipStream_t s) {
    hipLaunchKernelGGL(k_dec_tables, dim3(chunk_words ? 2 : 1), dim3(1024), 0, s, d_lens, sym_min, sym_count, t, radius, esc_sym, zero_word, chunk_words, n_chunks, group_off,
                       total_words);
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_decode(const uint8_t *payload, const szk_dec_params *p, uint16_t *codes, uint64_t *chunk_off,
                      uint64_t *total_words, hipStream_t s) {
    szk_layout_params no_layout{};
    (void)no_layout;  // (the group offsets are made by the second workgroup of the tables' launch: szk_launch_dec_tables)
    (void)chunk_off;
    (void)total_words;
    const uint64_t n_units = (p->n + SZH_UNIT_SYMS - 1) / SZH_UNIT_SYMS;
    const uint64_t nb = (n_units + 255) / 256;
    if (nb > 0x7FFFFFFFull) return -1;
    static const uint32_t pad = [] {  // (lab: extra LDS per workgroup = fewer resident workgroups; SZ3HIP_LAB_DEC_LDS bytes)
        const char *e = getenv("SZ3HIP_LAB_DEC_LDS");
        return e ? (uint32_t)atoi(e) : 0u;
    }();
    if (!p->scan_row) hipLaunchKernelGGL((k_decode<0>), dim3((uint32_t)nb), dim3(256), pad, s, payload, *p, codes);
    else if (p->q_bytes == 8 && p->half) hipLaunchKernelGGL((k_decode<8, true>), dim3((uint32_t)nb), dim3(256), pad, s, payload, *p, codes);
    else if (p->q_bytes == 8) hipLaunchKernelGGL((k_decode<8>), dim3((uint32_t)nb), dim3(256), pad, s, payload, *p, codes);
#ifdef SZ3HIP_LAB  // (the multi-symbol table form — round 5, slower than the one-symbol table — lab build only)
    else if (p->half && p->ms) hipLaunchKernelGGL((k_decode<4, true, true>), dim3((uint32_t)nb), dim3(256), pad, s, payload, *p, codes);
#endif
    else if (p->half) hipLaunchKernelGGL((k_decode<4, true>), dim3((uint32_t)nb), dim3(256), pad, s, payload, *p, codes);
    else hipLaunchKernelGGL((k_decode<4>), dim3((uint32_t)nb), dim3(256), pad, s, payload, *p, codes);
    if (p->scan_row && p->carry && p->carry_pass) {
        const uint64_t cb = (n_units + 3) / 4;
        if (p->q_bytes == 8 && p->half)
            hipLaunchKernelGGL((k_scan_carry_half<int32_t, int64_t>), dim3((uint32_t)cb), dim3(256), 0, s, (int32_t *)p->q_out, (const int64_t *)p->carry, p->n,
                               p->scan_row, p->ovf);
        else if (p->q_bytes == 8)
            hipLaunchKernelGGL(k_scan_carry<int64_t>, dim3((uint32_t)cb), dim3(256), 0, s, (int64_t *)p->q_out, (const int64_t *)p->carry, p->n, p->scan_row,
                               p->gate);
        else if (p->half)
            hipLaunchKernelGGL((k_scan_carry_half<int16_t, int32_t>), dim3((uint32_t)cb), dim3(256), 0, s, (int16_t *)p->q_out, (const int32_t *)p->carry, p->n,
                               p->scan_row, p->ovf);
        else
            hipLaunchKernelGGL(k_scan_carry<int32_t>, dim3((uint32_t)cb), dim3(256), 0, s, (int32_t *)p->q_out, (const int32_t *)p->carry, p->n, p->scan_row,
                               p->gate);
    }
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
static int launch_reconstruct(bool x_done, const uint8_t *payload, const szh_header &h, const szh_offsets &o, const uint16_t *codes,
                              void *d_out, void *d_segtot, hipStream_t s, const uint32_t *gate, const void *carry_v) {
    using Q = typename QTraits<T>::Q;
    const Q *carry = reinterpret_cast<const Q *>(carry_v);  // the decoder's unit carries, added by the first strided pass (nullptr: none)
    Q *q = reinterpret_cast<Q *>(d_out);
    const uint64_t n = h.n;
    const uint64_t L = h.dims[3], nrows = n / L;
    const bool wave_rows = L <= SCANX_SEG;  // one wave per row, no segment totals
    const uint32_t wgrid = grid_for(nrows, 4, 16384);
    if (x_done) {
        // the decoder wrote the x-scanned lattice values itself
    } else if (wave_rows && h.n_dout == 0) {
        // codes -> deltas -> x-scan in one pass
        hipLaunchKernelGGL((k_scan_x_wave<Q, true>), dim3(wgrid), dim3(256), 0, s, q, codes, (int)h.radius, L, nrows);
    } else {
        hipLaunchKernelGGL(k_expand_codes<Q>, dim3(grid_for(n, 256, 65536)), dim3(256), 0, s, codes, n, (int)h.radius, q);
        if (h.n_dout)
            hipLaunchKernelGGL(k_scatter_dout<Q>, dim3(grid_for(h.n_dout, 256, 4096)), dim3(256), 0, s, payload, o.dout_idx,
                               o.dout_val, h.n_dout, n, q);
        if (wave_rows) {
            hipLaunchKernelGGL((k_scan_x_wave<Q, false>), dim3(wgrid), dim3(256), 0, s, q, codes, (int)h.radius, L, nrows);
        } else {
            int rc = scan_rows<Q>(q, L, nrows, (Q *)d_segtot, s);
            if (rc) return rc;
        }
    }
    // strided axes y, z, w; the last one that exists also dequantises
    int last_ax = -1;
    for (int ax = 2; ax >= 0; ax--)
        if (h.dims[ax] > 1) last_ax = ax;
    uint64_t inner = L;
    for (int ax = 2; ax >= 0; ax--) {
        const uint64_t La = h.dims[ax];
        if (La > 1) {
            const uint64_t nlines = n / La;
            // few lines: cut them into segments (totals first) so that ~64K threads run
            uint32_t S = 1;
            if (nlines < 32768 && La >= 64) {
                S = (uint32_t)(65536 / nlines);
                if (S > La / 16) S = (uint32_t)(La / 16);
                if (S < 1) S = 1;
            }
            const uint64_t Lseg = (La + S - 1) / S;
            Q *totals = (Q *)d_segtot;
            if (S > 1)
                hipLaunchKernelGGL(k_strided_totals<Q>, dim3(grid_for(nlines * S, 256, 0x7FFFFFFF)), dim3(256), 0, s, (const Q *)q, La, inner, nlines, S,
                                   Lseg, totals, carry, (uint32_t)L);
            if (ax == last_ax)
                hipLaunchKernelGGL(k_scan_strided_dequant<T>, dim3(grid_for(nlines * S, 256, 0x7FFFFFFF)), dim3(256), 0, s, d_out, La, inner,
                                   nlines, S, Lseg, (const Q *)totals, szk_make_lattice(h.eb), gate, carry, (uint32_t)L);
            else
                hipLaunchKernelGGL(k_scan_strided<Q>, dim3(grid_for(nlines * S, 256, 0x7FFFFFFF)), dim3(256), 0, s, q, La, inner, nlines, S, Lseg,
                                   (const Q *)totals, gate, carry, (uint32_t)L);
            carry = nullptr;  // (the first pass took them)
        }
        inner *= La;
    }
    if (last_ax < 0)
        hipLaunchKernelGGL(k_dequant<T>, dim3(grid_for(n, 256, 65536)), dim3(256), 0, s, d_out, n, szk_make_lattice(h.eb), gate);
    if (h.n_vout)
        hipLaunchKernelGGL(k_patch_vout<T>, dim3(grid_for(h.n_vout, 256, 4096)), dim3(256), 0, s, payload, o.vout_idx,
                           o.vout_val, h.n_vout, n, (T *)d_out);
    SZK_CHECK_LAUNCH();
    return 0;
}
// the delta outliers alone, scattered to their code positions (a decoder that reads the codes itself looks here where a code is 0)
int szk_launch_scatter_deltas(int dtype, uint64_t n, const uint8_t *payload, const szh_offsets *o, uint64_t n_dout, void *d_out, hipStream_t s) {
    if (n_dout) {
        if (dtype == 0)
            hipLaunchKernelGGL(k_scatter_dout<int32_t>, dim3(grid_for(n_dout, 256, 4096)), dim3(256), 0, s, payload, o->dout_idx, o->dout_val, n_dout, n,
                               (int32_t *)d_out);
        else
            hipLaunchKernelGGL(k_scatter_dout<int64_t>, dim3(grid_for(n_dout, 256, 4096)), dim3(256), 0, s, payload, o->dout_idx, o->dout_val, n_dout, n,
                               (int64_t *)d_out);
    }
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_expand_deltas(int dtype, const uint16_t *codes, uint64_t n, int radius, const uint8_t *payload, const szh_offsets *o,
                             uint64_t n_dout, void *d_out, hipStream_t s) {
    if (dtype == 0) {
        hipLaunchKernelGGL(k_expand_codes<int32_t>, dim3(grid_for(n, 256, 65536)), dim3(256), 0, s, codes, n, radius, (int32_t *)d_out);
        if (n_dout)
            hipLaunchKernelGGL(k_scatter_dout<int32_t>, dim3(grid_for(n_dout, 256, 4096)), dim3(256), 0, s, payload, o->dout_idx, o->dout_val, n_dout, n,
                               (int32_t *)d_out);
    } else {
        hipLaunchKernelGGL(k_expand_codes<int64_t>, dim3(grid_for(n, 256, 65536)), dim3(256), 0, s, codes, n, radius, (int64_t *)d_out);
        if (n_dout)
            hipLaunchKernelGGL(k_scatter_dout<int64_t>, dim3(grid_for(n_dout, 256, 4096)), dim3(256), 0, s, payload, o->dout_idx, o->dout_val, n_dout, n,
                               (int64_t *)d_out);
    }
    SZK_CHECK_LAUNCH();
    return 0;
}
int szk_launch_reconstruct(int x_done, const uint8_t *payload, const szh_header *h, const szh_offsets *o, const uint16_t *codes,
                           void *d_out, void *d_segtot, hipStream_t s, const uint32_t *gate, const void *carry) {
    return h->dtype == 0 ? launch_reconstruct<float>(x_done != 0, payload, *h, *o, codes, d_out, d_segtot, s, gate, carry)
                         : launch_reconstruct<double>(x_done != 0, payload, *h, *o, codes, d_out, d_segtot, s, gate, carry);
}
// the shape gives every strided axis enough lines for one thread per line pair (no segment totals) and an even x extent
int szk_half_scans_ok(const szh_header *h) {
    if (h->dims[3] % 2) return 0;
    for (int ax = 2; ax >= 0; ax--)
        if (h->dims[ax] > 1 && h->n / h->dims[ax] < 32768) return 0;
    return 1;
}
int szk_launch_reconstruct_half(const uint8_t *payload, const szh_header *h, const szh_offsets *o, void *d_half_v, void *d_out, uint32_t *ovf,
                                hipStream_t s, const void *carry_v) {
    const uint64_t n = h->n;
    if (h->dtype != 0) {  // f64 data: int32 storage
        int32_t *d_half = reinterpret_cast<int32_t *>(d_half_v);
        const int64_t *carry = reinterpret_cast<const int64_t *>(carry_v);
        int last_ax = -1;
        for (int ax = 2; ax >= 0; ax--)
            if (h->dims[ax] > 1) last_ax = ax;
        uint64_t inner = h->dims[3];
        const szk_lattice lat = szk_make_lattice(h->eb);
        for (int ax = 2; ax >= 0; ax--) {
            const uint64_t La = h->dims[ax];
            if (La > 1) {
                const uint64_t npairs = n / La / 2;
                const dim3 g(grid_for(npairs, 256, 0x7FFFFFFF));
                const uint32_t row = (uint32_t)h->dims[3];
                if (ax == last_ax && carry) hipLaunchKernelGGL((k_scan_strided_half64<true, true>), g, dim3(256), 0, s, d_half, d_out, La, inner, n / La, lat, ovf, carry, row);
                else if (ax == last_ax) hipLaunchKernelGGL((k_scan_strided_half64<true, false>), g, dim3(256), 0, s, d_half, d_out, La, inner, n / La, lat, ovf, carry, row);
                else if (carry) hipLaunchKernelGGL((k_scan_strided_half64<false, true>), g, dim3(256), 0, s, d_half, (void *)d_half, La, inner, n / La, lat, ovf, carry, row);
                else hipLaunchKernelGGL((k_scan_strided_half64<false, false>), g, dim3(256), 0, s, d_half, (void *)d_half, La, inner, n / La, lat, ovf, carry, row);
                carry = nullptr;
            }
            inner *= La;
        }
        if (last_ax < 0) hipLaunchKernelGGL(k_dequant_half64, dim3(grid_for(n, 256, 65536)), dim3(256), 0, s, d_half, (double *)d_out, n, lat);
        SZK_CHECK_LAUNCH();
        return 0;
    }
    int16_t *d_half = reinterpret_cast<int16_t *>(d_half_v);
    const int32_t *carry = reinterpret_cast<const int32_t *>(carry_v);
    int last_ax = -1;
    for (int ax = 2; ax >= 0; ax--)
        if (h->dims[ax] > 1) last_ax = ax;
    uint64_t inner = h->dims[3];
    const szk_lattice lat = szk_make_lattice(h->eb);
    for (int ax = 2; ax >= 0; ax--) {
        const uint64_t La = h->dims[ax];
        if (La > 1) {
            const uint64_t npairs = n / La / 2;
            const dim3 g(grid_for(npairs, 256, 0x7FFFFFFF));
            const uint32_t row = (uint32_t)h->dims[3];
            if (ax == last_ax && carry) hipLaunchKernelGGL((k_scan_strided_half<true, true>), g, dim3(256), 0, s, d_half, d_out, La, inner, n / La, lat, ovf, carry, row);
            else if (ax == last_ax) hipLaunchKernelGGL((k_scan_strided_half<true, false>), g, dim3(256), 0, s, d_half, d_out, La, inner, n / La, lat, ovf, carry, row);
            else if (carry) hipLaunchKernelGGL((k_scan_strided_half<false, true>), g, dim3(256), 0, s, d_half, (void *)d_half, La, inner, n / La, lat, ovf, carry, row);
            else hipLaunchKernelGGL((k_scan_strided_half<false, false>), g, dim3(256), 0, s, d_half, (void *)d_half, La, inner, n / La, lat, ovf, carry, row);
            carry = nullptr;  // (the first pass took them)
        }
        inner *= La;
    }
    if (last_ax < 0) hipLaunchKernelGGL(k_dequant_half, dim3(grid_for(n, 256, 65536)), dim3(256), 0, s, d_half, (float *)d_out, n, lat);
    (void)payload;
    (void)o;
    SZK_CHECK_LAUNCH();
    return 0;
}

void szk_host_offsets(const szh_header *h, szh_offsets *o) { szh_compute_offsets(*h, *o); }
