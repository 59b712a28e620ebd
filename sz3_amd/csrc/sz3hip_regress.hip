// sz3_amd/csrc/sz3hip_regress.hip — the block-composed predictor path on gfx950: per-block choice among first-order
// Lorenzo, second-order Lorenzo and linear regression (SZH1 payload, predictor id 2; 3-D arrays, block edge 4..8).
//
// Reference (paths relative to /root/reference/include/SZ3):
//   predictor/RegressionPredictor.hpp:28-55    per-block fit: sum[i] = S idx_i*v, sum[N] = S v (double), coefficients in T
//   predictor/RegressionPredictor.hpp:77-92    predict = c0*i0 + c1*i1 + c2*i2 + c3 with block-local indices
//   predictor/RegressionPredictor.hpp:22-26    coefficient precision: linear terms eb/(N+1)/blockSize, constant eb/(N+1)
//   predictor/LorenzoPredictor.hpp:17-38,56-95 Lorenzo stencils (L = 1, 2) and the noise term of their error estimate
//   predictor/ComposedPredictor.hpp:25-40      selection: sampled S estimate_error per predictor, first minimum wins
//   utils/BlockwiseIterator.hpp:151-184        the sample points: the four diagonals (i,i,i) (i,i,j) (i,j,i) (i,j,j)
//   decomposition/BlockwiseDecomposition.hpp:28-67  block walk, fallback to Lorenzo-1 when the choice is not valid
//   quantizer/LinearQuantizer.hpp:43-86        quantize_and_overwrite / recover (regression residuals: used as is)
//
// What is re-designed. The reference walks the blocks one after the other and predicts from reconstructed values. Here
//   * Lorenzo blocks live on the lattice q = rint(x / 2eb) like the plain Lorenzo stream (sz3hip_kernels.hip): the code
//     is an exact integer stencil over q~, where q~ = q for Lorenzo elements and rint(x^ / 2eb) for the elements of
//     regression blocks (x^ = their reconstruction) — known to the encoder before any Lorenzo delta is formed, so the
//     whole encoder is two embarrassingly parallel block passes (k_blk_fit, k_blk_lorenzo), one wave per block, the block
//     and its low halo staged in LDS, the fit's four sums reduced across the wave in double.
//   * Regression blocks use the reference's own quantizer on x - pred (the point of regression — no differencing of the
//     noise — would be lost on the lattice); coefficients are snapped to their own lattice (dual quantisation again) and
//     delta-coded between consecutive regression blocks by a scan instead of the reference's running prev_coeffs.
//   * The decoder cannot be a global prefix sum any more (a regression block supplies values, not deltas): blocks are
//     solved in anti-diagonal wavefronts (bz + by + bx = const, one launch per front, one wave per block), every
//     Lorenzo block inverting its stencil from its low halo by three (six for L = 2) passes of line scans in LDS.
// Selection uses the reference's estimator with the reference's inputs: original values inside the block, reconstructed
// ones (here: the lattice reconstruction) outside it; the share of regression blocks is asserted against the oracle's.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "sz3hip_devutil.h"
#include "sz3hip_format.h"
#include "sz3hip_kernels.h"

#define BLK_MAXE 10                               // tile edge: block edge (<= 8) + 2 halo layers
#define BLK_TILE (BLK_MAXE * BLK_MAXE * BLK_MAXE)  // elements of a wave's tile
#define BLK_HWIN 4096   // LDS histogram window (bins around the radius) of a workgroup; BLK_HWIN_WIDE when the previous call of the
#define BLK_HWIN_WIDE 16384  // context saw an alphabet wider than the small one (C4-like fields: deltas of thousands of lattice steps)
#define BLK_GRID 2048u
// the encoder's persistent grids (round 6: 2048 -> 1024). Every workgroup of a coding kernel ends by adding its LDS histogram to the global
// one, one device-scope atomic per non-empty bin — a few hundred addresses that EVERY workgroup hits, performed one after another at the
// memory side (~20 ns each: 21 of 27 us of C1's stencil pass with 1024 workgroups). Half the workgroups, half the queue; C4a's slab runs
// the same to the microsecond (k_blk_fit 240, k_blk_rows 549 us), 512 would be 40 % slower there.
#define BLK_GRID_ENC 1024u

#define SZK_CHECK_LAUNCH()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return -1;      \
    } while (0)

namespace {

// wave-wide sum of doubles, the total in every lane. DPP lane moves (row_shr 1 2 4 8 inside the rows of 16 lanes, row_bcast 15 / 31
// across them: the last lane ends with the total), no LDS round trips: the butterfly of __shfl_xor it replaces is twelve
// ds_bpermute per sum, a dependent chain of ~1.5 k cycles — seven sums per block were 4 of the fit pass's 11 us per block.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov0_f64(double v) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));  // (lanes without a source read +0.0)
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_mov0_f64<0x111, 0xf>(v);
    v += dpp_mov0_f64<0x112, 0xf>(v);
    v += dpp_mov0_f64<0x114, 0xf>(v);
    v += dpp_mov0_f64<0x118, 0xf>(v);
    v += dpp_mov0_f64<0x142, 0xa>(v);
    v += dpp_mov0_f64<0x143, 0xc>(v);
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, WAVE - 1);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), WAVE - 1);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
template <typename T> __device__ __forceinline__ T lane_bcast(T v, int src);
template <> __device__ __forceinline__ float lane_bcast<float>(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
template <> __device__ __forceinline__ double lane_bcast<double>(double v, int src) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), src);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// lane l <- lane l - 1 across the wave (DPP wave_shr:1), lane 0 keeps `old`
__device__ __forceinline__ int32_t dpp_shr1_q(int32_t old, int32_t src) { return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int64_t dpp_shr1_q(int64_t old, int64_t src) {
    const int32_t lo = __builtin_amdgcn_update_dpp((int32_t)old, (int32_t)src, 0x138, 0xf, 0xf, false);
    const int32_t hi = __builtin_amdgcn_update_dpp((int32_t)(old >> 32), (int32_t)(src >> 32), 0x138, 0xf, 0xf, false);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// ... lane 0 reads zero (bound_ctrl)
// (the result passes through an empty asm: left to itself the compiler folds the lane move into the subtraction that follows —
// `v - dpp(v)` came out as `dpp(v) - v`, measured with a device printf: every x difference negated)
__device__ __forceinline__ int32_t dpp_shr1_z(int32_t src) {
    int32_t r = __builtin_amdgcn_update_dpp(0, src, 0x138, 0xf, 0xf, true);
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ int64_t dpp_shr1_z(int64_t src) {
    int32_t lo = __builtin_amdgcn_update_dpp(0, (int32_t)src, 0x138, 0xf, 0xf, true);
    int32_t hi = __builtin_amdgcn_update_dpp(0, (int32_t)(src >> 32), 0x138, 0xf, 0xf, true);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

struct BlkGeom {
    uint32_t bz, by, bx;
    uint32_t oz, oy, ox;  // origin of the block in the array
    uint32_t ez, ey, ex;  // extents (ragged at the high end)
    uint64_t coff;        // position of the block's first code: the codes are stored block by block (block raster order, elements in
                          // raster order inside the block) like the reference emits them (BlockwiseDecomposition.hpp:33-44) — a
                          // block's codes are one contiguous run, and the lossless stage finds 15 % more in that order
                          // (C2 field at 5e-2, regression: zstd of the Huffman stream 138 -> 118 KB)
};
__device__ __forceinline__ BlkGeom blk_geom_at(const szk_blk_params &p, uint32_t bz, uint32_t by, uint32_t bx) {
    BlkGeom g;
    g.bz = bz;
    g.by = by;
    g.bx = bx;
    g.oz = g.bz * p.B;
    g.oy = g.by * p.B;
    g.ox = g.bx * p.B;
    g.ez = min(p.B, (uint32_t)p.d[0] - g.oz);
    g.ey = min(p.B, (uint32_t)p.d[1] - g.oy);
    g.ex = min(p.B, (uint32_t)p.d[2] - g.ox);
    g.coff = (uint64_t)g.oz * p.d[1] * p.d[2] + (uint64_t)g.ez * ((uint64_t)g.oy * p.d[2] + (uint64_t)g.ey * g.ox);
    return g;
}
__device__ __forceinline__ BlkGeom blk_geom(const szk_blk_params &p, uint32_t task) {
    BlkGeom g;
    g.bx = task % p.nb[2];
    const uint32_t r = task / p.nb[2];
    g.by = r % p.nb[1];
    g.bz = r / p.nb[1];
    g.oz = g.bz * p.B;
    g.oy = g.by * p.B;
    g.ox = g.bx * p.B;
    g.ez = min(p.B, (uint32_t)p.d[0] - g.oz);
    g.ey = min(p.B, (uint32_t)p.d[1] - g.oy);
    g.ex = min(p.B, (uint32_t)p.d[2] - g.ox);
    // whole slabs of blocks below, whole rows of blocks in this slab, the blocks left of this one
    g.coff = (uint64_t)g.oz * p.d[1] * p.d[2] + (uint64_t)g.ez * ((uint64_t)g.oy * p.d[2] + (uint64_t)g.ey * g.ox);
    return g;
}
// t-th element of a block in raster order -> (i0, i1, i2). CB != 0: the kernel is compiled for that block edge — whole
// blocks (all but the array's high faces) then divide by constants (the run-time divisions were most of the block passes'
// instructions)
template <int CB>
__device__ __forceinline__ void own_index(const BlkGeom &g, uint32_t t, uint32_t &i0, uint32_t &i1, uint32_t &i2) {
    if (CB && g.ex == CB && g.ey == CB) {
        i2 = t % CB;
        i1 = (t / CB) % CB;
        i0 = t / (CB * CB);
    } else {
        i2 = t % g.ex;
        i1 = (t / g.ex) % g.ey;
        i0 = t / (g.ex * g.ey);
    }
}

// LDS histogram of a workgroup: BLK_HWIN bins around the radius, the rest straight to the global histogram; code 0
// (one address for the whole grid) is counted per wave
template <uint32_t HW>
__device__ __forceinline__ void blk_count(uint32_t *lh, const szk_blk_params &p, uint32_t code, bool active) {
    // (code 0 into the workgroup's own counter behind the window, lh[HW]: one global address for the whole grid took an atomic per wave
    // with a far delta — 5.8 of the code pass's 6.2 ms on a 1-D f64 series at 1e-6, same-address atomics run at ~90 per us)
    const unsigned long long zm = __ballot(active && code == 0);
    if (zm && lane_id() == __ffsll((long long)zm) - 1) atomicAdd(&lh[HW], (uint32_t)__popcll(zm));
    // the three codes around the radius are counted per wave (one lane adds the wave's number): at high ratios nearly every lane
    // of a wave has the SAME code, and 64 atomics on one LDS address are 64 serial ones (C4a: 2.1 of the element pass's 2.9 ms)
    bool mine = active && code != 0;
#pragma unroll
    for (int d = -1; d <= 1; d++) {
        const uint32_t c = p.radius + (uint32_t)d;
        const bool is = mine && code == c;
        const unsigned long long m = __ballot(is);
        if (m && lane_id() == __ffsll((long long)m) - 1) atomicAdd(&lh[c - (p.radius - HW / 2)], (uint32_t)__popcll(m));
        mine = mine && !is;
    }
    if (!mine) return;
    const uint32_t bin = code - (p.radius - HW / 2);
    if (bin < HW) atomicAdd(&lh[bin], 1u);
    else atomicAdd((unsigned long long *)&p.hist[code], 1ull);
}
template <uint32_t HW>
__device__ __forceinline__ void blk_flush(const uint32_t *lh, const szk_blk_params &p) {
#if defined(LAB_BLK) && (LAB_BLK & 1)  // (lab, wrong results: no flush of the workgroup's histogram)
    if (lh[0] == 0x12345678u) atomicAdd((unsigned long long *)&p.hist[1], 1ull);
    return;
#endif
    if (threadIdx.x == 0 && lh[HW]) atomicAdd((unsigned long long *)&p.hist[0], (unsigned long long)lh[HW]);
    for (uint32_t b = threadIdx.x; b < HW; b += blockDim.x) {
        const uint32_t v = lh[b];
        const uint32_t sym = p.radius - HW / 2 + b;
        if (v && sym < SZH_HIST_BINS) atomicAdd((unsigned long long *)&p.hist[sym], (unsigned long long)v);
    }
}
template <typename T>
__device__ __forceinline__ void blk_vout(const szk_blk_params &p, bool want, uint64_t gi, T raw) {
    const unsigned long long pos = wave_append_slot(want, p.n_vout);
    if (want && pos < p.out_cap) {
        p.vout_idx[pos] = gi;
        reinterpret_cast<T *>(p.vout_val)[pos] = raw;
    }
}

// Lorenzo stencil weights along one dimension: L = 1 -> (1, -1), L = 2 -> (1, -2, 1)
__device__ __forceinline__ int lz_w(int order, int j) { return order == 1 ? (j == 0 ? 1 : -1) : (j == 1 ? -2 : 1); }

// the reference's prediction from the ORIGINAL values of the tile (T arithmetic, the reference's term order):
// LorenzoPredictor.hpp:66-68 (L = 1) and :75-91 (L = 2, terms in lexicographic (k, j, i) order, coefficient -w(k)w(j)w(i))
// where a block sits in an LDS tile: index of tile coordinate (tz, ty, tx) of THE BLOCK's (B + 2)^3 neighbourhood (two low halo layers)
struct TileView {
    uint32_t pz, py, o;  // plane / row pitch of the tile, offset of the neighbourhood's origin
};
__device__ __forceinline__ uint32_t tv_at(const TileView &v, uint32_t tz, uint32_t ty, uint32_t tx) { return v.o + tz * v.pz + ty * v.py + tx; }
// rd(tz, ty, tx): the value the estimate sees at that tile coordinate
template <typename T, typename RD>
__device__ __forceinline__ T lorenzo_pred_orig(const RD &rd, uint32_t tz, uint32_t ty, uint32_t tx, int order) {
    if (order == 1) {
        return rd(tz, ty, tx - 1) + rd(tz, ty - 1, tx) + rd(tz - 1, ty, tx) - rd(tz, ty - 1, tx - 1) - rd(tz - 1, ty, tx - 1) - rd(tz - 1, ty - 1, tx) +
               rd(tz - 1, ty - 1, tx - 1);
    }
    T acc = 0;
    bool first = true;
    for (int k = 0; k <= 2; k++)
        for (int j = 0; j <= 2; j++)
            for (int i = 0; i <= 2; i++) {
                if ((k | j | i) == 0) continue;
                const int c = -(lz_w(2, k) * lz_w(2, j) * lz_w(2, i));
                const T term = (T)c * rd(tz - k, ty - j, tx - i);
                acc = first ? term : acc + term;
                first = false;
            }
    return acc;
}

// coefficient lattices (RegressionPredictor.hpp:22-26: quantizer_liner eb/(N+1)/block_size, quantizer_independent eb/(N+1))
struct CoefLat {
    double step_lin, step_ind;  // 2 * eb of the two quantizers
};
__device__ __host__ __forceinline__ CoefLat coef_lat(double eb, uint32_t B, uint32_t ndim) {
    CoefLat c;
    c.step_ind = 2.0 * (eb / (double)(ndim + 1));
    c.step_lin = 2.0 * (eb / (double)(ndim + 1) / (double)B);
    return c;
}
template <typename T>
__device__ __forceinline__ void coef_recover(const int64_t *lc, const CoefLat &cl, T (&rc)[4]) {
    for (int i = 0; i < 3; i++) rc[i] = (T)((double)lc[i] * cl.step_lin);
    rc[3] = (T)((double)lc[3] * cl.step_ind);
}
template <typename T>
__device__ __forceinline__ T reg_predict(const T (&c)[4], uint32_t i0, uint32_t i1, uint32_t i2) {  // RegressionPredictor.hpp:82-84
    return c[0] * (T)i0 + c[1] * (T)i1 + c[2] * (T)i2 + c[3];
}

// ------------------------------------------------------------------------------------------------------------
// encoder pass 1: fit, select, regression blocks coded; q~ of every element written to qwork
// ------------------------------------------------------------------------------------------------------------
// one block of the fit pass, by one wave, from an LDS tile that holds the block's originals and two low halo layers (zero outside
// the array, like the reference's padding). The selection's error estimates see what the reference's see (ComposedPredictor.hpp:
// 29-33 on a block whose predecessors are already compressed): ORIGINAL values inside the block, RECONSTRUCTED ones outside it —
// here the lattice reconstruction the decoder will hold for a Lorenzo neighbour. HALO_RAW: the tile holds originals everywhere
// (it is shared by several blocks) and the halo is put on the lattice when read; else the loader already did.
template <typename T, uint32_t HW, int CB, bool HALO_RAW>
__device__ __forceinline__ void blk_fit_block(const T *sx, const TileView &tv, const BlkGeom &g, uint32_t task, int lane, uint32_t *lh,
                                              const szk_blk_params &p, uint16_t *__restrict__ codes) {
    using Q = typename QTraits<T>::Q;
    const Lattice<T> lat(p.lat);
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    Q *qwork = reinterpret_cast<Q *>(p.qwork);
    const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
    const double eb_recip = 1.0 / p.eb;
    const bool has_l1 = p.mask & 1u, has_l2 = p.mask & 2u, has_r = p.mask & 4u;
    auto rd = [&](uint32_t tz, uint32_t ty, uint32_t tx) -> T {
        T v = sx[tv_at(tv, tz, ty, tx)];
        if (HALO_RAW && (tz < 2 || ty < 2 || tx < 2)) {
            bool bad;
            const Q qh = lat.quant(v, bad);
            if (!bad) v = lat.dequant(qh);
        }
        return v;
    };
    const uint32_t nown = g.ez * g.ey * g.ex;
    // The choice and the coefficients come from the selection pass (k_blk_select: the reference's summation order, a block per
    // lane) unless it did not run (p.sel_given == 0: this pass's own fit and estimates below, sums reduced across the wave).
    int sid = 0;
    int64_t lc[4] = {0, 0, 0, 0};
    if (p.sel_given) {
        sid = p.sel[task];
        if (sid == 2)
            for (int i = 0; i < 4; i++) lc[i] = p.coef[(uint64_t)task * 4 + i];
    } else {
    // ---- regression fit (RegressionPredictor.hpp:28-55) ----
    bool r_valid = has_r && g.ez > 1 && g.ey > 1 && g.ex > 1;
    T cf[4] = {0, 0, 0, 0};
    if (r_valid) {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (uint32_t t = lane; t < nown; t += WAVE) {
            uint32_t i0, i1, i2;
        own_index<CB>(g, t, i0, i1, i2);
            const T v = sx[tv_at(tv, i0 + 2, i1 + 2, i2 + 2)];
            s0 += (double)((T)i0 * v);  // sum[i] += index[i] * (*c): size_t * T is evaluated in T, accumulated in double
            s1 += (double)((T)i1 * v);
            s2 += (double)((T)i2 * v);
            s3 += (double)v;
        }
        s0 = wave_sum_f64(s0);
        s1 = wave_sum_f64(s1);
        s2 = wave_sum_f64(s2);
        s3 = wave_sum_f64(s3);
        // the three slopes have the same expression (RegressionPredictor.hpp:43-52): lanes 0..2 evaluate one each (a double
        // division is ~40 instructions, ten of them were a fifth of the pass), then everybody takes the results
        const double dz = g.ez, dy = g.ey, dx = g.ex, num = dz * dy * dx;
        const double sk = lane == 0 ? s0 : (lane == 1 ? s1 : s2), dk = lane == 0 ? dz : (lane == 1 ? dy : dx);
        const T ck = (T)((2 * sk / (dk - 1) - s3) * 6 / num / (dk + 1));
        cf[0] = lane_bcast<T>(ck, 0);
        cf[1] = lane_bcast<T>(ck, 1);
        cf[2] = lane_bcast<T>(ck, 2);
        cf[3] = (T)(s3 / num);
        cf[3] = (T)((double)cf[3] - (dz - 1) * (double)cf[0] / 2);
        cf[3] = (T)((double)cf[3] - (dy - 1) * (double)cf[1] / 2);
        cf[3] = (T)((double)cf[3] - (dx - 1) * (double)cf[2] / 2);
    }
    // ---- selection (ComposedPredictor.hpp:25-40 over foreach_sampling) ----
    sid = has_l1 ? 0 : (has_l2 ? 1 : 2);
    const int npred = (int)has_l1 + (int)has_l2 + (int)has_r;
    if (npred > 1) {
        const uint32_t m = min(g.ez, min(g.ey, g.ex));
        double e1 = 0, e2 = 0, er = 0;
        if ((uint32_t)lane < 4 * m) {
            const uint32_t i = (uint32_t)lane / 4, kind = (uint32_t)lane % 4, j = m - 1 - i;
            const uint32_t i0 = i, i1 = (kind & 2) ? j : i, i2 = (kind & 1) ? j : i;
            const uint32_t tz = i0 + 2, ty = i1 + 2, tx = i2 + 2;
            const T v = sx[tv_at(tv, tz, ty, tx)];
            if (has_l1) e1 = (double)(T)(fabs((double)(T)(v - lorenzo_pred_orig<T>(rd, tz, ty, tx, 1))) + (T)(1.22 * p.eb));
            if (has_l2) e2 = (double)(T)(fabs((double)(T)(v - lorenzo_pred_orig<T>(rd, tz, ty, tx, 2))) + (T)(6.8 * p.eb));
            if (r_valid) er = (double)(T)fabs((double)(T)(v - reg_predict(cf, i0, i1, i2)));
        }
        e1 = wave_sum_f64(e1);
        e2 = wave_sum_f64(e2);
        er = wave_sum_f64(er);
        double best = 1.7976931348623157e308;
        sid = -1;
        if (has_l1) { best = e1; sid = 0; }
        if (has_l2 && (sid < 0 || e2 < best)) { best = e2; sid = 1; }
        if (has_r && r_valid && (sid < 0 || er < best)) { best = er; sid = 2; }
        if (sid < 0) sid = 0;  // (regression the only candidate and not valid: the fallback predictor, Lorenzo-1)
    } else if (sid == 2 && !r_valid) {
        sid = 0;  // BlockwiseDecomposition.hpp:35-37
    }
    // ---- regression: coefficients onto their lattices; a coefficient the lattice cannot hold -> Lorenzo-1 ----
    if (sid == 2) {
        bool ok = true;
        for (int i = 0; i < 4; i++) {
            const double s = (double)cf[i] / (i < 3 ? cl.step_lin : cl.step_ind);
            if (!(fabs(s) < 4503599627370496.0)) ok = false;
            else lc[i] = (int64_t)rint(s);
        }
        if (!ok) sid = 0;
    }
    }  // (own selection)
    if (sid == 2) {
        T rc[4];
        coef_recover(lc, cl, rc);
        for (uint32_t t0 = 0; t0 < nown; t0 += WAVE) {
            const uint32_t t = t0 + lane;
            const bool act = t < nown;
            const uint32_t tt = act ? t : 0;
            uint32_t i0, i1, i2;
        own_index<CB>(g, tt, i0, i1, i2);
            const uint64_t gi = ((uint64_t)(g.oz + i0) * d1 + (g.oy + i1)) * d2 + (g.ox + i2);
            const T raw = sx[tv_at(tv, i0 + 2, i1 + 2, i2 + 2)];
            T v = raw;
            const int code = act ? ref_quantize(v, reg_predict(rc, i0, i1, i2), p.eb, eb_recip, (int)p.radius) : 1;
            Q qt = 0;
            if (code != 0) {
                bool bad;
                qt = lat.quant(v, bad);
                if (bad) qt = 0;
            }
            if (act) {
                codes[g.coff + t] = (uint16_t)code;
                qwork[gi] = qt;
            }
            blk_count<HW>(lh, p, (uint32_t)code, act);
            blk_vout<T>(p, act && code == 0, gi, raw);  // unpredictable: the raw value, LinearQuantizer.hpp:66-69
        }
        if (lane == 0) {
            for (int i = 0; i < 4; i++) p.coef[(uint64_t)task * 4 + i] = lc[i];
            // (no count here: the rank pass counts the regression blocks; one same-address atomic per block was serial work for the L2)
        }
    } else {
        for (uint32_t t0 = 0; t0 < nown; t0 += WAVE) {
            const uint32_t t = t0 + lane;
            const bool act = t < nown;
            const uint32_t tt = act ? t : 0;
            uint32_t i0, i1, i2;
        own_index<CB>(g, tt, i0, i1, i2);
            const uint64_t gi = ((uint64_t)(g.oz + i0) * d1 + (g.oy + i1)) * d2 + (g.ox + i2);
            const T raw = sx[tv_at(tv, i0 + 2, i1 + 2, i2 + 2)];
            bool bad;
            Q q = lat.quant(raw, bad);
            if (bad) q = 0;
            if (act) qwork[gi] = q;
            blk_vout<T>(p, act && bad, gi, raw);
        }
    }
    if (lane == 0) p.sel[task] = (uint8_t)sid;
}

// ------------------------------------------------------------------------------------------------------------
// the selection alone, a block per LANE: how many blocks would not be coded by first-order Lorenzo
// ------------------------------------------------------------------------------------------------------------
// On fields where regression (and Lorenzo-2) never win — C4b: 23 of 636 056 blocks in the reference, 21 of 643 302 here — the
// block-composed stream is the plain Lorenzo stream plus 160 KB of selection bits, made by two block passes at a third of the
// plain kernel's speed and decoded front by front. This pass answers the question first: the fit, the sampled estimates and
// the choice of every block, nothing else. A lane owns a block and walks it row by row in the reference's order
// (RegressionPredictor.hpp:28-55: the sums in raster order, in double; ComposedPredictor.hpp:25-40 over
// BlockwiseIterator.hpp:151-184: the sample points i = 0.., four diagonals each, summed as they come) — no tile in LDS, no
// cross-lane sum (the wave-per-block fit spends ~800 wave instructions per block, most of them on both), 64 blocks' rows of
// B values side by side in memory = whole cache lines per wave. Values outside the block are seen as the decoder will hold
// them (on the lattice), outside the array as zero, like k_blk_fit's tile loader does it.
template <typename T, int CB, bool L2>  // L2: the predictor set holds second-order Lorenzo (its 26-point estimate costs the kernel a third of its waves)
__global__ __launch_bounds__(256) void k_blk_select(const T *__restrict__ in, szk_blk_params p, uint32_t nblocks, unsigned long long *__restrict__ n_other) {
    using Q = typename QTraits<T>::Q;
    const uint32_t task = blockIdx.x * 256 + threadIdx.x;
    bool other = false;
    if (task < nblocks) {
        const BlkGeom g = blk_geom(p, task);
        const Lattice<T> lat(p.lat);
        const uint64_t d1 = p.d[1], d2 = p.d[2];
        const bool has_l1 = p.mask & 1u, has_l2 = L2 && (p.mask & 2u), has_r = p.mask & 4u;
        const bool whole = CB && g.ez == CB && g.ey == CB && g.ex == CB;
        using V2 = typename std::conditional<sizeof(T) == 8, double2, float2>::type;
        const bool pairs = d2 % 2 == 0 && (reinterpret_cast<uintptr_t>(in) % sizeof(V2)) == 0;
        const T *blk = in + ((uint64_t)g.oz * d1 + g.oy) * d2 + g.ox;
        // (tile coordinates of k_blk_fit: the block's origin is (2, 2, 2), two low halo layers)
        auto rd = [&](uint32_t tz, uint32_t ty, uint32_t tx) -> T {
            // (no branch around the load or the lattice round trip: a conditional load is waited for where it stands, and the
            // four sample points of a step have 28 - 104 of them)
            const int64_t z = (int64_t)g.oz + tz - 2, y = (int64_t)g.oy + ty - 2, x = (int64_t)g.ox + tx - 2;
            const bool in_arr = z >= 0 && y >= 0 && x >= 0;  // (never beyond the block's high faces)
            const T raw = in[((uint64_t)(z < 0 ? 0 : z) * d1 + (uint64_t)(y < 0 ? 0 : y)) * d2 + (uint64_t)(x < 0 ? 0 : x)];
            const T v = in_arr ? raw : (T)0;
            bool bad;
            const Q qh = lat.quant(v, bad);
            const T vl = bad ? v : lat.dequant(qh);
            return (tz < 2 || ty < 2 || tx < 2) ? vl : v;
        };
        bool r_valid = has_r && g.ez > 1 && g.ey > 1 && g.ex > 1;
        T cf[4] = {0, 0, 0, 0};
        if (r_valid) {
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            if (whole) {
                // (a plane at a time: 427 us at C4's slab. Measured and dropped: rows one after the other with the next row's loads
                // in flight, 92 instead of 115 registers: 466; the sample points' neighbours below as four pair loads (x - 1, x)
                // instead of eight single ones: 447)
#pragma unroll 1
                for (uint32_t i0 = 0; i0 < (uint32_t)CB; i0++) {
#pragma unroll
                    for (uint32_t i1 = 0; i1 < (uint32_t)CB; i1++) {
                        const T *row = blk + ((uint64_t)i0 * d1 + i1) * d2;
                        T v[CB ? CB : 1];
                        if (CB % 2 == 0 && pairs) {  // (rows of an even edge start on pair boundaries when the array's rows do: half the loads)
                            const V2 *r2 = reinterpret_cast<const V2 *>(row);
#pragma unroll
                            for (uint32_t i2 = 0; i2 < (uint32_t)CB / 2; i2++) {
                                const V2 w = r2[i2];
                                v[2 * i2] = w.x;
                                v[(2 * i2 + 1) % (CB ? CB : 1)] = w.y;
                            }
                        } else {
#pragma unroll
                            for (uint32_t i2 = 0; i2 < (uint32_t)CB; i2++) v[i2] = row[i2];
                        }
#pragma unroll
                        for (uint32_t i2 = 0; i2 < (uint32_t)CB; i2++) {
                            s0 += (double)((T)i0 * v[i2]);  // sum[i] += index[i] * (*c): size_t * T is evaluated in T, accumulated in double
                            s1 += (double)((T)i1 * v[i2]);
                            s2 += (double)((T)i2 * v[i2]);
                            s3 += (double)v[i2];
                        }
                    }
                }
            } else {
                for (uint32_t i0 = 0; i0 < g.ez; i0++)
                    for (uint32_t i1 = 0; i1 < g.ey; i1++) {
                        const T *row = blk + ((uint64_t)i0 * d1 + i1) * d2;
                        for (uint32_t i2 = 0; i2 < g.ex; i2++) {
                            const T v = row[i2];
                            s0 += (double)((T)i0 * v);
                            s1 += (double)((T)i1 * v);
                            s2 += (double)((T)i2 * v);
                            s3 += (double)v;
                        }
                    }
            }
            const double dz = g.ez, dy = g.ey, dx = g.ex, num = dz * dy * dx;
            cf[0] = (T)((2 * s0 / (dz - 1) - s3) * 6 / num / (dz + 1));
            cf[1] = (T)((2 * s1 / (dy - 1) - s3) * 6 / num / (dy + 1));
            cf[2] = (T)((2 * s2 / (dx - 1) - s3) * 6 / num / (dx + 1));
            cf[3] = (T)(s3 / num);
            cf[3] = (T)((double)cf[3] - (dz - 1) * (double)cf[0] / 2);
            cf[3] = (T)((double)cf[3] - (dy - 1) * (double)cf[1] / 2);
            cf[3] = (T)((double)cf[3] - (dx - 1) * (double)cf[2] / 2);
        }
        int sid = has_l1 ? 0 : (has_l2 ? 1 : 2);
        const int npred = (int)has_l1 + (int)has_l2 + (int)has_r;
        if (npred > 1) {
            const uint32_t m = min(g.ez, min(g.ey, g.ex));
            double e1 = 0, e2 = 0, er = 0;
#pragma unroll 1
            for (uint32_t i = 0; i < m; i++) {
                const uint32_t j = m - 1 - i;
#pragma unroll
                for (uint32_t kind = 0; kind < 4; kind++) {  // (unrolled: the four points' neighbour loads are in flight together)
                    const uint32_t i0 = i, i1 = (kind & 2) ? j : i, i2 = (kind & 1) ? j : i;
                    const uint32_t tz = i0 + 2, ty = i1 + 2, tx = i2 + 2;
                    const T v = blk[((uint64_t)i0 * d1 + i1) * d2 + i2];
                    if (has_l1) e1 += (double)(T)(fabs((double)(T)(v - lorenzo_pred_orig<T>(rd, tz, ty, tx, 1))) + (T)(1.22 * p.eb));
                    if (L2 && has_l2) e2 += (double)(T)(fabs((double)(T)(v - lorenzo_pred_orig<T>(rd, tz, ty, tx, 2))) + (T)(6.8 * p.eb));
                    if (r_valid) er += (double)(T)fabs((double)(T)(v - reg_predict(cf, i0, i1, i2)));
                }
            }
            double best = 1.7976931348623157e308;
            sid = -1;
            if (has_l1) { best = e1; sid = 0; }
            if (has_l2 && (sid < 0 || e2 < best)) { best = e2; sid = 1; }
            if (has_r && r_valid && (sid < 0 || er < best)) { best = er; sid = 2; }
            if (sid < 0) sid = 0;
        } else if (sid == 2 && !r_valid) {
            sid = 0;
        }
        if (sid == 2) {  // the coefficients onto their lattices; one its lattice cannot hold: Lorenzo-1
            const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
            int64_t lc[4];
            for (int i = 0; i < 4; i++) {
                const double sc = (double)cf[i] / (i < 3 ? cl.step_lin : cl.step_ind);
                if (!(fabs(sc) < 4503599627370496.0)) sid = 0;
                lc[i] = (int64_t)rint(sc);
            }
            if (sid == 2)
                for (int i = 0; i < 4; i++) p.coef[(uint64_t)task * 4 + i] = lc[i];
        }
        p.sel[task] = (uint8_t)sid;  // (k_blk_fit takes the choice and the coefficients from here)
        other = sid != 0;
    }
    const unsigned long long mo = __ballot(other);
    if (mo && lane_id() == 0) atomicAdd(n_other, (unsigned long long)__popcll(mo));
}

// The lattice values of everything that is not in a regression block: q~ = rint(x / 2eb), element by element (the choices are known
// since the selection pass) — what the fit pass used to do block by block from its tiles for the 86 % of C4a's blocks that only
// needed this. A thread per element, rows walked by the workgroups; the block of an element is (z / B, y / B, x / B).
template <typename T>
__global__ __launch_bounds__(256) void k_blk_lattice(const T *__restrict__ in, szk_blk_params p, uint64_t nrows) {
    using Q = typename QTraits<T>::Q;
    const Lattice<T> lat(p.lat);
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    Q *qwork = reinterpret_cast<Q *>(p.qwork);
    const uint32_t x = blockIdx.x * 256 + threadIdx.x;
    const bool xok = x < d2;
    const uint32_t bx = (xok ? x : 0u) / p.B;
    for (uint64_t row = blockIdx.y; row < nrows; row += gridDim.y) {
        const uint32_t z = (uint32_t)(row / d1), y = (uint32_t)(row % d1);
        const uint32_t task = ((z / p.B) * p.nb[1] + y / p.B) * p.nb[2] + bx;
        const uint64_t gi = row * d2 + x;
        bool bad = false;
        T raw = 0;
        if (xok && p.sel[task] != 2) {
            raw = in[gi];
            const Q q = lat.quant(raw, bad);
            qwork[gi] = bad ? (Q)0 : q;
        }
        blk_vout<T>(p, bad, gi, raw);  // (unpredictable: the raw value; the append is a wave operation, all lanes take part)
    }
}

// NW: waves (= blocks in flight) per workgroup; they share the LDS histogram, so the wide form (64 KB of bins) takes 16 of them to
// keep four waves per SIMD busy (with 4 the two passes ran at two waves per SIMD, bound by the latency of their tile loads).
// Measured and dropped (round 2, C4's slab): groups of 2 x 2 x 4 blocks sharing one tile per workgroup (1.47 x instead of 2.37 x
// the volume read, rows of 26 values): fit 1.91 against 1.76 ms, Lorenzo pass 1.15 against 1.04 — the barriers around the shared
// load cost more than the smaller read saves; a thread per ELEMENT for the Lorenzo pass (no idle lanes, block bookkeeping in an
// LDS table): 1.08 ms — not bound by its instructions, then; by its 8-byte lattice values (1.07 GB written by the fit, read back
// with halo) — or so it seemed: storing q~ in 4 bytes (behind gated full-width passes for values that do not fit) changed nothing either (fit 1.75, Lorenzo pass 1.00 ms). Round 3, ablations of the fit pass at C4's slab (tools/blk_lab.sh): 1682 us as is;
// 1593 without the lattice stores, 1558 without the selection, 1244 without the fit and the regression blocks; the seven wave
// sums on DPP instead of ds_bpermute butterflies 1780 -> 1704. No single piece dominates: ~800 wave instructions per block.
template <typename T, uint32_t HW, int CB, int NW>
__global__ __launch_bounds__(NW * 64) void k_blk_fit(const T *__restrict__ in, uint16_t *__restrict__ codes, szk_blk_params p, uint32_t nblocks,
                                                     const uint32_t *__restrict__ comp = nullptr, const uint64_t *__restrict__ n_reg = nullptr) {
    using Q = typename QTraits<T>::Q;
    __shared__ T s_x[NW][CB ? (CB + 2) * (CB + 2) * (CB + 2) : BLK_TILE];
    __shared__ uint32_t lh[HW + 1];  // (+ the count of code 0: blk_count)
    for (uint32_t b = threadIdx.x; b <= HW; b += NW * 64) lh[b] = 0;
    __syncthreads();
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    const uint32_t wv = threadIdx.x / WAVE;
    T *sx = s_x[wv];
    const uint32_t B = CB ? (uint32_t)CB : p.B, E = B + 2;
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    const TileView tv{E * E, E, 0};
    // With the list of the regression blocks at hand (comp, made by the rank pass in front of this launch: round 5) a wave walks that list —
    // walking all blocks and skipping 86 % of them at C4a was a dependent one-byte load per skipped block, 135 of them per wave for 22 blocks of work.
    const uint32_t n_items = comp ? (uint32_t)*n_reg : nblocks;
    for (uint32_t item = blockIdx.x * NW + wv; item < n_items; item += gridDim.x * NW) {
        const uint32_t task = comp ? comp[item] : item;
        if (!comp && p.sel_given && p.sel[task] != 2) continue;  // (not a regression block: k_blk_lattice wrote its lattice values)
        const BlkGeom g = blk_geom(p, task);
        if (p.sel_given) {
            // a regression block coded from the selection pass's coefficients: its own elements are all it reads (no estimates, no halo)
            const uint32_t nown = g.ez * g.ey * g.ex;
            for (uint32_t t = lane; t < nown; t += WAVE) {
                uint32_t i0, i1, i2;
                own_index<CB>(g, t, i0, i1, i2);
                sx[tv_at(tv, i0 + 2, i1 + 2, i2 + 2)] = in[((uint64_t)(g.oz + i0) * d1 + (g.oy + i1)) * d2 + (g.ox + i2)];
            }
        } else
        // ---- originals of the block and two low halo layers, the halo on the lattice ----
        for (uint32_t t = lane; t < E * E * E; t += WAVE) {
            const uint32_t tx = t % E, ty = (t / E) % E, tz = t / (E * E);
            const int64_t z = (int64_t)g.oz + tz - 2, y = (int64_t)g.oy + ty - 2, x = (int64_t)g.ox + tx - 2;
            T v = 0;
            if (z >= 0 && y >= 0 && x >= 0 && z < (int64_t)p.d[0] && y < (int64_t)d1 && x < (int64_t)d2) v = in[((uint64_t)z * d1 + (uint64_t)y) * d2 + (uint64_t)x];
            if (tz < 2 || ty < 2 || tx < 2) {
                bool bad;
                const Q qh = lat.quant(v, bad);
                if (!bad) v = lat.dequant(qh);
            }
            sx[t] = v;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's LDS writes are visible to its own reads
        blk_fit_block<T, HW, CB, false>(sx, tv, g, task, lane, lh, p, codes);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    blk_flush<HW>(lh, p);
}

// ------------------------------------------------------------------------------------------------------------
// encoder pass 2: Lorenzo blocks — integer stencil over q~ (block + two low halo layers in LDS)
// ------------------------------------------------------------------------------------------------------------
// one Lorenzo block of the second pass, by one wave, from an LDS tile of q~ (the block and its two low halo layers)
template <typename T, uint32_t HW, int CB>
__device__ __forceinline__ void blk_lorenzo_block(const typename QTraits<T>::Q *sq, const TileView &tv, const BlkGeom &g, int order, int lane, uint32_t *lh,
                                                  const szk_blk_params &p, uint16_t *__restrict__ codes) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    const uint32_t nown = g.ez * g.ey * g.ex;
    for (uint32_t t0 = 0; t0 < nown; t0 += WAVE) {
        const uint32_t t = t0 + lane;
        const bool act = t < nown;
        const uint32_t tt = act ? t : 0;
        uint32_t i0, i1, i2;
        own_index<CB>(g, tt, i0, i1, i2);
        UQ delta = 0;  // wrap-around arithmetic like the plain Lorenzo stream
        if (order == 1) {  // (wave-uniform; the stencils unrolled with their constant weights)
#pragma unroll
            for (int k = 0; k <= 1; k++)
#pragma unroll
                for (int j = 0; j <= 1; j++)
#pragma unroll
                    for (int i = 0; i <= 1; i++) {
                        const UQ v = (UQ)sq[tv_at(tv, i0 + 2 - k, i1 + 2 - j, i2 + 2 - i)];
                        delta = ((k + j + i) & 1) ? delta - v : delta + v;
                    }
        } else {
#pragma unroll
            for (int k = 0; k <= 2; k++)
#pragma unroll
                for (int j = 0; j <= 2; j++)
#pragma unroll
                    for (int i = 0; i <= 2; i++) {
                        const int w = lz_w(2, k) * lz_w(2, j) * lz_w(2, i);
                        delta += (UQ)((Q)w * sq[tv_at(tv, i0 + 2 - k, i1 + 2 - j, i2 + 2 - i)]);
                    }
        }
        const bool inr = (UQ)(delta + (UQ)(p.radius - 1)) <= (UQ)(2 * p.radius - 2);
        const uint32_t code = inr ? (uint32_t)(delta + (UQ)p.radius) : 0u;
        if (act) codes[g.coff + t] = (uint16_t)code;
        blk_count<HW>(lh, p, code, act);
        const unsigned long long pd = wave_append_slot(act && !inr, p.n_dout);
        if (act && !inr && pd < p.out_cap) {
            p.dout_idx[pd] = g.coff + t;  // (position of the code, not of the element: the decoder expands the codes in place)
            reinterpret_cast<Q *>(p.dout_val)[pd] = (Q)delta;
        }
    }
}

template <typename T, uint32_t HW, int CB, int NW>
__global__ __launch_bounds__(NW * 64) void k_blk_lorenzo(uint16_t *__restrict__ codes, szk_blk_params p, uint32_t nblocks) {
    using Q = typename QTraits<T>::Q;
    __shared__ Q s_q[NW][CB ? (CB + 2) * (CB + 2) * (CB + 2) : BLK_TILE];
    __shared__ uint32_t lh[HW + 1];  // (+ the count of code 0: blk_count)
    for (uint32_t b = threadIdx.x; b <= HW; b += NW * 64) lh[b] = 0;
    __syncthreads();
    const int lane = lane_id();
    const uint32_t wv = threadIdx.x / WAVE;
    Q *sq = s_q[wv];
    const uint32_t B = CB ? (uint32_t)CB : p.B, E = B + 2;
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    const Q *qwork = reinterpret_cast<const Q *>(p.qwork);
    const TileView tv{E * E, E, 0};
    for (uint32_t task = blockIdx.x * NW + wv; task < nblocks; task += gridDim.x * NW) {
        const int sid = p.sel[task];
        if (sid == 2) continue;
        const int order = sid == 1 ? 2 : 1;
        const BlkGeom g = blk_geom(p, task);
        for (uint32_t t = lane; t < E * E * E; t += WAVE) {
            const uint32_t tx = t % E, ty = (t / E) % E, tz = t / (E * E);
            const int64_t z = (int64_t)g.oz + tz - 2, y = (int64_t)g.oy + ty - 2, x = (int64_t)g.ox + tx - 2;
            Q v = 0;
            if (z >= 0 && y >= 0 && x >= 0 && z < (int64_t)p.d[0] && y < (int64_t)d1 && x < (int64_t)d2) v = qwork[((uint64_t)z * d1 + (uint64_t)y) * d2 + (uint64_t)x];
            sq[t] = v;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        blk_lorenzo_block<T, HW, CB>(sq, tv, g, order, lane, lh, p, codes);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    blk_flush<HW>(lh, p);
}
// The same pass by ROWS OF BLOCKS (round 3, taken when the selection pass ran and the set has no second-order Lorenzo): a Lorenzo
// element's code is a stencil over q~ of itself and its lower neighbours wherever they lie, and q~ of an element is a function of
// that element alone — rint(x / 2eb) outside regression blocks, the lattice value k_blk_fit stored inside them. So nothing is
// staged per block and nothing is written for the 86 % of C4a's blocks that are Lorenzo blocks: a workgroup takes a run of
// blocks along x of one block row (bz, by) — a thread per x — and marches through the run's planes and rows (one halo plane, one
// halo row per plane: 49 row loads for 36 rows of codes at B = 6, every one coalesced, every value put on the lattice once),
// keeps the previous plane's and row's lattice values in registers, takes the left neighbour from the previous lane (DPP; a
// wave's first lane fetches its own), and collects the codes of the run in LDS: the blocks of a run are neighbours in the
// block-major code order, so they leave as ONE contiguous stretch. (A thread per element storing each code where it belongs —
// 12 bytes here, 12 bytes there, every cache line of codes touched 36 times — took 2.2 - 2.9 ms at C4's slab: the stores, not
// the loads or the histogram.) k_blk_lattice + k_blk_lorenzo, which this replaces: 0.51 + 0.98 ms.
template <typename T, uint32_t HW, int CB>
__global__ __launch_bounds__(256) void k_blk_rows(const T *__restrict__ in, uint16_t *__restrict__ codes, szk_blk_params p, uint32_t ntasks, uint32_t xchunks) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    constexpr int MJ = CB ? CB + 1 : 9;  // rows of a plane the march keeps: the halo row + the block's
    __shared__ uint32_t lh[HW + 1];  // (+ the count of code 0: blk_count)
    __shared__ uint16_t s_codes[256 * (CB ? CB * CB : 64)];
    __shared__ uint8_t s_reg[256];
    for (uint32_t b = threadIdx.x; b <= HW; b += 256) lh[b] = 0;
    __syncthreads();
    const Lattice<T> lat(p.lat);
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    const uint32_t B = CB ? (uint32_t)CB : p.B, nb1 = p.nb[1], nb2 = p.nb[2];
    const Q *qwork = reinterpret_cast<const Q *>(p.qwork);
    // A wave's first lane is a HALO lane: it holds the column left of the wave's 63 and codes nothing — the x difference of
    // every column then comes from the neighbouring lane (DPP), no column is loaded twice, none behind a branch. 4 x 63 = 252
    // columns per workgroup = 42 blocks of 6. (PMC of the first form, which loaded and quantised a left value per lane and
    // counted, listed and indexed per element with wave operations: 163 vector + 160 scalar instructions per element, 1.15 ms.)
    const uint32_t TPB = (252u / B) * B, BPC = 252u / B;  // columns coded per run, blocks per run
    const int lane = lane_id();
    const uint32_t wv = threadIdx.x / WAVE;
    const int col = (int)(wv * 63u) + lane - 1;  // column inside the run (-1: the run's own left halo)
    const bool coder = lane > 0 && (uint32_t)col < TPB;
    const uint32_t cu = coder ? (uint32_t)col : 0u, bl = cu / B, i2 = cu - bl * B;
    const uint32_t peak_bin = HW / 2;  // bin of the code `radius` in the LDS window
    for (uint32_t task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const uint32_t xci = task % xchunks, r = task / xchunks, by = r % nb1, bz = r / nb1;  // (workgroup-uniform)
        const uint32_t oz = bz * B, oy = by * B, ez = min(B, (uint32_t)p.d[0] - oz), ey = min(B, (uint32_t)d1 - oy);
        const int64_t xs = (int64_t)xci * TPB + col;  // this lane's column of the array (-1 left of it, >= d2 right of it: zero)
        const bool xin = xs >= 0 && xs < (int64_t)d2;
        const bool xok = coder && xs < (int64_t)d2;   // codes an element
        const uint32_t xc = xin ? (uint32_t)xs : 0u;  // (a lane outside the array reads column 0: a valid address, its value unused)
        const uint32_t bx = xc / B, ox = bx * B, ex = min(B, (uint32_t)d2 - ox);
        // the predictor of the four blocks this lane's values come from: its own, the one above (by - 1), behind (bz - 1), both
        const uint32_t tk = (bz * nb1 + by) * nb2;
        // (four unconditional loads, all in flight — behind their conditions each was waited for where it stood; a block that does not
        // exist reads the lane's own block's byte)
        const uint32_t up = by ? nb2 : 0u, back = bz ? nb1 * nb2 : 0u;
        const uint8_t s_own = p.sel[tk + bx], s_up = p.sel[tk - up + bx], s_back = p.sel[tk - back + bx], s_bu = p.sel[tk - back - up + bx];
        const bool reg_own = xin && s_own == 2, reg_up = xin && by && s_up == 2, reg_back = xin && bz && s_back == 2,
                   reg_bu = xin && by && bz && s_bu == 2;
        const bool any_reg = __ballot(reg_own || reg_up || reg_back || reg_bu) != 0;
        const bool act = xok && !reg_own;
        if (threadIdx.x < BPC) s_reg[threadIdx.x] = 0;
        __syncthreads();
        if (xok && i2 == 0 && reg_own) s_reg[bl] = 1;
        // row offsets and validity: per task, not per plane
        uint64_t rowoff[MJ];
        bool rowin[MJ];
#pragma unroll
        for (int j = 0; j < MJ; j++) {
            rowin[j] = (uint32_t)j <= ey && (j > 0 || by > 0);
            rowoff[j] = (uint64_t)(rowin[j] ? oy + (uint32_t)j - 1 : 0u) * d2;
        }
        const uint32_t per = ez * ey * B;
        const uint32_t li0 = bl * per + i2;  // position of the block's column inside the run's stretch of codes
        uint32_t n_peak = 0, n_zero = 0;     // this lane's codes equal to the radius / zero (counted per lane, added per task)
        UQ PD[MJ];  // the previous plane: x differences of q~, rows oy - 1 ..
#pragma unroll
        for (int j = 0; j < MJ; j++) PD[j] = 0;
        for (uint32_t kz = 0; kz <= ez; kz++) {  // planes oz - 1 .. oz + ez - 1
            const bool zin = kz > 0 || bz > 0;    // (the plane exists: below the array's first plane everything is zero)
            const uint64_t pb = (uint64_t)(zin ? oz + kz - 1 : 0u) * d1 * d2;
            T raw[MJ];
#pragma unroll
            for (int j = 0; j < MJ; j++) raw[j] = in[pb + rowoff[j] + xc];  // the plane's rows oy - 1 .. oy + ey - 1: every load first
            // (round 5) ... the lattice values the fit pass stored for regression blocks among them, lanes of such blocks only: asked for where
            // they were used, behind a ballot each, they were 49 exposed memory latencies per task — 375 of the pass's 780 us at C4a's slab
            UQ qreg[MJ];
            bool isreg[MJ];
#pragma unroll
            for (int j = 0; j < MJ; j++) {
                isreg[j] = any_reg && zin && rowin[j] && (kz == 0 ? (j == 0 ? reg_bu : reg_back) : (j == 0 ? reg_up : reg_own));
                qreg[j] = 0;
            }
            if (any_reg) {  // (wave-uniform; every lane loads — the lanes of other blocks the array's first word, one cache line per wave:
                            // loads behind per-lane conditions came out of the compiler with a wait for everything in flight after each)
#pragma unroll
                for (int j = 0; j < MJ; j++) qreg[j] = (UQ)qwork[isreg[j] ? pb + rowoff[j] + xc : 0];
            }
            UQ D[MJ];
            uint32_t badmask = 0, outmask = 0;
            UQ deltas[MJ];
#pragma unroll
            for (int j = 0; j < MJ; j++) {
                const bool rin = zin && rowin[j] && xin;
                bool bad;
                const Q q = lat.quant(raw[j], bad);
                UQ v = bad ? (UQ)0 : (UQ)q;
                v = isreg[j] ? qreg[j] : v;  // (a regression block's elements)
                v = rin ? v : (UQ)0;
                const UQ vleft = (UQ)dpp_shr1_z((Q)v);  // the previous lane holds the column on the left (the halo lane: nothing, zero)
                D[j] = v - vleft;
                deltas[j] = 0;
                if (kz > 0 && j > 0 && (uint32_t)j <= ey) {  // an element of the run: plane kz - 1, row j - 1 of its block
                    const UQ delta = D[j] - D[j - 1] - PD[j] + PD[j - 1];  // wrap-around arithmetic like the plain Lorenzo stream
                    const bool inr = (UQ)(delta + (UQ)(p.radius - 1)) <= (UQ)(2 * p.radius - 2);
                    const uint32_t code = inr ? (uint32_t)(delta + (UQ)p.radius) : 0u;
                    const uint32_t li = li0 + ((kz - 1) * ey + ((uint32_t)j - 1)) * ex;
                    if (act) {
                        s_codes[li] = (uint16_t)code;
                        const bool is_peak = code == p.radius;
                        n_peak += is_peak ? 1u : 0u;
                        n_zero += code == 0u ? 1u : 0u;
#if defined(LAB_ROWS) && (LAB_ROWS & 1)  // (lab, wrong results: no histogram)
                        if (code == 0xFFFFFFu) {
#else
                        if (!is_peak && code != 0u) {  // (the peak is counted per lane: 64 lanes on one LDS address are 64 serial atomics)
#endif
                            const uint32_t bin = code - (p.radius - HW / 2);
                            if (bin < HW) atomicAdd(&lh[bin], 1u);
                            else atomicAdd((unsigned long long *)&p.hist[code], 1ull);
                        }
                        badmask |= bad ? (1u << j) : 0u;
                        outmask |= inr ? 0u : (1u << j);
                    }
                    deltas[j] = delta;
                }
            }
            // the rare elements: unpredictable values (their raw value is listed), deltas beyond the radius (listed with the code's position)
            if (__ballot((badmask | outmask) != 0)) {
#pragma unroll
                for (int j = 1; j < MJ; j++) {
                    const bool wb = (badmask >> j) & 1u, wo = (outmask >> j) & 1u;
                    if (!__ballot(wb || wo)) continue;
                    const uint64_t gi = pb + rowoff[j] + xc;
                    blk_vout<T>(p, wb, gi, raw[j]);
                    const unsigned long long pd = wave_append_slot(wo, p.n_dout);
                    if (wo && pd < p.out_cap) {
                        const uint64_t base = (uint64_t)oz * d1 * d2 + (uint64_t)ez * ((uint64_t)oy * d2 + (uint64_t)ey * (xci * TPB));
                        p.dout_idx[pd] = base + li0 + ((kz - 1) * ey + ((uint32_t)j - 1)) * ex;  // (position of the code, not of the element)
                        reinterpret_cast<Q *>(p.dout_val)[pd] = (Q)deltas[j];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < MJ; j++) PD[j] = D[j];
        }
        // the per-lane counts of the two special codes
        {
            const uint32_t wp = wave_sum(n_peak), wz = wave_sum(n_zero);
            if (lane == 0) {
                if (wp) atomicAdd(&lh[peak_bin], wp);
                if (wz) atomicAdd((unsigned long long *)&p.hist[0], (unsigned long long)wz);
            }
        }
        __syncthreads();
        // the run's codes leave as one stretch (the blocks of a row of blocks follow each other in the code order); a regression
        // block's codes are the fit pass's
        {
            const uint32_t x0 = xci * TPB, xend = min((uint32_t)d2, x0 + TPB);
            const uint32_t total = ez * ey * (xend - x0);
            const uint64_t base = (uint64_t)oz * d1 * d2 + (uint64_t)ez * ((uint64_t)oy * d2 + (uint64_t)ey * x0);
#if defined(LAB_ROWS) && (LAB_ROWS & 2)  // (lab, wrong results: no code stores)
            for (uint32_t i = threadIdx.x; i < total; i += 256 * 4096)
#else
            for (uint32_t i = threadIdx.x; i < total; i += 256)
#endif
                if (!s_reg[i / per]) codes[base + i] = s_codes[i];
        }
        __syncthreads();
    }
    blk_flush<HW>(lh, p);
}
// ------------------------------------------------------------------------------------------------------------
// side information of a block stream:
//   [u32 coding = 1][u32 sel_bits = 2][u64 n_blocks][u64 n_reg]
//   [selection: 2 bits per block, padded to 8 bytes]
//   [u8 k[4]][u32 n_groups]                      Rice parameters of the four coefficients, groups of 64 regression blocks
//   [u32 bit offset of every group][bits: u32 words, MSB first]
// A regression block's four coefficient lattice values are coded as differences to the previous regression block's (block
// raster order — the chain the reference's prev_coeffs follows, RegressionPredictor.hpp:148-156, but made of lattice
// integers so that it is a scan): zigzag, then Rice with the coefficient's own parameter; quotients >= 24 escape to the
// 64 raw bits. Groups of 64 blocks start at recorded bit offsets: the decoder parses the groups in parallel.
// (The reference Huffman-codes these values with a tree of their own, RegressionPredictor.hpp:94-107; raw 16-bit
// differences were 8 bytes per block = 2.4 % of the C4a stream, Rice codes are ~3.)
// ------------------------------------------------------------------------------------------------------------
#define SIDE_HDR 24u
#define RICE_ESC 24u
#define RICE_GROUP 64u
__device__ __forceinline__ uint64_t side_sel_bytes(uint64_t nblocks) { return ((nblocks + 3) / 4 + 7) & ~7ull; }
__device__ __forceinline__ uint64_t zigzag(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }
__device__ __forceinline__ int64_t unzigzag(uint64_t u) { return (int64_t)(u >> 1) ^ -(int64_t)(u & 1); }
__device__ __forceinline__ uint32_t rice_len(uint64_t u, uint32_t k) {
    const uint64_t q = u >> k;
    return q < RICE_ESC ? (uint32_t)q + 1u + k : RICE_ESC + 64u;
}
// rank of every block among the regression blocks (exclusive), the compacted list, the count — three small launches: the
// regression blocks of every run of 8192 blocks are counted, one workgroup turns the counts into offsets, every run is walked
// again with its offset (one workgroup over all flags took 0.38 ms for C4's 643 302 blocks)
#define RANK_RUN 8192u
__global__ __launch_bounds__(256) void k_blk_rank_count(const uint8_t *__restrict__ sel, uint32_t nblocks, uint32_t *__restrict__ run_cnt) {
    __shared__ uint32_t s_c;
    if (threadIdx.x == 0) s_c = 0;
    __syncthreads();
    const uint32_t b0 = blockIdx.x * RANK_RUN;
    uint32_t c = 0;
    for (uint32_t b = b0 + threadIdx.x; b < b0 + RANK_RUN && b < nblocks; b += 256) c += sel[b] == 2;
    c = wave_sum(c);
    if (lane_id() == 0 && c) atomicAdd(&s_c, c);
    __syncthreads();
    if (threadIdx.x == 0) run_cnt[blockIdx.x] = s_c;
}
__global__ __launch_bounds__(1024) void k_blk_rank_offsets(uint32_t *__restrict__ run_cnt, uint32_t nruns, uint64_t *n_reg_out) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nruns; base += 1024) {
        const uint32_t r = base + threadIdx.x;
        const uint32_t mine = r < nruns ? run_cnt[r] : 0u;
        const uint32_t incl = wave_incl_scan(mine);
        if (lane_id() == WAVE - 1) s_w[threadIdx.x / WAVE] = incl;
        __syncthreads();
        uint32_t run = s_carry + incl - mine, tot = 0;
        for (uint32_t w = 0; w < 16; w++) {
            if (w < threadIdx.x / WAVE) run += s_w[w];
            tot += s_w[w];
        }
        if (r < nruns) run_cnt[r] = run;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_reg_out = s_carry;
}
__global__ __launch_bounds__(256) void k_blk_rank_write(const uint8_t *__restrict__ sel, uint32_t nblocks, const uint32_t *__restrict__ run_off,
                                                        uint32_t *__restrict__ rank, uint32_t *__restrict__ comp) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = run_off[blockIdx.x];
    __syncthreads();
    constexpr uint32_t PER = 8;
    const uint32_t r0 = blockIdx.x * RANK_RUN;
    for (uint32_t base = r0; base < r0 + RANK_RUN && base < nblocks; base += 256 * PER) {
        const uint32_t b0 = base + threadIdx.x * PER;
        uint32_t f[PER], mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            f[k] = (b0 + k < nblocks && sel[b0 + k] == 2) ? 1u : 0u;
            mine += f[k];
        }
        const uint32_t incl = wave_incl_scan(mine);
        if (lane_id() == WAVE - 1) s_w[threadIdx.x / WAVE] = incl;
        __syncthreads();
        uint32_t run = s_carry + incl - mine, tot = 0;
        for (uint32_t w = 0; w < 4; w++) {
            if (w < threadIdx.x / WAVE) run += s_w[w];
            tot += s_w[w];
        }
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            if (b0 + k < nblocks) {
                rank[b0 + k] = run;
                if (f[k] && comp) comp[run] = b0 + k;
            }
            run += f[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
}
// up to RANK_SMALL blocks (round 6): the three launches in one workgroup (two launches and their 9 us off a 4 MB array's step)
#define RANK_SMALL 16384u
__global__ __launch_bounds__(1024) void k_blk_rank_small(const uint8_t *__restrict__ sel, uint32_t nblocks, uint32_t *__restrict__ rank, uint32_t *__restrict__ comp,
                                                         uint64_t *n_reg_out) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    constexpr uint32_t PER = 8;
    for (uint32_t base = 0; base < nblocks; base += 1024 * PER) {
        const uint32_t b0 = base + threadIdx.x * PER;
        uint32_t f[PER], mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            f[k] = (b0 + k < nblocks && sel[b0 + k] == 2) ? 1u : 0u;
            mine += f[k];
        }
        const uint32_t incl = wave_incl_scan(mine);
        if (lane_id() == WAVE - 1) s_w[threadIdx.x / WAVE] = incl;
        __syncthreads();
        uint32_t run = s_carry + incl - mine, tot = 0;
        for (uint32_t w = 0; w < 16; w++) {
            if (w < threadIdx.x / WAVE) run += s_w[w];
            tot += s_w[w];
        }
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            if (b0 + k < nblocks) {
                rank[b0 + k] = run;
                if (f[k] && comp) comp[run] = b0 + k;
            }
            run += f[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_reg_out = s_carry;
}
static void launch_blk_rank(const uint8_t *sel, uint32_t nblocks, uint32_t *rank, uint32_t *comp, uint32_t *run_scratch, uint64_t *n_reg, hipStream_t s) {
    if (nblocks <= RANK_SMALL && !(szk_dbg_flags & 2048)) {  // (debug flag 2048: the three launches whatever the block count)
        hipLaunchKernelGGL(k_blk_rank_small, dim3(1), dim3(1024), 0, s, sel, nblocks, rank, comp, n_reg);
        return;
    }
    const uint32_t nruns = (nblocks + RANK_RUN - 1) / RANK_RUN;
    hipLaunchKernelGGL(k_blk_rank_count, dim3(nruns), dim3(256), 0, s, sel, nblocks, run_scratch);
    hipLaunchKernelGGL(k_blk_rank_offsets, dim3(1), dim3(1024), 0, s, run_scratch, nruns, n_reg);
    hipLaunchKernelGGL(k_blk_rank_write, dim3(nruns), dim3(256), 0, s, sel, nblocks, (const uint32_t *)run_scratch, rank, comp);
}
// NC: coefficients of a regression block — N + 1: four for the arrays the kernels see in three dimensions (1-D and 2-D arrays leave the
// first ones zero), five for 4-D arrays (round 4). The parameter block behind the selection bits: NC Rice parameters, the number of
// groups in its last four bytes — 8 bytes for NC = 4 (the layout of rounds 2 and 3), 16 for NC = 5.
template <int NC> __device__ __host__ constexpr uint32_t side_par_bytes() { return NC <= 4 ? 8u : 16u; }
template <int NC>
__device__ __forceinline__ void coef_delta(const int64_t *__restrict__ coef, const uint32_t *__restrict__ comp, uint64_t r, uint64_t (&u)[NC]) {
    const int64_t *cur = coef + (uint64_t)comp[r] * NC;
    const int64_t *prv = r ? coef + (uint64_t)comp[r - 1] * NC : nullptr;
    for (int i = 0; i < NC; i++) u[i] = zigzag(cur[i] - (prv ? prv[i] : 0));
}
// sum of the zigzagged differences per coefficient -> its Rice parameter (stats[0..3]; doubles: no overflow worries)
template <int NC>
__global__ __launch_bounds__(256) void k_blk_coef_stats(const int64_t *__restrict__ coef, const uint32_t *__restrict__ comp, const uint64_t *n_reg,
                                                        double *stats) {
    const uint64_t nr = *n_reg;
    double s[NC];
    for (int i = 0; i < NC; i++) s[i] = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < nr; r += (uint64_t)gridDim.x * 256) {
        uint64_t u[NC];
        coef_delta<NC>(coef, comp, r, u);
        for (int i = 0; i < NC; i++) s[i] += (double)u[i];
    }
    // (32 workgroups: 4096 waves adding to the same NC words were ~40 of this launch's 52 us at C4a's 90 000 regression blocks; the sums are
    // of integers below 2^53: exact, whatever the order)
    for (int i = 0; i < NC; i++) {
        s[i] = wave_sum_f64(s[i]);
        if (lane_id() == 0 && s[i] != 0) atomicAdd(&stats[i], s[i]);
    }
}
__device__ __forceinline__ uint32_t rice_param(double sum, uint64_t n) {
    const double mean = n ? sum / (double)n : 0.0;
    uint32_t k = 0;
    while (k < 40 && (double)(1ull << (k + 1)) <= mean + 1.0) k++;  // 2^k ~ mean: within a fraction of a bit of the best choice
    return k;
}
// bits of every group of 64 regression blocks (one wave per group)
template <int NC>
__global__ __launch_bounds__(256) void k_blk_coef_len(const int64_t *__restrict__ coef, const uint32_t *__restrict__ comp, const uint64_t *n_reg,
                                                      const double *__restrict__ stats, uint32_t *__restrict__ group_bits) {
    const uint64_t nr = *n_reg;
    const uint64_t ngroups = (nr + RICE_GROUP - 1) / RICE_GROUP;
    uint32_t k[NC];
    for (int i = 0; i < NC; i++) k[i] = rice_param(stats[i], nr);
    for (uint64_t g = (uint64_t)blockIdx.x * 4 + threadIdx.x / WAVE; g < ngroups; g += (uint64_t)gridDim.x * 4) {
        const uint64_t r = g * RICE_GROUP + lane_id();
        uint32_t bits = 0;
        if (r < nr) {
            uint64_t u[NC];
            coef_delta<NC>(coef, comp, r, u);
            for (int i = 0; i < NC; i++) bits += rice_len(u[i], k[i]);
        }
        bits = wave_sum(bits);
        if (lane_id() == 0) group_bits[g] = bits;
    }
}
__device__ __forceinline__ void put_bits(uint32_t *words, uint64_t pos, uint64_t v, uint32_t nb) {  // nb <= 64, MSB first
    while (nb) {
        const uint32_t room = 32u - (uint32_t)(pos & 31);
        const uint32_t take = nb < room ? nb : room;
        const uint32_t chunk = (uint32_t)((v >> (nb - take)) & (take == 32 ? 0xFFFFFFFFull : ((1ull << take) - 1ull)));
        atomicOr(&words[pos >> 5], chunk << (room - take));
        pos += take;
        nb -= take;
    }
}
// header, selection bits, Rice parameters, group offsets (one workgroup scans the group sizes), then the bits
// selection section of the side information: 2 bits per block, four blocks per byte (its place does not depend on anything counted)
__global__ __launch_bounds__(256) void k_blk_sel_pack(const uint8_t *__restrict__ sel, uint32_t nblocks, uint8_t *__restrict__ side) {
    const uint64_t sel_bytes = side_sel_bytes(nblocks);
    for (uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < sel_bytes; b += (uint64_t)gridDim.x * 256) {
        uint32_t v = 0;
        for (uint32_t k = 0; k < 4; k++) {
            const uint64_t blk = b * 4 + k;
            if (blk < nblocks) v |= (uint32_t)(sel[blk] & 3u) << (2 * k);
        }
        side[SIDE_HDR + b] = (uint8_t)v;
    }
}
template <int NC>
__global__ __launch_bounds__(1024) void k_blk_side_layout(uint32_t nblocks, const uint64_t *n_reg,
                                                          const double *__restrict__ stats, const uint32_t *__restrict__ group_bits,
                                                          uint8_t *__restrict__ side, uint64_t *side_bytes) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const uint64_t nr = *n_reg;
    const uint64_t sel_bytes = side_sel_bytes(nblocks);
    const uint32_t ngroups = (uint32_t)((nr + RICE_GROUP - 1) / RICE_GROUP);
    uint8_t *kp = side + SIDE_HDR + sel_bytes;
    constexpr uint32_t PB = side_par_bytes<NC>();
    uint32_t *goff = reinterpret_cast<uint32_t *>(kp + PB);
    if (threadIdx.x == 0) {
        const uint32_t h0[2] = {1u, 2u};
        const uint64_t h1[2] = {nblocks, nr};
        memcpy(side, h0, 8);
        memcpy(side + 8, h1, 16);
        for (uint32_t i = 0; i < PB - 4; i++) kp[i] = i < (uint32_t)NC ? (uint8_t)rice_param(stats[i], nr) : (uint8_t)0;
        memcpy(kp + PB - 4, &ngroups, 4);
        s_carry = 0;
    }
    // (the selection bits are packed by k_blk_sel_pack, a launch of its own: one workgroup walking 155 KB of them was 100 of this
    // kernel's 129 us at C4's slab)
    __syncthreads();
    for (uint32_t base = 0; base < ngroups; base += 1024) {
        const uint32_t g = base + threadIdx.x;
        const uint32_t mine = g < ngroups ? group_bits[g] : 0u;
        const uint32_t incl = wave_incl_scan(mine);
        if (lane_id() == WAVE - 1) s_w[threadIdx.x / WAVE] = incl;
        __syncthreads();
        uint32_t run = s_carry + incl - mine, tot = 0;
        for (uint32_t w = 0; w < 16; w++) {
            if (w < threadIdx.x / WAVE) run += s_w[w];
            tot += s_w[w];
        }
        if (g < ngroups) goff[g] = run;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    const uint64_t words = ((uint64_t)s_carry + 31) / 32;
    uint32_t *bits = goff + ngroups;
    for (uint64_t i = threadIdx.x; i < words; i += 1024) bits[i] = 0;
    if (threadIdx.x == 0) *side_bytes = SIDE_HDR + sel_bytes + PB + 4ull * ngroups + 4 * words;
}
template <int NC>
__global__ __launch_bounds__(256) void k_blk_coef_write(const int64_t *__restrict__ coef, const uint32_t *__restrict__ comp, const uint64_t *n_reg,
                                                        uint32_t nblocks, uint8_t *__restrict__ side) {
    const uint64_t nr = *n_reg;
    const uint64_t ngroups = (nr + RICE_GROUP - 1) / RICE_GROUP;
    const uint8_t *kp = side + SIDE_HDR + side_sel_bytes(nblocks);
    uint32_t k[NC];
    for (int i = 0; i < NC; i++) k[i] = kp[i];
    const uint32_t *goff = reinterpret_cast<const uint32_t *>(kp + side_par_bytes<NC>());
    uint32_t *bits = const_cast<uint32_t *>(goff) + ngroups;
    for (uint64_t g = (uint64_t)blockIdx.x * 4 + threadIdx.x / WAVE; g < ngroups; g += (uint64_t)gridDim.x * 4) {
        const uint64_t r = g * RICE_GROUP + lane_id();
        uint64_t u[NC];
        for (int i = 0; i < NC; i++) u[i] = 0;
        uint32_t len = 0;
        if (r < nr) {
            coef_delta<NC>(coef, comp, r, u);
            for (int i = 0; i < NC; i++) len += rice_len(u[i], k[i]);
        }
        uint64_t pos = (uint64_t)goff[g] + (wave_incl_scan(len) - len);
        if (r < nr)
            for (int i = 0; i < NC; i++) {
                const uint64_t q = u[i] >> k[i];
                if (q < RICE_ESC) {
                    put_bits(bits, pos, ((1ull << q) - 1ull) << 1, (uint32_t)q + 1u);  // q ones, a zero
                    pos += q + 1;
                    if (k[i]) put_bits(bits, pos, u[i] & ((1ull << k[i]) - 1ull), k[i]);
                    pos += k[i];
                } else {
                    put_bits(bits, pos, (1ull << RICE_ESC) - 1ull, RICE_ESC);
                    pos += RICE_ESC;
                    put_bits(bits, pos, u[i], 64);
                    pos += 64;
                }
            }
    }
}

// Small block counts (round 6; up to SIDE_SMALL_BLOCKS: C1's 8192 blocks of 128 values, a 4 MB HDF5 chunk): the eight launches above in ONE
// workgroup, step by step with a barrier between two steps — each of them is a few microseconds of work behind 4.5 us of launch, 41 us
// of C1's 190. Same arrays, same arithmetic, same bytes (the coefficient sums are sums of integers below 2^53: exact in any order).
// Everything a step reads from memory was written by an earlier step of this workgroup and never read before (no stale line in the
// unit's L1). Up to SIDE_SMALL_REG regression blocks whose zigzagged differences all fit 32 bits (C1: every block of 8192 is a regression
// block) a thread keeps the differences of its (at most eight) blocks in registers from the statistics to the bits, and the groups' sizes
// and offsets live in LDS: the coefficient arrays are read once instead of three times (each time a chain of two dependent loads per
// step of a loop that nothing overlaps).
#define SIDE_SMALL_BLOCKS 16384u
#define SIDE_SMALL_REG 8192u
template <int NC>
__device__ __forceinline__ void side_put_block(uint32_t *bits, uint64_t pos, const uint64_t (&u)[NC], const uint32_t (&k)[NC]) {
    for (int i = 0; i < NC; i++) {
        const uint64_t q = u[i] >> k[i];
        if (q < RICE_ESC) {
            put_bits(bits, pos, ((1ull << q) - 1ull) << 1, (uint32_t)q + 1u);  // q ones, a zero
            pos += q + 1;
            if (k[i]) put_bits(bits, pos, u[i] & ((1ull << k[i]) - 1ull), k[i]);
            pos += k[i];
        } else {
            put_bits(bits, pos, (1ull << RICE_ESC) - 1ull, RICE_ESC);
            pos += RICE_ESC;
            put_bits(bits, pos, u[i], 64);
            pos += 64;
        }
    }
}
#ifdef LAB_SIDE_TS  // (lab: where the one-workgroup side section spends its time)
#define SIDE_TS(i) do { if (threadIdx.x == 0) lab_ts[i] = wall_clock64(); } while (0)
#define SIDE_TS_PRINT() do { if (threadIdx.x == 0) printf("side_small: nr %u in_regs %d | sel %.1f rank %.1f stats %.1f len %.1f hdr+scan %.1f zero %.1f bits %.1f us\n", (unsigned)nr, (int)in_regs, (lab_ts[1]-lab_ts[0])/100.0, (lab_ts[2]-lab_ts[1])/100.0, (lab_ts[3]-lab_ts[2])/100.0, (lab_ts[4]-lab_ts[3])/100.0, (lab_ts[5]-lab_ts[4])/100.0, (lab_ts[6]-lab_ts[5])/100.0, (lab_ts[7]-lab_ts[6])/100.0); } while (0)
#else
#define SIDE_TS(i) do { } while (0)
#define SIDE_TS_PRINT() do { } while (0)
#endif
template <int NC, bool RANK>
__global__ __launch_bounds__(1024) void k_blk_side_small(const uint8_t *sel, uint32_t nblocks, uint32_t *rank, uint32_t *comp, uint64_t *n_reg_io,
                                                         const int64_t *coef, double *stats, uint32_t *group_bits, uint8_t *side, uint64_t *side_bytes,
                                                         const uint64_t *range_hist, uint32_t *range) {
#ifdef LAB_SIDE_TS
    uint64_t lab_ts[8];
#endif
    if (blockIdx.x > 0) {  // (workgroups 1 .. 64: the range of the histogram's non-empty bins — k_hist_range's work, which stage 2 then finds done)
        const uint32_t i = (blockIdx.x - 1) * 1024u + threadIdx.x;
        const bool nz = range_hist[i] != 0;
        const unsigned long long m = __ballot(nz);
        if (m && lane_id() == 0) {
            const uint32_t base = i;  // lane 0's bin
            const uint32_t lo = base + (uint32_t)__ffsll((long long)m) - 1, hi = base + 63u - (uint32_t)__clzll((long long)m);
            atomicMax(&range[0], 0xFFFFu - lo);
            atomicMax(&range[1], hi);
            atomicAdd(&range[2], (uint32_t)__popcll(m));
        }
        return;
    }
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    __shared__ double s_stats[NC];
    __shared__ uint32_t s_gbits[SIDE_SMALL_REG / RICE_GROUP], s_goff[SIDE_SMALL_REG / RICE_GROUP];
    const uint32_t t = threadIdx.x, wv = t / WAVE;
    SIDE_TS(0);
    if (t == 0) s_carry = 0;
    if (t < (uint32_t)NC) s_stats[t] = 0.0;
    // the selection bits (their place depends on nothing counted), and with RANK in the same walk over the choices: the rank of every block
    // among the regression blocks, their compacted list, their number (the ranks themselves are not stored: the side section's builder is
    // their only reader here, and it goes by the list)
    const uint64_t sel_bytes = side_sel_bytes(nblocks);
    if (!RANK) {
        for (uint64_t b = t; b < sel_bytes; b += 1024) {
            uint32_t v = 0;
            for (uint32_t q = 0; q < 4; q++) {
                const uint64_t blk = b * 4 + q;
                if (blk < nblocks) v |= (uint32_t)(sel[blk] & 3u) << (2 * q);
            }
            side[SIDE_HDR + b] = (uint8_t)v;
        }
    }
    __syncthreads();
    SIDE_TS(1);
    (void)rank;
    if (RANK) {
        constexpr uint32_t PER = 8;
        const uint32_t span = (uint32_t)sel_bytes * 4;  // blocks the selection section has room for (a multiple of 32): the padding is written too
        for (uint32_t base = 0; base < span; base += 1024 * PER) {
            const uint32_t b0 = base + t * PER;
            uint32_t f[PER], mine = 0, packed = 0;
            uint64_t w8 = 0;
            if (b0 + PER <= nblocks) w8 = *reinterpret_cast<const uint64_t *>(sel + b0);  // (eight choices, one load: the array is 8-byte aligned)
            else
                for (uint32_t k = 0; k < PER; k++)
                    if (b0 + k < nblocks) w8 |= (uint64_t)sel[b0 + k] << (8 * k);
#pragma unroll
            for (uint32_t k = 0; k < PER; k++) {
                const uint32_t c = (uint32_t)(w8 >> (8 * k)) & 0xFFu;
                f[k] = (b0 + k < nblocks && c == 2) ? 1u : 0u;
                mine += f[k];
                packed |= (c & 3u) << (2 * k);
            }
            if (b0 < span) *reinterpret_cast<uint16_t *>(side + SIDE_HDR + b0 / 4) = (uint16_t)packed;
            const uint32_t incl = wave_incl_scan(mine);
            if (lane_id() == WAVE - 1) s_w[wv] = incl;
            __syncthreads();
            uint32_t run = s_carry + incl - mine, tot = 0;
            for (uint32_t w = 0; w < 16; w++) {
                if (w < wv) run += s_w[w];
                tot += s_w[w];
            }
#pragma unroll
            for (uint32_t k = 0; k < PER; k++) {
                if (f[k]) comp[run] = b0 + k;
                run += f[k];
            }
            __syncthreads();
            if (t == 0) s_carry += tot;
            __syncthreads();
        }
        if (t == 0) *n_reg_io = s_carry;
    }
    const uint64_t nr = RANK ? (uint64_t)s_carry : *n_reg_io;
    const uint32_t ngroups = (uint32_t)((nr + RICE_GROUP - 1) / RICE_GROUP);
    uint8_t *kp = side + SIDE_HDR + sel_bytes;
    constexpr uint32_t PB = side_par_bytes<NC>();
    uint32_t *goff = reinterpret_cast<uint32_t *>(kp + PB);
    uint32_t *bits = goff + ngroups;
    __threadfence_block();
    __syncthreads();
    SIDE_TS(2);
    const bool few = nr <= SIDE_SMALL_REG;  // (workgroup-uniform)
    constexpr int J = SIDE_SMALL_REG / 1024;
    uint32_t ur[J][NC];
    __shared__ uint32_t s_big;
    if (t == 0) s_big = 0;
    // sums of the zigzagged differences per coefficient -> the Rice parameters
    {
        double sm[NC];
        for (int i = 0; i < NC; i++) sm[i] = 0;
        if (few) {
            uint32_t big = 0;
            constexpr int JB = 2;  // blocks whose loads are in flight together (two dependent round trips per batch; all eight at once spilled)
#pragma unroll
            for (int jb = 0; jb < J; jb += JB) {
                if ((uint64_t)jb * 1024 >= nr) {  // (workgroup-uniform: nothing of the list left)
#pragma unroll
                    for (int j = 0; j < JB; j++)
                        for (int i = 0; i < NC; i++) ur[jb + j][i] = 0;
                    continue;
                }
                uint32_t c0[JB], c1[JB];
#pragma unroll
                for (int j = 0; j < JB; j++) {
                    const uint64_t r = (uint64_t)(jb + j) * 1024 + t;
                    c0[j] = r < nr ? comp[r] : 0u;
                    c1[j] = r < nr && r ? comp[r - 1] : 0u;
                }
                int64_t cur[JB][NC], prv[JB][NC];
#pragma unroll
                for (int j = 0; j < JB; j++)
                    for (int i = 0; i < NC; i++) {
                        cur[j][i] = coef[(uint64_t)c0[j] * NC + i];  // (a thread past the list reads block 0's: a valid address, the value unused)
                        prv[j][i] = coef[(uint64_t)c1[j] * NC + i];
                    }
#pragma unroll
                for (int j = 0; j < JB; j++) {
                    const uint64_t r = (uint64_t)(jb + j) * 1024 + t;
                    for (int i = 0; i < NC; i++) {
                        const uint64_t u = r < nr ? zigzag(cur[j][i] - (r ? prv[j][i] : 0)) : 0ull;
                        ur[jb + j][i] = (uint32_t)u;
                        big |= (uint32_t)(u >> 32);
                        sm[i] += (double)u;
                    }
                }
            }
            if (__ballot(big != 0) && lane_id() == 0) atomicOr(&s_big, 1u);
        } else {
            for (uint64_t r = t; r < nr; r += 1024) {
                uint64_t u[NC];
                coef_delta<NC>(coef, comp, r, u);
                for (int i = 0; i < NC; i++) sm[i] += (double)u[i];
            }
        }
        for (int i = 0; i < NC; i++) {
            sm[i] = wave_sum_f64(sm[i]);
            if (lane_id() == 0 && sm[i] != 0) atomicAdd(&s_stats[i], sm[i]);
        }
    }
    __syncthreads();
    SIDE_TS(3);
    const bool in_regs = few && s_big == 0;  // (workgroup-uniform)
    if (t < (uint32_t)NC) stats[t] += s_stats[t];  // (zeroed by the caller, as for k_blk_coef_stats)
    uint32_t k[NC];
    for (int i = 0; i < NC; i++) k[i] = rice_param(s_stats[i], nr);
    // bits of every group of 64 regression blocks (a wave per group)
    uint32_t len_r[J];
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < J; j++) {
            const uint64_t r = (uint64_t)j * 1024 + t;
            uint32_t b = 0;
            if (r < nr)
                for (int i = 0; i < NC; i++) b += rice_len((uint64_t)ur[j][i], k[i]);
            len_r[j] = b;
            const uint32_t tot = wave_sum(b);
            const uint32_t g = (uint32_t)j * 16 + wv;
            if (lane_id() == 0 && g < ngroups) s_gbits[g] = tot;
        }
    } else {
        for (uint32_t g = wv; g < ngroups; g += 16) {
            const uint64_t r = (uint64_t)g * RICE_GROUP + lane_id();
            uint32_t b = 0;
            if (r < nr) {
                uint64_t u[NC];
                coef_delta<NC>(coef, comp, r, u);
                for (int i = 0; i < NC; i++) b += rice_len(u[i], k[i]);
            }
            b = wave_sum(b);
            if (lane_id() == 0) group_bits[g] = b;
        }
    }
    SIDE_TS(4);
    // header, parameters, group offsets
    if (t == 0) {
        const uint32_t h0[2] = {1u, 2u};
        const uint64_t h1[2] = {nblocks, nr};
        memcpy(side, h0, 8);
        memcpy(side + 8, h1, 16);
        for (uint32_t i = 0; i < PB - 4; i++) kp[i] = i < (uint32_t)NC ? (uint8_t)k[i < (uint32_t)NC ? i : 0] : (uint8_t)0;
        memcpy(kp + PB - 4, &ngroups, 4);
        s_carry = 0;
    }
    __threadfence_block();
    __syncthreads();
    for (uint32_t base = 0; base < ngroups; base += 1024) {
        const uint32_t g = base + t;
        const uint32_t mine = g < ngroups ? (in_regs ? s_gbits[g] : group_bits[g]) : 0u;
        const uint32_t incl = wave_incl_scan(mine);
        if (lane_id() == WAVE - 1) s_w[wv] = incl;
        __syncthreads();
        uint32_t run = s_carry + incl - mine, tot = 0;
        for (uint32_t w = 0; w < 16; w++) {
            if (w < wv) run += s_w[w];
            tot += s_w[w];
        }
        if (g < ngroups) {
            goff[g] = run;
            if (in_regs) s_goff[g] = run;
        }
        __syncthreads();
        if (t == 0) s_carry += tot;
        __syncthreads();
    }
    SIDE_TS(5);
    const uint64_t words = ((uint64_t)s_carry + 31) / 32;
    for (uint64_t i = t; i < words; i += 1024) bits[i] = 0;
    if (t == 0) *side_bytes = SIDE_HDR + sel_bytes + PB + 4ull * ngroups + 4 * words;
    __threadfence_block();
    __syncthreads();
    SIDE_TS(6);
    // the coefficient chain's bits
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < J; j++) {
            const uint64_t r = (uint64_t)j * 1024 + t;
            const uint32_t g = (uint32_t)j * 16 + wv;
            const uint32_t before = wave_incl_scan(len_r[j]) - len_r[j];
            uint64_t u[NC];
            for (int i = 0; i < NC; i++) u[i] = ur[j][i];
            if (r < nr) side_put_block<NC>(bits, (uint64_t)s_goff[g] + before, u, k);
        }
    } else {
        for (uint32_t g = wv; g < ngroups; g += 16) {
            const uint64_t r = (uint64_t)g * RICE_GROUP + lane_id();
            uint64_t u[NC];
            for (int i = 0; i < NC; i++) u[i] = 0;
            uint32_t len = 0;
            if (r < nr) {
                coef_delta<NC>(coef, comp, r, u);
                for (int i = 0; i < NC; i++) len += rice_len(u[i], k[i]);
            }
            const uint64_t pos = (uint64_t)goff[g] + (wave_incl_scan(len) - len);
            if (r < nr) side_put_block<NC>(bits, pos, u, k);
        }
    }
    SIDE_TS(7);
    SIDE_TS_PRINT();
}

// ---- decoder side: selection bits out, coefficient differences parsed (one thread per group) and summed up ----
__global__ __launch_bounds__(256) void k_blk_side_sel(const uint8_t *__restrict__ side, uint32_t nblocks, uint8_t *__restrict__ sel) {
    for (uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < nblocks; b += (uint64_t)gridDim.x * 256)
        sel[b] = (side[SIDE_HDR + b / 4] >> (2 * (b & 3))) & 3u;
}
__device__ __forceinline__ uint32_t get_bit(const uint32_t *words, uint64_t pos) { return (words[pos >> 5] >> (31u - (uint32_t)(pos & 31))) & 1u; }
__device__ __forceinline__ uint64_t get_bits(const uint32_t *words, uint64_t pos, uint32_t nb) {
    uint64_t v = 0;
    while (nb) {
        const uint32_t room = 32u - (uint32_t)(pos & 31);
        const uint32_t take = nb < room ? nb : room;
        const uint32_t w = words[pos >> 5];
        const uint32_t chunk = (w >> (room - take)) & (take == 32 ? 0xFFFFFFFFu : ((1u << take) - 1u));
        v = take == 64 ? chunk : ((v << take) | chunk);
        pos += take;
        nb -= take;
    }
    return v;
}
// 64 bits of the stream from bit `pos` on (first bit = bit 63), zeros beyond its end
__device__ __forceinline__ uint64_t peek64(const uint32_t *__restrict__ words, uint64_t pos, uint64_t nwords) {
    const uint64_t i = pos >> 5;
    const uint32_t sh = (uint32_t)(pos & 31);
    const uint64_t w0 = i < nwords ? words[i] : 0u, w1 = i + 1 < nwords ? words[i + 1] : 0u, w2 = i + 2 < nwords ? words[i + 2] : 0u;
    const uint64_t hi = (w0 << 32) | w1;
    return sh ? (hi << sh) | (w2 >> (32 - sh)) : hi;
}
// A WAVE per group of 64 regression blocks (round 5; the groups' bit offsets are in the section). A group's codes are one dependent chain —
// every code's length decides where the next begins — and a thread per group walked it through memory: 256 steps of ~1 us of load
// latency each, 0.32 ms at C4a's 1 400 groups (and 94 of C1's 250 us), whatever the group count. Now the wave keeps a window of the
// stream in REGISTERS — 64 consecutive words, one per lane, loaded coalesced — and walks the chain with wave-uniform values: the three
// words a 64-bit look needs come by v_readlane, a new window is loaded when the walk leaves the old one (every ~200 codes), a lane keeps
// every 64th result and the results leave 64 at a time. The unary parts are counted on the 64-bit look (count leading ones). It also
// leaves the sums of its group's differences: the chain over the regression blocks is then a scan over the groups (k_blk_coef_gscan)
// and a wave scan inside each (k_blk_coef_apply).
template <int NC>
__global__ __launch_bounds__(256) void k_blk_coef_parse(const uint8_t *__restrict__ side, uint32_t nblocks, uint64_t nr, uint64_t bit_words,
                                                        int64_t *__restrict__ delta_by_rank, int64_t *__restrict__ gsum) {
    const uint64_t ngroups = (nr + RICE_GROUP - 1) / RICE_GROUP;
    const uint8_t *kp = side + SIDE_HDR + side_sel_bytes(nblocks);
    uint32_t k[NC];
    for (int i = 0; i < NC; i++) k[i] = kp[i];
    const uint32_t *goff = reinterpret_cast<const uint32_t *>(kp + side_par_bytes<NC>());
    const uint32_t *bits = goff + ngroups;
    const uint64_t total_bits = bit_words * 32;
    const int lane = lane_id();
    for (uint64_t g = (uint64_t)blockIdx.x * 4 + threadIdx.x / WAVE; g < ngroups; g += (uint64_t)gridDim.x * 4) {
        uint64_t pos = goff[g];  // (wave-uniform from here on)
        const uint64_t r0 = g * RICE_GROUP, r1 = (g + 1) * RICE_GROUP < nr ? (g + 1) * RICE_GROUP : nr;
        uint64_t wbase = pos >> 5;
        uint32_t wreg = wbase + (uint64_t)lane < bit_words ? bits[wbase + lane] : 0u;
        // 64 bits of the stream from bit `at` on (first bit = bit 63), zeros beyond its end — peek64 out of the window
        auto look = [&](uint64_t at) -> uint64_t {
            const uint64_t i = at >> 5;
            if (i < wbase || i + 3 > wbase + WAVE) {  // (wave-uniform)
                wbase = i;
                wreg = wbase + (uint64_t)lane < bit_words ? bits[wbase + lane] : 0u;
            }
            const int j = __builtin_amdgcn_readfirstlane((int)(i - wbase));
            const uint64_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)wreg, j), w1 = (uint32_t)__builtin_amdgcn_readlane((int)wreg, j + 1),
                           w2 = (uint32_t)__builtin_amdgcn_readlane((int)wreg, j + 2);
            const uint32_t sh = (uint32_t)(at & 31);
            const uint64_t hi = (w0 << 32) | w1;
            return sh ? (hi << sh) | (w2 >> (32 - sh)) : hi;
        };
        uint64_t tot[NC];
        for (int i = 0; i < NC; i++) tot[i] = 0;
        int64_t keep = 0;       // this lane's result of the current batch of 64
        uint32_t staged = 0;    // results in the batch (wave-uniform)
        uint64_t out0 = r0 * NC;  // index of the batch's first result
        for (uint64_t r = r0; r < r1; r++)
            for (int i = 0; i < NC; i++) {
                uint64_t u = 0;
                const uint64_t win = look(pos);
                const uint32_t ones = ~win ? (uint32_t)__clzll((long long)~win) : 64u;
                const uint32_t q = ones < RICE_ESC ? ones : RICE_ESC;
                pos += q;
                if (q < RICE_ESC) {
                    pos++;  // the terminating zero
                    uint64_t low = 0;
                    if (k[i] && pos + k[i] <= total_bits) low = q + 1 + k[i] <= 64 ? (win << (q + 1)) >> (64 - k[i]) : look(pos) >> (64 - k[i]);
                    pos += k[i];
                    u = ((uint64_t)q << k[i]) | low;
                } else {
                    u = pos + 64 <= total_bits ? look(pos) : 0;
                    pos += 64;
                }
                const int64_t dl = unzigzag(u);
                tot[i] += (uint64_t)dl;
                keep = lane == (int)staged ? dl : keep;
                if (++staged == WAVE) {
                    delta_by_rank[out0 + lane] = keep;
                    out0 += WAVE;
                    staged = 0;
                }
            }
        if ((uint32_t)lane < staged) delta_by_rank[out0 + lane] = keep;
        if (lane == 0)
            for (int i = 0; i < NC; i++) gsum[g * NC + i] = (int64_t)tot[i];
    }
}
// exclusive scan of the groups' sums, per coefficient (one workgroup, 1024 groups a round)
template <int NC>
__global__ __launch_bounds__(1024) void k_blk_coef_gscan(uint64_t ngroups, int64_t *__restrict__ gsum) {
    __shared__ int64_t s_w[NC][16];
    __shared__ int64_t s_carry[NC];
    if (threadIdx.x < NC) s_carry[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t base = 0; base < ngroups; base += 1024) {
        const uint64_t g = base + threadIdx.x;
        int64_t d[NC], incl[NC];
        for (int i = 0; i < NC; i++) d[i] = g < ngroups ? gsum[g * NC + i] : 0;
        for (int i = 0; i < NC; i++) {
            incl[i] = (int64_t)wave_incl_scan((uint64_t)d[i]);
            if (lane_id() == WAVE - 1) s_w[i][threadIdx.x / WAVE] = incl[i];
        }
        __syncthreads();
        for (int i = 0; i < NC; i++) {
            int64_t run = s_carry[i] + incl[i] - d[i];
            for (uint32_t kk = 0; kk < threadIdx.x / WAVE; kk++) run += s_w[i][kk];
            if (g < ngroups) gsum[g * NC + i] = run;
        }
        __syncthreads();
        if (threadIdx.x < NC) {
            int64_t tot = 0;
            for (int kk = 0; kk < 16; kk++) tot += s_w[threadIdx.x][kk];
            s_carry[threadIdx.x] += tot;
        }
        __syncthreads();
    }
}
// differences -> coefficients: a wave per group of 64 regression blocks, the group's inflow from the scan above
template <int NC>
__global__ __launch_bounds__(256) void k_blk_coef_apply(uint64_t nr, const int64_t *__restrict__ gpre, int64_t *__restrict__ coef_by_rank) {
    const uint64_t ngroups = (nr + RICE_GROUP - 1) / RICE_GROUP;
    for (uint64_t g = (uint64_t)blockIdx.x * 4 + threadIdx.x / WAVE; g < ngroups; g += (uint64_t)gridDim.x * 4) {
        const uint64_t r = g * RICE_GROUP + lane_id();
        for (int i = 0; i < NC; i++) {
            const uint64_t d = r < nr ? (uint64_t)coef_by_rank[r * NC + i] : 0ull;
            const uint64_t v = wave_incl_scan(d) + (uint64_t)gpre[g * NC + i];
            if (r < nr) coef_by_rank[r * NC + i] = (int64_t)v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// decoder: one anti-diagonal front of blocks per launch, one wave per block. d_out holds lattice values q~ (Q) until the
// final pass turns them into T.
// ------------------------------------------------------------------------------------------------------------
// the three line-scan passes that invert a block's stencil from its low halo, the stencil order a compile-time constant
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}
// l -> (l / e, l % e); whole blocks of a compiled edge divide by the constant (a lone wave issues an instruction every four cycles:
// the run-time divisions of the three passes were ~250 of a pass's instructions, 1 us of an inner step's 3.8)
template <int CB>
__device__ __forceinline__ void line_split(uint32_t l, uint32_t e, uint32_t &hi, uint32_t &lo) {
    if (CB && e == (uint32_t)CB) {
        hi = l / (uint32_t)(CB ? CB : 1);
        lo = l - hi * (uint32_t)CB;
    } else {
        hi = l / e;
        lo = l - hi * e;
    }
}
template <typename T, int CB, int ORDER>
__device__ __forceinline__ void blk_invert(typename QTraits<T>::Q *sq, typename QTraits<T>::Q *sa, typename QTraits<T>::Q *qout, const BlkGeom &g,
                                           const TileView &tv, uint64_t d1, uint64_t d2, int lane, bool keep) {
    // (tv: where the block's neighbourhood sits in sq / sa; keep: the result also replaces the deltas in sq — the tile is shared by a
    // group of blocks and the next ones read it as their halo)
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    wave_lds_fence();
    // ---- pass along x: a = Dy^m Dz^m q on the two halo columns, then the recurrence over the own columns ----
    for (uint32_t l = lane; l < g.ez * g.ey; l += WAVE) {
        uint32_t tz, ty;
        line_split<CB>(l, g.ey, tz, ty);
        tz += 2;
        ty += 2;
        UQ a[2];
#pragma unroll
        for (uint32_t tx = 0; tx < 2; tx++) {
            UQ s = 0;
#pragma unroll
            for (int k = 0; k <= ORDER; k++)
#pragma unroll
                for (int j = 0; j <= ORDER; j++) s += (UQ)((Q)(lz_w(ORDER, k) * lz_w(ORDER, j)) * sq[tv_at(tv, tz - k, ty - j, tx)]);
            a[tx] = s;
        }
        UQ p2 = a[0], p1 = a[1];
        if (CB) {  // the line's values first (all reads in flight), the recurrence in registers, then the writes
            UQ in[CB ? CB : 1];
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)CB; i++) in[i] = i < g.ex ? (UQ)sq[tv_at(tv, tz, ty, 2 + i)] : (UQ)0;
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)CB; i++) {
                const UQ v = ORDER == 1 ? p1 + in[i] : (UQ)(2 * p1 - p2 + in[i]);
                in[i] = v;
                p2 = p1;
                p1 = v;
            }
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)CB; i++)
                if (i < g.ex) sa[tv_at(tv, tz, ty, 2 + i)] = (Q)in[i];
        } else
        for (uint32_t tx = 2; tx < 2 + g.ex; tx++) {
            const UQ in = (UQ)sq[tv_at(tv, tz, ty, tx)];
            const UQ v = ORDER == 1 ? p1 + in : (UQ)(2 * p1 - p2 + in);
            sa[tv_at(tv, tz, ty, tx)] = (Q)v;
            p2 = p1;
            p1 = v;
        }
    }
    wave_lds_fence();
    // ---- pass along y: b = Dz^m q on the two halo rows (own columns), recurrence over the own rows ----
    for (uint32_t l = lane; l < g.ez * g.ex; l += WAVE) {
        uint32_t tz, tx;
        line_split<CB>(l, g.ex, tz, tx);
        tz += 2;
        tx += 2;
        UQ b[2];
#pragma unroll
        for (uint32_t ty = 0; ty < 2; ty++) {
            UQ s = 0;
#pragma unroll
            for (int k = 0; k <= ORDER; k++) s += (UQ)((Q)lz_w(ORDER, k) * sq[tv_at(tv, tz - k, ty, tx)]);
            b[ty] = s;
        }
        UQ p2 = b[0], p1 = b[1];
        if (CB) {
            UQ in[CB ? CB : 1];
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)CB; i++) in[i] = i < g.ey ? (UQ)sa[tv_at(tv, tz, 2 + i, tx)] : (UQ)0;
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)CB; i++) {
                const UQ v = ORDER == 1 ? p1 + in[i] : (UQ)(2 * p1 - p2 + in[i]);
                in[i] = v;
                p2 = p1;
                p1 = v;
            }
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)CB; i++)
                if (i < g.ey) sa[tv_at(tv, tz, 2 + i, tx)] = (Q)in[i];
        } else
        for (uint32_t ty = 2; ty < 2 + g.ey; ty++) {
            const UQ in = (UQ)sa[tv_at(tv, tz, ty, tx)];
            const UQ v = ORDER == 1 ? p1 + in : (UQ)(2 * p1 - p2 + in);
            sa[tv_at(tv, tz, ty, tx)] = (Q)v;
            p2 = p1;
            p1 = v;
        }
    }
    wave_lds_fence();
    // ---- pass along z: inflow = q~ of the two halo planes; the result is q ----
    for (uint32_t l = lane; l < g.ey * g.ex; l += WAVE) {
        uint32_t ty, tx;
        line_split<CB>(l, g.ex, ty, tx);
        ty += 2;
        tx += 2;
        UQ p2 = (UQ)sq[tv_at(tv, 0, ty, tx)], p1 = (UQ)sq[tv_at(tv, 1, ty, tx)];
        if (CB) {
            UQ in[CB ? CB : 1];
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)CB; i++) in[i] = i < g.ez ? (UQ)sa[tv_at(tv, 2 + i, ty, tx)] : (UQ)0;
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)CB; i++) {
                const UQ v = ORDER == 1 ? p1 + in[i] : (UQ)(2 * p1 - p2 + in[i]);
                in[i] = v;
                p2 = p1;
                p1 = v;
            }
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)CB; i++)
                if (i < g.ez) {
                    // keep: the tile keeps the result and the caller writes it out once, after its last inner step — a store
                    // here is waited for by the next step's barrier (s_waitcnt vmcnt(0): ~2 us of write latency per step)
                    if (keep) sq[tv_at(tv, 2 + i, ty, tx)] = (Q)in[i];
                    else qout[((uint64_t)(g.oz + i) * d1 + (g.oy + ty - 2)) * d2 + (g.ox + tx - 2)] = (Q)in[i];
                }
        } else
        for (uint32_t tz = 2; tz < 2 + g.ez; tz++) {
            const UQ in = (UQ)sa[tv_at(tv, tz, ty, tx)];
            const UQ v = ORDER == 1 ? p1 + in : (UQ)(2 * p1 - p2 + in);
            if (keep) sq[tv_at(tv, tz, ty, tx)] = (Q)v;
            else qout[((uint64_t)(g.oz + tz - 2) * d1 + (g.oy + ty - 2)) * d2 + (g.ox + tx - 2)] = (Q)v;
            p2 = p1;
            p1 = v;
        }
    }
}

template <typename T, int CB>
__global__ __launch_bounds__(256) void k_blk_decode(const uint16_t *__restrict__ codes, const void *deltas_, void *d_out, szk_blk_params p, uint32_t diag, uint32_t bz_lo,
                                                    uint32_t npairs, const uint32_t *__restrict__ rank, const int64_t *__restrict__ coef_by_rank) {
    using Q = typename QTraits<T>::Q;
    __shared__ Q s_q[4][CB ? (CB + 2) * (CB + 2) * (CB + 2) : BLK_TILE];
    __shared__ Q s_a[4][CB ? (CB + 2) * (CB + 2) * (CB + 2) : BLK_TILE];
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    const uint32_t wv = threadIdx.x / WAVE;
    Q *sq = s_q[wv], *sa = s_a[wv];
    const uint32_t B = CB ? (uint32_t)CB : p.B, E = B + 2;
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    Q *qout = reinterpret_cast<Q *>(d_out);
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    const uint32_t pair = blockIdx.x * 4 + wv;
    bool live = pair < npairs;
    uint32_t bz = 0, by = 0, bx = 0;
    if (live) {
        bz = bz_lo + pair / p.nb[1];
        by = pair % p.nb[1];
        live = bz + by <= diag && diag - bz - by < p.nb[2] && bz < p.nb[0];
        bx = live ? diag - bz - by : 0;
    }
    const uint32_t task = (bz * p.nb[1] + by) * p.nb[2] + bx;
    const int sid = live ? (int)p.sel[task] : 0;
    const BlkGeom g = blk_geom(p, live ? task : 0);
    const uint32_t nown = live ? g.ez * g.ey * g.ex : 0;
    if (live && sid == 2) {  // regression: values, no dependency
        const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
        T rc[4];
        coef_recover(coef_by_rank + (uint64_t)rank[task] * 4, cl, rc);
        for (uint32_t t = lane; t < nown; t += WAVE) {
            uint32_t i0, i1, i2;
            own_index<CB>(g, t, i0, i1, i2);
            const uint64_t gi = ((uint64_t)(g.oz + i0) * d1 + (g.oy + i1)) * d2 + (g.ox + i2);
            const uint32_t code = codes[g.coff + t];
            Q qt = 0;
            if (code) {
                bool bad;
                qt = lat.quant(ref_recover(reg_predict(rc, i0, i1, i2), (int)code, p.eb, (int)p.radius), bad);
                if (bad) qt = 0;
            }
            qout[gi] = qt;
        }
        live = false;  // (still walks through the barriers below)
    }
    const int order = sid == 1 ? 2 : 1;
    // ---- tile: halo = finished q~ of the lower neighbours (d_out, element order), own region = this block's deltas (the
    // codes expanded in their own order, block by block: szk_launch_expand_deltas into `deltas`) ----
    if (live) {
        for (uint32_t t = lane; t < E * E * E; t += WAVE) {
            const uint32_t tx = t % E, ty = (t / E) % E, tz = t / (E * E);
            const int64_t z = (int64_t)g.oz + tz - 2, y = (int64_t)g.oy + ty - 2, x = (int64_t)g.ox + tx - 2;
            Q v = 0;
            if (tz >= 2 && ty >= 2 && tx >= 2) {
                if (tz - 2 < g.ez && ty - 2 < g.ey && tx - 2 < g.ex) v = deltas[g.coff + ((uint64_t)(tz - 2) * g.ey + (ty - 2)) * g.ex + (tx - 2)];
            } else if (z >= 0 && y >= 0 && x >= 0 && z < (int64_t)p.d[0] && y < (int64_t)d1 && x < (int64_t)d2) {
                v = qout[((uint64_t)z * d1 + (uint64_t)y) * d2 + (uint64_t)x];
            }
            sq[t] = v;
        }
    }
    // (the tile is the wave's own: its LDS writes only need to be visible to itself)
    if (live) {
        const TileView tv{E * E, E, 0};
        if (order == 1) blk_invert<T, CB, 1>(sq, sa, qout, g, tv, d1, d2, lane, false);
        else blk_invert<T, CB, 2>(sq, sa, qout, g, tv, d1, d2, lane, false);
    }
}

// (Round 3's decoder — groups of 2 x 2 x 2 blocks inverted by line scans, a launch per front — was superseded twice (closed-form groups,
// then the one-launch chain). It is kept for A/B builds only: -DSZ3HIP_LAB, tools/build_lab.sh; the product library does not carry it.)
#ifdef SZ3HIP_LAB
// the regression blocks' lattice values, all of them before the fronts start (no dependency: coefficients and codes are all they need)
template <typename T, int CB>
__global__ __launch_bounds__(256) void k_blk_pre3(const uint16_t *__restrict__ codes, void *d_out, szk_blk_params p, uint32_t nblocks,
                                                  const uint32_t *__restrict__ rank, const int64_t *__restrict__ coef_by_rank) {
    using Q = typename QTraits<T>::Q;
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    Q *qout = reinterpret_cast<Q *>(d_out);
    const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
    for (uint32_t task = blockIdx.x * 4 + threadIdx.x / WAVE; task < nblocks; task += gridDim.x * 4) {
        if (p.sel[task] != 2) continue;
        const BlkGeom g = blk_geom(p, task);
        const uint32_t nown = g.ez * g.ey * g.ex;
        T rc[4];
        coef_recover(coef_by_rank + (uint64_t)rank[task] * 4, cl, rc);
        for (uint32_t t = lane; t < nown; t += WAVE) {
            uint32_t i0, i1, i2;
            own_index<CB>(g, t, i0, i1, i2);
            const uint32_t code = codes[g.coff + t];
            Q qt = 0;
            if (code) {
                bool bad;
                qt = lat.quant(ref_recover(reg_predict(rc, i0, i1, i2), (int)code, p.eb, (int)p.radius), bad);
                if (bad) qt = 0;
            }
            qout[((uint64_t)(g.oz + i0) * d1 + (g.oy + i1)) * d2 + (g.ox + i2)] = qt;
        }
    }
}
// The decoder on GROUPS of 2 x 2 x 2 blocks: the chain of fronts is what a block stream's decoding costs (a front = a launch + a
// block's latency: tile load, three line-scan passes, store), and a group halves it — 181 instead of 362 fronts at C4's slab. A
// workgroup of eight waves loads the group's tile once (halo = finished q~ of the lower neighbours, own regions = deltas), the
// regression blocks fill in their values, then the Lorenzo blocks are inverted in four inner steps (lz + ly + lx = 0..3) with a
// barrier in between: an inverted block leaves its q~ in the shared tile, where the next step's blocks find their halo.
template <typename T, int CB>
__global__ __launch_bounds__(512) void k_blk_decode_g(const uint16_t *__restrict__ codes, const void *deltas_, void *d_out, szk_blk_params p, uint32_t diag,
                                                      uint32_t gz_lo, uint32_t npairs, const uint32_t *__restrict__ rank,
                                                      const int64_t *__restrict__ coef_by_rank) {
    using Q = typename QTraits<T>::Q;
    constexpr uint32_t TE = 2 * CB + 2;
    // ONE tile, inverted in place (a pass reads its own line and halo positions outside the block, writes its own line): with a
    // second array for the intermediate sums (44 KB) three workgroups fitted a CU, 768 on the chip — the widest fronts of C4's slab
    // have 946 groups and ran in two rounds; 22 KB lets the four that the thread count allows in (1024)
    __shared__ Q s_q[TE * TE * TE];
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    const uint32_t wv = threadIdx.x / WAVE;
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    Q *qout = reinterpret_cast<Q *>(d_out);
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    const uint32_t ng0 = (p.nb[0] + 1) / 2, ng1 = (p.nb[1] + 1) / 2, ng2 = (p.nb[2] + 1) / 2;
    const uint32_t pair = blockIdx.x;
    if (pair >= npairs) return;
    const uint32_t gz = gz_lo + pair / ng1, gy = pair % ng1;
    if (gz >= ng0 || gz + gy > diag || diag - gz - gy >= ng2) return;  // (workgroup-uniform)
    const uint32_t gx = diag - gz - gy;
    // this wave's block
    const uint32_t lz = wv >> 2, ly = (wv >> 1) & 1u, lx = wv & 1u;
    const uint32_t bz = 2 * gz + lz, by = 2 * gy + ly, bx = 2 * gx + lx;
    const bool live = bz < p.nb[0] && by < p.nb[1] && bx < p.nb[2];
    const uint32_t task = live ? (bz * p.nb[1] + by) * p.nb[2] + bx : 0;
    int sid = 0;
    const BlkGeom g = blk_geom(p, task);
    const TileView tv{TE * TE, TE, (lz * CB) * TE * TE + (ly * CB) * TE + lx * CB};
    // ---- ONE round trip to memory for everything the group needs: the regression blocks' q~ were written by k_blk_pre3 before the
    // fronts started, so an own element is either its delta or its q~ — both are requested, with the block's choice, and the choice
    // picks afterwards; the halo's loads go out in the same batch. (Choice -> rank -> coefficients -> codes, then the halo in a loop
    // of load -> LDS store, were five to ten dependent trips of ~2 us: most of a front's 19.9 us.) ----
    constexpr int OWN = (CB * CB * CB + WAVE - 1) / WAVE;
    constexpr int NH = (TE * TE * TE + 511) / 512;
    Q own[OWN], ownq[OWN], halo[NH];
    const uint32_t nown = live ? g.ez * g.ey * g.ex : 0;
    const int64_t z0 = (int64_t)gz * 2 * CB - 2, y0 = (int64_t)gy * 2 * CB - 2, x0 = (int64_t)gx * 2 * CB - 2;
    {
        const Q *pd[OWN], *pq[OWN], *ph[NH];
#pragma unroll
        for (int k = 0; k < OWN; k++) {
            const uint32_t t = (uint32_t)lane + k * WAVE;
            pd[k] = pq[k] = nullptr;
            if (t < nown) {
                uint32_t i0, i1, i2;
                own_index<CB>(g, t, i0, i1, i2);
                pd[k] = deltas + (g.coff + t);
                pq[k] = qout + (((uint64_t)(g.oz + i0) * d1 + (g.oy + i1)) * d2 + (g.ox + i2));
            }
        }
#pragma unroll
        for (int k = 0; k < NH; k++) {
            const uint32_t t = threadIdx.x + 512u * k;
            const uint32_t tx = t % TE, ty = (t / TE) % TE, tz = t / (TE * TE);
            ph[k] = nullptr;
            if (t < TE * TE * TE && (tz < 2 || ty < 2 || tx < 2)) {
                const int64_t z = z0 + tz, y = y0 + ty, x = x0 + tx;
                if (z >= 0 && y >= 0 && x >= 0 && z < (int64_t)p.d[0] && y < (int64_t)d1 && x < (int64_t)d2)
                    ph[k] = qout + (((uint64_t)z * d1 + (uint64_t)y) * d2 + (uint64_t)x);
            }
        }
        sid = live ? (int)p.sel[task] : 0;
#ifdef LAB_G3_NOLOAD
        if (p.B) {
            for (int k = 0; k < OWN; k++) pd[k] = pq[k] = nullptr;
            for (int k = 0; k < NH; k++) ph[k] = nullptr;
        }
#endif
#pragma unroll
        for (int k = 0; k < OWN; k++) {
            own[k] = *(pd[k] ? pd[k] : deltas);
            ownq[k] = *(pq[k] ? pq[k] : deltas);
        }
#pragma unroll
        for (int k = 0; k < NH; k++) halo[k] = *(ph[k] ? ph[k] : deltas);
#pragma unroll
        for (int k = 0; k < OWN; k++) own[k] = pd[k] ? (sid == 2 ? ownq[k] : own[k]) : (Q)0;
#pragma unroll
        for (int k = 0; k < NH; k++) {
            const uint32_t t = threadIdx.x + 512u * k;
            if (t < TE * TE * TE) s_q[t] = ph[k] ? halo[k] : (Q)0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < OWN; k++) {
        const uint32_t t = (uint32_t)lane + k * WAVE;
        if (t < nown) {
            uint32_t i0, i1, i2;
            own_index<CB>(g, t, i0, i1, i2);
            s_q[tv_at(tv, i0 + 2, i1 + 2, i2 + 2)] = own[k];
        }
    }
    // ---- the Lorenzo blocks, inner front by inner front ----
#ifdef LAB_G3_NOSTEPS
    if (p.B == 0)
#endif
    for (uint32_t step = 0; step < 4; step++) {
        __syncthreads();
        if (live && sid != 2 && lz + ly + lx == step) {
            if (sid == 1) blk_invert<T, CB, 2>(s_q, s_q, qout, g, tv, d1, d2, lane, true);
            else blk_invert<T, CB, 1>(s_q, s_q, qout, g, tv, d1, d2, lane, true);
        }
    }
    // the Lorenzo blocks' lattice values, from the tile (each wave its own block: written by itself, visible to itself)
    wave_lds_fence();
    if (live && sid != 2) {
#pragma unroll
        for (int k = 0; k < OWN; k++) {
            const uint32_t t = (uint32_t)lane + k * WAVE;
            if (t < nown) {
                uint32_t i0, i1, i2;
                own_index<CB>(g, t, i0, i1, i2);
                qout[((uint64_t)(g.oz + i0) * d1 + (g.oy + i1)) * d2 + (g.ox + i2)] = s_q[tv_at(tv, i0 + 2, i1 + 2, i2 + 2)];
            }
        }
    }
}
#endif  // SZ3HIP_LAB

// ---- round 4: a group decoded without recurrences on its critical path -------------------------------------------------------------
// On the lattice a Lorenzo block is linear and exact (integers mod 2^32 / 2^64), so its inverse splits: with D = Dz Dy Dx the block's
// stencil (order m per dimension) and E_d the operator that continues a line's two halo values into the block along dimension d
// (E f(i) = a_m(i) f(-1) + b_m(i) f(-2); first order: a = 1, b = 0; second order: a = i + 2, b = -(i + 1)),
//     q = P + [I - (I - Ez)(I - Ey)(I - Ex)] q,       P = the block inverted with a ZERO halo,
// because (I - E_d) q = D_d^-1 (zero inflow) D_d q along every line and operators of different dimensions commute. P needs nothing but
// the block's own deltas: k_blk_local3 computes it for every block of the array at once, before the fronts. What is left inside the
// chain of fronts is the bracket: 7 (first order) or 26 (second order) halo values with small integer weights per element, no
// dependency between the elements of a block. A group takes its faces first, step by step (the two top layers of a block in every
// dimension are what its upper neighbours read), and all the interiors after the last step. The inner step: 2.4 us of three
// line-scan passes -> a few hundred cycles; which lets a group be 3 x 3 x 3 blocks (seven inner steps, 121 fronts at C4's slab instead
// of 181). Same values bit for bit: both forms are the same sums mod 2^w (tests/test_gpu_regression.py::
// test_grouped_and_per_block_decoders_agree takes all three decoders).
// WORK: the results stay in the work array, in the codes' order (P over the deltas, a regression block's lattice values over its
// slots: what k_blk_wave3 reads), and a regression block's final values go to the output
// codes_in: the deltas were not expanded — a delta is its code - radius, and the work array holds the far deltas only (scattered from
// their list: code 0); only_ragged: the whole blocks were done by k_blk_local3v
template <typename T, int CB, bool WORK>
__global__ __launch_bounds__(256) void k_blk_local3(const uint16_t *__restrict__ codes, void *deltas_, void *d_out, szk_blk_params p, uint32_t nblocks,
                                                    const uint32_t *__restrict__ rank, const int64_t *__restrict__ coef_by_rank, int codes_in = 0,
                                                    int only_ragged = 0) {
    using Q = typename QTraits<T>::Q;
    constexpr uint32_t E = CB + 2;
    __shared__ Q s_q[4][E * E * E];
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    const uint32_t wv = threadIdx.x / WAVE;
    Q *sq = s_q[wv];
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    Q *qout = reinterpret_cast<Q *>(d_out);
    T *tout = reinterpret_cast<T *>(d_out);
    Q *deltas = reinterpret_cast<Q *>(deltas_);
    const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
    for (uint32_t t = lane; t < E * E * E; t += WAVE) sq[t] = 0;  // (the halo stays zero: the passes write own positions only)
    const TileView tv{E * E, E, 0};
    // only_ragged: the blocks of the array's high faces are enumerated directly — the last layer of blocks in z, then the last row in
    // y of the other layers, then the last column in x of what is left (visiting every block to find them was 84 us at C4's slab)
    const bool rz = p.d[0] % p.B != 0, ry = p.d[1] % p.B != 0, rx = p.d[2] % p.B != 0;
    const uint32_t zfree = p.nb[0] - (rz ? 1u : 0u), yfree = p.nb[1] - (ry ? 1u : 0u);
    const uint32_t nA = rz ? p.nb[1] * p.nb[2] : 0u, nB = ry ? zfree * p.nb[2] : 0u, nC = rx ? zfree * yfree : 0u;
    const uint32_t nloop = only_ragged ? nA + nB + nC : nblocks;
    for (uint32_t it = blockIdx.x * 4 + wv; it < nloop; it += gridDim.x * 4) {
        uint32_t task = it;
        if (only_ragged) {
            uint32_t bz, by, bx;
            if (it < nA) {
                bz = p.nb[0] - 1;
                by = it / p.nb[2];
                bx = it % p.nb[2];
            } else if (it < nA + nB) {
                const uint32_t q = it - nA;
                bz = q / p.nb[2];
                by = p.nb[1] - 1;
                bx = q % p.nb[2];
            } else {
                const uint32_t q = it - nA - nB;
                bz = q / yfree;
                by = q % yfree;
                bx = p.nb[2] - 1;
            }
            task = (bz * p.nb[1] + by) * p.nb[2] + bx;
        }
        const int sid = (int)p.sel[task];
        const BlkGeom g = blk_geom(p, task);
        const uint32_t nown = g.ez * g.ey * g.ex;
        if (only_ragged && nown == (uint32_t)(CB * CB * CB)) continue;
        if (sid == 2) {  // regression: the lattice value the neighbours will predict from (k_blk_pre3's part)
            T rc[4];
            coef_recover(coef_by_rank + (uint64_t)rank[task] * 4, cl, rc);
            for (uint32_t t = lane; t < nown; t += WAVE) {
                uint32_t i0, i1, i2;
                own_index<CB>(g, t, i0, i1, i2);
                const uint32_t code = codes[g.coff + t];
                Q qt = 0;
                T val = 0;  // (code 0: patched from the list)
                if (code) {
                    bool bad;
                    val = ref_recover(reg_predict(rc, i0, i1, i2), (int)code, p.eb, (int)p.radius);
                    qt = lat.quant(val, bad);
                    if (bad) qt = 0;
                }
                const uint64_t gi = ((uint64_t)(g.oz + i0) * d1 + (g.oy + i1)) * d2 + (g.ox + i2);
                if (WORK) {
                    deltas[g.coff + t] = qt;
                    tout[gi] = val;
                } else qout[gi] = qt;
            }
            continue;
        }
        for (uint32_t t = lane; t < nown; t += WAVE) {
            uint32_t i0, i1, i2;
            own_index<CB>(g, t, i0, i1, i2);
            Q dl;
            if (codes_in) {
                const uint32_t c = codes[g.coff + t];
                dl = c ? (Q)((int)c - (int)p.radius) : deltas[g.coff + t];
            } else dl = deltas[g.coff + t];
            sq[tv_at(tv, i0 + 2, i1 + 2, i2 + 2)] = dl;
        }
        if (sid == 1) blk_invert<T, CB, 2>(sq, sq, qout, g, tv, d1, d2, lane, WORK);
        else blk_invert<T, CB, 1>(sq, sq, qout, g, tv, d1, d2, lane, WORK);
        wave_lds_fence();
        if (WORK)
            for (uint32_t t = lane; t < nown; t += WAVE) {
                uint32_t i0, i1, i2;
                own_index<CB>(g, t, i0, i1, i2);
                deltas[g.coff + t] = sq[tv_at(tv, i0 + 2, i1 + 2, i2 + 2)];
            }
        wave_lds_fence();
    }
}
// the bracket of the identity above for one element (i0, i1, i2) of the block whose origin sits at tile coordinate (oz, oy, ox); P is
// read from the element's own position
template <int ORDER, typename Q, typename UQ>
__device__ __forceinline__ UQ blk_closed(const Q *s, uint32_t TE, uint32_t oz, uint32_t oy, uint32_t ox, uint32_t i0, uint32_t i1, uint32_t i2) {
    const uint32_t cz[3] = {(oz + i0) * TE * TE, (oz - 1) * TE * TE, (oz - 2) * TE * TE};
    const uint32_t cy[3] = {(oy + i1) * TE, (oy - 1) * TE, (oy - 2) * TE};
    const uint32_t cx[3] = {ox + i2, ox - 1, ox - 2};
    const int wz[3] = {1, ORDER == 2 ? (int)i0 + 2 : 1, -((int)i0 + 1)};
    const int wy[3] = {1, ORDER == 2 ? (int)i1 + 2 : 1, -((int)i1 + 1)};
    const int wx[3] = {1, ORDER == 2 ? (int)i2 + 2 : 1, -((int)i2 + 1)};
    UQ acc = (UQ)s[cz[0] + cy[0] + cx[0]];
#pragma unroll
    for (int a = 0; a <= ORDER; a++)
#pragma unroll
        for (int b = 0; b <= ORDER; b++)
#pragma unroll
            for (int c = 0; c <= ORDER; c++) {
                if ((a | b | c) == 0) continue;
                const int members = (a != 0) + (b != 0) + (c != 0);
                const int w = wz[a] * wy[b] * wx[c] * ((members & 1) ? 1 : -1);
                acc += (UQ)((Q)w * s[cz[a] + cy[b] + cx[c]]);
            }
    return acc;
}
// the same for the element at tile index o whose indices are packed in w3 (i0 << 16 | i1 << 8 | i2): offsets from o instead of
// coordinates (k_blk_wave3, where a lane's element is the same in every block)
template <int ORDER, int TE, typename Q, typename UQ>
__device__ __forceinline__ UQ blk_closed_at(const Q *s, uint32_t o, uint32_t w3) {
    const uint32_t i0 = (w3 >> 16) & 255u, i1 = (w3 >> 8) & 255u, i2 = w3 & 255u;
    const uint32_t hz = (i0 + 1) * TE * TE, hy = (i1 + 1) * TE, hx = i2 + 1;  // (down to the first halo layer)
    if (ORDER == 1) {
        const uint32_t oz = o - hz, oy = o - hy, ozy = oz - hy;
        return (UQ)s[o] + (UQ)s[oz] + (UQ)s[oy] + (UQ)s[o - hx] - (UQ)s[ozy] - (UQ)s[oz - hx] - (UQ)s[oy - hx] + (UQ)s[ozy - hx];
    }
    const uint32_t fz[3] = {0u, hz, hz + TE * TE}, fy[3] = {0u, hy, hy + TE}, fx[3] = {0u, hx, hx + 1u};
    const int wz[3] = {1, (int)i0 + 2, -((int)i0 + 1)}, wy[3] = {1, (int)i1 + 2, -((int)i1 + 1)}, wx[3] = {1, (int)i2 + 2, -((int)i2 + 1)};
    UQ acc = (UQ)s[o];
#pragma unroll
    for (int a = 0; a <= 2; a++)
#pragma unroll
        for (int b = 0; b <= 2; b++)
#pragma unroll
            for (int c = 0; c <= 2; c++) {
                if ((a | b | c) == 0) continue;
                const int members = (a != 0) + (b != 0) + (c != 0);
                const int w = wz[a] * wy[b] * wx[c] * ((members & 1) ? 1 : -1);
                acc += (UQ)((Q)w * s[o - fz[a] - fy[b] - fx[c]]);
            }
    return acc;
}
template <typename T, int CB, int G>
__global__ __launch_bounds__(512, 4) void k_blk_decode_gf(void *d_out, szk_blk_params p, uint32_t diag, uint32_t gz_lo, uint32_t npairs) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    constexpr uint32_t TE = G * CB + 2, NT = TE * TE * TE, NB = G * G * G, CB3 = CB * CB * CB;
    constexpr int NH = (NT + 511) / 512;
    __shared__ Q s_q[NT];
    __shared__ uint8_t s_sel[NB];     // a block's choice; 255: no such block
    __shared__ uint8_t s_list[CB3];   // the elements of a block: faces first, then the interior
    __shared__ uint32_t s_nface;
    __shared__ uint8_t s_order[NB];          // the group's blocks by inner front (lz + ly + lx), ...
    __shared__ uint8_t s_first[3 * G];       // ... and where each front starts in that list
    const int lane = lane_id();
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    Q *qout = reinterpret_cast<Q *>(d_out);
    const uint32_t ng0 = (p.nb[0] + G - 1) / G, ng1 = (p.nb[1] + G - 1) / G, ng2 = (p.nb[2] + G - 1) / G;
    const uint32_t pair = blockIdx.x;
    if (pair >= npairs) return;
    const uint32_t gz = gz_lo + pair / ng1, gy = pair % ng1;
    if (gz >= ng0 || gz + gy > diag || diag - gz - gy >= ng2) return;  // (workgroup-uniform)
    const uint32_t gx = diag - gz - gy;
    const int64_t z0 = (int64_t)gz * G * CB - 2, y0 = (int64_t)gy * G * CB - 2, x0 = (int64_t)gx * G * CB - 2;
    // ---- ONE round trip: every tile position from the array (halo: finished lattice values; own positions: P of a Lorenzo block,
    // the lattice value of a regression block — both written by k_blk_local3), the blocks' choices beside them ----
    Q val[NH];
    uint8_t my_sel = 255;
    if (threadIdx.x < NB) {
        const uint32_t lz = threadIdx.x / (G * G), ly = (threadIdx.x / G) % G, lx = threadIdx.x % G;
        const uint32_t bz = G * gz + lz, by = G * gy + ly, bx = G * gx + lx;
        if (bz < p.nb[0] && by < p.nb[1] && bx < p.nb[2]) my_sel = p.sel[(bz * p.nb[1] + by) * p.nb[2] + bx];
    }
    // A thread's positions t = threadIdx.x + 512 k, walked by increments (the divisions by the tile's edge and the 64-bit products per
    // position were most of a lone group's 8 us of loading and storing). Unconditional loads, the choice afterwards: a load under a
    // branch is waited for before the next one is issued.
    constexpr uint32_t DX = 512u % TE, DY = (512u / TE) % TE, DZ = 512u / (TE * TE);
    const uint32_t zlo = z0 < 0 ? 2u : 0u, ylo = y0 < 0 ? 2u : 0u, xlo = x0 < 0 ? 2u : 0u;
    const uint32_t zn = (uint32_t)min((int64_t)TE, (int64_t)p.d[0] - z0) - zlo, yn = (uint32_t)min((int64_t)TE, (int64_t)d1 - y0) - ylo,
                   xn = (uint32_t)min((int64_t)TE, (int64_t)d2 - x0) - xlo;
    const uint64_t pz = d1 * d2;
    const int64_t g0 = (z0 * (int64_t)d1 + y0) * (int64_t)d2 + x0;
    struct Pos {
        uint32_t tz, ty, tx;
    };
    auto pos0 = [&](uint32_t t) { return Pos{t / (TE * TE), (t / TE) % TE, t % TE}; };
    auto advance = [&](Pos &q) {
        q.tx += DX;
        const uint32_t cx = q.tx >= TE ? 1u : 0u;
        q.tx -= cx * TE;
        q.ty += DY + cx;
        const uint32_t cy = q.ty >= TE ? 1u : 0u;
        q.ty -= cy * TE;
        q.tz += DZ + cy;
    };
    auto inside = [&](const Pos &q) { return q.tz - zlo < zn && q.ty - ylo < yn && q.tx - xlo < xn; };  // (tz >= TE: beyond the tile, zn <= TE - zlo)
    auto gindex = [&](const Pos &q) { return (uint64_t)(g0 + (int64_t)((uint64_t)q.tz * pz + (uint64_t)q.ty * d2 + q.tx)); };
    {
        Pos q = pos0(threadIdx.x);
#pragma unroll
        for (int k = 0; k < NH; k++) {
#if defined(LAB_GF) && LAB_GF == 5
            val[k] = p.B ? (Q)k : qout[inside(q) ? gindex(q) : 0];
#else
            val[k] = qout[inside(q) ? gindex(q) : 0];
#endif
            advance(q);
        }
    }
    // the element order of a block: which of its CB^3 positions are faces (the top `nl` layers of any dimension). One wave compacts.
    const uint32_t nl = (p.mask & 2u) ? 2u : 1u;
    if (wv == 7) {
        uint32_t nf = 0;
        for (int pass = 0; pass < 2; pass++) {
            for (uint32_t t0 = 0; t0 < CB3; t0 += WAVE) {
                const uint32_t t = t0 + lane;
                const uint32_t i2 = t % CB, i1 = (t / CB) % CB, i0 = t / (CB * CB);
                const bool face = i0 + nl >= (uint32_t)CB || i1 + nl >= (uint32_t)CB || i2 + nl >= (uint32_t)CB;
                const bool want = t < CB3 && (pass == 0 ? face : !face);
                const unsigned long long m = __ballot(want);
                if (want) s_list[nf + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)t;
                nf += (uint32_t)__popcll(m);
            }
            if (pass == 0 && lane == 0) s_nface = nf;
        }
    }
    if (threadIdx.x < NB) {
        s_sel[threadIdx.x] = my_sel;
        const uint32_t mine = threadIdx.x / (G * G) + (threadIdx.x / G) % G + threadIdx.x % G;
        uint32_t before = 0;
        for (uint32_t o = 0; o < NB; o++) {
            const uint32_t so = o / (G * G) + (o / G) % G + o % G;
            before += (so < mine || (so == mine && o < threadIdx.x)) ? 1u : 0u;
        }
        s_order[before] = (uint8_t)threadIdx.x;
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + 3 * G) {
        const uint32_t st = threadIdx.x - 64;
        uint32_t before = 0;
        for (uint32_t o = 0; o < NB; o++) before += (o / (G * G) + (o / G) % G + o % G < st) ? 1u : 0u;
        s_first[st] = (uint8_t)before;
    }
    {
        uint32_t t0 = threadIdx.x;
        asm volatile("" : "+v"(t0));  // (walked again, not carried in registers from the loads: sixteen positions' worth of them)
        Pos q = pos0(t0);
#pragma unroll
        for (int k = 0; k < NH; k++) {
            if (q.tz < TE) s_q[t0 + 512u * k] = inside(q) ? val[k] : (Q)0;
            advance(q);
        }
    }
    __syncthreads();

    const uint32_t nface = s_nface;
    const uint32_t frounds = (nface + WAVE - 1) / WAVE, irounds = (CB3 - nface + WAVE - 1) / WAVE;
    // one (block, round of 64 elements) item: the value and where it goes (false: nothing to write). Items are taken two (faces) or
    // four (interiors) at a time, all their reads before the first write: an item alone is a chain of four dependent LDS round trips
    auto gather = [&](uint32_t b, uint32_t first, uint32_t last, uint32_t &dst, UQ &v) {
        const uint32_t lz = b / (G * G), ly = (b / G) % G, lx = b % G;
        const uint32_t sid = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_sel[b]);  // (a scalar: the two orders are branches, not both computed)
        const uint32_t e = first + (uint32_t)lane;
        const uint32_t t = s_list[e < last ? e : first];
        const uint32_t i2 = t % CB, i1 = (t / CB) % CB, i0 = t / (CB * CB);
        const uint32_t oz = lz * CB + 2, oy = ly * CB + 2, ox = lx * CB + 2;
        dst = ((oz + i0) * TE + (oy + i1)) * TE + (ox + i2);
        v = 0;
        if (sid > 1 || e >= last) return false;  // regression (its values are final) / no block / no element
        // (positions beyond a ragged block's extents hold zeros from outside the array: computed like the rest, never stored)
        if (sid == 1) v = blk_closed<2, Q, UQ>(s_q, TE, oz, oy, ox, i0, i1, i2);
        else v = blk_closed<1, Q, UQ>(s_q, TE, oz, oy, ox, i0, i1, i2);
        return true;
    };
    // ---- the faces, inner front by inner front: the (block, round) items of a step go round the eight waves ----
#if defined(LAB_GF) && (LAB_GF == 3 || LAB_GF == 1 || LAB_GF >= 4)  // (lab builds: what a front's time is made of; results wrong)
    if (p.B == 0)
#endif
    for (uint32_t step = 0; step <= 3u * (G - 1); step++) {
        const uint32_t b0 = s_first[step], nitems = (s_first[step + 1] - b0) * frounds;
        for (uint32_t it = wv; it < nitems; it += 16) {
            uint32_t dst[2];
            UQ v[2];
            bool w[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t iu = it + 8u * u;
                const bool have = iu < nitems;
                const uint32_t ic = have ? iu : it, r = ic % frounds;
                w[u] = gather((uint32_t)__builtin_amdgcn_readfirstlane((int)s_order[b0 + ic / frounds]), r * WAVE, min(nface, (r + 1) * WAVE), dst[u], v[u]) && have;
            }
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (w[u]) s_q[dst[u]] = (Q)v[u];
        }
        __syncthreads();
    }
    // ---- the interiors: every block at once ----
#if defined(LAB_GF) && (LAB_GF == 2 || LAB_GF == 1 || LAB_GF >= 4)
    if (p.B == 0)
#endif
    for (uint32_t it = wv; it < NB * irounds; it += 32) {
        uint32_t dst[4];
        UQ v[4];
        bool w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t iu = it + 8u * u;
            const bool have = iu < NB * irounds;
            const uint32_t ic = have ? iu : it, r = ic % irounds;
            w[u] = gather(ic / irounds, nface + r * WAVE, min((uint32_t)CB3, nface + (r + 1) * WAVE), dst[u], v[u]) && have;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (w[u]) s_q[dst[u]] = (Q)v[u];
    }
    __syncthreads();
    // ---- out: the Lorenzo blocks' lattice values, rows of the tile ----
    {
        uint32_t t0 = threadIdx.x;
        asm volatile("" : "+v"(t0));
        Pos q = pos0(t0);
#pragma unroll
        for (int k = 0; k < NH; k++) {
            if (inside(q) && q.tz >= 2 && q.ty >= 2 && q.tx >= 2) {
                const uint32_t b = (((q.tz - 2) / CB) * G + (q.ty - 2) / CB) * G + (q.tx - 2) / CB;
#if defined(LAB_GF) && LAB_GF == 4
                if (s_sel[b] <= 1 && s_q[t0 + 512u * k] == 0x5a5a5a5a) qout[gindex(q)] = s_q[t0 + 512u * k];
#else
                if (s_sel[b] <= 1) qout[gindex(q)] = s_q[t0 + 512u * k];
#endif
            }
            advance(q);
        }
    }
}

// k_blk_local3<WORK> for whole blocks, sixteen at a time: a wave per block left 28 of 64 lanes idle in every line pass (36 lines of six)
// and spent most of its instructions on halo terms that are zero here. A workgroup of 192 threads takes 16 consecutive blocks (3456
// values, 576 lines per pass: three per thread), straight from the codes (a delta is code - radius; code 0: the far delta, scattered
// into the work array from its list beforehand — no expanded copy of the deltas is made), runs the three passes as plain running sums
// (first order) or second-order recurrences with zero inflow, by the block's choice, and leaves P in the work array; a regression
// block's values are computed element by element. Ragged blocks (the array's high faces) are left to k_blk_local3(only_ragged).
#define BLK3V_NB 16
template <typename T, int CB>
__global__ __launch_bounds__(192, 4) void k_blk_local3v(const uint16_t *__restrict__ codes, void *work_, void *d_out, szk_blk_params p, uint32_t nblocks,
                                                     const uint32_t *__restrict__ rank, const int64_t *__restrict__ coef_by_rank) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    constexpr uint32_t CB3 = CB * CB * CB, NL = CB * CB, PER = BLK3V_NB * CB3 / 192;  // (18 values per thread)
    static_assert(BLK3V_NB * CB3 % 192 == 0 && BLK3V_NB * NL % 192 == 0, "a workgroup's values and lines divide among its threads");
    __shared__ Q s_v[BLK3V_NB * CB3];
    __shared__ uint64_t s_coff[BLK3V_NB];
    __shared__ uint32_t s_org[BLK3V_NB][3];
    __shared__ uint8_t s_kind[BLK3V_NB];  // 0 / 1: first- / second-order Lorenzo, 2: regression, 255: not a whole block (skipped here)
    const Lattice<T> lat(p.lat);
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    Q *work = reinterpret_cast<Q *>(work_);
    T *tout = reinterpret_cast<T *>(d_out);
    const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
    const uint32_t nbatch = (nblocks + BLK3V_NB - 1) / BLK3V_NB;
    for (uint32_t batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        __syncthreads();
        if (threadIdx.x < BLK3V_NB) {
            const uint32_t task = batch * BLK3V_NB + threadIdx.x;
            uint8_t kind = 255;
            if (task < nblocks) {
                const BlkGeom g = blk_geom(p, task);
                if (g.ez * g.ey * g.ex == CB3) {
                    const uint8_t sl = p.sel[task];
                    kind = sl == 2 ? 2 : (sl == 1 ? 1 : 0);
                }
                s_coff[threadIdx.x] = g.coff;
                s_org[threadIdx.x][0] = g.oz;
                s_org[threadIdx.x][1] = g.oy;
                s_org[threadIdx.x][2] = g.ox;
            }
            s_kind[threadIdx.x] = kind;
        }
        __syncthreads();
        // ---- in: codes -> deltas (all of a thread's loads before the first use) ----
        uint16_t c[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t e = threadIdx.x + 192u * k, b = e / CB3, t = e - b * CB3;
            c[k] = s_kind[b] != 255 ? codes[s_coff[b] + t] : (uint16_t)1;
        }
        uint32_t tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));  // (positions computed again, not carried in registers from loop to loop: eighteen of them)
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t e = tid2 + 192u * k, b = e / CB3, t = e - b * CB3;
            Q dl = (Q)((int)c[k] - (int)p.radius);
            if (c[k] == 0 && s_kind[b] < 2) dl = work[s_coff[b] + t];  // (a far delta: rare)
            s_v[e] = dl;
        }
        __syncthreads();
        // ---- the three passes: x (contiguous lines), y, z ----
#pragma unroll
        for (int pass = 0; pass < 3; pass++) {
            uint32_t tidp = threadIdx.x;
            asm volatile("" : "+v"(tidp));
#pragma unroll
            for (uint32_t k = 0; k < BLK3V_NB * NL / 192; k++) {
                const uint32_t l = tidp + 192u * k, b = l / NL, r = l - b * NL, a0 = r / CB, a1 = r - a0 * CB;
                const uint32_t kind = s_kind[b];
                if (kind > 1) continue;
                // first element and stride of the line
                const uint32_t base = b * CB3 + (pass == 0 ? (a0 * CB + a1) * CB : (pass == 1 ? a0 * CB * CB + a1 : a0 * CB + a1));
                const uint32_t stride = pass == 0 ? 1u : (pass == 1 ? (uint32_t)CB : (uint32_t)(CB * CB));
                UQ in[CB];
#pragma unroll
                for (uint32_t i = 0; i < (uint32_t)CB; i++) in[i] = (UQ)s_v[base + i * stride];
                UQ p1 = 0, p2 = 0;
#pragma unroll
                for (uint32_t i = 0; i < (uint32_t)CB; i++) {
                    const UQ v = kind == 1 ? (UQ)(2 * p1 - p2 + in[i]) : (UQ)(p1 + in[i]);
                    in[i] = v;
                    p2 = p1;
                    p1 = v;
                }
#pragma unroll
                for (uint32_t i = 0; i < (uint32_t)CB; i++) s_v[base + i * stride] = (Q)in[i];
            }
            __syncthreads();
        }
        // ---- the regression blocks: lattice values for the neighbours (into the tile), final values into the array ----
        for (uint32_t b = 0; b < BLK3V_NB; b++) {
            if (s_kind[b] != 2) continue;  // (workgroup-uniform)
            T rc[4];
            coef_recover(coef_by_rank + (uint64_t)rank[batch * BLK3V_NB + b] * 4, cl, rc);
            for (uint32_t t = threadIdx.x; t < CB3; t += 192) {
                const uint32_t i2 = t % CB, i1 = (t / CB) % CB, i0 = t / (CB * CB);
                const uint32_t code = codes[s_coff[b] + t];
                Q qt = 0;
                T val = 0;  // (code 0: patched from the list)
                if (code) {
                    bool bad;
                    val = ref_recover(reg_predict(rc, i0, i1, i2), (int)code, p.eb, (int)p.radius);
                    qt = lat.quant(val, bad);
                    if (bad) qt = 0;
                }
                s_v[b * CB3 + t] = qt;
                tout[((uint64_t)(s_org[b][0] + i0) * d1 + (s_org[b][1] + i1)) * d2 + (s_org[b][2] + i2)] = val;
            }
        }
        __syncthreads();
        // ---- out: P / the lattice values, in the codes' order ----
        uint32_t tid3 = threadIdx.x;
        asm volatile("" : "+v"(tid3));
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t e = tid3 + 192u * k, b = e / CB3, t = e - b * CB3;
            if (s_kind[b] != 255) work[s_coff[b] + t] = s_v[e];
        }
    }
}

// ---- round 4: the whole chain of fronts in ONE launch ------------------------------------------------------------------------------
// A launch per front costs ~9 us whatever the front does (120 of them: 1.1 ms of the C4a slab's decompression, and the wide fronts wait
// for their slowest group). Here the groups are handed out by a ticket counter in the fronts' order (slot = position in the
// enumeration front by front; a slot without a group is skipped) to workgroups that stay: a group waits for the flags of its seven
// lower neighbours, and every group it can wait for holds a smaller ticket — taken by a workgroup that is running — so the waits end
// (the poll is bounded all the same: ctl[1] is raised should a flag never arrive, the host fetches the word behind the launch and sends
// the stream through the launch-per-front decoders instead: szk_launch_blk_decompress). What travels
// between groups is a group's outer shell (the two top layers of lattice values in every dimension), written over its P in the work
// array, in the codes' order, released with the group's flag; everything else a group computes goes out as final values (no
// k_blk_final pass over the array). k_blk_local3<WORK> ran before: P of every Lorenzo block and the lattice values of the regression
// blocks are in the work array, the regression blocks' final values in the output.
// ctl: [0] ticket, [1] a wait gave up, [4 ...] the groups' flags (zeroed by the caller).
// Memory order of the exchange — a contract with THIS target, not the HIP memory model's release / acquire pair (which costs a write-back
// of the L2's dirty lines per group: 16 ms, see below): a shell travels through relaxed agent-scope atomic stores and loads, which on
// gfx950 go past the per-XCD L2s to memory, the flag's store is issued behind s_waitcnt vmcnt(0) of the shell's stores (in-order return
// on one counter), and a reader's shell loads are issued behind its flag load's return. Another target must revisit this.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_blk_wave3 / k_blkn_wave2: the inter-workgroup shell exchange is written for gfx950's memory path (see the comment above)"
#endif
template <typename T, int CB, int G, int NL>  // NL: the layers of a block that are faces — 2 when the set holds second-order Lorenzo
__global__ __launch_bounds__(512, 4) void k_blk_wave3(void *work_, void *d_out, szk_blk_params p, uint32_t *ctl, uint32_t nslots) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    constexpr uint32_t TE = G * CB + 2, NT = TE * TE * TE, NB = G * G * G, CB3 = CB * CB * CB;
    constexpr int NH = (NT + 511) / 512;
    constexpr uint32_t HZ = 2 * TE * TE, HY = (TE - 2) * 2 * TE, HX = (TE - 2) * (TE - 2) * 2, NHALO = HZ + HY + HX;
    constexpr int NHL = (NHALO + 511) / 512;
    constexpr uint32_t NFACE = CB3 - (CB - NL) * (CB - NL) * (CB - NL), FR = (NFACE + WAVE - 1) / WAVE, IR = (CB3 - NFACE + WAVE - 1) / WAVE;
    __shared__ Q s_q[NT];
    __shared__ uint8_t s_sel[NB];
    __shared__ uint8_t s_list[CB3];
    __shared__ uint32_t s_nface, s_ticket;
    __shared__ uint8_t s_order[NB];
    __shared__ uint8_t s_first[3 * G];
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    Q *work = reinterpret_cast<Q *>(work_);
    T *tout = reinterpret_cast<T *>(d_out);
    uint32_t *flags = ctl + 4;
    const uint32_t ng0 = (p.nb[0] + G - 1) / G, ng1 = (p.nb[1] + G - 1) / G, ng2 = (p.nb[2] + G - 1) / G;
    constexpr uint32_t nl = NL;
    // ---- once: the element order of a block (faces first), the blocks of a group by inner front ----
    if (wv == 7) {
        uint32_t nf = 0;
        for (int pass = 0; pass < 2; pass++) {
            for (uint32_t t0 = 0; t0 < CB3; t0 += WAVE) {
                const uint32_t t = t0 + lane;
                const uint32_t i2 = t % CB, i1 = (t / CB) % CB, i0 = t / (CB * CB);
                const bool face = i0 + nl >= (uint32_t)CB || i1 + nl >= (uint32_t)CB || i2 + nl >= (uint32_t)CB;
                const bool want = t < CB3 && (pass == 0 ? face : !face);
                const unsigned long long m = __ballot(want);
                if (want) s_list[nf + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)t;
                nf += (uint32_t)__popcll(m);
            }
            if (pass == 0 && lane == 0) s_nface = nf;
        }
    }
    if (threadIdx.x < NB) {
        const uint32_t mine = threadIdx.x / (G * G) + (threadIdx.x / G) % G + threadIdx.x % G;
        uint32_t before = 0;
        for (uint32_t o = 0; o < NB; o++) {
            const uint32_t so = o / (G * G) + (o / G) % G + o % G;
            before += (so < mine || (so == mine && o < threadIdx.x)) ? 1u : 0u;
        }
        s_order[before] = (uint8_t)threadIdx.x;
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + 3 * G) {
        const uint32_t st = threadIdx.x - 64;
        uint32_t before = 0;
        for (uint32_t o = 0; o < NB; o++) before += (o / (G * G) + (o / G) % G + o % G < st) ? 1u : 0u;
        s_first[st] = (uint8_t)before;
    }
    __syncthreads();
    // What a lane computes in round r of a block (rounds 0 .. FR - 1: the faces, then IR rounds of interior) is the same element of
    // every block: its offset inside the block's part of the tile and its indices are worked out once (the element list, the divisions
    // and the products per item were most of this kernel's instructions — it is bound by their issue, not by memory: 351 instructions
    // per element before)
    uint32_t rel[FR + IR], ijk[FR + IR];
#pragma unroll
    for (uint32_t r = 0; r < FR + IR; r++) {
        const uint32_t e = r < FR ? r * WAVE + (uint32_t)lane : NFACE + (r - FR) * WAVE + (uint32_t)lane;
        const bool valid = r < FR ? e < NFACE : e < CB3;
        const uint32_t t = s_list[valid ? e : 0];
        const uint32_t i2 = t % CB, i1 = (t / CB) % CB, i0 = t / (CB * CB);
        rel[r] = (i0 * TE + i1) * TE + i2;
        ijk[r] = (valid ? 0x80000000u : 0u) | (i0 << 16) | (i1 << 8) | i2;
    }
    // ... and so is the k-th value a lane moves in or out of a whole block (t = lane + 64 k in the block's raster order)
    constexpr int OWNK = (CB3 + WAVE - 1) / WAVE;
    uint32_t orel[OWNK], oijk[OWNK];
#pragma unroll
    for (int k = 0; k < OWNK; k++) {
        const uint32_t t = (uint32_t)lane + k * WAVE;
        const uint32_t i2 = t % CB, i1 = (t / CB) % CB, i0 = t / (CB * CB);
        orel[k] = (i0 * TE + i1) * TE + i2;
        oijk[k] = (t < CB3 ? 0x80000000u : 0u) | (i0 << 16) | (i1 << 8) | i2;
    }
    // one (block b, round r) item: the value and where it goes (false: nothing to write)
    auto gather = [&](uint32_t b, uint32_t relr, uint32_t w3, uint32_t &dst, UQ &v) {
        asm volatile("" : "+v"(relr), "+v"(w3));  // (the offsets and weights that follow from them are a few instructions: computed here, not kept — for
                                                     // every round and both orders they were two hundred values per lane in scratch memory)
        const uint32_t lz = b / (G * G), ly = (b / G) % G, lx = b % G;
        const uint32_t sid = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_sel[b]);
        dst = ((lz * CB + 2) * TE + (ly * CB + 2)) * TE + (lx * CB + 2) + relr;
        v = 0;
        if (sid > 1 || !(w3 >> 31)) return false;  // regression (its values are final) / no block / no element
        if (sid == 1) v = blk_closed_at<2, (int)TE, Q, UQ>(s_q, dst, w3);
        else v = blk_closed_at<1, (int)TE, Q, UQ>(s_q, dst, w3);
        return true;
    };
    // the fronts' enumeration: front dd = the groups with gz + gy + gx = dd, by gz, then gy (row_groups: how many for one gz)
    auto row_groups = [&](uint32_t dd, uint32_t gz, uint32_t &gy_lo) {
        const uint32_t r = dd - gz;  // gy + gx
        gy_lo = r > ng2 - 1 ? r - (ng2 - 1) : 0;
        const uint32_t gy_hi = r < ng1 - 1 ? r : ng1 - 1;
        return gy_lo > gy_hi ? 0u : gy_hi - gy_lo + 1;
    };
    auto front_groups = [&](uint32_t dd, uint32_t &gz_lo, uint32_t &gz_hi) {
        const uint32_t rest = (ng1 - 1) + (ng2 - 1);
        gz_lo = dd > rest ? dd - rest : 0;
        gz_hi = dd < ng0 - 1 ? dd : ng0 - 1;
        uint32_t c = 0, gy_lo;
        for (uint32_t z = gz_lo; z <= gz_hi && gz_lo <= gz_hi; z++) c += row_groups(dd, z, gy_lo);
        return c;
    };
    uint32_t dcur = 0, start = 0;  // this workgroup's place in the enumeration: its tickets only grow
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ctl[0], 1u);
    for (;;) {
        // (the thread's and the lane's number afresh in every round: what is computed from them stays inside the round — hoisted out of
        // this loop it was several hundred values per lane kept in scratch memory)
        uint32_t tid = threadIdx.x, ln = (uint32_t)lane;
        asm volatile("" : "+v"(tid), "+v"(ln));
        __syncthreads();  // (the ticket — taken while the previous group was being finished — is there; the tile is free again)
        const uint32_t ticket = s_ticket;
        if (ticket >= nslots) break;
        uint32_t gz_lo, gz_hi, cnt;
        while (ticket >= start + (cnt = front_groups(dcur, gz_lo, gz_hi))) {
            start += cnt;
            dcur++;
        }
        uint32_t rest = ticket - start, gz = gz_lo, gy_lo, rc;
        while (rest >= (rc = row_groups(dcur, gz, gy_lo))) {
            rest -= rc;
            gz++;
        }
        const uint32_t gy = gy_lo + rest, gx = dcur - gz - gy;
        const int64_t z0 = (int64_t)gz * G * CB - 2, y0 = (int64_t)gy * G * CB - 2, x0 = (int64_t)gx * G * CB - 2;
        // ---- (1) nothing to wait for yet: the blocks' choices, their P / lattice values from the work array (a wave per block) ----
        if (tid < NB) {
            const uint32_t lz = tid / (G * G), ly = (tid / G) % G, lx = tid % G;
            const uint32_t bz = G * gz + lz, by = G * gy + ly, bx = G * gx + lx;
            uint8_t sl = 255;
            if (bz < p.nb[0] && by < p.nb[1] && bx < p.nb[2]) {
                sl = p.sel[(bz * p.nb[1] + by) * p.nb[2] + bx];
                sl = sl == 2 ? 2 : (sl == 1 ? 1 : 0);  // (a crafted 3 is read like k_blk_local3 reads it)
            }
            s_sel[tid] = sl;
        }
        {   // (every block's loads in flight before the first use: a block after the other was four dependent trips to memory)
            constexpr int OWN = (CB3 + WAVE - 1) / WAVE, NBW = (NB + 7) / 8;
            Q own[NBW][OWN];
#pragma unroll
            for (int j = 0; j < NBW; j++) {
                const uint32_t b = wv + 8u * j;
                const uint32_t lz = b / (G * G), ly = (b / G) % G, lx = b % G;
                const uint32_t bz = G * gz + lz, by = G * gy + ly, bx = G * gx + lx;
                const bool have = b < NB && bz < p.nb[0] && by < p.nb[1] && bx < p.nb[2];
                const BlkGeom g = blk_geom_at(p, have ? bz : 0, have ? by : 0, have ? bx : 0);
                const uint32_t nown = have ? g.ez * g.ey * g.ex : 0;
#pragma unroll
                for (int k = 0; k < OWN; k++) {
                    const uint32_t t = ln + k * WAVE;
                    own[j][k] = work[g.coff + (t < nown ? t : 0)];
                }
            }
#pragma unroll
            for (int j = 0; j < NBW; j++) {
                uint32_t b = wv + 8u * j;
                asm volatile("" : "+s"(b));  // (the geometry again, not carried in registers across the loads)
                const uint32_t lz = b / (G * G), ly = (b / G) % G, lx = b % G;
                const uint32_t bz = G * gz + lz, by = G * gy + ly, bx = G * gx + lx;
                const bool have = b < NB && bz < p.nb[0] && by < p.nb[1] && bx < p.nb[2];
                const BlkGeom g = blk_geom_at(p, have ? bz : 0, have ? by : 0, have ? bx : 0);
                const uint32_t nown = have ? g.ez * g.ey * g.ex : 0;
                const uint32_t base = ((lz * CB + 2) * TE + (ly * CB + 2)) * TE + (lx * CB + 2);
                if (nown == CB3) {  // (a whole block: the lane's places are known)
#pragma unroll
                    for (int k = 0; k < OWN; k++)
                        if (oijk[k] >> 31) s_q[base + orel[k]] = own[j][k];
                } else {
#pragma unroll
                    for (int k = 0; k < OWN; k++) {
                        const uint32_t t = ln + k * WAVE;
                        if (t < nown) {
                            uint32_t i0, i1, i2;
                            own_index<CB>(g, t, i0, i1, i2);
                            s_q[base + (i0 * TE + i1) * TE + i2] = own[j][k];
                        }
                    }
                }
            }
        }
        // ---- (2) the seven lower neighbours' flags ----
#if defined(LAB_W) && LAB_W == 4  // (lab builds: what the launch's time is made of; results wrong)
        if (p.B == 0)
#endif
        if (tid < 7) {
            const uint32_t k = tid + 1, dz = k >> 2, dy = (k >> 1) & 1u, dx = k & 1u;
            if (gz >= dz && gy >= dy && gx >= dx) {
                const uint32_t *f = flags + ((uint64_t)(gz - dz) * ng1 + (gy - dy)) * ng2 + (gx - dx);
                uint32_t spins = 0;
                while (__hip_atomic_load(const_cast<uint32_t *>(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1u << 22)) {
                        atomicExch(&ctl[1], 1u);
                        break;
                    }
                }
            }
        }
        __syncthreads();
        // (no cache-wide acquire / release around the exchange — a write-back of the L2's dirty lines per wave and group, with the groups'
        // output streaming through that L2, made this kernel 16 ms: what travels between groups goes through agent-scope atomic stores
        // and loads, which pass the L2 by, and the flag follows them after s_waitcnt vmcnt(0))
        // ---- (3) the halo: the neighbours' shells, from the work array (position -> block -> its place in the codes' order) ----
        {
            Q hv[NHL];
            uint32_t hpos[NHL];
            bool hin[NHL];
#pragma unroll
            for (int k = 0; k < NHL; k++) {
                const uint32_t h = tid + 512u * k;
                uint32_t tz, ty, tx;
                if (h < HZ) {
                    tz = h / (TE * TE);
                    ty = (h / TE) % TE;
                    tx = h % TE;
                } else if (h < HZ + HY) {
                    const uint32_t r = h - HZ;
                    tz = 2 + r / (2 * TE);
                    ty = (r / TE) % 2;
                    tx = r % TE;
                } else {
                    const uint32_t r = h - HZ - HY;
                    tz = 2 + r / ((TE - 2) * 2);
                    ty = 2 + (r / 2) % (TE - 2);
                    tx = r % 2;
                }
                const int64_t z = z0 + tz, y = y0 + ty, x = x0 + tx;
                const bool in = h < NHALO && z >= 0 && y >= 0 && x >= 0 && z < (int64_t)p.d[0] && y < (int64_t)d1 && x < (int64_t)d2;
                uint64_t addr = 0;
                if (in) {
                    const uint32_t bz = (uint32_t)z / CB, by = (uint32_t)y / CB, bx = (uint32_t)x / CB;
                    const BlkGeom g = blk_geom_at(p, bz, by, bx);
                    addr = g.coff + ((uint64_t)((uint32_t)z - g.oz) * g.ey + ((uint32_t)y - g.oy)) * g.ex + ((uint32_t)x - g.ox);
                }
                hin[k] = in;
                hpos[k] = h < NHALO ? (tz * TE + ty) * TE + tx : 0xFFFFFFFFu;
                hv[k] = __hip_atomic_load(work + addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < NHL; k++)
                if (hpos[k] != 0xFFFFFFFFu) s_q[hpos[k]] = hin[k] ? hv[k] : (Q)0;
        }
        __syncthreads();
        // ---- (4) the faces, inner front by inner front: item (round r, j-th block of the front) goes to wave (r * blocks + j) mod 8 ----
#if defined(LAB_W) && LAB_W == 3
        if (p.B == 0)
#endif
        for (uint32_t step = 0; step <= 3u * (G - 1); step++) {
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_first[step]);
            const uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_first[step + 1]) - b0;  // (at most 7 < 8: one block per round and wave)
            uint32_t dst[FR];
            UQ v[FR];
            bool w[FR];
#pragma unroll
            for (uint32_t r = 0; r < FR; r++) {
                const uint32_t j = (wv + 32u - r * nb) & 7u;
                const bool have = j < nb;
                const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_order[b0 + (have ? j : 0u)]);
                w[r] = gather(b, rel[r], ijk[r], dst[r], v[r]) && have;
            }
#pragma unroll
            for (uint32_t r = 0; r < FR; r++)
                if (w[r]) s_q[dst[r]] = (Q)v[r];
            __syncthreads();
        }
        // ---- (5) the shell out, then the flag: the upper neighbours may go ----
        for (uint32_t b = wv; b < NB; b += 8) {
            const uint32_t lz = b / (G * G), ly = (b / G) % G, lx = b % G;
            if (lz != G - 1 && ly != G - 1 && lx != G - 1) continue;
            if (s_sel[b] > 1) continue;  // (a regression block's lattice values are in the work array already)
            const BlkGeom g = blk_geom_at(p, G * gz + lz, G * gy + ly, G * gx + lx);
            const uint32_t nown = g.ez * g.ey * g.ex;
            const uint32_t base = ((lz * CB + 2) * TE + (ly * CB + 2)) * TE + (lx * CB + 2);
            if (nown == CB3) {
#pragma unroll
                for (int k = 0; k < OWNK; k++) {
                    const uint32_t w3 = oijk[k], i0 = (w3 >> 16) & 255u, i1 = (w3 >> 8) & 255u, i2 = w3 & 255u;
                    const bool shell = (lz == G - 1 && i0 + nl >= (uint32_t)CB) || (ly == G - 1 && i1 + nl >= (uint32_t)CB) || (lx == G - 1 && i2 + nl >= (uint32_t)CB);
                    if ((w3 >> 31) && shell) __hip_atomic_store(work + g.coff + (ln + k * WAVE), s_q[base + orel[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else
            for (uint32_t t = (int)ln; t < nown; t += WAVE) {
                uint32_t i0, i1, i2;
                own_index<CB>(g, t, i0, i1, i2);
                const bool shell = (lz == G - 1 && i0 + nl >= (uint32_t)CB) || (ly == G - 1 && i1 + nl >= (uint32_t)CB) || (lx == G - 1 && i2 + nl >= (uint32_t)CB);
                if (shell) __hip_atomic_store(work + g.coff + t, s_q[base + (i0 * TE + i1) * TE + i2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(flags + ((uint64_t)gz * ng1 + gy) * ng2 + gx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ticket = atomicAdd(&ctl[0], 1u);  // (the next group's ticket: on its way while this one's interiors and output are done — everybody read the last one long ago)
        }
        // ---- (6) the interiors: item (round q, block b) goes to wave (q * 27 + b) mod 8 ----
#if defined(LAB_W) && LAB_W == 1
        if (p.B == 0)
#endif
        {
            constexpr uint32_t IB = sizeof(Q) == 8 ? 2 : 4;  // (items in flight: registers)
#pragma unroll
            for (uint32_t q = 0; q < IR; q++) {
                const uint32_t bfirst = (wv + 64u - ((q * NB) & 63u)) & 7u;
#pragma unroll 1
                for (uint32_t k0 = 0; k0 < (NB + 7) / 8; k0 += IB) {
                    uint32_t dst[IB];
                    UQ v[IB];
                    bool w[IB];
#pragma unroll
                    for (uint32_t u = 0; u < IB; u++) {
                        const uint32_t b = bfirst + 8u * (k0 + u);
                        const bool have = b < NB;
                        w[u] = gather(have ? b : bfirst, rel[FR + q], ijk[FR + q], dst[u], v[u]) && have;
                    }
#pragma unroll
                    for (uint32_t u = 0; u < IB; u++)
                        if (w[u]) s_q[dst[u]] = (Q)v[u];
                }
            }
        }
        __syncthreads();
        // ---- (7) out: the Lorenzo blocks' final values (the regression blocks' were written by k_blk_local3). A thread keeps its (y, x) of
        // the tile and walks z: the position, the block's choice and the address are per-thread constants or increments ----
#if defined(LAB_W) && (LAB_W == 1 || LAB_W == 2)
        if (p.B == 0)
#endif
        if (tid < TE * TE) {
            const uint32_t ty = tid / TE, tx = tid % TE;
            const uint32_t zn = (uint32_t)min((int64_t)TE, (int64_t)p.d[0] - z0), yn = (uint32_t)min((int64_t)TE, (int64_t)d1 - y0),
                           xn = (uint32_t)min((int64_t)TE, (int64_t)d2 - x0);
            if (ty >= 2 && tx >= 2 && ty < yn && tx < xn) {
                const uint32_t byx = ((ty - 2) / CB) * G + (tx - 2) / CB;
                const uint64_t pz = d1 * d2;
                T *o = tout + ((z0 + 2) * (int64_t)d1 + (y0 + (int64_t)ty)) * (int64_t)d2 + (x0 + (int64_t)tx);
#pragma unroll
                for (uint32_t lz = 0; lz < (uint32_t)G; lz++) {
                    const bool mine = s_sel[lz * G * G + byx] <= 1;
#pragma unroll
                    for (uint32_t i = 0; i < (uint32_t)CB; i++) {
                        const uint32_t tz = 2 + lz * CB + i;
                        if (mine && tz < zn) o[(uint64_t)(lz * CB + i) * pz] = lat.dequant(s_q[tz * TE * TE + tid]);
                    }
                }
            }
        }
    }
}

// final pass: lattice value -> T for Lorenzo blocks; regression blocks are recomputed from their codes (their value is
// pred + 2*code*eb, not a lattice point). (Measured and dropped, round 3: the same pass by rows of the array — a thread per x, whole
// rows read and written instead of segments of B values — 831 against 807 us at C4a's slab.)
template <typename T, int CB>
__global__ __launch_bounds__(256) void k_blk_final(const uint16_t *__restrict__ codes, void *d_out, szk_blk_params p, uint32_t nblocks,
                                                   const uint32_t *__restrict__ rank, const int64_t *__restrict__ coef_by_rank) {
    using Q = typename QTraits<T>::Q;
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    Q *qv = reinterpret_cast<Q *>(d_out);
    T *ov = reinterpret_cast<T *>(d_out);
    const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
    for (uint32_t task = blockIdx.x * 4 + threadIdx.x / WAVE; task < nblocks; task += gridDim.x * 4) {
        const BlkGeom g = blk_geom(p, task);
        const uint32_t nown = g.ez * g.ey * g.ex;
        const bool reg = p.sel[task] == 2;
        T rc[4] = {0, 0, 0, 0};
        if (reg) coef_recover(coef_by_rank + (uint64_t)rank[task] * 4, cl, rc);
        for (uint32_t t = lane; t < nown; t += WAVE) {
            uint32_t i0, i1, i2;
            own_index<CB>(g, t, i0, i1, i2);
            const uint64_t gi = ((uint64_t)(g.oz + i0) * d1 + (g.oy + i1)) * d2 + (g.ox + i2);
            if (reg) {
                const uint32_t code = codes[g.coff + t];
                ov[gi] = code ? ref_recover(reg_predict(rc, i0, i1, i2), (int)code, p.eb, (int)p.radius) : (T)0;  // (code 0: patched from the list)
            } else {
                const Q q = qv[gi];
                ov[gi] = lat.dequant(q);
            }
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_blk_patch(const uint8_t *__restrict__ payload, uint64_t idx_off, uint64_t val_off, uint64_t cnt, uint64_t n,
                                                   T *__restrict__ out) {
    const uint64_t *idx = reinterpret_cast<const uint64_t *>(payload + idx_off);
    const T *val = reinterpret_cast<const T *>(payload + val_off);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (uint64_t)gridDim.x * 256) {
        const uint64_t k = idx[i];
        if (k < n) out[k] = val[i];
    }
}

// ------------------------------------------------------------------------------------------------------------
// arrays of one and two dimensions (default blocks: 128 values / 16 x 16, Config.hpp:175)
// ------------------------------------------------------------------------------------------------------------
// The kernels see the array as d = (1, 1, n) or (1, dy, dx): the block walk, the code order (block by block, raster order inside),
// the side section and the final pass are the 3-D path's with extents of one; a stencil term over a unit dimension reads the
// zero halo and drops out, so the integer Lorenzo stencil IS the 1-D / 2-D one. What the dimension count changes: the fit's
// sums and the coefficient lattices' steps (eb / (N + 1) [/ B], RegressionPredictor.hpp:22-26, 28-55), the selection's sample
// points (BlockwiseIterator.hpp:151-184: a block's two ends in 1-D, the two diagonals in 2-D) and its Lorenzo noise term
// (LorenzoPredictor.hpp:17-38: 0.5 eb / 0.81 eb) — and the decoder: blocks of 128 values do not fit the 3-D path's tiles, a chain
// of 1-D blocks is not a front, so 1-D is a segmented prefix sum over the blocks (a regression block restarts the sum with the
// lattice value of its last element) and 2-D runs anti-diagonal fronts of blocks with the block in LDS.
// Encoder: k_blkn_fit (wave per block: fit, estimates, choice, regression blocks coded, q~ of every element to qwork), then
// k_blkn_lorenzo (thread per CODE position: the stencil over q~, codes written where they lie — coalesced).

// what the selection's Lorenzo estimate sees at (y, x): original inside the block, lattice reconstruction outside, zero outside the array
template <typename T>
__device__ __forceinline__ T blkn_seen(const T *__restrict__ in, const szk_blk_params &p, const Lattice<T> &lat, const BlkGeom &g, int64_t y, int64_t x) {
    using Q = typename QTraits<T>::Q;
    if (y < 0 || x < 0) return (T)0;
    T v = in[(uint64_t)y * p.d[2] + (uint64_t)x];
    if (y < (int64_t)g.oy || x < (int64_t)g.ox) {
        bool bad;
        const Q qh = lat.quant(v, bad);
        if (!bad) v = lat.dequant(qh);
    }
    return v;
}

// position t of a block's raster order -> (row, column); 1-D blocks have one row, whole 2-D blocks of the default edge shift
template <bool TWO>
__device__ __forceinline__ void blkn_own(const BlkGeom &g, uint32_t t, uint32_t &i1, uint32_t &i2) {
    if (!TWO) {
        i1 = 0;
        i2 = t;
    } else {
        i1 = g.ex == 16 ? t >> 4 : t / g.ex;
        i2 = t - i1 * g.ex;
    }
}
// 1-D, a set of Lorenzo members only ({L1, L2}: what the reference's tuner takes for a 1-D array; {L2}): the choice needs the block's two
// ends and the two values left of it — a thread per block, seven loads, no fit and no work array: the code pass then reads the array
// itself (its `direct` form, p.sel_given)
template <typename T>
__global__ __launch_bounds__(256) void k_blkn_sel12(const T *__restrict__ in, szk_blk_params p, uint32_t nblocks) {
    const Lattice<T> lat(p.lat);
    const T n1 = (T)(0.5 * p.eb), n2 = (T)(1.08 * p.eb);
    const uint32_t n = (uint32_t)p.d[2];
    for (uint32_t task = blockIdx.x * 256 + threadIdx.x; task < nblocks; task += gridDim.x * 256) {
        uint8_t sid = 1;
        if (p.mask & 1u) {
            BlkGeom g;
            g.oy = g.oz = 0;
            g.ox = task * p.B;
            g.ex = min(p.B, n - g.ox);
            double e1 = 0, e2 = 0;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int64_t x = (int64_t)g.ox + (k ? g.ex - 1 : 0u);
                const T v = in[x];
                const T s1 = blkn_seen(in, p, lat, g, 0, x - 1), s2 = blkn_seen(in, p, lat, g, 0, x - 2);
                e1 += (double)(T)((T)fabs((double)(T)(v - s1)) + n1);
                e2 += (double)(T)((T)fabs((double)(T)(v - (T)((T)(2 * s1) - s2))) + n2);
            }
            sid = e2 < e1 ? 1 : 0;
        }
        p.sel[task] = sid;
    }
}

// NW: waves of a workgroup — they share the LDS histogram window, so the wide window (64 KB) takes 16 of them to keep the CU occupied
// SELECT: the selection alone (choices and coefficients to sel[] / coef[], the number of blocks that would not be coded by
// first-order Lorenzo to *n_other; HW is then the one-word window of that count) — the 3-D path's question, asked first: a
// field on which only Lorenzo-1 is chosen goes to the plain Lorenzo path. With p.sel_given the coding launch takes the choices
// from there instead of fitting again.
template <typename T, uint32_t HW, int NW, bool TWO, bool SELECT>
__global__ __launch_bounds__(NW * 64) void k_blkn_fit(const T *__restrict__ in, uint16_t *__restrict__ codes, szk_blk_params p, uint32_t nblocks,
                                                      unsigned long long *__restrict__ n_other) {
    using Q = typename QTraits<T>::Q;
    __shared__ uint32_t lh[HW + 1];  // (+ the count of code 0: blk_count)
    for (uint32_t b = threadIdx.x; b <= HW; b += NW * 64) lh[b] = 0;
    __syncthreads();
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    const uint32_t wv = threadIdx.x / WAVE;
    const uint64_t d2 = p.d[2];
    Q *qwork = reinterpret_cast<Q *>(p.qwork);
    const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
    const double eb_recip = 1.0 / p.eb;
    const bool has_l1 = p.mask & 1u, has_r = p.mask & 4u;
    constexpr bool two = TWO;
    const T noise = (T)((two ? 0.81 : 0.5) * p.eb);
    for (uint32_t task = blockIdx.x * NW + wv; task < nblocks; task += gridDim.x * NW) {
        const BlkGeom g = blk_geom(p, task);
        const uint32_t nown = g.ey * g.ex;
        int sid = 0;
        int64_t lc[4] = {0, 0, 0, 0};
        if (!SELECT && p.sel_given) {
            sid = p.sel[task];
            if (sid != 2) continue;  // (a Lorenzo block's elements are put on the lattice by the code pass itself, straight from the array)
            if (sid == 2)
                for (int i = 1; i < 4; i++) lc[i] = p.coef[(uint64_t)task * 4 + i];
        } else {
        // ---- regression fit (RegressionPredictor.hpp:28-55, N = 1 / 2) ----
        const bool r_valid = has_r && g.ex > 1 && (!two || g.ey > 1);
        T cf[4] = {0, 0, 0, 0};
        if (r_valid) {
            double s1 = 0, s2 = 0, s3 = 0;
            for (uint32_t t = lane; t < nown; t += WAVE) {
                uint32_t i1, i2;
                blkn_own<TWO>(g, t, i1, i2);
                const T v = in[(uint64_t)(g.oy + i1) * d2 + (g.ox + i2)];
                if (two) s1 += (double)((T)i1 * v);
                s2 += (double)((T)i2 * v);
                s3 += (double)v;
            }
            if (two) s1 = wave_sum_f64(s1);
            s2 = wave_sum_f64(s2);
            s3 = wave_sum_f64(s3);
            const double dy = g.ey, dx = g.ex, num = dy * dx;
            if (two) cf[1] = (T)((2 * s1 / (dy - 1) - s3) * 6 / num / (dy + 1));
            cf[2] = (T)((2 * s2 / (dx - 1) - s3) * 6 / num / (dx + 1));
            cf[3] = (T)(s3 / num);
            if (two) cf[3] = (T)((double)cf[3] - (dy - 1) * (double)cf[1] / 2);
            cf[3] = (T)((double)cf[3] - (dx - 1) * (double)cf[2] / 2);
        }
        // ---- selection (ComposedPredictor.hpp:25-40 over foreach_sampling) ----
        sid = has_l1 ? 0 : 2;
        if (p.mask & 2u) {
            // second-order Lorenzo in the set (round 4: 1-D and 2-D arrays): the members in the set's order [Lorenzo-1, Lorenzo-2,
            // regression] (SZAlgoLorenzoReg.hpp:28-64), the estimates at the block's sample points (its two ends; its two diagonals),
            // the first minimum wins (std::min_element). LorenzoPredictor.hpp:69-71 / 75-79: 2 d[-1] - d[-2] and the eight-term
            // stencil in the reference's order of terms (prev2(d, ds, j, i): j rows up, i columns left); noise 1.08 / 2.76 eb (:17-38)
            const T noise2 = (T)((two ? 2.76 : 1.08) * p.eb);
            const uint32_t m = two ? min(g.ey, g.ex) : g.ex;
            const uint32_t npts = two ? 2 * m : 2;
            double e1 = 0, e2 = 0, er = 0;
            for (uint32_t k = lane; k < npts; k += WAVE) {
                uint32_t i1 = 0, i2;
                if (two) {
                    i1 = k / 2;
                    i2 = (k & 1) ? m - 1 - i1 : i1;
                } else {
                    i2 = k ? m - 1 : 0;
                }
                const int64_t y = (int64_t)g.oy + i1, x = (int64_t)g.ox + i2;
                const T v = in[(uint64_t)y * d2 + (uint64_t)x];
                auto P = [&](int j, int i) -> T { return blkn_seen(in, p, lat, g, y - j, x - i); };
                T pr1, pr2;
                if (two) {
                    pr1 = (T)((T)(P(0, 1) + P(1, 0)) - P(1, 1));
                    T s2 = (T)(2 * P(0, 1));
                    s2 = (T)(s2 - P(0, 2));
                    s2 = (T)(s2 + (T)(2 * P(1, 0)));
                    s2 = (T)(s2 - (T)(4 * P(1, 1)));
                    s2 = (T)(s2 + (T)(2 * P(1, 2)));
                    s2 = (T)(s2 - P(2, 0));
                    s2 = (T)(s2 + (T)(2 * P(2, 1)));
                    s2 = (T)(s2 - P(2, 2));
                    pr2 = s2;
                } else {
                    pr1 = P(0, 1);
                    pr2 = (T)((T)(2 * P(0, 1)) - P(0, 2));
                }
                e1 += (double)(T)((T)fabs((double)(T)(v - pr1)) + noise);
                e2 += (double)(T)((T)fabs((double)(T)(v - pr2)) + noise2);
                if (r_valid) er += (double)(T)fabs((double)(T)(v - reg_predict(cf, 0u, i1, i2)));
            }
            e1 = wave_sum_f64(e1);
            e2 = wave_sum_f64(e2);
            er = wave_sum_f64(er);
            int best = -1;
            double eb_best = 0;
            if (has_l1) {
                best = 0;
                eb_best = e1;
            }
            if (best < 0 || e2 < eb_best) {
                best = 1;
                eb_best = e2;
            }
            if (has_r) {
                // (an invalid regression's estimate is DBL_MAX: it wins only when it is the set's single... never here, Lorenzo-2 is in the set)
                if (r_valid && er < eb_best) best = 2;
            }
            sid = best;
        } else
        if (has_l1 && has_r) {
            const uint32_t m = two ? min(g.ey, g.ex) : g.ex;
            const uint32_t npts = two ? 2 * m : 2;
            double e1 = 0, er = 0;
            for (uint32_t k = lane; k < npts; k += WAVE) {
                uint32_t i1 = 0, i2;
                if (two) {
                    i1 = k / 2;
                    i2 = (k & 1) ? m - 1 - i1 : i1;
                } else {
                    i2 = k ? m - 1 : 0;
                }
                const int64_t y = (int64_t)g.oy + i1, x = (int64_t)g.ox + i2;
                const T v = in[(uint64_t)y * d2 + (uint64_t)x];
                T pr;  // LorenzoPredictor.hpp:61-64: N = 1: d[-1]; N = 2: (y, x-1) + (y-1, x) - (y-1, x-1)
                if (two) pr = (T)((T)(blkn_seen(in, p, lat, g, y, x - 1) + blkn_seen(in, p, lat, g, y - 1, x)) - blkn_seen(in, p, lat, g, y - 1, x - 1));
                else pr = blkn_seen(in, p, lat, g, y, x - 1);
                e1 += (double)(T)((T)fabs((double)(T)(v - pr)) + noise);
                if (r_valid) er += (double)(T)fabs((double)(T)(v - reg_predict(cf, 0u, i1, i2)));
            }
            e1 = wave_sum_f64(e1);
            er = wave_sum_f64(er);
            sid = (r_valid && er < e1) ? 2 : 0;
        } else if (sid == 2 && !r_valid) {
            sid = 0;  // BlockwiseDecomposition.hpp:35-37
        }
        if (sid == 2) {  // coefficients onto their lattices; a coefficient the lattice cannot hold -> Lorenzo-1
            bool ok = true;
            for (int i = 1; i < 4; i++) {
                const double sc = (double)cf[i] / (i < 3 ? cl.step_lin : cl.step_ind);
                if (!(fabs(sc) < 4503599627370496.0)) ok = false;
                else lc[i] = (int64_t)rint(sc);
            }
            if (!ok) sid = 0;
        }
        }  // (own selection)
        if (SELECT) {
            if (lane == 0) {
                p.sel[task] = (uint8_t)sid;
                if (sid == 2)
                    for (int i = 0; i < 4; i++) p.coef[(uint64_t)task * 4 + i] = lc[i];
                if (sid != 0) atomicAdd(&lh[0], 1u);
            }
            continue;
        }
        if (sid == 2) {
            T rc[4];
            coef_recover(lc, cl, rc);
            for (uint32_t t0 = 0; t0 < nown; t0 += WAVE) {
                const uint32_t t = t0 + lane;
                const bool act = t < nown;
                const uint32_t tt = act ? t : 0;
                uint32_t i1, i2;
                blkn_own<TWO>(g, tt, i1, i2);
                const uint64_t gi = (uint64_t)(g.oy + i1) * d2 + (g.ox + i2);
                const T raw = in[gi];
                T v = raw;
                const int code = act ? ref_quantize(v, reg_predict(rc, 0u, i1, i2), p.eb, eb_recip, (int)p.radius) : 1;
                Q qt = 0;
                if (code != 0) {
                    bool bad;
                    qt = lat.quant(v, bad);
                    if (bad) qt = 0;
                }
                if (act) {
                    codes[g.coff + t] = (uint16_t)code;
                    qwork[gi] = qt;
                }
                blk_count<HW>(lh, p, (uint32_t)code, act);
                blk_vout<T>(p, act && code == 0, gi, raw);
            }
            if (lane == 0)
                for (int i = 0; i < 4; i++) p.coef[(uint64_t)task * 4 + i] = lc[i];
        } else {
            for (uint32_t t0 = 0; t0 < nown; t0 += WAVE) {
                const uint32_t t = t0 + lane;
                const bool act = t < nown;
                const uint32_t tt = act ? t : 0;
                uint32_t i1, i2;
                blkn_own<TWO>(g, tt, i1, i2);
                const uint64_t gi = (uint64_t)(g.oy + i1) * d2 + (g.ox + i2);
                const T raw = in[gi];
                bool bad;
                Q q = lat.quant(raw, bad);
                if (bad) q = 0;
                if (act) qwork[gi] = q;
                blk_vout<T>(p, act && bad, gi, raw);
            }
        }
        if (lane == 0) p.sel[task] = (uint8_t)sid;
    }
    __syncthreads();
    if (SELECT) {
        if (threadIdx.x == 0 && lh[0]) atomicAdd(n_other, (unsigned long long)lh[0]);
    } else {
        blk_flush<HW>(lh, p);
    }
}

// 1-D, the fit pass by ROWS OF 16 LANES: four blocks per wave, a block per DPP row. A wave per block of 128 values spends ~600
// instructions on it — the fit's double divisions, the coefficient snapping, the block's geometry, the loops' bookkeeping — and
// only the sums are work that all 64 lanes share: here the per-block arithmetic is done once per ROW (each of its lanes holds the
// same values), four blocks for the price of one, and the two sums are reduced inside the rows. The selection's two sample
// points (the block's ends) are evaluated by every lane of the row in the reference's order (no reduction at all).
__device__ __forceinline__ double row16_sum_f64(double v) {  // the sum over the lane's row of 16, in every lane of it
    for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
template <typename T, uint32_t HW, int NW>
__global__ __launch_bounds__(NW * 64) void k_blkn_fit_rows(const T *__restrict__ in, uint16_t *__restrict__ codes, szk_blk_params p, uint32_t nblocks) {
    using Q = typename QTraits<T>::Q;
    __shared__ uint32_t lh[HW + 1];  // (+ the count of code 0: blk_count)
    for (uint32_t b = threadIdx.x; b <= HW; b += NW * 64) lh[b] = 0;
    __syncthreads();
    const Lattice<T> lat(p.lat);
    const uint32_t lane = (uint32_t)lane_id(), row = lane >> 4, li = lane & 15u;
    const uint32_t wv = threadIdx.x / WAVE;
    Q *qwork = reinterpret_cast<Q *>(p.qwork);
    const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
    const double eb_recip = 1.0 / p.eb;
    const bool has_l1 = p.mask & 1u, has_r = p.mask & 4u;
    const T noise = (T)(0.5 * p.eb);
    const uint32_t n = (uint32_t)p.d[2];
    for (uint32_t base = (blockIdx.x * NW + wv) * 4; base < nblocks; base += gridDim.x * NW * 4) {  // (wave-uniform)
        const uint32_t task = base + row;
        const bool live = task < nblocks;
        const uint32_t ox = live ? task * p.B : 0u;
        const uint32_t ex = live ? min(p.B, n - ox) : 0u;
        // ---- regression fit (RegressionPredictor.hpp:28-55, N = 1) ----
        const bool r_valid = has_r && ex > 1;
        double s2 = 0, s3 = 0;
        // blocks of up to 128 values (the default): the lane's eight values are requested together and kept for the coding loop
        constexpr int NV = 8;
        const bool kept = p.B <= 16u * NV;
        T vals[NV];
        if (kept) {
#pragma unroll
            for (int m = 0; m < NV; m++) {
                const uint32_t t = li + 16u * m;
                vals[m] = t < ex ? in[ox + t] : (T)0;
            }
#pragma unroll
            for (int m = 0; m < NV; m++) {
                const uint32_t t = li + 16u * m;
                if (t < ex) {
                    s2 += (double)((T)t * vals[m]);
                    s3 += (double)vals[m];
                }
            }
        } else
        for (uint32_t t0 = 0; t0 < p.B; t0 += 16) {
            const uint32_t t = t0 + li;
            if (t < ex) {
                const T v = in[ox + t];
                s2 += (double)((T)t * v);
                s3 += (double)v;
            }
        }
        s2 = row16_sum_f64(s2);
        s3 = row16_sum_f64(s3);
        T c1 = 0, c0 = 0;
        if (r_valid) {
            const double dx = ex;
            c1 = (T)((2 * s2 / (dx - 1) - s3) * 6 / dx / (dx + 1));
            c0 = (T)(s3 / dx);
            c0 = (T)((double)c0 - (dx - 1) * (double)c1 / 2);
        }
        // ---- selection (ComposedPredictor.hpp:25-40; BlockwiseIterator.hpp:151-184, N = 1: the points 0 and m - 1, in that order) ----
        int sid = has_l1 ? 0 : 2;
        const T cfv[4] = {0, 0, c1, c0};
        if (has_l1 && has_r && live) {
            double e1 = 0, er = 0;
            for (int k = 0; k < 2; k++) {
                const uint32_t i2 = k ? ex - 1 : 0u;
                const uint32_t x = ox + i2;
                const T v = in[x];
                T pr = 0;  // LorenzoPredictor.hpp:61: d[-1] — the original inside the block, the lattice reconstruction left of it, zero left of the array
                if (x) {
                    pr = in[x - 1];
                    if (i2 == 0) {
                        bool bad;
                        const Q qh = lat.quant(pr, bad);
                        if (!bad) pr = lat.dequant(qh);
                    }
                }
                e1 += (double)(T)((T)fabs((double)(T)(v - pr)) + noise);
                if (r_valid) er += (double)(T)fabs((double)(T)(v - reg_predict(cfv, 0u, 0u, i2)));
            }
            sid = (r_valid && er < e1) ? 2 : 0;
        } else if (sid == 2 && !r_valid) {
            sid = 0;  // BlockwiseDecomposition.hpp:35-37
        }
        int64_t l1 = 0, l0 = 0;
        if (sid == 2) {  // coefficients onto their lattices; a coefficient the lattice cannot hold -> Lorenzo-1
            const double a1 = (double)c1 / cl.step_lin, a0 = (double)c0 / cl.step_ind;
            if (!(fabs(a1) < 4503599627370496.0) || !(fabs(a0) < 4503599627370496.0)) sid = 0;
            else {
                l1 = (int64_t)rint(a1);
                l0 = (int64_t)rint(a0);
            }
        }
        const T rcv[4] = {0, 0, (T)((double)l1 * cl.step_lin), (T)((double)l0 * cl.step_ind)};  // (coef_recover's arithmetic)
        // ---- codes of the regression blocks, q~ of every element ----
        for (uint32_t t0 = 0; t0 < p.B; t0 += 16) {  // (the same trip count in every row: the list appends and counts are wave operations)
            const uint32_t t = t0 + li;
            const bool act = t < ex;
            const uint32_t gi = ox + (act ? t : 0u);
            T raw = 0;
            if (kept) {
#pragma unroll
                for (int m = 0; m < NV; m++)
                    if (t0 == 16u * m) raw = vals[m];
            } else if (act) {
                raw = in[gi];
            }
            bool bad = false;
            Q qt = 0;
            int code = 1;
            if (sid == 2) {
                T v = raw;
                code = act ? ref_quantize(v, reg_predict(rcv, 0u, 0u, t), p.eb, eb_recip, (int)p.radius) : 1;
                if (code != 0) {
                    qt = lat.quant(v, bad);
                    if (bad) qt = 0;
                }
                bad = code == 0;  // unpredictable: the raw value, LinearQuantizer.hpp:66-69
                if (act) codes[gi] = (uint16_t)code;  // (1-D: the block-major code order is the element order)
            } else {
                qt = lat.quant(raw, bad);
                if (bad) qt = 0;
            }
            if (act) qwork[gi] = qt;
            blk_count<HW>(lh, p, (uint32_t)code, act && sid == 2);
            blk_vout<T>(p, act && bad, gi, raw);
        }
        if (live && li == 0) {
            p.sel[task] = (uint8_t)sid;
            if (sid == 2) {
                p.coef[(uint64_t)task * 4 + 0] = 0;
                p.coef[(uint64_t)task * 4 + 1] = 0;
                p.coef[(uint64_t)task * 4 + 2] = l1;
                p.coef[(uint64_t)task * 4 + 3] = l0;
            }
        }
    }
    __syncthreads();
    blk_flush<HW>(lh, p);
}

// code position -> block and element of the (1, dy, dx) view: the bands of B rows hold B * dx codes each, a band's blocks ey * B
struct BlknPos {
    uint32_t task, y, x, ox, oy;
};
__device__ __forceinline__ BlknPos blkn_pos(const szk_blk_params &p, uint64_t c) {
    BlknPos r;
    const uint64_t band = (uint64_t)p.B * p.d[2];
    const uint32_t by = (uint32_t)(c / band);
    const uint64_t rem = c - (uint64_t)by * band;
    r.oy = by * p.B;
    const uint32_t ey = min(p.B, (uint32_t)p.d[1] - r.oy);
    const uint32_t bx = (uint32_t)(rem / ((uint64_t)ey * p.B));
    const uint32_t rem2 = (uint32_t)(rem - (uint64_t)bx * ey * p.B);
    r.ox = bx * p.B;
    const uint32_t ex = min(p.B, (uint32_t)p.d[2] - r.ox);
    const uint32_t j = rem2 / ex;
    r.y = r.oy + j;
    r.x = r.ox + (rem2 - j * ex);
    r.task = by * p.nb[2] + bx;
    return r;
}

// With the selection pass's choices (p.sel_given) q~ of an element outside a regression block is rint(x / 2eb), a function of that
// element alone: the pass takes it from the array (its own and its lower neighbours'), the work array holds the regression blocks'
// values only — nothing is written or read back for the Lorenzo blocks, and the fit pass does not visit them.
template <typename T, uint32_t HW, int TB, bool TWO>
__global__ __launch_bounds__(TB) void k_blkn_lorenzo(const T *__restrict__ in, uint16_t *__restrict__ codes, szk_blk_params p, uint64_t n) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    const Lattice<T> lat(p.lat);
    const bool direct = p.sel_given != 0;
    __shared__ uint32_t lh[HW + 1];  // (+ the count of code 0: blk_count)
    for (uint32_t b = threadIdx.x; b <= HW; b += TB) lh[b] = 0;
    __syncthreads();
    const Q *__restrict__ qw = reinterpret_cast<const Q *>(p.qwork);
    const uint64_t d2 = p.d[2];
    const uint64_t band = (uint64_t)p.B * d2;
    // where the workgroup's first code lies (1-D: block and offset in it; 2-D: band and offset in it): one long division before
    // the loop, carried from round to round by additions — a 64-bit division per round was a third of the pass's instructions
    const uint64_t unit = TWO ? band : (uint64_t)p.B, stride = (uint64_t)gridDim.x * TB;
    uint64_t u0 = ((uint64_t)blockIdx.x * TB) / unit, o0 = ((uint64_t)blockIdx.x * TB) % unit;
    const uint64_t us = stride / unit, os = stride % unit;
    const float rcp_b = 1.0f / (float)p.B;
    for (uint64_t c0 = (uint64_t)blockIdx.x * TB; c0 < n; c0 += stride, u0 += us, o0 += os) {
        if (o0 >= unit) {
            o0 -= unit;
            u0++;
        }
        const uint64_t c = c0 + threadIdx.x;
        bool act = c < n;
        BlknPos e;
        if (!TWO) {  // the thread's own block: a short division (offsets below B + TB: exact through the float reciprocal + one correction)
            const uint32_t r = (uint32_t)o0 + threadIdx.x;
            uint32_t q = (uint32_t)((float)r * rcp_b);
            if (q * p.B > r) q--;
            else if ((q + 1) * p.B <= r) q++;
            e.task = (uint32_t)u0 + q;
            e.y = e.oy = 0;
            e.x = act ? (uint32_t)c : 0u;
            if (!act) e.task = 0;
            e.ox = e.task * p.B;
        } else {
            // (the workgroup's first band likewise; a workgroup's codes span at most two bands when a band has TB codes or more)
            const uint64_t b0 = u0;
            const uint64_t off = o0 + threadIdx.x;
            if (off < 2 * band && band < 0x80000000ull && act) {
                const uint32_t by = (uint32_t)b0 + (off >= band ? 1u : 0u);
                const uint32_t rem = (uint32_t)(off >= band ? off - band : off);
                const uint32_t oy = by * p.B, ey = min(p.B, (uint32_t)p.d[1] - oy);
                const uint32_t per = ey * p.B, bx = rem / per, rem2 = rem - bx * per;
                const uint32_t ox = bx * p.B, ex = min(p.B, (uint32_t)d2 - ox);
                const uint32_t j = ex == 16 ? rem2 >> 4 : rem2 / ex;
                e.y = oy + j;
                e.x = ox + (rem2 - j * ex);
                e.oy = oy;
                e.ox = ox;
                e.task = by * p.nb[2] + bx;
            } else {
                e = blkn_pos(p, act ? c : 0);
            }
        }
        const uint8_t own_sel = p.sel[e.task];
        act = act && own_sel != 2;
        const bool second = !TWO && own_sel == 1;  // 1-D, second-order Lorenzo: q - 2 q[-1] + q[-2] (LorenzoPredictor.hpp:69-71 on the lattice)
        const bool second2 = TWO && own_sel == 1;  // 2-D: the nine-term stencil (1, -2, 1) x (1, -2, 1) (LorenzoPredictor.hpp:75-79)
        UQ delta = 0;
        bool own_bad = false;
        T own_raw = 0;
        const uint64_t gi = (uint64_t)e.y * d2 + e.x;
        if (act && second2) {
            // (the general form only: q~ of every element through the work array, or — with the selection pass's choices — q~ of an
            // element outside a regression block from the array itself; nine taps, zeros outside the array)
            auto q9 = [&](uint32_t yy, uint32_t xx, bool own) -> UQ {
                const uint64_t at = (uint64_t)yy * d2 + xx;
                if (!direct) return (UQ)qw[at];
                const uint32_t tk = (yy / p.B) * p.nb[2] + xx / p.B;
                if (!own && p.sel[tk] == 2) return (UQ)qw[at];
                const T v = in[at];
                bool bad;
                const Q q = lat.quant(v, bad);
                if (own) {
                    own_bad = bad;
                    own_raw = v;
                }
                return bad ? (UQ)0 : (UQ)q;
            };
            const int w3[3] = {1, -2, 1};
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    if (e.y < (uint32_t)j || e.x < (uint32_t)i) continue;
                    const UQ q = q9(e.y - j, e.x - i, (j | i) == 0);
                    delta += (UQ)(Q)(w3[j] * w3[i]) * q;
                }
        } else
        if (act && !direct) {
            delta = (UQ)qw[gi];
            if (e.x) delta -= (UQ)qw[gi - 1];
            if (second) {
                if (e.x) delta -= (UQ)qw[gi - 1];
                if (e.x > 1) delta += (UQ)qw[gi - 2];
            }
            if (e.y) {
                delta -= (UQ)qw[gi - d2];
                if (e.x) delta += (UQ)qw[gi - d2 - 1];
            }
        } else if (act) {
            // q~ at an element: the regression blocks' from the work array, everything else from the array itself
            auto qt = [&](uint64_t at, uint32_t task, bool own) -> UQ {
                if (!own && p.sel[task] == 2) return (UQ)qw[at];
                const T v = in[at];
                bool bad;
                const Q q = lat.quant(v, bad);
                if (own) {
                    own_bad = bad;
                    own_raw = v;
                }
                return bad ? (UQ)0 : (UQ)q;
            };
            const uint32_t tl = e.x == e.ox ? e.task - 1 : e.task;  // the block of the left neighbour
            delta = qt(gi, e.task, true);
            if (e.x) {
                const UQ l1 = qt(gi - 1, tl, false);
                delta -= l1;
                if (second) {
                    delta -= l1;
                    if (e.x > 1) delta += qt(gi - 2, e.x - e.ox < 2 ? e.task - 1 : e.task, false);  // (blocks hold four values or more)
                }
            }
            if (TWO && e.y) {
                const uint32_t up = e.y == e.oy ? p.nb[2] : 0u;
                delta -= qt(gi - d2, e.task - up, false);
                if (e.x) delta += qt(gi - d2 - 1, tl - up, false);
            }
        }
        blk_vout<T>(p, act && own_bad, gi, own_raw);  // unpredictable: the raw value (a wave operation, all lanes take part)
        const bool inr = (UQ)(delta + (UQ)(p.radius - 1)) <= (UQ)(2 * p.radius - 2);
        const uint32_t code = inr ? (uint32_t)(delta + (UQ)p.radius) : 0u;
        if (act) codes[c] = (uint16_t)code;
        blk_count<HW>(lh, p, code, act);
        const unsigned long long pd = wave_append_slot(act && !inr, p.n_dout);
        if (act && !inr && pd < p.out_cap) {
            p.dout_idx[pd] = c;
            reinterpret_cast<Q *>(p.dout_val)[pd] = (Q)delta;
        }
    }
    __syncthreads();
    blk_flush<HW>(lh, p);
}

// 1-D, the work array holds q~ of every element (the fit pass chose and coded in one go): FOUR codes per thread — one 16-byte load
// of lattice values, the left neighbours from registers, one 8-byte store of codes; the block of an element from one short
// division per thread (a thread's four elements lie in at most two blocks when B >= 4).
template <typename T, uint32_t HW, int TB>
__global__ __launch_bounds__(TB) void k_blkn_lorenzo1v(uint16_t *__restrict__ codes, szk_blk_params p, uint64_t n) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    __shared__ uint32_t lh[HW + 1];  // (+ the count of code 0: blk_count)
    for (uint32_t b = threadIdx.x; b <= HW; b += TB) lh[b] = 0;
    __syncthreads();
    const Q *__restrict__ qw = reinterpret_cast<const Q *>(p.qwork);
    const uint64_t stride = (uint64_t)gridDim.x * TB * 4;
    for (uint64_t c0 = (uint64_t)blockIdx.x * TB * 4; c0 < n; c0 += stride) {  // (workgroup-uniform: every lane takes part in the wave operations)
        const uint64_t c = c0 + (uint64_t)threadIdx.x * 4;
        const bool any = c < n;
        const uint32_t task0 = any ? (uint32_t)c / p.B : 0u;  // (positions of a 1-D block stream fit 32 bits: blk_shape_ok)
        const uint64_t next = ((uint64_t)task0 + 1) * p.B;  // first element of the following block
        Q q[4] = {0, 0, 0, 0};
        Q left = 0;
        if (any) {
            if (c + 3 < n) {
                if (sizeof(Q) == 4) {
                    const int4 v = *reinterpret_cast<const int4 *>(qw + c);
                    q[0] = (Q)v.x; q[1] = (Q)v.y; q[2] = (Q)v.z; q[3] = (Q)v.w;
                } else {
                    const longlong2 v0 = *reinterpret_cast<const longlong2 *>(qw + c), v1 = *reinterpret_cast<const longlong2 *>(qw + c + 2);
                    q[0] = (Q)v0.x; q[1] = (Q)v0.y; q[2] = (Q)v1.x; q[3] = (Q)v1.y;
                }
            } else {
                for (int j = 0; j < 4; j++)
                    if (c + j < n) q[j] = qw[c + j];
            }
            if (c) left = qw[c - 1];
        }
        const uint8_t s0 = any ? p.sel[task0] : (uint8_t)2, s1 = any && next < n ? p.sel[task0 + 1] : (uint8_t)2;
        uint32_t code4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t cj = c + j;
            const bool act = cj < n && (cj < next ? s0 : s1) != 2;
            const UQ delta = (UQ)q[j] - (UQ)(j ? q[j - 1] : left);
            const bool inr = (UQ)(delta + (UQ)(p.radius - 1)) <= (UQ)(2 * p.radius - 2);
            const uint32_t code = inr ? (uint32_t)(delta + (UQ)p.radius) : 0u;
            code4[j] = code;
            blk_count<HW>(lh, p, code, act);
            const unsigned long long pd = wave_append_slot(act && !inr, p.n_dout);
            if (act && !inr && pd < p.out_cap) {
                p.dout_idx[pd] = cj;
                reinterpret_cast<Q *>(p.dout_val)[pd] = (Q)delta;
            }
            if (!act) code4[j] = 0xFFFFFFFFu;  // (a regression block's code: written by the fit pass, left alone)
        }
        if (any) {
            const bool all = code4[0] != 0xFFFFFFFFu && code4[1] != 0xFFFFFFFFu && code4[2] != 0xFFFFFFFFu && code4[3] != 0xFFFFFFFFu;
            if (all) {
                ushort4 o;
                o.x = (uint16_t)code4[0]; o.y = (uint16_t)code4[1]; o.z = (uint16_t)code4[2]; o.w = (uint16_t)code4[3];
                *reinterpret_cast<ushort4 *>(codes + c) = o;
            } else {
                for (int j = 0; j < 4; j++)
                    if (code4[j] != 0xFFFFFFFFu) codes[c + j] = (uint16_t)code4[j];
            }
        }
    }
    __syncthreads();
    blk_flush<HW>(lh, p);
}

// 1-D, Lorenzo members only (the choices from k_blkn_sel12): FOUR codes per thread straight from the array — one 16-byte load of values,
// the two left neighbours, six lattice values in registers, one 8-byte store of codes (a thread's four elements lie in at most two
// blocks when B >= 4)
template <typename T, uint32_t HW, int TB>
__global__ __launch_bounds__(TB) void k_blkn_lorenzo12v(const T *__restrict__ in, uint16_t *__restrict__ codes, szk_blk_params p, uint64_t n) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    const Lattice<T> lat(p.lat);
    __shared__ uint32_t lh[HW + 1];  // (+ the count of code 0: blk_count)
    for (uint32_t b = threadIdx.x; b <= HW; b += TB) lh[b] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * TB * 4;
    for (uint64_t c0 = (uint64_t)blockIdx.x * TB * 4; c0 < n; c0 += stride) {  // (workgroup-uniform: every lane takes part in the wave operations)
        const uint64_t c = c0 + (uint64_t)threadIdx.x * 4;
        const bool any = c < n;
        const uint32_t task0 = any ? (uint32_t)c / p.B : 0u;  // (positions of a 1-D block stream fit 32 bits: blk_shape_ok)
        const uint64_t next = ((uint64_t)task0 + 1) * p.B;  // first element of the following block
        T v[6] = {0, 0, 0, 0, 0, 0};  // [0], [1]: the two values left of the thread's four
        if (any) {
            if (c + 3 < n) {
                if (sizeof(T) == 4) {
                    const float4 f = *reinterpret_cast<const float4 *>(in + c);
                    v[2] = (T)f.x; v[3] = (T)f.y; v[4] = (T)f.z; v[5] = (T)f.w;
                } else {
                    const double2 f0 = *reinterpret_cast<const double2 *>(in + c), f1 = *reinterpret_cast<const double2 *>(in + c + 2);
                    v[2] = (T)f0.x; v[3] = (T)f0.y; v[4] = (T)f1.x; v[5] = (T)f1.y;
                }
            } else {
                for (int j = 0; j < 4; j++)
                    if (c + j < n) v[2 + j] = in[c + j];
            }
            if (c) v[1] = in[c - 1];
            if (c > 1) v[0] = in[c - 2];
        }
        const uint8_t s0 = any ? p.sel[task0] : (uint8_t)0, s1 = any && next < n ? p.sel[task0 + 1] : (uint8_t)0;
        Q q[6];
        bool bad[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            q[j] = lat.quant(v[j], bad[j]);
            if (bad[j]) q[j] = 0;
        }
        uint32_t code4[4];
        UQ delta4[4];
        uint32_t n_far = 0, n_bad = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t cj = c + j;
            const bool act = cj < n;
            UQ delta = (UQ)q[2 + j] - (UQ)q[1 + j];
            if ((cj < next ? s0 : s1) == 1) delta = delta - (UQ)q[1 + j] + (UQ)q[j];
            const bool inr = (UQ)(delta + (UQ)(p.radius - 1)) <= (UQ)(2 * p.radius - 2);
            const uint32_t code = inr ? (uint32_t)(delta + (UQ)p.radius) : 0u;
            code4[j] = code;
            delta4[j] = delta;
            n_far += act && !inr;
            n_bad += act && bad[2 + j];
            blk_count<HW>(lh, p, code, act);
        }
        // the list appends: ONE atomic per wave and list for the four elements of all its lanes (an atomic per element position made a
        // series with a far delta in most waves crawl: same-address atomics run at ~90 per us)
        unsigned long long pd = ~0ull, pv = ~0ull;
        if (HW > BLK_HWIN) {
            // the wide form (the context's previous call met a wide alphabet: rough data, or a bound far below the signal's slope — far
            // deltas in most waves): one atomic per WORKGROUP and list. Two barriers per round of the loop, which every thread takes.
            __shared__ uint32_t s_cnt[2][2][TB / WAVE];
            __shared__ unsigned long long s_base[2][2];
            const uint32_t par = (uint32_t)(((c0 / stride) & 1u)), wv = threadIdx.x / WAVE;
            const int lane = lane_id();
            const unsigned long long lt = (1ull << lane) - 1ull;
            uint32_t pre[2], tot[2];
            const uint32_t cn[2] = {n_far, n_bad};
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const unsigned long long b0 = __ballot(cn[k] & 1u), b1 = __ballot(cn[k] & 2u), b2 = __ballot(cn[k] & 4u);
                tot[k] = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
                pre[k] = (uint32_t)__popcll(b0 & lt) + 2u * (uint32_t)__popcll(b1 & lt) + 4u * (uint32_t)__popcll(b2 & lt);
                if (lane == 0) s_cnt[par][k][wv] = tot[k];
            }
            __syncthreads();
            if (threadIdx.x < 2) {
                uint32_t sum = 0;
                for (uint32_t w2 = 0; w2 < TB / WAVE; w2++) sum += s_cnt[par][threadIdx.x][w2];
                s_base[par][threadIdx.x] = sum ? atomicAdd((unsigned long long *)(threadIdx.x ? p.n_vout : p.n_dout), (unsigned long long)sum) : 0ull;
            }
            __syncthreads();
            uint32_t before[2] = {0, 0};
            for (uint32_t w2 = 0; w2 < wv; w2++) {
                before[0] += s_cnt[par][0][w2];
                before[1] += s_cnt[par][1][w2];
            }
            pd = s_base[par][0] + before[0] + pre[0];
            pv = s_base[par][1] + before[1] + pre[1];
        } else {
            if (__ballot(n_far != 0)) pd = wave_append_run(n_far, p.n_dout);
            if (__ballot(n_bad != 0)) pv = wave_append_run(n_bad, p.n_vout);
        }
        if (n_far) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (c + j < n && code4[j] == 0) {
                    if (pd < p.out_cap) {
                        p.dout_idx[pd] = c + j;
                        reinterpret_cast<Q *>(p.dout_val)[pd] = (Q)delta4[j];
                    }
                    pd++;
                }
        }
        if (n_bad) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (c + j < n && bad[2 + j]) {
                    if (pv < p.out_cap) {
                        p.vout_idx[pv] = c + j;
                        reinterpret_cast<T *>(p.vout_val)[pv] = v[2 + j];
                    }
                    pv++;
                }
        }
        if (any) {
            if (c + 3 < n) {
                ushort4 o;
                o.x = (uint16_t)code4[0]; o.y = (uint16_t)code4[1]; o.z = (uint16_t)code4[2]; o.w = (uint16_t)code4[3];
                *reinterpret_cast<ushort4 *>(codes + c) = o;
            } else {
                for (int j = 0; j < 4; j++)
                    if (c + j < n) codes[c + j] = (uint16_t)code4[j];
            }
        }
    }
    __syncthreads();
    blk_flush<HW>(lh, p);
}

// ---- decoder ----
// regression blocks: q~ of their elements into the output (as lattice words; k_blk_final turns everything into T). 1-D: every
// block leaves its aggregate for the scan over the blocks — a regression block the lattice value of its last element, a Lorenzo
// block the sum of its deltas.
template <typename T>
__global__ __launch_bounds__(256) void k_blkn_pre(const uint16_t *__restrict__ codes, const void *deltas_, void *d_out, szk_blk_params p, uint32_t nblocks,
                                                  const uint32_t *__restrict__ rank, const int64_t *__restrict__ coef_by_rank, int to_work = 0) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    const uint64_t d2 = p.d[2];
    Q *qout = reinterpret_cast<Q *>(d_out);
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    Q *agg = reinterpret_cast<Q *>(p.carry);
    const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
    for (uint32_t task = blockIdx.x * 4 + threadIdx.x / WAVE; task < nblocks; task += gridDim.x * 4) {
        const BlkGeom g = blk_geom(p, task);
        const uint32_t nown = g.ey * g.ex;
        if (p.sel[task] == 2) {
            T rc[4];
            coef_recover(coef_by_rank + (uint64_t)rank[task] * 4, cl, rc);
            for (uint32_t t = lane; t < nown; t += WAVE) {
                const uint32_t i1 = t / g.ex, i2 = t - i1 * g.ex;
                const uint32_t code = codes[g.coff + t];
                Q qt = 0;
                T val = 0;  // (code 0: patched from the list)
                if (code) {
                    bool bad;
                    val = ref_recover(reg_predict(rc, 0u, i1, i2), (int)code, p.eb, (int)p.radius);
                    qt = lat.quant(val, bad);
                    if (bad) qt = 0;
                }
                // 1-D: nothing reads a regression block's lattice values but the scan over the blocks (its last element's, below):
                // the element's final value goes out here and the final pass is not run
                if (p.ndim == 1) reinterpret_cast<T *>(d_out)[g.ox + i2] = val;
                else if (to_work) {  // (k_blkn_wave2: the lattice value over the block's slot in the work array, the final value out)
                    const_cast<Q *>(deltas)[g.coff + t] = qt;
                    reinterpret_cast<T *>(d_out)[(uint64_t)(g.oy + i1) * d2 + (g.ox + i2)] = val;
                } else qout[(uint64_t)(g.oy + i1) * d2 + (g.ox + i2)] = qt;
                if (p.ndim == 1 && t == nown - 1) agg[2 * (uint64_t)task] = qt;
            }
        } else if (p.ndim == 1) {
            // (1-D: the codes are read where they lie — delta = code - radius — and only the outliers' deltas, code 0, come from the
            // array they were scattered to: the dense codes -> deltas pass is not run)
            UQ sum = 0;
            for (uint32_t t = lane; t < nown; t += WAVE) {
                const uint32_t code = codes[g.coff + t];
                sum += code ? (UQ)(Q)((int)code - (int)p.radius) : (UQ)deltas[g.coff + t];
            }
            sum = wave_sum(sum);
            if (lane == 0) agg[2 * (uint64_t)task] = (Q)sum;
        }
    }
}
// 1-D: the value left of every block — a segmented exclusive scan of the blocks' aggregates, in two levels: a workgroup per tile of
// 1024 blocks (prefix inside the tile, the tile's own aggregate, and whether a regression block closes the prefix before a block),
// one workgroup over the tiles, and the consumer (k_blkn_apply1) adds the tile's inflow where the prefix is still open.
// carry buffer: Q agg[2 * nblocks] (aggregate, prefix), Q tile[2 * ntiles] (aggregate / inflow, restart flag), u8 closed[nblocks]
#define BLKN_TILE 1024u
template <typename Q>
__device__ __forceinline__ Q *blkn_tiles(void *carry, uint32_t nblocks) { return reinterpret_cast<Q *>(carry) + 2 * (uint64_t)nblocks; }
template <typename Q>
__device__ __forceinline__ uint8_t *blkn_closed(void *carry, uint32_t nblocks) {
    return reinterpret_cast<uint8_t *>(blkn_tiles<Q>(carry, nblocks) + 2 * (uint64_t)((nblocks + BLKN_TILE - 1) / BLKN_TILE));
}
// segmented inclusive scan of (f, a) over a workgroup of 1024 threads; pa: inflow of the workgroup. Returns the inclusive value,
// `f` becomes "a restart at or before this thread", `excl` / `fex` the same for the threads before this one
template <typename UQ>
__device__ __forceinline__ UQ blkn_wg_scan(uint32_t &f, UQ a, UQ inflow, UQ &excl, uint32_t &fex, UQ *wa, uint32_t *wf) {
    const int lane = lane_id();
    const uint32_t w = threadIdx.x / WAVE;
    for (int off = 1; off < WAVE; off <<= 1) {
        const UQ a2 = (UQ)__shfl_up((long long)a, off);
        const uint32_t f2 = (uint32_t)__shfl_up((int)f, off);
        if (lane >= off) {
            if (!f) a += a2;
            f |= f2;
        }
    }
    if (lane == WAVE - 1) {
        wa[w] = a;
        wf[w] = f;
    }
    __syncthreads();
    UQ pa = inflow;
    uint32_t pf = 0;
    for (uint32_t k = 0; k < w; k++) {
        pa = wf[k] ? wa[k] : pa + wa[k];
        pf |= wf[k];
    }
    const UQ incl = f ? a : pa + a;
    const uint32_t fincl = f | pf;
    excl = (UQ)__shfl_up((long long)incl, 1);
    fex = (uint32_t)__shfl_up((int)fincl, 1);
    if (lane == 0) {
        excl = pa;
        fex = pf;
    }
    f = fincl;
    __syncthreads();
    return incl;
}
template <typename Q>
__global__ __launch_bounds__(1024) void k_blkn_scan_tile(const uint8_t *__restrict__ sel, uint32_t nblocks, void *carry) {
    using UQ = typename std::make_unsigned<Q>::type;
    __shared__ UQ wa[16];
    __shared__ uint32_t wf[16];
    Q *agg = reinterpret_cast<Q *>(carry);
    Q *tile = blkn_tiles<Q>(carry, nblocks);
    uint8_t *closed = blkn_closed<Q>(carry, nblocks);
    const uint32_t b = blockIdx.x * BLKN_TILE + threadIdx.x;
    const bool live = b < nblocks;
    uint32_t f = live && sel[b] == 2 ? 1u : 0u, fex;
    UQ excl;
    const UQ incl = blkn_wg_scan<UQ>(f, live ? (UQ)agg[2 * (uint64_t)b] : (UQ)0, (UQ)0, excl, fex, wa, wf);
    if (live) {
        agg[2 * (uint64_t)b + 1] = (Q)excl;
        closed[b] = (uint8_t)fex;
    }
    if (threadIdx.x == BLKN_TILE - 1) {
        tile[2 * (uint64_t)blockIdx.x] = (Q)incl;
        tile[2 * (uint64_t)blockIdx.x + 1] = (Q)f;
    }
}
template <typename Q>
__global__ __launch_bounds__(1024) void k_blkn_scan_top(uint32_t ntiles, Q *__restrict__ tile) {
    using UQ = typename std::make_unsigned<Q>::type;
    __shared__ UQ wa[16];
    __shared__ uint32_t wf[16];
    __shared__ UQ run_s;
    UQ run = 0;
    for (uint32_t base = 0; base < ntiles; base += 1024) {
        const uint32_t t = base + threadIdx.x;
        const bool live = t < ntiles;
        uint32_t f = live && tile[2 * (uint64_t)t + 1] != 0 ? 1u : 0u, fex;
        UQ excl;
        const UQ incl = blkn_wg_scan<UQ>(f, live ? (UQ)tile[2 * (uint64_t)t] : (UQ)0, run, excl, fex, wa, wf);
        if (live) tile[2 * (uint64_t)t + 1] = (Q)excl;  // (the tile's inflow replaces its flag: this thread was the flag's only reader)
        if (threadIdx.x == 1023) run_s = incl;
        __syncthreads();
        run = run_s;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_blkn_apply1(const uint16_t *__restrict__ codes, const void *deltas_, void *d_out, szk_blk_params p, uint32_t nblocks) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    const int lane = lane_id();
    const Lattice<T> lat(p.lat);
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    const Q *agg = reinterpret_cast<const Q *>(p.carry);
    const Q *tile = blkn_tiles<Q>(p.carry, nblocks);
    const uint8_t *closed = blkn_closed<Q>(p.carry, nblocks);
    for (uint32_t task = blockIdx.x * 4 + threadIdx.x / WAVE; task < nblocks; task += gridDim.x * 4) {
        if (p.sel[task] == 2) continue;
        const BlkGeom g = blk_geom(p, task);
        UQ run = (UQ)agg[2 * (uint64_t)task + 1];
        if (!closed[task]) run += (UQ)tile[2 * (uint64_t)(task / BLKN_TILE) + 1];
        for (uint32_t t0 = 0; t0 < g.ex; t0 += WAVE) {
            const uint32_t t = t0 + lane;
            UQ dl = 0;
            if (t < g.ex) {
                const uint32_t code = codes[g.coff + t];
                dl = code ? (UQ)(Q)((int)code - (int)p.radius) : (UQ)deltas[g.coff + t];
            }
            const UQ incl = wave_incl_scan(dl) + run;
            if (t < g.ex) reinterpret_cast<T *>(d_out)[g.ox + t] = lat.dequant((Q)incl);  // (the final value: no pass over the array after this one)
            run = (UQ)__shfl((long long)incl, WAVE - 1);
        }
    }
}
template <typename UQ>
__device__ __forceinline__ UQ row16_incl_scan(UQ u) {
    u += dpp_mov0<0x111, 0xf>(u);
    u += dpp_mov0<0x112, 0xf>(u);
    u += dpp_mov0<0x114, 0xf>(u);
    u += dpp_mov0<0x118, 0xf>(u);
    return u;
}
// The same by ROWS OF 16 LANES for blocks of up to 128 values that are multiples of 8 (the default 128): four blocks per wave, a
// lane takes eight consecutive values — one 16-byte load of codes, the prefix in registers, the lanes' totals scanned inside the
// DPP row, eight values written — instead of two rounds of a full-wave scan with 64 two-byte loads each.
template <typename T>
__global__ __launch_bounds__(256) void k_blkn_apply1_rows(const uint16_t *__restrict__ codes, const void *deltas_, void *d_out, szk_blk_params p, uint32_t nblocks) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    const uint32_t lane = (uint32_t)lane_id(), row = lane >> 4, li = lane & 15u;
    const Lattice<T> lat(p.lat);
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    const Q *agg = reinterpret_cast<const Q *>(p.carry);
    const Q *tile = blkn_tiles<Q>(p.carry, nblocks);
    const uint8_t *closed = blkn_closed<Q>(p.carry, nblocks);
    T *out = reinterpret_cast<T *>(d_out);
    const bool out_aligned = (reinterpret_cast<uintptr_t>(d_out) & 15u) == 0;  // (the caller's array: only as aligned as its element type for sure)
    const uint32_t n = (uint32_t)p.d[2];
    for (uint32_t base = (blockIdx.x * 4 + threadIdx.x / WAVE) * 4; base < nblocks; base += gridDim.x * 16) {  // (wave-uniform)
        const uint32_t task = base + row;
        const bool live = task < nblocks && p.sel[task] != 2;
        const uint32_t ox = task < nblocks ? task * p.B : 0u;
        const uint32_t ex = live ? min(p.B, n - ox) : 0u;
        const uint32_t t = li * 8;  // the lane's first value in the block
        UQ d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t + 8 <= ex) {
            const uint4 w = *reinterpret_cast<const uint4 *>(codes + ox + t);  // (ox and t are multiples of 8: 16-byte aligned)
            const uint32_t c[8] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16, w.z & 0xFFFFu, w.z >> 16, w.w & 0xFFFFu, w.w >> 16};
#pragma unroll
            for (int j = 0; j < 8; j++) d[j] = c[j] ? (UQ)(Q)((int)c[j] - (int)p.radius) : (UQ)deltas[ox + t + j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (t + j < ex) {
                    const uint32_t c = codes[ox + t + j];
                    d[j] = c ? (UQ)(Q)((int)c - (int)p.radius) : (UQ)deltas[ox + t + j];
                }
        }
#pragma unroll
        for (int j = 1; j < 8; j++) d[j] += d[j - 1];
        UQ inflow = 0;
        if (live) {
            inflow = (UQ)agg[2 * (uint64_t)task + 1];
            if (!closed[task]) inflow += (UQ)tile[2 * (uint64_t)(task / BLKN_TILE) + 1];
        }
        const UQ before = row16_incl_scan(d[7]) - d[7] + inflow;  // the values left of this lane's eight
        T v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = lat.dequant((Q)(d[j] + before));
        if (t + 8 <= ex && out_aligned) {  // 16-byte stores (eight predicated scalar stores per lane held the pass at 2.3 TB/s)
            if (sizeof(T) == 4) {
                float4 *o4 = reinterpret_cast<float4 *>(out + ox + t);
                o4[0] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
                o4[1] = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
            } else {
                double2 *o2 = reinterpret_cast<double2 *>(out + ox + t);
#pragma unroll
                for (int j = 0; j < 4; j++) o2[j] = make_double2((double)v[2 * j], (double)v[2 * j + 1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (t + j < ex) out[ox + t + j] = v[j];
        }
    }
}
// k_blkn_pre's 1-D work by rows of 16 lanes (blocks of up to 128 values, a multiple of 8; four blocks per wave, eight values per
// lane): a regression block's final values and the lattice value of its last element, a Lorenzo block's sum of deltas
template <typename T>
__global__ __launch_bounds__(256) void k_blkn_pre1_rows(const uint16_t *__restrict__ codes, const void *deltas_, void *d_out, szk_blk_params p, uint32_t nblocks,
                                                        const uint32_t *__restrict__ rank, const int64_t *__restrict__ coef_by_rank) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    const uint32_t lane = (uint32_t)lane_id(), row = lane >> 4, li = lane & 15u;
    const Lattice<T> lat(p.lat);
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    Q *agg = reinterpret_cast<Q *>(p.carry);
    T *out = reinterpret_cast<T *>(d_out);
    const bool out_aligned = (reinterpret_cast<uintptr_t>(d_out) & 15u) == 0;
    const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
    const uint32_t n = (uint32_t)p.d[2];
    for (uint32_t base = (blockIdx.x * 4 + threadIdx.x / WAVE) * 4; base < nblocks; base += gridDim.x * 16) {  // (wave-uniform)
        const uint32_t task = base + row;
        const bool live = task < nblocks;
        const bool reg = live && p.sel[task] == 2;
        const uint32_t ox = live ? task * p.B : 0u;
        const uint32_t ex = live ? min(p.B, n - ox) : 0u;
        const uint32_t t = li * 8;
        uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t + 8 <= ex) {
            const uint4 w = *reinterpret_cast<const uint4 *>(codes + ox + t);
            c[0] = w.x & 0xFFFFu; c[1] = w.x >> 16; c[2] = w.y & 0xFFFFu; c[3] = w.y >> 16;
            c[4] = w.z & 0xFFFFu; c[5] = w.z >> 16; c[6] = w.w & 0xFFFFu; c[7] = w.w >> 16;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (t + j < ex) c[j] = codes[ox + t + j];
        }
        if (reg) {
            T rc[4];
            coef_recover(coef_by_rank + (uint64_t)rank[task] * 4, cl, rc);
            T v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                v[j] = 0;  // (code 0: patched from the list)
                if (t + j < ex && c[j]) v[j] = ref_recover(reg_predict(rc, 0u, 0u, t + j), (int)c[j], p.eb, (int)p.radius);
                if (t + j + 1 == ex) {  // the block's last element: its lattice value restarts the sum over the blocks
                    Q qt = 0;
                    if (c[j]) {
                        bool bad;
                        qt = lat.quant(v[j], bad);
                        if (bad) qt = 0;
                    }
                    agg[2 * (uint64_t)task] = qt;
                }
            }
            if (t + 8 <= ex && out_aligned) {
                if (sizeof(T) == 4) {
                    float4 *o4 = reinterpret_cast<float4 *>(out + ox + t);
                    o4[0] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
                    o4[1] = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
                } else {
                    double2 *o2 = reinterpret_cast<double2 *>(out + ox + t);
#pragma unroll
                    for (int j = 0; j < 4; j++) o2[j] = make_double2((double)v[2 * j], (double)v[2 * j + 1]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (t + j < ex) out[ox + t + j] = v[j];
            }
        }
        UQ sum = 0;
        if (!reg) {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (t + j < ex) sum += c[j] ? (UQ)(Q)((int)c[j] - (int)p.radius) : (UQ)deltas[ox + t + j];
        }
        const UQ tot = row16_incl_scan(sum);  // (lane 15 of the row holds the block's sum)
        if (live && !reg && li == 15) agg[2 * (uint64_t)task] = (Q)tot;
    }
}
// ---- 1-D with second-order Lorenzo in the predictor set (round 4; LorenzoPredictor.hpp:69-71, chosen by the reference's tuner for
// 1-D arrays: SZAlgoInterp.hpp:232-247) ----
// A second-order block needs the TWO lattice values left of it. The state carried along the array is the pair (a, b) = (q~ of the
// element left of the block, q~ of the one before that); a block of m elements maps it to the pair behind its last element:
//   regression:  (a, b) -> (r_m, r_{m-1})                                   its own lattice values, nothing of the inflow
//   Lorenzo-1:   (a, b) -> (a + S_m, a + S_{m-1})                           S_k: running sum of the block's deltas, S_0 = 0
//   Lorenzo-2:   (a, b) -> ((m + 1) a - m b + P_m, m a - (m - 1) b + P_{m-1})   P_k: running sum of S (q_k = (k+1) a - k b + P_k)
// — affine maps x -> M x + v with a matrix that the block's choice and length give, and affine maps compose associatively (in
// the wrap-around integers the lattice lives in): the segmented prefix sum of the first-order path becomes a scan of compositions.
// Reduce-then-scan over tiles of 1024 blocks: k_blkn2_pre leaves every block's v (two words), k_blkn2_tile composes a tile,
// k_blkn2_top runs over the tiles, k_blkn2_apply scans its tile again from the tile's inflow and rebuilds the Lorenzo blocks.
// carry buffer: Q v[2 * nblocks], then per tile eight words (M, v of the tile; the state in front of it).
template <typename UQ>
struct BlkAff {
    UQ m00, m01, m10, m11, v0, v1;
};
template <typename UQ>
__device__ __forceinline__ BlkAff<UQ> aff_after(const BlkAff<UQ> &g, const BlkAff<UQ> &f) {  // first f, then g
    BlkAff<UQ> r;
    r.m00 = g.m00 * f.m00 + g.m01 * f.m10;
    r.m01 = g.m00 * f.m01 + g.m01 * f.m11;
    r.m10 = g.m10 * f.m00 + g.m11 * f.m10;
    r.m11 = g.m10 * f.m01 + g.m11 * f.m11;
    r.v0 = g.m00 * f.v0 + g.m01 * f.v1 + g.v0;
    r.v1 = g.m10 * f.v0 + g.m11 * f.v1 + g.v1;
    return r;
}
template <typename UQ>
__device__ __forceinline__ BlkAff<UQ> aff_ident() {
    return BlkAff<UQ>{1, 0, 0, 1, 0, 0};
}
template <typename UQ>
__device__ __forceinline__ BlkAff<UQ> aff_const(UQ v0, UQ v1) {
    return BlkAff<UQ>{0, 0, 0, 0, v0, v1};
}
template <typename UQ>
__device__ __forceinline__ BlkAff<UQ> aff_of_block(uint32_t sel, uint32_t m, UQ v0, UQ v1) {
    if (sel == 2) return aff_const<UQ>(v0, v1);
    if (sel == 1) return BlkAff<UQ>{(UQ)m + 1, (UQ)0 - (UQ)m, (UQ)m, (UQ)0 - (UQ)(m - 1), v0, v1};
    return BlkAff<UQ>{1, 0, 1, 0, v0, v1};
}
template <typename UQ>
__device__ __forceinline__ UQ shfl_up_uq(UQ v, int off) {
    if (sizeof(UQ) == 8) return (UQ)__shfl_up((long long)v, off);
    return (UQ)__shfl_up((int)v, off);
}
template <typename UQ>
__device__ __forceinline__ BlkAff<UQ> aff_shfl_up(const BlkAff<UQ> &x, int off) {
    return BlkAff<UQ>{shfl_up_uq(x.m00, off), shfl_up_uq(x.m01, off), shfl_up_uq(x.m10, off), shfl_up_uq(x.m11, off), shfl_up_uq(x.v0, off), shfl_up_uq(x.v1, off)};
}
// inclusive scan of compositions over a workgroup of 1024 threads (thread order = block order); inflow: the map in front of thread 0;
// excl: the composition in front of this thread. ws: [16] in LDS.
template <typename UQ>
__device__ __forceinline__ BlkAff<UQ> aff_wg_scan(BlkAff<UQ> x, const BlkAff<UQ> &inflow, BlkAff<UQ> &excl, BlkAff<UQ> *ws) {
    const int lane = lane_id();
    const uint32_t w = threadIdx.x / WAVE;
    for (int off = 1; off < WAVE; off <<= 1) {
        const BlkAff<UQ> y = aff_shfl_up(x, off);
        if (lane >= off) x = aff_after(x, y);
    }
    if (lane == WAVE - 1) ws[w] = x;
    __syncthreads();
    BlkAff<UQ> pre = inflow;
    for (uint32_t k = 0; k < w; k++) pre = aff_after(ws[k], pre);
    const BlkAff<UQ> incl = aff_after(x, pre);
    excl = aff_shfl_up(incl, 1);
    if (lane == 0) excl = pre;
    __syncthreads();
    return incl;
}
template <typename Q>
__device__ __forceinline__ Q *blkn2_tiles(void *carry, uint32_t nblocks) { return reinterpret_cast<Q *>(carry) + 2 * (uint64_t)nblocks; }

// every block's v: a regression block is decoded here (final values out, the lattice values of its last two elements kept); a Lorenzo
// block leaves the sums of its deltas. A wave per block.
template <typename T>
__global__ __launch_bounds__(256) void k_blkn2_pre(const uint16_t *__restrict__ codes, const void *deltas_, void *d_out, szk_blk_params p, uint32_t nblocks,
                                                   const uint32_t *__restrict__ rank, const int64_t *__restrict__ coef_by_rank) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    Q *agg = reinterpret_cast<Q *>(p.carry);
    const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
    for (uint32_t task = blockIdx.x * 4 + threadIdx.x / WAVE; task < nblocks; task += gridDim.x * 4) {
        const BlkGeom g = blk_geom(p, task);
        const uint32_t m = g.ex, sel = p.sel[task];
        if (sel == 2) {
            T rc[4];
            coef_recover(coef_by_rank + (uint64_t)rank[task] * 4, cl, rc);
            for (uint32_t t = lane; t < m; t += WAVE) {
                const uint32_t code = codes[g.coff + t];
                Q qt = 0;
                T val = 0;  // (code 0: patched from the list)
                if (code) {
                    bool bad;
                    val = ref_recover(reg_predict(rc, 0u, 0u, t), (int)code, p.eb, (int)p.radius);
                    qt = lat.quant(val, bad);
                    if (bad) qt = 0;
                }
                reinterpret_cast<T *>(d_out)[g.ox + t] = val;
                if (t == m - 1) agg[2 * (uint64_t)task] = qt;
                if (t + 2 == m) agg[2 * (uint64_t)task + 1] = qt;
            }
        } else {
            UQ s = 0, w = 0, last = 0;  // S_m; sum of (m - t) d_t = P_m; the last delta
            for (uint32_t t = lane; t < m; t += WAVE) {
                const uint32_t code = codes[g.coff + t];
                const UQ d = code ? (UQ)(Q)((int)code - (int)p.radius) : (UQ)deltas[g.coff + t];
                s += d;
                w += (UQ)(m - t) * d;
                if (t == m - 1) last = d;
            }
            s = wave_sum(s);
            w = wave_sum(w);
            last = wave_sum(last);
            if (lane == 0) {
                agg[2 * (uint64_t)task] = (Q)(sel == 1 ? w : s);
                agg[2 * (uint64_t)task + 1] = (Q)(sel == 1 ? w - s : s - last);  // P_{m-1} = P_m - S_m; S_{m-1} = S_m - d_m
            }
        }
    }
}
// k_blkn2_pre by rows of 16 lanes (blocks of up to 128 values, a multiple of 8: the default 128; four blocks per wave, eight values per
// lane from one 16-byte load of codes — the wave-per-block form reads two bytes per lane and load: 262 us at 2^27 values)
template <typename T>
__global__ __launch_bounds__(256) void k_blkn2_pre_rows(const uint16_t *__restrict__ codes, const void *deltas_, void *d_out, szk_blk_params p, uint32_t nblocks,
                                                        const uint32_t *__restrict__ rank, const int64_t *__restrict__ coef_by_rank) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    const uint32_t lane = (uint32_t)lane_id(), row = lane >> 4, li = lane & 15u;
    const Lattice<T> lat(p.lat);
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    Q *agg = reinterpret_cast<Q *>(p.carry);
    T *out = reinterpret_cast<T *>(d_out);
    const bool out_aligned = (reinterpret_cast<uintptr_t>(d_out) & 15u) == 0;
    const CoefLat cl = coef_lat(p.eb, p.B, p.ndim);
    const uint32_t n = (uint32_t)p.d[2];
    for (uint32_t base = (blockIdx.x * 4 + threadIdx.x / WAVE) * 4; base < nblocks; base += gridDim.x * 16) {  // (wave-uniform)
        const uint32_t task = base + row;
        const bool live = task < nblocks;
        const uint32_t sel = live ? p.sel[task] : 0u;
        const bool reg = live && sel == 2;
        const uint32_t ox = live ? task * p.B : 0u;
        const uint32_t ex = live ? min(p.B, n - ox) : 0u;
        const uint32_t t = li * 8;
        uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t + 8 <= ex) {
            const uint4 w = *reinterpret_cast<const uint4 *>(codes + ox + t);
            c[0] = w.x & 0xFFFFu; c[1] = w.x >> 16; c[2] = w.y & 0xFFFFu; c[3] = w.y >> 16;
            c[4] = w.z & 0xFFFFu; c[5] = w.z >> 16; c[6] = w.w & 0xFFFFu; c[7] = w.w >> 16;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (t + j < ex) c[j] = codes[ox + t + j];
        }
        if (reg) {
            T rc[4];
            coef_recover(coef_by_rank + (uint64_t)rank[task] * 4, cl, rc);
            T v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                v[j] = 0;  // (code 0: patched from the list)
                if (t + j < ex && c[j]) v[j] = ref_recover(reg_predict(rc, 0u, 0u, t + j), (int)c[j], p.eb, (int)p.radius);
                if (t + j + 1 == ex || t + j + 2 == ex) {  // the block's last two elements: their lattice values are what it leaves behind
                    Q qt = 0;
                    if (c[j]) {
                        bool bad;
                        qt = lat.quant(v[j], bad);
                        if (bad) qt = 0;
                    }
                    agg[2 * (uint64_t)task + (t + j + 1 == ex ? 0 : 1)] = qt;
                }
            }
            if (t + 8 <= ex && out_aligned) {
                if (sizeof(T) == 4) {
                    float4 *o4 = reinterpret_cast<float4 *>(out + ox + t);
                    o4[0] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
                    o4[1] = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
                } else {
                    double2 *o2 = reinterpret_cast<double2 *>(out + ox + t);
#pragma unroll
                    for (int j = 0; j < 4; j++) o2[j] = make_double2((double)v[2 * j], (double)v[2 * j + 1]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (t + j < ex) out[ox + t + j] = v[j];
            }
        }
        UQ s = 0, w = 0, last = 0;  // S_m; sum of (m - t) d_t = P_m; the last delta
        if (!reg) {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (t + j < ex) {
                    const UQ d = c[j] ? (UQ)(Q)((int)c[j] - (int)p.radius) : (UQ)deltas[ox + t + j];
                    s += d;
                    w += (UQ)(ex - (t + j)) * d;
                    if (t + j + 1 == ex) last = d;
                }
        }
        s = row16_incl_scan(s);  // (lane 15 of the row holds the block's sums)
        w = row16_incl_scan(w);
        last = row16_incl_scan(last);
        if (live && !reg && li == 15) {
            agg[2 * (uint64_t)task] = (Q)(sel == 1 ? w : s);
            agg[2 * (uint64_t)task + 1] = (Q)(sel == 1 ? w - s : s - last);
        }
    }
}
template <typename Q>
__global__ __launch_bounds__(1024) void k_blkn2_tile(const uint8_t *__restrict__ sel, uint32_t nblocks, uint32_t B, uint32_t n, void *carry) {
    using UQ = typename std::make_unsigned<Q>::type;
    __shared__ BlkAff<UQ> ws[16];
    const Q *agg = reinterpret_cast<const Q *>(carry);
    Q *tile = blkn2_tiles<Q>(carry, nblocks);
    const uint32_t b = blockIdx.x * BLKN_TILE + threadIdx.x;
    BlkAff<UQ> x = aff_ident<UQ>(), excl;
    if (b < nblocks) x = aff_of_block<UQ>(sel[b], min(B, n - b * B), (UQ)agg[2 * (uint64_t)b], (UQ)agg[2 * (uint64_t)b + 1]);
    const BlkAff<UQ> incl = aff_wg_scan<UQ>(x, aff_ident<UQ>(), excl, ws);
    if (threadIdx.x == BLKN_TILE - 1) {
        Q *t8 = tile + 8 * (uint64_t)blockIdx.x;
        t8[0] = (Q)incl.m00;
        t8[1] = (Q)incl.m01;
        t8[2] = (Q)incl.m10;
        t8[3] = (Q)incl.m11;
        t8[4] = (Q)incl.v0;
        t8[5] = (Q)incl.v1;
    }
}
template <typename Q>
__global__ __launch_bounds__(1024) void k_blkn2_top(uint32_t ntiles, Q *__restrict__ tile) {
    using UQ = typename std::make_unsigned<Q>::type;
    __shared__ BlkAff<UQ> ws[16];
    __shared__ UQ run_s[2];
    UQ r0 = 0, r1 = 0;  // the state in front of the array: zeros (LorenzoPredictor.hpp: the padding)
    for (uint32_t base = 0; base < ntiles; base += 1024) {
        const uint32_t t = base + threadIdx.x;
        BlkAff<UQ> x = aff_ident<UQ>(), excl;
        if (t < ntiles) {
            const Q *t8 = tile + 8 * (uint64_t)t;
            x = BlkAff<UQ>{(UQ)t8[0], (UQ)t8[1], (UQ)t8[2], (UQ)t8[3], (UQ)t8[4], (UQ)t8[5]};
        }
        const BlkAff<UQ> incl = aff_wg_scan<UQ>(x, aff_const<UQ>(r0, r1), excl, ws);
        if (t < ntiles) {
            tile[8 * (uint64_t)t + 6] = (Q)excl.v0;  // (behind a constant map every composition is constant: v IS the state)
            tile[8 * (uint64_t)t + 7] = (Q)excl.v1;
        }
        if (threadIdx.x == 1023) {
            run_s[0] = incl.v0;
            run_s[1] = incl.v1;
        }
        __syncthreads();
        r0 = run_s[0];
        r1 = run_s[1];
        __syncthreads();
    }
}
template <typename T, bool ROWS>
__global__ __launch_bounds__(1024) void k_blkn2_apply(const uint16_t *__restrict__ codes, const void *deltas_, void *d_out, szk_blk_params p, uint32_t nblocks) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    __shared__ BlkAff<UQ> ws[16];
    __shared__ UQ sa[BLKN_TILE], sb[BLKN_TILE];
    const Lattice<T> lat(p.lat);
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    const Q *agg = reinterpret_cast<const Q *>(p.carry);
    const Q *tile = blkn2_tiles<Q>(p.carry, nblocks);
    const uint32_t n = (uint32_t)p.d[2];
    const uint32_t b = blockIdx.x * BLKN_TILE + threadIdx.x;
    BlkAff<UQ> x = aff_ident<UQ>(), excl;
    if (b < nblocks) x = aff_of_block<UQ>(p.sel[b], min(p.B, n - b * p.B), (UQ)agg[2 * (uint64_t)b], (UQ)agg[2 * (uint64_t)b + 1]);
    (void)aff_wg_scan<UQ>(x, aff_const<UQ>((UQ)tile[8 * (uint64_t)blockIdx.x + 6], (UQ)tile[8 * (uint64_t)blockIdx.x + 7]), excl, ws);
    sa[threadIdx.x] = excl.v0;
    sb[threadIdx.x] = excl.v1;
    __syncthreads();
    const int lane = lane_id();
    if (ROWS) {  // (the launcher: blocks of up to 128 values, a multiple of 8)
        // by rows of 16 lanes: four blocks per wave, eight values per lane (one 16-byte load of codes, the running sums in registers,
        // the lanes' totals scanned inside the DPP row — twice for a second-order block: S, then P = the running sum of S)
        T *out = reinterpret_cast<T *>(d_out);
        const bool out_aligned = (reinterpret_cast<uintptr_t>(d_out) & 15u) == 0;
        const uint32_t row = (uint32_t)lane >> 4, li = (uint32_t)lane & 15u;
        for (uint32_t k0 = (threadIdx.x / WAVE) * 4; k0 < BLKN_TILE; k0 += 64) {  // (wave-uniform)
            const uint32_t k = k0 + row, task = blockIdx.x * BLKN_TILE + k;
            const uint32_t sel = task < nblocks ? p.sel[task] : 2u;
            const bool live = task < nblocks && sel != 2;
            const uint32_t ox = task < nblocks ? task * p.B : 0u;
            const uint32_t ex = live ? min(p.B, n - ox) : 0u;
            const uint32_t t = li * 8;
            UQ d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (t + 8 <= ex) {
                const uint4 w = *reinterpret_cast<const uint4 *>(codes + ox + t);  // (ox and t are multiples of 8: 16-byte aligned)
                const uint32_t c[8] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16, w.z & 0xFFFFu, w.z >> 16, w.w & 0xFFFFu, w.w >> 16};
#pragma unroll
                for (int j = 0; j < 8; j++) d[j] = c[j] ? (UQ)(Q)((int)c[j] - (int)p.radius) : (UQ)deltas[ox + t + j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (t + j < ex) {
                        const uint32_t c = codes[ox + t + j];
                        d[j] = c ? (UQ)(Q)((int)c - (int)p.radius) : (UQ)deltas[ox + t + j];
                    }
            }
#pragma unroll
            for (int j = 1; j < 8; j++) d[j] += d[j - 1];  // S inside the lane
            const UQ s_before = row16_incl_scan(d[7]) - d[7];
            const UQ a = live ? sa[k] : (UQ)0, bq = live ? sb[k] : (UQ)0;
            UQ q[8];
            if (sel == 1) {
                UQ ps[8];
                ps[0] = d[0] + s_before;
#pragma unroll
                for (int j = 1; j < 8; j++) ps[j] = ps[j - 1] + d[j] + s_before;  // P inside the lane (every S carries the lanes before)
                const UQ p_before = row16_incl_scan(ps[7]) - ps[7];
#pragma unroll
                for (int j = 0; j < 8; j++) q[j] = (UQ)(t + j + 2) * a - (UQ)(t + j + 1) * bq + ps[j] + p_before;
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) q[j] = a + d[j] + s_before;
            }
            T v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = lat.dequant((Q)q[j]);
            if (t + 8 <= ex && out_aligned) {
                if (sizeof(T) == 4) {
                    float4 *o4 = reinterpret_cast<float4 *>(out + ox + t);
                    o4[0] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
                    o4[1] = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
                } else {
                    double2 *o2 = reinterpret_cast<double2 *>(out + ox + t);
#pragma unroll
                    for (int j = 0; j < 4; j++) o2[j] = make_double2((double)v[2 * j], (double)v[2 * j + 1]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (t + j < ex) out[ox + t + j] = v[j];
            }
        }
        return;
    }
    for (uint32_t k = threadIdx.x / WAVE; k < BLKN_TILE; k += 16) {
        const uint32_t task = blockIdx.x * BLKN_TILE + k;
        if (task >= nblocks) break;
        const uint32_t sel = p.sel[task];
        if (sel == 2) continue;
        const BlkGeom g = blk_geom(p, task);
        const UQ a = sa[k], bq = sb[k];
        UQ c1 = 0, cp = 0;  // running S and P behind the elements done so far
        for (uint32_t t0 = 0; t0 < g.ex; t0 += WAVE) {
            const uint32_t t = t0 + lane;
            UQ d = 0;
            if (t < g.ex) {
                const uint32_t code = codes[g.coff + t];
                d = code ? (UQ)(Q)((int)code - (int)p.radius) : (UQ)deltas[g.coff + t];
            }
            const UQ s_k = wave_incl_scan(d) + c1;  // S_{t+1}
            UQ q;
            if (sel == 1) {
                const UQ p_k = wave_incl_scan(s_k) + cp;  // P_{t+1}
                q = (UQ)(t + 2) * a - (UQ)(t + 1) * bq + p_k;
                cp = (UQ)__shfl((long long)p_k, WAVE - 1);
            } else {
                q = a + s_k;
            }
            c1 = (UQ)__shfl((long long)s_k, WAVE - 1);
            if (t < g.ex) reinterpret_cast<T *>(d_out)[g.ox + t] = lat.dequant((Q)q);
        }
    }
}
// 2-D: the Lorenzo blocks of one front (by + bx = diag), a wave per block: the deltas in LDS, sums along x (inflow: Dy q~ of the
// column left of the block), then along y (inflow: q~ of the row above it)
#define BLKN_MAXB2 32u
template <typename T>
__global__ __launch_bounds__(256) void k_blkn_decode2(const void *deltas_, void *d_out, szk_blk_params p, uint32_t diag, uint32_t by_lo, uint32_t nfront) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    __shared__ Q s_a[4][BLKN_MAXB2 * (BLKN_MAXB2 + 1)];
    const int lane = lane_id();
    const uint32_t wv = threadIdx.x / WAVE;
    Q *sa = s_a[wv];
    const uint32_t k = blockIdx.x * 4 + wv;
    if (k >= nfront) return;
    const uint32_t by = by_lo + k, bx = diag - by;
    const uint32_t task = by * p.nb[2] + bx;
    if (p.sel[task] == 2) return;
    const BlkGeom g = blk_geom(p, task);
    const uint64_t d2 = p.d[2];
    Q *qout = reinterpret_cast<Q *>(d_out);
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    const uint32_t pitch = g.ex + 1, nown = g.ey * g.ex;
    for (uint32_t t = lane; t < nown; t += WAVE) {
        const uint32_t j = t / g.ex, i = t - j * g.ex;
        sa[j * pitch + i] = deltas[g.coff + t];
    }
    wave_lds_fence();
    for (uint32_t j = lane; j < g.ey; j += WAVE) {
        const uint64_t y = g.oy + j;
        UQ a = 0;
        if (g.ox) {
            a = (UQ)qout[y * d2 + g.ox - 1];
            if (y) a -= (UQ)qout[(y - 1) * d2 + g.ox - 1];
        }
        for (uint32_t i = 0; i < g.ex; i++) {
            a += (UQ)sa[j * pitch + i];
            sa[j * pitch + i] = (Q)a;
        }
    }
    wave_lds_fence();
    for (uint32_t i = lane; i < g.ex; i += WAVE) {
        const uint64_t x = g.ox + i;
        UQ q = g.oy ? (UQ)qout[(uint64_t)(g.oy - 1) * d2 + x] : (UQ)0;
        for (uint32_t j = 0; j < g.ey; j++) {
            q += (UQ)sa[j * pitch + i];
            qout[(uint64_t)(g.oy + j) * d2 + x] = (Q)q;
        }
    }
}

// 2-D with second-order Lorenzo in the set (round 4): the plain form — a wave per Lorenzo block of one front, the block's deltas and TWO
// halo rows / columns of finished q~ in LDS; the pass along x takes the inflow Dy^m q~ of the two halo columns and runs the block's
// recurrence (order m = 1 or 2 by the block's choice) along its rows, the pass along y the same along its columns from the halo
// rows. In place: the x pass writes own entries only, and everything it reads of the rows above lies in the halo columns.
template <typename T>
__global__ __launch_bounds__(256) void k_blkn_decode2s(const void *deltas_, void *d_out, szk_blk_params p, uint32_t diag, uint32_t by_lo, uint32_t nfront) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    constexpr uint32_t TE = BLKN_MAXB2 + 2, PITCH = TE + 1;
    __shared__ Q s_q[4][TE * PITCH];
    const int lane = lane_id();
    const uint32_t wv = threadIdx.x / WAVE;
    Q *sq = s_q[wv];
    const uint32_t k = blockIdx.x * 4 + wv;
    if (k >= nfront) return;
    const uint32_t by = by_lo + k, bx = diag - by;
    const uint32_t task = by * p.nb[2] + bx;
    const uint32_t sel = p.sel[task];
    if (sel == 2) return;
    const BlkGeom g = blk_geom(p, task);
    const uint64_t d2 = p.d[2];
    Q *qout = reinterpret_cast<Q *>(d_out);
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    const uint32_t th = g.ey + 2, tw = g.ex + 2;
    for (uint32_t l = lane; l < th * tw; l += WAVE) {
        const uint32_t j = l / tw, i = l - j * tw;
        Q v = 0;
        if (j >= 2 && i >= 2) {
            v = deltas[g.coff + (j - 2) * g.ex + (i - 2)];
        } else {
            const int64_t y = (int64_t)g.oy + j - 2, x = (int64_t)g.ox + i - 2;
            if (y >= 0 && x >= 0) v = qout[(uint64_t)y * d2 + (uint64_t)x];
        }
        sq[j * PITCH + i] = v;
    }
    wave_lds_fence();
    const int m = sel == 1 ? 2 : 1;
    for (uint32_t j = 2 + lane; j < th; j += WAVE) {  // along x
        UQ a[2];
#pragma unroll
        for (uint32_t c = 0; c < 2; c++) {
            UQ sx = 0;
            for (int kk = 0; kk <= m; kk++) sx += (UQ)((Q)lz_w(m, kk) * sq[(j - kk) * PITCH + c]);
            a[c] = sx;
        }
        UQ p2 = a[0], p1 = a[1];
        for (uint32_t i = 2; i < tw; i++) {
            const UQ in = (UQ)sq[j * PITCH + i];
            const UQ v = m == 1 ? p1 + in : (UQ)(2 * p1 - p2 + in);
            sq[j * PITCH + i] = (Q)v;
            p2 = p1;
            p1 = v;
        }
    }
    wave_lds_fence();
    for (uint32_t i = 2 + lane; i < tw; i += WAVE) {  // along y: the result is q~
        UQ p2 = (UQ)sq[i], p1 = (UQ)sq[PITCH + i];
        for (uint32_t j = 2; j < th; j++) {
            const UQ in = (UQ)sq[j * PITCH + i];
            const UQ v = m == 1 ? p1 + in : (UQ)(2 * p1 - p2 + in);
            qout[(uint64_t)(g.oy + j - 2) * d2 + (g.ox + i - 2)] = (Q)v;
            p2 = p1;
            p1 = v;
        }
    }
}
// 2-D, block edges up to 16: GROUPS of 4 x 4 blocks per workgroup, like the 3-D decoder's groups — the chain of fronts is what a
// block stream's decoding costs (a launch + a block's latency each), and a group quarters it: 255 launches instead of 1023 at
// 8192^2. The group's tile (its deltas, the q~ of its regression blocks, one halo row and column of finished q~) sits in LDS;
// seven inner steps (ly + lx = 0..6, a barrier in between) invert the Lorenzo blocks in place, a wave per block: sums along x in
// the DPP rows of 16 lanes (four rows of the block at a time), then along y the same way on the transposed assignment.
#define BLKN_G 4u
template <typename T>
__global__ __launch_bounds__(256) void k_blkn_decode2g(const void *deltas_, void *d_out, szk_blk_params p, uint32_t diag, uint32_t gy_lo) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    constexpr uint32_t TE = BLKN_G * 16 + 1, PITCH = TE + 1;
    __shared__ Q sq[TE * PITCH];
    const uint32_t gy = gy_lo + blockIdx.x, gx = diag - gy;
    const uint32_t B = p.B;
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    const uint32_t y0 = gy * BLKN_G * B, x0 = gx * BLKN_G * B;
    const uint32_t hy = (uint32_t)min((uint64_t)(BLKN_G * B), d1 - y0), hx = (uint32_t)min((uint64_t)(BLKN_G * B), d2 - x0);
    Q *qout = reinterpret_cast<Q *>(d_out);
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    // the tile: 17 elements per thread, every load issued before the first LDS store (a loop of load -> LDS store waits for every
    // load in turn: 17 round trips were 20 of the kernel's 28 us)
    __shared__ uint8_t s_sel[BLKN_G * BLKN_G];
    {
        // ONE round trip to memory: an element's two possible sources (its delta; its q~ if the block is a regression block or
        // the element lies in the halo) and its block's choice are requested together, for all 17 elements of the thread,
        // and the choice picks afterwards (choice -> address -> value were three dependent trips of ~2 us each)
        constexpr int NR = (TE * TE + 255) / 256;
        Q v[NR], vq[NR];
        uint8_t sl[NR];
        const Q *sd[NR], *sv[NR];
        const uint8_t *ss[NR];
        const uint8_t *selp = p.sel;  // (threads 0..15: the choice of a block of the group, for the inner steps)
        if (threadIdx.x < BLKN_G * BLKN_G) {
            const uint32_t by = gy * BLKN_G + threadIdx.x / BLKN_G, bx = gx * BLKN_G + threadIdx.x % BLKN_G;
            if (by < p.nb[1] && bx < p.nb[2]) selp = p.sel + (by * p.nb[2] + bx);
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const uint32_t idx = threadIdx.x + 256u * r;
            const uint32_t ty = idx / TE, tx = idx - ty * TE;
            sd[r] = sv[r] = nullptr;
            ss[r] = nullptr;
            if (ty > hy || tx > hx) continue;
#ifdef LAB_G_NOLOAD
            if (p.B) continue;
#endif
            const int64_t y = (int64_t)y0 + ty - 1, x = (int64_t)x0 + tx - 1;
            if (y >= 0 && x >= 0) sv[r] = qout + ((uint64_t)y * d2 + (uint64_t)x);
            if (ty && tx) {
                const uint32_t by = (uint32_t)y / B, bx = (uint32_t)x / B;
                const uint32_t oy = by * B, ey = min(B, (uint32_t)d1 - oy);
                const uint32_t ox = bx * B, ex = min(B, (uint32_t)d2 - ox);
                sd[r] = deltas + ((uint64_t)oy * d2 + (uint64_t)ey * ox + ((uint32_t)y - oy) * ex + ((uint32_t)x - ox));
                ss[r] = p.sel + (by * p.nb[2] + bx);
            }
        }
        const uint8_t mysel = *selp;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            v[r] = *(sd[r] ? sd[r] : deltas);
            vq[r] = *(sv[r] ? sv[r] : deltas);
            sl[r] = *(ss[r] ? ss[r] : p.sel);
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            if (!sd[r] || sl[r] == 2) v[r] = sv[r] ? vq[r] : (Q)0;  // halo / regression block: q~ (zero outside the array)
        }
        if (threadIdx.x < BLKN_G * BLKN_G) s_sel[threadIdx.x] = mysel;  // (blocks beyond the array's edge are never visited)
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const uint32_t idx = threadIdx.x + 256u * r;
            const uint32_t ty = idx / TE, tx = idx - ty * TE;
            if (ty <= hy && tx <= hx) sq[ty * PITCH + tx] = v[r];
        }
    }
    __syncthreads();
    const uint32_t nly = (hy + B - 1) / B, nlx = (hx + B - 1) / B;
    const int lane = lane_id();
    const uint32_t w = threadIdx.x / WAVE, rr = (uint32_t)lane >> 4, cc = (uint32_t)lane & 15u;
#ifdef LAB_G_NOSTEPS
    if (p.B == 0)
#endif
    for (uint32_t step = 0; step + 1 < nly + nlx; step++) {
        const uint32_t ly = w, lx = step - ly;
        if (ly < nly && ly <= step && lx < nlx && s_sel[ly * BLKN_G + lx] != 2) {
            const uint32_t ty0 = 1 + ly * B, tx0 = 1 + lx * B;
            const uint32_t ey = min(B, hy - ly * B), ex = min(B, hx - lx * B);
            // (four independent chains of LDS read -> row scan -> LDS write, unrolled with predicates: as a loop with a data-dependent
            // trip count they ran one after the other, 1.07 us per inner step)
            UQ v[4], in[4];
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) {  // along x: a row of the block per DPP row
                const uint32_t j = rr + 4 * m, at = (ty0 + min(j, ey - 1)) * PITCH + tx0;
                v[m] = cc < ex ? (UQ)sq[at + cc] : (UQ)0;
                in[m] = (UQ)sq[at - 1] - (UQ)sq[at - PITCH - 1];
            }
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) v[m] = row16_incl_scan(v[m]) + in[m];
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) {
                const uint32_t j = rr + 4 * m;
                if (j < ey && cc < ex) sq[(ty0 + j) * PITCH + tx0 + cc] = (Q)v[m];
            }
            wave_lds_fence();
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) {  // along y: a column per DPP row
                const uint32_t i = min(rr + 4 * m, ex - 1), at = ty0 * PITCH + tx0 + i;
                v[m] = cc < ey ? (UQ)sq[at + cc * PITCH] : (UQ)0;
                in[m] = (UQ)sq[at - PITCH];
            }
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) v[m] = row16_incl_scan(v[m]) + in[m];
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) {
                const uint32_t i = rr + 4 * m;
                if (i < ex && cc < ey) sq[ty0 * PITCH + tx0 + i + cc * PITCH] = (Q)v[m];
            }
        }
        __syncthreads();
    }
#ifdef LAB_G_NOSTORE
    if (p.B == 0)
#endif
    for (uint32_t ty = threadIdx.x / WAVE; ty < hy; ty += 4) {
        const uint32_t tx = threadIdx.x % WAVE;
        if (tx < hx) qout[(uint64_t)(y0 + ty) * d2 + (x0 + tx)] = sq[(ty + 1) * PITCH + tx + 1];
    }
}

// 2-D, the chain of fronts in ONE launch (the 2-D form of k_blk_wave3, first-order Lorenzo + regression, block edges up to 16): groups
// of 4 x 4 blocks handed out by a ticket counter in the fronts' order to workgroups that stay; a group inverts its Lorenzo blocks with
// a zero halo while it has nothing to wait for (P: row sums, then column sums, in the DPP rows), waits for the flags of its three
// lower neighbours, takes their shells (a group's last row and column, left in the work array in the codes' order) and closes the
// blocks in the 2-D form of the identity above k_blk_local3: q(j, i) = P(j, i) + q(-1, i) + q(j, -1) - q(-1, -1) — the blocks' last
// rows and columns first, inner front by inner front, the shell out and the flag, then every interior at once, and the final values
// straight to the output (no k_blk_final pass). k_blkn_pre(to_work) ran before: the regression blocks' lattice values are in the work
// array, their final values in the output. ctl as for k_blk_wave3.
template <typename T>
__global__ __launch_bounds__(256, 4) void k_blkn_wave2(const uint16_t *__restrict__ codes, void *work_, void *d_out, szk_blk_params p, uint32_t *ctl, uint32_t nslots) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    constexpr uint32_t OW = BLKN_G * 16, TE = OW + 1, PITCH = TE + 1;
    constexpr int NR = (OW * OW) / 256;
    __shared__ Q sq[TE * PITCH];
    __shared__ uint8_t s_sel[BLKN_G * BLKN_G];
    __shared__ uint32_t s_ticket;
    const Lattice<T> lat(p.lat);
    const uint32_t B = p.B, inv = (65536u + B - 1) / B;  // (t / B for t < 64: (t * inv) >> 16)
    const uint64_t d1 = p.d[1], d2 = p.d[2];
    const uint32_t ng1 = (p.nb[1] + BLKN_G - 1) / BLKN_G, ng2 = (p.nb[2] + BLKN_G - 1) / BLKN_G;
    Q *work = reinterpret_cast<Q *>(work_);
    T *tout = reinterpret_cast<T *>(d_out);
    uint32_t *flags = ctl + 4;
    const int lane = lane_id();
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)), rr = (uint32_t)lane >> 4, cc = (uint32_t)lane & 15u;
    // where array position (y, x) lies in the work array (the codes' order: block by block, a block's elements in raster order)
    auto work_at = [&](uint32_t y, uint32_t x) {
        const uint32_t by = y / B, bx = x / B;
        const uint32_t oy = by * B, ey = min(B, (uint32_t)d1 - oy), ox = bx * B, ex = min(B, (uint32_t)d2 - ox);
        return (uint64_t)oy * d2 + (uint64_t)ey * ox + (uint64_t)(y - oy) * ex + (x - ox);
    };
    uint32_t dcur = 0, start = 0;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ctl[0], 1u);
    for (;;) {
        __syncthreads();
        const uint32_t ticket = s_ticket;
        if (ticket >= nslots) break;
        uint32_t gy_lo, cnt;
        for (;;) {
            gy_lo = dcur >= ng2 ? dcur - (ng2 - 1) : 0;
            const uint32_t gy_hi = dcur < ng1 - 1 ? dcur : ng1 - 1;
            cnt = gy_hi - gy_lo + 1;
            if (ticket < start + cnt) break;
            start += cnt;
            dcur++;
        }
        const uint32_t gy = gy_lo + (ticket - start), gx = dcur - gy;
        const uint32_t y0 = gy * BLKN_G * B, x0 = gx * BLKN_G * B;
        const uint32_t hy = (uint32_t)min((uint64_t)(BLKN_G * B), d1 - y0), hx = (uint32_t)min((uint64_t)(BLKN_G * B), d2 - x0);
        const uint32_t nly = (hy + B - 1) / B, nlx = (hx + B - 1) / B;
        // ---- (1) nothing to wait for yet: the group's own values from the work array, the blocks' choices ----
        if (threadIdx.x < BLKN_G * BLKN_G) {
            const uint32_t by = gy * BLKN_G + threadIdx.x / BLKN_G, bx = gx * BLKN_G + threadIdx.x % BLKN_G;
            uint8_t sl = 255;
            if (by < p.nb[1] && bx < p.nb[2]) sl = p.sel[by * p.nb[2] + bx] == 2 ? 2 : 0;  // (this path has no second-order member: anything else is first-order Lorenzo, as in k_blkn_pre)
            s_sel[threadIdx.x] = sl;
        }
        {
            // (the codes themselves: a delta is code - radius; the work array holds what the codes do not — the far deltas, scattered from
            // their list, and the regression blocks' lattice values — so no expanded copy of the deltas is made)
            Q v[NR];
            uint16_t cd[NR];
            uint8_t sl[NR];
            uint32_t tid1 = threadIdx.x;
            asm volatile("" : "+v"(tid1));  // (positions computed again, not carried from loop to loop)
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const uint32_t idx = tid1 + 256u * r, ty = idx / OW, tx = idx % OW;
                const bool in = ty < hy && tx < hx;
                // (in-tile block coordinates by the reciprocal: the group starts on a block boundary)
                const uint32_t ly = (ty * inv) >> 16, lx = (tx * inv) >> 16;
                const uint32_t oy = y0 + ly * B, ey = min(B, (uint32_t)d1 - oy), ox = x0 + lx * B, ex = min(B, (uint32_t)d2 - ox);
                const uint64_t at = (uint64_t)oy * d2 + (uint64_t)ey * ox + (uint64_t)(ty - ly * B) * ex + (tx - lx * B);
                v[r] = work[in ? at : 0];
                cd[r] = codes[in ? at : 0];
                sl[r] = p.sel[in ? (gy * BLKN_G + ly) * p.nb[2] + (gx * BLKN_G + lx) : 0];
            }
#pragma unroll
            for (int r = 0; r < NR; r++)
                if (sl[r] != 2 && cd[r]) v[r] = (Q)((int)cd[r] - (int)p.radius);
            uint32_t tid2 = threadIdx.x;
            asm volatile("" : "+v"(tid2));  // (positions computed again, not carried from loop to loop)
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const uint32_t idx = tid2 + 256u * r, ty = idx / OW, tx = idx % OW;
                sq[(ty + 1) * PITCH + tx + 1] = (ty < hy && tx < hx) ? v[r] : (Q)0;
            }
        }
        __syncthreads();
        // ---- (1b) P of the Lorenzo blocks: wave w takes the blocks of its row of blocks; sums along x, then along y, zero inflow ----
        for (uint32_t lx = 0; lx < nlx; lx++) {
            if (w >= nly || s_sel[w * BLKN_G + lx] > 1) continue;
            const uint32_t ty0 = 1 + w * B, tx0 = 1 + lx * B;
            const uint32_t ey = min(B, hy - w * B), ex = min(B, hx - lx * B);
            UQ v[4];
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) {
                const uint32_t j = rr + 4 * m, at = (ty0 + min(j, ey - 1)) * PITCH + tx0;
                v[m] = cc < ex ? (UQ)sq[at + cc] : (UQ)0;
            }
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) v[m] = row16_incl_scan(v[m]);
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) {
                const uint32_t j = rr + 4 * m;
                if (j < ey && cc < ex) sq[(ty0 + j) * PITCH + tx0 + cc] = (Q)v[m];
            }
            wave_lds_fence();
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) {
                const uint32_t i = min(rr + 4 * m, ex - 1), at = ty0 * PITCH + tx0 + i;
                v[m] = cc < ey ? (UQ)sq[at + cc * PITCH] : (UQ)0;
            }
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) v[m] = row16_incl_scan(v[m]);
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) {
                const uint32_t i = rr + 4 * m;
                if (i < ex && cc < ey) sq[ty0 * PITCH + tx0 + i + cc * PITCH] = (Q)v[m];
            }
        }
        // ---- (2) the three lower neighbours' flags ----
        if (threadIdx.x < 3) {
            const uint32_t k = threadIdx.x + 1, dy = k >> 1, dx = k & 1u;
            if (gy >= dy && gx >= dx) {
                uint32_t *f = flags + (uint64_t)(gy - dy) * ng2 + (gx - dx);
                uint32_t spins = 0;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1u << 22)) {
                        atomicExch(&ctl[1], 1u);
                        break;
                    }
                }
            }
        }
        __syncthreads();
        // ---- (3) the halo: the row above (threads 0 .. 64) and the column to the left (65 .. 128), from the neighbours' shells ----
        if (threadIdx.x <= 2 * OW) {
            const bool top = threadIdx.x <= OW;
            const uint32_t ty = top ? 0u : threadIdx.x - OW, tx = top ? threadIdx.x : 0u;
            const int64_t y = (int64_t)y0 + ty - 1, x = (int64_t)x0 + tx - 1;
            const bool in = y >= 0 && x >= 0 && ty <= hy && tx <= hx;
            const Q hv = __hip_atomic_load(work + (in ? work_at((uint32_t)y, (uint32_t)x) : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sq[ty * PITCH + tx] = in ? hv : (Q)0;
        }
        __syncthreads();
        // ---- (4) the blocks' last rows and columns, inner front by inner front (a wave per row of blocks) ----
        for (uint32_t step = 0; step + 1 < nly + nlx; step++) {
            const uint32_t ly = w, lx = step - ly;
            if (ly < nly && ly <= step && lx < nlx && s_sel[ly * BLKN_G + lx] <= 1) {
                const uint32_t ty0 = 1 + ly * B, tx0 = 1 + lx * B;
                const uint32_t ey = min(B, hy - ly * B), ex = min(B, hx - lx * B);
                const uint32_t l = (uint32_t)lane;
                if (l < ex + ey - 1) {
                    const uint32_t j = l < ex ? ey - 1 : l - ex, i = l < ex ? l : ex - 1;
                    const uint32_t at = (ty0 + j) * PITCH + tx0 + i;
                    const UQ v = (UQ)sq[at] + (UQ)sq[(ty0 - 1) * PITCH + tx0 + i] + (UQ)sq[(ty0 + j) * PITCH + tx0 - 1] - (UQ)sq[(ty0 - 1) * PITCH + tx0 - 1];
                    sq[at] = (Q)v;
                }
            }
            __syncthreads();
        }
        // ---- (5) the shell out (the group's last row: threads 0 .. 63, its last column: 64 .. 127), then the flag ----
        if (threadIdx.x < 2 * OW) {
            const bool row = threadIdx.x < OW;
            const uint32_t ty = row ? hy - 1 : threadIdx.x - OW, tx = row ? threadIdx.x : hx - 1;
            if (ty < hy && tx < hx && (row || ty != hy - 1)) {
                const uint32_t ly = (ty * inv) >> 16, lx = (tx * inv) >> 16;
                if (s_sel[ly * BLKN_G + lx] <= 1)  // (a regression block's lattice values are in the work array already)
                    __hip_atomic_store(work + work_at(y0 + ty, x0 + tx), sq[(ty + 1) * PITCH + tx + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(flags + (uint64_t)gy * ng2 + gx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ticket = atomicAdd(&ctl[0], 1u);
        }
        // ---- (6) the interiors, (7) the final values out: rows of the tile ----
        {
            Q q[NR];
            uint32_t tid3 = threadIdx.x;
            asm volatile("" : "+v"(tid3));  // (positions computed again, not carried from loop to loop)
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const uint32_t idx = tid3 + 256u * r, ty = idx / OW, tx = idx % OW;
                const uint32_t ly = (ty * inv) >> 16, lx = (tx * inv) >> 16;
                const uint32_t ty0 = 1 + ly * B, tx0 = 1 + lx * B;
                const uint32_t ey = min(B, hy - min(hy, ly * B)), ex = min(B, hx - min(hx, lx * B));
                const uint32_t j = ty + 1 - ty0, i = tx + 1 - tx0;
                const bool in = ty < hy && tx < hx;
                UQ v = (UQ)sq[(ty + 1) * PITCH + tx + 1];
                if (in && j + 1 < ey && i + 1 < ex)
                    v += (UQ)sq[(ty0 - 1) * PITCH + tx + 1] + (UQ)sq[(ty + 1) * PITCH + tx0 - 1] - (UQ)sq[(ty0 - 1) * PITCH + tx0 - 1];
                q[r] = (Q)v;
            }
            uint32_t tid4 = threadIdx.x;
            asm volatile("" : "+v"(tid4));  // (positions computed again, not carried from loop to loop)
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const uint32_t idx = tid4 + 256u * r, ty = idx / OW, tx = idx % OW;
                const uint32_t ly = (ty * inv) >> 16, lx = (tx * inv) >> 16;
                if (ty < hy && tx < hx && s_sel[ly * BLKN_G + lx] <= 1) tout[(uint64_t)(y0 + ty) * d2 + (x0 + tx)] = lat.dequant(q[r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// 4-D arrays (round 4): Lorenzo-1 / linear regression with FIVE coefficients per block of B^4 values (B = 4..6, default 6:
// Config.hpp:175) — RegressionPredictor.hpp:28-55, 77-92 for N = 4, LorenzoPredictor.hpp:69-74 (fifteen neighbours, noise 1.79 eb),
// the sample points of BlockwiseIterator.hpp:151-184 (eight per step of the diagonal). The same design as in three dimensions, in its
// plain form: k_blk4_fit (a wave per block: fit, estimates, choice, regression blocks coded, q~ of every element to the work array),
// k_blk4_lorenzo (a thread per code position: the sixteen-term stencil over q~), the side section with five coefficients per
// regression block, and a decoder of anti-diagonal fronts (bw + bz + by + bx = const, a wave per block, the block and its low halo
// in LDS, four passes of line sums). p.dw / p.nbw: the slowest dimension; p.d / p.nb: the other three.
// ------------------------------------------------------------------------------------------------------------
struct Blk4 {
    uint32_t ow, oz, oy, ox;  // origin
    uint32_t ew, ez, ey, ex;  // extents (ragged at the high end)
    uint64_t coff;            // position of the block's first code (block raster order, raster order inside a block)
};
__device__ __forceinline__ Blk4 blk4_geom(const szk_blk_params &p, uint32_t bw, uint32_t bz, uint32_t by, uint32_t bx) {
    Blk4 g;
    g.ow = bw * p.B;
    g.oz = bz * p.B;
    g.oy = by * p.B;
    g.ox = bx * p.B;
    g.ew = min(p.B, (uint32_t)p.dw - g.ow);
    g.ez = min(p.B, (uint32_t)p.d[0] - g.oz);
    g.ey = min(p.B, (uint32_t)p.d[1] - g.oy);
    g.ex = min(p.B, (uint32_t)p.d[2] - g.ox);
    // whole slabs of blocks along w below, whole slabs along z in this one, whole rows of blocks, the blocks left of this one
    g.coff = (uint64_t)g.ow * p.d[0] * p.d[1] * p.d[2] +
             (uint64_t)g.ew * ((uint64_t)g.oz * p.d[1] * p.d[2] + (uint64_t)g.ez * ((uint64_t)g.oy * p.d[2] + (uint64_t)g.ey * g.ox));
    return g;
}
__device__ __forceinline__ Blk4 blk4_of_task(const szk_blk_params &p, uint32_t task) {
    const uint32_t bx = task % p.nb[2];
    uint32_t r = task / p.nb[2];
    const uint32_t by = r % p.nb[1];
    r /= p.nb[1];
    const uint32_t bz = r % p.nb[0], bw = r / p.nb[0];
    return blk4_geom(p, bw, bz, by, bx);
}
__device__ __forceinline__ void blk4_own(const Blk4 &g, uint32_t t, uint32_t (&i)[4]) {
    i[3] = t % g.ex;
    t /= g.ex;
    i[2] = t % g.ey;
    t /= g.ey;
    i[1] = t % g.ez;
    i[0] = t / g.ez;
}
__device__ __forceinline__ uint64_t blk4_at(const szk_blk_params &p, uint64_t w, uint64_t z, uint64_t y, uint64_t x) {
    return ((w * p.d[0] + z) * p.d[1] + y) * p.d[2] + x;
}
template <typename T>
__device__ __forceinline__ T reg_predict4(const T (&c)[5], const uint32_t (&i)[4]) {  // RegressionPredictor.hpp:77-92, left to right in T
    T s = c[0] * (T)i[0];
    s = s + c[1] * (T)i[1];
    s = s + c[2] * (T)i[2];
    s = s + c[3] * (T)i[3];
    return s + c[4];
}
template <typename T>
__device__ __forceinline__ void coef_recover4(const int64_t *lc, const CoefLat &cl, T (&rc)[5]) {
    for (int i = 0; i < 4; i++) rc[i] = (T)((double)lc[i] * cl.step_lin);
    rc[4] = (T)((double)lc[4] * cl.step_ind);
}
template <typename T, uint32_t HW, int NW>
__global__ __launch_bounds__(NW * 64) void k_blk4_fit(const T *__restrict__ in, uint16_t *__restrict__ codes, szk_blk_params p, uint32_t nblocks) {
    using Q = typename QTraits<T>::Q;
    __shared__ uint32_t lh[HW + 1];  // (+ the count of code 0: blk_count)
    for (uint32_t b = threadIdx.x; b <= HW; b += NW * 64) lh[b] = 0;
    __syncthreads();
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    const uint32_t wv = threadIdx.x / WAVE;
    Q *qwork = reinterpret_cast<Q *>(p.qwork);
    const CoefLat cl = coef_lat(p.eb, p.B, 4u);
    const double eb_recip = 1.0 / p.eb;
    const bool has_l1 = p.mask & 1u, has_r = p.mask & 4u;
    const T noise = (T)(1.79 * p.eb);
    for (uint32_t task = blockIdx.x * NW + wv; task < nblocks; task += gridDim.x * NW) {
        const Blk4 g = blk4_of_task(p, task);
        const uint32_t nown = g.ew * g.ez * g.ey * g.ex;
        // ---- fit (RegressionPredictor.hpp:28-55, N = 4) ----
        const bool r_valid = has_r && g.ew > 1 && g.ez > 1 && g.ey > 1 && g.ex > 1;
        T cf[5] = {0, 0, 0, 0, 0};
        if (r_valid) {
            double s[5] = {0, 0, 0, 0, 0};
            for (uint32_t t = lane; t < nown; t += WAVE) {
                uint32_t i[4];
                blk4_own(g, t, i);
                const T v = in[blk4_at(p, g.ow + i[0], g.oz + i[1], g.oy + i[2], g.ox + i[3])];
                for (int k = 0; k < 4; k++) s[k] += (double)((T)i[k] * v);
                s[4] += (double)v;
            }
            for (int k = 0; k < 5; k++) s[k] = wave_sum_f64(s[k]);
            const double dims[4] = {(double)g.ew, (double)g.ez, (double)g.ey, (double)g.ex};
            const double num = dims[0] * dims[1] * dims[2] * dims[3];
            cf[4] = (T)(s[4] / num);
            for (int k = 0; k < 4; k++) {
                cf[k] = (T)((2 * s[k] / (dims[k] - 1) - s[4]) * 6 / num / (dims[k] + 1));
                cf[4] = (T)((double)cf[4] - (dims[k] - 1) * (double)cf[k] / 2);
            }
        }
        // ---- selection (ComposedPredictor.hpp:25-40 over foreach_sampling) ----
        int sid = has_l1 ? 0 : 2;
        if (has_l1 && has_r) {
            const uint32_t m = min(min(g.ew, g.ez), min(g.ey, g.ex));
            double e1 = 0, er = 0;
            // what the Lorenzo estimate sees: the original inside the block, the lattice reconstruction outside it, zero outside the array
            auto seen = [&](int64_t w, int64_t z, int64_t y, int64_t x) -> T {
                if (w < 0 || z < 0 || y < 0 || x < 0) return (T)0;
                T v = in[blk4_at(p, (uint64_t)w, (uint64_t)z, (uint64_t)y, (uint64_t)x)];
                if (w < (int64_t)g.ow || z < (int64_t)g.oz || y < (int64_t)g.oy || x < (int64_t)g.ox) {
                    bool bad;
                    const Q qh = lat.quant(v, bad);
                    if (!bad) v = lat.dequant(qh);
                }
                return v;
            };
            for (uint32_t k = lane; k < 8 * m; k += WAVE) {
                const uint32_t i = k >> 3, cmb = k & 7u, j = m - 1 - i;
                const uint32_t idx[4] = {i, (cmb & 4u) ? j : i, (cmb & 2u) ? j : i, (cmb & 1u) ? j : i};
                const int64_t w = (int64_t)g.ow + idx[0], z = (int64_t)g.oz + idx[1], y = (int64_t)g.oy + idx[2], x = (int64_t)g.ox + idx[3];
                const T v = in[blk4_at(p, (uint64_t)w, (uint64_t)z, (uint64_t)y, (uint64_t)x)];
                // LorenzoPredictor.hpp:69-74: prev4(d, ds, t, k, j, i) steps t along y, k along z, j along w, i along x (ds: the strides, slowest first)
                auto P = [&](int t_, int k_, int j_, int i_) -> T { return seen(w - j_, z - k_, y - t_, x - i_); };
                T s = P(0, 0, 0, 1);
                s = s + P(0, 0, 1, 0);
                s = s - P(0, 0, 1, 1);
                s = s + P(0, 1, 0, 0);
                s = s - P(0, 1, 0, 1);
                s = s - P(0, 1, 1, 0);
                s = s + P(0, 1, 1, 1);
                s = s + P(1, 0, 0, 0);
                s = s - P(1, 0, 0, 1);
                s = s - P(1, 0, 1, 0);
                s = s + P(1, 0, 1, 1);
                s = s - P(1, 1, 0, 0);
                s = s + P(1, 1, 0, 1);
                s = s + P(1, 1, 1, 0);
                s = s - P(1, 1, 1, 1);
                e1 += (double)(T)((T)fabs((double)(T)(v - s)) + noise);
                if (r_valid) er += (double)(T)fabs((double)(T)(v - reg_predict4(cf, idx)));
            }
            e1 = wave_sum_f64(e1);
            er = wave_sum_f64(er);
            sid = (r_valid && er < e1) ? 2 : 0;
        } else if (sid == 2 && !r_valid) {
            sid = 0;  // BlockwiseDecomposition.hpp:35-37
        }
        int64_t lc[5] = {0, 0, 0, 0, 0};
        if (sid == 2) {  // coefficients onto their lattices; a coefficient the lattice cannot hold -> Lorenzo-1
            bool ok = true;
            for (int k = 0; k < 5; k++) {
                const double sc = (double)cf[k] / (k < 4 ? cl.step_lin : cl.step_ind);
                if (!(fabs(sc) < 4503599627370496.0)) ok = false;
                else lc[k] = (int64_t)rint(sc);
            }
            if (!ok) sid = 0;
        }
        T rc[5];
        coef_recover4(lc, cl, rc);
        for (uint32_t t0 = 0; t0 < nown; t0 += WAVE) {  // (uniform trip count: the counts and list appends are wave operations)
            const uint32_t t = t0 + lane;
            const bool act = t < nown;
            uint32_t i[4];
            blk4_own(g, act ? t : 0u, i);
            const uint64_t gi = blk4_at(p, g.ow + i[0], g.oz + i[1], g.oy + i[2], g.ox + i[3]);
            const T raw = in[gi];
            bool bad = false;
            Q qt = 0;
            int code = 1;
            if (sid == 2) {
                T v = raw;
                code = act ? ref_quantize(v, reg_predict4(rc, i), p.eb, eb_recip, (int)p.radius) : 1;
                if (code != 0) {
                    qt = lat.quant(v, bad);
                    if (bad) qt = 0;
                }
                bad = code == 0;  // unpredictable: the raw value (LinearQuantizer.hpp:66-69)
                if (act) codes[g.coff + t] = (uint16_t)code;
            } else {
                qt = lat.quant(raw, bad);
                if (bad) qt = 0;
            }
            if (act) qwork[gi] = qt;
            blk_count<HW>(lh, p, (uint32_t)code, act && sid == 2);
            blk_vout<T>(p, act && bad, gi, raw);
        }
        if (lane == 0) {
            p.sel[task] = (uint8_t)sid;
            if (sid == 2)
                for (int k = 0; k < 5; k++) p.coef[(uint64_t)task * 5 + k] = lc[k];
        }
    }
    __syncthreads();
    blk_flush<HW>(lh, p);
}
// code position -> block and element (the codes lie block by block)
struct Blk4Pos {
    uint32_t task;
    uint64_t w, z, y, x;
};
__device__ __forceinline__ Blk4Pos blk4_pos(const szk_blk_params &p, uint64_t c) {
    Blk4Pos r;
    const uint64_t D3 = p.d[0] * p.d[1] * p.d[2];
    const uint32_t bw = (uint32_t)(c / ((uint64_t)p.B * D3));
    uint64_t rem = c - (uint64_t)bw * p.B * D3;
    const uint32_t ow = bw * p.B, ew = min(p.B, (uint32_t)p.dw - ow);
    const uint64_t zs = (uint64_t)ew * p.B * p.d[1] * p.d[2];
    const uint32_t bz = (uint32_t)(rem / zs);
    rem -= (uint64_t)bz * zs;
    const uint32_t oz = bz * p.B, ez = min(p.B, (uint32_t)p.d[0] - oz);
    const uint64_t ys = (uint64_t)ew * ez * p.B * p.d[2];
    const uint32_t by = (uint32_t)(rem / ys);
    rem -= (uint64_t)by * ys;
    const uint32_t oy = by * p.B, ey = min(p.B, (uint32_t)p.d[1] - oy);
    const uint32_t xs = ew * ez * ey * p.B;
    const uint32_t bx = (uint32_t)(rem / xs);
    uint32_t t = (uint32_t)(rem - (uint64_t)bx * xs);
    const uint32_t ox = bx * p.B, ex = min(p.B, (uint32_t)p.d[2] - ox);
    const uint32_t i3 = t % ex;
    t /= ex;
    const uint32_t i2 = t % ey;
    t /= ey;
    const uint32_t i1 = t % ez, i0 = t / ez;
    r.task = ((bw * p.nb[0] + bz) * p.nb[1] + by) * p.nb[2] + bx;
    r.w = ow + i0;
    r.z = oz + i1;
    r.y = oy + i2;
    r.x = ox + i3;
    return r;
}
template <typename T, uint32_t HW, int TB>
__global__ __launch_bounds__(TB) void k_blk4_lorenzo(uint16_t *__restrict__ codes, szk_blk_params p, uint64_t n) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    __shared__ uint32_t lh[HW + 1];  // (+ the count of code 0: blk_count)
    for (uint32_t b = threadIdx.x; b <= HW; b += TB) lh[b] = 0;
    __syncthreads();
    const Q *__restrict__ qw = reinterpret_cast<const Q *>(p.qwork);
    const uint64_t stride = (uint64_t)gridDim.x * TB;
    for (uint64_t c0 = (uint64_t)blockIdx.x * TB; c0 < n; c0 += stride) {  // (workgroup-uniform: the counts and appends are wave operations)
        const uint64_t c = c0 + threadIdx.x;
        bool act = c < n;
        const Blk4Pos e = blk4_pos(p, act ? c : 0);
        act = act && p.sel[e.task] != 2;
        UQ delta = 0;
        if (act) {
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const uint32_t a = (m >> 3) & 1, b = (m >> 2) & 1, cc = (m >> 1) & 1, d = m & 1;
                if (e.w < a || e.z < b || e.y < cc || e.x < d) continue;  // (zeros outside the array)
                const UQ q = (UQ)qw[blk4_at(p, e.w - a, e.z - b, e.y - cc, e.x - d)];
                delta = ((a + b + cc + d) & 1) ? delta - q : delta + q;
            }
        }
        const bool inr = (UQ)(delta + (UQ)(p.radius - 1)) <= (UQ)(2 * p.radius - 2);
        const uint32_t code = inr ? (uint32_t)(delta + (UQ)p.radius) : 0u;
        if (act) codes[c] = (uint16_t)code;
        blk_count<HW>(lh, p, code, act);
        const unsigned long long pd = wave_append_slot(act && !inr, p.n_dout);
        if (act && !inr && pd < p.out_cap) {
            p.dout_idx[pd] = c;
            reinterpret_cast<Q *>(p.dout_val)[pd] = (Q)delta;
        }
    }
    __syncthreads();
    blk_flush<HW>(lh, p);
}
// ---- decoder ----
// regression blocks: q~ of their elements into the output (as lattice words: what their neighbours' halos read)
template <typename T>
__global__ __launch_bounds__(256) void k_blk4_pre(const uint16_t *__restrict__ codes, void *d_out, szk_blk_params p, uint32_t nblocks,
                                                  const uint32_t *__restrict__ rank, const int64_t *__restrict__ coef_by_rank) {
    using Q = typename QTraits<T>::Q;
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    Q *qout = reinterpret_cast<Q *>(d_out);
    const CoefLat cl = coef_lat(p.eb, p.B, 4u);
    for (uint32_t task = blockIdx.x * 4 + threadIdx.x / WAVE; task < nblocks; task += gridDim.x * 4) {
        if (p.sel[task] != 2) continue;
        const Blk4 g = blk4_of_task(p, task);
        const uint32_t nown = g.ew * g.ez * g.ey * g.ex;
        T rc[5];
        coef_recover4(coef_by_rank + (uint64_t)rank[task] * 5, cl, rc);
        for (uint32_t t = lane; t < nown; t += WAVE) {
            uint32_t i[4];
            blk4_own(g, t, i);
            const uint32_t code = codes[g.coff + t];
            Q qt = 0;
            if (code) {
                bool bad;
                const T val = ref_recover(reg_predict4(rc, i), (int)code, p.eb, (int)p.radius);
                qt = lat.quant(val, bad);
                if (bad) qt = 0;
            }
            qout[blk4_at(p, g.ow + i[0], g.oz + i[1], g.oy + i[2], g.ox + i[3])] = qt;
        }
    }
}
// one front (bw + bz + by + bx = diag), a wave per Lorenzo block: the block's deltas and its low halo (one layer of finished q~, zeros
// outside the array) in LDS; four passes of line sums undo the four differences — along x with the inflow Dw Dz Dy q~ of the halo
// column, along y with Dw Dz q~ of the halo row, along z with Dw q~, along w with q~ itself
#define BLK4_MAXE 7u  // block edge (<= 6) + the halo layer
#define BLK4_DT 256u
template <typename T>
__global__ __launch_bounds__(BLK4_DT) void k_blk4_decode(const void *deltas_, void *d_out, szk_blk_params p, uint32_t diag) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    __shared__ Q sq[BLK4_MAXE * BLK4_MAXE * BLK4_MAXE * BLK4_MAXE], sa[BLK4_MAXE * BLK4_MAXE * BLK4_MAXE * BLK4_MAXE];
    const uint32_t by = blockIdx.x % p.nb[1];
    uint32_t r = blockIdx.x / p.nb[1];
    const uint32_t bz = r % p.nb[0], bw = r / p.nb[0];
    if (bw + bz + by > diag) return;
    const uint32_t bx = diag - bw - bz - by;
    if (bx >= p.nb[2]) return;
    const uint32_t task = ((bw * p.nb[0] + bz) * p.nb[1] + by) * p.nb[2] + bx;
    if (p.sel[task] == 2) return;
    const Blk4 g = blk4_geom(p, bw, bz, by, bx);
    const uint32_t lane = threadIdx.x;  // (a workgroup per block: the tile's 2401 words and the 216 lines of a pass by 256 threads)
    Q *qout = reinterpret_cast<Q *>(d_out);
    const Q *deltas = reinterpret_cast<const Q *>(deltas_);
    const uint32_t tw = g.ew + 1, tz = g.ez + 1, ty = g.ey + 1, tx = g.ex + 1;  // tile extents; coordinate 0 = the halo
    auto at = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) -> uint32_t { return ((a * tz + b) * ty + c) * tx + d; };
    for (uint32_t l = lane; l < tw * tz * ty * tx; l += BLK4_DT) {
        uint32_t q = l;
        const uint32_t d = q % tx;
        q /= tx;
        const uint32_t c = q % ty;
        q /= ty;
        const uint32_t b = q % tz, a = q / tz;
        Q v = 0;
        if (a && b && c && d) {
            v = deltas[g.coff + (((a - 1) * g.ez + (b - 1)) * g.ey + (c - 1)) * g.ex + (d - 1)];
        } else {
            const int64_t w = (int64_t)g.ow + a - 1, z = (int64_t)g.oz + b - 1, y = (int64_t)g.oy + c - 1, x = (int64_t)g.ox + d - 1;
            if (w >= 0 && z >= 0 && y >= 0 && x >= 0) v = qout[blk4_at(p, (uint64_t)w, (uint64_t)z, (uint64_t)y, (uint64_t)x)];
        }
        sq[l] = v;
    }
    __syncthreads();
    // along x: lines (a, b, c) of the block
    for (uint32_t l = lane; l < g.ew * g.ez * g.ey; l += BLK4_DT) {
        uint32_t q = l;
        const uint32_t c = q % g.ey + 1;
        q /= g.ey;
        const uint32_t b = q % g.ez + 1, a = q / g.ez + 1;
        UQ run = 0;
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const uint32_t da = (m >> 2) & 1, db = (m >> 1) & 1, dc = m & 1;
            const UQ v = (UQ)sq[at(a - da, b - db, c - dc, 0)];
            run = ((da + db + dc) & 1) ? run - v : run + v;
        }
        for (uint32_t d = 1; d <= g.ex; d++) {
            run += (UQ)sq[at(a, b, c, d)];
            sa[at(a, b, c, d)] = (Q)run;
        }
    }
    __syncthreads();
    // along y: lines (a, b, d)
    for (uint32_t l = lane; l < g.ew * g.ez * g.ex; l += BLK4_DT) {
        uint32_t q = l;
        const uint32_t d = q % g.ex + 1;
        q /= g.ex;
        const uint32_t b = q % g.ez + 1, a = q / g.ez + 1;
        UQ run = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const uint32_t da = (m >> 1) & 1, db = m & 1;
            const UQ v = (UQ)sq[at(a - da, b - db, 0, d)];
            run = ((da + db) & 1) ? run - v : run + v;
        }
        for (uint32_t c = 1; c <= g.ey; c++) {
            run += (UQ)sa[at(a, b, c, d)];
            sa[at(a, b, c, d)] = (Q)run;
        }
    }
    __syncthreads();
    // along z: lines (a, c, d)
    for (uint32_t l = lane; l < g.ew * g.ey * g.ex; l += BLK4_DT) {
        uint32_t q = l;
        const uint32_t d = q % g.ex + 1;
        q /= g.ex;
        const uint32_t c = q % g.ey + 1, a = q / g.ey + 1;
        UQ run = (UQ)sq[at(a, 0, c, d)] - (UQ)sq[at(a - 1, 0, c, d)];
        for (uint32_t b = 1; b <= g.ez; b++) {
            run += (UQ)sa[at(a, b, c, d)];
            sa[at(a, b, c, d)] = (Q)run;
        }
    }
    __syncthreads();
    // along w: lines (b, c, d); the result is q~
    for (uint32_t l = lane; l < g.ez * g.ey * g.ex; l += BLK4_DT) {
        uint32_t q = l;
        const uint32_t d = q % g.ex + 1;
        q /= g.ex;
        const uint32_t c = q % g.ey + 1, b = q / g.ey + 1;
        UQ run = (UQ)sq[at(0, b, c, d)];
        for (uint32_t a = 1; a <= g.ew; a++) {
            run += (UQ)sa[at(a, b, c, d)];
            qout[blk4_at(p, g.ow + a - 1, g.oz + b - 1, g.oy + c - 1, g.ox + d - 1)] = (Q)run;
        }
    }
}
// lattice values -> values; a regression block's values from its codes
template <typename T>
__global__ __launch_bounds__(256) void k_blk4_final(const uint16_t *__restrict__ codes, void *d_out, szk_blk_params p, uint32_t nblocks,
                                                    const uint32_t *__restrict__ rank, const int64_t *__restrict__ coef_by_rank) {
    using Q = typename QTraits<T>::Q;
    const Lattice<T> lat(p.lat);
    const int lane = lane_id();
    Q *qv = reinterpret_cast<Q *>(d_out);
    T *ov = reinterpret_cast<T *>(d_out);
    const CoefLat cl = coef_lat(p.eb, p.B, 4u);
    for (uint32_t task = blockIdx.x * 4 + threadIdx.x / WAVE; task < nblocks; task += gridDim.x * 4) {
        const Blk4 g = blk4_of_task(p, task);
        const uint32_t nown = g.ew * g.ez * g.ey * g.ex;
        const bool reg = p.sel[task] == 2;
        T rc[5] = {0, 0, 0, 0, 0};
        if (reg) coef_recover4(coef_by_rank + (uint64_t)rank[task] * 5, cl, rc);
        for (uint32_t t = lane; t < nown; t += WAVE) {
            uint32_t i[4];
            blk4_own(g, t, i);
            const uint64_t gi = blk4_at(p, g.ow + i[0], g.oz + i[1], g.oy + i[2], g.ox + i[3]);
            if (reg) {
                const uint32_t code = codes[g.coff + t];
                ov[gi] = code ? ref_recover(reg_predict4(rc, i), (int)code, p.eb, (int)p.radius) : (T)0;  // (code 0: patched from the list)
            } else {
                ov[gi] = lat.dequant(qv[gi]);
            }
        }
    }
}

// The tuner's Lorenzo trial (SZAlgoInterp.hpp:232-247, lorenzo_compress_test: a 1-D array's sample blocks coded in blocks of FIVE values
// by the set [Lorenzo-1, Lorenzo-2]): the choice per block by the reference's estimate at the block's two ends, the codes on the
// lattice, their histogram for the pricing kernel. A thread per block of five (seven loads), every sample block an array of its own
// (zeros left of its first value). extra[0] += blocks that chose second order, extra[1] += blocks.
template <typename T>
__global__ __launch_bounds__(256) void k_trial_lorenzo12(const T *__restrict__ samples, uint64_t per, uint64_t nsb, szk_lattice latp, double eb, uint32_t radius,
                                                         unsigned long long *__restrict__ hist, unsigned long long *__restrict__ counters,
                                                         unsigned long long *__restrict__ extra) {
    using Q = typename QTraits<T>::Q;
    using UQ = typename QTraits<T>::UQ;
    constexpr uint32_t HW = 4096;
    __shared__ uint32_t lh[HW];
    for (uint32_t b = threadIdx.x; b < HW; b += 256) lh[b] = 0;
    __syncthreads();
    const Lattice<T> lat(latp);
    const uint64_t nb5 = (per + 4) / 5, total = nsb * nb5;
    const T n1 = (T)(0.5 * eb), n2 = (T)(1.08 * eb);
    const uint32_t lo = radius > HW / 2 ? radius - HW / 2 : 0u;
    uint32_t n_bad = 0, n_out = 0, n_l2 = 0, n_blk = 0;
    for (uint64_t task = (uint64_t)blockIdx.x * 256 + threadIdx.x; task < total; task += (uint64_t)gridDim.x * 256) {
        const uint64_t k = task / nb5, x0 = (task - k * nb5) * 5;
        const uint32_t m = (uint32_t)min((uint64_t)5, per - x0);
        const T *sb = samples + k * per;
        T v[7], seen[7];
        Q q[7];
        bool bad[7];
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const int64_t x = (int64_t)x0 - 2 + i;
            const bool have = x >= 0 && x < (int64_t)(x0 + m);
            v[i] = have ? sb[x] : (T)0;
            q[i] = lat.quant(v[i], bad[i]);
            if (bad[i]) q[i] = 0;
            if (!have) bad[i] = false;
            // what the estimate sees: the original inside the block, the lattice reconstruction left of it (ComposedPredictor.hpp:25-40
            // runs before the block is coded), zero left of the array
            seen[i] = i >= 2 ? v[i] : (have && !bad[i] ? lat.dequant(q[i]) : v[i]);
        }
        double e1 = 0, e2 = 0;
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
            const int pi = 2 + (pt ? (int)m - 1 : 0);
            T s1 = 0, s2 = 0, pv = 0;
#pragma unroll
            for (int i = 0; i < 7; i++) {  // (constant indices: the arrays stay in registers)
                if (i == pi) pv = v[i];
                if (i == pi - 1) s1 = seen[i];
                if (i == pi - 2) s2 = seen[i];
            }
            e1 += (double)(T)((T)fabs((double)(T)(pv - s1)) + n1);
            e2 += (double)(T)((T)fabs((double)(T)(pv - (T)((T)(2 * s1) - s2))) + n2);
        }
        const bool second = e2 < e1;  // (std::min_element: the first minimum — Lorenzo-1 on a tie)
        n_l2 += second ? 1u : 0u;
        n_blk++;
#pragma unroll
        for (int i = 2; i < 7; i++) {
            if ((uint32_t)(i - 2) >= m) break;
            UQ delta = (UQ)q[i] - (UQ)q[i - 1];
            if (second) delta = delta - (UQ)q[i - 1] + (UQ)q[i - 2];
            const bool inr = (UQ)(delta + (UQ)(radius - 1)) <= (UQ)(2 * radius - 2);
            const uint32_t code = inr ? (uint32_t)(delta + (UQ)radius) : 0u;
            n_bad += bad[i] ? 1u : 0u;
            n_out += inr ? 0u : 1u;
            if (code - lo < HW) atomicAdd(&lh[code - lo], 1u);
            else atomicAdd(&hist[code], 1ull);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < HW; b += 256)
        if (lh[b]) atomicAdd(&hist[lo + b], (unsigned long long)lh[b]);
    n_bad = wave_sum(n_bad);
    n_out = wave_sum(n_out);
    n_l2 = wave_sum(n_l2);
    n_blk = wave_sum(n_blk);
    if (lane_id() == 0) {
        if (n_bad) atomicAdd(&counters[0], (unsigned long long)n_bad);
        if (n_out) atomicAdd(&counters[1], (unsigned long long)n_out);
        if (n_l2) atomicAdd(&extra[0], (unsigned long long)n_l2);
        if (n_blk) atomicAdd(&extra[1], (unsigned long long)n_blk);
    }
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------
static uint32_t blk_count_blocks(const szk_blk_params *p) { return p->nb[0] * p->nb[1] * p->nb[2] * (p->nbw ? p->nbw : 1u); }

static int launch_blk_side_build(const szk_blk_params *p, const szk_blk_scratch *sc, uint32_t nblocks, hipStream_t s, bool rank_done = false);
// arrays of one and two dimensions: fit / selection / regression blocks, then the Lorenzo codes over q~
static int launch_blkn_compress(int dtype, const void *d_in, uint16_t *codes, const szk_blk_params *p, const szk_blk_scratch *sc, hipStream_t s) {
    const uint32_t nblocks = blk_count_blocks(p);
    const uint64_t n = p->d[1] * p->d[2];
    if (p->ndim == 1 && (p->mask & 2u) && !(p->mask & 4u) && !p->sel_given && !(szk_dbg_flags & 134217728)) {
        // Lorenzo members only (debug flag 134217728: the general fit pass): choices by k_blkn_sel12, codes straight from the array
        szk_blk_params q = *p;
        q.sel_given = 1;
        const uint32_t gsel = (uint32_t)std::min<uint64_t>(BLK_GRID_ENC, ((uint64_t)nblocks + 255) / 256);
        if (dtype == 0) hipLaunchKernelGGL(k_blkn_sel12<float>, dim3(gsel), dim3(256), 0, s, (const float *)d_in, q, nblocks);
        else hipLaunchKernelGGL(k_blkn_sel12<double>, dim3(gsel), dim3(256), 0, s, (const double *)d_in, q, nblocks);
#define BLKN_L12(T, HW, NW)                                                                                                     \
    do {                                                                                                                        \
        const uint32_t g4 = (uint32_t)std::min<uint64_t>(BLK_GRID_ENC * 8 / NW, (n + NW * 256 - 1) / (NW * 256));                    \
        hipLaunchKernelGGL((k_blkn_lorenzo12v<T, HW, NW * 64>), dim3(g4), dim3(NW * 64), 0, s, (const T *)d_in, codes, q, n);      \
    } while (0)
        // (workgroups of 16 waves whatever the window: a quarter of the workgroups queue at the global histogram's bins — see below)
        if (dtype == 0) {
            if (sc->wide_hist) BLKN_L12(float, BLK_HWIN_WIDE, 16);
            else BLKN_L12(float, BLK_HWIN, 16);
        } else {
            if (sc->wide_hist) BLKN_L12(double, BLK_HWIN_WIDE, 16);
            else BLKN_L12(double, BLK_HWIN, 16);
        }
#undef BLKN_L12
        return launch_blk_side_build(p, sc, nblocks, s);
    }
#define BLKN_ENC1(T, HW, NW, TWO)                                                                                               \
    do {                                                                                                                        \
        const uint32_t gfit = (uint32_t)std::min<uint64_t>(BLK_GRID_ENC * 4 / NW, ((uint64_t)nblocks + NW - 1) / NW);                 \
        const uint32_t glor = (uint32_t)std::min<uint64_t>(BLK_GRID_ENC * 8 / NW, (n + NW * 64 - 1) / (NW * 64));                     \
        if (!TWO && !p->sel_given && !(p->mask & 2u) && !(szk_dbg_flags & 134217728)) { /* 1-D: four blocks per wave (debug flag 134217728: a wave per block) */ \
            const uint32_t grow = (uint32_t)std::min<uint64_t>(BLK_GRID_ENC * 4 / NW, ((uint64_t)nblocks + NW * 4 - 1) / (NW * 4));    \
            hipLaunchKernelGGL((k_blkn_fit_rows<T, HW, NW>), dim3(grow), dim3(NW * 64), 0, s, (const T *)d_in, codes, *p, nblocks); \
        } else                                                                                                                  \
        hipLaunchKernelGGL((k_blkn_fit<T, HW, NW, TWO, false>), dim3(gfit), dim3(NW * 64), 0, s, (const T *)d_in, codes, *p, nblocks, \
                           (unsigned long long *)nullptr);                                                                          \
        if (!TWO && !p->sel_given && !(p->mask & 2u) && !(szk_dbg_flags & 134217728)) { /* 1-D, q~ of everything in the work array: four codes per thread */ \
            const uint32_t g4 = (uint32_t)std::min<uint64_t>(BLK_GRID_ENC * 8 / NW, (n + NW * 256 - 1) / (NW * 256));                  \
            hipLaunchKernelGGL((k_blkn_lorenzo1v<T, HW, NW * 64>), dim3(g4), dim3(NW * 64), 0, s, codes, *p, n);                   \
        } else                                                                                                                  \
        hipLaunchKernelGGL((k_blkn_lorenzo<T, HW, NW * 64, TWO>), dim3(glor), dim3(NW * 64), 0, s, (const T *)d_in, codes, *p, n); \
    } while (0)
#define BLKN_ENC(T, HW, NW)                        \
    do {                                           \
        if (p->ndim == 2) BLKN_ENC1(T, HW, NW, true); \
        else BLKN_ENC1(T, HW, NW, false);          \
    } while (0)
    // (round 6) 1-D arrays: workgroups of 16 waves whatever the window. Every workgroup ends by adding its LDS histogram to the global one — one
    // device-scope atomic per non-empty bin, ~400 addresses that EVERY workgroup hits: they are performed one after another at the memory
    // side, ~20 ns each, and C1's 1024 workgroups of four waves spent 21 of the stencil pass's 27 us and 7 of the fit pass's 20 queueing
    // there (measured with the flush switched off). A quarter of the workgroups, a quarter of the queue.
    const bool wg16 = p->ndim == 1 && !(szk_dbg_flags & 134217728);
    if (dtype == 0) {
        if (sc->wide_hist) BLKN_ENC(float, BLK_HWIN_WIDE, 16);
        else if (wg16) BLKN_ENC1(float, BLK_HWIN, 16, false);
        else BLKN_ENC(float, BLK_HWIN, 4);
    } else {
        if (sc->wide_hist) BLKN_ENC(double, BLK_HWIN_WIDE, 16);
        else if (wg16) BLKN_ENC1(double, BLK_HWIN, 16, false);
        else BLKN_ENC(double, BLK_HWIN, 4);
    }
#undef BLKN_ENC
#undef BLKN_ENC1
    return launch_blk_side_build(p, sc, nblocks, s);
}

// 4-D arrays: the fit pass (choices, regression blocks, q~ of every element), then the stencil pass over q~
static int launch_blk4_compress(int dtype, const void *d_in, uint16_t *codes, const szk_blk_params *p, const szk_blk_scratch *sc, hipStream_t s) {
    const uint32_t nblocks = blk_count_blocks(p);
    const uint64_t n = p->dw * p->d[0] * p->d[1] * p->d[2];
#define BLK4_ENC(T, HW, NW)                                                                                                    \
    do {                                                                                                                       \
        const uint32_t gfit = (uint32_t)std::min<uint64_t>(BLK_GRID_ENC * 4 / NW, ((uint64_t)nblocks + NW - 1) / NW);                \
        const uint32_t glor = (uint32_t)std::min<uint64_t>(BLK_GRID_ENC * 8 / NW, (n + NW * 64 - 1) / (NW * 64));                    \
        hipLaunchKernelGGL((k_blk4_fit<T, HW, NW>), dim3(gfit), dim3(NW * 64), 0, s, (const T *)d_in, codes, *p, nblocks);      \
        hipLaunchKernelGGL((k_blk4_lorenzo<T, HW, NW * 64>), dim3(glor), dim3(NW * 64), 0, s, codes, *p, n);                    \
    } while (0)
    if (dtype == 0) {
        if (sc->wide_hist) BLK4_ENC(float, BLK_HWIN_WIDE, 16);
        else BLK4_ENC(float, BLK_HWIN, 4);
    } else {
        if (sc->wide_hist) BLK4_ENC(double, BLK_HWIN_WIDE, 16);
        else BLK4_ENC(double, BLK_HWIN, 4);
    }
#undef BLK4_ENC
    return launch_blk_side_build(p, sc, nblocks, s);
}

int szk_launch_blk_compress(int dtype, const void *d_in, uint16_t *codes, const szk_blk_params *p, const szk_blk_scratch *sc, hipStream_t s) {
    const uint32_t nblocks = blk_count_blocks(p);
    if (p->ndim == 4) return launch_blk4_compress(dtype, d_in, codes, p, sc, s);
    if (p->ndim < 3) return launch_blkn_compress(dtype, d_in, codes, p, sc, s);
    // With the selection pass's choices: the fit pass codes the regression blocks (and leaves their lattice values), the stencil
    // pass every other element straight from the array. Without (development switch): fit and selection by the fit pass, the
    // lattice values of everything through qwork, Lorenzo blocks from tiles.
    const bool by_element = p->sel_given && !(p->mask & 2u) && !(szk_dbg_flags & 67108864);  // (k_blk_rows: first-order Lorenzo only)
    const uint64_t nrows = p->d[0] * p->d[1];
    // the regression blocks' list before the fit pass walks it (the side section wants the same ranks afterwards: made once)
    const bool rank_first = p->sel_given && !(szk_dbg_flags & 1073741824);
    if (rank_first) launch_blk_rank(p->sel, nblocks, sc->rank, sc->comp, sc->run_scratch, sc->counters + 0, s);
    if (p->sel_given && !by_element) {
        const dim3 g((uint32_t)((p->d[2] + 255) / 256), (uint32_t)std::min<uint64_t>(nrows, 32768));
        if (dtype == 0) hipLaunchKernelGGL(k_blk_lattice<float>, g, dim3(256), 0, s, (const float *)d_in, *p, nrows);
        else hipLaunchKernelGGL(k_blk_lattice<double>, g, dim3(256), 0, s, (const double *)d_in, *p, nrows);
    }
#define BLK_ENC1(T, HW, CBV, NW)                                                                                                       \
    do {                                                                                                                               \
        const uint32_t grid = (uint32_t)std::min<uint64_t>(BLK_GRID_ENC * 4 / NW, ((uint64_t)nblocks + NW - 1) / NW);                        \
        hipLaunchKernelGGL((k_blk_fit<T, HW, CBV, NW>), dim3(grid), dim3(NW * 64), 0, s, (const T *)d_in, codes, *p, nblocks,           \
                           rank_first ? (const uint32_t *)sc->comp : (const uint32_t *)nullptr,                                          \
                           rank_first ? (const uint64_t *)(sc->counters + 0) : (const uint64_t *)nullptr);                               \
        if (by_element) {                                                                                                              \
            const uint32_t tpb = (252u / p->B) * p->B, xchunks = (uint32_t)((p->d[2] + tpb - 1) / tpb);                                  \
            const uint64_t ntasks = (uint64_t)p->nb[0] * p->nb[1] * xchunks;                                                            \
            hipLaunchKernelGGL((k_blk_rows<T, HW, CBV>), dim3((uint32_t)std::min<uint64_t>(ntasks, BLK_GRID_ENC)), dim3(256), 0, s,           \
                               (const T *)d_in, codes, *p, (uint32_t)ntasks, xchunks);                                                  \
        } else {                                                                                                                       \
            hipLaunchKernelGGL((k_blk_lorenzo<T, HW, CBV, NW>), dim3(grid), dim3(NW * 64), 0, s, codes, *p, nblocks);                   \
        }                                                                                                                              \
    } while (0)
    // (LDS: NW tiles of (B + 2)^3 values + the histogram window; the generic-edge form's tiles hold 1000 values)
#define BLK_ENC(T, HW, NW6, NW0)                  \
    do {                                          \
        if (p->B == 6) BLK_ENC1(T, HW, 6, NW6);   \
        else BLK_ENC1(T, HW, 0, NW0);             \
    } while (0)
    if (dtype == 0) {
        if (sc->wide_hist) BLK_ENC(float, BLK_HWIN_WIDE, 16, 16);
        else BLK_ENC(float, BLK_HWIN, 4, 4);
    } else {
        if (sc->wide_hist) BLK_ENC(double, BLK_HWIN_WIDE, 16, 8);
        else BLK_ENC(double, BLK_HWIN, 4, 4);
    }
#undef BLK_ENC
#undef BLK_ENC1
    return launch_blk_side_build(p, sc, nblocks, s, rank_first);
}
// the side section (selection bits + Rice-coded coefficient chain) from sel[] / coef[]
static int launch_blk_side_build(const szk_blk_params *p, const szk_blk_scratch *sc, uint32_t nblocks, hipStream_t s, bool rank_done) {
    // (counters: [0] regression blocks, [2] side bytes, [4..7] as doubles: sum of the zigzagged differences per coefficient;
    // the group sizes are staged in the rank array, which the encoder needs no more once comp is written)
    double *stats = reinterpret_cast<double *>(sc->counters + 4);
    uint32_t *group_bits = sc->rank;
    if (nblocks <= SIDE_SMALL_BLOCKS && !(szk_dbg_flags & 2048)) {  // (debug flag 2048: the eight launches whatever the block count)
        double *st = p->ndim == 4 ? sc->stats5 : stats;
#define SIDE_SMALL(NC, RK) hipLaunchKernelGGL((k_blk_side_small<NC, RK>), dim3(sc->range ? 1 + SZH_HIST_BINS / 1024 : 1), dim3(1024), 0, s, (const uint8_t *)p->sel, nblocks, sc->rank, sc->comp, sc->counters + 0, (const int64_t *)p->coef, st, group_bits, sc->side, sc->counters + 2, sc->range_hist, sc->range)
        if (p->ndim == 4) {
            if (rank_done) SIDE_SMALL(5, false);
            else SIDE_SMALL(5, true);
        } else {
            if (rank_done) SIDE_SMALL(4, false);
            else SIDE_SMALL(4, true);
        }
#undef SIDE_SMALL
        SZK_CHECK_LAUNCH();
        return 0;
    }
    if (!rank_done) launch_blk_rank(p->sel, nblocks, sc->rank, sc->comp, sc->run_scratch, sc->counters + 0, s);
    if (p->ndim == 4) {  // five coefficients: their Rice statistics have a place of their own (sc->stats5, zeroed by the caller)
        double *st5 = sc->stats5;
        hipLaunchKernelGGL(k_blk_coef_stats<5>, dim3(32), dim3(256), 0, s, p->coef, sc->comp, sc->counters + 0, st5);
        hipLaunchKernelGGL(k_blk_coef_len<5>, dim3(256), dim3(256), 0, s, p->coef, sc->comp, sc->counters + 0, st5, group_bits);
        hipLaunchKernelGGL(k_blk_sel_pack, dim3(256), dim3(256), 0, s, p->sel, nblocks, sc->side);
        hipLaunchKernelGGL(k_blk_side_layout<5>, dim3(1), dim3(1024), 0, s, nblocks, sc->counters + 0, st5, group_bits, sc->side, sc->counters + 2);
        hipLaunchKernelGGL(k_blk_coef_write<5>, dim3(256), dim3(256), 0, s, p->coef, sc->comp, sc->counters + 0, nblocks, sc->side);
        SZK_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(k_blk_coef_stats<4>, dim3(32), dim3(256), 0, s, p->coef, sc->comp, sc->counters + 0, stats);
    hipLaunchKernelGGL(k_blk_coef_len<4>, dim3(256), dim3(256), 0, s, p->coef, sc->comp, sc->counters + 0, stats, group_bits);
    hipLaunchKernelGGL(k_blk_sel_pack, dim3(256), dim3(256), 0, s, p->sel, nblocks, sc->side);
    hipLaunchKernelGGL(k_blk_side_layout<4>, dim3(1), dim3(1024), 0, s, nblocks, sc->counters + 0, stats, group_bits, sc->side, sc->counters + 2);
    hipLaunchKernelGGL(k_blk_coef_write<4>, dim3(256), dim3(256), 0, s, p->coef, sc->comp, sc->counters + 0, nblocks, sc->side);
    SZK_CHECK_LAUNCH();
    return 0;
}

int szk_blk_side_small(uint64_t nblocks) { return nblocks <= SIDE_SMALL_BLOCKS && !(szk_dbg_flags & 2048) ? 1 : 0; }
int szk_launch_blk_select(int dtype, const void *d_in, const szk_blk_params *p, uint64_t *n_other, hipStream_t s) {
    const uint32_t nblocks = blk_count_blocks(p);
    const dim3 g((nblocks + 255) / 256), b(256);
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(n_other);
    if (p->ndim < 3) {  // 1-D / 2-D: the fit kernel's selection form, a wave per block
        const uint32_t gfit = (uint32_t)std::min<uint64_t>(BLK_GRID, ((uint64_t)nblocks + 3) / 4);
#define BLKN_SEL(T)                                                                                                                      \
    do {                                                                                                                                 \
        if (p->ndim == 2)                                                                                                                \
            hipLaunchKernelGGL((k_blkn_fit<T, 1u, 4, true, true>), dim3(gfit), dim3(256), 0, s, (const T *)d_in, (uint16_t *)nullptr, *p, nblocks, cnt); \
        else                                                                                                                             \
            hipLaunchKernelGGL((k_blkn_fit<T, 1u, 4, false, true>), dim3(gfit), dim3(256), 0, s, (const T *)d_in, (uint16_t *)nullptr, *p, nblocks, cnt); \
    } while (0)
        if (dtype == 0) BLKN_SEL(float);
        else BLKN_SEL(double);
#undef BLKN_SEL
        SZK_CHECK_LAUNCH();
        return 0;
    }
#define BLK_SEL(T, CBV)                                                                                                       \
    do {                                                                                                                      \
        if (p->mask & 2u) hipLaunchKernelGGL((k_blk_select<T, CBV, true>), g, b, 0, s, (const T *)d_in, *p, nblocks, cnt);     \
        else hipLaunchKernelGGL((k_blk_select<T, CBV, false>), g, b, 0, s, (const T *)d_in, *p, nblocks, cnt);                 \
    } while (0)
    if (dtype == 0) {
        if (p->B == 6) BLK_SEL(float, 6);
        else BLK_SEL(float, 0);
    } else {
        if (p->B == 6) BLK_SEL(double, 6);
        else BLK_SEL(double, 0);
    }
#undef BLK_SEL
    SZK_CHECK_LAUNCH();
    return 0;
}

// worst case: every block a regression block, every coefficient escaped (4 x 88 bits)
size_t szk_blk_side_bound(uint64_t nblocks) { return SIDE_HDR + ((nblocks + 3) / 4 + 16) + 16 + 4 * (nblocks / RICE_GROUP + 1) + nblocks * 55 + 64; }  // (five coefficients of up to 88 bits each: 4-D)

// the side section of a block stream -> choices, ranks, coefficients (0.7 ms of small, serial kernels at C4's slab: the decoder runs
// them on a stream of their own beside the Huffman decoder, which they do not depend on)
int szk_launch_blk_side(const szk_blk_params *p, const szk_blk_scratch *sc, const uint8_t *payload, const szh_offsets *o, int64_t *coef_by_rank,
                        hipStream_t s) {
    const uint32_t nblocks = blk_count_blocks(p);
    const uint8_t *side = payload + o->side;
    // the side header was validated by the host (coding, counts, lengths)
    uint64_t nr, bit_words;
    memcpy(&nr, &sc->side_hdr[16], 8);
    memcpy(&bit_words, &sc->side_hdr[24], 8);
    hipLaunchKernelGGL(k_blk_side_sel, dim3((nblocks + 255) / 256 < 1024 ? (nblocks + 255) / 256 : 1024), dim3(256), 0, s, side, nblocks, p->sel);
    launch_blk_rank(p->sel, nblocks, sc->rank, (uint32_t *)nullptr, sc->run_scratch, sc->counters + 0, s);
    if (nr) {
        const uint64_t ngroups = (nr + RICE_GROUP - 1) / RICE_GROUP;
        int64_t *gsum = reinterpret_cast<int64_t *>(sc->comp);  // (the decoder has no other use for the compacted list's array: ngroups * 32 <= nblocks * 4 bytes)
        const dim3 gp((uint32_t)std::min<uint64_t>(4096, (ngroups + 3) / 4)), ga((uint32_t)std::min<uint64_t>(2048, (ngroups + 3) / 4));  // (a wave per group in both)
        if (p->ndim == 4) {
            hipLaunchKernelGGL(k_blk_coef_parse<5>, gp, dim3(256), 0, s, side, nblocks, nr, bit_words, coef_by_rank, gsum);
            hipLaunchKernelGGL(k_blk_coef_gscan<5>, dim3(1), dim3(1024), 0, s, ngroups, gsum);
            hipLaunchKernelGGL(k_blk_coef_apply<5>, ga, dim3(256), 0, s, nr, gsum, coef_by_rank);
        } else {
            hipLaunchKernelGGL(k_blk_coef_parse<4>, gp, dim3(256), 0, s, side, nblocks, nr, bit_words, coef_by_rank, gsum);
            hipLaunchKernelGGL(k_blk_coef_gscan<4>, dim3(1), dim3(1024), 0, s, ngroups, gsum);
            hipLaunchKernelGGL(k_blk_coef_apply<4>, ga, dim3(256), 0, s, nr, gsum, coef_by_rank);
        }
    }
    SZK_CHECK_LAUNCH();
    return 0;
}
// (dbg = the debug flags in force: the caller's retry ORs in the switches that take the launch-per-front decoders; *ctl_out: the
// control words of a one-launch decoder, when one was taken — [1] is raised by a flag poll that gave up)
static int blk_decompress_impl(int dtype, const uint16_t *codes, void *d_out, const szk_blk_params *p, const szk_blk_scratch *sc,
                               const uint8_t *payload, const szh_header *h, const szh_offsets *o, int64_t *coef_by_rank, hipStream_t s,
                               hipEvent_t side_done, const int dbg, uint32_t **ctl_out) {
    *ctl_out = nullptr;
    const uint32_t nblocks = blk_count_blocks(p);
    // (k_blk_local3v reads the codes themselves: the far deltas alone go to the work array; debug flag 16: the expanded copy and the
    // wave-per-block pass)
    const bool fusedv = p->ndim == 3 && p->B == 6 && p->carry && !(dbg & (32768 | 65536 | 8388608 | 16));
    const bool wave2 = p->ndim == 2 && !(p->mask & 2u) && p->B <= 16 && p->carry && !(dbg & (8388608 | 65536));  // (k_blkn_wave2 reads the codes too)
    if (p->ndim == 1 || fusedv || wave2) {
        if (szk_launch_scatter_deltas(dtype, h->n, payload, o, h->n_dout, p->qwork, s)) return -1;
    } else if (szk_launch_expand_deltas(dtype, codes, h->n, (int)h->radius, payload, o, h->n_dout, p->qwork, s)) return -1;
    // (the side section's kernels ran on another stream: the fronts are the first to need what they made)
    if (side_done && hipStreamWaitEvent(s, side_done, 0) != hipSuccess) return -1;
    if (p->ndim == 4) {  // fronts of blocks, bw + bz + by + bx = const
        const uint32_t g4 = (uint32_t)std::min<uint64_t>(BLK_GRID, ((uint64_t)nblocks + 3) / 4);
        const uint32_t ndiag = p->nbw + p->nb[0] + p->nb[1] + p->nb[2] - 3, gfront = p->nbw * p->nb[0] * p->nb[1];
#define BLK4_DEC(T)                                                                                                                   \
    do {                                                                                                                              \
        hipLaunchKernelGGL(k_blk4_pre<T>, dim3(g4), dim3(256), 0, s, codes, d_out, *p, nblocks, sc->rank, coef_by_rank);               \
        for (uint32_t d = 0; d < ndiag; d++) hipLaunchKernelGGL(k_blk4_decode<T>, dim3(gfront), dim3(BLK4_DT), 0, s, p->qwork, d_out, *p, d); \
        hipLaunchKernelGGL(k_blk4_final<T>, dim3(g4), dim3(256), 0, s, codes, d_out, *p, nblocks, sc->rank, coef_by_rank);             \
        if (h->n_vout) hipLaunchKernelGGL(k_blk_patch<T>, dim3(256), dim3(256), 0, s, payload, o->vout_idx, o->vout_val, h->n_vout, h->n, (T *)d_out); \
    } while (0)
        if (dtype == 0) BLK4_DEC(float);
        else BLK4_DEC(double);
#undef BLK4_DEC
        SZK_CHECK_LAUNCH();
        return 0;
    }
    if (p->ndim == 1 && (p->mask & 2u)) {  // second-order Lorenzo in the set: the scan of affine maps (k_blkn2_*)
        const uint32_t gpre = (uint32_t)std::min<uint64_t>(BLK_GRID, ((uint64_t)nblocks + 3) / 4);
        const uint32_t ntiles = (nblocks + BLKN_TILE - 1) / BLKN_TILE;
        // blocks of up to 128 values, a multiple of 8: four blocks per wave (debug flag 134217728: a wave per block)
        const bool rows = p->B <= 128 && p->B % 8 == 0 && !(dbg & 134217728);
        const uint32_t grow = (uint32_t)std::min<uint64_t>(BLK_GRID, ((uint64_t)nblocks + 15) / 16);
#define BLKN2_DEC(T, QT)                                                                                                                          \
    do {                                                                                                                                          \
        if (rows) hipLaunchKernelGGL(k_blkn2_pre_rows<T>, dim3(grow), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank); \
        else hipLaunchKernelGGL(k_blkn2_pre<T>, dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank);         \
        hipLaunchKernelGGL(k_blkn2_tile<QT>, dim3(ntiles), dim3(1024), 0, s, p->sel, nblocks, p->B, (uint32_t)p->d[2], p->carry);                  \
        hipLaunchKernelGGL(k_blkn2_top<QT>, dim3(1), dim3(1024), 0, s, ntiles, (QT *)p->carry + 2 * (uint64_t)nblocks);                            \
        if (rows) hipLaunchKernelGGL((k_blkn2_apply<T, true>), dim3(ntiles), dim3(1024), 0, s, codes, p->qwork, d_out, *p, nblocks);               \
        else hipLaunchKernelGGL((k_blkn2_apply<T, false>), dim3(ntiles), dim3(1024), 0, s, codes, p->qwork, d_out, *p, nblocks);                   \
    } while (0)
        if (dtype == 0) BLKN2_DEC(float, int32_t);
        else BLKN2_DEC(double, int64_t);
#undef BLKN2_DEC
    } else
    if (wave2) {
        // 2-D, first-order Lorenzo + regression, block edges up to 16: ONE launch for the chain of fronts (k_blkn_wave2; debug flag
        // 65536: groups of 4 x 4 blocks with a launch per front, k_blkn_decode2g)
        const uint32_t gpre = (uint32_t)std::min<uint64_t>(BLK_GRID, ((uint64_t)nblocks + 3) / 4);
        const uint32_t ng1 = (p->nb[1] + BLKN_G - 1) / BLKN_G, ng2 = (p->nb[2] + BLKN_G - 1) / BLKN_G;
        const uint64_t nslots = (uint64_t)ng1 * ng2;
        uint32_t *ctl = reinterpret_cast<uint32_t *>(p->carry);
        if (hipMemsetAsync(ctl, 0, (4 + (size_t)nslots) * 4, s) != hipSuccess) return -1;
        *ctl_out = ctl;
        const uint32_t gw = (uint32_t)std::min<uint64_t>(nslots, dtype == 0 ? 2048 : 1024);
        if (dtype == 0) {
            hipLaunchKernelGGL(k_blkn_pre<float>, dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank, 1);
            hipLaunchKernelGGL(k_blkn_wave2<float>, dim3(gw), dim3(256), 0, s, codes, p->qwork, d_out, *p, ctl, (uint32_t)nslots);
            if (h->n_vout) hipLaunchKernelGGL(k_blk_patch<float>, dim3(256), dim3(256), 0, s, payload, o->vout_idx, o->vout_val, h->n_vout, h->n, (float *)d_out);
        } else {
            hipLaunchKernelGGL(k_blkn_pre<double>, dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank, 1);
            hipLaunchKernelGGL(k_blkn_wave2<double>, dim3(gw), dim3(256), 0, s, codes, p->qwork, d_out, *p, ctl, (uint32_t)nslots);
            if (h->n_vout) hipLaunchKernelGGL(k_blk_patch<double>, dim3(256), dim3(256), 0, s, payload, o->vout_idx, o->vout_val, h->n_vout, h->n, (double *)d_out);
        }
        SZK_CHECK_LAUNCH();
        return 0;
    } else
    if (p->ndim < 3) {
        const uint32_t gpre = (uint32_t)std::min<uint64_t>(BLK_GRID, ((uint64_t)nblocks + 3) / 4);
        // 1-D, blocks of up to 128 values, a multiple of 8: four blocks per wave (debug flag 134217728: a wave per block)
        const bool rows1 = p->ndim == 1 && p->B <= 128 && p->B % 8 == 0 && !(dbg & 134217728);
        const uint32_t grow1 = (uint32_t)std::min<uint64_t>(BLK_GRID, ((uint64_t)nblocks + 15) / 16);
        if (rows1) {
            if (dtype == 0) hipLaunchKernelGGL(k_blkn_pre1_rows<float>, dim3(grow1), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank);
            else hipLaunchKernelGGL(k_blkn_pre1_rows<double>, dim3(grow1), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank);
        } else
        if (dtype == 0) hipLaunchKernelGGL(k_blkn_pre<float>, dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank);
        else hipLaunchKernelGGL(k_blkn_pre<double>, dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank);
        if (p->ndim == 1) {
            const uint32_t ntiles = (nblocks + BLKN_TILE - 1) / BLKN_TILE;
            // blocks of up to 128 values, a multiple of 8: four blocks per wave (debug flag 134217728: a wave per block)
            const bool rows = p->B <= 128 && p->B % 8 == 0 && !(dbg & 134217728);
            const uint32_t grow = (uint32_t)std::min<uint64_t>(BLK_GRID, ((uint64_t)nblocks + 15) / 16);
            if (dtype == 0) {
                hipLaunchKernelGGL(k_blkn_scan_tile<int32_t>, dim3(ntiles), dim3(1024), 0, s, p->sel, nblocks, p->carry);
                hipLaunchKernelGGL(k_blkn_scan_top<int32_t>, dim3(1), dim3(1024), 0, s, ntiles, (int32_t *)p->carry + 2 * (uint64_t)nblocks);
                if (rows) hipLaunchKernelGGL(k_blkn_apply1_rows<float>, dim3(grow), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks);
                else hipLaunchKernelGGL(k_blkn_apply1<float>, dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks);
            } else {
                hipLaunchKernelGGL(k_blkn_scan_tile<int64_t>, dim3(ntiles), dim3(1024), 0, s, p->sel, nblocks, p->carry);
                hipLaunchKernelGGL(k_blkn_scan_top<int64_t>, dim3(1), dim3(1024), 0, s, ntiles, (int64_t *)p->carry + 2 * (uint64_t)nblocks);
                if (rows) hipLaunchKernelGGL(k_blkn_apply1_rows<double>, dim3(grow), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks);
                else hipLaunchKernelGGL(k_blkn_apply1<double>, dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks);
            }
        } else if (p->mask & 2u) {  // second-order Lorenzo in the set: the plain form with two halo layers
            const uint32_t ndiag = p->nb[1] + p->nb[2] - 1;
            for (uint32_t d = 0; d < ndiag; d++) {
                const uint32_t by_lo = d >= p->nb[2] ? d - (p->nb[2] - 1) : 0, by_hi = d < p->nb[1] - 1 ? d : p->nb[1] - 1;
                const uint32_t nfront = by_hi - by_lo + 1;
                if (dtype == 0) hipLaunchKernelGGL(k_blkn_decode2s<float>, dim3((nfront + 3) / 4), dim3(256), 0, s, p->qwork, d_out, *p, d, by_lo, nfront);
                else hipLaunchKernelGGL(k_blkn_decode2s<double>, dim3((nfront + 3) / 4), dim3(256), 0, s, p->qwork, d_out, *p, d, by_lo, nfront);
            }
        } else if (p->B <= 16 && !(dbg & 8388608)) {  // groups of 4 x 4 blocks per workgroup, a launch per front (debug flag 65536; 8388608: a block per wave)
            const uint32_t ng1 = (p->nb[1] + BLKN_G - 1) / BLKN_G, ng2 = (p->nb[2] + BLKN_G - 1) / BLKN_G;
            for (uint32_t d = 0; d < ng1 + ng2 - 1; d++) {
                const uint32_t gy_lo = d >= ng2 ? d - (ng2 - 1) : 0, gy_hi = d < ng1 - 1 ? d : ng1 - 1;
                if (dtype == 0) hipLaunchKernelGGL(k_blkn_decode2g<float>, dim3(gy_hi - gy_lo + 1), dim3(256), 0, s, p->qwork, d_out, *p, d, gy_lo);
                else hipLaunchKernelGGL(k_blkn_decode2g<double>, dim3(gy_hi - gy_lo + 1), dim3(256), 0, s, p->qwork, d_out, *p, d, gy_lo);
            }
        } else {
            const uint32_t ndiag = p->nb[1] + p->nb[2] - 1;
            for (uint32_t d = 0; d < ndiag; d++) {
                const uint32_t by_lo = d >= p->nb[2] ? d - (p->nb[2] - 1) : 0, by_hi = d < p->nb[1] - 1 ? d : p->nb[1] - 1;
                const uint32_t nfront = by_hi - by_lo + 1;
                if (dtype == 0) hipLaunchKernelGGL(k_blkn_decode2<float>, dim3((nfront + 3) / 4), dim3(256), 0, s, p->qwork, d_out, *p, d, by_lo, nfront);
                else hipLaunchKernelGGL(k_blkn_decode2<double>, dim3((nfront + 3) / 4), dim3(256), 0, s, p->qwork, d_out, *p, d, by_lo, nfront);
            }
        }
    } else
    if (p->B == 6 && p->carry && !(dbg & (32768 | 65536 | 8388608))) {  // one launch for the chain of fronts (k_blk_wave3)
        constexpr uint32_t G = 3;
        const uint32_t ng0 = (p->nb[0] + G - 1) / G, ng1 = (p->nb[1] + G - 1) / G, ng2 = (p->nb[2] + G - 1) / G;
        const uint32_t ngd = ng0 + ng1 + ng2 - 2;
        const uint64_t nslots = (uint64_t)ng0 * ng1 * ng2;  // (the groups, handed out front by front)
        if (nslots > 0xFFFFFFF0ull) return -1;
        uint32_t *ctl = reinterpret_cast<uint32_t *>(p->carry);
        if (hipMemsetAsync(ctl, 0, (4 + (size_t)ng0 * ng1 * ng2) * 4, s) != hipSuccess) return -1;
        *ctl_out = ctl;
        const uint32_t gpre = (uint32_t)std::min<uint64_t>(BLK_GRID, ((uint64_t)nblocks + 3) / 4);
        const uint32_t gw = (uint32_t)std::min<uint64_t>(nslots, dtype == 0 ? 1024 : 512);
        const bool ragged = p->d[0] % 6 || p->d[1] % 6 || p->d[2] % 6;
        const uint32_t gv = (uint32_t)std::min<uint64_t>(8192, ((uint64_t)nblocks + BLK3V_NB - 1) / BLK3V_NB);
        if (dtype == 0) {
            if (fusedv) {
                hipLaunchKernelGGL((k_blk_local3v<float, 6>), dim3(gv), dim3(192), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank);
                if (ragged) hipLaunchKernelGGL((k_blk_local3<float, 6, true>), dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank, 1, 1);
            } else hipLaunchKernelGGL((k_blk_local3<float, 6, true>), dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank, 0, 0);
            if (p->mask & 2u) hipLaunchKernelGGL((k_blk_wave3<float, 6, 3, 2>), dim3(gw), dim3(512), 0, s, p->qwork, d_out, *p, ctl, (uint32_t)nslots);
            else hipLaunchKernelGGL((k_blk_wave3<float, 6, 3, 1>), dim3(gw), dim3(512), 0, s, p->qwork, d_out, *p, ctl, (uint32_t)nslots);
            if (h->n_vout) hipLaunchKernelGGL(k_blk_patch<float>, dim3(256), dim3(256), 0, s, payload, o->vout_idx, o->vout_val, h->n_vout, h->n, (float *)d_out);
        } else {
            if (fusedv) {
                hipLaunchKernelGGL((k_blk_local3v<double, 6>), dim3(gv), dim3(192), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank);
                if (ragged) hipLaunchKernelGGL((k_blk_local3<double, 6, true>), dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank, 1, 1);
            } else hipLaunchKernelGGL((k_blk_local3<double, 6, true>), dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank, 0, 0);
            if (p->mask & 2u) hipLaunchKernelGGL((k_blk_wave3<double, 6, 3, 2>), dim3(gw), dim3(512), 0, s, p->qwork, d_out, *p, ctl, (uint32_t)nslots);
            else hipLaunchKernelGGL((k_blk_wave3<double, 6, 3, 1>), dim3(gw), dim3(512), 0, s, p->qwork, d_out, *p, ctl, (uint32_t)nslots);
            if (h->n_vout) hipLaunchKernelGGL(k_blk_patch<double>, dim3(256), dim3(256), 0, s, payload, o->vout_idx, o->vout_val, h->n_vout, h->n, (double *)d_out);
        }
        SZK_CHECK_LAUNCH();
        return 0;
    } else
    if (p->B == 6 && (dbg & 32768)) {  // debug flag 32768: groups of 3 x 3 x 3 blocks per workgroup, closed form, a launch per front (k_blk_decode_gf)
        constexpr uint32_t G = 3;
        const uint32_t ng0 = (p->nb[0] + G - 1) / G, ng1 = (p->nb[1] + G - 1) / G, ng2 = (p->nb[2] + G - 1) / G;
        const uint32_t ngd = ng0 + ng1 + ng2 - 2;
        {
            const uint32_t gpre = (uint32_t)std::min<uint64_t>(BLK_GRID, ((uint64_t)nblocks + 3) / 4);
            if (dtype == 0) hipLaunchKernelGGL((k_blk_local3<float, 6, false>), dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank);
            else hipLaunchKernelGGL((k_blk_local3<double, 6, false>), dim3(gpre), dim3(256), 0, s, codes, p->qwork, d_out, *p, nblocks, sc->rank, coef_by_rank);
        }
        for (uint32_t d = 0; d < ngd; d++) {
            const uint32_t rest = (ng1 - 1) + (ng2 - 1);
            const uint32_t gz_lo = d > rest ? d - rest : 0, gz_hi = d < ng0 - 1 ? d : ng0 - 1;
            if (gz_lo > gz_hi) continue;
            const uint32_t npairs = (gz_hi - gz_lo + 1) * ng1;
            if (dtype == 0) hipLaunchKernelGGL((k_blk_decode_gf<float, 6, 3>), dim3(npairs), dim3(512), 0, s, d_out, *p, d, gz_lo, npairs);
            else hipLaunchKernelGGL((k_blk_decode_gf<double, 6, 3>), dim3(npairs), dim3(512), 0, s, d_out, *p, d, gz_lo, npairs);
        }
    } else
#ifdef SZ3HIP_LAB
    if (p->B == 6 && !(dbg & 8388608)) {  // round 3's form (lab builds, debug flag 65536): groups of 2 x 2 x 2 blocks, line scans (debug flag 8388608: a block per wave)
        const uint32_t ng0 = (p->nb[0] + 1) / 2, ng1 = (p->nb[1] + 1) / 2, ng2 = (p->nb[2] + 1) / 2;
        const uint32_t ngd = ng0 + ng1 + ng2 - 2;
        {
            const uint32_t gpre = (uint32_t)std::min<uint64_t>(BLK_GRID, ((uint64_t)nblocks + 3) / 4);
            if (dtype == 0) hipLaunchKernelGGL((k_blk_pre3<float, 6>), dim3(gpre), dim3(256), 0, s, codes, d_out, *p, nblocks, sc->rank, coef_by_rank);
            else hipLaunchKernelGGL((k_blk_pre3<double, 6>), dim3(gpre), dim3(256), 0, s, codes, d_out, *p, nblocks, sc->rank, coef_by_rank);
        }
        for (uint32_t d = 0; d < ngd; d++) {
            const uint32_t rest = (ng1 - 1) + (ng2 - 1);
            const uint32_t gz_lo = d > rest ? d - rest : 0, gz_hi = d < ng0 - 1 ? d : ng0 - 1;
            if (gz_lo > gz_hi) continue;
            const uint32_t npairs = (gz_hi - gz_lo + 1) * ng1;
            if (dtype == 0)
                hipLaunchKernelGGL((k_blk_decode_g<float, 6>), dim3(npairs), dim3(512), 0, s, codes, p->qwork, d_out, *p, d, gz_lo, npairs, sc->rank, coef_by_rank);
            else
                hipLaunchKernelGGL((k_blk_decode_g<double, 6>), dim3(npairs), dim3(512), 0, s, codes, p->qwork, d_out, *p, d, gz_lo, npairs, sc->rank, coef_by_rank);
        }
    } else
#endif
    {
    const uint32_t ndiag = p->nb[0] + p->nb[1] + p->nb[2] - 2;
    for (uint32_t d = 0; d < ndiag; d++) {
        // blocks of the front: bz + by + bx = d; only the bz that can have a partner (by, bx) are enumerated
        const uint32_t rest = (p->nb[1] - 1) + (p->nb[2] - 1);
        const uint32_t bz_lo = d > rest ? d - rest : 0, bz_hi = d < p->nb[0] - 1 ? d : p->nb[0] - 1;
        if (bz_lo > bz_hi) continue;
        const uint32_t npairs = (bz_hi - bz_lo + 1) * p->nb[1];
#define BLK_DEC(T, CBV) hipLaunchKernelGGL((k_blk_decode<T, CBV>), dim3((npairs + 3) / 4), dim3(256), 0, s, codes, p->qwork, d_out, *p, d, bz_lo, npairs, sc->rank, coef_by_rank)
        if (dtype == 0) {
            if (p->B == 6) BLK_DEC(float, 6);
            else BLK_DEC(float, 0);
        } else {
            if (p->B == 6) BLK_DEC(double, 6);
            else BLK_DEC(double, 0);
        }
#undef BLK_DEC
    }
    }
    const uint32_t grid = (uint32_t)std::min<uint64_t>(BLK_GRID, ((uint64_t)nblocks + 3) / 4);
    if (p->ndim == 1) {  // (the 1-D passes above wrote final values)
        if (h->n_vout) {
            if (dtype == 0) hipLaunchKernelGGL(k_blk_patch<float>, dim3(256), dim3(256), 0, s, payload, o->vout_idx, o->vout_val, h->n_vout, h->n, (float *)d_out);
            else hipLaunchKernelGGL(k_blk_patch<double>, dim3(256), dim3(256), 0, s, payload, o->vout_idx, o->vout_val, h->n_vout, h->n, (double *)d_out);
        }
        SZK_CHECK_LAUNCH();
        return 0;
    }
    if (dtype == 0) {
        if (p->B == 6) hipLaunchKernelGGL((k_blk_final<float, 6>), dim3(grid), dim3(256), 0, s, codes, d_out, *p, nblocks, sc->rank, coef_by_rank);
        else hipLaunchKernelGGL((k_blk_final<float, 0>), dim3(grid), dim3(256), 0, s, codes, d_out, *p, nblocks, sc->rank, coef_by_rank);
        if (h->n_vout) hipLaunchKernelGGL(k_blk_patch<float>, dim3(256), dim3(256), 0, s, payload, o->vout_idx, o->vout_val, h->n_vout, h->n, (float *)d_out);
    } else {
        if (p->B == 6) hipLaunchKernelGGL((k_blk_final<double, 6>), dim3(grid), dim3(256), 0, s, codes, d_out, *p, nblocks, sc->rank, coef_by_rank);
        else hipLaunchKernelGGL((k_blk_final<double, 0>), dim3(grid), dim3(256), 0, s, codes, d_out, *p, nblocks, sc->rank, coef_by_rank);
        if (h->n_vout) hipLaunchKernelGGL(k_blk_patch<double>, dim3(256), dim3(256), 0, s, payload, o->vout_idx, o->vout_val, h->n_vout, h->n, (double *)d_out);
    }
    SZK_CHECK_LAUNCH();
    return 0;
}
// The one-launch decoders (k_blk_wave3, k_blkn_wave2) wait for their lower neighbours' flags with a bounded poll; a poll that gives up
// (it never has: tickets are handed out in the fronts' order, every group a poll waits for is running or done) raises ctl[1] and
// leaves the array wrong. The word is fetched behind the launch — one host synchronisation on a decoder of milliseconds — and a
// raised word sends the stream through the launch-per-front decoders (the same arithmetic, no inter-workgroup waits), from the codes.
int szk_launch_blk_decompress(int dtype, const uint16_t *codes, void *d_out, const szk_blk_params *p, const szk_blk_scratch *sc,
                              const uint8_t *payload, const szh_header *h, const szh_offsets *o, int64_t *coef_by_rank, hipStream_t s,
                              hipEvent_t side_done) {
    uint32_t *ctl = nullptr;
    int rc = blk_decompress_impl(dtype, codes, d_out, p, sc, payload, h, o, coef_by_rank, s, side_done, szk_dbg_flags, &ctl);
    if (rc || !ctl) return rc;
    uint32_t gave_up = 0;
    if (hipMemcpyAsync(&gave_up, ctl + 1, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
    if (!gave_up && !(szk_dbg_flags & 4)) return 0;  // (debug flag 4: take the retry as if a poll had given up — tests)
    return blk_decompress_impl(dtype, codes, d_out, p, sc, payload, h, o, coef_by_rank, s, nullptr, szk_dbg_flags | 32768 | 65536, &ctl);
}

int szk_launch_trial_lorenzo12(int dtype, const void *d_samples, uint64_t per, uint64_t nsb, double eb, int radius, uint64_t *hist, uint64_t *counters,
                               uint64_t *extra, hipStream_t s) {
    const uint64_t total = nsb * ((per + 4) / 5);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(2048, (total + 255) / 256);
    if (!grid) return 0;
    const szk_lattice lat = szk_make_lattice(eb);
    if (dtype == 0)
        hipLaunchKernelGGL(k_trial_lorenzo12<float>, dim3(grid), dim3(256), 0, s, (const float *)d_samples, per, nsb, lat, eb, (uint32_t)radius,
                           (unsigned long long *)hist, (unsigned long long *)counters, (unsigned long long *)extra);
    else
        hipLaunchKernelGGL(k_trial_lorenzo12<double>, dim3(grid), dim3(256), 0, s, (const double *)d_samples, per, nsb, lat, eb, (uint32_t)radius,
                           (unsigned long long *)hist, (unsigned long long *)counters, (unsigned long long *)extra);
    SZK_CHECK_LAUNCH();
    return 0;
}
