// sz3_amd/csrc/sz3hip_interp.hip — multilevel spline-interpolation predictor on gfx950.
//
// Reference: InterpolationDecomposition<T,N,LinearQuantizer<T>> (include/SZ3/decomposition/InterpolationDecomposition.hpp)
// driven by SZ_compress_Interp (include/SZ3/api/impl/SZAlgoInterp.hpp:17-40), stencils of
// include/SZ3/utils/Interpolators.hpp:12-39, quantiser include/SZ3/quantizer/LinearQuantizer.hpp:43-86.
//
// Unlike the Lorenzo path this predictor has NO loop-carried dependency inside a directional pass: the reference
// walks level -> block (32*stride) -> direction, but every point predicted in pass k of a level reads only points of
// coarser levels or of earlier passes of the same level, whichever block they belong to (blocks only decide where a
// line is cut, i.e. which boundary stencil a point gets).  Re-ordering to level -> pass -> all points therefore
// reproduces the reference's predictions, quantisation codes and reconstructed values BIT FOR BIT, with one launch
// per (level, pass) and one thread per predicted point.  The single exception — the last point of an even-length line
// in linear mode (N >= 3) extrapolates from a point of the same pass (InterpolationDecomposition.hpp:345-351) — runs
// as a second, tiny launch of that pass.
// The arithmetic is the reference's: predictions in T, operand order of Interpolators.hpp; quantiser in double
// exactly as LinearQuantizer (no FMA contraction: -ffp-contract=off).  Codes are stored in element order (the
// reference emits them in traversal order — same multiset, same histogram, same Huffman cost).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "sz3hip_format.h"
#include "sz3hip_kernels.h"
#include "sz3hip_devutil.h"

#define IH_WIN 1024  // LDS histogram window (bins) around the radius


// ---- Interpolators.hpp:12-39 ---------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T ip_linear(T a, T b) { return (a + b) / 2; }
template <typename T> __device__ __forceinline__ T ip_linear1(T a, T b) { return (T)(-0.5 * (double)a + 1.5 * (double)b); }
template <typename T> __device__ __forceinline__ T ip_quad_1(T a, T b, T c) { return (3 * a + 6 * b - c) / 8; }
template <typename T> __device__ __forceinline__ T ip_quad_2(T a, T b, T c) { return (-a + 6 * b + 3 * c) / 8; }
template <typename T> __device__ __forceinline__ T ip_quad_3(T a, T b, T c) { return (3 * a - 10 * b + 15 * c) / 8; }
template <typename T> __device__ __forceinline__ T ip_cubic(T a, T b, T c, T d) { return (-a + 9 * b + 9 * c - d) / 16; }

// Unpredictable values (code 0: the raw value stays in the array) are NOT appended by the pass kernels: the histogram pass
// that reads every code anyway (k_hist_codes) collects their indices and values into the list, through per-wave LDS queues
// (a field with NaN / fill-value masks makes millions of them, and same-address global atomics run at ~90/us).
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2); giving every XCD a contiguous
// range of logical blocks keeps neighbouring rows / planes (read by several blocks) inside one L2
__device__ __forceinline__ uint32_t xcd_block() {
    const uint32_t g = gridDim.x;
    return g % 8u == 0 ? (blockIdx.x % 8u) * (g / 8u) + blockIdx.x / 8u : blockIdx.x;
}

// ---- one directional pass of one level: one thread per predicted point ----------------------------------------
// tuner trials in LDS: the code of a point goes straight into the trial's histogram (LDS window, global tail)
struct TrialSink {
    uint32_t *lh;
    unsigned long long *hist;
    uint32_t win_lo;
};
__device__ __forceinline__ void sink_code(const TrialSink *sk, uint32_t code) {
    const uint32_t bin = code - sk->win_lo;
    if (bin < IH_WIN) atomicAdd(&sk->lh[bin], 1u);
    else atomicAdd(&sk->hist[code], 1ull);
}
// IT: type of the point counter arithmetic (u32 when the pass has fewer than 2^32 points: 64-bit divisions cost ~100
// instructions each, and a tuner trial block is worked by ONE compute unit)
template <typename T, bool DEC, typename IT = uint64_t, bool SINK = false>
__device__ __forceinline__ void interp_point(T *__restrict__ w, uint16_t *__restrict__ codes, const szk_interp_pass &p, uint64_t t,
                                             uint64_t boff, const TrialSink *sink = nullptr, const T *__restrict__ orig = nullptr) {
    IT r = (IT)t;
    uint64_t idx = 0, cd = 0;
#pragma unroll
    for (int j = 3; j >= 0; j--) {
        if (j >= p.N) continue;
        const IT cj = (IT)p.cnt[j];
        const IT q = r % cj;
        r /= cj;
        const uint64_t c = p.start[j] + (uint64_t)q * p.step[j];
        idx += c * p.off[j];
        if (j == p.dir) cd = c;
    }
    const uint64_t D = p.dims[p.dir];
    const int ls = __ffsll((long long)p.s) - 1;  // s and bsz = 32 s are powers of two
    const uint64_t begin = cd & ~(p.bsz - 1);
    uint64_t end = begin + p.bsz;
    if (end > D - 1) end = D - 1;
    const uint64_t n = ((end - begin) >> ls) + 1, i = (cd - begin) >> ls;  // i is odd, 1 <= i <= n-1
    const int64_t st = (int64_t)(p.s * p.off[p.dir]);
    T *d = w + idx;
    bool deferred = false;
    T pred;
    if (p.old_api) {  // interpolation_1d, InterpolationDecomposition.hpp:248-293 (N <= 2)
        if (p.interp_id == 0 || n < 5) {
            if (i + 1 < n) pred = ip_linear<T>(d[-st], d[st]);
            else pred = n < 4 ? d[-st] : ip_linear1<T>(d[-3 * st], d[-st]);
        } else {
            if (i == 1) pred = ip_quad_1<T>(d[-st], d[st], d[3 * st]);
            else if (i + 3 < n) pred = ip_cubic<T>(d[-3 * st], d[-st], d[st], d[3 * st]);
            else if (i + 1 < n) pred = ip_quad_2<T>(d[-3 * st], d[-st], d[st]);
            else pred = ip_quad_3<T>(d[-5 * st], d[-3 * st], d[-st]);
        }
    } else if (p.interp_id == 0) {  // interpolation_1d_fastest_dim_first, linear branch :334-351
        if (i + 1 < n) {
            pred = ip_linear<T>(d[-st], d[st]);
        } else if (n < 3) {
            pred = d[-st];
        } else {
            deferred = true;  // reads d[-2*st]: a point of this same pass -> second launch
            pred = p.subpass ? ip_linear1<T>(d[-2 * st], d[-st]) : (T)0;
        }
    } else {  // cubic branch :352-399
        if (i >= 3) {
            if (i + 3 < n) pred = ip_cubic<T>(d[-3 * st], d[-st], d[st], d[3 * st]);
            else if (i + 1 < n) pred = ip_quad_2<T>(d[-3 * st], d[-st], d[st]);
            else pred = ip_linear1<T>(d[-3 * st], d[-st]);
        } else {
            if (i + 3 < n) pred = ip_quad_1<T>(d[-st], d[st], d[3 * st]);
            else if (i + 1 < n) pred = ip_linear<T>(d[-st], d[st]);
            else pred = d[-st];
        }
    }
    if ((p.subpass != 0) != deferred) return;
    if (DEC) {
        const int code = codes[idx];
        if (code) *d = ref_recover<T>(pred, code, p.eb, p.radius);  // code 0: raw value already scattered in place
    } else {
        // orig: the work array holds reconstructions only (no working copy of the input was made), the original comes from
        // the caller's array and an unpredictable point's raw value is stored like a reconstruction
        T v = orig ? orig[idx] : *d;
        const int code = ref_quantize<T>(v, pred, p.eb, p.eb_recip, p.radius);
        if (SINK) {
            sink_code(sink, (uint32_t)code);
            if (codes) codes[idx] = (uint16_t)code;  // (the tuner's exact pricing wants the trial's codes per element besides their histogram)
        } else {
            codes[idx] = (uint16_t)code;
        }
        if ((code || orig) && !p.no_store) *d = v;  // (unpredictable, code 0: the raw value stays; LinearQuantizer "unpred")
    }
}
template <typename T, bool DEC>
__global__ __launch_bounds__(256) void k_interp_pass(T *__restrict__ w, uint16_t *__restrict__ codes, szk_interp_pass p, const T *__restrict__ orig) {
    const uint64_t t = (uint64_t)xcd_block() * 256 + threadIdx.x;
    if (t >= p.total) return;
    const uint64_t boff = (uint64_t)blockIdx.y * p.batch_stride;  // independent arrays of one batch
    if (p.total <= 0xFFFFFFFFull) interp_point<T, DEC, uint32_t>(w + boff, codes + boff, p, t, boff, nullptr, orig);  // (uniform branch)
    else interp_point<T, DEC, uint64_t>(w + boff, codes + boff, p, t, boff, nullptr, orig);
}

// ---- level 1 (stride 1), cubic, N >= 3, row length a multiple of 4: 8 consecutive x per thread (a row may end in a half group) ----
// The finest level holds 7/8 of all points. One thread owns 8 consecutive elements of a row (two 16-byte accesses per
// array row it touches) instead of one 4-byte access per neighbour; operands, formulas and their order are exactly those
// of interp_point, so codes and reconstruction stay bit-identical.
//   XDIR = false: the pass runs along a slower dimension: the four neighbour rows are loaded as vectors, the case
//                 (cubic / quad / linear at the line ends) is uniform for the thread; only every xstep-th x is a point.
//   XDIR = true:  the pass runs along x: a 16-element window [x0 - 4, x0 + 12) supplies the even neighbours of the four odd
//                 points; the case is chosen per point.
template <typename T>
struct Vec8 {
    T v[8];
};
template <typename T>
__device__ __forceinline__ void ld8(const T *p, T (&o)[8]) {
    if (sizeof(T) == 4) {
        const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
        o[0] = (T)a.x; o[1] = (T)a.y; o[2] = (T)a.z; o[3] = (T)a.w; o[4] = (T)b.x; o[5] = (T)b.y; o[6] = (T)b.z; o[7] = (T)b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double2 a = reinterpret_cast<const double2 *>(p)[k];
            o[2 * k] = (T)a.x;
            o[2 * k + 1] = (T)a.y;
        }
    }
}
// rows whose length is a multiple of 4 but not of 8 end in a half group (4 valid elements): its upper half is read from the
// lower half's address again (never past the row, never conditional) and not written
template <typename T>
__device__ __forceinline__ void ld8m(const T *p, T (&o)[8], bool full) {
    const T *q = p + (full ? 4 : 0);
    if (sizeof(T) == 4) {
        const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(q)[0];
        o[0] = (T)a.x; o[1] = (T)a.y; o[2] = (T)a.z; o[3] = (T)a.w; o[4] = (T)b.x; o[5] = (T)b.y; o[6] = (T)b.z; o[7] = (T)b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const double2 a = reinterpret_cast<const double2 *>(p)[k], b = reinterpret_cast<const double2 *>(q)[k];
            o[2 * k] = (T)a.x;
            o[2 * k + 1] = (T)a.y;
            o[4 + 2 * k] = (T)b.x;
            o[4 + 2 * k + 1] = (T)b.y;
        }
    }
}
template <typename T>
__device__ __forceinline__ void ld4(const T *p, T (&o)[4]) {
    if (sizeof(T) == 4) {
        const float4 a = reinterpret_cast<const float4 *>(p)[0];
        o[0] = (T)a.x; o[1] = (T)a.y; o[2] = (T)a.z; o[3] = (T)a.w;
    } else {
        const double2 a = reinterpret_cast<const double2 *>(p)[0], b = reinterpret_cast<const double2 *>(p)[1];
        o[0] = (T)a.x; o[1] = (T)a.y; o[2] = (T)b.x; o[3] = (T)b.y;
    }
}
template <typename T>
__device__ __forceinline__ void st8m(T *p, const T (&o)[8], bool full) {
    if (sizeof(T) == 4) {
        reinterpret_cast<float4 *>(p)[0] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        if (full) reinterpret_cast<float4 *>(p)[1] = make_float4((float)o[4], (float)o[5], (float)o[6], (float)o[7]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k < 2 || full) reinterpret_cast<double2 *>(p)[k] = make_double2((double)o[2 * k], (double)o[2 * k + 1]);
    }
}
template <typename T>
__device__ __forceinline__ void st8(T *p, const T (&o)[8]) {
    if (sizeof(T) == 4) {
        reinterpret_cast<float4 *>(p)[0] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        reinterpret_cast<float4 *>(p)[1] = make_float4((float)o[4], (float)o[5], (float)o[6], (float)o[7]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) reinterpret_cast<double2 *>(p)[k] = make_double2((double)o[2 * k], (double)o[2 * k + 1]);
    }
}
// the 1-D / 2-D rules (interpolation_1d, InterpolationDecomposition.hpp:248-293) on gathered neighbours
template <typename T>
__device__ __forceinline__ T old_rule(uint64_t i, uint64_t n, T m5, T m3, T m1, T p1, T p3) {
    if (n < 5) {
        if (i + 1 < n) return ip_linear<T>(m1, p1);
        return n < 4 ? m1 : ip_linear1<T>(m3, m1);
    }
    if (i == 1) return ip_quad_1<T>(m1, p1, p3);
    if (i + 3 < n) return ip_cubic<T>(m3, m1, p1, p3);
    if (i + 1 < n) return ip_quad_2<T>(m3, m1, p1);
    return ip_quad_3<T>(m5, m3, m1);
}
// OLD: the case rules of the 1-D / 2-D interface (same block structure, different boundary formulas)
// IT: type of the thread-index arithmetic (u32 when the pass has fewer than 2^32 groups: a 64-bit division costs ~100
// instructions, and the three of them were most of the pass along x's instruction count)
template <typename T, bool DEC, bool XDIR, bool OLD = false, typename IT = uint64_t>
__global__ __launch_bounds__(256) void k_interp_vec(T *__restrict__ w, uint16_t *__restrict__ codes, szk_interp_pass p) {
    const int N = p.N;
    const uint64_t dx = p.dims[N - 1];  // dx is a multiple of 4: the last group of a row may hold 4 elements
    const IT xg = (IT)((dx + 7) / 8);
    const uint64_t t0 = (uint64_t)xcd_block() * 256 + threadIdx.x;
    const bool valid = t0 < p.total;  // total = rows * xg here
    const IT t = valid ? (IT)t0 : (IT)0;
    const IT tx = t % xg;
    IT r = t / xg;
    uint64_t idx = 0, cd = 0;
#pragma unroll
    for (int j = 2; j >= 0; j--) {
        if (j >= N - 1) continue;
        const IT cj = (IT)p.cnt[j];
        const IT q = r % cj;
        r /= cj;
        const uint64_t c = p.start[j] + (uint64_t)q * p.step[j];
        idx += c * p.off[j];
        if (j == p.dir) cd = c;
    }
    const uint64_t x0 = (uint64_t)tx * 8;
    const bool full = x0 + 8 <= dx;  // else 4 valid elements: the upper halves below are dummies, never stored
    idx += x0;
    T o[8];
    ld8m<T>(w + idx, o, full);
    // Codes of the 8 elements. Decompression reads them. A compression pass along x keeps the even-x codes of earlier
    // passes; a pass along a slower dimension owns every slot it writes (with xstep = 2 the odd-x slots belong to the
    // later pass along x, which overwrites them), so it does not read the old words at all.
    uint32_t cw[4] = {0u, 0u, 0u, 0u};
    if (DEC || XDIR) {
        const uint2 c0 = *reinterpret_cast<const uint2 *>(codes + idx), c1 = *reinterpret_cast<const uint2 *>(codes + idx + (full ? 4 : 0));
        cw[0] = c0.x; cw[1] = c0.y; cw[2] = c1.x; cw[3] = c1.y;
    }
    auto get_code = [&](int e) -> int { return (int)((cw[e >> 1] >> (16 * (e & 1))) & 0xFFFFu); };
    auto set_code = [&](int e, int c) { cw[e >> 1] = (cw[e >> 1] & ~(0xFFFFu << (16 * (e & 1)))) | ((uint32_t)c << (16 * (e & 1))); };
    auto finish = [&](int e, T pred) {
        if (DEC) {
            const int code = get_code(e);
            if (code) o[e] = ref_recover<T>(pred, code, p.eb, p.radius);
        } else {
            T v = o[e];
            const int code = ref_quantize<T>(v, pred, p.eb, p.eb_recip, p.radius);
            set_code(e, code);
            if (code) o[e] = v;  // (code 0: the raw value stays in o[e])
        }
    };
    if (!XDIR) {
        const uint64_t D = p.dims[p.dir];
        const uint64_t begin = (cd / 32) * 32;
        uint64_t end = begin + 32;
        if (end > D - 1) end = D - 1;
        const uint64_t n = end - begin + 1, i = cd - begin;
        const int64_t st = (int64_t)p.off[p.dir];
        const int xstep = (int)p.step[N - 1];
        T a[8], b[8], c[8], d[8];
        const T *base = w + idx;
        if (OLD) {
            T a5[8];
            if (i >= 3) ld8m<T>(base - 3 * st, a, full);
            ld8m<T>(base - st, b, full);
            if (i + 1 < n) ld8m<T>(base + st, c, full);
            if (i + 3 < n) ld8m<T>(base + 3 * st, d, full);
            if (i + 1 >= n && n >= 5) ld8m<T>(base - 5 * st, a5, full);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (e % xstep || (!full && e >= 4)) continue;
                finish(e, old_rule<T>(i, n, a5[e], a[e], b[e], c[e], d[e]));
            }
        } else if (i >= 3) {
            ld8m<T>(base - 3 * st, a, full);
            ld8m<T>(base - st, b, full);
            if (i + 1 < n) ld8m<T>(base + st, c, full);
            if (i + 3 < n) ld8m<T>(base + 3 * st, d, full);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (e % xstep || (!full && e >= 4)) continue;
                T pred;
                if (i + 3 < n) pred = ip_cubic<T>(a[e], b[e], c[e], d[e]);
                else if (i + 1 < n) pred = ip_quad_2<T>(a[e], b[e], c[e]);
                else pred = ip_linear1<T>(a[e], b[e]);
                finish(e, pred);
            }
        } else {
            ld8m<T>(base - st, b, full);
            if (i + 1 < n) ld8m<T>(base + st, c, full);
            if (i + 3 < n) ld8m<T>(base + 3 * st, d, full);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (e % xstep || (!full && e >= 4)) continue;
                T pred;
                if (i + 3 < n) pred = ip_quad_1<T>(b[e], c[e], d[e]);
                else if (i + 1 < n) pred = ip_linear<T>(b[e], c[e]);
                else pred = b[e];
                finish(e, pred);
            }
        }
    } else {
        // window win[k] = row[x0 - 4 + k], k = 0..15 (outside the row: never used by the case rules below)
        T win[16];
        T lo4[4], hi4[4];
        const T *row = w + idx;  // row + x0
        if (x0 >= 8) ld4<T>(row - 4, lo4);
        if (x0 + 8 < dx) ld4<T>(row + 8, hi4);  // (dx % 4 == 0: these four lie inside the row)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            win[k] = x0 >= 8 ? lo4[k] : (T)0;
            win[12 + k] = x0 + 8 < dx ? hi4[k] : (T)0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) win[4 + k] = o[k];
        // the 8 elements lie in one block of 32 (x0 is a multiple of 8): its bounds once, in 32 bits (dx < 2^31 here)
        const uint32_t xb = (uint32_t)x0 & ~31u;
        uint32_t xe = xb + 32;
        if (xe > (uint32_t)dx - 1) xe = (uint32_t)dx - 1;
        const uint32_t n = xe - xb + 1, i0 = (uint32_t)x0 - xb;
#pragma unroll
        for (int e = 1; e < 8; e += 2) {
            if (!full && e >= 4) continue;
            const uint32_t i = i0 + e;
            const T m3 = win[4 + e - 3], m1 = win[4 + e - 1], p1 = win[4 + e + 1], p3 = win[4 + e + 3];
            T pred;
            if (OLD) {
                pred = old_rule<T>(i, n, win[4 + e - 5], m3, m1, p1, p3);
            } else if (i >= 3) {
                if (i + 3 < n) pred = ip_cubic<T>(m3, m1, p1, p3);
                else if (i + 1 < n) pred = ip_quad_2<T>(m3, m1, p1);
                else pred = ip_linear1<T>(m3, m1);
            } else {
                if (i + 3 < n) pred = ip_quad_1<T>(m1, p1, p3);
                else if (i + 1 < n) pred = ip_linear<T>(m1, p1);
                else pred = m1;
            }
            finish(e, pred);
        }
    }
    if (valid) {
        if (DEC || !p.no_store) st8m<T>(w + idx, o, full);
        if (!DEC) {
            *reinterpret_cast<uint2 *>(codes + idx) = make_uint2(cw[0], cw[1]);
            if (full) *reinterpret_cast<uint2 *>(codes + idx + 4) = make_uint2(cw[2], cw[3]);
        }
    }
}

// ---- one LEVEL per launch (N == 3): the three directional passes of a tile of 32^3 grid intervals, in LDS ----------------
// The per-pass kernels above move every level's rows through HBM three times (once per direction). Here a workgroup owns one
// block of the reference's decomposition — 32 grid intervals of the level's stride in each dimension, 33^3 grid points with its
// faces, exactly the range [begin, end] inside which the reference cuts its stencils (InterpolationDecomposition.hpp:121-135) —
// and runs the three passes on it in LDS. What a pass reads is either coarser (final when the launch starts) or an earlier
// pass's point of the same block range along the pass axis: so the closed block is self-contained. The points on a face belong
// to two blocks (four on an edge, eight at a corner); all of them compute it — same inputs, same arithmetic, same bits — and
// one writes it: the block in which every coordinate of the point lies below the block's end (or the block is the axis' last).
//   LDS: the points with an even coordinate along the LAST pass's axis (the odd ones are that pass's own points: nobody reads
//        them) — 33 x 33 x 17 values, 74 KB for f32 (two workgroups per CU), 148 KB for f64.
//   in:  the originals are read from the caller's array (never written: a neighbour block may already have replaced a face
//        point's original in the work array by its reconstruction), coarser points from the work array w; nothing else of w is
//        read, so the working copy of the input the per-pass path starts from is not needed at all.
struct szk_interp_level {
    uint32_t off[2];   // element strides of the two slower dimensions (the fastest has stride 1)
    uint32_t s;        // level stride
    uint32_t g[3];     // grid points per dimension at this level: (D - 1) / s + 1
    uint32_t nt[3];    // blocks per dimension
    int perm[3];       // pass k runs along dimension perm[k]
    int interp_id, radius, no_store;
    int pair;          // decoder, finest level, x last, rows of even length: the last pass stores (even, odd) pairs, the earlier ones nothing
    double eb, eb_recip;
    // Compression, round 5: the hand-over between two levels through a DENSE array of the coarser level's grid. In place, the level of
    // stride 2 stores its reconstruction 4 bytes here, 4 bytes not — partial lines that the memory system reads, merges and writes back
    // (54 of that launch's 144 us at C3) — and the finest level gathers its coarse points with a stride of two elements.
    const void *coarse;  // this level's coarse points (the grid of stride 2 s) as a dense array with the strides coff[]; nullptr: from w
    void *dense;         // this level's reconstruction and the coarse points it loaded go to a dense array of ITS grid, strides doff[]; nullptr: into w
    uint32_t coff[2], doff[2];
};
#define LV_MAIN (33 * 33 * 17)
#define LV_SIDE (33 * 33)
#define LV_NT 768
struct LvMagic {
    uint32_t m[40];
};
__device__ __forceinline__ uint32_t lv_div(uint32_t t, uint32_t d, uint32_t magic) { return d <= 1 ? t : __umulhi(t, magic); }
static LvMagic lv_magic_host() {
    LvMagic g;
    g.m[0] = g.m[1] = 0;
    for (uint32_t d = 2; d < 40; d++) g.m[d] = 0xFFFFFFFFu / d + 1;  // exact quotients for t * d < 2^32
    return g;
}
__device__ __forceinline__ uint32_t sel3(int i, uint32_t a, uint32_t b, uint32_t c) { return i == 0 ? a : (i == 1 ? b : c); }

// the stencil rules of interpolation_1d_fastest_dim_first (:334-399) on gathered neighbours (unused ones hold anything);
// linear mode's extrapolated last point of an even-length line (reads the previous point of its own pass) is the caller's
template <typename T>
__device__ __forceinline__ T lv_rule(uint32_t i, uint32_t n, T m3, T m1, T p1, T p3, int interp_id) {
    if (interp_id == 0) return i + 1 < n ? ip_linear<T>(m1, p1) : m1;
    if (i >= 3) {
        if (i + 3 < n) return ip_cubic<T>(m3, m1, p1, p3);
        if (i + 1 < n) return ip_quad_2<T>(m3, m1, p1);
        return ip_linear1<T>(m3, m1);
    }
    if (i + 3 < n) return ip_quad_1<T>(m1, p1, p3);
    if (i + 1 < n) return ip_linear<T>(m1, p1);
    return m1;
}

// cubic mode without per-point branches: the three stencils a full block meets are computed and selected; the others (ragged
// blocks' line ends) take the branch
// FULL: the line has 33 points (every block but an axis' last one): the three stencils are all there is
template <typename T, bool FULL>
__device__ __forceinline__ T lv_cubic_sel(uint32_t i, uint32_t n, T m3, T m1, T p1, T p3) {
    const T c = ip_cubic<T>(m3, m1, p1, p3), q1 = ip_quad_1<T>(m1, p1, p3), q2 = ip_quad_2<T>(m3, m1, p1);
    const bool lo = i < 3, hi3 = i + 3 >= n;
    T r = lo ? q1 : c;
    r = (!lo && hi3) ? q2 : r;
    if (!FULL && (i + 1 >= n || (lo && hi3))) r = lv_rule<T>(i, n, m3, m1, p1, p3, 1);
    return r;
}
// ref_quantize with selects instead of its early returns (same arithmetic, same results)
template <typename T>
__device__ __forceinline__ int lv_quantize(T &data, T pred, double eb, double recip, int radius) {
    const T diff = data - pred;
    const double scaled = fabs((double)diff) * recip;
    const bool inr = scaled < (double)(2 * radius - 1);  // (false for NaN)
    int qi = (int)(inr ? scaled : 0.0) + 1;
    const int half = qi >> 1;
    qi = half << 1;
    const bool neg = diff < 0;
    const int sq = neg ? -qi : qi;
    const int shifted = neg ? radius - half : radius + half;
    const T dec = (T)((double)pred + (double)sq * eb);
    const T ad = dec - data;
    const bool ok = inr && fabs((double)ad) <= eb;
    data = ok ? dec : data;
    return ok ? shifted : 0;
}

// what a pass's item loop needs (all wave-uniform)
template <typename T>
struct LvPass {
    const T *inb;       // originals, at the block's origin
    T *wb;              // work array, at the block's origin
    uint16_t *cb;       // codes, at the block's origin
    T *L, *Ls;
    uint32_t na, nW, c2, cU, cW, seglen, items, mg_c2, mg_cU;
    uint32_t st2, sp2, stU, spU, stW, spW;
    uint32_t strW, strU, dW;
    uint32_t hX, hU, hW;   // 1: the LDS coordinate along that axis is half the index
    int o_m3, o_m1, o_p1, o_p3;
    uint32_t gX, gU, gW, gstep;  // element strides inside the block (32 bits: the host checked the block's span)
    uint32_t ownU_lim, ownX_lim, ownW_lim;  // an index below the limit is owned (limit = n - 1, or n in the axis' last block)
    int defer, no_store, radius, pair;
    uint32_t n2;
    double eb, eb_recip;
    T *dn;                      // compression: the level's reconstruction goes here (a dense array of the level's grid, at the block's origin) instead of wb
    uint32_t eX, eU, eW, estep;  // its element strides inside the block
};

// SLIDE: the walk runs along the pass axis (sliding stencil); CUBIC: interp_id 1; LASTP: last pass of the level (its points
// are not kept in LDS, only owned ones are computed)
template <typename T, bool DEC, bool SLIDE, bool CUBIC, bool LASTP, bool FULL = false>
__device__ __forceinline__ void lv_items(const LvPass<T> &q, uint32_t tid, int sub) {
    T *__restrict__ L = q.L;
    for (uint32_t it = tid; it < q.items; it += LV_NT) {
        const uint32_t q2 = lv_div(it, q.c2, q.mg_c2), xi = it - q2 * q.c2;
        const uint32_t sg = lv_div(q2, q.cU, q.mg_cU), ui = q2 - sg * q.cU;
        const uint32_t iX = q.st2 + xi * q.sp2, iU = q.stU + ui * q.spU;
        const uint32_t lw0 = sg * q.seglen;
        const uint32_t cnt = q.cW - lw0 < q.seglen ? q.cW - lw0 : q.seglen;
        uint32_t iW = q.stW + lw0 * q.spW;
        if (!SLIDE && !CUBIC && (q.defer && iX + 1 == q.na) != (sub != 0)) continue;  // (pass along x: a row's deferred point is an item's)
        int addr = (int)((iW >> q.hW) * q.strW + (iU >> q.hU) * q.strU + (iX >> q.hX));  // the point (passes 0, 1) or its lower neighbour
        uint32_t go = iW * q.gW + iU * q.gU + iX * q.gX;
        uint32_t gd = iW * q.eW + iU * q.eU + iX * q.eX;  // (the same point in the dense array, when there is one)
        const bool own_item = LASTP || (iU < q.ownU_lim && iX < q.ownX_lim);
        T wm3 = (T)0, wm1 = (T)0, wp1 = (T)0, prev = (T)0;
        if (SLIDE) {  // (addresses outside the line fall back to a valid one: the rules do not use what they return)
            wm3 = L[addr + (iW >= 3 ? q.o_m3 : q.o_m1)];
            wm1 = L[addr + q.o_m1];
            wp1 = L[addr + (iW + 1 < q.na ? q.o_p1 : q.o_m1)];
        }
        // pass along x: the item's position along the line is fixed, and so is what may be read
        const int e_m3 = iX >= 3 ? q.o_m3 : q.o_m1, e_p1 = iX + 1 < q.na ? q.o_p1 : q.o_m1, e_p3 = iX + 3 < q.na ? q.o_p3 : q.o_m1;
        // originals (codes) of the walk, four points ahead; beyond the walk's end the last point is read again
        T ov[4];
        int cv[4];
        const uint32_t go_last = go + (cnt - 1) * q.gstep;
        uint32_t gf = go;
        auto fetch = [&]() {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t ga = gf < go_last ? gf : go_last;
                if (DEC) cv[u] = q.cb[ga];
                else ov[u] = q.inb[ga];
                gf += q.gstep;
            }
        };
        auto point = [&](T orig, int code_in) {
            T pred;
            if (SLIDE) {
                const T wp3 = L[addr + (iW + 3 < q.na ? q.o_p3 : q.o_m1)];
                if (CUBIC) pred = lv_cubic_sel<T, FULL>(iW, FULL ? 33u : q.na, wm3, wm1, wp1, wp3);
                else if (q.defer && iW + 1 == q.na) pred = ip_linear1<T>(prev, wm1);
                else pred = lv_rule<T>(iW, q.na, wm3, wm1, wp1, wp3, 0);
                wm3 = wm1;
                wm1 = wp1;
                wp1 = wp3;
            } else if (!CUBIC && sub) {
                pred = ip_linear1<T>(LASTP ? q.Ls[iW * 33u + iU] : L[addr - 2], L[addr + q.o_m1]);
            } else {
                const T xm3 = L[addr + e_m3], xm1 = L[addr + q.o_m1], xp1 = L[addr + e_p1], xp3 = L[addr + e_p3];
                if (CUBIC) pred = lv_cubic_sel<T, FULL>(iX, FULL ? 33u : q.na, xm3, xm1, xp1, xp3);
                else pred = lv_rule<T>(iX, q.na, xm3, xm1, xp1, xp3, 0);
            }
            // (SLIDE: the walk's indices are odd, the last one of an axis lies in its last block: owned whenever the item is)
            const bool owned = own_item && (SLIDE || LASTP || iW < q.ownW_lim);
            T v;
            if (DEC) {
                v = code_in ? ref_recover<T>(pred, code_in, q.eb, q.radius) : q.wb[go];  // code 0: the raw value, scattered in place before
                if (sizeof(T) == 4 && q.pair) {
                    // the finest level's even-x points are stored by the last pass together with their odd neighbours (whole
                    // 8-byte pairs, rows written once) instead of 4 bytes here and 4 there; what has no odd neighbour — the last
                    // column of a row of odd length — is stored where it is decoded
                    if (LASTP) {
                        const float2 pr = make_float2((float)L[addr + q.o_m1], (float)v);
                        *reinterpret_cast<float2 *>(reinterpret_cast<float *>(q.wb) + go - 1) = pr;
                    } else if (owned && iX + 1 == q.n2) {
                        q.wb[go] = v;
                    }
                } else if (q.dn) {
                    // (decoder, the level of stride 2 under a finest level that stores whole pairs: that level writes these positions
                    // again from the dense array — nothing goes to the output here, 4 bytes in 16 of its lines)
                    if (owned) q.dn[gd] = v;
                } else if (owned && code_in) {
                    q.wb[go] = v;
                }
            } else {
                v = orig;
                const int code = lv_quantize<T>(v, pred, q.eb, q.eb_recip, q.radius);
                if (owned) {
                    q.cb[go] = (uint16_t)code;
                    if (!q.no_store) {
                        if (q.dn) q.dn[gd] = v;
                        else q.wb[go] = v;
                    }
                }
            }
            if (!LASTP) L[addr] = v;
            else if (!SLIDE && !CUBIC && q.defer && iX + 3 == q.na) q.Ls[iW * 33u + iU] = v;
            prev = v;
            iW += q.spW;
            addr += (int)q.dW;
            go += q.gstep;
            gd += q.estep;
        };
        fetch();
        uint32_t c = 0;
        for (; c + 4 <= cnt; c += 4) {
            T cur[4];
            int ccur[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                cur[u] = ov[u];
                ccur[u] = cv[u];
            }
            fetch();
#pragma unroll
            for (int u = 0; u < 4; u++) point(cur[u], ccur[u]);
        }
        if (c < cnt) {  // (the last fetch holds the tail's points)
            point(ov[0], cv[0]);
            if (c + 1 < cnt) point(ov[1], cv[1]);
            if (c + 2 < cnt) point(ov[2], cv[2]);
        }
    }
}

// Work of a pass: ITEMS = (index along x) x (index along the other non-walk axis) x (segment of the walk); a thread takes an
// item and walks along z or y — never along x, so that the lanes of a wave lie along x (coalesced global accesses, LDS rows
// without bank conflicts), all per-item index arithmetic is paid once per walk, and a pass along the walk axis slides its
// four-point stencil (one LDS read per point instead of four). The walk's originals (codes, when decoding) are requested four
// points ahead of their use.
template <typename T, bool DEC>
__global__ __launch_bounds__(LV_NT) void k_interp_level(const T *__restrict__ in, T *w, uint16_t *codes, szk_interp_level p, LvMagic mg) {
    extern __shared__ __align__(16) unsigned char lv_smem[];
    T *L = reinterpret_cast<T *>(lv_smem);
    T *Ls = L + LV_MAIN;  // linear mode, last pass along x: the points the extrapolated last point of a row reads
    const uint32_t tid = threadIdx.x;
    const uint32_t b = xcd_block();
    const uint32_t tq = b / p.nt[2];
    const uint32_t t2 = b - tq * p.nt[2];
    const uint32_t t0 = tq / p.nt[1], t1 = tq - t0 * p.nt[1];
    const int aL = p.perm[2];
    // block geometry (all wave-uniform)
    const uint32_t r0 = p.g[0] - 1 - t0 * 32, r1 = p.g[1] - 1 - t1 * 32, r2 = p.g[2] - 1 - t2 * 32;
    const uint32_t n0 = (r0 < 32 ? r0 : 32) + 1, n1 = (r1 < 32 ? r1 : 32) + 1, n2 = (r2 < 32 ? r2 : 32) + 1;
    const bool last0 = t0 == p.nt[0] - 1, last1 = t1 == p.nt[1] - 1, last2 = t2 == p.nt[2] - 1;
    const uint32_t m1 = aL == 1 ? (n1 + 1) >> 1 : n1, m2 = aL == 2 ? (n2 + 1) >> 1 : n2;
    const uint32_t str0 = m1 * m2, str1 = m2;  // LDS strides (the fastest dimension has 1)
    const uint64_t gs0 = (uint64_t)p.s * p.off[0], gs1 = (uint64_t)p.s * p.off[1];
    const uint64_t gbase = (uint64_t)(t0 * 32) * gs0 + (uint64_t)(t1 * 32) * gs1 + (uint64_t)(t2 * 32) * p.s;
    // ---- load: the coarse points (every coordinate even) from w ----
    {
        const uint32_t e0 = (n0 + 1) >> 1, e1 = (n1 + 1) >> 1, e2 = (n2 + 1) >> 1;
        const uint32_t total = e0 * e1 * e2;
        const uint32_t k0 = (aL == 0 ? 1u : 2u) * str0, k1 = (aL == 1 ? 1u : 2u) * str1, k2 = aL == 2 ? 1u : 2u;
        const uint32_t mg_e2 = mg.m[e2], mg_e1 = mg.m[e1];
        for (uint32_t base = tid; base < total; base += 4 * LV_NT) {
            T v[4];
            int ad[4];
            bool own[4] = {false, false, false, false};
            uint64_t da[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t t = base + u * LV_NT;
                ad[u] = -1;
                if (t < total) {
                    const uint32_t q2 = lv_div(t, e2, mg_e2), l2 = t - q2 * e2;
                    const uint32_t l0 = lv_div(q2, e1, mg_e1), l1 = q2 - l0 * e1;
                    if (p.coarse)  // (the dense array of the coarser grid: this block's origin there is half its origin here)
                        v[u] = reinterpret_cast<const T *>(p.coarse)[(uint64_t)(t0 * 16 + l0) * p.coff[0] + (uint64_t)(t1 * 16 + l1) * p.coff[1] + (t2 * 16 + l2)];
                    else v[u] = w[gbase + (uint64_t)(2 * l0) * gs0 + (uint64_t)(2 * l1) * gs1 + (uint64_t)(2 * l2 * p.s)];
                    ad[u] = (int)(l0 * k0 + l1 * k1 + l2 * k2);
                    own[u] = (2 * l0 < 32 || last0) && (2 * l1 < 32 || last1) && (2 * l2 < 32 || last2);
                    da[u] = (uint64_t)(t0 * 32 + 2 * l0) * p.doff[0] + (uint64_t)(t1 * 32 + 2 * l1) * p.doff[1] + (t2 * 32 + 2 * l2);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (ad[u] >= 0) {
                    L[ad[u]] = v[u];
                    // the coarse points belong to this level's grid too: whoever reads the dense array finds them there (written by their owners)
                    if (p.dense && own[u]) reinterpret_cast<T *>(p.dense)[da[u]] = v[u];
                }
        }
    }
    for (int k = 0; k < 3; k++) {
        __syncthreads();
        const int a = p.perm[k];
        const uint32_t na = sel3(a, n0, n1, n2);
        if (na < 2) continue;
        const bool defer = p.interp_id == 0 && na >= 3 && !(na & 1u);
        // point set of the pass: odd along a, every index along the axes of earlier passes, even ones along later axes;
        // the last pass computes owned points only (nobody reads the others)
        uint32_t st0, st1, st2, sp0, sp1, sp2, c0, c1, c2;
        auto axis = [&](int j, uint32_t n, bool last, uint32_t &st, uint32_t &sp, uint32_t &c) {
            if (j == a) {
                st = 1; sp = 2; c = n >> 1;
            } else if (k == 2) {
                st = 0; sp = 1; c = last ? n : n - 1;
            } else if (k == 1 && j == p.perm[0]) {
                st = 0; sp = 1; c = n;
            } else {
                st = 0; sp = 2; c = (n + 1) >> 1;
            }
        };
        axis(0, n0, last0, st0, sp0, c0);
        axis(1, n1, last1, st1, sp1, c1);
        axis(2, n2, last2, st2, sp2, c2);
        const int W = a == 2 ? 0 : a, U = 1 - W;   // walk axis, the other slow axis
        const bool slide = W == a;                  // the walk runs along the pass axis: sliding stencil
        const uint32_t nW = W == 0 ? n0 : n1, nU = U == 0 ? n0 : n1;
        const bool lastW = W == 0 ? last0 : last1, lastU = U == 0 ? last0 : last1;
        const uint32_t stW = W == 0 ? st0 : st1, spW = W == 0 ? sp0 : sp1, cW = W == 0 ? c0 : c1;
        const uint32_t stU = U == 0 ? st0 : st1, spU = U == 0 ? sp0 : sp1, cU = U == 0 ? c0 : c1;
        const uint32_t strW = W == 0 ? str0 : str1, strU = U == 0 ? str0 : str1;
        const uint64_t gsW = W == 0 ? gs0 : gs1, gsU = U == 0 ? gs0 : gs1;
        // LDS: coordinates along the last pass's axis are halved
        const uint32_t dW = (aL == W ? spW >> 1 : spW) * strW;  // address step of one walk step (spW = 2 whenever aL == W)
        const int sa = (int)sel3(a, str0, str1, 1u);
        const int o_m3 = k < 2 ? -3 * sa : -sa, o_m1 = k < 2 ? -sa : 0, o_p1 = sa, o_p3 = k < 2 ? 3 * sa : 2 * sa;
        const uint64_t gstep = (uint64_t)spW * gsW;
        // segment length of the walk: what keeps the workgroup's threads busiest (a deferred point needs its predecessor in
        // the same walk)
        uint32_t seglen = 32;
        if (!(defer && slide)) {
            uint32_t best = 0xFFFFFFFFu;
            for (uint32_t sl = 32; sl >= 4; sl >>= 1) {
                const uint32_t items = c2 * cU * ((cW + sl - 1) / sl);
                const uint32_t cost = ((items + LV_NT - 1) / LV_NT) * (sl + 2);
                if (cost < best) {
                    best = cost;
                    seglen = sl;
                }
            }
        }
        const uint32_t nseg = (cW + seglen - 1) / seglen;
        LvPass<T> q;
        q.inb = in + gbase;
        q.wb = w + gbase;
        q.cb = codes + gbase;
        q.L = L;
        q.Ls = Ls;
        q.na = na;
        q.nW = nW;
        q.c2 = c2;
        q.cU = cU;
        q.cW = cW;
        q.seglen = seglen;
        q.items = c2 * cU * nseg;
        q.mg_c2 = mg.m[c2];
        q.mg_cU = mg.m[cU];
        q.st2 = st2; q.sp2 = sp2; q.stU = stU; q.spU = spU; q.stW = stW; q.spW = spW;
        q.strW = strW; q.strU = strU; q.dW = dW;
        q.hX = aL == 2; q.hU = aL == U; q.hW = aL == W;
        q.o_m3 = o_m3; q.o_m1 = o_m1; q.o_p1 = o_p1; q.o_p3 = o_p3;
        q.gX = p.s; q.gU = (uint32_t)gsU; q.gW = (uint32_t)gsW; q.gstep = (uint32_t)gstep;
        q.ownU_lim = lastU ? nU : nU - 1;
        q.ownX_lim = last2 ? n2 : n2 - 1;
        q.ownW_lim = lastW ? nW : nW - 1;
        q.defer = defer; q.no_store = p.no_store; q.radius = p.radius; q.pair = p.pair; q.n2 = n2;
        q.eb = p.eb; q.eb_recip = p.eb_recip;
        q.dn = nullptr; q.eX = q.eU = q.eW = q.estep = 0;
        if (p.dense) {  // the block's origin in the dense array of this level's grid, the strides of the walk there (x: 1)
            q.dn = reinterpret_cast<T *>(p.dense) + ((uint64_t)(t0 * 32) * p.doff[0] + (uint64_t)(t1 * 32) * p.doff[1] + t2 * 32);
            q.eX = 1;
            q.eU = U == 0 ? p.doff[0] : p.doff[1];
            q.eW = W == 0 ? p.doff[0] : p.doff[1];
            q.estep = spW * q.eW;
        }
        const bool cubic = p.interp_id != 0;
        const bool full = cubic && na == 33;
        if (slide) {
            if (k == 2) {
                if (full) lv_items<T, DEC, true, true, true, true>(q, tid, 0);
                else if (cubic) lv_items<T, DEC, true, true, true>(q, tid, 0);
                else lv_items<T, DEC, true, false, true>(q, tid, 0);
            } else {
                if (full) lv_items<T, DEC, true, true, false, true>(q, tid, 0);
                else if (cubic) lv_items<T, DEC, true, true, false>(q, tid, 0);
                else lv_items<T, DEC, true, false, false>(q, tid, 0);
            }
        } else {
            for (int sub = 0; sub < (defer ? 2 : 1); sub++) {
                if (sub) __syncthreads();
                if (k == 2) {
                    if (full) lv_items<T, DEC, false, true, true, true>(q, tid, sub);
                    else if (cubic) lv_items<T, DEC, false, true, true>(q, tid, sub);
                    else lv_items<T, DEC, false, false, true>(q, tid, sub);
                } else {
                    if (full) lv_items<T, DEC, false, true, false, true>(q, tid, sub);
                    else if (cubic) lv_items<T, DEC, false, true, false>(q, tid, sub);
                    else lv_items<T, DEC, false, false, false>(q, tid, sub);
                }
            }
        }
    }
}

// anchor grid (build_anchor_grid :215-221): every anchor_stride-th point in each dimension is stored losslessly;
// without anchors (anchor_stride == 0) the first element is quantised against 0 (:92-93)
template <typename T, typename IT = uint64_t, bool SINK = false>
__device__ __forceinline__ void anchor_point(T *__restrict__ w, uint16_t *__restrict__ codes, const szk_interp_pass &p, uint64_t t,
                                             uint64_t boff, const TrialSink *sink = nullptr, const T *orig = nullptr) {
    IT r = (IT)t;
    uint64_t idx = 0;
#pragma unroll
    for (int j = 3; j >= 0; j--) {
        if (j >= p.N) continue;
        const IT cj = (IT)p.cnt[j];
        const IT q = r % cj;
        r /= cj;
        idx += (p.start[j] + (uint64_t)q * p.step[j]) * p.off[j];
    }
    T v = orig ? orig[idx] : w[idx];
    int code = 0;
    if (p.subpass) {  // "no anchor" mode: one point, predicted by 0
        code = ref_quantize<T>(v, (T)0, p.eb, p.eb_recip, p.radius);
        if (code) w[idx] = v;
    }
    if (orig && !code) w[idx] = v;  // (level kernels: the work array holds nothing but what the launches put there)
    if (SINK) {
        sink_code(sink, (uint32_t)code);
        if (codes) codes[idx] = (uint16_t)code;
    } else {
        codes[idx] = (uint16_t)code;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_interp_anchors(T *__restrict__ w, uint16_t *__restrict__ codes, szk_interp_pass p, const T *orig) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.total) return;
    const uint64_t boff = (uint64_t)blockIdx.y * p.batch_stride;
    anchor_point<T>(w + boff, codes + boff, p, t, boff, nullptr, orig);
}
template <typename T>
__global__ __launch_bounds__(256) void k_interp_first_dec(T *__restrict__ w, const uint16_t *__restrict__ codes, double eb, int radius) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && codes[0]) w[0] = ref_recover<T>((T)0, codes[0], eb, radius);
}

// histogram of u16 codes: persistent workgroups, LDS window [bin][4 copies] around the radius, flushed with one
// 64-bit atomic per non-empty bin and workgroup. The same pass builds the list of unpredictable values (code 0): their
// indices collect in a per-wave LDS queue and go to the global list in batches (one global atomic per batch), the values are
// gathered from the work array, where an unpredictable point keeps its raw value.
#define IHW_WIN 8192
#define IH_OQ 128  // indices per wave in the staging queue
// BIGW: second tier of 16384 instead of 8192 bins (84 KB of LDS: one workgroup per CU) for alphabets that spread wider
// (every code outside the tier is a global atomic); *far_cnt receives the number of codes outside +-4096 in either form, from
// which the host picks the form of the context's next call
template <typename T, bool SMALLR, bool BIGW>  // SMALLR: radius <= IH_WIN / 2, code 0 would fall inside the window
__global__ __launch_bounds__(BIGW ? 1024 : 512) void k_hist_codes(const uint16_t *__restrict__ codes, uint64_t n, int radius,
                                                    uint64_t *__restrict__ hist, const T *__restrict__ work,
                                                    uint64_t *__restrict__ n_vout, uint64_t *__restrict__ vout_idx,
                                                    T *__restrict__ vout_val, uint64_t out_cap, uint32_t *__restrict__ far_cnt,
                                                    uint32_t drop_far) {
    // drop_far (BIGW only): codes beyond the second tier are not counted here (every one a global atomic, ~1.2 G/s for the
    // whole chip): k_hist_tail counts them, window by window; far_cnt[1] receives their number either way
    constexpr uint32_t WWIN = BIGW ? 2 * IHW_WIN : IHW_WIN;
    __shared__ uint32_t lh[IH_WIN * 4];
    __shared__ uint32_t l_zero[4];  // code 0 (unpredictable): far from the window and ONE address for all of them
    __shared__ uint32_t lw[WWIN];  // second tier, one copy: the tails (tight bounds spread the codes over thousands of bins)
    __shared__ uint32_t s_far, s_far2;
    uint32_t my_far = 0;  // codes beyond the plain tier (global atomics)
    uint32_t my_far2 = 0; // codes beyond this form's second tier
    constexpr uint32_t NT = BIGW ? 1024 : 512;  // the large tier leaves room for one workgroup per CU (a big one), the plain one for two
    __shared__ uint64_t s_oq[NT / 64][IH_OQ];
    for (int i = threadIdx.x; i < IH_WIN * 4; i += NT) lh[i] = 0;
    for (uint32_t i = threadIdx.x; i < WWIN; i += NT) lw[i] = 0;
    if (threadIdx.x < 4) l_zero[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_far = s_far2 = 0;
    __syncthreads();
    const uint32_t win_lo = (uint32_t)(radius - IH_WIN / 2), wide_lo = (uint32_t)radius - WWIN / 2, copy = threadIdx.x & 3u;
    const int lane = threadIdx.x & 63;
    uint64_t *oq = s_oq[threadIdx.x >> 6];
    uint32_t oq_n = 0;  // wave-uniform fill level
    auto oq_flush = [&]() {  // (called by whatever lanes are active)
        oq_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)oq_n);
        if (oq_n == 0) return;
        // lane 0 has the smallest index of its wave, so it is active whenever any lane is (and its oq_n is current)
        const unsigned long long act = __ballot(1);
        const uint32_t nact = (uint32_t)__popcll(act), rank = (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd((unsigned long long *)n_vout, (unsigned long long)oq_n);
        const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
        const unsigned long long b0 = ((unsigned long long)bhi << 32) | blo;
        for (uint32_t k = rank; k < oq_n; k += nact) {
            const unsigned long long pos = b0 + k;
            if (pos < out_cap) {
                const uint64_t id = oq[k];
                vout_idx[pos] = id;
                vout_val[pos] = work[id];
            }
        }
        oq_n = 0;
    };
    const uint64_t nth = (uint64_t)gridDim.x * NT;
    // (lanes may leave the loop one iteration apart: the queue level is re-read from the first active lane before use)
    // the next iteration's 8 codes are requested before this iteration's are counted (clamped address, never conditional)
    const uint64_t i_first = ((uint64_t)blockIdx.x * NT + threadIdx.x) * 8;
    const uint64_t last8 = n >= 8 ? n - 8 : 0;
    uint4 nxt = make_uint4(0, 0, 0, 0);
    if (n >= 8) nxt = *reinterpret_cast<const uint4 *>(codes + (i_first < last8 ? i_first : last8));
    for (uint64_t i = i_first; i < n; i += nth * 8) {
        uint16_t c[8];
        const uint4 v = nxt;
        {
            const uint64_t in = i + nth * 8;
            if (n >= 8) nxt = *reinterpret_cast<const uint4 *>(codes + (in < last8 ? in : last8));
        }
        if (i + 8 <= n) {
            const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c[2 * k] = (uint16_t)(wv[k] & 0xFFFF);
                c[2 * k + 1] = (uint16_t)(wv[k] >> 16);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) c[k] = (i + k < n) ? codes[i + k] : (uint16_t)0xFFFF;
        }
        uint32_t zmask = 0;
        // what lies outside the window: code 0, the second tier, the far codes (every one of them a branch of its own)
        auto beyond = [&](int k) {
            if (c[k] == 0) {
                atomicAdd(&l_zero[copy], 1u);
                zmask |= 1u << k;
            } else if ((uint32_t)c[k] - wide_lo < WWIN) atomicAdd(&lw[(uint32_t)c[k] - wide_lo], 1u);
            else {
                if (!(BIGW && drop_far)) atomicAdd((unsigned long long *)&hist[c[k]], 1ull);
                my_far++;
                my_far2++;
            }
        };
        if (i + 8 <= n) {
            // a whole group of eight (every group but the array's last): the window's counts first, with no test but the window's own —
            // per code a 64-bit bound test and three levels of else branches were ~12 vector and as many scalar instructions around
            // one LDS add (k_hist_codes 92 us at C3 for a pass that reads 268 MB) —, the rest behind one wave-uniform test
            uint32_t rare = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t bin = (uint32_t)c[k] - win_lo;
                // (code 0 never counts in the window: with a small quantbinCnt it would lie inside it)
                const bool inw = bin < IH_WIN && (!SMALLR || c[k] != 0);
                if (inw) atomicAdd(&lh[bin * 4 + copy], 1u);
                rare |= inw ? 0u : 1u << k;
            }
            if (__ballot(rare != 0)) {
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if ((rare >> k) & 1u) beyond(k);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (i + k >= n) break;
                const uint32_t bin = (uint32_t)c[k] - win_lo;
                if (bin < IH_WIN && (!SMALLR || c[k] != 0)) atomicAdd(&lh[bin * 4 + copy], 1u);
                else beyond(k);
            }
        }
        if (__ballot(zmask != 0)) {  // some lane met unpredictable points: queue their indices
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool z = (zmask >> k) & 1u;
                const unsigned long long m = __ballot(z);
                if (m) {
                    oq_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)oq_n);
                    if (oq_n + 64 > IH_OQ) oq_flush();
                    if (z) oq[oq_n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i + k;
                    oq_n += (uint32_t)__popcll(m);
                }
            }
        }
    }
    oq_flush();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t z = l_zero[0] + l_zero[1] + l_zero[2] + l_zero[3];
        if (z) atomicAdd((unsigned long long *)&hist[0], (unsigned long long)z);
    }
    for (uint32_t b = threadIdx.x; b < WWIN; b += NT) {
        const uint32_t v = lw[b];
        const uint32_t sym = wide_lo + b;
        if (v && sym < SZH_HIST_BINS) atomicAdd((unsigned long long *)&hist[sym], (unsigned long long)v);
        if (BIGW && (b < IHW_WIN / 2 || b >= WWIN - IHW_WIN / 2)) my_far += v;  // what the plain tier would have missed
    }
    my_far = wave_sum32(my_far);
    my_far2 = wave_sum32(my_far2);
    if ((threadIdx.x & 63) == 0 && my_far) atomicAdd(&s_far, my_far);
    if ((threadIdx.x & 63) == 0 && my_far2) atomicAdd(&s_far2, my_far2);
    __syncthreads();
    if (threadIdx.x == 0 && s_far) atomicAdd(far_cnt, s_far);
    if (threadIdx.x == 0 && BIGW && s_far2) atomicAdd(far_cnt + 1, s_far2);
    for (int b = threadIdx.x; b < IH_WIN; b += NT) {
        const uint32_t s = lh[b * 4] + lh[b * 4 + 1] + lh[b * 4 + 2] + lh[b * 4 + 3];
        const int sym = (int)win_lo + b;
        if (s && sym >= 0 && sym < (int)SZH_HIST_BINS) atomicAdd((unsigned long long *)&hist[sym], (unsigned long long)s);
    }
}

// the codes beyond k_hist_codes' 16384-bin tier, counted in LDS: blockIdx.y picks one of the three 16384-bin windows that
// cover the rest of the 65536 symbols (cyclically, starting at radius + 8192); every workgroup reads all codes and counts
// the ones of its window. Code 0 (unpredictable) is k_hist_codes' business.
#define IHT_WIN 16384
__global__ __launch_bounds__(256) void k_hist_tail(const uint16_t *__restrict__ codes, uint64_t n, int radius, uint64_t *__restrict__ hist) {
    __shared__ uint32_t lw[IHT_WIN];
    for (uint32_t i = threadIdx.x; i < IHT_WIN; i += 256) lw[i] = 0;
    __syncthreads();
    const uint32_t lo = ((uint32_t)radius + IHT_WIN / 2 + blockIdx.y * IHT_WIN) & 0xFFFFu;
    const uint64_t nth = (uint64_t)gridDim.x * 256;
    const uint64_t i_first = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    const uint64_t last8 = n >= 8 ? n - 8 : 0;
    uint4 nxt = make_uint4(0, 0, 0, 0);
    if (n >= 8) nxt = *reinterpret_cast<const uint4 *>(codes + (i_first < last8 ? i_first : last8));
    for (uint64_t i = i_first; i < n; i += nth * 8) {
        const uint4 v = nxt;
        {
            const uint64_t in = i + nth * 8;
            if (n >= 8) nxt = *reinterpret_cast<const uint4 *>(codes + (in < last8 ? in : last8));
        }
        uint32_t c[8];
        if (i + 8 <= n) {
            const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c[2 * k] = wv[k] & 0xFFFFu;
                c[2 * k + 1] = wv[k] >> 16;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) c[k] = (i + k < n) ? codes[i + k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t b = (c[k] - lo) & 0xFFFFu;
            if (b < IHT_WIN && c[k] != 0) atomicAdd(&lw[b], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < IHT_WIN; b += 256) {
        const uint32_t v = lw[b];
        if (v) atomicAdd((unsigned long long *)&hist[(lo + b) & 0xFFFFu], (unsigned long long)v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_scatter_raw(const uint8_t *__restrict__ payload, uint64_t idx_off, uint64_t val_off,
                                                     uint64_t cnt, uint64_t n, T *__restrict__ out) {
    const uint64_t *idx = reinterpret_cast<const uint64_t *>(payload + idx_off);
    const T *val = reinterpret_cast<const T *>(payload + val_off);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (uint64_t)gridDim.x * 256) {
        const uint64_t k = idx[i];
        if (k < n) out[k] = val[i];
    }
}

// ---- host side: the level / pass schedule (InterpolationDecomposition::init :176-213, compress :79-147) -----------
static void nth_permutation(int N, int id, int *perm) {  // lexicographic order = std::next_permutation sequence
    int p[4] = {0, 1, 2, 3};
    for (int k = 0; k < id; k++) {
        int i = N - 2;
        while (i >= 0 && p[i] > p[i + 1]) i--;
        if (i < 0) break;
        int j = N - 1;
        while (p[j] < p[i]) j--;
        int t = p[i];
        p[i] = p[j];
        p[j] = t;
        for (int a = i + 1, b = N - 1; a < b; a++, b--) {
            t = p[a];
            p[a] = p[b];
            p[b] = t;
        }
    }
    for (int i = 0; i < N; i++) perm[i] = p[i];
}

int szk_interp_novec = 0;  // test hook: force the one-point-per-thread kernels

// the level / pass schedule as a list (kind 0: anchor grid, 1: first point without anchors, 2: directional pass)
static int build_schedule(const szk_interp_params &ip, bool dec, uint32_t nbatch, std::vector<szk_interp_pass> &out) {
    const int N = ip.N;
    szk_interp_pass p;
    memset(&p, 0, sizeof(p));
    p.N = N;
    uint64_t num = 1;
    for (int i = 0; i < N; i++) {
        p.dims[i] = ip.dims[i];
        num *= ip.dims[i];
    }
    p.off[N - 1] = 1;
    for (int i = N - 2; i >= 0; i--) p.off[i] = p.off[i + 1] * p.dims[i + 1];
    p.batch_stride = nbatch > 1 ? num : 0;
    p.radius = ip.radius;
    p.interp_id = ip.interp_id;
    p.old_api = N <= 2;
    p.n_vout = ip.n_vout;
    p.vout_idx = ip.vout_idx;
    p.vout_val = ip.vout_val;
    p.out_cap = ip.out_cap;
    // init(): levels and whether anchors are used
    uint64_t anchor = ip.anchor_stride;
    int interp_level = -1;
    bool use_anchor = false;
    for (int i = 0; i < N; i++) {
        const int lv = (int)ceil(log2((double)p.dims[i]));
        if (interp_level < lv) interp_level = lv;
        if (p.dims[i] > anchor) use_anchor = true;
    }
    if (!use_anchor) anchor = 0;
    if (anchor > 0) {
        const int maxl = (int)log2((double)anchor) + 1;
        if (maxl <= interp_level) interp_level = maxl;
    }
    int perm[4], pos[4];
    nth_permutation(N, ip.direction, perm);
    for (int k = 0; k < N; k++) pos[perm[k]] = k;
    p.eb = ip.eb;
    p.eb_recip = 1.0 / ip.eb;
    if (anchor == 0) {  // first point, predicted by 0
        p.kind = 1;
        p.total = 1;
        for (int j = 0; j < N; j++) {
            p.start[j] = 0;
            p.step[j] = 1;
            p.cnt[j] = 1;
        }
        p.subpass = 1;
        out.push_back(p);
    } else {
        if (!dec) {  // anchors are lossless: the decoder finds them among the scattered raw values
            p.kind = 0;
            p.total = 1;
            for (int j = 0; j < N; j++) {
                p.start[j] = 0;
                p.step[j] = anchor;
                p.cnt[j] = (p.dims[j] - 1) / anchor + 1;
                p.total *= p.cnt[j];
            }
            p.subpass = 0;
            out.push_back(p);
        }
        interp_level--;
    }
    p.kind = 2;
    for (int level = interp_level; level > 0; level--) {
        double cur_eb = ip.eb;  // per-level bound :103-117
        if (ip.alpha < 0) {
            cur_eb = level >= 3 ? ip.eb * 0.5 : ip.eb;
        } else if (ip.alpha >= 1) {
            double r = pow(ip.alpha, level - 1);
            if (r > ip.beta) r = ip.beta;
            cur_eb = ip.eb / r;
        }
        p.eb = cur_eb;
        p.eb_recip = 1.0 / cur_eb;
        p.s = 1ull << (level - 1);
        p.bsz = 32ull * p.s;
        for (int k = 0; k < N; k++) {
            p.dir = perm[k];
            p.total = 1;
            for (int j = 0; j < N; j++) {
                const uint64_t Dj = p.dims[j];
                if (j == p.dir) {
                    p.start[j] = p.s;
                    p.step[j] = 2 * p.s;
                    p.cnt[j] = ((Dj - 1) / p.s + 1) / 2;
                } else if (pos[j] < k) {
                    p.start[j] = 0;
                    p.step[j] = p.s;
                    p.cnt[j] = (Dj - 1) / p.s + 1;
                } else {
                    p.start[j] = 0;
                    p.step[j] = 2 * p.s;
                    p.cnt[j] = (Dj - 1) / (2 * p.s) + 1;
                }
                p.total *= p.cnt[j];
            }
            if (p.total == 0) continue;
            if ((p.total + 255) / 256 > 0x7FFFFFFFull) return -1;
            p.subpass = 0;
            out.push_back(p);
            if (!p.old_api && p.interp_id == 0) {  // the deferred last point of even-length lines (linear, fastest-dim-first rule)
                p.subpass = 1;
                out.push_back(p);
            }
        }
    }
    if (!dec && !out.empty() && out.back().kind == 2) out.back().no_store = 1;  // (a deferred sub-pass, if any, is that last entry)
    return 0;
}

// the level kernels apply to 3-D arrays whose index arithmetic fits their 32-bit fields and whose finest level has enough
// blocks to fill the chip (smaller arrays: the per-pass kernels on a working copy)
#define LV_MIN_BLOCKS 256
int szk_interp_min_blocks = LV_MIN_BLOCKS;  // test hook: 1 sends every level of every 3-D array through the level kernel
int szk_interp_levels_ok(const szk_interp_params *ip) {
    if (ip->N != 3 || szk_interp_novec) return 0;
    uint64_t blocks = 1;
    for (int j = 0; j < 3; j++) {
        if (ip->dims[j] >= (1ull << 26)) return 0;
        blocks *= ip->dims[j] > 1 ? (ip->dims[j] - 1 + 31) / 32 : 1;
    }
    if (ip->dims[1] * ip->dims[2] >= (1ull << 32)) return 0;
    return blocks >= (uint64_t)szk_interp_min_blocks;
}
// a level with few blocks leaves most of the chip idle in the level kernel (one workgroup per block, ~60 us each whatever
// their number): such levels run pass by pass, one thread per point
static uint64_t level_blocks(const szk_interp_pass &p) {
    uint64_t tiles = 1;
    for (int j = 0; j < 3; j++) {
        const uint64_t g = (p.dims[j] - 1) / p.s + 1;
        tiles *= g > 1 ? (g - 1 + 31) / 32 : 1;
    }
    // (the kernel addresses a block's elements with 32-bit offsets from the block's origin)
    if (33 * p.s * (p.off[0] + p.off[1] + 1) >= (1ull << 31)) return 0;
    return tiles;
}
template <typename T, bool DEC>
static int launch_level(const szk_interp_pass &p, const int *perm, const T *in, T *w, uint16_t *codes, hipStream_t s, const T *coarse = nullptr, T *dense = nullptr) {
    static const LvMagic mg = lv_magic_host();
    szk_interp_level L;
    memset(&L, 0, sizeof(L));
    L.off[0] = (uint32_t)p.off[0];
    L.off[1] = (uint32_t)p.off[1];
    L.s = (uint32_t)p.s;
    uint64_t tiles = 1;
    for (int j = 0; j < 3; j++) {
        L.g[j] = (uint32_t)((p.dims[j] - 1) / p.s + 1);
        L.nt[j] = L.g[j] > 1 ? (L.g[j] - 1 + 31) / 32 : 1;
        L.perm[j] = perm[j];
        tiles *= L.nt[j];
    }
    if (tiles > 0x7FFFFFFFull) return -1;
    L.interp_id = p.interp_id;
    L.radius = p.radius;
    L.no_store = !DEC && p.s == 1;  // the finest level's reconstruction is read by nobody
    L.pair = DEC && sizeof(T) == 4 && p.s == 1 && perm[2] == 2 && p.dims[2] % 2 == 0 && (reinterpret_cast<uintptr_t>(w) & 7) == 0;
    L.eb = p.eb;
    L.eb_recip = p.eb_recip;
    if (dense) {  // this level's grid as a dense array (strides of the two slower dimensions)
        L.dense = dense;
        L.doff[0] = L.g[1] * L.g[2];
        L.doff[1] = L.g[2];
    }
    if (coarse) {  // the coarser level's grid, dense: (D - 1) / (2 s) + 1 points per dimension
        L.coarse = coarse;
        const uint32_t c1 = (uint32_t)((p.dims[1] - 1) / (2 * p.s) + 1), c2 = (uint32_t)((p.dims[2] - 1) / (2 * p.s) + 1);
        L.coff[0] = c1 * c2;
        L.coff[1] = c2;
    }
    const size_t lds = (size_t)(LV_MAIN + LV_SIDE) * sizeof(T);
    // (per device, and a context may sit on any of them: asked for at every launch, a host-side table lookup)
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_interp_level<T, DEC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return -2;
    hipLaunchKernelGGL((k_interp_level<T, DEC>), dim3((uint32_t)tiles), dim3(LV_NT), lds, s, in, w, codes, L, mg);
    return 0;
}

// in: the originals (compression with the level kernels: never written; nullptr = w holds a working copy of them)
template <typename T, bool DEC>
static int run_interp(const szk_interp_params &ip, const T *in, T *w, uint16_t *codes, hipStream_t s, uint32_t nbatch = 1) {
    std::vector<szk_interp_pass> sched;
    if (build_schedule(ip, DEC, nbatch, sched)) return -1;
    const bool levels = nbatch == 1 && szk_interp_levels_ok(&ip) && (DEC || in != nullptr);
    int perm[4];
    nth_permutation(ip.N, ip.direction, perm);
    uint64_t level_done = 0;  // stride of the level the last level launch covered
    // compression: the level of stride 2 hands its grid to the finest level as a dense array when both run as level launches and the
    // caller gave room for it (szk_interp_params::dense2)
    bool dense2 = false;
    // (decoder: only under a finest level that stores whole (even, odd) pairs — it then writes the grid of stride 2 itself, from the dense array)
    const bool dec_pairs = sizeof(T) == 4 && perm[2] == 2 && ip.dims[2] % 2 == 0 && (reinterpret_cast<uintptr_t>(w) & 7) == 0;
    if ((!DEC || dec_pairs) && levels && ip.dense2 && ip.N == 3) {
        bool l1 = false, l2 = false;
        for (const szk_interp_pass &p : sched)
            if (p.kind == 2 && level_blocks(p) >= (uint64_t)szk_interp_min_blocks) {
                l1 |= p.s == 1;
                l2 |= p.s == 2;
            }
        const uint64_t need = ((ip.dims[0] - 1) / 2 + 1) * ((ip.dims[1] - 1) / 2 + 1) * ((ip.dims[2] - 1) / 2 + 1);
        dense2 = l1 && l2 && need <= ip.dense2_elems;
    }
    for (const szk_interp_pass &p : sched) {
        const uint32_t nb = (uint32_t)((p.total + 255) / 256);
        if (levels && p.kind == 2 && level_blocks(p) >= (uint64_t)szk_interp_min_blocks) {
            if (p.s == level_done) continue;  // the other passes of a level already launched
            level_done = p.s;
            const int rc = launch_level<T, DEC>(p, perm, in, w, codes, s, dense2 && p.s == 1 ? reinterpret_cast<const T *>(ip.dense2) : (const T *)nullptr,
                                                dense2 && p.s == 2 ? reinterpret_cast<T *>(ip.dense2) : (T *)nullptr);
            if (rc) return rc;
            continue;
        }
        const uint64_t dxl = p.dims[p.N - 1];
        const bool vec = p.kind == 2 && nbatch == 1 && p.interp_id == 1 && p.s == 1 && dxl % 4 == 0 && dxl >= 16 && dxl < (1ull << 31) &&
                         (reinterpret_cast<uintptr_t>(w) & 15) == 0 && (reinterpret_cast<uintptr_t>(codes) & 15) == 0 && !szk_interp_novec;
        if (vec) {
            szk_interp_pass q = p;
            uint64_t rows = 1;
            for (int j = 0; j < p.N - 1; j++) rows *= p.cnt[j];
            q.total = rows * ((dxl + 7) / 8);
            const uint64_t vb = (q.total + 255) / 256;
            if (vb > 0x7FFFFFFFull) return -1;
            const dim3 g((uint32_t)vb), b(256);
            const bool small = q.total <= 0xFFFFFFFFull;  // (then every count of the decomposition fits 32 bits too)
#define SZK_VEC_LAUNCH(XD, OL)                                                                                             \
    do {                                                                                                                   \
        if (small) hipLaunchKernelGGL((k_interp_vec<T, DEC, XD, OL, uint32_t>), g, b, 0, s, w, codes, q);                   \
        else hipLaunchKernelGGL((k_interp_vec<T, DEC, XD, OL, uint64_t>), g, b, 0, s, w, codes, q);                         \
    } while (0)
            if (p.old_api) {
                if (p.dir == p.N - 1) SZK_VEC_LAUNCH(true, true);
                else SZK_VEC_LAUNCH(false, true);
            } else if (p.dir == p.N - 1) SZK_VEC_LAUNCH(true, false);
            else SZK_VEC_LAUNCH(false, false);
#undef SZK_VEC_LAUNCH
        } else if (p.kind == 2) {
            hipLaunchKernelGGL((k_interp_pass<T, DEC>), dim3(nb, nbatch), dim3(256), 0, s, w, codes, p, levels ? in : (const T *)nullptr);
        } else if (DEC) {
            hipLaunchKernelGGL((k_interp_first_dec<T>), dim3(1), dim3(64), 0, s, w, codes, ip.eb, ip.radius);
        } else {
            hipLaunchKernelGGL((k_interp_anchors<T>), dim3(nb, nbatch), dim3(256), 0, s, w, codes, p, levels ? in : (const T *)nullptr);
        }
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int szk_launch_interp_compress(int dtype, const szk_interp_params *ip, const void *d_in, void *d_work, uint16_t *codes,
                               uint64_t *hist, hipStream_t s) {
    uint64_t num = 1;
    for (int i = 0; i < ip->N; i++) num *= ip->dims[i];
    const size_t tsz = dtype == 0 ? 4 : 8;
    hipError_t e = hipSuccess;
    const bool levels = d_in && szk_interp_levels_ok(ip);  // level kernels: no working copy, the originals stay where they are
    if (d_in && !levels) e = hipMemcpyAsync(d_work, d_in, num * tsz, hipMemcpyDeviceToDevice, s);  // the dispatcher's dataCopy
    if (e != hipSuccess) return (int)e;
    const void *orig = levels ? d_in : nullptr;
    int rc = dtype == 0 ? run_interp<float, false>(*ip, (const float *)orig, (float *)d_work, codes, s)
                        : run_interp<double, false>(*ip, (const double *)orig, (double *)d_work, codes, s);
    if (rc) return rc;
    const void *raw_src = levels ? d_in : d_work;  // an unpredictable point's value: the original (kept in place by the passes)
    const bool smallr = ip->radius <= IH_WIN / 2;  // window start <= 0: code 0 lies inside it
    // tail passes: only with the large second tier (which covers radius +- 8192) and an alphabet that reaches beyond it
    const bool tails = ip->hist_big && ip->hist_tail && ip->radius > IHW_WIN && SZH_HIST_BINS == 65536;
#define SZK_HIST_LAUNCH(T, SR)                                                                                                    \
    do {                                                                                                                          \
        if (ip->hist_big)                                                                                                         \
            hipLaunchKernelGGL((k_hist_codes<T, SR, true>), dim3(256), dim3(1024), 0, s, codes, num, ip->radius, hist, (const T *)raw_src, \
                               ip->n_vout, ip->vout_idx, (T *)ip->vout_val, ip->out_cap, ip->far_cnt, tails ? 1u : 0u);           \
        else                                                                                                                      \
            hipLaunchKernelGGL((k_hist_codes<T, SR, false>), dim3(512), dim3(512), 0, s, codes, num, ip->radius, hist, (const T *)raw_src, \
                               ip->n_vout, ip->vout_idx, (T *)ip->vout_val, ip->out_cap, ip->far_cnt, 0u);                        \
    } while (0)
    if (dtype == 0) {
        if (smallr) SZK_HIST_LAUNCH(float, true);
        else SZK_HIST_LAUNCH(float, false);
    } else {
        if (smallr) SZK_HIST_LAUNCH(double, true);
        else SZK_HIST_LAUNCH(double, false);
    }
#undef SZK_HIST_LAUNCH
    if (tails) hipLaunchKernelGGL(k_hist_tail, dim3(170, 3), dim3(256), 0, s, codes, num, ip->radius, hist);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int szk_launch_interp_decompress(int dtype, const szk_interp_params *ip, const uint8_t *payload, uint64_t vout_idx_off,
                                 uint64_t vout_val_off, uint64_t n_vout, uint16_t *codes, void *d_out, hipStream_t s) {
    uint64_t num = 1;
    for (int i = 0; i < ip->N; i++) num *= ip->dims[i];
    if (n_vout) {
        const uint32_t g = (uint32_t)((n_vout + 255) / 256 < 4096 ? (n_vout + 255) / 256 : 4096);
        if (dtype == 0) hipLaunchKernelGGL((k_scatter_raw<float>), dim3(g), dim3(256), 0, s, payload, vout_idx_off, vout_val_off, n_vout, num, (float *)d_out);
        else hipLaunchKernelGGL((k_scatter_raw<double>), dim3(g), dim3(256), 0, s, payload, vout_idx_off, vout_val_off, n_vout, num, (double *)d_out);
    }
    return dtype == 0 ? run_interp<float, true>(*ip, (const float *)nullptr, (float *)d_out, codes, s)
                      : run_interp<double, true>(*ip, (const double *)nullptr, (double *)d_out, codes, s);
}

// ---- ALGO_INTERP_LORENZO tuner: device side (SZ_compress_Interp_lorenzo, api/impl/SZAlgoInterp.hpp:122-286) ---------
// profiling_block (utils/Sample.hpp:9-136): one thread per candidate block origin (multiples of bs below dim - bs);
// flags the blocks whose strided samples span more than abseb (same min / else-if-max walk as the reference)
struct szk_prof_params {
    int N;
    uint64_t off[4], cnt[4], total, bs, stride;
    double abseb;
};
template <typename T>
__global__ __launch_bounds__(256) void k_profile_blocks(const T *__restrict__ data, szk_prof_params p, uint8_t *__restrict__ flags) {
    // one wave per candidate block; lanes share the strided sample points. The reference's sequential
    // "if (v < min) min = v; else if (v > max) max = v" walk equals the plain min / max of the samples (NaN samples never
    // update either, in both forms)
    const uint64_t t = (uint64_t)blockIdx.x * 4 + threadIdx.x / 64;
    if (t >= p.total) return;
    const int lane = threadIdx.x & 63;
    uint64_t r = t, start = 0;
    for (int j = p.N - 1; j >= 0; j--) {
        start += (r % p.cnt[j]) * p.bs * p.off[j];
        r /= p.cnt[j];
    }
    const uint64_t m = p.bs / p.stride + 1;  // sample points per dimension: 0, stride, ..., <= bs
    uint64_t npts = 1;
    for (int j = 0; j < p.N; j++) npts *= m;
    const T first = data[start];
    T mn = first, mx = first;
    for (uint64_t q = lane; q < npts; q += 64) {
        uint64_t rr = q, idx = start;
        for (int j = p.N - 1; j >= 0; j--) {
            idx += (rr % m) * p.stride * p.off[j];
            rr /= m;
        }
        const T v = data[idx];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const T a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
        if (a < mn) mn = a;
        if (b > mx) mx = b;
    }
    if (lane == 0) flags[t] = (mx - mn > p.abseb) ? 1 : 0;
}
// sample_blocks (utils/Sample.hpp:138-219): copy the edge^N block at starts[b] into the b-th slot of the batch
template <typename T>
__global__ __launch_bounds__(256) void k_gather_blocks(const T *__restrict__ data, szk_prof_params p, uint64_t edge, uint64_t per,
                                                       const uint64_t *__restrict__ starts, T *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= per) return;
    const uint64_t *st = starts + (uint64_t)blockIdx.y * 4;
    uint64_t r = t, idx = 0;
    for (int j = p.N - 1; j >= 0; j--) {
        idx += (st[j] + r % edge) * p.off[j];
        r /= edge;
    }
    out[(uint64_t)blockIdx.y * per + t] = data[idx];
}
// Priced size of a trial's code stream from its histogram alone: res[0] = sum over the alphabet of f * log2(total / f) in
// 1/256-bit fixed point (integer atomics: the sum does not depend on arrival order), res[1] = symbols in use. The
// entropy tracks the Huffman-coded size closely enough for the tuner's ratio comparisons (tests/checks/estimator_study.py:
// 29 vs 30 of 41 decisions equal to the reference's) and needs no code book.
__global__ __launch_bounds__(256) void k_code_cost(const uint64_t *__restrict__ hist, const uint64_t *__restrict__ counters,
                                                   unsigned long long *res, double total, int unpred_is_code0) {
    const size_t book = blockIdx.y;  // batch of trials: histograms sliced per book, results 4 words apart, counters 8 apart
    hist += book * SZH_HIST_BINS;
    counters += book * 8;
    res += book * 4;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {  // res[2] = unpredictable values (interpolation: the points coded 0), res[3] = delta outliers
        res[2] = unpred_is_code0 ? hist[0] : counters[0];
        res[3] = counters[1];
    }
    const uint64_t f = hist[i];
    unsigned long long v = 0, c = f != 0;
    if (f) v = (unsigned long long)((double)f * log2(total / (double)f) * 256.0 + 0.5);
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_xor(v, off, 64);
        c += __shfl_xor(c, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && c) {
        atomicAdd(res, v);
        atomicAdd(res + 1, c);
    }
}

int szk_launch_profile_blocks(int dtype, const void *d_in, int N, const uint64_t *dims, uint64_t bs, uint64_t stride, double abseb,
                              uint8_t *d_flags, uint64_t *total_out, hipStream_t s) {
    szk_prof_params p;
    memset(&p, 0, sizeof(p));
    p.N = N;
    p.off[N - 1] = 1;
    for (int i = N - 2; i >= 0; i--) p.off[i] = p.off[i + 1] * dims[i + 1];
    p.total = 1;
    for (int i = 0; i < N; i++) {
        if (dims[i] < bs) {
            *total_out = 0;
            return 0;
        }
        p.cnt[i] = (dims[i] - bs + bs - 1) / bs;  // origins 0, bs, 2 bs, ... strictly below dim - bs
        p.total *= p.cnt[i];
    }
    p.bs = bs;
    p.stride = stride ? stride : bs;
    p.abseb = abseb;
    *total_out = p.total;
    if (p.total == 0) return 0;
    const uint32_t g = (uint32_t)((p.total + 3) / 4);
    if (dtype == 0) hipLaunchKernelGGL((k_profile_blocks<float>), dim3(g), dim3(256), 0, s, (const float *)d_in, p, d_flags);
    else hipLaunchKernelGGL((k_profile_blocks<double>), dim3(g), dim3(256), 0, s, (const double *)d_in, p, d_flags);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
int szk_launch_gather_blocks(int dtype, const void *d_in, int N, const uint64_t *dims, uint64_t edge, const uint64_t *d_starts,
                             uint32_t nblocks, void *d_out, hipStream_t s) {
    szk_prof_params p;
    memset(&p, 0, sizeof(p));
    p.N = N;
    p.off[N - 1] = 1;
    for (int i = N - 2; i >= 0; i--) p.off[i] = p.off[i + 1] * dims[i + 1];
    uint64_t per = 1;
    for (int i = 0; i < N; i++) per *= edge;
    const dim3 g((uint32_t)((per + 255) / 256), nblocks);
    if (dtype == 0) hipLaunchKernelGGL((k_gather_blocks<float>), g, dim3(256), 0, s, (const float *)d_in, p, edge, per, d_starts, (float *)d_out);
    else hipLaunchKernelGGL((k_gather_blocks<double>), g, dim3(256), 0, s, (const double *)d_in, p, edge, per, d_starts, (double *)d_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
// Tuner trials (interp_compress_test, SZAlgoInterp.hpp:42-78, decomposition part) in ONE launch: workgroup (b, j) runs the
// whole pass schedule of trial j over sample block b (a private copy of the block in `work`; the passes of a block only
// need workgroup-level ordering), then adds its codes to trial j's histogram through an LDS window. Unpredictables are
// only counted (out_cap = 0 in the schedules).
#define TRIAL_MAX_PASSES 64
__device__ unsigned long long g_trial_ts[TRIAL_MAX_PASSES + 8];  // development: per-pass time stamps of workgroup (0, 1)
extern "C" int szk_debug_trial_ts(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trial_ts), sizeof(g_trial_ts));
}
template <typename T>
__global__ __launch_bounds__(1024) void k_interp_trials(const T *__restrict__ samples, T *__restrict__ work, uint16_t *__restrict__ codes,
                                                        const szk_interp_pass *__restrict__ passes, const uint32_t *__restrict__ npasses,
                                                        uint64_t per, uint64_t *__restrict__ hists) {
    __shared__ szk_interp_pass sps[TRIAL_MAX_PASSES];  // the whole schedule of this trial (one fetch, not one per pass)
    __shared__ uint32_t lh[IH_WIN];
    const uint32_t b = blockIdx.x, j = blockIdx.y, nb = gridDim.x, tid = threadIdx.x;
    const uint64_t base = ((uint64_t)j * nb + b) * per;
    T *w = work + base;
    uint16_t *c = codes + base;
    const T *in = samples + (uint64_t)b * per;
    const uint32_t np = npasses[j];
    {
        const uint32_t words = np * (uint32_t)(sizeof(szk_interp_pass) / 4);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(passes + (size_t)j * TRIAL_MAX_PASSES);
        for (uint32_t i = tid; i < words; i += 1024) reinterpret_cast<uint32_t *>(sps)[i] = src[i];
    }
    for (uint64_t i = tid; i < per; i += 1024) w[i] = in[i];
    for (uint32_t i = tid; i < IH_WIN; i += 1024) lh[i] = 0;
    for (uint32_t k = 0; k < np; k++) {
        __syncthreads();  // previous pass complete
        if (tid == 0 && b == 0 && j == 1) g_trial_ts[k] = wall_clock64();
        const szk_interp_pass &sp = sps[k];
        if (sp.kind == 2) {
            for (uint64_t t = tid; t < sp.total; t += 1024) interp_point<T, false, uint32_t>(w, c, sp, t, base);  // (a block has < 2^32 points)
        } else {
            for (uint64_t t = tid; t < sp.total; t += 1024) anchor_point<T, uint32_t>(w, c, sp, t, base);
        }
    }
    const int radius = sps[0].radius;
    __syncthreads();
    if (tid == 0 && b == 0 && j == 1) g_trial_ts[np] = wall_clock64();
    uint64_t *hist = hists + (size_t)j * SZH_HIST_BINS;
    const uint32_t win_lo = (uint32_t)(radius - IH_WIN / 2);
    for (uint64_t i = tid; i < per; i += 1024) {
        const uint32_t code = c[i];
        const uint32_t bin = code - win_lo;
        if (bin < IH_WIN) atomicAdd(&lh[bin], 1u);
        else atomicAdd((unsigned long long *)&hist[code], 1ull);
    }
    __syncthreads();
    for (uint32_t bb = tid; bb < IH_WIN; bb += 1024) {
        const uint32_t v = lh[bb];
        const uint32_t sym = win_lo + bb;
        if (v && sym < SZH_HIST_BINS) atomicAdd((unsigned long long *)&hist[sym], (unsigned long long)v);
    }
    if (tid == 0 && b == 0 && j == 1) {
        g_trial_ts[np + 1] = wall_clock64();
        g_trial_ts[TRIAL_MAX_PASSES + 7] = np;
    }
}

// The same trials with the sample block resident in LDS (blocks up to 140 KB: 33^3 f32, 129^2 f32 / f64, 1-D): neighbour reads
// cost LDS latency instead of an L2 round trip per point (the passes of a block are latency-bound: 17 points per thread in
// the last pass), and codes go straight into the histogram window instead of through a code array.
#define TRIAL_LDS_BYTES 143752
#define TRIAL_LDS_PASSES 48
template <typename T>
__global__ __launch_bounds__(1024) void k_interp_trials_lds(const T *__restrict__ samples, const szk_interp_pass *__restrict__ passes,
                                                            const uint32_t *__restrict__ npasses, uint32_t per,
                                                            uint64_t *__restrict__ hists, uint16_t *__restrict__ codes) {
    __shared__ __align__(16) T w[TRIAL_LDS_BYTES / sizeof(T)];
    __shared__ szk_interp_pass sps[TRIAL_LDS_PASSES];
    __shared__ uint32_t lh[IH_WIN];
    const uint32_t b = blockIdx.x, j = blockIdx.y, nb = gridDim.x, tid = threadIdx.x;
    const uint64_t base = ((uint64_t)j * nb + b) * per;  // (offsets the indices of counted unpredictables, and the codes when they are kept)
    uint16_t *c = codes ? codes + base : nullptr;
    const T *in = samples + (uint64_t)b * per;
    const uint32_t np = npasses[j];
    {
        const uint32_t words = np * (uint32_t)(sizeof(szk_interp_pass) / 4);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(passes + (size_t)j * TRIAL_MAX_PASSES);
        for (uint32_t i = tid; i < words; i += 1024) reinterpret_cast<uint32_t *>(sps)[i] = src[i];
    }
    for (uint32_t i = tid; i < per; i += 1024) w[i] = in[i];
    for (uint32_t i = tid; i < IH_WIN; i += 1024) lh[i] = 0;
    __syncthreads();
    TrialSink sink;
    sink.lh = lh;
    sink.hist = reinterpret_cast<unsigned long long *>(hists + (size_t)j * SZH_HIST_BINS);
    sink.win_lo = (uint32_t)(sps[0].radius - IH_WIN / 2);
    for (uint32_t k = 0; k < np; k++) {
        const szk_interp_pass &sp = sps[k];
        if (sp.kind == 2) {
            for (uint32_t t = tid; t < (uint32_t)sp.total; t += 1024) interp_point<T, false, uint32_t, true>(w, c, sp, t, base, &sink);
        } else {
            for (uint32_t t = tid; t < (uint32_t)sp.total; t += 1024) anchor_point<T, uint32_t, true>(w, c, sp, t, base, &sink);
        }
        __syncthreads();
    }
    for (uint32_t bb = tid; bb < IH_WIN; bb += 1024) {
        const uint32_t v = lh[bb];
        const uint32_t sym = sink.win_lo + bb;
        if (v && sym < SZH_HIST_BINS) atomicAdd(&sink.hist[sym], (unsigned long long)v);
    }
}

// ips[j]: parameters of trial j (dims = the sample block's, n_vout = that trial's counter, out_cap = 0); the schedules are
// built on the host into h_passes (pinned) and copied to d_passes
int szk_launch_interp_trials(int dtype, const szk_interp_params *ips, uint32_t ntrials, const void *d_samples, void *d_work,
                             uint16_t *codes, uint32_t nblocks, uint64_t *d_hists, szk_interp_pass *h_passes, szk_interp_pass *d_passes,
                             uint32_t *h_np, uint32_t *d_np, int keep_codes, hipStream_t s) {
    uint64_t per = 1;
    for (int i = 0; i < ips[0].N; i++) per *= ips[0].dims[i];
    for (uint32_t j = 0; j < ntrials; j++) {
        std::vector<szk_interp_pass> sched;
        if (build_schedule(ips[j], false, 1, sched)) return -1;
        if (sched.size() > TRIAL_MAX_PASSES) return -2;
        h_np[j] = (uint32_t)sched.size();
        memcpy(h_passes + (size_t)j * TRIAL_MAX_PASSES, sched.data(), sched.size() * sizeof(szk_interp_pass));
    }
    hipError_t e = hipMemcpyAsync(d_passes, h_passes, (size_t)ntrials * TRIAL_MAX_PASSES * sizeof(szk_interp_pass), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(d_np, h_np, ntrials * 4, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return (int)e;
    const size_t tsz = dtype == 0 ? 4 : 8;
    bool lds = per * tsz <= TRIAL_LDS_BYTES;
    for (uint32_t j = 0; j < ntrials; j++) lds = lds && h_np[j] <= TRIAL_LDS_PASSES;
    if (lds && dtype == 0)
        hipLaunchKernelGGL((k_interp_trials_lds<float>), dim3(nblocks, ntrials), dim3(1024), 0, s, (const float *)d_samples, d_passes, d_np,
                           (uint32_t)per, d_hists, keep_codes ? codes : nullptr);  // (keep_codes: per-element codes besides the histograms, for the exact pricing)
    else if (lds)
        hipLaunchKernelGGL((k_interp_trials_lds<double>), dim3(nblocks, ntrials), dim3(1024), 0, s, (const double *)d_samples, d_passes, d_np,
                           (uint32_t)per, d_hists, keep_codes ? codes : nullptr);
    else if (dtype == 0)
        hipLaunchKernelGGL((k_interp_trials<float>), dim3(nblocks, ntrials), dim3(1024), 0, s, (const float *)d_samples, (float *)d_work, codes,
                           d_passes, d_np, per, d_hists);
    else
        hipLaunchKernelGGL((k_interp_trials<double>), dim3(nblocks, ntrials), dim3(1024), 0, s, (const double *)d_samples, (double *)d_work, codes,
                           d_passes, d_np, per, d_hists);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
int szk_launch_code_cost(const uint64_t *hist, const uint64_t *counters, uint64_t *d_res, uint32_t n_books, uint64_t total,
                         int unpred_is_code0, hipStream_t s) {
    hipLaunchKernelGGL(k_code_cost, dim3(SZH_HIST_BINS / 256, n_books), dim3(256), 0, s, hist, counters, (unsigned long long *)d_res,
                       (double)total, unpred_is_code0);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
