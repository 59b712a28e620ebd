// sz3_amd/csrc/sz3hip_devutil.h — device-side helpers shared by the kernel translation units (wave64, gfx950)
#ifndef SZ3HIP_DEVUTIL_H
#define SZ3HIP_DEVUTIL_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "sz3hip_kernels.h"

#define WAVE 64

// ------------------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// Wave-wide inclusive scan / sum with DPP lane moves (no LDS round trip; __shfl_up lowers to ds_bpermute, ~100 cycles
// per step): Kogge-Stone inside the rows of 16 lanes (row_shr 1, 2, 4, 8), then row_bcast:15 carries the row totals into
// rows 1 and 3 and row_bcast:31 the half-wave total into rows 2 and 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov0(uint32_t v) {  // lanes without a source (or outside ROW_MASK) read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_mov0(uint64_t v) {
    const uint32_t lo = dpp_mov0<CTRL, ROW_MASK>((uint32_t)v), hi = dpp_mov0<CTRL, ROW_MASK>((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <typename V>
__device__ __forceinline__ V wave_incl_scan(V v) {
    using U = typename std::conditional<sizeof(V) == 8, uint64_t, uint32_t>::type;
    U u = (U)v;
    u += dpp_mov0<0x111, 0xf>(u);  // row_shr:1
    u += dpp_mov0<0x112, 0xf>(u);  // row_shr:2
    u += dpp_mov0<0x114, 0xf>(u);  // row_shr:4
    u += dpp_mov0<0x118, 0xf>(u);  // row_shr:8
    u += dpp_mov0<0x142, 0xa>(u);  // row_bcast:15 -> rows 1, 3
    u += dpp_mov0<0x143, 0xc>(u);  // row_bcast:31 -> rows 2, 3
    return (V)u;
}
template <typename V>
__device__ __forceinline__ V wave_sum(V v) {  // total in every lane
    const V incl = wave_incl_scan(v);
    if (sizeof(V) == 8) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)incl, WAVE - 1);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)incl >> 32), WAVE - 1);
        return (V)(((uint64_t)hi << 32) | lo);
    }
    return (V)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)incl, WAVE - 1);
}

// reserve one slot of an append-only list for every active lane with want == true: ONE atomic per wave (same-address
// global atomics run at ~90/us: a field with a third of NaNs would otherwise spend half a second here). Any set of active
// lanes may call it together. Returns the lane's slot (meaningful only where want).
__device__ __forceinline__ unsigned long long wave_append_slot(bool want, uint64_t *counter) {
    const unsigned long long m = __ballot(want);
    if (m == 0) return ~0ull;
    const int lane = lane_id();
    const int leader = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd((unsigned long long *)counter, (unsigned long long)__popcll(m));
    const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, leader);
    const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(base >> 32), leader);
    return (((unsigned long long)bhi << 32) | blo) + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
}

// the same for a RUN of slots per lane (cnt of them, 0 .. 7, 0 for none): one atomic per wave for all of them — prefix counts from three
// ballots (one per bit of cnt), so any set of active lanes may call it together. Returns the lane's first slot.
__device__ __forceinline__ unsigned long long wave_append_run(uint32_t cnt, uint64_t *counter) {
    const int lane = lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u);
    const unsigned long long any = b0 | b1 | b2;
    if (any == 0) return ~0ull;
    const uint32_t total = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
    const uint32_t before = (uint32_t)__popcll(b0 & lt) + 2u * (uint32_t)__popcll(b1 & lt) + 4u * (uint32_t)__popcll(b2 & lt);
    const int leader = __ffsll((long long)any) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd((unsigned long long *)counter, (unsigned long long)total);
    const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, leader);
    const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(base >> 32), leader);
    return (((unsigned long long)bhi << 32) | blo) + (unsigned long long)before;
}

template <typename T> struct QTraits;
template <> struct QTraits<float> {
    using Q = int32_t;
    using UQ = uint32_t;
};
template <> struct QTraits<double> {
    using Q = int64_t;
    using UQ = uint64_t;
};

// The quantisation lattice.  q = rint(x / 2eb), clamped to +-LIM; the reconstruction x^ = q * 2eb is verified against the
// bound (same acceptance test as quantizer/LinearQuantizer.hpp:57-60: |dec - data| evaluated in T, compared with eb) and the
// raw value is kept losslessly when the check fails.  The arithmetic type is the data type: f32 data use f32 multiplies
// (one rounding each, no FMA contraction: built with -ffp-contract=off), f64 data f64 — the decoder applies the identical
// expression, so the bound that the encoder verified is the bound the user gets.
//
// Rounding goes through the "magic number" M = 1.5 * 2^(mantissa bits): fl(s + M) has the integer rint(s) in its low
// mantissa bits for |s| <= LIM = 2^(mantissa bits - 1), so the BIT PATTERN of s + M is C + rint(s) (C = bits of M) — no
// rounding instruction, no float-to-integer conversion, and r = fl(s + M) - M is rint(s) as a float again (exact). Values
// beyond the lattice (|x / 2eb| > LIM: fill values like 1e35, +-Inf) and NaN take q = 0 — like a neighbour outside the array,
// so that the points around a masked region still predict sanely; their reconstruction fails the check: raw value kept.
// `qbits` returns that offset pattern: the stencils difference it as it is (the offset cancels; a neighbour outside the
// array is C), which saves the subtraction too.
template <typename T> struct Lattice;
template <> struct Lattice<float> {
    using B = int32_t;
    static constexpr B C = 0x4B400000, LIM = 1 << 22;
    float recip, two_eb, eb_lo;  // (float)(1/(2eb)), (float)(2eb), largest float <= eb
    __device__ __forceinline__ explicit Lattice(const szk_lattice &l) : recip(l.recip_f), two_eb(l.two_eb_f), eb_lo(l.eb_lo_f) {}
    __device__ __forceinline__ B qbits(float x) const {
        const float s = x * recip;
        const float tm = s + 12582912.0f;  // (two roundings: -ffp-contract=off)
        return fabsf(s) <= 4194304.0f ? __float_as_int(tm) : C;  // beyond the lattice, Inf, NaN: 0, so that neighbours still predict sanely
    }
    __device__ __forceinline__ float rounded(B bits) const { return __int_as_float(bits) - 12582912.0f; }  // rint(s) as a float
    __device__ __forceinline__ bool bad(float x, B bits) const { return !(fabsf(rounded(bits) * two_eb - x) <= eb_lo); }
    __device__ __forceinline__ int32_t quant(float x, bool &is_bad) const {
        const B b = qbits(x);
        is_bad = bad(x, b);
        return b - C;
    }
    __device__ __forceinline__ float dequant(int32_t q) const { return (float)q * two_eb; }
};
template <> struct Lattice<double> {
    using B = int64_t;
    static constexpr B C = 0x4338000000000000ll, LIM = 1ll << 51;
    double recip, two_eb, eb;
    __device__ __forceinline__ explicit Lattice(const szk_lattice &l) : recip(l.recip), two_eb(l.two_eb), eb(l.eb) {}
    __device__ __forceinline__ B qbits(double x) const {
        const double s = x * recip;
        const double tm = s + 6755399441055744.0;
        return fabs(s) <= 2251799813685248.0 ? __double_as_longlong(tm) : C;
    }
    __device__ __forceinline__ double rounded(B bits) const { return __longlong_as_double(bits) - 6755399441055744.0; }
    __device__ __forceinline__ bool bad(double x, B bits) const { return !(fabs(rounded(bits) * two_eb - x) <= eb); }
    __device__ __forceinline__ int64_t quant(double x, bool &is_bad) const {
        const B b = qbits(x);
        is_bad = bad(x, b);
        return b - C;
    }
    __device__ __forceinline__ double dequant(int64_t q) const { return (double)q * two_eb; }
};


// ---- LinearQuantizer<T>::quantize_and_overwrite / recover (LinearQuantizer.hpp:43-86) ---------------------------
template <typename T>
__device__ __forceinline__ int ref_quantize(T &data, T pred, double eb, double recip, int radius) {
    const T diff = data - pred;
    const double scaled = fabs((double)diff) * recip;
    // the reference casts to int64, adds 1 and asks "< 2 * radius": true exactly when scaled < 2 * radius - 1 (a NaN or an
    // overflowing quotient fails it: unpredictable). Inside that range the quotient fits 32 bits, where the conversions
    // are single instructions (the 64-bit ones are emulated: they were a third of the pass kernels' time).
    if (!(scaled < (double)(2 * radius - 1))) return 0;
    int qi = (int)scaled + 1;
    const int half = qi >> 1;
    qi = half << 1;
    int shifted;
    if (diff < 0) {
        qi = -qi;
        shifted = radius - half;
    } else {
        shifted = radius + half;
    }
    const T dec = (T)((double)pred + (double)qi * eb);
    const T ad = dec - data;
    const double adiff = fabs((double)ad);
    if (adiff <= eb) {
        data = dec;
        return shifted;
    }
    return 0;
}
template <typename T>
__device__ __forceinline__ T ref_recover(T pred, int code, double eb, int radius) {
    return (T)((double)pred + (double)(2 * (code - radius)) * eb);
}


#endif
