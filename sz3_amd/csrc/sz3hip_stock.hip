// sz3_amd/csrc/sz3hip_stock.hip — the device side of stock-stream interoperability: the codes of an ALGO_INTERP
// stream live in the reference's EMISSION order (sz3hip_stock_geom.h), this library's interpolation kernels keep them per ELEMENT.
// Two permutations and the bookkeeping of the unpredictable values connect the two:
//   k_stock_to_elem    emission-order codes (+ the quantizer's list of unpredictable values, in the order their zero codes were
//                      emitted: quantizer/LinearQuantizer.hpp:74-86 recover() pops them one by one) -> codes[element] and the
//                      (index, value) lists szk_launch_interp_decompress scatters
//   k_stock_from_elem  codes[element] -> emission order;  k_stock_unpred_from_lists: the (index, value) list of a compression ->
//                      the quantizer's list in emission order
// The ordinal of a zero code among the zero codes before it is a two-level count: zeros per tile of 1024 emission positions
// (k_stock_zero_tiles), their exclusive scan (k_stock_tile_scan, one workgroup), and a walk over the tile's own positions — taken
// only by the zero codes themselves, which are rare (the anchor grid and the few points no interpolation predicts).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "sz3hip_devutil.h"
#include "sz3hip_kernels.h"
#include "sz3hip_stock_geom.h"

#define STOCK_TILE 1024u

__device__ __forceinline__ void stock_coords(const szg_geom &g, uint64_t e, uint64_t *x) {
    for (int i = g.N - 1; i >= 0; i--) {
        x[i] = e % g.d[i];
        e /= g.d[i];
    }
}
__device__ __forceinline__ uint64_t stock_ordinal(const uint16_t *__restrict__ em, const uint64_t *__restrict__ tile_base, uint64_t r) {
    const uint64_t t0 = r & ~(uint64_t)(STOCK_TILE - 1);
    uint64_t k = tile_base[r / STOCK_TILE];
    for (uint64_t q = t0; q < r; q++) k += em[q] == 0;
    return k;
}

__global__ __launch_bounds__(256) void k_stock_rank(szg_geom g, const uint64_t *__restrict__ blk_base, uint64_t *__restrict__ rank) {
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < g.n; e += (uint64_t)gridDim.x * 256) {
        uint64_t x[4];
        stock_coords(g, e, x);
        rank[e] = szg_rank(g, blk_base, x);
    }
}
__global__ __launch_bounds__(256) void k_stock_zero_tiles(const uint16_t *__restrict__ em, uint64_t n, uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t s_c;
    for (uint64_t t = blockIdx.x; t * STOCK_TILE < n; t += gridDim.x) {
        if (threadIdx.x == 0) s_c = 0;
        __syncthreads();
        uint32_t c = 0;
        for (uint32_t k = threadIdx.x; k < STOCK_TILE; k += 256) {
            const uint64_t r = t * STOCK_TILE + k;
            c += r < n && em[r] == 0;
        }
        c = wave_sum(c);
        if (lane_id() == 0 && c) atomicAdd(&s_c, c);
        __syncthreads();
        if (threadIdx.x == 0) tile_cnt[t] = s_c;
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_stock_tile_scan(const uint32_t *__restrict__ tile_cnt, uint64_t ntiles, uint64_t *__restrict__ tile_base, uint64_t *total) {
    __shared__ uint64_t s_w[16];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t t0 = 0; t0 < ntiles; t0 += 1024) {
        const uint64_t t = t0 + threadIdx.x;
        const uint64_t mine = t < ntiles ? tile_cnt[t] : 0;
        const uint64_t incl = wave_incl_scan(mine);
        if (lane_id() == WAVE - 1) s_w[threadIdx.x / WAVE] = incl;
        __syncthreads();
        uint64_t run = s_carry + incl - mine, tot = 0;
        for (int w = 0; w < 16; w++) {
            if (w < (int)(threadIdx.x / WAVE)) run += s_w[w];
            tot += s_w[w];
        }
        if (t < ntiles) tile_base[t] = run;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}
// emission order -> per element, and the unpredictable values to the (index, value) lists (list position = the zero code's ordinal)
template <typename V>
__global__ __launch_bounds__(256) void k_stock_to_elem(szg_geom g, const uint64_t *__restrict__ blk_base, const uint16_t *__restrict__ em,
                                                       const uint64_t *__restrict__ tile_base, const V *__restrict__ unpred, uint64_t n_unpred,
                                                       uint16_t *__restrict__ codes, uint64_t *__restrict__ vout_idx, V *__restrict__ vout_val,
                                                       uint32_t *bad) {
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < g.n; e += (uint64_t)gridDim.x * 256) {
        uint64_t x[4];
        stock_coords(g, e, x);
        const uint64_t r = szg_rank(g, blk_base, x);
        const uint16_t c = r < g.n ? em[r] : (uint16_t)0;
        if (r >= g.n) atomicOr(bad, 1u);
        codes[e] = c;
        if (c == 0) {
            const uint64_t k = stock_ordinal(em, tile_base, r);
            if (k < n_unpred) {
                vout_idx[k] = e;
                vout_val[k] = unpred[k];
            } else {
                atomicOr(bad, 2u);  // more zero codes than the stream lists values for
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_stock_from_elem(szg_geom g, const uint64_t *__restrict__ blk_base, const uint16_t *__restrict__ codes,
                                                         uint16_t *__restrict__ em) {
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < g.n; e += (uint64_t)gridDim.x * 256) {
        uint64_t x[4];
        stock_coords(g, e, x);
        em[szg_rank(g, blk_base, x)] = codes[e];
    }
}
template <typename V>
__global__ __launch_bounds__(256) void k_stock_unpred_from_lists(szg_geom g, const uint64_t *__restrict__ blk_base, const uint16_t *__restrict__ em,
                                                                 const uint64_t *__restrict__ tile_base, const uint64_t *__restrict__ vout_idx,
                                                                 const V *__restrict__ vout_val, uint64_t n_vout, V *__restrict__ unpred) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n_vout; k += (uint64_t)gridDim.x * 256) {
        uint64_t x[4];
        stock_coords(g, vout_idx[k], x);
        const uint64_t r = szg_rank(g, blk_base, x);
        unpred[stock_ordinal(em, tile_base, r)] = vout_val[k];
    }
}

// ---- host: the geometry of an array and its tables (InterpolationDecomposition::init, :176-213) --------------------------------
int szk_stock_geom_build(int N, const uint64_t *dims, int interp_id, int direction, uint64_t anchor_stride, szg_geom *gp, std::vector<uint64_t> *blk_base) {
    szg_geom &g = *gp;
    memset(&g, 0, sizeof(g));
    if (N < 1 || N > 4) return -1;
    g.N = N;
    g.interp_id = interp_id ? 1 : 0;
    int level = -1;
    bool use_anchor = false;
    g.n = 1;
    for (int i = 0; i < N; i++) {
        if (dims[i] == 0) return -1;
        g.d[i] = dims[i];
        int cl = 0;  // ceil(log2(d))
        while (((uint64_t)1 << cl) < dims[i]) cl++;
        if (level < cl) level = cl;
        if (dims[i] > anchor_stride) use_anchor = true;
        g.n *= dims[i];
    }
    if (!use_anchor) anchor_stride = 0;
    if (anchor_stride & (anchor_stride - 1)) return -1;
    if (anchor_stride > 0) {
        int ml = 0;
        while (((uint64_t)1 << ml) < anchor_stride) ml++;
        ml += 1;  // log2(anchor_stride) + 1
        if (ml <= level) level = ml;
    }
    g.anchor = anchor_stride;
    // the direction's permutation: std::next_permutation order from the identity (:205-212)
    int p[4] = {0, 1, 2, 3};
    int nperm = 1;
    for (int i = 2; i <= N; i++) nperm *= i;
    if (direction < 0 || direction >= nperm) return -1;
    for (int k = 0; k < direction; k++) {
        int i = N - 2;
        while (i >= 0 && p[i] > p[i + 1]) i--;
        if (i < 0) break;
        int j = N - 1;
        while (p[j] < p[i]) j--;
        int t = p[i]; p[i] = p[j]; p[j] = t;
        for (int a = i + 1, b = N - 1; a < b; a++, b--) { t = p[a]; p[a] = p[b]; p[b] = t; }
    }
    for (int i = 0; i < N; i++) g.seq[i] = p[i];
    if (anchor_stride) {
        g.head = 1;
        for (int i = 0; i < N; i++) g.head *= (dims[i] - 1) / anchor_stride + 1;
        level--;
    } else {
        g.head = 1;
    }
    if (level >= SZG_MAX_LEVELS) return -1;
    g.top = level < 0 ? 0 : level;
    blk_base->clear();
    uint64_t run = g.head;
    for (int l = g.top; l >= 1; l--) {
        const uint64_t s = (uint64_t)1 << (l - 1), bsz = 32 * s;
        uint64_t total = 1;
        for (int i = 0; i < N; i++) {
            g.nb[l][i] = (dims[i] - 1) / bsz + 1;
            total *= g.nb[l][i];
        }
        g.level_base[l] = run;
        g.blk_off[l] = blk_base->size();
        uint64_t b[4] = {0, 0, 0, 0};
        for (uint64_t t = 0; t < total; t++) {
            uint64_t begin[4], end[4];
            szg_block_box(g, l, b, begin, end);
            blk_base->push_back(run);
            run += szg_block_total(g, begin, end, s);
            for (int i = N - 1; i >= 0; i--) {
                if (++b[i] < g.nb[l][i]) break;
                b[i] = 0;
            }
        }
    }
    return run == g.n ? 0 : -2;  // every element is emitted exactly once
}
// test hook (CPU, no device): the emission rank of every element of a small array
extern "C" int sz3hip_debug_stock_ranks(int N, const uint64_t *dims, int interp_id, int direction, uint64_t anchor_stride, uint64_t *ranks) {
    szg_geom g;
    std::vector<uint64_t> bb;
    const int rc = szk_stock_geom_build(N, dims, interp_id, direction, anchor_stride, &g, &bb);
    if (rc) return rc;
    for (uint64_t e = 0; e < g.n; e++) {
        uint64_t x[4], q = e;
        for (int i = N - 1; i >= 0; i--) {
            x[i] = q % g.d[i];
            q /= g.d[i];
        }
        ranks[e] = szg_rank(g, bb.data(), x);
    }
    return 0;
}

static inline uint32_t stock_grid(uint64_t n) { return (uint32_t)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536); }
// zero ordinals of an emission-order code array: tile counts + their scan (tile_cnt: ceil(n / 1024) u32, tile_base: as many u64 + 1 for the total)
static int stock_zero_scan(const uint16_t *d_em, uint64_t n, uint32_t *d_tile_cnt, uint64_t *d_tile_base, hipStream_t s) {
    const uint64_t ntiles = (n + STOCK_TILE - 1) / STOCK_TILE;
    hipLaunchKernelGGL(k_stock_zero_tiles, dim3((uint32_t)(ntiles < 65536 ? ntiles : 65536)), dim3(256), 0, s, d_em, n, d_tile_cnt);
    hipLaunchKernelGGL(k_stock_tile_scan, dim3(1), dim3(1024), 0, s, d_tile_cnt, ntiles, d_tile_base, d_tile_base + ntiles);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int szk_launch_stock_to_elem(int dtype, const szg_geom *g, const uint64_t *d_blk_base, const uint16_t *d_em, const void *d_unpred, uint64_t n_unpred,
                             uint32_t *d_tile_cnt, uint64_t *d_tile_base, uint16_t *d_codes, uint64_t *d_vout_idx, void *d_vout_val, uint32_t *d_bad,
                             hipStream_t s) {
    if (stock_zero_scan(d_em, g->n, d_tile_cnt, d_tile_base, s)) return -1;
    if (dtype == 0)
        hipLaunchKernelGGL((k_stock_to_elem<uint32_t>), dim3(stock_grid(g->n)), dim3(256), 0, s, *g, d_blk_base, d_em, d_tile_base, (const uint32_t *)d_unpred, n_unpred,
                           d_codes, d_vout_idx, (uint32_t *)d_vout_val, d_bad);
    else
        hipLaunchKernelGGL((k_stock_to_elem<uint64_t>), dim3(stock_grid(g->n)), dim3(256), 0, s, *g, d_blk_base, d_em, d_tile_base, (const uint64_t *)d_unpred, n_unpred,
                           d_codes, d_vout_idx, (uint64_t *)d_vout_val, d_bad);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int szk_launch_stock_from_elem(int dtype, const szg_geom *g, const uint64_t *d_blk_base, const uint16_t *d_codes, const uint64_t *d_vout_idx,
                               const void *d_vout_val, uint64_t n_vout, uint32_t *d_tile_cnt, uint64_t *d_tile_base, uint16_t *d_em, void *d_unpred,
                               hipStream_t s) {
    hipLaunchKernelGGL(k_stock_from_elem, dim3(stock_grid(g->n)), dim3(256), 0, s, *g, d_blk_base, d_codes, d_em);
    if (stock_zero_scan(d_em, g->n, d_tile_cnt, d_tile_base, s)) return -1;
    if (n_vout) {
        if (dtype == 0)
            hipLaunchKernelGGL((k_stock_unpred_from_lists<uint32_t>), dim3(stock_grid(n_vout)), dim3(256), 0, s, *g, d_blk_base, d_em, d_tile_base, d_vout_idx,
                               (const uint32_t *)d_vout_val, n_vout, (uint32_t *)d_unpred);
        else
            hipLaunchKernelGGL((k_stock_unpred_from_lists<uint64_t>), dim3(stock_grid(n_vout)), dim3(256), 0, s, *g, d_blk_base, d_em, d_tile_base, d_vout_idx,
                               (const uint64_t *)d_vout_val, n_vout, (uint64_t *)d_unpred);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int szk_launch_stock_ranks(const szg_geom *g, const uint64_t *d_blk_base, uint64_t *d_rank, hipStream_t s) {
    hipLaunchKernelGGL(k_stock_rank, dim3(stock_grid(g->n)), dim3(256), 0, s, *g, d_blk_base, d_rank);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
