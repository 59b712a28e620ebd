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
    // (a 1-D array has one order whatever the field says: the reference's OpenMP path hands a 2-D array's one-row slabs — a dimension dropped,
    // Config::setDims — to the 1-D interpolation with the caller's interpDirection, which then indexes past its one-entry table of orders
    // and, on both sides, finds the identity; tests/checks/wild_data_sweep.py, leg 3)
    if (N == 1) direction = 0;
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

// ------------------------------------------------------------------------------------------------------------------------------
// The Huffman stage of a stock stream on the device (encoder/HuffmanEncoder.hpp:140-255). The reference's bit stream is ONE
// unindexed MSB-first string over a tree it serialises node by node:
//   * coding is embarrassingly parallel once the code words' lengths are scanned: a workgroup takes a tile of 2048 symbols, places
//     every code word at its bit offset inside an LDS stage (ds_or), and copies the stage out shifted by the tile's bit offset in
//     the stream (the two words it shares with its neighbours by atomic OR into the zeroed stream);
//   * decoding has no entry points, but a Huffman decoder that starts at a wrong bit re-synchronises with the true code word
//     boundaries within a few symbols (the property Weissenberger & Schmidt's parallel decoder builds on): the stream is cut into
//     subsequences of 4096 bits, thread i decodes from where thread i - 1's last pass ENDED (initially from the subsequence's first
//     bit) to the first code word boundary at or beyond the next subsequence, passes repeat until no end moves — thread 0 is right
//     from the start, every pass extends the right prefix by at least one subsequence, and in practice two or three passes settle
//     everything — then the symbol counts are scanned and a last pass writes the symbols.
// Code words are looked up in a 12-bit table held in LDS (node reached, bits used, leaf?); longer ones walk the tree's child arrays.
// ------------------------------------------------------------------------------------------------------------------------------
#define STOCK_ENC_TILE 2048u
#define STOCK_SUB_BITS 4096u
#define STOCK_LUT_BITS 12u

// 32 bits of the MSB-first stream from bit `pos` on (zeros beyond its end)
__device__ __forceinline__ uint32_t stock_peek32(const uint32_t *__restrict__ words, uint64_t nwords, uint64_t pos) {
    const uint64_t w = pos >> 5;
    const uint32_t sh = (uint32_t)(pos & 31);
    const uint32_t a = w < nwords ? __builtin_bswap32(words[w]) : 0u, b = w + 1 < nwords ? __builtin_bswap32(words[w + 1]) : 0u;
    return sh ? (a << sh) | (b >> (32 - sh)) : a;
}
// one symbol from bit pos; returns the leaf's node (or 0xFFFFFFFF: the walk left the tree / the stream) and advances pos
__device__ __forceinline__ uint32_t stock_decode_one(const szk_stock_tree_dev &tr, const uint32_t *s_lut, const uint32_t *__restrict__ words, uint64_t nwords,
                                                     uint64_t total_bits, uint64_t &pos) {
    uint32_t win = stock_peek32(words, nwords, pos);
    const uint32_t e = s_lut[win >> (32 - STOCK_LUT_BITS)];
    uint32_t node = e >> 8, used = (e >> 1) & 127u;
    if (e & 1u) {
        pos += used;
        return pos <= total_bits ? node : 0xFFFFFFFFu;
    }
    if (used == 0) return 0xFFFFFFFFu;  // (a child missing right below the root: corrupt tree)
    pos += used;
    win <<= used;
    uint32_t have = 32 - used;
    for (;;) {  // code words beyond the table: bit by bit
        if (have == 0) {
            win = stock_peek32(words, nwords, pos);
            have = 32;
        }
        if (pos >= total_bits) return 0xFFFFFFFFu;
        node = (win >> 31) ? tr.R[node] : tr.L[node];
        win <<= 1;
        have--;
        pos++;
        if (node == 0) return 0xFFFFFFFFu;
        if (tr.t[node]) return node;
    }
}
__global__ __launch_bounds__(256) void k_stock_huff_sync(szk_stock_tree_dev tr, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t total_bits, uint64_t nsub,
                                                         const uint64_t *__restrict__ start, uint64_t *__restrict__ last_start, uint64_t *__restrict__ next_start,
                                                         uint32_t *__restrict__ count, uint32_t *changed) {
    __shared__ uint32_t s_lut[1u << STOCK_LUT_BITS];
    for (uint32_t i = threadIdx.x; i < (1u << STOCK_LUT_BITS); i += 256) s_lut[i] = tr.lut[i];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nsub; i += (uint64_t)gridDim.x * 256) {
        const uint64_t st = start[i];
        if (last_start[i] == st) {  // decoded from here already: its end stands
            continue;
        }
        last_start[i] = st;
        const uint64_t bound = (i + 1) * STOCK_SUB_BITS < total_bits ? (i + 1) * STOCK_SUB_BITS : total_bits;
        uint64_t pos = st;
        uint32_t cnt = 0;
        while (pos < bound) {
            uint64_t p2 = pos;
            if (stock_decode_one(tr, s_lut, words, nwords, total_bits, p2) == 0xFFFFFFFFu) break;  // (the stream's padding, or garbage: ends here)
            pos = p2;
            cnt++;
        }
        if (pos < bound) pos = bound > pos ? bound : pos;  // (nothing decodable up to the boundary: the next thread starts at the boundary)
        count[i] = cnt;
        if (next_start[i + 1] != pos) {
            next_start[i + 1] = pos;
            *changed = 1u;
        }
    }
}
__global__ __launch_bounds__(256) void k_stock_huff_write(szk_stock_tree_dev tr, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t total_bits, uint64_t nsub,
                                                          const uint64_t *__restrict__ start, const uint64_t *__restrict__ out_base, uint64_t n,
                                                          uint16_t *__restrict__ em, uint32_t *bad) {
    __shared__ uint32_t s_lut[1u << STOCK_LUT_BITS];
    for (uint32_t i = threadIdx.x; i < (1u << STOCK_LUT_BITS); i += 256) s_lut[i] = tr.lut[i];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nsub; i += (uint64_t)gridDim.x * 256) {
        const uint64_t bound = (i + 1) * STOCK_SUB_BITS < total_bits ? (i + 1) * STOCK_SUB_BITS : total_bits;
        uint64_t pos = start[i], k = out_base[i];
        while (pos < bound && k < n) {
            uint64_t p2 = pos;
            const uint32_t node = stock_decode_one(tr, s_lut, words, nwords, total_bits, p2);
            if (node == 0xFFFFFFFFu) break;
            pos = p2;
            const int32_t v = tr.C[node] + tr.offset;
            if (v < 0 || v > 65535) {
                atomicOr(bad, 4u);
                break;
            }
            em[k++] = (uint16_t)v;
        }
    }
}
// exclusive scan of u32 counts into u64 bases (one workgroup), total behind the last
__global__ __launch_bounds__(1024) void k_stock_scan32(const uint32_t *__restrict__ cnt, uint64_t m, uint64_t *__restrict__ base) {
    __shared__ uint64_t s_w[16];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t t0 = 0; t0 < m; t0 += 1024) {
        const uint64_t t = t0 + threadIdx.x;
        const uint64_t mine = t < m ? cnt[t] : 0;
        const uint64_t incl = wave_incl_scan(mine);
        if (lane_id() == WAVE - 1) s_w[threadIdx.x / WAVE] = incl;
        __syncthreads();
        uint64_t run = s_carry + incl - mine, tot = 0;
        for (int w = 0; w < 16; w++) {
            if (w < (int)(threadIdx.x / WAVE)) run += s_w[w];
            tot += s_w[w];
        }
        if (t < m) base[t] = run;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) base[m] = s_carry;
}

// ---- coding: code-word table by symbol (length, bits), tiles of 2048 symbols ----
__global__ __launch_bounds__(256) void k_stock_enc_bits(const uint16_t *__restrict__ em, uint64_t n, const uint8_t *__restrict__ clen, uint32_t *__restrict__ tile_bits) {
    __shared__ uint32_t s_c;
    for (uint64_t t = blockIdx.x; t * STOCK_ENC_TILE < n; t += gridDim.x) {
        if (threadIdx.x == 0) s_c = 0;
        __syncthreads();
        uint32_t c = 0;
        for (uint32_t k = threadIdx.x; k < STOCK_ENC_TILE; k += 256) {
            const uint64_t r = t * STOCK_ENC_TILE + k;
            if (r < n) c += clen[em[r]];
        }
        c = wave_sum(c);
        if (lane_id() == 0 && c) atomicAdd(&s_c, c);
        __syncthreads();
        if (threadIdx.x == 0) tile_bits[t] = s_c;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_stock_enc_pack(const uint16_t *__restrict__ em, uint64_t n, const uint8_t *__restrict__ clen, const uint64_t *__restrict__ cbits,
                                                        const uint64_t *__restrict__ tile_base, uint32_t *__restrict__ out_words) {
    constexpr uint32_t PER = STOCK_ENC_TILE / 256;  // consecutive symbols per thread
    constexpr uint32_t STAGE = STOCK_ENC_TILE * 2 + 4;  // words: 64 bits per symbol at worst
    __shared__ uint32_t s_stage[STAGE];
    __shared__ uint32_t s_wave[4];
    for (uint64_t t = blockIdx.x; t * STOCK_ENC_TILE < n; t += gridDim.x) {
        const uint64_t bit0 = tile_base[t];
        const uint32_t tbits = (uint32_t)(tile_base[t + 1] - bit0);
        const uint32_t nw = (tbits + 31) >> 5;
        for (uint32_t i = threadIdx.x; i < nw + 2; i += 256) s_stage[i] = 0;
        uint32_t len[PER];
        uint64_t bits[PER];
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint64_t r = t * STOCK_ENC_TILE + (uint64_t)threadIdx.x * PER + k;
            const uint32_t sym = r < n ? em[r] : 0u;
            len[k] = r < n ? clen[sym] : 0u;
            bits[k] = r < n ? cbits[sym] : 0ull;
            mine += len[k];
        }
        const uint32_t incl = wave_incl_scan(mine);
        if (lane_id() == WAVE - 1) s_wave[threadIdx.x / WAVE] = incl;
        __syncthreads();
        uint32_t pos = incl - mine;
        for (uint32_t w = 0; w < threadIdx.x / WAVE; w++) pos += s_wave[w];
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            if (len[k]) {
                const uint64_t v = bits[k] << (64u - len[k]);  // left-aligned
                const uint32_t word = pos >> 5, sh = pos & 31u;
                const uint64_t tv = v >> sh;
                atomicOr(&s_stage[word], (uint32_t)(tv >> 32));
                if ((uint32_t)tv) atomicOr(&s_stage[word + 1], (uint32_t)tv);
                const uint32_t w2 = (uint32_t)(((uint64_t)(uint32_t)v << 32) >> sh);
                if (w2) atomicOr(&s_stage[word + 2], w2);
                pos += len[k];
            }
        }
        __syncthreads();
        // out: the stage shifted right by the tile's bit offset in the stream; its first and last words are shared with the neighbours
        const uint64_t w0 = bit0 >> 5;
        const uint32_t sh = (uint32_t)(bit0 & 31);
        const uint32_t now = (tbits + sh + 31) >> 5;  // words touched in the stream
        for (uint32_t j = threadIdx.x; j < now; j += 256) {
            const uint32_t hi = j ? s_stage[j - 1] : 0u, lo = j < nw ? s_stage[j] : 0u;
            const uint32_t v = sh ? (hi << (32 - sh)) | (lo >> sh) : lo;
            if (!v) continue;
            const uint32_t be = __builtin_bswap32(v);  // bytes in stream order
            if (j == 0 || j + 1 >= now) atomicOr(&out_words[w0 + j], be);
            else out_words[w0 + j] = be;
        }
        __syncthreads();
    }
}

int szk_launch_stock_huff_encode(const uint16_t *d_em, uint64_t n, const uint8_t *d_clen, const uint64_t *d_cbits, uint32_t *d_tile_bits, uint64_t *d_tile_base,
                                 uint32_t *d_out_words, uint64_t out_words_cap, uint64_t *total_bits, hipStream_t s) {
    const uint64_t ntiles = (n + STOCK_ENC_TILE - 1) / STOCK_ENC_TILE;
    const uint32_t g = (uint32_t)(ntiles < 8192 ? ntiles : 8192);
    hipLaunchKernelGGL(k_stock_enc_bits, dim3(g), dim3(256), 0, s, d_em, n, d_clen, d_tile_bits);
    hipLaunchKernelGGL(k_stock_scan32, dim3(1), dim3(1024), 0, s, d_tile_bits, ntiles, d_tile_base);
    uint64_t tb = 0;
    if (hipMemcpyAsync(&tb, d_tile_base + ntiles, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
    *total_bits = tb;
    const uint64_t words = (tb + 31) / 32;
    if (words + 2 > out_words_cap) return -2;
    if (hipMemsetAsync(d_out_words, 0, (words + 2) * 4, s) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_stock_enc_pack, dim3(g), dim3(256), 0, s, d_em, n, d_clen, d_cbits, d_tile_base, d_out_words);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// scratch: start / last_start / next_start / out_base: (nsub + 2) u64 each, count: nsub u32, flag: 2 u32. Returns 0, -3 when the counts do
// not add up to n, -4 when the subsequences' starts have not settled after STOCK_SYNC_MAX_PASSES passes (each a launch and a host
// synchronisation; an honest stream settles in two or three — a code built to synchronise slowly could ask for one per subsequence)
#define STOCK_SYNC_MAX_PASSES 12
int szk_launch_stock_huff_decode(const szk_stock_tree_dev *tr, const uint32_t *d_words, uint64_t nbytes, uint64_t n, uint64_t *d_start, uint64_t *d_last, uint64_t *d_next,
                                 uint64_t *d_base, uint32_t *d_count, uint32_t *d_flags, uint16_t *d_em, int *passes, hipStream_t s) {
    const uint64_t total_bits = nbytes * 8, nwords = (nbytes + 3) / 4;
    const uint64_t nsub = (total_bits + STOCK_SUB_BITS - 1) / STOCK_SUB_BITS;
    if (nsub == 0) return n == 0 ? 0 : -3;
    const uint32_t g = (uint32_t)((nsub + 255) / 256 < 4096 ? (nsub + 255) / 256 : 4096);
    // first guess: every subsequence starts at its own first bit
    std::vector<uint64_t> h((size_t)nsub + 2);
    for (uint64_t i = 0; i <= nsub; i++) h[i] = i * STOCK_SUB_BITS < total_bits ? i * STOCK_SUB_BITS : total_bits;
    if (hipMemcpyAsync(d_start, h.data(), (nsub + 1) * 8, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    if (hipMemcpyAsync(d_next, h.data(), (nsub + 1) * 8, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    if (hipMemsetAsync(d_last, 0xFF, (nsub + 1) * 8, s) != hipSuccess) return -1;
    if (hipMemsetAsync(d_count, 0, nsub * 4, s) != hipSuccess) return -1;
    int it = 0;
    for (;; it++) {
        if (it >= STOCK_SYNC_MAX_PASSES || (uint64_t)it > nsub + 1) return -4;  // (not settled: the caller decodes on the host instead)
        if (hipMemsetAsync(d_flags, 0, 8, s) != hipSuccess) return -1;
        hipLaunchKernelGGL(k_stock_huff_sync, dim3(g), dim3(256), 0, s, *tr, d_words, nwords, total_bits, nsub, d_start, d_last, d_next, d_count, d_flags);
        uint32_t changed = 0;
        if (hipMemcpyAsync(&changed, d_flags, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
        if (!changed) break;
        if (hipMemcpyAsync(d_start, d_next, (nsub + 1) * 8, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
    }
    if (passes) *passes = it + 1;
    hipLaunchKernelGGL(k_stock_scan32, dim3(1), dim3(1024), 0, s, d_count, nsub, d_base);
    uint64_t total = 0;
    if (hipMemcpyAsync(&total, d_base + nsub, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
    if (total < n) return -3;  // (more than n: the last byte's padding decoded as symbols — the write pass stops at n)
    hipLaunchKernelGGL(k_stock_huff_write, dim3(g), dim3(256), 0, s, *tr, d_words, nwords, total_bits, nsub, d_start, d_base, n, d_em, d_flags + 1);
    uint32_t bad = 0;
    if (hipMemcpyAsync(&bad, d_flags + 1, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
    return bad ? -3 : 0;
}

// ------------------------------------------------------------------------------------------------------------
// Stock ALGO_LORENZO_REG streams, read side (round 4). BlockwiseDecomposition::decompress (decomposition/BlockwiseDecomposition.hpp:48-67)
// walks the blocks in raster order and predicts every element from RECONSTRUCTED values in T arithmetic (LorenzoPredictor.hpp:60-95,
// RegressionPredictor.hpp:77-92), then LinearQuantizer::recover (:77-86): nothing of it is associative, so the values must be made in
// an order that respects the reference's dependencies — a block after its low neighbours (fronts of blocks, a launch each), an element
// after its low neighbours (hyperplanes i + j + k = s inside the block, a wave per block, the block and two halo layers in LDS).
// 1-D arrays are a single chain: one wave walks the blocks, one lane the elements of a Lorenzo block (k_slr_chain: correct, and as slow
// as a chain is). The predictions spell the reference's order of terms (prev3(d, ds, k, j, i) steps k along y, j along z, i along x:
// LorenzoPredictor.hpp:102-108 with ds = the strides slowest first).
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T slr_value(const szk_slr_params &p, T pred, uint64_t c) {  // LinearQuantizer::recover
    const uint32_t code = p.codes[c];
    if (code) return ref_recover(pred, (int)code, p.eb, (int)p.radius);
    const uint64_t k = stock_ordinal(p.codes, p.tile_base, c);
    if (k >= p.n_unpred) {
        *p.bad = 1u;
        return (T)0;
    }
    return reinterpret_cast<const T *>(p.unpred)[k];
}
__device__ __forceinline__ int slr_w(int j) { return j == 1 ? -2 : 1; }  // (1, -2, 1)
template <typename T, int N>
__global__ __launch_bounds__(256) void k_slr_front(szk_slr_params p, uint32_t diag) {
    constexpr uint32_t MAXT = N == 3 ? 10u * 10u * 10u : 34u * 34u;  // (B + 2)^N
    __shared__ T s_t[4][MAXT];
    const int lane = lane_id();
    T *tl = s_t[threadIdx.x / WAVE];
    const uint32_t cand = blockIdx.x * 4 + threadIdx.x / WAVE;  // candidates: (bz, by) pairs in 3-D, by in 2-D; bx follows from the front
    uint32_t bz = 0, by, bx;
    if (N == 3) {
        if (cand >= p.nb[0] * p.nb[1]) return;
        bz = cand / p.nb[1];
        by = cand - bz * p.nb[1];
    } else {
        if (cand >= p.nb[1]) return;
        by = cand;
    }
    if (bz + by > diag) return;
    bx = diag - bz - by;
    if (bx >= p.nb[2]) return;
    const uint32_t task = (bz * p.nb[1] + by) * p.nb[2] + bx;
    const uint32_t oz = bz * p.B, oy = by * p.B, ox = bx * p.B;
    const uint32_t ez = N == 3 ? min(p.B, (uint32_t)p.d[0] - oz) : 1u, ey = min(p.B, (uint32_t)p.d[1] - oy), ex = min(p.B, (uint32_t)p.d[2] - ox);
    const uint64_t coff = (uint64_t)oz * p.d[1] * p.d[2] + (uint64_t)ez * ((uint64_t)oy * p.d[2] + (uint64_t)ey * ox);
    const uint32_t nown = ez * ey * ex;
    const uint32_t hz = N == 3 ? 2u : 0u;  // halo layers along z
    const uint32_t tz = ez + hz, ty = ey + 2, tx = ex + 2;
    auto at = [&](uint32_t a, uint32_t b, uint32_t c) -> uint32_t { return (a * ty + b) * tx + c; };
    T *out = reinterpret_cast<T *>(p.out);
    for (uint32_t l = lane; l < tz * ty * tx; l += WAVE) {  // the halo: finished values, zeros outside the array (the reference's padding)
        const uint32_t c = l % tx, b = (l / tx) % ty, a = l / (tx * ty);
        if (a >= hz && b >= 2 && c >= 2) continue;
        const int64_t z = (int64_t)oz + a - hz, y = (int64_t)oy + b - 2, x = (int64_t)ox + c - 2;
        tl[l] = (z >= 0 && y >= 0 && x >= 0) ? out[((uint64_t)z * p.d[1] + (uint64_t)y) * p.d[2] + (uint64_t)x] : (T)0;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const uint32_t kind = p.kind[task];
    if (kind == 2) {  // regression: nothing of the neighbours
        const T *cf = reinterpret_cast<const T *>(p.coef) + (uint64_t)task * 4;
        for (uint32_t t = lane; t < nown; t += WAVE) {
            const uint32_t i2 = t % ex, i1 = (t / ex) % ey, i0 = t / (ex * ey);
            T pr;
            if (N == 3) {
                pr = (T)(cf[0] * (T)i0);
                pr = (T)(pr + (T)(cf[1] * (T)i1));
                pr = (T)(pr + (T)(cf[2] * (T)i2));
                pr = (T)(pr + cf[3]);
            } else {
                pr = (T)(cf[0] * (T)i1);
                pr = (T)(pr + (T)(cf[1] * (T)i2));
                pr = (T)(pr + cf[2]);
            }
            tl[at(i0 + hz, i1 + 2, i2 + 2)] = slr_value<T>(p, pr, coff + t);
        }
    } else {
        const uint32_t smax = (ez - 1) + (ey - 1) + (ex - 1);
        for (uint32_t s = 0; s <= smax; s++) {
            for (uint32_t t = lane; t < nown; t += WAVE) {
                const uint32_t i2 = t % ex, i1 = (t / ex) % ey, i0 = t / (ex * ey);
                if (i0 + i1 + i2 != s) continue;
                const uint32_t a = i0 + hz, b = i1 + 2, c = i2 + 2;
                T pr;
                if (N == 3) {
                    auto P = [&](int k, int j, int i) -> T { return tl[at(a - j, b - k, c - i)]; };
                    if (kind == 0) {
                        pr = (T)(P(0, 0, 1) + P(0, 1, 0));
                        pr = (T)(pr + P(1, 0, 0));
                        pr = (T)(pr - P(0, 1, 1));
                        pr = (T)(pr - P(1, 0, 1));
                        pr = (T)(pr - P(1, 1, 0));
                        pr = (T)(pr + P(1, 1, 1));
                    } else {
                        pr = 0;
                        bool first = true;
                        for (int k = 0; k <= 2; k++)
                            for (int j = 0; j <= 2; j++)
                                for (int i = 0; i <= 2; i++) {
                                    if ((k | j | i) == 0) continue;
                                    const T term = (T)((T)(-(slr_w(k) * slr_w(j) * slr_w(i))) * P(k, j, i));
                                    pr = first ? term : (T)(pr + term);
                                    first = false;
                                }
                    }
                } else {
                    auto P = [&](int j, int i) -> T { return tl[at(a, b - j, c - i)]; };
                    if (kind == 0) {
                        pr = (T)((T)(P(0, 1) + P(1, 0)) - P(1, 1));
                    } else {
                        pr = 0;
                        bool first = true;
                        for (int j = 0; j <= 2; j++)
                            for (int i = 0; i <= 2; i++) {
                                if ((j | i) == 0) continue;
                                const T term = (T)((T)(-(slr_w(j) * slr_w(i))) * P(j, i));
                                pr = first ? term : (T)(pr + term);
                                first = false;
                            }
                    }
                }
                tl[at(a, b, c)] = slr_value<T>(p, pr, coff + t);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t t = lane; t < nown; t += WAVE) {
        const uint32_t i2 = t % ex, i1 = (t / ex) % ey, i0 = t / (ex * ey);
        out[((uint64_t)(oz + i0) * p.d[1] + (oy + i1)) * p.d[2] + (ox + i2)] = tl[at(i0 + hz, i1 + 2, i2 + 2)];
    }
}
// 4-D arrays (round 5): the same walk over four block indices — fronts bw + bz + by + bx, hyperplanes i0 + i1 + i2 + i3 inside a block,
// one low halo layer (first-order Lorenzo: LorenzoPredictor.hpp:69-74, fifteen terms in the reference's order; the reference defines no
// second-order member for N = 4 — its predict() returns 0 there, :92-94 — and a regression block has five coefficients, :86-88).
template <typename T>
__global__ __launch_bounds__(256) void k_slr_front4(szk_slr_params p, uint32_t diag) {
    constexpr uint32_t MAXT = 9u * 9u * 9u * 9u;  // (B + 1)^4, B <= 8
    extern __shared__ __align__(8) unsigned char s_raw[];
    const int lane = lane_id();
    const uint32_t B = p.B, te = B + 1;
    T *tl = reinterpret_cast<T *>(s_raw) + (size_t)(threadIdx.x / WAVE) * te * te * te * te;
    (void)MAXT;
    const uint32_t cand = blockIdx.x * 4 + threadIdx.x / WAVE;  // (bw, bz, by); bx follows from the front
    if (cand >= p.nbw * p.nb[0] * p.nb[1]) return;
    const uint32_t by = cand % p.nb[1], bz = (cand / p.nb[1]) % p.nb[0], bw = cand / (p.nb[1] * p.nb[0]);
    if (bw + bz + by > diag) return;
    const uint32_t bx = diag - bw - bz - by;
    if (bx >= p.nb[2]) return;
    const uint32_t task = ((bw * p.nb[0] + bz) * p.nb[1] + by) * p.nb[2] + bx;
    const uint32_t ow = bw * B, oz = bz * B, oy = by * B, ox = bx * B;
    const uint32_t ew = min(B, (uint32_t)p.dw - ow), ez = min(B, (uint32_t)p.d[0] - oz), ey = min(B, (uint32_t)p.d[1] - oy), ex = min(B, (uint32_t)p.d[2] - ox);
    const uint64_t vol3 = p.d[0] * p.d[1] * p.d[2];
    const uint64_t coff = (uint64_t)ow * vol3 + (uint64_t)ew * ((uint64_t)oz * p.d[1] * p.d[2] + (uint64_t)ez * ((uint64_t)oy * p.d[2] + (uint64_t)ey * ox));
    const uint32_t nown = ew * ez * ey * ex;
    const uint32_t tw = ew + 1, tz = ez + 1, ty = ey + 1, tx = ex + 1;
    auto at = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) -> uint32_t { return ((a * tz + b) * ty + c) * tx + d; };
    T *out = reinterpret_cast<T *>(p.out);
    for (uint32_t l = lane; l < tw * tz * ty * tx; l += WAVE) {  // the halo: finished values, zeros outside the array
        const uint32_t d = l % tx, c = (l / tx) % ty, b = (l / (tx * ty)) % tz, a = l / (tx * ty * tz);
        if (a >= 1 && b >= 1 && c >= 1 && d >= 1) continue;
        const int64_t w = (int64_t)ow + a - 1, z = (int64_t)oz + b - 1, y = (int64_t)oy + c - 1, x = (int64_t)ox + d - 1;
        tl[l] = (w >= 0 && z >= 0 && y >= 0 && x >= 0) ? out[(((uint64_t)w * p.d[0] + (uint64_t)z) * p.d[1] + (uint64_t)y) * p.d[2] + (uint64_t)x] : (T)0;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const uint32_t kind = p.kind[task];
    auto split = [&](uint32_t t, uint32_t &i0, uint32_t &i1, uint32_t &i2, uint32_t &i3) {
        i3 = t % ex;
        i2 = (t / ex) % ey;
        i1 = (t / (ex * ey)) % ez;
        i0 = t / (ex * ey * ez);
    };
    if (kind == 2) {
        const T *cf = reinterpret_cast<const T *>(p.coef) + (uint64_t)task * 8;
        for (uint32_t t = lane; t < nown; t += WAVE) {
            uint32_t i0, i1, i2, i3;
            split(t, i0, i1, i2, i3);
            T pr = (T)(cf[0] * (T)i0);
            pr = (T)(pr + (T)(cf[1] * (T)i1));
            pr = (T)(pr + (T)(cf[2] * (T)i2));
            pr = (T)(pr + (T)(cf[3] * (T)i3));
            pr = (T)(pr + cf[4]);
            tl[at(i0 + 1, i1 + 1, i2 + 1, i3 + 1)] = slr_value<T>(p, pr, coff + t);
        }
    } else {
        const uint32_t smax = (ew - 1) + (ez - 1) + (ey - 1) + (ex - 1);
        for (uint32_t s = 0; s <= smax; s++) {
            for (uint32_t t = lane; t < nown; t += WAVE) {
                uint32_t i0, i1, i2, i3;
                split(t, i0, i1, i2, i3);
                if (i0 + i1 + i2 + i3 != s) continue;
                const uint32_t a = i0 + 1, b = i1 + 1, c = i2 + 1, d = i3 + 1;
                T pr = 0;
                if (kind == 0) {
                    // prev4(d, ds, t, k, j, i) = *(d - (t ds[2] + k ds[1] + j ds[0] + i)), ds = the strides slowest first: t steps along y, k
                    // along z, j along w, i along x (LorenzoPredictor.hpp:69-74, 109-111)
                    auto P = [&](int tt, int k, int j, int i) -> T { return tl[at(a - j, b - k, c - tt, d - i)]; };
                    pr = (T)(P(0, 0, 0, 1) + P(0, 0, 1, 0));
                    pr = (T)(pr - P(0, 0, 1, 1));
                    pr = (T)(pr + P(0, 1, 0, 0));
                    pr = (T)(pr - P(0, 1, 0, 1));
                    pr = (T)(pr - P(0, 1, 1, 0));
                    pr = (T)(pr + P(0, 1, 1, 1));
                    pr = (T)(pr + P(1, 0, 0, 0));
                    pr = (T)(pr - P(1, 0, 0, 1));
                    pr = (T)(pr - P(1, 0, 1, 0));
                    pr = (T)(pr + P(1, 0, 1, 1));
                    pr = (T)(pr - P(1, 1, 0, 0));
                    pr = (T)(pr + P(1, 1, 0, 1));
                    pr = (T)(pr + P(1, 1, 1, 0));
                    pr = (T)(pr - P(1, 1, 1, 1));
                }  // (kind 1: the reference's second-order member predicts 0 for N = 4)
                tl[at(a, b, c, d)] = slr_value<T>(p, pr, coff + t);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t t = lane; t < nown; t += WAVE) {
        uint32_t i0, i1, i2, i3;
        split(t, i0, i1, i2, i3);
        out[(((uint64_t)(ow + i0) * p.d[0] + (oz + i1)) * p.d[1] + (oy + i2)) * p.d[2] + (ox + i3)] = tl[at(i0 + 1, i1 + 1, i2 + 1, i3 + 1)];
    }
}
// 1-D: the chain. One workgroup of sixteen waves works through the array in rounds of sixteen UNITS (a unit: a block, or 256 elements
// of a longer one). What does not depend on the chain is made by all waves first, a unit each: the quantizer's term 2 (code - radius) eb
// in double, the unpredictable values of the zero codes (ordinals from the tile scan), a regression block's values outright. Then ONE
// lane walks the round's Lorenzo units in order — convert, add, convert back per element — and all waves store. (First versions: one
// wave doing everything block by block, 16 us of exposed memory latency per block: 528 ms for 2^22 values.)
#define SLR_UNIT 256u
template <typename T>
__global__ __launch_bounds__(1024) void k_slr_chain(szk_slr_params p) {
    __shared__ double s_add[16][SLR_UNIT];
    __shared__ T s_v[16][SLR_UNIT];
    __shared__ uint8_t s_z[16][SLR_UNIT];
    __shared__ uint32_t s_m[16], s_kind[16];
    __shared__ T s_pair[2];
    const int lane = lane_id();
    const uint32_t w = threadIdx.x / WAVE;
    const uint32_t n = (uint32_t)p.d[2];
    T *out = reinterpret_cast<T *>(p.out);
    const T *un = reinterpret_cast<const T *>(p.unpred);
    const uint32_t upb = (p.B + SLR_UNIT - 1) / SLR_UNIT;  // units per block
    const uint64_t nunits = (uint64_t)p.nb[2] * upb;
    if (threadIdx.x == 0) s_pair[0] = s_pair[1] = 0;  // the two values left of the next element (zeros in front of the array)
    __syncthreads();
    for (uint64_t u0 = 0; u0 < nunits; u0 += 16) {
        const uint64_t u = u0 + w;
        uint32_t m = 0, kind = 0, x0 = 0;
        if (u < nunits) {
            const uint32_t task = (uint32_t)(u / upb), seg = (uint32_t)(u - (uint64_t)task * upb);
            const uint32_t ox = task * p.B, ex = min(p.B, n - ox);
            const uint32_t t0 = seg * SLR_UNIT;
            if (t0 < ex) {
                m = min(SLR_UNIT, ex - t0);
                x0 = ox + t0;
                kind = p.kind[task];
                const T *cf = reinterpret_cast<const T *>(p.coef) + (uint64_t)task * 4;
                const T c0 = cf[0], c1 = cf[1];
                for (uint32_t t = lane; t < m; t += WAVE) {
                    const uint32_t code = p.codes[x0 + t];
                    T uv = 0;
                    if (code == 0) {
                        const uint64_t zi = stock_ordinal(p.codes, p.tile_base, (uint64_t)x0 + t);
                        if (zi < p.n_unpred) uv = un[zi];
                        else *p.bad = 1u;
                    }
                    const double add = (double)(2 * ((int)code - (int)p.radius)) * p.eb;
                    s_z[w][t] = code == 0;
                    s_add[w][t] = add;
                    if (kind == 2) {  // regression: nothing of the chain
                        const T pr = (T)((T)(c0 * (T)(t0 + t)) + c1);
                        s_v[w][t] = code ? (T)((double)pr + add) : uv;
                    } else {
                        s_v[w][t] = uv;
                    }
                }
            }
        }
        if (lane == 0) {
            s_m[w] = m;
            s_kind[w] = kind;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            T p1 = s_pair[0], p2 = s_pair[1];
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t mk = s_m[k], kk = s_kind[k];
                if (!mk) continue;
                if (kk == 2) {
                    p2 = mk > 1 ? s_v[k][mk - 2] : p1;
                    p1 = s_v[k][mk - 1];
                } else {
                    // eight elements' operands out of LDS at once (a lone lane waits out every read it issues one by one: 235 cycles
                    // per element with three reads inside the step), then eight steps in registers
                    for (uint32_t tb = 0; tb < mk; tb += 8) {
                        double a8[8];
                        T u8[8];
                        uint8_t z8[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const uint32_t t = tb + j < mk ? tb + j : mk - 1;
                            a8[j] = s_add[k][t];
                            u8[j] = s_v[k][t];
                            z8[j] = s_z[k][t];
                        }
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const bool live = tb + j < mk;  // (no early exit: the arrays stay in registers)
                            const T pr = kk == 1 ? (T)((T)(2 * p1) - p2) : p1;
                            const T v = z8[j] ? u8[j] : (T)((double)pr + a8[j]);
                            u8[j] = v;
                            p2 = live ? p1 : p2;
                            p1 = live ? v : p1;
                        }
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if (tb + j < mk) s_v[k][tb + j] = u8[j];
                    }
                }
            }
            s_pair[0] = p1;
            s_pair[1] = p2;
        }
        __syncthreads();
        for (uint32_t t = lane; t < m; t += WAVE) out[x0 + t] = s_v[w][t];
        __syncthreads();
    }
}
// ------------------------------------------------------------------------------------------------------------
// Stock ALGO_LORENZO_REG streams, WRITE side (round 5; szk_slw_params in sz3hip_kernels.h says what is the writer's to choose).
// k_slw_select: a wave per block, the block and two low halo layers of ORIGINAL values in LDS; the regression member's fit
// (RegressionPredictor::precompress, :28-55: sums in double, coefficients stored in T), the members' sampled error estimates
// (ComposedPredictor::precompress :25-40 over BlockwiseIterator's sample points :151-184, LorenzoPredictor::estimate_error with its
// noise term :17-38), the first minimum. k_slw_front: the coding, front by front of blocks — k_slr_front run forward.
// Round 6: the selection is repeated behind the coding pass (szk_slw_params::reselect: the halo then holds the values as the reader will have
// them, the reference's view of a block's lower neighbours) and the pass with it while a choice moves — sz3hip_host.cpp,
// stock_encode_lorenzo_reg: a vector the repetition leaves alone is the reference's own.
// ------------------------------------------------------------------------------------------------------------
template <typename T, int N, int L>
__device__ __forceinline__ T slw_lorenzo(const T *tl, uint32_t ty, uint32_t tx, uint32_t a, uint32_t b, uint32_t c) {  // LorenzoPredictor::predict, :60-95
    auto at = [&](uint32_t z, uint32_t y, uint32_t x) -> uint32_t { return (z * ty + y) * tx + x; };
    T pr;
    if (N == 3) {
        auto P = [&](int k, int j, int i) -> T { return tl[at(a - j, b - k, c - i)]; };
        if (L == 1) {
            pr = (T)(P(0, 0, 1) + P(0, 1, 0));
            pr = (T)(pr + P(1, 0, 0));
            pr = (T)(pr - P(0, 1, 1));
            pr = (T)(pr - P(1, 0, 1));
            pr = (T)(pr - P(1, 1, 0));
            pr = (T)(pr + P(1, 1, 1));
        } else {
            pr = 0;
            bool first = true;
            for (int k = 0; k <= 2; k++)
                for (int j = 0; j <= 2; j++)
                    for (int i = 0; i <= 2; i++) {
                        if ((k | j | i) == 0) continue;
                        const T term = (T)((T)(-(slr_w(k) * slr_w(j) * slr_w(i))) * P(k, j, i));
                        pr = first ? term : (T)(pr + term);
                        first = false;
                    }
        }
    } else {
        auto P = [&](int j, int i) -> T { return tl[at(a, b - j, c - i)]; };
        if (L == 1) {
            pr = (T)((T)(P(0, 1) + P(1, 0)) - P(1, 1));
        } else {
            pr = 0;
            bool first = true;
            for (int j = 0; j <= 2; j++)
                for (int i = 0; i <= 2; i++) {
                    if ((j | i) == 0) continue;
                    const T term = (T)((T)(-(slr_w(j) * slr_w(i))) * P(j, i));
                    pr = first ? term : (T)(pr + term);
                    first = false;
                }
        }
    }
    return pr;
}
template <typename T, int N>
__device__ __forceinline__ T slw_regression(const T *cf, uint32_t i0, uint32_t i1, uint32_t i2) {  // RegressionPredictor::predict, :81-92
    T pr;
    if (N == 3) {
        pr = (T)(cf[0] * (T)i0);
        pr = (T)(pr + (T)(cf[1] * (T)i1));
        pr = (T)(pr + (T)(cf[2] * (T)i2));
        pr = (T)(pr + cf[3]);
    } else {
        pr = (T)(cf[0] * (T)i1);
        pr = (T)(pr + (T)(cf[1] * (T)i2));
        pr = (T)(pr + cf[2]);
    }
    return pr;
}
struct SlwGeom {
    uint32_t task, oz, oy, ox, ez, ey, ex, nown, hz, tz, ty, tx;
    uint64_t coff;
};
template <int N>
__device__ __forceinline__ void slw_geom(const uint64_t (&d)[3], const uint32_t (&nb)[3], uint32_t B, uint32_t bz, uint32_t by, uint32_t bx, SlwGeom &g) {
    g.task = (bz * nb[1] + by) * nb[2] + bx;
    g.oz = bz * B;
    g.oy = by * B;
    g.ox = bx * B;
    g.ez = N == 3 ? min(B, (uint32_t)d[0] - g.oz) : 1u;
    g.ey = min(B, (uint32_t)d[1] - g.oy);
    g.ex = min(B, (uint32_t)d[2] - g.ox);
    g.coff = (uint64_t)g.oz * d[1] * d[2] + (uint64_t)g.ez * ((uint64_t)g.oy * d[2] + (uint64_t)g.ey * g.ox);
    g.nown = g.ez * g.ey * g.ex;
    g.hz = N == 3 ? 2u : 0u;
    g.tz = g.ez + g.hz;
    g.ty = g.ey + 2;
    g.tx = g.ex + 2;
}
// The reference's fit runs on x86, and where a block holds NaN or Inf its coefficients are NaN whose SIGN is the instruction set's: an invalid
// operation (0 * Inf, Inf - Inf) makes the negative "default" NaN, an operation with one NaN operand returns that operand, with two the first
// (the accumulator of `sum += x`, the minuend of a subtraction). gfx950 makes positive NaNs and has rules of its own — and a NaN coefficient is
// stored as it is (unpredictable), so the sign is in the file. The fit's operations spelled out with x86's rules (finite results are the
// plain operation's, bit for bit): tools/r6/nan_sign_1d.py shows the one kind that differed on the host — Inf at a block's first element
// (0 * Inf: negative), a NaN later (positive): the sum keeps its own.
template <int OP> __device__ __forceinline__ double x86_op(double a, double b) {  // 0 add, 1 sub, 2 mul, 3 div; a = the first (destination) operand
    if (a != a) return a;
    if (b != b) return b;
    const double r = OP == 0 ? a + b : OP == 1 ? a - b : OP == 2 ? a * b : a / b;
    return r != r ? __longlong_as_double((long long)0xFFF8000000000000ull) : r;
}
template <typename T> __device__ __forceinline__ double x86_to_double(T v) {  // (cvtss2sd keeps a NaN's sign)
    if (v != v) return __longlong_as_double((long long)(signbit(v) ? 0xFFF8000000000000ull : 0x7FF8000000000000ull));
    return (double)v;
}
template <typename T> __device__ __forceinline__ T x86_from_double(double v);
template <> __device__ __forceinline__ float x86_from_double<float>(double v) {
    if (v != v) return __uint_as_float(signbit(v) ? 0xFFC00000u : 0x7FC00000u);
    return (float)v;
}
template <> __device__ __forceinline__ double x86_from_double<double>(double v) { return v; }
template <typename T> __device__ __forceinline__ T x86_index_times(uint32_t ix, T v) {  // index[i] * (*c): a product in T; the index is never NaN
    if (v != v) return v;
    const T r = (T)ix * v;
    if (r != r) return x86_from_double<T>(__longlong_as_double((long long)0xFFF8000000000000ull));
    return r;
}
// coefficient i of RegressionPredictor.hpp:48-53 from its sums
template <typename T> __device__ __forceinline__ T x86_slope(double sm, double sn, double dm, double num) {
    double v = x86_op<3>(x86_op<2>(2.0, sm), dm - 1);
    v = x86_op<1>(v, sn);
    v = x86_op<3>(x86_op<3>(x86_op<2>(v, 6.0), num), dm + 1);
    return x86_from_double<T>(v);
}
template <typename T> __device__ __forceinline__ T x86_intercept_step(T cn, T ci, double dm) {  // current_coeffs[N] -= (dims[i] - 1) * current_coeffs[i] / 2
    return x86_from_double<T>(x86_op<1>(x86_to_double(cn), x86_op<3>(x86_op<2>(dm - 1, x86_to_double(ci)), 2.0)));
}
template <typename T, int N>
__global__ __launch_bounds__(256) void k_slw_select(szk_slw_params p) {
    constexpr uint32_t MAXT = N == 3 ? 10u * 10u * 10u : 34u * 34u;
    __shared__ T s_t[4][MAXT];
    const int lane = lane_id();
    T *tl = s_t[threadIdx.x / WAVE];
    const uint32_t nblocks = p.nb[0] * p.nb[1] * p.nb[2];
    const uint32_t task = blockIdx.x * 4 + threadIdx.x / WAVE;
    if (task >= nblocks) return;
    const uint32_t bx = task % p.nb[2], by = (task / p.nb[2]) % p.nb[1], bz = task / (p.nb[2] * p.nb[1]);
    SlwGeom g;
    slw_geom<N>(p.d, p.nb, p.B, bz, by, bx, g);
    auto at = [&](uint32_t a, uint32_t b, uint32_t c) -> uint32_t { return (a * g.ty + b) * g.tx + c; };
    const T *in = reinterpret_cast<const T *>(p.in);
    const T *halo = p.reselect ? reinterpret_cast<const T *>(p.recon) : in;  // (the repeated selection: the halo as the reader will have it)
    for (uint32_t l = lane; l < g.tz * g.ty * g.tx; l += WAVE) {  // original values (the block's own always), zeros outside the array (the reference's padding)
        const uint32_t c = l % g.tx, b = (l / g.tx) % g.ty, a = l / (g.tx * g.ty);
        const int64_t z = (int64_t)g.oz + a - g.hz, y = (int64_t)g.oy + b - 2, x = (int64_t)g.ox + c - 2;
        const bool own = a >= g.hz && b >= 2 && c >= 2;
        const uint64_t e = ((uint64_t)z * p.d[1] + (uint64_t)y) * p.d[2] + (uint64_t)x;
        tl[l] = (z >= 0 && y >= 0 && x >= 0) ? (own ? in[e] : halo[e]) : (T)0;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // the regression member: valid when no extent of the block is 1 (RegressionPredictor.hpp:33-37)
    const bool reg_on = (p.set_mask & 4u) != 0;
    const bool reg_valid = reg_on && (N == 3 ? g.ez > 1 : true) && g.ey > 1 && g.ex > 1;
    T cf[4] = {0, 0, 0, 0};
    if (reg_valid) {
        // the four sums one lane each, in the reference's own order (block_iter::foreach: the last index fastest). A sum of doubles commits to an
        // order as soon as one addition rounds: f32 values 2^29 apart (a spike of 1e30 among ones: tests/checks/wild_data_sweep.py), any f64 array.
        // Lane-strided partial sums + a butterfly gave another coefficient there, which shows wherever a coefficient is stored as it is (unpredictable).
        // (sum[i] += index[i] * (*c), RegressionPredictor.hpp:43: size_t * T is a product in T; in double it is another coefficient once in
        // ~10^5 blocks of f32 data: round 6, found by the byte sweep's 1-D twin of this line)
        double acc = 0;
        if (lane < 4)
            for (uint32_t i0 = 0; i0 < g.ez; i0++)
                for (uint32_t i1 = 0; i1 < g.ey; i1++)
                    for (uint32_t i2 = 0; i2 < g.ex; i2++) {
                        const T tv = tl[at(i0 + g.hz, i1 + 2, i2 + 2)];
                        const uint32_t ix = lane == 0 ? i0 : lane == 1 ? i1 : i2;
                        acc = x86_op<0>(acc, x86_to_double(lane == 3 ? tv : x86_index_times<T>(ix, tv)));
                    }
        const double s0 = __shfl(acc, 0), s1 = __shfl(acc, 1), s2 = __shfl(acc, 2), sn = __shfl(acc, 3);
        const double num = (double)g.nown;
        const double dm[3] = {(double)g.ez, (double)g.ey, (double)g.ex};
        const double sm[3] = {s0, s1, s2};
        T cn = x86_from_double<T>(x86_op<3>(sn, num));
        T ci[3] = {0, 0, 0};
        for (int i = 3 - N; i < 3; i++) {
            ci[i] = x86_slope<T>(sm[i], sn, dm[i], num);
            cn = x86_intercept_step<T>(cn, ci[i], dm[i]);
        }
        if (N == 3) {
            cf[0] = ci[0];
            cf[1] = ci[1];
            cf[2] = ci[2];
            cf[3] = cn;
        } else {
            cf[0] = ci[1];
            cf[1] = ci[2];
            cf[2] = cn;
        }
    }
    // the members' estimates at the block's sample points (2 per step of the diagonal in 2-D, 4 in 3-D), one point per lane
    const uint32_t msz = N == 3 ? min(g.ez, min(g.ey, g.ex)) : min(g.ey, g.ex);
    constexpr uint32_t PPS = N == 3 ? 4u : 2u;
    // (predict_error[i] += estimate_error(...), ComposedPredictor.hpp:30-33: a member's estimates summed in the sampling order, one lane per member)
    double eacc = 0;
    if (lane < 3 && (lane == 0 ? (p.set_mask & 1u) != 0 : lane == 1 ? (p.set_mask & 2u) != 0 : reg_valid))
        for (uint32_t q = 0; q < msz * PPS; q++) {
            const uint32_t i = q / PPS, w = q % PPS, j = msz - 1 - i;
            uint32_t i0, i1, i2;
            if (N == 3) {
                i0 = i;
                i1 = (w & 2u) ? j : i;
                i2 = (w & 1u) ? j : i;
            } else {
                i0 = 0;
                i1 = i;
                i2 = w ? j : i;
            }
            const uint32_t a = i0 + g.hz, b = i1 + 2, c = i2 + 2;
            const T v = tl[at(a, b, c)];
            if (lane == 0) eacc += (double)(T)(fabs((double)(T)(v - slw_lorenzo<T, N, 1>(tl, g.ty, g.tx, a, b, c))) + (N == 3 ? 1.22 : 0.81) * p.eb);
            else if (lane == 1) eacc += (double)(T)(fabs((double)(T)(v - slw_lorenzo<T, N, 2>(tl, g.ty, g.tx, a, b, c))) + (N == 3 ? 6.8 : 2.76) * p.eb);
            else eacc += (double)(T)fabs((double)(T)(v - slw_regression<T, N>(cf, i0, i1, i2)));
        }
    const double e1 = __shfl(eacc, 0), e2 = __shfl(eacc, 1), er = __shfl(eacc, 2);
    if (lane == 0) {
        // first minimum in the set's order (std::min_element); an invalid member counts as the largest double
        // (std::min_element's own walk: the first member is the minimum until a later one compares LESS — an estimate that is not a number,
        // a block with NaN in it, is never less and, in front, never beaten)
        const double big = 1.7976931348623157e308;
        double best = 0;
        uint32_t kind = 0, idx = 0, k = 0;
        auto member = [&](double e, uint32_t kd) {
            if (k == 0 || e < best) {
                best = e;
                kind = kd;
                idx = k;
            }
            k++;
        };
        if (p.set_mask & 1u) member(e1, 0u);
        if (p.set_mask & 2u) member(e2, 1u);
        if (p.set_mask & 4u) member(reg_valid ? er : big, 2u);
        // (a regression-only set on a block with an extent of 1: the reference falls back to Lorenzo-1; the launcher refuses such arrays)
        if (kind == 2 && !reg_valid) kind = 0;
        if (p.reselect) {  // (the fits are functions of the block's own values: unchanged)
            p.kind_new[task] = (uint8_t)kind;
            p.sel_new[task] = (uint8_t)idx;
            if (p.kind[task] != (uint8_t)kind || p.sel[task] != (uint8_t)idx) atomicAdd(p.n_changed, 1u);
            return;
        }
        p.kind[task] = (uint8_t)kind;
        p.sel[task] = (uint8_t)idx;
        T *o = reinterpret_cast<T *>(p.coef_fit) + (uint64_t)task * 4;
        o[0] = cf[0];
        o[1] = cf[1];
        o[2] = cf[2];
        o[3] = cf[3];
    }
}
template <typename T, int N>
__global__ __launch_bounds__(256) void k_slw_front(szk_slw_params p, uint32_t diag) {
    constexpr uint32_t MAXT = N == 3 ? 10u * 10u * 10u : 34u * 34u;
    __shared__ T s_t[4][MAXT];
    const int lane = lane_id();
    T *tl = s_t[threadIdx.x / WAVE];
    const uint32_t cand = blockIdx.x * 4 + threadIdx.x / WAVE;
    uint32_t bz = 0, by, bx;
    if (N == 3) {
        if (cand >= p.nb[0] * p.nb[1]) return;
        bz = cand / p.nb[1];
        by = cand - bz * p.nb[1];
    } else {
        if (cand >= p.nb[1]) return;
        by = cand;
    }
    if (bz + by > diag) return;
    bx = diag - bz - by;
    if (bx >= p.nb[2]) return;
    SlwGeom g;
    slw_geom<N>(p.d, p.nb, p.B, bz, by, bx, g);
    auto at = [&](uint32_t a, uint32_t b, uint32_t c) -> uint32_t { return (a * g.ty + b) * g.tx + c; };
    const T *in = reinterpret_cast<const T *>(p.in);
    T *recon = reinterpret_cast<T *>(p.recon);
    T *uval = reinterpret_cast<T *>(p.uval);
    for (uint32_t l = lane; l < g.tz * g.ty * g.tx; l += WAVE) {  // the halo: values as the reader will have them; the block: the caller's
        const uint32_t c = l % g.tx, b = (l / g.tx) % g.ty, a = l / (g.tx * g.ty);
        const int64_t z = (int64_t)g.oz + a - g.hz, y = (int64_t)g.oy + b - 2, x = (int64_t)g.ox + c - 2;
        const bool own = a >= g.hz && b >= 2 && c >= 2;
        const uint64_t e = ((uint64_t)z * p.d[1] + (uint64_t)y) * p.d[2] + (uint64_t)x;
        tl[l] = (z >= 0 && y >= 0 && x >= 0) ? (own ? in[e] : recon[e]) : (T)0;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const uint32_t kind = p.kind[g.task];
    const double recip = 1.0 / p.eb;
    auto code_one = [&](uint32_t t, uint32_t a, uint32_t b, uint32_t c, T pr) {
        T v = tl[at(a, b, c)];
        const T orig = v;
        const int code = ref_quantize<T>(v, pr, p.eb, recip, (int)p.radius);  // LinearQuantizer::quantize_and_overwrite, :43-71
        tl[at(a, b, c)] = v;
        p.codes[g.coff + t] = (uint16_t)code;
        if (code == 0) uval[g.coff + t] = orig;
    };
    if (kind == 2) {
        const T *cf = reinterpret_cast<const T *>(p.coef) + (uint64_t)g.task * 4;
        for (uint32_t t = lane; t < g.nown; t += WAVE) {
            const uint32_t i2 = t % g.ex, i1 = (t / g.ex) % g.ey, i0 = t / (g.ex * g.ey);
            code_one(t, i0 + g.hz, i1 + 2, i2 + 2, slw_regression<T, N>(cf, i0, i1, i2));
        }
    } else {
        const uint32_t smax = (g.ez - 1) + (g.ey - 1) + (g.ex - 1);
        for (uint32_t s = 0; s <= smax; s++) {
            for (uint32_t t = lane; t < g.nown; t += WAVE) {
                const uint32_t i2 = t % g.ex, i1 = (t / g.ex) % g.ey, i0 = t / (g.ex * g.ey);
                if (i0 + i1 + i2 != s) continue;
                const uint32_t a = i0 + g.hz, b = i1 + 2, c = i2 + 2;
                const T pr = kind == 0 ? slw_lorenzo<T, N, 1>(tl, g.ty, g.tx, a, b, c) : slw_lorenzo<T, N, 2>(tl, g.ty, g.tx, a, b, c);
                code_one(t, a, b, c, pr);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t t = lane; t < g.nown; t += WAVE) {
        const uint32_t i2 = t % g.ex, i1 = (t / g.ex) % g.ey, i0 = t / (g.ex * g.ey);
        recon[((uint64_t)(g.oz + i0) * p.d[1] + (g.oy + i1)) * p.d[2] + (g.ox + i2)] = tl[at(i0 + g.hz, i1 + 2, i2 + 2)];
    }
}
// 4-D arrays (round 5, second half): the write side of k_slr_front4. Tiles of (B + 1)^4 values with ONE low halo layer (first-order
// Lorenzo is the only member that reads neighbours: the reference's second-order member predicts 0 for N = 4, LorenzoPredictor.hpp:92-94,
// with a noise term of 0, :17-38), five regression coefficients (RegressionPredictor.hpp:28-55, 86-88), eight sample points per step of
// the block's diagonal (BlockwiseIterator.hpp:170-179), the fifteen Lorenzo terms in the reference's order (:69-74).
struct SlwGeom4 {
    uint32_t task, ow, oz, oy, ox, ew, ez, ey, ex, nown, tw, tz, ty, tx;
    uint64_t coff;
};
__device__ __forceinline__ void slw_geom4(const szk_slw_params &p, uint32_t bw, uint32_t bz, uint32_t by, uint32_t bx, SlwGeom4 &g) {
    const uint32_t B = p.B;
    g.task = ((bw * p.nb[0] + bz) * p.nb[1] + by) * p.nb[2] + bx;
    g.ow = bw * B;
    g.oz = bz * B;
    g.oy = by * B;
    g.ox = bx * B;
    g.ew = min(B, (uint32_t)p.dw - g.ow);
    g.ez = min(B, (uint32_t)p.d[0] - g.oz);
    g.ey = min(B, (uint32_t)p.d[1] - g.oy);
    g.ex = min(B, (uint32_t)p.d[2] - g.ox);
    const uint64_t vol3 = p.d[0] * p.d[1] * p.d[2];
    g.coff = (uint64_t)g.ow * vol3 + (uint64_t)g.ew * ((uint64_t)g.oz * p.d[1] * p.d[2] + (uint64_t)g.ez * ((uint64_t)g.oy * p.d[2] + (uint64_t)g.ey * g.ox));
    g.nown = g.ew * g.ez * g.ey * g.ex;
    g.tw = g.ew + 1;
    g.tz = g.ez + 1;
    g.ty = g.ey + 1;
    g.tx = g.ex + 1;
}
template <typename T>
__device__ __forceinline__ T slw_lorenzo4(const T *tl, const SlwGeom4 &g, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    // prev4(d, ds, t, k, j, i): t steps along y, k along z, j along w, i along x (LorenzoPredictor.hpp:69-74, 109-111)
    auto P = [&](int tt, int k, int j, int i) -> T { return tl[(((a - j) * g.tz + (b - k)) * g.ty + (c - tt)) * g.tx + (d - i)]; };
    T pr = (T)(P(0, 0, 0, 1) + P(0, 0, 1, 0));
    pr = (T)(pr - P(0, 0, 1, 1));
    pr = (T)(pr + P(0, 1, 0, 0));
    pr = (T)(pr - P(0, 1, 0, 1));
    pr = (T)(pr - P(0, 1, 1, 0));
    pr = (T)(pr + P(0, 1, 1, 1));
    pr = (T)(pr + P(1, 0, 0, 0));
    pr = (T)(pr - P(1, 0, 0, 1));
    pr = (T)(pr - P(1, 0, 1, 0));
    pr = (T)(pr + P(1, 0, 1, 1));
    pr = (T)(pr - P(1, 1, 0, 0));
    pr = (T)(pr + P(1, 1, 0, 1));
    pr = (T)(pr + P(1, 1, 1, 0));
    pr = (T)(pr - P(1, 1, 1, 1));
    return pr;
}
template <typename T>
__device__ __forceinline__ T slw_regression4(const T *cf, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3) {  // RegressionPredictor::predict, :86-88
    T pr = (T)(cf[0] * (T)i0);
    pr = (T)(pr + (T)(cf[1] * (T)i1));
    pr = (T)(pr + (T)(cf[2] * (T)i2));
    pr = (T)(pr + (T)(cf[3] * (T)i3));
    pr = (T)(pr + cf[4]);
    return pr;
}
template <typename T>
__device__ __forceinline__ void slw_fill4(T *tl, const SlwGeom4 &g, const szk_slw_params &p, const T *inner, const T *halo, int lane) {
    // the block's own elements from `inner`, its low halo layer from `halo`, zeros outside the array (the reference's padding)
    for (uint32_t l = lane; l < g.tw * g.tz * g.ty * g.tx; l += WAVE) {
        const uint32_t d = l % g.tx, c = (l / g.tx) % g.ty, b = (l / (g.tx * g.ty)) % g.tz, a = l / (g.tx * g.ty * g.tz);
        const bool own = a >= 1 && b >= 1 && c >= 1 && d >= 1;
        const int64_t w = (int64_t)g.ow + a - 1, z = (int64_t)g.oz + b - 1, y = (int64_t)g.oy + c - 1, x = (int64_t)g.ox + d - 1;
        const uint64_t e = (((uint64_t)w * p.d[0] + (uint64_t)z) * p.d[1] + (uint64_t)y) * p.d[2] + (uint64_t)x;
        tl[l] = (w >= 0 && z >= 0 && y >= 0 && x >= 0) ? (own ? inner[e] : halo[e]) : (T)0;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}
template <typename T>
__global__ __launch_bounds__(256) void k_slw_select4(szk_slw_params p) {
    extern __shared__ __align__(8) unsigned char s_raw4[];
    const int lane = lane_id();
    const uint32_t te = p.B + 1;
    T *tl = reinterpret_cast<T *>(s_raw4) + (size_t)(threadIdx.x / WAVE) * te * te * te * te;
    const uint32_t nblocks = p.nbw * p.nb[0] * p.nb[1] * p.nb[2];
    const uint32_t task = blockIdx.x * 4 + threadIdx.x / WAVE;
    if (task >= nblocks) return;
    const uint32_t bx = task % p.nb[2], by = (task / p.nb[2]) % p.nb[1], bz = (task / (p.nb[2] * p.nb[1])) % p.nb[0], bw = task / (p.nb[2] * p.nb[1] * p.nb[0]);
    SlwGeom4 g;
    slw_geom4(p, bw, bz, by, bx, g);
    const T *in = reinterpret_cast<const T *>(p.in);
    slw_fill4<T>(tl, g, p, in, p.reselect ? reinterpret_cast<const T *>(p.recon) : in, lane);  // (the repeated selection: the halo as the reader will have it)
    auto at = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) -> uint32_t { return ((a * g.tz + b) * g.ty + c) * g.tx + d; };
    const bool reg_on = (p.set_mask & 4u) != 0;
    const bool reg_valid = reg_on && g.ew > 1 && g.ez > 1 && g.ey > 1 && g.ex > 1;  // RegressionPredictor.hpp:33-37
    T cf[5] = {0, 0, 0, 0, 0};
    if (reg_valid) {
        double acc = 0;  // (one lane per sum, the reference's order, products in T: see k_slw_select)
        if (lane < 5)
            for (uint32_t i0 = 0; i0 < g.ew; i0++)
                for (uint32_t i1 = 0; i1 < g.ez; i1++)
                    for (uint32_t i2 = 0; i2 < g.ey; i2++)
                        for (uint32_t i3 = 0; i3 < g.ex; i3++) {
                            const T tv = tl[at(i0 + 1, i1 + 1, i2 + 1, i3 + 1)];
                            const uint32_t ix = lane == 0 ? i0 : lane == 1 ? i1 : lane == 2 ? i2 : i3;
                            acc = x86_op<0>(acc, x86_to_double(lane == 4 ? tv : x86_index_times<T>(ix, tv)));
                        }
        const double sm[4] = {__shfl(acc, 0), __shfl(acc, 1), __shfl(acc, 2), __shfl(acc, 3)}, sn = __shfl(acc, 4);
        const double num = (double)g.nown;
        const double dm[4] = {(double)g.ew, (double)g.ez, (double)g.ey, (double)g.ex};
        cf[4] = x86_from_double<T>(x86_op<3>(sn, num));
        for (int i = 0; i < 4; i++) {
            cf[i] = x86_slope<T>(sm[i], sn, dm[i], num);
            cf[4] = x86_intercept_step<T>(cf[4], cf[i], dm[i]);
        }
    }
    const uint32_t msz = min(min(g.ew, g.ez), min(g.ey, g.ex));
    double eacc = 0;  // (one lane per member, the sampling order: see k_slw_select)
    if (lane < 3 && (lane == 0 ? (p.set_mask & 1u) != 0 : lane == 1 ? (p.set_mask & 2u) != 0 : reg_valid))
        for (uint32_t q = 0; q < msz * 8u; q++) {
            const uint32_t i = q / 8u, w = q % 8u, j = msz - 1 - i;
            const uint32_t i0 = i, i1 = (w & 4u) ? j : i, i2 = (w & 2u) ? j : i, i3 = (w & 1u) ? j : i;
            const uint32_t a = i0 + 1, b = i1 + 1, c = i2 + 1, d = i3 + 1;
            const T v = tl[at(a, b, c, d)];
            if (lane == 0) eacc += (double)(T)(fabs((double)(T)(v - slw_lorenzo4<T>(tl, g, a, b, c, d))) + 1.79 * p.eb);
            else if (lane == 1) eacc += (double)(T)fabs((double)v);  // (the member predicts 0 and has no noise term for N = 4)
            else eacc += (double)(T)fabs((double)(T)(v - slw_regression4<T>(cf, i0, i1, i2, i3)));
        }
    const double e1 = __shfl(eacc, 0), e2 = __shfl(eacc, 1), er = __shfl(eacc, 2);
    if (lane == 0) {
        const double big = 1.7976931348623157e308;
        double best = 0;
        uint32_t kind = 0, idx = 0, k = 0;
        auto member = [&](double e, uint32_t kd) {  // (std::min_element's own walk, see k_slw_select)
            if (k == 0 || e < best) {
                best = e;
                kind = kd;
                idx = k;
            }
            k++;
        };
        if (p.set_mask & 1u) member(e1, 0u);
        if (p.set_mask & 2u) member(e2, 1u);
        if (p.set_mask & 4u) member(reg_valid ? er : big, 2u);
        if (kind == 2 && !reg_valid) kind = 0;  // (a regression-only set on a thin block: the launcher refuses such arrays)
        if (p.reselect) {
            p.kind_new[task] = (uint8_t)kind;
            p.sel_new[task] = (uint8_t)idx;
            if (p.kind[task] != (uint8_t)kind || p.sel[task] != (uint8_t)idx) atomicAdd(p.n_changed, 1u);
            return;
        }
        p.kind[task] = (uint8_t)kind;
        p.sel[task] = (uint8_t)idx;
        T *o = reinterpret_cast<T *>(p.coef_fit) + (uint64_t)task * 8;
        for (int i = 0; i < 5; i++) o[i] = cf[i];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_slw_front4(szk_slw_params p, uint32_t diag) {
    extern __shared__ __align__(8) unsigned char s_raw4[];
    const int lane = lane_id();
    const uint32_t te = p.B + 1;
    T *tl = reinterpret_cast<T *>(s_raw4) + (size_t)(threadIdx.x / WAVE) * te * te * te * te;
    const uint32_t cand = blockIdx.x * 4 + threadIdx.x / WAVE;  // (bw, bz, by); bx follows from the front
    if (cand >= p.nbw * p.nb[0] * p.nb[1]) return;
    const uint32_t by = cand % p.nb[1], bz = (cand / p.nb[1]) % p.nb[0], bw = cand / (p.nb[1] * p.nb[0]);
    if (bw + bz + by > diag) return;
    const uint32_t bx = diag - bw - bz - by;
    if (bx >= p.nb[2]) return;
    SlwGeom4 g;
    slw_geom4(p, bw, bz, by, bx, g);
    const T *in = reinterpret_cast<const T *>(p.in);
    T *recon = reinterpret_cast<T *>(p.recon);
    T *uval = reinterpret_cast<T *>(p.uval);
    slw_fill4<T>(tl, g, p, in, recon, lane);  // the halo: values as the reader will have them; the block: the caller's
    auto at = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) -> uint32_t { return ((a * g.tz + b) * g.ty + c) * g.tx + d; };
    auto split = [&](uint32_t t, uint32_t &i0, uint32_t &i1, uint32_t &i2, uint32_t &i3) {
        i3 = t % g.ex;
        i2 = (t / g.ex) % g.ey;
        i1 = (t / (g.ex * g.ey)) % g.ez;
        i0 = t / (g.ex * g.ey * g.ez);
    };
    const uint32_t kind = p.kind[g.task];
    const double recip = 1.0 / p.eb;
    auto code_one = [&](uint32_t t, uint32_t l, T pr) {
        T v = tl[l];
        const T orig = v;
        const int code = ref_quantize<T>(v, pr, p.eb, recip, (int)p.radius);  // LinearQuantizer::quantize_and_overwrite, :43-71
        tl[l] = v;
        p.codes[g.coff + t] = (uint16_t)code;
        if (code == 0) uval[g.coff + t] = orig;
    };
    if (kind != 0) {  // regression, or the second-order member's prediction of 0: nothing of the neighbours
        const T *cf = reinterpret_cast<const T *>(p.coef) + (uint64_t)g.task * 8;
        for (uint32_t t = lane; t < g.nown; t += WAVE) {
            uint32_t i0, i1, i2, i3;
            split(t, i0, i1, i2, i3);
            code_one(t, at(i0 + 1, i1 + 1, i2 + 1, i3 + 1), kind == 2 ? slw_regression4<T>(cf, i0, i1, i2, i3) : (T)0);
        }
    } else {
        const uint32_t smax = (g.ew - 1) + (g.ez - 1) + (g.ey - 1) + (g.ex - 1);
        for (uint32_t s = 0; s <= smax; s++) {
            for (uint32_t t = lane; t < g.nown; t += WAVE) {
                uint32_t i0, i1, i2, i3;
                split(t, i0, i1, i2, i3);
                if (i0 + i1 + i2 + i3 != s) continue;
                code_one(t, at(i0 + 1, i1 + 1, i2 + 1, i3 + 1), slw_lorenzo4<T>(tl, g, i0 + 1, i1 + 1, i2 + 1, i3 + 1));
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t t = lane; t < g.nown; t += WAVE) {
        uint32_t i0, i1, i2, i3;
        split(t, i0, i1, i2, i3);
        recon[(((uint64_t)(g.ow + i0) * p.d[0] + (g.oz + i1)) * p.d[1] + (g.oy + i2)) * p.d[2] + (g.ox + i3)] = tl[at(i0 + 1, i1 + 1, i2 + 1, i3 + 1)];
    }
}
// the codes' histogram (a window of 2048 bins around the radius in LDS, the rest straight to memory) ...
__global__ __launch_bounds__(256) void k_slw_hist(const uint16_t *__restrict__ codes, uint64_t n, uint32_t radius, unsigned long long *__restrict__ hist) {
    __shared__ uint32_t s_h[2048];
    for (uint32_t i = threadIdx.x; i < 2048; i += 256) s_h[i] = 0;
    __syncthreads();
    const uint32_t lo = radius > 1024 ? radius - 1024 : 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t c = codes[i], r = c - lo;
        if (r < 2048) atomicAdd(&s_h[r], 1u);
        else atomicAdd(&hist[c], 1ull);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 2048; i += 256)
        if (s_h[i] && lo + i < 65536) atomicAdd(&hist[lo + i], (unsigned long long)s_h[i]);
}
// ... and the unpredictable values in the order of their zero codes (LinearQuantizer's list, :57-66)
template <typename T>
__global__ __launch_bounds__(256) void k_slw_unpred(const uint16_t *__restrict__ codes, uint64_t n, const uint64_t *__restrict__ tile_base, const T *__restrict__ uval,
                                                    T *__restrict__ unpred) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        if (codes[i] == 0) unpred[stock_ordinal(codes, tile_base, i)] = uval[i];
}
// ---- stock ALGO_NOPRED streams (round 5; decomposition/NoPredictionDecomposition.hpp:17-33): every value quantized against a prediction of 0 ----
template <typename T>
__global__ __launch_bounds__(256) void k_snp_decode(const uint16_t *__restrict__ codes, uint64_t n, double eb, uint32_t radius, const uint64_t *__restrict__ tile_base,
                                                    const T *__restrict__ unpred, uint64_t n_unpred, T *__restrict__ out, uint32_t *bad) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t code = codes[i];
        T v;
        if (code) {
            v = ref_recover<T>((T)0, (int)code, eb, (int)radius);  // LinearQuantizer::recover, :77-86
        } else {
            const uint64_t k = stock_ordinal(codes, tile_base, i);
            v = k < n_unpred ? unpred[k] : (T)0;
            if (k >= n_unpred) *bad = 1u;
        }
        out[i] = v;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_snp_encode(const T *__restrict__ in, uint64_t n, double eb, uint32_t radius, uint16_t *__restrict__ codes) {
    const double recip = 1.0 / eb;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        T v = in[i];
        codes[i] = (uint16_t)ref_quantize<T>(v, (T)0, eb, recip, (int)radius);  // (a zero code: the value itself goes to the list, k_slw_unpred)
    }
}
int szk_launch_stock_nopred_decode(int dtype, const uint16_t *d_codes, uint64_t n, double eb, uint32_t radius, uint32_t *d_tile_cnt, uint64_t *d_tile_base,
                                   const void *d_unpred, uint64_t n_unpred, void *d_out, uint32_t *d_bad, hipStream_t s) {
    if (stock_zero_scan(d_codes, n, d_tile_cnt, d_tile_base, s)) return -1;
    const uint32_t g = stock_grid(n);
    if (dtype == 0) hipLaunchKernelGGL(k_snp_decode<float>, dim3(g), dim3(256), 0, s, d_codes, n, eb, radius, d_tile_base, (const float *)d_unpred, n_unpred, (float *)d_out, d_bad);
    else hipLaunchKernelGGL(k_snp_decode<double>, dim3(g), dim3(256), 0, s, d_codes, n, eb, radius, d_tile_base, (const double *)d_unpred, n_unpred, (double *)d_out, d_bad);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int szk_launch_stock_nopred_encode(int dtype, const void *d_in, uint64_t n, double eb, uint32_t radius, uint16_t *d_codes, hipStream_t s) {
    const uint32_t g = stock_grid(n);
    if (dtype == 0) hipLaunchKernelGGL(k_snp_encode<float>, dim3(g), dim3(256), 0, s, (const float *)d_in, n, eb, radius, d_codes);
    else hipLaunchKernelGGL(k_snp_encode<double>, dim3(g), dim3(256), 0, s, (const double *)d_in, n, eb, radius, d_codes);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
static size_t slw_lds4(const szk_slw_params *p, int dtype) {  // four waves' tiles of (B + 1)^4 values: 77 KB for f64 at B = 6
    const uint32_t te = p->B + 1;
    return (size_t)4 * te * te * te * te * (dtype == 0 ? 4 : 8);
}
int szk_launch_stock_lr_select(int dtype, const szk_slw_params *p, hipStream_t s) {
    if (p->N == 4) {
        const uint32_t nb4 = p->nbw * p->nb[0] * p->nb[1] * p->nb[2];
        const size_t lds = slw_lds4(p, dtype);
        if (dtype == 0) {
            (void)hipFuncSetAttribute((const void *)k_slw_select4<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_slw_select4<float>, dim3((nb4 + 3) / 4), dim3(256), lds, s, *p);
        } else {
            (void)hipFuncSetAttribute((const void *)k_slw_select4<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_slw_select4<double>, dim3((nb4 + 3) / 4), dim3(256), lds, s, *p);
        }
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    const uint32_t nblocks = p->nb[0] * p->nb[1] * p->nb[2];
    const dim3 grid((nblocks + 3) / 4), blk(256);
    if (p->N == 3) {
        if (dtype == 0) hipLaunchKernelGGL((k_slw_select<float, 3>), grid, blk, 0, s, *p);
        else hipLaunchKernelGGL((k_slw_select<double, 3>), grid, blk, 0, s, *p);
    } else {
        if (dtype == 0) hipLaunchKernelGGL((k_slw_select<float, 2>), grid, blk, 0, s, *p);
        else hipLaunchKernelGGL((k_slw_select<double, 2>), grid, blk, 0, s, *p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int szk_launch_stock_lr_code(int dtype, const szk_slw_params *p, hipStream_t s) {
    if (p->N == 4) {
        const size_t lds = slw_lds4(p, dtype);
        const uint32_t nd4 = p->nbw + p->nb[0] + p->nb[1] + p->nb[2] - 3;
        const dim3 g4((p->nbw * p->nb[0] * p->nb[1] + 3) / 4), b4(256);
        if (dtype == 0) (void)hipFuncSetAttribute((const void *)k_slw_front4<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else (void)hipFuncSetAttribute((const void *)k_slw_front4<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (uint32_t d = 0; d < nd4; d++) {
            if (dtype == 0) hipLaunchKernelGGL(k_slw_front4<float>, g4, b4, lds, s, *p, d);
            else hipLaunchKernelGGL(k_slw_front4<double>, g4, b4, lds, s, *p, d);
        }
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    const uint32_t ndiag = p->nb[0] + p->nb[1] + p->nb[2] - 2;
    const uint32_t cand = p->N == 3 ? p->nb[0] * p->nb[1] : p->nb[1];
    const dim3 grid((cand + 3) / 4), blk(256);
    for (uint32_t d = 0; d < ndiag; d++) {
        if (p->N == 3) {
            if (dtype == 0) hipLaunchKernelGGL((k_slw_front<float, 3>), grid, blk, 0, s, *p, d);
            else hipLaunchKernelGGL((k_slw_front<double, 3>), grid, blk, 0, s, *p, d);
        } else {
            if (dtype == 0) hipLaunchKernelGGL((k_slw_front<float, 2>), grid, blk, 0, s, *p, d);
            else hipLaunchKernelGGL((k_slw_front<double, 2>), grid, blk, 0, s, *p, d);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int szk_launch_stock_lr_finish(int dtype, const szk_slw_params *p, uint64_t n, uint64_t *d_hist, uint32_t *d_tile_cnt, uint64_t *d_tile_base, void *d_unpred,
                               uint64_t *h_n_unpred, hipStream_t s) {
    hipLaunchKernelGGL(k_slw_hist, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, p->codes, n, p->radius, (unsigned long long *)d_hist);
    if (stock_zero_scan(p->codes, n, d_tile_cnt, d_tile_base, s)) return -1;
    const uint64_t ntiles = (n + 1023) / 1024;
    if (hipMemcpyAsync(h_n_unpred, d_tile_base + ntiles, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
    if (*h_n_unpred) {
        const uint32_t g = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
        if (dtype == 0) hipLaunchKernelGGL(k_slw_unpred<float>, dim3(g), dim3(256), 0, s, p->codes, n, d_tile_base, (const float *)p->uval, (float *)d_unpred);
        else hipLaunchKernelGGL(k_slw_unpred<double>, dim3(g), dim3(256), 0, s, p->codes, n, d_tile_base, (const double *)p->uval, (double *)d_unpred);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int szk_launch_stock_lorenzo_reg(int dtype, const szk_slr_params *p, uint64_t n, uint32_t *d_tile_cnt, uint64_t *d_tile_base, hipStream_t s) {
    if (stock_zero_scan(p->codes, n, d_tile_cnt, d_tile_base, s)) return -1;
    if (p->N == 1) {
        if (dtype == 0) hipLaunchKernelGGL(k_slr_chain<float>, dim3(1), dim3(1024), 0, s, *p);
        else hipLaunchKernelGGL(k_slr_chain<double>, dim3(1), dim3(1024), 0, s, *p);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if (p->N == 4) {
        const uint32_t te = p->B + 1;
        const size_t lds = (size_t)4 * te * te * te * te * (dtype == 0 ? 4 : 8);
        const uint32_t nd4 = p->nbw + p->nb[0] + p->nb[1] + p->nb[2] - 3;
        const dim3 g4((p->nbw * p->nb[0] * p->nb[1] + 3) / 4), b4(256);
        // (four waves' tiles of (B + 1)^4 values: 77 KB for f64 at B = 6 — beyond the 64 KB a launch gets without asking)
        if (dtype == 0) (void)hipFuncSetAttribute((const void *)k_slr_front4<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else (void)hipFuncSetAttribute((const void *)k_slr_front4<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (uint32_t d = 0; d < nd4; d++) {
            if (dtype == 0) hipLaunchKernelGGL(k_slr_front4<float>, g4, b4, lds, s, *p, d);
            else hipLaunchKernelGGL(k_slr_front4<double>, g4, b4, lds, s, *p, d);
        }
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    const uint32_t ndiag = p->nb[0] + p->nb[1] + p->nb[2] - 2;
    const uint32_t cand = p->N == 3 ? p->nb[0] * p->nb[1] : p->nb[1];
    const dim3 grid((cand + 3) / 4), blk(256);
    for (uint32_t d = 0; d < ndiag; d++) {
        if (p->N == 3) {
            if (dtype == 0) hipLaunchKernelGGL((k_slr_front<float, 3>), grid, blk, 0, s, *p, d);
            else hipLaunchKernelGGL((k_slr_front<double, 3>), grid, blk, 0, s, *p, d);
        } else {
            if (dtype == 0) hipLaunchKernelGGL((k_slr_front<float, 2>), grid, blk, 0, s, *p, d);
            else hipLaunchKernelGGL((k_slr_front<double, 2>), grid, blk, 0, s, *p, d);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
