// sz3_amd/csrc/sz3hip_stock_geom.h — where the reference's interpolation decomposition EMITS the code of every element.
//
// Stock SZ3 streams of ALGO_INTERP (api/impl/SZAlgoInterp.hpp:17-30) hold their quantisation codes in the order
// InterpolationDecomposition::compress visits the points (decomposition/InterpolationDecomposition.hpp:79-147): the anchor grid
// (build_anchor_grid, :215-221; or the first point, :93), then level by level (coarse to fine), block by block (row-major over
// the level's grid of 32-interval blocks, :118-143), pass by pass (:429-450 for N >= 3, :409-428 for N = 2, interpolation_1d for
// N = 1), and inside a pass group by group — the interior points of the lines, then up to three boundary positions (:352-399) —
// each group a row-major loop nest over the block (foreach, utils/Iterator / IP order: dimension 0 outermost).
// This library's kernels keep a code per ELEMENT (codes[element index], bit-identical to the reference's code of that element:
// DESIGN.md §2). Reading or writing a stock stream therefore needs one function: the emission rank of an element. It is computed
// here in closed form per element — level, block, pass, group and position inside the group from the coordinates; the number of
// points emitted before the block from a per-level table of block totals the host fills with the same counting function.
// Host and device share this header (the host builds the tables and the CPU tests evaluate ranks without a device).
#ifndef SZ3HIP_STOCK_GEOM_H
#define SZ3HIP_STOCK_GEOM_H
#include <stdint.h>

#if defined(__HIPCC__)
#define SZG_HD __host__ __device__ __forceinline__
#else
#define SZG_HD inline
#endif

#define SZG_MAX_LEVELS 48

struct szg_geom {
    int N;             // 1..4
    int interp_id;     // 0 linear, 1 cubic
    int seq[4];        // the direction's permutation of the dimensions (dim_sequences[direction], :205-212)
    uint64_t d[4];     // extents, slowest first (exactly N)
    uint64_t anchor;   // effective anchor stride (0: none — the first point is coded instead)
    int top;           // levels run top .. 1; stride of level l is 2^(l-1), its blocks have 32 of those per edge
    uint64_t head;     // codes emitted before the level loop: the anchor grid's points, or 1
    uint64_t n;        // elements
    uint64_t level_base[SZG_MAX_LEVELS];  // rank of the first code of level l
    uint64_t nb[SZG_MAX_LEVELS][4];       // blocks per dimension at level l
    uint64_t blk_off[SZG_MAX_LEVELS];     // where level l's per-block bases start in the table
};

// points interpolation_1d (:248-293; also the lines of the N = 2 passes) codes on a line of n grid points, and the position of grid
// point i (odd) among them
SZG_HD uint64_t szg_line_count(uint64_t n, int interp_id) {
    if (n <= 1) return 0;
    if (interp_id == 0 || n < 5) return (n - 1) / 2 + ((n & 1) ? 0 : 1);
    const uint64_t c0 = n > 6 ? (n - 5) / 2 : 0;  // i = 3, 5, ... while i + 3 < n
    return c0 + 2 + ((n & 1) ? 0 : 1);            // + quad_1 at 1, quad_2 at the loop's exit, quad_3 at n - 1 (n even)
}
SZG_HD uint64_t szg_line_pos(uint64_t i, uint64_t n, int interp_id) {
    if (interp_id == 0 || n < 5) return (i + 1 < n) ? (i - 1) / 2 : (n - 1) / 2;  // (the last point of an even line comes behind the pairs)
    const uint64_t c0 = n > 6 ? (n - 5) / 2 : 0;
    if (i == 1) return c0;
    if (i >= 3 && i + 3 < n) return (i - 3) / 2;
    if (i == 3 + 2 * c0) return c0 + 1;
    return c0 + 2;  // i == n - 1, n even
}

// one pass of interpolation() over one block for N >= 3 (interpolation_1d_fastest_dim_first, :310-402): the loop nest's lower ends
// and steps outside the pass's own dimension, the grid points n along it, and the counts
struct szg_pass {
    bool empty;
    int dm;
    uint64_t n;         // grid points of a line along dm
    uint64_t lo[4], step[4], cnt[4];  // (dm: unused)
    uint64_t other;     // product of cnt over the dimensions other than dm
    uint64_t c0;        // points of the first group along dm (interior / pairs)
    uint64_t ngroups;   // groups with at least one point: the first (if c0) + the boundary positions
};
SZG_HD uint64_t szg_bnd(const szg_pass &p, int interp_id, uint64_t *bnd) {  // boundary positions in emission order; returns how many
    uint64_t k = 0;
    const uint64_t n = p.n;
    if (interp_id == 0) {
        if ((n & 1) == 0) bnd[k++] = n - 1;
        return k;
    }
    bnd[k++] = 1;
    if ((n & 1) == 1 && n > 3) bnd[k++] = n - 2;
    if ((n & 1) == 0 && n > 4) bnd[k++] = n - 3;
    if ((n & 1) == 0 && n > 2) bnd[k++] = n - 1;
    return k;
}
SZG_HD void szg_pass_geom(const szg_geom &g, const uint64_t *begin, const uint64_t *end, uint64_t s, int k, szg_pass &p) {
    p.empty = false;
    p.dm = g.seq[k];
    p.other = 1;
    for (int j = 0; j < g.N; j++) {
        const int dd = g.seq[j];
        if (j == k) continue;
        const uint64_t st = j < k ? s : 2 * s;  // dimensions already interpolated at this level run over every grid point, the others over every second
        p.step[dd] = st;
        p.lo[dd] = begin[dd] ? begin[dd] + st : 0;
        if (end[dd] < p.lo[dd]) {
            p.empty = true;
            p.cnt[dd] = 0;
        } else {
            p.cnt[dd] = (end[dd] - p.lo[dd]) / st + 1;
        }
        p.other *= p.cnt[dd];
    }
    p.n = (end[p.dm] - begin[p.dm]) / s + 1;
    if (p.n <= 1) p.empty = true;
    if (g.interp_id == 0) p.c0 = p.n >= 2 ? (p.n - 1) / 2 : 0;                  // i = 1, 3, ... < n - 1
    else p.c0 = p.n > 6 ? (p.n - 5) / 2 : 0;                                   // i = 3, 5, ... < n - 3
}
SZG_HD uint64_t szg_pass_total(const szg_geom &g, const szg_pass &p) {
    if (p.empty) return 0;
    uint64_t bnd[4];
    return p.other * (p.c0 + szg_bnd(p, g.interp_id, bnd));
}
// codes one block emits (all its passes)
SZG_HD uint64_t szg_block_total(const szg_geom &g, const uint64_t *begin, const uint64_t *end, uint64_t s) {
    if (g.N == 1) return szg_line_count((end[0] - begin[0]) / s + 1, g.interp_id);
    if (g.N == 2) {  // (:409-428) lines along seq[0] at every second grid point of seq[1], then lines along seq[1] at every grid point of seq[0]
        const int a = g.seq[0], b = g.seq[1];
        const uint64_t j0 = begin[b] ? begin[b] + 2 * s : 0, i0 = begin[a] ? begin[a] + s : 0;
        const uint64_t lines0 = end[b] >= j0 ? (end[b] - j0) / (2 * s) + 1 : 0, lines1 = end[a] >= i0 ? (end[a] - i0) / s + 1 : 0;
        return lines0 * szg_line_count((end[a] - begin[a]) / s + 1, g.interp_id) + lines1 * szg_line_count((end[b] - begin[b]) / s + 1, g.interp_id);
    }
    uint64_t t = 0;
    for (int k = 0; k < g.N; k++) {
        szg_pass p;
        szg_pass_geom(g, begin, end, s, k, p);
        t += szg_pass_total(g, p);
    }
    return t;
}
SZG_HD void szg_block_box(const szg_geom &g, int level, const uint64_t *b, uint64_t *begin, uint64_t *end) {
    const uint64_t bsz = (uint64_t)32 << (level - 1);
    for (int i = 0; i < g.N; i++) {
        begin[i] = b[i] * bsz;
        end[i] = begin[i] + bsz;
        if (end[i] > g.d[i] - 1) end[i] = g.d[i] - 1;
    }
}
SZG_HD int szg_tz(uint64_t x) {  // trailing zeros, 64 for 0
    if (x == 0) return 64;
    int t = 0;
    while (!(x & 1)) {
        x >>= 1;
        t++;
    }
    return t;
}

// the emission rank of the element at coordinates x[] (slowest first); blk_base: the per-block bases (see szg_geom::blk_off)
SZG_HD uint64_t szg_rank(const szg_geom &g, const uint64_t *blk_base, const uint64_t *x) {
    int mt = 64;
    for (int i = 0; i < g.N; i++) {
        const int t = szg_tz(x[i]);
        mt = t < mt ? t : mt;
    }
    if (mt >= g.top) {  // on the anchor grid (or the first point)
        if (!g.anchor) return 0;
        uint64_t r = 0;
        for (int i = 0; i < g.N; i++) r = r * ((g.d[i] - 1) / g.anchor + 1) + x[i] / g.anchor;
        return r;
    }
    const int level = mt + 1;
    const uint64_t s = (uint64_t)1 << mt, bsz = 32 * s;
    // the pass: the LAST dimension of the sequence whose coordinate is an odd multiple of s (the later ones are still on the coarser grid)
    int k = 0;
    for (int j = 0; j < g.N; j++)
        if ((x[g.seq[j]] >> mt) & 1) k = j;
    const int dm = g.seq[k];
    // the block: along dm the point lies strictly inside; a point ON a block's lower face belongs to the block below (whose upper face it is)
    uint64_t b[4], begin[4], end[4];
    for (int i = 0; i < g.N; i++) b[i] = i == dm ? x[i] / bsz : (x[i] ? (x[i] - 1) / bsz : 0);
    szg_block_box(g, level, b, begin, end);
    uint64_t bi = 0;
    for (int i = 0; i < g.N; i++) bi = bi * g.nb[level][i] + b[i];
    uint64_t r = blk_base[g.blk_off[level] + bi];
    const uint64_t idm = (x[dm] - begin[dm]) / s;  // the point's grid index on its line
    if (g.N == 1) return r + szg_line_pos(idm, (end[0] - begin[0]) / s + 1, g.interp_id);
    if (g.N == 2) {
        const int a = g.seq[0], bb = g.seq[1];
        const uint64_t na = (end[a] - begin[a]) / s + 1, nbb = (end[bb] - begin[bb]) / s + 1;
        const uint64_t j0 = begin[bb] ? begin[bb] + 2 * s : 0, i0 = begin[a] ? begin[a] + s : 0;
        if (k == 0) return r + ((x[bb] - j0) / (2 * s)) * szg_line_count(na, g.interp_id) + szg_line_pos(idm, na, g.interp_id);
        const uint64_t lines0 = end[bb] >= j0 ? (end[bb] - j0) / (2 * s) + 1 : 0;
        return r + lines0 * szg_line_count(na, g.interp_id) + ((x[a] - i0) / s) * szg_line_count(nbb, g.interp_id) + szg_line_pos(idm, nbb, g.interp_id);
    }
    szg_pass p;
    for (int j = 0; j < k; j++) {
        szg_pass_geom(g, begin, end, s, j, p);
        r += szg_pass_total(g, p);
    }
    szg_pass_geom(g, begin, end, s, k, p);
    // the group: interior / pairs first, then the boundary positions one by one
    uint64_t gcount, pos, before;
    const bool first = g.interp_id == 0 ? (idm + 1 < p.n) : (idm >= 3 && idm + 3 < p.n);
    if (first) {
        gcount = p.c0;
        pos = g.interp_id == 0 ? (idm - 1) / 2 : (idm - 3) / 2;
        before = 0;
    } else {
        uint64_t bnd[4];
        const uint64_t nbnd = szg_bnd(p, g.interp_id, bnd);
        uint64_t q = 0;
        while (q + 1 < nbnd && bnd[q] != idm) q++;
        gcount = 1;
        pos = 0;
        before = p.other * (p.c0 + q);
    }
    uint64_t idx = 0;  // row-major over the group's loop nest, dimension 0 outermost
    for (int i = 0; i < g.N; i++) {
        if (i == dm) idx = idx * gcount + pos;
        else idx = idx * p.cnt[i] + (x[i] - p.lo[i]) / p.step[i];
    }
    return r + before + idx;
}

#endif
