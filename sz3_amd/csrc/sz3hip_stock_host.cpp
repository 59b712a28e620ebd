// sz3_amd/csrc/sz3hip_stock_host.cpp — stock SZ3 streams of ALGO_INTERP read and written by this library.
//
// A stock stream's pre-zstd buffer (compressor/SZGenericCompressor.hpp:51-57) is
//     [InterpolationDecomposition::save (decomposition/InterpolationDecomposition.hpp:149-159): u64 dims x N, u32 blocksize = 32,
//      i32 interp id, i32 direction, u64 anchor stride, f64 alpha, f64 beta, LinearQuantizer::save (quantizer/LinearQuantizer.hpp:95-104):
//      u8 uid = 2, f64 eb, i32 radius, u64 count, T x count]
//     [HuffmanEncoder::save (encoder/HuffmanEncoder.hpp:108-125, 601-628): i32 offset, BE i32 nodeCount, BE i32 stateNum / 2, u8 endian,
//      L / R child indices (1 / 2 / 4 bytes by nodeCount), i32 C x nodeCount, u8 t x nodeCount — node 0 the root, pre-order]
//     [u64 number of codes][u64 bytes of the bit stream][bits, MSB first: left = 0, right = 1]
// What runs where: prediction, quantisation and reconstruction are this library's interpolation kernels (bit-identical codes per
// element) plus the permutation between element order and the reference's emission order (sz3hip_stock.hip); this
// file is the container around them — the byte layout above and the Huffman stage in the reference's tree format, on the host
// like the zstd stage behind it (lossless/Lossless_zstd.hpp:29-45). Any valid tree decodes in stock SZ3 (its decoder walks the
// serialised tree, :225-255), so the tree built here need not be the reference's own; reconstruction is the reference's bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <queue>
#include <thread>
#include <array>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/sz3hip.h"
#include "sz3hip_internal.h"
#include "sz3hip_stock_geom.h"
#include "sz3hip_stock_host.h"

int szk_stock_geom_build(int N, const uint64_t *dims, int interp_id, int direction, uint64_t anchor_stride, szg_geom *gp, std::vector<uint64_t> *blk_base);

namespace stock {
namespace {
struct W {
    std::vector<uint8_t> &b;
    template <typename V> void put(V v) {
        const size_t at = b.size();
        b.resize(at + sizeof(V));
        memcpy(b.data() + at, &v, sizeof(V));
    }
    void be32(uint32_t v) {
        const uint8_t q[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
        b.insert(b.end(), q, q + 4);
    }
    void bytes(const void *p, size_t n) {
        const uint8_t *q = static_cast<const uint8_t *>(p);
        b.insert(b.end(), q, q + n);
    }
};
struct R {
    const uint8_t *p, *end;
    bool ok = true;
    template <typename V> V get() {
        V v{};
        if ((size_t)(end - p) < sizeof(V)) {
            ok = false;
            return v;
        }
        memcpy(&v, p, sizeof(V));
        p += sizeof(V);
        return v;
    }
    uint32_t be32() {
        if (end - p < 4) {
            ok = false;
            return 0;
        }
        const uint32_t v = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
        p += 4;
        return v;
    }
};
unsigned threads() {
    unsigned t = std::thread::hardware_concurrency();
    return t == 0 ? 4 : (t > 32 ? 32 : t);
}
template <typename F> void parallel(size_t n_items, F f) {  // f(begin, end, thread index)
    const unsigned nt = (unsigned)std::min<size_t>(threads(), std::max<size_t>(1, n_items / (1u << 20)));
    if (nt <= 1) {
        f((size_t)0, n_items, 0u);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back([=] { f(n_items * t / nt, n_items * (t + 1) / nt, t); });
    for (auto &x : th) x.join();
}

// ---- Huffman in the reference's container (Tree, Code: sz3hip_stock_host.h) ----------------------------------------------
// an optimal prefix code over the symbols with freq > 0 (heap merge), as a pre-order tree + the code of every symbol
void build_tree(const std::vector<uint64_t> &freq, Tree &tr, std::vector<Code> &codes, bool ref_heap = false) {
    struct N {
        uint64_t f;
        int l, r, sym;
    };
    std::vector<N> nodes;
    int root = -1;
    if (ref_heap) {
        // the reference's own queue (encoder/HuffmanEncoder.hpp:402-432, 516-561): a 1-based binary heap whose insertion climbs only past
        // STRICTLY larger parents and whose removal prefers the right child only when strictly smaller — which of several equal
        // frequencies merges first decides the tree's shape (not its cost), and a tuner trial's zstd size is taken over the tree's bytes
        struct HE {  // (a queue entry carries its frequency: the sift loops compare entries, not nodes behind an index — a wide alphabet's 20 000
            uint64_t f;  //  symbols are a million comparisons)
            int n;
        };
        std::vector<HE> heap(1);  // (1-based: slot 0 unused)
        size_t K = 0;
        for (size_t s = 0; s < freq.size(); s++) K += freq[s] != 0;
        nodes.reserve(2 * K + 1);
        heap.reserve(K + 2);
        auto insert = [&](int n) {
            const uint64_t f = nodes[n].f;
            heap.push_back(HE{f, n});
            size_t i = heap.size() - 1;
            for (size_t j = i >> 1; j; j = i >> 1) {
                if (heap[j].f <= f) break;
                heap[i] = heap[j];
                i = j;
            }
            heap[i] = HE{f, n};
        };
        auto remove = [&]() {
            const int top = heap[1].n;
            heap[1] = heap.back();
            heap.pop_back();
            const size_t end = heap.size();
            size_t i = 1;
            for (size_t l = i << 1; l < end; l = i << 1) {
                if (l + 1 < end && heap[l + 1].f < heap[l].f) l++;
                if (heap[i].f > heap[l].f) {
                    std::swap(heap[i], heap[l]);
                    i = l;
                } else {
                    break;
                }
            }
            return top;
        };
        for (size_t s = 0; s < freq.size(); s++)
            if (freq[s]) {
                nodes.push_back({freq[s], -1, -1, (int)s});
                insert((int)nodes.size() - 1);
            }
        while (heap.size() > 2) {
            const int a = remove(), b = remove();
            nodes.push_back({nodes[a].f + nodes[b].f, a, b, -1});
            insert((int)nodes.size() - 1);
        }
        root = heap[1].n;
    } else {
        typedef std::pair<uint64_t, int> QE;  // (frequency, node): ties by creation order
        std::priority_queue<QE, std::vector<QE>, std::greater<QE>> q;
        for (size_t s = 0; s < freq.size(); s++)
            if (freq[s]) {
                nodes.push_back({freq[s], -1, -1, (int)s});
                q.push({freq[s], (int)nodes.size() - 1});
            }
        while (q.size() > 1) {
            const QE a = q.top();
            q.pop();
            const QE b = q.top();
            q.pop();
            nodes.push_back({a.first + b.first, a.second, b.second, -1});
            q.push({a.first + b.first, (int)nodes.size() - 1});
        }
        root = q.top().second;
    }
    const size_t nc = nodes.size();
    tr.L.assign(nc, 0);
    tr.R.assign(nc, 0);
    tr.C.assign(nc, 0);
    tr.t.assign(nc, 0);
    codes.assign(freq.size(), Code{0, 0});
    // pre-order numbering like pad_tree (:601-616): a node, its left subtree, its right subtree — a node's number is its rank in that walk,
    // which a stack that takes the right child before the left one pops in (one visit per node: a wide alphabet's tree has 40 000)
    struct Fr {
        int node, parent;  // parent: the parent's NUMBER (-1: the root)
        uint64_t bits;
        uint32_t len;
        bool right;
    };
    std::vector<Fr> st;
    st.reserve(256);
    uint32_t next = 0;
    st.push_back({root, -1, 0, 0, false});
    while (!st.empty()) {
        const Fr f = st.back();
        st.pop_back();
        const uint32_t id = next++;
        if (f.parent >= 0) (f.right ? tr.R : tr.L)[f.parent] = id;
        const N &nd = nodes[f.node];
        if (nd.sym >= 0) {
            tr.t[id] = 1;
            tr.C[id] = nd.sym;
            codes[nd.sym] = Code{f.bits, f.len};
            continue;
        }
        st.push_back({nd.r, (int)id, (f.bits << 1) | 1, f.len + 1, true});
        st.push_back({nd.l, (int)id, f.bits << 1, f.len + 1, false});
    }
}
void save_tree(W &w, const Tree &tr, int32_t offset, uint32_t state_num) {
    const uint32_t nc = (uint32_t)tr.t.size();
    w.put<int32_t>(offset);
    w.be32(nc);
    w.be32(state_num / 2);
    w.put<uint8_t>(0);  // little endian
    auto idx = [&](const std::vector<uint32_t> &v) {
        for (uint32_t x : v) {
            if (nc <= 256) w.put<uint8_t>((uint8_t)x);
            else if (nc <= 65536) w.put<uint16_t>((uint16_t)x);
            else w.put<uint32_t>(x);
        }
    };
    idx(tr.L);
    idx(tr.R);
    w.bytes(tr.C.data(), nc * 4);
    w.bytes(tr.t.data(), nc);
}
bool load_tree(R &r, Tree &tr, int32_t &offset) {
    offset = r.get<int32_t>();
    const uint32_t nc = r.be32();
    (void)r.be32();  // stateNum / 2: sizes the reference's node pool
    (void)r.get<uint8_t>();
    if (!r.ok || nc == 0 || nc > (1u << 18)) return false;  // (at most 65536 symbols: 131071 nodes)
    const size_t w = nc <= 256 ? 1 : (nc <= 65536 ? 2 : 4);
    if ((size_t)(r.end - r.p) < (size_t)nc * (2 * w + 5)) return false;
    tr.L.resize(nc);
    tr.R.resize(nc);
    tr.C.resize(nc);
    tr.t.resize(nc);
    auto idx = [&](std::vector<uint32_t> &v) {
        for (uint32_t i = 0; i < nc; i++) {
            uint32_t x = 0;
            memcpy(&x, r.p, w);
            r.p += w;
            v[i] = x;
        }
    };
    idx(tr.L);
    idx(tr.R);
    memcpy(tr.C.data(), r.p, (size_t)nc * 4);
    r.p += (size_t)nc * 4;
    memcpy(tr.t.data(), r.p, nc);
    r.p += nc;
    for (uint32_t i = 0; i < nc; i++)
        if (tr.L[i] >= nc || tr.R[i] >= nc) return false;
    // The arrays come from a file: they must BE a tree before anything walks them (a cycle, a node reached twice or a node with one
    // child would make every walk of the decoders — one per restart point on the device — run to the end of the bit stream).
    // Pre-order walk from the root, the shape the reference writes (HuffmanEncoder.hpp:601-628): every node is reached exactly
    // once, an inner node (t == 0) has two different children, neither of them the root, a tree of nc nodes has (nc + 1) / 2
    // leaves, and no leaf lies deeper than 64 (a code word the reference's own encoder could not have counted: a leaf at depth d
    // needs more than Fib(d) elements).
    if ((nc & 1u) == 0) return false;
    std::vector<uint8_t> seen(nc, 0);
    std::vector<std::pair<uint32_t, uint32_t>> st;  // (node, depth)
    st.reserve(130);
    st.emplace_back(0u, 0u);
    uint32_t reached = 0, leaves = 0;
    while (!st.empty()) {
        const uint32_t nd = st.back().first, dep = st.back().second;
        st.pop_back();
        if (seen[nd]) return false;
        seen[nd] = 1;
        reached++;
        if (tr.t[nd]) {
            leaves++;
            continue;
        }
        const uint32_t l = tr.L[nd], r2 = tr.R[nd];
        if (l == 0 || r2 == 0 || l == r2 || dep >= 64) return false;
        st.emplace_back(r2, dep + 1);
        st.emplace_back(l, dep + 1);
    }
    if (reached != nc || leaves != (nc + 1) / 2) return false;
    return true;
}
// bits of n codes, MSB first. Threads code contiguous ranges into buffers of their own; the ranges are then joined with the
// bit shift their predecessors' total leaves them.
// (the two halves of encode_bits, also a tuner trial's steps: a range of symbols into a buffer of its own, the buffers joined)
void encode_range(const uint16_t *em, uint64_t a, uint64_t b, int32_t offset, const std::vector<Code> &codes, std::vector<uint8_t> &o, uint64_t &bits_out) {
    // a 64-bit word fills from its top and leaves as eight bytes at once (a byte pushed per eight bits was 13 ns per symbol: most of a
    // tuner trial's exact price, and of the host twin of the device coder)
    o.resize((size_t)(b - a) / 2 + 64);
    size_t at = 0;
    uint64_t acc = 0;   // bits collected, from the top
    uint32_t have = 0;  // ... how many (< 64)
    uint64_t total = 0;
    for (uint64_t i = a; i < b; i++) {
        const Code &c = codes[(int32_t)em[i] - offset];
        const uint32_t len = c.len;  // (<= 64: the callers refuse longer code words)
        if (!len) continue;
        total += len;
        const uint32_t room = 64 - have;
        if (len < room) {
            acc |= c.bits << (room - len);
            have += len;
        } else {
            const uint32_t rest = len - room;  // (< 64)
            acc |= rest ? (c.bits >> rest) : c.bits;
            if (at + 16 > o.size()) o.resize(o.size() * 2);
            const uint64_t be = __builtin_bswap64(acc);
            memcpy(o.data() + at, &be, 8);
            at += 8;
            acc = rest ? (c.bits << (64 - rest)) : 0;
            have = rest;
        }
    }
    if (at + 16 > o.size()) o.resize(o.size() + 16);
    for (uint32_t k = 0; k < have; k += 8) o[at++] = (uint8_t)(acc >> (56 - k));
    o.resize(at);
    bits_out = total;
}
void join_parts(const std::vector<uint8_t> *part, const uint64_t *pbits, unsigned nt, std::vector<uint8_t> &out) {
    uint64_t total = 0;
    for (unsigned t = 0; t < nt; t++) total += pbits[t];
    out.assign((total + 7) / 8 + 8, 0);
    uint64_t at = 0;
    for (unsigned t = 0; t < nt; t++) {
        const uint32_t sh = (uint32_t)(at & 7);
        uint8_t *d = out.data() + (at >> 3);
        const std::vector<uint8_t> &o = part[t];
        const size_t nb = (pbits[t] + 7) / 8;
        if (sh == 0) {
            memcpy(d, o.data(), nb);
        } else {
            for (size_t k = 0; k < nb; k++) {
                d[k] |= (uint8_t)(o[k] >> sh);
                d[k + 1] |= (uint8_t)(o[k] << (8 - sh));
            }
        }
        at += pbits[t];
    }
    out.resize((total + 7) / 8);
}
void encode_bits(const uint16_t *em, uint64_t n, int32_t offset, const std::vector<Code> &codes, std::vector<uint8_t> &out) {
    const unsigned nt = (unsigned)std::min<uint64_t>(threads(), std::max<uint64_t>(1, n / (1u << 20)));
    std::vector<std::vector<uint8_t>> part(nt);
    std::vector<uint64_t> pbits(nt, 0);
    auto work = [&](unsigned t) { encode_range(em, n * t / nt, n * (t + 1) / nt, offset, codes, part[t], pbits[t]); };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    join_parts(part.data(), pbits.data(), nt, out);
}
// HuffmanEncoder::decode (:225-255): walk the tree bit by bit — through a 12-bit table of where twelve bits lead from the root
bool decode_bits(const Tree &tr, int32_t offset, const uint8_t *bits, size_t nbytes, uint64_t n, uint16_t *em) {
    if (tr.t[0]) {  // a single symbol: no bits at all (:233-237)
        const int32_t v = tr.C[0] + offset;
        if (v < 0 || v > 65535) return false;
        for (uint64_t i = 0; i < n; i++) em[i] = (uint16_t)v;
        return true;
    }
    const uint32_t LB = 12;
    struct E {
        uint32_t node;  // where the bits lead (a leaf: the symbol's node)
        uint8_t used;   // bits consumed to get there (a leaf may be reached early)
        uint8_t leaf;
    };
    std::vector<E> lut(1u << LB);
    for (uint32_t v = 0; v < (1u << LB); v++) {
        uint32_t nd = 0, used = 0;
        while (used < LB && !tr.t[nd]) {
            nd = ((v >> (LB - 1 - used)) & 1) ? tr.R[nd] : tr.L[nd];
            used++;
            if (nd == 0) break;  // (a missing child: corrupt tree — stops at the root, caught below as an endless walk)
        }
        lut[v] = {nd, (uint8_t)used, (uint8_t)(tr.t[nd] ? 1 : 0)};
    }
    const uint64_t total_bits = (uint64_t)nbytes * 8;
    uint64_t pos = 0;
    auto peek = [&](uint64_t at) -> uint32_t {  // twelve bits from `at`, zeros beyond the end
        uint32_t v = 0;
        const uint64_t by = at >> 3;
        for (int k = 0; k < 3; k++) v = (v << 8) | (by + k < nbytes ? bits[by + k] : 0u);
        return (v >> (24 - LB - (at & 7))) & ((1u << LB) - 1);
    };
    for (uint64_t i = 0; i < n; i++) {
        E e = lut[peek(pos)];
        pos += e.used;
        uint32_t nd = e.node;
        while (!tr.t[nd]) {  // code words beyond twelve bits: one by one
            if (pos >= total_bits || e.used == 0) return false;
            const uint32_t bit = (bits[pos >> 3] >> (7 - (pos & 7))) & 1;
            nd = bit ? tr.R[nd] : tr.L[nd];
            if (nd == 0) return false;
            pos++;
        }
        if (pos > total_bits) return false;
        const int32_t v = tr.C[nd] + offset;
        if (v < 0 || v > 65535) return false;
        em[i] = (uint16_t)v;
    }
    return true;
}
}  // namespace

// the 12-bit table of the device decoder (sz3hip_stock.hip): (node reached << 8) | (bits used << 1) | leaf
void make_lut(const Tree &tr, std::vector<uint32_t> &lut) {
    const uint32_t LB = 12;
    lut.assign(1u << LB, 0);
    for (uint32_t v = 0; v < (1u << LB); v++) {
        uint32_t nd = 0, used = 0;
        while (used < LB && !tr.t[nd]) {
            const uint32_t nx = ((v >> (LB - 1 - used)) & 1) ? tr.R[nd] : tr.L[nd];
            if (nx == 0) {  // a missing child: the table sends the decoder nowhere (used = 0 at the root means "corrupt")
                nd = 0;
                used = 0;
                break;
            }
            nd = nx;
            used++;
        }
        lut[v] = (nd << 8) | (used << 1) | (tr.t[nd] ? 1u : 0u);
    }
}
// code book of a stock stream from the histogram of its codes: frequencies over [min, max] (HuffmanEncoder::init, :516-537)
bool book_from_hist(const uint64_t *hist65536, Tree &tr, std::vector<uint8_t> &clen, std::vector<uint64_t> &cbits, int &lo, int &hi) {
    lo = 0;
    hi = 65535;
    while (lo < 65535 && !hist65536[lo]) lo++;
    while (hi > lo && !hist65536[hi]) hi--;
    if (!hist65536[lo]) return false;
    std::vector<uint64_t> freq(hist65536 + lo, hist65536 + hi + 1);
    std::vector<Code> codes;
    build_tree(freq, tr, codes, true);  // (the reference's own queue: which of two equal frequencies merges first — a written stream is then the reference's, byte for byte)
    clen.assign(65536, 0);
    cbits.assign(65536, 0);
    for (int s = lo; s <= hi; s++) {
        if (codes[s - lo].len > 64) return false;
        clen[s] = (uint8_t)codes[s - lo].len;
        cbits[s] = codes[s - lo].bits;
    }
    return true;
}
// everything of the pre-zstd buffer in front of the bit stream: decomposition header, quantizer, tree, the two counters
void write_head(const szi_stock_params &p, uint64_t anchor_effective, const void *unpred, uint64_t n_unpred, size_t tsize, const Tree &tr, int lo, int hi,
                uint64_t n, uint64_t bit_bytes, std::vector<uint8_t> &raw) {
    raw.clear();
    W w{raw};
    for (int i = 0; i < p.N; i++) w.put<uint64_t>(p.dims[i]);
    w.put<uint32_t>(32);  // blocksize (:87)
    w.put<int32_t>(p.interp_id);
    w.put<int32_t>(p.direction);
    w.put<uint64_t>(anchor_effective);
    w.put<double>(p.alpha);
    w.put<double>(p.beta);
    w.put<uint8_t>(2);  // LinearQuantizer uid (:131)
    w.put<double>(p.eb);
    w.put<int32_t>(p.radius);
    w.put<uint64_t>(n_unpred);
    if (n_unpred) w.bytes(unpred, (size_t)n_unpred * tsize);
    save_tree(w, tr, lo, (uint32_t)(hi - lo + 2));
    w.put<uint64_t>(n);
    w.put<uint64_t>(bit_bytes);
}
// host coder (A/B partner of the device's, SZ3HIP_STOCK_HOST_HUFFMAN=1)
void host_encode(const uint16_t *em, uint64_t n, const std::vector<uint8_t> &clen, const std::vector<uint64_t> &cbits, std::vector<uint8_t> &bits) {
    std::vector<Code> codes(65536);
    for (int s = 0; s < 65536; s++) codes[s] = Code{cbits[s], clen[s]};
    encode_bits(em, n, 0, codes, bits);
}
bool host_decode(const Tree &tr, int32_t offset, const uint8_t *bits, size_t nbytes, uint64_t n, uint16_t *em) { return decode_bits(tr, offset, bits, nbytes, n, em); }
// the inverse of write_head; false: not a well-formed stream of this kind. bits / bit_bytes: where the bit stream lies in raw
bool parse_head(const uint8_t *raw, size_t len, int N, size_t tsize, szi_stock_params &p, const uint8_t *&unpred, uint64_t &n_unpred, Tree &tr, int32_t &offset,
                uint64_t &n, const uint8_t *&bits, uint64_t &bit_bytes) {
    R r{raw, raw + len};
    memset(&p, 0, sizeof(p));
    p.N = N;
    for (int i = 0; i < N; i++) p.dims[i] = r.get<uint64_t>();
    const uint32_t blocksize = r.get<uint32_t>();
    p.interp_id = r.get<int32_t>();
    p.direction = r.get<int32_t>();
    p.anchor_stride = r.get<uint64_t>();
    p.alpha = r.get<double>();
    p.beta = r.get<double>();
    const uint8_t uid = r.get<uint8_t>();
    p.eb = r.get<double>();
    p.radius = r.get<int32_t>();
    n_unpred = r.get<uint64_t>();
    if (!r.ok || blocksize != 32 || uid != 2 || p.interp_id < 0 || p.interp_id > 1 || !(p.eb > 0) || p.radius < 1 || p.radius > 32768) return false;
    if ((uint64_t)(r.end - r.p) / tsize < n_unpred) return false;
    unpred = r.p;
    r.p += (size_t)n_unpred * tsize;
    if (!load_tree(r, tr, offset)) return false;
    n = r.get<uint64_t>();
    bit_bytes = r.get<uint64_t>();
    uint64_t want = 1;
    for (int i = 0; i < N; i++) want *= p.dims[i];
    if (!r.ok || n != want || (uint64_t)(r.end - r.p) < bit_bytes) return false;
    bits = r.p;
    return true;
}

namespace {
bool read_quant(R &r, size_t tsize, Quant &q) {
    const uint8_t uid = r.get<uint8_t>();
    q.eb = r.get<double>();
    q.radius = r.get<int32_t>();
    q.n_unpred = r.get<uint64_t>();
    if (!r.ok || uid != 2 || !(q.eb > 0) || q.radius < 1 || q.radius > 32768) return false;
    if ((uint64_t)(r.end - r.p) / tsize < q.n_unpred) return false;
    q.unpred = r.p;
    r.p += (size_t)q.n_unpred * tsize;
    return true;
}
// a side vector: HuffmanEncoder::save + encode as RegressionPredictor / ComposedPredictor call them (tree, [u64 bytes], bits); the
// reference does not step over the byte count's payload for a one-symbol tree (encoder/HuffmanEncoder.hpp:233-237: nothing was written)
bool read_vector(R &r, uint64_t n, std::vector<uint16_t> &out) {
    Tree tr;
    int32_t offset;
    if (!load_tree(r, tr, offset)) return false;
    const uint64_t bytes = r.get<uint64_t>();
    if (!r.ok || n > (1ull << 32)) return false;
    out.resize((size_t)n);
    if (tr.t[0]) return decode_bits(tr, offset, nullptr, 0, n, out.data());
    if ((uint64_t)(r.end - r.p) < bytes) return false;
    if (!decode_bits(tr, offset, r.p, (size_t)bytes, n, out.data())) return false;
    r.p += (size_t)bytes;
    return true;
}
}  // namespace

bool parse_lorenzo_reg(const uint8_t *raw, size_t len, size_t tsize, bool has_regression, bool composed, uint64_t nblocks, int N, LorenzoReg &o) {
    R r{raw, raw + len};
    if (has_regression) {  // RegressionPredictor::save (:94-107)
        const uint64_t nc = r.get<uint64_t>();
        if (!r.ok || nc > nblocks * (uint64_t)(N + 1)) return false;  // (N + 1 codes per regression block: a corrupt count must not size a vector)
        if (nc) {
            if (!read_quant(r, tsize, o.q_indep) || !read_quant(r, tsize, o.q_lin)) return false;
            if (!read_vector(r, nc, o.coef_codes)) return false;
        }
    }
    if (composed) {  // ComposedPredictor::save (:52-64)
        const uint64_t ns = r.get<uint64_t>();
        if (!r.ok || ns > nblocks) return false;
        if (ns && !read_vector(r, ns, o.selection)) return false;
    }
    if (!read_quant(r, tsize, o.q)) return false;
    if (!load_tree(r, o.tree, o.offset)) return false;
    o.n = r.get<uint64_t>();
    o.bit_bytes = r.get<uint64_t>();
    if (!r.ok || (uint64_t)(r.end - r.p) < o.bit_bytes) return false;
    o.bits = r.p;
    return true;
}
// ---- the WRITE side of a stock ALGO_LORENZO_REG stream (round 5) ------------------------------------------------------------
namespace {
// LinearQuantizer<T>::quantize_and_overwrite (quantizer/LinearQuantizer.hpp:43-71) — the host twin of ref_quantize (sz3hip_devutil.h)
template <typename T> int quantize_and_overwrite(T &data, T pred, double eb, double recip, int radius) {
    const T diff = data - pred;
    const double scaled = fabs((double)diff) * recip;
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
    if (fabs((double)ad) <= eb) {
        data = dec;
        return shifted;
    }
    return 0;
}
void write_quant(W &w, double eb, int32_t radius, const void *unpred, uint64_t n_unpred, size_t tsize) {  // LinearQuantizer::save, :95-105
    w.put<uint8_t>(2);
    w.put<double>(eb);
    w.put<int32_t>(radius);
    w.put<uint64_t>(n_unpred);
    if (n_unpred) w.bytes(unpred, (size_t)n_unpred * tsize);
}
// a side vector as RegressionPredictor / ComposedPredictor save it: HuffmanEncoder<int>::save + encode (tree over [min, max], byte count, bits)
void write_vector(W &w, const std::vector<uint16_t> &v) {
    uint32_t lo = 65535, hi = 0;
    for (uint16_t x : v) {
        lo = std::min<uint32_t>(lo, x);
        hi = std::max<uint32_t>(hi, x);
    }
    std::vector<uint64_t> freq(hi - lo + 1, 0);
    for (uint16_t x : v) freq[x - lo]++;
    Tree tr;
    std::vector<Code> codes;
    build_tree(freq, tr, codes, true);  // (the reference's own queue: which of two equal frequencies merges first — a written stream is then the reference's, byte for byte)
    save_tree(w, tr, (int32_t)lo, hi - lo + 2);
    std::vector<uint8_t> bits;
    if (!tr.t[0]) encode_bits(v.data(), v.size(), (int32_t)lo, codes, bits);
    w.put<uint64_t>(bits.size());
    if (!bits.empty()) w.bytes(bits.data(), bits.size());
}
}  // namespace

// the regression blocks' coefficient chain (RegressionPredictor::pred_and_quantize_coefficients, :142-149): every chosen block's fit is
// quantized against the previous chosen block's RECOVERED coefficients, in block order; coef (in: the fits, [blocks][4]) comes back holding
// what the reader will recover, codes / the two quantizers' unpredictable values are what the stream stores
template <typename T>
void lorenzo_reg_chain(int N, uint32_t B, double eb, const uint8_t *kind, uint64_t nblocks, T *coef, std::vector<uint16_t> &codes, std::vector<T> &un_indep,
                       std::vector<T> &un_lin) {
    const size_t CS = N == 4 ? 8 : 4;  // coefficients per block in the array (N + 1 used)
    const double eb_ind = eb / (N + 1), eb_lin = eb / (N + 1) / B;
    const double r_ind = 1.0 / eb_ind, r_lin = 1.0 / eb_lin;
    const int radius = 32768;
    T prev[5] = {0, 0, 0, 0, 0};
    for (uint64_t b = 0; b < nblocks; b++) {
        if (kind[b] != 2) continue;
        T *c = coef + b * CS;
        for (int i = 0; i < N; i++) {
            const T orig = c[i];
            const int q = quantize_and_overwrite<T>(c[i], prev[i], eb_lin, r_lin, radius);
            if (q == 0) un_lin.push_back(orig);
            codes.push_back((uint16_t)q);
        }
        const T orig = c[N];
        const int q = quantize_and_overwrite<T>(c[N], prev[N], eb_ind, r_ind, radius);
        if (q == 0) un_indep.push_back(orig);
        codes.push_back((uint16_t)q);
        for (int i = 0; i <= N; i++) prev[i] = c[i];
    }
}
template void lorenzo_reg_chain<float>(int, uint32_t, double, const uint8_t *, uint64_t, float *, std::vector<uint16_t> &, std::vector<float> &, std::vector<float> &);
template void lorenzo_reg_chain<double>(int, uint32_t, double, const uint8_t *, uint64_t, double *, std::vector<uint16_t> &, std::vector<double> &, std::vector<double> &);

// everything of the pre-zstd buffer in front of the main bit stream (BlockwiseDecomposition::save :68-72 = fallback Lorenzo: nothing,
// the predictor — RegressionPredictor::save :94-107, ComposedPredictor::save :52-64 —, the quantizer; then SZGenericCompressor.hpp:52-57)
void write_lorenzo_reg_head(int N, uint32_t B, double eb, size_t tsize, bool has_regression, bool composed, const std::vector<uint16_t> &coef_codes,
                            const void *un_indep, uint64_t n_un_indep, const void *un_lin, uint64_t n_un_lin, const std::vector<uint16_t> &selection, int32_t radius,
                            const void *unpred, uint64_t n_unpred, const Tree &tr, int lo, int hi, uint64_t n, uint64_t bit_bytes, std::vector<uint8_t> &raw) {
    raw.clear();
    W w{raw};
    if (has_regression) {
        w.put<uint64_t>(coef_codes.size());
        if (!coef_codes.empty()) {
            write_quant(w, eb / (N + 1), 32768, un_indep, n_un_indep, tsize);
            write_quant(w, eb / (N + 1) / B, 32768, un_lin, n_un_lin, tsize);
            write_vector(w, coef_codes);
        }
    }
    if (composed) {
        w.put<uint64_t>(selection.size());
        if (!selection.empty()) write_vector(w, selection);
    }
    write_quant(w, eb, radius, unpred, n_unpred, tsize);
    save_tree(w, tr, lo, (uint32_t)(hi - lo + 2));
    w.put<uint64_t>(n);
    w.put<uint64_t>(bit_bytes);
}
// ---- 1-D stock ALGO_LORENZO_REG streams: the chain on the host (round 5) -------------------------------------------------------
// A 1-D array under Lorenzo prediction is ONE dependent chain of roundings — value i needs value i - 1 as the reader will have it —
// and nothing of it is associative (T arithmetic, LinearQuantizer's double detour). On one GPU lane it ran 322 ms for 2^22 values
// (round 4, k_slr_chain), 3.6 x slower than one host core; the walk therefore runs here, over codes the device decoded / for the
// device to code. Same arithmetic as k_slr_front's: LorenzoPredictor::predict (:60-64, :75-76), RegressionPredictor::predict (:81-83),
// LinearQuantizer::recover (:77-86) / quantize_and_overwrite (:43-71).
template <typename T>
bool lorenzo_reg_read_1d(uint64_t n, uint32_t B, double eb, int radius, const uint16_t *codes, const uint8_t *kind, const T *coef, const T *unpred, uint64_t n_unpred,
                         T *out) {
    T p1 = 0, p2 = 0;  // the two values left of the next element (the reference pads the array with zeros)
    uint64_t u = 0;
    for (uint64_t x0 = 0, b = 0; x0 < n; x0 += B, b++) {
        const uint32_t ex = (uint32_t)std::min<uint64_t>(B, n - x0);
        const uint32_t k = kind[b];
        const T c0 = coef[b * 4], c1 = coef[b * 4 + 1];
        for (uint32_t t = 0; t < ex; t++) {
            const uint32_t code = codes[x0 + t];
            T v;
            if (code == 0) {
                if (u >= n_unpred) return false;
                v = unpred[u++];
            } else {
                const T pr = k == 2 ? (T)((T)(c0 * (T)t) + c1) : (k == 1 ? (T)((T)(2 * p1) - p2) : p1);
                v = (T)((double)pr + (double)(2 * ((int)code - radius)) * eb);
            }
            out[x0 + t] = v;
            p2 = p1;
            p1 = v;
        }
    }
    return true;
}
template bool lorenzo_reg_read_1d<float>(uint64_t, uint32_t, double, int, const uint16_t *, const uint8_t *, const float *, const float *, uint64_t, float *);
template bool lorenzo_reg_read_1d<double>(uint64_t, uint32_t, double, int, const uint16_t *, const uint8_t *, const double *, const double *, uint64_t, double *);

// the write side for a 1-D array: BlockwiseDecomposition::compress (:28-46) as it stands — block after block the members' sampled
// estimates on the array as it is at that moment (ComposedPredictor::precompress :25-40, the block's two ends: BlockwiseIterator.hpp:154-157),
// the first minimum, the regression coefficients quantized against the previous regression block's, every element quantized against its
// prediction from reconstructed values and overwritten. data: the array, overwritten with what the reader will decode.
template <typename T>
void lorenzo_reg_write_1d(uint64_t n, uint32_t B, double eb, int radius, uint32_t set_mask, T *data, std::vector<uint16_t> &codes, std::vector<T> &unpred,
                          std::vector<uint16_t> &selection, std::vector<uint16_t> &coef_codes, std::vector<T> &un_indep, std::vector<T> &un_lin) {
    const double recip = 1.0 / eb;
    const double eb_ind = eb / 2, eb_lin = eb / 2 / B;  // (N + 1 = 2)
    const double r_ind = 1.0 / eb_ind, r_lin = 1.0 / eb_lin;
    const int members = ((set_mask & 1u) ? 1 : 0) + ((set_mask & 2u) ? 1 : 0) + ((set_mask & 4u) ? 1 : 0);
    codes.resize((size_t)n);
    T prev_c[2] = {0, 0};
    auto at = [&](int64_t i) -> T { return i >= 0 ? data[i] : (T)0; };
    for (uint64_t x0 = 0; x0 < n; x0 += B) {
        const uint32_t ex = (uint32_t)std::min<uint64_t>(B, n - x0);
        const bool reg_valid = (set_mask & 4u) && ex > 1;
        T cf[2] = {0, 0};
        if (reg_valid) {
            double s0 = 0, sn = 0;
            // (a NaN's sign is in the file where a coefficient is stored as it is: the operand an x86 instruction returns when BOTH are NaN is its
            // first — the accumulator of `sum += x`, the minuend — whatever order this compiler gives a commutative operation: sz3hip_stock.hip, x86_op)
            auto keep = [](double a, double r) { return a != a ? a : r; };
            for (uint32_t t = 0; t < ex; t++) {
                s0 = keep(s0, s0 + (double)((T)t * data[x0 + t]));  // (sum[i] += index[i] * (*c): size_t * T is a product in T, RegressionPredictor.hpp:43 — in double
                                                                    // it is another coefficient once in ~10^5 blocks of f32 data, and the chain behind it another stream)
                sn = keep(sn, sn + (double)data[x0 + t]);
            }
            const double num = ex, d = ex;
            cf[1] = (T)(sn / num);
            const double lead = 2 * s0 / (d - 1);
            cf[0] = (T)(keep(lead, lead - sn) * 6 / num / (d + 1));
            cf[1] = (T)keep((double)cf[1], (double)cf[1] - (d - 1) * (double)cf[0] / 2);
        }
        uint32_t kind = 0, idx = 0;
        if (members > 1) {
            const double big = 1.7976931348623157e308;
            double e1 = 0, e2 = 0, er = 0;
            const uint32_t pts[2] = {0, ex - 1};
            for (int q = 0; q < 2; q++) {
                const int64_t i = (int64_t)x0 + pts[q];
                const T v = data[i];
                if (set_mask & 1u) e1 += (double)(T)(fabs((double)(T)(v - at(i - 1))) + 0.5 * eb);
                if (set_mask & 2u) e2 += (double)(T)(fabs((double)(T)(v - (T)((T)(2 * at(i - 1)) - at(i - 2)))) + 1.08 * eb);
                if (reg_valid) er += (double)(T)fabs((double)(T)(v - (T)((T)(cf[0] * (T)pts[q]) + cf[1])));
            }
            // (std::min_element's own walk, ComposedPredictor.hpp:38: the first member is the minimum until a later one compares LESS — an
            // estimate that is not a number, a block with NaN or Inf - Inf in it, is never less and, in front, never beaten)
            double best = 0;
            uint32_t k = 0;
            auto member = [&](double e, uint32_t kd) {
                if (k == 0 || e < best) best = e, kind = kd, idx = k;
                k++;
            };
            if (set_mask & 1u) member(e1, 0u);
            if (set_mask & 2u) member(e2, 1u);
            if (set_mask & 4u) member(reg_valid ? er : big, 2u);
            selection.push_back((uint16_t)idx);
        } else {
            kind = (set_mask & 1u) ? 0u : ((set_mask & 2u) ? 1u : 2u);
        }
        if (kind == 2 && !reg_valid) kind = 0;  // the fallback predictor (BlockwiseDecomposition.hpp:35-37; 1-D: nothing of the padding question)
        if (kind == 2) {
            const T o0 = cf[0], o1 = cf[1];
            const int q0 = quantize_and_overwrite<T>(cf[0], prev_c[0], eb_lin, r_lin, 32768);
            if (q0 == 0) un_lin.push_back(o0);
            const int q1 = quantize_and_overwrite<T>(cf[1], prev_c[1], eb_ind, r_ind, 32768);
            if (q1 == 0) un_indep.push_back(o1);
            coef_codes.push_back((uint16_t)q0);
            coef_codes.push_back((uint16_t)q1);
            prev_c[0] = cf[0];
            prev_c[1] = cf[1];
        }
        for (uint32_t t = 0; t < ex; t++) {
            const int64_t i = (int64_t)x0 + t;
            const T pr = kind == 2 ? (T)((T)(cf[0] * (T)t) + cf[1]) : (kind == 1 ? (T)((T)(2 * at(i - 1)) - at(i - 2)) : at(i - 1));
            const T orig = data[i];
            const int q = quantize_and_overwrite<T>(data[i], pr, eb, recip, radius);
            if (q == 0) unpred.push_back(orig);
            codes[(size_t)i] = (uint16_t)q;
        }
    }
}
template void lorenzo_reg_write_1d<float>(uint64_t, uint32_t, double, int, uint32_t, float *, std::vector<uint16_t> &, std::vector<float> &, std::vector<uint16_t> &,
                                          std::vector<uint16_t> &, std::vector<float> &, std::vector<float> &);
template void lorenzo_reg_write_1d<double>(uint64_t, uint32_t, double, int, uint32_t, double *, std::vector<uint16_t> &, std::vector<double> &, std::vector<uint16_t> &,
                                           std::vector<uint16_t> &, std::vector<double> &, std::vector<double> &);
// code book + bits of a host-side code array (the 1-D writer's)
bool encode_codes_host(const std::vector<uint16_t> &codes, Tree &tr, int &lo, int &hi, std::vector<uint8_t> &bits) {
    std::vector<uint64_t> hist(65536, 0);
    for (uint16_t c : codes) hist[c]++;
    std::vector<uint8_t> clen;
    std::vector<uint64_t> cbits;
    if (!book_from_hist(hist.data(), tr, clen, cbits, lo, hi)) return false;
    bits.clear();
    if (!tr.t[0]) host_encode(codes.data(), codes.size(), clen, cbits, bits);
    return true;
}
// The pre-zstd buffer of one trial of the ALGO_INTERP_LORENZO tuner, exactly as interp_compress_test makes it (api/impl/SZAlgoInterp.hpp:
// 42-78): ONE decomposition object codes all sampled blocks, so its quantizer's list holds the unpredictable values (anchors included) of
// all of them in emission order; the codes of all blocks are concatenated and coded with one tree; the buffer is decomposition header,
// quantizer, tree, count, bits — a stock ALGO_INTERP stream's body (write_head) over an array of the sample block's extents. codes: what the
// trial kernel left per ELEMENT of every block ([nb][per], zero = unpredictable), samples: the blocks themselves. The tree is built with
// the reference's own queue (build_tree, ref_heap): the trial's price is zstd's size of these very bytes.
// In steps (TrialWork, sz3hip_stock_host.h) so that the tuner spreads a group's trials over the pool's threads, two per trial where
// the step divides: prepare, order(0 | 1), book, bits(0 | 1), finish.
template <typename T>
bool TrialWork<T>::prepare() {
    std::vector<uint64_t> bb;
    if (szk_stock_geom_build(p.N, p.dims, p.interp_id, p.direction, p.anchor_stride, &g, &bb)) return false;
    per = g.n;
    if (!per || !nb) return false;
    // the element emitted r-th: a function of the block's geometry alone — kept from trial to trial and call to call (36 000 closed-form ranks
    // are milliseconds of one host thread; the trials of a tuning share two or three geometries, a series of calls all of them)
    static std::mutex mu;
    static std::map<std::array<uint64_t, 8>, std::shared_ptr<const std::vector<uint32_t>>> kept;
    const std::array<uint64_t, 8> key = {(uint64_t)p.N, p.dims[0], p.N > 1 ? p.dims[1] : 0, p.N > 2 ? p.dims[2] : 0, p.N > 3 ? p.dims[3] : 0, (uint64_t)g.interp_id,
                                         (uint64_t)p.direction, g.anchor};
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = kept.find(key);
        if (it != kept.end()) perm = it->second;
    }
    if (!perm) {
        auto v = std::make_shared<std::vector<uint32_t>>(per);
        for (uint64_t e = 0; e < per; e++) {
            uint64_t x[4], q = e;
            for (int i = p.N - 1; i >= 0; i--) {
                x[i] = q % g.d[i];
                q /= g.d[i];
            }
            (*v)[szg_rank(g, bb.data(), x)] = (uint32_t)e;
        }
        perm = v;
        std::lock_guard<std::mutex> lk(mu);
        if (kept.size() >= 32) kept.clear();
        kept[key] = perm;
    }
    em.resize((size_t)(per * nb));
    return true;
}
template <typename T>
void TrialWork<T>::order(int h) {  // blocks [nb h / 2, nb (h + 1) / 2): emission order, the quantizer's list, the symbols' counts
    const std::vector<uint32_t> &at = *perm;
    std::vector<uint32_t> &f = cnt[h];
    f.assign(65536, 0);
    std::vector<T> &u = un[h];
    u.clear();
    for (uint64_t k = nb * h / 2; k < nb * (h + 1) / 2; k++) {
        const uint16_t *c = codes + k * per;
        const T *v = samples + k * per;
        uint16_t *o = em.data() + k * per;
        for (uint64_t r = 0; r < per; r++) {
            const uint16_t x = c[at[r]];
            o[r] = x;
            if (!x) u.push_back(v[at[r]]);
            f[x]++;
        }
    }
}
template <typename T>
bool TrialWork<T>::book() {
    lo = 0;
    hi = 65535;
    while (lo < 65535 && !(cnt[0][lo] + cnt[1][lo])) lo++;
    while (hi > lo && !(cnt[0][hi] + cnt[1][hi])) hi--;
    std::vector<uint64_t> freq(hi - lo + 1, 0);
    for (uint32_t x = lo; x <= hi; x++) freq[x - lo] = (uint64_t)cnt[0][x] + cnt[1][x];
    build_tree(freq, tr, cw, true);
    for (const Code &c : cw)
        if (c.len > 64) return false;
    return true;
}
template <typename T>
void TrialWork<T>::bits(int h) {
    const uint64_t n = per * nb;
    pbits[h] = 0;
    part[h].clear();
    if (!tr.t[0]) encode_range(em.data(), n * h / 2, n * (h + 1) / 2, (int32_t)lo, cw, part[h], pbits[h]);
}
template <typename T>
void TrialWork<T>::finish(std::vector<uint8_t> &raw) {
    std::vector<uint8_t> joined;
    if (!tr.t[0]) join_parts(part, pbits, 2, joined);
    std::vector<T> u(un[0]);
    u.insert(u.end(), un[1].begin(), un[1].end());
    write_head(p, g.anchor, u.data(), u.size(), sizeof(T), tr, (int)lo, (int)hi, per * nb, joined.size(), raw);
    raw.insert(raw.end(), joined.begin(), joined.end());
}
template struct TrialWork<float>;
template struct TrialWork<double>;
template <typename T>
bool trial_buffer(const szi_stock_params &p, const uint16_t *codes, const T *samples, uint64_t nb, std::vector<uint8_t> &raw) {
    TrialWork<T> w;
    w.p = p;
    w.codes = codes;
    w.samples = samples;
    w.nb = nb;
    if (!w.prepare()) return false;
    w.order(0);
    w.order(1);
    if (!w.book()) return false;
    w.bits(0);
    w.bits(1);
    w.finish(raw);
    return true;
}
template bool trial_buffer<float>(const szi_stock_params &, const uint16_t *, const float *, uint64_t, std::vector<uint8_t> &);
template bool trial_buffer<double>(const szi_stock_params &, const uint16_t *, const double *, uint64_t, std::vector<uint8_t> &);
// ... and of the tuner's 1-D Lorenzo trial (lorenzo_compress_test, api/impl/SZAlgoInterp.hpp:80-120): ONE blockwise decomposition with the
// composed set {Lorenzo-1, Lorenzo-2} in blocks of five codes every sampled block as an array of its own (padding of zeros in front), its
// selection vector and its quantizer's list growing from block to block; the codes concatenated, one tree, the reference's buffer — a stock
// ALGO_LORENZO_REG stream's body (write_lorenzo_reg_head, composed, no regression). The walk is lorenzo_reg_write_1d's: the reference's order.
template <typename T>
bool lorenzo_trial_buffer(double eb, int radius, const T *samples, uint64_t per, uint64_t nb, std::vector<uint8_t> &raw) {
    std::vector<uint16_t> all, codes, sel, cc;
    std::vector<T> un, ui, ul, blk((size_t)per);
    all.reserve((size_t)(per * nb));
    for (uint64_t k = 0; k < nb; k++) {
        memcpy(blk.data(), samples + k * per, (size_t)per * sizeof(T));
        lorenzo_reg_write_1d<T>(per, 5, eb, radius, 3u, blk.data(), codes, un, sel, cc, ui, ul);
        all.insert(all.end(), codes.begin(), codes.end());
    }
    if (all.empty()) return false;
    Tree tr;
    int lo = 0, hi = 0;
    std::vector<uint8_t> bits;
    if (!encode_codes_host(all, tr, lo, hi, bits)) return false;
    write_lorenzo_reg_head(1, 5, eb, sizeof(T), false, true, cc, nullptr, 0, nullptr, 0, sel, radius, un.data(), un.size(), tr, lo, hi, all.size(), bits.size(), raw);
    raw.insert(raw.end(), bits.begin(), bits.end());
    return true;
}
template bool lorenzo_trial_buffer<float>(double, int, const float *, uint64_t, uint64_t, std::vector<uint8_t> &);
template bool lorenzo_trial_buffer<double>(double, int, const double *, uint64_t, uint64_t, std::vector<uint8_t> &);
}  // namespace stock
// test hook (CPU, no device): the buffer of a trial from per-element codes; returns its size, -1 when the geometry is refused, -2 when `cap` is short
extern "C" int64_t sz3hip_debug_trial_buffer(int N, const uint64_t *dims, int interp_id, int direction, uint64_t anchor_stride, double alpha, double beta,
                                             double eb, int radius, int dtype, const uint16_t *codes, const void *samples, uint64_t nb, uint8_t *out,
                                             uint64_t cap) {
    szi_stock_params p;
    memset(&p, 0, sizeof(p));
    p.N = N;
    for (int i = 0; i < N && i < 4; i++) p.dims[i] = dims[i];
    p.interp_id = interp_id;
    p.direction = direction;
    p.anchor_stride = anchor_stride;
    p.alpha = alpha;
    p.beta = beta;
    p.eb = eb;
    p.radius = radius;
    std::vector<uint8_t> raw;
    const bool made = dtype == 0 ? stock::trial_buffer<float>(p, codes, (const float *)samples, nb, raw) : stock::trial_buffer<double>(p, codes, (const double *)samples, nb, raw);
    if (!made) return -1;
    if (raw.size() > cap) return -2;
    memcpy(out, raw.data(), raw.size());
    return (int64_t)raw.size();
}
