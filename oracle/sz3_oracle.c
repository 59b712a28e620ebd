/*
 * oracle/sz3_oracle.c — TEST INFRASTRUCTURE ONLY (see sz3_oracle.h for the rules).
 *
 * Plain-C restatement of the reference hot path of szcompressor/SZ3 v3.3.2:
 *   SZ_compress<T> -> dispatcher -> {Lorenzo/regression blockwise | interpolation} decomposition
 *   -> LinearQuantizer -> HuffmanEncoder<int> -> Lossless_zstd, and the inverse.
 * Every function cites the reference file:line it follows (relative to /root/reference).
 * libzstd is the one third-party dependency of this path (not under /root/reference; the reference pins
 * v1.4.5 for its bundled build, tools/zstd/CMakeLists.txt:8, otherwise whatever the system has): we dlopen the
 * image's libzstd.so.1 (1.4.8) and declare the four prototypes we need ourselves (zstd.h is not in /usr/include).
 */
#define _GNU_SOURCE
#include "sz3_oracle.h"

#include <dlfcn.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------------ */
static _Thread_local char g_err[256];
/* diagnostics for the parity tests: the predictor the composed predictor chose for every block (0 Lorenzo-1, 1 Lorenzo-2,
 * 2 regression), in the order BlockwiseDecomposition visits the blocks; not part of the reference API */
static int8_t *g_sel_sink;
static size_t g_sel_cap;
void szo_debug_selection_sink(int8_t *buf, size_t cap) {
    g_sel_sink = buf;
    g_sel_cap = cap;
}
static _Thread_local int zstd_cap_error; /* mirrors std::length_error(SZ3_ERROR_COMP_BUFFER_NOT_LARGE_ENOUGH) */
static int set_err(const char *m) {
    snprintf(g_err, sizeof(g_err), "%s", m);
    return -1;
}
const char *szo_last_error(void) { return g_err; }
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* little-endian byte I/O — utils/MemoryUtil.hpp:75-145 (host is little-endian: plain memcpy) */
static void wr_bytes(uint8_t **c, const void *p, size_t n) {
    memcpy(*c, p, n);
    *c += n;
}
static void rd_bytes(const uint8_t **c, void *p, size_t n) {
    memcpy(p, *c, n);
    *c += n;
}
static void wr_u8(uint8_t **c, uint8_t v) { wr_bytes(c, &v, 1); }
static void wr_i32(uint8_t **c, int32_t v) { wr_bytes(c, &v, 4); }
static void wr_u64(uint8_t **c, uint64_t v) { wr_bytes(c, &v, 8); }
static void wr_f64(uint8_t **c, double v) { wr_bytes(c, &v, 8); }
static uint8_t rd_u8(const uint8_t **c) { uint8_t v; rd_bytes(c, &v, 1); return v; }
static int32_t rd_i32(const uint8_t **c) { int32_t v; rd_bytes(c, &v, 4); return v; }
static uint32_t rd_u32(const uint8_t **c) { uint32_t v; rd_bytes(c, &v, 4); return v; }
static uint64_t rd_u64(const uint8_t **c) { uint64_t v; rd_bytes(c, &v, 8); return v; }
static double rd_f64(const uint8_t **c) { double v; rd_bytes(c, &v, 8); return v; }
/* big-endian int32 used by the Huffman header — utils/ByteUtil.hpp:75-96, 146-151 */
static void wr_be32(uint8_t **c, uint32_t v) {
    uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
    wr_bytes(c, b, 4);
}
static int32_t rd_be32(const uint8_t **c) {
    const uint8_t *b = *c;
    *c += 4;
    return (int32_t)(((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | (uint32_t)b[3]);
}

/* ------------------------------------------------------------------------------------------------
 * Lossless_zstd (include/SZ3/lossless/Lossless_zstd.hpp:29-45): [u64 srcLen][ZSTD frame], level 3 (:48)
 * ---------------------------------------------------------------------------------------------- */
typedef size_t (*zstd_compress_fn)(void *, size_t, const void *, size_t, int);
typedef size_t (*zstd_decompress_fn)(void *, size_t, const void *, size_t);
typedef size_t (*zstd_bound_fn)(size_t);
typedef unsigned (*zstd_iserr_fn)(size_t);
typedef const char *(*zstd_ver_fn)(void);
static struct {
    void *h;
    zstd_compress_fn compress;
    zstd_decompress_fn decompress;
    zstd_bound_fn bound;
    zstd_iserr_fn is_error;
    zstd_ver_fn version;
} Z;
static int zstd_load(void) {
    if (Z.h) return 0;
    const char *names[] = {"libzstd.so.1", "libzstd.so", "/usr/lib/x86_64-linux-gnu/libzstd.so.1", NULL};
    for (int i = 0; names[i] && !Z.h; i++) Z.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!Z.h) return set_err("oracle: libzstd.so.1 not found");
    Z.compress = (zstd_compress_fn)dlsym(Z.h, "ZSTD_compress");
    Z.decompress = (zstd_decompress_fn)dlsym(Z.h, "ZSTD_decompress");
    Z.bound = (zstd_bound_fn)dlsym(Z.h, "ZSTD_compressBound");
    Z.is_error = (zstd_iserr_fn)dlsym(Z.h, "ZSTD_isError");
    Z.version = (zstd_ver_fn)dlsym(Z.h, "ZSTD_versionString");
    if (!Z.compress || !Z.decompress || !Z.bound || !Z.is_error) return set_err("oracle: libzstd symbols missing");
    return 0;
}
const char *szo_zstd_version(void) { return zstd_load() ? "" : (Z.version ? Z.version() : "?"); }
size_t szo_zstd_bound(size_t n) { return zstd_load() ? 0 : Z.bound(n); }
size_t szo_zstd_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {
    zstd_cap_error = 0;
    if (zstd_load()) return 0;
    if (cap < 8 || cap - 8 < Z.bound(n)) { /* :31-34 */
        zstd_cap_error = 1;
        set_err("The buffer for compressed data is not large enough.");
        return 0;
    }
    uint64_t len = n;
    memcpy(dst, &len, 8);
    size_t r = Z.compress(dst + 8, cap - 8, src, n, 3);
    if (Z.is_error(r)) {
        set_err("ZSTD_compress failed");
        return 0;
    }
    return r + 8;
}
size_t szo_zstd_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {
    if (zstd_load()) return 0;
    uint64_t len;
    memcpy(&len, src, 8);
    if (len > cap) {
        set_err("zstd: destination too small");
        return 0;
    }
    size_t r = Z.decompress(dst, (size_t)len, src + 8, n - 8); /* :39-45 */
    if (Z.is_error(r)) {
        set_err("ZSTD_decompress failed");
        return 0;
    }
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * HuffmanEncoder<int>  (include/SZ3/encoder/HuffmanEncoder.hpp)
 * Nodes live in a pool and are referred to by index (the reference uses pointers into the same pool).
 * ---------------------------------------------------------------------------------------------- */
typedef struct hnode {
    int32_t left, right; /* pool indices, -1 = none */
    size_t freq;
    uint8_t leaf; /* node_t::t */
    int32_t c;
} hnode;
typedef struct htree {
    uint32_t state_num;
    hnode *pool;
    int32_t n_nodes;
    int32_t *heap; /* 1-based binary min-heap of pool indices (qq) */
    int32_t qend;
    uint64_t *code; /* MSB-aligned code word per state (code[state][0] of the reference; lengths <= 64 only) */
    uint8_t *len;   /* cout */
    uint8_t *has;
    int32_t root;
    int32_t offset;      /* min symbol */
    uint32_t node_count; /* 2*leaves-1 */
} htree;
typedef struct huff_times {
    double t_tree, t_encode;
} huff_times;

static void htree_free(htree *h) {
    free(h->pool);
    free(h->heap);
    free(h->code);
    free(h->len);
    free(h->has);
    memset(h, 0, sizeof(*h));
}
/* createHuffmanTree :54-77 (pool of 2*allNodes nodes, heap of the same capacity) */
static void htree_alloc(htree *h, uint32_t state_num) {
    memset(h, 0, sizeof(*h));
    h->state_num = state_num;
    size_t all = 2 * (size_t)state_num;
    h->pool = (hnode *)calloc(all * 2 + 2, sizeof(hnode));
    h->heap = (int32_t *)calloc(all * 2 + 2, sizeof(int32_t));
    h->code = (uint64_t *)calloc(state_num, sizeof(uint64_t));
    h->len = (uint8_t *)calloc(state_num, 1);
    h->has = (uint8_t *)calloc(state_num, 1);
    h->qend = 1;
    h->root = -1;
}
/* qinsert :402-410 — sift up; a parent with freq <= the new node's freq stops the climb */
static void heap_insert(htree *h, int32_t n) {
    int32_t i = h->qend++;
    for (int32_t j = i >> 1; j; j = i >> 1) {
        if (h->pool[h->heap[j]].freq <= h->pool[n].freq) break;
        h->heap[i] = h->heap[j];
        i = j;
    }
    h->heap[i] = n;
}
/* qremove :412-432 — pop min, move last to the root, sift down preferring the right child only when strictly
 * smaller, swapping only when the parent is strictly larger */
static int32_t heap_remove(htree *h) {
    if (h->qend < 2) return -1;
    int32_t top = h->heap[1];
    h->qend--;
    h->heap[1] = h->heap[h->qend];
    int32_t i = 1;
    for (int32_t l = i << 1; l < h->qend; l = i << 1) {
        if (l + 1 < h->qend && h->pool[h->heap[l + 1]].freq < h->pool[h->heap[l]].freq) l++;
        if (h->pool[h->heap[i]].freq > h->pool[h->heap[l]].freq) {
            int32_t t = h->heap[i];
            h->heap[i] = h->heap[l];
            h->heap[l] = t;
            i = l;
        } else
            break;
    }
    return top;
}
/* build_code :441-470 — 0 to the left, 1 to the right, code stored MSB-aligned in 64 bits.
 * Iterative pre-order (the reference recurses). Codes longer than 64 bits use a second word in the reference;
 * the oracle rejects them (never seen for quantisation codes; would need > 2^44 symbols of Fibonacci skew). */
static int build_codes(htree *h) {
    typedef struct { int32_t node; int32_t len; uint64_t bits; } frame;
    frame *stack = (frame *)malloc(sizeof(frame) * (size_t)(h->n_nodes + 2));
    int sp = 0, rc = 0;
    stack[sp++] = (frame){h->root, 0, 0};
    while (sp) {
        frame f = stack[--sp];
        hnode *n = &h->pool[f.node];
        if (n->leaf) {
            if (f.len > 64) {
                rc = set_err("oracle: Huffman code longer than 64 bits");
                break;
            }
            h->code[n->c] = f.len == 0 ? 0 : f.bits << (64 - f.len);
            h->len[n->c] = (uint8_t)f.len;
            h->has[n->c] = 1;
            continue;
        }
        if (f.len >= 64) {
            rc = set_err("oracle: Huffman code longer than 64 bits");
            break;
        }
        /* push right first so that left is expanded first (order is irrelevant for the result) */
        stack[sp++] = (frame){n->right, f.len + 1, (f.bits << 1) | 1};
        stack[sp++] = (frame){n->left, f.len + 1, (f.bits << 1)};
    }
    free(stack);
    return rc;
}
/* init :516-561 — dense frequency table over [min,max], leaves inserted in increasing symbol order, then
 * repeatedly merge the two smallest (left = first removed, right = second removed) */
static int htree_build(htree *h, const int32_t *s, size_t n) {
    int32_t mx = s[0], mn = s[0];
    for (size_t i = 1; i < n; i++) {
        if (s[i] > mx) mx = s[i];
        if (s[i] < mn) mn = s[i];
    }
    uint32_t state_num = (uint32_t)(mx - mn + 2); /* :536 */
    htree_alloc(h, state_num);
    h->offset = mn;
    size_t *freq = (size_t *)calloc(state_num, sizeof(size_t));
    for (size_t i = 0; i < n; i++) freq[s[i] - mn]++;
    uint32_t leaves = 0;
    for (uint32_t i = 0; i < state_num; i++) {
        if (!freq[i]) continue;
        hnode *nd = &h->pool[h->n_nodes]; /* new_node(freq, c, 0, 0) :382-393 */
        nd->c = (int32_t)i;
        nd->freq = freq[i];
        nd->leaf = 1;
        nd->left = nd->right = -1;
        heap_insert(h, h->n_nodes++);
        leaves++;
    }
    free(freq);
    while (h->qend > 2) { /* :551-555 */
        int32_t a = heap_remove(h), b = heap_remove(h);
        hnode *nd = &h->pool[h->n_nodes];
        nd->left = a;
        nd->right = b;
        nd->freq = h->pool[a].freq + h->pool[b].freq;
        nd->leaf = 0;
        nd->c = 0;
        heap_insert(h, h->n_nodes++);
    }
    h->root = h->heap[1];
    h->node_count = leaves * 2 - 1; /* preprocess_encode :101-104 */
    return build_codes(h);
}
/* save :108-125 + convert_HuffTree_to_bytes_anyStates :601-628 + pad_tree :565-581 (pre-order numbering) */
static void htree_save(const htree *h, uint8_t **c) {
    uint32_t nc = h->node_count;
    wr_i32(c, h->offset);
    wr_be32(c, nc);
    wr_be32(c, h->state_num / 2);
    uint32_t *L = (uint32_t *)calloc(nc, 4), *R = (uint32_t *)calloc(nc, 4);
    int32_t *C = (int32_t *)calloc(nc, 4);
    uint8_t *t = (uint8_t *)calloc(nc, 1);
    /* pre-order walk: a node's index is its pre-order rank; children indices are assigned on descent */
    typedef struct { int32_t node; uint32_t idx; int stage; } frame;
    frame *stack = (frame *)malloc(sizeof(frame) * (nc + 2));
    int sp = 0;
    uint32_t counter = 0;
    stack[sp++] = (frame){h->root, 0, 0};
    C[0] = h->pool[h->root].c;
    t[0] = h->pool[h->root].leaf;
    while (sp) {
        frame *f = &stack[sp - 1];
        const hnode *n = &h->pool[f->node];
        if (f->stage == 0) {
            f->stage = 1;
            if (n->left >= 0) {
                uint32_t ci = ++counter;
                L[f->idx] = ci;
                C[ci] = h->pool[n->left].c;
                t[ci] = h->pool[n->left].leaf;
                stack[sp++] = (frame){n->left, ci, 0};
            }
        } else if (f->stage == 1) {
            f->stage = 2;
            if (n->right >= 0) {
                uint32_t ci = ++counter;
                R[f->idx] = ci;
                C[ci] = h->pool[n->right].c;
                t[ci] = h->pool[n->right].leaf;
                stack[sp++] = (frame){n->right, ci, 0};
            }
        } else
            sp--;
    }
    free(stack);
    wr_u8(c, 0); /* sysEndianType: little endian :46-52 */
    int w = nc <= 256 ? 1 : (nc <= 65536 ? 2 : 4);
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t *A = pass ? R : L;
        for (uint32_t i = 0; i < nc; i++) {
            if (w == 1) wr_u8(c, (uint8_t)A[i]);
            else if (w == 2) {
                uint16_t v = (uint16_t)A[i];
                wr_bytes(c, &v, 2);
            } else
                wr_bytes(c, &A[i], 4);
        }
    }
    wr_bytes(c, C, (size_t)nc * 4);
    wr_bytes(c, t, nc);
    free(L);
    free(R);
    free(C);
    free(t);
}
/* encode :140-218 — MSB-first concatenation of the code words, prefixed by u64 outSize = bytes touched.
 * (The reference's byte-wise writer is equivalent to a plain bit concatenation for code lengths <= 64.) */
static size_t huff_encode_bits(const htree *h, const int32_t *s, size_t n, uint8_t **c) {
    uint8_t *base = *c + 8, *p = base;
    uint64_t acc = 0; /* pending bits, MSB-aligned */
    int nacc = 0;
    for (size_t i = 0; i < n; i++) {
        int32_t st = s[i] - h->offset;
        uint64_t code = h->code[st];
        int len = h->len[st];
        while (len > 0) {
            int room = 64 - nacc, take = len < room ? len : room;
            acc |= (take == 64 ? code : (code >> (64 - take)) << (room - take));
            nacc += take;
            code = take == 64 ? 0 : code << take;
            len -= take;
            if (nacc == 64) {
                for (int b = 0; b < 8; b++) *p++ = (uint8_t)(acc >> (56 - 8 * b));
                acc = 0;
                nacc = 0;
            }
        }
    }
    int rem_bytes = (nacc + 7) / 8;
    for (int b = 0; b < rem_bytes; b++) *p++ = (uint8_t)(acc >> (56 - 8 * b));
    uint64_t out_size = (uint64_t)(p - base);
    memcpy(*c, &out_size, 8);
    *c = p;
    return (size_t)out_size;
}
/* preprocess_encode + save + encode, as used for side streams (RegressionPredictor.hpp:99-105,
 * ComposedPredictor.hpp:57-62) */
static void huffman_encode_all(const int32_t *s, size_t n, uint8_t **c) {
    htree h;
    htree_build(&h, s, n);
    htree_save(&h, c);
    huff_encode_bits(&h, s, n, c);
    htree_free(&h);
}
/* the main stream: encoder.save, write<size_t>(n), encoder.encode (SZGenericCompressor.hpp:51-57) */
static void huffman_encode_main(const int32_t *s, size_t n, uint8_t **c, huff_times *ht, uint32_t *node_count,
                                uint64_t *enc_bytes) {
    htree h;
    double t0 = now_s();
    htree_build(&h, s, n);
    double t1 = now_s();
    htree_save(&h, c);
    wr_u64(c, (uint64_t)n);
    size_t nb = huff_encode_bits(&h, s, n, c);
    double t2 = now_s();
    if (ht) {
        ht->t_tree = t1 - t0;
        ht->t_encode = t2 - t1;
    }
    if (node_count) *node_count = h.node_count;
    if (enc_bytes) *enc_bytes = nb;
    htree_free(&h);
}
/* load :261-279 + decode :225-255 (bit-serial tree walk over the serialised L/R/C/t arrays) */
typedef struct hload {
    int32_t offset;
    uint32_t nc;
    uint32_t *L, *R;
    int32_t *C;
    uint8_t *t;
} hload;
static void hload_read(hload *h, const uint8_t **c) {
    h->offset = rd_i32(c);
    h->nc = (uint32_t)rd_be32(c);
    (void)rd_be32(c); /* stateNum/2 */
    (void)rd_u8(c);   /* endian byte */
    uint32_t nc = h->nc;
    int w = nc <= 256 ? 1 : (nc <= 65536 ? 2 : 4);
    h->L = (uint32_t *)calloc(nc, 4);
    h->R = (uint32_t *)calloc(nc, 4);
    h->C = (int32_t *)calloc(nc, 4);
    h->t = (uint8_t *)calloc(nc, 1);
    for (int pass = 0; pass < 2; pass++) {
        uint32_t *A = pass ? h->R : h->L;
        for (uint32_t i = 0; i < nc; i++) {
            if (w == 1) A[i] = rd_u8(c);
            else if (w == 2) {
                uint16_t v;
                rd_bytes(c, &v, 2);
                A[i] = v;
            } else
                A[i] = rd_u32(c);
        }
    }
    rd_bytes(c, h->C, (size_t)nc * 4);
    rd_bytes(c, h->t, nc);
}
static void hload_free(hload *h) {
    free(h->L);
    free(h->R);
    free(h->C);
    free(h->t);
}
static void hload_decode(const hload *h, const uint8_t **c, size_t n, int32_t *out) {
    uint64_t enc_len = rd_u64(c);
    if (h->t[0]) { /* single-symbol tree :233-237 */
        for (size_t i = 0; i < n; i++) out[i] = h->C[0] + h->offset;
        return;
    }
    const uint8_t *b = *c;
    uint32_t node = 0;
    size_t cnt = 0;
    for (size_t i = 0; cnt < n; i++) {
        int bit = (b[i >> 3] >> (7 - (i & 7))) & 1;
        node = bit ? h->R[node] : h->L[node];
        if (h->t[node]) {
            out[cnt++] = h->C[node] + h->offset;
            node = 0;
        }
    }
    *c += enc_len;
}
static void huffman_decode_all(const uint8_t **c, size_t n, int32_t *out) {
    hload h;
    hload_read(&h, c);
    hload_decode(&h, c, n, out);
    hload_free(&h);
}
static void huffman_decode_main(const uint8_t **c, int32_t *out, size_t n_expected) {
    hload h;
    hload_read(&h, c);
    uint64_t n = rd_u64(c); /* SZGenericCompressor.hpp:76-78 */
    (void)n_expected;
    hload_decode(&h, c, (size_t)n, out);
    hload_free(&h);
}
size_t szo_huffman_encode(const int32_t *codes, size_t n, uint8_t *out, size_t cap) {
    (void)cap;
    if (n == 0) { /* :98-100 */
        set_err("Huffman bins should not be empty");
        return 0;
    }
    uint8_t *p = out;
    huffman_encode_all(codes, n, &p);
    return (size_t)(p - out);
}
size_t szo_huffman_decode(const uint8_t *in, size_t n, int32_t *codes) {
    const uint8_t *p = in;
    huffman_decode_all(&p, n, codes);
    return (size_t)(p - in);
}

/* ------------------------------------------------------------------------------------------------
 * SZ3::Config  (include/SZ3/utils/Config.hpp)
 * ---------------------------------------------------------------------------------------------- */
void szo_config_init(szo_config *c, int ndims, const uint64_t *dims) { /* ctor :146-150, setDims :161-177, defaults :452-478 */
    memset(c, 0, sizeof(*c));
    int n = 0;
    for (int i = 0; i < ndims; i++)
        if (dims[i] > 1) c->dims[n++] = dims[i];
    if (n == 0) c->dims[n++] = 1;
    c->N = n;
    c->num = 1;
    for (int i = 0; i < n; i++) c->num *= c->dims[i];
    c->predDim = (uint8_t)n;
    c->blockSize = n == 1 ? 128 : (n == 2 ? 16 : 6);
    c->cmprAlgo = SZO_ALGO_INTERP_LORENZO;
    c->errorBoundMode = SZO_EB_ABS;
    c->absErrorBound = 1e-3;
    c->quantbinCnt = 65536;
    c->dataType = SZO_FLOAT;
    c->lorenzo = 1;
    c->regression = 1;
    c->interpAlgo = SZO_INTERP_CUBIC;
    c->interpDirection = 0;
    c->interpAnchorStride = -1;
    c->interpAlpha = 1.25;
    c->interpBeta = 2.0;
}
size_t szo_config_save(const szo_config *c, uint8_t *out) { /* save :312-354 */
    uint8_t *p = out + 1;
    wr_u8(&p, (uint8_t)c->N);
    uint64_t mx = 0; /* vector_bit_width, utils/ByteUtil.hpp:195-204 */
    for (int i = 0; i < c->N; i++)
        if (c->dims[i] > mx) mx = c->dims[i];
    uint8_t bw = 0;
    while (mx > 0) {
        mx >>= 1;
        bw++;
    }
    wr_u8(&p, bw);
    /* vector2bytes, ByteUtil.hpp:206-238: values packed LSB-first, bw bits each */
    size_t total_bits = (size_t)bw * (size_t)c->N, nbytes = (total_bits + 7) / 8;
    memset(p, 0, nbytes);
    for (int i = 0; i < c->N; i++)
        for (int j = 0; j < bw; j++) {
            size_t bit = (size_t)i * bw + (size_t)j;
            if ((c->dims[i] >> j) & 1) p[bit >> 3] |= (uint8_t)(1u << (bit & 7));
        }
    p += nbytes;
    wr_u64(&p, c->num);
    wr_u8(&p, c->cmprAlgo);
    wr_u8(&p, c->errorBoundMode);
    switch (c->errorBoundMode) {
        case SZO_EB_ABS: wr_f64(&p, c->absErrorBound); break;
        case SZO_EB_REL: wr_f64(&p, c->relErrorBound); break;
        case SZO_EB_PSNR: wr_f64(&p, c->psnrErrorBound); break;
        case SZO_EB_L2NORM: wr_f64(&p, c->l2normErrorBound); break;
        case SZO_EB_ABS_OR_REL:
        case SZO_EB_ABS_AND_REL:
            wr_f64(&p, c->absErrorBound);
            wr_f64(&p, c->relErrorBound);
            break;
        default: break;
    }
    uint8_t bools = (uint8_t)((c->lorenzo & 1) << 7 | (c->lorenzo2 & 1) << 6 | (c->regression & 1) << 5 |
                              (c->regression2 & 1) << 4 | (c->openmp & 1) << 3);
    wr_u8(&p, bools);
    wr_u8(&p, c->dataType);
    wr_i32(&p, c->quantbinCnt);
    wr_i32(&p, c->blockSize);
    wr_u8(&p, c->predDim);
    out[0] = (uint8_t)(p - out);
    return (size_t)(p - out);
}
size_t szo_config_load(szo_config *c, const uint8_t *in) { /* load :361-413 */
    const uint8_t *p = in;
    uint8_t conf_size = rd_u8(&p);
    const uint8_t *end = p + conf_size; /* "c1 = c + confSize" measured after the size byte, as in the reference */
    uint64_t dummy = 1;
    szo_config_init(c, 1, &dummy); /* a default-constructed Config receives the loaded fields */
    c->N = (int8_t)rd_u8(&p);
    uint8_t bw = rd_u8(&p);
    size_t total_bits = (size_t)bw * (size_t)c->N, nbytes = (total_bits + 7) / 8;
    for (int i = 0; i < c->N && i < 4; i++) { /* bytes2vector, ByteUtil.hpp:240-264 */
        uint64_t v = 0;
        for (int j = 0; j < bw; j++) {
            size_t bit = (size_t)i * bw + (size_t)j;
            v |= (uint64_t)((p[bit >> 3] >> (bit & 7)) & 1) << j;
        }
        c->dims[i] = v;
    }
    p += nbytes;
    c->num = rd_u64(&p);
    c->cmprAlgo = rd_u8(&p);
    c->errorBoundMode = rd_u8(&p);
    switch (c->errorBoundMode) {
        case SZO_EB_ABS: c->absErrorBound = rd_f64(&p); break;
        case SZO_EB_REL: c->relErrorBound = rd_f64(&p); break;
        case SZO_EB_PSNR: c->psnrErrorBound = rd_f64(&p); break;
        case SZO_EB_L2NORM: c->l2normErrorBound = rd_f64(&p); break;
        case SZO_EB_ABS_OR_REL:
        case SZO_EB_ABS_AND_REL:
            c->absErrorBound = rd_f64(&p);
            c->relErrorBound = rd_f64(&p);
            break;
        default: break;
    }
    if (p < end) {
        uint8_t b = rd_u8(&p);
        c->lorenzo = (b >> 7) & 1;
        c->lorenzo2 = (b >> 6) & 1;
        c->regression = (b >> 5) & 1;
        c->regression2 = (b >> 4) & 1;
        c->openmp = (b >> 3) & 1;
    }
    if (p < end) c->dataType = rd_u8(&p);
    if (p < end) c->quantbinCnt = rd_i32(&p);
    if (p < end) c->blockSize = rd_i32(&p);
    if (p < end) c->predDim = rd_u8(&p);
    return (size_t)(p - in);
}

/* number of OpenMP "threads" = slabs the serial oracle emulates (SZImplOMP.hpp:27-36) */
static int g_omp_slabs = 8;
void szo_set_omp_slabs(int n) { g_omp_slabs = n > 0 ? n : 1; }

/* ------------------------------------------------------------------------------------------------
 * type-generic part, instantiated for float and double
 * ---------------------------------------------------------------------------------------------- */
#define T float
#define SUF(x) x##_f32
#include "sz3_oracle_impl.h"
#undef T
#undef SUF
#define T double
#define SUF(x) x##_f64
#include "sz3_oracle_impl.h"
#undef T
#undef SUF

int32_t szo_quantize_f32(float *data, float pred, double eb, int32_t radius) {
    quantizer_f32 q;
    quantizer_init_f32(&q, eb, radius);
    int32_t c = quantize_and_overwrite_f32(&q, data, pred);
    quantizer_free_f32(&q);
    return c;
}
int32_t szo_quantize_f64(double *data, double pred, double eb, int32_t radius) {
    quantizer_f64 q;
    quantizer_init_f64(&q, eb, radius);
    int32_t c = quantize_and_overwrite_f64(&q, data, pred);
    quantizer_free_f64(&q);
    return c;
}
float szo_recover_f32(float pred, int32_t code, double eb, int32_t radius) {
    quantizer_f32 q;
    quantizer_init_f32(&q, eb, radius);
    return quantizer_recover_f32(&q, pred, code);
}
double szo_recover_f64(double pred, int32_t code, double eb, int32_t radius) {
    quantizer_f64 q;
    quantizer_init_f64(&q, eb, radius);
    return quantizer_recover_f64(&q, pred, code);
}

/* SZ_compress_size_bound, api/impl/SZImpl.hpp:34-44 and SZImplOMP.hpp:189-209 */
size_t szo_compress_bound(const szo_config *c, int dtype) {
    size_t es = dtype == SZO_FLOAT ? 4 : 8;
    uint8_t tmp[128];
    size_t ce = szo_config_save(c, tmp);
    if (c->openmp) {
        size_t nt = (size_t)g_omp_slabs;
        if (c->dims[0] < nt) nt = (size_t)c->dims[0];
        size_t chunk = (size_t)(c->dims[0] / nt * (c->num / c->dims[0]));
        size_t last = (size_t)((c->dims[0] - c->dims[0] / nt * (nt - 1)) * (c->num / c->dims[0]));
        return 4096 + 4 + nt * ce + nt * 8 + (nt - 1) * szo_zstd_bound(chunk * es) + szo_zstd_bound(last * es);
    }
    return 4096 + ce + szo_zstd_bound((size_t)c->num * es);
}

/* SZ_compress<T>, api/sz.hpp:43-82 */
size_t szo_compress(const szo_config *config, int dtype, const void *data, uint8_t *out, size_t cap, szo_stats *st) {
    szo_config conf = *config;
    if (st) memset(st, 0, sizeof(*st));
    if (zstd_load()) return 0;
    if (conf.N < 1 || conf.N > 4) {
        set_err("Data dimension higher than 4 is not supported.");
        return 0;
    }
    if (cap < szo_compress_bound(&conf, dtype)) {
        set_err("The buffer for compressed data is not large enough.");
        return 0;
    }
    uint8_t *p = out;
    uint32_t magic = 0xF342F310u, ver = (3u << 24) | (3u << 16) | (2u << 8); /* version.hpp.in:10,21-26 with 3.3.2 */
    wr_bytes(&p, &magic, 4);
    wr_bytes(&p, &ver, 4);
    uint8_t *size_pos = p;
    p += 8;
    uint8_t tmp[128];
    size_t cmp_cap = cap - 16 - szo_config_save(&conf, tmp) * 2; /* :58 */
    size_t sz;
    if (conf.openmp) /* SZ_compress_impl, api/impl/SZImpl.hpp:10-20 */
        sz = dtype == SZO_FLOAT ? compress_omp_f32(&conf, (const float *)data, p, cmp_cap, g_omp_slabs, st)
                                : compress_omp_f64(&conf, (const double *)data, p, cmp_cap, g_omp_slabs, st);
    else
        sz = dtype == SZO_FLOAT ? compress_dispatch_f32(&conf, (const float *)data, p, cmp_cap, st)
                                : compress_dispatch_f64(&conf, (const double *)data, p, cmp_cap, st);
    if (!sz) return 0;
    uint64_t sz64 = sz;
    memcpy(size_pos, &sz64, 8);
    p += sz;
    p += szo_config_save(&conf, p);
    return (size_t)(p - out);
}

/* SZ_decompress<T>, api/sz.hpp:117-157 */
size_t szo_decompress(int dtype, const uint8_t *cmp, size_t cmp_size, void *dec, szo_config *conf_out) {
    if (zstd_load()) return 0;
    const uint8_t *p = cmp;
    uint32_t magic = rd_u32(&p);
    if (magic != 0xF342F310u) {
        set_err("magic number mismatch, the input data is not compressed by SZ3");
        return 0;
    }
    uint32_t ver = rd_u32(&p);
    if ((ver >> 8) != ((3u << 16) | (3u << 8) | 2u)) {
        set_err("data version mismatch");
        return 0;
    }
    uint64_t payload = rd_u64(&p);
    szo_config conf;
    szo_config_load(&conf, p + payload);
    (void)cmp_size;
    int rc;
    if (conf.openmp)
        rc = dtype == SZO_FLOAT ? decompress_omp_f32(&conf, p, (size_t)payload, (float *)dec)
                                : decompress_omp_f64(&conf, p, (size_t)payload, (double *)dec);
    else
        rc = dtype == SZO_FLOAT ? decompress_dispatch_f32(&conf, p, (size_t)payload, (float *)dec)
                                : decompress_dispatch_f64(&conf, p, (size_t)payload, (double *)dec);
    if (conf_out) *conf_out = conf;
    return rc ? 0 : (size_t)conf.num;
}

size_t szo_decomposition_codes(const szo_config *config, int dtype, const void *data, int32_t *codes) {
    szo_config conf = *config;
    szo_stats st;
    memset(&st, 0, sizeof(st));
    if (dtype == SZO_FLOAT) {
        if (cal_abs_eb_f32(&conf, (const float *)data)) return (size_t)-1;
        if (conf.cmprAlgo == SZO_ALGO_LORENZO_REG) compress_lorenzo_reg_f32(&conf, (const float *)data, NULL, 0, &st, codes);
        else if (conf.cmprAlgo == SZO_ALGO_INTERP) compress_interp_f32(&conf, (const float *)data, NULL, 0, &st, codes);
        else return (size_t)-1;
    } else {
        if (cal_abs_eb_f64(&conf, (const double *)data)) return (size_t)-1;
        if (conf.cmprAlgo == SZO_ALGO_LORENZO_REG) compress_lorenzo_reg_f64(&conf, (const double *)data, NULL, 0, &st, codes);
        else if (conf.cmprAlgo == SZO_ALGO_INTERP) compress_interp_f64(&conf, (const double *)data, NULL, 0, &st, codes);
        else return (size_t)-1;
    }
    return (size_t)st.n_unpred;
}

int szo_tune_interp_lorenzo(szo_config *conf, int dtype, const void *data, szo_tuner_report *rep) {
    if (conf->cmprAlgo != SZO_ALGO_INTERP_LORENZO) return -1;
    if (dtype == SZO_FLOAT) {
        if (cal_abs_eb_f32(conf, (const float *)data)) return -1;
        return tune_interp_lorenzo_f32(conf, (const float *)data, rep);
    }
    if (cal_abs_eb_f64(conf, (const double *)data)) return -1;
    return tune_interp_lorenzo_f64(conf, (const double *)data, rep);
}

size_t szo_interp_codes(const szo_config *config, int dtype, const void *data, int32_t *codes, uint64_t *order, void *recon) {
    szo_config conf = *config;
    if (conf.interpAnchorStride < 0) {
        static const int def[4] = {4096, 128, 32, 16};
        conf.interpAnchorStride = def[conf.N - 1];
    }
    if (dtype == SZO_FLOAT) {
        if (cal_abs_eb_f32(&conf, (const float *)data)) return (size_t)-1;
        return interp_codes_f32(&conf, (const float *)data, codes, order, (float *)recon);
    }
    if (cal_abs_eb_f64(&conf, (const double *)data)) return (size_t)-1;
    return interp_codes_f64(&conf, (const double *)data, codes, order, (double *)recon);
}
