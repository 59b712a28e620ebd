/*
 * oracle/sz3_oracle_interp.h — TEST INFRASTRUCTURE ONLY (see sz3_oracle.h).
 * Restatement of InterpolationDecomposition<T,N,LinearQuantizer<T>> and SZ_compress_Interp / SZ_decompress_Interp
 * (include/SZ3/decomposition/InterpolationDecomposition.hpp, include/SZ3/api/impl/SZAlgoInterp.hpp:17-40,
 * include/SZ3/utils/Interpolators.hpp:12-39).  Type-generic: included from sz3_oracle_impl.h with T / SUF defined.
 */

/* Interpolators.hpp:12-39 — arithmetic in T with int multipliers, except interp_linear1 (double constants) */
static inline T SUF(ip_linear)(T a, T b) { return (T)((T)(a + b) / 2); }
static inline T SUF(ip_linear1)(T a, T b) { return (T)(-0.5 * (double)a + 1.5 * (double)b); }
static inline T SUF(ip_quad_1)(T a, T b, T c) { return (T)((T)((T)((T)(3 * a) + (T)(6 * b)) - c) / 8); }
static inline T SUF(ip_quad_2)(T a, T b, T c) { return (T)((T)((T)((T)(-a) + (T)(6 * b)) + (T)(3 * c)) / 8); }
static inline T SUF(ip_quad_3)(T a, T b, T c) { return (T)((T)((T)((T)(3 * a) - (T)(10 * b)) + (T)(15 * c)) / 8); }
static inline T SUF(ip_cubic)(T a, T b, T c, T d) { return (T)((T)((T)((T)((T)(-a) + (T)(9 * b)) + (T)(9 * c)) - d) / 16); }

typedef struct SUF(interp) {
    int N;
    size_t dims[4], off[4]; /* original_dimensions, original_dim_offsets */
    int interp_level, interp_id, direction;
    uint32_t blocksize;
    size_t anchor_stride;
    double alpha, beta;
    size_t num;
    int seqs[24][4], nseq; /* dim_sequences (std::next_permutation order) */
    SUF(quantizer) q;
    int32_t *codes; /* quant_inds */
    size_t qi;      /* quant_index */
    uint64_t *order; /* optional: element index of every emitted code (oracle extension for the GPU parity tests) */
    int decompress;
    T *base;
} SUF(interp);

/* the quantize_func lambdas of compress (:135-139) and decompress (:68-70) */
static inline void SUF(ip_emit)(SUF(interp) * c, T *d, T pred) {
    if (c->decompress) {
        *d = SUF(quantizer_recover)(&c->q, pred, c->codes[c->qi++]);
    } else {
        if (c->order) c->order[c->qi] = (uint64_t)(d - c->base);
        c->codes[c->qi++] = SUF(quantize_and_overwrite)(&c->q, d, pred);
    }
}

/* init, InterpolationDecomposition.hpp:176-213 */
static void SUF(ip_init)(SUF(interp) * c) {
    c->qi = 0;
    c->num = 1;
    c->interp_level = -1;
    int use_anchor = 0;
    for (int i = 0; i < c->N; i++) {
        int lv = (int)ceil(log2((double)c->dims[i]));
        if (c->interp_level < lv) c->interp_level = lv;
        if (c->dims[i] > c->anchor_stride) use_anchor = 1;
        c->num *= c->dims[i];
    }
    if (!use_anchor) c->anchor_stride = 0;
    if (c->anchor_stride > 0) {
        int maxl = (int)log2((double)c->anchor_stride) + 1;
        if (maxl <= c->interp_level) c->interp_level = maxl;
    }
    c->off[c->N - 1] = 1;
    for (int i = c->N - 2; i >= 0; i--) c->off[i] = c->off[i + 1] * c->dims[i + 1];
    /* all permutations of (0..N-1) in lexicographic order = std::next_permutation sequence */
    int p[4] = {0, 1, 2, 3};
    c->nseq = 0;
    for (;;) {
        memcpy(c->seqs[c->nseq++], p, sizeof(p));
        int i = c->N - 2;
        while (i >= 0 && p[i] > p[i + 1]) i--;
        if (i < 0) break;
        int j = c->N - 1;
        while (p[j] < p[i]) j--;
        int t = p[i]; p[i] = p[j]; p[j] = t;
        for (int a = i + 1, b = c->N - 1; a < b; a++, b--) { t = p[a]; p[a] = p[b]; p[b] = t; }
    }
}

/* free function foreach<T,N> (utils/BlockwiseIterator.hpp:282-322): nested loops, dim 0 outermost, ends exclusive */
#define IP_FOREACH(c, data, offset, begins, ends, strides, doffs, BODY)                                       \
    do {                                                                                                      \
        size_t b_[4] = {0, 0, 0, 0}, e_[4] = {1, 1, 1, 1}, s_[4] = {1, 1, 1, 1}, o_[4] = {0, 0, 0, 0};         \
        int sh_ = 4 - (c)->N;                                                                                 \
        for (int k_ = 0; k_ < (c)->N; k_++) {                                                                 \
            b_[sh_ + k_] = (begins)[k_];                                                                      \
            e_[sh_ + k_] = (ends)[k_];                                                                        \
            s_[sh_ + k_] = (strides)[k_];                                                                     \
            o_[sh_ + k_] = (doffs)[k_];                                                                       \
        }                                                                                                     \
        for (size_t i0_ = b_[0]; i0_ < e_[0]; i0_ += s_[0])                                                   \
            for (size_t i1_ = b_[1]; i1_ < e_[1]; i1_ += s_[1])                                               \
                for (size_t i2_ = b_[2]; i2_ < e_[2]; i2_ += s_[2])                                           \
                    for (size_t i3_ = b_[3]; i3_ < e_[3]; i3_ += s_[3]) {                                     \
                        T *d = (data) + (offset) + i0_ * o_[0] + i1_ * o_[1] + i2_ * o_[2] + i3_ * o_[3];     \
                        BODY                                                                                  \
                    }                                                                                         \
    } while (0)

/* build_anchor_grid / recover_anchor_grid, :215-233 */
static void SUF(ip_anchors)(SUF(interp) * c, T *data) {
    size_t begins[4] = {0, 0, 0, 0}, strides[4];
    for (int i = 0; i < 4; i++) strides[i] = c->anchor_stride;
    IP_FOREACH(c, data, (size_t)0, begins, c->dims, strides, c->off, {
        if (c->decompress) {
            *d = c->q.unpred[c->q.index++]; /* recover_unpred */
            c->qi++;
        } else {
            if (c->order) c->order[c->qi] = (uint64_t)(d - c->base);
            SUF(quantizer_push_unpred)(&c->q, *d); /* force_save_unpred returns 0 */
            c->codes[c->qi++] = 0;
        }
    });
}

/* interpolation_1d (old API, N <= 2), :248-293 */
static void SUF(ip_1d)(SUF(interp) * c, T *data, size_t begin, size_t end, size_t stride) {
    size_t n = (end - begin) / stride + 1;
    if (n <= 1) return;
    size_t s3 = 3 * stride, s5 = 5 * stride;
    if (c->interp_id == 0 || n < 5) {
        for (size_t i = 1; i + 1 < n; i += 2) {
            T *d = data + begin + i * stride;
            SUF(ip_emit)(c, d, SUF(ip_linear)(*(d - stride), *(d + stride)));
        }
        if (n % 2 == 0) {
            T *d = data + begin + (n - 1) * stride;
            if (n < 4) SUF(ip_emit)(c, d, *(d - stride));
            else SUF(ip_emit)(c, d, SUF(ip_linear1)(*(d - s3), *(d - stride)));
        }
    } else {
        T *d;
        size_t i;
        for (i = 3; i + 3 < n; i += 2) {
            d = data + begin + i * stride;
            SUF(ip_emit)(c, d, SUF(ip_cubic)(*(d - s3), *(d - stride), *(d + stride), *(d + s3)));
        }
        d = data + begin + stride;
        SUF(ip_emit)(c, d, SUF(ip_quad_1)(*(d - stride), *(d + stride), *(d + s3)));
        d = data + begin + i * stride;
        SUF(ip_emit)(c, d, SUF(ip_quad_2)(*(d - s3), *(d - stride), *(d + stride)));
        if (n % 2 == 0) {
            d = data + begin + (n - 1) * stride;
            SUF(ip_emit)(c, d, SUF(ip_quad_3)(*(d - s5), *(d - s3), *(d - stride)));
        }
    }
}

/* interpolation_1d_fastest_dim_first (new API, N >= 3), :310-402. `strides` is modified like the reference's
 * by-reference parameter */
static void SUF(ip_fdf)(SUF(interp) * c, T *data, const size_t *begin_idx, const size_t *end_idx, int direction,
                        size_t *strides, size_t math_stride) {
    int N = c->N;
    for (int i = 0; i < N; i++)
        if (end_idx[i] < begin_idx[i]) return;
    size_t mb = begin_idx[direction], me = end_idx[direction];
    size_t n = (me - mb) / math_stride + 1;
    if (n <= 1) return;
    size_t offset = 0, stride = math_stride * c->off[direction];
    size_t begins[4], ends[4], doffs[4];
    for (int i = 0; i < N; i++) {
        begins[i] = 0;
        ends[i] = end_idx[i] - begin_idx[i] + 1;
        doffs[i] = c->off[i];
        offset += c->off[i] * begin_idx[i];
    }
    doffs[direction] = stride;
    size_t s2 = 2 * stride;
    if (c->interp_id == 0) {
        begins[direction] = 1;
        ends[direction] = n - 1;
        strides[direction] = 2;
        IP_FOREACH(c, data, offset, begins, ends, strides, doffs,
                   { SUF(ip_emit)(c, d, SUF(ip_linear)(*(d - stride), *(d + stride))); });
        if (n % 2 == 0) {
            begins[direction] = n - 1;
            ends[direction] = n;
            IP_FOREACH(c, data, offset, begins, ends, strides, doffs, {
                if (n < 3) SUF(ip_emit)(c, d, *(d - stride));
                else SUF(ip_emit)(c, d, SUF(ip_linear1)(*(d - s2), *(d - stride)));
            });
        }
    } else {
        size_t s3 = 3 * stride;
        begins[direction] = 3;
        ends[direction] = (n >= 3) ? (n - 3) : 0;
        strides[direction] = 2;
        IP_FOREACH(c, data, offset, begins, ends, strides, doffs,
                   { SUF(ip_emit)(c, d, SUF(ip_cubic)(*(d - s3), *(d - stride), *(d + stride), *(d + s3))); });
        size_t bnd[4];
        int nb = 0;
        bnd[nb++] = 1;
        if (n % 2 == 1 && n > 3) bnd[nb++] = n - 2;
        if (n % 2 == 0 && n > 4) bnd[nb++] = n - 3;
        if (n % 2 == 0 && n > 2) bnd[nb++] = n - 1;
        for (int k = 0; k < nb; k++) {
            size_t boundary = bnd[k];
            begins[direction] = boundary;
            ends[direction] = boundary + 1;
            IP_FOREACH(c, data, offset, begins, ends, strides, doffs, {
                if (boundary >= 3) {
                    if (boundary + 3 < n) SUF(ip_emit)(c, d, SUF(ip_cubic)(*(d - s3), *(d - stride), *(d + stride), *(d + s3)));
                    else if (boundary + 1 < n) SUF(ip_emit)(c, d, SUF(ip_quad_2)(*(d - s3), *(d - stride), *(d + stride)));
                    else SUF(ip_emit)(c, d, SUF(ip_linear1)(*(d - s3), *(d - stride)));
                } else {
                    if (boundary + 3 < n) SUF(ip_emit)(c, d, SUF(ip_quad_1)(*(d - stride), *(d + stride), *(d + s3)));
                    else if (boundary + 1 < n) SUF(ip_emit)(c, d, SUF(ip_linear)(*(d - stride), *(d + stride)));
                    else SUF(ip_emit)(c, d, *(d - stride));
                }
            });
        }
    }
}

/* interpolation, :405-454 */
static void SUF(ip_block)(SUF(interp) * c, T *data, const size_t *begin, const size_t *end, size_t stride) {
    const int *dims = c->seqs[c->direction];
    size_t s2 = stride * 2;
    if (c->N == 1) {
        SUF(ip_1d)(c, data, begin[0], end[0], stride);
    } else if (c->N == 2) {
        for (size_t j = (begin[dims[1]] ? begin[dims[1]] + s2 : 0); j <= end[dims[1]]; j += s2) {
            size_t bo = begin[dims[0]] * c->off[dims[0]] + j * c->off[dims[1]];
            SUF(ip_1d)(c, data, bo, bo + (end[dims[0]] - begin[dims[0]]) * c->off[dims[0]], stride * c->off[dims[0]]);
        }
        for (size_t i = (begin[dims[0]] ? begin[dims[0]] + stride : 0); i <= end[dims[0]]; i += stride) {
            size_t bo = i * c->off[dims[0]] + begin[dims[1]] * c->off[dims[1]];
            SUF(ip_1d)(c, data, bo, bo + (end[dims[1]] - begin[dims[1]]) * c->off[dims[1]], stride * c->off[dims[1]]);
        }
    } else {
        size_t strides[4], bi[4], ei[4];
        for (int i = 0; i < c->N; i++) {
            bi[i] = begin[i];
            ei[i] = end[i];
        }
        strides[dims[0]] = 1;
        for (int i = 1; i < c->N; i++) {
            bi[dims[i]] = begin[dims[i]] ? begin[dims[i]] + s2 : 0;
            strides[dims[i]] = s2;
        }
        SUF(ip_fdf)(c, data, bi, ei, dims[0], strides, stride);
        for (int i = 1; i < c->N; i++) {
            bi[dims[i]] = begin[dims[i]];
            bi[dims[i - 1]] = begin[dims[i - 1]] ? begin[dims[i - 1]] + stride : 0;
            strides[dims[i - 1]] = stride;
            SUF(ip_fdf)(c, data, bi, ei, dims[i], strides, stride);
        }
    }
}

/* the level / block loops shared by compress (:79-147) and decompress (:26-76) */
static void SUF(ip_run)(SUF(interp) * c, T *data) {
    SUF(ip_init)(c);
    c->base = data;
    double eb = c->q.eb;
    if (c->anchor_stride == 0) {
        if (c->decompress) {
            *data = SUF(quantizer_recover)(&c->q, 0, c->codes[c->qi++]);
        } else {
            if (c->order) c->order[c->qi] = 0;
            c->codes[c->qi++] = SUF(quantize_and_overwrite)(&c->q, data, 0);
        }
    } else {
        SUF(ip_anchors)(c, data);
        c->interp_level--;
    }
    int top = c->interp_level;
    for (int level = top; level > 0 && level <= top; level--) {
        double cur_eb = eb;
        if (c->alpha < 0) {
            cur_eb = level >= 3 ? eb * 0.5 : eb; /* eb_ratio = 0.5 (:475) */
        } else if (c->alpha >= 1) {
            double r = pow(c->alpha, level - 1);
            if (r > c->beta) r = c->beta;
            cur_eb = eb / r;
        }
        c->q.eb = cur_eb; /* quantizer.set_eb */
        c->q.eb_recip = 1.0 / cur_eb;
        size_t stride = (size_t)1 << (level - 1);
        size_t bsz = (size_t)c->blocksize * stride;
        /* multi_dimensional_range over block origins, row-major (utils/Iterator.hpp:72-83, 241-252) */
        size_t nb[4] = {1, 1, 1, 1}, b[4] = {0, 0, 0, 0};
        for (int i = 0; i < c->N; i++) nb[i] = (c->dims[i] - 1) / bsz + 1;
        size_t total = 1;
        for (int i = 0; i < c->N; i++) total *= nb[i];
        for (size_t t = 0; t < total; t++) {
            size_t begin[4], end[4];
            for (int i = 0; i < c->N; i++) {
                begin[i] = b[i] * bsz;
                end[i] = begin[i] + bsz;
                if (end[i] > c->dims[i] - 1) end[i] = c->dims[i] - 1;
            }
            SUF(ip_block)(c, data, begin, end, stride);
            for (int i = c->N - 1; i >= 0; i--) {
                if (++b[i] < nb[i]) break;
                b[i] = 0;
            }
        }
    }
    c->q.eb = eb;
    c->q.eb_recip = 1.0 / eb;
}

/* save / load, :149-171: [u64 dims x N][u32 blocksize][i32 interp_id][i32 direction][u64 anchor_stride][f64 a][f64 b] quantizer */
static void SUF(ip_save)(SUF(interp) * c, uint8_t **p) {
    for (int i = 0; i < c->N; i++) wr_u64(p, (uint64_t)c->dims[i]);
    wr_bytes(p, &c->blocksize, 4);
    wr_i32(p, c->interp_id);
    wr_i32(p, c->direction);
    wr_u64(p, (uint64_t)c->anchor_stride);
    wr_f64(p, c->alpha);
    wr_f64(p, c->beta);
    SUF(quantizer_save)(&c->q, p);
}
static int SUF(ip_load)(SUF(interp) * c, const uint8_t **p) {
    for (int i = 0; i < c->N; i++) c->dims[i] = (size_t)rd_u64(p);
    c->blocksize = rd_u32(p);
    c->interp_id = rd_i32(p);
    c->direction = rd_i32(p);
    c->anchor_stride = (size_t)rd_u64(p);
    c->alpha = rd_f64(p);
    c->beta = rd_f64(p);
    return SUF(quantizer_load)(&c->q, p);
}

/* decomposition only: codes in emission order (+ optional element index of every code); returns #unpredictable */
static size_t SUF(interp_codes)(const szo_config *conf, const T *data, int32_t *codes, uint64_t *order, T *recon) {
    SUF(interp) c;
    memset(&c, 0, sizeof(c));
    c.N = conf->N;
    for (int i = 0; i < c.N; i++) c.dims[i] = (size_t)conf->dims[i];
    c.interp_id = conf->interpAlgo;
    c.direction = conf->interpDirection;
    c.anchor_stride = (size_t)conf->interpAnchorStride;
    c.blocksize = 32;
    c.alpha = conf->interpAlpha;
    c.beta = conf->interpBeta;
    SUF(quantizer_init)(&c.q, conf->absErrorBound, conf->quantbinCnt / 2);
    size_t n = (size_t)conf->num;
    T *work = recon ? recon : (T *)malloc(n * sizeof(T));
    memcpy(work, data, n * sizeof(T));
    c.codes = codes;
    c.order = order;
    SUF(ip_run)(&c, work);
    size_t nun = c.q.n_unpred;
    if (!recon) free(work);
    SUF(quantizer_free)(&c.q);
    return nun;
}

/* SZ_compress_Interp (api/impl/SZAlgoInterp.hpp:17-30) through SZGenericCompressor::compress */
static size_t SUF(compress_interp)(szo_config *conf, const T *data, uint8_t *out, size_t cap, szo_stats *st,
                                   int32_t *codes_out) {
    if (conf->interpAnchorStride < 0) { /* default anchor stride :20-24 */
        static const int def[4] = {4096, 128, 32, 16};
        conf->interpAnchorStride = def[conf->N - 1];
    }
    SUF(interp) c;
    memset(&c, 0, sizeof(c));
    c.N = conf->N;
    for (int i = 0; i < c.N; i++) c.dims[i] = (size_t)conf->dims[i];
    c.interp_id = conf->interpAlgo;
    c.direction = conf->interpDirection;
    c.anchor_stride = (size_t)conf->interpAnchorStride;
    c.blocksize = 32;
    c.alpha = conf->interpAlpha;
    c.beta = conf->interpBeta;
    SUF(quantizer_init)(&c.q, conf->absErrorBound, conf->quantbinCnt / 2);
    size_t n = (size_t)conf->num;
    T *work = (T *)malloc(n * sizeof(T)); /* the dispatcher's dataCopy (SZDispatcher.hpp:27) */
    memcpy(work, data, n * sizeof(T));
    int32_t *codes = codes_out ? codes_out : (int32_t *)malloc(n * sizeof(int32_t));
    c.codes = codes;
    double t0 = now_s();
    SUF(ip_run)(&c, work);
    double t1 = now_s();
    free(work);
    size_t result = 0;
    if (st) {
        st->n_unpred = c.q.n_unpred;
        st->t_decomp = t1 - t0;
    }
    if (out) {
        size_t bufsz = 4096 + 2 * (sizeof(T) * n + c.q.n_unpred * sizeof(T)) + 16 * n;
        uint8_t *buf = (uint8_t *)malloc(bufsz), *p = buf;
        SUF(ip_save)(&c, &p);
        huff_times ht = {0, 0};
        uint32_t node_count = 0;
        uint64_t enc_bytes = 0;
        huffman_encode_main(codes, n, &p, &ht, &node_count, &enc_bytes);
        double t2 = now_s();
        if (st) {
            st->raw_bytes = (uint64_t)(p - buf);
            st->huff_bytes = enc_bytes;
            st->huff_node_count = node_count;
            st->t_hist_tree = ht.t_tree;
            st->t_encode = ht.t_encode;
        }
        result = szo_zstd_compress(buf, (size_t)(p - buf), out, cap);
        if (st) st->t_zstd = now_s() - t2;
        free(buf);
    }
    if (!codes_out) free(codes);
    SUF(quantizer_free)(&c.q);
    return result;
}

/* SZ_decompress_Interp (SZAlgoInterp.hpp:32-40) through SZGenericCompressor::decompress */
static int SUF(decompress_interp)(const szo_config *conf, const uint8_t *cmp, size_t cmp_size, T *dec) {
    uint64_t raw_len;
    memcpy(&raw_len, cmp, 8);
    uint8_t *raw = (uint8_t *)malloc(raw_len ? raw_len : 1);
    if (szo_zstd_decompress(cmp, cmp_size, raw, raw_len) != raw_len) {
        free(raw);
        return set_err("zstd decompress failed");
    }
    SUF(interp) c;
    memset(&c, 0, sizeof(c));
    c.N = conf->N;
    SUF(quantizer_init)(&c.q, conf->absErrorBound, conf->quantbinCnt / 2);
    const uint8_t *p = raw;
    int rc = SUF(ip_load)(&c, &p);
    if (!rc) {
        size_t n = (size_t)conf->num;
        int32_t *codes = (int32_t *)malloc(n * sizeof(int32_t));
        huffman_decode_main(&p, codes, n);
        c.codes = codes;
        c.decompress = 1;
        SUF(ip_run)(&c, dec);
        free(codes);
    }
    SUF(quantizer_free)(&c.q);
    free(raw);
    return rc;
}
