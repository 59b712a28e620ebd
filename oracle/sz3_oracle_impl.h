/*
 * oracle/sz3_oracle_impl.h — TEST INFRASTRUCTURE ONLY (see sz3_oracle.h).
 * Type-generic part of the restatement; included twice by sz3_oracle.c with
 *     #define T float  / SUF(x) x##_f32      and      #define T double / SUF(x) x##_f64
 * All floating-point expressions keep the reference's operand types and evaluation order
 * (built with -ffp-contract=off; the reference build has no FMA on baseline x86-64 either).
 */

/* ---------------------------------------------------------------------------------------------
 * LinearQuantizer<T>  (include/SZ3/quantizer/LinearQuantizer.hpp)
 * ------------------------------------------------------------------------------------------- */
typedef struct SUF(quantizer) {
    double eb, eb_recip; /* :129-130 */
    int32_t radius;      /* :131 */
    T *unpred;           /* :124 std::vector<T> unpred */
    size_t n_unpred, cap_unpred;
    size_t index; /* :125 decompression cursor */
} SUF(quantizer);

static void SUF(quantizer_init)(SUF(quantizer) * q, double eb, int32_t radius) { /* :23-29 */
    memset(q, 0, sizeof(*q));
    q->eb = eb;
    q->eb_recip = 1.0 / eb;
    q->radius = radius;
}
static void SUF(quantizer_free)(SUF(quantizer) * q) {
    free(q->unpred);
    q->unpred = NULL;
}
static void SUF(quantizer_push_unpred)(SUF(quantizer) * q, T v) {
    if (q->n_unpred == q->cap_unpred) {
        q->cap_unpred = q->cap_unpred ? q->cap_unpred * 2 : 64;
        q->unpred = (T *)realloc(q->unpred, q->cap_unpred * sizeof(T));
    }
    q->unpred[q->n_unpred++] = v;
}

/* quantize_and_overwrite, LinearQuantizer.hpp:43-71.  Operand types as in the reference:
 * diff in T; |diff|*recip in double; cast to int64 (x86 cvttsd2si semantics for NaN/overflow);
 * reconstruction pred + q*eb in double then rounded to T; check |dec-data| (T) <= eb (double). */
static inline int32_t SUF(quantize_and_overwrite)(SUF(quantizer) * q, T *data, T pred) {
    T diff = *data - pred;
    double scaled = fabs((double)diff) * q->eb_recip;
    int64_t quant_index;
    /* C makes an out-of-range double->int64 conversion undefined; the reference binary (x86-64) yields
     * INT64_MIN there and for NaN, after which "+1" gives a negative index that enters the branch and
     * fails the bound check.  State that explicitly so the restatement is well-defined C. */
    if (!(scaled < 9223372036854775808.0)) /* also true for NaN */
        quant_index = INT64_MIN + 1;
    else
        quant_index = (int64_t)scaled + 1;
    if (quant_index < (int64_t)q->radius * 2) {
        quant_index >>= 1;
        int32_t half_index = (int32_t)quant_index;
        quant_index = (int64_t)((uint64_t)quant_index << 1);
        int32_t shifted;
        if (diff < 0) {
            quant_index = -quant_index;
            shifted = q->radius - half_index;
        } else {
            shifted = q->radius + half_index;
        }
        T dec = (T)((double)pred + (double)quant_index * q->eb);
        T adiff = (T)fabs((double)(T)(dec - *data));
        if ((double)adiff <= q->eb) {
            *data = dec;
            return shifted;
        }
        SUF(quantizer_push_unpred)(q, *data);
        return 0;
    }
    SUF(quantizer_push_unpred)(q, *data);
    return 0;
}

/* recover, LinearQuantizer.hpp:74-86: pred + 2*(code-radius)*eb, int product then double math, cast to T */
static inline T SUF(quantizer_recover)(SUF(quantizer) * q, T pred, int32_t code) {
    if (code) return (T)((double)pred + (double)(2 * (code - q->radius)) * q->eb);
    return q->unpred[q->index++];
}

/* save, LinearQuantizer.hpp:95-104: [u8 uid=0b10][f64 eb][i32 radius][u64 n][T x n] */
static void SUF(quantizer_save)(const SUF(quantizer) * q, uint8_t **c) {
    wr_u8(c, 2);
    wr_f64(c, q->eb);
    wr_i32(c, q->radius);
    wr_u64(c, (uint64_t)q->n_unpred);
    if (q->n_unpred) wr_bytes(c, q->unpred, q->n_unpred * sizeof(T));
}
/* load, LinearQuantizer.hpp:106-122 */
static int SUF(quantizer_load)(SUF(quantizer) * q, const uint8_t **c) {
    uint8_t uid = rd_u8(c);
    if (uid != 2) return set_err("LinearQuantizer uid mismatch");
    q->eb = rd_f64(c);
    q->eb_recip = 1.0 / q->eb;
    q->radius = rd_i32(c);
    uint64_t n = rd_u64(c);
    free(q->unpred);
    q->unpred = NULL;
    q->n_unpred = q->cap_unpred = (size_t)n;
    if (n) {
        q->unpred = (T *)malloc(n * sizeof(T));
        rd_bytes(c, q->unpred, n * sizeof(T));
    }
    q->index = 0;
    return 0;
}

#include "sz3_oracle_interp.h"

/* ---------------------------------------------------------------------------------------------
 * block_data<T,N> (include/SZ3/utils/BlockwiseIterator.hpp:200-271): zero-initialised buffer of
 * (dims[i]+padding) per dim, data at offset `padding` in every dim; block walk :48-56, :104-141.
 * ------------------------------------------------------------------------------------------- */
typedef struct SUF(blockdata) {
    int N;
    size_t dims[4], ds[4], ds_pad[4];
    size_t padding, num, num_pad;
    T *buf;  /* internal_buffer */
    T *data; /* data_padding */
    size_t block_size, offset[4];
} SUF(blockdata);

static void SUF(bd_copy)(SUF(blockdata) * b, T *dst, const size_t *dstr, const T *src, const size_t *sstr) {
    /* copy_data_with_padding, BlockwiseIterator.hpp:240-271 (row-wise memcpy) */
    size_t d[4] = {1, 1, 1, 1}, ds_[4] = {0, 0, 0, 0}, ss_[4] = {0, 0, 0, 0};
    int sh = 4 - b->N;
    for (int i = 0; i < b->N; i++) {
        d[sh + i] = b->dims[i];
        ds_[sh + i] = dstr[i];
        ss_[sh + i] = sstr[i];
    }
    for (size_t i = 0; i < d[0]; i++)
        for (size_t j = 0; j < d[1]; j++)
            for (size_t k = 0; k < d[2]; k++)
                memcpy(dst + i * ds_[0] + j * ds_[1] + k * ds_[2], src + i * ss_[0] + j * ss_[1] + k * ss_[2],
                       d[3] * sizeof(T));
}

static void SUF(bd_init)(SUF(blockdata) * b, int N, const uint64_t *dims, size_t padding, const T *src,
                         size_t block_size) {
    memset(b, 0, sizeof(*b));
    b->N = N;
    b->padding = padding;
    b->block_size = block_size;
    size_t cs = 1, csp = 1; /* cal_dim_strides :226-238 */
    for (int i = N - 1; i >= 0; i--) {
        b->dims[i] = (size_t)dims[i];
        b->ds[i] = cs;
        b->ds_pad[i] = csp;
        cs *= b->dims[i];
        csp *= b->dims[i] + padding;
    }
    b->num = cs;
    b->num_pad = csp;
    b->buf = (T *)calloc(b->num_pad, sizeof(T));
    size_t off = 0;
    for (int i = 0; i < N; i++) off += b->ds_pad[i];
    b->data = b->buf + padding * off;
    if (src) SUF(bd_copy)(b, b->data, b->ds_pad, src, b->ds);
}
/* ~block_data(): copy the padded buffer back to the user array (decompression) :194-198 */
static void SUF(bd_copy_out)(SUF(blockdata) * b, T *dst) { SUF(bd_copy)(b, dst, b->ds, b->data, b->ds_pad); }
static void SUF(bd_free)(SUF(blockdata) * b) { free(b->buf); }

static int SUF(bd_next)(SUF(blockdata) * b) { /* block_iterator::next :48-56 */
    int i = b->N - 1;
    b->offset[i] += b->block_size;
    while (i && b->offset[i] >= b->dims[i]) {
        b->offset[i] = 0;
        b->offset[--i] += b->block_size;
    }
    return b->offset[0] < b->dims[0];
}
static void SUF(bd_range)(const SUF(blockdata) * b, size_t *len) { /* get_block_range :63-70, as lengths */
    for (int i = 0; i < b->N; i++) {
        size_t e = b->offset[i] + b->block_size;
        if (e > b->dims[i]) e = b->dims[i];
        len[i] = e - b->offset[i];
    }
}
static T *SUF(bd_ptr)(const SUF(blockdata) * b, const size_t *idx) { /* get_block_data :78-87 */
    size_t off = 0;
    for (int i = 0; i < b->N; i++) off += (idx[i] + b->offset[i]) * b->ds_pad[i];
    return b->data + off;
}

/* ---------------------------------------------------------------------------------------------
 * LorenzoPredictor<T,N,L>::predict (include/SZ3/predictor/LorenzoPredictor.hpp:60-95); arithmetic in T,
 * left-to-right exactly as written there.  ds = padded strides (ds[N-1] == 1).
 * ------------------------------------------------------------------------------------------- */
#define P1(i) (d[-(ptrdiff_t)(i)])
#define P2(j, i) (d[-(ptrdiff_t)((j)*ds[0] + (i))])
#define P3(k, j, i) (d[-(ptrdiff_t)((k)*ds[1] + (j)*ds[0] + (i))])
#define P4(t, k, j, i) (d[-(ptrdiff_t)((t)*ds[2] + (k)*ds[1] + (j)*ds[0] + (i))])
/* NOTE the reference indexes ds[] so that ds[0] is the stride of the *second-fastest* dim as seen by
 * prev2/prev3/prev4 (LorenzoPredictor.hpp:102-108): prev3(d,ds,k,j,i) = *(d - (k*ds[1] + j*ds[0] + i)).
 * With ds = block.get_dim_strides() = ds_padding (slowest first), ds[0] is the SLOWEST stride for N=3.
 * That is what the reference computes, and it is symmetric in the Lorenzo stencil, so we restate it verbatim. */
static inline T SUF(lorenzo_predict)(int N, int L, const T *d, const size_t *ds) {
    if (L == 1) {
        switch (N) {
            case 1: return P1(1);
            case 2: return (T)((T)(P2(0, 1) + P2(1, 0)) - P2(1, 1));
            case 3:
                return (T)((T)((T)((T)((T)((T)(P3(0, 0, 1) + P3(0, 1, 0)) + P3(1, 0, 0)) - P3(0, 1, 1)) - P3(1, 0, 1)) -
                               P3(1, 1, 0)) +
                           P3(1, 1, 1));
            default: {
                T s = P4(0, 0, 0, 1);
                s = (T)(s + P4(0, 0, 1, 0));
                s = (T)(s - P4(0, 0, 1, 1));
                s = (T)(s + P4(0, 1, 0, 0));
                s = (T)(s - P4(0, 1, 0, 1));
                s = (T)(s - P4(0, 1, 1, 0));
                s = (T)(s + P4(0, 1, 1, 1));
                s = (T)(s + P4(1, 0, 0, 0));
                s = (T)(s - P4(1, 0, 0, 1));
                s = (T)(s - P4(1, 0, 1, 0));
                s = (T)(s + P4(1, 0, 1, 1));
                s = (T)(s - P4(1, 1, 0, 0));
                s = (T)(s + P4(1, 1, 0, 1));
                s = (T)(s + P4(1, 1, 1, 0));
                s = (T)(s - P4(1, 1, 1, 1));
                return s;
            }
        }
    }
    /* L == 2, :75-91 ("2 * x" is int*T -> T) */
    switch (N) {
        case 1: return (T)((T)(2 * P1(1)) - P1(2));
        case 2: {
            T s = (T)(2 * P2(0, 1));
            s = (T)(s - P2(0, 2));
            s = (T)(s + (T)(2 * P2(1, 0)));
            s = (T)(s - (T)(4 * P2(1, 1)));
            s = (T)(s + (T)(2 * P2(1, 2)));
            s = (T)(s - P2(2, 0));
            s = (T)(s + (T)(2 * P2(2, 1)));
            s = (T)(s - P2(2, 2));
            return s;
        }
        default: {
            T s = (T)(2 * P3(0, 0, 1));
            s = (T)(s - P3(0, 0, 2));
            s = (T)(s + (T)(2 * P3(0, 1, 0)));
            s = (T)(s - (T)(4 * P3(0, 1, 1)));
            s = (T)(s + (T)(2 * P3(0, 1, 2)));
            s = (T)(s - P3(0, 2, 0));
            s = (T)(s + (T)(2 * P3(0, 2, 1)));
            s = (T)(s - P3(0, 2, 2));
            s = (T)(s + (T)(2 * P3(1, 0, 0)));
            s = (T)(s - (T)(4 * P3(1, 0, 1)));
            s = (T)(s + (T)(2 * P3(1, 0, 2)));
            s = (T)(s - (T)(4 * P3(1, 1, 0)));
            s = (T)(s + (T)(8 * P3(1, 1, 1)));
            s = (T)(s - (T)(4 * P3(1, 1, 2)));
            s = (T)(s + (T)(2 * P3(1, 2, 0)));
            s = (T)(s - (T)(4 * P3(1, 2, 1)));
            s = (T)(s + (T)(2 * P3(1, 2, 2)));
            s = (T)(s - P3(2, 0, 0));
            s = (T)(s + (T)(2 * P3(2, 0, 1)));
            s = (T)(s - P3(2, 0, 2));
            s = (T)(s + (T)(2 * P3(2, 1, 0)));
            s = (T)(s - (T)(4 * P3(2, 1, 1)));
            s = (T)(s + (T)(2 * P3(2, 1, 2)));
            s = (T)(s - P3(2, 2, 0));
            s = (T)(s + (T)(2 * P3(2, 2, 1)));
            s = (T)(s - P3(2, 2, 2));
            return s;
        }
    }
}
#undef P1
#undef P2
#undef P3
#undef P4

/* noise term of estimate_error, LorenzoPredictor.hpp:17-38 (stored in T) */
static T SUF(lorenzo_noise)(int N, int L, double eb) {
    static const double n1[5] = {0, 0.5, 0.81, 1.22, 1.79}, n2[5] = {0, 1.08, 2.76, 6.8, 0};
    return (T)((L == 1 ? n1[N] : n2[N]) * eb);
}

/* ---------------------------------------------------------------------------------------------
 * RegressionPredictor<T,N>  (include/SZ3/predictor/RegressionPredictor.hpp)
 * ------------------------------------------------------------------------------------------- */
typedef struct SUF(regression) {
    SUF(quantizer) q_indep, q_lin; /* :137 (eb/(N+1), eb/(N+1)/block_size; radius 32768) */
    int32_t *coeff_codes;          /* :138 regression_coeff_quant_inds */
    size_t n_codes, cap_codes, code_index;
    T prev[5], cur[5]; /* :140-141 */
    int N;
} SUF(regression);

static void SUF(reg_init)(SUF(regression) * r, int N, uint32_t block_size, double eb) { /* :22-26 */
    memset(r, 0, sizeof(*r));
    r->N = N;
    SUF(quantizer_init)(&r->q_indep, eb / (N + 1), 32768);
    SUF(quantizer_init)(&r->q_lin, eb / (N + 1) / block_size, 32768);
}
static void SUF(reg_free)(SUF(regression) * r) {
    SUF(quantizer_free)(&r->q_indep);
    SUF(quantizer_free)(&r->q_lin);
    free(r->coeff_codes);
}
/* precompress :28-55 — sums in double; index*value product is (float)(size_t)*T -> T */
static int SUF(reg_precompress)(SUF(regression) * r, const SUF(blockdata) * b) {
    int N = r->N;
    size_t len[4];
    SUF(bd_range)(b, len);
    double dims[4], num_elements = 1;
    for (int i = 0; i < N; i++) {
        dims[i] = (double)len[i];
        if (dims[i] <= 1) return 0;
        num_elements *= dims[i];
    }
    double sum[5] = {0, 0, 0, 0, 0};
    size_t L[4] = {1, 1, 1, 1};
    int sh = 4 - N;
    for (int i = 0; i < N; i++) L[sh + i] = len[i];
    size_t idx4[4];
    for (idx4[0] = 0; idx4[0] < L[0]; idx4[0]++)
        for (idx4[1] = 0; idx4[1] < L[1]; idx4[1]++)
            for (idx4[2] = 0; idx4[2] < L[2]; idx4[2]++) {
                size_t idx[4];
                for (int i = 0; i < N - 1; i++) idx[i] = idx4[sh + i];
                idx[N - 1] = 0;
                const T *c = SUF(bd_ptr)(b, idx);
                for (idx4[3] = 0; idx4[3] < L[3]; idx4[3]++, c++) {
                    for (int i = 0; i < N; i++) sum[i] += (T)((T)idx4[sh + i] * (*c));
                    sum[N] += *c;
                }
            }
    for (int i = 0; i <= N; i++) r->cur[i] = 0;
    r->cur[N] = (T)(sum[N] / num_elements);
    for (int i = 0; i < N; i++) {
        r->cur[i] = (T)((2 * sum[i] / (dims[i] - 1) - sum[N]) * 6 / num_elements / (dims[i] + 1));
        r->cur[N] = (T)((double)r->cur[N] - (dims[i] - 1) * (double)r->cur[i] / 2);
    }
    return 1;
}
static void SUF(reg_push_code)(SUF(regression) * r, int32_t c) {
    if (r->n_codes == r->cap_codes) {
        r->cap_codes = r->cap_codes ? r->cap_codes * 2 : 256;
        r->coeff_codes = (int32_t *)realloc(r->coeff_codes, r->cap_codes * sizeof(int32_t));
    }
    r->coeff_codes[r->n_codes++] = c;
}
/* precompress_block_commit :57-60 + pred_and_quantize_coefficients :148-155 */
static void SUF(reg_commit)(SUF(regression) * r) {
    for (int i = 0; i < r->N; i++)
        SUF(reg_push_code)(r, SUF(quantize_and_overwrite)(&r->q_lin, &r->cur[i], r->prev[i]));
    SUF(reg_push_code)(r, SUF(quantize_and_overwrite)(&r->q_indep, &r->cur[r->N], r->prev[r->N]));
    memcpy(r->prev, r->cur, sizeof(r->cur));
}
/* predecompress :62-71 + pred_and_recover_coefficients :157-164 */
static int SUF(reg_predecompress)(SUF(regression) * r, const SUF(blockdata) * b) {
    size_t len[4];
    SUF(bd_range)(b, len);
    for (int i = 0; i < r->N; i++)
        if (len[i] <= 1) return 0;
    for (int i = 0; i < r->N; i++)
        r->cur[i] = SUF(quantizer_recover)(&r->q_lin, r->cur[i], r->coeff_codes[r->code_index++]);
    r->cur[r->N] = SUF(quantizer_recover)(&r->q_indep, r->cur[r->N], r->coeff_codes[r->code_index++]);
    return 1;
}
/* predict :77-92 — T * (T)size_t products summed left to right in T */
static inline T SUF(reg_predict)(const SUF(regression) * r, const size_t *index) {
    T s = (T)(r->cur[0] * (T)index[0]);
    for (int i = 1; i < r->N; i++) s = (T)(s + (T)(r->cur[i] * (T)index[i]));
    return (T)(s + r->cur[r->N]);
}
/* save :94-107 */
static void SUF(reg_save)(SUF(regression) * r, uint8_t **c) {
    wr_u64(c, (uint64_t)r->n_codes);
    if (r->n_codes) {
        SUF(quantizer_save)(&r->q_indep, c);
        SUF(quantizer_save)(&r->q_lin, c);
        huffman_encode_all(r->coeff_codes, r->n_codes, c);
    }
}
/* load :109-123 */
static int SUF(reg_load)(SUF(regression) * r, const uint8_t **c) {
    uint64_t n = rd_u64(c);
    if (n) {
        if (SUF(quantizer_load)(&r->q_indep, c)) return -1;
        if (SUF(quantizer_load)(&r->q_lin, c)) return -1;
        free(r->coeff_codes);
        r->coeff_codes = (int32_t *)malloc(n * sizeof(int32_t));
        r->n_codes = r->cap_codes = (size_t)n;
        huffman_decode_all(c, (size_t)n, r->coeff_codes);
        for (int i = 0; i <= r->N; i++) r->cur[i] = 0;
        r->code_index = 0;
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Predictor set = what make_compressor_lorenzo_regression assembles (api/impl/SZAlgoLorenzoReg.hpp:22-64):
 * one predictor directly, or a ComposedPredictor (predictor/ComposedPredictor.hpp) over
 * [lorenzo, lorenzo2, regression] in that order.
 * ------------------------------------------------------------------------------------------- */
#ifndef SZO_PK_ENUM
#define SZO_PK_ENUM
enum { PK_LORENZO1 = 0, PK_LORENZO2 = 1, PK_REGRESSION = 2 };
#endif
typedef struct SUF(predset) {
    int N, n_pred, kinds[3], composed;
    T noise[3];
    SUF(regression) reg;
    int has_reg;
    int32_t *selection; /* ComposedPredictor::selection :112 */
    size_t n_sel, cap_sel, sel_index;
    int sid;
    T fallback_noise;
    size_t n_reg_blocks, n_blocks;
} SUF(predset);

static int SUF(ps_init)(SUF(predset) * p, const szo_config *conf) {
    memset(p, 0, sizeof(*p));
    p->N = conf->N;
    int cnt = (conf->lorenzo != 0) + (conf->lorenzo2 != 0) + (conf->regression != 0);
    if (cnt == 0) return set_err("All lorenzo and regression methods are disabled.");
    p->composed = cnt > 1;
    if (conf->lorenzo) {
        p->noise[p->n_pred] = SUF(lorenzo_noise)(p->N, 1, conf->absErrorBound);
        p->kinds[p->n_pred++] = PK_LORENZO1;
    }
    if (conf->lorenzo2) {
        if (p->N == 4) return set_err("2nd-order Lorenzo is not defined for N=4 (LorenzoPredictor.hpp:92)");
        p->noise[p->n_pred] = SUF(lorenzo_noise)(p->N, 2, conf->absErrorBound);
        p->kinds[p->n_pred++] = PK_LORENZO2;
    }
    if (conf->regression) {
        SUF(reg_init)(&p->reg, p->N, (uint32_t)conf->blockSize, conf->absErrorBound);
        p->has_reg = 1;
        p->kinds[p->n_pred++] = PK_REGRESSION;
    }
    return 0;
}
static void SUF(ps_free)(SUF(predset) * p) {
    if (p->has_reg) SUF(reg_free)(&p->reg);
    free(p->selection);
}
static size_t SUF(ps_padding)(const SUF(predset) * p) {
    /* get_padding: Lorenzo 2 (LorenzoPredictor.hpp:54); regression inherits the interface default 0
     * (predictor/Predictor.hpp); Composed = max (ComposedPredictor.hpp:101-107) */
    size_t m = 0;
    for (int i = 0; i < p->n_pred; i++)
        if (p->kinds[i] != PK_REGRESSION) m = 2;
    return m;
}
static inline T SUF(ps_predict_kind)(const SUF(predset) * p, int kind, const SUF(blockdata) * b, const T *d,
                                     const size_t *index) {
    if (kind == PK_REGRESSION) return SUF(reg_predict)(&p->reg, index);
    return SUF(lorenzo_predict)(p->N, kind == PK_LORENZO1 ? 1 : 2, d, b->ds_pad);
}

/* foreach_sampling (BlockwiseIterator.hpp:151-184): the diagonal sample points of a block */
static size_t SUF(sample_points)(int N, const size_t *len, size_t (*pts)[4], size_t cap) {
    size_t m = (size_t)-1, n = 0;
    for (int i = 0; i < N; i++)
        if (len[i] < m) m = len[i];
    if (N == 1) {
        pts[n++][0] = 0;
        pts[n++][0] = m - 1;
        return n;
    }
    for (size_t i = 0; i < m; i++) {
        size_t j = m - 1 - i;
        int combos = 1 << (N - 1);
        for (int cmb = 0; cmb < combos; cmb++) {
            if (n >= cap) return n;
            pts[n][0] = i;
            /* order {i..i},{i..j},... = binary counting with the LAST index toggling fastest */
            for (int k = 1; k < N; k++) pts[n][k] = ((cmb >> (N - 1 - k)) & 1) ? j : i;
            n++;
        }
    }
    return n;
}

/* ComposedPredictor::precompress (:25-40) or the single predictor's precompress.
 * returns 1 if the selected predictor is valid for the block (else the caller falls back to Lorenzo-1,
 * BlockwiseDecomposition.hpp:35-37) */
static int SUF(ps_precompress)(SUF(predset) * p, const SUF(blockdata) * b) {
    if (!p->composed) {
        p->sid = 0;
        if (p->kinds[0] == PK_REGRESSION) return SUF(reg_precompress)(&p->reg, b);
        return 1;
    }
    double err[3];
    int valid[3];
    size_t len[4];
    SUF(bd_range)(b, len);
    static size_t pts[8 * 4096][4];
    size_t npts = 0;
    int have_pts = 0;
    for (int i = 0; i < p->n_pred; i++) {
        err[i] = 0;
        valid[i] = p->kinds[i] == PK_REGRESSION ? SUF(reg_precompress)(&p->reg, b) : 1;
        if (valid[i]) {
            if (!have_pts) {
                npts = SUF(sample_points)(p->N, len, pts, 8 * 4096);
                have_pts = 1;
            }
            for (size_t s = 0; s < npts; s++) {
                const T *c = SUF(bd_ptr)(b, pts[s]);
                T pr = SUF(ps_predict_kind)(p, p->kinds[i], b, c, pts[s]);
                T e = (T)fabs((double)(T)(*c - pr)); /* estimate_error: LorenzoPredictor.hpp:56-58, Regression :73-75 */
                if (p->kinds[i] != PK_REGRESSION) e = (T)(e + p->noise[i]);
                err[i] += e;
            }
        } else {
            err[i] = DBL_MAX;
        }
    }
    int best = 0; /* std::min_element: first minimum */
    for (int i = 1; i < p->n_pred; i++)
        if (err[i] < err[best]) best = i;
    p->sid = best;
    return valid[best];
}
/* precompress_block_commit: Composed :42-45 (selection.push_back + winner commits) */
static void SUF(ps_commit)(SUF(predset) * p) {
    if (p->composed) {
        if (p->n_sel == p->cap_sel) {
            p->cap_sel = p->cap_sel ? p->cap_sel * 2 : 1024;
            p->selection = (int32_t *)realloc(p->selection, p->cap_sel * sizeof(int32_t));
        }
        p->selection[p->n_sel++] = p->sid;
    }
    if (p->kinds[p->sid] == PK_REGRESSION) {
        SUF(reg_commit)(&p->reg);
        p->n_reg_blocks++;
    }
}
static int SUF(ps_predecompress)(SUF(predset) * p, const SUF(blockdata) * b) { /* Composed :47-50 */
    if (p->composed) p->sid = p->selection[p->sel_index++];
    if (p->kinds[p->sid] == PK_REGRESSION) return SUF(reg_predecompress)(&p->reg, b);
    return 1;
}
/* save: Composed :52-64 (each predictor's save, then [u64 nSel]{Huffman}); Lorenzo::save writes nothing */
static void SUF(ps_save)(SUF(predset) * p, uint8_t **c) {
    if (p->has_reg) SUF(reg_save)(&p->reg, c);
    if (p->composed) {
        wr_u64(c, (uint64_t)p->n_sel);
        if (p->n_sel) huffman_encode_all(p->selection, p->n_sel, c);
    }
}
static int SUF(ps_load)(SUF(predset) * p, const uint8_t **c) { /* Composed :66-78 */
    if (p->has_reg && SUF(reg_load)(&p->reg, c)) return -1;
    if (p->composed) {
        uint64_t n = rd_u64(c);
        free(p->selection);
        p->selection = NULL;
        p->n_sel = p->cap_sel = (size_t)n;
        p->sel_index = 0;
        if (n) {
            p->selection = (int32_t *)malloc(n * sizeof(int32_t));
            huffman_decode_all(c, (size_t)n, p->selection);
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * BlockwiseDecomposition<T,N,Predictor,Quantizer>  (include/SZ3/decomposition/BlockwiseDecomposition.hpp)
 * ------------------------------------------------------------------------------------------- */
/* compress :28-46 — data is consumed (copied into the padded buffer); emits conf.num codes block-major */
static int SUF(blockwise_compress)(const szo_config *conf, SUF(predset) * ps, SUF(quantizer) * q, const T *data,
                                   int32_t *codes) {
    SUF(blockdata) b;
    SUF(bd_init)(&b, conf->N, conf->dims, SUF(ps_padding)(ps), data, (size_t)conf->blockSize);
    int N = conf->N, sh = 4 - N;
    size_t pos = 0;
    do {
        int ok = SUF(ps_precompress)(ps, &b);
        int kind = ok ? ps->kinds[ps->sid] : PK_LORENZO1; /* fallback_predictor :35-37 */
        if (ok) SUF(ps_commit)(ps);
        if (g_sel_sink && ps->n_blocks < g_sel_cap) g_sel_sink[ps->n_blocks] = (int8_t)kind;
        /* NOTE: when precompress fails the reference calls fallback_predictor.precompress_block_commit()
         * (a no-op, LorenzoPredictor.hpp:44) and the composed predictor's selection is NOT extended. */
        ps->n_blocks++;
        size_t len[4], L[4] = {1, 1, 1, 1}, idx4[4];
        SUF(bd_range)(&b, len);
        for (int i = 0; i < N; i++) L[sh + i] = len[i];
        for (idx4[0] = 0; idx4[0] < L[0]; idx4[0]++)
            for (idx4[1] = 0; idx4[1] < L[1]; idx4[1]++)
                for (idx4[2] = 0; idx4[2] < L[2]; idx4[2]++) {
                    size_t idx[4];
                    for (int i = 0; i < N - 1; i++) idx[i] = idx4[sh + i];
                    idx[N - 1] = 0;
                    T *c = SUF(bd_ptr)(&b, idx);
                    for (idx4[3] = 0; idx4[3] < L[3]; idx4[3]++, c++) {
                        idx[N - 1] = idx4[3];
                        T pred = SUF(ps_predict_kind)(ps, kind, &b, c, idx);
                        codes[pos++] = SUF(quantize_and_overwrite)(q, c, pred);
                    }
                }
    } while (SUF(bd_next)(&b));
    SUF(bd_free)(&b);
    return 0;
}
/* decompress :48-67 */
static int SUF(blockwise_decompress)(const szo_config *conf, SUF(predset) * ps, SUF(quantizer) * q,
                                     const int32_t *codes, T *dec) {
    SUF(blockdata) b;
    SUF(bd_init)(&b, conf->N, conf->dims, SUF(ps_padding)(ps), NULL, (size_t)conf->blockSize);
    int N = conf->N, sh = 4 - N;
    size_t pos = 0;
    do {
        int ok = SUF(ps_predecompress)(ps, &b);
        int kind = ok ? ps->kinds[ps->sid] : PK_LORENZO1;
        size_t len[4], L[4] = {1, 1, 1, 1}, idx4[4];
        SUF(bd_range)(&b, len);
        for (int i = 0; i < N; i++) L[sh + i] = len[i];
        for (idx4[0] = 0; idx4[0] < L[0]; idx4[0]++)
            for (idx4[1] = 0; idx4[1] < L[1]; idx4[1]++)
                for (idx4[2] = 0; idx4[2] < L[2]; idx4[2]++) {
                    size_t idx[4];
                    for (int i = 0; i < N - 1; i++) idx[i] = idx4[sh + i];
                    idx[N - 1] = 0;
                    T *c = SUF(bd_ptr)(&b, idx);
                    for (idx4[3] = 0; idx4[3] < L[3]; idx4[3]++, c++) {
                        idx[N - 1] = idx4[3];
                        T pred = SUF(ps_predict_kind)(ps, kind, &b, c, idx);
                        *c = SUF(quantizer_recover)(q, pred, codes[pos++]);
                    }
                }
    } while (SUF(bd_next)(&b));
    if (SUF(ps_padding)(ps) > 0) SUF(bd_copy_out)(&b, dec);
    else memcpy(dec, b.data, b.num * sizeof(T)); /* padding==0: block_data aliases the user array :218 */
    SUF(bd_free)(&b);
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * SZ_compress_LorenzoReg / SZGenericCompressor::compress
 * (api/impl/SZAlgoLorenzoReg.hpp:67-84, compressor/SZGenericCompressor.hpp:38-63)
 * raw := [decomposition.save][encoder.save][u64 n][u64 encBytes][bits]  ->  Lossless_zstd
 * ------------------------------------------------------------------------------------------- */
static size_t SUF(compress_lorenzo_reg)(const szo_config *conf, const T *data, uint8_t *out, size_t cap,
                                        szo_stats *st, int32_t *codes_out) {
    SUF(predset) ps;
    SUF(quantizer) q;
    if (SUF(ps_init)(&ps, conf)) return 0;
    SUF(quantizer_init)(&q, conf->absErrorBound, conf->quantbinCnt / 2); /* :72 */
    size_t n = (size_t)conf->num;
    int32_t *codes = codes_out ? codes_out : (int32_t *)malloc(n * sizeof(int32_t));
    double t0 = now_s();
    SUF(blockwise_compress)(conf, &ps, &q, data, codes);
    double t1 = now_s();
    size_t result = 0;
    if (st) {
        st->n_unpred = q.n_unpred;
        st->n_regression_blocks = ps.n_reg_blocks;
        st->n_blocks = ps.n_blocks;
        st->t_decomp = t1 - t0;
    }
    if (out) {
        /* bufferSize = max(1000, 2*(size_est + sizeof(T)*n)) — SZGenericCompressor.hpp:45-48; we size generously */
        size_t bufsz = 4096 + 2 * (sizeof(T) * n + q.n_unpred * sizeof(T)) + 16 * n / 1 + (ps.has_reg ? 64 * ps.reg.n_codes : 0);
        uint8_t *buf = (uint8_t *)malloc(bufsz), *p = buf;
        SUF(ps_save)(&ps, &p);          /* fallback Lorenzo saves nothing, then predictor.save */
        SUF(quantizer_save)(&q, &p);    /* BlockwiseDecomposition::save :69-73 */
        huff_times ht = {0, 0};
        uint32_t node_count = 0;
        uint64_t enc_bytes = 0;
        uint8_t *p_before = p;
        (void)p_before;
        huffman_encode_main(codes, n, &p, &ht, &node_count, &enc_bytes); /* encoder.save, write(n), encode */
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
    SUF(quantizer_free)(&q);
    SUF(ps_free)(&ps);
    return result;
}

/* SZ_decompress_LorenzoReg / SZGenericCompressor::decompress (SZAlgoLorenzoReg.hpp:76-84, SZGenericCompressor.hpp:65-84) */
static int SUF(decompress_lorenzo_reg)(const szo_config *conf, const uint8_t *cmp, size_t cmp_size, T *dec) {
    uint64_t raw_len;
    memcpy(&raw_len, cmp, 8);
    uint8_t *raw = (uint8_t *)malloc(raw_len ? raw_len : 1);
    if (szo_zstd_decompress(cmp, cmp_size, raw, raw_len) != raw_len) {
        free(raw);
        return set_err("zstd decompress failed");
    }
    SUF(predset) ps;
    SUF(quantizer) q;
    if (SUF(ps_init)(&ps, conf)) {
        free(raw);
        return -1;
    }
    SUF(quantizer_init)(&q, 1, 32768); /* default-constructed quantizer :80, then load */
    const uint8_t *p = raw;
    int rc = SUF(ps_load)(&ps, &p);
    if (!rc) rc = SUF(quantizer_load)(&q, &p);
    if (!rc) {
        size_t n = (size_t)conf->num;
        int32_t *codes = (int32_t *)malloc(n * sizeof(int32_t));
        huffman_decode_main(&p, codes, n);
        rc = SUF(blockwise_decompress)(conf, &ps, &q, codes, dec);
        free(codes);
    }
    SUF(quantizer_free)(&q);
    SUF(ps_free)(&ps);
    free(raw);
    return rc;
}

/* data_range, utils/Statistic.hpp:12-21 */
static T SUF(data_range)(const T *data, size_t num) {
    T max = data[0], min = data[0];
    for (size_t i = 1; i < num; i++) {
        if (max < data[i]) max = data[i];
        if (min > data[i]) min = data[i];
    }
    return (T)(max - min);
}
/* calAbsErrorBound, utils/Statistic.hpp:32-56 (range argument 0 => computed) */
static int SUF(cal_abs_eb)(szo_config *conf, const T *data) {
    if (conf->errorBoundMode == SZO_EB_ABS) return 0;
    switch (conf->errorBoundMode) {
        case SZO_EB_REL:
            conf->absErrorBound = conf->relErrorBound * (double)SUF(data_range)(data, (size_t)conf->num);
            break;
        case SZO_EB_PSNR: { /* computeABSErrBoundFromPSNR :25-30 with threshold 0.99 */
            double range = (double)SUF(data_range)(data, (size_t)conf->num);
            double v1 = conf->psnrErrorBound + 10 * log10(1 - 2.0 / 3.0 * 0.99);
            conf->absErrorBound = range * pow(10, v1 / (-20));
            break;
        }
        case SZO_EB_L2NORM: conf->absErrorBound = sqrt(3.0 / (double)conf->num) * conf->l2normErrorBound; break;
        case SZO_EB_ABS_AND_REL: {
            double r = conf->relErrorBound * (double)SUF(data_range)(data, (size_t)conf->num);
            conf->absErrorBound = conf->absErrorBound < r ? conf->absErrorBound : r;
            break;
        }
        case SZO_EB_ABS_OR_REL: {
            double r = conf->relErrorBound * (double)SUF(data_range)(data, (size_t)conf->num);
            conf->absErrorBound = conf->absErrorBound > r ? conf->absErrorBound : r;
            break;
        }
        default: return set_err("Error bound mode not supported");
    }
    conf->errorBoundMode = SZO_EB_ABS;
    return 0;
}

#include "sz3_oracle_tuner.h"

/* SZ_compress_dispatcher, api/impl/SZDispatcher.hpp:13-76 (serial path; conf is modified like the reference's copy) */
static size_t SUF(compress_dispatch)(szo_config *conf, const T *data, uint8_t *out, size_t cap, szo_stats *st) {
    if (SUF(cal_abs_eb)(conf, data)) return 0;
    size_t cmp = 0;
    size_t raw_size = (size_t)conf->num * sizeof(T);
    if (conf->absErrorBound == 0) conf->cmprAlgo = SZO_ALGO_LOSSLESS; /* :19-21 */
    int cap_ok = 1;
    if (conf->cmprAlgo != SZO_ALGO_LOSSLESS) {
        if (cap < 8 + szo_zstd_bound(0)) cap_ok = 0;
        if (conf->cmprAlgo == SZO_ALGO_LORENZO_REG) {
            cmp = SUF(compress_lorenzo_reg)(conf, data, out, cap, st, NULL);
        } else if (conf->cmprAlgo == SZO_ALGO_INTERP) {
            cmp = SUF(compress_interp)(conf, data, out, cap, st, NULL);
        } else if (conf->cmprAlgo == SZO_ALGO_INTERP_LORENZO) {
            cmp = SUF(compress_interp_lorenzo)(conf, data, out, cap, st);
        } else {
            set_err("oracle: compression algorithm outside the hot-path scope");
            return 0;
        }
        if (cmp == 0) {
            if (zstd_cap_error) cap_ok = 0; /* std::length_error => lossless fallback :44-59 */
            else return 0;
        }
    }
    if (conf->cmprAlgo == SZO_ALGO_LOSSLESS || !cap_ok) {
        conf->cmprAlgo = SZO_ALGO_LOSSLESS;
        return szo_zstd_compress((const uint8_t *)data, raw_size, out, cap);
    }
    if ((double)raw_size / 1.0 / (double)cmp < 3) { /* :62-74 */
        size_t zcap = szo_zstd_bound(raw_size) + 8;
        uint8_t *z = (uint8_t *)malloc(zcap);
        size_t zs = szo_zstd_compress((const uint8_t *)data, raw_size, z, zcap);
        if (zs && zs < cmp && zs <= cap) {
            conf->cmprAlgo = SZO_ALGO_LOSSLESS;
            memcpy(out, z, zs);
            cmp = zs;
        }
        free(z);
    }
    return cmp;
}

/* SZ_decompress_dispatcher, api/impl/SZDispatcher.hpp:79-100 */
static int SUF(decompress_dispatch)(const szo_config *conf, const uint8_t *cmp, size_t cmp_size, T *dec) {
    if (conf->cmprAlgo == SZO_ALGO_LOSSLESS) {
        uint64_t n;
        memcpy(&n, cmp, 8);
        if (n != conf->num * sizeof(T)) return set_err("Decompressed data size does not match the original data size");
        if (szo_zstd_decompress(cmp, cmp_size, (uint8_t *)dec, (size_t)n) != n) return set_err("zstd decompress failed");
        return 0;
    }
    if (conf->cmprAlgo == SZO_ALGO_LORENZO_REG) return SUF(decompress_lorenzo_reg)(conf, cmp, cmp_size, dec);
    if (conf->cmprAlgo == SZO_ALGO_INTERP) return SUF(decompress_interp)(conf, cmp, cmp_size, dec);
    return set_err("Unknown compression algorithm");
}

/* SZ_compress_OMP container (api/impl/SZImplOMP.hpp:16-117) evaluated serially slab by slab with a fixed
 * slab count `nslabs` (the reference uses omp_get_num_threads(); the layout depends only on that count):
 *   [i32 n][Config x n][u64 size x n][blob x n]   */
static size_t SUF(compress_omp)(szo_config *conf, const T *data, uint8_t *out, size_t cap, int nslabs, szo_stats *st) {
    if ((uint64_t)nslabs > conf->dims[0]) nslabs = (int)conf->dims[0]; /* :33-36 */
    size_t base = 1;
    for (int i = 1; i < conf->N; i++) base *= (size_t)conf->dims[i];
    if (conf->errorBoundMode != SZO_EB_ABS) { /* :57-69: global range from per-slab min/max == global range */
        if (SUF(cal_abs_eb)(conf, data)) return 0;
    }
    szo_config *ct = (szo_config *)calloc((size_t)nslabs, sizeof(szo_config));
    uint8_t **blob = (uint8_t **)calloc((size_t)nslabs, sizeof(uint8_t *));
    uint64_t *sz = (uint64_t *)calloc((size_t)nslabs, sizeof(uint64_t));
    size_t total = 0;
    int fail = 0;
    for (int t = 0; t < nslabs && !fail; t++) {
        int lo = (int)((uint64_t)t * conf->dims[0] / (uint64_t)nslabs);
        int hi = (int)((uint64_t)(t + 1) * conf->dims[0] / (uint64_t)nslabs);
        uint64_t d[4];
        for (int i = 0; i < conf->N; i++) d[i] = conf->dims[i];
        d[0] = (uint64_t)(hi - lo);
        ct[t] = *conf;
        /* conf_t[tid].setDims(...) resets N/num/predDim/blockSize (:71-72, Config.hpp:161-177) */
        szo_config tmp;
        szo_config_init(&tmp, conf->N, d);
        ct[t].N = tmp.N;
        memcpy(ct[t].dims, tmp.dims, sizeof(tmp.dims));
        ct[t].num = tmp.num;
        ct[t].predDim = tmp.predDim;
        ct[t].blockSize = tmp.blockSize;
        size_t ccap = szo_zstd_bound((size_t)ct[t].num * sizeof(T)); /* :73 */
        blob[t] = (uint8_t *)malloc(ccap + 64);
        sz[t] = SUF(compress_dispatch)(&ct[t], data + (size_t)lo * base, blob[t], ccap, t == 0 ? st : NULL);
        if (!sz[t]) fail = 1;
        total += (size_t)sz[t];
    }
    size_t written = 0;
    if (!fail) {
        uint8_t *p = out;
        wr_i32(&p, nslabs);
        for (int t = 0; t < nslabs; t++) p += szo_config_save(&ct[t], p);
        for (int t = 0; t < nslabs; t++) wr_u64(&p, sz[t]);
        if ((size_t)(p - out) + total > cap) fail = 1;
        else {
            for (int t = 0; t < nslabs; t++) {
                memcpy(p, blob[t], (size_t)sz[t]);
                p += sz[t];
            }
            written = (size_t)(p - out);
        }
    }
    for (int t = 0; t < nslabs; t++) free(blob[t]);
    free(blob);
    free(sz);
    free(ct);
    return fail ? 0 : written;
}
/* SZ_decompress_OMP (api/impl/SZImplOMP.hpp:120-186) */
static int SUF(decompress_omp)(const szo_config *conf, const uint8_t *cmp, size_t cmp_size, T *dec) {
    const uint8_t *p = cmp;
    int32_t n = rd_i32(&p);
    if (n <= 0 || n > 65536) return set_err("bad slab count");
    szo_config *ct = (szo_config *)calloc((size_t)n, sizeof(szo_config));
    for (int t = 0; t < n; t++) p += szo_config_load(&ct[t], p);
    uint64_t *sz = (uint64_t *)malloc((size_t)n * 8);
    for (int t = 0; t < n; t++) sz[t] = rd_u64(&p);
    size_t off = 0;
    int rc = 0;
    for (int t = 0; t < n && !rc; t++) {
        rc = SUF(decompress_dispatch)(&ct[t], p, (size_t)sz[t], dec + off);
        p += sz[t];
        off += (size_t)ct[t].num;
    }
    (void)cmp_size;
    (void)conf;
    free(sz);
    free(ct);
    return rc;
}
