/* sz3_oracle_tuner.h — TEST INFRASTRUCTURE ONLY. Restatement of the ALGO_INTERP_LORENZO sampling auto-tuner:
 *   SZ_compress_Interp_lorenzo   api/impl/SZAlgoInterp.hpp:122-286
 *   interp_compress_test         api/impl/SZAlgoInterp.hpp:42-78
 *   lorenzo_compress_test        api/impl/SZAlgoInterp.hpp:80-120
 *   profiling_block              utils/Sample.hpp:9-136
 *   sample_blocks / sampleBlocks utils/Sample.hpp:138-289
 * Included by sz3_oracle_impl.h once per element type (T, SUF). */

/* profiling_block: origins (multiples of bs, strictly below dim - bs) of blocks whose strided samples span more
 * than abseb; lexicographic order. Returns the number found; *starts_out is malloc'ed [count][N]. */
static size_t SUF(tn_profile)(const T *data, int N, const uint64_t *dims, size_t bs, double abseb, size_t stride,
                              size_t **starts_out) {
    *starts_out = NULL;
    if (stride == 0) stride = bs;
    size_t off[4], cnt[4], total = 1;
    for (int i = 0; i < N; i++)
        if (dims[i] < bs) return 0;
    off[N - 1] = 1;
    for (int i = N - 2; i >= 0; i--) off[i] = off[i + 1] * (size_t)dims[i + 1];
    for (int i = 0; i < N; i++) {
        size_t lim = (size_t)dims[i] - bs; /* for (i = 0; i < dim - bs; i += bs) */
        cnt[i] = (lim + bs - 1) / bs;
        total *= cnt[i];
    }
    if (total == 0) return 0;
    size_t *starts = (size_t *)malloc(sizeof(size_t) * (size_t)N * total), found = 0;
    size_t b[4] = {0, 0, 0, 0};
    for (size_t t = 0; t < total; t++) {
        size_t start_idx = 0;
        for (int i = 0; i < N; i++) start_idx += b[i] * bs * off[i];
        T mn = data[start_idx], mx = data[start_idx];
        size_t k[4] = {0, 0, 0, 0};
        for (;;) { /* ii, jj, ... from 0 to bs inclusive, step stride */
            size_t idx = start_idx;
            for (int i = 0; i < N; i++) idx += k[i] * off[i];
            T v = data[idx];
            if (v < mn) mn = v;
            else if (v > mx) mx = v;
            int i = N - 1;
            for (; i >= 0; i--) {
                k[i] += stride;
                if (k[i] <= bs) break;
                k[i] = 0;
            }
            if (i < 0) break;
        }
        if (mx - mn > abseb) {
            for (int i = 0; i < N; i++) starts[found * (size_t)N + i] = b[i] * bs;
            found++;
        }
        for (int i = N - 1; i >= 0; i--) {
            if (++b[i] < cnt[i]) break;
            b[i] = 0;
        }
    }
    *starts_out = starts;
    return found;
}

/* sample_blocks: copy the (edge)^N block at `start` */
static T *SUF(tn_copy_block)(const T *data, int N, const uint64_t *dims, const size_t *start, size_t edge) {
    size_t off[4], per = 1;
    off[N - 1] = 1;
    for (int i = N - 2; i >= 0; i--) off[i] = off[i + 1] * (size_t)dims[i + 1];
    for (int i = 0; i < N; i++) per *= edge;
    T *blk = (T *)malloc(per * sizeof(T));
    size_t k[4] = {0, 0, 0, 0};
    for (size_t s = 0; s < per; s++) {
        size_t idx = 0;
        for (int i = 0; i < N; i++) idx += (start[i] + k[i]) * off[i];
        blk[s] = data[idx];
        for (int i = N - 1; i >= 0; i--) {
            if (++k[i] < edge) break;
            k[i] = 0;
        }
    }
    return blk;
}

/* sampleBlocks: returns the number of blocks; *blocks_out malloc'ed array of malloc'ed blocks of (sbs+1)^N */
static size_t SUF(tn_sample)(const T *data, int N, const uint64_t *dims, size_t sbs, double rate, int profiling,
                             const size_t *starts, size_t n_starts, T ***blocks_out) {
    *blocks_out = NULL;
    for (int i = 0; i < N; i++)
        if (dims[i] < sbs) return 0;
    size_t totalblock = 1;
    for (int i = 0; i < N; i++) totalblock *= (size_t)(int)(((size_t)dims[i] - 1) / sbs);
    size_t cap = 16, nb = 0;
    T **blocks = (T **)malloc(cap * sizeof(T *));
    if (profiling) {
        size_t stride = (size_t)((double)n_starts / ((double)totalblock * rate));
        if (stride <= 0) stride = 1;
        for (size_t i = 0; i < n_starts; i += stride) {
            if (nb == cap) blocks = (T **)realloc(blocks, (cap *= 2) * sizeof(T *));
            blocks[nb++] = SUF(tn_copy_block)(data, N, dims, starts + i * (size_t)N, sbs + 1);
        }
    } else {
        size_t stride = (size_t)(1.0 / rate);
        if (stride <= 0) stride = 1;
        size_t cnt[4], total = 1, b[4] = {0, 0, 0, 0};
        for (int i = 0; i < N; i++) {
            cnt[i] = ((size_t)dims[i] - sbs + sbs - 1) / sbs; /* x_start < dims - sbs, step sbs */
            total *= cnt[i];
        }
        for (size_t idx = 0; idx < total; idx++) {
            if (idx % stride == 0) {
                size_t st[4];
                for (int i = 0; i < N; i++) st[i] = b[i] * sbs;
                if (nb == cap) blocks = (T **)realloc(blocks, (cap *= 2) * sizeof(T *));
                blocks[nb++] = SUF(tn_copy_block)(data, N, dims, st, sbs + 1);
            }
            for (int i = N - 1; i >= 0; i--) {
                if (++b[i] < cnt[i]) break;
                b[i] = 0;
            }
        }
    }
    *blocks_out = blocks;
    return nb;
}

/* interp_compress_test: one decomposition object (its quantizer keeps the unpredictables of all blocks), codes of all
 * blocks concatenated, one Huffman tree, zstd; ratio = raw bytes of the samples / compressed bytes */
static double SUF(tn_interp_test)(const szo_config *tc, T **blocks, size_t nb, size_t per, uint64_t *raw_bytes) {
    /* raw_bytes (optional): [0] pre-zstd size, [8] Huffman bit-stream bytes, [16] node count, [24] unpredictables,
     * [32] (as double) Shannon entropy of the codes in bits */
    SUF(interp) c;
    memset(&c, 0, sizeof(c));
    c.N = tc->N;
    SUF(quantizer_init)(&c.q, tc->absErrorBound, tc->quantbinCnt / 2);
    int32_t *codes = (int32_t *)malloc(sizeof(int32_t) * per * nb);
    T *work = (T *)malloc(sizeof(T) * per);
    for (size_t k = 0; k < nb; k++) {
        for (int i = 0; i < c.N; i++) c.dims[i] = (size_t)tc->dims[i];
        c.interp_id = tc->interpAlgo;
        c.direction = tc->interpDirection;
        c.anchor_stride = (size_t)tc->interpAnchorStride;
        c.blocksize = 32;
        c.alpha = tc->interpAlpha;
        c.beta = tc->interpBeta;
        memcpy(work, blocks[k], sizeof(T) * per);
        c.codes = codes + k * per;
        SUF(ip_run)(&c, work);
    }
    size_t n = per * nb;
    size_t bufsz = 4096 + 2 * (sizeof(T) * n + c.q.n_unpred * sizeof(T)) + 16 * n;
    uint8_t *buf = (uint8_t *)malloc(bufsz), *p = buf;
    SUF(ip_save)(&c, &p);
    uint32_t node_count = 0;
    uint64_t enc_bytes = 0;
    huffman_encode_main(codes, n, &p, NULL, &node_count, &enc_bytes);
    if (raw_bytes) {
        raw_bytes[0] = (uint64_t)(p - buf);
        raw_bytes[8] = enc_bytes;
        raw_bytes[16] = node_count;
        raw_bytes[24] = c.q.n_unpred;
        uint64_t *hh = (uint64_t *)calloc(65536 * 2, sizeof(uint64_t));
        for (size_t i = 0; i < n; i++) hh[(uint32_t)codes[i] & 0x1FFFF]++;
        double H = 0;
        for (size_t i = 0; i < 65536 * 2; i++)
            if (hh[i]) H += (double)hh[i] * log2((double)n / (double)hh[i]);
        free(hh);
        memcpy(&raw_bytes[32], &H, 8);
    }
    size_t zcap = szo_zstd_bound((size_t)(p - buf)) + 8;
    uint8_t *z = (uint8_t *)malloc(zcap);
    size_t cs = szo_zstd_compress(buf, (size_t)(p - buf), z, zcap);
    free(z);
    free(buf);
    free(work);
    free(codes);
    SUF(quantizer_free)(&c.q);
    return (double)tc->num * (double)nb * sizeof(T) * 1.0 / (double)cs;
}

/* lorenzo_compress_test: blockwise decomposition with ComposedPredictor{Lorenzo 1st, 2nd order} */
static double SUF(tn_lorenzo_test)(const szo_config *lc, T **blocks, size_t nb, size_t per) {
    SUF(predset) ps;
    SUF(quantizer) q;
    szo_config c2 = *lc;
    c2.lorenzo = 1;
    c2.lorenzo2 = 1;
    c2.regression = 0;
    c2.regression2 = 0;
    if (SUF(ps_init)(&ps, &c2)) return 0;
    SUF(quantizer_init)(&q, lc->absErrorBound, lc->quantbinCnt / 2);
    int32_t *codes = (int32_t *)malloc(sizeof(int32_t) * per * nb);
    for (size_t k = 0; k < nb; k++) SUF(blockwise_compress)(&c2, &ps, &q, blocks[k], codes + k * per);
    size_t n = per * nb;
    size_t bufsz = 4096 + 2 * (sizeof(T) * n + q.n_unpred * sizeof(T)) + 16 * n;
    uint8_t *buf = (uint8_t *)malloc(bufsz), *p = buf;
    SUF(ps_save)(&ps, &p);
    SUF(quantizer_save)(&q, &p);
    huffman_encode_main(codes, n, &p, NULL, NULL, NULL);
    size_t zcap = szo_zstd_bound((size_t)(p - buf)) + 8;
    uint8_t *z = (uint8_t *)malloc(zcap);
    size_t cs = szo_zstd_compress(buf, (size_t)(p - buf), z, zcap);
    free(z);
    free(buf);
    free(codes);
    SUF(quantizer_free)(&q);
    SUF(ps_free)(&ps);
    return (double)lc->num * (double)nb * sizeof(T) * 1.0 / (double)cs;
}

static void SUF(tn_set_dims)(szo_config *c, int N, const uint64_t *dims) { /* Config::setDims :160-177 */
    uint64_t d[4];
    for (int i = 0; i < N; i++) d[i] = dims[i];
    uint8_t keep_algo = c->cmprAlgo;
    szo_config fresh;
    szo_config_init(&fresh, N, d);
    c->N = fresh.N;
    for (int i = 0; i < 4; i++) c->dims[i] = fresh.dims[i];
    c->num = fresh.num;
    c->predDim = fresh.predDim;
    c->blockSize = fresh.blockSize;
    c->cmprAlgo = keep_algo;
}

/* the decisions only (shared by compress and by the GPU parity tests): fills conf with the tuned settings;
 * returns 1 when the tuner ran, 0 when it was skipped (then conf.cmprAlgo = ALGO_INTERP with the user's settings) */
static int SUF(tune_interp_lorenzo)(szo_config *conf, const T *data, szo_tuner_report *rep) {
    const int N = conf->N;
    if (rep) memset(rep, 0, sizeof(*rep));
    if (conf->interpAnchorStride < 0) {
        static const int def[4] = {4096, 128, 32, 16};
        conf->interpAnchorStride = def[N - 1];
    }
    const double rate = 0.005;
    static const size_t sb_def[4] = {4096, 128, 32, 16};
    size_t sbs = sb_def[N - 1];
    size_t shortest = (size_t)conf->dims[0];
    for (int i = 0; i < N; i++)
        if ((size_t)conf->dims[i] < shortest) shortest = (size_t)conf->dims[i];
    while (sbs >= shortest) sbs /= 2;
    while (sbs >= 16 && (pow((double)(sbs + 1), N) / (double)conf->num) > 1.5 * rate) sbs /= 2;
    if (sbs < 8) sbs = 8;
    int to_tune = pow((double)(sbs + 1), N) <= 0.05 * (double)conf->num;
    for (int i = 0; i < N; i++)
        if ((size_t)conf->dims[i] < sbs) to_tune = 0;
    if (rep) rep->sample_block_size = sbs;
    if (!to_tune) {
        conf->cmprAlgo = SZO_ALGO_INTERP;
        return 0;
    }
    size_t per = (size_t)pow((double)(sbs + 1), N);
    size_t *starts = NULL;
    size_t nf = SUF(tn_profile)(data, N, conf->dims, sbs, conf->absErrorBound, sbs / 4, &starts);
    int profiling = (double)(nf * per) >= 0.5 * rate * (double)conf->num;
    T **blocks = NULL;
    size_t nb = SUF(tn_sample)(data, N, conf->dims, sbs, rate, profiling, starts, nf, &blocks);
    free(starts);
    size_t sampling_num = nb * per;
    if (rep) {
        rep->n_filtered = nf;
        rep->profiling = profiling;
        rep->n_blocks = nb;
    }
    int ran = 0;
    if (sampling_num == 0 || (double)sampling_num >= (double)conf->num * 0.2) {
        conf->cmprAlgo = SZO_ALGO_INTERP;
    } else {
        ran = 1;
        double best_lorenzo = 0, best_interp = 0, ratio;
        szo_config lorenzo_config = *conf;
        conf->interpDirection = 0;
        conf->interpAlpha = 1.25;
        conf->interpBeta = 2.0;
        szo_config tc = *conf;
        uint64_t sd[4];
        for (int i = 0; i < N; i++) sd[i] = sbs + 1;
        SUF(tn_set_dims)(&tc, N, sd);
        int nr = 0;
        for (int op = 0; op < 2; op++) { /* INTERP_ALGO_LINEAR, INTERP_ALGO_CUBIC */
            tc.interpAlgo = (uint8_t)op;
            ratio = SUF(tn_interp_test)(&tc, blocks, nb, per, rep && nr < 8 ? &rep->raw_bytes[nr] : NULL);
            if (rep && nr < 8) rep->ratios[nr++] = ratio;
            if (ratio > best_interp) {
                best_interp = ratio;
                conf->interpAlgo = (uint8_t)op;
            }
        }
        tc.interpAlgo = conf->interpAlgo;
        int fact = 1;
        for (int i = 2; i <= N; i++) fact *= i;
        tc.interpDirection = (uint8_t)(fact - 1);
        ratio = SUF(tn_interp_test)(&tc, blocks, nb, per, rep && nr < 8 ? &rep->raw_bytes[nr] : NULL);
        if (rep && nr < 8) rep->ratios[nr++] = ratio;
        if (ratio > best_interp * 1.02) {
            best_interp = ratio;
            conf->interpDirection = tc.interpDirection;
        }
        tc.interpDirection = conf->interpDirection;
        static const double alphas[3] = {1.0, 1.5, 2.0}, betas[3] = {1.0, 2.5, 3.0};
        for (int i = 0; i < 3; i++) {
            tc.interpAlpha = alphas[i];
            tc.interpBeta = betas[i];
            ratio = SUF(tn_interp_test)(&tc, blocks, nb, per, rep && nr < 8 ? &rep->raw_bytes[nr] : NULL);
            if (rep && nr < 8) rep->ratios[nr++] = ratio;
            if (ratio > best_interp * 1.02) {
                best_interp = ratio;
                conf->interpAlpha = alphas[i];
                conf->interpBeta = betas[i];
            }
        }
        if (N == 1 && best_interp < 50) { /* Lorenzo is only tried in 1-D */
            lorenzo_config.cmprAlgo = SZO_ALGO_LORENZO_REG;
            SUF(tn_set_dims)(&lorenzo_config, N, sd);
            lorenzo_config.lorenzo = 1;
            lorenzo_config.lorenzo2 = 1;
            lorenzo_config.regression = 0;
            lorenzo_config.regression2 = 0;
            lorenzo_config.openmp = 0;
            lorenzo_config.blockSize = 5;
            best_lorenzo = SUF(tn_lorenzo_test)(&lorenzo_config, blocks, nb, per);
        }
        if (rep) {
            rep->best_interp = best_interp;
            rep->best_lorenzo = best_lorenzo;
        }
        int use_interp = !(best_lorenzo >= best_interp * 1.1 && best_lorenzo < 50 && best_interp < 50);
        if (use_interp) {
            conf->cmprAlgo = SZO_ALGO_INTERP;
        } else {
            if (conf->relErrorBound < 1.01e-6 && best_lorenzo > 5 && lorenzo_config.quantbinCnt != 16384) {
                int qn = lorenzo_config.quantbinCnt;
                lorenzo_config.quantbinCnt = 16384;
                ratio = SUF(tn_lorenzo_test)(&lorenzo_config, blocks, nb, per);
                if (ratio > best_lorenzo * 1.02) best_lorenzo = ratio;
                else lorenzo_config.quantbinCnt = qn;
            }
            SUF(tn_set_dims)(&lorenzo_config, N, conf->dims);
            *conf = lorenzo_config;
        }
    }
    for (size_t k = 0; k < nb; k++) free(blocks[k]);
    free(blocks);
    return ran;
}

/* SZ_compress_Interp_lorenzo (SZAlgoInterp.hpp:122-286) */
static size_t SUF(compress_interp_lorenzo)(szo_config *conf, const T *data, uint8_t *out, size_t cap, szo_stats *st) {
    SUF(tune_interp_lorenzo)(conf, data, NULL);
    if (conf->cmprAlgo == SZO_ALGO_INTERP) return SUF(compress_interp)(conf, data, out, cap, st, NULL);
    return SUF(compress_lorenzo_reg)(conf, data, out, cap, st, NULL);
}
