/*
 * oracle/sz3_oracle_interp.h — TEST INFRASTRUCTURE ONLY (see sz3_oracle.h).
 * InterpolationDecomposition + SZ_compress_Interp(_lorenzo) restatement. (type-generic; included from
 * sz3_oracle_impl.h after the quantizer so that T / SUF are defined)
 */
static size_t SUF(compress_interp)(const szo_config *conf, const T *data, uint8_t *out, size_t cap, szo_stats *st,
                                   int32_t *codes_out) {
    (void)conf; (void)data; (void)out; (void)cap; (void)st; (void)codes_out;
    set_err("oracle: interpolation not restated yet");
    return 0;
}
static size_t SUF(compress_interp_lorenzo)(szo_config *conf, const T *data, uint8_t *out, size_t cap, szo_stats *st) {
    (void)conf; (void)data; (void)out; (void)cap; (void)st;
    set_err("oracle: interpolation not restated yet");
    return 0;
}
static int SUF(decompress_interp)(const szo_config *conf, const uint8_t *cmp, size_t cmp_size, T *dec) {
    (void)conf; (void)cmp; (void)cmp_size; (void)dec;
    return set_err("oracle: interpolation not restated yet");
}
