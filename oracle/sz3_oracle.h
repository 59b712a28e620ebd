/*
 * oracle/sz3_oracle.h — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the
 * product (sz3_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * oracle/libsz3oracle.so, and there only as the checker / timed CPU baseline.
 *
 * A plain-C (C11) restatement of the reference algorithm szcompressor/SZ3 v3.3.2 for the hot path
 *   predictor -> linear quantizer -> Huffman -> zstd          
 * Each function in sz3_oracle.c / sz3_oracle_impl.h cites the reference file:line it follows
 * (paths relative to /root/reference).  Parity pin: byte-identical compressed streams against the
 * reference itself built by `make -C oracle ref` (tests/test_oracle_vs_ref.py) and against the committed
 * golden fixtures in tests/golden/ (generated from that reference build by tests/golden/make_golden.py).
 */
#ifndef SZ3_ORACLE_H
#define SZ3_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/SZ3/utils/Config.hpp:66-91 */
enum { SZO_EB_ABS = 0, SZO_EB_REL, SZO_EB_PSNR, SZO_EB_L2NORM, SZO_EB_ABS_AND_REL, SZO_EB_ABS_OR_REL };
enum { SZO_ALGO_LORENZO_REG = 0, SZO_ALGO_INTERP_LORENZO, SZO_ALGO_INTERP, SZO_ALGO_NOPRED, SZO_ALGO_LOSSLESS };
enum { SZO_INTERP_LINEAR = 0, SZO_INTERP_CUBIC = 1 };
enum { SZO_FLOAT = 0, SZO_DOUBLE = 1 };

/* mirror of SZ3::Config's serialised + algorithm fields (include/SZ3/utils/Config.hpp:441-478) */
typedef struct szo_config {
    int32_t N;
    uint64_t dims[4]; /* slowest first, size-1 dims already dropped (Config.hpp:161-177) */
    uint64_t num;
    uint8_t cmprAlgo, errorBoundMode;
    double absErrorBound, relErrorBound, psnrErrorBound, l2normErrorBound;
    uint8_t openmp;
    int32_t quantbinCnt, blockSize;
    uint8_t predDim, dataType;
    uint8_t lorenzo, lorenzo2, regression, regression2;
    uint8_t interpAlgo, interpDirection;
    int32_t interpAnchorStride;
    double interpAlpha, interpBeta;
} szo_config;

/* diagnostics filled by szo_compress (not part of the reference API) */
typedef struct szo_stats {
    uint64_t n_unpred;       /* LinearQuantizer unpred.size() of the data quantizer */
    uint64_t raw_bytes;      /* size of the pre-zstd buffer */
    uint64_t huff_bytes;     /* encoded bit-stream bytes (outSize) */
    uint64_t n_regression_blocks, n_blocks;
    uint32_t huff_node_count;
    double t_decomp, t_hist_tree, t_encode, t_zstd; /* seconds */
} szo_stats;

/* what the ALGO_INTERP_LORENZO tuner saw and decided (diagnostics for the parity tests) */
typedef struct szo_tuner_report {
    uint64_t sample_block_size, n_filtered, n_blocks;
    int32_t profiling, reserved;
    double ratios[8]; /* trial ratios in the reference's order: linear, cubic, reversed direction, 3 x (alpha, beta) */
    double best_interp, best_lorenzo;
    uint64_t raw_bytes[8], huff_bytes[8], node_count[8], n_unpred[8]; /* per interpolation trial (pre-zstd size, ...) */
    double entropy_bits[8]; /* Shannon entropy of the trial's quantisation codes, in bits (estimator studies, tools/) */
} szo_tuner_report;

/* Config ctor semantics: setDims drops dims==1, sets N/num/predDim/blockSize defaults (Config.hpp:161-177, 452-478) */
void szo_config_init(szo_config *c, int ndims, const uint64_t *dims_slowest_first);
size_t szo_config_save(const szo_config *c, uint8_t *out);                     /* Config.hpp:312-354 */
size_t szo_config_load(szo_config *c, const uint8_t *in);                      /* Config.hpp:361-413 */

/* SZ_compress<T> / SZ_decompress<T> (include/SZ3/api/sz.hpp:43-82, 117-157) incl. 16-byte header and trailer.
 * returns compressed size, 0 on error (message in szo_last_error()). conf is taken by value semantics (copied). */
size_t szo_compress_bound(const szo_config *c, int dtype);                     /* api/impl/SZImpl.hpp:34-44 */
size_t szo_compress(const szo_config *c, int dtype, const void *data, uint8_t *out, size_t cap, szo_stats *st);
/* decData must hold conf.num elements; conf_out receives the trailer config. returns num elements, 0 on error */
size_t szo_decompress(int dtype, const uint8_t *cmp, size_t cmp_size, void *dec, szo_config *conf_out);
const char *szo_last_error(void);
/* diagnostics: the next szo_compress calls write the predictor chosen for block i (visiting order) to buf[i] (NULL: off) */
void szo_debug_selection_sink(int8_t *buf, size_t cap);

/* stage-level entry points (unit tests restating tools/test/modules/test_{quantizer,encoder,lossless}.cpp) */
/* LinearQuantizer<T>::quantize_and_overwrite / recover on one value (quantizer/LinearQuantizer.hpp:43-86).
 * returns code (0 = unpredictable); *data is overwritten with the reconstructed value when code != 0 */
int32_t szo_quantize_f32(float *data, float pred, double eb, int32_t radius);
int32_t szo_quantize_f64(double *data, double pred, double eb, int32_t radius);
float szo_recover_f32(float pred, int32_t code, double eb, int32_t radius);
double szo_recover_f64(double pred, int32_t code, double eb, int32_t radius);
/* HuffmanEncoder<int>: preprocess_encode+save+encode into out ([tree][u64 outSize][bits]); returns bytes written.
 * (encoder/HuffmanEncoder.hpp:96-125,140-218) */
size_t szo_huffman_encode(const int32_t *codes, size_t n, uint8_t *out, size_t cap);
/* load+decode; returns bytes consumed (encoder/HuffmanEncoder.hpp:225-279) */
size_t szo_huffman_decode(const uint8_t *in, size_t n, int32_t *codes);
/* Lossless_zstd::compress/decompress (lossless/Lossless_zstd.hpp:29-45): [u64 srcLen][zstd frame] */
size_t szo_zstd_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap);
size_t szo_zstd_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap);
size_t szo_zstd_bound(size_t n);
const char *szo_zstd_version(void);

/* slab count the serial oracle uses where the reference uses omp_get_num_threads() (SZImplOMP.hpp:27-36) */
void szo_set_omp_slabs(int n);

/* quantization codes of the decomposition stage only, in the reference's emission order
 * (BlockwiseDecomposition::compress / InterpolationDecomposition::compress). codes must hold conf.num ints.
 * data is NOT modified (internally copied). returns number of unpredictable values. */
size_t szo_decomposition_codes(const szo_config *c, int dtype, const void *data, int32_t *codes);

/* InterpolationDecomposition::compress only (ALGO_INTERP parameters from conf; interpAnchorStride must be >= 0):
 * codes in the reference's emission order, the element index of every code (order, may be NULL) and the reconstructed
 * array the encoder ends up with (recon, may be NULL).  returns the number of unpredictable values (incl. anchors) */
size_t szo_interp_codes(const szo_config *c, int dtype, const void *data, int32_t *codes, uint64_t *order, void *recon);


/* SZ_compress_Interp_lorenzo's decisions only (api/impl/SZAlgoInterp.hpp:122-262): conf (cmprAlgo must be
 * ALGO_INTERP_LORENZO) is updated exactly as the reference updates it before the final compress call — cmprAlgo becomes
 * ALGO_INTERP (interpAlgo / interpDirection / interpAlpha / interpBeta tuned) or ALGO_LORENZO_REG (1-D only).
 * returns 1 if the sampling trials ran, 0 if the tuner was skipped, -1 on error */
int szo_tune_interp_lorenzo(szo_config *conf, int dtype, const void *data, szo_tuner_report *rep);

#ifdef __cplusplus
}
#endif
#endif
