/*
 * include/sz3hip.h — C ABI of libsz3hip.so: the MI355X (gfx950) implementation of the SZ3 hot path
 *     predictor -> linear quantizer -> Huffman (-> zstd on the host side of the boundary)
 *
 * Plain pointers and sizes only; no C++/torch types.  Three groups of entry points:
 *
 *  (1) the reference's own C ABI for this path, tools/sz3c/include/sz3c.h:52-59
 *      (SZ_compress_args / SZ_decompress / free_buf) — same names, argument meaning, malloc ownership and
 *      "unsupported => printf + exit(0)" behaviour — declared in include/sz3c.h of this repository.
 *  (2) sz3hip_compress / sz3hip_decompress (+ bound / config save / load): what the reference's C++ templates
 *      SZ_compress<T>(conf, data, cmpData, cmpCap) and SZ_decompress<T>(conf, cmpData, cmpSize, decData)
 *      (include/SZ3/api/sz.hpp:43,117) do, with the full SZ3::Config passed as the POD `sz3hip_config`
 *      (mirror of include/SZ3/utils/Config.hpp:441-478).  Host pointers in, host pointers out; the stream is the
 *      reference container (16-byte header + payload + Config trailer, sz.hpp:53-81) whose trailer carries the
 *      new cmprAlgo id SZ3HIP_ALGO_LORENZO (stock SZ3 rejects it with "Unknown compression algorithm",
 *      api/impl/SZDispatcher.hpp:98 — the honest behaviour: GPU streams are not decodable by the CPU reference).
 *  (3) the device-resident API (sz3hip_ctx_*, sz3hip_*_device): input already in HBM, payload left in HBM,
 *      split into stage1 (predict+quantize+histogram) and stage2 (codebook+encode) so that a multi-GPU caller
 *      can all-reduce the histogram between them.
 *
 * Error handling: functions returning int return 0 on success, a negative SZ3HIP_E* code otherwise;
 * functions returning size_t return 0 on error. sz3hip_last_error() gives the message (thread-local).
 */
#ifndef SZ3HIP_H
#define SZ3HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* data types: include/SZ3/utils/Config.hpp:27-36 */
#define SZ3HIP_FLOAT 0
#define SZ3HIP_DOUBLE 1
#define SZ3HIP_INT32 7 /* host-buffer API only: integers ride the f64 pipeline (exact; int64 beyond 2^53 -> lossless) */
#define SZ3HIP_INT64 9

/* error-bound modes: include/SZ3/utils/Config.hpp:66 (enum EB) */
enum { SZ3HIP_EB_ABS = 0, SZ3HIP_EB_REL, SZ3HIP_EB_PSNR, SZ3HIP_EB_L2NORM, SZ3HIP_EB_ABS_AND_REL, SZ3HIP_EB_ABS_OR_REL };

/* algorithms: include/SZ3/utils/Config.hpp:80 (enum ALGO) + the ids of the GPU stream formats */
enum {
    SZ3HIP_ALGO_LORENZO_REG = 0,
    SZ3HIP_ALGO_INTERP_LORENZO = 1,
    SZ3HIP_ALGO_INTERP = 2,
    SZ3HIP_ALGO_NOPRED = 3,
    SZ3HIP_ALGO_LOSSLESS = 4,
    SZ3HIP_ALGO_HIP_LORENZO = 16, /* dual-quantisation integer Lorenzo + chunked canonical Huffman (this library) */
    SZ3HIP_ALGO_HIP_INTERP = 17   /* the reference's multilevel interpolation, pass-parallel, same codes bit for bit */
};

enum {
    SZ3HIP_OK = 0,
    SZ3HIP_EINVAL = -1,      /* std::invalid_argument in the reference */
    SZ3HIP_ECAPACITY = -2,   /* buffer too small (SZ3_ERROR_COMP_BUFFER_NOT_LARGE_ENOUGH) */
    SZ3HIP_EFORMAT = -3,     /* bad magic / version / corrupt stream */
    SZ3HIP_EHIP = -4,        /* HIP runtime error */
    SZ3HIP_EUNSUPPORTED = -5,
    SZ3HIP_EOUTLIERS = -6,   /* outlier lists overflowed: caller falls back to lossless like SZDispatcher.hpp:44-59 */
    SZ3HIP_EZSTD = -7
};

/* POD mirror of SZ3::Config (include/SZ3/utils/Config.hpp:441-478); dims slowest first, like Config::dims */
typedef struct sz3hip_config {
    int32_t N;
    uint64_t dims[4];
    uint64_t num;
    uint8_t cmprAlgo, errorBoundMode;
    double absErrorBound, relErrorBound, psnrErrorBound, l2normErrorBound;
    uint8_t openmp; /* here: "slab container" flag — set when the payload holds several independent slabs */
    int32_t quantbinCnt, blockSize;
    uint8_t predDim, dataType;
    uint8_t lorenzo, lorenzo2, regression, regression2;
    uint8_t interpAlgo, interpDirection;
    int32_t interpAnchorStride;
    double interpAlpha, interpBeta;
} sz3hip_config;

const char *sz3hip_last_error(void);
int sz3hip_last_error_code(void); /* SZ3HIP_E* of the last failure on this thread (for wrappers that map codes to exceptions) */
const char *sz3hip_version(void);

/* ---- (2) Config + host-buffer API ------------------------------------------------------------------------ */
/* SZ3::Config(dims...) : drops size-1 dims, sets N/num/predDim/blockSize and all defaults (Config.hpp:146-177,452-478) */
void sz3hip_config_init(sz3hip_config *c, int ndims, const uint64_t *dims_slowest_first);
/* Config::save / Config::load (Config.hpp:312-413); return bytes written / consumed */
size_t sz3hip_config_save(const sz3hip_config *c, unsigned char *out);
size_t sz3hip_config_load(sz3hip_config *c, const unsigned char *in);
/* SZ_compress_size_bound<T> (api/impl/SZImpl.hpp:34-44) */
size_t sz3hip_compress_bound(const sz3hip_config *c, int dataType);
/* SZ_compress<T>(conf, data, cmpData, cmpCap) -> size (api/sz.hpp:43); `conf` is not modified (copied, sz.hpp:45).
 * nslabs >= 1 splits dims[0] into independent slabs exactly like SZ_compress_OMP (api/impl/SZImplOMP.hpp:48-55)
 * and stores them in the reference's multi-slab container (SZImplOMP.hpp:100-107). */
size_t sz3hip_compress(const sz3hip_config *conf, int dataType, const void *data, char *cmpData, size_t cmpCap);
/* SZ_decompress<T>(conf, cmpData, cmpSize, decData) (api/sz.hpp:117): conf is overwritten from the trailer;
 * decData must hold conf.num elements (query with sz3hip_peek_config first). Also decodes ALGO_LOSSLESS streams. */
int sz3hip_decompress(sz3hip_config *conf, int dataType, const char *cmpData, size_t cmpSize, void *decData);
/* reads only header + trailer (what SZ_decompress does before dispatching, sz.hpp:119-141) */
int sz3hip_peek_config(sz3hip_config *conf, const char *cmpData, size_t cmpSize);

/* ---- (3) device-resident API -------------------------------------------------------------------------------- */
typedef struct sz3hip_ctx sz3hip_ctx;

/* workspace for arrays of up to max_elems elements of dataType on HIP device `device` (all device memory is
 * allocated here, nothing is allocated inside the compress/decompress calls) */
sz3hip_ctx *sz3hip_ctx_create(int device, uint64_t max_elems, int dataType);
void sz3hip_ctx_destroy(sz3hip_ctx *ctx);
/* upper bound of the device payload for n elements (outlier lists of up to n / 32 entries: a compress call that needs more
 * returns SZ3HIP_EOUTLIERS when the buffer is this size) */
size_t sz3hip_payload_bound(const sz3hip_ctx *ctx, uint64_t n);
/* the bound with the largest outlier lists the library will build (n / 8 entries: rough fields at tight bounds, small
 * quantbinCnt). With a buffer this large sz3hip_compress_device grows its lists on demand and retries instead of
 * returning SZ3HIP_EOUTLIERS (the reference keeps any number of unpredictable values, LinearQuantizer.hpp:43-66). */
size_t sz3hip_payload_bound_max(const sz3hip_ctx *ctx, uint64_t n);

/* global min/max of a device array (K0: utils/Statistic.hpp:12-21); result written to host doubles (synchronises) */
int sz3hip_minmax_device(sz3hip_ctx *ctx, const void *d_in, uint64_t n, double *min_out, double *max_out, void *stream);

/* stage 1: prequantise + integer Lorenzo + code emission + outlier capture + histogram (K1+K4).
 * conf: N/dims/absErrorBound(must already be absolute)/quantbinCnt are used. Asynchronous on `stream`. */
int sz3hip_compress_stage1(sz3hip_ctx *ctx, const sz3hip_config *conf, const void *d_in, void *stream);
/* device pointer to the code histogram: uint64_t[sz3hip_histogram_len()] — the buffer a multi-GPU caller
 * all-reduces (sum) between stage1 and stage2 */
void *sz3hip_histogram_ptr(sz3hip_ctx *ctx);
size_t sz3hip_histogram_len(const sz3hip_ctx *ctx);
/* let the caller own the histogram buffer (uint64_t[sz3hip_histogram_len()] in device memory), e.g. a tensor that its
 * communication library can all-reduce in place; NULL restores the internal buffer */
int sz3hip_ctx_set_histogram(sz3hip_ctx *ctx, void *d_hist);
/* stage 2: canonical codebook from the histogram (K5), chunked Huffman bit-pack (K6), payload assembly into
 * d_payload (device, capacity cap bytes). Asynchronous on `stream`. */
int sz3hip_compress_stage2(sz3hip_ctx *ctx, void *d_payload, size_t cap, void *stream);
/* waits for `stream`, returns the payload size in *payload_size (host), SZ3HIP_EOUTLIERS / SZ3HIP_ECAPACITY on overflow */
int sz3hip_compress_finish(sz3hip_ctx *ctx, size_t *payload_size, void *stream);
/* stage1 + stage2 + finish */
int sz3hip_compress_device(sz3hip_ctx *ctx, const sz3hip_config *conf, const void *d_in, void *d_payload, size_t cap,
                           size_t *payload_size, void *stream);
/* inverse: payload (device) -> d_out (device, n elements). Synchronises once to read the 128-byte header. */
int sz3hip_decompress_device(sz3hip_ctx *ctx, const void *d_payload, size_t payload_size, void *d_out, void *stream);

/* diagnostics of the last compress on this ctx (valid after sz3hip_compress_finish) */
typedef struct sz3hip_stats {
    uint64_t n, n_value_outliers, n_delta_outliers, n_chunks, bitstream_bytes, payload_bytes;
    uint32_t n_symbols, max_code_len;
    uint32_t narrow_codes; /* 1 when stage 1 kept the intermediate codes as one byte each (internal, not a format property) */
    uint32_t reserved;
} sz3hip_stats;
int sz3hip_get_stats(sz3hip_ctx *ctx, sz3hip_stats *st);

/* what the ALGO_INTERP_LORENZO sampling auto-tuner (SZ_compress_Interp_lorenzo, api/impl/SZAlgoInterp.hpp:122-286) saw and
 * decided in the last sz3hip_compress_stage1 / sz3hip_compress_device call of this context */
typedef struct sz3hip_tuner_report {
    int32_t ran;        /* 1: the sampling trials ran; 0: skipped like the reference (:149-162, :176-179) or another cmprAlgo */
    int32_t use_interp; /* 1: interpolation chosen; 0: Lorenzo (possible in 1-D only, :232-250) */
    uint64_t sample_block_size, n_filtered, n_blocks;
    int32_t profiling;
    int32_t interpAlgo, interpDirection, reserved;
    double interpAlpha, interpBeta;
    double est_bytes[8]; /* priced size of the trials: linear, cubic, reversed direction, 3 x (alpha, beta), [6] Lorenzo (1-D) */
} sz3hip_tuner_report;
int sz3hip_get_tuner_report(sz3hip_ctx *ctx, sz3hip_tuner_report *rep);

/* per-stage kernel time of the last compress / decompress when profiling is on (hipEvents on `stream`):
 * names[i] / ms[i] for i < returned count; count 0 if profiling is off */
void sz3hip_set_profiling(sz3hip_ctx *ctx, int on);
int sz3hip_get_stage_times(sz3hip_ctx *ctx, const char **names, float *ms, int max);
/* test hooks: copy internal device arrays to host (quantisation codes as uint16, histogram as uint64) */
int sz3hip_debug_copy_codes(sz3hip_ctx *ctx, uint16_t *host_codes, uint64_t n);
/* test hook: non-zero routes every shape through the generic (any-shape) stage-1 kernel instead of the tuned one */
void sz3hip_debug_force_generic(int on);
/* development switches (bit mask, process-wide; 0 = product behaviour). Bits 1..16: ablations of the stage-1 kernel for
 * tools/k1_lab.py - results are WRONG. The others force one of two equivalent paths, results unchanged (tests compare them):
 * 32 no marching kernel, 64 no one-byte codes, 128 interpolation without the 8-wide level-1 kernels, 256 no stage-1
 * specialisation by code width, 512 decoder without the fused x prefix sum, 1024 code book without the two-class
 * construction, 4096 stage 1 without the XCD-aware task order, 8192 interpolation histogram with the large tier and the
 * windowed tail passes */
void sz3hip_debug_flags(int flags);

#ifdef __cplusplus
}
#endif
#endif
