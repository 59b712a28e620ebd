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
 *  (4) the multi-GPU exchange (sz3hip_comm_*): an RCCL communicator over xGMI — one process driving all its GPUs
 *      (what sz3hip_compress does by itself when conf.openmp is set: slabs along dims[0] like SZ_compress_OMP,
 *      api/impl/SZImplOMP.hpp:16-117) or one process per GPU — with the sum all-reduce of the code histogram and the
 *      min/max all-reduce of the value range as its two collectives.
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
/* host-buffer API only: integers ride the f64 pipeline (exact; 64-bit values beyond 2^53 -> the array stays lossless). The numbers
 * are the reference's SZ_UINT8 .. SZ_INT64 (include/SZ3/def.hpp:27-36), the element types of its HDF5 filter (H5Z_SZ3.cpp:195-227) */
#define SZ3HIP_UINT8 2
#define SZ3HIP_INT8 3
#define SZ3HIP_UINT16 4
#define SZ3HIP_INT16 5
#define SZ3HIP_UINT32 6
#define SZ3HIP_INT32 7
#define SZ3HIP_UINT64 8
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
    uint8_t openmp; /* compress: split dims[0] into one slab per visible GPU (SZ_compress_OMP's slabs, SZImplOMP.hpp:48-55)
                     * and write the multi-slab container; trailer bit: the payload is that container */
    int32_t quantbinCnt, blockSize;
    uint8_t predDim, dataType;
    uint8_t lorenzo, lorenzo2, regression, regression2;
    uint8_t interpAlgo, interpDirection; /* interpDirection: 0 .. N! - 1, the order of the dimensions (std::next_permutation steps from the identity);
                                          * a 1-D array takes any value (one order exists); elsewhere a value beyond N! - 1 is SZ3HIP_EINVAL */
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
/* Stock-stream interoperability. READING needs no switch: sz3hip_decompress decodes stock SZ3 streams of the
 * interpolation compressor (cmprAlgo ALGO_INTERP = 2, float / double, 1-D .. 4-D — what the reference's default ALGO_INTERP_LORENZO
 * writes for anything but some 1-D arrays) and of ALGO_LOSSLESS, next to this library's own ids 16 / 17; the reconstruction is the
 * reference's bit for bit. WRITING: sz3hip_set_stock_format(1) (or SZ3HIP_STOCK_FORMAT=1 in the environment) makes sz3hip_compress —
 * and everything on top of it — write ALGO_INTERP streams stock SZ3 reads whenever the interpolation predictor is chosen (a 1-D array whose
 * default-algorithm tuner takes Lorenzo gets the reference's ALGO_LORENZO_REG container with the Config the reference goes on with); calls that
 * name ALGO_LORENZO_REG / ALGO_NOPRED get those containers; what has no stock form here keeps this library's ids. Prediction, quantisation, reconstruction and the Huffman bit stream run on the
 * GPU either way; the tree's serialisation and zstd are host stages. The tree is built with the reference's own queue (which of two
 * equal frequencies merges first, encoder/HuffmanEncoder.hpp:402-432): wherever the codes are the reference's — ALGO_INTERP, ALGO_NOPRED,
 * the default algorithm with SZ3HIP_TUNER_EXACT=1, ALGO_LORENZO_REG (round 6: the writer repeats its per-block choices against the coded array until they are the reference's) — the container
 * written IS the reference's file byte for byte, as long as its buffer leaves in one zstd frame: up to 1 MB by default (larger buffers are
 * cut into 1 MB frames for the pool's threads; stock SZ3 reads them), any size with SZ3HIP_STOCK_ONE_FRAME=1 (one host thread, ~0.4 GB/s). */
void sz3hip_set_stock_format(int on);
int sz3hip_get_stock_format(void);
/* the same over at most `avail` readable bytes: 0 when the serialised Config does not fit in them (truncated stream) */
size_t sz3hip_config_load_n(sz3hip_config *c, const unsigned char *in, size_t avail);
/* SZ_compress_size_bound<T> (api/impl/SZImpl.hpp:34-44) */
size_t sz3hip_compress_bound(const sz3hip_config *c, int dataType);
/* SZ_compress<T>(conf, data, cmpData, cmpCap) -> size (api/sz.hpp:43); `conf` is not modified (copied, sz.hpp:45).
 * conf->openmp (SZ_compress_impl, api/impl/SZImpl.hpp:10-20): dims[0] is split into independent slabs exactly like
 * SZ_compress_OMP (api/impl/SZImplOMP.hpp:48-55), one per visible GPU (SZ3HIP_GPUS caps the GPUs, SZ3HIP_SLABS sets
 * another slab count; never more slabs than dims[0]); range-based bounds use the global value range (:57-69); the code
 * histograms of all slabs are summed (RCCL all-reduce across GPUs) so that every slab is coded with the same code book;
 * the slabs are stored in the reference's multi-slab container [i32 G][Config x G][u64 size x G][blob x G] (:100-107),
 * every blob what a single-slab call would have produced.
 * Large plain calls (round 5: >= 96 MB, absolute or L2-norm bound, ALGO_LORENZO_REG / ALGO_NOPRED, no conf->openmp) are written in the
 * same container by ONE GPU as a pipeline of up to 8 pieces — the copy in of piece k + 1 beside the kernels of piece k beside the copy
 * out + zstd of piece k - 1 —, every piece with its own code book; SZ3HIP_PIECES=0 keeps them whole (INTEGRATION.md sections 6, 7).
 * The trailer's dataType field: this library's own streams (ids 16 / 17) name the element type of the call there (the decoder refuses a
 * request for another one); a stock container keeps conf->dataType as the caller left it, like the reference (Config.hpp:312-354 saves the
 * field as it finds it; neither the reference's CLI nor SZ_compress<T> sets it). */
size_t sz3hip_compress(const sz3hip_config *conf, int dataType, const void *data, char *cmpData, size_t cmpCap);
/* SZ_decompress<T>(conf, cmpData, cmpSize, decData) (api/sz.hpp:117): conf is overwritten from the trailer;
 * decData must hold conf.num elements (query with sz3hip_peek_config first). Also decodes ALGO_LOSSLESS streams and
 * multi-slab containers (trailer bit openmp; SZ_decompress_OMP, SZImplOMP.hpp:120-186), slab g on GPU g % visible GPUs; with one GPU
 * the slabs are unpacked and decoded side by side and copied out in order. Arrays of 32 MB and more leave the device through a ring
 * of pinned staging buffers whose chunks host threads copy on into decData (a fresh array's pages are faulted in by those threads). */
int sz3hip_decompress(sz3hip_config *conf, int dataType, const char *cmpData, size_t cmpSize, void *decData);
/* One algorithm of the reference's dispatcher (SZ_compress_LorenzoReg / SZ_compress_Interp / ..., SZDispatcher.hpp:28-42 and
 * their SZ_decompress_* counterparts :89-99): only the bytes between the container's 16-byte header and its Config trailer.
 * compress: returns their number (0 on error); *conf is updated like the reference updates it (absolute bound resolved,
 * cmprAlgo = id of the stream written: SZ3HIP_ALGO_HIP_LORENZO / _HIP_INTERP, or ALGO_LOSSLESS after a fallback).
 * These are what include/SZ3/api/impl/SZAlgoHip.hpp binds inside the reference's own header tree. */
size_t sz3hip_compress_blob(sz3hip_config *conf, int dataType, const void *data, char *blob, size_t cap);
int sz3hip_decompress_blob(const sz3hip_config *conf, int dataType, const char *blob, size_t size, void *decData);
/* reads only header + trailer (what SZ_decompress does before dispatching, sz.hpp:119-141) */
int sz3hip_peek_config(sz3hip_config *conf, const char *cmpData, size_t cmpSize);

/* ---- (3) device-resident API -------------------------------------------------------------------------------- */
typedef struct sz3hip_ctx sz3hip_ctx;

/* workspace for arrays of up to max_elems elements of dataType on HIP device `device`: the arrays every call uses are
 * allocated here; a few that only some paths use (the block predictor's per-block arrays, the interpolation work array, the f64
 * decoder's int32 intermediates, the decoder's carry array) with the first call that takes that path — later calls allocate nothing */
sz3hip_ctx *sz3hip_ctx_create(int device, uint64_t max_elems, int dataType);
void sz3hip_ctx_destroy(sz3hip_ctx *ctx);
/* upper bound of the device payload for n elements (outlier lists of up to n / 32 entries: a compress call that needs more
 * returns SZ3HIP_EOUTLIERS when the buffer is this size) */
size_t sz3hip_payload_bound(const sz3hip_ctx *ctx, uint64_t n);
/* the bound with the largest outlier lists the library will build (n / 8 entries: rough fields at tight bounds, small
 * quantbinCnt). With a buffer this large sz3hip_compress_device grows its lists on demand and retries instead of
 * returning SZ3HIP_EOUTLIERS (the reference keeps any number of unpredictable values, LinearQuantizer.hpp:43-66). */
size_t sz3hip_payload_bound_max(const sz3hip_ctx *ctx, uint64_t n);
/* The bound for one Config: like the two above (worst_case != 0: the second), and also sufficient for the block-composed predictor's
 * side section on thin arrays (an extent below 9), which have more blocks per element than the shape-blind bounds assume. */
size_t sz3hip_payload_bound_conf(const sz3hip_ctx *ctx, const sz3hip_config *conf, int worst_case);

/* global min/max of a device array (K0: utils/Statistic.hpp:12-21); result written to host doubles (synchronises) */
int sz3hip_minmax_device(sz3hip_ctx *ctx, const void *d_in, uint64_t n, double *min_out, double *max_out, void *stream);

/* stage 1: prequantise + integer Lorenzo + code emission + outlier capture + histogram (K1+K4).
 * conf: N/dims/absErrorBound(must already be absolute)/quantbinCnt are used. Asynchronous on `stream` — except for the paths
 * that decide on the host from a device result before going on: ALGO_INTERP_LORENZO (the auto-tuner's outcome) and
 * ALGO_LORENZO_REG with lorenzo2 / regression on 3-D arrays (the selection pass's count: a field on which only first-order Lorenzo
 * is chosen is handed to the plain Lorenzo path) synchronise the stream once inside this call.
 * d_in must stay valid and unchanged until sz3hip_compress_finish returns: a context that took a shortcut from what its
 * previous call found (the one-launch form of stage 1 assumes the previous call's code width) repeats the call from stage 1
 * inside finish() when this call's data says otherwise.
 * Predictor sets of ALGO_LORENZO_REG (api/impl/SZAlgoLorenzoReg.hpp:22-64): lorenzo alone = the plain Lorenzo stream (N = 1..4);
 * with regression: per-block choice in 1-D (blockSize 4..65535, default 128), 2-D (4..32, default 16) and 3-D (4..8, default 6);
 * lorenzo2: 3-D. Elsewhere a set that contains lorenzo is coded by it alone (the trailer records that), others are refused
 * with SZ3HIP_EUNSUPPORTED. */
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
/* inverse: payload (device) -> d_out (device, n elements). Synchronises once to read the 160-byte header. */
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
    int32_t interpAlgo, interpDirection;
    int32_t speculated; /* 0: stage 1 waited for the tuner; 1: it had started with the previous call's outcome and the tuner confirmed it;
                         * 2: started, not confirmed, enqueued again with this call's outcome */
    double interpAlpha, interpBeta;
    double est_bytes[8]; /* priced size of the trials: linear, cubic, reversed direction, 3 x (alpha, beta), [6] Lorenzo (1-D),
                          * [7] Lorenzo with 16384 quantization bins (1-D, only when Lorenzo won: SZAlgoInterp.hpp:268-277) */
} sz3hip_tuner_report;
int sz3hip_get_tuner_report(sz3hip_ctx *ctx, sz3hip_tuner_report *rep);

/* per-stage kernel time of the last compress / decompress when profiling is on (hipEvents on `stream`):
 * names[i] / ms[i] for i < returned count; count 0 if profiling is off */
void sz3hip_set_profiling(sz3hip_ctx *ctx, int on);
int sz3hip_get_stage_times(sz3hip_ctx *ctx, const char **names, float *ms, int max);
/* What a context remembers between calls only ever changes the TIME of a call, never its payload: which form of a kernel the
 * previous call's data took, histogram windows, the auto-tuner's previous outcome (stage 1 starts with it beside the tuner) and
 * the previous code book (stage 2 packs with it while this call's is built by one workgroup of the same launch; finish() repeats
 * the encoder when the two differ and lets the next 1, 2, 4, 8 calls sit out). sz3hip_ctx_forget drops all of it: the next call
 * behaves like a context's first (bench.py's cold numbers). sz3hip_ctx_set_speculation(ctx, off): 0 on, 1 off, 2 on without the
 * back-off after a miss (tests); sz3hip_get_spec_stats counts the code-book speculation's hits / misses. */
void sz3hip_ctx_forget(sz3hip_ctx *ctx);
void sz3hip_ctx_set_speculation(sz3hip_ctx *ctx, int off);
/* Round 4: one exception to "never its payload", and the switch that removes it. The verdict on the previous call's code book
 * accepts it when it is a complete prefix code over this call's alphabet AND codes this call's symbols within 1/1024 of the size
 * this call's own book would give (the format stores code lengths, a decoder cannot tell): a series of similar but not identical
 * arrays keeps the shortcut, and a payload then depends on what the context coded before (its size by < 0.1 %, its decoded
 * values not at all). sz3hip_ctx_set_deterministic(ctx, 1): the previous book stands only when it IS this call's book — the
 * payload is a pure function of the input again (what the host API — sz3hip_compress, the CLI, the HDF5 filter — always sets;
 * the reference builds a tree per call, encoder/HuffmanEncoder.hpp:96-105). Default of a device context: 0.
 * Round 6: the streams this mattered most for no longer speculate at all. A Lorenzo stream (ALGO_LORENZO_REG with Lorenzo-1 alone,
 * ALGO_NOPRED) of an array of at least 2^22 elements in rows of whole 256-element segments (1-D ... 3-D) that turns out to have
 * one-byte codes is coded with a book built from a SAMPLE of the array (262 144 values at places the extents alone decide) — inside
 * stage 1's own launch once the context knows the stream's form, by a launch of its own otherwise. Sample and book are functions of
 * the input: such a payload is the same whatever the context coded before and whatever this switch says, in the time the speculative
 * path took on a hit (512^3 f32: 0.24 ms either way; 0.30 ms with this switch on before). Not for contexts whose histogram is
 * exchanged between the stages (sz3hip_histogram_ptr / sz3hip_ctx_set_histogram: the ranks share one book from the summed histogram).
 * The payload header's anchor_stride field of such a stream names the symbol that stands for a listed delta (sz3hip_format.h). */
void sz3hip_ctx_set_deterministic(sz3hip_ctx *ctx, int on);
/* Round 5: the ALGO_INTERP_LORENZO tuner's trials priced the reference's way. By default a trial is priced on the device from the
 * histogram of its codes (entropy + a model of the serialised tree: 0.2 ms per tuning, the reference's decision in about three of four
 * cases, a neighbouring (alpha, beta) otherwise). With this switch every trial is priced as interp_compress_test does
 * (api/impl/SZAlgoInterp.hpp:42-78): the trial kernel's per-element codes of all sampled blocks come to the host, are put into the order
 * the reference's decomposition emits them, coded with one Huffman tree built with the reference's own queue, serialised like its buffer and
 * compressed with ZSTD_compress at level 3 — est_bytes[0..5] of the tuner report are then the reference's own compressed sizes byte for
 * byte (same libzstd) and the decisions the reference's, at a few milliseconds per tuning (a host thread per trial). The 1-D Lorenzo
 * trials (est_bytes[6], [7]) are then walked on the host in the reference's own order and priced the same way. Where a trial cannot be
 * priced this way (an anchor stride that is no power of two) the tuning goes on with the estimates. The HOST API's contexts (sz3hip_compress, and through it
 * the C++ / C wrappers, the CLI, the HDF5 filter) have it ON by default — a caller of the reference's boundary gets the reference's
 * decisions; at no cost for arrays of 16 MB and more under an absolute bound: the tuner then runs from the host's copy of the array beside its copy to
 * the device (512^3 f32 host to host: 15.1 ms per call either way; SZ3HIP_NO_PRETUNE=1: inside stage 1 as before, 18.1) —, a device context (sz3hip_ctx_create)
 * OFF. Environment, read per call, for every context this function was never called on: SZ3HIP_TUNER_EXACT=1 on, =0 off. */
void sz3hip_ctx_set_tuner_exact(sz3hip_ctx *ctx, int on);
void sz3hip_get_spec_stats(const sz3hip_ctx *ctx, uint32_t *hits, uint32_t *misses);
/* 1 when the last finished compression of this context ran the fused stage 1 (round 4: a context whose previous call left a small
 * code book codes with it INSIDE the predictor kernel — one-byte codes, rows that are multiples of 256 elements, 1-D..3-D — and the
 * encoder only moves the rows' bit strings to their places; the verdict on the book is as for the unfused form, a miss repeats
 * the whole call in the two-pass form: the input must stay valid until sz3hip_compress_finish returns) */
int sz3hip_last_call_fused(const sz3hip_ctx *ctx);
/* 1 when the last finished compression of this context ran the 16-bit form of the one-byte stage-1 kernel (round 5: f32 data, 1-D ... 3-D
 * arrays, behind a call whose probe found every lattice value within +-2047 steps): same bytes out as the one-byte kernel at 13 instead
 * of 21 vector instructions per element. A lattice value beyond +-4095 (or a value that is not finite) voids the launch and the call is
 * repeated with the one-byte kernel (the input must stay valid until sz3hip_compress_finish returns, as for every one-launch form). */
int sz3hip_last_call_q16(const sz3hip_ctx *ctx);
/* opt in to (1) / out of (0, the default) that form for this context; SZ3HIP_FUSED=1 in the environment makes 1 the default of every
 * context created afterwards. It halves the encoder's HBM traffic (no code array) and is byte-identical to the two-pass form, but on
 * MI355X it measured slower ("Round 4: the single-pass encoder"): the default stays the two-pass form. */
void sz3hip_ctx_set_fused(sz3hip_ctx *ctx, int on);
/* 1: this library carries the superseded forms kept for reference — the fused stage 1 above and the decoder's multi-symbol table
 * (sz3hip_debug_flags(2)) — i.e. it is the lab build (python -m sz3_amd.build --lab: sz3_amd/libsz3hip_lab.so). The product build
 * (libsz3hip.so) leaves them out: 0, and both switches do nothing. */
int sz3hip_lab_build(void);
/* test hooks: copy internal device arrays to host (quantisation codes as uint16, histogram as uint64) */
int sz3hip_debug_copy_codes(sz3hip_ctx *ctx, uint16_t *host_codes, uint64_t n);
/* test hook: which chain the last sz3hip_decompress_device took: out4[0] half-width intermediates, [1] rows that cross chunk
 * boundaries (carry pass), [2] calls left before the half-width chain is tried again after an overflow, [3] reserved */
int sz3hip_debug_decode_info(sz3hip_ctx *ctx, uint32_t *out4);
/* test hook: non-zero routes every shape through the generic (any-shape) stage-1 kernel instead of the tuned one */
void sz3hip_debug_force_generic(int on);
/* development switches (bit mask, process-wide; 0 = product behaviour). They force one of two equivalent paths, results unchanged (tests compare them):
 * 1 the wide code book compacts the histogram inside its own workgroup (round 5's default: k_cb_compact over the whole chip in front of it),
 * 2 Lorenzo decoder with the multi-symbol lookup table for small code books (round 5: up to three code words per 12-bit window; same output,
 * measured slower than the one-symbol table in its first form: opt-in),
 * 4 the one-launch block decoders' retry through the launch-per-front decoders, taken as if a flag poll had given up,
 * 8 no 16-bit form of the one-byte stage-1 kernel (round 5), 32 no marching kernel, 64 no one-byte codes, 128 interpolation pass by pass, one point per thread (no 8-wide level-1
 * kernels, no level kernels), 256 no stage-1 specialisation by code width, 512 decoder without the fused x prefix sum,
 * 1024 code book without the two-class construction, 4096 stage 1 without the XCD-aware task order, 8192 interpolation
 * histogram with the large tier and the windowed tail passes, 16384 predictor sets with Lorenzo-2 / regression fall back
 * to plain Lorenzo (no block path), 131072 contexts do not remember the previous call's code width / code-book form,
 * 262144 round-parallel Huffman merge for small alphabets, 2097152 Lorenzo decoder without half-width intermediates,
 * 4194304 interpolation level kernels whatever the array's size (normally from 256 blocks up), 536870912 the level kernels hand the grid of
 * stride 2 over in place (round 5's default: as a dense array — no partial-line stores at the level of stride 2, no strided gather at the finest). The 3-D block decoder (blocks of
 * 6^3; the product path is ONE launch for the chain of fronts, k_blk_wave3, after a local pass straight from the codes): 16 the
 * local pass a wave per block from an expanded copy of the deltas, 32768 groups of 3 x 3 x 3 blocks in closed form with a launch
 * per front (also what a one-launch decoder whose flag poll gave up falls back to), 8388608 a block per wave (65536 — round 3's groups
 * of 2 x 2 x 2 blocks inverted by line scans — exists in -DSZ3HIP_LAB builds only since round 5; in the product library it takes the block-per-wave form). The 2-D one (block edges
 * up to 16, no second-order member; product: k_blkn_wave2, one launch): 65536 groups of 4 x 4 blocks with a launch per front,
 * 8388608 a block per wave.
 * Experiments with WRONG or slower results (tools/dec_lab.py): 524288 decoder without stores, 1048576 decoder with direct stores. */
void sz3hip_debug_flags(int flags);

/* ---- (4) multi-GPU exchange over RCCL / xGMI ------------------------------------------------------------------- */
typedef struct sz3hip_comm sz3hip_comm;
#define SZ3HIP_COMM_ID_BYTES 128
/* one process, `ndev` GPUs (devices[i], or 0..ndev-1 when NULL; ndev <= 0: all visible): ncclCommInitAll. The
 * communicator has ndev members, all local. */
sz3hip_comm *sz3hip_comm_create_local(int ndev, const int *devices);
/* one process per GPU: rank 0 makes the id (ncclGetUniqueId), the launcher ships its SZ3HIP_COMM_ID_BYTES bytes to
 * every rank, each rank joins with its device (ncclCommInitRank). The communicator has one local member. */
int sz3hip_comm_unique_id(unsigned char *id128);
sz3hip_comm *sz3hip_comm_create_rank(int nranks, int rank, int device, const unsigned char *id128);
void sz3hip_comm_destroy(sz3hip_comm *comm);
int sz3hip_comm_size(const sz3hip_comm *comm);       /* ranks RCCL reports for the communicator */
int sz3hip_comm_rank(const sz3hip_comm *comm);       /* rank of this process's first member (0 for a local communicator) */
int sz3hip_comm_local_size(const sz3hip_comm *comm); /* members this process drives */
int sz3hip_comm_device(const sz3hip_comm *comm, int member);
/* Between stage1 and stage2: in-place sum all-reduce of the code histogram of ctxs[m] (one context per local member, on
 * that member's device), enqueued on streams[m] — stage2 on the same stream sees the global histogram, and every member
 * builds the same code book. Asynchronous. */
int sz3hip_comm_allreduce_histogram(sz3hip_comm *comm, sz3hip_ctx *const *ctxs, void *const *streams);
/* the same for any uint64 device buffers (d_bufs[m] on member m's device) */
int sz3hip_comm_allreduce_u64(sz3hip_comm *comm, void *const *d_bufs, size_t count, void *const *streams);
/* global value range for REL / PSNR / ABS_AND_REL / ABS_OR_REL bounds (api/impl/SZImplOMP.hpp:57-69): mins[m] / maxs[m]
 * hold member m's local pair on entry and the global pair on return (synchronises the streams) */
int sz3hip_comm_allreduce_minmax(sz3hip_comm *comm, double *mins, double *maxs, void *const *streams);


/* One process per GPU (communicator from sz3hip_comm_create_rank): this rank's share of SZ_compress_OMP. `global_conf`
 * describes the WHOLE array; rank r of G owns the slab [r*dims[0]/G, (r+1)*dims[0]/G) (SZImplOMP.hpp:48-50) and passes
 * a host pointer to that slab. Collective: every rank of the communicator must call it (value range for range-based
 * bounds, a status word, the code histogram). Writes this slab's blob — what the container stores for it — to `blob`
 * (capacity >= sz3hip_compress_bound of the slab's Config) and the slab's Config to *slab_conf; returns the blob size,
 * 0 on error (then on every rank). The launcher gathers blobs + Configs and rank 0 calls sz3hip_assemble_container. */
size_t sz3hip_compress_rank(sz3hip_comm *comm, const sz3hip_config *global_conf, int dataType, const void *slab_data,
                            char *blob, size_t cap, sz3hip_config *slab_conf);
/* [magic][version][u64 body][ i32 G | Config x G | u64 size x G | blob x G ][outer Config, openmp = 1] (api/sz.hpp:53-81
 * around SZImplOMP.hpp:100-107): the stream sz3hip_decompress / SZ_decompress<T> read back. Returns its size, 0 on error. */
size_t sz3hip_assemble_container(const sz3hip_config *global_conf, int dataType, int G, const sz3hip_config *slab_confs,
                                 const char *const *blobs, const size_t *blob_sizes, char *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
