// sz3_amd/csrc/sz3hip_api.cpp — host side of libsz3hip.so: the C ABI declared in include/sz3hip.h and include/sz3c.h.
//
// Mirrors, for the GPU path, what these reference pieces do on the CPU (paths relative to /root/reference):
//   include/SZ3/api/sz.hpp:43-82,117-157        container: 16-byte header + payload + Config trailer
//   include/SZ3/utils/Config.hpp:161-177,312-413 Config::setDims / save / load
//   include/SZ3/api/impl/SZDispatcher.hpp:13-100 eb-mode conversion, eb==0 => lossless, lossless fallback,
//                                                "ratio < 3 => also try zstd alone"
//   include/SZ3/lossless/Lossless_zstd.hpp:29-45 [u64 rawLen][zstd frames]  (we emit several concatenated frames,
//                                                compressed by a thread pool; any zstd decoder reads them)
//   tools/sz3c/src/sz3c.cpp:11-94                SZ_compress_args / SZ_decompress / free_buf
// There is NO CPU implementation of the predictor/quantizer/Huffman stages in this library: if the HIP device
// or kernels are unavailable every entry point fails loudly.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sz3c.h"
#include "../../include/sz3hip.h"
#include "sz3hip_format.h"
#include "sz3hip_kernels.h"

// ------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512];
static thread_local int g_err_code = 0;
static int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    g_err_code = code;
    return code;
}
extern "C" const char *sz3hip_last_error(void) { return g_err; }
extern "C" int sz3hip_last_error_code(void) { return g_err_code; }
extern "C" const char *sz3hip_version(void) { return "sz3hip 0.1 (gfx950; data format SZ3 3.3.2 container, payload SZH1)"; }

#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail(SZ3HIP_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------------------------
// libzstd (third-party, the reference's lossless stage; not vendored there either — CMakeLists.txt:69-75).
// zstd.h is not installed in /usr/include of this image, so the five prototypes are declared here and the
// library is dlopen'ed; a missing library is a hard error.
// ------------------------------------------------------------------------------------------------------------
namespace zs {
typedef size_t (*compress_fn)(void *, size_t, const void *, size_t, int);
typedef size_t (*decompress_fn)(void *, size_t, const void *, size_t);
typedef size_t (*bound_fn)(size_t);
typedef unsigned (*iserr_fn)(size_t);
typedef size_t (*framesize_fn)(const void *, size_t);
typedef unsigned long long (*contentsize_fn)(const void *, size_t);
static void *h;
static compress_fn compress;
static decompress_fn decompress;
static bound_fn bound;
static iserr_fn is_error;
static framesize_fn frame_csize;
static contentsize_fn frame_content;
static std::once_flag once;
static bool ok;
static void load_once() {
    const char *names[] = {"libzstd.so.1", "libzstd.so", "/usr/lib/x86_64-linux-gnu/libzstd.so.1", nullptr};
    for (int i = 0; names[i] && !h; i++) h = dlopen(names[i], RTLD_NOW);
    if (!h) return;
    compress = (compress_fn)dlsym(h, "ZSTD_compress");
    decompress = (decompress_fn)dlsym(h, "ZSTD_decompress");
    bound = (bound_fn)dlsym(h, "ZSTD_compressBound");
    is_error = (iserr_fn)dlsym(h, "ZSTD_isError");
    frame_csize = (framesize_fn)dlsym(h, "ZSTD_findFrameCompressedSize");
    frame_content = (contentsize_fn)dlsym(h, "ZSTD_getFrameContentSize");
    ok = compress && decompress && bound && is_error;
}
static int load() {
    std::call_once(once, load_once);
    return ok ? 0 : fail(SZ3HIP_EZSTD, "libzstd.so.1 not found or incomplete");
}
static const size_t FRAME = 4u << 20;  // bytes of input per zstd frame
static unsigned nthreads() {
    const char *e = getenv("SZ3HIP_ZSTD_THREADS");
    unsigned n = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return n;
}
static size_t bound_frames(size_t n) {
    size_t nf = (n + FRAME - 1) / FRAME;
    if (nf == 0) nf = 1;
    return nf * bound(std::min(n, FRAME)) + 8;
}
// [u64 srcLen][frame]...  level 3 (lossless/Lossless_zstd.hpp:48); returns 0 on error
static size_t compress_frames(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {
    if (load()) return 0;
    if (cap < 8) {
        fail(SZ3HIP_ECAPACITY, "The buffer for compressed data is not large enough.");
        return 0;
    }
    uint64_t len = n;
    memcpy(dst, &len, 8);
    const size_t nf = std::max<size_t>(1, (n + FRAME - 1) / FRAME);
    const size_t fb = bound(std::min(n, FRAME));
    std::vector<std::vector<uint8_t>> out(nf);
    std::vector<size_t> sz(nf, 0);
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&]() {
        for (;;) {
            size_t f = next.fetch_add(1);
            if (f >= nf) break;
            size_t lo = f * FRAME, l = std::min(FRAME, n - lo);
            out[f].resize(fb);
            size_t r = compress(out[f].data(), fb, src + lo, l, 3);
            if (is_error(r)) bad = 1;
            sz[f] = r;
        }
    };
    unsigned nt = (unsigned)std::min<size_t>(nthreads(), nf);
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (bad) {
        fail(SZ3HIP_EZSTD, "ZSTD_compress failed");
        return 0;
    }
    size_t total = 8;
    for (size_t f = 0; f < nf; f++) total += sz[f];
    if (total > cap) {
        fail(SZ3HIP_ECAPACITY, "The buffer for compressed data is not large enough.");
        return 0;
    }
    uint8_t *p = dst + 8;
    for (size_t f = 0; f < nf; f++) {
        memcpy(p, out[f].data(), sz[f]);
        p += sz[f];
    }
    return total;
}
// inverse; frames are located with ZSTD_findFrameCompressedSize and decoded in parallel. returns bytes produced
static size_t decompress_frames(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {
    if (load()) return 0;
    if (n < 8) {
        fail(SZ3HIP_EFORMAT, "truncated lossless block");
        return 0;
    }
    uint64_t len;
    memcpy(&len, src, 8);
    if (len > cap) {
        fail(SZ3HIP_ECAPACITY, "lossless block larger than the destination");
        return 0;
    }
    const uint8_t *p = src + 8;
    size_t rem = n - 8;
    struct Fr { const uint8_t *p; size_t c, off, d; };
    std::vector<Fr> frames;
    bool split = frame_csize && frame_content;
    if (split) {
        size_t off = 0;
        while (rem > 0) {
            size_t c = frame_csize(p, rem);
            if (is_error(c)) { split = false; break; }
            unsigned long long d = frame_content(p, c);
            if (d == (unsigned long long)-1 || d == (unsigned long long)-2) { split = false; break; }
            frames.push_back({p, c, off, (size_t)d});
            off += (size_t)d;
            p += c;
            rem -= c;
        }
        if (split && off != len) split = false;
    }
    if (!split || frames.size() <= 1) {
        size_t r = decompress(dst, (size_t)len, src + 8, n - 8);
        if (is_error(r) || r != len) {
            fail(SZ3HIP_EZSTD, "ZSTD_decompress failed");
            return 0;
        }
        return r;
    }
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&]() {
        for (;;) {
            size_t f = next.fetch_add(1);
            if (f >= frames.size()) break;
            size_t r = decompress(dst + frames[f].off, frames[f].d, frames[f].p, frames[f].c);
            if (is_error(r) || r != frames[f].d) bad = 1;
        }
    };
    unsigned nt = (unsigned)std::min<size_t>(nthreads(), frames.size());
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (bad) {
        fail(SZ3HIP_EZSTD, "ZSTD_decompress failed");
        return 0;
    }
    return (size_t)len;
}
}  // namespace zs

// ------------------------------------------------------------------------------------------------------------
// Config (include/SZ3/utils/Config.hpp)
// ------------------------------------------------------------------------------------------------------------
extern "C" void sz3hip_config_init(sz3hip_config *c, int ndims, const uint64_t *dims) {
    memset(c, 0, sizeof(*c));
    int n = 0;
    for (int i = 0; i < ndims && n < 4; i++)  // setDims drops extents of 1 (Config.hpp:164-168)
        if (dims[i] > 1) c->dims[n++] = dims[i];
    if (n == 0) c->dims[n++] = 1;
    c->N = n;
    c->num = 1;
    for (int i = 0; i < n; i++) c->num *= c->dims[i];
    c->predDim = (uint8_t)n;
    c->blockSize = n == 1 ? 128 : (n == 2 ? 16 : 6);  // Config.hpp:175
    c->cmprAlgo = SZ3HIP_ALGO_INTERP_LORENZO;           // defaults Config.hpp:452-478
    c->errorBoundMode = SZ3HIP_EB_ABS;
    c->absErrorBound = 1e-3;
    c->quantbinCnt = 65536;
    c->dataType = SZ3HIP_FLOAT;
    c->lorenzo = 1;
    c->regression = 1;
    c->interpAlgo = 1;
    c->interpAnchorStride = -1;
    c->interpAlpha = 1.25;
    c->interpBeta = 2.0;
}

namespace {
struct Writer {
    unsigned char *p;
    template <class V> void put(V v) {
        memcpy(p, &v, sizeof(V));
        p += sizeof(V);
    }
};
struct Reader {
    const unsigned char *p;
    template <class V> V get() {
        V v;
        memcpy(&v, p, sizeof(V));
        p += sizeof(V);
        return v;
    }
};
}  // namespace

extern "C" size_t sz3hip_config_save(const sz3hip_config *c, unsigned char *out) {  // Config.hpp:312-354
    Writer w{out + 1};
    w.put<int8_t>((int8_t)c->N);
    uint64_t mx = 0;
    for (int i = 0; i < c->N; i++) mx = std::max(mx, c->dims[i]);
    uint8_t bw = 0;  // vector_bit_width, utils/ByteUtil.hpp:195-204
    for (; mx > 0; mx >>= 1) bw++;
    w.put<uint8_t>(bw);
    const size_t nbytes = ((size_t)bw * (size_t)c->N + 7) / 8;  // vector2bytes, ByteUtil.hpp:206-238 (LSB first)
    memset(w.p, 0, nbytes);
    for (int i = 0; i < c->N; i++)
        for (int j = 0; j < bw; j++)
            if ((c->dims[i] >> j) & 1) {
                size_t bit = (size_t)i * bw + j;
                w.p[bit >> 3] |= (unsigned char)(1u << (bit & 7));
            }
    w.p += nbytes;
    w.put<uint64_t>(c->num);
    w.put<uint8_t>(c->cmprAlgo);
    w.put<uint8_t>(c->errorBoundMode);
    switch (c->errorBoundMode) {
        case SZ3HIP_EB_ABS: w.put<double>(c->absErrorBound); break;
        case SZ3HIP_EB_REL: w.put<double>(c->relErrorBound); break;
        case SZ3HIP_EB_PSNR: w.put<double>(c->psnrErrorBound); break;
        case SZ3HIP_EB_L2NORM: w.put<double>(c->l2normErrorBound); break;
        case SZ3HIP_EB_ABS_AND_REL:
        case SZ3HIP_EB_ABS_OR_REL:
            w.put<double>(c->absErrorBound);
            w.put<double>(c->relErrorBound);
            break;
        default: break;
    }
    w.put<uint8_t>((uint8_t)((c->lorenzo & 1) << 7 | (c->lorenzo2 & 1) << 6 | (c->regression & 1) << 5 |
                             (c->regression2 & 1) << 4 | (c->openmp & 1) << 3));
    w.put<uint8_t>(c->dataType);
    w.put<int32_t>(c->quantbinCnt);
    w.put<int32_t>(c->blockSize);
    w.put<uint8_t>(c->predDim);
    out[0] = (unsigned char)(w.p - out);
    return (size_t)(w.p - out);
}

extern "C" size_t sz3hip_config_load(sz3hip_config *c, const unsigned char *in) {  // Config.hpp:361-413
    Reader r{in};
    const uint8_t conf_size = r.get<uint8_t>();
    const unsigned char *end = r.p + conf_size;
    uint64_t one = 1;
    sz3hip_config_init(c, 1, &one);
    c->N = r.get<int8_t>();
    if (c->N < 0 || c->N > 4) c->N = 0;
    const uint8_t bw = r.get<uint8_t>();
    const size_t nbytes = ((size_t)bw * (size_t)c->N + 7) / 8;
    for (int i = 0; i < c->N; i++) {  // bytes2vector, ByteUtil.hpp:240-264
        uint64_t v = 0;
        for (int j = 0; j < bw && j < 64; j++) {
            size_t bit = (size_t)i * bw + j;
            v |= (uint64_t)((r.p[bit >> 3] >> (bit & 7)) & 1) << j;
        }
        c->dims[i] = v;
    }
    r.p += nbytes;
    c->num = r.get<uint64_t>();
    c->cmprAlgo = r.get<uint8_t>();
    c->errorBoundMode = r.get<uint8_t>();
    switch (c->errorBoundMode) {
        case SZ3HIP_EB_ABS: c->absErrorBound = r.get<double>(); break;
        case SZ3HIP_EB_REL: c->relErrorBound = r.get<double>(); break;
        case SZ3HIP_EB_PSNR: c->psnrErrorBound = r.get<double>(); break;
        case SZ3HIP_EB_L2NORM: c->l2normErrorBound = r.get<double>(); break;
        case SZ3HIP_EB_ABS_AND_REL:
        case SZ3HIP_EB_ABS_OR_REL:
            c->absErrorBound = r.get<double>();
            c->relErrorBound = r.get<double>();
            break;
        default: break;
    }
    if (r.p < end) {
        uint8_t b = r.get<uint8_t>();
        c->lorenzo = (b >> 7) & 1;
        c->lorenzo2 = (b >> 6) & 1;
        c->regression = (b >> 5) & 1;
        c->regression2 = (b >> 4) & 1;
        c->openmp = (b >> 3) & 1;
    }
    if (r.p < end) c->dataType = r.get<uint8_t>();
    if (r.p < end) c->quantbinCnt = r.get<int32_t>();
    if (r.p < end) c->blockSize = r.get<int32_t>();
    if (r.p < end) c->predDim = r.get<uint8_t>();
    return (size_t)(r.p - in);
}

// ------------------------------------------------------------------------------------------------------------
// device context
// ------------------------------------------------------------------------------------------------------------
enum { ST_K1 = 0, ST_CODEBOOK, ST_ENCODE, ST_ASSEMBLE, ST_DEC_HUFF, ST_DEC_RECON, ST_TUNER, ST_K1_KERNEL, ST_COUNT };
static const char *const kStageNames[ST_COUNT] = {"lorenzo_quant_hist", "codebook", "encode", "assemble",
                                                  "huffman_decode",     "reconstruct", "tuner", "k1_kernel"};

struct sz3hip_ctx {
    int device;
    int dtype;
    uint64_t max_n, out_cap, cur_out_cap, max_chunks;
    uint64_t out_alloc;      // entries the four outlier arrays hold (>= out_cap; grown on demand)
    uint64_t force_out_cap;  // != 0: list capacity of the retry after an overflow
    // device buffers
    uint16_t *d_codes;
    uint64_t *d_hist;      // histogram in use (internal or caller-owned)
    uint64_t *d_hist_own;  // internal allocation
    uint64_t *d_counters;  // [0]=n_vout [1]=n_dout [2]=total_words [3]=decoder [4..6]=probe words [8..9]=code book's symbol range (inside d_hist_own's block)
    uint32_t *d_hist_partial;
    void *d_work;          // interpolation: the array being overwritten with reconstructed values (lazy)
    uint64_t *d_vout_idx, *d_dout_idx;
    void *d_vout_val, *d_dout_val;
    uint32_t *d_enc;
    uint8_t *d_lens;
    uint64_t *d_keys, *d_ifreq;
    uint16_t *d_syms, *d_pleaf, *d_pint, *d_depth, *d_aux2, *d_pint2;
    uint32_t *d_range;
    szk_cb_info *d_info;
    uint16_t *d_chunk_words;
    uint64_t *d_chunk_off;
    szk_state *d_state;
    szk_dec_tables *d_tables;
    void *d_segtot;
    double *d_minmax;
    szk_state *h_state;  // pinned
    szk_mode mode;       // of the pending / last compress
    double *h_minmax;    // pinned
    // pending compress
    szh_header proto;
    bool stage1_done, stage2_done;
    sz3hip_stats stats;
    // ALGO_INTERP_LORENZO tuner scratch (lazy)
    uint8_t *d_flags;
    size_t flags_cap;
    uint64_t *d_starts;
    size_t starts_cap;
    void *d_samples;
    size_t samples_cap;
    void *d_trial_work;  // scratch of the trial kernel's global-memory variant (blocks too large for LDS)
    size_t trial_work_cap;
    hipStream_t side;    // the working copy of the input is made here while the tuner runs on the caller's stream
    hipEvent_t ev_fork, ev_join;
    bool copy_ahead;     // d_work already holds this call's input (joined into the caller's stream)
    int hist_tail;       // with hist_big: codes beyond the large tier counted by windowed passes (from the previous call's count)
    int hist_big;        // interpolation histogram pass with the 16384-bin second tier (from the previous call's far count)
    int pack_wide;       // the packer's LDS table window: 8192 instead of 4096 entries (from the previous call's probe)
    int wide16;          // -1: not decided yet (f64 starts with the 16384-bin stage-1 window, f32 with 8192); else 0 / 1,
                         // adapted after every Lorenzo call from the width of the alphabet it saw
    uint64_t *d_trial;  // [8][4]: bits, symbols, unpredictables, delta outliers
    uint64_t *h_trial;  // pinned
    uint64_t *d_trial_hist;      // [SZK_MAX_TRIALS][65536] histograms of the trials of one group
    uint64_t *d_trial_counters;  // [SZK_MAX_TRIALS][8]
    szk_interp_pass *d_passes, *h_passes;  // [SZK_MAX_TRIALS][SZK_TRIAL_MAX_PASSES] pass schedules (h: pinned)
    uint32_t *d_np, *h_np;
    sz3hip_tuner_report tuner;
    // profiling
    bool profiling;
    hipEvent_t ev[ST_COUNT][2];
    bool ev_used[ST_COUNT];
};

#define SZ_COUNTER_BYTES 128
// histogram and counters of a call start at zero; the internal histogram and the counters share one block (one fill launch)
static hipError_t clear_hist_counters(sz3hip_ctx *c, hipStream_t s) {
    if (c->d_hist == c->d_hist_own) return hipMemsetAsync(c->d_hist, 0, SZH_HIST_BINS * 8 + SZ_COUNTER_BYTES, s);
    hipError_t e = hipMemsetAsync(c->d_hist, 0, SZH_HIST_BINS * 8, s);  // caller-owned histogram (multi-GPU all-reduce buffer)
    return e != hipSuccess ? e : hipMemsetAsync(c->d_counters, 0, SZ_COUNTER_BYTES, s);
}
static void ctx_free(sz3hip_ctx *c) {
    if (!c) return;
    void *bufs[] = {c->d_work, c->d_hist_partial, c->d_codes, c->d_hist_own, c->d_vout_idx, c->d_dout_idx, c->d_vout_val, c->d_dout_val,
                    c->d_enc, c->d_lens, c->d_keys, c->d_ifreq, c->d_syms, c->d_pleaf, c->d_pint, c->d_depth, c->d_aux2, c->d_pint2, c->d_range, c->d_info,
                    c->d_chunk_words, c->d_chunk_off, c->d_state, c->d_tables, c->d_segtot, c->d_minmax, c->d_samples, c->d_trial_work,
                    c->d_trial, c->d_passes, c->d_np};  // (d_trial_counters / d_trial_hist live inside d_trial's block)
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->d_flags) (void)hipHostFree(c->d_flags);
    if (c->d_starts) (void)hipHostFree(c->d_starts);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->h_minmax) (void)hipHostFree(c->h_minmax);
    if (c->h_trial) (void)hipHostFree(c->h_trial);
    if (c->h_passes) (void)hipHostFree(c->h_passes);
    if (c->h_np) (void)hipHostFree(c->h_np);
    for (int i = 0; i < ST_COUNT; i++)
        for (int j = 0; j < 2; j++)
            if (c->ev[i][j]) (void)hipEventDestroy(c->ev[i][j]);
    delete c;
}

extern "C" sz3hip_ctx *sz3hip_ctx_create(int device, uint64_t max_elems, int dataType) {
    if (dataType != SZ3HIP_FLOAT && dataType != SZ3HIP_DOUBLE) {
        fail(SZ3HIP_EUNSUPPORTED, "dataType %d not supported by the HIP path (float / double only)", dataType);
        return nullptr;
    }
    if (max_elems == 0) max_elems = 1;
    if (hipSetDevice(device) != hipSuccess) {
        fail(SZ3HIP_EHIP, "hipSetDevice(%d) failed — no usable HIP device; this library has no CPU path", device);
        return nullptr;
    }
    sz3hip_ctx *c = new sz3hip_ctx();
    memset(c, 0, sizeof(*c));
    c->wide16 = -1;
    c->device = device;
    c->dtype = dataType;
    c->max_n = max_elems;
    c->out_cap = std::max<uint64_t>(4096, max_elems / 32);
    c->out_alloc = c->out_cap;
    c->max_chunks = (max_elems + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS;
    const size_t tsz = dataType == SZ3HIP_FLOAT ? 4 : 8;
    bool ok = true;
    auto alloc = [&](void **p, size_t bytes) {
        if (!ok) return;
        if (hipMalloc(p, bytes ? bytes : 16) != hipSuccess) {
            ok = false;
            fail(SZ3HIP_EHIP, "hipMalloc of %zu bytes failed", bytes);
        }
    };
    alloc((void **)&c->d_codes, (max_elems + 64) * 2);
    alloc((void **)&c->d_hist_own, SZH_HIST_BINS * 8 + SZ_COUNTER_BYTES);  // the counters follow the histogram: one memset per call
    c->d_hist = c->d_hist_own;
    c->d_counters = c->d_hist_own ? c->d_hist_own + SZH_HIST_BINS : nullptr;
    alloc((void **)&c->d_hist_partial, (size_t)SZK_K1_GRID * 1024 * 4);
    alloc((void **)&c->d_vout_idx, c->out_cap * 8);
    alloc((void **)&c->d_dout_idx, c->out_cap * 8);
    alloc(&c->d_vout_val, c->out_cap * 8);
    alloc(&c->d_dout_val, c->out_cap * 8);
    alloc((void **)&c->d_enc, SZK_MAX_BOOKS * SZH_HIST_BINS * 4);
    alloc((void **)&c->d_lens, SZK_MAX_BOOKS * SZH_HIST_BINS);
    alloc((void **)&c->d_keys, SZK_MAX_BOOKS * SZH_HIST_BINS * 8);
    alloc((void **)&c->d_ifreq, SZK_MAX_BOOKS * SZH_HIST_BINS * 8);
    alloc((void **)&c->d_syms, SZK_MAX_BOOKS * SZH_HIST_BINS * 2);
    alloc((void **)&c->d_pleaf, SZK_MAX_BOOKS * SZH_HIST_BINS * 2);
    alloc((void **)&c->d_pint, SZK_MAX_BOOKS * SZH_HIST_BINS * 2);
    alloc((void **)&c->d_depth, SZK_MAX_BOOKS * SZH_HIST_BINS * 2);
    alloc((void **)&c->d_aux2, SZK_MAX_BOOKS * SZH_HIST_BINS * 2);
    alloc((void **)&c->d_pint2, SZK_MAX_BOOKS * SZH_HIST_BINS * 2);
    alloc((void **)&c->d_range, SZK_MAX_BOOKS * 16);
    alloc((void **)&c->d_info, SZK_MAX_BOOKS * sizeof(szk_cb_info));
    alloc((void **)&c->d_chunk_words, (c->max_chunks + 8) * 2);
    alloc((void **)&c->d_chunk_off, (c->max_chunks + 8) * 8);
    alloc((void **)&c->d_state, sizeof(szk_state));
    alloc((void **)&c->d_tables, sizeof(szk_dec_tables));
    alloc(&c->d_segtot, (4 * max_elems / 16384 + 65536) * 8);
    alloc((void **)&c->d_minmax, (2 * 1024 + 2) * 8);
    if (ok && hipHostMalloc((void **)&c->h_state, sizeof(szk_state)) != hipSuccess) ok = false;
    if (ok && hipHostMalloc((void **)&c->h_minmax, 16) != hipSuccess) ok = false;
    (void)tsz;
    if (!ok) {
        if (!g_err[0]) fail(SZ3HIP_EHIP, "device allocation failed");
        ctx_free(c);
        return nullptr;
    }
    return c;
}
extern "C" void sz3hip_ctx_destroy(sz3hip_ctx *ctx) {
    if (ctx) (void)hipSetDevice(ctx->device);
    ctx_free(ctx);
}

static size_t payload_bound_n(uint64_t n, uint64_t out_cap) {
    const uint64_t n_chunks = (n + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS;
    return (size_t)(sizeof(szh_header) + SZH_HIST_BINS + 16 + 2 * n_chunks + 16 + 2 * (out_cap * 16 + 16) +
                    4 * n_chunks * (SZH_CHUNK_SYMS * SZH_MAX_LEN / 32) + 64);
}
extern "C" size_t sz3hip_payload_bound(const sz3hip_ctx *ctx, uint64_t n) { return payload_bound_n(n, ctx->out_cap); }
// lists of up to n / 8 entries: beyond that the stream cannot beat the lossless fallback any more
static uint64_t out_cap_limit(uint64_t n) { return std::max<uint64_t>(1024, n / 8); }
extern "C" size_t sz3hip_payload_bound_max(const sz3hip_ctx *ctx, uint64_t n) {
    return payload_bound_n(n, std::max<uint64_t>(ctx->out_cap, out_cap_limit(n)));
}
extern "C" void *sz3hip_histogram_ptr(sz3hip_ctx *ctx) { return ctx->d_hist; }
extern "C" size_t sz3hip_histogram_len(const sz3hip_ctx *) { return SZH_HIST_BINS; }
extern "C" int sz3hip_ctx_set_histogram(sz3hip_ctx *ctx, void *d_hist) {
    ctx->d_hist = d_hist ? (uint64_t *)d_hist : ctx->d_hist_own;
    return 0;
}
extern "C" void sz3hip_set_profiling(sz3hip_ctx *ctx, int on) {
    ctx->profiling = on != 0;
    if (on)
        for (int i = 0; i < ST_COUNT; i++)
            for (int j = 0; j < 2; j++)
                if (!ctx->ev[i][j]) (void)hipEventCreate(&ctx->ev[i][j]);
}
static void prof_begin(sz3hip_ctx *c, int st, hipStream_t s) {
    if (c->profiling) {
        (void)hipEventRecord(c->ev[st][0], s);
        c->ev_used[st] = true;
    }
}
static void prof_end(sz3hip_ctx *c, int st, hipStream_t s) {
    if (c->profiling) (void)hipEventRecord(c->ev[st][1], s);
}
extern "C" int sz3hip_get_stage_times(sz3hip_ctx *ctx, const char **names, float *ms, int max) {
    int k = 0;
    if (!ctx->profiling) return 0;
    for (int i = 0; i < ST_COUNT && k < max; i++) {
        if (!ctx->ev_used[i]) continue;
        float t = 0;
        if (hipEventElapsedTime(&t, ctx->ev[i][0], ctx->ev[i][1]) != hipSuccess) {
            (void)hipGetLastError();  // (e.g. the kernel-level pair when another predictor kernel ran: never recorded)
            continue;
        }
        names[k] = kStageNames[i];
        ms[k] = t;
        k++;
    }
    return k;
}

extern "C" int sz3hip_minmax_device(sz3hip_ctx *ctx, const void *d_in, uint64_t n, double *mn, double *mx, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return fail(SZ3HIP_EINVAL, "empty array");
    int rc = szk_launch_minmax(ctx->dtype, d_in, n, ctx->d_minmax + 2, ctx->d_minmax, s);
    if (rc) return fail(SZ3HIP_EHIP, "minmax kernel launch failed (%d)", rc);
    HIPCHK(hipMemcpyAsync(ctx->h_minmax, ctx->d_minmax, 16, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *mn = ctx->h_minmax[0];
    *mx = ctx->h_minmax[1];
    return 0;
}

// ---- stage 1, interpolation predictor with explicit parameters (SZ_compress_Interp, api/impl/SZAlgoInterp.hpp:17-30) ----
static int interp_params_from(const sz3hip_config *conf, double eb, int radius, szk_interp_params &ip) {
    memset(&ip, 0, sizeof(ip));
    ip.N = conf->N;
    for (int i = 0; i < conf->N; i++) ip.dims[i] = conf->dims[i];
    ip.interp_id = conf->interpAlgo ? 1 : 0;
    ip.direction = conf->interpDirection;
    static const int def_anchor[4] = {4096, 128, 32, 16};  // SZAlgoInterp.hpp:20-24
    ip.anchor_stride = conf->interpAnchorStride < 0 ? (uint64_t)def_anchor[conf->N - 1] : (uint64_t)conf->interpAnchorStride;
    if (ip.anchor_stride & (ip.anchor_stride - 1)) return fail(SZ3HIP_EINVAL, "Anchor stride should be 0 or 2's exponentials");
    int nperm = 1;
    for (int i = 2; i <= conf->N; i++) nperm *= i;
    if (ip.direction < 0 || ip.direction >= nperm) return fail(SZ3HIP_EINVAL, "interpDirection out of range");
    ip.alpha = conf->interpAlpha;
    ip.beta = conf->interpBeta;
    ip.eb = eb;
    ip.radius = radius;
    return 0;
}
static void cb_params_from(sz3hip_ctx *ctx, szk_cb_params &cb, uint64_t out_cap) {
    cb.enc = ctx->d_enc;
    cb.lens = ctx->d_lens;
    cb.keys = ctx->d_keys;
    cb.syms = ctx->d_syms;
    cb.ifreq = ctx->d_ifreq;
    cb.pleaf = ctx->d_pleaf;
    cb.pint = ctx->d_pint;
    cb.depth = ctx->d_depth;
    cb.aux2 = ctx->d_aux2;
    cb.pint2 = ctx->d_pint2;
    cb.range = reinterpret_cast<uint32_t *>(ctx->d_counters + 8);  // (zeroed with the counters)
    cb.vout_idx = ctx->d_vout_idx;
    cb.dout_idx = ctx->d_dout_idx;
    cb.vout_val = ctx->d_vout_val;
    cb.dout_val = ctx->d_dout_val;
    cb.n_vout = ctx->d_counters + 0;
    cb.n_dout = ctx->d_counters + 1;
    cb.out_cap = out_cap;
    cb.t_is_32bit = cb.q_is_32bit = ctx->dtype == SZ3HIP_FLOAT;
    cb.info = ctx->d_info;
    cb.n_books = 1;
}

static int stage1_interp(sz3hip_ctx *ctx, const sz3hip_config *conf, const void *d_in, double eb, int radius, uint64_t num, hipStream_t s) {
    if (!ctx->d_work) HIPCHK(hipMalloc(&ctx->d_work, ctx->max_n * (ctx->dtype == SZ3HIP_FLOAT ? 4 : 8)));
    szk_interp_params ip;
    int rcp = interp_params_from(conf, eb, radius, ip);
    if (rcp) return rcp;
    ip.n_vout = ctx->d_counters + 0;
    ip.vout_idx = ctx->d_vout_idx;
    ip.vout_val = ctx->d_vout_val;
    ip.out_cap = ctx->cur_out_cap;
    ip.hist_big = (uint32_t)ctx->hist_big;
    ip.hist_tail = ctx->hist_tail > 0 || (szk_dbg_flags & 8192) ? 1u : 0u;
    if (szk_dbg_flags & 8192) ip.hist_big = 1;  // (test hook: large tier + tail passes whatever the history)
    ip.far_cnt = reinterpret_cast<uint32_t *>(ctx->d_counters + 6);  // (zeroed with the counters, fetched with the probe words)
    prof_begin(ctx, ST_K1, s);
    int rci = szk_launch_interp_compress(ctx->dtype, &ip, ctx->copy_ahead ? nullptr : d_in, ctx->d_work, ctx->d_codes, ctx->d_hist, s);
    ctx->copy_ahead = false;
    prof_end(ctx, ST_K1, s);
    if (rci) return fail(SZ3HIP_EHIP, "interpolation kernel launch failed (%d)", rci);
    memset(&ctx->mode, 0, sizeof(ctx->mode));
    ctx->mode.probe_big = reinterpret_cast<uint32_t *>(ctx->d_counters + 4);
    szh_header &hh = ctx->proto;
    memset(&hh, 0, sizeof(hh));
    hh.magic = SZH_MAGIC;
    hh.version = SZH_VERSION;
    hh.dtype = (uint8_t)ctx->dtype;
    hh.ndim = (uint8_t)conf->N;
    hh.qbytes = ctx->dtype == SZ3HIP_FLOAT ? 4 : 8;
    hh.predictor = 1;
    hh.radius = (uint32_t)radius;
    for (int i = 0; i < 4; i++) hh.dims[i] = 1;
    for (int i = 0; i < conf->N; i++) hh.dims[4 - conf->N + i] = conf->dims[i];
    hh.eb = eb;
    hh.n = num;
    hh.chunk_syms = SZH_CHUNK_SYMS;
    hh.n_chunks = (num + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS;
    hh.interp_alpha = ip.alpha;
    hh.interp_beta = ip.beta;
    hh.interp_id = (uint32_t)ip.interp_id;
    hh.interp_dir = (uint32_t)ip.direction;
    hh.anchor_stride = ip.anchor_stride;
    ctx->stage1_done = true;
    ctx->stage2_done = false;
    return 0;
}

// ---- stage 1, integer Lorenzo on the prequantised lattice ----
static int lorenzo_k1(sz3hip_ctx *ctx, int N, const uint64_t *dims, const void *d_in, double eb, int radius, uint64_t num,
                      uint64_t out_cap, bool allow_narrow, szk_k1_params &p, hipStream_t s) {
    memset(&p, 0, sizeof(p));
    for (int i = 0; i < 4; i++) p.d[i] = 1;
    for (int i = 0; i < N; i++) p.d[4 - N + i] = dims[i];
    p.lat = szk_make_lattice(eb);
    p.radius = (uint32_t)radius;
    p.out_cap = out_cap;  // lists larger than n/32 entries can never pay off: overflow => lossless fallback
    p.hist = ctx->d_hist;
    p.hist_partial = ctx->d_hist_partial;
    p.n_vout = ctx->d_counters + 0;
    p.n_dout = ctx->d_counters + 1;
    p.vout_idx = ctx->d_vout_idx;
    p.dout_idx = ctx->d_dout_idx;
    p.vout_val = ctx->d_vout_val;
    p.dout_val = ctx->d_dout_val;
    p.mode.probe_big = reinterpret_cast<uint32_t *>(ctx->d_counters + 4);  // zeroed with the counters
    p.mode.n_total = num;
    p.mode.n_samples = (num / SZK_PROBE_STRIDE) * 64 + std::min<uint64_t>(64, num % SZK_PROBE_STRIDE);
    p.mode.allow = allow_narrow && radius >= 128;
    p.mode.pack_wide = allow_narrow ? (uint32_t)ctx->pack_wide : 0u;
    p.wide16 = ctx->wide16 < 0 ? (ctx->dtype == SZ3HIP_DOUBLE ? 1u : 0u) : (uint32_t)ctx->wide16;
    p.prof_ev0 = p.prof_ev1 = nullptr;
    if (ctx->profiling && allow_narrow) {  // (the production call, not the tuner's trial): events around the kernel itself
        p.prof_ev0 = ctx->ev[ST_K1_KERNEL][0];
        p.prof_ev1 = ctx->ev[ST_K1_KERNEL][1];
        ctx->ev_used[ST_K1_KERNEL] = true;
    }
    return szk_launch_k1(ctx->dtype, N, d_in, ctx->d_codes, &p, s);
}
static int stage1_lorenzo(sz3hip_ctx *ctx, const sz3hip_config *conf, const void *d_in, double eb, int radius, uint64_t num, hipStream_t s) {
    szk_k1_params p;
    prof_begin(ctx, ST_K1, s);
    int rc = lorenzo_k1(ctx, conf->N, conf->dims, d_in, eb, radius, num, ctx->cur_out_cap, true, p, s);
    ctx->mode = p.mode;
    prof_end(ctx, ST_K1, s);
    if (rc) return fail(SZ3HIP_EHIP, "lorenzo_quant kernel launch failed (%d)", rc);
    szh_header &h = ctx->proto;
    memset(&h, 0, sizeof(h));
    h.magic = SZH_MAGIC;
    h.version = SZH_VERSION;
    h.dtype = (uint8_t)ctx->dtype;
    h.ndim = (uint8_t)conf->N;
    h.qbytes = ctx->dtype == SZ3HIP_FLOAT ? 4 : 8;
    h.radius = (uint32_t)radius;
    for (int i = 0; i < 4; i++) h.dims[i] = p.d[i];
    h.eb = eb;
    h.n = num;
    h.chunk_syms = SZH_CHUNK_SYMS;
    h.n_chunks = (num + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS;
    ctx->stage1_done = true;
    ctx->stage2_done = false;
    return 0;
}

// ---- ALGO_INTERP_LORENZO: the sampling auto-tuner (SZ_compress_Interp_lorenzo, api/impl/SZAlgoInterp.hpp:122-286) ----
// Same sampling geometry, trial list and decision rules as the reference. The six interpolation trials (and, in 1-D,
// the Lorenzo trial) run on the GPU over the batch of sampled blocks. What differs is the size estimate of a trial:
// the reference Huffman-codes and zstd-compresses every trial on the CPU; here a trial's size is priced on the device
// as   optimal-Huffman bits / 8  +  0.45 x serialised tree bytes  +  unpredictable values  +  80
// (0.45 = measured zstd gain on the reference's tree bytes; the bit stream itself is taken as incompressible).
// Decisions agree with the reference in most cases and can differ near its 2 % thresholds or at ratios > 30 where
// zstd matters; any decision yields a valid stream.
static int tuner_reserve(sz3hip_ctx *ctx, size_t flags, size_t starts, size_t samples) {
    if (ctx->flags_cap < flags) {
        // flags and block origins live in pinned host memory the kernels access directly: a few KB each way, and the
        // tuner saves two staged copies and one synchronisation
        if (ctx->d_flags) (void)hipHostFree(ctx->d_flags);
        ctx->d_flags = nullptr;
        HIPCHK(hipHostMalloc((void **)&ctx->d_flags, flags));
        ctx->flags_cap = flags;
    }
    if (ctx->starts_cap < starts) {
        if (ctx->d_starts) (void)hipHostFree(ctx->d_starts);
        ctx->d_starts = nullptr;
        HIPCHK(hipHostMalloc((void **)&ctx->d_starts, starts));
        ctx->starts_cap = starts;
    }
    if (ctx->samples_cap < samples) {
        if (ctx->d_samples) (void)hipFree(ctx->d_samples);
        ctx->d_samples = nullptr;
        HIPCHK(hipMalloc(&ctx->d_samples, samples));
        ctx->samples_cap = samples;
    }
    if (ctx->trial_work_cap < samples * SZK_MAX_TRIALS) {  // (d_work itself is being filled on the side stream)
        if (ctx->d_trial_work) (void)hipFree(ctx->d_trial_work);
        ctx->d_trial_work = nullptr;
        HIPCHK(hipMalloc(&ctx->d_trial_work, samples * SZK_MAX_TRIALS));
        ctx->trial_work_cap = samples * SZK_MAX_TRIALS;
    }
    if (!ctx->d_trial) {  // one block [results 256 B][counters 512 B][pad][histograms]: a group zeroes it with one memset
        HIPCHK(hipMalloc(&ctx->d_trial, 1024 + SZK_MAX_TRIALS * SZH_HIST_BINS * 8));
        ctx->d_trial_counters = ctx->d_trial + 32;
        ctx->d_trial_hist = ctx->d_trial + 128;
    }
    if (!ctx->h_trial) HIPCHK(hipHostMalloc((void **)&ctx->h_trial, 8 * 4 * 8));
    const size_t pbytes = SZK_MAX_TRIALS * SZK_TRIAL_MAX_PASSES * sizeof(szk_interp_pass);
    if (!ctx->d_passes) HIPCHK(hipMalloc(&ctx->d_passes, pbytes));
    if (!ctx->h_passes) HIPCHK(hipHostMalloc((void **)&ctx->h_passes, pbytes));
    if (!ctx->d_np) HIPCHK(hipMalloc(&ctx->d_np, 4 * SZK_MAX_TRIALS));
    if (!ctx->h_np) HIPCHK(hipHostMalloc((void **)&ctx->h_np, 4 * SZK_MAX_TRIALS));
    if (!ctx->d_work) HIPCHK(hipMalloc(&ctx->d_work, ctx->max_n * (ctx->dtype == SZ3HIP_FLOAT ? 4 : 8)));
    return 0;
}
// Priced size of one trial. r: entropy of the codes in 1/256 bit, symbols in use, unpredictables, delta outliers. The code
// stream is priced at its entropy (a Huffman code spends at most a few per cent more; no code book is built for a trial).
static double trial_bytes(const uint64_t *r, size_t tsz) {
    const double nc = r[1] ? 2.0 * (double)r[1] - 1.0 : 0.0;
    const double w = nc <= 256 ? 1 : (nc <= 65536 ? 2 : 4);
    const double tree = 13.0 + nc * (2 * w + 5);  // HuffmanEncoder::save: [i32][i32][i32][u8] L R C t (HuffmanEncoder.hpp:108-125)
    return std::ceil((double)r[0] / 2048.0) + 0.45 * tree + (double)r[2] * (double)tsz + (double)r[3] * 12.0 + 80.0;
}
// one group of up to SZK_MAX_TRIALS independent trials: one interpolation launch for all of them, one cost launch over their
// histograms; trial j's priced size lands in result slot slot0 + j
static int tuner_interp_group(sz3hip_ctx *ctx, const sz3hip_config *tcs, int ntr, double eb, int radius, uint32_t nb, int slot0,
                              hipStream_t s) {
    HIPCHK(hipMemsetAsync(ctx->d_trial, 0, 1024 + (size_t)ntr * SZH_HIST_BINS * 8, s));  // (earlier groups' results were fetched)
    szk_interp_params ips[SZK_MAX_TRIALS];
    for (int j = 0; j < ntr; j++) {
        int rc = interp_params_from(&tcs[j], eb, radius, ips[j]);
        if (rc) return rc;
        ips[j].n_vout = ctx->d_trial_counters + 8 * j;
        ips[j].vout_idx = ctx->d_vout_idx;
        ips[j].vout_val = ctx->d_vout_val;
        ips[j].out_cap = 0;  // count only
    }
    int rc = szk_launch_interp_trials(ctx->dtype, ips, (uint32_t)ntr, ctx->d_samples, ctx->d_trial_work, ctx->d_codes, nb, ctx->d_trial_hist,
                                      ctx->h_passes, ctx->d_passes, ctx->h_np, ctx->d_np, s);
    if (rc) return fail(SZ3HIP_EHIP, "tuner: interpolation trial launch failed (%d)", rc);
    rc = szk_launch_code_cost(ctx->d_trial_hist, ctx->d_trial_counters, ctx->d_trial + 4 * slot0, (uint32_t)ntr, tcs[0].num * nb, 1, s);
    if (rc) return fail(SZ3HIP_EHIP, "tuner: cost kernel launch failed (%d)", rc);
    return 0;
}
static int tuner_fetch(sz3hip_ctx *ctx, hipStream_t s) {
    HIPCHK(hipMemcpyAsync(ctx->h_trial, ctx->d_trial, 8 * 4 * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}
// fills `conf` like the reference does before its final compress call: cmprAlgo becomes ALGO_INTERP (interpAlgo,
// interpDirection, interpAlpha, interpBeta tuned) or ALGO_LORENZO_REG (1-D only)
static int tune_interp_lorenzo(sz3hip_ctx *ctx, sz3hip_config &conf, const void *d_in, double eb, int radius, hipStream_t s) {
    const int N = conf.N;
    const size_t tsz = ctx->dtype == SZ3HIP_FLOAT ? 4 : 8;
    sz3hip_tuner_report &rep = ctx->tuner;
    memset(&rep, 0, sizeof(rep));
    rep.use_interp = 1;
    static const int def_anchor[4] = {4096, 128, 32, 16};
    if (conf.interpAnchorStride < 0) conf.interpAnchorStride = def_anchor[N - 1];
    const double rate = 0.005;                               // SZAlgoInterp.hpp:133-135
    uint64_t sbs = (uint64_t)def_anchor[N - 1];              // sampleBlock_Sizes :136-138
    uint64_t shortest = conf.dims[0];
    for (int i = 0; i < N; i++) shortest = std::min<uint64_t>(shortest, conf.dims[i]);
    while (sbs >= shortest) sbs /= 2;                        // :144-147
    while (sbs >= 16 && (std::pow((double)(sbs + 1), N) / (double)conf.num) > 1.5 * rate) sbs /= 2;
    if (sbs < 8) sbs = 8;
    bool to_tune = std::pow((double)(sbs + 1), N) <= 0.05 * (double)conf.num;
    for (int i = 0; i < N; i++)
        if (conf.dims[i] < sbs) to_tune = false;
    rep.sample_block_size = sbs;
    auto fall_back = [&]() {
        conf.cmprAlgo = SZ3HIP_ALGO_INTERP;
        rep.interpAlgo = conf.interpAlgo;
        rep.interpDirection = conf.interpDirection;
        rep.interpAlpha = conf.interpAlpha;
        rep.interpBeta = conf.interpBeta;
        return 0;
    };
    if (!to_tune) return fall_back();
    const uint64_t per = (uint64_t)std::pow((double)(sbs + 1), N);
    // profiling_block: candidate origins whose strided samples are not constant within eb
    uint64_t cand = 1;
    for (int i = 0; i < N; i++) cand *= (conf.dims[i] - sbs + sbs - 1) / sbs;
    int rc = tuner_reserve(ctx, std::max<uint64_t>(cand, 1), 4096 * 32, 0);
    if (rc) return rc;
    uint64_t total = 0;
    rc = szk_launch_profile_blocks(ctx->dtype, d_in, N, conf.dims, sbs, sbs / 4, eb, ctx->d_flags, &total, s);
    if (rc) return fail(SZ3HIP_EHIP, "tuner: profiling launch failed (%d)", rc);
    if (total) HIPCHK(hipStreamSynchronize(s));
    const uint8_t *flags = ctx->d_flags;
    uint64_t cnt[4] = {1, 1, 1, 1};
    for (int i = 0; i < N; i++) cnt[i] = (conf.dims[i] - sbs + sbs - 1) / sbs;
    std::vector<uint64_t> filtered;  // linear candidate indices, lexicographic
    for (uint64_t t = 0; t < total; t++)
        if (flags[t]) filtered.push_back(t);
    const uint64_t nf = filtered.size();
    const bool profiling = (double)(nf * per) >= 0.5 * rate * (double)conf.num;  // :169
    // sampleBlocks (utils/Sample.hpp:221-289)
    uint64_t totalblock = 1;
    for (int i = 0; i < N; i++) totalblock *= (uint64_t)(int)((conf.dims[i] - 1) / sbs);
    std::vector<uint64_t> chosen;
    if (profiling) {
        uint64_t stride = (uint64_t)((double)nf / ((double)totalblock * rate));
        if (stride == 0) stride = 1;
        for (uint64_t i = 0; i < nf; i += stride) chosen.push_back(filtered[i]);
    } else {
        uint64_t stride = (uint64_t)(1.0 / rate);
        if (stride == 0) stride = 1;
        for (uint64_t idx = 0; idx < total; idx += stride) chosen.push_back(idx);
    }
    const uint64_t nb = chosen.size();
    rep.n_filtered = nf;
    rep.profiling = profiling;
    rep.n_blocks = nb;
    const uint64_t sampling_num = nb * per;
    if (sampling_num == 0 || (double)sampling_num >= (double)conf.num * 0.2) return fall_back();  // :176-179
    if (nb > 0x7FFFFFFFull) return fall_back();
    rc = tuner_reserve(ctx, 0, nb * 32, sampling_num * tsz);
    if (rc) return rc;
    uint64_t *starts = ctx->d_starts;  // (the previous call's gather finished before this call's profiling synchronised)
    memset(starts, 0, nb * 32);
    for (uint64_t b = 0; b < nb; b++) {
        uint64_t r = chosen[b];
        for (int j = N - 1; j >= 0; j--) {
            starts[b * 4 + j] = (r % cnt[j]) * sbs;
            r /= cnt[j];
        }
    }
    rc = szk_launch_gather_blocks(ctx->dtype, d_in, N, conf.dims, sbs + 1, ctx->d_starts, (uint32_t)nb, ctx->d_samples, s);
    if (rc) return fail(SZ3HIP_EHIP, "tuner: gather launch failed (%d)", rc);

    const double raw = (double)sampling_num * (double)tsz;
    double best_interp = 0, best_lorenzo = 0;
    sz3hip_config lorenzo_config = conf;
    conf.interpDirection = 0;  // :186-189
    conf.interpAlpha = 1.25;
    conf.interpBeta = 2.0;
    sz3hip_config tc = conf;
    tc.N = N;
    for (int i = 0; i < N; i++) tc.dims[i] = sbs + 1;
    tc.num = per;
    // linear and cubic, each with the identity and with the reversed dimension order, in one batch (the reference runs the
    // reversed-order trial only for the better formula: slots 2 / 3 hold that trial for linear / cubic)
    int fact = 1;
    for (int i = 2; i <= N; i++) fact *= i;
    // The (alpha, beta) trials of the most common outcome (cubic, identity order) ride along speculatively in slots 4..6 when
    // all trial workgroups of the launch still fit the chip at once (one workgroup per compute unit): the launch takes no
    // longer, and the second round trip (launch + fetch) is saved whenever the first group confirms that outcome.
    static const double alphas[3] = {1.0, 1.5, 2.0}, betas[3] = {1.0, 2.5, 3.0};
    const bool speculate = nb * 7 <= 256;
    {
        sz3hip_config g[7] = {tc, tc, tc, tc, tc, tc, tc};
        for (int k = 0; k < 4; k++) {
            g[k].interpAlgo = (uint8_t)(k & 1);
            g[k].interpDirection = (uint8_t)(k < 2 ? 0 : fact - 1);
        }
        for (int i = 0; i < 3; i++) {
            g[4 + i].interpAlgo = 1;
            g[4 + i].interpDirection = 0;
            g[4 + i].interpAlpha = alphas[i];
            g[4 + i].interpBeta = betas[i];
        }
        rc = tuner_interp_group(ctx, g, speculate ? 7 : 4, eb, radius, (uint32_t)nb, 0, s);
        if (rc) return rc;
    }
    rc = tuner_fetch(ctx, s);
    if (rc) return rc;
    double dir_bytes[2];
    for (int op = 0; op < 2; op++) {
        rep.est_bytes[op] = trial_bytes(ctx->h_trial + 4 * op, tsz);
        dir_bytes[op] = trial_bytes(ctx->h_trial + 4 * (2 + op), tsz);
        const double ratio = raw / rep.est_bytes[op];
        if (ratio > best_interp) {
            best_interp = ratio;
            conf.interpAlgo = (uint8_t)op;
        }
    }
    tc.interpAlgo = conf.interpAlgo;
    rep.est_bytes[2] = dir_bytes[conf.interpAlgo ? 1 : 0];
    if (raw / rep.est_bytes[2] > best_interp * 1.02) {
        best_interp = raw / rep.est_bytes[2];
        conf.interpDirection = (uint8_t)(fact - 1);
    }
    tc.interpDirection = conf.interpDirection;
    // (alpha, beta) pairs
    int ab_slot = 4;
    if (!(speculate && conf.interpAlgo == 1 && conf.interpDirection == 0)) {
        sz3hip_config g[3] = {tc, tc, tc};
        for (int i = 0; i < 3; i++) {
            g[i].interpAlpha = alphas[i];
            g[i].interpBeta = betas[i];
        }
        rc = tuner_interp_group(ctx, g, 3, eb, radius, (uint32_t)nb, 3, s);
        if (rc) return rc;
        rc = tuner_fetch(ctx, s);
        if (rc) return rc;
        ab_slot = 3;
    }
    for (int i = 0; i < 3; i++) {
        rep.est_bytes[3 + i] = trial_bytes(ctx->h_trial + 4 * (ab_slot + i), tsz);
        const double ratio = raw / rep.est_bytes[3 + i];
        if (ratio > best_interp * 1.02) {
            best_interp = ratio;
            conf.interpAlpha = alphas[i];
            conf.interpBeta = betas[i];
        }
    }
    if (N == 1 && best_interp < 50) {  // :232-247 — here: this library's own Lorenzo coder over the concatenated samples
        HIPCHK(clear_hist_counters(ctx, s));
        HIPCHK(hipMemsetAsync(ctx->d_trial + 24, 0, 32, s));
        szk_k1_params p;
        const uint64_t d1[1] = {sampling_num};
        rc = lorenzo_k1(ctx, 1, d1, ctx->d_samples, eb, radius, sampling_num, 0, false, p, s);
        if (rc) return fail(SZ3HIP_EHIP, "tuner: Lorenzo trial launch failed (%d)", rc);
        rc = szk_launch_code_cost(ctx->d_hist, ctx->d_counters, ctx->d_trial + 24, 1, sampling_num, 0, s);
        if (rc) return fail(SZ3HIP_EHIP, "tuner: cost kernel launch failed (%d)", rc);
        rc = tuner_fetch(ctx, s);
        if (rc) return rc;
        rep.est_bytes[6] = trial_bytes(ctx->h_trial + 24, tsz);
        best_lorenzo = raw / rep.est_bytes[6];
    }
    rep.ran = 1;
    const bool use_interp = !(best_lorenzo >= best_interp * 1.1 && best_lorenzo < 50 && best_interp < 50);  // :249-250
    rep.use_interp = use_interp;
    if (use_interp) {
        conf.cmprAlgo = SZ3HIP_ALGO_INTERP;
    } else {
        lorenzo_config.cmprAlgo = SZ3HIP_ALGO_LORENZO_REG;
        conf = lorenzo_config;
    }
    rep.interpAlgo = conf.interpAlgo;
    rep.interpDirection = conf.interpDirection;
    rep.interpAlpha = conf.interpAlpha;
    rep.interpBeta = conf.interpBeta;
    return 0;
}
extern "C" int sz3hip_get_tuner_report(sz3hip_ctx *ctx, sz3hip_tuner_report *rep) {
    *rep = ctx->tuner;
    return 0;
}

extern "C" int sz3hip_compress_stage1(sz3hip_ctx *ctx, const sz3hip_config *conf_in, const void *d_in, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    sz3hip_config conf_copy = *conf_in;
    sz3hip_config *conf = &conf_copy;
    if (conf->N < 1 || conf->N > 4) return fail(SZ3HIP_EINVAL, "Data dimension higher than 4 is not supported.");
    uint64_t num = 1;
    for (int i = 0; i < conf->N; i++) num *= conf->dims[i];
    if (num != conf->num || num == 0) return fail(SZ3HIP_EINVAL, "conf.num does not match conf.dims");
    if (num > ctx->max_n) return fail(SZ3HIP_EINVAL, "array of %llu elements exceeds the context capacity %llu",
                                      (unsigned long long)num, (unsigned long long)ctx->max_n);
    if (conf->errorBoundMode != SZ3HIP_EB_ABS) return fail(SZ3HIP_EINVAL, "stage1 needs an absolute error bound");
    const double eb = conf->absErrorBound;
    if (!(eb > 0) || !isfinite(eb)) return fail(SZ3HIP_EINVAL, "absErrorBound must be positive and finite");
    const int radius = conf->quantbinCnt / 2;  // api/impl/SZAlgoLorenzoReg.hpp:72
    if (radius < 2 || radius > 32768) return fail(SZ3HIP_EINVAL, "quantbinCnt must be in [4, 65536]");
    ctx->cur_out_cap = ctx->force_out_cap ? ctx->force_out_cap : std::min<uint64_t>(ctx->out_cap, std::max<uint64_t>(1024, num / 32));
    memset(&ctx->tuner, 0, sizeof(ctx->tuner));
    for (int i = 0; i < ST_COUNT; i++) ctx->ev_used[i] = false;  // stage times describe this call only
    ctx->copy_ahead = false;
    if (conf->cmprAlgo == SZ3HIP_ALGO_INTERP_LORENZO) {  // the reference's default: sampling auto-tuner, then one of the two paths
        // The tuner is a chain of small launches and host round trips (the chip is mostly idle), and its outcome is almost
        // always interpolation, which starts from a working copy of the input: make that copy meanwhile on a side stream.
        if (!ctx->d_work) HIPCHK(hipMalloc(&ctx->d_work, ctx->max_n * (ctx->dtype == SZ3HIP_FLOAT ? 4 : 8)));
        if (!ctx->side) {
            HIPCHK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(ctx->ev_fork, s));
        HIPCHK(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        HIPCHK(hipMemcpyAsync(ctx->d_work, d_in, num * (ctx->dtype == SZ3HIP_FLOAT ? 4 : 8), hipMemcpyDeviceToDevice, ctx->side));
        HIPCHK(hipEventRecord(ctx->ev_join, ctx->side));
        prof_begin(ctx, ST_TUNER, s);
        int rct = tune_interp_lorenzo(ctx, *conf, d_in, eb, radius, s);
        prof_end(ctx, ST_TUNER, s);
        hipError_t ej = hipStreamWaitEvent(s, ctx->ev_join, 0);  // whatever the outcome: the caller's stream owns d_in again
        if (ej != hipSuccess) {
            (void)hipStreamSynchronize(ctx->side);
            return fail(SZ3HIP_EHIP, "joining the side stream failed: %s", hipGetErrorString(ej));
        }
        if (rct) return rct;
        ctx->copy_ahead = true;
    }
    HIPCHK(clear_hist_counters(ctx, s));
    if (conf->cmprAlgo == SZ3HIP_ALGO_INTERP || conf->cmprAlgo == SZ3HIP_ALGO_HIP_INTERP)
        return stage1_interp(ctx, conf, d_in, eb, radius, num, s);
    return stage1_lorenzo(ctx, conf, d_in, eb, radius, num, s);
}

extern "C" int sz3hip_compress_stage2(sz3hip_ctx *ctx, void *d_payload, size_t cap, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->stage1_done) return fail(SZ3HIP_EINVAL, "stage2 called before stage1");
    if (ctx->stage2_done) return fail(SZ3HIP_EINVAL, "stage2 called twice for one stage1 (finish the call first)");
    const uint64_t n = ctx->proto.n;
    if (cap < payload_bound_n(n, std::max<uint64_t>(ctx->out_cap, ctx->cur_out_cap)))
        return fail(SZ3HIP_ECAPACITY, "The buffer for compressed data is not large enough.");
    szk_cb_params cb;
    cb_params_from(ctx, cb, ctx->cur_out_cap);
    prof_begin(ctx, ST_CODEBOOK, s);
    int rc = szk_launch_codebook(ctx->d_hist, &cb, s);
    if (rc) return fail(SZ3HIP_EHIP, "codebook kernel launch failed (%d)", rc);
    szk_layout_params lp;
    lp.proto = ctx->proto;
    lp.n_vout = ctx->d_counters + 0;
    lp.n_dout = ctx->d_counters + 1;
    lp.out_cap = ctx->cur_out_cap;
    lp.info = ctx->d_info;
    lp.state = ctx->d_state;
    prof_end(ctx, ST_CODEBOOK, s);
    prof_begin(ctx, ST_ENCODE, s);  // (the payload layout is computed inside the encoder's scan launch)
    rc = szk_launch_encode(ctx->d_codes, n, ctx->d_enc, ctx->d_info, (int)ctx->proto.radius, ctx->mode, ctx->d_chunk_words, ctx->d_chunk_off,
                           ctx->d_counters + 2, ctx->d_state, (uint8_t *)d_payload, &lp, s);
    prof_end(ctx, ST_ENCODE, s);
    if (rc) return fail(SZ3HIP_EHIP, "encode kernel launch failed (%d)", rc);
    szk_asm_params ap;
    ap.n_vout = ctx->d_counters + 0;
    ap.n_dout = ctx->d_counters + 1;
    ap.out_cap = ctx->cur_out_cap;
    ap.t_is_32bit = ap.q_is_32bit = ctx->dtype == SZ3HIP_FLOAT;
    ap.state = ctx->d_state;
    ap.payload = (uint8_t *)d_payload;
    ap.cap = cap;
    ap.total_words = ctx->d_counters + 2;
    ap.lens = ctx->d_lens;
    ap.chunk_words = ctx->d_chunk_words;
    ap.vout_idx = ctx->d_vout_idx;
    ap.dout_idx = ctx->d_dout_idx;
    ap.vout_val = ctx->d_vout_val;
    ap.dout_val = ctx->d_dout_val;
    prof_begin(ctx, ST_ASSEMBLE, s);
    rc = szk_launch_assemble(&ap, s);
    prof_end(ctx, ST_ASSEMBLE, s);
    if (rc) return fail(SZ3HIP_EHIP, "assemble kernel launch failed (%d)", rc);
    HIPCHK(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(szk_state), hipMemcpyDeviceToHost, s));
    // (h_state->probe = the probe counters: |delta| > 127, in [4096, 8192), in [2048, 4096); [4] = interpolation codes beyond +-4096)
    ctx->stage2_done = true;
    return 0;
}

extern "C" int sz3hip_compress_finish(sz3hip_ctx *ctx, size_t *payload_size, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->stage2_done) return fail(SZ3HIP_EINVAL, "finish called before stage2");
    HIPCHK(hipStreamSynchronize(s));
    ctx->stage1_done = ctx->stage2_done = false;
    const szk_state &st = *ctx->h_state;
    if (st.hdr.magic != SZH_MAGIC) return fail(SZ3HIP_EHIP, "device did not produce a payload header (kernel fault?)");
    ctx->stats.n = st.hdr.n;
    ctx->stats.n_value_outliers = st.hdr.n_vout;
    ctx->stats.n_delta_outliers = st.hdr.n_dout;
    ctx->stats.n_chunks = st.hdr.n_chunks;
    ctx->stats.bitstream_bytes = st.hdr.bitstream_words * 4;
    ctx->stats.payload_bytes = st.hdr.payload_bytes;
    ctx->stats.max_code_len = st.hdr.max_len;
    ctx->stats.n_symbols = 0;
    ctx->stats.narrow_codes = ctx->mode.allow && (uint64_t)st.probe[0] * 4096ull <= ctx->mode.n_samples;
    ctx->stats.reserved = (ctx->wide16 > 0 ? 1u : 0u) | (st.probe[1] << 1);  // (development: window used, far-delta count)
    if (st.hdr.predictor == 1)  // interpolation: second histogram tier of the next call (one workgroup per CU against three)
    {
        const bool was_big = ctx->hist_big > 0;
        // (> 1 %. Re-measured with the 1024-thread form: equal to the plain form at C3, but 0.19 ms slower at 1e-5, where 0.5 %
        // of the codes lie beyond the plain tier and 0.3 % are unpredictable - the threshold stays)
        ctx->hist_big = (uint64_t)st.probe[4] * 100ull > st.hdr.n ? 1 : 0;
        // beyond +-8192 every code is a global atomic (~1.2 G/s for the chip): from 2^18 of them on, three more passes over
        // the codes with LDS windows are cheaper (measured: 1.5 M of them cost 1.15 ms, the passes 0.2 ms)
        ctx->hist_tail = was_big && ctx->hist_big && st.probe[5] > (1u << 18) ? 1 : (was_big ? 0 : ctx->hist_tail);
    }
    if (st.hdr.predictor == 0 && ctx->mode.allow && ctx->mode.n_samples) {
        // stage-1 window of the next Lorenzo call: the large one (half the occupancy) when the probe saw more than 1/300 of
        // the deltas between the two windows (each costs a global atomic with the small one; measured break-even ~0.2 %:
        // 0.1 % -> 0.20 vs 0.28 ms in favour of the small window, 0.5 % -> 0.57 vs 0.36 ms in favour of the large one)
        ctx->wide16 = (uint64_t)st.probe[1] * 300ull > ctx->mode.n_samples ? 1 : 0;
        // the packers' table window likewise: doubled (3 instead of 5 workgroups per CU) when > 2 % of the symbols lie between
        ctx->pack_wide = (uint64_t)st.probe[2] * 50ull > ctx->mode.n_samples ? 1 : 0;
    }
    if (st.overflow)
        return fail(SZ3HIP_EOUTLIERS, "outlier capacity exceeded (%llu per list): data not compressible at this bound",
                    (unsigned long long)ctx->cur_out_cap);
    if (st.cap_exceeded) return fail(SZ3HIP_ECAPACITY, "The buffer for compressed data is not large enough.");
    if (payload_size) *payload_size = (size_t)st.hdr.payload_bytes;
    return 0;
}

extern "C" int sz3hip_compress_device(sz3hip_ctx *ctx, const sz3hip_config *conf, const void *d_in, void *d_payload,
                                      size_t cap, size_t *payload_size, void *stream) {
    int rc = sz3hip_compress_stage1(ctx, conf, d_in, stream);
    if (rc) return rc;
    rc = sz3hip_compress_stage2(ctx, d_payload, cap, stream);
    if (rc) return rc;
    rc = sz3hip_compress_finish(ctx, payload_size, stream);
    if (rc != SZ3HIP_EOUTLIERS) return rc;
    // More unpredictable values than the default lists hold (n / 32): a rough field at a tight bound, or a small
    // quantbinCnt. The reference keeps any number of them; here the lists grow to what this input needs (up to n / 8,
    // where the stream stops beating the lossless fallback) when the caller's buffer allows it
    // (sz3hip_payload_bound_max), and the call runs once more.
    const uint64_t n = ctx->proto.n;
    uint64_t cnt[2] = {0, 0};
    HIPCHK(hipMemcpy(cnt, ctx->d_counters, 16, hipMemcpyDeviceToHost));
    const uint64_t need = std::max(cnt[0], cnt[1]);
    if (need > out_cap_limit(n)) return rc;
    const uint64_t want = std::min<uint64_t>(out_cap_limit(n), need + need / 16 + 1024);
    if (cap < payload_bound_n(n, std::max<uint64_t>(ctx->out_cap, want))) return rc;
    if (want > ctx->out_alloc) {  // (finish() synchronised the stream: nothing uses the old arrays)
        void **arr[4] = {(void **)&ctx->d_vout_idx, (void **)&ctx->d_dout_idx, &ctx->d_vout_val, &ctx->d_dout_val};
        void *fresh[4] = {nullptr, nullptr, nullptr, nullptr};
        bool ok = true;
        for (int i = 0; i < 4 && ok; i++) ok = hipMalloc(&fresh[i], want * 8) == hipSuccess;
        if (!ok) {
            for (void *f : fresh)
                if (f) (void)hipFree(f);
            (void)hipGetLastError();
            return rc;  // no memory for larger lists: the caller falls back to lossless
        }
        for (int i = 0; i < 4; i++) {
            (void)hipFree(*arr[i]);
            *arr[i] = fresh[i];
        }
        ctx->out_alloc = want;
    }
    ctx->force_out_cap = want;
    rc = sz3hip_compress_stage1(ctx, conf, d_in, stream);
    ctx->force_out_cap = 0;
    if (rc) return rc;
    rc = sz3hip_compress_stage2(ctx, d_payload, cap, stream);
    if (rc) return rc;
    return sz3hip_compress_finish(ctx, payload_size, stream);
}

extern "C" int sz3hip_get_stats(sz3hip_ctx *ctx, sz3hip_stats *st) {
    *st = ctx->stats;
    return 0;
}
extern "C" int sz3hip_debug_copy_codes(sz3hip_ctx *ctx, uint16_t *host_codes, uint64_t n) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n > ctx->max_n) return fail(SZ3HIP_EINVAL, "n exceeds capacity");
    HIPCHK(hipDeviceSynchronize());
    uint32_t big = 0;
    HIPCHK(hipMemcpy(&big, ctx->d_counters + 4, 4, hipMemcpyDeviceToHost));
    const bool narrow = ctx->mode.allow && (uint64_t)big * 4096ull <= ctx->mode.n_samples;
    if (!narrow) {
        HIPCHK(hipMemcpy(host_codes, ctx->d_codes, n * 2, hipMemcpyDeviceToHost));
        return 0;
    }
    std::vector<uint8_t> b(n);  // one-byte codes: delta + 128, 0 = outlier -> symbols
    HIPCHK(hipMemcpy(b.data(), ctx->d_codes, n, hipMemcpyDeviceToHost));
    const uint32_t add = ctx->h_state->hdr.radius ? ctx->h_state->hdr.radius - 128u : ctx->proto.radius - 128u;
    for (uint64_t i = 0; i < n; i++) host_codes[i] = (uint16_t)(b[i] ? b[i] + add : 0u);
    return 0;
}

extern "C" void sz3hip_debug_force_generic(int on) { szk_force_generic = on; }
extern "C" void sz3hip_debug_flags(int flags) {
    szk_dbg_flags = flags;
    szk_interp_novec = (flags & 128) != 0;  // 128: interpolation without the 8-wide level-1 kernels
}
extern "C" int sz3hip_debug_codebook_info(sz3hip_ctx *ctx, uint64_t *out16) {
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    szk_cb_info info;
    HIPCHK(hipMemcpy(&info, ctx->d_info, sizeof(info), hipMemcpyDeviceToHost));
    out16[0] = info.n_symbols; out16[1] = info.max_len; out16[2] = info.sym_min; out16[3] = info.sym_count;
    for (int i = 0; i < 12; i++) out16[4 + i] = info.ts[i];
    return 0;
}

extern "C" int sz3hip_decompress_device(sz3hip_ctx *ctx, const void *d_payload, size_t payload_size, void *d_out,
                                        void *stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    if (payload_size < sizeof(szh_header)) return fail(SZ3HIP_EFORMAT, "payload shorter than its header");
    szh_header h;
    HIPCHK(hipMemcpyAsync(&ctx->h_state->hdr, d_payload, sizeof(szh_header), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    h = ctx->h_state->hdr;
    if (h.magic != SZH_MAGIC || h.version != SZH_VERSION) return fail(SZ3HIP_EFORMAT, "not an SZH1 payload");
    if (h.dtype != ctx->dtype) return fail(SZ3HIP_EINVAL, "payload data type does not match the context");
    if (h.n == 0 || h.n > ctx->max_n) return fail(SZ3HIP_EINVAL, "payload element count exceeds the context capacity");
    if (h.dims[0] * h.dims[1] * h.dims[2] * h.dims[3] != h.n || h.chunk_syms != SZH_CHUNK_SYMS ||
        h.n_chunks != (h.n + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS || h.sym_count > SZH_HIST_BINS ||
        h.sym_min + h.sym_count > SZH_HIST_BINS || h.max_len > SZH_MAX_LEN || h.radius < 2 || h.radius > 32768 ||
        h.qbytes != (h.dtype == 0 ? 4 : 8))
        return fail(SZ3HIP_EFORMAT, "corrupt SZH1 header");
    szh_offsets o;
    szk_host_offsets(&h, &o);
    if (o.end > payload_size || h.payload_bytes != o.end) return fail(SZ3HIP_EFORMAT, "truncated SZH1 payload");
    const uint8_t *pl = (const uint8_t *)d_payload;
    prof_begin(ctx, ST_DEC_HUFF, s);
    int rc = szk_launch_dec_tables(pl + o.lens, h.sym_min, h.sym_count, ctx->d_tables, s);
    if (rc) return fail(SZ3HIP_EHIP, "dec_tables kernel launch failed (%d)", rc);
    szk_dec_params dp;
    dp.n = h.n;
    dp.n_chunks = h.n_chunks;
    dp.bitstream_off = o.bitstream;
    dp.total_words = h.bitstream_words;
    dp.chunk_words = (const uint16_t *)(pl + o.chunkwords);
    dp.group_off = ctx->d_chunk_off;
    dp.tables = ctx->d_tables;
    dp.single_sym = h.sym_min;
    // Lorenzo stream, rows of at most one chunk, a sorted delta-outlier list: the decoder also does the x prefix sum
    const uint64_t row = h.dims[3];
    const bool fuse_x = h.predictor == 0 && h.n_dout <= 32768 && row >= 1 && row <= SZH_CHUNK_SYMS && !(szk_dbg_flags & 512);  // (lists that long are sorted)
    dp.scan_row = fuse_x ? (uint32_t)row : 0u;
    dp.radius = h.radius;
    dp.q_bytes = h.qbytes;
    dp.reserved = 0;
    dp.q_out = d_out;
    dp.dout_idx = reinterpret_cast<const uint64_t *>(pl + o.dout_idx);
    dp.dout_val = pl + o.dout_val;
    dp.n_dout = h.n_dout;
    dp.carry = fuse_x && (SZH_CHUNK_SYMS % row) != 0 ? (void *)ctx->d_codes : nullptr;  // (the code array is idle in this mode)
    rc = szk_launch_decode(pl, &dp, ctx->d_codes, ctx->d_chunk_off, ctx->d_counters + 3, s);
    prof_end(ctx, ST_DEC_HUFF, s);
    if (rc) return fail(SZ3HIP_EHIP, "decode kernel launch failed (%d)", rc);
    prof_begin(ctx, ST_DEC_RECON, s);
    if (h.predictor == 1) {
        szk_interp_params ip;
        memset(&ip, 0, sizeof(ip));
        ip.N = h.ndim;
        if (ip.N < 1 || ip.N > 4) return fail(SZ3HIP_EFORMAT, "corrupt SZH1 header");
        for (int i = 0; i < ip.N; i++) ip.dims[i] = h.dims[4 - ip.N + i];
        ip.interp_id = (int)h.interp_id;
        ip.direction = (int)h.interp_dir;
        ip.anchor_stride = h.anchor_stride;
        ip.alpha = h.interp_alpha;
        ip.beta = h.interp_beta;
        ip.eb = h.eb;
        ip.radius = (int)h.radius;
        rc = szk_launch_interp_decompress(ctx->dtype, &ip, pl, o.vout_idx, o.vout_val, h.n_vout, ctx->d_codes, d_out, s);
    } else {
        rc = szk_launch_reconstruct(fuse_x ? 1 : 0, pl, &h, &o, ctx->d_codes, d_out, ctx->d_segtot, s);
    }
    prof_end(ctx, ST_DEC_RECON, s);
    if (rc) return fail(SZ3HIP_EHIP, "reconstruct kernel launch failed (%d)", rc);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// host-buffer API: SZ_compress<T> / SZ_decompress<T> equivalents
// ------------------------------------------------------------------------------------------------------------
static const uint32_t kMagic = 0xF342F310u;                          // include/SZ3/version.hpp.in:10
static const uint32_t kDataVer = (3u << 24) | (3u << 16) | (2u << 8);  // SZ3_DATA_VERSION 3.3.2 (CMakeLists.txt:7)

static inline bool dtype_ok(int dt) { return dt == SZ3HIP_FLOAT || dt == SZ3HIP_DOUBLE || dt == SZ3HIP_INT32 || dt == SZ3HIP_INT64; }
static inline bool dtype_is_int(int dt) { return dt == SZ3HIP_INT32 || dt == SZ3HIP_INT64; }
static inline size_t dtype_size(int dt) { return (dt == SZ3HIP_FLOAT || dt == SZ3HIP_INT32) ? 4 : 8; }
static inline int dtype_compute(int dt) { return dtype_is_int(dt) ? SZ3HIP_DOUBLE : dt; }  // integers ride the f64 pipeline

extern "C" size_t sz3hip_compress_bound(const sz3hip_config *c, int dataType) {  // api/impl/SZImpl.hpp:34-44
    if (zs::load()) return 0;
    unsigned char tmp[160];
    const size_t es = dtype_size(dataType);
    return 4096 + sz3hip_config_save(c, tmp) + zs::bound_frames((size_t)c->num * es);
}

namespace {
// one cached context per (device, dtype); grown on demand. The host API is serialised per process.
std::mutex g_ctx_mu;
sz3hip_ctx *g_ctx[2];
void *g_dev_in[2], *g_dev_payload[2];
size_t g_dev_in_bytes[2], g_dev_payload_bytes[2];

int host_device() {
    const char *e = getenv("SZ3HIP_DEVICE");
    return e ? atoi(e) : 0;
}
sz3hip_ctx *get_ctx(int dtype, uint64_t n) {
    sz3hip_ctx *&c = g_ctx[dtype];
    if (c && c->max_n >= n) return c;
    if (c) sz3hip_ctx_destroy(c);
    c = sz3hip_ctx_create(host_device(), n, dtype);
    return c;
}
// pinned host staging for the payload (the device <-> host hop of the host API): DMA at link speed, no page faults
void *g_pin;
size_t g_pin_bytes;
int ensure_pin(size_t want) {
    if (g_pin_bytes >= want) return 0;
    if (g_pin) (void)hipHostFree(g_pin);
    g_pin = nullptr;
    g_pin_bytes = 0;
    want += want / 4;  // (payload sizes vary from call to call)
    HIPCHK(hipHostMalloc(&g_pin, want));
    g_pin_bytes = want;
    return 0;
}
int ensure_dev(void **p, size_t *have, size_t want) {
    if (*have >= want) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    HIPCHK(hipMalloc(p, want));
    *have = want;
    return 0;
}
}  // namespace

// utils/Statistic.hpp:32-56 with the range taken from the device min/max kernel
static int cal_abs_eb(sz3hip_config &conf, sz3hip_ctx *ctx, const void *d_in) {
    if (conf.errorBoundMode == SZ3HIP_EB_ABS) return 0;
    double range = 0;
    if (conf.errorBoundMode != SZ3HIP_EB_L2NORM) {
        double mn, mx;
        int rc = sz3hip_minmax_device(ctx, d_in, conf.num, &mn, &mx, nullptr);
        if (rc) return rc;
        // data_range computes max - min in T (Statistic.hpp:12-21)
        range = ctx->dtype == SZ3HIP_FLOAT ? (double)((float)mx - (float)mn) : mx - mn;
    }
    switch (conf.errorBoundMode) {
        case SZ3HIP_EB_REL: conf.absErrorBound = conf.relErrorBound * range; break;
        case SZ3HIP_EB_PSNR: {  // computeABSErrBoundFromPSNR, Statistic.hpp:25-30, threshold 0.99
            double v1 = conf.psnrErrorBound + 10 * log10(1 - 2.0 / 3.0 * 0.99);
            conf.absErrorBound = range * pow(10, v1 / (-20));
            break;
        }
        case SZ3HIP_EB_L2NORM: conf.absErrorBound = sqrt(3.0 / (double)conf.num) * conf.l2normErrorBound; break;
        case SZ3HIP_EB_ABS_AND_REL: conf.absErrorBound = std::min(conf.absErrorBound, conf.relErrorBound * range); break;
        case SZ3HIP_EB_ABS_OR_REL: conf.absErrorBound = std::max(conf.absErrorBound, conf.relErrorBound * range); break;
        default: return fail(SZ3HIP_EINVAL, "Error bound mode not supported");
    }
    conf.errorBoundMode = SZ3HIP_EB_ABS;
    return 0;
}

// SZ3HIP_TIMING=1: wall-clock breakdown of the host API on stderr (development aid)
struct HostTimer {
    bool on;
    std::chrono::steady_clock::time_point t0;
    HostTimer() : on(getenv("SZ3HIP_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void lap(const char *what) {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[sz3hip] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
extern "C" size_t sz3hip_compress(const sz3hip_config *config, int dataType, const void *data, char *cmpData,
                                  size_t cmpCap) {
    HostTimer tm;
    if (!dtype_ok(dataType)) {
        fail(SZ3HIP_EUNSUPPORTED, "dataType %d not supported by the HIP path (float, double, int32, int64)", dataType);
        return 0;
    }
    const bool is_int = dtype_is_int(dataType);
    const int cdt = dtype_compute(dataType);  // the type the kernels compute in
    sz3hip_config conf = *config;  // sz.hpp:45
    if (conf.N < 1 || conf.N > 4) {
        fail(SZ3HIP_EINVAL, "Data dimension higher than 4 is not supported.");
        return 0;
    }
    if (zs::load()) return 0;
    if (cmpCap < sz3hip_compress_bound(&conf, dataType)) {  // sz.hpp:47-49
        fail(SZ3HIP_ECAPACITY, "The buffer for compressed data is not large enough.");
        return 0;
    }
    const size_t es = dtype_size(dataType);
    const size_t raw_bytes = (size_t)conf.num * es;
    unsigned char *out = reinterpret_cast<unsigned char *>(cmpData);
    Writer w{out};
    w.put<uint32_t>(kMagic);
    w.put<uint32_t>(kDataVer);
    unsigned char *size_pos = w.p;
    w.p += 8;
    unsigned char tmp[160];
    const size_t payload_cap = cmpCap - 16 - 2 * sz3hip_config_save(&conf, tmp);
    size_t payload_size = 0;

    std::lock_guard<std::mutex> lock(g_ctx_mu);
    bool lossless = conf.cmprAlgo == SZ3HIP_ALGO_LOSSLESS;
    if (!lossless) {
        sz3hip_ctx *ctx = get_ctx(cdt, conf.num);
        if (!ctx) return 0;
        if (ensure_dev(&g_dev_in[cdt], &g_dev_in_bytes[cdt], (size_t)conf.num * (cdt == SZ3HIP_FLOAT ? 4 : 8))) return 0;
        const size_t pb = sz3hip_payload_bound(ctx, conf.num);
        if (ensure_dev(&g_dev_payload[cdt], &g_dev_payload_bytes[cdt], pb)) return 0;
        tm.lap("setup");
        if (!is_int) {
            if (hipMemcpy(g_dev_in[cdt], data, raw_bytes, hipMemcpyHostToDevice) != hipSuccess) {
                fail(SZ3HIP_EHIP, "host->device copy failed");
                return 0;
            }
            tm.lap("host->device");
        } else {
            // integers: staged in the (still unused) payload buffer, widened to f64 on the device
            if (pb < raw_bytes + 16 && ensure_dev(&g_dev_payload[cdt], &g_dev_payload_bytes[cdt], raw_bytes + 16)) return 0;
            if (hipMemcpy(g_dev_payload[cdt], data, raw_bytes, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemsetAsync(ctx->d_counters + 5, 0, 8, nullptr) != hipSuccess) {
                fail(SZ3HIP_EHIP, "host->device copy failed");
                return 0;
            }
            if (szk_launch_int_to_f64(dataType == SZ3HIP_INT64, g_dev_payload[cdt], conf.num, (double *)g_dev_in[cdt],
                                      reinterpret_cast<uint32_t *>(ctx->d_counters + 5), nullptr)) {
                fail(SZ3HIP_EHIP, "integer widening kernel failed");
                return 0;
            }
            uint32_t big = 0;
            if (hipMemcpy(&big, ctx->d_counters + 5, 4, hipMemcpyDeviceToHost) != hipSuccess) {
                fail(SZ3HIP_EHIP, "device->host copy failed");
                return 0;
            }
            if (big) lossless = true;  // |x| > 2^53 is not exact in f64: keep such arrays lossless
        }
        if (!lossless && cal_abs_eb(conf, ctx, g_dev_in[cdt])) return 0;
        if (is_int) {
            // |x - x^| <= eb between integers means <= floor(eb); the lattice 2*floor(eb) keeps every reconstruction integral
            conf.absErrorBound = std::floor(conf.absErrorBound);
            conf.errorBoundMode = SZ3HIP_EB_ABS;
        }
        if (conf.absErrorBound == 0) lossless = true;  // SZDispatcher.hpp:19-21
        if (!lossless) {
            // ALGO_LORENZO_REG / NOPRED -> HIP Lorenzo stream (16); ALGO_INTERP / ALGO_INTERP_LORENZO -> HIP interpolation (17)
            size_t dsize = 0;
            int rc = sz3hip_compress_device(ctx, &conf, g_dev_in[cdt], g_dev_payload[cdt], g_dev_payload_bytes[cdt], &dsize, nullptr);
            tm.lap("device compress");
            if (rc == SZ3HIP_EOUTLIERS && g_dev_payload_bytes[cdt] < sz3hip_payload_bound_max(ctx, conf.num)) {
                // room for the largest lists, then once more (the device call grows them to what the input needs)
                if (ensure_dev(&g_dev_payload[cdt], &g_dev_payload_bytes[cdt], sz3hip_payload_bound_max(ctx, conf.num))) return 0;
                rc = sz3hip_compress_device(ctx, &conf, g_dev_in[cdt], g_dev_payload[cdt], g_dev_payload_bytes[cdt], &dsize, nullptr);
            }
            if (rc == SZ3HIP_EOUTLIERS) {
                lossless = true;  // same policy as the reference's length_error fallback, SZDispatcher.hpp:44-59
            } else if (rc) {
                return 0;
            } else if (dsize + 64 >= raw_bytes) {
                lossless = true;  // the GPU stream would not even beat the raw array (tiny or incompressible input)
            } else {
                if (ensure_pin(dsize)) return 0;
                if (hipMemcpy(g_pin, g_dev_payload[cdt], dsize, hipMemcpyDeviceToHost) != hipSuccess) {
                    fail(SZ3HIP_EHIP, "device->host copy failed");
                    return 0;
                }
                tm.lap("device->host");
                payload_size = zs::compress_frames((const uint8_t *)g_pin, dsize, w.p, payload_cap);
                if (!payload_size) return 0;
                tm.lap("zstd");
                conf.cmprAlgo = ctx->h_state->hdr.predictor == 1 ? SZ3HIP_ALGO_HIP_INTERP : SZ3HIP_ALGO_HIP_LORENZO;
                if ((double)raw_bytes / (double)payload_size < 3) {  // SZDispatcher.hpp:62-74
                    std::vector<uint8_t> z(zs::bound_frames(raw_bytes) + 8);
                    size_t zsz = zs::compress_frames((const uint8_t *)data, raw_bytes, z.data(), z.size());
                    if (zsz && zsz < payload_size && zsz <= payload_cap) {
                        memcpy(w.p, z.data(), zsz);
                        payload_size = zsz;
                        conf.cmprAlgo = SZ3HIP_ALGO_LOSSLESS;
                    }
                }
            }
        }
    }
    if (lossless) {
        conf.cmprAlgo = SZ3HIP_ALGO_LOSSLESS;
        payload_size = zs::compress_frames((const uint8_t *)data, raw_bytes, w.p, payload_cap);
        if (!payload_size) return 0;
    }
    uint64_t ps = payload_size;
    memcpy(size_pos, &ps, 8);
    w.p += payload_size;
    conf.openmp = 0;
    conf.dataType = (uint8_t)dataType;  // lets the decoder refuse a request for another element type
    w.p += sz3hip_config_save(&conf, w.p);
    return (size_t)(w.p - out);
}

extern "C" int sz3hip_peek_config(sz3hip_config *conf, const char *cmpData, size_t cmpSize) {
    if (cmpSize < 16 + 8) return fail(SZ3HIP_EFORMAT, "compressed buffer too small");
    Reader r{reinterpret_cast<const unsigned char *>(cmpData)};
    if (r.get<uint32_t>() != kMagic)  // sz.hpp:122-125
        return fail(SZ3HIP_EFORMAT, "magic number mismatch, the input data is not compressed by SZ3");
    const uint32_t ver = r.get<uint32_t>();
    if ((ver >> 8) != (kDataVer >> 8))  // sz.hpp:127-135 compares major.minor.patch
        return fail(SZ3HIP_EFORMAT, "Please use SZ3 v%u.%u.%u to decompress the data", ver >> 24, (ver >> 16) & 255,
                    (ver >> 8) & 255);
    const uint64_t payload = r.get<uint64_t>();
    if (payload > cmpSize - 16) return fail(SZ3HIP_EFORMAT, "payload size exceeds the buffer");
    sz3hip_config_load(conf, r.p + payload);
    return 0;
}

extern "C" int sz3hip_decompress(sz3hip_config *conf, int dataType, const char *cmpData, size_t cmpSize, void *decData) {
    if (!dtype_ok(dataType))
        return fail(SZ3HIP_EUNSUPPORTED, "dataType %d not supported by the HIP path (float, double, int32, int64)", dataType);
    int rc = sz3hip_peek_config(conf, cmpData, cmpSize);
    if (rc) return rc;
    const bool is_int = dtype_is_int(dataType);
    const int cdt = dtype_compute(dataType);
    if (dtype_is_int(conf->dataType) != is_int)
        return fail(SZ3HIP_EINVAL, "the stream holds %s data but %s output was requested", dtype_is_int(conf->dataType) ? "integer" : "floating-point",
                    is_int ? "integer" : "floating-point");
    if (zs::load()) return SZ3HIP_EZSTD;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(cmpData) + 8;
    uint64_t payload;
    memcpy(&payload, p, 8);
    p += 8;
    const size_t es = dtype_size(dataType);
    const size_t raw_bytes = (size_t)conf->num * es;
    if (conf->cmprAlgo == SZ3HIP_ALGO_LOSSLESS) {  // SZDispatcher.hpp:81-88
        uint64_t len = 0;
        if (payload >= 8) memcpy(&len, p, 8);
        if (len != raw_bytes)
            return fail(SZ3HIP_EFORMAT, "Decompressed data size does not match the original data size");
        return zs::decompress_frames(p, payload, (uint8_t *)decData, raw_bytes) == raw_bytes ? 0 : SZ3HIP_EZSTD;
    }
    if (conf->cmprAlgo != SZ3HIP_ALGO_HIP_LORENZO && conf->cmprAlgo != SZ3HIP_ALGO_HIP_INTERP)
        return fail(SZ3HIP_EUNSUPPORTED,
                    "stream uses cmprAlgo %d of the CPU reference; this library decodes only its own GPU streams (ids %d, %d) "
                    "and ALGO_LOSSLESS",
                    conf->cmprAlgo, SZ3HIP_ALGO_HIP_LORENZO, SZ3HIP_ALGO_HIP_INTERP);
    if (payload < 8) return fail(SZ3HIP_EFORMAT, "truncated payload");
    uint64_t raw_len;
    memcpy(&raw_len, p, 8);
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    if ((rc = ensure_pin(raw_len))) return rc;
    if (zs::decompress_frames(p, payload, (uint8_t *)g_pin, raw_len) != raw_len) return SZ3HIP_EZSTD;
    sz3hip_ctx *ctx = get_ctx(cdt, conf->num);
    if (!ctx) return SZ3HIP_EHIP;
    const size_t cbytes = (size_t)conf->num * (cdt == SZ3HIP_FLOAT ? 4 : 8);
    if ((rc = ensure_dev(&g_dev_in[cdt], &g_dev_in_bytes[cdt], cbytes))) return rc;
    if ((rc = ensure_dev(&g_dev_payload[cdt], &g_dev_payload_bytes[cdt], std::max<size_t>(raw_len + 64, is_int ? raw_bytes : 0)))) return rc;
    HIPCHK(hipMemcpy(g_dev_payload[cdt], g_pin, raw_len, hipMemcpyHostToDevice));
    rc = sz3hip_decompress_device(ctx, g_dev_payload[cdt], raw_len, g_dev_in[cdt], nullptr);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    if (ctx->h_state->hdr.n != conf->num) return fail(SZ3HIP_EFORMAT, "payload element count does not match the trailer");
    if (ctx->h_state->hdr.dtype != (uint8_t)cdt) return fail(SZ3HIP_EINVAL, "the stream's element type does not match the requested one");
    if (!is_int) {
        HIPCHK(hipMemcpy(decData, g_dev_in[cdt], raw_bytes, hipMemcpyDeviceToHost));  // (the runtime pins large pageable buffers
                                                                                       // itself: a hand-made pinned pipeline was slower)
    } else {
        rc = szk_launch_f64_to_int(dataType == SZ3HIP_INT64, (const double *)g_dev_in[cdt], conf->num, g_dev_payload[cdt], nullptr);
        if (rc) return fail(SZ3HIP_EHIP, "integer narrowing kernel failed");
        HIPCHK(hipMemcpy(decData, g_dev_payload[cdt], raw_bytes, hipMemcpyDeviceToHost));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// the reference's C ABI (tools/sz3c/include/sz3c.h:52-59, tools/sz3c/src/sz3c.cpp:11-94)
// ------------------------------------------------------------------------------------------------------------
extern "C" unsigned char *SZ_compress_args(int dataType, void *data, size_t *outSize, int errBoundMode,
                                           double absErrBound, double relBoundRatio, double pwrBoundRatio, size_t r5,
                                           size_t r4, size_t r3, size_t r2, size_t r1) {
    (void)pwrBoundRatio;  // sz3c.cpp:29 ignores it too
    uint64_t d[4];
    int nd;
    if (r2 == 0) { nd = 1; d[0] = r1; }
    else if (r3 == 0) { nd = 2; d[0] = r2; d[1] = r1; }
    else if (r4 == 0) { nd = 3; d[0] = r3; d[1] = r2; d[2] = r1; }
    else if (r5 == 0) { nd = 4; d[0] = r4; d[1] = r3; d[2] = r2; d[3] = r1; }
    else { nd = 4; d[0] = r5 * r4; d[1] = r3; d[2] = r2; d[3] = r1; }  // sz3c.cpp:24
    sz3hip_config conf;
    sz3hip_config_init(&conf, nd, d);
    conf.absErrorBound = absErrBound;
    conf.relErrorBound = relBoundRatio;
    if (errBoundMode == ABS) conf.errorBoundMode = SZ3HIP_EB_ABS;
    else if (errBoundMode == REL) conf.errorBoundMode = SZ3HIP_EB_REL;
    else if (errBoundMode == ABS_AND_REL) conf.errorBoundMode = SZ3HIP_EB_ABS_AND_REL;
    else if (errBoundMode == ABS_OR_REL) conf.errorBoundMode = SZ3HIP_EB_ABS_OR_REL;
    else {
        printf("errBoundMode %d not support\n ", errBoundMode);  // sz3c.cpp:39-40
        exit(0);
    }
    if (dataType != SZ_FLOAT && dataType != SZ_DOUBLE) {
        printf("dataType %d not support\n", dataType);  // sz3c.cpp:51-52
        exit(0);
    }
    const size_t cap = sz3hip_compress_bound(&conf, dataType);
    unsigned char *buf = static_cast<unsigned char *>(malloc(cap));  // C memory, released by free_buf (sz3c.cpp:56-58)
    if (!buf) return nullptr;
    const size_t n = sz3hip_compress(&conf, dataType, data, reinterpret_cast<char *>(buf), cap);
    if (n == 0) {
        fprintf(stderr, "SZ_compress_args: %s\n", sz3hip_last_error());
        free(buf);
        *outSize = 0;
        return nullptr;
    }
    *outSize = n;
    unsigned char *shrunk = static_cast<unsigned char *>(realloc(buf, n));
    return shrunk ? shrunk : buf;
}

extern "C" void *SZ_decompress(int dataType, unsigned char *bytes, size_t byteLength, size_t r5, size_t r4, size_t r3,
                               size_t r2, size_t r1) {
    size_t n;  // sz3c.cpp:66-77
    if (r2 == 0) n = r1;
    else if (r3 == 0) n = r1 * r2;
    else if (r4 == 0) n = r1 * r2 * r3;
    else if (r5 == 0) n = r1 * r2 * r3 * r4;
    else n = r1 * r2 * r3 * r4 * r5;
    if (dataType != SZ_FLOAT && dataType != SZ_DOUBLE) {
        printf("dataType %d not support\n", dataType);  // sz3c.cpp:90-91
        exit(0);
    }
    sz3hip_config conf;
    if (sz3hip_peek_config(&conf, reinterpret_cast<const char *>(bytes), byteLength)) {
        fprintf(stderr, "SZ_decompress: %s\n", sz3hip_last_error());
        return nullptr;
    }
    if (conf.num > n) n = (size_t)conf.num;
    void *dec = malloc(n * (dataType == SZ_FLOAT ? 4 : 8));
    if (!dec) return nullptr;
    if (sz3hip_decompress(&conf, dataType, reinterpret_cast<const char *>(bytes), byteLength, dec)) {
        fprintf(stderr, "SZ_decompress: %s\n", sz3hip_last_error());
        free(dec);
        return nullptr;
    }
    return dec;
}

extern "C" void free_buf(void *p) { free(p); }  // sz3c.cpp:94
