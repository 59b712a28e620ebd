// sz3_amd/csrc/sz3hip_api.cpp — Config and the device-resident API of libsz3hip.so (include/sz3hip.h groups 2 and 3).
//
// Mirrors, for the GPU path, what these reference pieces do on the CPU (paths relative to /root/reference):
//   include/SZ3/utils/Config.hpp:161-177,312-413     Config::setDims / save / load
//   include/SZ3/compressor/SZGenericCompressor.hpp:38-84  stage glue: decomposition -> encoder (-> lossless on the host)
//   include/SZ3/api/impl/SZAlgoInterp.hpp:122-286    the ALGO_INTERP_LORENZO sampling tuner
// The host-buffer API (container, dispatcher policies, slab-parallel path, sz3c.h) is sz3hip_host.cpp, the RCCL
// communicator sz3hip_comm.cpp. There is NO CPU implementation of the predictor/quantizer/Huffman stages in this
// library: if the HIP device or kernels are unavailable every entry point fails loudly.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sched.h>
#include <unistd.h>
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
#include "sz3hip_internal.h"
#include "sz3hip_stock_geom.h"
#include "sz3hip_kernels.h"
#include "sz3hip_stock_host.h"

// ------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512];
static thread_local int g_err_code = 0;
int szi_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    g_err_code = code;
    return code;
}
#define fail szi_fail
extern "C" int sz3hip_lab_build(void);
extern "C" const char *sz3hip_last_error(void) { return g_err; }
extern "C" int sz3hip_last_error_code(void) { return g_err_code; }
extern "C" const char *sz3hip_version(void) { return "sz3hip 0.1 (gfx950; data format SZ3 3.3.2 container, payload SZH1)"; }

#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail(SZ3HIP_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)


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

// Config::load (Config.hpp:361-413) over at most `avail` bytes: 0 when the leading size byte or any field runs past them
extern "C" size_t sz3hip_config_load_n(sz3hip_config *c, const unsigned char *in, size_t avail) {
    uint64_t one = 1;
    sz3hip_config_init(c, 1, &one);
    if (avail < 1) return 0;
    Reader r{in};
    const uint8_t conf_size = r.get<uint8_t>();
    // (the stored size counts its own byte: Config::save writes `pos - start` over the first byte, Config.hpp:351-353)
    if (conf_size < 1 || conf_size > avail) return 0;
    const unsigned char *end = in + conf_size;
    auto room = [&](size_t k) { return (size_t)(end - r.p) >= k; };
    if (!room(2)) return 0;
    c->N = r.get<int8_t>();
    if (c->N < 0 || c->N > 4) c->N = 0;
    const uint8_t bw = r.get<uint8_t>();
    const size_t nbytes = ((size_t)bw * (size_t)c->N + 7) / 8;
    if (!room(nbytes)) return 0;
    for (int i = 0; i < c->N; i++) {  // bytes2vector, ByteUtil.hpp:240-264
        uint64_t v = 0;
        for (int j = 0; j < bw && j < 64; j++) {
            size_t bit = (size_t)i * bw + j;
            v |= (uint64_t)((r.p[bit >> 3] >> (bit & 7)) & 1) << j;
        }
        c->dims[i] = v;
    }
    r.p += nbytes;
    if (!room(10)) return 0;
    c->num = r.get<uint64_t>();
    c->cmprAlgo = r.get<uint8_t>();
    c->errorBoundMode = r.get<uint8_t>();
    switch (c->errorBoundMode) {
        case SZ3HIP_EB_ABS: if (!room(8)) return 0; c->absErrorBound = r.get<double>(); break;
        case SZ3HIP_EB_REL: if (!room(8)) return 0; c->relErrorBound = r.get<double>(); break;
        case SZ3HIP_EB_PSNR: if (!room(8)) return 0; c->psnrErrorBound = r.get<double>(); break;
        case SZ3HIP_EB_L2NORM: if (!room(8)) return 0; c->l2normErrorBound = r.get<double>(); break;
        case SZ3HIP_EB_ABS_AND_REL:
        case SZ3HIP_EB_ABS_OR_REL:
            if (!room(16)) return 0;
            c->absErrorBound = r.get<double>();
            c->relErrorBound = r.get<double>();
            break;
        default: break;
    }
    if (room(1)) {
        uint8_t b = r.get<uint8_t>();
        c->lorenzo = (b >> 7) & 1;
        c->lorenzo2 = (b >> 6) & 1;
        c->regression = (b >> 5) & 1;
        c->regression2 = (b >> 4) & 1;
        c->openmp = (b >> 3) & 1;
    }
    if (room(1)) c->dataType = r.get<uint8_t>();
    if (room(4)) c->quantbinCnt = r.get<int32_t>();
    if (room(4)) c->blockSize = r.get<int32_t>();
    if (room(1)) c->predDim = r.get<uint8_t>();
    return (size_t)conf_size;
}
extern "C" size_t sz3hip_config_load(sz3hip_config *c, const unsigned char *in) {  // the caller vouches for in[0] readable bytes
    return sz3hip_config_load_n(c, in, in[0]);
}

// ------------------------------------------------------------------------------------------------------------
// device context
// ------------------------------------------------------------------------------------------------------------
static const char *const kStageNames[ST_COUNT] = {"lorenzo_quant_hist", "codebook", "encode", "assemble",
                                                  "huffman_decode",     "reconstruct", "tuner", "k1_kernel", "step_span"};

#define SZ_COUNTER_BYTES (128 + 4 * SZK_SAMP_WORDS)  // 16 counter words, then the sampled book's state words (szk_samp)
// histogram and counters of a call start at zero; the internal histogram and the counters share one block (one fill launch)
static hipError_t clear_hist_counters(sz3hip_ctx *c, hipStream_t s) {
    if (c->pre_cleared && c->d_hist == c->d_hist_own && c->pre_stream == s) {  // (done behind the previous call, on this stream)
        c->pre_cleared = false;
        return hipSuccess;
    }
    if (c->pre_cleared && c->pre_stream != s) {
        // the zeroing behind the previous call is pending on ANOTHER stream: this call's memset (and its stage 1) must not overtake it
        hipError_t e = hipStreamSynchronize(c->pre_stream);
        if (e != hipSuccess) return e;
    }
    c->pre_cleared = false;
    if (c->d_hist == c->d_hist_own) return hipMemsetAsync(c->d_hist, 0, SZH_HIST_BINS * 8 + SZ_COUNTER_BYTES, s);
    hipError_t e = hipMemsetAsync(c->d_hist, 0, SZH_HIST_BINS * 8, s);  // caller-owned histogram (multi-GPU all-reduce buffer)
    return e != hipSuccess ? e : hipMemsetAsync(c->d_counters, 0, SZ_COUNTER_BYTES, s);
}
static void ctx_free(sz3hip_ctx *c) {
    if (!c) return;
    void *bufs[] = {c->d_dense2, c->d_work, c->d_hist_partial, c->d_codes, c->d_hist_own, c->d_vout_idx, c->d_dout_idx, c->d_vout_val, c->d_dout_val,
                    c->d_enc, c->d_lens, c->d_keys, c->d_ifreq, c->d_syms, c->d_pleaf, c->d_pint, c->d_depth, c->d_aux2, c->d_pint2, c->d_range, c->d_info,
                    c->d_chunk_words, c->d_chunk_off, c->d_carry, c->d_state, c->d_tables, c->d_segtot, c->d_minmax, c->d_samples, c->d_trial_work, c->d_trial_codes,
                    c->d_trial, c->d_passes, c->d_np,  // (d_trial_counters / d_trial_hist live inside d_trial's block)
                    c->d_blk_sel, c->d_blk_coef, c->d_blk_rank, c->d_blk_comp, c->d_blk_side, c->d_blk_counters,
                    c->bk[1].enc, c->bk[1].lens, c->bk[1].info, c->d_seg_bits, c->d_seg_base, c->d_half32, c->d_sub_bits, c->d_fuse_scratch};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (c->book_stream) {
        (void)hipStreamSynchronize(c->book_stream);
        (void)hipStreamDestroy(c->book_stream);
        if (c->ev_s1) (void)hipEventDestroy(c->ev_s1);
        if (c->ev_book) (void)hipEventDestroy(c->ev_book);
    }
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->d_flags) (void)hipHostFree(c->d_flags);
    if (c->d_starts) (void)hipHostFree(c->d_starts);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->h_minmax) (void)hipHostFree(c->h_minmax);
    if (c->h_trial) (void)hipHostFree(c->h_trial);
    if (c->h_trial_codes) (void)hipHostFree(c->h_trial_codes);
    if (c->h_samples) (void)hipHostFree(c->h_samples);
    if (c->h_passes) (void)hipHostFree(c->h_passes);
    if (c->h_np) (void)hipHostFree(c->h_np);
    if (c->d_blk_carry) (void)hipFree(c->d_blk_carry);
    if (c->h_blk_side_hdr) (void)hipHostFree(c->h_blk_side_hdr);
    if (c->d_blk_stats5) (void)hipFree(c->d_blk_stats5);
    if (c->h_ovf) (void)hipHostFree(c->h_ovf);
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
    c->narrow_hint = c->cb_hint = c->cb_part = -1;
    c->q16_hint = c->q16_block = 0;
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
    {   // (the fused stage 1 is opt-in: measured slower than the two-pass form on this chip; SZ3HIP_FUSED=1 turns it on for every context)
        const char *fe = getenv("SZ3HIP_FUSED");
        c->fuse_on = fe && fe[0] == '1' && sz3hip_lab_build();
    }
    alloc((void **)&c->d_seg_bits, (max_elems / 256 + 8) * 2);
    alloc((void **)&c->d_seg_base, (max_elems / 256 + 8) * 4);
    alloc((void **)&c->bk[1].enc, SZH_HIST_BINS * 4);
    alloc((void **)&c->bk[1].lens, SZH_HIST_BINS);
    alloc((void **)&c->bk[1].info, sizeof(szk_cb_info));
    // (a speculative stage 2 may look symbols up that the book it runs with never held: such entries must read "no code", not
    // whatever the allocation contained — a length beyond the format's limit would overrun the packer's LDS stage)
    if (ok) ok = hipMemset(c->d_enc, 0, SZK_MAX_BOOKS * SZH_HIST_BINS * 4) == hipSuccess && hipMemset(c->d_lens, 0, SZK_MAX_BOOKS * SZH_HIST_BINS) == hipSuccess &&
                 hipMemset(c->bk[1].enc, 0, SZH_HIST_BINS * 4) == hipSuccess && hipMemset(c->bk[1].lens, 0, SZH_HIST_BINS) == hipSuccess &&
                 hipMemset(c->d_info, 0, SZK_MAX_BOOKS * sizeof(szk_cb_info)) == hipSuccess && hipMemset(c->bk[1].info, 0, sizeof(szk_cb_info)) == hipSuccess;
    c->bk[0].enc = c->d_enc;
    c->bk[0].lens = c->d_lens;
    c->bk[0].info = c->d_info;
    c->book_idx = c->book_pending = -1;
    alloc((void **)&c->d_chunk_words, (c->max_chunks + 8) * 2);
    alloc((void **)&c->d_sub_bits, (c->max_chunks + 8) * 2 * (SZH_SUBS - 1));
    alloc((void **)&c->d_chunk_off, (c->max_chunks + 8) * 8);
    alloc((void **)&c->d_state, sizeof(szk_state));
    alloc((void **)&c->d_tables, sizeof(szk_dec_tables));
    alloc(&c->d_segtot, (4 * max_elems / 16384 + 65536) * 8);
    alloc((void **)&c->d_minmax, (2 * 1024 + 2) * 8);
    // (the device writes this copy itself, k_publish: coherent host memory, the sequence word behind the state block)
    if (ok && hipHostMalloc((void **)&c->h_state, sizeof(szk_state) + 128, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) ok = false;
    if (ok) {
        memset(c->h_state, 0, sizeof(szk_state) + 128);
        c->h_pub_seq = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(c->h_state) + ((sizeof(szk_state) + 63) & ~(size_t)63));
        c->pub_seq = 0;
    }
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

// side_blocks: blocks of the block-composed predictor the side section must hold
static size_t payload_bound_blocks(uint64_t n, uint64_t out_cap, uint64_t side_blocks) {
    const uint64_t n_chunks = (n + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS;
    return (size_t)(sizeof(szh_header) + SZH_HIST_BINS + 16 + 2 * n_chunks + 16 + 2 * (SZH_SUBS - 1) * n_chunks + 16 + 2 * (out_cap * 16 + 16) +
                    4 * n_chunks * (SZH_CHUNK_SYMS * SZH_MAX_LEN / 32) + 64 + szk_blk_side_bound(side_blocks));
}
// The shape-blind bound: blocks have an edge of at least 4, so a 3-D array whose extents are all >= 9 has at most n / 27 of
// them (ceil(d / 4) <= d / 3). Thin arrays (an extent below 9: a one-plane slab, say) can have more — up to n / 4; stage 2
// checks the caller's capacity against the blocks the call really has, and sz3hip_payload_bound_conf sizes a buffer for them.
static size_t payload_bound_n(uint64_t n, uint64_t out_cap) { return payload_bound_blocks(n, out_cap, n / 27 + 64); }
// shapes the block-composed predictor is built for: 3-D with block edges 4..8 (tiles in LDS), 1-D with blocks of 4..65535 values,
// 2-D with block edges 4..32 (the decoder's block in LDS); second-order Lorenzo in 1-D, 2-D and 3-D (decided where the set is known)
static bool blk_shape_ok(const sz3hip_config *conf) {
    if (conf->N == 3) return conf->blockSize >= 4 && conf->blockSize <= 8;
    if (conf->N == 2) return conf->blockSize >= 4 && conf->blockSize <= 32 && conf->dims[0] < 0xFFFFFFFFull && conf->dims[1] < 0xFFFFFFFFull;
    if (conf->N == 1) return conf->blockSize >= 4 && conf->blockSize <= 65535 && conf->dims[0] < 0xFFFFFFFFull;  // (positions in 32 bits)
    if (conf->N == 4) {  // (round 4: blocks of up to 6^4 values — the decoder's tile of (B + 1)^4 lattice words sits in LDS twice)
        for (int i = 0; i < 4; i++)
            if (conf->dims[i] >= 0xFFFFFFFFull) return false;
        return conf->blockSize >= 4 && conf->blockSize <= 6;
    }
    return false;
}
// the kernels' view of the array: three extents, the caller's right-aligned (1-D: (1, 1, n), 2-D: (1, dy, dx))
// (a 4-D array: its three fast extents; the slowest travels beside them — blk_view_w)
static void blk_view(int N, const uint64_t *dims, uint64_t *d3) {
    for (int i = 0; i < 3; i++) d3[i] = 1;
    if (N == 4) {
        for (int i = 0; i < 3; i++) d3[i] = dims[i + 1];
        return;
    }
    for (int i = 0; i < N && i < 3; i++) d3[3 - N + i] = dims[i];
}
static uint64_t blk_view_w(int N, const uint64_t *dims) { return N == 4 ? dims[0] : 1; }
static uint64_t conf_blocks(const sz3hip_config *conf) {  // blocks the block-composed predictor would cut this array into (0: not its shape)
    if (!blk_shape_ok(conf)) return 0;
    uint64_t nb = 1;
    for (int i = 0; i < conf->N; i++) nb *= (conf->dims[i] + (uint64_t)conf->blockSize - 1) / (uint64_t)conf->blockSize;
    return nb;
}
extern "C" size_t sz3hip_payload_bound(const sz3hip_ctx *ctx, uint64_t n) { return payload_bound_n(n, ctx->out_cap); }
// lists of up to n / 8 entries: beyond that the stream cannot beat the lossless fallback any more
static uint64_t out_cap_limit(uint64_t n) { return std::max<uint64_t>(1024, n / 8); }
extern "C" size_t sz3hip_payload_bound_max(const sz3hip_ctx *ctx, uint64_t n) {
    return payload_bound_n(n, std::max<uint64_t>(ctx->out_cap, out_cap_limit(n)));
}
extern "C" size_t sz3hip_payload_bound_conf(const sz3hip_ctx *ctx, const sz3hip_config *conf, int worst_case) {
    const uint64_t n = conf->num;
    const uint64_t lists = worst_case ? std::max<uint64_t>(ctx->out_cap, out_cap_limit(n)) : ctx->out_cap;
    return std::max(payload_bound_n(n, lists), payload_bound_blocks(n, lists, conf_blocks(conf)));
}
extern "C" void *sz3hip_histogram_ptr(sz3hip_ctx *ctx) {
    ctx->hist_exposed = true;  // (whoever holds the pointer may change the histogram between the stages)
    return ctx->d_hist;
}
// The library's own exchange (sz3hip_comm_allreduce_histogram): the histogram is summed over the ranks ON THE COMPRESS STREAM between
// stage 1 and stage 2 and by nobody else — known well enough for stage 2 to speculate on the previous call's (global) code book:
// stage 1 completes its histogram itself (the exchange needs it whole) and waits for the probe (the one-launch form's repeat of a
// whole call could not redo the exchange); stage 2 recounts the alphabet's range from the summed histogram.
void *szi_histogram_for_exchange(sz3hip_ctx *ctx) {
    ctx->hist_reduced = true;
    return ctx->d_hist;
}
extern "C" size_t sz3hip_histogram_len(const sz3hip_ctx *) { return SZH_HIST_BINS; }
extern "C" int sz3hip_ctx_set_histogram(sz3hip_ctx *ctx, void *d_hist) {
    ctx->d_hist = d_hist ? (uint64_t *)d_hist : ctx->d_hist_own;
    ctx->hist_exposed = d_hist != nullptr;
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
    if (conf->N == 1) ip.direction = 0;  // (one order whatever the field says: a one-row slab of a 2-D array under SZ_compress_OMP's split comes here with the caller's value)
    if (ip.direction < 0 || ip.direction >= nperm) return fail(SZ3HIP_EINVAL, "interpDirection out of range");
    ip.alpha = conf->interpAlpha;
    ip.beta = conf->interpBeta;
    ip.eb = eb;
    ip.radius = radius;
    return 0;
}
static void cb_params_from(sz3hip_ctx *ctx, szk_cb_params &cb, uint64_t out_cap, int slot) {
    memset(&cb, 0, sizeof(cb));
    cb.enc = ctx->bk[slot].enc;
    cb.lens = ctx->bk[slot].lens;
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
    cb.info = ctx->bk[slot].info;
    cb.n_books = 1;
    cb.range_ready = ctx->range_ready && !ctx->hist_exposed && !ctx->hist_reduced;
    cb.part_hint = (szk_dbg_flags & 131072) ? -1 : ctx->cb_part;
    cb.mispredict = reinterpret_cast<uint32_t *>(ctx->d_counters + 7);  // (zeroed with the counters)
}

// the level kernels hand the grid of stride 2 over as a dense array (sz3hip_interp.hip, szk_interp_level::dense): room for it in the context
static void dense2_for(sz3hip_ctx *ctx, szk_interp_params &ip) {
    ip.dense2 = nullptr;
    ip.dense2_elems = 0;
    if (ip.N != 3 || !szk_interp_levels_ok(&ip) || (szk_dbg_flags & 536870912)) return;
    const size_t need = (size_t)(((ip.dims[0] - 1) / 2 + 1) * ((ip.dims[1] - 1) / 2 + 1) * ((ip.dims[2] - 1) / 2 + 1));
    if (ctx->dense2_elems < need) {
        if (ctx->d_dense2) (void)hipFree(ctx->d_dense2);
        ctx->d_dense2 = nullptr;
        ctx->dense2_elems = 0;
        if (hipMalloc(&ctx->d_dense2, need * (ctx->dtype == SZ3HIP_FLOAT ? 4 : 8)) == hipSuccess) ctx->dense2_elems = need;
        else (void)hipGetLastError();  // (no room: the hand-over stays in place)
    }
    if (ctx->dense2_elems >= need) {
        ip.dense2 = ctx->d_dense2;
        ip.dense2_elems = ctx->dense2_elems;
    }
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
    if (!ctx->copy_ahead) dense2_for(ctx, ip);
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

// may stage 2 of a call with this predictor and radius run with the context's previous code book? (decided once per call)
static bool book_spec_ok(const sz3hip_ctx *ctx, uint32_t predictor, uint32_t radius) {
    // (spec_skip: calls left to sit out after a miss — a series whose books keep changing pays for one failed attempt in
    // 2, 4, 8 calls, not in every call; spec_off == 2 switches the back-off off for tests)
    return ctx->book_idx >= 0 && ctx->spec_off != 1 && (ctx->spec_skip == 0 || ctx->spec_off == 2) && !(szk_dbg_flags & 131072) &&
           ctx->book_pred == predictor && ctx->book_radius == radius;
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
    // range words of the alphabet kept by stage 1 itself (whoever finds a bin empty enters it): the one-launch form a context takes
    // after a one-byte call does (few bins); the two-byte forms flush thousands of bins per workgroup, where the returning atomic
    // that takes costs more than the k_hist_range launch it saves (szk_launch_k1 reports through range_kept what was done)
    p.range = reinterpret_cast<uint32_t *>(ctx->d_counters + 8);
    p.hint_narrow = allow_narrow ? ctx->narrow_hint : -1;
    // (a caller that holds the histogram exchanges it between the stages — multi-GPU: the one-launch form's repeat of a whole
    // call from inside finish() could not redo that exchange, so such contexts keep the form that waits for the probe)
    if ((ctx->hist_exposed || ctx->hist_reduced) && p.hint_narrow > 0) p.hint_narrow = -1;
    p.hint_q16 = allow_narrow && ctx->dtype == SZ3HIP_FLOAT && ctx->q16_hint > 0 && ctx->q16_block == 0 ? 1 : 0;
    p.q16_flag = reinterpret_cast<uint32_t *>(ctx->d_counters + 11) + 1;  // zeroed with the counters
    // Round 6, the sampled book: an array of at least SZK_SAMP_MIN_ELEMS elements in rows of whole 256-element segments (1-D ... 3-D)
    // whose stream turns out to have one-byte codes is coded with a book built from a sample of it (sz3hip_kernels.h, szk_samp) — a
    // function of the input alone, whatever this context coded before and whichever form of stage 1 it takes. Not for contexts whose
    // histogram is exchanged between the stages (multi-GPU: the ranks share one book made from the summed histogram).
    const bool samp = allow_narrow && radius >= 128 && N <= 3 && p.d[3] % 256 == 0 && num >= SZK_SAMP_MIN_ELEMS && !ctx->hist_exposed && !ctx->hist_reduced &&
                      !(szk_dbg_flags & 65536);
    if (samp) {
        const int fresh = ctx->book_idx < 0 ? 0 : 1 - ctx->book_idx;  // (stage2_launch's slot for this call's book)
        p.samp.words = reinterpret_cast<uint32_t *>(ctx->d_counters + 16);  // zeroed with the counters
        p.samp.enc = ctx->bk[fresh].enc;
        p.samp.lens = ctx->bk[fresh].lens;
        p.samp.info = ctx->bk[fresh].info;
        p.seg_bits = ctx->d_seg_bits;
        p.seg_made = reinterpret_cast<uint32_t *>(ctx->d_counters + 10) + 1;  // zeroed with the counters
    }
    if (!samp && allow_narrow && book_spec_ok(ctx, 0, (uint32_t)radius) && ctx->cb_hint == 0 && !ctx->lists_long && !ctx->hist_exposed) {
        // stage 2 will pack with the previous call's book: its code lengths let the one-byte kernel sum the code bits of the
        // 256-element segments (no bits pass), and the fold of the histogram rows moves into the scan's launch of stage 2
        p.spec_lens = ctx->bk[ctx->book_idx].lens;
        p.seg_bits = ctx->d_seg_bits;
        p.seg_made = reinterpret_cast<uint32_t *>(ctx->d_counters + 10) + 1;  // zeroed with the counters
        p.defer_fold = ctx->hist_exposed || ctx->hist_reduced || (szk_dbg_flags & 16777216) ? 0 : 1;  // (whoever exchanges the histogram wants it complete after stage 1)
        // Round 4: stage 1 may code with that book itself (the fused form, k_lorenzo_quant_march3f: the launcher takes it for the
        // one-launch form on rows of whole segments when the scratch — the code array's memory — holds a slot per task). A verdict
        // miss then costs the whole call once more: a context that wants its payloads bit-identical to a fresh context's
        // (deterministic mode) asks for it only behind a call whose book was confirmed.
        const uint64_t need = szk_fuse_scratch_words(N, p.d);
        if (need && ctx->fuse_on && !ctx->hist_reduced && (!ctx->spec_exact || ctx->last_spec_hit)) {
            p.fuse_slots = reinterpret_cast<uint32_t *>(ctx->d_codes);
            p.fuse_cap_words = (ctx->max_n + 64) / 2;  // (the array holds (max_n + 64) two-byte codes)
            if (need > p.fuse_cap_words && need <= 2 * num + (1u << 18)) {
                // extents that are not multiples of the tasks' 4 rows x 16 planes: the slots of the tasks that end beyond the array
                // do not fit the code array's 2 bytes per element — a scratch of the context's own, for up to 4 x that
                if (ctx->fuse_scratch_words < need) {
                    if (ctx->d_fuse_scratch) (void)hipFree(ctx->d_fuse_scratch);
                    ctx->d_fuse_scratch = nullptr;
                    ctx->fuse_scratch_words = 0;
                    if (hipMalloc((void **)&ctx->d_fuse_scratch, need * 4) == hipSuccess) ctx->fuse_scratch_words = need;
                    else (void)hipGetLastError();
                }
                if (ctx->fuse_scratch_words >= need) {
                    p.fuse_slots = ctx->d_fuse_scratch;
                    p.fuse_cap_words = ctx->fuse_scratch_words;
                }
            }
            p.fuse = 1;
            p.fuse_enc = ctx->bk[ctx->book_idx].enc;
            p.fuse_info = ctx->bk[ctx->book_idx].info;
            p.seg_base = ctx->d_seg_base;
            p.fuse_flag = reinterpret_cast<uint32_t *>(ctx->d_counters + 11);  // zeroed with the counters
        }
    }
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
    ctx->range_ready = p.range_kept != 0;
    ctx->s1_assumed_narrow = p.assumed_narrow != 0;
    ctx->s1_q16 = p.assumed_q16 != 0;
    ctx->s1_spec = p.spec_lens != nullptr;
    ctx->seg_expected = p.seg_expected != 0 && !(szk_dbg_flags & 33554432);
    ctx->fold_rows = p.defer_fold ? p.fold_rows : 0;
    ctx->fold_range = p.range;
    ctx->s1_fused = p.fused != 0;
    ctx->s1_slots = p.fuse_slots;
    ctx->s1_samp = p.samp.words != nullptr;  // (the launcher drops it for forms that have no one-byte codes)
    ctx->s1_samp_in = ctx->s1_samp && p.samp_in_launch != 0;
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

// ---- stage 1, block-composed predictor: Lorenzo-1 / Lorenzo-2 / regression chosen per block
// (make_compressor_lorenzo_regression, api/impl/SZAlgoLorenzoReg.hpp:22-64) ----
static int blk_reserve_select(sz3hip_ctx *ctx, uint64_t nblocks) {  // (all the selection pass needs: counters, choices, coefficients)
    if (!ctx->d_blk_counters) HIPCHK(hipMalloc((void **)&ctx->d_blk_counters, 64 + 4 * (0x7FFFFFF0ull / 8192 + 2)));
    if (!ctx->d_blk_stats5) HIPCHK(hipMalloc((void **)&ctx->d_blk_stats5, 128));
    if (!ctx->h_blk_side_hdr) HIPCHK(hipHostMalloc((void **)&ctx->h_blk_side_hdr, 64));
    if (ctx->blk_sel_cap >= nblocks) return 0;
    void **arr[2] = {(void **)&ctx->d_blk_sel, (void **)&ctx->d_blk_coef};
    for (void **a : arr) {
        if (*a) (void)hipFree(*a);
        *a = nullptr;
    }
    ctx->blk_sel_cap = 0;
    HIPCHK(hipMalloc((void **)&ctx->d_blk_sel, nblocks));
    HIPCHK(hipMalloc((void **)&ctx->d_blk_coef, nblocks * 40));  // (up to five coefficients a block: 4-D arrays)
    ctx->blk_sel_cap = nblocks;
    return 0;
}
static int blk_reserve(sz3hip_ctx *ctx, uint64_t nblocks) {
    if (!ctx->d_work) HIPCHK(hipMalloc(&ctx->d_work, ctx->max_n * (ctx->dtype == SZ3HIP_FLOAT ? 4 : 8)));
    int rcc = blk_reserve_select(ctx, nblocks);
    if (rcc) return rcc;
    if (ctx->blk_cap >= nblocks) return 0;
    void **arr[3] = {(void **)&ctx->d_blk_rank, (void **)&ctx->d_blk_comp, (void **)&ctx->d_blk_side};
    for (void **a : arr) {
        if (*a) (void)hipFree(*a);
        *a = nullptr;
    }
    ctx->blk_cap = 0;
    HIPCHK(hipMalloc((void **)&ctx->d_blk_rank, nblocks * 4));
    HIPCHK(hipMalloc((void **)&ctx->d_blk_comp, nblocks * 4 + 64));  // (+ 64: the decoder keeps the coefficient groups' sums here, 32 bytes per 64 regression blocks)
    HIPCHK(hipMalloc((void **)&ctx->d_blk_side, szk_blk_side_bound(nblocks)));
    ctx->blk_cap = nblocks;
    return 0;
}
static void blk_params_from(sz3hip_ctx *ctx, int ndim, const uint64_t *dims3, uint32_t B, uint32_t mask, double eb, int radius, uint64_t out_cap,
                            szk_blk_params &bp, szk_blk_scratch &sc, uint64_t dw = 1) {
    memset(&bp, 0, sizeof(bp));
    memset(&sc, 0, sizeof(sc));
    for (int i = 0; i < 3; i++) {
        bp.d[i] = dims3[i];
        bp.nb[i] = (uint32_t)((dims3[i] + B - 1) / B);
    }
    bp.dw = dw;
    bp.nbw = (uint32_t)((dw + B - 1) / B);
    sc.stats5 = reinterpret_cast<double *>(ctx->d_blk_stats5);
    bp.B = B;
    bp.ndim = (uint32_t)ndim;
    bp.carry = ctx->d_blk_carry;
    bp.mask = mask;
    bp.lat = szk_make_lattice(eb);
    bp.eb = eb;
    bp.radius = (uint32_t)radius;
    bp.out_cap = out_cap;
    bp.hist = ctx->d_hist;
    bp.n_vout = ctx->d_counters + 0;
    bp.n_dout = ctx->d_counters + 1;
    bp.vout_idx = ctx->d_vout_idx;
    bp.dout_idx = ctx->d_dout_idx;
    bp.vout_val = ctx->d_vout_val;
    bp.dout_val = ctx->d_dout_val;
    bp.sel = ctx->d_blk_sel;
    bp.coef = ctx->d_blk_coef;
    bp.qwork = ctx->d_work;
    bp.n_reg = ctx->d_blk_counters + 3;
    sc.rank = ctx->d_blk_rank;
    sc.comp = ctx->d_blk_comp;
    sc.counters = ctx->d_blk_counters;
    sc.side = ctx->d_blk_side;
    sc.run_scratch = reinterpret_cast<uint32_t *>(ctx->d_blk_counters + 8);  // (the counter block holds 8 words + a run table)
}
// The selection first (k_blk_select): when fewer than one block in 4096 would be coded by anything but first-order Lorenzo, every
// block is — the stream is then the plain Lorenzo stream (same lattice, same stencil: a Lorenzo block's neighbours are lattice
// values either way), made by the plain kernel and decoded by the global prefix sums instead of block fronts, and the selection
// bits (2 per block: more than those few blocks save) are not stored. The few blocks take the reference's own fallback
// predictor (BlockwiseDecomposition.hpp:35-37). One 8-byte read-back and a stream synchronisation decide.
#define BLK_EXIT_SHIFT 12
static inline uint64_t *blk_sel_count(sz3hip_ctx *ctx) { return reinterpret_cast<uint64_t *>(ctx->d_blk_stats5) + 8; }
static int blk_all_lorenzo(sz3hip_ctx *ctx, const sz3hip_config *conf, const void *d_in, double eb, int radius, uint32_t mask, hipStream_t s, bool *all) {
    *all = false;
    ctx->blk_sel_given = false;
    ctx->blk_spec = false;
    if (szk_dbg_flags & 2147483648u) return 0;  // (development: no selection pass, the fit pass chooses by its own wave sums)
    // (1-D: the fit pass chooses — the estimate looks at a block's two ends only and a line through 128 values wins there on every
    // field with noise above the bound, as in the reference: a selection pass of its own found no field to hand over and cost a
    // launch and a synchronisation, 2^27 values 1.97 -> 2.22 ms, C1 0.235 -> 0.27 ms)
    if (conf->N == 1 || conf->N == 4) return 0;  // (4-D: the plain form — the fit pass chooses)
    const uint32_t B = (uint32_t)conf->blockSize;
    uint64_t d3[3];
    blk_view(conf->N, conf->dims, d3);
    uint64_t nblocks = 1;
    for (int i = 0; i < 3; i++) nblocks *= (d3[i] + B - 1) / B;
    if (nblocks > 0x7FFFFFF0ull) return 0;
    int rc = blk_reserve_select(ctx, nblocks);
    if (rc) return rc;
    szk_blk_params bp;
    szk_blk_scratch sc;
    blk_params_from(ctx, conf->N, d3, B, mask, eb, radius, ctx->cur_out_cap, bp, sc);
    // (the count has a word of its own behind the Rice statistics: the counter block's eighth word is the fourth coefficient's statistic,
    // which a block stream's stage 1 adds up AFTER this pass — the speculative form below reads the count at the call's end)
    uint64_t *sel_cnt = blk_sel_count(ctx);
    if (!ctx->blk_cleared_now) HIPCHK(hipMemsetAsync(sel_cnt, 0, 8, s));  // (else: k_publish zeroed it behind the previous call)
    prof_begin(ctx, ST_TUNER, s);
    rc = szk_launch_blk_select(ctx->dtype, d_in, &bp, sel_cnt, s);
    prof_end(ctx, ST_TUNER, s);
    if (rc) return fail(SZ3HIP_EHIP, "block selection kernel launch failed (%d)", rc);
    ctx->blk_sel_given = true;  // (the choices and coefficients are in d_blk_sel / d_blk_coef: the fit pass codes what they say)
    // The decision — every block Lorenzo-1: the plain stream — is a function of the pass's count. A context whose previous call of the
    // same configuration made it from a count assumes the same outcome and goes on without the count's round trip to the host (a copy,
    // a synchronisation and the relaunch latency: ~65 us of an idle GPU at C4's slab); the count travels with the state block at the
    // call's end (k_publish) and finish() repeats the call in this waiting form when it decides otherwise.
    const sz3hip_config &k = ctx->blk_dec_conf;
    bool same = ctx->blk_dec_valid && k.N == conf->N && k.blockSize == conf->blockSize && k.absErrorBound == conf->absErrorBound && k.quantbinCnt == conf->quantbinCnt &&
                k.lorenzo == conf->lorenzo && k.lorenzo2 == conf->lorenzo2 && k.regression == conf->regression;
    for (int i = 0; same && i < conf->N; i++) same = k.dims[i] == conf->dims[i];
    if (same && ctx->spec_off != 1 && !(szk_dbg_flags & 1073741824)) {
        ctx->blk_spec = true;
        ctx->blk_spec_all = ctx->blk_dec_all;
        ctx->blk_spec_nblocks = nblocks;
        ctx->blk_spec_mask = mask;
        *all = ctx->blk_dec_all;
        return 0;
    }
    HIPCHK(hipMemcpyAsync(ctx->h_blk_side_hdr, sel_cnt, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    uint64_t others;
    memcpy(&others, ctx->h_blk_side_hdr, 8);
    ctx->blk_others = others;
    *all = (mask & 1u) && !(szk_dbg_flags & 1073741824) && (others << BLK_EXIT_SHIFT) < nblocks;
    ctx->blk_dec_valid = !(szk_dbg_flags & 1073741824);
    ctx->blk_dec_all = *all;
    ctx->blk_dec_conf = *conf;
    return 0;
}
static int stage1_blocks(sz3hip_ctx *ctx, const sz3hip_config *conf, const void *d_in, double eb, int radius, uint64_t num, uint32_t mask, hipStream_t s) {
    const uint32_t B = (uint32_t)conf->blockSize;
    uint64_t d3[3];
    blk_view(conf->N, conf->dims, d3);
    const uint64_t dw = blk_view_w(conf->N, conf->dims);
    uint64_t nblocks = (dw + B - 1) / B;
    for (int i = 0; i < 3; i++) nblocks *= (d3[i] + B - 1) / B;
    if (nblocks > 0x7FFFFFF0ull) return fail(SZ3HIP_EUNSUPPORTED, "too many blocks for the block-composed predictor");
    int rc = blk_reserve(ctx, nblocks);
    if (rc) return rc;
    szk_blk_params bp;
    szk_blk_scratch sc;
    blk_params_from(ctx, conf->N, d3, B, mask, eb, radius, ctx->cur_out_cap, bp, sc, dw);
    bp.sel_given = ctx->blk_sel_given ? 1u : 0u;
    sc.wide_hist = ctx->blk_wide;
    if (!ctx->blk_cleared_now) {  // (else: zeroed behind the previous call by its k_publish, on this stream)
        HIPCHK(hipMemsetAsync(ctx->d_blk_counters, 0, 64, s));
        HIPCHK(hipMemsetAsync(ctx->d_blk_stats5, 0, 64, s));
    }
    ctx->blk_cleared_now = false;
    // (round 6) a small array's side launch finds the range of the histogram's non-empty bins on the way: stage 2 launches no k_hist_range
    const bool range_made = szk_blk_side_small(nblocks) && ctx->d_hist == ctx->d_hist_own && !ctx->hist_exposed;
    if (range_made) {
        sc.range_hist = ctx->d_hist;
        sc.range = reinterpret_cast<uint32_t *>(ctx->d_counters + 8);  // (zeroed with the counters)
    }
    prof_begin(ctx, ST_K1, s);
    rc = szk_launch_blk_compress(ctx->dtype, d_in, ctx->d_codes, &bp, &sc, s);
    prof_end(ctx, ST_K1, s);
    if (rc) return fail(SZ3HIP_EHIP, "block predictor kernel launch failed (%d)", rc);
    ctx->range_ready = range_made;
    memset(&ctx->mode, 0, sizeof(ctx->mode));
    ctx->mode.probe_big = reinterpret_cast<uint32_t *>(ctx->d_counters + 4);
    szh_header &h = ctx->proto;
    memset(&h, 0, sizeof(h));
    h.magic = SZH_MAGIC;
    h.version = SZH_VERSION;
    h.dtype = (uint8_t)ctx->dtype;
    h.ndim = (uint8_t)conf->N;
    h.qbytes = ctx->dtype == SZ3HIP_FLOAT ? 4 : 8;
    h.predictor = 2;
    h.radius = (uint32_t)radius;
    h.dims[0] = dw;
    for (int i = 0; i < 3; i++) h.dims[1 + i] = d3[i];
    h.eb = eb;
    h.n = num;
    h.chunk_syms = SZH_CHUNK_SYMS;
    h.n_chunks = (num + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS;
    h.interp_id = B;
    h.interp_dir = mask;
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
// result words of the tuner's trials (the first 1024 bytes of d_trial): [4 * slot ..] the interpolation trials' costs (slots 0 .. 6), [24 .. 31] a
// Lorenzo trial priced on its own (cost, then the blocks that took second order and their number), [32 .. 39] and [40 .. 47] the two Lorenzo
// trials of a 1-D array when they ride behind the first interpolation group (round 6)
#define SZ_TRIAL_WORDS 48
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
    const size_t tsz_r = ctx->dtype == SZ3HIP_FLOAT ? 4 : 8;
    const size_t code_bytes = (samples / tsz_r) * SZK_MAX_TRIALS * 2;  // two bytes per sampled point and trial
    if (ctx->trial_codes_cap < code_bytes) {
        if (ctx->d_trial_codes) (void)hipFree(ctx->d_trial_codes);
        ctx->d_trial_codes = nullptr;
        HIPCHK(hipMalloc((void **)&ctx->d_trial_codes, code_bytes));
        ctx->trial_codes_cap = code_bytes;
    }
    if (!ctx->d_trial) {  // one block [results 256 B][counters 512 B][pad][histograms]: a group zeroes it with one memset
        HIPCHK(hipMalloc(&ctx->d_trial, 1024 + SZK_MAX_TRIALS * SZH_HIST_BINS * 8));
        ctx->d_trial_counters = ctx->d_trial + 32;
        ctx->d_trial_hist = ctx->d_trial + 128;
    }
    if (!ctx->h_trial) HIPCHK(hipHostMalloc((void **)&ctx->h_trial, SZ_TRIAL_WORDS * 8));
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
    int rc = szk_launch_interp_trials(ctx->dtype, ips, (uint32_t)ntr, ctx->d_samples, ctx->d_trial_work, ctx->d_trial_codes, nb, ctx->d_trial_hist,
                                      ctx->h_passes, ctx->d_passes, ctx->h_np, ctx->d_np, ctx->exact_now ? 1 : 0, s);
    if (rc) return fail(SZ3HIP_EHIP, "tuner: interpolation trial launch failed (%d)", rc);
    rc = szk_launch_code_cost(ctx->d_trial_hist, ctx->d_trial_counters, ctx->d_trial + 4 * slot0, (uint32_t)ntr, tcs[0].num * nb, 1, s);
    if (rc) return fail(SZ3HIP_EHIP, "tuner: cost kernel launch failed (%d)", rc);
    if (!ctx->exact_now) return 0;
    // The reference's own price of every trial (interp_compress_test, api/impl/SZAlgoInterp.hpp:42-78): the codes of all sampled blocks in
    // the order its decomposition emits them, one Huffman tree built with its queue, its serialised buffer, ZSTD_compress at level 3 — on
    // the host, a thread per trial, from the per-element codes the trial kernel's global-memory form leaves (bit-identical to the
    // reference's code of every element). Milliseconds per tuning instead of 0.2: a switch for callers who want the reference's decisions.
    const size_t tsz = ctx->dtype == SZ3HIP_FLOAT ? 4 : 8;
    const uint64_t per = tcs[0].num;
    const size_t cbytes = (size_t)ntr * nb * per * 2, sbytes = (size_t)nb * per * tsz;
    if (ctx->h_trial_codes_cap < cbytes) {
        if (ctx->h_trial_codes) (void)hipHostFree(ctx->h_trial_codes);  // (pinned: 8 MB of codes per group at C3, a staged copy into pageable memory took longer than the trials)
        ctx->h_trial_codes = nullptr;
        ctx->h_trial_codes_cap = 0;
        HIPCHK(hipHostMalloc((void **)&ctx->h_trial_codes, cbytes));
        ctx->h_trial_codes_cap = cbytes;
    }
    if (ctx->h_samples_cap < sbytes) {
        if (ctx->h_samples) (void)hipHostFree(ctx->h_samples);
        ctx->h_samples = nullptr;
        ctx->h_samples_cap = 0;
        ctx->h_samples_valid = false;
        HIPCHK(hipHostMalloc(&ctx->h_samples, sbytes));
        ctx->h_samples_cap = sbytes;
    }
    static const bool tt = getenv("SZ3HIP_TUNER_TIMING") != nullptr;  // (development: where an exactly priced group's time goes)
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipMemcpyAsync(ctx->h_trial_codes, ctx->d_trial_codes, cbytes, hipMemcpyDeviceToHost, s));
    if (!ctx->h_samples_valid) HIPCHK(hipMemcpyAsync(ctx->h_samples, ctx->d_samples, sbytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    ctx->h_samples_valid = true;
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<size_t> sizes((size_t)ntr, 0);
    auto params = [&](int j) {
        szi_stock_params sp;
        memset(&sp, 0, sizeof(sp));
        sp.N = tcs[j].N;
        for (int i = 0; i < sp.N; i++) sp.dims[i] = tcs[j].dims[i];
        sp.interp_id = tcs[j].interpAlgo;
        sp.direction = tcs[j].interpDirection;
        sp.anchor_stride = (uint64_t)tcs[j].interpAnchorStride;
        sp.alpha = tcs[j].interpAlpha;
        sp.beta = tcs[j].interpBeta;
        sp.eb = eb;
        sp.radius = radius;
        return sp;
    };
    // the steps of stock::TrialWork over the pool's threads, two per trial where a step divides (order, bits): 14 tasks for a group of seven
    auto priced_as = [&](auto zero) {
        typedef decltype(zero) T;
        std::vector<stock::TrialWork<T>> w((size_t)ntr);
        std::vector<char> ok((size_t)ntr, 1);
        for (int j = 0; j < ntr; j++) {
            w[j].p = params(j);
            w[j].codes = ctx->h_trial_codes + (size_t)j * nb * per;
            w[j].samples = (const T *)ctx->h_samples;
            w[j].nb = nb;
            ok[j] = w[j].prepare();
        }
        // (a trial from end to end on one thread: 3.3 against 3.0 ms per group at C3 — but a 4 MB array's trials are 8 K codes each, and four
        // hand-overs to the pool cost more than they share out: 0.42 -> ~0.3 ms per group; round 6. SZ3HIP_TUNER_PHASES = 0 / 1 forces either)
        static const int phases_env = getenv("SZ3HIP_TUNER_PHASES") ? atoi(getenv("SZ3HIP_TUNER_PHASES")) : -1;
        const int phases = phases_env >= 0 ? phases_env : (nb * per >= 65536 ? 1 : 0);
        double ph[4] = {0, 0, 0, 0};
        auto lap = [&](int k, const std::chrono::steady_clock::time_point &from) { ph[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - from).count(); };
        auto p0 = std::chrono::steady_clock::now();
        if (phases) {
            szi_run_parallel(2 * ntr, [&](int i) { if (ok[i / 2]) w[i / 2].order(i & 1); });
            lap(0, p0);
            p0 = std::chrono::steady_clock::now();
            szi_run_parallel(ntr, [&](int j) { if (ok[j]) ok[j] = w[j].book(); });
            lap(1, p0);
            p0 = std::chrono::steady_clock::now();
            szi_run_parallel(2 * ntr, [&](int i) { if (ok[i / 2]) w[i / 2].bits(i & 1); });
            lap(2, p0);
            p0 = std::chrono::steady_clock::now();
        }
        // (round 6) the exact Lorenzo trials of a 1-D array (tune_interp_lorenzo asks for them with the first group) are tasks of this batch:
        // the host's walk over the sampled blocks + zstd, 0.18 ms each, beside the interpolation trials instead of behind them
        const int lz_n = slot0 == 0 ? ((ctx->lz_want & 1) ? 1 : 0) + ((ctx->lz_want & 2) ? 1 : 0) : 0;
        szi_run_parallel(ntr + lz_n, [&](int j) {
            if (j >= ntr) {
                const int which = (j - ntr == 0 && (ctx->lz_want & 1)) ? 0 : 1;
                std::vector<uint8_t> buf;
                const bool made = stock::lorenzo_trial_buffer<T>(eb, which ? 8192 : radius, (const T *)ctx->h_samples, per, nb, buf);
                const size_t z = made ? szi_zstd_size(buf.data(), buf.size()) : 0;
                ctx->lz_bytes[which] = z ? (double)(z + 8) : 0.0;
                return;
            }
            if (!ok[j]) return;
            if (!phases) {  // a trial from end to end on one thread of the pool: its 1.2 MB of codes stay in that core's cache
                w[j].order(0);
                w[j].order(1);
                if (!(ok[j] = w[j].book())) return;
                w[j].bits(0);
                w[j].bits(1);
            }
            std::vector<uint8_t> raw;
            w[j].finish(raw);
            const size_t z = szi_zstd_size(raw.data(), raw.size());
            sizes[j] = z ? z + 8 : 0;  // (Lossless_zstd::compress: the length word in front of the frame)
        });
        lap(3, p0);
        if (tt) fprintf(stderr, "[sz3hip tuner] steps: order %.3f, book %.3f, bits %.3f, buffer + zstd %.3f ms\n", ph[0], ph[1], ph[2], ph[3]);
    };
    if (ctx->dtype == SZ3HIP_FLOAT) priced_as(0.0f);
    else priced_as(0.0);
    if (tt)
        fprintf(stderr, "[sz3hip tuner] exact group of %d: kernels + copy out %.3f ms, pricing %.3f ms\n", ntr, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    for (int j = 0; j < ntr; j++)
        if (!sizes[j]) {
            // a trial that cannot be priced this way (an anchor stride the emission-order geometry does not take, no libzstd): this tuning goes on
            // with the device-side estimates, which the cost launch above has made for every trial of the group
            ctx->exact_now = false;
            return 0;
        }
    for (int j = 0; j < ntr; j++) ctx->exact_bytes[slot0 + j] = (double)sizes[j];
    if (slot0 == 0) ctx->lz_have = ctx->lz_want;
    return 0;
}
static int tuner_fetch(sz3hip_ctx *ctx, hipStream_t s) {
    HIPCHK(hipMemcpyAsync(ctx->h_trial, ctx->d_trial, SZ_TRIAL_WORDS * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}
// profiling_block / sample_blocks on an array that still lies in HOST memory (szi_pretune_host: the tuner beside the array's copy in) — the
// device kernels' arithmetic (k_profile_blocks, k_gather_blocks): the strided samples' plain minimum / maximum in T, blocks copied element by element
template <typename T>
static uint64_t host_profile_blocks(const T *data, int N, const uint64_t *dims, uint64_t bs, uint64_t stride, double abseb, uint8_t *flags) {
    uint64_t off[4] = {0, 0, 0, 0}, cnt[4] = {1, 1, 1, 1}, total = 1;
    off[N - 1] = 1;
    for (int i = N - 2; i >= 0; i--) off[i] = off[i + 1] * dims[i + 1];
    for (int i = 0; i < N; i++) {
        if (dims[i] < bs) return 0;
        cnt[i] = (dims[i] - bs + bs - 1) / bs;
        total *= cnt[i];
    }
    if (!total) return 0;
    if (!stride) stride = bs;
    const uint64_t m = bs / stride + 1;
    uint64_t npts = 1;
    for (int j = 0; j < N; j++) npts *= m;
    const int parts = (int)std::min<uint64_t>(16, std::max<uint64_t>(1, total / 256));
    szi_run_parallel(parts, [&](int part) {
        for (uint64_t t = total * part / parts; t < total * (part + 1) / parts; t++) {
            uint64_t r = t, start = 0;
            for (int j = N - 1; j >= 0; j--) {
                start += (r % cnt[j]) * bs * off[j];
                r /= cnt[j];
            }
            T mn = data[start], mx = mn;
            for (uint64_t q = 0; q < npts; q++) {
                uint64_t rr = q, idx = start;
                for (int j = N - 1; j >= 0; j--) {
                    idx += (rr % m) * stride * off[j];
                    rr /= m;
                }
                const T v = data[idx];
                if (v < mn) mn = v;
                if (v > mx) mx = v;
            }
            flags[t] = (mx - mn > abseb) ? 1 : 0;
        }
    });
    return total;
}
template <typename T>
static void host_gather_blocks(const T *data, int N, const uint64_t *dims, uint64_t edge, const uint64_t *starts, uint64_t nb, T *out) {
    uint64_t off[4] = {0, 0, 0, 0}, per = 1;
    off[N - 1] = 1;
    for (int i = N - 2; i >= 0; i--) off[i] = off[i + 1] * dims[i + 1];
    for (int i = 0; i < N; i++) per *= edge;
    const uint64_t rows = per / edge;  // runs of `edge` consecutive elements
    for (uint64_t b = 0; b < nb; b++) {
        const uint64_t *st = starts + b * 4;
        T *o = out + b * per;
        for (uint64_t rw = 0; rw < rows; rw++) {
            uint64_t r = rw, idx = st[N - 1];
            for (int j = N - 2; j >= 0; j--) {
                idx += (st[j] + r % edge) * off[j];
                r /= edge;
            }
            memcpy(o + rw * edge, data + idx, edge * sizeof(T));
        }
    }
}
// fills `conf` like the reference does before its final compress call: cmprAlgo becomes ALGO_INTERP (interpAlgo,
// interpDirection, interpAlpha, interpBeta tuned) or ALGO_LORENZO_REG (1-D only)
static int tune_interp_lorenzo(sz3hip_ctx *ctx, sz3hip_config &conf, const void *d_in, double eb, int radius, hipStream_t s, const void *h_in = nullptr) {
    const int N = conf.N;
    const size_t tsz = ctx->dtype == SZ3HIP_FLOAT ? 4 : 8;
    sz3hip_tuner_report &rep = ctx->tuner;
    memset(&rep, 0, sizeof(rep));
    rep.use_interp = 1;
    {   // sz3hip_ctx_set_tuner_exact, or — for contexts nobody holds a handle of: the host API's, the CLI's, the HDF5 filter's — the environment
        const char *te = getenv("SZ3HIP_TUNER_EXACT");
        ctx->exact_now = ctx->tuner_exact == 1 || (ctx->tuner_exact == 0 && (te ? atoi(te) != 0 : ctx->exact_default));
    }
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
    if (h_in) {  // (the array is still on its way to the device: the candidates' strided samples read where it lies)
        total = ctx->dtype == SZ3HIP_FLOAT ? host_profile_blocks<float>((const float *)h_in, N, conf.dims, sbs, sbs / 4, eb, ctx->d_flags)
                                           : host_profile_blocks<double>((const double *)h_in, N, conf.dims, sbs, sbs / 4, eb, ctx->d_flags);
    } else {
        rc = szk_launch_profile_blocks(ctx->dtype, d_in, N, conf.dims, sbs, sbs / 4, eb, ctx->d_flags, &total, s);
        if (rc) return fail(SZ3HIP_EHIP, "tuner: profiling launch failed (%d)", rc);
        if (total) HIPCHK(hipStreamSynchronize(s));
    }
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
    ctx->h_samples_valid = false;  // (exact pricing: this call's sample blocks are fetched with the first group's codes — or gathered on the host right here)
    if (h_in) {
        const size_t sbytes = (size_t)sampling_num * tsz;
        if (ctx->h_samples_cap < sbytes) {
            if (ctx->h_samples) (void)hipHostFree(ctx->h_samples);
            ctx->h_samples = nullptr;
            ctx->h_samples_cap = 0;
            HIPCHK(hipHostMalloc(&ctx->h_samples, sbytes));
            ctx->h_samples_cap = sbytes;
        }
        if (ctx->dtype == SZ3HIP_FLOAT) host_gather_blocks<float>((const float *)h_in, N, conf.dims, sbs + 1, starts, nb, (float *)ctx->h_samples);
        else host_gather_blocks<double>((const double *)h_in, N, conf.dims, sbs + 1, starts, nb, (double *)ctx->h_samples);
        HIPCHK(hipMemcpyAsync(ctx->d_samples, ctx->h_samples, sbytes, hipMemcpyHostToDevice, s));
        ctx->h_samples_valid = true;
    } else {
        rc = szk_launch_gather_blocks(ctx->dtype, d_in, N, conf.dims, sbs + 1, ctx->d_starts, (uint32_t)nb, ctx->d_samples, s);
        if (rc) return fail(SZ3HIP_EHIP, "tuner: gather launch failed (%d)", rc);
    }

    const double raw = (double)sampling_num * (double)tsz;
    auto priced = [&](int slot) { return ctx->exact_now ? ctx->exact_bytes[slot] : trial_bytes(ctx->h_trial + 4 * slot, tsz); };
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
    ctx->lz_have = 0;
    ctx->lz_want = (N == 1 && ctx->exact_now) ? (1 | ((conf.relErrorBound < 1.01e-6 && lorenzo_config.quantbinCnt != 16384) ? 2 : 0)) : 0;
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
    // (round 6) 1-D arrays, device-side pricing: the Lorenzo trials the decision may ask for below — at the call's radius and, where the rule
    // of SZAlgoInterp.hpp:268-277 can apply, at 8192 — are enqueued behind the interpolation group and come back with ITS fetch: one
    // synchronisation instead of three (a copy, the host's wake-up and the next launch's latency each: ~20 us of an idle GPU, 40 of the
    // 245 us of a 4 MB series). They cost ~15 us of kernels when the decision then does not ask (interpolation beyond ratio 50).
    const bool lz_batched = N == 1 && !ctx->exact_now;
    const bool lz_second = lz_batched && conf.relErrorBound < 1.01e-6 && lorenzo_config.quantbinCnt != 16384;
    auto lorenzo_enqueue = [&](int rad, int base) -> int {
        HIPCHK(clear_hist_counters(ctx, s));
        int rl = szk_launch_trial_lorenzo12(ctx->dtype == SZ3HIP_FLOAT ? 0 : 1, ctx->d_samples, per, nb, eb, rad, ctx->d_hist, ctx->d_counters, ctx->d_trial + base + 4, s);
        if (rl) return fail(SZ3HIP_EHIP, "tuner: Lorenzo trial launch failed (%d)", rl);
        rl = szk_launch_code_cost(ctx->d_hist, ctx->d_counters, ctx->d_trial + base, 1, sampling_num, 0, s);
        if (rl) return fail(SZ3HIP_EHIP, "tuner: cost kernel launch failed (%d)", rl);
        return 0;
    };
    // a Lorenzo trial's bytes from its eight result words: the coded size + the choices' own cost (ComposedPredictor::save: the selection
    // vector Huffman-coded, ComposedPredictor.hpp:52-64): its entropy
    auto lorenzo_bytes = [&](const uint64_t *w) -> double {
        double bytes = trial_bytes(w, tsz);
        const double nblk = (double)w[5], n2 = (double)w[4];
        if (nblk > 0 && n2 > 0 && n2 < nblk) {
            const double p2 = n2 / nblk;
            bytes += nblk * -(p2 * std::log2(p2) + (1 - p2) * std::log2(1 - p2)) / 8.0;
        }
        return bytes;
    };
    uint64_t lz_words[16] = {0};
    if (lz_batched) {  // (the group's launch zeroed every result word)
        rc = lorenzo_enqueue(radius, 32);
        if (rc) return rc;
        if (lz_second) {
            rc = lorenzo_enqueue(8192, 40);
            if (rc) return rc;
        }
    }
    rc = tuner_fetch(ctx, s);
    if (rc) return rc;
    if (lz_batched) memcpy(lz_words, ctx->h_trial + 32, sizeof(lz_words));  // (a second interpolation group zeroes the result words again)
    double dir_bytes[2];
    for (int op = 0; op < 2; op++) {
        rep.est_bytes[op] = priced(op);
        dir_bytes[op] = priced(2 + op);
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
        rep.est_bytes[3 + i] = priced(ab_slot + i);
        const double ratio = raw / rep.est_bytes[3 + i];
        if (ratio > best_interp * 1.02) {
            best_interp = ratio;
            conf.interpAlpha = alphas[i];
            conf.interpBeta = betas[i];
        }
    }
    // the 1-D Lorenzo trial (lorenzo_compress_test, :80-120) at a given radius: this library's trial kernel and the device-side estimate, or —
    // exact pricing — the reference's own walk on the host over the sampled blocks, its buffer, zstd
    auto lorenzo_price = [&](int rad, double &bytes) -> int {
        // (round 4: the reference's own trial geometry — every sample block an array of its own, blocks of five values, Lorenzo-1 or
        // Lorenzo-2 per block — instead of first-order Lorenzo over the concatenated samples: on smooth 1-D series the second-order
        // blocks are what makes Lorenzo win, and the set chosen here is the one stage 1 then codes the array with)
        if (ctx->exact_now && ctx->h_samples_valid) {
            static const bool ttl = getenv("SZ3HIP_TUNER_TIMING") != nullptr;
            const auto tl0 = std::chrono::steady_clock::now();
            const int which = rad == radius ? 0 : 1;
            if ((ctx->lz_have >> which) & 1) {  // (priced beside the first interpolation group)
                if (ctx->lz_bytes[which] <= 0) return fail(SZ3HIP_EZSTD, "tuner: the Lorenzo trial could not be priced");
                bytes = ctx->lz_bytes[which];
                return 0;
            }
            std::vector<uint8_t> buf;
            const bool made = ctx->dtype == SZ3HIP_FLOAT ? stock::lorenzo_trial_buffer<float>(eb, rad, (const float *)ctx->h_samples, per, nb, buf)
                                                         : stock::lorenzo_trial_buffer<double>(eb, rad, (const double *)ctx->h_samples, per, nb, buf);
            const size_t z = made ? szi_zstd_size(buf.data(), buf.size()) : 0;
            if (!z) return fail(SZ3HIP_EZSTD, "tuner: the Lorenzo trial could not be priced");
            bytes = (double)(z + 8);
            if (ttl) fprintf(stderr, "[sz3hip tuner] exact Lorenzo trial (radius %d): %.3f ms\n", rad, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count());
            return 0;
        }
        if (lz_batched) {  // (priced behind the first interpolation group)
            bytes = lorenzo_bytes(lz_words + (rad == radius ? 0 : 8));
            return 0;
        }
        HIPCHK(hipMemsetAsync(ctx->d_trial + 24, 0, 64, s));
        int rl = lorenzo_enqueue(rad, 24);
        if (rl) return rl;
        rl = tuner_fetch(ctx, s);
        if (rl) return rl;
        bytes = lorenzo_bytes(ctx->h_trial + 24);
        return 0;
    };
    if (N == 1 && best_interp < 50) {  // :232-247
        rc = lorenzo_price(radius, rep.est_bytes[6]);
        if (rc) return rc;
        best_lorenzo = raw / rep.est_bytes[6];
    }
    rep.ran = 1;
    const bool use_interp = !(best_lorenzo >= best_interp * 1.1 && best_lorenzo < 50 && best_interp < 50);  // :249-250
    rep.use_interp = use_interp;
    if (use_interp) {
        conf.cmprAlgo = SZ3HIP_ALGO_INTERP;
    } else {
        // SZAlgoInterp.hpp:233-240, 282: the set the trial priced — Lorenzo-1 + Lorenzo-2, no regression — and, through setDims, the
        // default block size of a 1-D array again (the trial's blocks of five do not survive it)
        lorenzo_config.cmprAlgo = SZ3HIP_ALGO_LORENZO_REG;
        lorenzo_config.lorenzo = 1;
        lorenzo_config.lorenzo2 = 1;
        lorenzo_config.regression = 0;
        lorenzo_config.regression2 = 0;
        lorenzo_config.blockSize = 128;
        // :268-277 — a narrower quantizer (16384 bins) when the bound is not a tiny relative one and Lorenzo compresses at all: kept when
        // its trial is 2 % better. The caller takes the radius from what comes back in quantbinCnt.
        if (conf.relErrorBound < 1.01e-6 && best_lorenzo > 5 && lorenzo_config.quantbinCnt != 16384) {
            rc = lorenzo_price(8192, rep.est_bytes[7]);
            if (rc) return rc;
            if (raw / rep.est_bytes[7] > best_lorenzo * 1.02) {
                best_lorenzo = raw / rep.est_bytes[7];
                lorenzo_config.quantbinCnt = 16384;
            }
        }
        conf = lorenzo_config;
    }
    rep.interpAlgo = conf.interpAlgo;
    rep.interpDirection = conf.interpDirection;
    rep.interpAlpha = conf.interpAlpha;
    rep.interpBeta = conf.interpBeta;
    return 0;
}
void szi_ctx_exact_default(sz3hip_ctx *ctx, int on) { ctx->exact_default = on != 0; }
// 1: the last stage 1 was the default algorithm's and its tuner took Lorenzo (1-D arrays only); *quantbinCnt: the quantizer it ended with
int szi_tuner_took_lorenzo(sz3hip_ctx *ctx, int *quantbinCnt) {
    if (!ctx->tuner.ran || ctx->tuner.use_interp || !ctx->stage1_done) return 0;
    *quantbinCnt = (int)ctx->proto.radius * 2;
    return 1;
}
extern "C" int sz3hip_get_tuner_report(sz3hip_ctx *ctx, sz3hip_tuner_report *rep) {
    *rep = ctx->tuner;
    return 0;
}

static bool same_call(const sz3hip_config &a, const sz3hip_config &b) {
    if (a.N != b.N || a.num != b.num || a.cmprAlgo != b.cmprAlgo || a.errorBoundMode != b.errorBoundMode || a.absErrorBound != b.absErrorBound ||
        a.quantbinCnt != b.quantbinCnt || a.interpAnchorStride != b.interpAnchorStride || a.relErrorBound != b.relErrorBound)
        return false;
    for (int i = 0; i < a.N; i++)
        if (a.dims[i] != b.dims[i]) return false;
    return true;
}
void szi_pretune_cancel(sz3hip_ctx *ctx) { ctx->pre_valid = false; }
int szi_pretune_host(sz3hip_ctx *ctx, const sz3hip_config *conf_in, const void *h_in) {
    ctx->pre_valid = false;
    if (conf_in->cmprAlgo != SZ3HIP_ALGO_INTERP_LORENZO || conf_in->errorBoundMode != SZ3HIP_EB_ABS || conf_in->N < 1 || conf_in->N > 4) return -1;
    const double eb = conf_in->absErrorBound;
    const int radius = conf_in->quantbinCnt / 2;
    if (!(eb > 0) || !isfinite(eb) || radius < 2 || radius > 32768 || conf_in->num > ctx->max_n || conf_in->num == 0) return -1;
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->side) {  // (the side stream always comes with its two events: the other users test the stream alone)
        HIPCHK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    sz3hip_config conf = *conf_in;
    const sz3hip_tuner_report keep = ctx->tuner;
    const int rc = tune_interp_lorenzo(ctx, conf, nullptr, eb, radius, ctx->side, h_in);
    ctx->pre_report = ctx->tuner;
    ctx->tuner = keep;
    if (rc) return rc;
    ctx->pre_conf = conf;
    ctx->pre_key = *conf_in;
    ctx->pre_valid = true;
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
    int radius = conf->quantbinCnt / 2;  // api/impl/SZAlgoLorenzoReg.hpp:72
    if (radius < 2 || radius > 32768) return fail(SZ3HIP_EINVAL, "quantbinCnt must be in [4, 65536]");
    ctx->cur_out_cap = ctx->force_out_cap ? ctx->force_out_cap : std::min<uint64_t>(ctx->out_cap, std::max<uint64_t>(1024, num / 32));
    memset(&ctx->tuner, 0, sizeof(ctx->tuner));
    for (int i = 0; i < ST_COUNT; i++) ctx->ev_used[i] = false;  // stage times describe this call only
    ctx->copy_ahead = false;
    ctx->range_ready = false;
    ctx->s1_spec = ctx->seg_expected = ctx->s1_assumed_narrow = ctx->s1_fused = ctx->s1_samp = ctx->s1_samp_in = false;
    ctx->blk_spec = false;
    ctx->blk_cleared_now = ctx->blk_pre_cleared && ctx->blk_pre_stream == s && !ctx->hist_exposed;  // (this call only: whatever follows starts from memsets again)
    ctx->blk_pre_cleared = false;
    ctx->fold_rows = 0;
    ctx->s1_conf = *conf_in;
    ctx->s1_in = d_in;
    prof_begin(ctx, ST_SPAN, s);  // (closed at the end of stage 2: the device time of the whole step)
    const bool pretuned = conf->cmprAlgo == SZ3HIP_ALGO_INTERP_LORENZO && ctx->pre_valid && same_call(ctx->pre_key, *conf_in);
    ctx->pre_valid = false;  // (an outcome is for the very next call, or for none)
    if (pretuned) {  // szi_pretune_host tuned this call from the host's copy of the array while it was on its way here
        *conf = ctx->pre_conf;
        ctx->tuner = ctx->pre_report;
        radius = conf->quantbinCnt / 2;
        ctx->spec_valid = ctx->tuner.ran && conf->cmprAlgo == SZ3HIP_ALGO_INTERP;
        if (ctx->spec_valid) ctx->spec_conf = *conf;
    }
    if (conf->cmprAlgo == SZ3HIP_ALGO_INTERP_LORENZO) {  // the reference's default: sampling auto-tuner, then one of the two paths
        // The tuner is a chain of small launches and host round trips (the chip is mostly idle), and its outcome is almost
        // always interpolation, which starts from a working copy of the input: make that copy meanwhile on a side stream.
        // (3-D arrays run through the level kernels, which read the input where it lies: no copy)
        szk_interp_params shape;
        memset(&shape, 0, sizeof(shape));
        shape.N = conf->N;
        for (int i = 0; i < conf->N && i < 4; i++) shape.dims[i] = conf->dims[i];
        // (speculation, below: stage 1 itself runs beside the tuner; not for 1-D arrays, whose tuner shares the histogram with it)
        bool spec = ctx->spec_valid && !(szk_dbg_flags & 131072) && conf->N >= 2 && ctx->spec_conf.N == conf->N;
        for (int i = 0; spec && i < conf->N; i++) spec = ctx->spec_conf.dims[i] == conf->dims[i];
        const bool ahead = !szk_interp_levels_ok(&shape) && !spec;
        if (!ctx->d_work) HIPCHK(hipMalloc(&ctx->d_work, ctx->max_n * (ctx->dtype == SZ3HIP_FLOAT ? 4 : 8)));
        if (ahead) {
            if (!ctx->side) {
                HIPCHK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
            }
            HIPCHK(hipEventRecord(ctx->ev_fork, s));
            HIPCHK(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
            HIPCHK(hipMemcpyAsync(ctx->d_work, d_in, num * (ctx->dtype == SZ3HIP_FLOAT ? 4 : 8), hipMemcpyDeviceToDevice, ctx->side));
            HIPCHK(hipEventRecord(ctx->ev_join, ctx->side));
        }
        // Speculation: the tuner is 0.2 ms (f64 / 4-D: 1-3 ms) of small launches and host round trips during which the chip
        // idles, and on a series of similar arrays its outcome repeats. A context that has a previous outcome for this shape
        // starts stage 1 with it on the caller's stream and runs the tuner beside it on the side stream; the tuner's outcome
        // decides as ever — when it differs, stage 1 is enqueued again with it (the speculative launches run out first: wasted
        // time, same payload; tests/test_gpu_stages.py::test_speculative_stage1_follows_the_tuner).
        int rc_spec = 0;
        hipStream_t ts = s;  // the stream the tuner runs on
        if (spec) {
            if (!ctx->side) {
                HIPCHK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
            }
            HIPCHK(hipEventRecord(ctx->ev_fork, s));  // (the input is ready on the side stream when it is on the caller's)
            HIPCHK(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
            sz3hip_config sc = *conf;
            sc.cmprAlgo = SZ3HIP_ALGO_INTERP;
            sc.interpAlgo = ctx->spec_conf.interpAlgo;
            sc.interpDirection = ctx->spec_conf.interpDirection;
            sc.interpAlpha = ctx->spec_conf.interpAlpha;
            sc.interpBeta = ctx->spec_conf.interpBeta;
            sc.interpAnchorStride = ctx->spec_conf.interpAnchorStride;
            ctx->spec_used = sc;
            HIPCHK(clear_hist_counters(ctx, s));
            rc_spec = stage1_interp(ctx, &sc, d_in, eb, radius, num, s);
            ts = ctx->side;
        }
        prof_begin(ctx, ST_TUNER, ts);
        int rct = tune_interp_lorenzo(ctx, *conf, d_in, eb, radius, ts);
        prof_end(ctx, ST_TUNER, ts);
        if (spec) {
            hipError_t ej = hipEventRecord(ctx->ev_join, ctx->side);
            if (ej == hipSuccess) ej = hipStreamWaitEvent(s, ctx->ev_join, 0);  // the caller's stream owns d_in again
            if (ej != hipSuccess) {
                (void)hipStreamSynchronize(ctx->side);
                return fail(SZ3HIP_EHIP, "joining the side stream failed: %s", hipGetErrorString(ej));
            }
        }
        if (ahead) {
            hipError_t ej = hipStreamWaitEvent(s, ctx->ev_join, 0);  // whatever the outcome: the caller's stream owns d_in again
            if (ej != hipSuccess) {
                (void)hipStreamSynchronize(ctx->side);
                return fail(SZ3HIP_EHIP, "joining the side stream failed: %s", hipGetErrorString(ej));
            }
        }
        if (rct) {
            ctx->stage1_done = false;  // (a speculative stage 1 may be on its way: its results are not this call's)
            ctx->spec_valid = false;
            return rct;
        }
        ctx->copy_ahead = ahead;
        radius = conf->quantbinCnt / 2;  // (the tuner's Lorenzo outcome may have narrowed the quantizer, SZAlgoInterp.hpp:268-277)
        // what the next call of this context may start from
        ctx->spec_valid = ctx->tuner.ran && conf->cmprAlgo == SZ3HIP_ALGO_INTERP;
        if (ctx->spec_valid) ctx->spec_conf = *conf;
        if (spec) {
            const sz3hip_config &p0 = ctx->spec_conf;  // (= this call's outcome when valid)
            const bool hit = ctx->spec_valid && rc_spec == 0 && p0.interpAlgo == ctx->spec_used.interpAlgo &&
                             p0.interpDirection == ctx->spec_used.interpDirection && p0.interpAlpha == ctx->spec_used.interpAlpha &&
                             p0.interpBeta == ctx->spec_used.interpBeta && p0.interpAnchorStride == ctx->spec_used.interpAnchorStride;
            ctx->tuner.speculated = hit ? 1 : 2;
            if (hit) return 0;  // stage 1 is already on its way
        }
    }
    HIPCHK(clear_hist_counters(ctx, s));
    if (conf->cmprAlgo == SZ3HIP_ALGO_INTERP || conf->cmprAlgo == SZ3HIP_ALGO_HIP_INTERP)
        return stage1_interp(ctx, conf, d_in, eb, radius, num, s);
    if (conf->cmprAlgo == SZ3HIP_ALGO_LORENZO_REG) {
        // the predictor set of make_compressor_lorenzo_regression (api/impl/SZAlgoLorenzoReg.hpp:28-64). Lorenzo-1 alone is
        // the plain stream; anything with Lorenzo-2 or regression is the block-composed one, built for 3-D arrays with
        // block edges 4..8. Elsewhere the set falls back to its Lorenzo-1 member (the stream's header says which predictor
        // coded it, the host API clears the flags it did not honour in the trailer) and is refused when it has none.
        const uint32_t mask = (conf->lorenzo ? 1u : 0u) | (conf->lorenzo2 ? 2u : 0u) | (conf->regression ? 4u : 0u);
        if (mask == 0) return fail(SZ3HIP_EINVAL, "All lorenzo and regression methods are disabled.");
        if (mask != 1u) {
            if (blk_shape_ok(conf) && (conf->N != 4 || !(mask & 2u)) && !(szk_dbg_flags & 16384)) {  // (the reference has no second-order Lorenzo for N = 4: LorenzoPredictor.hpp:92)
                bool all_lorenzo = false;
                const int rcs = blk_all_lorenzo(ctx, conf, d_in, eb, radius, mask, s, &all_lorenzo);
                if (rcs) return rcs;
                if (!all_lorenzo) return stage1_blocks(ctx, conf, d_in, eb, radius, num, mask, s);
                return stage1_lorenzo(ctx, conf, d_in, eb, radius, num, s);  // (the selection chose first-order Lorenzo throughout)
            }
            if (!(mask & 1u))
                return fail(SZ3HIP_EUNSUPPORTED, "regression is built for 1-D (blockSize 4..65535), 2-D (4..32), 3-D (4..8) and 4-D arrays (4..6), 2nd-order "
                                                 "Lorenzo for 1-D, 2-D and 3-D ones (got N = %d, blockSize = %d)", conf->N, conf->blockSize);
        }
    }
    return stage1_lorenzo(ctx, conf, d_in, eb, radius, num, s);
}

enum { S2_CLASSIC = 0, S2_SPEC = 1, S2_REENCODE = 2, S2_SPEC_WIDE = 3, S2_SAMPLED = 4 };
static int stage2_launch(sz3hip_ctx *ctx, void *d_payload, size_t cap, hipStream_t s, int how);
extern "C" int sz3hip_compress_stage2(sz3hip_ctx *ctx, void *d_payload, size_t cap, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->stage1_done) return fail(SZ3HIP_EINVAL, "stage2 called before stage1");
    if (ctx->stage2_done) return fail(SZ3HIP_EINVAL, "stage2 called twice for one stage1 (finish the call first)");
    const uint64_t n = ctx->proto.n;
    {   // what this call can need: the shape-blind bound, or — block-composed stream of a thin array — the bound for its own block count
        const uint64_t lists = std::max<uint64_t>(ctx->out_cap, ctx->cur_out_cap);
        size_t need = payload_bound_n(n, lists);
        if (ctx->proto.predictor == 2) {
            uint64_t nb = 1;
            for (int i = 0; i < 4; i++) nb *= (ctx->proto.dims[i] + ctx->proto.interp_id - 1) / ctx->proto.interp_id;  // (dims[0]: 1, or a 4-D array's slowest extent)
            need = std::max(need, payload_bound_blocks(n, lists, nb));
        }
        if (cap < need) return fail(SZ3HIP_ECAPACITY, "The buffer for compressed data is not large enough.");
    }
    ctx->s2_payload = d_payload;
    ctx->s2_cap = cap;
    // Speculation: the code book of a series of similar arrays repeats (it is a function of the code lengths alone), and building
    // it is a serial chain the chip idles through. A context whose previous call left a book for the same predictor and radius
    // packs with THAT book while this call's is built from this call's histogram — by a workgroup of the packer's own launch
    // (alphabets <= 256 symbols) or on a stream of its own (wide ones); a verdict compares the two and finish() repeats the
    // encoder when they differ. The book a payload is coded with is always the one its own histogram gives.
    bool spec = ctx->proto.predictor == 0 ? ctx->s1_spec : book_spec_ok(ctx, ctx->proto.predictor, ctx->proto.radius);
    // The one-stream form: small alphabet, short outlier lists, range words kept by stage 1 (what a smooth field's previous
    // call left).
    // (Not for block streams — predictor 2: their side section is copied by the assembly's list workgroups, which the sort roles replace;
    // a context that went through the histogram exchange keeps hist_reduced set, which would open this gate for them.)
    if (spec && !(ctx->cb_hint == 0 && !ctx->lists_long && (ctx->range_ready || ctx->hist_reduced) && !ctx->hist_exposed && ctx->proto.predictor != 2))
        spec = false;
    // Wide alphabets (round 3, second attempt): the book needs a compute unit's whole LDS and 0.15 - 0.3 ms of ONE workgroup —
    // as long as the encoder's two passes take. It is built on a high-priority stream of its own, forked behind stage 1 and
    // enqueued BEFORE the encoder's launches (the workgroup is placed before the persistent packers fill the chip), and joined in
    // front of a one-workgroup verdict on the caller's stream; the list sorts ride in the packer's launch as in the small form.
    // (Lists too long for the packer's sort roles — C3's 4096 anchors — were tried with a sort launch of their own in front of the
    // encoder: C3 1.253 against 1.247 ms without, 1.33 with alternating fields. Not taken.)
    // (Not for block streams: their side section is copied by the assembly's list workgroups, which the sort roles replace.)
    const bool spec_wide = !spec && book_spec_ok(ctx, ctx->proto.predictor, ctx->proto.radius) && ctx->cb_part == 1 && !ctx->lists_long &&
                           ctx->proto.predictor != 2 && !(szk_dbg_flags & 4096);
    if (ctx->s1_fused && !spec) {
        // (cannot happen: the fused stage 1 is taken under the conditions of this stage's one-stream form; should they ever drift apart,
        // there is no code array to encode from — stage 1 once more, in the two-pass form)
        const int spec_was = ctx->spec_off;
        ctx->spec_off = 1;
        sz3hip_config conf = ctx->s1_conf;
        int rc1 = sz3hip_compress_stage1(ctx, &conf, ctx->s1_in, stream);
        ctx->spec_off = spec_was;
        if (rc1) return rc1;
    }
    // the sampled book (round 6) is in its slot when stage 1 ends: with the segments' sums made by stage 1 and short lists (the packer's
    // sort roles) stage 2 is the segment pass, the scan and the packer; otherwise the classic form, whose code-book launch then only
    // sorts the lists
    const bool samp_roles = ctx->proto.predictor == 0 && ctx->s1_samp_in && ctx->seg_expected && !ctx->lists_long;
    int rc = stage2_launch(ctx, d_payload, cap, s, samp_roles ? S2_SAMPLED : (ctx->proto.predictor == 0 && ctx->s1_samp) ? S2_CLASSIC : spec ? S2_SPEC : (spec_wide ? S2_SPEC_WIDE : S2_CLASSIC));
    if (rc) return rc;
    ctx->stage2_done = true;
    return 0;
}
static int stage2_launch(sz3hip_ctx *ctx, void *d_payload, size_t cap, hipStream_t s, int how) {
    const uint64_t n = ctx->proto.n;
    // the slot this call's book goes to (the other one holds the previous call's), and the slot the encoder reads
    const int fresh = how == S2_REENCODE ? ctx->book_pending : (ctx->book_idx < 0 ? 0 : 1 - ctx->book_idx);
    const bool wide = how == S2_SPEC_WIDE;
    const int used = (how == S2_SPEC || wide) ? ctx->book_idx : fresh;
    ctx->book_pending = fresh;
    ctx->s2_spec = how == S2_SPEC || wide;
    ctx->s2_wide = wide;
    ctx->s2_samp = how == S2_SAMPLED;
    const uint32_t *samp_words = ctx->proto.predictor == 0 && ctx->s1_samp ? reinterpret_cast<const uint32_t *>(ctx->d_counters + 16) : nullptr;
    szk_cb_params cb;
    cb_params_from(ctx, cb, ctx->cur_out_cap, fresh);
    cb.samp_words = samp_words;
    // Speculative form: this call's book, the verdict and the list sorts ride in the packer's own launch as three role
    // workgroups, the fold of stage 1's histogram rows in the scan's launch — one stream, two launches.
    const bool fused = how == S2_SPEC;
    szk_encode_roles er;
    memset(&er, 0, sizeof(er));
    if (fused) {
        if (!cb.range_ready) {  // (the histogram was summed over the ranks since stage 1: the range of the SUMMED alphabet)
            HIPCHK(hipMemsetAsync(ctx->d_counters + 8, 0, 16, s));
            if (szk_launch_hist_range(ctx->d_hist, cb.range, s)) return fail(SZ3HIP_EHIP, "histogram range launch failed");
            cb.range_ready = 1;
        }
        cb.part_hint = 0;
        er.roles = 1;
        er.hist = ctx->d_hist;
        er.cb = &cb;
        er.used_lens = ctx->bk[used].lens;
        er.flags = reinterpret_cast<uint32_t *>(ctx->d_counters + 10);
        er.exact = ctx->spec_exact;
        if (ctx->fold_rows) {
            er.fold_partial = ctx->d_hist_partial;
            er.fold_rows = ctx->fold_rows;
            er.fold_hist = ctx->d_hist;
            er.fold_range = ctx->fold_range;
            ctx->fold_rows = 0;
        }
    } else if (ctx->fold_rows) {  // (stage 1 deferred the fold and stage 2 does not speculate after all: the range words were not kept)
        if (szk_launch_hist_fold(ctx->d_hist_partial, ctx->fold_rows, (int)ctx->proto.radius, ctx->d_hist, ctx->fold_range, s))
            return fail(SZ3HIP_EHIP, "histogram fold launch failed");
        ctx->fold_rows = 0;
    }
    if (wide) {
        if (!ctx->book_stream) {
            int lo_pri = 0, hi_pri = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri);
            HIPCHK(hipStreamCreateWithPriority(&ctx->book_stream, hipStreamNonBlocking, hi_pri));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_s1, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_book, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(ctx->ev_s1, s));
        HIPCHK(hipStreamWaitEvent(ctx->book_stream, ctx->ev_s1, 0));
        if (ctx->range_ready && !cb.range_ready) HIPCHK(hipMemsetAsync(ctx->d_counters + 8, 0, 16, ctx->book_stream));
        cb.part_hint = 1;   // the wide form alone: a small alphabet raises `mispredict` (the verdict passes it on, stage 2 is repeated)
        cb.skip_sort = 1;   // (the lists are sorted by the packer's launch: its assembly copies them)
        int rcb = szk_launch_codebook(ctx->d_hist, &cb, ctx->book_stream);
        if (rcb) return fail(SZ3HIP_EHIP, "codebook kernel launch failed (%d)", rcb);
        HIPCHK(hipEventRecord(ctx->ev_book, ctx->book_stream));
        er.roles = 1;
        er.no_book = 1;
        er.hist = ctx->d_hist;
        er.cb = &cb;
        er.used_lens = ctx->bk[used].lens;
        er.flags = reinterpret_cast<uint32_t *>(ctx->d_counters + 10);
        er.exact = ctx->spec_exact;
    }
    if (how == S2_SAMPLED) {  // the book is in its slot (stage 1 built it): the packer's launch carries the two sort roles, nothing else
        er.roles = 1;
        er.no_book = 1;
        er.hist = ctx->d_hist;
        er.cb = &cb;
        er.used_lens = ctx->bk[used].lens;
        er.flags = reinterpret_cast<uint32_t *>(ctx->d_counters + 10);
        er.exact = 1;
    }
    if (how == S2_CLASSIC) {
        if (ctx->range_ready && !cb.range_ready)  // stage 1 kept the range of the LOCAL histogram, the caller then changed it (all-reduce):
            HIPCHK(hipMemsetAsync(ctx->d_counters + 8, 0, 16, s));  // k_hist_range starts from zero
        prof_begin(ctx, ST_CODEBOOK, s);
        int rcb = szk_launch_codebook(ctx->d_hist, &cb, s);
        prof_end(ctx, ST_CODEBOOK, s);
        if (rcb) return fail(SZ3HIP_EHIP, "codebook kernel launch failed (%d)", rcb);
    }
    szk_layout_params lp;
    lp.proto = ctx->proto;
    lp.n_vout = ctx->d_counters + 0;
    lp.n_dout = ctx->d_counters + 1;
    lp.out_cap = ctx->cur_out_cap;
    lp.info = ctx->bk[used].info;
    lp.state = ctx->d_state;
    lp.side_bytes = ctx->proto.predictor == 2 ? ctx->d_blk_counters + 2 : nullptr;
    szk_asm_params ap;
    memset(&ap, 0, sizeof(ap));
    ap.n_vout = ctx->d_counters + 0;
    ap.n_dout = ctx->d_counters + 1;
    ap.out_cap = ctx->cur_out_cap;
    ap.t_is_32bit = ap.q_is_32bit = ctx->dtype == SZ3HIP_FLOAT;
    ap.state = ctx->d_state;
    ap.payload = (uint8_t *)d_payload;
    ap.cap = cap;
    ap.total_words = ctx->d_counters + 2;
    ap.lens = ctx->bk[used].lens;
    ap.chunk_words = ctx->d_chunk_words;
    ap.sub_bits = ctx->d_sub_bits;
    ap.vout_idx = ctx->d_vout_idx;
    ap.dout_idx = ctx->d_dout_idx;
    ap.vout_val = ctx->d_vout_val;
    ap.dout_val = ctx->d_dout_val;
    ap.side = ctx->proto.predictor == 2 ? ctx->d_blk_side : nullptr;
    ap.assumed_narrow = ctx->s1_assumed_narrow ? 1 : 0;
    ap.q16_flag = ctx->s1_q16 && ctx->proto.predictor == 0 ? reinterpret_cast<const uint32_t *>(ctx->d_counters + 11) + 1 : nullptr;
    ap.mode = ctx->mode;
    ap.samp_words = samp_words;
    szk_merge_args mg;
    memset(&mg, 0, sizeof(mg));
    const bool merge = fused && ctx->s1_fused && ctx->seg_expected;
    if (ctx->s1_fused && !merge) return fail(SZ3HIP_EHIP, "internal: fused stage 1 without the merging encoder");
    if (merge) {
        mg.slots = ctx->s1_slots;
        mg.seg_base = ctx->d_seg_base;
        mg.fuse_flag = reinterpret_cast<const uint32_t *>(ctx->d_counters + 11);
    }
    prof_begin(ctx, ST_ENCODE, s);  // (the payload layout is computed inside the encoder's scan launch, the sections are assembled by the packer's)
    int rc = szk_launch_encode(ctx->d_codes, n, ctx->bk[used].enc, ctx->bk[used].info, (int)ctx->proto.radius, ctx->mode, ctx->d_chunk_words,
                               ctx->d_chunk_off, ctx->d_counters + 2, ctx->d_state, (uint8_t *)d_payload, &lp, &ap, s,
                               (fused || (samp_words && ctx->s1_samp_in)) && ctx->seg_expected ? ctx->d_seg_bits : nullptr, reinterpret_cast<uint32_t *>(ctx->d_counters + 10) + 1,
                               (fused || wide || how == S2_SAMPLED) ? &er : nullptr, merge ? &mg : nullptr);
    prof_end(ctx, ST_ENCODE, s);
    if (rc) return fail(SZ3HIP_EHIP, "encode kernel launch failed (%d)", rc);
    if (wide) {  // join: this call's book against the one the encoder used
        HIPCHK(hipStreamWaitEvent(s, ctx->ev_book, 0));
        if (szk_launch_book_verdict(ctx->bk[fresh].info, ctx->bk[fresh].lens, ctx->bk[used].info, ctx->bk[used].lens,
                                    reinterpret_cast<const uint32_t *>(ctx->d_counters + 7), reinterpret_cast<const uint32_t *>(ctx->d_counters + 8),
                                    ctx->d_state, ctx->d_hist, ctx->spec_exact, s))
            return fail(SZ3HIP_EHIP, "verdict kernel launch failed");
    }
    prof_end(ctx, ST_SPAN, s);
    // the state block to the host, and — when the call turns out to need no repeat — the next call's histogram and counters zeroed,
    // in one small launch behind the packer's (k_publish); finish() polls the sequence word
    ctx->pub_zero = ctx->d_hist == ctx->d_hist_own && !ctx->hist_exposed && !(szk_dbg_flags & 268435456);
    ctx->pub_seq++;
    ctx->pub_blk_zero = ctx->pub_zero && ctx->d_blk_counters && ctx->d_blk_stats5;
    if (szk_launch_publish(ctx->d_state, ctx->h_state, ctx->h_pub_seq, ctx->pub_seq, ctx->pub_zero ? (void *)ctx->d_hist : nullptr,
                           SZH_HIST_BINS * 8 + SZ_COUNTER_BYTES, s, ctx->blk_spec ? blk_sel_count(ctx) : nullptr,
                           ctx->pub_blk_zero ? (void *)ctx->d_blk_counters : nullptr, ctx->pub_blk_zero ? ctx->d_blk_stats5 : nullptr))
        return fail(SZ3HIP_EHIP, "publish kernel launch failed");
    ctx->pub_stream = s;
    // (h_state->probe = the probe counters: |delta| > 127, in [4096, 8192), in [2048, 4096); [4] = interpolation codes beyond +-4096)
    return 0;
}

// the host's side of k_publish: poll the sequence word in pinned memory (a wake-up within a microsecond of the device's store; an event
// behind a device-to-host copy took ~20). The stream is queried now and then so that a fault on the device ends the wait.
static int wait_published(sz3hip_ctx *ctx) {
    volatile uint32_t *seq = ctx->h_pub_seq;
    for (uint64_t it = 1;; it++) {
        if (*seq == ctx->pub_seq) break;
        if ((it & 0xFFFF) == 0) {
            const hipError_t q = hipStreamQuery(ctx->pub_stream);
            if (q == hipSuccess) {
                if (*seq == ctx->pub_seq) break;
                return fail(SZ3HIP_EHIP, "stage 2 finished without publishing its state");
            }
            if (q != hipErrorNotReady) return fail(SZ3HIP_EHIP, "HIP error while waiting for stage 2: %s", hipGetErrorString(q));
        }
        // (a pure spin for the first ~50 us — the device API's calls are that short and the wake-up latency is the step's —, then the
        // core is offered to whoever else wants it: several piece threads of the host API wait here beside the zstd pool, and under a
        // CPU quota a spinning waiter is what throttles the threads it waits for)
        if (it < 4096) __builtin_ia32_pause();
        else if (it < 65536) sched_yield();
        else usleep(20);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return 0;
}
extern "C" int sz3hip_compress_finish(sz3hip_ctx *ctx, size_t *payload_size, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->stage2_done) return fail(SZ3HIP_EINVAL, "finish called before stage2");
    if (int rw = wait_published(ctx)) return rw;  // (the payload is complete: the state's publication is the last thing stage 2 enqueued)
    bool blk_miss = false;
    if (ctx->blk_spec) {  // the call assumed the previous call's hand-over decision (blk_all_lorenzo): what its own count says
        const uint64_t others = ctx->h_state->blk_others;
        const bool all = (ctx->blk_spec_mask & 1u) && (others << BLK_EXIT_SHIFT) < ctx->blk_spec_nblocks;
        ctx->blk_others = others;
        blk_miss = all != ctx->blk_spec_all;
        ctx->blk_spec = false;
        if (blk_miss) ctx->blk_dec_valid = false;  // (the repeat below decides from its own count and records it)
    }
    const bool fused_miss = (ctx->s1_fused && (ctx->h_state->miss_kind != 0 || ctx->h_state->mispredict != 0)) || blk_miss;
    ctx->last_fused = ctx->s1_fused && !fused_miss;
    ctx->last_q16 = ctx->s1_q16 && !(ctx->h_state->miss_kind & (32u | 128u)) && !fused_miss;
    ctx->last_spec_hit = ctx->s2_spec && ctx->h_state->miss_kind == 0 && ctx->h_state->mispredict == 0;
    if ((ctx->h_state->miss_kind & (32u | 128u)) || fused_miss) {
        // stage 1 assumed one-byte codes (the form a context takes after a one-byte call) and this call's probe says two: the
        // whole call once more, in the form that waits for the probe. The input must still be where stage 1 found it.
        // Likewise behind a FUSED stage 1 whose book the verdict rejects (or that met a symbol the book has no code word for, or
        // whose lists were too long for the sort roles): there is no code array to encode again from.
        ctx->redo_calls++;
        ctx->spec_misses++;  // (counted with the other failed shortcuts)
        if (ctx->h_state->miss_kind & 32u) ctx->narrow_hint = 0;
        if (ctx->h_state->miss_kind & 128u) {  // the 16-bit stage 1 met a lattice value beyond its range: the one-byte form from here on
            ctx->q16_hint = 0;
            ctx->q16_block = 16;
        }
        if (ctx->h_state->mispredict || (ctx->h_state->miss_kind & 2u)) ctx->cb_hint = ctx->cb_part = -1;
        const int spec_was = ctx->spec_off;
        ctx->spec_off = 1;
        sz3hip_config conf = ctx->s1_conf;
        int rc1 = sz3hip_compress_stage1(ctx, &conf, ctx->s1_in, stream);
        ctx->spec_off = spec_was;
        if (rc1) return rc1;
        rc1 = stage2_launch(ctx, ctx->s2_payload, ctx->s2_cap, s, S2_CLASSIC);
        if (rc1) return rc1;
        if (int rw = wait_published(ctx)) return rw;
        ctx->spec_penalty = ctx->spec_penalty ? std::min(8, 2 * ctx->spec_penalty) : 1;
        ctx->spec_skip = ctx->spec_penalty;
    } else if (ctx->s2_spec) {
        if (ctx->h_state->miss_kind) {
            // the previous call's book is not this call's: the encoder once more, with the book the packer's book role built from
            // this call's histogram (or, when that role declined the alphabet or a list was too long to sort, the whole of stage 2)
            ctx->spec_misses++;
            const uint32_t kind = ctx->h_state->miss_kind;
            const bool redo_book = (kind & (2u | 4u)) != 0;  // no fresh book (form declined) / lists unsorted: stage 2 from its start
            if (redo_book) {
                if (kind & 2u) ctx->cb_hint = ctx->cb_part = -1;
                HIPCHK(hipMemsetAsync(ctx->d_counters + 7, 0, 24, s));  // mispredict flag and the range words (recomputed)
                ctx->range_ready = false;
            }
            int rc2 = stage2_launch(ctx, ctx->s2_payload, ctx->s2_cap, s, redo_book ? S2_CLASSIC : S2_REENCODE);
            if (rc2) return rc2;
            if (int rw = wait_published(ctx)) return rw;
            ctx->spec_penalty = ctx->spec_penalty ? std::min(8, 2 * ctx->spec_penalty) : 1;
            ctx->spec_skip = ctx->spec_penalty;
        } else {
            ctx->spec_hits++;
            ctx->spec_penalty = 0;
        }
    } else if (ctx->s2_samp && ctx->h_state->miss_kind) {
        // (miss kind 4) a list too long for the packer's sort roles: stage 2 once more in the classic form — codes, segment sums and the
        // sampled book are as stage 1 left them, the code-book launch sorts the lists
        HIPCHK(hipMemsetAsync(ctx->d_counters + 7, 0, 24, s));
        ctx->range_ready = false;
        int rc2 = stage2_launch(ctx, ctx->s2_payload, ctx->s2_cap, s, S2_CLASSIC);
        if (rc2) return rc2;
        if (int rw = wait_published(ctx)) return rw;
    } else if (ctx->spec_skip > 0) {
        ctx->spec_skip--;
    }
    if (ctx->h_state->mispredict) {
        // the code-book form launched alone met the other form's alphabet (the data changed character since the previous
        // call): stage 2 once more with both forms; histogram, range words and outlier lists are as stage 1 left them
        ctx->cb_hint = ctx->cb_part = -1;
        HIPCHK(hipMemsetAsync(ctx->d_counters + 7, 0, 24, s));  // the flag AND the range words (k_hist_range adds to what it finds: the count doubled, round 5)
        ctx->range_ready = false;
        int rc2 = stage2_launch(ctx, ctx->s2_payload, ctx->s2_cap, s, S2_CLASSIC);
        if (rc2) return rc2;
        if (int rw = wait_published(ctx)) return rw;
        if (ctx->h_state->mispredict) return fail(SZ3HIP_EHIP, "code book was not built (both forms declined)");
    }
    ctx->stage1_done = ctx->stage2_done = false;
    // the next call's histogram and counters start from zero: enqueue that now, behind this call, instead of in front of the
    // next one (a launch and its gap off the next call's critical path). Everything the host still wants is in h_state.
    if (ctx->pub_zero && ctx->h_state->miss_kind == 0 && ctx->h_state->mispredict == 0) {
        ctx->pre_cleared = true;  // (k_publish did it: the last launch of this call's stage 2 met the same state)
        ctx->pre_stream = ctx->pub_stream;
        ctx->blk_pre_cleared = ctx->pub_blk_zero;
        ctx->blk_pre_stream = ctx->pub_stream;
    } else if (ctx->d_hist == ctx->d_hist_own && !ctx->hist_exposed && !(szk_dbg_flags & 268435456)) {
        if (hipMemsetAsync(ctx->d_hist, 0, SZH_HIST_BINS * 8 + SZ_COUNTER_BYTES, s) == hipSuccess) {
            ctx->pre_cleared = true;
            ctx->pre_stream = s;
        }
    }
    const szk_state &st = *ctx->h_state;
    if (st.hdr.magic != SZH_MAGIC) return fail(SZ3HIP_EHIP, "device did not produce a payload header (kernel fault?)");
    // lists too long for the code book's sort workgroups went into the payload in arrival order: into index order now (the payload
    // is a function of the input; a rare path — a bound far below the data's noise — with a device-wide sort and a synchronisation)
    if (st.hdr.n_vout > 32768 || st.hdr.n_dout > 32768) {
        uint8_t *pl = (uint8_t *)ctx->s2_payload;
        const int tb = st.hdr.dtype == SZ3HIP_FLOAT ? 4 : 8;
        int rs = 0;
        if (st.hdr.n_vout > 32768) rs |= szk_sort_list_pairs((uint64_t *)(pl + st.off.vout_idx), pl + st.off.vout_val, st.hdr.n_vout, tb, s);
        if (st.hdr.n_dout > 32768) rs |= szk_sort_list_pairs((uint64_t *)(pl + st.off.dout_idx), pl + st.off.dout_val, st.hdr.n_dout, (int)st.hdr.qbytes, s);
        if (rs) return fail(SZ3HIP_EHIP, "sorting the payload's outlier lists failed");
    }
    // the book this payload was coded with is the context's reference from now on
    ctx->book_idx = ctx->book_pending;
    ctx->book_pred = st.hdr.predictor;
    ctx->book_radius = st.hdr.radius;
    ctx->lists_long = st.hdr.n_vout > 2048 || st.hdr.n_dout > 2048;  // (what the packer's sort roles take: ROLE_SORT_MAX)
    ctx->cb_hint = st.n_symbols > SZK_CB_SMALL_SYMS ? 1 : 0;
    ctx->cb_part = szk_cb_part(st.n_symbols, st.hdr.sym_count);
    if (st.hdr.predictor == 2) ctx->blk_wide = st.hdr.sym_count > 3000 ? 1 : 0;  // the block kernels' LDS histogram window of the next call
    ctx->stats.n = st.hdr.n;
    ctx->stats.n_value_outliers = st.hdr.n_vout;
    ctx->stats.n_delta_outliers = st.hdr.n_dout;
    ctx->stats.n_chunks = st.hdr.n_chunks;
    ctx->stats.bitstream_bytes = st.hdr.bitstream_words * 4;
    ctx->stats.payload_bytes = st.hdr.payload_bytes;
    ctx->stats.max_code_len = st.hdr.max_len;
    ctx->stats.n_symbols = st.n_symbols;
    ctx->stats.narrow_codes = ctx->mode.allow && (uint64_t)st.probe[0] * 4096ull <= ctx->mode.n_samples;
    if (st.hdr.predictor == 0 && ctx->mode.allow) ctx->narrow_hint = ctx->stats.narrow_codes ? 1 : 0;
    if (st.hdr.predictor == 0 && ctx->mode.allow) {
        if (ctx->q16_block > 0) ctx->q16_block--;
        ctx->q16_hint = ctx->stats.narrow_codes && st.probe[3] == 0 ? 1 : 0;
    }
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

static bool grow_lists(sz3hip_ctx *ctx, uint64_t want) {  // (the stream is idle: nothing uses the old arrays)
    if (want <= ctx->out_alloc) return true;
    void **arr[4] = {(void **)&ctx->d_vout_idx, (void **)&ctx->d_dout_idx, &ctx->d_vout_val, &ctx->d_dout_val};
    void *fresh[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ok = true;
    for (int i = 0; i < 4 && ok; i++) ok = hipMalloc(&fresh[i], want * 8) == hipSuccess;
    if (!ok) {
        for (void *f : fresh)
            if (f) (void)hipFree(f);
        (void)hipGetLastError();
        return false;
    }
    for (int i = 0; i < 4; i++) {
        (void)hipFree(*arr[i]);
        *arr[i] = fresh[i];
    }
    ctx->out_alloc = want;
    return true;
}
// stage 1 once more with lists that hold `need` unpredictable values (this library's own streams: up to n / 8, beyond that no stream beats the
// lossless fallback; `any_number` — a stock container's writer: as many as there are, the reference keeps lossy streams of nothing else)
int szi_stage1_with_larger_lists(sz3hip_ctx *ctx, const sz3hip_config *conf, const void *d_in, uint64_t need, void *stream, bool any_number) {
    const uint64_t n = conf->num;
    const uint64_t limit = any_number ? std::max<uint64_t>(1024, n) : out_cap_limit(n);
    if (need > limit) return fail(SZ3HIP_EOUTLIERS, "outlier capacity exceeded: data not compressible at this bound");
    const uint64_t want = std::min<uint64_t>(limit, need + need / 16 + 1024);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (!grow_lists(ctx, want)) return fail(SZ3HIP_EOUTLIERS, "no memory for larger outlier lists");
    ctx->force_out_cap = want;
    const int rc = sz3hip_compress_stage1(ctx, conf, d_in, stream);
    ctx->force_out_cap = 0;
    return rc;
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
    const uint64_t cnt[2] = {ctx->h_state->n_vout_raw, ctx->h_state->n_dout_raw};
    const uint64_t need = std::max(cnt[0], cnt[1]);
    if (need > out_cap_limit(n)) return rc;
    const uint64_t want = std::min<uint64_t>(out_cap_limit(n), need + need / 16 + 1024);
    if (cap < payload_bound_n(n, std::max<uint64_t>(ctx->out_cap, want))) return rc;
    if (!grow_lists(ctx, want)) return rc;  // no memory for larger lists: the caller falls back to lossless
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
    const uint32_t big = ctx->h_state->probe[0];  // (the device counters are zeroed behind every call)
    const bool narrow = ctx->mode.allow && (uint64_t)big * 4096ull <= ctx->mode.n_samples;
    if (!narrow) {
        HIPCHK(hipMemcpy(host_codes, ctx->d_codes, n * 2, hipMemcpyDeviceToHost));
        return 0;
    }
    std::vector<uint8_t> b(n);  // one-byte codes: delta + 128, 0 = outlier -> symbols
    HIPCHK(hipMemcpy(b.data(), ctx->d_codes, n, hipMemcpyDeviceToHost));
    const uint32_t add = ctx->h_state->hdr.radius ? ctx->h_state->hdr.radius - 127u : ctx->proto.radius - 127u;  // stored byte = delta + 127, 255 = outlier
    for (uint64_t i = 0; i < n; i++) host_codes[i] = (uint16_t)(b[i] != 255 ? b[i] + add : 0u);
    return 0;
}

extern "C" void sz3hip_debug_force_generic(int on) { szk_force_generic = on; }
// Forget what earlier calls of this context found (kernel forms, histogram windows, the tuner's outcome, the code book): the
// next call behaves like a context's first. The payload never depends on these; the time does (bench.py's cold numbers).
extern "C" void sz3hip_ctx_forget(sz3hip_ctx *ctx) {
    ctx->narrow_hint = ctx->cb_hint = ctx->cb_part = -1;
    ctx->q16_hint = ctx->q16_block = 0;
    ctx->wide16 = -1;
    ctx->pack_wide = ctx->hist_big = ctx->hist_tail = ctx->blk_wide = 0;
    ctx->spec_valid = false;
    ctx->book_idx = -1;
    ctx->spec_skip = ctx->spec_penalty = 0;
    ctx->lists_long = false;
    ctx->half_skip = 0;
    ctx->blk_dec_valid = false;
}
// speculation off (1) / on (0) / on without the back-off after a miss (2, for tests) for this context: with it off every
// stage 2 builds its code book before it encodes
extern "C" void sz3hip_ctx_set_speculation(sz3hip_ctx *ctx, int off) { ctx->spec_off = off; }
// 1: a context's shortcuts never change its payloads (the previous call's code book stands only when this call's histogram gives the
// same book); 0 (the device API's default): the previous book also stands when it is complete over this call's alphabet and codes
// it within 1/1024 of this call's own book's size — the payload then depends on the context's history, its size by < 0.1 %
extern "C" void sz3hip_ctx_set_deterministic(sz3hip_ctx *ctx, int on) { ctx->spec_exact = on ? 1 : 0; }
extern "C" void sz3hip_ctx_set_tuner_exact(sz3hip_ctx *ctx, int on) { ctx->tuner_exact = on ? 1 : 2; }
// ---- stock-stream interoperability: the device work between this library's per-element codes and the reference's emission order ----
int szi_stock_stage1_outcome(sz3hip_ctx *ctx, szi_stock_params *out, uint64_t *n_unpred, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->stage1_done) return fail(SZ3HIP_EINVAL, "no pending stage 1");
    if (ctx->proto.predictor != 1) return fail(SZ3HIP_EUNSUPPORTED, "stage 1 did not take the interpolation predictor");
    uint64_t nv = 0;
    HIPCHK(hipMemcpyAsync(&nv, ctx->d_counters + 0, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *n_unpred = nv;
    if (nv > ctx->cur_out_cap) return fail(SZ3HIP_EOUTLIERS, "outlier capacity exceeded (%llu): data not compressible at this bound", (unsigned long long)ctx->cur_out_cap);
    const szh_header &h = ctx->proto;
    memset(out, 0, sizeof(*out));
    out->N = h.ndim;
    for (int i = 0; i < h.ndim; i++) out->dims[i] = h.dims[4 - h.ndim + i];
    out->interp_id = (int)h.interp_id;
    out->direction = (int)h.interp_dir;
    out->anchor_stride = h.anchor_stride;
    out->alpha = h.interp_alpha;
    out->beta = h.interp_beta;
    out->eb = h.eb;
    out->radius = (int)h.radius;
    *n_unpred = nv;
    return 0;
}
int szi_stock_export(sz3hip_ctx *ctx, const szg_geom *g, const uint64_t *d_blk_base, uint16_t *d_em, void *d_unpred, uint64_t n_unpred,
                     uint32_t *d_tile_cnt, uint64_t *d_tile_base, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    if (szk_launch_stock_from_elem(ctx->dtype, g, d_blk_base, ctx->d_codes, ctx->d_vout_idx, ctx->d_vout_val, n_unpred, d_tile_cnt, d_tile_base, d_em, d_unpred, s))
        return fail(SZ3HIP_EHIP, "stock export kernels failed");
    ctx->stage1_done = ctx->stage2_done = false;  // (this call ends here: no device payload is made of it)
    return 0;
}
int szi_stock_import(sz3hip_ctx *ctx, const szi_stock_params *p, const szg_geom *g, const uint64_t *d_blk_base, const uint16_t *d_em,
                     const void *d_unpred, uint64_t n_unpred, uint32_t *d_tile_cnt, uint64_t *d_tile_base, uint64_t *d_vout_idx, void *d_vout_val,
                     uint32_t *d_bad, void *d_out, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t num = 1;
    for (int i = 0; i < p->N; i++) num *= p->dims[i];
    if (num > ctx->max_n) return fail(SZ3HIP_EINVAL, "array exceeds the context capacity");
    HIPCHK(hipMemsetAsync(d_bad, 0, 4, s));
    if (szk_launch_stock_to_elem(ctx->dtype, g, d_blk_base, d_em, d_unpred, n_unpred, d_tile_cnt, d_tile_base, ctx->d_codes, d_vout_idx, d_vout_val, d_bad, s))
        return fail(SZ3HIP_EHIP, "stock import kernels failed");
    szk_interp_params ip;
    memset(&ip, 0, sizeof(ip));
    ip.N = p->N;
    for (int i = 0; i < p->N; i++) ip.dims[i] = p->dims[i];
    ip.interp_id = p->interp_id;
    ip.direction = p->direction;
    ip.anchor_stride = p->anchor_stride;
    ip.alpha = p->alpha;
    ip.beta = p->beta;
    ip.eb = p->eb;
    ip.radius = p->radius;
    // (the lists are the caller's own arrays: handed over as offsets from a null base)
    if (szk_launch_interp_decompress(ctx->dtype, &ip, nullptr, (uint64_t)(uintptr_t)d_vout_idx, (uint64_t)(uintptr_t)d_vout_val, n_unpred, ctx->d_codes, d_out, s))
        return fail(SZ3HIP_EHIP, "interpolation decoder launch failed");
    uint32_t bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (bad) return fail(SZ3HIP_EFORMAT, "corrupt stock stream: its codes and its list of unpredictable values do not fit together");
    return 0;
}
extern "C" int sz3hip_last_call_fused(const sz3hip_ctx *ctx) { return ctx->last_fused ? 1 : 0; }
extern "C" int sz3hip_last_call_q16(const sz3hip_ctx *ctx) { return ctx->last_q16 ? 1 : 0; }
// 1: the library was built with the superseded forms (python -m sz3_amd.build --lab): the fused stage 1 (sz3hip_ctx_set_fused) and the
// decoder's multi-symbol table (sz3hip_debug_flags(2)); the product build leaves them out and both switches do nothing
extern "C" int sz3hip_lab_build(void) {
#ifdef SZ3HIP_LAB
    return 1;
#else
    return 0;
#endif
}
extern "C" void sz3hip_ctx_set_fused(sz3hip_ctx *ctx, int on) { ctx->fuse_on = on != 0 && sz3hip_lab_build(); }
extern "C" void sz3hip_get_spec_stats(const sz3hip_ctx *ctx, uint32_t *hits, uint32_t *misses) {
    *hits = ctx->spec_hits;
    *misses = ctx->spec_misses;
}
extern "C" void sz3hip_debug_flags(int flags) {
    szk_dbg_flags = flags;
    szk_interp_novec = (flags & 128) != 0;  // 128: interpolation without the 8-wide level-1 kernels and without the level kernels
    szk_interp_min_blocks = (flags & 4194304) ? 1 : 256;  // 4194304: level kernels whatever the array's size
}
extern "C" int sz3hip_debug_codebook_info(sz3hip_ctx *ctx, uint64_t *out16) {
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    szk_cb_info info;
    HIPCHK(hipMemcpy(&info, ctx->bk[ctx->book_idx < 0 ? 0 : ctx->book_idx].info, sizeof(info), hipMemcpyDeviceToHost));
    out16[0] = info.n_symbols; out16[1] = info.max_len; out16[2] = info.sym_min; out16[3] = info.sym_count;
    for (int i = 0; i < 12; i++) out16[4 + i] = info.ts[i];
    return 0;
}

// test hook: which chain the last device decompression took: out[0] half-width intermediates, out[1] chunk-crossing rows (carries),
// out[2] calls left before the half-width chain is tried again after an overflow
extern "C" int sz3hip_debug_decode_info(sz3hip_ctx *ctx, uint32_t *out4) {
    out4[0] = ctx->last_half;
    out4[1] = ctx->last_carry;
    out4[2] = (uint32_t)ctx->half_skip;
    out4[3] = 0;
    return 0;
}

extern "C" int sz3hip_decompress_device(sz3hip_ctx *ctx, const void *d_payload, size_t payload_size, void *d_out,
                                        void *stream) {
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->blk_pre_cleared = false;  // (a block stream's decoder counts in the same counter block)
    if (payload_size < sizeof(szh_header)) return fail(SZ3HIP_EFORMAT, "payload shorter than its header");
    szh_header h;
    uint32_t *ovf = reinterpret_cast<uint32_t *>(ctx->d_counters + 12);  // overflow flag of the half-width chain (see below)
    if (!ctx->h_ovf) {
        HIPCHK(hipHostMalloc((void **)&ctx->h_ovf, 8));
        *ctx->h_ovf = 0;
        HIPCHK(hipMemsetAsync(ovf, 0, 4, s));
    }
    HIPCHK(hipMemcpyAsync(&ctx->h_state->hdr, d_payload, sizeof(szh_header), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(ctx->h_ovf, ovf, 4, hipMemcpyDeviceToHost, s));  // (the previous call's: rides with the header's round trip)
    HIPCHK(hipStreamSynchronize(s));
    if (*ctx->h_ovf) ctx->half_skip = 8;  // that stream's values did not fit: the next calls go straight to full width
    else if (ctx->half_skip > 0) ctx->half_skip--;
    h = ctx->h_state->hdr;
    if (h.magic != SZH_MAGIC || (h.version != SZH_VERSION && h.version != SZH_VERSION_ESC)) return fail(SZ3HIP_EFORMAT, "not an SZH1 payload");
    if ((h.version == SZH_VERSION_ESC) != (h.predictor == 0 && h.anchor_stride != 0))  // (version 5 = version 4 + a Lorenzo stream's escape symbol)
        return fail(SZ3HIP_EFORMAT, "corrupt SZH1 header (version / escape symbol)");
    if (h.predictor > 2) return fail(SZ3HIP_EFORMAT, "unknown predictor id %d in the SZH1 header", h.predictor);
    if (h.predictor != 2 && h.side_bytes) return fail(SZ3HIP_EFORMAT, "corrupt SZH1 header (side section)");
    if (h.side_bytes > payload_size) return fail(SZ3HIP_EFORMAT, "corrupt SZH1 header (side section)");
    if (h.dtype != ctx->dtype) return fail(SZ3HIP_EINVAL, "payload data type does not match the context");
    if (h.n == 0 || h.n > ctx->max_n) return fail(SZ3HIP_EINVAL, "payload element count exceeds the context capacity");
    {   // the extents multiply to n — by successive division, so that crafted extents cannot wrap to it
        uint64_t rest = h.n;
        for (int i = 0; i < 4; i++) {
            if (h.dims[i] == 0 || rest % h.dims[i]) return fail(SZ3HIP_EFORMAT, "corrupt SZH1 header (extents)");
            rest /= h.dims[i];
        }
        if (rest != 1) return fail(SZ3HIP_EFORMAT, "corrupt SZH1 header (extents)");
    }
    if (h.chunk_syms != SZH_CHUNK_SYMS ||
        h.n_chunks != (h.n + SZH_CHUNK_SYMS - 1) / SZH_CHUNK_SYMS || h.sym_count > SZH_HIST_BINS ||
        h.sym_min + h.sym_count > SZH_HIST_BINS || h.max_len > SZH_MAX_LEN || h.radius < 2 || h.radius > 32768 ||
        h.qbytes != (h.dtype == 0 ? 4 : 8) || h.n_vout > h.n || h.n_dout > h.n || h.bitstream_words > payload_size / 4)
        return fail(SZ3HIP_EFORMAT, "corrupt SZH1 header");
    szh_offsets o;
    szk_host_offsets(&h, &o);
    if (o.end > payload_size || h.payload_bytes != o.end) return fail(SZ3HIP_EFORMAT, "truncated SZH1 payload");
    const uint8_t *pl = (const uint8_t *)d_payload;
    szk_blk_params bp;
    szk_blk_scratch sc;
    if (h.predictor == 2) {
        // block-composed stream: selection + coefficients from the side section — read, checked and unpacked first, on the side
        // stream, while the Huffman decoder runs on the caller's
        const uint32_t B = h.interp_id, mask = h.interp_dir;
        const bool fits32 = h.dims[1] < 0xFFFFFFFFull && h.dims[2] < 0xFFFFFFFFull && h.dims[3] < 0xFFFFFFFFull;  // (block positions are 32-bit)
        const bool shape_ok = fits32 && (h.ndim == 4 ? B >= 4 && B <= 6 && h.dims[0] < 0xFFFFFFFFull && !(mask & 2u)
                                                     : h.dims[0] == 1 && (h.ndim == 3 ? B >= 4 && B <= 8 : (h.ndim == 2 ? B >= 4 && B <= 32 && h.dims[1] == 1
                                                                 : h.ndim == 1 && B >= 4 && B <= 65535 && h.dims[1] == 1 && h.dims[2] == 1)));
        if (!shape_ok || mask == 0 || mask > 7 || h.side_bytes < 24 || h.n_dout > h.n)
            return fail(SZ3HIP_EFORMAT, "corrupt SZH1 header (block predictor fields)");
        uint64_t nblocks = 1;
        for (int i = 0; i < 4; i++) {
            nblocks *= (h.dims[i] + B - 1) / B;
            if (nblocks > 0x7FFFFFF0ull) return fail(SZ3HIP_EFORMAT, "corrupt SZH1 header (block count)");
        }
        int rb = blk_reserve(ctx, nblocks);
        if (rb) return rb;
        if (h.ndim <= 3 && ctx->blk_carry_cap < nblocks) {  // 1-D: aggregate + inflow of every block (two lattice words); 2-D / 3-D: the groups' flags (k_blkn_wave2, k_blk_wave3)
            if (ctx->d_blk_carry) HIPCHK(hipFree(ctx->d_blk_carry));
            ctx->d_blk_carry = nullptr;
            ctx->blk_carry_cap = 0;
            HIPCHK(hipMalloc(&ctx->d_blk_carry, nblocks * 17 + (nblocks / 1024 + 2) * 64 + 64));  // (+ the tiles' words — eight each for the second-order scan — and a flag byte per block)
            ctx->blk_carry_cap = nblocks;
        }
        const uint64_t sel_bytes = ((nblocks + 3) / 4 + 7) & ~7ull;
        const uint64_t par_bytes = h.ndim == 4 ? 16 : 8;  // the Rice parameters (one per coefficient: N + 1) and the group count
        if (h.side_bytes < 24 + sel_bytes + par_bytes) return fail(SZ3HIP_EFORMAT, "corrupt side section of a block-predictor stream");
        HIPCHK(hipMemcpyAsync(ctx->h_blk_side_hdr, pl + o.side, 24, hipMemcpyDeviceToHost, s));
        // the Rice parameters behind the selection bits travel with the header: they are shift counts in k_blk_coef_parse
        HIPCHK(hipMemcpyAsync(ctx->h_blk_side_hdr + 24, pl + o.side + 24 + sel_bytes, par_bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        uint32_t coding, sel_bits;
        uint64_t nb_side, nr;
        memcpy(&coding, ctx->h_blk_side_hdr, 4);
        memcpy(&sel_bits, ctx->h_blk_side_hdr + 4, 4);
        memcpy(&nb_side, ctx->h_blk_side_hdr + 8, 8);
        memcpy(&nr, ctx->h_blk_side_hdr + 16, 8);
        for (int i = 0; i < (h.ndim == 4 ? 5 : 4); i++)
            if (ctx->h_blk_side_hdr[24 + i] > 63) return fail(SZ3HIP_EFORMAT, "corrupt side section of a block-predictor stream (Rice parameter)");
        const uint64_t ngroups = (nr + 63) / 64;
        const uint64_t fixed = 24 + sel_bytes + par_bytes + 4 * ngroups;  // header, selection, Rice parameters, group offsets
        if (coding != 1 || sel_bits != 2 || nb_side != nblocks || nr > nblocks || h.side_bytes < fixed || (h.side_bytes - fixed) % 4)
            return fail(SZ3HIP_EFORMAT, "corrupt side section of a block-predictor stream");
        const uint64_t bit_words = (h.side_bytes - fixed) / 4;
        blk_params_from(ctx, h.ndim, h.dims + 1, B, mask, h.eb, (int)h.radius, 0, bp, sc, h.dims[0]);
        memcpy(sc.side_hdr, ctx->h_blk_side_hdr, 24);
        memcpy(sc.side_hdr + 24, &bit_words, 8);
        if (!ctx->side) {
            HIPCHK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(ctx->ev_fork, s));  // (the payload is ready on the side stream when it is on the caller's)
        HIPCHK(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        if (szk_launch_blk_side(&bp, &sc, pl, &o, ctx->d_blk_coef, ctx->side)) return fail(SZ3HIP_EHIP, "side section kernel launch failed");
        HIPCHK(hipEventRecord(ctx->ev_join, ctx->side));
    }
    prof_begin(ctx, ST_DEC_HUFF, s);
    // (a Lorenzo stream's small book — code words up to 16 bits, at most 1024 symbols: the tables' launch also makes the multi-symbol table)
    const bool ms_book = h.predictor == 0 && h.qbytes == 4 && h.max_len >= 1 && h.max_len <= 16 && h.sym_count <= 1024 && (szk_dbg_flags & 2) && sz3hip_lab_build();
    // (a Lorenzo stream's header names the symbol that stands for a listed delta in its anchor_stride field: 0 = symbol 0 itself)
    const uint32_t esc_sym = h.predictor == 0 ? (uint32_t)h.anchor_stride : 0u;
    if (h.predictor == 0 && h.anchor_stride && (h.anchor_stride < h.sym_min || h.anchor_stride >= (uint64_t)h.sym_min + h.sym_count))
        return fail(SZ3HIP_EFORMAT, "corrupt SZH1 header (escape symbol)");
    int rc = szk_launch_dec_tables(pl + o.lens, h.sym_min, h.sym_count, ctx->d_tables, ms_book ? h.radius : 0u, esc_sym, ovf, (const uint16_t *)(pl + o.chunkwords), h.n_chunks,
                                   ctx->d_chunk_off, ctx->d_counters + 3, s);
    if (rc) return fail(SZ3HIP_EHIP, "dec_tables kernel launch failed (%d)", rc);
    szk_dec_params dp;
    dp.n = h.n;
    dp.n_chunks = h.n_chunks;
    dp.bitstream_off = o.bitstream;
    dp.total_words = h.bitstream_words;
    dp.chunk_words = (const uint16_t *)(pl + o.chunkwords);
    dp.sub_bits = (const uint16_t *)(pl + o.subbits);
    dp.group_off = ctx->d_chunk_off;
    dp.tables = ctx->d_tables;
    dp.single_sym = h.sym_min;
    // Lorenzo stream, rows of at most one chunk, a sorted delta-outlier list: the decoder also does the x prefix sum
    const uint64_t row = h.dims[3];
    const bool fuse_x = h.predictor == 0 && h.n_dout <= 32768 && row >= 1 && row <= SZH_CHUNK_SYMS && !(szk_dbg_flags & 512);  // (lists that long are sorted)
    dp.scan_row = fuse_x ? (uint32_t)row : 0u;
    dp.radius = h.radius;
    dp.q_bytes = h.qbytes;
    dp.reserved = ((szk_dbg_flags & 524288) ? 1u : 0u) | ((szk_dbg_flags & 1048576) ? 2u : 0u);  // (experiments: no stores / direct stores)
    dp.q_out = d_out;
    dp.dout_idx = reinterpret_cast<const uint64_t *>(pl + o.dout_idx);
    dp.dout_val = pl + o.dout_val;
    dp.n_dout = h.n_dout;
    // rows that do not divide the decoder's unit: a unit that starts inside a row gets the running sums of the row's earlier units
    // (k_scan_carry); the carries have an array of their own (the code array, idle in this mode, holds the half-width chain's values)
    // Rows that are multiples of the unit (1024) with a strided axis behind them: the first strided scan adds the carries
    // as it reads (no pass of its own); other lengths: k_scan_carry behind the decoder.
    dp.carry = nullptr;
    dp.carry_pass = 0;
    const void *carry_in_scan = nullptr;
    if (fuse_x && (SZH_UNIT_SYMS % row) != 0) {
        if (!ctx->d_carry) HIPCHK(hipMalloc(&ctx->d_carry, (ctx->max_chunks * SZH_SUBS + 8) * 8));
        dp.carry = ctx->d_carry;
        if (row % SZH_UNIT_SYMS == 0 && h.n > row && !(szk_dbg_flags & 536870912)) carry_in_scan = ctx->d_carry;
        else dp.carry_pass = 1;
    }
    dp.half = 0;
    dp.ms = 0;
    dp.ovf = nullptr;
    dp.gate = nullptr;
    // Half-width intermediates: the x-scanned lattice differences of a smooth f32 field fit int16 (f64: int32), and the strided scans that
    // follow then move half the bytes (the code array, idle in the fused mode, holds them). The decoder and the scans raise a
    // flag on a value that does not fit; the full-width chain is enqueued right behind with that flag as its gate (its kernels
    // return at once while it is clear), so the call stays asynchronous and correct either way.
    const bool half = fuse_x && szk_half_scans_ok(&h) && ctx->half_skip == 0 && !(szk_dbg_flags & 2097152);
    ctx->last_half = half ? 1u : 0u;
    ctx->last_carry = dp.carry ? (dp.carry_pass ? 1u : 2u) : 0u;
    void *d_half = ctx->d_codes;  // f32 data: int16 values in the code array (2 bytes per element, idle in the fused mode)
    if (half && ctx->dtype != SZ3HIP_FLOAT) {  // f64 data: int32 values, an array of their own (allocated with the first such call)
        if (!ctx->d_half32) HIPCHK(hipMalloc(&ctx->d_half32, ctx->max_n * 4));
        d_half = ctx->d_half32;
    }
    if (half) {
        dp.half = 1;
        dp.ms = ms_book && ctx->dtype == SZ3HIP_FLOAT ? 1u : 0u;  // (debug flag 2 takes it: measured slower than the one-symbol table in its first form, 612 against 565 us at C2)
        dp.ovf = ovf;
        dp.q_out = d_half;
    }
    rc = szk_launch_decode(pl, &dp, ctx->d_codes, ctx->d_chunk_off, ctx->d_counters + 3, s);
    if (!rc && half) {
        rc = szk_launch_reconstruct_half(pl, &h, &o, d_half, d_out, ovf, s, carry_in_scan);
        dp.half = 0;
        dp.ms = 0;
        dp.ovf = nullptr;
        dp.gate = ovf;
        dp.q_out = d_out;
        if (!rc) rc = szk_launch_decode(pl, &dp, ctx->d_codes, ctx->d_chunk_off, ctx->d_counters + 3, s);
    }
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
        dense2_for(ctx, ip);
        rc = szk_launch_interp_decompress(ctx->dtype, &ip, pl, o.vout_idx, o.vout_val, h.n_vout, ctx->d_codes, d_out, s);
    } else if (h.predictor == 2) {
        // codes -> deltas, then the blocks in anti-diagonal fronts once the side stream has the choices and coefficients
        rc = szk_launch_blk_decompress(ctx->dtype, ctx->d_codes, d_out, &bp, &sc, pl, &h, &o, ctx->d_blk_coef, s, ctx->ev_join);
    } else {
        rc = szk_launch_reconstruct(fuse_x ? 1 : 0, pl, &h, &o, ctx->d_codes, d_out, ctx->d_segtot, s, half ? ovf : nullptr, carry_in_scan);
    }
    prof_end(ctx, ST_DEC_RECON, s);
    if (rc) return fail(SZ3HIP_EHIP, "reconstruct kernel launch failed (%d)", rc);
    return 0;
}

