// sz3_amd/csrc/sz3hip_internal.h — shared between the host-side translation units of libsz3hip.so (not installed):
//   sz3hip_api.cpp   Config, the device-resident API (contexts, stage 1 / 2, decode)
//   sz3hip_host.cpp  libzstd, the SZ3 container, the host-buffer API incl. the slab-parallel (multi-GPU) path, sz3c.h
//   sz3hip_comm.cpp  the RCCL communicator
#ifndef SZ3HIP_INTERNAL_H
#define SZ3HIP_INTERNAL_H
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/sz3hip.h"
#include "sz3hip_format.h"
#include "sz3hip_kernels.h"

// records code + message for sz3hip_last_error() (thread-local) and returns code
int szi_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
struct sz3hip_ctx;
// ---- stock-stream interoperability (sz3hip_stock.hip, sz3hip_stock_host.cpp) ----
struct szg_geom;
struct szi_stock_params {  // what InterpolationDecomposition::save holds besides the quantizer (decomposition/InterpolationDecomposition.hpp:149-159)
    int N;
    uint64_t dims[4];
    int interp_id, direction;
    uint64_t anchor_stride;
    double alpha, beta, eb;
    int radius;
};
// after sz3hip_compress_stage1 chose interpolation (header predictor 1): waits for it, reports its parameters and the number of
// unpredictable values (the anchor grid included); SZ3HIP_EUNSUPPORTED when stage 1 took another predictor
int szi_stock_stage1_outcome(sz3hip_ctx *ctx, szi_stock_params *out, uint64_t *n_unpred, void *stream);
// ... then the codes in the reference's emission order and the quantizer's list of unpredictable values, into the caller's buffers
int szi_stock_export(sz3hip_ctx *ctx, const szg_geom *g, const uint64_t *d_blk_base, uint16_t *d_em, void *d_unpred, uint64_t n_unpred,
                     uint32_t *d_tile_cnt, uint64_t *d_tile_base, void *stream);
// the inverse: emission-order codes + unpredictable values of a stock ALGO_INTERP stream -> the reconstructed array
int szi_stock_import(sz3hip_ctx *ctx, const szi_stock_params *p, const szg_geom *g, const uint64_t *d_blk_base, const uint16_t *d_em,
                     const void *d_unpred, uint64_t n_unpred, uint32_t *d_tile_cnt, uint64_t *d_tile_base, uint64_t *d_vout_idx, void *d_vout_val,
                     uint32_t *d_bad, void *d_out, void *stream);
int szi_stage1_with_larger_lists(sz3hip_ctx *ctx, const sz3hip_config *conf, const void *d_in, uint64_t need, void *stream, bool any_number = false);
// The default algorithm's tuner run from the HOST copy of an array while that array is being copied to the device (the host API: the tuner's
// launches, round trips and host-side pricing vanish behind the copy). conf: the call's Config with its absolute bound; the next
// sz3hip_compress_stage1 of this context with the same Config takes the outcome instead of tuning. Returns 0 when it did.
int szi_pretune_host(sz3hip_ctx *ctx, const sz3hip_config *conf, const void *h_in);
void szi_pretune_cancel(sz3hip_ctx *ctx);  // an outcome is for the call it was made for: every host call starts by dropping what an earlier one may have left
void szi_ctx_exact_default(sz3hip_ctx *ctx, int on);  // what a context does about the tuner's pricing when neither the setter nor the environment says
int szi_tuner_took_lorenzo(sz3hip_ctx *ctx, int *quantbinCnt);  // the default algorithm's tuner chose Lorenzo in the pending stage 1 (1-D), and with which quantizer
void szi_run_parallel(int n, const std::function<void(int)> &f);  // f(0 .. n - 1) over the host API's pool of threads (sz3hip_host.cpp), f(0) on the caller
size_t szi_zstd_size(const void *src, size_t n);  // ZSTD_compress(level 3)'s size of a buffer (sz3hip_host.cpp: libzstd lives there); 0 on error
void *szi_histogram_for_exchange(sz3hip_ctx *ctx);  // the histogram, for the library's own all-reduce between the stages (sz3hip_api.cpp)

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

enum { ST_K1 = 0, ST_CODEBOOK, ST_ENCODE, ST_ASSEMBLE, ST_DEC_HUFF, ST_DEC_RECON, ST_TUNER, ST_K1_KERNEL, ST_SPAN, ST_COUNT };

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
    void *d_dense2;        // interpolation, compression with the level kernels (round 5): the grid of stride 2 as a dense array (lazy, max_n / 8 + its faces)
    size_t dense2_elems;
    void *d_work;          // interpolation: the array being overwritten with reconstructed values (lazy);
                           // block-composed predictor: the lattice values q~ the Lorenzo stencils run on
    // block-composed predictor (sz3hip_regress.hip), allocated on first use for the call's block count
    // what the previous call of this context found, so that this call launches one form of a kernel instead of two
    int narrow_hint;   // Lorenzo code width: 1 one byte, 0 two bytes, -1 unknown
    int cb_hint;       // the previous call's alphabet: 0 up to SZK_CB_SMALL_SYMS symbols (what the packer's book role takes), 1 wider, -1 unknown
    int cb_part;       // the code book launch it called for (szk_cb_part): 0 k_codebook<0>, 1 the wide form, -1 unknown
    int q16_hint;      // 1: the previous Lorenzo call's probe saw lattice values within +-Q16_LIM / 2 only (f32): stage 1 may take its 16-bit form
    int q16_block;     // calls left to sit out after that form met a value beyond its range
    bool s1_q16;       // the pending call's stage 1 ran the 16-bit form
    bool last_q16;     // ... the last finished call's did (and was not repeated)
    bool range_ready;  // stage 1 of the pending call kept the alphabet's range words itself
    bool hist_exposed; // the caller holds a pointer to the histogram (multi-GPU exchange): its range is recomputed in stage 2
    int half_skip;     // decoder: calls left before half-width intermediates are tried again (after a call whose values overflowed)
    uint32_t *h_ovf;   // pinned: the previous decode's overflow flag, fetched with this call's header
    void *s2_payload;  // stage 2's arguments, kept for the repeat after a mispredicted code-book form
    size_t s2_cap;
    // Two code books: bk[book_idx] is the last one a call of this context completed with (-1: none yet). Stage 2 builds this
    // call's book into the other slot; when the previous book may still apply (same predictor / radius) the encoder runs with
    // it while the new one is built beside the encoder (a workgroup of the packer's launch / a stream of its own), and finish() repeats the encoder only when
    // the two differ (the payload is a function of the input alone either way).
    struct Book {
        uint32_t *enc;
        uint8_t *lens;
        szk_cb_info *info;
    } bk[2];
    int book_idx, book_pending;
    uint32_t book_pred, book_radius;  // what bk[book_idx] was built for
    bool s2_spec;                     // the pending stage 2 ran speculatively
    bool lists_long;                  // the previous call listed more than 2048 outliers: the speculative stage 2 takes the any-length sort
    int spec_off;                     // 1: never speculate; 2: speculate without the back-off (tests); 0: product behaviour
    int spec_exact;                   // 1: the previous call's book stands only when it IS this call's book (payload = a pure function of the input)
    int spec_skip, spec_penalty;      // calls to sit out after a miss; the count doubles with every miss in a row (up to 8)
    uint32_t spec_hits, spec_misses;  // statistics (sz3hip_get_spec_stats)
    hipEvent_t ev_done;  // (unused since round 5: the state is published by the device, below)
    uint32_t *h_pub_seq;  // pinned, behind h_state: the sequence word k_publish writes after the state block; finish() polls it
    uint32_t pub_seq;     // ... the value the pending stage 2 will write
    bool pub_zero;        // ... and whether that launch zeroes the next call's histogram and counters when the call needs no repeat
    hipStream_t pub_stream;
    hipStream_t pre_stream;
    bool pre_cleared;    // finish() of the previous call already enqueued the zeroing of histogram and counters
    // stage 1 of the pending Lorenzo call, when it ran with the previous book's code lengths (speculation decided there):
    bool s1_assumed_narrow;  // stage 1 ran the one-launch form (assumes one-byte codes; the probe rides in it)
    sz3hip_config s1_conf;   // stage 1's arguments, kept for the repeat of the whole call after a wrong assumption
    const void *s1_in;
    uint32_t redo_calls;     // statistics: calls repeated from stage 1
    bool s1_spec;          // stage 2 speculates (same condition, evaluated once per call)
    bool seg_expected;     // stage 1 sums the code bits per 256-element segment: stage 2 launches no bits pass
    uint32_t fold_rows;    // != 0: the fold of stage 1's histogram rows was left to stage 2 (it rides in the scan's launch)
    uint32_t *fold_range;
    uint16_t *d_seg_bits;  // [max_n / 256 + 8]
    uint32_t *d_seg_base;  // [max_n / 256 + 8] fused stage 1: index of the first word of a segment's bit string in the scratch
    uint32_t *d_fuse_scratch;  // the fused stage 1's slots when the code array's memory does not hold them (ragged extents), grown on demand
    uint64_t fuse_scratch_words;
    const uint32_t *s1_slots;  // the scratch this call's fused stage 1 wrote
    bool fuse_on;          // the context asks for the fused stage 1 where it applies (sz3hip_ctx_set_fused; default off)
    bool s1_fused;         // this call's stage 1 was the fused form (k_lorenzo_quant_march3f): no code array, the encoder merges
    bool s1_samp;          // (round 6) the pending call's one-byte stream is coded with a book built from a sample of the array (szk_samp)
    bool s1_samp_in;       // ... taken and built inside stage 1's launch, which also summed the segments' bits with it
    bool s2_samp;          // the pending stage 2 ran with that book and the packer's sort roles
    bool last_fused;       // ... the last finished call's
    bool last_spec_hit;    // the last finished call confirmed its speculation
    int blk_wide;           // block predictor: wide LDS histogram window (from the previous call's alphabet)
    uint64_t blk_cap;       // blocks the arrays below hold
    uint8_t *d_blk_sel;     // [blk_cap]
    int64_t *d_blk_coef;    // [blk_cap][4] (encode: per block; decode: per regression rank)
    uint32_t *d_blk_rank, *d_blk_comp;
    uint8_t *d_blk_side;    // szk_blk_side_bound(blk_cap) bytes
    uint64_t *d_blk_counters;  // [8]
    uint8_t *h_blk_side_hdr;   // pinned, 64 bytes
    void *d_blk_stats5;        // [5] doubles: Rice statistics of a 4-D array's five coefficients (64 bytes), then the selection pass's count (blk_all_lorenzo; 16 bytes)
    bool blk_pre_cleared;      // the two counter blocks were zeroed behind the previous call (k_publish, with the histogram: pre_cleared / pre_stream)
    bool pub_blk_zero;         // ... this call's k_publish was given them
    bool blk_cleared_now;      // ... and this call may rely on it (set at stage 1's entry, consumed by the block predictor's stage 1)
    hipStream_t blk_pre_stream;
    void *d_half32;            // f64 decoder: int32 intermediates of the half-width chain (max_n * 4 bytes, lazily)
    bool hist_reduced;         // the library's own exchange sums the histogram between the stages (szi_histogram_for_exchange)
    uint64_t blk_sel_cap;      // blocks d_blk_sel / d_blk_coef hold
    void *d_blk_carry;         // 1-D block streams, decoder: [blocks][2] lattice words
    uint64_t blk_carry_cap;
    bool blk_sel_given;        // the selection pass of this call wrote them
    uint64_t blk_others;       // the last selection pass: blocks that would not be coded by first-order Lorenzo
    // (round 6) the hand-over decision of a call — plain Lorenzo stream or block stream, a function of that count — is ASSUMED to be the
    // previous call's of the same configuration, so that nothing waits for the selection pass on the host; finish() compares and repeats
    bool blk_dec_valid, blk_dec_all;   // the last decision made from a count, and for which call ...
    sz3hip_config blk_dec_conf;
    bool blk_spec, blk_spec_all;       // this call assumed it; what it assumed
    uint64_t blk_spec_nblocks;
    uint32_t blk_spec_mask;
    uint64_t *d_vout_idx, *d_dout_idx;
    void *d_vout_val, *d_dout_val;
    uint32_t *d_enc;
    uint8_t *d_lens;
    uint64_t *d_keys, *d_ifreq;
    uint16_t *d_syms, *d_pleaf, *d_pint, *d_depth, *d_aux2, *d_pint2;
    uint32_t *d_range;
    szk_cb_info *d_info;
    uint16_t *d_chunk_words;
    uint16_t *d_sub_bits;      // the units' bit offsets inside their chunks (payload section subbits)
    uint64_t *d_chunk_off;
    bool spec_valid;               // spec_conf holds the previous call's tuner outcome (ALGO_INTERP_LORENZO, interpolation chosen)
    sz3hip_config spec_conf, spec_used;
    uint32_t last_half, last_carry;  // which chain the last decompression took (test hook)
    void *d_carry;  // decoder: running sums the chunks end with (rows that do not divide the chunk), allocated on first use
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
    uint16_t *d_trial_codes;  // that variant's code array, [SZK_MAX_TRIALS][sampled points] — never d_codes: a speculative stage 1 may
                              // be filling that one on the caller's stream while the tuner runs on the side stream
    size_t trial_codes_cap;
    hipStream_t side;    // the working copy of the input is made here while the tuner runs on the caller's stream
    hipEvent_t ev_fork, ev_join;
    hipStream_t book_stream;       // speculative stage 2, wide alphabets: this call's code book is built here (high priority) beside the encoder
    hipEvent_t ev_s1, ev_book;     // ... forked behind stage 1, joined in front of the verdict
    bool s2_wide;                  // this call's stage 2 took that form
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
    int tuner_exact;         // sz3hip_ctx_set_tuner_exact: trials priced the reference's way (tree + bits + zstd on the host); 0: as SZ3HIP_TUNER_EXACT says, 1 on, 2 off
    bool exact_now;          // ... this call's
    bool exact_default;      // ... with neither the setter nor SZ3HIP_TUNER_EXACT (the host API's contexts: on)
    bool pre_valid;          // szi_pretune_host ran for the call to come: its outcome (pre_conf, pre_report) for the array pre_key describes
    sz3hip_config pre_conf, pre_key;
    sz3hip_tuner_report pre_report;
    double exact_bytes[8];   // ... their sizes, by result slot
    int lz_want, lz_have;      // exact pricing, 1-D: the Lorenzo trials to price beside the first interpolation group (bit 0: at the call's radius, bit 1: at 8192) / priced
    double lz_bytes[2];
    uint16_t *h_trial_codes; // ... the trials' codes and the sampled blocks on the host
    size_t h_trial_codes_cap;
    void *h_samples;
    size_t h_samples_cap;
    bool h_samples_valid;
    // profiling
    bool profiling;
    hipEvent_t ev[ST_COUNT][2];
    bool ev_used[ST_COUNT];
};

#endif
