// sz3_amd/csrc/sz3hip_kernels.h — parameter blocks and host launchers of the gfx950 kernels (sz3hip_kernels.hip)
#ifndef SZ3HIP_KERNELS_H
#define SZ3HIP_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sz3hip_format.h"

// lattice constants derived from the absolute error bound (identical on the encode and the decode side)
struct szk_lattice {
    double recip, two_eb, eb;        // f64 data: 1/(2eb), 2eb, eb
    float recip_f, two_eb_f, eb_lo_f;  // f32 data: (float)(1/(2eb)), (float)(2eb), largest float <= eb
};
static inline szk_lattice szk_make_lattice(double eb) {
    szk_lattice l;
    l.eb = eb;
    l.two_eb = 2.0 * eb;
    l.recip = 1.0 / l.two_eb;
    l.recip_f = (float)l.recip;
    l.two_eb_f = (float)l.two_eb;
    float e = (float)eb;
    if ((double)e > eb) e = __builtin_nextafterf(e, 0.0f);
    l.eb_lo_f = e;
    return l;
}

#define SZK_K1_GRID 2048u  // rows of hist_partial = upper bound of the persistent stage-1 grid

#define SZK_PROBE_STRIDE 32768ull  // the probe looks at 64 consecutive elements of every 32768
// narrow-code mode (see k_probe): a pure function of the probe counter, evaluated on the device by every kernel
struct szk_mode {
    uint32_t *probe_big;          // number of probed deltas outside [-127, 127]
    uint32_t pack_wide;           // host-side choice: the packer caches 8192 instead of 4096 encode-table entries in LDS
    uint64_t n_samples;           // number of probed elements
    uint64_t n_total;             // elements of the array
    uint32_t allow;               // 0: always two-byte codes
};

struct szk_cb_info;
// Round 6: the code book of a one-byte Lorenzo stream from a SAMPLE of the array ("the sampled book"). The sample — SZK_SAMP_UNITS
// 256-element row segments at places that depend on the extents alone — and the book built from its counts are pure functions of the
// input, known a few microseconds into the call instead of after a pass over all codes: stage 1 sums the segments' code bits with THIS
// call's book (no bits pass, no speculation with the previous call's book, nothing to verify or repeat), and a payload does not depend
// on what its context coded before. Every byte value 0 .. 255 gets a code word (a value the sample did not meet counts a quarter of an
// occurrence: code words up to SZH_MAX_LEN bits, which k_pack_b takes); byte 255 — a listed delta — is coded as symbol radius + 128
// (szk_cb_info::esc_sym, the payload header's anchor_stride field), so that the lengths' table of the payload stays 256 entries.
#define SZK_SAMP_UNITS 1024u   // sample units of one array (262144 values)
#define SZK_SAMP_ROLES 128u    // workgroups that take the sample, two units per wave (the last one to finish builds the book)
#define SZK_SAMP_MIN_ELEMS (1ull << 22)
#define SZK_SAMP_SEEN_LEN 16u  // longest code word of a byte value the sample met (13 would make every pair of them fit an entry of the packer's pair table — and costs 0.2 % of the ratio on a 2-D field with a hundred symbols)
// device words of the sample's state, zeroed with the call's counters: [0..255] counts by byte value, [320] workgroups done; on a cache line
// of their own (the workers of stage 1 poll it): [384..447] the book's lengths by byte value, four per word, [448] 1 = the book is in its slot
#define SZK_SAMP_WORDS 512u
#define SZK_SAMP_TICKET 320u
#define SZK_SAMP_LENS 384u
#define SZK_SAMP_READY 448u
struct szk_samp {
    uint32_t *words;          // [SZK_SAMP_WORDS] (nullptr: no sampled book in this call)
    uint32_t *enc;            // the book's slot: [65536] (code word << 5) | length ...
    uint8_t *lens;            // ... [65536] lengths ...
    szk_cb_info *info;        // ... and its descriptor
};
struct szk_k1_params {
    uint64_t d[4];  // extents slowest first, left-padded with 1: [w][z][y][x]
    szk_lattice lat;
    uint32_t radius;
    uint32_t dbg;      // ablation switches for tools/k1_lab.py (0 in production)
    szk_mode mode;
    uint64_t out_cap;  // capacity of each outlier list
    uint64_t *hist;    // [SZH_HIST_BINS]
    uint32_t *hist_partial;  // [SZK_K1_GRID][1024] private histogram rows of the persistent stage-1 workgroups
    uint64_t *n_vout, *n_dout;
    uint64_t *vout_idx, *dout_idx;
    void *vout_val, *dout_val;
    // host side only (profiling): HIP events recorded right before / after the predictor kernel itself, so that its own
    // duration can be set against the per-kernel average of a rocprofv3 trace
    void *prof_ev0, *prof_ev1;
    // two-byte marching kernel: 16384-bin LDS histogram window instead of 8192 (64 KB: 2 workgroups per CU). Codes outside the
    // window cost a global atomic each; when the previous call of the context saw an alphabet wider than the small window the
    // large one pays (C4's f64 slab: 0.6 % of the deltas beyond +-4096, stage 1 0.57 -> 0.36 ms)
    uint32_t wide16;
    uint32_t *range;   // the alphabet's range words (see hist_add_ranged), updated with the histogram; nullptr: left to k_hist_range
    int hint_narrow;   // code width of the context's previous call: 1 one byte, 0 two bytes, -1 unknown (both forms are launched)
    // speculative stage 2: code lengths by symbol of the context's previous code book (nullptr: none). The one-byte marching
    // kernel then sums the code bits of every 256-element row segment into seg_bits[element offset / 256] (x extents that are
    // multiples of 256 only; seg_bits_made reports whether the launched form does it)
    const uint8_t *spec_lens;
    uint16_t *seg_bits;
    uint32_t *seg_made;   // device flag, raised by the form that sums the segments
    int seg_expected;     // out: the launched form sums the segments when this call's probe keeps one-byte codes
    int assumed_narrow;   // out: the one-launch form was taken: it assumes one-byte codes and runs the probe itself
    // the 16-bit form of the one-launch kernel (round 5, k_lorenzo_quant_march3q: f32, 1-D ... 3-D): taken when the context's previous
    // call probed lattice values within +-Q16_LIM / 2 only (hint_q16 > 0); it raises *q16_flag on a value it does not take
    int hint_q16;
    int assumed_q16;      // out: that form was launched
    uint32_t *q16_flag;   // device word, zeroed with the counters
    // defer_fold: the launcher leaves out k_hist_reduce (the fold rides in the encoder's scan launch, szk_encode_roles);
    // fold_rows (out): rows of hist_partial to fold, 0 when the launched kernels need no fold
    int defer_fold;
    uint32_t fold_rows;
    int range_kept;    // out: the launched kernels keep the range words (the register-marching forms do)
    // fused form (round 4, k_lorenzo_quant_march3f): stage 1 codes with the previous call's book and writes the rows' bit strings
    int fuse;                       // in: asked for (a small previous book, the one-launch form, x extent a multiple of 256, the scratch fits)
    int fused;                      // out: the launcher took it
    const uint32_t *fuse_enc;       // [65536] (code word << 5) | length of the previous book
    const szk_cb_info *fuse_info;   // ... its descriptor (max_len is read on the device)
    uint32_t *fuse_slots;           // the scratch: one slot of TY * MARCH_TZ * 128 words per task (the code array's memory)
    uint64_t fuse_cap_words;        // words the scratch holds
    uint32_t *seg_base;             // [n / 256] index of the first word of a segment's bit string in the scratch
    uint32_t *fuse_flag;            // device word, raised when a symbol had no code word in that book
    uint32_t fuse_geom[4];          // out: [3] = words of a task's slot
    // the sampled book (round 6): samp.words != nullptr = the array is one whose one-byte stream is coded with a sampled book. The
    // one-launch forms take the sample and build the book in SZK_SAMP_ROLES workgroups of their own launch and sum the segments' bits
    // with it (samp_in_launch, out); behind the two-launch form a launch of its own does (k_sample)
    szk_samp samp;
    int samp_in_launch;             // out
    // (the launcher's query, szk_fuse_scratch_words: words the scratch of a fused launch over this shape needs)
};

struct szk_cb_info {
    uint32_t n_symbols, max_len, sym_min, sym_count;
    uint32_t win_lo, reserved;  // first symbol of the packers' LDS window of the encode table
    uint32_t esc_sym, pad0;     // 0: a listed delta is symbol 0; else the symbol that stands for it (sampled books: radius + 128) — the decoder maps it back to 0
    uint64_t ts[12];  // phase timestamps (wall_clock64, 100 MHz) for tools/cb_lab.py
    uint32_t first_code[SZH_MAX_LEN + 2];  // (round 5) first code word of every length: what k_cb_assign needs of the book's workgroup (reserved == 0x5A5A while the code words are its to make)
};
struct szk_cb_params {
    uint32_t *enc;   // [65536] (code << 5) | len
    uint8_t *lens;   // [65536]
    uint64_t *keys;  // [65536] scratch (freq << 16 | sym)
    uint16_t *syms;  // [65536] compacted alphabet in symbol order
    uint64_t *ifreq;
    uint16_t *pleaf, *pint, *depth, *aux2, *pint2;  // [65536] scratch for alphabets > 2048 symbols
    uint32_t *range;                 // [4] see k_hist_range
    // the two outlier lists, sorted by blocks 1 and 2 of the same launch
    uint64_t *vout_idx, *dout_idx;
    void *vout_val, *dout_val;
    const uint64_t *n_vout, *n_dout;
    uint64_t out_cap;
    int t_is_32bit, q_is_32bit;
    szk_cb_info *info;
    uint32_t n_books;  // 0/1: one code book (+ the outlier sorts); 2..SZK_MAX_BOOKS: a batch, tables sliced per book
    uint32_t dbg;      // development switches, filled by the launcher from the debug flags (1: force the one-class fallback)
    int range_ready;   // the range words are already filled (stage 1 kept them with the histogram): no k_hist_range launch
    int part_hint;     // -1: both forms of k_codebook are launched; 0 / 1: only that form (small / wide alphabets), see mispredict
    uint32_t *mispredict;  // set to 1 by a form launched alone that meets the other form's alphabet
    int assign_later;      // (set by the launcher) the wide book leaves the code words to k_cb_assign behind its launch (info->first_code, depth[] = the lengths)
    int keys_ready;        // (set by the launcher) keys[] / syms[] were compacted by k_cb_compact in front of this launch, ifreq[0..63] holds its sums
    int skip_sort;         // the launch's two list-sorting workgroups return at once (the lists are sorted elsewhere: speculative stage 2)
    const uint32_t *samp_words;  // non-null: when the words say that this call's sampled book is in its slot, no book is built (the launch sorts the lists only)
};
#define SZK_CB_SMALL_SYMS 256  // alphabets up to this size: the small book with margins, the packer's book role (speculative stage 2), one-byte tables
// which launch builds a call's book: k_codebook<0> (one workgroup, everything in LDS, no launch in front or behind) for alphabets up to
// SZK_CB_PART0_SYMS symbols — beyond SZK_CB_SMALL_SYMS only when the occupied range is at most SZK_CB_PART0_RANGE bins (its compaction reads the
// range with 256 threads) —, k_cb_compact + k_codebook<1> + k_cb_assign otherwise. Round 6: 256 -> 640 (C1's 398 symbols: four launches and
// 85 us became two and ~50; the one-wave merge costs ~0.08 us per symbol, the wide form ~85 us whatever the alphabet)
#define SZK_CB_PART0_SYMS 640
#define SZK_CB_PART0_RANGE 4096
static inline int szk_cb_part(uint32_t n_symbols, uint32_t span) { return n_symbols <= SZK_CB_SMALL_SYMS || (n_symbols <= SZK_CB_PART0_SYMS && span <= SZK_CB_PART0_RANGE) ? 0 : 1; }
#define SZK_MAX_BOOKS 4
#define SZK_MAX_TRIALS 8  // tuner trials of one launch group

struct szk_state {
    szh_header hdr;
    szh_offsets off;
    uint32_t overflow, cap_exceeded;
    uint32_t mispredict, n_symbols;  // code book: wrong form launched alone (stage 2 is repeated); size of the alphabet
    uint32_t book_miss, miss_kind;   // (book_miss: unused) speculative stage 2: the encoder's output is void (stage 2 is repeated); why: 1 the previous call's code
                                     // book is not this call's, 2 code-book form declined, 4 outlier list too long for the short sort,
                                     // 8 stage 1 did not sum the segments' bits, 32 stage 1 assumed one-byte codes and the probe says two
                                     // (the whole call is repeated), 128 the 16-bit stage 1 met a lattice value beyond its range (likewise)
    uint64_t n_vout_raw, n_dout_raw;  // the outlier counters before capping at the lists' capacity (what an overflowing call needs)
    uint32_t probe[6];  // copy of the probe counters (d_counters + 4): one device-to-host copy brings everything the host reads
    uint32_t pad1[2];
    uint64_t blk_others;  // block-composed predictor, a call that ASSUMED the previous call's hand-over decision: the selection pass's count, written by k_publish (host copy only)
};
struct szk_layout_params {
    szh_header proto;
    const uint64_t *n_vout, *n_dout;
    uint64_t out_cap;
    const szk_cb_info *info;
    szk_state *state;
    const uint64_t *side_bytes;  // predictor 2: device word holding the length of the side section (else nullptr)
};
struct szk_asm_params {
    const uint64_t *n_vout, *n_dout;
    uint64_t out_cap;
    int t_is_32bit, q_is_32bit;
    szk_state *state;
    uint8_t *payload;
    uint64_t cap;
    const uint64_t *total_words;
    const uint8_t *lens;
    const uint16_t *chunk_words;
    uint16_t *sub_bits;  // [n_chunks * (SZH_SUBS - 1)] the units' bit offsets: written by the bits pass / k_seg_chunks, copied into the payload by the assembly
    const uint64_t *vout_idx, *dout_idx;
    const void *vout_val, *dout_val;
    const uint8_t *side;  // predictor 2: the side section as the block kernels left it (else nullptr)
    int lists_by_roles;   // the outlier lists are sorted AND copied by role workgroups of the packer's launch (speculative stage 2)
    int assumed_narrow;   // stage 1 ran the one-launch form, which assumes one-byte codes: a probe that says otherwise raises miss_kind bit 32
    const uint32_t *q16_flag;  // stage 1 ran the 16-bit form: the word it raises on a value it does not take (miss_kind bit 128), else nullptr
    szk_mode mode;
    const uint32_t *samp_words;  // the call codes with its sampled book when the words say so: the assembly then writes the state words a book role would
};
// speculative stage 2, small alphabets: work that rides in the encoder's two launches instead of a side stream
struct szk_merge_args {         // what the merge launch needs of a fused stage 1 (k_lorenzo_quant_march3f)
    const uint32_t *slots;      // the scratch its tasks wrote their rows' bit strings to
    const uint32_t *seg_base;   // [n / 256] index of the first word of a segment's string in the scratch
    const uint32_t *fuse_flag;  // raised by stage 1: a book it could not code with
};
struct szk_encode_roles {
    int roles;                  // the packer's launch carries the book role (this call's code book + the verdict) and the two sort roles
    int no_book;                // ... the sort roles only: the book is built elsewhere (wide alphabets: k_codebook<1> on the side stream)
    const uint64_t *hist;
    const szk_cb_params *cb;    // this call's book: fresh slot, part_hint = 0, range words ready
    const uint8_t *used_lens;   // code lengths of the book the packer runs with
    uint32_t *flags;            // [1]: stage 1 summed the segments' bits (device flag)
    int exact;                  // verdict: the used book must BE this call's book (else: complete over the alphabet and within 1/1024 of its size)
    // fold of stage 1's histogram rows in the scan's launch (fold_rows != 0)
    const uint32_t *fold_partial;
    uint32_t fold_rows;
    uint64_t *fold_hist;
    uint32_t *fold_range;
};

#define DEC_LUT_BITS 12u
struct szk_dec_tables {
    uint32_t first_code[SZH_MAX_LEN + 2], first_rank[SZH_MAX_LEN + 2], count[SZH_MAX_LEN + 2];
    uint32_t max_len, n_coded, lut_bits, reserved;
    uint16_t sorted_syms[65536];
    uint32_t lut[1u << DEC_LUT_BITS];  // next lut_bits bits -> (symbol << 8) | length, 0 = longer code word
    // round 5, small code books of Lorenzo streams (k_decode's MS form): the next 12 bits -> UP TO THREE code words at once, as what the
    // fused x prefix sum wants of them: bits 0-3 their total length - 1, 4-5 their number (0: the first code word is longer than 12
    // bits, is symbol 0 — a listed delta — or a delta beyond +-127: decoded on its own), 6-13 d1, 14-22 d1 + d2, 23-30 d3 (two's
    // complement; with fewer than three code words the later sums repeat the last one: the running sum is always "+ s2 + d3")
    uint32_t mlut[1u << DEC_LUT_BITS];
};
struct szk_dec_params {
    uint64_t n, n_chunks;
    uint64_t bitstream_off, total_words;  // the bit-stream section of the payload and its length in 32-bit words
    const uint16_t *chunk_words;  // inside the payload
    const uint16_t *sub_bits;     // inside the payload: bit offsets of a chunk's symbols 256, 512, 768 (the decoder's restart points)
    const uint64_t *group_off;    // word offset of every group of 32 chunks (k_scan_groups)
    const szk_dec_tables *tables;
    uint32_t single_sym;
    // Lorenzo streams with rows of at most one chunk and a sorted delta-outlier list (<= 32768 records): the decoder turns the codes into deltas and
    // prefix-sums them along x itself (scan_row = row length, 0 = plain code output). A lane decodes a unit of 256 symbols; a row
    // that starts in an earlier unit misses those units' running sums: every unit leaves its own in carry[] and k_scan_carry
    // (or the first strided scan, rows that are multiples of the unit) adds them afterwards (not needed when the row length
    // divides the unit).
    uint32_t scan_row, radius, q_bytes, reserved;  // q_bytes: 4 = int32 lattice (f32 data), 8 = int64 (f64 data)
    void *q_out;  // lattice deltas summed along x: int32 (f32 data) / int64 (f64 data), n elements
    void *carry;  // [units] running sum at the end of every unit (same type), nullptr when rows start on unit boundaries
    uint32_t carry_pass;  // 1: k_scan_carry adds them behind the decoder; 0: the first strided scan does (szk_launch_reconstruct*)
    // delta outliers of a fused stream (code 0): the sorted (index, delta) lists inside the payload, searched by index
    const uint64_t *dout_idx;
    const void *dout_val;  // int32 / int64 like q_out
    uint64_t n_dout;
    // Half-width intermediates (f32 data: q_out is int16, half = 1): the x-scanned lattice differences of smooth fields are
    // small, and the two strided scans that follow move half the bytes. A value that does not fit raises *ovf; the full-width
    // chain is enqueued behind the half-width one with gate = ovf: its kernels return at once while the flag is clear.
    uint32_t half;
    uint32_t ms;           // half, f32 data, a code book of at most 16-bit words and 1024 symbols: the multi-symbol table form (tables->mlut)
    uint32_t *ovf;         // half: raised on a value outside int16
    const uint32_t *gate;  // non-null: the kernel runs only when *gate != 0
};

// ---- block-composed predictor: Lorenzo-1 / Lorenzo-2 / regression per block (sz3hip_regress.hip) ----
struct szk_blk_params {
    uint64_t d[3];   // z, y, x extents
    uint32_t nb[3];  // blocks per dimension
    uint32_t B;      // block edge: 4..8 (3-D); low-dimensional arrays (ndim 1, 2: the kernels see d = (1, 1, n) / (1, dy, dx)) up to 65535 / 32
    uint32_t ndim;   // dimensions of the caller's array (1, 2, 3): the coefficient lattices' steps are eb / (ndim + 1) [/ B]
    uint32_t mask;   // enabled predictors: 1 Lorenzo-1 | 2 Lorenzo-2 | 4 regression
    szk_lattice lat;
    double eb;
    uint32_t radius;
    uint64_t out_cap;
    uint64_t *hist;
    uint64_t *n_vout, *n_dout;
    uint64_t *vout_idx, *dout_idx;
    void *vout_val, *dout_val;
    uint8_t *sel;    // [blocks] chosen predictor: 0 Lorenzo-1, 1 Lorenzo-2, 2 regression
    int64_t *coef;   // [blocks][4] coefficient lattice values of the regression blocks
    void *qwork;     // [n] lattice values q~ (int32 / int64) the Lorenzo stencils run on
    uint64_t *n_reg; // number of regression blocks (counted by the fit pass; the rank pass writes the same number)
    uint32_t sel_given;  // sel[] / coef[] were written by k_blk_select: k_blk_fit codes what they say instead of fitting again
    void *carry;         // low-dimensional decoder, ndim 1: [blocks][2] lattice words (a block's aggregate, then the value left of it)
    uint64_t dw;         // 4-D arrays (ndim 4, round 4): the extent of the slowest dimension, d[] holds the other three; 1 otherwise
    uint32_t nbw;        // ... and its blocks
};
struct szk_blk_scratch {
    uint32_t *rank, *comp;  // [blocks] rank among the regression blocks, compacted list of their ids
    uint32_t *run_scratch;  // [blocks / 8192 + 1] regression blocks per run of 8192 blocks, then the runs' offsets
    uint64_t *counters;     // [8]: [0] regression blocks, [2] side bytes, [3] fit pass's count, [4..7] Rice statistics
    uint8_t *side;          // the side section being built
    int wide_hist;          // encode: 16384-bin LDS histogram window instead of 4096 (the context's previous alphabet was wide)
    uint8_t side_hdr[32];   // decode: host copy of the side section's header + [24..31] words of its bit section (validated by the caller)
    double *stats5;         // encode, 4-D arrays: [5] Rice statistics of the five coefficients (zeroed by the caller)
    const uint64_t *range_hist;  // encode, optional: the call's histogram and its three range words (at zero) — a small array's side launch then also
    uint32_t *range;             // finds the range of the non-empty bins (k_hist_range's work in 64 more workgroups of that launch); see szk_blk_side_small()
};
// 1: the side section of a stream of `nblocks` blocks is built by the one-workgroup launch (which takes the range words along when given)
int szk_blk_side_small(uint64_t nblocks);
// the selection pass alone (a block per lane): *n_other += the blocks that would not be coded by first-order Lorenzo
// the tuner's Lorenzo trial for 1-D arrays: the set [Lorenzo-1, Lorenzo-2] in blocks of five over the sample blocks (sz3hip_regress.hip)
int szk_launch_trial_lorenzo12(int dtype, const void *d_samples, uint64_t per, uint64_t nsb, double eb, int radius, uint64_t *hist, uint64_t *counters,
                               uint64_t *extra, hipStream_t s);
int szk_launch_blk_select(int dtype, const void *d_in, const szk_blk_params *p, uint64_t *n_other, hipStream_t s);
int szk_launch_blk_compress(int dtype, const void *d_in, uint16_t *codes, const szk_blk_params *p, const szk_blk_scratch *sc, hipStream_t s);
// decoder, first part (stream-independent of the Huffman decoder): the side section -> sel[], rank[], coef_by_rank[]
int szk_launch_blk_side(const szk_blk_params *p, const szk_blk_scratch *sc, const uint8_t *payload, const szh_offsets *o, int64_t *coef_by_rank,
                        hipStream_t s);
int szk_launch_blk_decompress(int dtype, const uint16_t *codes, void *d_out, const szk_blk_params *p, const szk_blk_scratch *sc,
                              const uint8_t *payload, const szh_header *h, const szh_offsets *o, int64_t *coef_by_rank, hipStream_t s, hipEvent_t side_done = nullptr);
size_t szk_blk_side_bound(uint64_t nblocks);
// (index, value) records of an outlier list inside a finished payload into index order (lists beyond 32768 records: sz3hip_sortlists.hip)
int szk_sort_list_pairs(uint64_t *idx, void *val, uint64_t n, int val_bytes, hipStream_t s);
// codes -> lattice deltas (code - radius) in d_out, the delta outliers scattered over them (Lorenzo and block streams)
int szk_launch_expand_deltas(int dtype, const uint16_t *codes, uint64_t n, int radius, const uint8_t *payload, const szh_offsets *o,
                             uint64_t n_dout, void *d_out, hipStream_t s);
// the delta outliers alone, scattered to their code positions in d_out (nothing else of d_out is written)
int szk_launch_scatter_deltas(int dtype, uint64_t n, const uint8_t *payload, const szh_offsets *o, uint64_t n_dout, void *d_out, hipStream_t s);

// ---- interpolation predictor (sz3hip_interp.hip) ----
struct szk_interp_params {  // what InterpolationDecomposition keeps (decomposition/InterpolationDecomposition.hpp:456-477)
    int N;
    uint64_t dims[4];  // slowest first, exactly N entries
    int interp_id, direction;
    uint64_t anchor_stride;
    double alpha, beta, eb;
    int radius;
    uint64_t *n_vout, *vout_idx;
    void *vout_val;
    uint64_t out_cap;
    uint32_t hist_big;   // histogram pass with the 16384-bin second tier (host choice, from the previous call's far count)
    uint32_t hist_tail;  // with hist_big: the codes beyond +-8192 are counted by three windowed passes in LDS (k_hist_tail)
    uint32_t *far_cnt;   // [0] receives the number of codes outside +-4096 of the radius, [1] (hist_big form) outside +-8192
    void *dense2;        // compression with the level kernels (round 5): room for the grid of stride 2 as a dense array — the level of stride 2 leaves
                         // its reconstruction (and the coarse points it loaded) there, the finest level reads its coarse points from there; nullptr: in place
    uint64_t dense2_elems;
};
struct szk_interp_pass {
    int N, dir, interp_id, old_api, subpass, radius;
    int kind;     // 0: anchor grid, 1: first point (no anchors), 2: directional pass
    int no_store; // compression, final pass of the schedule: nothing reads its reconstruction, so it is not written
    uint64_t dims[4], off[4], start[4], step[4], cnt[4];
    uint64_t total, s, bsz;
    uint64_t batch_stride;  // elements between the independent arrays of a batch (grid.y), 0 = one array
    double eb, eb_recip;
    uint64_t *n_vout, *vout_idx;
    void *vout_val;
    uint64_t out_cap;
};
extern int szk_interp_novec;
extern int szk_interp_min_blocks;
// 1: the array runs through the level kernels (one launch per level, no working copy of the input)
int szk_interp_levels_ok(const szk_interp_params *ip);
// d_in == nullptr: d_work already holds the copy of the input (made on a side stream while the tuner ran)
int szk_launch_interp_compress(int dtype, const szk_interp_params *ip, const void *d_in, void *d_work, uint16_t *codes,
                               uint64_t *hist, hipStream_t s);
int szk_launch_interp_decompress(int dtype, const szk_interp_params *ip, const uint8_t *payload, uint64_t vout_idx_off,
                                 uint64_t vout_val_off, uint64_t n_vout, uint16_t *codes, void *d_out, hipStream_t s);

int szk_launch_profile_blocks(int dtype, const void *d_in, int N, const uint64_t *dims, uint64_t bs, uint64_t stride, double abseb,
                              uint8_t *d_flags, uint64_t *total_out, hipStream_t s);
int szk_launch_gather_blocks(int dtype, const void *d_in, int N, const uint64_t *dims, uint64_t edge, const uint64_t *d_starts,
                             uint32_t nblocks, void *d_out, hipStream_t s);
#define SZK_TRIAL_MAX_PASSES 64
int szk_launch_interp_trials(int dtype, const szk_interp_params *ips, uint32_t ntrials, const void *d_samples, void *d_work,
                             uint16_t *codes, uint32_t nblocks, uint64_t *d_hists, szk_interp_pass *h_passes, szk_interp_pass *d_passes,
                             uint32_t *h_np, uint32_t *d_np, int keep_codes, hipStream_t s);
// res (4 words per book): entropy of the histogram in 1/256 bit, symbols in use, counters[0], counters[1]
// unpred_is_code0: the unpredictable count of an interpolation trial is its number of points coded 0 (hist[0])
int szk_launch_code_cost(const uint64_t *hist, const uint64_t *counters, uint64_t *d_res, uint32_t n_books, uint64_t total,
                         int unpred_is_code0, hipStream_t s);

int szk_launch_int_to_f64(int sz_type /* SZ_UINT8 = 2 .. SZ_INT64 = 9 */, const void *d_in, uint64_t n, double *d_out, uint32_t *d_flag, hipStream_t s);
int szk_launch_f64_to_int(int sz_type, const double *d_in, uint64_t n, void *d_out, hipStream_t s);
int szk_launch_hist_add(uint64_t *d_dst, const uint64_t *d_src, uint32_t n, hipStream_t s);  // dst[i] += src[i]
int szk_launch_minmax(int dtype, const void *d_in, uint64_t n, double *d_partial /*[2*1024]*/, double *d_out /*[2]*/, hipStream_t s);
int szk_launch_k1(int dtype, int ndim, const void *d_in, uint16_t *codes, szk_k1_params *p, hipStream_t s);
int szk_launch_codebook(const uint64_t *d_hist, const szk_cb_params *p, hipStream_t s);
int szk_launch_encode(const uint16_t *codes, uint64_t n, const uint32_t *d_enc, const szk_cb_info *info, int radius,
                      szk_mode mode, uint16_t *chunk_words, uint64_t *group_off /*[n_chunks/32 + 1]*/, uint64_t *total_words,
                      const szk_state *state, uint8_t *payload, const szk_layout_params *layout,
                      const szk_asm_params *asmp /* non-null: the packer's launch also assembles the payload (no szk_launch_assemble) */, hipStream_t s,
                      const uint16_t *seg_bits = nullptr /* non-null: code bits per 256-element segment, summed by stage 1 (no bits pass) */,
                      const uint32_t *seg_made = nullptr /* device flag: stage 1 really made them */,
                      const szk_encode_roles *roles = nullptr,
                      const szk_merge_args *merge = nullptr /* stage 1 was the fused form: k_merge instead of the packer */);
// the fold of stage 1's per-workgroup histogram rows (k_hist_reduce), for a caller that deferred it (szk_k1_params::defer_fold)
// and does not run the encoder form that carries it
// speculative stage 2 with a wide alphabet: the book built on the side stream against the one the packer used (miss_kind bit 1:
// they differ; bit 2 + mispredict: the side launch met a small alphabet and built nothing) — one workgroup, on the encoder's stream
// once the side stream has joined
int szk_launch_book_verdict(const szk_cb_info *fresh, const uint8_t *fresh_lens, const szk_cb_info *used, const uint8_t *used_lens,
                            const uint32_t *mispredict, const uint32_t *range, szk_state *state,
                            const uint64_t *hist /* this call's histogram */, int exact /* see szk_encode_roles::exact */, hipStream_t s);
// stock-stream interoperability (sz3hip_stock.hip): permutations between the reference's emission order and element order
struct szg_geom;
int szk_launch_stock_to_elem(int dtype, const szg_geom *g, const uint64_t *d_blk_base, const uint16_t *d_em, const void *d_unpred, uint64_t n_unpred,
                             uint32_t *d_tile_cnt, uint64_t *d_tile_base, uint16_t *d_codes, uint64_t *d_vout_idx, void *d_vout_val, uint32_t *d_bad,
                             hipStream_t s);
int szk_launch_stock_from_elem(int dtype, const szg_geom *g, const uint64_t *d_blk_base, const uint16_t *d_codes, const uint64_t *d_vout_idx,
                               const void *d_vout_val, uint64_t n_vout, uint32_t *d_tile_cnt, uint64_t *d_tile_base, uint16_t *d_em, void *d_unpred,
                               hipStream_t s);
int szk_launch_stock_ranks(const szg_geom *g, const uint64_t *d_blk_base, uint64_t *d_rank, hipStream_t s);
// the Huffman stage of a stock stream on the device (sz3hip_stock.hip): coder (tile bit counts, scan, pack into a zeroed word array) and the
// self-synchronising decoder; the tree argument is sz3hip_stock.hip's szk_stock_tree (same layout as the caller's mirror struct)
int szk_launch_stock_huff_encode(const uint16_t *d_em, uint64_t n, const uint8_t *d_clen, const uint64_t *d_cbits, uint32_t *d_tile_bits, uint64_t *d_tile_base,
                                 uint32_t *d_out_words, uint64_t out_words_cap, uint64_t *total_bits, hipStream_t s);
struct szk_stock_tree_dev {  // HuffmanEncoder's serialised tree on the device (encoder/HuffmanEncoder.hpp:601-628)
    const uint32_t *L, *R;  // children by pre-order node index (0: none)
    const int32_t *C;       // leaf: symbol - offset
    const uint8_t *t;       // 1: leaf
    const uint32_t *lut;    // [4096] (node reached by twelve bits << 8) | (bits used << 1) | leaf
    uint32_t nc;
    int32_t offset;
};
int szk_launch_stock_huff_decode(const szk_stock_tree_dev *tr, const uint32_t *d_words, uint64_t nbytes, uint64_t n, uint64_t *d_start, uint64_t *d_last,
                                 uint64_t *d_next, uint64_t *d_base, uint32_t *d_count, uint32_t *d_flags, uint16_t *d_em, int *passes, hipStream_t s);
// a stock ALGO_LORENZO_REG stream on the device (round 4, read side: sz3hip_stock.hip k_slr_*): the reference's own arithmetic —
// predictions from reconstructed values in T, LinearQuantizer::recover — block by block in anti-diagonal fronts
struct szk_slr_params {
    uint64_t d[3];     // the array as (z, y, x): leading extents 1 for N < 3
    uint32_t nb[3];    // blocks per dimension
    uint32_t B, N;
    double eb;         // the main quantizer's bound and radius (from the stream)
    uint32_t radius;
    const uint16_t *codes;      // block by block (block raster order, raster order inside a block)
    const uint8_t *kind;        // [blocks] 0 Lorenzo-1, 1 Lorenzo-2, 2 regression
    const void *coef;           // [blocks][4] T: a regression block's N + 1 coefficients (recovered on the host: a chain over the blocks)
    const void *unpred;         // the quantizer's unpredictable values, in the order of their zero codes
    uint64_t n_unpred;
    const uint64_t *tile_base;  // zero codes in front of every tile of 1024 codes
    void *out;
    uint32_t *bad;              // raised by a zero code beyond the list
    uint64_t dw;                // 4-D arrays (N == 4, round 5): the slowest extent, d[] holds the other three; coef is then [blocks][8]
    uint32_t nbw;               // ... and its blocks
};
int szk_launch_stock_lorenzo_reg(int dtype, const szk_slr_params *p, uint64_t n, uint32_t *d_tile_cnt, uint64_t *d_tile_base, hipStream_t s);
// ... and the WRITE side (round 5, sz3hip_stock.hip k_slw_*; 2-D and 3-D arrays): a stream stock SZ3 decodes. The format leaves the choice of
// a block's predictor and of its regression coefficients to the writer — the reader follows the selection and the coefficient chain it is
// handed (ComposedPredictor::predecompress, RegressionPredictor::pred_and_recover_coefficients) — so the selection is made up front, in
// parallel, from the ORIGINAL neighbours (the reference decides from reconstructed ones, block after block: a chain through the whole
// array), the coefficient chain is quantized on the host (a chain over the regression blocks), and the values are coded front by front
// of blocks in the reference's own arithmetic: prediction from reconstructed values in T, LinearQuantizer::quantize_and_overwrite.
// Round 6: the selection is then repeated against the coded array and the pass with it while a choice moves (`reselect` below): what stands
// is the reference's own vector, the container its file.
struct szk_slw_params {
    uint64_t d[3];     // the array as (z, y, x): leading extents 1 for N < 3
    uint32_t nb[3];
    uint32_t B, N;
    double eb;
    uint32_t radius;
    uint32_t set_mask;        // the predictor set: 1 Lorenzo-1, 2 Lorenzo-2, 4 regression (the reference's order)
    const void *in;           // the caller's array
    void *recon;              // [n] T: values as the reader will have them (written block by block; a block's halo is read from here)
    uint16_t *codes;          // block by block (block raster order, raster order inside a block)
    void *uval;               // [n] T, by code position: the original value where the code is 0
    uint8_t *kind;            // [blocks] 0 Lorenzo-1, 1 Lorenzo-2, 2 regression
    uint8_t *sel;             // [blocks] the chosen member's index in the set's order (the selection vector of a composed set)
    void *coef_fit;           // [blocks][4] T: the fit of every block the regression member is valid for (selection pass)
    const void *coef;         // [blocks][4] T: the RECOVERED coefficients of the regression blocks (host chain), what predictions use
    uint64_t dw;              // 4-D arrays (N == 4): the slowest extent, d[] holds the other three; coef_fit / coef are then [blocks][8]
    uint32_t nbw;             // ... and its blocks
    // the selection repeated behind a coding pass (round 6): a block's halo then holds the values as the READER will have them (recon) — what the
    // reference's estimate sees, whose loop overwrites the array block by block (ComposedPredictor.hpp:25-40 inside
    // BlockwiseDecomposition.hpp:33-44) —, the choices go to kind_new / sel_new and *n_changed counts the blocks whose choice moved
    uint32_t reselect;
    uint8_t *kind_new, *sel_new;
    uint32_t *n_changed;
};
int szk_launch_stock_lr_select(int dtype, const szk_slw_params *p, hipStream_t s);
// stock ALGO_NOPRED streams (round 5): every value quantized against a prediction of 0 (NoPredictionDecomposition.hpp:17-33)
int szk_launch_stock_nopred_decode(int dtype, const uint16_t *d_codes, uint64_t n, double eb, uint32_t radius, uint32_t *d_tile_cnt, uint64_t *d_tile_base,
                                   const void *d_unpred, uint64_t n_unpred, void *d_out, uint32_t *d_bad, hipStream_t s);
int szk_launch_stock_nopred_encode(int dtype, const void *d_in, uint64_t n, double eb, uint32_t radius, uint16_t *d_codes, hipStream_t s);
int szk_launch_stock_lr_code(int dtype, const szk_slw_params *p, hipStream_t s);
// histogram of a stock stream's codes (u64[65536], zeroed by the caller) and the unpredictable values in the order of their zero codes
int szk_launch_stock_lr_finish(int dtype, const szk_slw_params *p, uint64_t n, uint64_t *d_hist, uint32_t *d_tile_cnt, uint64_t *d_tile_base, void *d_unpred,
                               uint64_t *h_n_unpred, hipStream_t s);
#ifdef __cplusplus
#include <vector>
// the geometry of an array under InterpolationDecomposition::init (:176-213) and the per-block bases of its emission order; 0 on success
int szk_stock_geom_build(int N, const uint64_t *dims, int interp_id, int direction, uint64_t anchor_stride, szg_geom *g, std::vector<uint64_t> *blk_base);
#endif
// words of scratch a fused stage 1 over this view (extents slowest first, left-padded with 1) needs; 0: the shape does not take that form
uint64_t szk_fuse_scratch_words(int ndim, const uint64_t d[4]);
int szk_launch_hist_range(const uint64_t *d_hist, uint32_t *range /* [4], zeroed */, hipStream_t s);  // range and count of the non-empty bins
int szk_launch_hist_fold(const uint32_t *partial, uint32_t nrows, int radius, uint64_t *hist, uint32_t *range, hipStream_t s);
int szk_launch_assemble(const szk_asm_params *p, hipStream_t s);
// the state block -> the host's pinned copy + a sequence word behind it, written by the device (finish() polls the word); d_zero != nullptr:
// the same launch zeroes zero_bytes (a multiple of 16) there when the state reports no miss and no mispredicted code-book form
int szk_launch_publish(const szk_state *d_state, void *h_state, uint32_t *h_seq, uint32_t seq, void *d_zero, uint64_t zero_bytes, hipStream_t s,
                       const uint64_t *d_blk_others = nullptr /* non-null: the word the host copy's blk_others receives */,
                       void *d_zero_blk0 = nullptr /* with d_zero: 64 bytes ... */, void *d_zero_blk1 = nullptr /* ... and 80 bytes zeroed under the same condition (the block predictor's counters) */);
int szk_launch_dec_tables(const uint8_t *d_lens, uint32_t sym_min, uint32_t sym_count, szk_dec_tables *t,
                          uint32_t radius /* != 0: the multi-symbol table of a Lorenzo stream's small book is made too (mlut) */,
                          uint32_t esc_sym /* != 0: the stored symbol that decodes as symbol 0 (a listed delta; sampled books) */,
                          uint32_t *zero_word /* a device word this launch clears (nullptr: none) */,
                          const uint16_t *chunk_words, uint64_t n_chunks, uint64_t *group_off, uint64_t *total_words /* the decoder's group
                          offsets, made by a second workgroup of the same launch (chunk_words == nullptr: not made) */, hipStream_t s);
int szk_launch_decode(const uint8_t *payload, const szk_dec_params *p, uint16_t *codes, uint64_t *chunk_off,
                      uint64_t *total_words, hipStream_t s);
// x_done: the decoder already produced the x-scanned lattice values in d_out (szk_dec_params::scan_row)
int szk_launch_reconstruct(int x_done, const uint8_t *payload, const szh_header *h, const szh_offsets *o, const uint16_t *codes,
                           void *d_out, void *d_segtot, hipStream_t s, const uint32_t *gate = nullptr,
                           const void *carry = nullptr /* the decoder's unit carries, to be added by the first strided pass (RowCarry) */);
// the strided scans of a Lorenzo stream whose decoder left int16 x-scanned values in d_half (see szk_dec_params::half);
// szk_half_scans_ok says whether the shape qualifies (even x extent, enough lines for one thread pair per line)
int szk_half_scans_ok(const szh_header *h);
int szk_launch_reconstruct_half(const uint8_t *payload, const szh_header *h, const szh_offsets *o, void *d_half /* int16 (f32 data) / int32 (f64) */,
                                void *d_out, uint32_t *ovf, hipStream_t s, const void *carry = nullptr /* int32 / int64 unit carries */);
void szk_host_offsets(const szh_header *h, szh_offsets *o);
extern int szk_force_generic;
extern int szk_dbg_flags;

#endif
