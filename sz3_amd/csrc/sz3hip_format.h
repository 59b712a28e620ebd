// sz3_amd/csrc/sz3hip_format.h — the device payload ("raw" buffer handed to the lossless stage) of the
// SZ3HIP_ALGO_HIP_LORENZO stream.  It plays the role of the reference's pre-zstd buffer
//   [decomposition.save][encoder.save][u64 n][u64 encBytes][bits]   (compressor/SZGenericCompressor.hpp:51-57)
// re-designed for chunk-parallel GPU encode/decode.  All fields little-endian.
//
//   header (160 B, struct szh_header)
//   lens       u8  [sym_count]      canonical-Huffman code length of symbol sym_min+i (0 = absent)   pad to 16
//                                   (symbols: code = delta + radius; symbol 0 = a listed delta / an unpredictable point. A Lorenzo stream
//                                   — predictor 0 — may name ANOTHER symbol as the stand-in of symbol 0, header.anchor_stride != 0: the
//                                   sampled books of round 6 code a listed delta as symbol radius + 128, so that the table spans the 256
//                                   symbols radius - 127 .. radius + 128 instead of 0 .. radius + 127; canonical code words are assigned
//                                   over the stored symbols, the decoder's tables map the stand-in back to symbol 0)
//   chunkwords u16 [n_chunks]       32-bit words used by chunk c (chunks of chunk_syms symbols)      pad to 16
//   subbits    u16 [n_chunks]       bit offset, from the chunk's start, of its symbol 512: the decoder's restart point (a chunk
//                                   is decoded by two lanes, not one: the serial chain of table lookups per lane bounds the
//                                   decoder, and 131 072 chunks are two waves per SIMD; 2 bytes per 1024 symbols — four
//                                   units per chunk decode no faster while LDS holds four workgroups per CU, and cost 6)  pad to 16
//   vout_idx   u64 [n_vout]         value outliers (Lorenzo) / anchors + unpredictable values (interpolation): index ...
//   vout_val   T   [n_vout]         ... and the raw value stored losslessly (LinearQuantizer "unpred")  pad to 16
//   dout_idx   u64 [n_dout]         delta outliers: element index whose Lorenzo delta does not fit the
//   dout_val   Q   [n_dout]         radius (code 0) and the delta itself (Q = i32 for f32, i64 for f64)  pad to 16
//   side       u8  [side_bytes]     predictor 2 only (block-composed Lorenzo / regression): per-block selection and the
//                                   regression coefficients, layout in sz3hip_regress.hip                 pad to 16
//   bitstream  u8  [4 * bitstream_words]  chunk c starts at byte 4 * sum(chunkwords[0..c)); bits are filled MSB-first and
//                                     the BYTES are in stream order (a 32-bit word of the packer is stored big-endian), as
//                                     in the reference's bit-stream: where the Huffman code saturates (smooth fields, one
//                                     bit per symbol) the lossless stage finds the repeats only in that order (measured:
//                                     C2 field at 5e-2, regression: zstd 0.69 -> 0.5x of the bit-stream);
//                                     code words are canonical (assigned by (len, symbol))
#ifndef SZ3HIP_FORMAT_H
#define SZ3HIP_FORMAT_H
#include <stdint.h>

#define SZH_MAGIC 0x31485A53u /* "SZH1" */
#define SZH_VERSION 4u /* 3: bit-stream bytes in stream order; side section (predictor 2); 4: restart offsets inside the chunks */
#define SZH_VERSION_ESC 5u /* version 4 + a Lorenzo stream whose header names a stand-in for symbol 0 (anchor_stride != 0; round 6's sampled books).
                            * Every other stream is still written as version 4; a decoder of round 6 reads both, an earlier one refuses 5 by name */
#define SZH_CHUNK_SYMS 1024u
#ifndef SZH_SUBS
#define SZH_SUBS 2u /* units per chunk: a unit starts at the chunk's start or at a restart offset */
#endif
#define SZH_UNIT_SYMS (SZH_CHUNK_SYMS / SZH_SUBS) /* symbols a lane of the decoder walks through */
#define SZH_MAX_LEN 24u /* longest code word; alphabets <= 512 symbols are limited to 16 (4 words per 64-bit register in the packer) — but a sampled book's
                           values the sample did not meet: class prefix (<= 16) + index (<= 8) */
#define SZH_HIST_BINS 65536u

typedef struct szh_header {
    uint32_t magic, version;
    uint8_t dtype, ndim, qbytes, predictor; /* predictor: 0 = dual-quantisation Lorenzo, 1 = multilevel interpolation,
                                             * 2 = per-block choice of Lorenzo-1 / Lorenzo-2 / regression (3-D) */
    uint32_t radius;
    uint64_t dims[4]; /* slowest first, left-padded with 1: [w][z][y][x] */
    double eb;
    uint64_t n;
    uint32_t chunk_syms, max_len;
    uint64_t n_chunks;
    uint32_t sym_min, sym_count;
    uint64_t n_vout, n_dout;
    uint64_t bitstream_words;
    uint64_t payload_bytes;
    uint64_t side_bytes; /* predictor == 2: length of the side section, else 0 */
    /* interpolation parameters (predictor == 1), InterpolationDecomposition::save fields
     * (decomposition/InterpolationDecomposition.hpp:149-159) */
    double interp_alpha, interp_beta;
    uint32_t interp_id, interp_dir; /* predictor == 2: block edge, enabled predictors (1 Lorenzo-1 | 2 Lorenzo-2 | 4 regression) */
    uint64_t anchor_stride; /* predictor == 1: the anchor stride; predictor == 0: the symbol that stands for symbol 0 in the lengths' table
                             * (0 = symbol 0 itself; else a symbol inside [sym_min, sym_min + sym_count)) */
} szh_header;

#ifdef __cplusplus
static_assert(sizeof(szh_header) == 160, "szh_header must be 160 bytes");
#endif

#define szh_align16(x) ((((uint64_t)(x)) + 15u) & ~(uint64_t)15u)

typedef struct szh_offsets {
    uint64_t lens, chunkwords, vout_idx, vout_val, dout_idx, dout_val, bitstream, end;
    uint64_t side; /* between dout_val and the bit-stream; empty unless predictor == 2 */
    uint64_t subbits; /* between chunkwords and vout_idx */
} szh_offsets;

#endif
