// sz3_amd/csrc/sz3hip_format.h — the device payload ("raw" buffer handed to the lossless stage) of the
// SZ3HIP_ALGO_HIP_LORENZO stream.  It plays the role of the reference's pre-zstd buffer
//   [decomposition.save][encoder.save][u64 n][u64 encBytes][bits]   (compressor/SZGenericCompressor.hpp:51-57)
// re-designed for chunk-parallel GPU encode/decode.  All fields little-endian.
//
//   header (160 B, struct szh_header)
//   lens       u8  [sym_count]      canonical-Huffman code length of symbol sym_min+i (0 = absent)   pad to 16
//   chunkwords u16 [n_chunks]       32-bit words used by chunk c (chunks of chunk_syms symbols)      pad to 16
//   vout_idx   u64 [n_vout]         value outliers (Lorenzo) / anchors + unpredictable values (interpolation): index ...
//   vout_val   T   [n_vout]         ... and the raw value stored losslessly (LinearQuantizer "unpred")  pad to 16
//   dout_idx   u64 [n_dout]         delta outliers: element index whose Lorenzo delta does not fit the
//   dout_val   Q   [n_dout]         radius (code 0) and the delta itself (Q = i32 for f32, i64 for f64)  pad to 16
//   bitstream  u32 [bitstream_words]  chunk c starts at word sum(chunkwords[0..c)); inside a word bits are
//                                     filled MSB-first; code words are canonical (assigned by (len, symbol))
#ifndef SZ3HIP_FORMAT_H
#define SZ3HIP_FORMAT_H
#include <stdint.h>

#define SZH_MAGIC 0x31485A53u /* "SZH1" */
#define SZH_VERSION 2u
#define SZH_CHUNK_SYMS 1024u
#define SZH_MAX_LEN 24u /* longest code word; alphabets <= 512 symbols are limited to 16 (4 words per 64-bit register in the packer) */
#define SZH_HIST_BINS 65536u

typedef struct szh_header {
    uint32_t magic, version;
    uint8_t dtype, ndim, qbytes, predictor; /* predictor: 0 = dual-quantisation Lorenzo, 1 = multilevel interpolation */
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
    uint64_t reserved1;
    /* interpolation parameters (predictor == 1), InterpolationDecomposition::save fields
     * (decomposition/InterpolationDecomposition.hpp:149-159) */
    double interp_alpha, interp_beta;
    uint32_t interp_id, interp_dir;
    uint64_t anchor_stride;
} szh_header;

#ifdef __cplusplus
static_assert(sizeof(szh_header) == 160, "szh_header must be 160 bytes");
#endif

#define szh_align16(x) ((((uint64_t)(x)) + 15u) & ~(uint64_t)15u)

typedef struct szh_offsets {
    uint64_t lens, chunkwords, vout_idx, vout_val, dout_idx, dout_val, bitstream, end;
} szh_offsets;

#endif
