// sz3_amd/csrc/sz3hip_stock_host.h — the container of a stock ALGO_INTERP stream (sz3hip_stock_host.cpp)
#ifndef SZ3HIP_STOCK_HOST_H
#define SZ3HIP_STOCK_HOST_H
#include <stdint.h>

#include <memory>
#include <vector>

#include "sz3hip_internal.h"
#include "sz3hip_stock_geom.h"

namespace stock {
struct Tree {                    // HuffmanEncoder's serialised tree (encoder/HuffmanEncoder.hpp:601-628)
    std::vector<uint32_t> L, R;  // children of node i in pre-order numbering (0: none)
    std::vector<int32_t> C;      // leaf: symbol - offset
    std::vector<uint8_t> t;      // 1: leaf
};
struct Code {
    uint64_t bits;
    uint32_t len;
};
bool book_from_hist(const uint64_t *hist65536, Tree &tr, std::vector<uint8_t> &clen, std::vector<uint64_t> &cbits, int &lo, int &hi);
void make_lut(const Tree &tr, std::vector<uint32_t> &lut);
void write_head(const szi_stock_params &p, uint64_t anchor_effective, const void *unpred, uint64_t n_unpred, size_t tsize, const Tree &tr, int lo, int hi,
                uint64_t n, uint64_t bit_bytes, std::vector<uint8_t> &raw);
bool parse_head(const uint8_t *raw, size_t len, int N, size_t tsize, szi_stock_params &p, const uint8_t *&unpred, uint64_t &n_unpred, Tree &tr, int32_t &offset,
                uint64_t &n, const uint8_t *&bits, uint64_t &bit_bytes);
// a stock ALGO_LORENZO_REG stream (round 4, read side): what BlockwiseDecomposition / ComposedPredictor / RegressionPredictor saved
// in front of the main code stream (compressor/SZGenericCompressor.hpp:38-84, predictor/ComposedPredictor.hpp:52-78,
// predictor/RegressionPredictor.hpp:94-123, quantizer/LinearQuantizer.hpp:95-122)
struct Quant {  // a LinearQuantizer as saved: bound, radius, its unpredictable values in the order their zero codes occur
    double eb = 0;
    int32_t radius = 0;
    const uint8_t *unpred = nullptr;
    uint64_t n_unpred = 0;
};
struct LorenzoReg {
    std::vector<uint16_t> coef_codes;  // regression: N + 1 codes per regression block, in block order
    Quant q_indep, q_lin;              // ... and their two quantizers
    std::vector<uint16_t> selection;   // composed sets: the chosen member per block (index into the set's order)
    Quant q;                           // the main quantizer
    Tree tree;                         // the main code stream
    int32_t offset = 0;
    uint64_t n = 0;
    const uint8_t *bits = nullptr;
    uint64_t bit_bytes = 0;
};
bool parse_lorenzo_reg(const uint8_t *raw, size_t len, size_t tsize, bool has_regression, bool composed, uint64_t nblocks, int N, LorenzoReg &out);
// ... and the WRITE side (round 5): the regression coefficients' chain (the fits of the chosen blocks in, what the reader recovers out) and
// the buffer in front of the main bit stream
template <typename T>
void lorenzo_reg_chain(int N, uint32_t B, double eb, const uint8_t *kind, uint64_t nblocks, T *coef, std::vector<uint16_t> &codes, std::vector<T> &un_indep,
                       std::vector<T> &un_lin);
void write_lorenzo_reg_head(int N, uint32_t B, double eb, size_t tsize, bool has_regression, bool composed, const std::vector<uint16_t> &coef_codes,
                            const void *un_indep, uint64_t n_un_indep, const void *un_lin, uint64_t n_un_lin, const std::vector<uint16_t> &selection, int32_t radius,
                            const void *unpred, uint64_t n_unpred, const Tree &tr, int lo, int hi, uint64_t n, uint64_t bit_bytes, std::vector<uint8_t> &raw);
// 1-D arrays (round 5): the chain of roundings is walked on the host, both ways (sz3hip_stock_host.cpp says why)
template <typename T>
bool lorenzo_reg_read_1d(uint64_t n, uint32_t B, double eb, int radius, const uint16_t *codes, const uint8_t *kind, const T *coef, const T *unpred, uint64_t n_unpred,
                         T *out);
template <typename T>
void lorenzo_reg_write_1d(uint64_t n, uint32_t B, double eb, int radius, uint32_t set_mask, T *data, std::vector<uint16_t> &codes, std::vector<T> &unpred,
                          std::vector<uint16_t> &selection, std::vector<uint16_t> &coef_codes, std::vector<T> &un_indep, std::vector<T> &un_lin);
bool encode_codes_host(const std::vector<uint16_t> &codes, Tree &tr, int &lo, int &hi, std::vector<uint8_t> &bits);
void host_encode(const uint16_t *em, uint64_t n, const std::vector<uint8_t> &clen, const std::vector<uint64_t> &cbits, std::vector<uint8_t> &bits);
bool host_decode(const Tree &tr, int32_t offset, const uint8_t *bits, size_t nbytes, uint64_t n, uint16_t *em);
// one trial of the ALGO_INTERP_LORENZO tuner priced the reference's way (sz3hip_ctx_set_tuner_exact): the buffer interp_compress_test
// hands to zstd (api/impl/SZAlgoInterp.hpp:42-78), from the trial kernel's per-element codes of all sampled blocks
template <typename T>
bool trial_buffer(const szi_stock_params &p, const uint16_t *codes, const T *samples, uint64_t nb, std::vector<uint8_t> &raw);
// ... the same in steps, for a caller that spreads a group of trials over host threads: prepare(); order(0), order(1) in any order or at once;
// book(); bits(0), bits(1) likewise; finish(raw)
template <typename T>
struct TrialWork {
    szi_stock_params p;
    const uint16_t *codes = nullptr;  // [nb][per] per element
    const T *samples = nullptr;       // [nb][per]
    uint64_t nb = 0;
    szg_geom g;
    uint64_t per = 0;
    std::shared_ptr<const std::vector<uint32_t>> perm;  // the element emitted r-th
    std::vector<uint16_t> em;                           // codes in emission order, all blocks
    std::vector<T> un[2];                               // the quantizer's list, by half of the blocks
    std::vector<uint32_t> cnt[2];                       // symbol counts, by half
    uint32_t lo = 0, hi = 0;
    Tree tr;
    std::vector<Code> cw;
    std::vector<uint8_t> part[2];                       // the bit stream's halves
    uint64_t pbits[2] = {0, 0};
    bool prepare();
    void order(int h);
    bool book();
    void bits(int h);
    void finish(std::vector<uint8_t> &raw);
};
// ... and the 1-D Lorenzo trial's (lorenzo_compress_test, :80-120), walked on the host over the sampled blocks
template <typename T>
bool lorenzo_trial_buffer(double eb, int radius, const T *samples, uint64_t per, uint64_t nb, std::vector<uint8_t> &raw);
}  // namespace stock
#endif
