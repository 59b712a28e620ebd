// sz3_amd/csrc/sz3hip_stock_host.h — the container of a stock ALGO_INTERP stream (sz3hip_stock_host.cpp)
#ifndef SZ3HIP_STOCK_HOST_H
#define SZ3HIP_STOCK_HOST_H
#include <stdint.h>

#include <vector>

#include "sz3hip_internal.h"

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
void host_encode(const uint16_t *em, uint64_t n, const std::vector<uint8_t> &clen, const std::vector<uint64_t> &cbits, std::vector<uint8_t> &bits);
bool host_decode(const Tree &tr, int32_t offset, const uint8_t *bits, size_t nbytes, uint64_t n, uint16_t *em);
}  // namespace stock
#endif
