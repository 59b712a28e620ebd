// sz3_amd/csrc/sz3hip_stock_host.h — the container of a stock ALGO_INTERP stream (sz3hip_stock_host.cpp)
#ifndef SZ3HIP_STOCK_HOST_H
#define SZ3HIP_STOCK_HOST_H
#include <stdint.h>

#include <vector>

#include "sz3hip_internal.h"

namespace stock {
void serialise_interp(const szi_stock_params &p, uint64_t anchor_effective, const uint16_t *em, uint64_t n, const void *unpred, uint64_t n_unpred, size_t tsize,
                      std::vector<uint8_t> &raw);
bool parse_interp(const uint8_t *raw, size_t len, int N, size_t tsize, szi_stock_params &p, std::vector<uint16_t> &em, const uint8_t *&unpred, uint64_t &n_unpred);
}  // namespace stock
#endif
