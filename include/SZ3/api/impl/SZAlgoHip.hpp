// include/SZ3/api/impl/SZAlgoHip.hpp — the MI355X path as ONE MORE ALGORITHM of a stock SZ3 tree.
//
// This header is for the REFERENCE's own include tree (szcompressor/SZ3 v3.3.x), following the recipe its authors document in
// tools/sz3/sz3_customized_demo.cpp:8-14 ("1. add a new ALGO in SZ3 Config, 2. add a new hpp in include/SZ3/api/impl/,
// 3. dispatch it in SZDispatcher.hpp, 4. select it with -c"). It is NOT used by this repository's own C++ face
// (include/SZ3/api/sz.hpp replaces the whole tree instead). The three edits a maintainer makes next to dropping this file in:
//
//   include/SZ3/utils/Config.hpp:80    enum ALGO { ..., ALGO_BIOMDXTC, ALGO_HIP_LORENZO = 16, ALGO_HIP_INTERP = 17 };
//                              :93-98  ALGO_MAP: {"ALGO_HIP_LORENZO", ALGO_HIP_LORENZO}, {"ALGO_HIP_INTERP", ALGO_HIP_INTERP},
//   include/SZ3/api/impl/SZDispatcher.hpp:4     #include "SZ3/api/impl/SZAlgoHip.hpp"
//                                        :38    } else if (conf.cmprAlgo == ALGO_HIP_LORENZO || conf.cmprAlgo == ALGO_HIP_INTERP) {
//                                                   cmpSize = SZ_compress_Hip<T, N>(conf, dataCopy.data(), cmpData, cmpCap);
//                                        :95    } else if (conf.cmprAlgo == ALGO_HIP_LORENZO || conf.cmprAlgo == ALGO_HIP_INTERP) {
//                                                   SZ_decompress_Hip<T, N>(conf, cmpData, cmpSize, decData);
//   link with -lsz3hip (sz3_amd/libsz3hip.so; include/sz3hip.h on the include path).
// `make -C oracle algohip` applies exactly these edits to a scratch copy of the reference's two headers and builds the
// reference's CLI with them (oracle/_ref/sz3_algohip); tests/test_capi_cpu.py and tests/test_gpu_parity.py drive it.
//
// ALGO_HIP_LORENZO asks for the Lorenzo / regression family (the reference's ALGO_LORENZO_REG with conf.lorenzo /
// lorenzo2 / regression), ALGO_HIP_INTERP for the reference's default (ALGO_INTERP_LORENZO: sampling tuner, then
// interpolation). The same two ids name the streams in the trailer, so a file written through this path opens through it.
// The dispatcher's own policies stay in force around the call (eb == 0, buffer too small, ratio < 3: SZDispatcher.hpp:17-74);
// the GPU library's internal fallback to a lossless stream is reported by setting conf.cmprAlgo = ALGO_LOSSLESS, which is
// what the dispatcher itself does (:57).
#ifndef SZ3_ALGO_HIP_HPP
#define SZ3_ALGO_HIP_HPP

#include <cstring>
#include <stdexcept>
#include <type_traits>

#include "SZ3/utils/Config.hpp"
#include "sz3hip.h"

namespace SZ3 {
namespace hip_detail {
template <class T>
constexpr int dtype_id() {
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value || std::is_same<T, int32_t>::value ||
                      std::is_same<T, int64_t>::value,
                  "the HIP path takes float, double, int32, int64");
    return std::is_same<T, float>::value ? SZ3HIP_FLOAT : std::is_same<T, double>::value ? SZ3HIP_DOUBLE
           : std::is_same<T, int32_t>::value ? SZ3HIP_INT32 : SZ3HIP_INT64;
}
inline sz3hip_config to_pod(const Config &c) {
    sz3hip_config p;
    uint64_t d[4] = {1, 1, 1, 1};
    const int nd = static_cast<int>(c.dims.size() > 4 ? 4 : c.dims.size());
    for (int i = 0; i < nd; i++) d[i] = c.dims[i];
    sz3hip_config_init(&p, nd, d);
    p.cmprAlgo = c.cmprAlgo;
    p.errorBoundMode = c.errorBoundMode;
    p.absErrorBound = c.absErrorBound;
    p.relErrorBound = c.relErrorBound;
    p.psnrErrorBound = c.psnrErrorBound;
    p.l2normErrorBound = c.l2normErrorBound;
    p.openmp = 0;  // (slabs are the reference's SZ_compress_OMP's business on this route)
    p.quantbinCnt = c.quantbinCnt;
    p.blockSize = c.blockSize;
    p.predDim = c.predDim;
    p.dataType = c.dataType;
    p.lorenzo = c.lorenzo;
    p.lorenzo2 = c.lorenzo2;
    p.regression = c.regression;
    p.regression2 = c.regression2;
    p.interpAlgo = c.interpAlgo;
    p.interpDirection = c.interpDirection;
    p.interpAnchorStride = c.interpAnchorStride;
    p.interpAlpha = c.interpAlpha;
    p.interpBeta = c.interpBeta;
    return p;
}
[[noreturn]] inline void raise_last() {
    const int code = sz3hip_last_error_code();
    const char *msg = sz3hip_last_error();
    if (code == SZ3HIP_ECAPACITY) throw std::length_error(msg);  // (the dispatcher turns this one into the lossless fallback, :44-51)
    if (code == SZ3HIP_EINVAL) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}
}  // namespace hip_detail

template <class T, uint N>
size_t SZ_compress_Hip(Config &conf, T *data, uchar *cmpData, size_t cmpCap) {
    sz3hip_config p = hip_detail::to_pod(conf);
    const bool interp = conf.cmprAlgo == 17;
    p.cmprAlgo = interp ? SZ3HIP_ALGO_INTERP_LORENZO : SZ3HIP_ALGO_LORENZO_REG;
    const size_t n = sz3hip_compress_blob(&p, hip_detail::dtype_id<T>(), data, reinterpret_cast<char *>(cmpData), cmpCap);
    if (n == 0) hip_detail::raise_last();
    // what the library resolved: the absolute bound (calAbsErrorBound rewrites conf the same way) and the stream's id
    conf.errorBoundMode = p.errorBoundMode;
    conf.absErrorBound = p.absErrorBound;
    conf.cmprAlgo = p.cmprAlgo;
    conf.lorenzo = p.lorenzo;
    conf.lorenzo2 = p.lorenzo2;
    conf.regression = p.regression;
    conf.dataType = p.dataType;  // (the reference leaves Config::dataType at its default; the trailer then names what the blob holds)
    return n;
}

template <class T, uint N>
void SZ_decompress_Hip(const Config &conf, const uchar *cmpData, size_t cmpSize, T *decData) {
    sz3hip_config p = hip_detail::to_pod(conf);
    p.dataType = (uint8_t)hip_detail::dtype_id<T>();  // the caller's T decides, as in the reference (streams written before dataType was recorded say 0)
    if (sz3hip_decompress_blob(&p, hip_detail::dtype_id<T>(), reinterpret_cast<const char *>(cmpData), cmpSize, decData)) hip_detail::raise_last();
}
}  // namespace SZ3
#endif
