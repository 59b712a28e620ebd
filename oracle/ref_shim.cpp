// oracle/ref_shim.cpp — TEST INFRASTRUCTURE ONLY.
// A few extern "C" entry points that instantiate the *reference's own* templates
// (SZ_compress<T>/SZ_decompress<T>, /root/reference/include/SZ3/api/sz.hpp:43,117) so that python/ctypes
// tests can drive the real reference with a full SZ3::Config (the reference's own C ABI, tools/sz3c,
// exposes only the error-bound mode and dims). This file contains no algorithm: it fills a Config
// and forwards.  Built only into oracle/_ref/libsz3ref.so by `make ref`.
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include "SZ3/api/sz.hpp"

namespace {
// fields the entry points' argument lists do not carry (set by ref_set_extra before a call, 0 = the Config's defaults)
int g_quantbin = 0;
double g_psnr = 0, g_l2norm = 0;
SZ3::Config make_conf(int N, const size_t *dims_slowest_first, int algo, int eb_mode, double abs_eb, double rel_eb,
                      int lorenzo, int lorenzo2, int regression, int openmp, int interp_algo, int block_size,
                      int interp_dir = -1, int anchor_stride = -2, double alpha = -2, double beta = -2) {
    std::vector<size_t> d(dims_slowest_first, dims_slowest_first + N);
    SZ3::Config conf;
    conf.setDims(d.begin(), d.end());
    conf.cmprAlgo = static_cast<uint8_t>(algo);
    conf.errorBoundMode = static_cast<uint8_t>(eb_mode);
    conf.absErrorBound = abs_eb;
    conf.relErrorBound = rel_eb;
    conf.lorenzo = lorenzo != 0;
    conf.lorenzo2 = lorenzo2 != 0;
    conf.regression = regression != 0;
    conf.openmp = openmp != 0;
    if (interp_algo >= 0) conf.interpAlgo = static_cast<uint8_t>(interp_algo);
    if (block_size > 0) conf.blockSize = block_size;
    if (interp_dir >= 0) conf.interpDirection = static_cast<uint8_t>(interp_dir);
    if (anchor_stride > -2) conf.interpAnchorStride = anchor_stride;
    if (alpha > -2) conf.interpAlpha = alpha;
    if (beta > -2) conf.interpBeta = beta;
    if (g_quantbin > 0) conf.quantbinCnt = g_quantbin;
    if (g_psnr > 0) conf.psnrErrorBound = g_psnr;
    if (g_l2norm > 0) conf.l2normErrorBound = g_l2norm;
    return conf;
}
template <class T>
size_t do_compress(const T *data, char *out, size_t cap, const SZ3::Config &conf, double *seconds) {
    auto t0 = std::chrono::steady_clock::now();
    size_t n = SZ_compress<T>(conf, data, out, cap);
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    return n;
}
}  // namespace

extern "C" {
// quantbinCnt, psnrErrorBound, l2normErrorBound of the Config the next calls fill (0 = leave the default); not part of ref_compress_bound
void ref_set_extra(int quantbin, double psnr, double l2norm) {
    g_quantbin = quantbin;
    g_psnr = psnr;
    g_l2norm = l2norm;
}
// dtype: 0 = f32, 1 = f64 (SZ_FLOAT / SZ_DOUBLE, include/SZ3/utils/Config.hpp:27-36)
size_t ref_compress(int dtype, const void *data, int N, const size_t *dims, int algo, int eb_mode, double abs_eb,
                    double rel_eb, int lorenzo, int lorenzo2, int regression, int openmp, int interp_algo, int block_size,
                    char *out, size_t cap, double *seconds) {
    try {
        SZ3::Config conf = make_conf(N, dims, algo, eb_mode, abs_eb, rel_eb, lorenzo, lorenzo2, regression, openmp,
                                     interp_algo, block_size);
        if (dtype == 0) return do_compress<float>(static_cast<const float *>(data), out, cap, conf, seconds);
        return do_compress<double>(static_cast<const double *>(data), out, cap, conf, seconds);
    } catch (std::exception &e) {
        fprintf(stderr, "ref_compress: %s\n", e.what());
        return 0;
    }
}
// same with the interpolation parameters of SZ3::Config (interpDirection, interpAnchorStride, interpAlpha, interpBeta)
size_t ref_compress_ex(int dtype, const void *data, int N, const size_t *dims, int algo, int eb_mode, double abs_eb,
                       double rel_eb, int lorenzo, int lorenzo2, int regression, int openmp, int interp_algo, int block_size,
                       int interp_dir, int anchor_stride, double alpha, double beta, char *out, size_t cap, double *seconds) {
    try {
        SZ3::Config conf = make_conf(N, dims, algo, eb_mode, abs_eb, rel_eb, lorenzo, lorenzo2, regression, openmp,
                                     interp_algo, block_size, interp_dir, anchor_stride, alpha, beta);
        if (dtype == 0) return do_compress<float>(static_cast<const float *>(data), out, cap, conf, seconds);
        return do_compress<double>(static_cast<const double *>(data), out, cap, conf, seconds);
    } catch (std::exception &e) {
        fprintf(stderr, "ref_compress_ex: %s\n", e.what());
        return 0;
    }
}
size_t ref_compress_bound(int dtype, int N, const size_t *dims) {
    SZ3::Config conf = make_conf(N, dims, 1, 0, 1e-3, 0, 1, 0, 0, 0, -1, 0);
    // the CLI allocates 2*num*sizeof(T) (tools/sz3/sz3.cpp:134); take the max with SZ_compress_size_bound
    size_t es = dtype == 0 ? 4 : 8;
    size_t b = dtype == 0 ? SZ3::SZ_compress_size_bound<float>(conf) : SZ3::SZ_compress_size_bound<double>(conf);
    return std::max(b, 2 * conf.num * es) + 4096;
}
// returns number of elements, fills dec (caller-allocated, num elements); 0 on error
size_t ref_decompress(int dtype, const char *cmp, size_t cmp_size, void *dec, double *seconds) {
    try {
        SZ3::Config conf;
        auto t0 = std::chrono::steady_clock::now();
        if (dtype == 0) {
            float *p = static_cast<float *>(dec);
            SZ_decompress<float>(conf, cmp, cmp_size, p);
        } else {
            double *p = static_cast<double *>(dec);
            SZ_decompress<double>(conf, cmp, cmp_size, p);
        }
        auto t1 = std::chrono::steady_clock::now();
        if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
        return conf.num;
    } catch (std::exception &e) {
        fprintf(stderr, "ref_decompress: %s\n", e.what());
        return 0;
    }
}
}
