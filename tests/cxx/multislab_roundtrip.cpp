// tests/cxx/multislab_roundtrip.cpp — the slab-parallel path through the C++ face (include/SZ3/api/sz.hpp), written the
// way the reference's smoke test drives it (tools/sz3/sz3_smoke_test.cpp:10-52: conf.openmp = true, SZ_compress<T>,
// SZ_decompress<T>, max error <= bound). argv: nz ny nx eb [algo]; prints "ok <ratio> <max_err> <openmp bit of the stream>".
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "SZ3/api/sz.hpp"

template <class T>
int run(size_t nz, size_t ny, size_t nx, double eb, int algo) {
    std::vector<T> data(nz * ny * nx);
    for (size_t z = 0; z < nz; z++)
        for (size_t y = 0; y < ny; y++)
            for (size_t x = 0; x < nx; x++)
                data[(z * ny + y) * nx + x] = (T)(std::sin(0.11 * x) * std::cos(0.07 * y) * std::sin(0.05 * z) + 1e-3 * ((x * 7 + y * 13 + z * 29) % 17));
    SZ3::Config conf({nz, ny, nx});
    conf.cmprAlgo = (uint8_t)algo;
    conf.errorBoundMode = SZ3::EB_ABS;
    conf.absErrorBound = eb;
    conf.openmp = true;
    size_t cmpSize = 0;
    char *cmp = SZ_compress<T>(conf, data.data(), cmpSize);
    SZ3::Config conf2;
    T *dec = nullptr;
    SZ_decompress<T>(conf2, cmp, cmpSize, dec);
    double err = 0;
    for (size_t i = 0; i < data.size(); i++) err = std::max(err, std::fabs((double)dec[i] - (double)data[i]));
    const bool ok = err <= eb && conf2.num == data.size() && conf2.openmp;
    printf("%s %.4f %.6g %d\n", ok ? "ok" : "FAIL", (double)(data.size() * sizeof(T)) / (double)cmpSize, err, (int)conf2.openmp);
    delete[] cmp;
    delete[] dec;
    return ok ? 0 : 1;
}

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const size_t nz = atoll(argv[1]), ny = atoll(argv[2]), nx = atoll(argv[3]);
    const double eb = atof(argv[4]);
    const int algo = argc > 5 ? atoi(argv[5]) : SZ3::ALGO_LORENZO_REG;
    const bool f64 = argc > 6 && argv[6][0] == 'd';
    try {
        return f64 ? run<double>(nz, ny, nx, eb, algo) : run<float>(nz, ny, nx, eb, algo);
    } catch (std::exception &e) {
        printf("EXCEPTION %s\n", e.what());
        return 3;
    }
}
