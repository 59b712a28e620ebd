// sz3_amd/csrc/sz3hip_h5z.cpp — HDF5 filter face (include/sz3hip_h5z.h): tools/H5Z-SZ3/src/H5Z_SZ3.cpp:11-24, 154-227 on top of the
// host-buffer API of this library. No HDF5 header is needed: HDF5 finds a plugin by the two H5PLget_* symbols and calls the filter
// through the record they return.
#include <stdlib.h>
#include <string.h>

#include "../../include/sz3hip.h"
#include "../../include/sz3hip_h5z.h"

namespace {
// SZ_FLOAT .. SZ_INT64 (include/SZ3/def.hpp:27-36): the element types the library has a path for, and their sizes
size_t elem_size(int dataType) {
    switch (dataType) {
        case 0: return 4;  // SZ_FLOAT
        case 1: return 8;  // SZ_DOUBLE
        case 7: return 4;  // SZ_INT32
        case 9: return 8;  // SZ_INT64
        default: return 0;
    }
}
}  // namespace

extern "C" size_t sz3hip_h5z_filter(unsigned int flags, size_t cd_nelmts, const unsigned int cd_values[], size_t nbytes, size_t *buf_size,
                                    void **buf) {
    if (cd_nelmts == 0) return nbytes;  // H5Z_SZ3.cpp:183-184: special data (strings): not values
    if (!cd_values || !buf || !*buf || !buf_size) return 0;
    sz3hip_config conf;
    if (sz3hip_config_load_n(&conf, reinterpret_cast<const unsigned char *>(cd_values), cd_nelmts * sizeof(unsigned int)) == 0) return 0;
    if (conf.num < 20) return nbytes;  // :192
    const size_t esz = elem_size(conf.dataType);
    if (esz == 0) return 0;
    if (flags & SZ3HIP_H5Z_FLAG_REVERSE) {  // process_data<T>, :156-161
        sz3hip_config in_stream;
        if (sz3hip_peek_config(&in_stream, static_cast<const char *>(*buf), nbytes) != 0) return 0;
        if (in_stream.num != conf.num) return 0;  // (the chunk the stream holds is not the chunk cd_values describe)
        void *out = malloc(conf.num * esz);
        if (!out) return 0;
        sz3hip_config c = conf;
        if (sz3hip_decompress(&c, conf.dataType, static_cast<const char *>(*buf), nbytes, out) != 0) {
            free(out);
            return 0;
        }
        free(*buf);
        *buf = out;
        *buf_size = conf.num * esz;
        return *buf_size;
    }
    if (nbytes < conf.num * esz) return 0;  // (fewer valid bytes than the Config's elements)
    const size_t cap = sz3hip_compress_bound(&conf, conf.dataType) > 2 * esz * conf.num ? sz3hip_compress_bound(&conf, conf.dataType) : 2 * esz * conf.num;  // :163
    char *out = static_cast<char *>(malloc(cap));
    if (!out) return 0;
    const size_t size = sz3hip_compress(&conf, conf.dataType, *buf, out, cap);
    if (size == 0) {
        free(out);
        return 0;
    }
    free(*buf);
    *buf = out;
    *buf_size = size;  // (the reference reports the valid bytes as the buffer's size too, :164)
    return size;
}

static const sz3hip_h5z_class2 g_sz3hip_h5z_class = {
    1, SZ3HIP_H5Z_FILTER_ID, 1, 1, "SZ3 compressor/decompressor for floating-point data (libsz3hip, MI355X).", nullptr, nullptr, sz3hip_h5z_filter,
};
extern "C" int H5PLget_plugin_type(void) { return 0; }
extern "C" const void *H5PLget_plugin_info(void) { return &g_sz3hip_h5z_class; }
