// sz3_amd/csrc/sz3hip_h5z.cpp — HDF5 filter face (include/sz3hip_h5z.h): tools/H5Z-SZ3/src/H5Z_SZ3.cpp on top of the host-buffer API of
// this library — the plugin record and its two lookup symbols (:11-24), set_SZ3_conf_to_H5 / get_SZ3_conf_from_H5 (:26-72), the
// "set local" callback (:74-150) and the filter function (:154-227).
// No HDF5 header and no link-time dependency on libhdf5: a filter plugin always lives in a process that has libhdf5 loaded (HDF5
// dlopens the plugin), so the handful of HDF5 functions set_local needs are resolved at run time from the objects already loaded
// (h5sym below), exactly as this library finds libzstd and librccl. The few HDF5 types and constants used are part of its stable
// public ABI (H5public.h, H5Ipublic.h, H5Tpublic.h, H5Zpublic.h; restated in include/sz3hip_h5z.h).
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <link.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sz3hip.h"
#include "../../include/sz3hip_h5z.h"

namespace {
// SZ_FLOAT .. SZ_INT64 (include/SZ3/def.hpp:27-36): sizes of the ten element types of the reference's filter (H5Z_SZ3.cpp:195-227)
size_t elem_size(int dataType) {
    static const size_t sz[10] = {4, 8, 1, 1, 2, 2, 4, 4, 8, 8};
    return dataType >= 0 && dataType <= 9 ? sz[dataType] : 0;
}

// ---- the HDF5 functions set_local and the two conf helpers call, found among the objects this process has loaded -------------
typedef sz3hip_hid_t hid_t;
typedef int herr_t;
typedef unsigned long long hsize_t;
struct H5 {
    herr_t (*Pget_filter_by_id2)(hid_t, int, unsigned *, size_t *, unsigned *, size_t, char *, unsigned *);
    herr_t (*Pmodify_filter)(hid_t, int, unsigned, size_t, const unsigned *);
    herr_t (*Pset_filter)(hid_t, int, unsigned, size_t, const unsigned *);
    int (*Pget_nfilters)(hid_t);
    int (*Pget_filter2)(hid_t, unsigned, unsigned *, size_t *, unsigned *, size_t, char *, unsigned *);
    int (*Tget_class)(hid_t);
    size_t (*Tget_size)(hid_t);
    int (*Tget_sign)(hid_t);
    int (*Sget_simple_extent_dims)(hid_t, hsize_t *, hsize_t *);
    bool ok;
};
struct FindLib {
    void *handle;
};
int find_hdf5(struct dl_phdr_info *info, size_t, void *data) {
    // (an application that dlopen'ed libhdf5 privately — h5py's bundled copy — does not export it to RTLD_DEFAULT)
    FindLib *f = static_cast<FindLib *>(data);
    if (f->handle || !info->dlpi_name || !strstr(info->dlpi_name, "libhdf5")) return 0;
    if (strstr(info->dlpi_name, "libhdf5_hl") || strstr(info->dlpi_name, "libhdf5_cpp")) return 0;
    f->handle = dlopen(info->dlpi_name, RTLD_NOLOAD | RTLD_NOW);
    return 0;
}
void *h5sym(const char *name) {
    if (void *p = dlsym(RTLD_DEFAULT, name)) return p;
    static FindLib lib = {nullptr};
    if (!lib.handle) dl_iterate_phdr(find_hdf5, &lib);
    return lib.handle ? dlsym(lib.handle, name) : nullptr;
}
const H5 &h5() {  // (resolved on first use; looked for again while HDF5 has not been found: it may be loaded later)
    static H5 t = {};
    if (t.ok) return t;
    H5 r;
    memset(&r, 0, sizeof(r));
    r.Pget_filter_by_id2 = reinterpret_cast<decltype(r.Pget_filter_by_id2)>(h5sym("H5Pget_filter_by_id2"));
    r.Pmodify_filter = reinterpret_cast<decltype(r.Pmodify_filter)>(h5sym("H5Pmodify_filter"));
    r.Pset_filter = reinterpret_cast<decltype(r.Pset_filter)>(h5sym("H5Pset_filter"));
    r.Pget_nfilters = reinterpret_cast<decltype(r.Pget_nfilters)>(h5sym("H5Pget_nfilters"));
    r.Pget_filter2 = reinterpret_cast<decltype(r.Pget_filter2)>(h5sym("H5Pget_filter2"));
    r.Tget_class = reinterpret_cast<decltype(r.Tget_class)>(h5sym("H5Tget_class"));
    r.Tget_size = reinterpret_cast<decltype(r.Tget_size)>(h5sym("H5Tget_size"));
    r.Tget_sign = reinterpret_cast<decltype(r.Tget_sign)>(h5sym("H5Tget_sign"));
    r.Sget_simple_extent_dims = reinterpret_cast<decltype(r.Sget_simple_extent_dims)>(h5sym("H5Sget_simple_extent_dims"));
    r.ok = r.Pget_filter_by_id2 && r.Pmodify_filter && r.Pset_filter && r.Pget_nfilters && r.Pget_filter2 && r.Tget_class && r.Tget_size && r.Tget_sign &&
           r.Sget_simple_extent_dims;
    t = r;
    return t;
}
int complain(const char *what) {  // (the reference pushes onto HDF5's error stack, H5Z_SZ_PUSH_AND_GOTO; the return value is what HDF5 acts on)
    fprintf(stderr, "H5Z-SZ3 (libsz3hip): %s\n", what);
    return -1;
}
// Config::setDims (utils/Config.hpp:152-177): extents of 1 dropped, N / num / predDim / blockSize follow; every other field stays
bool conf_set_dims(sz3hip_config *c, int ndims, const hsize_t *dims) {
    int n = 0;
    uint64_t d[4] = {0, 0, 0, 0};
    for (int i = 0; i < ndims; i++) {
        if (dims[i] <= 1) continue;
        if (n == 4) return false;  // api/sz.hpp:71: "Data dimension higher than 4 is not supported."
        d[n++] = dims[i];
    }
    if (n == 0) d[n++] = 1;
    c->N = n;
    c->num = 1;
    for (int i = 0; i < 4; i++) c->dims[i] = i < n ? d[i] : 0;
    for (int i = 0; i < n; i++) c->num *= d[i];
    c->predDim = (uint8_t)n;
    c->blockSize = n == 1 ? 128 : (n == 2 ? 16 : 6);
    return true;
}
bool on_list(const H5 &f, hid_t plist) {  // does the property list's filter pipeline hold filter 32024? (asked without raising an HDF5 error)
    const int n = f.Pget_nfilters(plist);
    for (int i = 0; i < n; i++) {
        unsigned flags = 0, fconf = 0;
        size_t nel = 0;
        if (f.Pget_filter2(plist, (unsigned)i, &flags, &nel, nullptr, 0, nullptr, &fconf) == SZ3HIP_H5Z_FILTER_ID) return true;
    }
    return false;
}
}  // namespace

// set_SZ3_conf_to_H5 (H5Z_SZ3.cpp:26-52): Config::save bytes as the filter's cd_values on a dataset creation property list
extern "C" int sz3hip_h5z_conf_to_H5(sz3hip_hid_t propertyList, const sz3hip_config *conf) {
    const H5 &f = h5();
    if (!f.ok) return complain("the HDF5 library is not loaded in this process (H5P / H5T / H5S / H5Z functions not found)");
    unsigned char bytes[256];
    memset(bytes, 0, sizeof(bytes));
    const size_t real = sz3hip_config_save(conf, bytes);
    const size_t cd_nelmts = (real + sizeof(unsigned) - 1) / sizeof(unsigned);  // :38
    unsigned cd_values[64];
    memcpy(cd_values, bytes, cd_nelmts * sizeof(unsigned));
    // (:40-51 decides between modify and set by H5Zfilter_avail — whether the filter is REGISTERED with the library; what matters is whether
    // the LIST carries it already: modifying a filter that is not on the list fails, setting it twice breaks decompression, :46-47)
    if (on_list(f, propertyList)) {
        if (f.Pmodify_filter(propertyList, SZ3HIP_H5Z_FILTER_ID, 0 /* H5Z_FLAG_MANDATORY */, cd_nelmts, cd_values) < 0) return complain("failed to modify cd_values");
        return 1;
    }
    if (f.Pset_filter(propertyList, SZ3HIP_H5Z_FILTER_ID, 0, cd_nelmts, cd_values) < 0) return complain("failed to set the filter's cd_values");
    return 1;
}
// get_SZ3_conf_from_H5 (H5Z_SZ3.cpp:54-72): the Config the list's cd_values hold; a default Config when there are none
extern "C" int sz3hip_h5z_conf_from_H5(sz3hip_hid_t propertyList, sz3hip_config *conf) {
    const H5 &f = h5();
    if (!f.ok) return complain("the HDF5 library is not loaded in this process (H5P / H5T / H5S / H5Z functions not found)");
    uint64_t one = 1;
    sz3hip_config_init(conf, 1, &one);  // SZ3::Config conf; (:80)
    unsigned cd_values[64];
    memset(cd_values, 0, sizeof(cd_values));
    size_t cd_nelmts = 64;
    unsigned flags = 0, fconf = 0;
    if (!on_list(f, propertyList)) return 1;  // (:61 asks H5Zfilter_avail; a list without the filter has no cd_values either way)
    if (f.Pget_filter_by_id2(propertyList, SZ3HIP_H5Z_FILTER_ID, &flags, &cd_nelmts, cd_values, 0, nullptr, &fconf) < 0) return 1;
    if (cd_nelmts > 64) cd_nelmts = 64;
    if (cd_nelmts > 0) {
        sz3hip_config loaded;
        if (sz3hip_config_load_n(&loaded, reinterpret_cast<const unsigned char *>(cd_values), cd_nelmts * sizeof(unsigned)) != 0) *conf = loaded;
    }
    return 1;
}

// H5Z_sz3_set_local (H5Z_SZ3.cpp:74-150): called by HDF5 when a dataset with this filter is created — the Config in the list's
// cd_values (the user's bounds and algorithm, or none at all: h5py's compression=32024) gets the dataset's element type and the chunk's
// extents, and goes back into the list; the filter function then finds everything it needs in its cd_values.
extern "C" int sz3hip_h5z_set_local(sz3hip_hid_t dcpl_id, sz3hip_hid_t type_id, sz3hip_hid_t chunk_space_id) {
    const H5 &f = h5();
    if (!f.ok) return complain("the HDF5 library is not loaded in this process (H5P / H5T / H5S / H5Z functions not found)");
    sz3hip_config conf;
    if (sz3hip_h5z_conf_from_H5(dcpl_id, &conf) < 0) return -1;
    const int dclass = f.Tget_class(type_id);
    if (dclass < 0) return complain("not a datatype");
    const size_t dsize = f.Tget_size(type_id);
    if (dsize == 0) return complain("size is smaller than 0!");
    hsize_t dims_all[32];  // H5S_MAX_RANK
    const int ndims = f.Sget_simple_extent_dims(chunk_space_id, dims_all, nullptr);
    if (ndims < 0 || ndims > 32) return complain("not a data space");
    conf.dataType = SZ3HIP_FLOAT;
    if (dclass == 1) {  // H5T_FLOAT
        if (dsize != 4 && dsize != 8) return complain("floating-point types of 4 and 8 bytes are supported");
        conf.dataType = dsize == 4 ? SZ3HIP_FLOAT : SZ3HIP_DOUBLE;
    } else if (dclass == 0) {  // H5T_INTEGER
        const int dsign = f.Tget_sign(type_id);
        if (dsign < 0) return complain("Error in calling H5Tget_sign(type_id)....");
        const bool uns = dsign == 0;  // H5T_SGN_NONE
        switch (dsize) {
            case 1: conf.dataType = uns ? SZ3HIP_UINT8 : SZ3HIP_INT8; break;
            case 2: conf.dataType = uns ? SZ3HIP_UINT16 : SZ3HIP_INT16; break;
            case 4: conf.dataType = uns ? SZ3HIP_UINT32 : SZ3HIP_INT32; break;
            case 8: conf.dataType = uns ? SZ3HIP_UINT64 : SZ3HIP_INT64; break;
            default: return complain("integer types of 1, 2, 4 and 8 bytes are supported");
        }
    } else {
        complain("datatype class must be H5T_FLOAT or H5T_INTEGER");
        return 0;  // (:139: the reference's return value for this case)
    }
    if (!conf_set_dims(&conf, ndims, dims_all)) return complain("Data dimension higher than 4 is not supported.");
    // (:146-147 refresh sz3MagicNumber / sz3DataVer: fields of the container's header here, written by sz3hip_compress, not of the Config)
    return sz3hip_h5z_conf_to_H5(dcpl_id, &conf) < 0 ? -1 : 1;
}

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
    1, SZ3HIP_H5Z_FILTER_ID, 1, 1, "SZ3 compressor/decompressor for floating-point data (libsz3hip, MI355X).", nullptr, sz3hip_h5z_set_local, sz3hip_h5z_filter,
};
extern "C" int H5PLget_plugin_type(void) { return 0; }
extern "C" const void *H5PLget_plugin_info(void) { return &g_sz3hip_h5z_class; }
