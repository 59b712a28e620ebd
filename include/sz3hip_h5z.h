/* include/sz3hip_h5z.h — the HDF5 dynamically-loaded-filter face of libsz3hip.so (filter id 32024, the reference's).
 *
 * Replaces /root/reference/tools/H5Z-SZ3/src/H5Z_SZ3.cpp:
 *   :11-20   the filter's class record (H5Z_class2_t: version, id, encoder / decoder present, name, can_apply, set_local, filter)
 *   :22-24   H5PLget_plugin_type / H5PLget_plugin_info — the two symbols HDF5 looks up in a plugin directory
 *   :154-168 process_data<T>: buffers swapped in place, malloc / free as HDF5's filter contract wants
 *   :179-227 H5Z_filter_sz3: cd_values = the bytes of Config::save; cd_nelmts == 0 and conf.num < 20 pass the chunk through
 *   :26-72   set_SZ3_conf_to_H5 / get_SZ3_conf_from_H5 (declared in tools/H5Z-SZ3/include/H5Z_SZ3.hpp:51,53): a Config to / from the
 *            filter's cd_values on a dataset creation property list — sz3hip_h5z_conf_to_H5 / _from_H5 here (C ABI, the POD config);
 *            include/H5Z_SZ3.hpp gives them the reference's names and SZ3::Config& signatures
 *   :74-150  H5Z_sz3_set_local: the dataset's element type (all ten: 8 ... 64-bit integers signed and unsigned, float, double) and
 *            the chunk's extents go into the Config in cd_values when a dataset is created — what makes plain
 *            `create_dataset(..., compression=32024, compression_opts=...)` work (tools/test/integration/test_h5_filter.py:19-35)
 * HDF5 is not in this image and the library does not link against it: the record's layout is restated below from HDF5's public,
 * stable plugin ABI (H5Zpublic.h: H5Z_class2_t, H5Z_CLASS_T_VERS = 1, H5Z_FLAG_REVERSE = 0x0100; H5PLpublic.h: H5PL_TYPE_FILTER = 0;
 * H5Ipublic.h: hid_t = int64_t since 1.10; H5Tpublic.h: H5T_INTEGER = 0, H5T_FLOAT = 1, H5T_SGN_NONE = 0), and the HDF5 functions
 * set_local calls (H5Pget_nfilters, H5Pget_filter2, H5Pget_filter_by_id2, H5Pmodify_filter, H5Pset_filter, H5Tget_class / _size /
 * _sign, H5Sget_simple_extent_dims) are looked up at run time in the process that loaded the plugin — it has libhdf5 by
 * construction. The library can be dropped into HDF5_PLUGIN_PATH as it is. What differs from the reference's filter: any failure
 * returns 0 — "filter failed" in HDF5's contract — instead of calling exit(). Chunks are compressed on HIP device 0 through
 * sz3hip_compress / sz3hip_decompress (host buffers in, host buffers out; integers ride the f64 pipeline, exact); without a device
 * the filter fails (no CPU path). tests/h5stub/ holds a small stand-in for libhdf5 (the nine functions above over an in-memory
 * property list) that the tests drive set_local through. */
#ifndef SZ3HIP_H5Z_H
#define SZ3HIP_H5Z_H
#include <stddef.h>

#include "sz3hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SZ3HIP_H5Z_FILTER_ID 32024 /* H5Z_FILTER_SZ3, tools/H5Z-SZ3/include/H5Z_SZ3.hpp:4 */
#define SZ3HIP_H5Z_FLAG_REVERSE 0x0100u /* H5Z_FLAG_REVERSE: the filter runs in the read direction */

typedef long long sz3hip_hid_t; /* hid_t: a 64-bit integer since HDF5 1.10 */
typedef size_t (*sz3hip_h5z_func_t)(unsigned int flags, size_t cd_nelmts, const unsigned int cd_values[], size_t nbytes, size_t *buf_size,
                                    void **buf);
/* H5Z_class2_t (H5Zpublic.h): hid_t arguments of the two callbacks are 64-bit integers since HDF5 1.10 */
typedef struct sz3hip_h5z_class2 {
    int version;                  /* H5Z_CLASS_T_VERS = 1 */
    int id;                       /* H5Z_filter_t */
    unsigned int encoder_present; /* 1 */
    unsigned int decoder_present; /* 1 */
    const char *name;
    int (*can_apply)(sz3hip_hid_t dcpl_id, sz3hip_hid_t type_id, sz3hip_hid_t space_id); /* NULL, as in the reference */
    int (*set_local)(sz3hip_hid_t dcpl_id, sz3hip_hid_t type_id, sz3hip_hid_t space_id); /* sz3hip_h5z_set_local */
    sz3hip_h5z_func_t filter;
} sz3hip_h5z_class2;

int H5PLget_plugin_type(void);          /* H5PL_TYPE_FILTER = 0 */
const void *H5PLget_plugin_info(void);  /* -> the sz3hip_h5z_class2 record */
/* H5Z_sz3_set_local (H5Z_SZ3.cpp:74-150): > 0 on success, < 0 on failure (0: an element class that is neither integer nor float) */
int sz3hip_h5z_set_local(sz3hip_hid_t dcpl_id, sz3hip_hid_t type_id, sz3hip_hid_t chunk_space_id);
/* set_SZ3_conf_to_H5 / get_SZ3_conf_from_H5 (H5Z_SZ3.cpp:26-72) over the POD config: 1 on success, < 0 on failure. `from` leaves a
 * default Config (SZ3::Config()) when the list does not carry the filter or carries it without cd_values */
int sz3hip_h5z_conf_to_H5(sz3hip_hid_t propertyList, const sz3hip_config *conf);
int sz3hip_h5z_conf_from_H5(sz3hip_hid_t propertyList, sz3hip_config *conf);
/* the filter function itself (what the record's `filter` points to), callable without HDF5 */
size_t sz3hip_h5z_filter(unsigned int flags, size_t cd_nelmts, const unsigned int cd_values[], size_t nbytes, size_t *buf_size, void **buf);

#ifdef __cplusplus
}
#endif
#endif
