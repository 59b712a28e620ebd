/* include/sz3hip_h5z.h — the HDF5 dynamically-loaded-filter face of libsz3hip.so (filter id 32024, the reference's).
 *
 * Replaces /root/reference/tools/H5Z-SZ3/src/H5Z_SZ3.cpp:
 *   :11-20   the filter's class record (H5Z_class2_t: version, id, encoder / decoder present, name, can_apply, set_local, filter)
 *   :22-24   H5PLget_plugin_type / H5PLget_plugin_info — the two symbols HDF5 looks up in a plugin directory
 *   :154-168 process_data<T>: buffers swapped in place, malloc / free as HDF5's filter contract wants
 *   :179-227 H5Z_filter_sz3: cd_values = the bytes of Config::save; cd_nelmts == 0 and conf.num < 20 pass the chunk through
 * HDF5 is not in this image: the record's layout is restated below from HDF5's public, stable plugin ABI (H5Zpublic.h: H5Z_class2_t,
 * H5Z_CLASS_T_VERS = 1, H5Z_FLAG_REVERSE = 0x0100; H5PLpublic.h: H5PL_TYPE_FILTER = 0) so that the library can be dropped into
 * HDF5_PLUGIN_PATH as it is. What differs from the reference's filter: element types this library has no path for (8 / 16-bit and
 * unsigned integers) and any failure return 0 — "filter failed" in HDF5's contract — instead of calling exit(). Chunks are
 * compressed on HIP device 0 through sz3hip_compress / sz3hip_decompress (host buffers in, host buffers out); without a device
 * the filter fails (no CPU path). */
#ifndef SZ3HIP_H5Z_H
#define SZ3HIP_H5Z_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SZ3HIP_H5Z_FILTER_ID 32024 /* H5Z_FILTER_SZ3, tools/H5Z-SZ3/include/H5Z_SZ3.hpp:4 */
#define SZ3HIP_H5Z_FLAG_REVERSE 0x0100u /* H5Z_FLAG_REVERSE: the filter runs in the read direction */

typedef size_t (*sz3hip_h5z_func_t)(unsigned int flags, size_t cd_nelmts, const unsigned int cd_values[], size_t nbytes, size_t *buf_size,
                                    void **buf);
/* H5Z_class2_t (H5Zpublic.h): hid_t arguments of the two callbacks are 64-bit integers since HDF5 1.10 */
typedef struct sz3hip_h5z_class2 {
    int version;                  /* H5Z_CLASS_T_VERS = 1 */
    int id;                       /* H5Z_filter_t */
    unsigned int encoder_present; /* 1 */
    unsigned int decoder_present; /* 1 */
    const char *name;
    int (*can_apply)(long long dcpl_id, long long type_id, long long space_id); /* NULL, as in the reference */
    int (*set_local)(long long dcpl_id, long long type_id, long long space_id); /* NULL here: the application stores Config::save bytes
                                                                                   with H5Pset_filter (the reference's set_local,
                                                                                   H5Z_SZ3.cpp:77-150, derives them from the
                                                                                   dataset's type and chunk shape and needs HDF5) */
    sz3hip_h5z_func_t filter;
} sz3hip_h5z_class2;

int H5PLget_plugin_type(void);          /* H5PL_TYPE_FILTER = 0 */
const void *H5PLget_plugin_info(void);  /* -> the sz3hip_h5z_class2 record */
/* the filter function itself (what the record's `filter` points to), callable without HDF5 */
size_t sz3hip_h5z_filter(unsigned int flags, size_t cd_nelmts, const unsigned int cd_values[], size_t nbytes, size_t *buf_size, void **buf);

#ifdef __cplusplus
}
#endif
#endif
