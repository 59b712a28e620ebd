// include/H5Z_SZ3.hpp — the reference's header name for its HDF5 filter (tools/H5Z-SZ3/include/H5Z_SZ3.hpp): the filter id and the two
// helpers an application uses to put a Config into / read it from a dataset creation property list (:51,53; H5Z_SZ3.cpp:26-72), with
// the reference's names and signatures, over libsz3hip's C ABI (include/sz3hip_h5z.h: sz3hip_h5z_conf_to_H5 / _from_H5). The filter
// itself — H5PLget_plugin_type / _info, the set_local callback, the filter function — lives in libsz3hip.so: put the library (or a
// link to it) into HDF5_PLUGIN_PATH and HDF5 finds it by id 32024.
// With HDF5's own headers on the include path (hdf5.h) hid_t / herr_t are HDF5's; without them they are restated from its stable
// public ABI (H5Ipublic.h: int64_t since 1.10; H5public.h: int).
#ifndef SZ3_H5Z_SZ3_H
#define SZ3_H5Z_SZ3_H

#define H5Z_FILTER_SZ3 32024

#include <cstdint>

#include "SZ3/api/sz.hpp"
#include "sz3hip_h5z.h"

#if defined(__has_include)
#if __has_include("hdf5.h")
#include "hdf5.h"
#define SZ3HIP_HAVE_HDF5_H 1
#endif
#endif
#ifndef SZ3HIP_HAVE_HDF5_H
typedef int64_t hid_t;
typedef int herr_t;
#endif

inline herr_t set_SZ3_conf_to_H5(const hid_t propertyList, SZ3::Config &conf) {  // H5Z_SZ3.cpp:26-52
    const sz3hip_config pod = conf.to_pod();
    return (herr_t)sz3hip_h5z_conf_to_H5((sz3hip_hid_t)propertyList, &pod);
}
inline herr_t get_SZ3_conf_from_H5(const hid_t propertyList, SZ3::Config &conf) {  // H5Z_SZ3.cpp:54-72
    sz3hip_config pod;
    const int rc = sz3hip_h5z_conf_from_H5((sz3hip_hid_t)propertyList, &pod);
    if (rc > 0) conf.from_pod(pod);
    return (herr_t)rc;
}

#endif  // SZ3_H5Z_SZ3_H
