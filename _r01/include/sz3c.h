/*
 * include/sz3c.h — the SZ2/SZ3 C ABI as exported by libsz3hip.so.
 *
 * Same symbol names, argument order/meaning, memory ownership and failure behaviour as the reference's
 * tools/sz3c/include/sz3c.h:52-59 and tools/sz3c/src/sz3c.cpp (file:line cited per entry), so that a C/Fortran/
 * ctypes caller of libSZ3c (e.g. tools/pysz/deprecated/cpysz.py:23-27) can be pointed at libsz3hip.so unchanged.
 * The work is done on the MI355X (sz3_amd/csrc/sz3hip_kernels.hip); there is no CPU fallback.
 */
#ifndef SZ3HIP_SZ3C_H
#define SZ3HIP_SZ3C_H
#include <stddef.h>
#include <stdio.h>

/* error-bound modes of SZ2 (reference sz3c.h:9-22); only the first four are accepted (sz3c.cpp:30-41) */
#define ABS 0
#define REL 1
#define VR_REL 1
#define ABS_AND_REL 2
#define ABS_OR_REL 3
#define PSNR 4
#define NORM 5
#define PW_REL 10
#define ABS_AND_PW_REL 11
#define ABS_OR_PW_REL 12
#define REL_AND_PW_REL 13
#define REL_OR_PW_REL 14

/* data types of SZ2 (reference sz3c.h:25-36); SZ_FLOAT and SZ_DOUBLE are accepted (sz3c.cpp:44-53) */
#define SZ_FLOAT 0
#define SZ_DOUBLE 1
#define SZ_UINT8 2
#define SZ_INT8 3
#define SZ_UINT16 4
#define SZ_INT16 5
#define SZ_UINT32 6
#define SZ_INT32 7
#define SZ_UINT64 8
#define SZ_INT64 9

#ifdef __cplusplus
extern "C" {
#endif

/* sz3c.h:52 / sz3c.cpp:11-61. r1 is the fastest dimension, unused leading dims are 0, 5-D is folded r5*r4
 * (sz3c.cpp:24). Returns malloc'ed memory of *outSize bytes (release with free_buf); unsupported mode or type
 * => printf + exit(0) exactly like sz3c.cpp:39-40,51-52. Default algorithm of a fresh Config. */
unsigned char *SZ_compress_args(int dataType, void *data, size_t *outSize, int errBoundMode, double absErrBound,
                                double relBoundRatio, double pwrBoundRatio, size_t r5, size_t r4, size_t r3, size_t r2,
                                size_t r1);

/* sz3c.h:56 / sz3c.cpp:63-92. Returns malloc'ed memory holding the decompressed array. */
void *SZ_decompress(int dataType, unsigned char *bytes, size_t byteLength, size_t r5, size_t r4, size_t r3, size_t r2,
                    size_t r1);

/* sz3c.h:59 / sz3c.cpp:94 */
void free_buf(void *p);

#ifdef __cplusplus
}
#endif
#endif
