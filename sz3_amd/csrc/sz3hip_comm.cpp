// sz3_amd/csrc/sz3hip_comm.cpp — the path's only exchange, in C++ over RCCL (xGMI): a communicator object and the
// collectives the slab-parallel path needs:
//   * sum all-reduce of the 65536 x u64 code histogram between stage 1 and stage 2 (one code book for all slabs)
//   * min / max all-reduce of one value each for the range-based error bounds (api/impl/SZImplOMP.hpp:57-69)
// Two ways to build a communicator, the rest of the code does not care which:
//   sz3hip_comm_create_local  one process drives all its GPUs (ncclCommInitAll): what sz3hip_compress does for conf.openmp
//   sz3hip_comm_create_rank   one process per GPU (ncclCommInitRank): the launcher ships the 128-byte id of rank 0
// librccl is dlopen'ed (like libzstd) so that the library still loads, and the CPU-side tests still run, where RCCL is
// absent; using a communicator without it is a hard error.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/sz3hip.h"
#include "sz3hip_internal.h"

namespace rc {
static void *h;
static ncclResult_t (*GetUniqueId)(ncclUniqueId *);
static ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
static ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
static ncclResult_t (*CommDestroy)(ncclComm_t);
static ncclResult_t (*CommCount)(const ncclComm_t, int *);
static ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
static ncclResult_t (*GroupStart)();
static ncclResult_t (*GroupEnd)();
static const char *(*GetErrorString)(ncclResult_t);
static std::once_flag once;
static bool ok;
static void load_once() {
    // by soname first: a process that already holds an RCCL (e.g. the one PyTorch ships) gets that same copy
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr};
    for (int i = 0; names[i] && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
#define SYM(f) *(void **)(&f) = dlsym(h, "nccl" #f)
    SYM(GetUniqueId);
    SYM(CommInitRank);
    SYM(CommInitAll);
    SYM(CommDestroy);
    SYM(CommCount);
    SYM(AllReduce);
    SYM(GroupStart);
    SYM(GroupEnd);
    SYM(GetErrorString);
#undef SYM
    ok = GetUniqueId && CommInitRank && CommInitAll && CommDestroy && CommCount && AllReduce && GroupStart && GroupEnd && GetErrorString;
}
static int load() {
    std::call_once(once, load_once);
    return ok ? 0 : szi_fail(SZ3HIP_EUNSUPPORTED, "librccl.so.1 not found or incomplete: no multi-GPU exchange without RCCL");
}
}  // namespace rc

#define RCCLCHK(call)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess) return szi_fail(SZ3HIP_EHIP, "%s failed: %s", #call, rc::GetErrorString(r_)); \
    } while (0)
#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return szi_fail(SZ3HIP_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

struct sz3hip_comm {
    int nranks;                     // size of the communicator
    int rank0;                      // rank of local member 0 (rank mode: this process's rank; local mode: 0)
    std::vector<int> devices;       // device of every local member
    std::vector<ncclComm_t> comms;  // one per local member
    std::vector<double *> d_mm;     // per member: [min, -max] staging for the range exchange (device) ...
    std::vector<double *> h_mm;     // ... and its pinned host mirror
};

extern "C" int sz3hip_comm_unique_id(unsigned char *id128) {
    if (rc::load()) return sz3hip_last_error_code();
    static_assert(sizeof(ncclUniqueId) == SZ3HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    RCCLCHK(rc::GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

static int comm_alloc_staging(sz3hip_comm *c) {
    c->d_mm.assign(c->devices.size(), nullptr);
    c->h_mm.assign(c->devices.size(), nullptr);
    for (size_t i = 0; i < c->devices.size(); i++) {
        HIPCHK(hipSetDevice(c->devices[i]));
        HIPCHK(hipMalloc((void **)&c->d_mm[i], 16));
        HIPCHK(hipHostMalloc((void **)&c->h_mm[i], 16));
    }
    return 0;
}

extern "C" sz3hip_comm *sz3hip_comm_create_local(int ndev, const int *devices) {
    if (rc::load()) return nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1) {
        szi_fail(SZ3HIP_EHIP, "no usable HIP device; this library has no CPU path");
        return nullptr;
    }
    if (ndev <= 0) ndev = have;
    int prev = 0;
    (void)hipGetDevice(&prev);
    sz3hip_comm *c = new sz3hip_comm();
    c->nranks = ndev;
    c->rank0 = 0;
    for (int i = 0; i < ndev; i++) c->devices.push_back(devices ? devices[i] : i);
    for (int d : c->devices)
        if (d < 0 || d >= have) {
            szi_fail(SZ3HIP_EINVAL, "device %d requested but %d visible", d, have);
            delete c;
            return nullptr;
        }
    c->comms.assign(ndev, nullptr);
    ncclResult_t r = rc::CommInitAll(c->comms.data(), ndev, c->devices.data());
    if (r != ncclSuccess) {
        szi_fail(SZ3HIP_EHIP, "ncclCommInitAll(%d) failed: %s", ndev, rc::GetErrorString(r));
        delete c;
        (void)hipSetDevice(prev);
        return nullptr;
    }
    if (comm_alloc_staging(c)) {
        sz3hip_comm_destroy(c);
        (void)hipSetDevice(prev);
        return nullptr;
    }
    (void)hipSetDevice(prev);
    return c;
}

extern "C" sz3hip_comm *sz3hip_comm_create_rank(int nranks, int rank, int device, const unsigned char *id128) {
    if (rc::load()) return nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) {
        szi_fail(SZ3HIP_EINVAL, "rank %d of %d", rank, nranks);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        szi_fail(SZ3HIP_EHIP, "hipSetDevice(%d) failed — no usable HIP device; this library has no CPU path", device);
        return nullptr;
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    sz3hip_comm *c = new sz3hip_comm();
    c->nranks = nranks;
    c->rank0 = rank;
    c->devices.push_back(device);
    c->comms.assign(1, nullptr);
    ncclResult_t r = rc::CommInitRank(&c->comms[0], nranks, id, rank);
    if (r != ncclSuccess) {
        szi_fail(SZ3HIP_EHIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, rc::GetErrorString(r));
        delete c;
        return nullptr;
    }
    if (comm_alloc_staging(c)) {
        sz3hip_comm_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" void sz3hip_comm_destroy(sz3hip_comm *c) {
    if (!c) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < c->comms.size(); i++) {
        (void)hipSetDevice(c->devices[i]);
        if (i < c->d_mm.size() && c->d_mm[i]) (void)hipFree(c->d_mm[i]);
        if (i < c->h_mm.size() && c->h_mm[i]) (void)hipHostFree(c->h_mm[i]);
        if (c->comms[i]) (void)rc::CommDestroy(c->comms[i]);
    }
    (void)hipSetDevice(prev);
    delete c;
}

extern "C" int sz3hip_comm_size(const sz3hip_comm *c) {
    // what RCCL itself reports (bench.py asserts it against --gpus)
    int n = 0;
    if (!c || c->comms.empty() || rc::CommCount(c->comms[0], &n) != ncclSuccess) return c ? c->nranks : 0;
    return n;
}
extern "C" int sz3hip_comm_rank(const sz3hip_comm *c) { return c ? c->rank0 : -1; }
extern "C" int sz3hip_comm_local_size(const sz3hip_comm *c) { return c ? (int)c->comms.size() : 0; }
extern "C" int sz3hip_comm_device(const sz3hip_comm *c, int member) {
    return c && member >= 0 && member < (int)c->devices.size() ? c->devices[member] : -1;
}

// every local member contributes bufs[i] (device memory of its GPU) on streams[i]; in place
static int allreduce_members(sz3hip_comm *c, void *const *bufs, size_t count, ncclDataType_t dt, ncclRedOp_t op, void *const *streams) {
    const size_t m = c->comms.size();
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (m > 1) RCCLCHK(rc::GroupStart());
    for (size_t i = 0; i < m; i++) {
        HIPCHK(hipSetDevice(c->devices[i]));
        ncclResult_t r = rc::AllReduce(bufs[i], bufs[i], count, dt, op, c->comms[i], (hipStream_t)streams[i]);
        if (r != ncclSuccess) {
            if (m > 1) (void)rc::GroupEnd();
            (void)hipSetDevice(prev);
            return szi_fail(SZ3HIP_EHIP, "ncclAllReduce failed: %s", rc::GetErrorString(r));
        }
    }
    if (m > 1) RCCLCHK(rc::GroupEnd());
    (void)hipSetDevice(prev);
    return 0;
}

extern "C" int sz3hip_comm_allreduce_u64(sz3hip_comm *c, void *const *d_bufs, size_t count, void *const *streams) {
    if (!c) return szi_fail(SZ3HIP_EINVAL, "no communicator");
    return allreduce_members(c, d_bufs, count, ncclUint64, ncclSum, streams);
}

extern "C" int sz3hip_comm_allreduce_histogram(sz3hip_comm *c, sz3hip_ctx *const *ctxs, void *const *streams) {
    if (!c) return szi_fail(SZ3HIP_EINVAL, "no communicator");
    std::vector<void *> bufs(c->comms.size());
    for (size_t i = 0; i < bufs.size(); i++) {
        if (!ctxs[i]) return szi_fail(SZ3HIP_EINVAL, "member %zu has no context", i);
        bufs[i] = szi_histogram_for_exchange(ctxs[i]);
    }
    return allreduce_members(c, bufs.data(), sz3hip_histogram_len(ctxs[0]), ncclUint64, ncclSum, streams);
}

// global (min, max) from every member's local pair: one MIN all-reduce of [min, -max] (synchronises the streams)
extern "C" int sz3hip_comm_allreduce_minmax(sz3hip_comm *c, double *mins, double *maxs, void *const *streams) {
    if (!c) return szi_fail(SZ3HIP_EINVAL, "no communicator");
    const size_t m = c->comms.size();
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < m; i++) {
        HIPCHK(hipSetDevice(c->devices[i]));
        c->h_mm[i][0] = mins[i];
        c->h_mm[i][1] = -maxs[i];
        HIPCHK(hipMemcpyAsync(c->d_mm[i], c->h_mm[i], 16, hipMemcpyHostToDevice, (hipStream_t)streams[i]));
    }
    std::vector<void *> bufs(m);
    for (size_t i = 0; i < m; i++) bufs[i] = c->d_mm[i];
    int rcx = allreduce_members(c, bufs.data(), 2, ncclFloat64, ncclMin, streams);
    if (rcx) return rcx;
    for (size_t i = 0; i < m; i++) {
        HIPCHK(hipSetDevice(c->devices[i]));
        HIPCHK(hipMemcpyAsync(c->h_mm[i], c->d_mm[i], 16, hipMemcpyDeviceToHost, (hipStream_t)streams[i]));
    }
    for (size_t i = 0; i < m; i++) {
        HIPCHK(hipSetDevice(c->devices[i]));
        HIPCHK(hipStreamSynchronize((hipStream_t)streams[i]));
        mins[i] = c->h_mm[i][0];
        maxs[i] = -c->h_mm[i][1];
    }
    (void)hipSetDevice(prev);
    return 0;
}
