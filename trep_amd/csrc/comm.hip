// comm.hip -- the one collective of the path, directly on RCCL (no torch, no MPI).
//
// trajectories / seeds are independent, so every rank owns a contiguous batch slice on its
// own GPU and nothing is exchanged inside a rollout.  What the host-side line search needs on every rank afterwards
// -- terminal states or per-seed costs of ALL ranks -- is one all-gather over the node's xGMI links, plus a scalar
// all-reduce for barriers and max-over-ranks timings.  One process per GPU; the 128-byte RCCL unique id is made
// by rank 0 (tg_comm_unique_id) and handed to the other ranks by the launcher-side code in any way it likes
// (trep_amd/rccl.py uses a file next to the rendezvous port; no GPU call is involved in that exchange).
//
// librccl is opened lazily with dlopen the first time a tg_comm_* entry point runs: single-GPU users of
// libtrepamd.so never load it, and the library has no link-time dependency on RCCL.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/trep_amd.h"

namespace tg_detail {
int fail(int code, const std::string &msg);
}

namespace {

// the slice of the RCCL ABI this file uses (rccl.h, NCCL 2.x compatible)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0, ncclFloat64 = 8 };
enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
    int (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl R;
    if (R.handle || !R.error.empty()) return R;
    const char *names[] = {std::getenv("TREPAMD_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        R.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (R.handle) break;
    }
    if (!R.handle) { R.error = std::string("cannot open librccl: ") + (dlerror() ? dlerror() : "not found"); return R; }
    auto sym = [&](const char *n) -> void * {
        void *p = dlsym(R.handle, n);
        if (!p && R.error.empty()) R.error = std::string("librccl lacks ") + n;
        return p;
    };
    R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
    R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
    R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
    R.CommCount = reinterpret_cast<decltype(R.CommCount)>(sym("ncclCommCount"));
    R.CommUserRank = reinterpret_cast<decltype(R.CommUserRank)>(sym("ncclCommUserRank"));
    R.AllGather = reinterpret_cast<decltype(R.AllGather)>(sym("ncclAllGather"));
    R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(sym("ncclAllReduce"));
    R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
    return R;
}

int rccl_fail(const char *what, int rc) {
    Rccl &R = rccl();
    return tg_detail::fail(TG_ERR_HIP, std::string(what) + ": " + (R.GetErrorString ? R.GetErrorString(rc) : "RCCL error"));
}

#define HIP_TRYC(expr)                                                                                          \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return tg_detail::fail(TG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

}  // namespace

struct tg_comm {
    int device = 0, world = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double *scratch = nullptr;    // device staging of the host-scalar reductions
    size_t scratch_n = 0;
    hipEvent_t handoff = nullptr; // producer stream -> communicator stream (tg_comm_wait_stream)
    hipEvent_t done = nullptr;    // communicator stream -> consumer stream (tg_comm_stream_wait_comm)
};

extern "C" {

int tg_comm_unique_id(uint8_t id_out[TG_COMM_ID_BYTES]) {
    Rccl &R = rccl();
    if (!R.error.empty()) return tg_detail::fail(TG_ERR_UNSUPPORTED, R.error);
    ncclUniqueId id;
    static_assert(sizeof(id) == TG_COMM_ID_BYTES, "RCCL unique id size");
    const int rc = R.GetUniqueId(&id);
    if (rc != ncclSuccess) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id_out, id.internal, TG_COMM_ID_BYTES);
    return TG_SUCCESS;
}

tg_comm *tg_comm_create(int32_t device, int32_t world, int32_t rank, const uint8_t id_in[TG_COMM_ID_BYTES]) {
    Rccl &R = rccl();
    if (!R.error.empty()) { tg_detail::fail(TG_ERR_UNSUPPORTED, R.error); return nullptr; }
    if (world < 1 || rank < 0 || rank >= world || !id_in) { tg_detail::fail(TG_ERR_INVALID, "bad communicator arguments"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { tg_detail::fail(TG_ERR_HIP, "hipSetDevice failed"); return nullptr; }
    tg_comm *c = new tg_comm();
    c->device = device; c->world = world; c->rank = rank;
    ncclUniqueId id;
    std::memcpy(id.internal, id_in, TG_COMM_ID_BYTES);
    const int rc = R.CommInitRank(&c->comm, world, id, rank);
    if (rc != ncclSuccess) { rccl_fail("ncclCommInitRank", rc); delete c; return nullptr; }
    // a blocking stream: implicitly ordered with the device's NULL stream only; work of any other stream (a tg_batch's) is
    // handed over with an event (tg_comm_wait_stream)
    if (hipStreamCreate(&c->stream) != hipSuccess || hipEventCreateWithFlags(&c->handoff, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        tg_detail::fail(TG_ERR_HIP, "stream / event creation failed");
        if (c->handoff) hipEventDestroy(c->handoff);
        if (c->stream) hipStreamDestroy(c->stream);
        R.CommDestroy(c->comm); delete c; return nullptr;
    }
    return c;
}

void tg_comm_destroy(tg_comm *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm) rccl().CommDestroy(c->comm);
    if (c->handoff) hipEventDestroy(c->handoff);
    if (c->done) hipEventDestroy(c->done);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->scratch) hipFree(c->scratch);
    delete c;
}

int tg_comm_info(const tg_comm *c, int32_t out[3]) {
    if (!c) return tg_detail::fail(TG_ERR_INVALID, "null communicator");
    // what RCCL itself says about the communicator (not what the caller passed in)
    int n = 0, r = -1;
    int rc = rccl().CommCount(c->comm, &n);
    if (rc != ncclSuccess) return rccl_fail("ncclCommCount", rc);
    rc = rccl().CommUserRank(c->comm, &r);
    if (rc != ncclSuccess) return rccl_fail("ncclCommUserRank", rc);
    out[0] = n; out[1] = r; out[2] = c->device;
    return TG_SUCCESS;
}

int tg_comm_all_gather(tg_comm *c, const void *send_dev, void *recv_dev, uint64_t bytes_per_rank) {
    if (!c || !send_dev || !recv_dev) return tg_detail::fail(TG_ERR_INVALID, "null argument");
    HIP_TRYC(hipSetDevice(c->device));
    if (bytes_per_rank == 0) return TG_SUCCESS;
    int rc;
    if (bytes_per_rank % 8 == 0) rc = rccl().AllGather(send_dev, recv_dev, (size_t)(bytes_per_rank / 8), ncclFloat64, c->comm, c->stream);
    else rc = rccl().AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, ncclInt8, c->comm, c->stream);
    if (rc != ncclSuccess) return rccl_fail("ncclAllGather", rc);
    return TG_SUCCESS;
}

int tg_comm_wait_stream(tg_comm *c, void *producer) {
    if (!c) return tg_detail::fail(TG_ERR_INVALID, "null communicator");
    HIP_TRYC(hipSetDevice(c->device));
    HIP_TRYC(hipEventRecord(c->handoff, (hipStream_t)producer));
    HIP_TRYC(hipStreamWaitEvent(c->stream, c->handoff, 0));
    return TG_SUCCESS;
}

int tg_comm_all_gather_after(tg_comm *c, void *producer, const void *send_dev, void *recv_dev, uint64_t bytes_per_rank) {
    const int rc = tg_comm_wait_stream(c, producer);
    return rc != TG_SUCCESS ? rc : tg_comm_all_gather(c, send_dev, recv_dev, bytes_per_rank);
}

int tg_comm_stream_wait_comm(tg_comm *c, void *consumer) {
    if (!c) return tg_detail::fail(TG_ERR_INVALID, "null communicator");
    HIP_TRYC(hipSetDevice(c->device));
    HIP_TRYC(hipEventRecord(c->done, c->stream));
    HIP_TRYC(hipStreamWaitEvent((hipStream_t)consumer, c->done, 0));
    return TG_SUCCESS;
}

int tg_comm_synchronize(tg_comm *c) {
    if (!c) return tg_detail::fail(TG_ERR_INVALID, "null communicator");
    HIP_TRYC(hipSetDevice(c->device));
    HIP_TRYC(hipStreamSynchronize(c->stream));
    return TG_SUCCESS;
}

int tg_comm_all_reduce_host(tg_comm *c, double *values, int32_t n, int32_t op) {
    if (!c || !values || n <= 0) return tg_detail::fail(TG_ERR_INVALID, "bad arguments");
    if (op != TG_REDUCE_SUM && op != TG_REDUCE_MAX && op != TG_REDUCE_MIN) return tg_detail::fail(TG_ERR_INVALID, "unknown reduction");
    HIP_TRYC(hipSetDevice(c->device));
    if ((size_t)n > c->scratch_n) {
        if (c->scratch) HIP_TRYC(hipFree(c->scratch));
        c->scratch = nullptr; c->scratch_n = 0;
        HIP_TRYC(hipMalloc(&c->scratch, (size_t)n * sizeof(double)));
        c->scratch_n = (size_t)n;
    }
    HIP_TRYC(hipMemcpyAsync(c->scratch, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int rop = op == TG_REDUCE_SUM ? ncclSum : (op == TG_REDUCE_MAX ? ncclMax : ncclMin);
    const int rc = rccl().AllReduce(c->scratch, c->scratch, (size_t)n, ncclFloat64, rop, c->comm, c->stream);
    if (rc != ncclSuccess) return rccl_fail("ncclAllReduce", rc);
    HIP_TRYC(hipMemcpyAsync(values, c->scratch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRYC(hipStreamSynchronize(c->stream));
    return TG_SUCCESS;
}

int tg_comm_barrier(tg_comm *c) {
    double one = 1.0;
    return tg_comm_all_reduce_host(c, &one, 1, TG_REDUCE_SUM);
}

}  // extern "C"
