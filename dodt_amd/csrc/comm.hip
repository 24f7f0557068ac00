// Multi-GPU exchange step of the path: one RCCL all-gather of the fixed detection
// record of every rank, on a SIDE stream, no PyTorch.
//
// The reference has no collective at all: avod/experiments/run_tracking_inference.py:109-128 pins ONE
// device through CUDA_VISIBLE_DEVICES and walks the sequences one after the other.  Here frame pairs
// shard over ranks (pair i -> rank i mod N) and the sequential temporal module
// (dt_evaluator_utils.py:189-362) consumes the gathered records, so the one exchange is
//     ncclAllGather( float32 [pairs][frames][100][17] ) + ncclAllGather( int32 [pairs][frames] )
// in one group (13.6 KB per rank and pair: latency bound).
//
// librccl.so is opened lazily with dlopen when the first communicator is made: a single-GPU process
// never maps it, and libdodt_hip.so has no link-time dependency on it.
//
// Stream pattern: the gather waits (event) for what the producer context has enqueued so far, runs
// on the communicator's own stream and signals one of four events (a ring indexed by the caller's
// slot); a consumer joins a slot's event only when it is about to overwrite that slot's send buffer
// (two steps later in the pipeline), so a slow rank delays the gather, not the next step's kernels.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

int load_rccl() {
    if (g_rccl.handle) return DODT_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) {
        dodt::set_error("dodt_comm: librccl.so not found (%s)", dlerror());
        return DODT_ERR_UNSUPPORTED;
    }
    RcclApi a;
    a.handle = h;
#define DODT_SYM(field, name)                                                  \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));             \
    if (!a.field) {                                                            \
        dodt::set_error("dodt_comm: librccl.so lacks %s", name);               \
        dlclose(h);                                                            \
        return DODT_ERR_UNSUPPORTED;                                           \
    }
    DODT_SYM(GetUniqueId, "ncclGetUniqueId")
    DODT_SYM(CommInitRank, "ncclCommInitRank")
    DODT_SYM(CommDestroy, "ncclCommDestroy")
    DODT_SYM(AllGather, "ncclAllGather")
    DODT_SYM(AllReduce, "ncclAllReduce")
    DODT_SYM(GroupStart, "ncclGroupStart")
    DODT_SYM(GroupEnd, "ncclGroupEnd")
    DODT_SYM(GetErrorString, "ncclGetErrorString")
#undef DODT_SYM
    g_rccl = a;
    return DODT_OK;
}

#define DODT_NCCL_CHECK(expr)                                                          \
    do {                                                                               \
        ncclResult_t r_ = (expr);                                                      \
        if (r_ != ncclSuccess) {                                                       \
            dodt::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), \
                            __FILE__, __LINE__);                                       \
            return DODT_ERR_HIP;                                                       \
        }                                                                              \
    } while (0)

constexpr int kSlots = 4;

// One wave that idles until `ticks` of the 100 MHz wall clock have passed (dodt_comm_set_late_peer).  The loop
// ends by the clock, and by an iteration cap should the clock ever stand still: an s_sleep of 64 x 64 cycles
// is >= 1.7 us at the highest shader clock, so `ticks / 64 + 1024` iterations always outlast `ticks`.
__global__ void __launch_bounds__(64) late_peer_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    const unsigned long long cap = ticks / 64 + 1024;
    for (unsigned long long i = 0; i < cap && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(64);
}

}  // namespace

struct dodt_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;          // the side stream every collective runs on
    bool owns_stream = true;               // false once attached to a context's stream (dodt_comm_attach)
    hipEvent_t produced = nullptr;         // recorded on the producer's stream
    hipEvent_t done[kSlots] = {};          // gather of slot s has left the side stream
    bool used[kSlots] = {};
    double* d_scalar = nullptr;            // 2 doubles for the host-value reductions
    double late_peer_us = 0.0;             // measurement aid: idle time in front of every gather
};

extern "C" {

int dodt_comm_unique_id(uint8_t* id_out) {
    DODT_REQUIRE(id_out, "dodt_comm_unique_id: NULL argument");
    if (int rc = load_rccl()) return rc;
    static_assert(DODT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId id;
    DODT_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return DODT_OK;
}

int dodt_comm_create(dodt_ctx* ctx, int rank, int world, const uint8_t* id, dodt_comm** out) {
    DODT_REQUIRE(ctx && id && out, "dodt_comm_create: NULL argument");
    DODT_REQUIRE(world >= 1 && rank >= 0 && rank < world, "dodt_comm_create: rank %d of %d", rank, world);
    if (int rc = load_rccl()) return rc;
    DODT_HIP_CHECK(hipSetDevice(ctx->device));
    dodt_comm* c = new dodt_comm();
    c->device = ctx->device;
    c->rank = rank;
    c->world = world;
    ncclUniqueId uid;
    memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        dodt::set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
        delete c;
        return DODT_ERR_HIP;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->produced, hipEventDisableTiming);
    for (int i = 0; i < kSlots && e == hipSuccess; ++i)
        e = hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_scalar), 2 * sizeof(double));
    if (e != hipSuccess) {
        dodt::set_error("dodt_comm_create: %s", hipGetErrorString(e));
        dodt_comm_destroy(c);
        return DODT_ERR_HIP;
    }
    *out = c;
    return DODT_OK;
}

int dodt_comm_destroy(dodt_comm* c) {
    if (!c) return DODT_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_scalar) (void)hipFree(c->d_scalar);
    if (c->produced) (void)hipEventDestroy(c->produced);
    for (int i = 0; i < kSlots; ++i)
        if (c->done[i]) (void)hipEventDestroy(c->done[i]);
    if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return DODT_OK;
}

int dodt_comm_attach(dodt_comm* c, dodt_ctx* ctx) {
    DODT_REQUIRE(c && ctx, "dodt_comm_attach: NULL argument");
    DODT_REQUIRE(ctx->device == c->device, "dodt_comm_attach: context of another device");
    DODT_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    c->stream = ctx->stream;
    c->owns_stream = false;
    return DODT_OK;
}

int dodt_comm_rank(const dodt_comm* c, int* rank, int* world) {
    DODT_REQUIRE(c && rank && world, "dodt_comm_rank: NULL argument");
    *rank = c->rank;
    *world = c->world;
    return DODT_OK;
}

int dodt_comm_set_late_peer(dodt_comm* c, double microseconds) {
    DODT_REQUIRE(c, "dodt_comm_set_late_peer: comm is NULL");
    DODT_REQUIRE(microseconds >= 0.0 && microseconds <= 1e5, "dodt_comm_set_late_peer: %g us not in 0..100000", microseconds);
    c->late_peer_us = microseconds;
    return DODT_OK;
}

int dodt_all_gather_records(dodt_comm* c, dodt_ctx* producer, int slot, const float* d_records,
                            const int32_t* d_counts, int pairs, int frames, int max_det, int cols,
                            float* d_all_records, int32_t* d_all_counts) {
    DODT_REQUIRE(c && producer && d_records && d_counts && d_all_records && d_all_counts,
                 "dodt_all_gather_records: NULL argument");
    DODT_REQUIRE(slot >= 0 && slot < kSlots, "dodt_all_gather_records: slot %d not in 0..%d", slot, kSlots - 1);
    DODT_REQUIRE(pairs >= 1 && frames >= 1 && max_det >= 1 && cols >= 1, "dodt_all_gather_records: bad shape");
    DODT_REQUIRE(producer->device == c->device, "dodt_all_gather_records: context of another device");
    DODT_HIP_CHECK(hipEventRecord(c->produced, producer->stream));
    DODT_HIP_CHECK(hipStreamWaitEvent(c->stream, c->produced, 0));
    const size_t n_rec = (size_t)pairs * frames * max_det * cols, n_cnt = (size_t)pairs * frames;
    if (c->late_peer_us > 0.0) {
        hipLaunchKernelGGL(late_peer_kernel, dim3(1), dim3(64), 0, c->stream,
                           (unsigned long long)(c->late_peer_us * 100.0));
        DODT_LAUNCH_CHECK();
    }
    DODT_NCCL_CHECK(g_rccl.GroupStart());
    ncclResult_t r1 = g_rccl.AllGather(d_records, d_all_records, n_rec, ncclFloat32, c->comm, c->stream);
    ncclResult_t r2 = g_rccl.AllGather(d_counts, d_all_counts, n_cnt, ncclInt32, c->comm, c->stream);
    DODT_NCCL_CHECK(g_rccl.GroupEnd());
    DODT_NCCL_CHECK(r1);
    DODT_NCCL_CHECK(r2);
    DODT_HIP_CHECK(hipEventRecord(c->done[slot], c->stream));
    c->used[slot] = true;
    return DODT_OK;
}

int dodt_comm_join(dodt_comm* c, int slot, dodt_ctx* consumer) {
    DODT_REQUIRE(c && consumer, "dodt_comm_join: NULL argument");
    DODT_REQUIRE(slot >= 0 && slot < kSlots, "dodt_comm_join: slot %d not in 0..%d", slot, kSlots - 1);
    if (c->used[slot]) DODT_HIP_CHECK(hipStreamWaitEvent(consumer->stream, c->done[slot], 0));
    return DODT_OK;
}

int dodt_comm_sync(dodt_comm* c) {
    DODT_REQUIRE(c, "dodt_comm_sync: comm is NULL");
    DODT_HIP_CHECK(hipStreamSynchronize(c->stream));
    return DODT_OK;
}

static int reduce_f64(dodt_comm* c, double* value, ncclRedOp_t op) {
    DODT_HIP_CHECK(hipMemcpyAsync(c->d_scalar, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    DODT_NCCL_CHECK(g_rccl.AllReduce(c->d_scalar, c->d_scalar + 1, 1, ncclFloat64, op, c->comm, c->stream));
    DODT_HIP_CHECK(hipMemcpyAsync(value, c->d_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    DODT_HIP_CHECK(hipStreamSynchronize(c->stream));
    return DODT_OK;
}

int dodt_comm_barrier(dodt_comm* c) {
    DODT_REQUIRE(c, "dodt_comm_barrier: comm is NULL");
    double one = 1.0;
    if (int rc = reduce_f64(c, &one, ncclSum)) return rc;
    DODT_REQUIRE((int)(one + 0.5) == c->world, "dodt_comm_barrier: %g of %d ranks answered", one, c->world);
    return DODT_OK;
}

int dodt_comm_max_f64(dodt_comm* c, double* value) {
    DODT_REQUIRE(c && value, "dodt_comm_max_f64: NULL argument");
    return reduce_f64(c, value, ncclMax);
}

}  // extern "C"
