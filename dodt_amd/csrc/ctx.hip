// Context, device memory and the HIP-event stopwatch of libdodt_hip.so.
#include "common.h"

namespace dodt {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int Scratch::reserve(size_t need) {
    if (need <= bytes) return DODT_OK;
    if (ptr) {
        DODT_HIP_CHECK(hipFree(ptr));
        ptr = nullptr;
        bytes = 0;
    }
    size_t want = align_up(need + need / 4, 256);
    DODT_HIP_CHECK(hipMalloc(&ptr, want));
    bytes = want;
    return DODT_OK;
}

void Scratch::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

}  // namespace dodt

extern "C" {

int dodt_version(void) { return 3; }

const char* dodt_last_error(void) { return dodt::g_last_error.c_str(); }

static int ctx_create(int device_id, hipStream_t stream, bool external, bool high_priority,
                      dodt_ctx** out) {
    DODT_REQUIRE(out != nullptr, "dodt_ctx_create: out is NULL");
    int ndev = 0;
    DODT_HIP_CHECK(hipGetDeviceCount(&ndev));
    DODT_REQUIRE(device_id >= 0 && device_id < ndev,
                 "dodt_ctx_create: device %d out of range (%d visible)", device_id, ndev);
    DODT_HIP_CHECK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    DODT_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    dodt_ctx* c = new dodt_ctx();
    c->device = device_id;
    c->num_cus = prop.multiProcessorCount;
    if (external) {
        c->stream = stream;
        c->owns_stream = false;
    } else {
        hipError_t e;
        if (high_priority) {
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest);
        } else {
            e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        }
        if (e != hipSuccess) {
            delete c;
            dodt::set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            return DODT_ERR_HIP;
        }
        c->owns_stream = true;
    }
    (void)hipEventCreate(&c->ev_start);
    (void)hipEventCreate(&c->ev_stop);
    (void)hipHostMalloc(reinterpret_cast<void**>(&c->pinned), kFetchSlots * 16 * sizeof(int32_t), 0);
    for (int i = 0; i < kFetchSlots; ++i) (void)hipEventCreateWithFlags(&c->fetch_ev[i], hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming);
    *out = c;
    return DODT_OK;
}

int dodt_ctx_create(int device_id, dodt_ctx** out) {
    return ctx_create(device_id, nullptr, false, false, out);
}

int dodt_ctx_create_high_priority(int device_id, dodt_ctx** out) {
    return ctx_create(device_id, nullptr, false, true, out);
}

int dodt_ctx_create_on_stream(int device_id, void* hip_stream, dodt_ctx** out) {
    return ctx_create(device_id, (hipStream_t)hip_stream, true, false, out);
}

int dodt_ctx_destroy(dodt_ctx* ctx) {
    if (!ctx) return DODT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->vox_ws.release();
    ctx->anchor_ws.release();
    ctx->nms_ws.release();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (int i = 0; i < kFetchSlots; ++i)
        if (ctx->fetch_ev[i]) (void)hipEventDestroy(ctx->fetch_ev[i]);
    if (ctx->join_ev) (void)hipEventDestroy(ctx->join_ev);
    for (int i = 0; i < kMarkSlots; ++i)
        if (ctx->mark_ev[i]) (void)hipEventDestroy(ctx->mark_ev[i]);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return DODT_OK;
}

int dodt_ctx_sync(dodt_ctx* ctx) {
    DODT_REQUIRE(ctx, "dodt_ctx_sync: ctx is NULL");
    DODT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return DODT_OK;
}

int dodt_ctx_wait_for(dodt_ctx* ctx, dodt_ctx* other) {
    DODT_REQUIRE(ctx && other, "dodt_ctx_wait_for: NULL argument");
    if (ctx->stream == other->stream) return DODT_OK;
    DODT_HIP_CHECK(hipEventRecord(other->join_ev, other->stream));
    DODT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, other->join_ev, 0));
    return DODT_OK;
}

int dodt_mark(dodt_ctx* ctx, int slot) {
    DODT_REQUIRE(ctx && slot >= 0 && slot < kMarkSlots, "dodt_mark: bad argument");
    if (!ctx->mark_ev[slot]) DODT_HIP_CHECK(hipEventCreate(&ctx->mark_ev[slot]));
    DODT_HIP_CHECK(hipEventRecord(ctx->mark_ev[slot], ctx->stream));
    return DODT_OK;
}

int dodt_ctx_wait_mark(dodt_ctx* ctx, dodt_ctx* other, int slot) {
    DODT_REQUIRE(ctx && other && slot >= 0 && slot < kMarkSlots && other->mark_ev[slot],
                 "dodt_ctx_wait_mark: bad argument or mark never recorded");
    DODT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, other->mark_ev[slot], 0));
    return DODT_OK;
}

int dodt_mark_elapsed(dodt_ctx* from, int from_slot, dodt_ctx* to, int to_slot, float* ms) {
    DODT_REQUIRE(from && to && ms && from_slot >= 0 && from_slot < kMarkSlots && to_slot >= 0 &&
                     to_slot < kMarkSlots && from->mark_ev[from_slot] && to->mark_ev[to_slot],
                 "dodt_mark_elapsed: bad argument or mark never recorded");
    DODT_HIP_CHECK(hipEventSynchronize(to->mark_ev[to_slot]));
    DODT_HIP_CHECK(hipEventElapsedTime(ms, from->mark_ev[from_slot], to->mark_ev[to_slot]));
    return DODT_OK;
}

int dodt_malloc(dodt_ctx* ctx, size_t bytes, void** d_out) {
    DODT_REQUIRE(ctx && d_out, "dodt_malloc: NULL argument");
    DODT_HIP_CHECK(hipSetDevice(ctx->device));
    DODT_HIP_CHECK(hipMalloc(d_out, bytes ? bytes : 1));
    return DODT_OK;
}

int dodt_free(dodt_ctx* ctx, void* d_ptr) {
    DODT_REQUIRE(ctx, "dodt_free: ctx is NULL");
    if (d_ptr) DODT_HIP_CHECK(hipFree(d_ptr));
    return DODT_OK;
}

int dodt_memcpy_h2d(dodt_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
    DODT_REQUIRE(ctx && (bytes == 0 || (d_dst && src)), "dodt_memcpy_h2d: NULL argument");
    if (bytes == 0) return DODT_OK;
    DODT_HIP_CHECK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    // the source is pageable host memory owned by the caller: finish before return
    DODT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return DODT_OK;
}

int dodt_memcpy_d2h(dodt_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
    DODT_REQUIRE(ctx && (bytes == 0 || (dst && d_src)), "dodt_memcpy_d2h: NULL argument");
    if (bytes == 0) return DODT_OK;
    DODT_HIP_CHECK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    DODT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return DODT_OK;
}

int dodt_pinned_alloc(dodt_ctx* ctx, size_t bytes, void** out) {
    DODT_REQUIRE(ctx && out, "dodt_pinned_alloc: NULL argument");
    DODT_HIP_CHECK(hipSetDevice(ctx->device));
    DODT_HIP_CHECK(hipHostMalloc(out, bytes ? bytes : 1, 0));
    return DODT_OK;
}

int dodt_pinned_free(dodt_ctx* ctx, void* ptr) {
    DODT_REQUIRE(ctx, "dodt_pinned_free: ctx is NULL");
    if (ptr) DODT_HIP_CHECK(hipHostFree(ptr));
    return DODT_OK;
}

int dodt_memcpy_h2d_async(dodt_ctx* ctx, void* d_dst, const void* pinned_src, size_t bytes) {
    DODT_REQUIRE(ctx && (bytes == 0 || (d_dst && pinned_src)), "dodt_memcpy_h2d_async: NULL argument");
    if (bytes == 0) return DODT_OK;
    DODT_HIP_CHECK(hipMemcpyAsync(d_dst, pinned_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return DODT_OK;
}

int dodt_memset(dodt_ctx* ctx, void* d_dst, int value, size_t bytes) {
    DODT_REQUIRE(ctx && (bytes == 0 || d_dst), "dodt_memset: NULL argument");
    if (bytes == 0) return DODT_OK;
    DODT_HIP_CHECK(hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return DODT_OK;
}

int dodt_fetch_i32_begin(dodt_ctx* ctx, const int32_t* d_src, int n, int slot) {
    DODT_REQUIRE(ctx && d_src && ctx->pinned, "dodt_fetch_i32_begin: NULL argument");
    DODT_REQUIRE(n >= 1 && n <= 16 && slot >= 0 && slot < kFetchSlots, "dodt_fetch_i32_begin: bad n/slot");
    DODT_HIP_CHECK(hipMemcpyAsync(ctx->pinned + slot * 16, d_src, n * sizeof(int32_t),
                                  hipMemcpyDeviceToHost, ctx->stream));
    DODT_HIP_CHECK(hipEventRecord(ctx->fetch_ev[slot], ctx->stream));
    return DODT_OK;
}

int dodt_fetch_i32_end(dodt_ctx* ctx, int slot, int32_t* dst, int n) {
    DODT_REQUIRE(ctx && dst && ctx->pinned, "dodt_fetch_i32_end: NULL argument");
    DODT_REQUIRE(n >= 1 && n <= 16 && slot >= 0 && slot < kFetchSlots, "dodt_fetch_i32_end: bad n/slot");
    DODT_HIP_CHECK(hipEventSynchronize(ctx->fetch_ev[slot]));
    for (int i = 0; i < n; ++i) dst[i] = ctx->pinned[slot * 16 + i];
    return DODT_OK;
}

int dodt_timer_start(dodt_ctx* ctx) {
    DODT_REQUIRE(ctx, "dodt_timer_start: ctx is NULL");
    DODT_HIP_CHECK(hipEventRecord(ctx->ev_start, ctx->stream));
    return DODT_OK;
}

int dodt_timer_stop(dodt_ctx* ctx, float* ms_out) {
    DODT_REQUIRE(ctx && ms_out, "dodt_timer_stop: NULL argument");
    DODT_HIP_CHECK(hipEventRecord(ctx->ev_stop, ctx->stream));
    DODT_HIP_CHECK(hipEventSynchronize(ctx->ev_stop));
    DODT_HIP_CHECK(hipEventElapsedTime(ms_out, ctx->ev_start, ctx->ev_stop));
    return DODT_OK;
}

}  // extern "C"
