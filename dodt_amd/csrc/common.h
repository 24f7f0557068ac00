// Internal declarations shared by the HIP translation units of libdodt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/dodt_hip.h"

namespace dodt {

void set_error(const char* fmt, ...);

#define DODT_HIP_CHECK(expr)                                                   \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            dodt::set_error("%s failed: %s (%s:%d)", #expr,                    \
                            hipGetErrorString(e_), __FILE__, __LINE__);        \
            return DODT_ERR_HIP;                                               \
        }                                                                      \
    } while (0)

#define DODT_REQUIRE(cond, ...)                                                \
    do {                                                                       \
        if (!(cond)) {                                                         \
            dodt::set_error(__VA_ARGS__);                                      \
            return DODT_ERR_INVALID;                                           \
        }                                                                      \
    } while (0)

#define DODT_LAUNCH_CHECK() DODT_HIP_CHECK(hipGetLastError())

// Scratch buffer that only ever grows; never freed inside a launch function.
struct Scratch {
    void* ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t need);
    void release();
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// bf16 <-> float on the host: round to nearest even, like the device's v_cvt_pk_bf16_f32
static inline uint16_t float_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_float(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace dodt

constexpr int kMarkSlots = 256;  // dodt_mark slots per context
constexpr int kFetchSlots = 32;   // dodt_fetch_i32_* slots per context

struct dodt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    dodt::Scratch vox_ws;    // voxeliser: touched list + counters
    dodt::Scratch anchor_ws; // anchor filter: mask + block counts
    dodt::Scratch nms_ws;    // NMS: keys, sorted boxes, suppression mask
    int num_cus = 256;
    int32_t* pinned = nullptr;       // kFetchSlots x 16 int32, hipHostMalloc
    hipEvent_t fetch_ev[kFetchSlots] = {};
    hipEvent_t mark_ev[kMarkSlots] = {};     // timing marks for tools/ (created on first use)
    hipEvent_t join_ev = nullptr;    // recorded on this stream for dodt_ctx_wait_for
};
