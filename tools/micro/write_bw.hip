// HBM write bandwidth by store pattern (what bounds the write-heavy layers: conv1_1, upconv1 write 143 MB):
//   dense     : a wave instruction writes 1 KiB contiguous (16 B per lane)
//   halves    : 16 B per lane at a 32-byte pitch (every other 16 B), the other halves by a second instruction
//               -- the Winograd epilogue's CB8 stores (a lane owns 4 of a cell's 8 channels)
//   quarter   : 16 B per lane at a 64-byte pitch, four instructions fill a line
//   *_nt      : the same with nontemporal stores
//   copy      : read 16 B + write 16 B dense
//   hipcc --offload-arch=gfx950 -O3 -o write_bw.bin write_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PITCH16, bool NT>   // PITCH16: lane pitch in 16-byte units (1, 2, 4)
__global__ __launch_bounds__(256) void writer(f32x4* out, size_t n16, float v) {
    const f32x4 val = {v, v + 1, v + 2, v + 3};
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const size_t waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    // a wave owns 64 * PITCH16 consecutive 16-byte slots per round and fills them with PITCH16 instructions
    for (size_t base = wave * 64 * PITCH16; base + 64 * PITCH16 <= n16; base += waves * 64 * PITCH16) {
#pragma unroll
        for (int k = 0; k < PITCH16; ++k) {
            f32x4* p = out + base + lane * PITCH16 + k;
            if (NT) __builtin_nontemporal_store(val, p);
            else *p = val;
        }
    }
}

__global__ __launch_bounds__(256) void copier(const f32x4* in, f32x4* out, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}

template <class F>
static float time_ms(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    for (size_t mb : {143, 1024}) {
        const size_t bytes = mb << 20, n16 = bytes / 16;
        f32x4 *d = nullptr, *s = nullptr;
        hipMalloc(&d, bytes); hipMalloc(&s, bytes);
        hipMemset(s, 0, bytes);
        const int grid = 256 * 8;
        auto report = [&](const char* name, float ms, double moved) {
            printf("%5zu MB  %-12s %7.1f us  %6.2f TB/s\n", mb, name, ms * 1e3, moved / ms * 1e-9);
        };
        report("dense", time_ms([&] { writer<1, false><<<grid, 256>>>(d, n16, 1.f); }, 10), (double)bytes);
        report("dense_nt", time_ms([&] { writer<1, true><<<grid, 256>>>(d, n16, 1.f); }, 10), (double)bytes);
        report("halves", time_ms([&] { writer<2, false><<<grid, 256>>>(d, n16, 1.f); }, 10), (double)bytes);
        report("halves_nt", time_ms([&] { writer<2, true><<<grid, 256>>>(d, n16, 1.f); }, 10), (double)bytes);
        report("quarter", time_ms([&] { writer<4, false><<<grid, 256>>>(d, n16, 1.f); }, 10), (double)bytes);
        report("quarter_nt", time_ms([&] { writer<4, true><<<grid, 256>>>(d, n16, 1.f); }, 10), (double)bytes);
        report("copy", time_ms([&] { copier<<<grid, 256>>>(s, d, n16); }, 10), 2.0 * bytes);
        hipFree(d); hipFree(s);
    }
    return 0;
}
