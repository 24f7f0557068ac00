// Measured ceiling of the fp32 / bf16 MFMA shapes on this chip: N waves per SIMD issuing
// independent MFMAs back to back.   hipcc --offload-arch=gfx950 -O3 -o mfma_peak.bin mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Kind { const char* name; double flop; };   // flop per wave per loop iteration
static const Kind kinds[] = {
    {"f32 16x16x4   (8 acc sets)", 8 * 2048.0},  {"f32 16x16x4   (16 acc sets)", 16 * 2048.0},
    {"f32 32x32x2   (4 acc sets)", 4 * 4096.0},  {"f32 16x16x1 4B (4 acc sets)", 4 * 2048.0},
    {"f32 32x32x1 2B (2 acc sets)", 2 * 4096.0}, {"f32 4x4x1 16B (8 acc sets)", 8 * 512.0},
    {"bf16 32x32x16 (4 acc sets)", 4 * 32768.0},
    {"f32 16x16x4   (1 acc set: back to back dependent)", 8 * 2048.0}, {"f32 16x16x4   (2 acc sets)", 8 * 2048.0},
    {"f32 16x16x4   (4 acc sets)", 8 * 2048.0}, {"f32 32x32x2   (1 acc set)", 4 * 4096.0}, {"f32 32x32x2   (2 acc sets)", 4 * 4096.0},
};

template <int kind>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* ticks) {
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x * 1e-3f, b = 1.f, s = 0;
    if constexpr (kind == 0 || kind == 1 || kind == 5) {
        constexpr int N = kind == 1 ? 16 : 8;
        f32x4 acc[N] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j)
                acc[j] = kind == 5 ? __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        for (int j = 0; j < N; ++j) s += acc[j][0];
    } else if constexpr (kind == 2 || kind == 3) {
        f32x16 acc[4] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = kind == 2 ? __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[j], 0, 0, 0);
        for (int j = 0; j < 4; ++j) s += acc[j][0];
    } else if constexpr (kind == 4) {
        f32x32 acc[2] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, acc[j], 0, 0, 0);
        for (int j = 0; j < 2; ++j) s += acc[j][0];
    } else if constexpr (kind >= 7 && kind <= 9) {
        constexpr int N = kind == 7 ? 1 : kind == 8 ? 2 : 4;
        f32x4 acc[N] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j % N] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j % N], 0, 0, 0);
        for (int j = 0; j < N; ++j) s += acc[j][0];
    } else if constexpr (kind == 10 || kind == 11) {
        constexpr int N = kind == 10 ? 1 : 2;
        f32x16 acc[N] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j % N] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j % N], 0, 0, 0);
        for (int j = 0; j < N; ++j) s += acc[j][0];
    } else {
        f32x16 acc[4] = {};
        bf16x8 av, bv;
        for (int e = 0; e < 8; ++e) { av[e] = (__bf16)a; bv[e] = (__bf16)b; }
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
        for (int j = 0; j < 4; ++j) s += acc[j][0];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ticks[0] = __builtin_amdgcn_s_memtime() - t0;
        ticks[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int kind>
void run(float* d, long long* t) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) {     // 1, 2, 4 waves per SIMD
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            k<kind><<<blocks, 256>>>(d, iters, t);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("%-50s %d waves/SIMD: %7.3f ms %7.1f TFLOP/s   first block: %.0f MHz shader clock\n", kinds[kind].name,
               blocks / 256, ms, kinds[kind].flop * iters * 4.0 * blocks / ms / 1e9, 100.0 * h[0] / h[1]);
    }
}

int main() {
    float* d; long long* t; hipMalloc(&d, 4096 * 256 * 4); hipMalloc(&t, 16);
    run<0>(d, t); run<1>(d, t); run<2>(d, t); run<3>(d, t); run<4>(d, t); run<5>(d, t); run<6>(d, t); run<7>(d, t); run<8>(d, t); run<9>(d, t); run<10>(d, t); run<11>(d, t);
    return 0;
}
