// Does an MFMA burst that follows a stretch of vector work start slowly?  One wave per SIMD:
// loop { K packed vector adds ; N MFMAs (16x16x4 f32, in place, 36 accumulators) }, stamps around
// the two parts.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int K, int N, int SLEEP>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, long long* ticks) {
    f32x4 acc[36];
    float va[8], vb[8];
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) { va[i] = threadIdx.x * 1e-3f + i; vb[i] = 1.f + i; v[i] = f32x2{va[i], vb[i]}; }
    for (int x = 0; x < 36; ++x) { acc[x] = f32x4{0, 0, 0, 0}; asm volatile("" : "+a"(acc[x])); }
    long long tv = 0, tm = 0;
    for (int i = 0; i < iters; ++i) {
        long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int j = 0; j < K; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(v[(j + 3) & 7]));
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int j = 0; j < N; ++j)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[j % 36]) : "v"(va[j & 7]), "v"(vb[j & 7]));
        long long t2 = __builtin_amdgcn_s_memtime();
        tv += t1 - t0; tm += t2 - t1;
    }
    float s = 0;
    for (int x = 0; x < 36; ++x) { f32x4 c = acc[x]; asm volatile("s_nop 15\n\ts_nop 3" : "+a"(c)); s += c[0]; }
    for (int j = 0; j < 8; ++j) s += v[j][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { ticks[0] = tv; ticks[1] = tm; }
}

template <int K, int N, int SLEEP = 0>
void run(float* d, long long* t) {
    const int iters = 300;
    for (int rep = 0; rep < 2; ++rep) { k<K, N, SLEEP><<<256, 256>>>(d, iters, t); hipDeviceSynchronize(); }
    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%4d vector ops (+ s_sleep %2d) then %3d MFMAs: vector part %6.0f cycles, MFMA burst %6.0f cycles = %5.1f per MFMA (issue of the last)\n",
           K, SLEEP, N, (double)h[0] / iters, (double)h[1] / iters, (double)h[1] / iters / N);
    fflush(stdout);
}

int main() {
    float* d; long long* t; hipMalloc(&d, 256 * 256 * 4); hipMalloc(&t, 16);
    run<0, 72>(d, t); run<16, 72>(d, t); run<64, 72>(d, t); run<144, 72>(d, t); run<256, 72>(d, t);
    run<144, 18>(d, t); run<144, 36>(d, t); run<144, 144>(d, t); run<144, 288>(d, t);
    run<0, 72, 8>(d, t); run<0, 72, 32>(d, t);
    return 0;
}
