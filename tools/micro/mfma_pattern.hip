// What paces the MFMA burst of the Winograd kernels?  One wave per SIMD, 36 accumulators in AGPRs,
// v_mfma_f32_16x16x4_f32 accumulating in place in the kernels' order (x0, x1, x0, x1), with
// variations: accumulators in VGPRs; operands from 2 or from 144 distinct VGPRs; a ds_read_b128
// per 4 MFMAs (prefetched, not waited for).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool AGPR, bool MANYOPS, bool LDSREAD, int ORDER>   // ORDER 0: x0 x1 x0 x1; 1: x0 x0 x1 x1; 2: all 36 then again
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, long long* ticks) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
    __syncthreads();
    f32x4 acc[36];
    float va[72], vb[72];
    for (int i = 0; i < 72; ++i) { va[i] = threadIdx.x * 1e-3f + i; vb[i] = 1.f + i; }
    for (int x = 0; x < 36; ++x) acc[x] = f32x4{0, 0, 0, 0};
    if (AGPR) { for (int x = 0; x < 36; ++x) asm volatile("" : "+a"(acc[x])); }
    const f32x4* p = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
    f32x4 sink = {0, 0, 0, 0};
    f32x4 q[3] = {{1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int xp = 0; xp < 18; ++xp) {
            // the fragment read here is an MFMA operand two groups later (like the kernels' weights)
            if (LDSREAD) q[(xp + 2) % 3] = p[xp * 64];
            __builtin_amdgcn_sched_barrier(0);
            const int x0 = 2 * xp, x1 = 2 * xp + 1;
            const int o0 = MANYOPS ? 2 * x0 : 0, o1 = MANYOPS ? 2 * x1 : 1;
            const f32x4 w = q[xp % 3];
            auto mf = [&](int x, int o, float wa) {
                const float aop = LDSREAD ? wa : va[o];
                if (AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[x]) : "v"(aop), "v"(vb[o]));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[x]) : "v"(aop), "v"(vb[o]));
            };
            if (ORDER == 0) { mf(x0, o0, w[0]); mf(x1, o1, w[2]); mf(x0, o0 + 1, w[1]); mf(x1, o1 + 1, w[3]); }
            else { mf(x0, o0, w[0]); mf(x0, o0 + 1, w[1]); mf(x1, o1, w[2]); mf(x1, o1 + 1, w[3]); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = sink[0];
    for (int x = 0; x < 36; ++x) { f32x4 c = acc[x]; if (AGPR) asm volatile("s_nop 15\n\ts_nop 3" : "+a"(c)); s += c[0]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <bool AGPR, bool MANYOPS, bool LDSREAD, int ORDER>
void run(float* d, long long* t, const char* name) {
    const int iters = 500;
    for (int rep = 0; rep < 2; ++rep) { k<AGPR, MANYOPS, LDSREAD, ORDER><<<256, 256>>>(d, iters, t); hipDeviceSynchronize(); }
    long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-70s %6.1f cycles per MFMA\n", name, (double)h / (iters * 72));
    fflush(stdout);
}

int main() {
    float* d; long long* t; hipMalloc(&d, 256 * 256 * 4); hipMalloc(&t, 16);
    run<false, false, false, 0>(d, t, "acc in VGPRs, 2 operand registers, order x0 x1 x0 x1");
    run<true, false, false, 0>(d, t, "acc in AGPRs, 2 operand registers, order x0 x1 x0 x1");
    run<true, true, false, 0>(d, t, "acc in AGPRs, 144 operand registers, order x0 x1 x0 x1");
    run<true, true, false, 1>(d, t, "acc in AGPRs, 144 operand registers, order x0 x0 x1 x1");
    run<true, true, true, 0>(d, t, "acc in AGPRs, 144 operand registers, x0 x1 x0 x1, + ds_read_b128 per 4");
    run<false, true, true, 0>(d, t, "acc in VGPRs, 144 operand registers, x0 x1 x0 x1, + ds_read_b128 per 4");
    return 0;
}
