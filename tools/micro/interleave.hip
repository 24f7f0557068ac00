// One wave per SIMD: fp32 MFMAs (32 cycles each) with N vector instructions between consecutive
// MFMAs of the same wave -- do they run in the MFMA's shadow?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, int KIND>   // KIND 0: v_pk_add_f32, 1: v_add_f32, 2: ds_read_b64 (NV reads), 3: v_pk_fma_f32
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* ticks) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float a = threadIdx.x * 1e-3f, b = 1.f;
    f32x4 acc[16] = {};
    f32x2 v[8];
    for (int j = 0; j < 8; ++j) v[j] = f32x2{a + j, b};
    const f32x2* p = reinterpret_cast<const f32x2*>(lds) + (threadIdx.x & 63);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int r = (j * NV + n) & 7;
                if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[(r + 3) & 7]));
                else if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r][0]) : "v"(v[(r + 3) & 7][0]));
                else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(v[(r + 3) & 7]));
                else { f32x2 q = p[((j * NV + n) & 15) * 64]; asm volatile("" : "+v"(q)); v[r] = q; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += acc[j][0];
    for (int j = 0; j < 8; ++j) s += v[j][0] + v[j][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int NV, int KIND>
void run(float* d, long long* t, const char* name) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) { k<NV, KIND><<<256, 256>>>(d, iters, t); hipDeviceSynchronize(); }
    long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-30s %d per MFMA: %6.1f cycles per MFMA\n", name, NV, (double)h / (iters * 16));
    fflush(stdout);
}

int main() {
    float* d; long long* t; hipMalloc(&d, 256 * 256 * 4); hipMalloc(&t, 16);
    run<0, 0>(d, t, "bare MFMAs");
    run<1, 0>(d, t, "v_pk_add_f32"); run<2, 0>(d, t, "v_pk_add_f32"); run<4, 0>(d, t, "v_pk_add_f32"); run<6, 0>(d, t, "v_pk_add_f32"); run<8, 0>(d, t, "v_pk_add_f32");
    run<2, 1>(d, t, "v_add_f32"); run<4, 1>(d, t, "v_add_f32"); run<6, 1>(d, t, "v_add_f32"); run<8, 1>(d, t, "v_add_f32");
    run<2, 3>(d, t, "v_pk_fma_f32"); run<4, 3>(d, t, "v_pk_fma_f32");
    run<1, 2>(d, t, "ds_read_b64"); run<2, 2>(d, t, "ds_read_b64");
    return 0;
}
