// Do a wave's fp32 MFMAs and its SIMD partner's vector / scalar / LDS instructions overlap?
// 512-thread workgroups (two waves per SIMD): waves 0-3 issue MFMAs, waves 4-7 a stream of one
// other kind; each alone, then together.   hipcc --offload-arch=gfx950 -O3 -o coissue.bin coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode bit 0: first half runs MFMAs (kind mk: 0 = f32 16x16x4, 1 = f32 32x32x2, 2 = bf16 32x32x16)
// mode bit 1: second half runs the side stream (kind sk: 0 = v_pk_add_f32, 1 = v_add_f32,
//             2 = ds_read_b64, 3 = s_add (SALU), 4 = v_readlane)
template <int mk, int sk>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, long long* ticks, int prio) {
    __shared__ float lds[4096];
    const int half = threadIdx.x >> 8;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    float a = threadIdx.x * 1e-3f, b = 1.f, s = 0;
    if (half == 0) {
        if (mode & 1) {
            if constexpr (mk == 0) {
                f32x4 acc[16] = {};
                for (int i = 0; i < iters; ++i)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
                for (int j = 0; j < 16; ++j) s += acc[j][0];
            } else if constexpr (mk == 1) {
                f32x16 acc[4] = {};
                for (int i = 0; i < 2 * iters; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                for (int j = 0; j < 4; ++j) s += acc[j][0];
            } else {
                f32x16 acc[4] = {};
                bf16x8 av, bv;
                for (int e = 0; e < 8; ++e) { av[e] = (__bf16)a; bv[e] = (__bf16)b; }
                for (int i = 0; i < 4 * iters; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
                for (int j = 0; j < 4; ++j) s += acc[j][0];
            }
        }
    } else if (mode & 2) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        if constexpr (sk == 0) {
            f32x2 v[8];
            for (int j = 0; j < 8; ++j) v[j] = f32x2{a + j, b};
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = v[j] + v[(j + 1) & 7];
            for (int j = 0; j < 8; ++j) s += v[j][0] + v[j][1];
        } else if constexpr (sk == 1) {
            float v[8];
            for (int j = 0; j < 8; ++j) v[j] = a + j;
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = v[j] + v[(j + 1) & 7];
            for (int j = 0; j < 8; ++j) s += v[j];
        } else if constexpr (sk == 2) {
            f32x2 v = {0, 0};
            const f32x2* p = reinterpret_cast<const f32x2*>(lds) + (threadIdx.x & 63);
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int r = 0; r < 64; ++r) {
                    f32x2 q = p[(r & 15) * 64];
                    asm volatile("" : "+v"(q));
                    v += q;
                }
            s = v[0] + v[1];
        } else if constexpr (sk == 3) {
            int x = __builtin_amdgcn_readfirstlane(threadIdx.x);
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int r = 0; r < 64; ++r) asm volatile("s_add_i32 %0, %0, 3\n\ts_xor_b32 %0, %0, 5" : "+s"(x) : : "scc");
            s = x;
        } else {
            int x = threadIdx.x, y = 0;
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int r = 0; r < 64; ++r) {
                    int t;
                    asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(t) : "v"(x));
                    y += t;
                }
            s = y;
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256))
        ticks[threadIdx.x >> 8] = __builtin_amdgcn_s_memtime() - t0;
}

template <int mk, int sk>
void run(float* d, long long* t, const char* name, int prio = 0) {
    long long h[3][2];
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            k<mk, sk><<<256, 512>>>(d, 2000, mode, t, prio);
            hipDeviceSynchronize();
        }
        hipMemcpy(h[mode - 1], t, 16, hipMemcpyDeviceToHost);
    }
    printf("%-52s MFMA alone %8lld  side alone %8lld  together: MFMA %8lld  side %8lld cycles\n", name, h[0][0], h[1][1],
           h[2][0], h[2][1]);
    fflush(stdout);
}

int main() {
    float* d; long long* t; hipMalloc(&d, 256 * 512 * 4); hipMalloc(&t, 16);
    run<0, 0>(d, t, "f32 16x16x4 | v_pk_add_f32");
    run<0, 1>(d, t, "f32 16x16x4 | v_add_f32");
    run<0, 2>(d, t, "f32 16x16x4 | ds_read_b64");
    run<0, 3>(d, t, "f32 16x16x4 | SALU");
    run<0, 4>(d, t, "f32 16x16x4 | v_readlane_b32");
    run<1, 0>(d, t, "f32 32x32x2 | v_pk_add_f32");
    run<1, 1>(d, t, "f32 32x32x2 | v_add_f32");
    run<2, 0>(d, t, "bf16 32x32x16 | v_pk_add_f32");
    run<2, 1>(d, t, "bf16 32x32x16 | v_add_f32");
    run<0, 0>(d, t, "f32 16x16x4 | v_pk_add_f32, side at s_setprio 3", 1);
    run<0, 2>(d, t, "f32 16x16x4 | ds_read_b64, side at s_setprio 3", 1);
    run<0, 4>(d, t, "f32 16x16x4 | v_readlane_b32, side at s_setprio 3", 1);
    run<1, 0>(d, t, "f32 32x32x2 | v_pk_add_f32, side at s_setprio 3", 1);
    run<2, 0>(d, t, "bf16 32x32x16 | v_pk_add_f32, side at s_setprio 3", 1);
    return 0;
}
