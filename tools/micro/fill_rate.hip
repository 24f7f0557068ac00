// Global -> LDS fill rate per CU (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction) by
// where the bytes come from: one region shared by every workgroup (L2 hits), a private region per
// workgroup that it re-reads (L2 / MALL by size), and a once-read stream (HBM).
//   hipcc --offload-arch=gfx950 -O3 -o fill_rate.bin fill_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void blds16(i32x4_t rsrc, int voffset, int soffset, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voffset), "s"(rsrc), "s"(soffset), "s"(lds_addr) : "memory");
}

// every wave copies `region` bytes (its workgroup's region, waves interleaved by KiB) `reps` times
__global__ __launch_bounds__(256) void fill(const char* base, size_t wg_stride, unsigned region, int reps,
                                            float* out) {
    extern __shared__ float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* p = base + (size_t)blockIdx.x * wg_stride;
    const unsigned long long b = (unsigned long long)p;
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xffff);
    r[2] = __builtin_amdgcn_readfirstlane((int)region);
    r[3] = 0x00020000;
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem + wave * 8192;
    for (int rep = 0; rep < reps; ++rep)
        for (unsigned off = wave * 1024; off < region; off += 8 * 4096) {
#pragma unroll
            for (int u = 0; u < 8; ++u)   // 8 KiB in flight per wave
                blds16(r, lane * 16, __builtin_amdgcn_readfirstlane((int)((off + u * 4096) % region)),
                       lds + u * 1024);
            __builtin_amdgcn_s_waitcnt(0x0f70 | 4);   // vmcnt(4): keep half in flight
        }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = smem[5];
}

int main() {
    const size_t total = 2ull << 30;
    char* d; float* o; hipMalloc(&d, total); hipMemset(d, 1, total); hipMalloc(&o, 4096 * 4);
    hipFuncSetAttribute((const void*)fill, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { const char* name; size_t stride; unsigned region; int blocks; };
    const Case cases[] = {
        {"shared 256 KiB (all WGs the same bytes)", 0, 256u << 10, 512},
        {"shared 2 MiB", 0, 2u << 20, 512},
        {"shared 16 MiB", 0, 16u << 20, 512},
        {"private 32 KiB per WG, re-read (16 MiB in all)", 32u << 10, 32u << 10, 512},
        {"private 256 KiB per WG, re-read (128 MiB in all)", 256u << 10, 256u << 10, 512},
        {"private 1 MiB per WG, re-read (512 MiB in all)", 1u << 20, 1u << 20, 512},
        {"private 4 MiB per WG, read once (2 GiB stream)", 4u << 20, 4u << 20, 512},
        {"shared 2 MiB, 256 WGs", 0, 2u << 20, 256},
    };
    for (const Case& c : cases) {
        const size_t per_wg = 64ull << 20;                      // bytes each WG moves
        const int reps = (int)(per_wg / c.region) > 0 ? (int)(per_wg / c.region) : 1;
        for (int it = 0; it < 2; ++it) {
            hipEventRecord(e0);
            fill<<<c.blocks, 256, 65536>>>(d, c.stride, c.region, reps, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)c.blocks * reps * c.region;
        printf("%-52s %8.3f ms  %6.2f TB/s  %5.1f B/clk/CU (at 2.4 GHz, 256 CUs)\n", c.name, ms, bytes / ms / 1e9,
               bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
    return 0;
}
