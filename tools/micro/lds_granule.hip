// LDS allocation granularity, from the occupancy calculator: blocks per CU of a 64-thread kernel
// against its dynamic LDS size.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) { extern __shared__ float s[]; s[threadIdx.x] = 1; __syncthreads(); o[threadIdx.x] = s[63 - threadIdx.x]; }
int main() {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int bytes : {54272, 54400, 54613, 40960, 40961, 32768, 32769, 81920, 81921, 114688, 49152}) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, bytes);
        printf("dynamic LDS %6d B -> %d blocks per CU\n", bytes, n);
    }
    return 0;
}
