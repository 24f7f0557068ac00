// Cost of a dependent kernel launch: a chain of N tiny kernels on one stream, as plain launches
// and as a captured graph.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void wide(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }   // 512 workgroups
int main() {
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 400;
    for (int grid : {1, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, s);
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d);
            hipEventRecord(e1, s); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("stream launches, grid %3d: %.2f us per dependent launch\n", grid, ms * 1e3 / N);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, s);
            hipGraphLaunch(ge, s);
            hipEventRecord(e1, s); hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        printf("graph of %d nodes, grid %3d: %.2f us per node\n", N, grid, ms * 1e3 / N);
    }
    return 0;
}
