// FlowNet-style correlation of two BEV feature maps for gfx950 (the
// T branch): the reference's only hand-written CUDA op, avod/core/ops/correlation/
// correlation_kernel.cu.cc:21-119 (CorrelateData) + pad.cu.cc:14-73 (PadData), called with
// kernel_size 1, stride_1 1, stride_2 2, max_displacement = pad = 5
// (avod/core/models/dt_rpn_model.py:324-333, corr_layers/correlation.py:7).
//
//   out[y, x, k] = 1/C * sum_c A[y', x', c] * B[y' + s2p, x' + s2o, c]     (zero padded)
//   y' = y + d - pad, k = (s2p/s2 + r) * (2r+1) + (s2o/s2 + r), r = d / s2
//
// The reference launches one 32-thread block per output pixel and loops serially over the
// 25 displacements.  Here a workgroup owns a 16x16 pixel tile: the (16+2R)^2 neighbourhood
// of B is staged in LDS in two passes of 16 channels (pixel stride 20 floats -> conflict-free
// b128 reads; 46 KB, three workgroups per CU so that one stages while the others compute),
// each lane keeps its A pixel (32 channels) in registers and produces all (2r+1)^2 outputs,
// which leave through LDS as full rows.  HBM-bound: reads A and B once (2 x 71.7 MB), writes
// 56 MB.  Channel sums are sequential in float32, like the reference's lane-0 reduction.
#include "common.h"

namespace {

constexpr int kT = 16;        // tile edge
constexpr int kC = 32;        // channels (the BEV pyramid's output depth)
constexpr int kCH = 16;       // channels staged per pass
constexpr int kPS = kCH + 4;  // LDS floats per pixel (16 + 4 pad: conflict-free b128 reads)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The displacement loop stays a runtime loop: fully unrolled (compile-time r, s2) hipcc hoists
// all 100 LDS reads of a pass and spills (measured 528 us against 101 us for this form).
__global__ void __launch_bounds__(256, 3)
correlation_kernel(const float* __restrict__ A, const float* __restrict__ B, int H, int W,
                   int d, int pad, int s2, int r, int OH, int OW, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int R = r * s2;              // neighbourhood radius in pixels
    const int PT = kT + 2 * R;         // patch edge
    const int gw = 2 * r + 1, K = gw * gw;
    const int tid = threadIdx.x;
    const int oy0 = blockIdx.y * kT, ox0 = blockIdx.x * kT;
    const int shift = d - pad;         // output (y,x) looks at input (y+shift, x+shift)
    const int ly = tid / kT, lx = tid % kT;
    const int oy = oy0 + ly, ox = ox0 + lx;
    const int ay = oy + shift, ax = ox + shift;
    f32x4 a[kC / 4];
    const bool a_in = ay >= 0 && ay < H && ax >= 0 && ax < W;
#pragma unroll
    for (int q = 0; q < kC / 4; ++q) {
        a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a_in) a[q] = *reinterpret_cast<const f32x4*>(A + ((size_t)ay * W + ax) * kC + q * 4);
    }
    float res[32];   // host checks K <= 32
#pragma unroll
    for (int k = 0; k < 32; ++k) res[k] = 0.0f;
#pragma unroll
    for (int pass = 0; pass < kC / kCH; ++pass) {
        if (pass) __syncthreads();     // everyone is done reading the previous channels
        // stage the B neighbourhood (zero outside the image = the reference's zero padding)
        for (int t = tid; t < PT * PT * (kCH / 4); t += 256) {
            const int q = t % (kCH / 4), p = t / (kCH / 4);
            const int py = p / PT, px = p - py * PT;
            const int gy = oy0 + shift - R + py, gx = ox0 + shift - R + px;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const f32x4*>(B + ((size_t)gy * W + gx) * kC + pass * kCH +
                                                    q * 4);
            *reinterpret_cast<f32x4*>(smem + p * kPS + q * 4) = v;
        }
        __syncthreads();
        // channel sums stay one sequential float32 chain per output, c = 0 .. 31
        for (int k = 0; k < K; ++k) {
            const int s2p = (k / gw - r) * s2, s2o = (k % gw - r) * s2;
            const float* b = smem + ((ly + R + s2p) * PT + (lx + R + s2o)) * kPS;
            float sum = res[k];
#pragma unroll
            for (int q = 0; q < kCH / 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(b + q * 4);
                const f32x4 av = a[pass * (kCH / 4) + q];
                sum += av[0] * bv[0];
                sum += av[1] * bv[1];
                sum += av[2] * bv[2];
                sum += av[3] * bv[3];
            }
            res[k] = sum;
        }
    }
    for (int k = 0; k < K; ++k) res[k] = res[k] / (float)kC;
    __syncthreads();   // the patch is dead: reuse LDS as the [pixel][K] output tile
    for (int k = 0; k < K; ++k) smem[tid * K + k] = res[k];
    __syncthreads();
    // rows of the tile are contiguous in the output: kT * K floats each
    for (int t = tid; t < kT * kT * K; t += 256) {
        const int row = t / (kT * K), off = t - row * (kT * K);
        const int y = oy0 + row, x = ox0 + off / K;
        if (y < OH && x < OW) out[((size_t)y * OW + ox0) * K + off] = smem[t];
    }
}

}  // namespace

extern "C" int dodt_correlation(dodt_ctx* ctx, const float* d_a, const float* d_b, int H, int W,
                                int C, int max_displacement, int stride_2, int pad,
                                float* d_out) {
    DODT_REQUIRE(ctx && d_a && d_b && d_out, "dodt_correlation: NULL argument");
    DODT_REQUIRE(C == kC, "dodt_correlation: C = %d, only %d channels are supported", C, kC);
    DODT_REQUIRE(H > 0 && W > 0 && max_displacement >= 0 && stride_2 >= 1 && pad >= 0,
                 "dodt_correlation: bad sizes");
    const int r = max_displacement / stride_2;
    const int K = (2 * r + 1) * (2 * r + 1);
    DODT_REQUIRE(K <= 32, "dodt_correlation: %d displacement channels exceed 32", K);
    // correlation_op.cc:36-40: out = ceil((in + 2*pad - 2*(max_displacement + 0)) / stride_1)
    const int OH = H + 2 * pad - 2 * max_displacement, OW = W + 2 * pad - 2 * max_displacement;
    DODT_REQUIRE(OH >= 1 && OW >= 1, "dodt_correlation: empty output");
    const int PT = kT + 2 * r * stride_2;
    size_t lds = (size_t)PT * PT * kPS * sizeof(float);
    const size_t lds_out = (size_t)kT * kT * K * sizeof(float);
    if (lds < lds_out) lds = lds_out;
    DODT_REQUIRE(lds <= 160 * 1024, "dodt_correlation: neighbourhood does not fit LDS");
    // measured: 101 us at (700,800,32), 2 TB/s of HBM traffic
    static bool prepared = false;
    if (!prepared) {
        DODT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&correlation_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024));
        prepared = true;
    }
    hipLaunchKernelGGL(correlation_kernel, dim3(dodt::ceil_div(OW, kT), dodt::ceil_div(OH, kT)), dim3(256),
                       lds, ctx->stream, d_a, d_b, H, W, max_displacement, pad, stride_2, r, OH,
                       OW, d_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}
