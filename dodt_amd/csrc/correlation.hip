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
// 25 displacements.  Here a workgroup owns a 16-column x 32-row pixel tile; HBM-bound by design
// (A and B read once, 2 x 71.7 MB, 56 MB written), so what the kernel has to keep small is the work
// per byte:
//   * the (16+2R) x (32+2R) neighbourhood of B is staged in LDS in two passes of 16 channels, pixel
//     stride 20 floats (16 + 4 pad): the 16 lanes one LDS cycle of a ds_read_b128 serves --
//     consecutive pixels of a row -- fall into 16 different 16-byte bank columns, and all 120 reads
//     of a pass are one lane address + immediates (77 KB: two workgroups per CU, one stages while the
//     other computes);
//   * a lane owns the TWO pixels (y, x) and (y + 2, x): with stride_2 = 2 their displacement rows
//     overlap, a B pixel read from LDS feeds both (6 neighbourhood rows instead of 10 per column of
//     displacements: 120 ds_read_b128 per output pixel instead of 200);
//   * the channel sum of an output is kept as two float32 chains (even / odd channels) advanced by one
//     v_pk_fma_f32 per channel pair -- the reference's CUDA kernel fuses its multiply-adds too -- and
//     added at the end: 400 vector instructions per pixel instead of 1600 (tests: 1e-5 of the scale
//     against the oracle's sequential float32 sum);
//   * results leave through LDS as full rows (16-byte stores);
//   * workgroup b takes tile (b % 8) * ceil(n / 8) + b / 8: the eight XCDs (blocks are dealt to them
//     round-robin) each sweep a contiguous band of the map, so that the halo a tile shares with its
//     neighbours is an L2 hit instead of a second fetch over the fabric (FETCH_SIZE x 2: 385 -> 299 MB
//     per launch).
// Measured at (700,800,32): 64 us = 3.1 TB/s of algorithmic bytes (round 2's one-pixel-per-lane kernel:
// 98 us); with 354 MB at the L2's fabric side that is 5.5 TB/s of real traffic, i.e. what is left is
// the halo: the rows a tile shares with the tile below are fetched again one round later.  Tried and
// slower: four 8-channel passes with three workgroups per CU (77 us), A read once for both passes with
// non-temporal loads and stores (89 us).
#include "common.h"

namespace {

constexpr int kTW = 16;       // tile columns = lanes of a row
constexpr int kTH = 32;       // tile rows: 16 lane rows x 2 pixels (y, y + 2)
constexpr int kC = 32;        // channels (the BEV pyramid's output depth)
constexpr int kCH = 16;       // channels staged per pass (general kernel)
constexpr int kPS = kCH + 4;  // LDS floats per pixel (16 + 4 pad: conflict-free b128 reads)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// R = 2 * S2 is the only shape the pairing is written for (max_displacement 4 or 5, stride_2 2)
// CH channels per pass, pixel stride CH + 4 floats (80 or 48 bytes: 5 p or 3 p mod 16 bank columns are
// distinct for 16 consecutive pixels p); WGS workgroups per CU (registers and LDS permitting)
template <int S2, int CH, int WGS>
__global__ void __launch_bounds__(256, WGS)
correlation_kernel(const float* __restrict__ A, const float* __restrict__ B, int H, int W,
                   int d, int pad, int OH, int OW, int tiles_x, int n_tiles, float* __restrict__ out) {
    constexpr int r = 2, R = r * S2, GW = 2 * r + 1, K = GW * GW, PS = CH + 4;
    constexpr int PW = kTW + 2 * R, PH = kTH + 2 * R;      // patch: 24 x 40 pixels
    static_assert(S2 == 2, "a lane's pixel pair (y, y + 2) shares neighbourhood rows only for stride_2 = 2");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    // XCD-aware tile order (see above); a bijection of [0, 8 * per) of which n_tiles are real
    const int per = (n_tiles + 7) / 8;
    const int tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (tile >= n_tiles) return;
    const int oy0 = (tile / tiles_x) * kTH, ox0 = (tile % tiles_x) * kTW;
    const int shift = d - pad;         // output (y,x) looks at input (y+shift, x+shift)
    // Lane -> pixel.  A ds_read_b128 is served in four groups of 16 lanes that are NOT consecutive --
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- and only
    // lanes of one group can conflict: each group is made one row of 16 consecutive pixels, whose
    // 80-byte pixel stride then spreads them over the 16 bank columns (5 p mod 16).  (Mapped the
    // plain way, lane = 16 row + column, a group mixes columns 0-3, 12-15 of one row with 4-11 of the
    // next, 24 pixels on: every bank column hit twice, SQ_LDS_BANK_CONFLICT 47 % of the LDS cycles.)
    const int l5 = tid & 31;
    const bool grp_a = l5 < 4 || (l5 >= 12 && l5 < 16) || (l5 >= 20 && l5 < 28);
    const int lx = grp_a ? (l5 < 4 ? l5 : l5 < 16 ? l5 - 8 : l5 - 12)
                         : (l5 < 12 ? l5 - 4 : l5 < 20 ? l5 - 8 : l5 - 16);
    const int lr = (tid >> 5) * 2 + (grp_a ? 0 : 1);     // lane row 0..15
    const int ly = 4 * (lr >> 1) + (lr & 1);              // first pixel row of the lane: 0,1,4,5,8,9,...
    f32x2 res[2][K];    // (even-channel chain, odd-channel chain) per output
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int k = 0; k < K; ++k) res[px][k] = f32x2{0.f, 0.f};
#pragma unroll 1
    for (int pass = 0; pass < kC / CH; ++pass) {
        if (pass) __syncthreads();     // everyone is done reading the previous channels
        // the lane's two A pixels, this pass's 16 channels (registers: the outputs' 100 accumulators
        // leave room for one pass of A and half a staging trip set at a time)
        f32x4 ah[2][CH / 4];
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            const int ay = oy0 + ly + 2 * px + shift, ax = ox0 + lx + shift;
            const bool in = ay >= 0 && ay < H && ax >= 0 && ax < W;
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) {
                ah[px][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (in)
                    ah[px][q] = *reinterpret_cast<const f32x4*>(A + ((size_t)ay * W + ax) * kC + pass * CH + q * 4);
            }
        }
        // stage the B neighbourhood (zero outside the image = the reference's zero padding): one
        // 16-byte quad per lane and trip, 15 trips in two groups, a group's loads issued before its
        // first write
        constexpr int kQuads = PW * PH * (CH / 4), kTrips = (kQuads + 255) / 256, kGroup = kTrips < 8 ? kTrips : 8;
        // (opaque copy of the lane id: the 15 trips' addresses are recomputed in every pass instead of
        //  living in registers across the compute loop, where the accumulators need them)
        int stid = tid;
        asm volatile("" : "+v"(stid));
#pragma unroll
        for (int g0 = 0; g0 < kTrips; g0 += kGroup) {
            f32x4 stage[kGroup];
#pragma unroll
            for (int i = 0; i < kGroup; ++i) {
                const int t = stid + 256 * (g0 + i);
                const int q = t % (CH / 4), p = t / (CH / 4);
                const int py = p / PW, pxl = p - py * PW;
                const int gy = oy0 + shift - R + py, gx = ox0 + shift - R + pxl;
                stage[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (g0 + i < kTrips && t < kQuads && gy >= 0 && gy < H && gx >= 0 && gx < W)
                    stage[i] = *reinterpret_cast<const f32x4*>(B + ((size_t)gy * W + gx) * kC + pass * CH + q * 4);
            }
#pragma unroll
            for (int i = 0; i < kGroup; ++i) {
                const int t = stid + 256 * (g0 + i);
                const int q = t % (CH / 4), p = t / (CH / 4);
                if (g0 + i < kTrips && t < kQuads)
                    *reinterpret_cast<f32x4*>(smem + p * PS + q * 4) = stage[i];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // neighbourhood rows y + 2 (tr - 2) of the lane's first pixel, tr = 0..5: row tr feeds output row
        // tr of pixel 0 (tr <= 4) and output row tr - 1 of pixel 1 (tr >= 1)
#pragma unroll
        for (int tr = 0; tr < GW + 1; ++tr) {
#pragma unroll
            for (int j = 0; j < GW; ++j) {
                const float* b = smem + (ly * PW + lx) * PS + (S2 * tr * PW + S2 * j) * PS;
                f32x4 bv[CH / 4];
#pragma unroll
                for (int q = 0; q < CH / 4; ++q) bv[q] = *reinterpret_cast<const f32x4*>(b + q * 4);
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const int krow = tr - px;
                    if (krow < 0 || krow >= GW) continue;          // compile time
                    f32x2 sum = res[px][krow * GW + j];
#pragma unroll
                    for (int q = 0; q < CH / 4; ++q) {
                        const f32x4 av = ah[px][q];
                        sum = __builtin_elementwise_fma(f32x2{av[0], av[1]}, f32x2{bv[q][0], bv[q][1]}, sum);
                        sum = __builtin_elementwise_fma(f32x2{av[2], av[3]}, f32x2{bv[q][2], bv[q][3]}, sum);
                    }
                    res[px][krow * GW + j] = sum;
                }
            }
            // keep the six neighbourhood rows apart (hoisted together, a pass's 120 LDS reads spill);
            // inside a row the scheduler runs the next pixels' reads under the current one's multiply-adds
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();   // the patch is dead: reuse LDS as the [pixel][K] output tile
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int k = 0; k < K; ++k)
            smem[((ly + 2 * px) * kTW + lx) * K + k] = (res[px][k][0] + res[px][k][1]) * (1.0f / (float)kC);   // exact: a power of two
    __syncthreads();
    // rows of the tile are contiguous in the output: kTW * K floats each
    constexpr int kRow = kTW * K;
    if ((OW * K) % 4 == 0 && kRow % 4 == 0 && ox0 + kTW <= OW) {
        for (int t = tid; t < kTH * (kRow / 4); t += 256) {
            const int row = t / (kRow / 4), off = t - row * (kRow / 4);
            const int y = oy0 + row;
            if (y < OH)
                *reinterpret_cast<f32x4*>(out + ((size_t)y * OW + ox0) * K + off * 4) =
                    *reinterpret_cast<const f32x4*>(smem + row * kRow + off * 4);
        }
    } else {
        for (int t = tid; t < kTH * kRow; t += 256) {
            const int row = t / kRow, off = t - row * kRow;
            const int y = oy0 + row, x = ox0 + off / K;
            if (y < OH && x < OW) out[((size_t)y * OW + ox0) * K + off] = smem[t];
        }
    }
}

// ---- any other displacement grid (<= 32 displacements): one pixel per lane, a 16 x 16 tile, the
//      neighbourhood padded to 20 floats per pixel; the form the fast kernel above grew out of -----------
constexpr int kT = 16;        // tile edge


// The displacement loop stays a runtime loop: fully unrolled (compile-time r, s2) hipcc hoists
// all 100 LDS reads of a pass and spills (measured 528 us against 101 us for this form).
__global__ void __launch_bounds__(256, 3)
correlation_generic_kernel(const float* __restrict__ A, const float* __restrict__ B, int H, int W,
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
    if (!(stride_2 == 2 && r == 2)) {
        // not the DODT configuration's 5 x 5 grid of displacements two pixels apart
        // (correlation.py:7: max_displacement 5, stride_2 2): the general kernel
        const int PT = kT + 2 * r * stride_2;
        size_t glds = (size_t)PT * PT * kPS * sizeof(float);
        const size_t lds_out = (size_t)kT * kT * K * sizeof(float);
        if (glds < lds_out) glds = lds_out;
        DODT_REQUIRE(glds <= 160 * 1024, "dodt_correlation: neighbourhood does not fit LDS");
        static bool gprepared = false;
        if (!gprepared) {
            DODT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&correlation_generic_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            gprepared = true;
        }
        hipLaunchKernelGGL(correlation_generic_kernel, dim3(dodt::ceil_div(OW, kT), dodt::ceil_div(OH, kT)),
                           dim3(256), glds, ctx->stream, d_a, d_b, H, W, max_displacement, pad, stride_2, r, OH,
                           OW, d_out);
        DODT_LAUNCH_CHECK();
        return DODT_OK;
    }
    const int R = r * stride_2;
    // (four 8-channel passes with three workgroups per CU were measured as well: 77 us against 65)
    const int tiles_x = dodt::ceil_div(OW, kTW), n_tiles = tiles_x * dodt::ceil_div(OH, kTH);
    auto go = [&](auto kernel, int CH) -> int {
        size_t lds = (size_t)(kTW + 2 * R) * (kTH + 2 * R) * (CH + 4) * sizeof(float);
        const size_t lds_out = (size_t)kTW * kTH * K * sizeof(float);
        if (lds < lds_out) lds = lds_out;
        static bool prepared = false;      // (one per instantiation of this lambda)
        if (!prepared) {
            DODT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            prepared = true;
        }
        hipLaunchKernelGGL(kernel, dim3(8 * dodt::ceil_div(n_tiles, 8)), dim3(256), lds, ctx->stream,
                           d_a, d_b, H, W, max_displacement, pad, OH, OW, tiles_x, n_tiles, d_out);
        return DODT_OK;
    };
    if (int rc = go(&correlation_kernel<2, 16, 2>, 16)) return rc;
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}
