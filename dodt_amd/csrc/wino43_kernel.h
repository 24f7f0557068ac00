// 3x3 stride-1 convolution in fp32 by Winograd F(4x4, 3x3) on the fp32 MFMA of gfx950.
//
// Why a second Winograd kernel: measured on this chip (tools/micro/coissue.hip, interleave.hip) the
// fp32 MFMA and the vector ALU of a SIMD do not overlap -- a wave's vector instructions wait for
// the MFMAs of its SIMD, whichever wave issued them -- so the F(2x2,3x3) kernel (wino_kernels.h) is
// bound by MFMA cycles + transform cycles, and only fewer MFMAs per output make it faster.  F(4x4,
// 3x3) computes a 4x4 block of outputs from a 6x6 block of inputs with 36 multiplications per
// (input channel, output channel) instead of 144: 2.25 per output against 4 (F(2x2)) and 9 (direct).
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A      with the interpolation points 0, +-2/3, +-3/2, inf
// fp32 throughout (fused multiply-adds in the transforms), the filter transform runs on the host in
// float64.  Error against the direct form: ~1.5e-6 of a layer's scale (round 2's points 0, +-1, +-2:
// 3-5e-6; F(2x2): 2e-7; the per-layer bar of tests/test_gpu_conv.py is 1e-4) -- which is why this
// kernel is the OPT-IN fast form (DODT_CONV_WINO=4) and F(2x2,3x3) the default: DESIGN.md 5.0.
//
// Work decomposition (one workgroup = 4 waves, ONE per CU: 512 registers per lane):
//   * workgroup tile 32 x 16 output pixels x 32 output channels; wave w: tile block w & 1 (16 x 16
//     pixels = 4 x 4 Winograd tiles) and channel block w >> 1 (16 channels): 36 points x one
//     16x16x4 accumulator = 144 registers.  Lane l: tile t = l % 16 (B operand column / the
//     output's pixel block), k-pair g = l / 16 (input channels 2g, 2g + 1 of the chunk).
//   * K in chunks of 8 input channels (one CB8 plane); per chunk the 18 x 34 pixel input patch and
//     the chunk's transformed weights [xi pair][g][cb][n][xi & 1][2] are copied global -> LDS by
//     LDS-DMA, double buffered, one barrier per chunk; persistent workgroups walk a work queue.
//   * per chunk a lane reads its tile's 6 x 6 pixels (36 ds_read_b64: base + immediate), applies
//     B^T d B (144 packed fused multiply-adds), then issues 72 MFMAs.
//   * epilogue: A^T M A per lane (4 x 4 pixels x 4 channels), batch-norm + ReLU, 16-byte stores
//     into the CB8 output; a lane's 4 x 4 outputs hold four windows of a following 2x2 max pool.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "conv_kernels.h"
#include "lds_dma.h"
#include "wino_kernels.h"

namespace dodt {

struct Wino43Cfg {
    static constexpr int TW = 32, TH = 16, BN = 32;
    static constexpr int PH = TH + 2, PW = TW + 2;
    // LDS patch image: pixel (py, px) lives in 32-byte cell py * 34 + px + (py >> 2), its two
    // 16-byte halves swapped when (px >> 3) & 1: for every tap (r, c) of the 6x6 input tile the 16
    // lanes x 2 that one LDS cycle of a ds_read_b64 serves (4 x 4 tiles, pixels four apart) fall
    // into 16 different 16-byte bank columns (tools/lds_swizzle_search.py reproduces the search and
    // checks this and the other kernels' images).
    // Filled by LDS-DMA (lane-linear): the permutation is applied to the SOURCE address of a slot.
    static constexpr int kPitch = PW;
    static constexpr int kCells = (PH - 1) * kPitch + (PW - 1) + ((PH - 1) >> 2) + 1;   // 616
    static constexpr int kPatchSlots = kCells * 2;
    static constexpr int kPatchInstr = (kPatchSlots + 63) / 64;       // 20
    static constexpr int kPatchFloats = kPatchInstr * 256;
    static constexpr int kWFloats = 36 * 8 * BN;                      // 9216: 36 copies
    static constexpr int kWInstr = kWFloats / 256;
    static constexpr int kBufFloats = kPatchFloats + kWFloats;
    // exactly 112 KB: a 48 KB workgroup of the FC kernel (gemm.hip) fits beside it on a CU, so the
    // tail's GEMMs can use the vector-pipe time this kernel's waves spend waiting.  The work queue's
    // ticket travels through a per-workgroup mailbox in global memory and the bottleneck's
    // exchange area aliases a patch image that is free during an epilogue.
    static constexpr int kLdsBytes = 2 * kBufFloats * 4;
    static constexpr int kPatchPerWave = (kPatchInstr + 3) / 4;       // 5
    static constexpr int kWPerWave = kWInstr / 4;                     // 9
};

constexpr int tap_off_c(int r, int c) {
    return (r * Wino43Cfg::kPitch + (r >> 2)) * 32 + (c & 3) * 32;
}
__device__ __forceinline__ f32x2_t pk_fma(float k, f32x2_t a, f32x2_t b) {   // k * a + b
    return __builtin_elementwise_fma(f32x2_t{k, k}, a, b);
}
// Interpolation points of F(4,3): 0, +-a, +-b, infinity with a = 2/3, b = 3/2 (a b = 1).  Round 2 used
// Lavin & Gray's 0, +-1, +-2, inf; what makes F(4x4,3x3) noisy in fp32 is the accumulation over the input
// channels in the TRANSFORMED domain, whose values are larger than the outputs they cancel to, and that
// amplification depends on the points: searched over symmetric sets (tests/experiments/wino_points.py),
// the best lie at a ~ 0.65, b ~ 1.5 with 2.2x less rms error than (1, 2) -- the layer error against the
// direct form falls from 4e-6 to 1.5e-6 of the scale (F(2x2,3x3): 2e-7), for two more packed operations
// per 1-D transform.  With a b = 1:
//     B^T = [ 1   0  -s   0   1   0 ]      s = a^2 + b^2 = 97/36         A^T = [ 1  1    1    1    1   0 ]
//           [ 0  -b  -b2  a   1   0 ]      b2 = b^2 = 9/4                      [ 0  a   -a    b   -b   0 ]
//           [ 0   b  -b2 -a   1   0 ]      a2 = a^2 = 4/9                      [ 0  a2   a2   b2   b2  0 ]
//           [ 0  -a  -a2  b   1   0 ]                                          [ 0  a3  -a3   b3  -b3  1 ]
//           [ 0   a  -a2 -b   1   0 ]
//           [ 0   1   0  -s   0   1 ]      G[j][k] = p_j^k / prod_{l != j} (p_j - p_l), last row (0 0 1)
// (the filter transform G g G^T runs on the host in float64: conv.hip; tests/test_wino_transforms.py checks
// these matrices against the direct convolution).
constexpr float kWa = 2.0f / 3.0f, kWb = 1.5f, kWa2 = 4.0f / 9.0f, kWb2 = 2.25f, kWs = 97.0f / 36.0f;
constexpr float kWa3 = 8.0f / 27.0f, kWb3 = 3.375f;
// B^T of F(4,3) applied to six values in place
__device__ __forceinline__ void wino43_bt(f32x2_t& x0, f32x2_t& x1, f32x2_t& x2, f32x2_t& x3, f32x2_t& x4,
                                          f32x2_t& x5) {
    const f32x2_t e1 = pk_fma(-kWb2, x2, x4), e2 = pk_fma(-kWa2, x2, x4);
    const f32x2_t o1 = pk_fma(-kWb, x1, f32x2_t{kWa, kWa} * x3), o2 = pk_fma(-kWa, x1, f32x2_t{kWb, kWb} * x3);
    const f32x2_t y0 = x0 + pk_fma(-kWs, x2, x4);
    const f32x2_t y5 = x1 + pk_fma(-kWs, x3, x5);
    x1 = e1 + o1;
    x2 = e1 - o1;
    x3 = e2 + o2;
    x4 = e2 - o2;
    x0 = y0;
    x5 = y5;
}
__device__ __forceinline__ f32x4 v4_fma(float k, f32x4 a, f32x4 b) {
    return __builtin_elementwise_fma(f32x4{k, k, k, k}, a, b);
}
// A^T of F(4,3): six values -> four
__device__ __forceinline__ void wino43_at(const f32x4& m0, const f32x4& m1, const f32x4& m2, const f32x4& m3,
                                          const f32x4& m4, const f32x4& m5, f32x4& o0, f32x4& o1, f32x4& o2,
                                          f32x4& o3) {
    const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    o0 = (m0 + s1) + s2;
    o1 = v4_fma(kWb, d2, f32x4{kWa, kWa, kWa, kWa} * d1);
    o2 = v4_fma(kWb2, s2, f32x4{kWa2, kWa2, kWa2, kWa2} * s1);
    o3 = v4_fma(kWb3, d2, f32x4{kWa3, kWa3, kWa3, kWa3} * d1) + m5;
}

// The accumulators live in AGPRs and every MFMA accumulates in place (vDst = SrcC), written as
// inline asm: left to the register allocator, 144 accumulator registers plus the 72 of V wander
// between VGPRs and AGPRs (hundreds of v_accvgpr_write / _read per step, all vector instructions
// that the MFMAs wait for).  The first MFMA of an item takes the inline constant 0 as SrcC.
__device__ __forceinline__ void mfma43_acc(float w, float v, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(v));
}
__device__ __forceinline__ void mfma43_first(float w, float v, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(c) : "v"(w), "v"(v));
}

template <int kForm>      // (0: the 32 x 16 tile; a template so that the header may be included twice)
__global__ void __launch_bounds__(256, 1)
wino43_f32_kernel(const ConvArgs a) {
    using Cfg = Wino43Cfg;
    constexpr int BN = Cfg::BN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* const mailbox = a.counter_base + 1024 + blockIdx.x;      // successor ticket of this workgroup

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tb = wave & 1, cbw = wave >> 1;
    const int t = lane & 15, g = lane >> 4;
    const int nchunks = a.Cin / 8;
    const int in_plane = a.H * a.W * 8;
    const int plane_bytes = in_plane * 4;

    struct Item { int frame, ntile, ty0, tx0; };
    auto decode = [&](int it) {
        const int4 v = a.items[__builtin_amdgcn_readfirstlane(it)];
        return Item{v.x, v.y, v.z, v.w};
    };

    // ---- copy plan of this wave: patch copies j = wave, wave + 4, ... < kPatchInstr, weight copies
    //      j = wave + 4 m < kWInstr (1 KB each) ------------------------------------------------------
    int p_off[Cfg::kPatchPerWave];
    i32x4_t in_rsrc, w_rsrc;
    auto setup_patch = [&](const Item& it) {
#pragma unroll
        for (int k = 0; k < Cfg::kPatchPerWave; ++k) {
            const int j = wave + 4 * k;
            const int s = j * 64 + lane;           // 16-byte slot of the image
            const int cell = s >> 1;
            // invert cell = py * 34 + px + (py >> 2): rows of four share an offset
            int py = cell / Cfg::kPitch;           // first guess, at most one too large
            if (py * Cfg::kPitch + (py >> 2) > cell) --py;
            const int px = cell - py * Cfg::kPitch - (py >> 2);
            const int hf = (s & 1) ^ ((px >> 3) & 1);
            const int gy = it.ty0 - 1 + py, gx = it.tx0 - 1 + px;
            const bool ok = j < Cfg::kPatchInstr && py < Cfg::PH && px >= 0 && px < Cfg::PW &&
                            gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            p_off[k] = ok ? ((gy * a.W + gx) * 8 + hf * 4) * 4 : kOob;
        }
        const float* in_item = a.in + (size_t)it.frame * a.in_frame_stride +
                               (size_t)(a.in_coff / 8) * in_plane;
        in_rsrc = make_rsrc(in_item, (unsigned)(nchunks * plane_bytes));
    };
    auto setup_w = [&](const Item& it) {
        const float* w_item = a.w + (size_t)it.ntile * nchunks * Cfg::kWFloats;
        w_rsrc = make_rsrc(w_item, (unsigned)(nchunks * Cfg::kWFloats * 4));
    };
    float* const sPB = smem;                             // two patch images
    float* const sWB = smem + 2 * Cfg::kPatchFloats;     // two weight images
    int pit = 0, pch = 0, wit = 0, wch = 0;              // copy cursors (item, chunk)
    constexpr int kCopies = Cfg::kPatchPerWave + Cfg::kWPerWave;    // 14 per wave and chunk
    // (scalar operands only: a vector instruction in the issue path would wait for the MFMAs the
    //  weight copies are interleaved with)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    const int w_voff = lane * 16;
    auto copy_n = [&](int n, int img) {
        if (n < Cfg::kPatchPerWave) {            // compile-time
            const int j = wave + 4 * n;          // kPatchInstr is a multiple of 4: every piece is real
            blds16s(in_rsrc, p_off[n], pch * plane_bytes,
                    lds0 + (unsigned)(img * Cfg::kPatchFloats + j * 256) * 4);
        } else {
            const int j = wave + 4 * (n - Cfg::kPatchPerWave);
            blds16s(w_rsrc, w_voff, wch * (Cfg::kWFloats * 4) + j * 1024,
                    lds0 + (unsigned)(2 * Cfg::kPatchFloats + img * Cfg::kWFloats + j * 256) * 4);
        }
    };
    static_assert(Cfg::kPatchInstr % 4 == 0 && Cfg::kWInstr % 4 == 0, "whole pieces per wave");

    int comp_item = blockIdx.x;
    if (comp_item >= a.n_items) return;
    int q0 = a.n_items;                          // successor of comp_item (fetched in its step 0)
    auto advance = [&](int& it, int& ch, bool patch) {
        if (it >= a.n_items) return;
        if (++ch == nchunks) {
            ch = 0;
            it = (it == comp_item) ? q0 : a.n_items;
            if (it < a.n_items) {
                if (patch) setup_patch(decode(it));
                else setup_w(decode(it));
            } else if (patch) {
                in_rsrc[2] = 0;       // nothing left: every copy reads zeros
            } else {
                w_rsrc[2] = 0;
            }
        }
    };

    // lane constants: byte offsets of the lane's 6x6 input tile inside a patch image.  Pixel
    // (4 ty + r, 16 tb + 4 tx + c): the row term and (py >> 2) = ty + (r >> 2) split into a
    // per-lane base and an immediate; the half swap (px >> 3) & 1 is constant over c = 0..3 and
    // over c = 4, 5 of a lane: two bases.
    const int tyl = t >> 2, txl = t & 3;
    const int hsel = g >> 1, sub = g & 1;
    int base_c[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int px = 16 * tb + 4 * txl + 4 * k;          // c = 0 / c = 4
        const int cell = (4 * tyl) * Cfg::kPitch + px + tyl;
        base_c[k] = cell * 32 + ((hsel ^ ((px >> 3) & 1)) * 16) + sub * 8;
    }
    // immediate part (bytes) of tap (r, c) relative to base_c[c >= 4]: tap_off_c below
    const int w_lane = ((g * 2 + cbw) * 16 + t) * 4;       // floats: [xp][g][cb][t][4]

    // ---- prologue: patch(0), W(0) ------------------------------------------------------------------
    pit = wit = comp_item;
    setup_patch(decode(pit));
    setup_w(decode(wit));
#pragma unroll
    for (int n = 0; n < kCopies; ++n) copy_n(n, 0);
    advance(pit, pch, true);
    advance(wit, wch, false);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    // One step: copies of chunk k+1 into images PAR ^ 1, input transform of chunk k from patch
    // image PAR, its 72 MFMAs with the weights of image PAR.
    int k_stamp = 0;
    auto step = [&](auto par, auto first, f32x4 (&acc)[36], int comp_ch) {
        constexpr int PAR = decltype(par)::value;
        constexpr bool FIRST = decltype(first)::value;      // the item's first chunk
        // (a.debug & 32, tools/: s_memtime stamps of one wave's first 12 steps, counter_base[32..])
        const bool stamp = (a.debug & 32) && blockIdx.x == 1 && tid == 0 && k_stamp < 12;
        int* stamps = a.counter_base + 32 + (k_stamp < 12 ? k_stamp : 0) * 6;
        if (stamp) stamps[0] = (int)__builtin_amdgcn_s_memtime();
        if (comp_ch == 0 && tid == 0)
            __hip_atomic_store(mailbox, (int)gridDim.x + atomicAdd(a.counter, 1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // patch(k+1) early (needed at the top of the next step: between the transform's column
        // passes), W(k+1) between the MFMAs below
        if (stamp) stamps[1] = (int)__builtin_amdgcn_s_memtime();
        // ---- V = B^T d B of the lane's tile, two channels (packed) ----
        f32x2_t v[6][6];
        {
            // 36 x ds_read_b64, written as asm: the compiler would pair them into ds_read2_b64, which
            // is served 16 lanes at a time over 32 banks at half the rate (MI355X_MICROARCH.md LDS
            // table) -- the image's swizzle is laid out for ds_read_b64's 32-lane groups over 64 banks
            const unsigned pimg = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(
                sPB + PAR * Cfg::kPatchFloats);
            const unsigned b0 = pimg + base_c[0], b1 = pimg + base_c[1];
            // column c + 1 is read while column c is transformed (lgkmcnt is a 4-bit counter: no
            // more than two columns in flight); patch(k+1) copies go between the column passes
            auto read_col = [&](int c) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    asm volatile("ds_read_b64 %0, %1 offset:%2"
                                 : "=v"(v[r][c])
                                 : "v"(c < 4 ? b0 : b1), "n"(tap_off_c(r, c)));
            };
            read_col(0);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                if (c + 1 < 6) {
                    read_col(c + 1);
                    // the reads are asynchronous: wait for column c (its destinations are operands so
                    // that no use moves above the wait)
                    asm volatile("s_waitcnt lgkmcnt(6)"
                                 : "+v"(v[0][c]), "+v"(v[1][c]), "+v"(v[2][c]), "+v"(v[3][c]), "+v"(v[4][c]),
                                   "+v"(v[5][c]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(v[0][c]), "+v"(v[1][c]), "+v"(v[2][c]), "+v"(v[3][c]), "+v"(v[4][c]),
                                   "+v"(v[5][c]));
                }
                wino43_bt(v[0][c], v[1][c], v[2][c], v[3][c], v[4][c], v[5][c]);
                if (c < Cfg::kPatchPerWave) copy_n(c, PAR ^ 1);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) wino43_bt(v[i][0], v[i][1], v[i][2], v[i][3], v[i][4], v[i][5]);
        }
        if (stamp) stamps[2] = (int)__builtin_amdgcn_s_memtime() + (v[5][5][1] == 12345.f);
        // ---- 36 points x 2 k-steps; weight fragments (one b128 = two points) read two ahead ----
        const float* sW = sWB + PAR * Cfg::kWFloats + w_lane;
        {
        constexpr int kAhead = 4;                 // fragments in flight (LDS latency under the DMA fills)
        f32x4 wq[kAhead + 1];
#pragma unroll
        for (int i = 0; i < kAhead; ++i) wq[i] = *reinterpret_cast<const f32x4*>(sW + i * 512);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xp = 0; xp < 18; ++xp) {
            if (xp + kAhead < 18)
                wq[(xp + kAhead) % (kAhead + 1)] = *reinterpret_cast<const f32x4*>(sW + (xp + kAhead) * 512);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 w = wq[xp % (kAhead + 1)];
            const int x0 = 2 * xp, x1 = 2 * xp + 1;
            if (FIRST) {
                mfma43_first(w[0], v[x0 / 6][x0 % 6][0], acc[x0]);
                mfma43_first(w[2], v[x1 / 6][x1 % 6][0], acc[x1]);
            } else {
                mfma43_acc(w[0], v[x0 / 6][x0 % 6][0], acc[x0]);
                mfma43_acc(w[2], v[x1 / 6][x1 % 6][0], acc[x1]);
            }
            mfma43_acc(w[1], v[x0 / 6][x0 % 6][1], acc[x0]);
            mfma43_acc(w[3], v[x1 / 6][x1 % 6][1], acc[x1]);
            // one weight copy of chunk k+1 per group of four MFMAs: scalar + VMEM issue in the
            // shadow of the matrix pipe
            if (xp < Cfg::kWPerWave) copy_n(Cfg::kPatchPerWave + xp, PAR ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (stamp) stamps[3] = (int)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0);
        if (stamp) stamps[4] = (int)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        if (stamp) stamps[5] = (int)__builtin_amdgcn_s_memtime();
        ++k_stamp;
        // (the store above completed ahead of the barrier: s_waitcnt vmcnt(0); read it at the L2)
        if (comp_ch == 0)
            q0 = __builtin_amdgcn_readfirstlane(
                __hip_atomic_load(mailbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        advance(pit, pch, true);
        advance(wit, wch, false);
    };

    while (comp_item < a.n_items) {
        f32x4 acc[36];
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        step(T0{}, std::true_type{}, acc, 0);
        step(T1{}, std::false_type{}, acc, 1);
        for (int comp_ch = 2; comp_ch < nchunks; comp_ch += 2) {     // Cin / 8 is even
            step(T0{}, std::false_type{}, acc, comp_ch);
            step(T1{}, std::false_type{}, acc, comp_ch + 1);
        }
        // the asm MFMAs are opaque to the compiler's hazard recogniser: their results may be read
        // by vector instructions no earlier than 11 wait states after the last one (the
        // accumulators are operands of the nops so that no reader moves above them)
#pragma unroll
        for (int x = 0; x < 36; x += 12)
            asm volatile("s_nop 15\n\ts_nop 3"
                         : "+a"(acc[x]), "+a"(acc[x + 1]), "+a"(acc[x + 2]), "+a"(acc[x + 3]), "+a"(acc[x + 4]),
                           "+a"(acc[x + 5]), "+a"(acc[x + 6]), "+a"(acc[x + 7]), "+a"(acc[x + 8]),
                           "+a"(acc[x + 9]), "+a"(acc[x + 10]), "+a"(acc[x + 11]));
        float bdot[4][4];      // bottleneck: this wave's partial dot per pixel of the lane's tile
        // ---- epilogue: Y = A^T M A (rows, then columns), batch-norm + ReLU, stores ----------------
        {
            const Item it = decode(comp_item);
            float* out = a.out + (size_t)it.frame * a.out_frame_stride;
            const int out_rows = a.H - a.out_y0;
            const long long plane = (long long)out_rows * a.W * 8;
            const long long pplane = (long long)(a.H >> 1) * (a.W >> 1) * 8;
            const int oy0 = it.ty0 + 4 * tyl, ox0 = it.tx0 + 16 * tb + 4 * txl;
            const int c0 = it.ntile * BN + cbw * 16 + 4 * g;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + c0);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + c0);
            f32x4 z[4][6];        // A^T M: rows
#pragma unroll
            for (int j = 0; j < 6; ++j)
                wino43_at(acc[0 * 6 + j], acc[1 * 6 + j], acc[2 * 6 + j], acc[3 * 6 + j], acc[4 * 6 + j],
                          acc[5 * 6 + j], z[0][j], z[1][j], z[2][j], z[3][j]);
            float* obase = out + (size_t)((a.out_coff + c0) >> 3) * plane + ((a.out_coff + c0) & 7);
            // fused 1x1 bottleneck of a stack's last layer (Cout == 32): per pixel a dot over the lane's
            // four channels, + the lanes l ^ 16, l ^ 32 (the wave's other twelve), + the other channel
            // block's wave through LDS
            f32x4 bw = {0.f, 0.f, 0.f, 0.f};
            if (a.bneck_w) bw = *reinterpret_cast<const f32x4*>(a.bneck_w + c0);
            float* const sDot = smem + Cfg::kPatchFloats + tb * 256;     // [tile t][4 x 4 pixels]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 y[4];
                wino43_at(z[i][0], z[i][1], z[i][2], z[i][3], z[i][4], z[i][5], y[0], y[1], y[2], y[3]);
                const int oy = oy0 + i;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float tv = y[j][k] * sc[k] + sh[k];
                        y[j][k] = a.relu ? fmaxf(tv, 0.0f) : tv;
                    }
                    const int ox = ox0 + j;
                    const bool ok = oy < a.H && ox < a.W && oy >= a.out_y0;
                    if (ok) {
                        if (a.out_nhwc)
                            *reinterpret_cast<f32x4*>(out + ((size_t)(oy - a.out_y0) * a.W + ox) * a.out_ld +
                                                      a.out_coff + c0) = y[j];
                        else
                            *reinterpret_cast<f32x4*>(obase + ((size_t)(oy - a.out_y0) * a.W + ox) * 8) = y[j];
                    }
                    if (a.bneck_w) {
                        float d = ((y[j][0] * bw[0] + y[j][1] * bw[1]) + y[j][2] * bw[2]) + y[j][3] * bw[3];
                        d += __shfl_xor(d, 16, 64);
                        d += __shfl_xor(d, 32, 64);
                        if (cbw == 1 && g == 0) sDot[t * 16 + i * 4 + j] = d;
                        bdot[i][j] = d;
                    }
                }
                // rows i = 0, 1 and 2, 3 hold the windows of a following VALID 2x2 max pool
                if (a.pool_out) {
                    if ((i & 1) == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) z[i][j] = y[j];      // keep the even row
                    } else {
#pragma unroll
                        for (int jw = 0; jw < 2; ++jw) {
                            f32x4 mx;
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                mx[k] = fmaxf(fmaxf(z[i - 1][2 * jw][k], z[i - 1][2 * jw + 1][k]),
                                              fmaxf(y[2 * jw][k], y[2 * jw + 1][k]));
                            const int py = oy0 + i - 1, px = ox0 + 2 * jw;
                            if (py + 1 < a.H && px + 1 < a.W)
                                *reinterpret_cast<f32x4*>(
                                    a.pool_out + (size_t)it.frame * a.pool_frame_stride +
                                    (size_t)(c0 >> 3) * pplane +
                                    ((size_t)(py >> 1) * (a.W >> 1) + (px >> 1)) * 8 + (c0 & 7)) = mx;
                        }
                    }
                }
            }
        }
        if (a.bneck_w) {
            __syncthreads();
            if (cbw == 0 && g == 0) {
                const Item it = decode(comp_item);
                const float* sDot = smem + Cfg::kPatchFloats + tb * 256;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int oy = it.ty0 + 4 * tyl + i, ox = it.tx0 + 16 * tb + 4 * txl + j;
                        if (oy < a.H && ox < a.W && oy >= a.out_y0)
                            a.bneck_out[(size_t)it.frame * a.bneck_frame_stride +
                                        (size_t)(oy - a.out_y0) * a.W + ox] =
                                fmaxf((bdot[i][j] + sDot[t * 16 + i * 4 + j]) * a.bneck_scale + a.bneck_shift, 0.0f);
                    }
            }
            __syncthreads();      // the exchange area is free for the next item
        }
        comp_item = q0;
    }
}

}  // namespace dodt
