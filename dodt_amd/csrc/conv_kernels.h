// 3x3 convolution / 3x3 stride-2 transposed convolution on the fp32 MFMA of gfx950.
//
// Implicit GEMM, M = output pixels, N = output channels, K = 9 * Cin, computed
// with v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain per output).
//
//   * A workgroup (4 waves, 256 threads) owns a TH x TW spatial tile of one
//     frame and BN output channels.  TH x TW pixels = MTB "M-tiles" of 32 pixels
//     (32/TW rows x TW columns each), stacked vertically.
//   * K is walked in chunks of CK input channels.  Per chunk the halo'd input
//     patch (TH+2) x (TW+2) x CK is staged ONCE from NHWC global memory into LDS
//     in channel-planar form [CK][PH*PW (+pad)], plus the chunk's weights
//     [9][CK][BN].  Each input element is fetched once per workgroup and then
//     serves 9 taps x BN channels from LDS: the kernel is MFMA-bound, not
//     HBM-bound (fp32: 2.25 MFMA clocks per staged float at BN = 32).
//   * A-fragment of tap (ky,kx), channels (c, c+1): lane l reads
//     patch[c + (l>>5)][row(l&31) + ky][col(l&31) + kx]; consecutive lanes are
//     consecutive pixels of a plane -> conflict-free ds_read_b32, and every
//     (tap, channel, M-tile) offset is an immediate added to one lane base.
//     B-fragment: lane l reads w[tap][c + (l>>5)][n0 + (l&31)].
//   * A wave holds MT x NT accumulator tiles (16 VGPRs each); B is reused over MT
//     M-tiles and A over NT N-tiles, so a k-step costs MT+NT LDS reads for MT*NT
//     MFMAs of 64 cycles each.
//   * Epilogue: inference batch-norm (scale, shift) + ReLU on the accumulators,
//     stored NHWC with a caller-given pixel stride / channel offset, so decoder
//     concats are written in place.  For a fixed accumulator register 32 lanes
//     write 32 consecutive channels of one pixel (128 B).
//   * Transposed conv (3x3, stride 2, SAME): M = INPUT pixels; the 9 taps fall
//     into the 4 output-parity classes (4 + 2 + 2 + 1 taps), one accumulator tile
//     per class; out[2i+py][2j+px].
#pragma once
#include <hip/hip_runtime.h>

namespace dodt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* in;      // NHWC, pixel stride in_ld floats, first channel in_coff
    float* out;           // NHWC, pixel stride out_ld floats, first channel out_coff
    const float* w;       // blocked weights [n_tile][chunk][9][CK][BN]
    const float* scale;   // [Cout] batch-norm scale  (rsqrt(var + eps))
    const float* shift;   // [Cout] batch-norm shift  (beta - mean * scale)
    int H, W;             // spatial size of the GEMM's M grid (conv: output = input
                          // size; transposed conv: INPUT size, output is 2H x 2W)
    int Cin, Cout;
    int in_ld, in_coff, out_ld, out_coff;
    long long in_frame_stride, out_frame_stride;  // floats between frames
    int tiles_x, tiles_y;  // tiles per frame
    int relu;
    int out_y0;  // conv only: rows < out_y0 are dropped, row y lands at y - out_y0
};

template <int TW, int MTB, int WM, int WN, int BN, int CK, bool DECONV>
struct ConvCfg {
    static constexpr int kRowsPerMT = 32 / TW;
    static constexpr int TH = MTB * kRowsPerMT;
    static constexpr int HALO_T = 1, HALO_L = 1;
    static constexpr int PH = DECONV ? TH + 1 : TH + 2;
    static constexpr int PW = DECONV ? TW + 1 : TW + 2;
    // plane stride == 2 (mod 8): the 4 channel groups of a staging store land on
    // disjoint bank octets (ds_write_b32, 32 banks)
    static constexpr int PS = ((PH * PW + 5) / 8) * 8 + 2;
    static constexpr int MT = MTB / WM;
    static constexpr int NT = BN / 32 / WN;
    static constexpr int kPatchFloats = CK * PS;
    static constexpr int kWFloats = 9 * CK * BN;
    static constexpr int kLdsBytes = (kPatchFloats + kWFloats) * 4;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(MTB % WM == 0 && (BN / 32) % WN == 0, "tile split");
    static_assert(CK % 2 == 0, "channel pairs");
    static_assert(!DECONV || (MT == 1 && NT == 1), "deconv: one M x N tile per wave");
};

template <int TW, int MTB, int WM, int WN, int BN, int CK, bool DECONV>
__global__ void __launch_bounds__(256)
conv3x3_mfma_kernel(const ConvArgs a) {
    using Cfg = ConvCfg<TW, MTB, WM, WN, BN, CK, DECONV>;
    constexpr int TH = Cfg::TH, PH = Cfg::PH, PW = Cfg::PW, PS = Cfg::PS;
    constexpr int MT = Cfg::MT, NT = Cfg::NT;
    constexpr int VEC = (CK % 4 == 0) ? 4 : 2;
    constexpr int NACC = DECONV ? 4 : MT * NT;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sP = smem;                      // [CK][PS]
    float* sW = smem + Cfg::kPatchFloats;  // [9][CK][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    int bid = blockIdx.x;
    const int tiles = a.tiles_x * a.tiles_y;
    const int frame = bid / tiles;
    bid -= frame * tiles;
    const int ty0 = (bid / a.tiles_x) * TH, tx0 = (bid % a.tiles_x) * TW;
    const int ntile = blockIdx.y;
    const int nchunks = a.Cin / CK;

    const float* in = a.in + (size_t)frame * a.in_frame_stride;
    const float* wblk = a.w + (size_t)ntile * nchunks * Cfg::kWFloats;

    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;

    // lane bases (floats)
    const int a_base = lh * PS + (li / TW) * PW + (li % TW) + wm * MT * Cfg::kRowsPerMT * PW;
    const int b_base = lh * BN + wn * NT * 32 + li;

    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();  // previous chunk fully consumed
        // ---- stage the input patch: NHWC global -> channel-planar LDS -----------
        {
            constexpr int CG = CK / VEC;
            constexpr int ITEMS = PH * PW * CG;
            const int c0 = a.in_coff + ch * CK;
            for (int t = tid; t < ITEMS; t += 256) {
                const int cg = t % CG;
                const int p = t / CG;
                const int py = p / PW, px = p - py * PW;
                const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
                float v[VEC];
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                    const float* src = in + ((size_t)gy * a.W + gx) * a.in_ld + c0 + cg * VEC;
                    if constexpr (VEC == 4) {
                        const float4 q = *reinterpret_cast<const float4*>(src);
                        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                    } else {
                        const float2 q = *reinterpret_cast<const float2*>(src);
                        v[0] = q.x; v[1] = q.y;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) v[k] = 0.0f;
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) sP[(cg * VEC + k) * PS + p] = v[k];
            }
        }
        // ---- stage the weights of this chunk (already blocked, contiguous) ------
        {
            const float4* src = reinterpret_cast<const float4*>(wblk + (size_t)ch * Cfg::kWFloats);
            float4* dst = reinterpret_cast<float4*>(sW);
            for (int t = tid; t < Cfg::kWFloats / 4; t += 256) dst[t] = src[t];
        }
        __syncthreads();
        // ---- MFMA over the chunk ---------------------------------------------------
        if constexpr (!DECONV) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
#pragma unroll
                for (int cp = 0; cp < CK / 2; ++cp) {
                    float bf[NT], af[MT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        bf[nt] = sW[b_base + (tap * CK + 2 * cp) * BN + nt * 32];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        af[mt] = sP[a_base + 2 * cp * PS + (mt * Cfg::kRowsPerMT + ky) * PW + kx];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt * NT + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                af[mt], bf[nt], acc[mt * NT + nt], 0, 0, 0);
                }
            }
        } else {
            // patch origin is (ty0-1, tx0-1): in[i][j] sits at patch (r+1, c+1)
#pragma unroll
            for (int cp = 0; cp < CK / 2; ++cp) {
                const float* pa = sP + a_base + 2 * cp * PS;
                const float a00 = pa[PW + 1];  // in[i  ][j  ]
                const float a10 = pa[1];       // in[i-1][j  ]
                const float a01 = pa[PW];      // in[i  ][j-1]
                const float a11 = pa[0];       // in[i-1][j-1]
                float bw[9];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) bw[tap] = sW[b_base + (tap * CK + 2 * cp) * BN];
                // taps indexed ky*3+kx; out[2i+ky-2*di][2j+kx-2*dj] += in[i-di][j-dj]*w[ky][kx]
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, bw[0], acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, bw[6], acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, bw[2], acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, bw[8], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, bw[1], acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, bw[7], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, bw[3], acc[2], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, bw[5], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, bw[4], acc[3], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: batch-norm + ReLU, NHWC store -------------------------------------
    float* out = a.out + (size_t)frame * a.out_frame_stride;
    if constexpr (!DECONV) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = ntile * BN + (wn * NT + nt) * 32 + li;
            const float sc = a.scale[co], sh = a.shift[co];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int y = ty0 + (wm * MT + mt) * Cfg::kRowsPerMT + m / TW;
                    const int x = tx0 + m % TW;
                    if (y < a.H && x < a.W && y >= a.out_y0) {
                        float v = acc[mt * NT + nt][r] * sc + sh;
                        if (a.relu) v = fmaxf(v, 0.0f);
                        out[((size_t)(y - a.out_y0) * a.W + x) * a.out_ld + a.out_coff + co] = v;
                    }
                }
            }
        }
    } else {
        const int co = ntile * BN + wn * 32 + li;
        const float sc = a.scale[co], sh = a.shift[co];
        const int OW = 2 * a.W;
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int y = ty0 + wm * Cfg::kRowsPerMT + m / TW;
                const int x = tx0 + m % TW;
                if (y < a.H && x < a.W) {
                    float v = acc[cls][r] * sc + sh;
                    if (a.relu) v = fmaxf(v, 0.0f);
                    out[((size_t)(2 * y + py) * OW + (2 * x + px)) * a.out_ld + a.out_coff + co] = v;
                }
            }
        }
    }
}

}  // namespace dodt
