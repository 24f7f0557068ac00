// 3x3 convolution / 3x3 stride-2 transposed convolution on the fp32 MFMA of gfx950.
//
// Implicit GEMM, M = output pixels, N = output channels, K = 9 * Cin, computed
// with v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain per output).
//
//   * A workgroup (4 waves, 256 threads) owns a TH x TW spatial tile of one
//     frame and BN output channels.  TH x TW pixels = MTB "M-tiles" of 32 pixels
//     (32/TW rows x TW columns each), stacked vertically.
//   * K is walked in chunks of CK = 8 input channels.  Per chunk the halo'd input
//     patch (TH+2) x (TW+2) x 8 is staged ONCE from NHWC global memory into LDS
//     as [pixel][8 channels + 4 pad] (48-byte pixels: straight 16-byte copies,
//     no transposition), plus the chunk's weights.  Each input element is
//     fetched once per workgroup and then serves 9 taps x BN channels from LDS:
//     the kernel is MFMA-bound, not HBM-bound.
//   * The MFMA's two k-lanes (lane>>5) take channels (s, s+4), s = 0..3, so ONE
//     ds_read_b128 per lane delivers the A operand of four consecutive MFMAs
//     (channels 4h..4h+3 of its pixel); the weights are pre-blocked on the host
//     as [tap][h][n][4] so the B operand is one ds_read_b128 as well.  A tap costs
//     MT + NT LDS reads for 4*MT*NT MFMAs of 64 cycles, and the reads of tap t+1
//     are issued before the MFMAs of tap t (register double buffer).
//   * The global loads of chunk c+1 are issued before the MFMAs of chunk c and
//     land in registers; they are written to LDS after the chunk's last MFMA
//     (T14 "issue early / write late"), so HBM/L2 latency hides under compute.
//   * Epilogue: inference batch-norm (scale, shift) + ReLU on the accumulators,
//     stored NHWC with a caller-given pixel stride / channel offset, so decoder
//     concats are written in place.  For a fixed accumulator register 32 lanes
//     write 32 consecutive channels of one pixel (128 B).
//   * Transposed conv (3x3, stride 2, SAME): M = INPUT pixels; the 9 taps fall
//     into the 4 output-parity classes (4 + 2 + 2 + 1 taps), one accumulator tile
//     per class; out[2i+py][2j+px].
//   * conv3x3_small_cin_kernel handles the first layer (Cin = 6 or 4): planar
//     LDS patch, ds_read_b32 fragments.
#pragma once
#include <hip/hip_runtime.h>

namespace dodt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // a true 4-register value (not the HIP struct)

struct ConvArgs {
    const float* in;      // NHWC, pixel stride in_ld floats, first channel in_coff
    float* out;           // NHWC, pixel stride out_ld floats, first channel out_coff
    const float* w;       // blocked weights, see conv.hip block_weights()
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

constexpr int kCK = 8;        // input channels per K chunk
constexpr int kPixStride = 12;  // floats per LDS pixel: 8 channels + 4 pad (48 B)

template <int TW, int MTB, int WM, int WN, int BN, bool DECONV>
struct ConvCfg {
    static constexpr int kRowsPerMT = 32 / TW;
    static constexpr int TH = MTB * kRowsPerMT;
    static constexpr int PH = DECONV ? TH + 1 : TH + 2;
    static constexpr int PW = DECONV ? TW + 1 : TW + 2;
    static constexpr int MT = MTB / WM;
    static constexpr int NT = BN / 32 / WN;
    static constexpr int kPatchFloats = PH * PW * kPixStride;
    static constexpr int kWFloats = 9 * kCK * BN;
    static constexpr int kLdsBytes = (kPatchFloats + kWFloats) * 4;
    static constexpr int kPatchItems = PH * PW * 2;      // float4 per chunk
    static constexpr int kWItems = kWFloats / 4;         // float4 per chunk
    static constexpr int NP = (kPatchItems + 255) / 256;  // per-thread prefetch regs
    static constexpr int NW = (kWItems + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(MTB % WM == 0 && (BN / 32) % WN == 0, "tile split");
    static_assert(!DECONV || (MT == 1 && NT == 1), "deconv: one M x N tile per wave");
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int TW, int MTB, int WM, int WN, int BN, bool DECONV>
__global__ void __launch_bounds__(256)
conv3x3_mfma_kernel(const ConvArgs a) {
    using Cfg = ConvCfg<TW, MTB, WM, WN, BN, DECONV>;
    constexpr int TH = Cfg::TH, PW = Cfg::PW;
    constexpr int MT = Cfg::MT, NT = Cfg::NT, NP = Cfg::NP, NW = Cfg::NW;
    constexpr int NACC = DECONV ? 4 : MT * NT;
    constexpr int PS = kPixStride;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sP = smem;                      // [PH*PW][12]
    float* sW = smem + Cfg::kPatchFloats;  // [9][2][BN][4]

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
    const int nchunks = a.Cin / kCK;

    const float* in = a.in + (size_t)frame * a.in_frame_stride;
    const f32x4* wblk =
        reinterpret_cast<const f32x4*>(a.w + (size_t)ntile * nchunks * Cfg::kWFloats);

    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;

    // per-thread staging slots: patch item t -> (pixel p, half g).  Offsets are kept
    // in scalar-indexed registers (loops fully unrolled, no address-taken arrays).
    // Loads are unconditional (halo / surplus items read a valid dummy address and
    // are zeroed or skipped at LDS-write time): a load under a branch would make
    // hipcc wait for it on the spot.
    int p_lds[NP];  // float offset in sP, or -1 (no item)
    int p_glb[NP];  // float offset in `in` of the chunk's channel 0 (0 when padded)
    unsigned p_ok = 0;  // bit k: item k reads real data (else zero padding)
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int t = tid + k * 256;
        p_lds[k] = -1;
        p_glb[k] = 0;
        if (t < Cfg::kPatchItems) {
            const int g = t & 1, p = t >> 1;
            const int py = p / PW, px = p - py * PW;
            const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
            p_lds[k] = p * PS + 4 * g;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                p_glb[k] = (gy * a.W + gx) * a.in_ld + a.in_coff + 4 * g;
                p_ok |= 1u << k;
            }
        }
    }
    f32x4 pre_p[NP], pre_w[NW];
#define DODT_ISSUE_LOADS(CH)                                                              \
    {                                                                                     \
        _Pragma("unroll") for (int k = 0; k < NP; ++k)                                    \
            pre_p[k] = *reinterpret_cast<const f32x4*>(in + p_glb[k] + (CH) * kCK);       \
        _Pragma("unroll") for (int k = 0; k < NW; ++k) {                                  \
            const int t = tid + k * 256;                                                  \
            pre_w[k] = wblk[(size_t)(CH) * Cfg::kWItems + min(t, Cfg::kWItems - 1)];     \
        }                                                                                 \
    }
#define DODT_WRITE_LDS()                                                                  \
    {                                                                                     \
        _Pragma("unroll") for (int k = 0; k < NP; ++k)                                    \
            if (p_lds[k] >= 0)                                                            \
                *reinterpret_cast<f32x4*>(sP + p_lds[k]) =                                \
                    ((p_ok >> k) & 1u) ? pre_p[k] : f32x4{0.f, 0.f, 0.f, 0.f};            \
        _Pragma("unroll") for (int k = 0; k < NW; ++k) {                                  \
            const int t = tid + k * 256;                                                  \
            if (t < Cfg::kWItems) reinterpret_cast<f32x4*>(sW)[t] = pre_w[k];            \
        }                                                                                 \
    }

    // lane bases (floats)
    const int a_base =
        ((li / TW + wm * MT * Cfg::kRowsPerMT) * PW + (li % TW)) * PS + 4 * lh;
    const int b_base = (lh * BN + wn * NT * 32 + li) * 4;

    DODT_ISSUE_LOADS(0)
    DODT_WRITE_LDS()
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) DODT_ISSUE_LOADS(ch + 1)  // in flight during the MFMAs below
        if constexpr (!DECONV) {
            f32x4 af[2][MT], bf[2][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                af[0][mt] = *reinterpret_cast<const f32x4*>(
                    sP + a_base + (mt * Cfg::kRowsPerMT * PW) * PS);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bf[0][nt] = *reinterpret_cast<const f32x4*>(sW + b_base + nt * 128);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int cur = tap & 1, nxt = cur ^ 1;
                if (tap + 1 < 9) {
                    const int ky = (tap + 1) / 3, kx = (tap + 1) % 3;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        af[nxt][mt] = *reinterpret_cast<const f32x4*>(
                            sP + a_base + ((mt * Cfg::kRowsPerMT + ky) * PW + kx) * PS);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        bf[nxt][nt] = *reinterpret_cast<const f32x4*>(
                            sW + b_base + (tap + 1) * 2 * BN * 4 + nt * 128);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            acc[mt * NT + nt] =
                                mfma32(af[cur][mt][s], bf[cur][nt][s], acc[mt * NT + nt]);
                        }
                // keep the next tap's LDS reads ahead of this tap's MFMAs
                if (tap + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT * NT, 0);
            }
        } else {
            // patch origin is (ty0-1, tx0-1): in[i][j] sits at patch (r+1, c+1)
            const float* pa = sP + a_base;
            const f32x4 a00 = *reinterpret_cast<const f32x4*>(pa + (PW + 1) * PS);  // in[i][j]
            const f32x4 a10 = *reinterpret_cast<const f32x4*>(pa + 1 * PS);         // in[i-1][j]
            const f32x4 a01 = *reinterpret_cast<const f32x4*>(pa + PW * PS);        // in[i][j-1]
            const f32x4 a11 = *reinterpret_cast<const f32x4*>(pa);                  // in[i-1][j-1]
            f32x4 bw[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                bw[tap] = *reinterpret_cast<const f32x4*>(sW + b_base + tap * 2 * BN * 4);
#define DODT_DECONV_STEP(S)                                   \
            acc[0] = mfma32(a00.S, bw[0].S, acc[0]);          \
            acc[1] = mfma32(a00.S, bw[1].S, acc[1]);          \
            acc[2] = mfma32(a00.S, bw[3].S, acc[2]);          \
            acc[3] = mfma32(a00.S, bw[4].S, acc[3]);          \
            acc[0] = mfma32(a10.S, bw[6].S, acc[0]);          \
            acc[1] = mfma32(a10.S, bw[7].S, acc[1]);          \
            acc[2] = mfma32(a01.S, bw[5].S, acc[2]);          \
            acc[0] = mfma32(a01.S, bw[2].S, acc[0]);          \
            acc[0] = mfma32(a11.S, bw[8].S, acc[0]);
            // taps indexed ky*3+kx; out[2i+ky-2*di][2j+kx-2*dj] += in[i-di][j-dj]*w[ky][kx]
            DODT_DECONV_STEP(x) DODT_DECONV_STEP(y) DODT_DECONV_STEP(z) DODT_DECONV_STEP(w)
#undef DODT_DECONV_STEP
        }
        if (ch + 1 < nchunks) {
            __syncthreads();  // every wave is done reading this chunk
            DODT_WRITE_LDS()
            __syncthreads();
        }
    }

#undef DODT_ISSUE_LOADS
#undef DODT_WRITE_LDS
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

// ---------------------------------------------------------------------------
// First layer (Cin = 6 for BEV, 4 for the padded image): one chunk of CK = Cin
// channels, planar LDS patch [CK][PH*PW (+pad)], weights [9][CK][32], scalar
// ds_read_b32 fragments.  1.5 % of the FLOPs.
// ---------------------------------------------------------------------------
template <int TW, int MTB, int CK>
struct SmallCfg {
    static constexpr int kRowsPerMT = 32 / TW;
    static constexpr int TH = MTB * kRowsPerMT;
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int PS = ((PH * PW + 5) / 8) * 8 + 2;  // == 2 (mod 8): conflict-free stores
    static constexpr int MT = MTB / 4;
    static constexpr int BN = 32;
    static constexpr int kPatchFloats = CK * PS;
    static constexpr int kWFloats = 9 * CK * BN;
    static constexpr int kLdsBytes = (kPatchFloats + kWFloats) * 4;
    static_assert(CK % 2 == 0 && MTB % 4 == 0, "shape");
};

template <int TW, int MTB, int CK>
__global__ void __launch_bounds__(256)
conv3x3_small_cin_kernel(const ConvArgs a) {
    using Cfg = SmallCfg<TW, MTB, CK>;
    constexpr int TH = Cfg::TH, PH = Cfg::PH, PW = Cfg::PW, PS = Cfg::PS, MT = Cfg::MT;
    constexpr int BN = 32;
    constexpr int VEC = (CK % 4 == 0) ? 4 : 2;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sP = smem;
    float* sW = smem + Cfg::kPatchFloats;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wm = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    const int tiles = a.tiles_x * a.tiles_y;
    const int frame = bid / tiles;
    bid -= frame * tiles;
    const int ty0 = (bid / a.tiles_x) * TH, tx0 = (bid % a.tiles_x) * TW;
    const float* in = a.in + (size_t)frame * a.in_frame_stride;

    f32x16 acc[MT];
#pragma unroll
    for (int k = 0; k < MT; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    {
        constexpr int CG = CK / VEC;
        constexpr int ITEMS = PH * PW * CG;
        for (int t = tid; t < ITEMS; t += 256) {
            const int cg = t % CG, p = t / CG;
            const int py = p / PW, px = p - py * PW;
            const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
            float v[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] = 0.0f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const float* src = in + ((size_t)gy * a.W + gx) * a.in_ld + a.in_coff + cg * VEC;
                if constexpr (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(src);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
                    const float2 q = *reinterpret_cast<const float2*>(src);
                    v[0] = q.x; v[1] = q.y;
                }
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) sP[(cg * VEC + k) * PS + p] = v[k];
        }
        const float4* src = reinterpret_cast<const float4*>(a.w);
        for (int t = tid; t < Cfg::kWFloats / 4; t += 256) reinterpret_cast<float4*>(sW)[t] = src[t];
    }
    __syncthreads();
    const int a_base = lh * PS + (li / TW) * PW + (li % TW) + wm * MT * Cfg::kRowsPerMT * PW;
    const int b_base = lh * BN + li;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int cp = 0; cp < CK / 2; ++cp) {
            const float bfv = sW[b_base + (tap * CK + 2 * cp) * BN];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float afv = sP[a_base + 2 * cp * PS + (mt * Cfg::kRowsPerMT + ky) * PW + kx];
                acc[mt] = mfma32(afv, bfv, acc[mt]);
            }
        }
    }
    float* out = a.out + (size_t)frame * a.out_frame_stride;
    const float sc = a.scale[li], sh = a.shift[li];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int y = ty0 + (wm * MT + mt) * Cfg::kRowsPerMT + m / TW;
            const int x = tx0 + m % TW;
            if (y < a.H && x < a.W) {
                float v = acc[mt][r] * sc + sh;
                if (a.relu) v = fmaxf(v, 0.0f);
                out[((size_t)y * a.W + x) * a.out_ld + a.out_coff + li] = v;
            }
        }
}

}  // namespace dodt
