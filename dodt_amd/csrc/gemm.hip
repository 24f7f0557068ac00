// Fully connected layers of the dense heads on the fp32 MFMA of gfx950.
//
// Reference call sites: the RPN anchor predictor (avod/core/models/dt_rpn_model.py:445-537:
// a 3x3 VALID conv = FC 9->256, then 1x1 convs 256->256->{2,6}, slim.conv2d defaults: bias +
// ReLU, none on the last layer) and the stage-2 heads (avod/core/avod_fc_layers/
// fusion_fc_layers.py:136-180 early fusion: mean of the BEV and image crops, flatten,
// three slim.fully_connected 2048 + cls/off/ang outputs; avod/builders/
// avod_corr_layers_builder.py:120-160 the same on the correlation crops -> 3 offsets).
//
// y[M][N] = act(x[M][K] . w[K][N] + b[N]), fp32, v_mfma_f32_32x32x2_f32.
//   * samples are the MFMA's A operand, weights its B operand: in the 32x32 result a lane
//     owns one output feature and 16 samples, so a store instruction writes 32 consecutive
//     features of a sample (128 B).
//   * weights are pre-blocked on the host as [n-tile][K/8][h][BN][4] (the two MFMA k-lanes
//     take k = s and s + 4): one ds_read_b128 per operand feeds four MFMAs, like the conv.
//   * a stage = 32 k; x tile [BM][32 + 4 pad] (conflict-free b128 reads), two LDS buffers;
//     while stage s computes, stage s+1 moves from registers to the other buffer and the
//     loads of stage s+2 are issued, both spread over the MFMAs; fragments of the next
//     k-block are read before the current one's MFMAs (one barrier per stage).
//   * optional second input: x = (x1 + x2) / 2, the heads' "mean" fusion, folded into the
//     load of the first layer (tf.reduce_sum over the two crops / 2.0).
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <vector>

#include <algorithm>

#include "common.h"
#include "lds_dma.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kKAlign = 64;   // K is padded to this (the largest stage depth)
constexpr int kSmallK = 16;   // up to here a layer runs on fc_smallk_kernel

struct GemmArgs {
    const float* x;
    const float* x2;   // may be NULL
    const float* w;    // blocked
    const float* bias; // [Npad]
    float* y;
    int M, K, Kp, N, ldx, ldy, relu;
    const int* d_m;    // may be NULL: rows >= *d_m are skipped
    unsigned long long* clock_probe = nullptr;   // tools/ (DODT_FC_CLOCK=1): shader and 100 MHz clock of one workgroup's K loop
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float pack_bf16(float lo, float hi) {   // round to nearest even
    return __builtin_bit_cast(float, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// the float value of half `which` (0 = low 16 bits) of a packed bf16 pair
__device__ __forceinline__ float bf16_value(float packed, int which) {
    const unsigned u = __builtin_bit_cast(unsigned, packed);
    return __builtin_bit_cast(float, which ? (u & 0xffff0000u) : (u << 16));
}

// kBK = k per stage.  BF16: x is rounded to bf16 on its way into LDS, the weights are stored
// as bf16, one v_mfma_f32_32x32x16_bf16 replaces four fp32 MFMAs; a 16-byte LDS fragment
// holds KB = 8 k-values instead of 4, so the LDS images of a stage have the geometry of an
// fp32 stage of half the depth (G).
template <int BM, int BN, int WM, int WN, bool XVEC, int WBN, int kBK, bool FUSE, bool BF16>
__global__ void __launch_bounds__(256)
fc_mfma_kernel(const GemmArgs a) {
    constexpr int MT = BM / 32 / WM, NT = BN / 32 / WN;
    constexpr int KB = BF16 ? 8 : 4;         // k per 16-byte fragment
    constexpr int G = kBK * 4 / KB;          // floats of LDS per x row and stage
    constexpr int kXS = G + 4;               // LDS floats per x row
    constexpr int XITEMS = BM * (G / 4);     // 16-byte fragments per stage
    constexpr int WITEMS = G * BN / 4;
    constexpr int NX = (XITEMS + 255) / 256, NW = (WITEMS + 255) / 256;
    constexpr int kBuf = BM * kXS + G * BN;  // floats per LDS buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * BM, nt0 = blockIdx.y;
    const int M = a.d_m ? min(*a.d_m, a.M) : a.M;
    if (m0 >= M) return;
    const int nstages = a.Kp / kBK;
    // weights are blocked WBN columns wide; this kernel's BN columns are a slice of them
    static_assert(WBN % BN == 0, "tile must divide the blocked width");
    const f32x4* wblk = reinterpret_cast<const f32x4*>(a.w) +
                        (size_t)(nt0 * BN / WBN) * (a.Kp / (2 * KB)) * 2 * WBN + (nt0 * BN) % WBN;

    // staging slot j < NX: x float4 (tid + 256 j); slot NX + j: weight float4 (tid + 256 j).
    // Loads are unconditional (clamped addresses), zero-fill happens by select.
    constexpr int XV = KB / 4;     // float4 loads per x fragment
    f32x4 prex[NX][XV];            // x slots: raw floats of one fragment
    f32x4 pre[NX + NW];            // weight slots (x slots of this array are unused)
    constexpr int NSLOT = NX + NW;
    // per-slot addressing, hoisted out of the k loop (the loop body must leave the issue
    // slots between MFMAs to the loads and LDS writes, not to integer arithmetic)
    int g_off[NSLOT];   // x: float offset of (row, k-quad) at stage 0; w: float4 offset
    int l_off[NSLOT];   // LDS float offset inside a buffer
    int k_q[NX];        // first k of the slot's quad within a stage
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        if (j < NX) {
            const int t = tid + j * 256;
            const int row = min(t / (G / 4), BM - 1), q = t % (G / 4);
            const int m = min(m0 + row, M - 1);          // clamp: surplus rows are never stored
            g_off[j] = m * a.ldx + q * KB;
            k_q[j < NX ? j : 0] = q * KB;
            l_off[j] = row * kXS + q * 4;
        } else {
            const int t = min(tid + (j - NX) * 256, WITEMS - 1);
            g_off[j] = (t / BN) * WBN + t % BN;
            l_off[j] = BM * kXS + t * 4;
        }
    }
    // loads only ISSUE here; averaging (FUSE) and the zero fill of the K padding happen when
    // the registers are written to LDS a stage later, so no load is waited for on the spot
    f32x4 pre2[FUSE ? NX : 1][XV];
    auto load_slot = [&](int j, int st) {
        if (j < NX) {
            const int jx = j < NX ? j : 0;
            const int kk = st * kBK + k_q[jx];
            if constexpr (XVEC) {
                const int back = max(kk + KB - a.K, 0);   // > 0 only in the padded last stage
#pragma unroll
                for (int i = 0; i < XV; ++i) {
                    prex[jx][i] = *reinterpret_cast<const f32x4*>(a.x + g_off[j] + st * kBK +
                                                                  4 * i - back);
                    if constexpr (FUSE)
                        pre2[jx][i] = *reinterpret_cast<const f32x4*>(a.x2 + g_off[j] + st * kBK +
                                                                      4 * i - back);
                }
            } else {
#pragma unroll
                for (int e = 0; e < KB; ++e) {
                    const int back = max(kk + e + 1 - a.K, 0);
                    prex[jx][e / 4][e % 4] = a.x[g_off[j] + st * kBK + e - back];
                    if constexpr (FUSE)
                        pre2[jx][e / 4][e % 4] = a.x2[g_off[j] + st * kBK + e - back];
                }
            }
        } else {
            pre[j] = wblk[(size_t)st * (G / 4 * WBN) + g_off[j]];
        }
    };
    auto store_slot = [&](int j, int buf, int st) {   // st: the stage the registers hold
        if (j < NX) {
            const int jx = j < NX ? j : 0;
            const int kk = st * kBK + k_q[jx];
            float v[KB];
#pragma unroll
            for (int e = 0; e < KB; ++e) {
                float t = prex[jx][e / 4][e % 4];
                if constexpr (FUSE) t = (t + pre2[jx][e / 4][e % 4]) / 2.0f;
                v[e] = (kk + e < a.K) ? t : 0.0f;
            }
            f32x4 o;
            if constexpr (BF16)
                o = f32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]),
                          pack_bf16(v[4 % KB], v[5 % KB]), pack_bf16(v[6 % KB], v[7 % KB])};
            else
                o = f32x4{v[0], v[1], v[2], v[3]};
            if (XITEMS % 256 == 0 || tid + j * 256 < XITEMS)
                *reinterpret_cast<f32x4*>(smem + buf * kBuf + l_off[j]) = o;
        } else {
            if (WITEMS % 256 == 0 || tid + (j - NX) * 256 < WITEMS)
                *reinterpret_cast<f32x4*>(smem + buf * kBuf + l_off[j]) = pre[j];
        }
    };

    f32x16 acc[MT * NT];
#pragma unroll
    for (int k = 0; k < MT * NT; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;

    // pipeline: stage st computes from LDS buffer st&1; in the shadow of its MFMAs the
    // registers (stage st+1) are written to the other buffer and refilled with stage st+2
    const int last = nstages - 1;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) load_slot(j, 0);
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) store_slot(j, 0, 0);
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) load_slot(j, min(1, last));
    __syncthreads();
    constexpr int NQ = G / 8;
    constexpr int SPQ = (NSLOT + NQ - 1) / NQ;   // staging slots per k-block of 8
    for (int st = 0; st < nstages; ++st) {
        const int buf = st & 1;
        const float* sX = smem + buf * kBuf;
        const float* sW = sX + BM * kXS;
        const int st2 = min(st + 2, last);       // surplus loads re-read the last stage
        f32x4 xf[2][MT], wf[2][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            xf[0][mt] = *reinterpret_cast<const f32x4*>(sX + ((wm * MT + mt) * 32 + li) * kXS + lh * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            wf[0][nt] = *reinterpret_cast<const f32x4*>(sW + (lh * BN + (wn * NT + nt) * 32 + li) * 4);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int cb = q & 1, nb = cb ^ 1;
            if (q + 1 < NQ) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    xf[nb][mt] = *reinterpret_cast<const f32x4*>(
                        sX + ((wm * MT + mt) * 32 + li) * kXS + (q + 1) * 8 + lh * 4);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    wf[nb][nt] = *reinterpret_cast<const f32x4*>(
                        sW + (((q + 1) * 2 + lh) * BN + (wn * NT + nt) * 32 + li) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the next fragments' reads above the MFMAs
            if constexpr (BF16) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt * NT + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, xf[cb][mt]),
                            __builtin_bit_cast(bf16x8, wf[cb][nt]), acc[mt * NT + nt], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt * NT + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                xf[cb][mt][s], wf[cb][nt][s], acc[mt * NT + nt], 0, 0, 0);
            }
#pragma unroll
            for (int j = q * SPQ; j < (q + 1) * SPQ && j < NSLOT; ++j) {
                store_slot(j, buf ^ 1, min(st + 1, last));
                load_slot(j, st2);
            }
            if constexpr (!BF16) {
#pragma unroll
                for (int s = 2; s < 4; ++s)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt * NT + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                xf[cb][mt][s], wf[cb][nt][s], acc[mt * NT + nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();   // buffer buf^1 complete, buffer buf free
    }
    // epilogue: bias + activation; lane = feature, registers = samples
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt0 * BN + (wn * NT + nt) * 32 + li;
        const float b = a.bias[n];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M && n < a.N) {
                    float v = acc[mt * NT + nt][r] + b;
                    if (a.relu) v = fmaxf(v, 0.0f);
                    a.y[(size_t)m * a.ldy + n] = v;
                }
            }
    }
}

template <int BM, int BN, int WM, int WN, int WBN = BN, int kBK = 32, bool BF16 = false>
int launch_fc(hipStream_t s, const GemmArgs& a, int Npad) {
    constexpr int G = BF16 ? kBK / 2 : kBK;
    constexpr size_t lds = 2 * (size_t)(BM * (G + 4) + G * BN) * sizeof(float);
    const bool vec = (a.ldx % 4 == 0) && (a.K % (BF16 ? 8 : 4) == 0);
    dim3 grid(dodt::ceil_div(a.M, BM), Npad / BN);
    auto go = [&](auto kernel) -> hipError_t {
        static std::mutex mu;
        static std::set<const void*> prepared;   // one attribute call per kernel
        {
            std::lock_guard<std::mutex> lock(mu);
            if (!prepared.count(reinterpret_cast<const void*>(kernel))) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)lds);
                if (e != hipSuccess) return e;
                prepared.insert(reinterpret_cast<const void*>(kernel));
            }
        }
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, a);
        return hipSuccess;
    };
    hipError_t e;
    if (vec && a.x2) e = go(&fc_mfma_kernel<BM, BN, WM, WN, true, WBN, kBK, true, BF16>);
    else if (vec) e = go(&fc_mfma_kernel<BM, BN, WM, WN, true, WBN, kBK, false, BF16>);
    else if (a.x2) e = go(&fc_mfma_kernel<BM, BN, WM, WN, false, WBN, kBK, true, BF16>);
    else e = go(&fc_mfma_kernel<BM, BN, WM, WN, false, WBN, kBK, false, BF16>);
    DODT_HIP_CHECK(e);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

// ---- the large layers (stage-2 / correlation FC 2048-wide): LDS-DMA staged form ------------------
// Same arithmetic and weight blocking as fc_mfma_kernel; what changed is how a stage reaches LDS and
// how many waves share a SIMD:
//   * x rows and weight blocks go global -> LDS by buffer_load_dwordx4 ... lds (no staging
//     registers, no ds_write pass, no vector instructions: on this chip a wave's vector work does
//     not overlap the fp32 MFMAs of its SIMD, tools/micro/coissue.hip), three stages of 32 k in a
//     ring, the copies of stage s + 2 issued while stage s computes; rows beyond M and the tail of
//     the weight block read zeros through the descriptor's bound.
//   * the x image is stored unpadded, the 16-byte k-quads of row r XOR-swizzled by (r >> 1) & 7 (the
//     permutation is applied to the source address of each slot): the 16 rows one LDS pass serves
//     fall into 16 different bank groups.
//   * 64 x 64 tiles (one 32x32 MFMA tile per wave): M = 1024, N = 2048 gives 512 workgroups of
//     48 KB (72 KB with the fused mean: the second crop has its own image, the average is taken
//     on the fragment) -- two per CU, two waves per SIMD, one covering the other's waits.
//   * the workgroup -> tile map gives every XCD a contiguous range of tiles with the n-tile
//     running fastest: its workgroups share x rows and sweep K together, so most fills hit its L2
//     (fill rate from L2 ~50 B/clk/CU against ~10.5 from beyond it, tools/micro/fill_rate.hip).
// Needs K % 32 == 0, ldx % 4 == 0 (16-byte aligned rows) and the 128-wide weight blocking.
// (Round 3 tried a FOUR-stage ring, 64 KB and still two workgroups per CU, in which a wave reads the
//  fragments of stage s + 1 under the MFMAs of stage s instead of behind the barrier: 103 TFLOP/s against
//  this form's 109 at M = 1024, N = K = 2048 -- the LDS latency in front of the MFMAs is not what is
//  left; removed again.  Once the copies were interleaved and the fragments read two k-quads ahead (115 TFLOP/s),
//  a four-stage ring with the copies three stages ahead was tried again: 109.)
constexpr int kDmaBM = 64, kDmaBK = 32, kDmaStages = 3;
constexpr int kDmaXFloats = kDmaBM * kDmaBK;

// NT: 32x32 MFMA tiles per wave along N (tile 64 x 64 NT): 2 when M is large enough for 64 x 128 tiles
// to fill two workgroups per CU (both frames' proposals in one launch), else 1.
template <bool FUSE, int NT>
__global__ void __launch_bounds__(256, 2)
fc_dma_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
    constexpr int kDmaBN = 64 * NT, kDmaWFloats = kDmaBK * kDmaBN;
    using dodt::blds16;
    using dodt::i32x4_t;
    using dodt::kOob;
    using dodt::make_rsrc;
    constexpr int kStage = kDmaXFloats * (FUSE ? 2 : 1) + kDmaWFloats;   // floats
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware tile order: workgroup id -> XCD id % 8 (round-robin dispatch); XCD j takes the j-th
    // eighth of the tile list (n fastest)
    const int T = tiles_m * tiles_n;
    int tile = blockIdx.x;
    if (T % 8 == 0) tile = (int)(blockIdx.x % 8) * (T / 8) + (int)(blockIdx.x / 8);
    const int mt = tile / tiles_n, nt0 = tile % tiles_n;
    const int m0 = mt * kDmaBM;
    const int M = a.d_m ? min(*a.d_m, a.M) : a.M;
    if (m0 >= M) return;
    const int nstages = a.K / kDmaBK;

    // ---- copy plan: per stage 8 x pieces (8 rows of 128 B each) [+ 8 of the second input] and
    //      8 weight pieces ((q, h) rows of 64 features); wave w issues pieces w, w + 4 ---------------
    const int rows = min(M - m0, kDmaBM);
    const i32x4_t x_rsrc = make_rsrc(a.x + (size_t)m0 * a.ldx, (unsigned)(((size_t)(rows - 1) * a.ldx + a.K) * 4));
    i32x4_t x2_rsrc = x_rsrc;
    if (FUSE) x2_rsrc = make_rsrc(a.x2 + (size_t)m0 * a.ldx, (unsigned)(((size_t)(rows - 1) * a.ldx + a.K) * 4));
    // weights: block of 128 features [K/8][h][128][4]; this tile's 64 are columns n0 % 128 ..
    const int n0 = nt0 * kDmaBN;
    const float* wblk = a.w + ((size_t)(n0 / 128) * (a.Kp / 8) * 2 * 128 + (n0 % 128)) * 4;
    const i32x4_t w_rsrc = make_rsrc(wblk, (unsigned)(((size_t)(a.Kp / 8) * 2 * 128 - (n0 % 128)) * 16));
    int x_off[2], w_off[2 * NT];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int piece = wave + 4 * k;                  // 0 .. 7
        const int row = piece * 8 + (lane >> 3), cq = (lane & 7) ^ ((row >> 1) & 7);
        x_off[k] = row < rows ? (row * a.ldx + cq * 4) * 4 : kOob;
    }
#pragma unroll
    for (int k = 0; k < 2 * NT; ++k) {
        const int piece = wave + 4 * k;                  // 0 .. 8 NT - 1: (q, h) row and 64-feature half
        w_off[k] = ((piece / NT) * 128 + (piece % NT) * 64 + lane) * 16;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    constexpr int kPerStage = (FUSE ? 4 : 2) + 2 * NT;   // copies per wave and stage
    // copy number n (0 .. kPerStage - 1) of stage st into ring buffer buf: scalar arithmetic only (see blds16s)
    auto issue_one = [&](int st, int buf, int n) {
        const unsigned sX = lds0 + (unsigned)(buf * kStage) * 4;
        constexpr int kX = FUSE ? 4 : 2;
        if (n < kX) {
            const int k = FUSE ? n >> 1 : n;
            const unsigned piece = (unsigned)(wave + 4 * k) * 1024;
            if (FUSE && (n & 1)) dodt::blds16s(x2_rsrc, x_off[k], st * (kDmaBK * 4), sX + kDmaXFloats * 4 + piece);
            else dodt::blds16s(x_rsrc, x_off[k], st * (kDmaBK * 4), sX + piece);
        } else {
            const int k = n - kX;
            dodt::blds16s(w_rsrc, w_off[k], st * (8 * 128 * 16),
                          sX + kDmaXFloats * (FUSE ? 8 : 4) + (unsigned)(wave + 4 * k) * 1024);
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
    // fragment addresses inside a stage: x row (wm * 32 + li), k-quad 2 q + lh, swizzled;
    // weights row (2 q + lh), feature wn * 32 + li
    const int xrow = wm * 32 + li;
    const int xsw = (xrow >> 1) & 7;
    const int x_base = xrow * kDmaBK;
    const int w_base = kDmaXFloats * (FUSE ? 2 : 1) + (lh * kDmaBN + wn * 32 * NT + li) * 4;

    // per-lane fragment offsets inside a stage (floats): with the buffer number a compile-time
    // constant (ring walked by an unrolled-by-three loop) every LDS read is base + immediate
    int xo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xo[q] = x_base + (((2 * q + lh) ^ xsw) * 4);
    auto stage = [&](auto bufc, int st) {
        constexpr int BUF = decltype(bufc)::value;
        // stage st has landed once all but the copies of stage st + 1 are done
        if (st + 1 < nstages) __builtin_amdgcn_s_waitcnt(0x0f70 | kPerStage);   // vmcnt(kPerStage)
        else __builtin_amdgcn_s_waitcnt(0x0f70);                                  // vmcnt(0)
        __builtin_amdgcn_s_barrier();     // ... for every wave; buffer (BUF + 2) % 3 is free
        const bool more = st + 2 < nstages;
        const float* sS = smem + BUF * kStage;
        f32x4 xf[4], wf[4][NT], x2[FUSE ? 4 : 1];
        auto read_q = [&](int q) {
            xf[q] = *reinterpret_cast<const f32x4*>(sS + xo[q]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wf[q][nt] = *reinterpret_cast<const f32x4*>(sS + w_base + q * (2 * kDmaBN * 4) + nt * 128);
            if (FUSE) x2[q] = *reinterpret_cast<const f32x4*>(sS + kDmaXFloats + xo[q]);
        };
        // fragments two k-quads ahead of the MFMAs that use them: only the first pair's LDS latency stands in
        // front of the stage's first MFMA; the copies of stage st + 2 go out one per group of four MFMAs, in the
        // matrix pipe's shadow (issued in one burst behind the barrier they held the wave's MFMAs back)
        read_q(0);
        read_q(1);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int kPerQ = (kPerStage + 3) / 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q + 2 < 4) read_q(q + 2);
            if (FUSE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) xf[q][e] = (xf[q][e] + x2[q][e]) / 2.0f;
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[q][s2], wf[q][nt][s2], acc[nt], 0, 0, 0);
            if (more) {
#pragma unroll
                for (int n = q * kPerQ; n < (q + 1) * kPerQ && n < kPerStage; ++n) issue_one(st + 2, (BUF + 2) % kDmaStages, n);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto issue = [&](int st, int buf) {
#pragma unroll
        for (int n = 0; n < kPerStage; ++n) issue_one(st, buf, n);
    };
    issue(0, 0);
    if (nstages > 1) issue(1, 1);
    const bool probe = a.clock_probe != nullptr && blockIdx.x == 17 && tid == 0;
    unsigned long long c0 = 0, r0 = 0;
    if (probe) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int st = 0; st < nstages; st += 3) {
        stage(std::integral_constant<int, 0>{}, st);
        if (st + 1 < nstages) stage(std::integral_constant<int, 1>{}, st + 1);
        if (st + 2 < nstages) stage(std::integral_constant<int, 2>{}, st + 2);
    }
    if (probe) {
        a.clock_probe[0] = __builtin_amdgcn_s_memtime() - c0;
        a.clock_probe[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    // epilogue: bias + activation; lane = feature, registers = samples
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn * 32 * NT + nt * 32 + li;
        const float b = a.bias[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < M && n < a.N) {
                float v = acc[nt][r] + b;
                if (a.relu) v = fmaxf(v, 0.0f);
                a.y[(size_t)m * a.ldy + n] = v;
            }
        }
    }
}

int launch_fc_dma(hipStream_t s, const GemmArgs& a_in, int Npad) {
    GemmArgs a = a_in;
    // DODT_FC_CLOCK=1 (tools/gemm_one.py): shader clock inside the K loop of workgroup 17, from s_memtime against the
    // 100 MHz s_memrealtime -- the clock the fp32 matrix pipe actually runs at under this kernel
    static const bool want_clock = getenv("DODT_FC_CLOCK") != nullptr;
    static unsigned long long* d_probe = nullptr;
    if (want_clock) {
        if (!d_probe) DODT_HIP_CHECK(hipMalloc(&d_probe, 2 * sizeof(unsigned long long)));
        a.clock_probe = d_probe;
    }
    // 64 x 128 tiles (two MFMA tiles per wave) when that still gives two workgroups per CU
    static const int force_nt = getenv("DODT_FC_NT") ? atoi(getenv("DODT_FC_NT")) : 0;
    const int nt = force_nt ? force_nt
                            : (!a.x2 && (long)dodt::ceil_div(a.M, kDmaBM) * (Npad / 128) >= 2 * 256 ? 2 : 1);
    const int tiles_m = dodt::ceil_div(a.M, kDmaBM), tiles_n = Npad / (64 * nt);
    auto go = [&](auto kernel, size_t lds) -> hipError_t {
        static std::mutex mu;
        static std::set<const void*> prepared;
        {
            std::lock_guard<std::mutex> lock(mu);
            if (!prepared.count(reinterpret_cast<const void*>(kernel))) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                prepared.insert(reinterpret_cast<const void*>(kernel));
            }
        }
        hipLaunchKernelGGL(kernel, dim3(tiles_m * tiles_n), dim3(256), lds, s, a, tiles_m, tiles_n);
        return hipSuccess;
    };
    hipError_t e;
    if (a.x2) e = go(&fc_dma_kernel<true, 1>, (size_t)kDmaStages * (2 * kDmaXFloats + kDmaBK * 64) * 4);
    else if (nt == 2) e = go(&fc_dma_kernel<false, 2>, (size_t)kDmaStages * (kDmaXFloats + kDmaBK * 128) * 4);
    else e = go(&fc_dma_kernel<false, 1>, (size_t)kDmaStages * (kDmaXFloats + kDmaBK * 64) * 4);
    DODT_HIP_CHECK(e);
    DODT_LAUNCH_CHECK();
    if (want_clock) {
        unsigned long long h[2] = {0, 0};
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, d_probe, sizeof(h), hipMemcpyDeviceToHost);
        if (h[1])
            fprintf(stderr, "[dodt] fc_dma_kernel M %d K %d N %d: K loop of workgroup 17: %llu shader cycles in %.2f us = %.3f GHz\n",
                    a.M, a.K, a.N, h[0], h[1] / 100.0, (double)h[0] / h[1] * 0.1);
    }
    return DODT_OK;
}

// ---- bf16 heads, round 4: bf16 activations in HBM, both operands by LDS-DMA ---------------------------------
// fc_mfma_kernel's bf16 instantiation reads float32 x, rounds it on its way into LDS (a register-staged ds_write
// pass with the conversions on the vector ALU) and prefetches two 32-k stages ahead: at the heads' M = 1024 a stage is
// ~130 matrix cycles per SIMD against > 1000 cycles of load latency, so the K loop waits for memory: 32 us at
// N = K = 2048 with an intercept of ~25 us per launch (M = 1024 / 2048 / 4096: 32 / 38 / 60 us).  Here the hidden
// activations of a head stay bf16 in HBM -- the layer that produces them rounds them (to nearest even) in its
// epilogue, which is the same rounding the next layer's load applied, so the arithmetic of oracle/heads.py's bf16
// restatement is unchanged -- and a stage (64 k: x 64 rows x 128 B, weights 128 columns x 128 B, 24 KB) travels
// global -> LDS by buffer_load_dwordx4 ... lds with scalar operands, three stages in a ring, two workgroups per CU
// (six stages in flight per CU ~ what the L2 -> LDS fill rate of ~50 B/clk/CU takes).  Both images are unpadded
// with the 16-byte k-slots of row / column r XOR-swizzled by (r >> 1) & 7 (x: on the source address; weights:
// pre-swizzled on the host, a stage image is one contiguous 16 KB block): conflict-free ds_read_b128 fragments.
// 64 x 128 tiles, wave = 32 x 64 (one x fragment, two weight fragments, two v_mfma_f32_32x32x16_bf16 per 16 k).
// Needs x rows of >= Kd = ceil(K / 64) 64 bf16 whose tail beyond K is zero, 16-byte aligned, N % 128 == 0.
// BK = k per stage: 64 (24 KB stages, 72 KB: the fastest layer alone) or 32 (12 KB stages, 36 KB: fits into the
// 41 KB two resident bf16 conv workgroups leave of a CU's LDS, like fc_mfma_kernel's 32-k form did).  A row /
// column of a stage image is BK / 8 slots of 16 bytes; slot s of row r sits in slot s ^ ((r / (16 / S)) & (S - 1)),
// S = BK / 8: 16 consecutive rows then cover the 16 bank columns for every s.
constexpr int kBfBM = 64, kBfBN = 128, kBfStages = 3;
__host__ __device__ constexpr int bf_swz(int r, int S) { return (r / (16 / S)) & (S - 1); }

template <bool YBF16, int BK, int R = kBfStages>
__global__ void __launch_bounds__(256, 2)
fc_bf16_dma_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
    // R: ring of stage images (three; six with DODT_FC_BF16_DMA_RING=6 and BK = 32: at M = 1024 a layer is 256 workgroups,
    // one per CU, and two 12 KB stages on their way per CU are less than the memory system's latency x bandwidth -- see
    // launch_fc_bf16_dma for what that bought)
    static_assert((R - 2) * ((kBfBM + kBfBN) * BK * 2 / 1024 / 4) <= 63, "vmcnt is six bits");
    using dodt::i32x4_t;
    using dodt::kOob;
    using dodt::make_rsrc;
    constexpr int S = BK / 8;                                  // 16-byte slots per row of an image
    constexpr int kXBytes = kBfBM * BK * 2, kWBytes = kBfBN * BK * 2, kStage = kXBytes + kWBytes;
    constexpr int kXPieces = kXBytes / 1024 / 4, kWPieces = kWBytes / 1024 / 4;   // 1 KB copies per wave and stage
    constexpr int kPerStage = kXPieces + kWPieces;
    constexpr int kRowsPerPiece = 1024 / (BK * 2);
    constexpr int NQ = BK / 16;                                // MFMA k-steps per stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware tile order (as fc_dma_kernel): XCD j takes the j-th eighth of the tile list, n fastest
    const int T = tiles_m * tiles_n;
    int tile = blockIdx.x;
    if (T % 8 == 0) tile = (int)(blockIdx.x % 8) * (T / 8) + (int)(blockIdx.x / 8);
    const int mt = tile / tiles_n, nt0 = tile % tiles_n;
    const int m0 = mt * kBfBM;
    const int M = a.d_m ? min(*a.d_m, a.M) : a.M;
    if (m0 >= M) return;
    const int nstages = a.Kp / BK;          // (Kp: K rounded up to 64 for this kernel)
    const int rows = min(M - m0, kBfBM);
    const unsigned short* x16 = reinterpret_cast<const unsigned short*>(a.x);
    const i32x4_t x_rsrc = make_rsrc(x16 + (size_t)m0 * a.ldx, (unsigned)(((size_t)(rows - 1) * a.ldx + a.Kp) * 2));
    const char* wblk = reinterpret_cast<const char*>(a.w) + (size_t)nt0 * nstages * kWBytes;
    const i32x4_t w_rsrc = make_rsrc(wblk, (unsigned)((size_t)nstages * kWBytes));
    // copy plan: per stage the x image (rows of BK bf16) and the weight image (pre-swizzled, contiguous) in 1 KB
    // pieces; wave w issues pieces w, w + 4, ...
    int x_off[kXPieces];
#pragma unroll
    for (int k = 0; k < kXPieces; ++k) {
        const int row = (wave + 4 * k) * kRowsPerPiece + lane / S, slot = (lane % S) ^ bf_swz(row, S);
        x_off[k] = row < rows ? (row * a.ldx + slot * 8) * 2 : kOob;
    }
    const int w_off = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    auto issue_one = [&](int st, int buf, int n) {
        const unsigned sX = lds0 + (unsigned)(buf * kStage);
        if (n < kXPieces) dodt::blds16s(x_rsrc, x_off[n < kXPieces ? n : 0], st * (BK * 2), sX + (unsigned)(wave + 4 * n) * 1024);
        else dodt::blds16s(w_rsrc, w_off, st * kWBytes + (wave + 4 * (n - kXPieces)) * 1024,
                           sX + kXBytes + (unsigned)(wave + 4 * (n - kXPieces)) * 1024);
    };
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
    // fragment addresses inside a stage (bytes): x row wm 32 + li, weight columns wn 64 + nt 32 + li; both swizzles
    // depend on li only (the bases are multiples of 32), so the k-steps share their slot offsets
    const int sw = bf_swz(li, S);
    int so[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) so[q] = ((2 * q + lh) ^ sw) * 16;
    const int x_base = (wm * 32 + li) * (BK * 2);
    const int w_base = kXBytes + (wn * 64 + li) * (BK * 2);
    auto stage = [&](auto bufc, int st) {
        constexpr int BUF = decltype(bufc)::value;
        // stage st has landed once all but the copies of stages st + 1 .. st + R - 2 are done (fewer near the end: drain)
        if (st + R - 2 < nstages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * kPerStage) : "memory");
        else __builtin_amdgcn_s_waitcnt(0x0f70);                                  // vmcnt(0)
        __builtin_amdgcn_s_barrier();     // ... for every wave; buffer (BUF + R - 1) % R is free
        const bool more = st + R - 1 < nstages;
        const char* sS = reinterpret_cast<const char*>(smem) + BUF * kStage;
        f32x4 xf[NQ], wf[NQ][2];
        auto read_q = [&](int q) {
            xf[q] = *reinterpret_cast<const f32x4*>(sS + x_base + so[q]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                wf[q][nt] = *reinterpret_cast<const f32x4*>(sS + w_base + nt * 32 * (BK * 2) + so[q]);
        };
        read_q(0);
        read_q(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + 2 < NQ) read_q(q + 2);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xf[q]),
                                                                  __builtin_bit_cast(bf16x8, wf[q][nt]), acc[nt], 0, 0, 0);
            if (more) {
                // the copies of stage st + 2 spread over the k-steps
                constexpr int kPerQ = (kPerStage + NQ - 1) / NQ;
#pragma unroll
                for (int n = q * kPerQ; n < (q + 1) * kPerQ && n < kPerStage; ++n) issue_one(st + R - 1, (BUF + R - 1) % R, n);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto issue = [&](int st, int buf) {
#pragma unroll
        for (int n = 0; n < kPerStage; ++n) issue_one(st, buf, n);
    };
#pragma unroll
    for (int b = 0; b < R - 1; ++b)
        if (b < nstages) issue(b, b);
    for (int st = 0; st < nstages; st += R) {
        stage(std::integral_constant<int, 0>{}, st);
        if (st + 1 < nstages) stage(std::integral_constant<int, 1>{}, st + 1);
        if (st + 2 < nstages) stage(std::integral_constant<int, 2>{}, st + 2);
        if constexpr (R > 3) {
            if (st + 3 < nstages) stage(std::integral_constant<int, 3 % R>{}, st + 3);
            if (st + 4 < nstages) stage(std::integral_constant<int, 4 % R>{}, st + 4);
            if (st + 5 < nstages) stage(std::integral_constant<int, 5 % R>{}, st + 5);
        }
        static_assert(R == 3 || R == 6, "the stage loop is unrolled for rings of three or six");
    }
    // epilogue: bias + activation; lane = feature, registers = samples; bf16 output: round to nearest even
    const int n0 = nt0 * kBfBN;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + wn * 64 + nt * 32 + li;
        const float b = a.bias[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < M && n < a.N) {
                float v = acc[nt][r] + b;
                if (a.relu) v = fmaxf(v, 0.0f);
                if constexpr (YBF16)
                    reinterpret_cast<unsigned short*>(a.y)[(size_t)m * a.ldy + n] =
                        (unsigned short)(__builtin_bit_cast(unsigned, pack_bf16(v, 0.0f)) & 0xffffu);
                else
                    a.y[(size_t)m * a.ldy + n] = v;
            }
        }
    }
}

int launch_fc_bf16_dma(hipStream_t s, const GemmArgs& a, int Npad, bool y_bf16, int bk) {
    const int tiles_m = dodt::ceil_div(a.M, kBfBM), tiles_n = Npad / kBfBN;
    // (DODT_FC_BF16_DMA_RING=6: six stage images instead of three -- measured at the end of round 4: 16.7 -> 14.8 us at
    //  M = 1024 (one workgroup per CU), 27.8 -> 29.5 / 52.6 -> 59.2 us at M = 2048 / 4096 (two per CU), and 980-992 against
    //  995-1 012 pairs/s in the bf16 pipeline: opt-in)
    static const int ring = getenv("DODT_FC_BF16_DMA_RING") && atoi(getenv("DODT_FC_BF16_DMA_RING")) == 6 ? 6 : 3;
    const int stages = bk == 64 ? kBfStages : ring;
    const size_t lds = (size_t)stages * (kBfBM + kBfBN) * bk * 2;
    auto go = [&](auto kernel) -> hipError_t {
        static std::mutex mu;
        static std::set<const void*> prepared;
        {
            std::lock_guard<std::mutex> lock(mu);
            if (!prepared.count(reinterpret_cast<const void*>(kernel))) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                prepared.insert(reinterpret_cast<const void*>(kernel));
            }
        }
        hipLaunchKernelGGL(kernel, dim3(tiles_m * tiles_n), dim3(256), lds, s, a, tiles_m, tiles_n);
        return hipSuccess;
    };
    hipError_t e;
    if (bk == 64) e = y_bf16 ? go(&fc_bf16_dma_kernel<true, 64>) : go(&fc_bf16_dma_kernel<false, 64>);
    else if (stages == 6) e = y_bf16 ? go(&fc_bf16_dma_kernel<true, 32, 6>) : go(&fc_bf16_dma_kernel<false, 32, 6>);
    else e = y_bf16 ? go(&fc_bf16_dma_kernel<true, 32>) : go(&fc_bf16_dma_kernel<false, 32>);
    DODT_HIP_CHECK(e);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

// stage depth of fc_bf16_dma_kernel in this process (DODT_FC_BF16_DMA_BK = 32 | 64; the weights are blocked for it)
int bf16_dma_bk() {
    static const int bk = getenv("DODT_FC_BF16_DMA_BK") && atoi(getenv("DODT_FC_BF16_DMA_BK")) == 64 ? 64 : 32;
    return bk;
}

// ---- the skinny layers (N <= 32: the heads' output layers, 2048 -> 2 / 10 / 2 / 3, and the RPN's
//      256 -> 2 / 6) ---------------------------------------------------------------------------------
// On the tiled kernels such a layer is ceil(M / 128) workgroups that each walk all of K: 8 workgroups
// and 49 us at M = 1024, K = 2048 -- latency, not work (the layer reads 8 MB).  Here a workgroup is 16
// samples x all 32 columns with K SPLIT over its eight waves (wave w takes the 16-k steps w, w + 8, ...: the
// eight waves read 512 contiguous bytes of a sample row), operands straight from global memory four steps
// ahead, v_mfma_f32_16x16x4_f32 with the samples as A operand; the eight partial tiles are added in
// wave order through LDS.  The columns may go to up to three dense arrays (`SplitOut`): the output layers
// of a head -- cls | offsets | angle vectors -- are then ONE layer with concatenated weights and one launch.
struct SplitOut {
    float* y[3];
    int n_end[3];     // columns [n_end[p-1], n_end[p]) go to y[p]
    int ld[3];
};

__device__ __forceinline__ f32x4 mfma16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// (BF16: the heads' bf16 mode -- x rounded to bf16 on load, weights stored as bf16 [K/16][h][32][8], k = 16q + 8h + s;
//  one v_mfma_f32_16x16x16_bf16 per step and column block, k-slot g = k 4g .. 4g+3 of the step)
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// (XB16: x is already bf16 in memory -- the bf16 heads' hidden activations --, ldx in elements)
template <bool BF16, bool XB16 = false>
__global__ void __launch_bounds__(512)
fc_skinny_kernel(const GemmArgs a, const SplitOut so) {
    static_assert(BF16 || !XB16, "bf16 rows feed the bf16 arithmetic only");
    __shared__ f32x4 s_part[8][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int M = a.d_m ? min(*a.d_m, a.M) : a.M;
    const int m0 = blockIdx.x * 16;
    if (m0 >= M) return;
    // A operand: lane (g, i) holds sample m0 + i at k-slot g; a 16-byte load covers the slot's four k of a step
    const float* xr = a.x + (size_t)min(m0 + i, M - 1) * a.ldx + 4 * g;
    const unsigned short* xr16 = reinterpret_cast<const unsigned short*>(a.x) + (size_t)min(m0 + i, M - 1) * a.ldx + 4 * g;
    const int steps = a.K / 16;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    if constexpr (BF16) {
        // B operand: lane (g, n): column n (and n + 16), 8 bytes = k 4g .. 4g+3 of the step
        const f32x2* w2 = reinterpret_cast<const f32x2*>(a.w) + ((g >> 1) * 32 + i) * 2 + (g & 1);
        for (int j0 = wave; j0 < steps; j0 += 8 * U) {
            f32x4 xv[U];
            f32x2 wa[U], wb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 8 * u;
                const bool ok = j < steps;
                const int jj = ok ? j : wave;
                if constexpr (XB16) {      // 8 bytes = the slot's four k, already rounded
                    const f32x2 x2 = *reinterpret_cast<const f32x2*>(xr16 + 16 * jj);
                    xv[u] = f32x4{x2[0], x2[1], 0.f, 0.f};
                } else {
                    xv[u] = *reinterpret_cast<const f32x4*>(xr + 16 * jj);
                }
                const f32x2 z = {0.f, 0.f};
                wa[u] = ok ? w2[(size_t)jj * 128] : z;           // a step = 2 h-planes of 32 x 16 bytes
                wb[u] = ok ? w2[(size_t)jj * 128 + 32] : z;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f32x2 xb = XB16 ? f32x2{xv[u][0], xv[u][1]}
                                      : f32x2{pack_bf16(xv[u][0], xv[u][1]), pack_bf16(xv[u][2], xv[u][3])};
                acc0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, xb),
                                                                 __builtin_bit_cast(s16x4, wa[u]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, xb),
                                                                 __builtin_bit_cast(s16x4, wb[u]), acc1, 0, 0, 0);
            }
        }
    } else {
        // B operand: lane (g, n) holds column n (and n + 16) at k-slot g: blocked [K/8][h][32][4], k = 8q + 4h + s
        const f32x4* w4 = reinterpret_cast<const f32x4*>(a.w) + g * 32 + i;
        for (int j0 = wave; j0 < steps; j0 += 8 * U) {
            f32x4 xv[U], wa[U], wb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 8 * u;
                const bool ok = j < steps;
                const int jj = ok ? j : wave;
                xv[u] = *reinterpret_cast<const f32x4*>(xr + 16 * jj);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                wa[u] = ok ? w4[(size_t)jj * 128] : z;          // a step = 2 q = 4 (q, h) planes of 32 float4
                wb[u] = ok ? w4[(size_t)jj * 128 + 16] : z;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc0 = mfma16x4(xv[u][s], wa[u][s], acc0);
                    acc1 = mfma16x4(xv[u][s], wb[u][s], acc1);
                }
        }
    }
    s_part[wave][0][lane] = acc0;
    s_part[wave][1][lane] = acc1;
    __syncthreads();
    if (tid >= 128) return;
    const int nb = tid >> 6;
    f32x4 sum = s_part[0][nb][lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) sum += s_part[w][nb][lane];
    const int n = nb * 16 + i;                  // lane (g, i): samples m0 + 4 g + r, column n
    if (n >= a.N) return;
    const float b = a.bias[n];
    const int p = (n >= so.n_end[0]) + (n >= so.n_end[1]);
    float* y = p == 0 ? so.y[0] : (p == 1 ? so.y[1] : so.y[2]);
    const int ld = p == 0 ? so.ld[0] : (p == 1 ? so.ld[1] : so.ld[2]);
    const int nn = n - (p == 0 ? 0 : (p == 1 ? so.n_end[0] : so.n_end[1]));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * g + r;
        if (m < M) {
            float v = sum[r] + b;
            if (a.relu) v = fmaxf(v, 0.0f);
            y[(size_t)m * ld + nn] = v;
        }
    }
}

// ---- layers with a tiny K (the RPN's 3x3x1 "conv" = FC 9 -> 512 on 5 500 anchors) --------------------------------
// On the tiled kernel K is padded to a 64-deep stage (seven eighths of the MFMAs multiply zeros) and the register-
// staged scalar loads keep it at 35 us; the layer is 51 MFLOP and an 11 MB write.  Here a thread owns four
// consecutive columns of one sample: K multiply-adds each on the vector ALU (float32, k ascending, no fusion), weights
// [K][N] read as float4 (L1 hits), 16-byte stores along the row.  BF16: the sample (after the mean of the two inputs)
// and the weights are rounded to bf16 like the tiled bf16 kernel's, products and sums in float32.
template <bool BF16>
__global__ void __launch_bounds__(256)
fc_smallk_kernel(const GemmArgs a, const float* __restrict__ w_plain) {
    const int M = a.d_m ? min(*a.d_m, a.M) : a.M;
    const int n4 = a.N >> 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int m = (int)(t / n4), c = (int)(t - (long long)m * n4) * 4;
    if (m >= M) return;
    const float* x = a.x + (size_t)m * a.ldx;
    const float* x2 = a.x2 ? a.x2 + (size_t)m * a.ldx : nullptr;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < a.K; ++k) {
        float xv = x[k];
        if (x2) xv = (xv + x2[k]) / 2.0f;
        if constexpr (BF16) xv = bf16_value(pack_bf16(xv, 0.0f), 0);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w_plain + (size_t)k * a.N + c);
        acc += wv * xv;
    }
    const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + c);
    acc += b;
    if (a.relu) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaxf(acc[i], 0.0f);
    }
    float* y = a.y + (size_t)m * a.ldy + c;
    if ((a.ldy & 3) == 0) *reinterpret_cast<f32x4*>(y) = acc;
    else {
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = acc[i];
    }
}

// the conditions of fc_skinny_kernel (weights blocked for BN = 32, no second input)
bool skinny_ok(int Npad, int K, const GemmArgs& a) {
    static const bool on = !(getenv("DODT_FC_SKINNY") && atoi(getenv("DODT_FC_SKINNY")) == 0);
    return on && Npad == 32 && K % 16 == 0 && a.ldx % 4 == 0 && (size_t)a.x % 16 == 0 && !a.x2;
}

int launch_fc_skinny(hipStream_t s, const GemmArgs& a, const SplitOut& so, bool bf16, bool x_bf16 = false) {
    if (x_bf16) hipLaunchKernelGGL((fc_skinny_kernel<true, true>), dim3(dodt::ceil_div(a.M, 16)), dim3(512), 0, s, a, so);
    else if (bf16) hipLaunchKernelGGL(fc_skinny_kernel<true>, dim3(dodt::ceil_div(a.M, 16)), dim3(512), 0, s, a, so);
    else hipLaunchKernelGGL(fc_skinny_kernel<false>, dim3(dodt::ceil_div(a.M, 16)), dim3(512), 0, s, a, so);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

}  // namespace

// (a + b) / 2 of two row-major blocks, rows limited by *d_n: the heads' "mean" fusion of BEV and image
// crops as its own pass (6.4 MB each at 1024 x 7 x 7 x 32), so that the GEMM behind it is the plain one
namespace {
__global__ void __launch_bounds__(256)
mean_fusion_kernel(const float4* __restrict__ a, const float4* __restrict__ b, int rows, const int* __restrict__ d_n,
                   int row4, float4* __restrict__ out) {
    const int lim = d_n ? min(*d_n, rows) : rows;
    const long long total = (long long)lim * row4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const float4 x = a[i], y = b[i];
        out[i] = make_float4((x.x + y.x) / 2.0f, (x.y + y.y) / 2.0f, (x.z + y.z) / 2.0f, (x.w + y.w) / 2.0f);
    }
}
}  // namespace

extern "C" int dodt_mean_fusion(dodt_ctx* ctx, const float* d_a, const float* d_b, int rows, const int32_t* d_n,
                                int row_floats, float* d_out) {
    DODT_REQUIRE(ctx && d_a && d_b && d_out, "dodt_mean_fusion: NULL argument");
    DODT_REQUIRE(rows >= 0 && row_floats >= 4 && row_floats % 4 == 0,
                 "dodt_mean_fusion: rows of a multiple of four floats, got %d", row_floats);
    DODT_REQUIRE(((size_t)d_a | (size_t)d_b | (size_t)d_out) % 16 == 0, "dodt_mean_fusion: 16-byte aligned blocks");
    if (rows == 0) return DODT_OK;
    const long long total = (long long)rows * (row_floats / 4);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(mean_fusion_kernel, dim3(blocks), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float4*>(d_a), reinterpret_cast<const float4*>(d_b), rows, d_n,
                       row_floats / 4, reinterpret_cast<float4*>(d_out));
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

// The input rows of a bf16 head: out[r][0 .. row_floats) = bf16((a[r] + b[r]) / 2) (or bf16(a[r]) without b),
// out[r][row_floats .. out_ld) = 0; a / b rows in_ld floats apart, one lane per four elements.
namespace {
__global__ void __launch_bounds__(256)
rows_to_bf16_kernel(const float* __restrict__ a, const float* __restrict__ b, int rows, const int* __restrict__ d_n,
                    int row_floats, int in_ld, unsigned short* __restrict__ out, int out_ld) {
    const int lim = d_n ? min(*d_n, rows) : rows;
    const int q4 = out_ld / 4;
    const long long total = (long long)lim * q4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / q4), c = (int)(i - (long long)r * q4) * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = 0.0f;
            if (c + e < row_floats) {
                t = a[(size_t)r * in_ld + c + e];
                if (b) t = (t + b[(size_t)r * in_ld + c + e]) / 2.0f;
            }
            v[e] = t;
        }
        f32x2 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
        *reinterpret_cast<f32x2*>(out + (size_t)r * out_ld + c) = o;
    }
}
}  // namespace

extern "C" int dodt_rows_to_bf16(dodt_ctx* ctx, const float* d_a, const float* d_b, int rows, const int32_t* d_n,
                                 int row_floats, int in_ld, void* d_out_bf16, int out_ld) {
    DODT_REQUIRE(ctx && d_a && d_out_bf16, "dodt_rows_to_bf16: NULL argument");
    DODT_REQUIRE(rows >= 0 && row_floats >= 1 && in_ld >= row_floats && out_ld >= row_floats && out_ld % 4 == 0,
                 "dodt_rows_to_bf16: rows of %d floats, %d apart, into rows of %d bf16 (a multiple of 4)", row_floats,
                 in_ld, out_ld);
    DODT_REQUIRE((size_t)d_out_bf16 % 8 == 0, "dodt_rows_to_bf16: 8-byte aligned output");
    if (rows == 0) return DODT_OK;
    const long long total = (long long)rows * (out_ld / 4);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(rows_to_bf16_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_a, d_b, rows, d_n, row_floats,
                       in_ld, reinterpret_cast<unsigned short*>(d_out_bf16), out_ld);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

struct dodt_fc {
    dodt_ctx* ctx = nullptr;
    int K = 0, Kp = 0, N = 0, Npad = 0, BN = 0, relu = 0;
    bool bf16 = false;
    float* d_w = nullptr;
    float* d_b = nullptr;
    float* d_w_plain = nullptr;   // [K][N] as given (bf16 layers: rounded), for K <= kSmallK
    void* d_w_dma = nullptr;      // bf16 layers with N % 128 == 0: stage images for fc_bf16_dma_kernel
    int Kd = 0;                   //   ... of K rounded up to 64
    int dma_bk = 0;               //   ... blocked for stages of this many k
};

extern "C" {

int dodt_fc_create(dodt_ctx* ctx, int K, int N, const float* w, const float* bias, int relu,
                   dodt_fc** out) {
    return dodt_fc_create_ex(ctx, K, N, w, bias, relu ? DODT_FC_RELU : 0, out);
}

int dodt_fc_create_ex(dodt_ctx* ctx, int K, int N, const float* w, const float* bias, int flags,
                      dodt_fc** out) {
    const int relu = (flags & DODT_FC_RELU) != 0;
    const bool bf16 = (flags & DODT_FC_BF16) != 0;
    DODT_REQUIRE(ctx && w && bias && out, "dodt_fc_create: NULL argument");
    DODT_REQUIRE(K >= 1 && N >= 1, "dodt_fc_create: bad sizes");
    dodt_fc* f = new dodt_fc();
    f->ctx = ctx;
    f->K = K;
    f->N = N;
    f->relu = relu;
    f->bf16 = bf16;
    f->Kp = (int)dodt::align_up((size_t)K, bf16 ? 2 * kKAlign : kKAlign);
    f->BN = (N >= 128) ? 128 : 32;
    f->Npad = (int)dodt::align_up((size_t)N, (size_t)f->BN);
    // blocked weights [n-tile][Kp/8][h][BN][4]; k = 8q + 4h + s
    std::vector<float> blk((size_t)f->Kp * f->Npad, 0.0f), b(f->Npad, 0.0f);
    //   bf16: [n-tile][Kp/16][h][BN][8] bf16 (k = 16q + 8h + s), in the first half of blk
    uint16_t* blk16 = reinterpret_cast<uint16_t*>(blk.data());
    const int kb = bf16 ? 8 : 4, KQ = f->Kp / (2 * kb);
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) {
            const int q = k / (2 * kb), h = (k % (2 * kb)) / kb, s = k % kb;
            const int nt = n / f->BN, nn = n % f->BN;
            const size_t idx = ((((size_t)nt * KQ + q) * 2 + h) * f->BN + nn) * kb + s;
            if (bf16) blk16[idx] = dodt::float_to_bf16(w[(size_t)k * N + n]);
            else blk[idx] = w[(size_t)k * N + n];
        }
    for (int n = 0; n < N; ++n) b[n] = bias[n];
    hipError_t e1 = hipMalloc(&f->d_w, blk.size() * sizeof(float));
    hipError_t e2 = hipMalloc(&f->d_b, b.size() * sizeof(float));
    if (e1 != hipSuccess || e2 != hipSuccess) {
        dodt::set_error("dodt_fc_create: hipMalloc failed");
        dodt_fc_destroy(f);
        return DODT_ERR_HIP;
    }
    DODT_HIP_CHECK(hipMemcpyAsync(f->d_w, blk.data(), blk.size() * sizeof(float),
                                  hipMemcpyHostToDevice, ctx->stream));
    DODT_HIP_CHECK(hipMemcpyAsync(f->d_b, b.data(), b.size() * sizeof(float),
                                  hipMemcpyHostToDevice, ctx->stream));
    if (bf16 && N % kBfBN == 0 && K >= 128) {
        // fc_bf16_dma_kernel's weights: [n-tile of 128][stage of BK k][column c][slot p] 16 bytes = 8 bf16, slot p
        // holding k = BK stage + 8 (p ^ swz(c)) .. + 7 (the conflict-free LDS image of a stage, contiguous)
        const int BK = bf16_dma_bk(), S = BK / 8;
        f->Kd = (int)dodt::align_up((size_t)K, (size_t)64);
        f->dma_bk = BK;
        const int ns = f->Kd / BK;
        std::vector<uint16_t> img((size_t)(N / kBfBN) * ns * kBfBN * BK, 0);
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < N; ++n) {
                const int t = n / kBfBN, c = n % kBfBN, st = k / BK, kk = k % BK;
                const int slot = (kk / 8) ^ bf_swz(c, S);
                img[((((size_t)t * ns + st) * kBfBN + c) * S + slot) * 8 + kk % 8] = dodt::float_to_bf16(w[(size_t)k * N + n]);
            }
        if (hipMalloc(&f->d_w_dma, img.size() * sizeof(uint16_t)) != hipSuccess) {
            dodt::set_error("dodt_fc_create: hipMalloc failed");
            dodt_fc_destroy(f);
            return DODT_ERR_HIP;
        }
        DODT_HIP_CHECK(hipMemcpy(f->d_w_dma, img.data(), img.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    std::vector<float> plain;
    if (K <= kSmallK && N % 4 == 0) {
        plain.assign(w, w + (size_t)K * N);
        if (bf16)
            for (float& v : plain) v = dodt::bf16_to_float(dodt::float_to_bf16(v));
        if (hipMalloc(&f->d_w_plain, plain.size() * sizeof(float)) != hipSuccess) {
            dodt::set_error("dodt_fc_create: hipMalloc failed");
            dodt_fc_destroy(f);
            return DODT_ERR_HIP;
        }
        DODT_HIP_CHECK(hipMemcpyAsync(f->d_w_plain, plain.data(), plain.size() * sizeof(float),
                                      hipMemcpyHostToDevice, ctx->stream));
    }
    DODT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *out = f;
    return DODT_OK;
}

int dodt_fc_destroy(dodt_fc* f) {
    if (!f) return DODT_OK;
    if (f->d_w) (void)hipFree(f->d_w);
    if (f->d_b) (void)hipFree(f->d_b);
    if (f->d_w_plain) (void)hipFree(f->d_w_plain);
    if (f->d_w_dma) (void)hipFree(f->d_w_dma);
    delete f;
    return DODT_OK;
}

int dodt_fc_forward(dodt_fc* f, dodt_ctx* ctx, const float* d_x, const float* d_x2, int ldx,
                    int M, const int32_t* d_m, float* d_y, int ldy) {
    DODT_REQUIRE(f && d_x && d_y, "dodt_fc_forward: NULL argument");
    DODT_REQUIRE(M >= 0 && ldx >= f->K && ldy >= f->N, "dodt_fc_forward: bad strides");
    if (M == 0) return DODT_OK;
    GemmArgs a;
    a.x = d_x; a.x2 = d_x2; a.w = f->d_w; a.bias = f->d_b; a.y = d_y;
    a.M = M; a.K = f->K; a.Kp = f->Kp; a.N = f->N; a.ldx = ldx; a.ldy = ldy; a.relu = f->relu;
    a.d_m = d_m;
    hipStream_t s = (ctx ? ctx : f->ctx)->stream;
    static const bool smallk = !(getenv("DODT_FC_SMALLK") && atoi(getenv("DODT_FC_SMALLK")) == 0);
    if (smallk && f->d_w_plain && f->N >= 64 && (size_t)d_y % 16 == 0) {
        const long long threads = (long long)M * (f->N / 4);
        if (f->bf16)
            hipLaunchKernelGGL(fc_smallk_kernel<true>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a,
                               f->d_w_plain);
        else
            hipLaunchKernelGGL(fc_smallk_kernel<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a,
                               f->d_w_plain);
        DODT_LAUNCH_CHECK();
        return DODT_OK;
    }
    if (skinny_ok(f->Npad, f->K, a)) {
        SplitOut so = {{d_y, d_y, d_y}, {f->N, f->N, f->N}, {ldy, ldy, ldy}};
        return launch_fc_skinny(s, a, so, f->bf16);
    }
    if (f->bf16) {
        if (f->BN == 128) {
            // Stage depth (DODT_FC_BF16_BK).  128 k is the fastest layer alone (343 TFLOP/s at M = 1024, N = K = 2048;
            // 64: 320, 32: 264) but its two 50 KB buffers leave room for one workgroup per CU and none beside a conv
            // workgroup; in the frame-pair pipeline, where the heads' GEMMs run in what two resident conv workgroups leave
            // of a CU (42 KB of LDS, 224 registers), 32 k (27 KB) wins: 802 -> 823 pairs/s with bf16 convs and heads.
            static const int bk = getenv("DODT_FC_BF16_BK") ? atoi(getenv("DODT_FC_BF16_BK")) : 32;
            if (bk == 64) return launch_fc<64, 128, 2, 2, 128, 64, true>(s, a, f->Npad);
            if (bk == 32) return launch_fc<64, 128, 2, 2, 128, 32, true>(s, a, f->Npad);
            return launch_fc<64, 128, 2, 2, 128, 128, true>(s, a, f->Npad);
        }
        return launch_fc<128, 32, 4, 1, 32, 64, true>(s, a, f->Npad);
    }
    if (f->BN == 128) {
        // the 2048-wide layers: LDS-DMA staged kernel (DODT_FC_DMA=0: the register-staged one)
        static const bool dma = !(getenv("DODT_FC_DMA") && atoi(getenv("DODT_FC_DMA")) == 0);
        if (dma && f->K % kDmaBK == 0 && ldx % 4 == 0 && f->K >= 2 * kDmaBK &&
            ((size_t)d_x % 16 == 0) && (!d_x2 || (size_t)d_x2 % 16 == 0))
            return launch_fc_dma(s, a, f->Npad);
        // tile shapes measured at the heads' sizes (M = 1024, N = K = 2048): 64x128 with a
        // 64-deep stage 97 TFLOP/s; 64x64 tiles and 32-deep stages within 3 % of it; 128x128
        // (2x2 MFMA tiles per wave) 106 at M = 4096 but only 128 workgroups at M = 1024
        static const int tile = getenv("DODT_FC_TILE") ? atoi(getenv("DODT_FC_TILE")) : 0;
        if (tile == 256) return launch_fc<128, 128, 2, 2, 128, 32>(s, a, f->Npad);
        if (tile == 64) return launch_fc<64, 64, 2, 2, 128>(s, a, f->Npad);
        if (tile == 32) return launch_fc<64, 128, 2, 2, 128, 32>(s, a, f->Npad);
        return launch_fc<64, 128, 2, 2, 128, 64>(s, a, f->Npad);
    }
    return launch_fc<128, 32, 4, 1>(s, a, f->Npad);
}

int dodt_fc_forward_split(dodt_fc* f, dodt_ctx* ctx, const float* d_x, int ldx, int M, const int32_t* d_m,
                          int parts, const int* widths, float* const* d_ys) {
    DODT_REQUIRE(f && d_x && widths && d_ys, "dodt_fc_forward_split: NULL argument");
    DODT_REQUIRE(parts >= 1 && parts <= 3, "dodt_fc_forward_split: %d parts (1..3)", parts);
    DODT_REQUIRE(M >= 0 && ldx >= f->K, "dodt_fc_forward_split: bad strides");
    GemmArgs a;
    a.x = d_x; a.x2 = nullptr; a.w = f->d_w; a.bias = f->d_b; a.y = nullptr;
    a.M = M; a.K = f->K; a.Kp = f->Kp; a.N = f->N; a.ldx = ldx; a.ldy = 0; a.relu = f->relu;
    a.d_m = d_m;
    SplitOut so;
    int end = 0;
    for (int p = 0; p < 3; ++p) {
        const int q = p < parts ? p : parts - 1;
        DODT_REQUIRE(widths[q] >= 1 && d_ys[q], "dodt_fc_forward_split: part %d is empty", q);
        if (p < parts) end += widths[p];
        so.y[p] = d_ys[q];
        so.n_end[p] = p < parts ? end : f->N + 1;
        so.ld[p] = widths[q];
    }
    DODT_REQUIRE(end == f->N, "dodt_fc_forward_split: the parts have %d columns, the layer %d", end, f->N);
    if (!skinny_ok(f->Npad, f->K, a)) {
        dodt::set_error("dodt_fc_forward_split: layers with N <= 32, K %% 16 == 0 and 16-byte aligned rows only");
        return DODT_ERR_UNSUPPORTED;
    }
    if (M == 0) return DODT_OK;
    return launch_fc_skinny((ctx ? ctx : f->ctx)->stream, a, so, f->bf16);
}

int dodt_fc_bf16_row_elems(const dodt_fc* f) {
    // bf16 elements an input row of dodt_fc_forward_bf16 must hold (zeros beyond K): 0 = the layer has no such path
    if (!f || !f->bf16) return 0;
    if (f->d_w_dma) return f->Kd;
    return (f->Npad == 32 && f->K % 16 == 0) ? f->K : 0;
}

int dodt_fc_forward_bf16(dodt_fc* f, dodt_ctx* ctx, const void* d_x_bf16, int ldx, int M, const int32_t* d_m,
                         void* d_y, int ldy, int y_bf16) {
    DODT_REQUIRE(f && d_x_bf16 && d_y, "dodt_fc_forward_bf16: NULL argument");
    DODT_REQUIRE(f->bf16, "dodt_fc_forward_bf16: the layer was not created with DODT_FC_BF16");
    DODT_REQUIRE(M >= 0 && ldy >= f->N, "dodt_fc_forward_bf16: bad strides");
    if (M == 0) return DODT_OK;
    GemmArgs a;
    a.x = reinterpret_cast<const float*>(d_x_bf16); a.x2 = nullptr; a.bias = f->d_b; a.y = reinterpret_cast<float*>(d_y);
    a.M = M; a.K = f->K; a.N = f->N; a.ldx = ldx; a.ldy = ldy; a.relu = f->relu; a.d_m = d_m;
    hipStream_t s = (ctx ? ctx : f->ctx)->stream;
    if (f->d_w_dma) {
        DODT_REQUIRE(ldx >= f->Kd && ldx % 8 == 0 && (size_t)d_x_bf16 % 16 == 0,
                     "dodt_fc_forward_bf16: rows of >= %d bf16 (zeros beyond K = %d), 16-byte aligned", f->Kd, f->K);
        DODT_REQUIRE((size_t)M * ldx * 2 < (1ull << 31), "dodt_fc_forward_bf16: x block beyond 2 GB");
        a.w = reinterpret_cast<const float*>(f->d_w_dma);
        a.Kp = f->Kd;
        return launch_fc_bf16_dma(s, a, f->Npad, y_bf16 != 0, f->dma_bk);
    }
    if (f->Npad == 32 && f->K % 16 == 0 && ldx % 4 == 0 && (size_t)d_x_bf16 % 8 == 0 && !y_bf16) {
        DODT_REQUIRE(ldx >= f->K, "dodt_fc_forward_bf16: bad strides");
        a.w = f->d_w; a.Kp = f->Kp;
        float* y = reinterpret_cast<float*>(d_y);
        SplitOut so = {{y, y, y}, {f->N, f->N, f->N}, {ldy, ldy, ldy}};
        return launch_fc_skinny(s, a, so, true, true);
    }
    dodt::set_error("dodt_fc_forward_bf16: layers with N %% 128 == 0 and K >= 128, or N <= 32 and K %% 16 == 0 with "
                    "float32 output, only (K = %d, N = %d)", f->K, f->N);
    return DODT_ERR_UNSUPPORTED;
}

int dodt_fc_forward_split_bf16(dodt_fc* f, dodt_ctx* ctx, const void* d_x_bf16, int ldx, int M, const int32_t* d_m,
                               int parts, const int* widths, float* const* d_ys) {
    DODT_REQUIRE(f && d_x_bf16 && widths && d_ys, "dodt_fc_forward_split_bf16: NULL argument");
    DODT_REQUIRE(f->bf16, "dodt_fc_forward_split_bf16: the layer was not created with DODT_FC_BF16");
    DODT_REQUIRE(parts >= 1 && parts <= 3, "dodt_fc_forward_split_bf16: %d parts (1..3)", parts);
    DODT_REQUIRE(M >= 0 && ldx >= f->K, "dodt_fc_forward_split_bf16: bad strides");
    GemmArgs a;
    a.x = reinterpret_cast<const float*>(d_x_bf16); a.x2 = nullptr; a.w = f->d_w; a.bias = f->d_b; a.y = nullptr;
    a.M = M; a.K = f->K; a.Kp = f->Kp; a.N = f->N; a.ldx = ldx; a.ldy = 0; a.relu = f->relu;
    a.d_m = d_m;
    SplitOut so;
    int end = 0;
    for (int p = 0; p < 3; ++p) {
        const int q = p < parts ? p : parts - 1;
        DODT_REQUIRE(widths[q] >= 1 && d_ys[q], "dodt_fc_forward_split_bf16: part %d is empty", q);
        if (p < parts) end += widths[p];
        so.y[p] = d_ys[q];
        so.n_end[p] = p < parts ? end : f->N + 1;
        so.ld[p] = widths[q];
    }
    DODT_REQUIRE(end == f->N, "dodt_fc_forward_split_bf16: the parts have %d columns, the layer %d", end, f->N);
    if (!(f->Npad == 32 && f->K % 16 == 0 && ldx % 4 == 0 && (size_t)d_x_bf16 % 8 == 0)) {
        dodt::set_error("dodt_fc_forward_split_bf16: layers with N <= 32, K %% 16 == 0 and 8-byte aligned rows only");
        return DODT_ERR_UNSUPPORTED;
    }
    if (M == 0) return DODT_OK;
    return launch_fc_skinny((ctx ? ctx : f->ctx)->stream, a, so, true, true);
}

double dodt_fc_flops(const dodt_fc* f, int M) {
    return f ? 2.0 * M * (double)f->K * f->N : 0.0;
}

}  // extern "C"
