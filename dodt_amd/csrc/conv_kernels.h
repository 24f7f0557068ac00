// 3x3 convolution / 3x3 stride-2 transposed convolution on the MFMA of gfx950: fp32
// (v_mfma_f32_32x32x2_f32, the reference's arithmetic, described below) or, template
// parameter BF16, bf16 with fp32 accumulation (v_mfma_f32_32x32x16_bf16 over CB16 bf16
// maps; same tiling, same byte geometry of every LDS image).
//
// Implicit GEMM, M = output pixels, N = output channels, K = 9 * Cin, computed
// with v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain per output).
//
//   * Activations live in HBM channel-blocked: [frame][C/8][H][W][8] ("CB8").  K is
//     walked in chunks of 8 input channels = one plane.  Per chunk the halo'd
//     patch (TH+2) x (TW+2) x 8 is staged ONCE into LDS as [pixel][8 ch + 4 pad]:
//     straight 16-byte copies of rows that are contiguous in HBM (an NHWC layout
//     would touch 32 cache lines per wave load and use a quarter of each).  Each
//     input element is fetched once per workgroup and then serves 9 taps x BN
//     channels from LDS: the kernel is MFMA-bound, not HBM-bound.
//   * A workgroup (4 waves, 256 threads) owns a TH x TW spatial tile of one
//     frame and BN output channels.  TH x TW pixels = MTB "M-tiles" of 32 pixels
//     (32/TW rows x TW columns each), stacked vertically.
//   * The MFMA's two k-lanes (lane>>5) take channels (s, s+4), s = 0..3, so ONE
//     ds_read_b128 per lane delivers the pixel operand of four consecutive MFMAs
//     (channels 4h..4h+3 of its pixel); the weights are pre-blocked on the host
//     as [tap][h][n][4] so the weight operand is one ds_read_b128 as well.  A tap
//     costs MT + NT LDS reads for 4*MT*NT MFMAs of 64 cycles.
//   * Software pipeline, one barrier per chunk: the global loads of chunk c+2 are
//     issued (into registers) before the MFMAs of chunk c, and chunk c+1 is written
//     to the other LDS buffer at the top of step c (T14 "issue early / write
//     late"), so HBM/L2 latency and the LDS writes hide under the MFMAs.  Workgroups
//     are persistent and the chunk sequence runs across their work items, so an
//     item's first loads and its epilogue stores are covered as well.
//   * The weights are the MFMA's A operand and the pixels its B operand, so in the
//     32x32 result a lane owns ONE pixel and 4 x 4 consecutive output channels:
//     the epilogue (inference batch-norm scale/shift + ReLU) stores 16 bytes per
//     lane, 1 KB contiguous per wave instruction in the CB8 output (channel-plane
//     offset given by the caller, so decoder concats are written in place).  The
//     last layer of a net can store NHWC instead (the layout of the public API).
//   * Transposed conv (3x3, stride 2, SAME): M = INPUT pixels; the 9 taps fall
//     into the 4 output-parity classes (4 + 2 + 2 + 1 taps), one accumulator tile
//     per class; out[2i+py][2j+px].
//   * conv3x3_small_cin_kernel handles the first layer (Cin = 6 or 4, NHWC input
//     as the API delivers it): planar LDS patch, ds_read_b32 fragments, CB8 out.
#pragma once
#include <hip/hip_runtime.h>

namespace dodt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // 4 registers (not the HIP struct)

struct ConvArgs {
    const float* in;      // CB8 [C/8][H][W][8], first plane in_coff/8
                          // (small-cin kernel: NHWC, pixel stride in_ld floats)
    float* out;           // CB8, first plane out_coff/8; or NHWC (out_nhwc) with pixel
                          // stride out_ld floats and first channel out_coff
    const float* w;       // blocked weights, see conv.hip dodt_extractor_set_layer
    const float* scale;   // [Cout] batch-norm scale  (rsqrt(var + eps))
    const float* shift;   // [Cout] batch-norm shift  (beta - mean * scale)
    int H, W;             // spatial size of the GEMM's M grid (conv: output = input
                          // size; transposed conv: INPUT size, output is 2H x 2W)
    int Cin, Cout;
    int in_ld, in_coff, out_ld, out_coff;
    long long in_frame_stride, out_frame_stride;  // floats between frames
    int tiles_x, tiles_y;  // spatial tiles per frame (first-layer kernel)
    const int4* items;     // MFMA kernel: work items {frame, n-tile, tile y0, tile x0}
    int n_items;
    int relu;
    int out_y0;    // conv only: rows < out_y0 are dropped, row y lands at y - out_y0
    int out_nhwc;  // 1: NHWC output (last layer), 0: CB8
    int* counter;  // work-item counter of this launch (zeroed by the host beforehand)
    int* counter_base;  // start of the counter block (words 32.. are diagnostics)
    // 2x2 max pool fused into the epilogue of the layer a pool follows (conv only, CB8):
    // pool_out is the pooled map (H/2, W/2, planes from 0), NULL when not fused
    float* pool_out;
    long long pool_frame_stride;
    // 1x1 bottleneck fused into the last layer's epilogue (NHWC output, Cout == BN == 32):
    // bneck_out[pixel] = relu(bneck_scale * sum_c bneck_w[c] * y[pixel][c] + bneck_shift)
    const float* bneck_w;          // NULL: no bottleneck
    float* bneck_out;              // (frames, H - out_y0, W)
    float bneck_scale, bneck_shift;
    long long bneck_frame_stride;  // floats between frames of bneck_out
    // split mode (PARTS = 2): a map is a hi bf16 map followed, part_stride floats further, by
    // the lo map (value = hi + lo); strides of the input, output and pooled maps
    long long in_part_stride, out_part_stride, pool_part_stride;
    const float* zeros;   // >= 16 bytes of zeros (Winograd kernel: source of out-of-image pixels)
    int debug;     // ablation switches for tools/ (0 in production): 1 = no epilogue
                   // stores, 2 = no global loads in the K loop, 4 = no MFMAs
    // bf16 LDS-DMA kernel, round 4: XCD-grouped work queue.  xcd_counters != NULL: the blocks b with the same
    // b % 8 (they share an XCD and its L2) walk the contiguous eighth b % 8 of the item table with a counter of
    // their own (xcd_counters[16 (b % 8)], zeroed by the host) -- the table then lists the channel tiles of a pixel
    // tile next to each other, so that a tile's input patch is fetched over the fabric once and is an L2 hit for
    // the other channel tiles.
    int* xcd_counters;
    // conv3x3_bf16_first2_kernel (conv1_1 folded into conv1_2): `in` is the net's NHWC fp32 input; the first layer's weights
    // as hi + lo bf16 MFMA fragments (conv.hip dodt_extractor_set_layer) and its batch-norm scale / shift
    const float* first_w;
    const float* first_scale;
    const float* first_shift;
};

constexpr int kCK = 8;          // input channels per K chunk = one CB8 plane
constexpr int kPixStride = 12;  // floats per LDS pixel: 8 channels + 4 pad (48 B)

// PARTS = 2: the "split" mode.  fp32 values are carried as hi + lo bf16 (16 mantissa bits),
// every LDS image exists twice (hi | lo) and a product is three bf16 MFMAs
// (w_hi x_hi + w_hi x_lo + w_lo x_hi, fp32 accumulate): fp32-grade results from the bf16 pipe.
template <int TW, int MTB, int WM, int WN, int BN, bool DECONV, int PARTS = 1>
struct ConvCfg {
    static constexpr int kRowsPerMT = 32 / TW;
    static constexpr int TH = MTB * kRowsPerMT;
    static constexpr int PH = DECONV ? TH + 1 : TH + 2;
    static constexpr int PW = DECONV ? TW + 1 : TW + 2;
    static constexpr int MT = MTB / WM;
    static constexpr int NT = BN / 32 / WN;
    // Row pitch of the LDS patch in floats.  A ds_read_b128 is served in lane groups
    // {0-3,12-15,20-27} / {4-11,16-19,28-31} (MI355X_MICROARCH.md, LDS): with 32 / TW patch
    // rows inside one 32-pixel fragment the pitch decides whether the 16 lanes of a group hit
    // 16 different 4-bank columns.  Pixels are 12 floats apart (3 columns), so a row of the
    // fragment must start (pitch / 4) mod 16 = 0 (TW 16), 8 (TW 8) or 4 (TW 4) columns after
    // the previous one for the groups to be conflict-free (TW 32: one row, no constraint).
    static constexpr int kPitchTarget = TW == 16 ? 0 : TW == 8 ? 8 : TW == 4 ? 4 : -1;
    static constexpr int pitch_for(int pw) {
        int pr = pw * kPixStride;
        if (kPitchTarget >= 0)
            while ((pr / 4) % 16 != kPitchTarget) pr += 4;
        return pr;
    }
    // ... unless the padded patch costs a resident workgroup per CU (LDS is the limit there)
    static constexpr int kMinWaves0 = PARTS == 2 ? 1 : 2;
    static constexpr int resident_for(int pr) {
        const int lds = 2 * PARTS * (PH * pr + 9 * kCK * BN) * 4 + 16;
        const int r = (160 * 1024) / lds;
        return r < kMinWaves0 ? r : kMinWaves0;
    }
    static constexpr int PR = resident_for(pitch_for(PW)) >= resident_for(PW * kPixStride)
                                  ? pitch_for(PW) : PW * kPixStride;
    static constexpr int kPatchFloats1 = PH * PR;   // one part
    static constexpr int kWFloats1 = 9 * kCK * BN;
    static constexpr int kPatchFloats = PARTS * kPatchFloats1;
    static constexpr int kWFloats = PARTS * kWFloats1;
    static constexpr int kBufFloats = kPatchFloats + kWFloats;  // one chunk: patch | weights
    static constexpr int kLdsBytes = 2 * kBufFloats * 4 + 16;    // double buffered + control
    static constexpr int kPatchItems = PARTS * PH * PW * 2;   // float4 per chunk
    static constexpr int kWItems = kWFloats / 4;          // float4 per chunk
    static constexpr int NP = (kPatchItems + 255) / 256;  // per-thread prefetch regs
    static constexpr int NW = (kWItems + 255) / 256;
    // accumulator registers per lane and the residency they allow (unified 512-entry
    // VGPR+AGPR file per SIMD): ask the register allocator for that many waves
    static constexpr int kAccRegs = (DECONV ? 4 * NT : MT * NT) * 16;
    // persistent workgroups hide their own prologue/epilogue, so two per CU suffice:
    // give the register allocator the full 256-register budget of 2 waves per SIMD
    // (split mode: the doubled LDS images leave room for one workgroup per CU anyway, and
    // the doubled fragments and staging registers need more than 256 registers)
    static constexpr int kMinWaves = PARTS == 2 ? 1 : 2;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(MTB % WM == 0 && (BN / 32) % WN == 0, "tile split");
    static_assert(!DECONV || MT == 1, "deconv: one 32-pixel tile per wave (x 4 parity classes)");
};

// D[row = output channel][col = pixel] += W[channel][k] * X[k][pixel]
__device__ __forceinline__ f32x16 mfma32(float w, float x, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, c, 0, 0, 0);
}

// bf16 path: one v_mfma_f32_32x32x16_bf16 takes the 8 + 8 channels that the fp32 path walks
// in four k-steps; the operands are the same 16-byte LDS reads (8 bf16 instead of 4 floats)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x16 mfma_bf16(f32x4 w, f32x4 x, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w),
                                                   __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}
// two floats -> two bf16 (round to nearest even) in one register
__device__ __forceinline__ float pack_bf16(float lo, float hi) {
    return __builtin_bit_cast(float, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
// the float value of half `which` (0 = low 16 bits) of a packed bf16 pair
__device__ __forceinline__ float bf16_value(float packed, int which) {
    const unsigned u = __builtin_bit_cast(unsigned, packed);
    return __builtin_bit_cast(float, which ? (u & 0xffff0000u) : (u << 16));
}
// Channel of accumulator register group g (4 consecutive channels) of lane half lh inside a
// 32-channel tile.  fp32 kernels: MFMA row order, 8g + 4lh.  bf16 kernels (PERM): the host
// permutes the weight rows so that a lane's 16 channels are two runs of 8 consecutive ones
// (16(g>>1) + 8lh + 4(g&1)): two 16-byte bf16 stores per lane.
template <bool PERM>
__device__ __forceinline__ int group_channel(int g, int lh) {
    return PERM ? 16 * (g >> 1) + 8 * lh + 4 * (g & 1) : 8 * g + 4 * lh;
}

// Batch-norm + ReLU + store of one 32(channel) x 32(pixel) accumulator tile.
// Lane (li = pixel, lh): register group g = r>>2 holds channels c0 + 8g + 4lh + (r&3).
// KEEP: the activated values replace the accumulators (for the fused pool).
// PERM: channel order of bf16 kernels; OUT16: store bf16 into a CB16 map (needs PERM).
// SPLIT (with OUT16): write hi = bf16(v) and lo = bf16(v - hi) maps.
template <bool KEEP = false, bool PERM = false, bool OUT16 = false, bool SPLIT = false>
__device__ __forceinline__ void store_tile(const ConvArgs& a, float* out, f32x16& acc,
                                           int c0, int lh, int y, int x, int out_w,
                                           long long plane_stride, bool ok, int frame = 0) {
    static_assert(!OUT16 || PERM, "bf16 output needs the permuted channel order");
    float dot = 0.0f;   // this lane's 16 channels of the fused 1x1 bottleneck
    f32x2 half = {0.f, 0.f};   // OUT16: the even group's four channels, packed
    f32x2 half_lo = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = c0 + group_channel<PERM>(g, lh);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + c);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + c);
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = acc[4 * g + k] * sc[k] + sh[k];
            v[k] = a.relu ? fmaxf(t, 0.0f) : t;
        }
        if constexpr (KEEP) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[4 * g + k] = v[k];
        }
        if (a.bneck_w) {
            const f32x4 bw = *reinterpret_cast<const f32x4*>(a.bneck_w + c);
            dot += v[0] * bw[0];
            dot += v[1] * bw[1];
            dot += v[2] * bw[2];
            dot += v[3] * bw[3];
        }
        if constexpr (OUT16) {
            // groups (0,1) and (2,3) are 8 consecutive channels each: one 16-byte store
            f32x2 hi = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            f32x2 lo = {0.f, 0.f};
            if constexpr (SPLIT) {
                float r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)   // residual against the rounded hi part
                    r[k] = v[k] - bf16_value(hi[k >> 1], k & 1);
                lo = f32x2{pack_bf16(r[0], r[1]), pack_bf16(r[2], r[3])};
            }
            if (!(g & 1)) {
                half = hi;
                half_lo = lo;
            } else if (ok) {
                float* dst = out + (size_t)((a.out_coff + c) >> 4) * plane_stride +
                             ((size_t)y * out_w + x) * 8 + 4 * lh;
                *reinterpret_cast<f32x4*>(dst) = f32x4{half[0], half[1], hi[0], hi[1]};
                if constexpr (SPLIT)
                    *reinterpret_cast<f32x4*>(dst + a.out_part_stride) =
                        f32x4{half_lo[0], half_lo[1], lo[0], lo[1]};
            }
        } else if (ok) {
            float* dst;
            if (a.out_nhwc)
                dst = out + ((size_t)y * out_w + x) * a.out_ld + a.out_coff + c;
            else
                dst = out + (size_t)((a.out_coff + c) >> 3) * plane_stride +
                      ((size_t)y * out_w + x) * 8 + 4 * lh;
            *reinterpret_cast<f32x4*>(dst) = v;
        }
    }
    if (a.bneck_w) {
        // the other 16 channels of this pixel live in lane ^ 32
        dot += __shfl_xor(dot, 32, 64);
        const float t = dot * a.bneck_scale + a.bneck_shift;
        if (ok && lh == 0)
            a.bneck_out[(size_t)frame * a.bneck_frame_stride + (size_t)y * out_w + x] =
                fmaxf(t, 0.0f);
    }
}

// 2x2/2 max pool of one activated 32(channel) x 32(pixel) tile: the horizontal neighbour
// is lane ^ 1, the vertical one lane ^ TW (two or more rows per 32-pixel tile) or the same
// lane of the next row's tile (`below`, TW == 32).  Lanes at even (y, x) store.
template <int TW, bool PERM = false, bool OUT16 = false, bool SPLIT = false>
__device__ __forceinline__ void pool_tile(const ConvArgs& a, const f32x16& v, const f32x16& below,
                                          int c0, int lh, int y, int x, int frame, bool ok) {
    const int OW = a.W >> 1;
    const long long plane = (long long)(a.H >> 1) * OW * 8;
    float* base = a.pool_out + (size_t)frame * a.pool_frame_stride +
                  ((size_t)(y >> 1) * OW + (x >> 1)) * 8 + 4 * lh;
    // VALID pooling: a trailing odd row / column has no window (plain-VGG levels such as
    // 175 x 200 -> 87 x 100)
    const bool writer = ok && !(y & 1) && !(x & 1) && y + 1 < a.H && x + 1 < a.W;
    f32x2 half = {0.f, 0.f}, half_lo = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 m;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = v[4 * g + k];
            if constexpr (TW < 32) t = fmaxf(t, __shfl_xor(t, TW, 64));
            else t = fmaxf(t, below[4 * g + k]);
            m[k] = fmaxf(t, __shfl_xor(t, 1, 64));
        }
        const int c = c0 + group_channel<PERM>(g, lh);
        if constexpr (OUT16) {
            f32x2 hi = {pack_bf16(m[0], m[1]), pack_bf16(m[2], m[3])};
            f32x2 lo = {0.f, 0.f};
            if constexpr (SPLIT) {
                float r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = m[k] - bf16_value(hi[k >> 1], k & 1);
                lo = f32x2{pack_bf16(r[0], r[1]), pack_bf16(r[2], r[3])};
            }
            if (!(g & 1)) {
                half = hi;
                half_lo = lo;
            } else if (writer) {
                float* dst = base + (size_t)(c >> 4) * plane;
                *reinterpret_cast<f32x4*>(dst) = f32x4{half[0], half[1], hi[0], hi[1]};
                if constexpr (SPLIT)
                    *reinterpret_cast<f32x4*>(dst + a.pool_part_stride) =
                        f32x4{half_lo[0], half_lo[1], lo[0], lo[1]};
            }
        } else if (writer) {
            *reinterpret_cast<f32x4*>(base + (size_t)(c >> 3) * plane) = m;
        }
    }
}

// BF16: activations are CB16 bf16 maps (32 bytes per pixel and plane, like CB8 fp32: all
// staging below is byte-identical), a K chunk is 16 channels and one MFMA per tap and tile.
template <int TW, int MTB, int WM, int WN, int BN, bool DECONV, bool BF16 = false, int PARTS = 1>
__global__ void
__launch_bounds__(256, (ConvCfg<TW, MTB, WM, WN, BN, DECONV, PARTS>::kMinWaves))
conv3x3_mfma_kernel(const ConvArgs a) {
    static_assert(PARTS == 1 || BF16, "the split mode runs on the bf16 MFMA");
    constexpr bool SPLIT = PARTS == 2;
    using Cfg = ConvCfg<TW, MTB, WM, WN, BN, DECONV, PARTS>;
    constexpr int PW = Cfg::PW;
    constexpr int MT = Cfg::MT, NT = Cfg::NT, NP = Cfg::NP, NW = Cfg::NW;
    constexpr int NACC = DECONV ? 4 * NT : MT * NT;
    constexpr int PS = kPixStride;
    constexpr int NSLOT = NP + NW;            // staging registers (float4) per thread
    constexpr int SPT = (NSLOT + 8) / 9;      // staging slots handled per tap

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sP = smem;                      // buffer b: [PH*PW][12] at + b * kBufFloats
    float* sW = smem + Cfg::kPatchFloats;  //           [9][2][BN][4] behind it
    int* s_ctrl = reinterpret_cast<int*>(smem + 2 * Cfg::kBufFloats);  // [0] next work item

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    constexpr int kChunkCh = BF16 ? 16 : 8;   // channels per K chunk = one plane
    const int nchunks = a.Cin / kChunkCh;
    const int in_plane = a.H * a.W * 8;  // floats per input plane

    // Persistent workgroup.  Work item = (frame, n-tile, spatial tile), listed by the host
    // (a layer is a main launch of big tiles plus, where that evens out the last round, a
    // tail launch of smaller ones); the first item is blockIdx.x, further ones come from an
    // atomic counter (a workgroup that starts late because another stream holds its CU
    // simply takes fewer items).
    struct Item { int frame, ntile, ty0, tx0; };
    auto decode = [&](int it) {
        const int4 v = a.items[__builtin_amdgcn_readfirstlane(it)];
        return Item{v.x, v.y, v.z, v.w};
    };

    // Staging slot k < NP: patch float4 (tid + 256 k) -> (pixel p, half g); slot NP + k:
    // weight float4 (tid + 256 k).  Loads are unconditional (halo / surplus slots read a
    // valid dummy address and are zeroed or skipped at LDS-write time): a load under a
    // branch would make hipcc wait for it on the spot.
    constexpr int PR = Cfg::PR;
    int p_lds[NP];        // float offset of this thread's staging slot k inside the LDS patch
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int t = tid + k * 256;
        int p = t >> 1, part = 0;
        if (SPLIT && p >= Cfg::PH * PW) {
            p -= Cfg::PH * PW;
            part = Cfg::kPatchFloats1;
        }
        const int py = p / PW, px = p - py * PW;
        p_lds[k] = part + py * PR + px * PS + (t & 1) * 4;
    }
    int p_glb[NP];        // float offset inside a plane (0 when padded), current load item
    unsigned ok_issue = 0;  // zero-pad mask of the item whose loads are being issued
    const float* in_item = a.in;  // plane 0 of the load item's input
    const f32x4* w_item = reinterpret_cast<const f32x4*>(a.w);
    auto setup_loads = [&](const Item& it) {
        ok_issue = 0;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int t = tid + k * 256;
            const int g = t & 1;
            int p = t >> 1;
            int part_off = 0;                       // split mode: the second half is the lo map
            if (SPLIT && p >= Cfg::PH * PW) {
                p -= Cfg::PH * PW;
                part_off = (int)a.in_part_stride;
            }
            const int py = p / PW, px = p - py * PW;
            const int gy = it.ty0 - 1 + py, gx = it.tx0 - 1 + px;
            const bool ok = t < Cfg::kPatchItems && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            p_glb[k] = ok ? part_off + (gy * a.W + gx) * 8 + g * 4 : 0;
            ok_issue |= (ok ? 1u : 0u) << k;
        }
        in_item = a.in + (size_t)it.frame * a.in_frame_stride +
                  (size_t)(a.in_coff / kChunkCh) * in_plane;
        w_item = reinterpret_cast<const f32x4*>(a.w + (size_t)it.ntile * nchunks * Cfg::kWFloats);
    };
    f32x4 pre[NSLOT];
    // load staging slot J of chunk CH of the load item into its register
#define DODT_LOAD_SLOT(J, CH)                                                                 \
    {                                                                                         \
        if constexpr ((J) < NP) {                                                             \
            pre[J] = *reinterpret_cast<const f32x4*>(in_item + (size_t)(CH) * in_plane +      \
                                                     p_glb[(J) < NP ? (J) : 0]);              \
        } else if constexpr ((J) < NSLOT) {                                                   \
            const int t_ = tid + ((J) - NP) * 256;                                            \
            pre[J] = w_item[(size_t)(CH) * Cfg::kWItems + min(t_, Cfg::kWItems - 1)];         \
        }                                                                                     \
    }
    // write staging slot J (data of the chunk after the one being computed) to buffer BUF
#define DODT_STORE_SLOT(J, BUF, OKMASK)                                                       \
    {                                                                                         \
        if constexpr ((J) < NP) {                                                             \
            const int t_ = tid + (J) * 256;                                                   \
            if (t_ < Cfg::kPatchItems)                                                        \
                *reinterpret_cast<f32x4*>(sP + (BUF) * Cfg::kBufFloats +                      \
                                          p_lds[(J) < NP ? (J) : 0]) =                        \
                    (((OKMASK) >> (J)) & 1u) ? pre[J] : f32x4{0.f, 0.f, 0.f, 0.f};            \
        } else if constexpr ((J) < NSLOT) {                                                   \
            const int t_ = tid + ((J) - NP) * 256;                                            \
            if (t_ < Cfg::kWItems)                                                            \
                reinterpret_cast<f32x4*>(sW + (BUF) * Cfg::kBufFloats)[t_] = pre[J];          \
        }                                                                                     \
    }

    // lane bases (floats)
    const int x_base =
        (li / TW + wm * MT * Cfg::kRowsPerMT) * PR + (li % TW) * PS + 4 * lh;
    const int w_base = (lh * BN + wn * NT * 32 + li) * 4;

    // ---- flat pipeline over (item, chunk) steps --------------------------------------------
    //   step k:  MFMAs of chunk k from LDS buffer k&1; in their shadow, spread over the nine
    //            taps: chunk k+1 (in registers since step k-1) is written to buffer (k+1)&1
    //            and the global loads of chunk k+2 are issued into the freed registers;
    //            epilogue if chunk k ends an item; ONE barrier.
    // The chunk sequence runs across item boundaries, so neither the first loads of an
    // item nor its stores ever leave the matrix pipe idle.
    int comp_item = blockIdx.x, comp_ch = 0;   // chunk the MFMAs work on
    if (comp_item >= a.n_items) return;        // uniform per workgroup
    // diagnostic build switch (debug & 8): shader clock of workgroup 0, written to words
    // that nothing else reads (counter[32..35]); see MI355X_MICROARCH 'DVFS give-back'
    unsigned long long dbg_t0 = 0, dbg_r0 = 0;
    if ((a.debug & 8) && blockIdx.x == 0 && tid == 0) {
        dbg_t0 = __builtin_amdgcn_s_memtime();
        dbg_r0 = __builtin_amdgcn_s_memrealtime();
    }
    int load_item = comp_item, load_ch = 0;    // chunk whose loads are issued next
    // Items after comp_item come from the counter, one fetch during every item's first step.
    // With >= 4 chunks per item that fetch is the item's successor q0 (needed by the loads two
    // chunks before the item ends).  Items of 2-3 chunks (bf16 layers with 32 input channels;
    // fp32 layers always have >= 4) need their successor from the start: then q0 is fetched
    // before the prologue's barrier and each later fetch is the item behind it, q1 ("deep").
    // Claiming no further ahead than necessary keeps the last round even.
    const bool deep = BF16 && nchunks < 4;
    int q0 = a.n_items, q1 = a.n_items;
    if (deep && tid == 0) s_ctrl[0] = (int)gridDim.x + atomicAdd(a.counter, 1);
    setup_loads(decode(load_item));
    unsigned ok_regs = ok_issue;               // mask of the data sitting in pre[]
    // prologue: chunk 0 -> buffer 0, chunk 1 -> registers
#define DODT_ADVANCE_LOAD()                                                                   \
    {                                                                                         \
        if (++load_ch == nchunks) {                                                           \
            load_ch = 0;                                                                      \
            load_item = (load_item == comp_item) ? q0 : q1;                                   \
            if (load_item < a.n_items) setup_loads(decode(load_item));                        \
        }                                                                                     \
    }
#define DODT_FOR_SLOTS(BODY) \
    { BODY(0) BODY(1) BODY(2) BODY(3) BODY(4) BODY(5) BODY(6) BODY(7) BODY(8) BODY(9) BODY(10) \
      BODY(11) BODY(12) BODY(13) BODY(14) BODY(15) BODY(16) BODY(17) }
    static_assert(NSLOT <= 18, "raise DODT_FOR_SLOTS");
#define DODT_PRO_LOAD(J) DODT_LOAD_SLOT(J, 0)
#define DODT_PRO_STORE(J) DODT_STORE_SLOT(J, 0, ok_regs)
    DODT_FOR_SLOTS(DODT_PRO_LOAD)
    DODT_FOR_SLOTS(DODT_PRO_STORE)
    __syncthreads();
    if (deep) {
        q0 = s_ctrl[0];
        __syncthreads();                       // s_ctrl[0] is rewritten in the first step
    }
    DODT_ADVANCE_LOAD()
    ok_regs = ok_issue;
    {
        const int ch_ = load_ch;
#define DODT_PRO_LOAD1(J) DODT_LOAD_SLOT(J, ch_)
        DODT_FOR_SLOTS(DODT_PRO_LOAD1)   // harmless re-read of a valid address if no chunk is left
    }
    bool more_loads = load_item < a.n_items;   // pre[] holds a real chunk
    if (more_loads) DODT_ADVANCE_LOAD()

    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;

    int buf = 0;
    while (comp_item < a.n_items) {
        const float* bP = sP + buf * Cfg::kBufFloats;
        const float* bW = sW + buf * Cfg::kBufFloats;
        // staging plan of this step: write the registers (chunk k+1) to buffer buf^1, then
        // refill them with chunk k+2 = (load_item, load_ch) -- all unconditional
        const unsigned ok_store = ok_regs;
        const int ld_ch = load_ch;
        const bool more = load_item < a.n_items;   // a real chunk is left to load
        const bool issue = !(a.debug & 2);         // production: always (a surplus load
                                                   // re-reads a valid address, no branch)
        if (comp_ch == 0 && tid == 0)   // fetch the successor (deep: the item behind q0)
            s_ctrl[0] = (int)gridDim.x + atomicAdd(a.counter, 1);

        if (a.debug & 4) {
            // ablation: no MFMAs (staging still runs)
#define DODT_ABL(J) DODT_STORE_SLOT(J, buf ^ 1, ok_store) DODT_LOAD_SLOT(J, ld_ch)
            DODT_FOR_SLOTS(DODT_ABL)
        } else if constexpr (!DECONV) {
            f32x4 xf[2][MT], wf[2][NT];
            f32x4 xl[2][SPLIT ? MT : 1], wl[2][SPLIT ? NT : 1];   // lo parts (split mode)
            constexpr int kXlo = Cfg::kPatchFloats1, kWlo = Cfg::kWFloats1;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                xf[0][mt] = *reinterpret_cast<const f32x4*>(
                    bP + x_base + (mt * Cfg::kRowsPerMT) * PR);
                if constexpr (SPLIT)
                    xl[0][mt] = *reinterpret_cast<const f32x4*>(
                        bP + kXlo + x_base + (mt * Cfg::kRowsPerMT) * PR);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                wf[0][nt] = *reinterpret_cast<const f32x4*>(bW + w_base + nt * 128);
                if constexpr (SPLIT)
                    wl[0][nt] = *reinterpret_cast<const f32x4*>(bW + kWlo + w_base + nt * 128);
            }
#define DODT_TAP(TAP)                                                                         \
            {                                                                                 \
                constexpr int cb = (TAP) & 1, nb = cb ^ 1;                                    \
                if constexpr ((TAP) + 1 < 9) {                                                \
                    constexpr int ky = ((TAP) + 1) / 3, kx = ((TAP) + 1) % 3;                 \
                    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                       \
                        xf[nb][mt] = *reinterpret_cast<const f32x4*>(                         \
                            bP + x_base + (mt * Cfg::kRowsPerMT + ky) * PR + kx * PS);        \
                        if constexpr (SPLIT)                                                  \
                            xl[nb][mt] = *reinterpret_cast<const f32x4*>(                     \
                                bP + kXlo + x_base +                                          \
                                (mt * Cfg::kRowsPerMT + ky) * PR + kx * PS);                  \
                    }                                                                         \
                    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                       \
                        wf[nb][nt] = *reinterpret_cast<const f32x4*>(                         \
                            bW + w_base + ((TAP) + 1) * 2 * BN * 4 + nt * 128);               \
                        if constexpr (SPLIT)                                                  \
                            wl[nb][nt] = *reinterpret_cast<const f32x4*>(                     \
                                bW + kWlo + w_base + ((TAP) + 1) * 2 * BN * 4 + nt * 128);    \
                    }                                                                         \
                }                                                                             \
                /* the next tap's LDS reads stay ABOVE this tap's MFMAs (hipcc would */       \
                /* otherwise sink them below to save registers and expose their latency) */   \
                __builtin_amdgcn_sched_barrier(0);                                            \
                if constexpr (BF16) {                                                         \
                    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                         \
                        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                     \
                            acc[mt * NT + nt] =                                               \
                                mfma_bf16(wf[cb][nt], xf[cb][mt], acc[mt * NT + nt]);         \
                    DODT_STAGE_TAP(TAP)                                                       \
                    if constexpr (SPLIT) {   /* the two cross terms, small ones last */        \
                        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                     \
                            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                 \
                                acc[mt * NT + nt] = mfma_bf16(                                \
                                    wf[cb][nt], xl[cb][mt < (SPLIT ? MT : 1) ? mt : 0],       \
                                    acc[mt * NT + nt]);                                       \
                        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                     \
                            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                 \
                                acc[mt * NT + nt] = mfma_bf16(                                \
                                    wl[cb][nt < (SPLIT ? NT : 1) ? nt : 0], xf[cb][mt],       \
                                    acc[mt * NT + nt]);                                       \
                    }                                                                         \
                } else {                                                                      \
                    _Pragma("unroll") for (int s = 0; s < 2; ++s)                             \
                        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                     \
                            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                 \
                                acc[mt * NT + nt] =                                           \
                                    mfma32(wf[cb][nt][s], xf[cb][mt][s], acc[mt * NT + nt]);  \
                    /* staging in the shadow of the MFMAs just issued */                      \
                    DODT_STAGE_TAP(TAP)                                                       \
                    _Pragma("unroll") for (int s = 2; s < 4; ++s)                             \
                        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                     \
                            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                 \
                                acc[mt * NT + nt] =                                           \
                                    mfma32(wf[cb][nt][s], xf[cb][mt][s], acc[mt * NT + nt]);  \
                }                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                            \
            }
#define DODT_STAGE_ONE(J)                                                                     \
            DODT_STORE_SLOT(J, buf ^ 1, ok_store) if (issue) DODT_LOAD_SLOT(J, ld_ch)
#define DODT_STAGE_TAP(TAP)                                                                   \
            {                                                                                 \
                if constexpr (SPT >= 1) { DODT_STAGE_ONE((TAP) * SPT + 0) }                   \
                if constexpr (SPT >= 2) { DODT_STAGE_ONE((TAP) * SPT + 1) }                   \
            }
            static_assert(SPT <= 2, "more than two staging slots per tap");
            DODT_TAP(0) DODT_TAP(1) DODT_TAP(2) DODT_TAP(3) DODT_TAP(4)
            DODT_TAP(5) DODT_TAP(6) DODT_TAP(7) DODT_TAP(8)
        } else {
            // patch origin is (ty0-1, tx0-1): in[i][j] sits at patch (r+1, c+1)
            const float* pa = bP + x_base;
            f32x4 af[4], al[SPLIT ? 4 : 1];
            constexpr int kXlo = Cfg::kPatchFloats1, kWlo = Cfg::kWFloats1;
            af[0] = *reinterpret_cast<const f32x4*>(pa + PR + PS);        // in[i][j]
            af[1] = *reinterpret_cast<const f32x4*>(pa + 1 * PS);         // in[i-1][j]
            af[2] = *reinterpret_cast<const f32x4*>(pa + PR);             // in[i][j-1]
            af[3] = *reinterpret_cast<const f32x4*>(pa);                  // in[i-1][j-1]
            if constexpr (SPLIT) {
                al[0] = *reinterpret_cast<const f32x4*>(pa + kXlo + PR + PS);
                al[1] = *reinterpret_cast<const f32x4*>(pa + kXlo + 1 * PS);
                al[2] = *reinterpret_cast<const f32x4*>(pa + kXlo + PR);
                al[3] = *reinterpret_cast<const f32x4*>(pa + kXlo);
            }
            // taps ky*3+kx; out[2i+ky-2*di][2j+kx-2*dj] += in[i-di][j-dj] * w[ky][kx]: every
            // tap feeds one output-parity class from one of the four input pixels.  Order:
            // classes alternate so that consecutive taps use different accumulators.
            f32x4 bw[2][NT], bl[2][SPLIT ? NT : 1];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bw[0][nt] = *reinterpret_cast<const f32x4*>(bW + w_base + nt * 128);   // tap 0
                if constexpr (SPLIT)
                    bl[0][nt] = *reinterpret_cast<const f32x4*>(bW + kWlo + w_base + nt * 128);
            }
#define DODT_DTAP(I, TAP, CLS, AF, NEXT_TAP)                                                  \
            {                                                                                 \
                constexpr int cb = (I) & 1, nb = cb ^ 1;                                      \
                if constexpr ((NEXT_TAP) >= 0) {                                              \
                    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                       \
                        bw[nb][nt] = *reinterpret_cast<const f32x4*>(                         \
                            bW + w_base + (NEXT_TAP) * 2 * BN * 4 + nt * 128);                \
                        if constexpr (SPLIT)                                                  \
                            bl[nb][nt] = *reinterpret_cast<const f32x4*>(                     \
                                bW + kWlo + w_base + (NEXT_TAP) * 2 * BN * 4 + nt * 128);     \
                    }                                                                         \
                }                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                            \
                if constexpr (BF16) {                                                         \
                    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                         \
                        acc[(CLS) * NT + nt] =                                                \
                            mfma_bf16(bw[cb][nt], af[AF], acc[(CLS) * NT + nt]);              \
                    DODT_STAGE_TAP(I)                                                         \
                    if constexpr (SPLIT) {                                                    \
                        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                     \
                            acc[(CLS) * NT + nt] = mfma_bf16(                                 \
                                bw[cb][nt], al[SPLIT ? (AF) : 0], acc[(CLS) * NT + nt]);      \
                        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                     \
                            acc[(CLS) * NT + nt] = mfma_bf16(                                 \
                                bl[cb][SPLIT ? nt : 0], af[AF], acc[(CLS) * NT + nt]);        \
                    }                                                                         \
                } else {                                                                      \
                    _Pragma("unroll") for (int s = 0; s < 2; ++s)                             \
                        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                     \
                            acc[(CLS) * NT + nt] =                                            \
                                mfma32(bw[cb][nt][s], af[AF][s], acc[(CLS) * NT + nt]);       \
                    DODT_STAGE_TAP(I)                                                         \
                    _Pragma("unroll") for (int s = 2; s < 4; ++s)                             \
                        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                     \
                            acc[(CLS) * NT + nt] =                                            \
                                mfma32(bw[cb][nt][s], af[AF][s], acc[(CLS) * NT + nt]);       \
                }                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                            \
            }
            //        step tap class input  next tap
            DODT_DTAP(0, 0, 0, 0, 1)
            DODT_DTAP(1, 1, 1, 0, 3)
            DODT_DTAP(2, 3, 2, 0, 4)
            DODT_DTAP(3, 4, 3, 0, 6)
            DODT_DTAP(4, 6, 0, 1, 7)
            DODT_DTAP(5, 7, 1, 1, 5)
            DODT_DTAP(6, 5, 2, 2, 2)
            DODT_DTAP(7, 2, 0, 2, 8)
            DODT_DTAP(8, 8, 0, 3, -1)
        }
        if (more) {
            ok_regs = ok_issue;       // pre[] now holds (load_item, ld_ch)
            DODT_ADVANCE_LOAD()       // uses q0 / q1 read after earlier barriers
        }
        const bool item_done = (comp_ch + 1 == nchunks);
        if (item_done) {
            // ---- epilogue: the stores drain under the next item's MFMAs ---------------
            const Item cur = decode(comp_item);
            float* out = a.out + (size_t)cur.frame * a.out_frame_stride;
            const bool st = !(a.debug & 1);
            if constexpr (!DECONV) {
                const int out_rows = a.H - a.out_y0;
                const long long plane = (long long)out_rows * a.W * 8;
                // pooling in registers needs both rows of a 2x2 window in this wave
                constexpr bool kCanPool = Cfg::kRowsPerMT >= 2 || MT % 2 == 0;
                const bool pool = kCanPool && a.pool_out != nullptr;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int y = cur.ty0 + (wm * MT + mt) * Cfg::kRowsPerMT + li / TW;
                    const int x = cur.tx0 + li % TW;
                    const bool ok = st && y < a.H && x < a.W && y >= a.out_y0;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int c0 = cur.ntile * BN + (wn * NT + nt) * 32;
                        // bf16 kernels store bf16 CB16 maps, except the net's last layer (NHWC fp32)
                        if (pool)
                            store_tile<true, BF16, BF16, SPLIT>(a, out, acc[mt * NT + nt], c0, lh,
                                                                y - a.out_y0, x, a.W, plane, ok,
                                                                cur.frame);
                        else if (BF16 && !a.out_nhwc)
                            store_tile<false, BF16, BF16, SPLIT>(a, out, acc[mt * NT + nt], c0, lh,
                                                                 y - a.out_y0, x, a.W, plane, ok,
                                                                 cur.frame);
                        else
                            store_tile<false, BF16, false>(a, out, acc[mt * NT + nt], c0, lh,
                                                           y - a.out_y0, x, a.W, plane, ok, cur.frame);
                    }
                }
                if constexpr (kCanPool) {
                    if (pool) {
#pragma unroll
                        for (int mt = 0; mt < MT; mt += (Cfg::kRowsPerMT >= 2 ? 1 : 2)) {
                            const int y = cur.ty0 + (wm * MT + mt) * Cfg::kRowsPerMT + li / TW;
                            const int x = cur.tx0 + li % TW;
                            const bool ok = st && y < a.H && x < a.W;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                pool_tile<TW, BF16, BF16, SPLIT>(a, acc[mt * NT + nt],
                                              acc[(Cfg::kRowsPerMT >= 2 ? mt : mt + 1) * NT + nt],
                                              cur.ntile * BN + (wn * NT + nt) * 32, lh, y, x,
                                              cur.frame, ok);
                        }
                    }
                }
            } else {
                const int y = cur.ty0 + wm * Cfg::kRowsPerMT + li / TW;
                const int x = cur.tx0 + li % TW;
                const bool ok = st && y < a.H && x < a.W;
                const long long plane = (long long)(2 * a.H) * (2 * a.W) * 8;
#pragma unroll
                for (int cls = 0; cls < 4; ++cls)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        store_tile<false, BF16, BF16, SPLIT>(a, out, acc[cls * NT + nt],
                                                      cur.ntile * BN + (wn * NT + nt) * 32, lh,
                                                      2 * y + (cls >> 1), 2 * x + (cls & 1),
                                                      2 * a.W, plane, ok);
            }
#pragma unroll
            for (int k = 0; k < NACC; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
        }
        __syncthreads();  // buffer buf^1 is complete, buffer buf is free, s_ctrl is visible
        buf ^= 1;
        if (comp_ch == 0) {                        // fetched at the top of this step
            if (deep) q1 = s_ctrl[0];
            else q0 = s_ctrl[0];
        }
        if (item_done) {
            comp_ch = 0;
            comp_item = q0;
            if (deep) q0 = q1;
        } else {
            ++comp_ch;
        }
    }
    if ((a.debug & 8) && blockIdx.x == 0 && tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.counter + 32 - (a.counter - a.counter_base));
        o[0] = t1 - dbg_t0;
        o[1] = r1 - dbg_r0;
    }
#undef DODT_LOAD_SLOT
#undef DODT_STORE_SLOT
#undef DODT_ADVANCE_LOAD
#undef DODT_FOR_SLOTS
#undef DODT_TAP
#undef DODT_STAGE_ONE
#undef DODT_STAGE_TAP
#undef DODT_DTAP
}

// ---------------------------------------------------------------------------
// First layer (Cin = 6 for BEV, 4 for the padded image): NHWC input, one chunk of
// CK = Cin channels, planar LDS patch [CK][PH*PW (+pad)], weights [9][CK][32],
// scalar ds_read_b32 fragments, CB8 output.  1.5 % of the FLOPs.
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
    static constexpr int kLdsBytes = (2 * kPatchFloats + kWFloats) * 4;      // two patch images (round 4)
    static_assert(CK % 2 == 0 && MTB % 4 == 0, "shape");
};

// OUT16: the output is a CB16 bf16 map (the arithmetic stays fp32; the host permutes the
// weights' output channels like for the bf16 kernels).
template <int TW, int MTB, int CK, bool OUT16 = false, bool SPLIT = false>
__global__ void __launch_bounds__(256)
conv3x3_small_cin_kernel(const ConvArgs a) {
    using Cfg = SmallCfg<TW, MTB, CK>;
    constexpr int TH = Cfg::TH, PH = Cfg::PH, PW = Cfg::PW, PS = Cfg::PS, MT = Cfg::MT;
    constexpr int BN = 32;
    constexpr int VEC = (CK % 4 == 0) ? 4 : 2;

    // Round 4: persistent workgroups (grid = what stays resident, items taken with the grid's stride) with the weights
    // staged once and TWO patch images: while the MFMAs of item i run, the loads of item i + 1 are in flight into
    // registers; one barrier per item.  (Round 1-3: one item per workgroup -- fill, 108 MFMAs, stores in sequence,
    // three workgroups per CU: 0.29-0.34 of the matrix pipe.)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW = smem + 2 * Cfg::kPatchFloats;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wm = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int tiles = a.tiles_x * a.tiles_y;
    const int n_items = a.n_items;
    constexpr int CG = CK / VEC;
    constexpr int ITEMS = PH * PW * CG;
    constexpr int NIT = (ITEMS + 255) / 256;
    constexpr int NW = (Cfg::kWFloats / 4 + 255) / 256;
    float v[NIT][VEC];
    // every load of the thread is issued before the first LDS write (a loop with a load and its write per trip ran
    // one global-memory latency per trip)
    auto load_patch = [&](int bid) {
        const int frame = bid / tiles, t2 = bid - frame * tiles;
        const int ty0 = (t2 / a.tiles_x) * TH, tx0 = (t2 % a.tiles_x) * TW;
        const float* in = a.in + (size_t)frame * a.in_frame_stride;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int t = tid + 256 * i;
            const int cg = t % CG, p = t / CG;
            const int py = p / PW, px = p - py * PW;
            const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[i][k] = 0.0f;
            if (t < ITEMS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const float* src = in + ((size_t)gy * a.W + gx) * a.in_ld + a.in_coff + cg * VEC;
                if constexpr (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(src);
                    v[i][0] = q.x; v[i][1] = q.y; v[i][2] = q.z; v[i][3] = q.w;
                } else {
                    const float2 q = *reinterpret_cast<const float2*>(src);
                    v[i][0] = q.x; v[i][1] = q.y;
                }
            }
        }
    };
    int bid = blockIdx.x;
    if (bid >= n_items) return;
    {
        float4 wv[NW];
        const float4* wsrc = reinterpret_cast<const float4*>(a.w);
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int t = tid + 256 * i;
            wv[i] = t < Cfg::kWFloats / 4 ? wsrc[t] : float4{0.f, 0.f, 0.f, 0.f};
        }
        load_patch(bid);
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int t = tid + 256 * i;
            if (t < Cfg::kWFloats / 4) reinterpret_cast<float4*>(sW)[t] = wv[i];
        }
    }
    const int x_base = lh * PS + (li / TW) * PW + (li % TW) + wm * MT * Cfg::kRowsPerMT * PW;
    const int w_base = lh * BN + li;
    const long long plane = (long long)a.H * a.W * 8;
    int buf = 0;
    for (; bid < n_items; bid += gridDim.x) {
        float* sP = smem + buf * Cfg::kPatchFloats;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int t = tid + 256 * i;
            const int cg = t % CG, p = t / CG;
            if (t < ITEMS) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) sP[(cg * VEC + k) * PS + p] = v[i][k];
            }
        }
        __syncthreads();     // this item's patch (and, the first time, the weights) complete; the other image is free
        if (bid + (int)gridDim.x < n_items) load_patch(bid + gridDim.x);     // in flight under the MFMAs
        f32x16 acc[MT];
#pragma unroll
        for (int k = 0; k < MT; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int cp = 0; cp < CK / 2; ++cp) {
                const float wv = sW[w_base + (tap * CK + 2 * cp) * BN];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float xv = sP[x_base + 2 * cp * PS + (mt * Cfg::kRowsPerMT + ky) * PW + kx];
                    acc[mt] = mfma32(wv, xv, acc[mt]);
                }
            }
        }
        const int frame = bid / tiles, t2 = bid - frame * tiles;
        const int ty0 = (t2 / a.tiles_x) * TH, tx0 = (t2 % a.tiles_x) * TW;
        float* out = a.out + (size_t)frame * a.out_frame_stride;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int y = ty0 + (wm * MT + mt) * Cfg::kRowsPerMT + li / TW;
            const int x = tx0 + li % TW;
            store_tile<false, OUT16, OUT16, SPLIT>(a, out, acc[mt], 0, lh, y, x, a.W, plane,
                                            y < a.H && x < a.W);
        }
        buf ^= 1;
    }
}

}  // namespace dodt
