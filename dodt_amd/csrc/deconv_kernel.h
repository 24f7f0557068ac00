// 3x3 stride-2 transposed convolution (slim.conv2d_transpose, SAME: output 2H x 2W) on the MFMA of
// gfx950, LDS-DMA staged -- the upconv layers of the pyramid extractors.  fp32 (v_mfma_f32_16x16x4_f32,
// CB8 maps) and, since round 3, the bf16 conv path (BF16: v_mfma_f32_16x16x16_bf16 on CB16 bf16 maps).
// The two are byte-identical in everything but the MFMA: a CB16 bf16 pixel is 32 bytes like a CB8 fp32
// one, a lane's 8-byte LDS read holds input channels 4g .. 4g+3 as bf16 where it held 2g, 2g+1 as
// floats, and ONE 16x16x16 MFMA takes the 16 channels of a chunk that two 16x16x4 ones take 8 of.
//
// out[2i + ky - 2 di][2j + kx - 2 dj] += in[i - di][j - dj] . w[ky][kx]: every tap feeds one
// output-parity class from one of the four input pixels (i, j), (i-1, j), (i, j-1), (i-1, j-1);
// per input pixel 9 multiplications per channel pair (1 + 2 + 2 + 4 over the four classes).
// There is no input transform, so the K loop holds no vector instruction at all (which on this
// chip would wait for the MFMAs): LDS reads, LDS-DMA copies with scalar operands
// and MFMAs that accumulate in place in AGPRs (inline asm, as in wino43_kernel.h).
//
// Work decomposition (one workgroup = 4 waves = 16 x 16 INPUT pixels x 32 output channels; two
// workgroups per CU):
//   * wave w: input rows 4w .. 4w + 3 (four groups of 16 pixels) x both 16-channel blocks:
//     4 rows x 4 classes x 2 blocks of v_mfma_f32_16x16x4_f32 accumulators = 128 AGPRs.
//     Lane l: pixel t = l % 16 of a row (B operand column), k-pair g = l / 16.
//   * K in chunks of 8 input channels; per chunk the 17 x 17 patch (origin one up / left of the
//     tile) and the chunk's weights [tap][g][n][channel block][2] are copied global -> LDS (19
//     KB), double buffered, one barrier per chunk; persistent workgroups on a work queue.
//   * per chunk a lane reads 5 rows x 2 column shifts of its pixel (ds_read_b64) and 9 weight
//     fragments (ds_read_b128 = both channel blocks of a tap), and issues 144 MFMAs.
//   * epilogue: batch-norm + ReLU, 16-byte stores of the four classes into the CB8 output (which
//     may be a channel range of a concat buffer).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "conv_kernels.h"
#include "lds_dma.h"
#include "wino_kernels.h"
#include "wino43_kernel.h"

namespace dodt {

template <int CB>      // channel blocks of 16 per workgroup tile: 2, or 1 for layers with few items
struct DeconvCfg {
    static constexpr int TW = 16, TH = 16, BN = 16 * CB;
    static constexpr int PH = TH + 1, PW = TW + 1;
    // patch image: pixel (py, px) in 32-byte cell py * 17 + px, halves swapped when (px >> 3) & 1:
    // the 16 consecutive pixels x 2 channel pairs one LDS cycle of a ds_read_b64 serves fall into
    // 16 different 16-byte bank columns for both column shifts
    static constexpr int kPitch = PW;
    static constexpr int kPatchSlots = PH * kPitch * 2;               // 578
    static constexpr int kPatchInstr = (kPatchSlots + 63) / 64;       // 10
    static constexpr int kPatchFloats = kPatchInstr * 256;
    static constexpr int kWFloats = (9 * 8 * BN + 255) / 256 * 256;   // 2304 (CB = 1: 1152 padded to 1280)
    static constexpr int kWInstr = kWFloats / 256;
    static constexpr int kBufFloats = kPatchFloats + kWFloats;
    static constexpr int kLdsBytes = 2 * kBufFloats * 4 + 16 + 1024;  // + control word + dummy slot
    static constexpr int kPatchPerWave = (kPatchInstr + 3) / 4;       // 3
    static constexpr int kWPerWave = (kWInstr + 3) / 4;               // 3
};

// (step, tap, class, input): classes alternate so that consecutive steps use different
// accumulators; input 0 = (i, j), 1 = (i-1, j), 2 = (i, j-1), 3 = (i-1, j-1)
struct DeconvTap { int tap, cls, di, dj; };
constexpr DeconvTap kDeconvTaps[9] = {{0, 0, 0, 0}, {1, 1, 0, 0}, {3, 2, 0, 0}, {4, 3, 0, 0}, {6, 0, 1, 0},
                                      {7, 1, 1, 0}, {5, 2, 0, 1}, {2, 0, 0, 1}, {8, 0, 1, 1}};

// bf16 twins of mfma43_acc / mfma43_first: operands are register pairs holding four bf16 each
__device__ __forceinline__ void mfma_bf16x4_acc(f32x2_t w, f32x2_t v, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(v));
}
__device__ __forceinline__ void mfma_bf16x4_first(f32x2_t w, f32x2_t v, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(w), "v"(v));
}

template <int CB, bool BF16>
__global__ void __launch_bounds__(256, 2)
deconv3x3_dma_kernel(const ConvArgs a) {
    using Cfg = DeconvCfg<CB>;
    constexpr int kChunkCh = BF16 ? 16 : 8;        // input channels per chunk = per 32-byte pixel cell
    constexpr int BN = Cfg::BN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* s_ctrl = reinterpret_cast<int*>(smem + 2 * Cfg::kBufFloats);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 15, g = lane >> 4;
    const int nchunks = a.Cin / kChunkCh;
    const int in_plane = a.H * a.W * 8;          // floats per plane (32 bytes per pixel either way)
    const int plane_bytes = in_plane * 4;

    struct Item { int frame, ntile, ty0, tx0; };
    auto decode = [&](int it) {
        const int4 v = a.items[__builtin_amdgcn_readfirstlane(it)];
        return Item{v.x, v.y, v.z, v.w};
    };

    int p_off[Cfg::kPatchPerWave];
    i32x4_t in_rsrc, w_rsrc;
    auto setup_patch = [&](const Item& it) {
#pragma unroll
        for (int k = 0; k < Cfg::kPatchPerWave; ++k) {
            const int j = wave + 4 * k;
            const int s = j * 64 + lane;
            const int cell = s >> 1;
            const int py = cell / Cfg::kPitch, px = cell - py * Cfg::kPitch;
            const int hf = (s & 1) ^ ((px >> 3) & 1);
            const int gy = it.ty0 - 1 + py, gx = it.tx0 - 1 + px;
            const bool ok = j < Cfg::kPatchInstr && py < Cfg::PH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            p_off[k] = ok ? ((gy * a.W + gx) * 8 + hf * 4) * 4 : kOob;
        }
        const float* in_item = a.in + (size_t)it.frame * a.in_frame_stride +
                               (size_t)(a.in_coff / kChunkCh) * in_plane;
        in_rsrc = make_rsrc(in_item, (unsigned)(nchunks * plane_bytes));
    };
    auto setup_w = [&](const Item& it) {
        const float* w_item = a.w + (size_t)it.ntile * nchunks * Cfg::kWFloats;
        w_rsrc = make_rsrc(w_item, (unsigned)(nchunks * Cfg::kWFloats * 4));
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    const unsigned dummy = lds0 + (2 * Cfg::kBufFloats + 4) * 4;
    const int w_voff = lane * 16;
    int pit = 0, pch = 0, wit = 0, wch = 0;
    constexpr int kCopies = Cfg::kPatchPerWave + Cfg::kWPerWave;     // 6 (some are dummies)
    auto copy_n = [&](int n, int img) {          // scalar operands only (issued between MFMAs)
        if (n < Cfg::kPatchPerWave) {
            const int j = wave + 4 * n;
            const unsigned dst = j < Cfg::kPatchInstr ? lds0 + (unsigned)(img * Cfg::kPatchFloats + j * 256) * 4
                                                      : dummy;
            blds16s(in_rsrc, p_off[n], pch * plane_bytes, dst);
        } else {
            const int j = wave + 4 * (n - Cfg::kPatchPerWave);
            const bool real = j < Cfg::kWInstr;
            const unsigned dst = real ? lds0 + (unsigned)(2 * Cfg::kPatchFloats + img * Cfg::kWFloats + j * 256) * 4
                                      : dummy;
            blds16s(w_rsrc, real ? w_voff : kOob, wch * (Cfg::kWFloats * 4) + j * 1024, dst);
        }
    };

    // the queue: one counter for the launch, or (a.xcd_counters, bf16 layers: conv_bf16_dma.h) one per group of
    // blocks that share an XCD, each walking its own eighth [q_lo, q_hi) of the table
    const bool grouped = a.xcd_counters != nullptr;
    const int vx = grouped ? (int)(blockIdx.x & 7) : 0;
    const int q_lo = grouped ? vx * (a.n_items / 8) + min(vx, a.n_items % 8) : 0;
    const int q_hi = grouped ? q_lo + a.n_items / 8 + (vx < a.n_items % 8 ? 1 : 0) : a.n_items;
    const int q_first = q_lo + (grouped ? ((int)gridDim.x - vx + 7) / 8 : (int)gridDim.x);      // item of ticket 0
    int* const q_counter = grouped ? a.xcd_counters + 16 * vx : a.counter;
    int comp_item = q_lo + (grouped ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    if (comp_item >= q_hi) return;
    int q0 = a.n_items;
    auto advance = [&](int& it, int& ch, bool patch) {
        if (it >= a.n_items) return;
        if (++ch == nchunks) {
            ch = 0;
            it = (it == comp_item) ? q0 : a.n_items;
            if (it < a.n_items) {
                if (patch) setup_patch(decode(it));
                else setup_w(decode(it));
            } else if (patch) {
                in_rsrc[2] = 0;
            } else {
                w_rsrc[2] = 0;
            }
        }
    };

    // lane constants: LDS byte offsets of the lane's pixel for the two column shifts, patch row
    // 4w (= input row 4w - 1); rows add an immediate
    const int hsel = g >> 1, sub = g & 1;
    int base_dj[2];
#pragma unroll
    for (int dj = 0; dj < 2; ++dj) {
        const int px = t + 1 - dj;
        base_dj[dj] = ((4 * wave) * Cfg::kPitch + px) * 32 + ((hsel ^ ((px >> 3) & 1)) * 16) + sub * 8;
    }
    const int w_lane = (g * 16 + t) * 8 * CB;      // bytes: [tap][g][t][cb][2]

    pit = wit = comp_item;
    setup_patch(decode(pit));
    setup_w(decode(wit));
#pragma unroll
    for (int n = 0; n < kCopies; ++n) copy_n(n, 0);
    advance(pit, pch, true);
    advance(wit, wch, false);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    const char* const xbase0 = reinterpret_cast<const char*>(smem) + base_dj[0];
    const char* const xbase1 = reinterpret_cast<const char*>(smem) + base_dj[1];
    const char* const wbase = reinterpret_cast<const char*>(smem) + 2 * Cfg::kPatchFloats * 4 + w_lane;

    // One step: the 144 MFMAs of chunk k (inputs and weights from images PAR), the copies of chunk
    // k+1 into images PAR ^ 1 between them.
    typedef float wfrag_t __attribute__((ext_vector_type(2 * CB)));     // a tap's fragment: CB x 2 channels
    constexpr int kTapBytes = 4 * 16 * 8 * CB;
    auto step = [&](auto par, auto first, f32x4 (&acc)[4][4][CB], int comp_ch) {
        constexpr int PAR = decltype(par)::value;
        constexpr bool FIRST = decltype(first)::value;       // the item's first chunk
        if (comp_ch == 0 && tid == 0) {
            const int t = q_first + atomicAdd(q_counter, 1);
            s_ctrl[0] = t < q_hi ? t : a.n_items;
        }
        // inputs: patch rows 4w .. 4w + 4 (input rows 4w - 1 .. 4w + 3) x column shifts 0, 1
        f32x2_t xin[5][2];
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            xin[pr][0] = *reinterpret_cast<const f32x2_t*>(xbase0 + PAR * Cfg::kPatchFloats * 4 +
                                                           pr * Cfg::kPitch * 32);
            xin[pr][1] = *reinterpret_cast<const f32x2_t*>(xbase1 + PAR * Cfg::kPatchFloats * 4 +
                                                           pr * Cfg::kPitch * 32);
        }
        // weight fragments (one b128 = both channel blocks of a tap) two taps ahead
        const char* sW = wbase + PAR * Cfg::kWFloats * 4;
        wfrag_t wq[3];
        wq[0] = *reinterpret_cast<const wfrag_t*>(sW + kDeconvTaps[0].tap * kTapBytes);
        wq[1] = *reinterpret_cast<const wfrag_t*>(sW + kDeconvTaps[1].tap * kTapBytes);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            if (s + 2 < 9)
                wq[(s + 2) % 3] = *reinterpret_cast<const wfrag_t*>(sW + kDeconvTaps[s + 2].tap * kTapBytes);
            __builtin_amdgcn_sched_barrier(0);
            const wfrag_t w = wq[s % 3];
            const int cls = kDeconvTaps[s].cls, di = kDeconvTaps[s].di, dj = kDeconvTaps[s].dj;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2_t x = xin[r + 1 - di][dj];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    if constexpr (BF16) {
                        const f32x2_t wv = {w[cb * 2 + 0], w[cb * 2 + 1]};       // four bf16: channels 4g .. 4g+3
                        if (FIRST && s < 4) mfma_bf16x4_first(wv, x, acc[r][cls][cb]);
                        else mfma_bf16x4_acc(wv, x, acc[r][cls][cb]);
                    } else {
                        if (FIRST && s < 4) mfma43_first(w[cb * 2 + 0], x[0], acc[r][cls][cb]);
                        else mfma43_acc(w[cb * 2 + 0], x[0], acc[r][cls][cb]);
                        mfma43_acc(w[cb * 2 + 1], x[1], acc[r][cls][cb]);
                    }
                }
            }
            if (s < kCopies) copy_n(s, PAR ^ 1);     // scalar + VMEM issue in the MFMAs' shadow
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_s_barrier();
        if (comp_ch == 0) q0 = s_ctrl[0];
        advance(pit, pch, true);
        advance(wit, wch, false);
    };

    while (comp_item < a.n_items) {
        f32x4 acc[4][4][CB];
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        step(T0{}, std::true_type{}, acc, 0);
        step(T1{}, std::false_type{}, acc, 1);
        for (int comp_ch = 2; comp_ch < nchunks; comp_ch += 2) {     // Cin / 8 is even
            step(T0{}, std::false_type{}, acc, comp_ch);
            step(T1{}, std::false_type{}, acc, comp_ch + 1);
        }
        // the asm MFMAs are opaque to the compiler's hazard recogniser (see wino43_kernel.h)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                asm volatile("s_nop 15\n\ts_nop 3"
                             : "+a"(acc[r][0][cb]), "+a"(acc[r][1][cb]), "+a"(acc[r][2][cb]), "+a"(acc[r][3][cb]));
        // ---- epilogue: batch-norm + ReLU, the four parity classes of every input pixel ------------
        {
            // (arguments from the kernarg segment again, packed multiply / add, one cell offset per lane and no
            //  per-pixel tests on interior tiles: see the epilogue of wino3x3_f32_kernel)
            typedef const ConvArgs __attribute__((address_space(4))) KernArgs;
            KernArgs* ep_ = (KernArgs*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ep_));
            KernArgs& e = *ep_;
            const Item it = decode(comp_item);
            float* out = e.out + (size_t)it.frame * e.out_frame_stride;
            const long long plane = (long long)(2 * e.H) * (2 * e.W) * 8;
            const int x = it.tx0 + t;
            const bool interior = it.ty0 + 16 <= e.H && it.tx0 + 16 <= e.W;
            const float relu_floor = e.relu ? 0.0f : -__builtin_inff();
            const int row8 = 2 * e.W * 8;                 // floats per output row of a plane
            // the lane's cell: output pixel (2 y0, 2 x) of its first row, in floats from the plane pair's base
            const int y0 = it.ty0 + 4 * wave;
            const int cell = (2 * y0 * 2 * e.W + 2 * x) * 8;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int c0 = it.ntile * BN + cb * 16 + 4 * g;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(e.scale + c0);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(e.shift + c0);
                const f32x2_t sc_lo = {sc[0], sc[1]}, sc_hi = {sc[2], sc[3]}, sh_lo = {sh[0], sh[1]}, sh_hi = {sh[2], sh[3]};
                // CB8 fp32: plane (channel >> 3), float (channel & 7) of the pixel's 8; CB16 bf16: plane
                // (channel >> 4), the lane's four channels are floats ((channel & 15) >> 1) .. + 1
                float* obase = BF16 ? out + (size_t)((e.out_coff + c0) >> 4) * plane + (((e.out_coff + c0) & 15) >> 1)
                                    : out + (size_t)((e.out_coff + c0) >> 3) * plane + ((e.out_coff + c0) & 7);
                float* cellp = obase + cell;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = interior || (y0 + r < e.H && x < e.W);
#pragma unroll
                    for (int cls = 0; cls < 4; ++cls) {
                        const f32x4 m = acc[r][cls][cb];
                        const f32x2_t lo = f32x2_t{m[0], m[1]} * sc_lo + sh_lo, hi = f32x2_t{m[2], m[3]} * sc_hi + sh_hi;
                        const f32x4 v = {fmaxf(lo[0], relu_floor), fmaxf(lo[1], relu_floor), fmaxf(hi[0], relu_floor),
                                         fmaxf(hi[1], relu_floor)};
                        float* dst = cellp + (2 * r + (cls >> 1)) * row8 + (cls & 1) * 8;
                        if (ok) {
                            if constexpr (BF16)     // round to nearest even, as every map of the bf16 path
                                *reinterpret_cast<f32x2_t*>(dst) = f32x2_t{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
                            else
                                *reinterpret_cast<f32x4*>(dst) = v;
                        }
                    }
                }
            }
        }
        comp_item = q0;
    }
}

}  // namespace dodt
