// 3x3 stride-1 convolution on the bf16 MFMA (v_mfma_f32_32x32x16_bf16) over CB16 bf16 maps --
// round 2's kernel for the bf16 conv path (BASELINE.json configs[2]); it replaces the fp32
// kernel template's BF16 instantiation (conv_kernels.h), whose fragment reads and register
// staging paced the loop instead of the matrix pipe.
//
// At bf16 an MFMA takes 32 cycles for the work an fp32 kernel spends 4 x 64 on, so the
// kernel is shaped by bytes, not by multiplications:
//   * staging: the halo'd patch of a 16-channel chunk and the chunk's weights [tap][h][n][16 B]
//     go global -> LDS by buffer_load_dwordx4 ... lds (LDS-DMA: no staging registers, no
//     ds_write pass; out-of-image pixels are out of the descriptor's range and read zeros),
//     double buffered, one barrier per chunk, two workgroups per CU so that one's copies and
//     barrier hide behind the other's MFMAs.
//   * LDS reads: a wave owns MT rows of 32 pixels and 32 NT output channels.  For one kx it
//     reads the MT + 2 patch rows its three ky taps share (row-sliding reuse: 3 (MT + 2)
//     instead of 9 MT pixel fragments per chunk) and 3 NT weight fragments; 9 MT NT MFMAs.
//   * the patch image is conflict-free: a pixel cell is 32 bytes (two 16-byte k-halves); the
//     halves of the cells with (column >> 3) & 1 are swapped -- by permuting the SOURCE address
//     of the lane-linear LDS-DMA -- so that the 16 lanes of one ds_read_b128 group
//     ({0-3,12-15,20-27} / {4-11,16-19,28-31}, columns 8 or 24 apart) hit 16 different bank
//     columns for every kx.
// The accumulator layout (weights = A operand: rows = output channels in the permuted order of
// group_channel<true>, pixels = B operand) is the fp32 template's, so its epilogue (batch-norm
// + ReLU, bf16 CB16 / fp32 NHWC stores, fused 2x2 max pool, fused 1x1 bottleneck) is reused
// unchanged (store_tile / pool_tile of conv_kernels.h).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "conv_kernels.h"
#include "wino_kernels.h"   // blds16, make_rsrc, kOob

namespace dodt {

// Epilogue of the kernels of this file for one item: batch-norm + ReLU, stores, fused 2x2 pool / 1x1 bottleneck -- what
// store_tile / pool_tile of conv_kernels.h do for the other kernels, with the per-channel parameters (s_par: scale[32],
// shift[32], bottleneck weights[32]) read from LDS in one batch (an item is 1-2 us here: eight dependent parameter loads per
// tile were twice the MFMAs' time) and the pool's horizontal neighbour taken by DPP.
// MODE 0: CB16 bf16 map, 1: the same + its 2x2 max pool, 2: NHWC fp32 (+ bottleneck).  Returns nothing; `mid` (optional
// diagnostic) receives s_memtime between the stores and the pool.
struct StreamTile { int frame, ty0, tx0; };
// p_scale / p_shift / p_bw: the channel tile's batch-norm scale, shift and bottleneck weights (LDS or global memory: read in
// one batch either way); c0: the tile's first channel; acc: tile mt of channel tile NTI is acc[mt * NT + NTI].
template <int MODE, int MT, int NT = 1, int NTI = 0>
__device__ __forceinline__ void stream_epilogue(const ConvArgs& a, const float* p_scale, const float* p_shift, const float* p_bw,
                                                f32x16 (&acc_all)[MT * NT], const StreamTile cur, int c0, int wave, int li, int lh,
                                                bool stamp, int* stamps, float* s_tile = nullptr) {
    auto acc = [&](int mt) -> f32x16& { return acc_all[mt * NT + NTI]; };
    const float floor_v = a.relu ? 0.0f : -3.0e38f;
    {
        f32x4 sc[4], sh[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = group_channel<true>(g, lh);
            sc[g] = *reinterpret_cast<const f32x4*>(p_scale + c);
            sh[g] = *reinterpret_cast<const f32x4*>(p_shift + c);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc(mt)[r] = fmaxf(acc(mt)[r] * sc[r >> 2][r & 3] + sh[r >> 2][r & 3], floor_v);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int out_rows = a.H - a.out_y0;
    const long long plane = (long long)out_rows * a.W * 8;
    float* out = a.out + (size_t)cur.frame * a.out_frame_stride;
    const int x = cur.tx0 + li;
    f32x4 bw[4];      // (read when the scales are dead: the 64-channel instance has 144 registers of weights)
    if (MODE == 2 && a.bneck_w) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bw[g] = *reinterpret_cast<const f32x4*>(p_bw + group_channel<true>(g, lh));
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int y = cur.ty0 + wave * MT + mt;
        const bool ok = y < a.H && x < a.W && y >= a.out_y0 && !(a.debug & 1);      // (tools/: 1 = no stores)
        const size_t px = (size_t)(y - a.out_y0) * a.W + x;
        if (MODE == 2) {
            if (s_tile && a.out_ld == 32 && a.out_coff == 0 && c0 == 0) {
                // whole 128-byte pixels per store: the tile goes through the wave's own LDS scratch (32 pixels x 36 dwords)
                // and comes back with eight lanes per pixel, so that a store instruction writes 1 KB of consecutive bytes
                // (the accumulator layout gives a lane 16 bytes of every 128: 64 sixteen-byte pieces per instruction --
                // pyramid_fusion1 spent a third of its time on them)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(s_tile + li * 36 + group_channel<true>(g, lh)) =
                        f32x4{acc(mt)[4 * g], acc(mt)[4 * g + 1], acc(mt)[4 * g + 2], acc(mt)[4 * g + 3]};
                const int lane = lh * 32 + li, sub = lane & 7;
                const bool row_ok = y < a.H && y >= a.out_y0 && !(a.debug & 1);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int pp = 8 * m + (lane >> 3);                  // pixel of the tile row
                    const f32x4 v = *reinterpret_cast<const f32x4*>(s_tile + pp * 36 + 4 * sub);
                    if (row_ok && cur.tx0 + pp < a.W)
                        *reinterpret_cast<f32x4*>(out + ((size_t)(y - a.out_y0) * a.W + cur.tx0 + pp) * 32 + 4 * sub) = v;
                }
            } else if (ok) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(out + px * a.out_ld + a.out_coff + c0 + group_channel<true>(g, lh)) =
                        f32x4{acc(mt)[4 * g], acc(mt)[4 * g + 1], acc(mt)[4 * g + 2], acc(mt)[4 * g + 3]};
            }
            if (a.bneck_w) {
                float dot = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) dot += acc(mt)[r] * bw[r >> 2][r & 3];
                dot += __shfl_xor(dot, 32, 64);          // the other 16 channels of this pixel live in lane ^ 32
                if (ok && lh == 0)
                    a.bneck_out[(size_t)cur.frame * a.bneck_frame_stride + px] =
                        fmaxf(dot * a.bneck_scale + a.bneck_shift, 0.0f);
            }
        } else if (ok) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp)       // groups (2 gp, 2 gp + 1): 8 consecutive channels = one 16-byte store
                *reinterpret_cast<f32x4*>(out + (size_t)((a.out_coff + c0 + 16 * gp) >> 4) * plane + px * 8 + 4 * lh) =
                    f32x4{pack_bf16(acc(mt)[8 * gp], acc(mt)[8 * gp + 1]), pack_bf16(acc(mt)[8 * gp + 2], acc(mt)[8 * gp + 3]),
                          pack_bf16(acc(mt)[8 * gp + 4], acc(mt)[8 * gp + 5]), pack_bf16(acc(mt)[8 * gp + 6], acc(mt)[8 * gp + 7])};
        }
    }
    if (stamp) stamps[3] = (int)__builtin_amdgcn_s_memtime();
    if (MODE == 1) {
        // VALID 2x2 pool of the activated rows (mt, mt + 1): vertical neighbour = the same lane of the next row,
        // horizontal = lane ^ 1 (DPP quad_perm [1,0,3,2]); lanes at even (y, x) store
        const int OW = a.W >> 1;
        const long long pplane = (long long)(a.H >> 1) * OW * 8;
#pragma unroll
        for (int mt = 0; mt < MT; mt += 2) {
            const int y = cur.ty0 + wave * MT + mt;
            const bool writer = y < a.H && x < a.W && !(y & 1) && !(x & 1) && y + 1 < a.H && x + 1 < a.W && !(a.debug & 1);
            float m[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float t = fmaxf(acc(mt)[r], acc(mt + 1)[r]);
                const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                    0, __builtin_bit_cast(int, t), 0xB1, 0xf, 0xf, false));
                m[r] = fmaxf(t, o);
            }
            if (writer) {
                float* base = a.pool_out + (size_t)cur.frame * a.pool_frame_stride +
                              ((size_t)(y >> 1) * OW + (x >> 1)) * 8 + 4 * lh;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp)
                    *reinterpret_cast<f32x4*>(base + (size_t)((c0 >> 4) + gp) * pplane) =
                        f32x4{pack_bf16(m[8 * gp], m[8 * gp + 1]), pack_bf16(m[8 * gp + 2], m[8 * gp + 3]),
                              pack_bf16(m[8 * gp + 4], m[8 * gp + 5]), pack_bf16(m[8 * gp + 6], m[8 * gp + 7])};
            }
        }
    }
}

template <int MT, int NT, int S>
struct Bf16DmaCfg {
    static_assert(S >= 2 && S <= 4, "ring of 2..4 chunk images");
    static constexpr int TW = 32, TH = 4 * MT, BN = 32 * NT;
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int kPatchSlots = PH * PW * 2;                 // 16-byte slots
    static constexpr int kPatchInstr = (kPatchSlots + 63) / 64;     // 1 KB wave-level copies
    static constexpr int kPatchFloats = kPatchInstr * 256;
    static constexpr int kWFloats = 9 * 2 * BN * 4;                 // [tap][h][n][16 B]
    static constexpr int kWInstr = kWFloats / 256;
    static constexpr int kBufFloats = kPatchFloats + kWFloats;
    static constexpr int kLdsBytes = S * kBufFloats * 4 + 16 + 1024;
    static constexpr int kPatchPerWave = (kPatchInstr + 3) / 4;
    static constexpr int kWPerWave = (kWInstr + 3) / 4;
    static constexpr int kCopies = kPatchPerWave + kWPerWave;
    static_assert(kWFloats % 256 == 0, "weight image is a whole number of 1 KB copies");
    static_assert((S - 2) * kCopies <= 63, "vmcnt is six bits");
    static_assert(MT % 2 == 0, "the fused pool pairs the rows of a wave");
};

// (MT = 2: 8-row tiles, 41 KB, three workgroups per CU -- for the small maps of the deepest level, whose 16-row
//  tiles do not give every CU two workgroups)
template <int MT, int NT, int S>
__global__ void __launch_bounds__(256, (S == 2 ? (MT == 2 ? 3 : 2) : 1))
conv3x3_bf16_dma_kernel(const ConvArgs a) {
    using Cfg = Bf16DmaCfg<MT, NT, S>;
    constexpr int BN = Cfg::BN, PW = Cfg::PW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // ring of S chunk images: [patch | weights] each
    int* s_ctrl = reinterpret_cast<int*>(smem + S * Cfg::kBufFloats);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int nchunks = a.Cin / 16;
    const int in_plane = a.H * a.W * 8;          // floats per CB16 plane (32 B per pixel)
    const int plane_bytes = in_plane * 4;

    struct Item { int frame, ntile, ty0, tx0; };
    auto decode = [&](int it) {
        const int4 v = a.items[__builtin_amdgcn_readfirstlane(it)];
        return Item{v.x, v.y, v.z, v.w};
    };

    int p_off[Cfg::kPatchPerWave];
    i32x4_t in_rsrc, w_rsrc;
    auto setup = [&](const Item& it) {
#pragma unroll
        for (int k = 0; k < Cfg::kPatchPerWave; ++k) {
            const int j = wave + 4 * k;
            const int s = j * 64 + lane;
            const int q = s >> 1;
            const int py = q / PW, px = q - py * PW;
            const int half = (s & 1) ^ ((px >> 3) & 1);
            const int gy = it.ty0 - 1 + py, gx = it.tx0 - 1 + px;
            const bool ok = j < Cfg::kPatchInstr && py < Cfg::PH && gy >= 0 && gy < a.H &&
                            gx >= 0 && gx < a.W;
            p_off[k] = ok ? ((gy * a.W + gx) * 8 + half * 4) * 4 : kOob;
        }
        const float* in_item = a.in + (size_t)it.frame * a.in_frame_stride +
                               (size_t)(a.in_coff / 16) * in_plane;
        in_rsrc = make_rsrc(in_item, (unsigned)(nchunks * plane_bytes));
        const float* w_item = a.w + (size_t)it.ntile * nchunks * Cfg::kWFloats;
        w_rsrc = make_rsrc(w_item, (unsigned)(nchunks * Cfg::kWFloats * 4));
    };
    // Work items known to this workgroup: k0 is being computed, k1 follows it and k2 follows k1
    // (k2 is claimed from the atomic counter during the first step of every item, in time for
    // the copy cursor, which runs S - 1 <= 3 chunks = at most two items ahead of the MFMAs).
    // The queue: one counter for the launch, or (a.xcd_counters) one per group of blocks that share an XCD, each
    // group walking its own contiguous eighth [q_lo, q_hi) of the table.  Ticket t of a queue is item q_first + t;
    // tickets beyond the range mean "nothing left" (a.n_items).
    const bool grouped = a.xcd_counters != nullptr;
    const int vx = grouped ? (int)(blockIdx.x & 7) : 0;
    // (eighths that differ by at most one item, the larger ones first -- like the groups' block counts, so that a
    //  launch with one item per block stays one round in every group)
    const int q_lo = grouped ? vx * (a.n_items / 8) + min(vx, a.n_items % 8) : 0;
    const int q_hi = grouped ? q_lo + a.n_items / 8 + (vx < a.n_items % 8 ? 1 : 0) : a.n_items;
    const int q_blocks = grouped ? ((int)gridDim.x - vx + 7) / 8 : (int)gridDim.x;    // blocks drawing from this queue
    const int q_first = q_lo + q_blocks;                                              // item of ticket 0
    int* const q_counter = grouped ? a.xcd_counters + 16 * vx : a.counter;
    auto ticket_item = [&](int t) { return q_first + t < q_hi ? q_first + t : a.n_items; };
    int k0 = q_lo + (grouped ? (int)(blockIdx.x >> 3) : (int)blockIdx.x), k1 = a.n_items, k2 = a.n_items;
    if (k0 >= q_hi) return;
    // the successor's ticket is drawn now and read behind the prologue's copies (one barrier, and
    // the atomic's round trip hides behind the first fills)
    int pend = 0;
    if (tid == 0) pend = atomicAdd(q_counter, 1);
    // (a.debug & 128, tools/: start | first step | end on the chip-wide 100 MHz clock and the item
    //  count of every workgroup, 4 words each from counter_base[256])
    const bool wgstamp = (a.debug & 128) && tid == 0;
    int* wgs = a.counter_base + 256 + 4 * blockIdx.x;
    int n_done = 0;
    if (wgstamp) {
        wgs[0] = (int)__builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 1) a.counter_base[250] = (int)__builtin_amdgcn_s_memtime();
    }
    int cslot = 0, cch = 0;
    bool cur_live = true;            // the cursor points at a real chunk
    // copy n of this wave for the cursor's chunk into image b; scalar operands only: the copies
    // are issued between the MFMAs, where a vector instruction would wait for the SIMD's matrix
    // work
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    const unsigned dummy_addr = lds0 + (S * Cfg::kBufFloats + 4) * 4;
    const int w_voff = lane * 16;
    auto copy_n = [&](int n, int b) {
        const unsigned img = lds0 + (unsigned)(b * Cfg::kBufFloats) * 4;
        if (n < Cfg::kPatchPerWave) {            // compile-time
            const int j = wave + 4 * n;
            blds16s(in_rsrc, p_off[n], cch * plane_bytes, j < Cfg::kPatchInstr ? img + j * 1024 : dummy_addr);
        } else {
            const int j = wave + 4 * (n - Cfg::kPatchPerWave);
            const bool real = j < Cfg::kWInstr;
            blds16s(w_rsrc, real ? w_voff : kOob, cch * (Cfg::kWFloats * 4) + j * 1024,
                    real ? img + Cfg::kPatchFloats * 4 + j * 1024 : dummy_addr);
        }
    };
    auto copies = [&](int b) {
#pragma unroll
        for (int n = 0; n < Cfg::kCopies; ++n) copy_n(n, b);
    };
    auto advance = [&]() {
        if (!cur_live) return;
        if (++cch == nchunks) {
            cch = 0;
            ++cslot;
            const int nxt = cslot == 1 ? k1 : cslot == 2 ? k2 : a.n_items;
            if (nxt < a.n_items) {
                setup(decode(nxt));
            } else {
                cur_live = false;
                in_rsrc[2] = 0;       // nothing left: the remaining copies read zeros
                w_rsrc[2] = 0;
            }
        }
    };

    // lane constants: float offset of (row 0, column li + kx) with the half swap; weights
    int col_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int c = li + kx;
        col_off[kx] = c * 8 + ((lh ^ ((c >> 3) & 1)) * 4);
    }
    const int row0 = wave * MT * PW * 8;
    const int w_lane = (lh * BN + li) * 4;

    // ---- prologue: chunks 0 .. S-2 of the flat (item, chunk) sequence -----------------------
    setup(decode(k0));
#pragma unroll
    for (int b = 0; b < S - 1; ++b) {
        copies(b);
        advance();
    }
    if (tid == 0) s_ctrl[1] = ticket_item(pend);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_s_barrier();
    k1 = s_ctrl[1];           // (word 1: word 0 is rewritten by thread 0 at the top of step 0)
    if (wgstamp) wgs[1] = (int)__builtin_amdgcn_s_memrealtime();

    static_assert(S == 2, "the chunk loop below is unrolled for two images");
    // One step = the 9 MT NT MFMAs of chunk k from image PAR (accumulating in place in AGPRs: inline
    // asm, see wino43_kernel.h) with the copies of chunk k+1 into image PAR ^ 1 issued one per
    // tap in their shadow; patch rows of the next kx and weight fragments of the next tap are read
    // one group ahead (pinned by scheduling barriers).
    auto mfma_acc = [&](const f32x4& w, const f32x4& x, f32x16& c) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
    };
    auto mfma_first = [&](const f32x4& w, const f32x4& x, f32x16& c) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(w), "v"(x));
    };
    int k_stamp = 0;
    auto step = [&](auto par, auto first, f32x16 (&acc)[MT * NT], int comp_ch) {
        constexpr int PAR = decltype(par)::value;
        constexpr bool FIRST = decltype(first)::value;
        // (a.debug & 32, tools/: s_memtime stamps of one wave's first 12 steps, counter_base[32..])
        const bool stamp = (a.debug & 32) && blockIdx.x == 1 && tid == 0 && k_stamp < 12;
        int* stamps = a.counter_base + 32 + (k_stamp < 12 ? k_stamp : 0) * 6;
        if (stamp) { stamps[0] = (int)__builtin_amdgcn_s_memtime(); stamps[1] = stamps[0]; stamps[2] = stamps[0]; }
        // (the ticket is drawn here and published at the end of the step: the atomic's round trip hides behind the MFMAs)
        int ticket = 0;
        if (comp_ch == 0 && tid == 0) ticket = atomicAdd(q_counter, 1);
        const float* sP = smem + PAR * Cfg::kBufFloats + row0;
        const float* sW = smem + PAR * Cfg::kBufFloats + Cfg::kPatchFloats + w_lane;
        f32x4 x[MT + 2], w[2][NT];
#pragma unroll
        for (int r = 0; r < MT + 2; ++r)
            x[r] = *reinterpret_cast<const f32x4*>(sP + r * PW * 8 + col_off[0]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) w[0][nt] = *reinterpret_cast<const f32x4*>(sW + nt * 32 * 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g9 = 0; g9 < 9; ++g9) {             // group g9 = kx * 3 + ky
            const int kx = g9 / 3, ky = g9 % 3;
            if (g9 + 1 < 9) {
                const int kx1 = (g9 + 1) / 3, ky1 = (g9 + 1) % 3;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    w[(g9 + 1) & 1][nt] = *reinterpret_cast<const f32x4*>(
                        sW + ((ky1 * 3 + kx1) * 2 * BN + nt * 32) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (FIRST && g9 == 0) mfma_first(w[g9 & 1][nt], x[mt + ky], acc[mt * NT + nt]);
                    else mfma_acc(w[g9 & 1][nt], x[mt + ky], acc[mt * NT + nt]);
                }
            if (ky == 2 && kx < 2) {     // the next kx's rows (the partner wave covers their latency)
#pragma unroll
                for (int r = 0; r < MT + 2; ++r)
                    x[r] = *reinterpret_cast<const f32x4*>(sP + r * PW * 8 + col_off[kx + 1]);
            }
            if (g9 < Cfg::kCopies) copy_n(g9, PAR ^ 1);
            if (g9 == 8) {
#pragma unroll
                for (int n = 9; n < Cfg::kCopies; ++n) copy_n(n, PAR ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (stamp) stamps[3] = (int)__builtin_amdgcn_s_memtime();
        if (comp_ch == 0 && tid == 0) s_ctrl[0] = ticket_item(ticket);
        __builtin_amdgcn_s_waitcnt(0);        // the copies of chunk k+1 have landed
        if (stamp) stamps[4] = (int)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        if (stamp) stamps[5] = (int)__builtin_amdgcn_s_memtime();
        ++k_stamp;
        if (comp_ch == 0) k2 = s_ctrl[0];     // the item claimed in this step: third in line
        advance();
    };
    while (k0 < a.n_items) {
        f32x16 acc[MT * NT];
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        step(T0{}, std::true_type{}, acc, 0);
        step(T1{}, std::false_type{}, acc, 1);
        for (int comp_ch = 2; comp_ch < nchunks; comp_ch += 2) {      // Cin / 16 is even
            step(T0{}, std::false_type{}, acc, comp_ch);
            step(T1{}, std::false_type{}, acc, comp_ch + 1);
        }
        // the asm MFMAs are opaque to the hazard recogniser: 16-pass results need 18 wait states
#pragma unroll
        for (int k = 0; k < MT * NT; ++k) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[k]));
        if constexpr (NT == 1) {
            // ---- epilogue: BN + ReLU, stores, fused pool / bottleneck (stream_epilogue above: a channel tile's parameters in one
            // batch of loads -- store_tile's two loads and a wait per group of four channels, sixteen times per item, were a
            // third of a mid layer's item time)
            const Item cur = decode(k0);
            const StreamTile cur_t{cur.frame, cur.ty0, cur.tx0};
            const bool pool = a.pool_out != nullptr;
            auto channel_tile = [&](auto nt_c) {
                constexpr int NTI = decltype(nt_c)::value;
                const int c0 = cur.ntile * BN + NTI * 32;
                const float* bw = a.bneck_w ? a.bneck_w + c0 : a.scale + c0;      // (read only with a bottleneck)
                if (pool) stream_epilogue<1, MT, NT, NTI>(a, a.scale + c0, a.shift + c0, bw, acc, cur_t, c0, wave, li, lh, false, nullptr);
                else if (!a.out_nhwc) stream_epilogue<0, MT, NT, NTI>(a, a.scale + c0, a.shift + c0, bw, acc, cur_t, c0, wave, li, lh, false, nullptr);
                else stream_epilogue<2, MT, NT, NTI>(a, a.scale + c0, a.shift + c0, bw, acc, cur_t, c0, wave, li, lh, false, nullptr);
                __builtin_amdgcn_sched_barrier(0);
            };
            channel_tile(std::integral_constant<int, 0>{});
            if constexpr (NT == 2) channel_tile(std::integral_constant<int, 1>{});
        } else {
            // (two channel tiles per wave: 128 accumulators leave no room for a batch of parameters -- store_tile's form)
            // ---- epilogue (as in conv3x3_mfma_kernel): BN + ReLU, stores, fused pool / bottleneck ---
            const Item cur = decode(k0);
            float* out = a.out + (size_t)cur.frame * a.out_frame_stride;
            const int out_rows = a.H - a.out_y0;
            const long long plane = (long long)out_rows * a.W * 8;
            const bool pool = a.pool_out != nullptr;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int y = cur.ty0 + wave * MT + mt;
                const int x = cur.tx0 + li;
                const bool ok = y < a.H && x < a.W && y >= a.out_y0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int c0 = cur.ntile * BN + nt * 32;
                    if (pool)
                        store_tile<true, true, true, false>(a, out, acc[mt * NT + nt], c0, lh,
                                                            y - a.out_y0, x, a.W, plane, ok, cur.frame);
                    else if (!a.out_nhwc)
                        store_tile<false, true, true, false>(a, out, acc[mt * NT + nt], c0, lh,
                                                             y - a.out_y0, x, a.W, plane, ok, cur.frame);
                    else
                        store_tile<false, true, false>(a, out, acc[mt * NT + nt], c0, lh, y - a.out_y0,
                                                       x, a.W, plane, ok, cur.frame);
                    // one accumulator tile at a time: they live in AGPRs and pass through VGPRs here
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (pool) {
#pragma unroll
                for (int mt = 0; mt < MT; mt += 2) {
                    const int y = cur.ty0 + wave * MT + mt;
                    const int x = cur.tx0 + li;
                    const bool ok = y < a.H && x < a.W;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                    {
                        pool_tile<32, true, true, false>(a, acc[mt * NT + nt], acc[(mt + 1) * NT + nt],
                                                         cur.ntile * BN + nt * 32, lh, y, x, cur.frame, ok);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        // next item: the queue moves up, the cursor's slot with it
        k0 = k1;
        k1 = k2;
        k2 = a.n_items;
        --cslot;
        ++n_done;
    }
    if (wgstamp) {
        wgs[2] = (int)__builtin_amdgcn_s_memrealtime();
        wgs[3] = n_done;
        if (blockIdx.x == 1) a.counter_base[251] = (int)__builtin_amdgcn_s_memtime();
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Round 4, late: the level-1 layers (32 output channels, 32 or 64 input channels: conv1_2, pyramid_fusion1 -- 0.25 of
// the bf16 stacks' time) are HBM-bound at ~3 TB/s with the kernel above: it has ONE 20 KB chunk per workgroup on its
// way (two per CU = 40 KB against the ~60 KB per CU that 5 TB/s times the loaded memory latency asks for), and a
// deeper ring does not help it, because a wave's epilogue stores sit between its copies in the same in-order counter
// (vmcnt): the first wait after an epilogue drains everything older.  This kernel separates the roles instead:
//   * wave 3 is the PRODUCER: it alone issues the LDS-DMA copies (a ring of S patch images, S - 1 chunks ahead of the
//     MFMAs), has nothing else in its vmcnt queue, and so can wait for exactly "chunk k + 1 has landed"
//     (vmcnt((S - 2) x copies per chunk)) before the barrier that ends step k;
//   * waves 0-2 are the CONSUMERS: MT rows of 32 pixels each (tile = 3 MT rows x 32 pixels), weights for ALL chunks in
//     registers (9 NCH fragments of 16 bytes per lane, loaded once per persistent workgroup: no weight image, no
//     weight copies, no weight reads from LDS), epilogue as above;
//   * the work queue is the grouped one above; the tickets are drawn by lane 0 of wave 0 (whose vmcnt queue holds its
//     stores only) and published through a 16-entry ring in LDS, D items ahead.
// MT = 2, S = 8: 6 x 32 pixel tiles, 9 KB images, 72 KB of LDS, two workgroups per CU, 7 x 8.7 KB in flight each.
template <int MT, int NCH, int S>
struct Bf16StreamCfg {
    static constexpr int TW = 32, TH = 3 * MT, BN = 32;
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int kPatchSlots = PH * PW * 2;                 // 16-byte slots
    static constexpr int kCopies = (kPatchSlots + 63) / 64;         // 1 KB copies per chunk, all by the producer
    static constexpr int kPatchFloats = kCopies * 256;
    static constexpr int kWFloats = 9 * 2 * BN * 4;                 // a chunk's weights in global memory: [tap][h][n][16 B]
    static constexpr int WR = NCH <= 3 ? NCH : 3;                   // chunks whose weights stay in registers (36 each); the others' in LDS
    static constexpr int kWLdsFloats = (NCH - WR) * kWFloats;
    static constexpr int kTileFloats = 32 * 36;                     // a wave's scratch for the NHWC stores (stream_epilogue)
    static constexpr int kLdsBytes = S * kPatchFloats * 4 + 64 + 384 + kWLdsFloats * 4 +
                                     (NCH == 4 ? 3 * kTileFloats * 4 : 0);   // images | item ring | scale, shift, bottleneck weights | weights | scratch
    static constexpr int kAhead = (NCH - 1 + S) / NCH + 3;          // D: the prologue publishes ring entries 0 .. D - 2, item n >= 1 entry n + D - 2
    static_assert((S - 2) * kCopies <= 63, "vmcnt is six bits");
    static_assert(MT % 2 == 0, "the fused pool pairs the rows of a wave");
    static_assert(kAhead <= 8, "item ring");
};

template <int MT, int NCH, int S>
__global__ void __launch_bounds__(256, 2)
conv3x3_bf16_stream_kernel(const ConvArgs a) {
    using Cfg = Bf16StreamCfg<MT, NCH, S>;
    constexpr int BN = Cfg::BN, PW = Cfg::PW, D = Cfg::kAhead, KC = Cfg::kCopies;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* s_q = reinterpret_cast<int*>(smem + S * Cfg::kPatchFloats);      // item ring: entry n & 15 = this workgroup's n-th item

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave == 3;
    const int li = lane & 31, lh = lane >> 5;
    const int in_plane = a.H * a.W * 8;          // floats per CB16 plane (32 B per pixel)
    const int plane_bytes = in_plane * 4;

    // item -> tile, by arithmetic on the table's order (frame, tile row, tile column; one channel tile): the producer must
    // not load from memory (a vector load's wait would drain its copy ring), and the consumers need not
    struct Item { int frame, ty0, tx0; };
    const int tiles_per_frame = a.tiles_x * a.tiles_y;
    auto decode = [&](int it) {
        it = __builtin_amdgcn_readfirstlane(it);
        const int f = it / tiles_per_frame, r = it - f * tiles_per_frame;
        const int ty = r / a.tiles_x;
        return Item{f, ty * Cfg::TH, (r - ty * a.tiles_x) * Cfg::TW};
    };
    // the grouped queue of conv3x3_bf16_dma_kernel
    const bool grouped = a.xcd_counters != nullptr;
    const int vx = grouped ? (int)(blockIdx.x & 7) : 0;
    const int q_lo = grouped ? vx * (a.n_items / 8) + min(vx, a.n_items % 8) : 0;
    const int q_hi = grouped ? q_lo + a.n_items / 8 + (vx < a.n_items % 8 ? 1 : 0) : a.n_items;
    const int q_blocks = grouped ? ((int)gridDim.x - vx + 7) / 8 : (int)gridDim.x;
    const int q_first = q_lo + q_blocks;
    int* const q_counter = grouped ? a.xcd_counters + 16 * vx : a.counter;
    auto ticket_item = [&](int t) { return q_first + t < q_hi ? q_first + t : a.n_items; };
    const int k_first = q_lo + (grouped ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    if (k_first >= q_hi) return;
    // batch-norm scale / shift and the bottleneck's weights of the 32 output channels: in LDS, so that an item's epilogue
    // does not wait for global loads (an item is 1-2 us of MFMAs here)
    float* s_par = reinterpret_cast<float*>(s_q + 16);
    if (tid < 96) s_par[tid] = tid < 32 ? a.scale[tid] : tid < 64 ? a.shift[tid - 32] : a.bneck_w ? a.bneck_w[tid - 64] : 0.0f;
    auto queue_at = [&](int n) { return __builtin_amdgcn_readfirstlane(s_q[n & 15]); };

    // items 0 .. D-2 of this workgroup: the first by its block index, the others by one draw of D - 2 tickets
    if (tid == 0) {
        s_q[0] = k_first;
        if (D > 2) {
            const int t = atomicAdd(q_counter, D - 2);
#pragma unroll
            for (int i = 0; i < D - 2; ++i) s_q[1 + i] = ticket_item(t + i);
        }
    }

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;

    if (producer) {
        // ---- wave 3: the copy stream -----------------------------------------------------------------------
        int p_off[KC];
        i32x4_t in_rsrc;
        auto setup = [&](int item) {
            if (item >= a.n_items) {      // nothing left: the remaining copies read zeros (they keep the count exact)
                in_rsrc = make_rsrc(a.in, 0u);
                return;
            }
            const Item it = decode(item);
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int s = k * 64 + lane;
                const int q = s >> 1;
                const int py = q / PW, px = q - py * PW;
                const int half = (s & 1) ^ ((px >> 3) & 1);
                const int gy = it.ty0 - 1 + py, gx = it.tx0 - 1 + px;
                const bool ok = py < Cfg::PH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                p_off[k] = ok ? ((gy * a.W + gx) * 8 + half * 4) * 4 : kOob;
            }
            const float* in_item = a.in + (size_t)it.frame * a.in_frame_stride + (size_t)(a.in_coff / 16) * in_plane;
            in_rsrc = make_rsrc(in_item, (a.debug & 2) ? 0u : (unsigned)(NCH * plane_bytes));     // (tools/: 2 = no loads)
        };
        int c_n = 0, c_ch = 0, c_slot = 0;       // the cursor: item sequence number, chunk, ring slot
        auto copies = [&]() {
            const unsigned img = lds0 + (unsigned)(c_slot * Cfg::kPatchFloats) * 4;
#pragma unroll
            for (int k = 0; k < KC; ++k) blds16s(in_rsrc, p_off[k], c_ch * plane_bytes, img + k * 1024);
            c_slot = c_slot + 1 == S ? 0 : c_slot + 1;
        };
        auto advance = [&]() {       // (behind a barrier: the ring entry it may read was published before it)
            if (++c_ch == NCH) {
                c_ch = 0;
                ++c_n;
                setup(queue_at(c_n));
            }
        };
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("s_barrier" ::: "memory");             // B0: the item ring is visible
        setup(k_first);
#pragma unroll 1
        for (int b = 0; b < S - 1; ++b) {
            copies();
            advance();
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * KC) : "memory");      // chunk 0 has landed
        asm volatile("s_barrier" ::: "memory");             // B1
        for (int n = 0; queue_at(n) < a.n_items; ++n) {
            // (stamps, producer's view: item top | first copies issued | first wait over | first barrier over | last barrier over | set-up done)
            const bool stamp = (a.debug & 32) && (a.debug & 64) && blockIdx.x == 1 && lane == 0 && n < 12;
            int* stamps = a.counter_base + 32 + (n < 12 ? n : 0) * 6;
            if (stamp) stamps[0] = (int)__builtin_amdgcn_s_memtime();
#pragma unroll 1
            for (int ch = 0; ch < NCH; ++ch) {
                copies();                          // chunk k + S - 1 into the image step k - 1 released
                if (stamp && ch == 0) stamps[1] = (int)__builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * KC) : "memory");      // chunk k + 1 has landed
                if (stamp && ch == 0) stamps[2] = (int)__builtin_amdgcn_s_memtime();
                asm volatile("s_barrier" ::: "memory");
                if (stamp && ch == 0) stamps[3] = (int)__builtin_amdgcn_s_memtime();
                if (stamp && ch == NCH - 1) stamps[4] = (int)__builtin_amdgcn_s_memtime();
                advance();
            }
            if (stamp) stamps[5] = (int)__builtin_amdgcn_s_memtime();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing may land in this LDS after the workgroup has left
        return;
    }

    // ---- waves 0-2: weights in registers, MFMAs, epilogue ---------------------------------------------------
    constexpr int WR = Cfg::WR;
    f32x4 wreg[WR * 9];
    const int w_lane = (lh * BN + li) * 4;
    {
        const float* wl = a.w + w_lane;
#pragma unroll
        for (int i = 0; i < WR * 9; ++i)
            wreg[i] = *reinterpret_cast<const f32x4*>(wl + (size_t)(i / 9) * Cfg::kWFloats + (i % 9) * 2 * BN * 4);
    }
    float* s_w = s_par + 96;          // the weights of chunks WR .. NCH - 1, as they lie in global memory
    for (int i = tid; i < Cfg::kWLdsFloats / 4; i += 192)
        reinterpret_cast<f32x4*>(s_w)[i] = reinterpret_cast<const f32x4*>(a.w + (size_t)WR * Cfg::kWFloats)[i];
    int col_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int c = li + kx;
        col_off[kx] = c * 8 + ((lh ^ ((c >> 3) & 1)) * 4);
    }
    const int row0 = wave * MT * PW * 8;
    auto mfma_acc = [&](const f32x4& w, const f32x4& x, f32x16& c) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
    };
    auto mfma_first = [&](const f32x4& w, const f32x4& x, f32x16& c) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(w), "v"(x));
    };
    __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
    for (int i = 0; i < WR * 9; ++i) asm volatile("" : "+v"(wreg[i]));
    asm volatile("s_barrier" ::: "memory");                 // B0
    asm volatile("s_barrier" ::: "memory");                 // B1: chunk 0 is in image 0
    int slot = 0, ticket = 0;
    for (int n = 0;; ++n) {
        const int item = queue_at(n);
        if (item >= a.n_items) break;
        // (a.debug & 32, tools/: s_memtime stamps of the first 12 items of block 1 -- wave 0's view, or with a.debug & 64 the
        //  producer's: item top | first step done | last step done | stores issued | pool issued)
        const bool stamp = (a.debug & 32) && !(a.debug & 64) && blockIdx.x == 1 && tid == 0 && n < 12;
        int* stamps = a.counter_base + 32 + (n < 12 ? n : 0) * 6;
        if (stamp) stamps[0] = (int)__builtin_amdgcn_s_memtime();
        f32x16 acc[MT];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            // queue tickets: at the top of every item's last step the ticket drawn an item ago is published (visible behind
            // this step's barrier) and the next one drawn -- an atomic's round trip is longer than a step, and the compiler's
            // wait for the result also waits for the wave's stores, which are an item old by then
            if (ch == NCH - 1 && tid == 0) {
                if (n >= 1) s_q[(n + D - 2) & 15] = ticket_item(ticket);
                ticket = atomicAdd(q_counter, 1);
            }
            const float* sP = smem + slot * Cfg::kPatchFloats + row0;
            const float* sW = s_w + (ch >= WR ? (ch - WR) * Cfg::kWFloats : 0) + w_lane;
            f32x4 x[MT + 2], wl[2];
#pragma unroll
            for (int r = 0; r < MT + 2; ++r) x[r] = *reinterpret_cast<const f32x4*>(sP + r * PW * 8 + col_off[0]);
            if (ch >= WR) wl[0] = *reinterpret_cast<const f32x4*>(sW);
            __builtin_amdgcn_sched_barrier(0);
            if (!(a.debug & 4) || ch == 0)                // (tools/: 4 = the first chunk's MFMAs only)
#pragma unroll
            for (int g9 = 0; g9 < 9; ++g9) {             // group g9 = kx * 3 + ky
                const int kx = g9 / 3, ky = g9 % 3;
                if (ch >= WR && g9 + 1 < 9)              // the next tap's fragment, one group ahead
                    wl[(g9 + 1) & 1] = *reinterpret_cast<const f32x4*>(sW + (((g9 + 1) % 3) * 3 + (g9 + 1) / 3) * 2 * BN * 4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f32x4& w = ch >= WR ? wl[g9 & 1] : wreg[(ch < WR ? ch : 0) * 9 + ky * 3 + kx];
                    if (ch == 0 && g9 == 0) mfma_first(w, x[mt + ky], acc[mt]);
                    else mfma_acc(w, x[mt + ky], acc[mt]);
                }
                if (ky == 2 && kx < 2) {
#pragma unroll
                    for (int r = 0; r < MT + 2; ++r)
                        x[r] = *reinterpret_cast<const f32x4*>(sP + r * PW * 8 + col_off[kx + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): this step's LDS reads and the ring entry are done
            asm volatile("s_barrier" ::: "memory");
            slot = slot + 1 == S ? 0 : slot + 1;
            if (stamp && ch == 0) stamps[1] = (int)__builtin_amdgcn_s_memtime();
            if (stamp && ch == NCH - 1) stamps[2] = (int)__builtin_amdgcn_s_memtime();
        }
        // the asm MFMAs are opaque to the hazard recogniser: 16-pass results need 18 wait states
#pragma unroll
        for (int k = 0; k < MT; ++k) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[k]));
        const Item cur = decode(item);
        const StreamTile cur_t{cur.frame, cur.ty0, cur.tx0};
        if (a.pool_out) stream_epilogue<1, MT>(a, s_par, s_par + 32, s_par + 64, acc, cur_t, 0, wave, li, lh, stamp, stamps);
        else if (!a.out_nhwc) stream_epilogue<0, MT>(a, s_par, s_par + 32, s_par + 64, acc, cur_t, 0, wave, li, lh, stamp, stamps);
        else stream_epilogue<2, MT>(a, s_par, s_par + 32, s_par + 64, acc, cur_t, 0, wave, li, lh, stamp, stamps,
                                    NCH == 4 ? s_w + Cfg::kWLdsFloats + wave * Cfg::kTileFloats : nullptr);
        if (stamp) { stamps[4] = (int)__builtin_amdgcn_s_memtime(); stamps[5] = stamps[4]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// conv1_1 folded into conv1_2 (bf16 conv path): one launch reads the net's NHWC fp32 input and writes conv1_2's map and its
// 2x2 pool; conv1_1's map (72 MB per BEV pair written and read back, a third of the two layers' traffic) never leaves the CU.
// The structure is conv3x3_bf16_stream_kernel's -- wave 3 streams raw input patches (a ring of S, 10 x 36 pixels each) by
// LDS-DMA, waves 0-2 compute, the grouped ticket queue feeds both -- with two phases per item:
//   A. conv1_1 on the 8 x 34 pixels conv1_2's tile needs (272 = 8.5 MFMA tiles of 32, three per wave) on the bf16 MFMA at
//      fp32 grade: x and w as hi + lo bf16 pairs (v = hi + lo to 16 mantissa bits), a product is
//      w_hi x_hi + w_lo x_hi + w_hi x_lo in three MFMAs (what the split mode of DESIGN.md 5a' does for every layer; the
//      dropped terms are 2^-17 of a product, the map is rounded to bf16 = 2^-9 behind it).  On the fp32 MFMA the same
//      phase is 27 x 64 cycles per tile against 15 x 32: measured, it made the folded launch as slow as the two launches.
//      The PRODUCER wave, idle between its copies, splits every raw patch once, while the consumers are in phase B of the
//      item before: pixel records [hi: the pixel's channels, zero-padded to 16 (CIN0 = 6) or 8 bytes | lo: the same], two
//      buffers by item parity.  A consumer lane's B operands of a K = 16 step are then one or two 16-byte LDS reads
//      (CIN0 = 6: one tap's record per lane half, five steps; CIN0 = 4: two taps' per lane half, three steps; taps beyond
//      the ninth read a zero record) -- splitting in the consumers, per tap, was 2.5x the MFMAs' time in VALU work.
//      Batch-norm + ReLU, zero outside the image (conv1_2's SAME padding), round to bf16, two 16-byte LDS writes per lane
//      and tile into the two chunk images conv1_2 reads -- the same swizzled layout the DMA kernels stage from memory.
//   B. conv1_2 from those images (both chunks resident: 36 MFMAs back to back), epilogue with the fused pool.
// Two barriers per item: X (raw patch n landed, phase B of n - 1 over) and Y (phase A of n written).
template <int CIN0, int S>
struct Bf16First2Cfg {
    static constexpr int MT = 2, TW = 32, TH = 3 * MT, BN = 32;
    static constexpr int PH = TH + 2, PW = TW + 2;                  // conv1_2's input patch = conv1_1's output tile
    static constexpr int RH = TH + 4, RW = TW + 4;                  // the raw input patch
    static constexpr int kPixBytes = CIN0 * 4;
    static constexpr int kRowSlots = RW * kPixBytes / 16;           // 16-byte slots per raw patch row
    static constexpr int kRawSlots = RH * kRowSlots;
    static constexpr int kCopies = (kRawSlots + 63) / 64;
    static constexpr int kRawFloats = kCopies * 256;
    static constexpr int kImgFloats = ((PH * PW * 2 + 63) / 64) * 256;     // a chunk image, as in the other kernels
    static constexpr int kSteps = CIN0 == 6 ? 5 : 3;                // K = 16 steps of phase A
    static constexpr int kWFloats = 9 * 2 * BN * 4;                 // conv1_2: a chunk's weights in global memory
    static constexpr int kRecFloats = CIN0 == 6 ? 8 : 4;           // a split pixel record: hi | lo
    static constexpr int kSplitFloats = (RH * RW + 1) * kRecFloats; // ... of every raw pixel, and a zero record
    static constexpr int kLdsBytes = (S * kRawFloats + 2 * kSplitFloats + 2 * kImgFloats) * 4 + 64 + 4 * 32 * 4;   // | item ring | scale, shift x 2
    static_assert(CIN0 == 6 || CIN0 == 4, "BEV maps (6) or the padded image (4)");
    static_assert((RW * kPixBytes) % 16 == 0, "whole slots per row");
    static_assert((S - 2) * kCopies <= 63, "vmcnt is six bits");
    static_assert(S + 2 <= 16, "item ring");
};

template <int CIN0, int S>
__global__ void __launch_bounds__(256, 2)
conv3x3_bf16_first2_kernel(const ConvArgs a) {
    using Cfg = Bf16First2Cfg<CIN0, S>;
    constexpr int MT = Cfg::MT, BN = Cfg::BN, PW = Cfg::PW, RW = Cfg::RW, KC = Cfg::kCopies, NS = Cfg::kSteps;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_split = smem + S * Cfg::kRawFloats;                      // split raw patches, by item parity
    float* s_img = s_split + 2 * Cfg::kSplitFloats;                   // conv1_2's two chunk images
    int* s_q = reinterpret_cast<int*>(s_img + 2 * Cfg::kImgFloats);   // item ring
    float* s_par = reinterpret_cast<float*>(s_q + 16);                // conv1_2: scale[32], shift[32]; then conv1_1's

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the producer is wave 3 in the first workgroup of a CU and wave 2 in the second (blocks b and b + CUs share a CU when
    // the grid is two per CU): the consumers' MFMAs then spread over all four SIMDs
    const int producer_wave = 3 - (int)((blockIdx.x / ((gridDim.x + 1) / 2)) & 1);
    const bool producer = wave == producer_wave;
    const int cw = wave < producer_wave ? wave : wave - 1;            // consumer index 0..2
    const int li = lane & 31, lh = lane >> 5;

    const int tiles_per_frame = a.tiles_x * a.tiles_y;
    auto decode = [&](int it) {
        it = __builtin_amdgcn_readfirstlane(it);
        const int f = it / tiles_per_frame, r = it - f * tiles_per_frame;
        const int ty = r / a.tiles_x;
        return StreamTile{f, ty * Cfg::TH, (r - ty * a.tiles_x) * Cfg::TW};
    };
    const bool grouped = a.xcd_counters != nullptr;
    const int vx = grouped ? (int)(blockIdx.x & 7) : 0;
    const int q_lo = grouped ? vx * (a.n_items / 8) + min(vx, a.n_items % 8) : 0;
    const int q_hi = grouped ? q_lo + a.n_items / 8 + (vx < a.n_items % 8 ? 1 : 0) : a.n_items;
    const int q_blocks = grouped ? ((int)gridDim.x - vx + 7) / 8 : (int)gridDim.x;
    const int q_first = q_lo + q_blocks;
    int* const q_counter = grouped ? a.xcd_counters + 16 * vx : a.counter;
    auto ticket_item = [&](int t) { return q_first + t < q_hi ? q_first + t : a.n_items; };
    const int k_first = q_lo + (grouped ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    if (k_first >= q_hi) return;
    if (tid < 128)
        s_par[tid] = tid < 32 ? a.scale[tid] : tid < 64 ? a.shift[tid - 32]
                   : tid < 96 ? a.first_scale[tid - 64] : a.first_shift[tid - 96];
    auto queue_at = [&](int n) { return __builtin_amdgcn_readfirstlane(s_q[n & 15]); };
    // ring entries 0 .. S of this workgroup (the producer sets up item n + S at the top of item n); item n publishes entry
    // n + S + 1 with the ticket drawn an item before (the first of them here)
    int ticket = 0;
    if (tid == 0) {
        s_q[0] = k_first;
        const int t = atomicAdd(q_counter, S + 1);
#pragma unroll
        for (int i = 0; i < S; ++i) s_q[1 + i] = ticket_item(t + i);
        ticket = t + S;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;

    if (producer) {
        int p_off[KC];
        i32x4_t in_rsrc;
        const int row_bytes = a.W * Cfg::kPixBytes;
        auto setup = [&](int item) {
            if (item >= a.n_items) {
                in_rsrc = make_rsrc(a.in, 0u);
                return;
            }
            const StreamTile it = decode(item);
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int s = k * 64 + lane;
                const int r = s / Cfg::kRowSlots, j = s - r * Cfg::kRowSlots;
                const int gy = it.ty0 - 2 + r;
                const int xb = (it.tx0 - 2) * Cfg::kPixBytes + 16 * j;       // byte in the image row: whole slots in or out
                const bool ok = r < Cfg::RH && gy >= 0 && gy < a.H && xb >= 0 && xb < row_bytes;
                p_off[k] = ok ? gy * row_bytes + xb : kOob;
            }
            in_rsrc = make_rsrc(a.in + (size_t)it.frame * a.in_frame_stride,
                                (a.debug & 2) ? 0u : (unsigned)(a.H * row_bytes));
        };
        int c_n = 0, c_slot = 0;
        auto copies = [&]() {
            const unsigned img = lds0 + (unsigned)(c_slot * Cfg::kRawFloats) * 4;
#pragma unroll
            for (int k = 0; k < KC; ++k) blds16s(in_rsrc, p_off[k], 0, img + k * 1024);
            c_slot = c_slot + 1 == S ? 0 : c_slot + 1;
            ++c_n;
            setup(queue_at(c_n));        // (the entry was published at least a barrier ago)
        };
        // raw patch m (ring slot m % S) -> hi | lo records in split buffer m & 1
        auto split_patch = [&](int m_slot, int par) {
            const float* raw = smem + m_slot * Cfg::kRawFloats;
            float* dst = s_split + par * Cfg::kSplitFloats;
#pragma unroll
            for (int k = 0; k < (Cfg::RH * Cfg::RW + 63) / 64; ++k) {
                const int q = k * 64 + lane;
                if (q < Cfg::RH * Cfg::RW) {
                    if constexpr (CIN0 == 6) {
                        const f32x2 v0 = *reinterpret_cast<const f32x2*>(raw + q * 6),
                                    v1 = *reinterpret_cast<const f32x2*>(raw + q * 6 + 2),
                                    v2 = *reinterpret_cast<const f32x2*>(raw + q * 6 + 4);
                        const float h0 = pack_bf16(v0[0], v0[1]), h1 = pack_bf16(v1[0], v1[1]), h2 = pack_bf16(v2[0], v2[1]);
                        *reinterpret_cast<f32x4*>(dst + q * 8) = f32x4{h0, h1, h2, 0.f};
                        *reinterpret_cast<f32x4*>(dst + q * 8 + 4) =
                            f32x4{pack_bf16(v0[0] - bf16_value(h0, 0), v0[1] - bf16_value(h0, 1)),
                                  pack_bf16(v1[0] - bf16_value(h1, 0), v1[1] - bf16_value(h1, 1)),
                                  pack_bf16(v2[0] - bf16_value(h2, 0), v2[1] - bf16_value(h2, 1)), 0.f};
                    } else {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(raw + q * 4);
                        const float h0 = pack_bf16(v[0], v[1]), h1 = pack_bf16(v[2], v[3]);
                        *reinterpret_cast<f32x4*>(dst + q * 4) =
                            f32x4{h0, h1, pack_bf16(v[0] - bf16_value(h0, 0), v[1] - bf16_value(h0, 1)),
                                  pack_bf16(v[2] - bf16_value(h1, 0), v[3] - bf16_value(h1, 1))};
                    }
                }
            }
        };
        if (lane < 2 * Cfg::kRecFloats)      // the zero records of both buffers
            s_split[(lane / Cfg::kRecFloats) * Cfg::kSplitFloats + Cfg::RH * Cfg::RW * Cfg::kRecFloats + lane % Cfg::kRecFloats] = 0.f;
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("s_barrier" ::: "memory");             // B0: the item ring is visible
        setup(k_first);
#pragma unroll 1
        for (int b = 0; b < S - 1; ++b) copies();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * KC) : "memory");      // patch 0 has landed
        split_patch(0, 0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        asm volatile("s_barrier" ::: "memory");             // X_0
        int r_slot = 0;
        for (int n = 0; queue_at(n) < a.n_items; ++n) {
            copies();                                        // patch n + S - 1 into the image whose split is long done
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * KC) : "memory");      // patch n + 1 has landed
            asm volatile("s_barrier" ::: "memory");         // Y_n: phase A of item n is over (split buffer (n + 1) & 1 is free)
            r_slot = r_slot + 1 == S ? 0 : r_slot + 1;
            split_patch(r_slot, (n + 1) & 1);                // under phase B and the epilogue of item n
            __builtin_amdgcn_s_waitcnt(0xc07f);
            asm volatile("s_barrier" ::: "memory");         // X_{n+1}
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- waves 0-2 ------------------------------------------------------------------------------------------
    // conv1_2's weights (two chunks) and conv1_1's hi / lo fragments: registers
    f32x4 wreg[2 * 9], w1[NS][2];
    {
        const float* wl = a.w + (lh * BN + li) * 4;
#pragma unroll
        for (int i = 0; i < 18; ++i)
            wreg[i] = *reinterpret_cast<const f32x4*>(wl + (size_t)(i / 9) * Cfg::kWFloats + (i % 9) * 2 * BN * 4);
        // [step][hi, lo][lane half][32 MFMA rows][8 bf16]
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                w1[s][part] = *reinterpret_cast<const f32x4*>(a.first_w + (((s * 2 + part) * 2 + lh) * 32 + li) * 4);
    }
    // phase A lane constants: the wave's three MFMA tiles cover patch pixels p = 32 t + li, t = 3 wave + i
    int a_raw[3], a_cell[3], a_rc[3];       // (a_rc: patch row << 8 | column)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = 32 * (3 * cw + i) + li;
        const int pq = p < Cfg::PH * PW ? p : Cfg::PH * PW - 1;            // (the last tile is half empty: reads stay inside)
        const int pr = pq / PW, pc = pq - pr * PW;
        a_raw[i] = (pr * RW + pc) * Cfg::kRecFloats;                       // floats: the record of tap (0, 0)
        a_cell[i] = p < Cfg::PH * PW ? (pr * PW + pc) * 8 + ((lh ^ ((pc >> 3) & 1)) * 4) : -1;
        a_rc[i] = pr << 8 | pc;
    }
    // this lane half's tap(s) of step s: float offset of its record from tap (0, 0)'s, or -1 (beyond the ninth tap)
    auto tap_offset = [&](int s, int u) {
        auto of = [](int tap) { return tap < 9 ? ((tap / 3) * RW + tap % 3) * Cfg::kRecFloats : -1; };
        const int t0 = CIN0 == 6 ? 2 * s : 4 * s + u, t1 = CIN0 == 6 ? 2 * s + 1 : 4 * s + 2 + u;     // lane half 0 / 1
        return lh ? of(t1) : of(t0);
    };
    int col_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int c = li + kx;
        col_off[kx] = c * 8 + ((lh ^ ((c >> 3) & 1)) * 4);
    }
    const int row0 = cw * MT * PW * 8;
    auto mfma_acc = [&](const f32x4& w, const f32x4& x, f32x16& c) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
    };
    auto mfma_first = [&](const f32x4& w, const f32x4& x, f32x16& c) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(w), "v"(x));
    };
    __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
    for (int i = 0; i < 18; ++i) asm volatile("" : "+v"(wreg[i]));
#pragma unroll
    for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(w1[s][0]), "+v"(w1[s][1]));
    asm volatile("s_barrier" ::: "memory");                 // B0
    asm volatile("s_barrier" ::: "memory");                 // X_0: raw patch 0 is in image 0
    int slot = 0;
    for (int n = 0;; ++n) {
        const int item = queue_at(n);
        if (item >= a.n_items) break;
        const bool stamp = (a.debug & 32) && blockIdx.x == 1 && tid == 0 && n < 12;
        int* stamps = a.counter_base + 32 + (n < 12 ? n : 0) * 6;
        if (stamp) stamps[0] = (int)__builtin_amdgcn_s_memtime();
        const StreamTile cur = decode(item);
        // ---- phase A: conv1_1 of the wave's three tiles ----
        const float* rec = s_split + (n & 1) * Cfg::kSplitFloats;
        {
            // (compiler-visible MFMAs here, not the inline-asm ones of phase B: the operands are assembled in registers, and
            //  only the compiler's hazard recogniser keeps a VALU write away from a register an MFMA in flight still reads)
            f32x16 c1[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) c1[i][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // (a tap beyond the ninth: the zero record behind the patch's; the three tiles' accumulators take turns)
                f32x4 xh[3], xl[3];
                if constexpr (CIN0 == 6) {
                    const int off = tap_offset(s, 0);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float* q = rec + (off < 0 ? Cfg::RH * RW * 8 : a_raw[i] + off);
                        xh[i] = *reinterpret_cast<const f32x4*>(q);
                        xl[i] = *reinterpret_cast<const f32x4*>(q + 4);
                    }
                } else {
                    const int o0 = tap_offset(s, 0), o1 = tap_offset(s, 1);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rec + (o0 < 0 ? Cfg::RH * RW * 4 : a_raw[i] + o0));
                        const f32x4 r1 = *reinterpret_cast<const f32x4*>(rec + (o1 < 0 ? Cfg::RH * RW * 4 : a_raw[i] + o1));
                        xh[i] = f32x4{r0[0], r0[1], r1[0], r1[1]};
                        xl[i] = f32x4{r0[2], r0[3], r1[2], r1[3]};
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) c1[i] = mfma_bf16(w1[s][0], xh[i], c1[i]);
#pragma unroll
                for (int i = 0; i < 3; ++i) c1[i] = mfma_bf16(w1[s][1], xh[i], c1[i]);
#pragma unroll
                for (int i = 0; i < 3; ++i) c1[i] = mfma_bf16(w1[s][0], xl[i], c1[i]);
            }
            // batch-norm + ReLU, zero outside the image, bf16, into the chunk images
            f32x4 sc[4], sh[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = group_channel<true>(g, lh);
                sc[g] = *reinterpret_cast<const f32x4*>(s_par + 64 + c);
                sh[g] = *reinterpret_cast<const f32x4*>(s_par + 96 + c);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int gy = cur.ty0 - 1 + (a_rc[i] >> 8), gx = cur.tx0 - 1 + (a_rc[i] & 255);
                const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                float r[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float v = fmaxf(c1[i][k] * sc[k >> 2][k & 3] + sh[k >> 2][k & 3], 0.0f);
                    r[k] = inside ? v : 0.0f;
                }
                if (a_cell[i] >= 0) {
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch)
                        *reinterpret_cast<f32x4*>(s_img + ch * Cfg::kImgFloats + a_cell[i]) =
                            f32x4{pack_bf16(r[8 * ch], r[8 * ch + 1]), pack_bf16(r[8 * ch + 2], r[8 * ch + 3]),
                                  pack_bf16(r[8 * ch + 4], r[8 * ch + 5]), pack_bf16(r[8 * ch + 6], r[8 * ch + 7])};
                }
            }
        }
        // queue tickets as in conv3x3_bf16_stream_kernel: publish the one drawn an item ago, draw the next (here, behind
        // phase A: the wait for the old ticket also waits for the previous epilogue's stores)
        if (tid == 0) {
            s_q[(n + S + 1) & 15] = ticket_item(ticket);
            ticket = atomicAdd(q_counter, 1);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the images are written, the ring entry too
        asm volatile("s_barrier" ::: "memory");   // Y_n
        if (stamp) stamps[1] = (int)__builtin_amdgcn_s_memtime();
        // ---- phase B: conv1_2, both chunks ----
        f32x16 acc[MT];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const float* sP = s_img + ch * Cfg::kImgFloats + row0;
            f32x4 x[MT + 2];
#pragma unroll
            for (int r = 0; r < MT + 2; ++r) x[r] = *reinterpret_cast<const f32x4*>(sP + r * PW * 8 + col_off[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g9 = 0; g9 < 9; ++g9) {
                const int kx = g9 / 3, ky = g9 % 3;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (ch == 0 && g9 == 0) mfma_first(wreg[ky * 3 + kx], x[mt + ky], acc[mt]);
                    else mfma_acc(wreg[ch * 9 + ky * 3 + kx], x[mt + ky], acc[mt]);
                }
                if (ky == 2 && kx < 2) {
#pragma unroll
                    for (int r = 0; r < MT + 2; ++r)
                        x[r] = *reinterpret_cast<const f32x4*>(sP + r * PW * 8 + col_off[kx + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int k = 0; k < MT; ++k) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[k]));
        if (stamp) stamps[2] = (int)__builtin_amdgcn_s_memtime();
        if (a.pool_out) stream_epilogue<1, MT>(a, s_par, s_par + 32, s_par + 64, acc, cur, 0, cw, li, lh, stamp, stamps);
        else stream_epilogue<0, MT>(a, s_par, s_par + 32, s_par + 64, acc, cur, 0, cw, li, lh, stamp, stamps);
        if (stamp) { stamps[4] = (int)__builtin_amdgcn_s_memtime(); stamps[5] = stamps[4]; }
        slot = slot + 1 == S ? 0 : slot + 1;
        asm volatile("s_barrier" ::: "memory");   // X_{n+1}
    }
}

}  // namespace dodt
