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
        if (comp_ch == 0 && tid == 0) s_ctrl[0] = ticket_item(atomicAdd(q_counter, 1));
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

}  // namespace dodt
