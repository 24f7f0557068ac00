// 3x3 stride-1 convolution in fp32 by Winograd F(2x2, 3x3) on the fp32 MFMA of gfx950.
//
// Why: the fp32 matrix pipe (v_mfma_f32_16x16x4_f32 / 32x32x2) runs at the fp32 vector rate,
// 1/16 of the bf16 rate, so an fp32 conv is bound by the NUMBER of multiplications.  The
// minimal-filtering form computes every 2x2 block of outputs from a 4x4 block of inputs with
// 16 multiplications per (input channel, output channel) instead of 36:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray 2016; Winograd 1980)
// i.e. 16 independent GEMMs  M_xi[cout][tile] = sum_cin U_xi[cout][cin] V_xi[cin][tile]  over
// the 16 points xi of the transformed domain -- 2.25x fewer MFMA cycles for the same layer.
// Everything stays fp32 (the transforms are +-, the filter transform (x 1/2, x 1/4) runs on
// the host in float64): results agree with the direct form to ~1e-6 of a layer's scale (the
// per-layer bar of tests/test_gpu_conv.py is 1e-4).
//
// Work decomposition (one workgroup = 4 waves = one 16-column x 16*TB-row output tile of one
// frame x BN = 16*CB output channels; TB * CB = 4):
//   * a wave owns TB "tile blocks" (2 x 8 Winograd tiles = 4 x 16 output pixels each) and all
//     BN channels: 16 points x TB x CB accumulators of v_mfma_f32_16x16x4_f32 = 256 registers.
//     Lane l: tile t = l % 16 of the block (B operand column), k-pair g = l / 16: the MFMA's
//     four k lanes take the chunk's input channels (2g) in one instruction and (2g + 1) in the
//     next, so every operand is ONE 8-byte LDS read (ds_read_b64) per two MFMAs.
//   * K is walked in chunks of 8 input channels = one CB8 plane.  Per chunk the halo'd input
//     patch and the chunk's transformed weights [xi][h][n][4] are copied global -> LDS by
//     global_load_lds_dwordx4 (no staging registers, no ds_write pass; out-of-image pixels read
//     a zero page), double buffered: chunk k+1 lands while chunk k is computed, one barrier
//     per chunk.  The chunk sequence runs across work items (persistent workgroups).
//   * per chunk a lane reads its tile's 4x4 pixels (16 ds_read_b64), applies B^T d B (64 VALU
//     adds, in the shadow of the other tile block's MFMAs), then issues 32 MFMAs per tile block
//     and channel block pair.
//   * epilogue: A^T M A per lane (it holds all 16 points of its tile and 4 consecutive output
//     channels per channel block), batch-norm scale/shift + ReLU, 16-byte stores into the CB8
//     (or NHWC) output.  A lane's 2x2 outputs are exactly one window of a following 2x2 max
//     pool, so the fused pool is a max of four registers; the 1x1 bottleneck of the last
//     layer is a dot over the lane's channels + two cross-lane adds.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "conv_kernels.h"
#include "lds_dma.h"

namespace dodt {

typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int TB, int CB>
struct WinoCfg {
    static_assert(TB * CB == 4 || TB * CB == 2, "256 or 128 accumulator registers per wave");
    // 256 accumulator registers: one wave per SIMD, the step is software-pipelined inside the
    // wave (kPipe).  128: two workgroups per CU, the partner wave on the SIMD fills the gaps, the
    // step is the plain sequence copies -> input transform -> MFMAs.
    static constexpr bool kPipe = TB * CB == 4;
    static constexpr int TW = 16;            // output columns of a workgroup tile
    static constexpr int TH = 16 * TB;       // output rows (4 waves x TB tile blocks x 4 rows)
    static constexpr int BN = 16 * CB;       // output channels of a workgroup tile
    static constexpr int PH = TH + 2, PW = TW + 2;
    // LDS patch image: pixel (py, px) lives in 32-byte cell py * kPitch + px + ((py >> 1) & 1),
    // its two 16-byte halves swapped when (px >> 3) & 1.  With that placement the 16 lanes
    // that one LDS cycle of a ds_read_b64 serves (2 tile rows x 8 tile columns, pixels two
    // apart) fall into 16 different 16-byte bank columns for every tap (r, c) of the 4x4 input
    // tile.  The image is filled by LDS-DMA, which writes lane-linear: the permutation is
    // applied to the SOURCE address of each 16-byte slot.
    static constexpr int kPitch = PW + 1;
    static constexpr int kPatchSlots = (PH * kPitch + 1) * 2;         // 16-byte slots
    static constexpr int kPatchInstr = (kPatchSlots + 63) / 64;       // wave-level copies
    static constexpr int kPatchFloats = kPatchInstr * 64 * 4;
    static constexpr int kWFloats = 16 * 8 * BN;   // [xi][g][cb pair][t][cb & 1][k]: 16 B per lane
    static constexpr int kWInstr = kWFloats / 256;
    static constexpr int kBufFloats = kPatchFloats + kWFloats;
    static constexpr int kLdsBytes = 2 * kBufFloats * 4 + 16 + 1024;   // + control word + dummy copy slot
    static constexpr int kInstr = kPatchInstr + kWInstr;              // copies per chunk
    static constexpr int kPerWave = (kInstr + 3) / 4;
    static constexpr int kPatchPerWave = (kPatchInstr + 3) / 4;
};

// a - b on a register pair in one instruction (the compiler splits a v2f32 subtraction into two v_sub_f32)
typedef float wino_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wino_f32x2 pk_sub(wino_f32x2 a, wino_f32x2 b) {
    wino_f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}


template <int TB, int CB>
__global__ void __launch_bounds__(256, (TB * CB == 4 ? 1 : 2))
wino3x3_f32_kernel(const ConvArgs a) {
    using Cfg = WinoCfg<TB, CB>;
    constexpr int BN = Cfg::BN, PW = Cfg::PW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* s_ctrl = reinterpret_cast<int*>(smem + 2 * Cfg::kBufFloats);   // behind the four images
    float* const sDummy = smem + 2 * Cfg::kBufFloats + 4;   // 1 KB: target of a wave's surplus copy

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: addresses, branches
    const int t = lane & 15, g = lane >> 4;
    const int nchunks = a.Cin / 8;
    const int in_plane = a.H * a.W * 8;

    struct Item { int frame, ntile, ty0, tx0; };
    auto decode = [&](int it) {
        const int4 v = a.items[__builtin_amdgcn_readfirstlane(it)];
        return Item{v.x, v.y, v.z, v.w};
    };

    // ---- copy plan of this wave: patch copies j = wave, wave + 4, ... < kPatchInstr (64 LDS
    //      slots each), weight copies likewise over kWInstr (256 floats each) -------------------
    int p_off[Cfg::kPatchPerWave];      // byte offset inside the plane, kOob: zero padding
    i32x4_t in_rsrc, w_rsrc;
    const int plane_bytes = in_plane * 4;
    auto setup_patch = [&](const Item& it) {
#pragma unroll
        for (int k = 0; k < Cfg::kPatchPerWave; ++k) {
            const int j = wave + 4 * k;
            const int s = j * 64 + lane;
            const int q = s >> 1;                         // 32-byte cell of the LDS image
            const int py = q / Cfg::kPitch;
            const int px = q - py * Cfg::kPitch - ((py >> 1) & 1);
            const int half = (s & 1) ^ ((px >> 3) & 1);   // the halves of a cell may be swapped
            const int gy = it.ty0 - 1 + py, gx = it.tx0 - 1 + px;
            const bool ok = j < Cfg::kPatchInstr && py < Cfg::PH && px >= 0 && px < PW &&
                            gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            p_off[k] = ok ? ((gy * a.W + gx) * 8 + half * 4) * 4 : kOob;
        }
        const float* in_item = a.in + (size_t)it.frame * a.in_frame_stride +
                               (size_t)(a.in_coff / 8) * in_plane;
        in_rsrc = make_rsrc(in_item, (unsigned)(nchunks * plane_bytes));
    };
    auto setup_w = [&](const Item& it) {
        const float* w_item = a.w + (size_t)it.ntile * nchunks * Cfg::kWFloats;
        w_rsrc = make_rsrc(w_item, (unsigned)(nchunks * Cfg::kWFloats * 4));
    };
    // LDS: two patch images, two weight images
    float* const sPB = smem;
    float* const sWB = smem + 2 * Cfg::kPatchFloats;
    // copy number n of this wave for the step: n < kPatchPerWave -> patch piece of chunk pch
    // into patch image pb; else weight piece of chunk wch into weight image wb
    int pit = 0, pch = 0, wit = 0, wch = 0;       // copy cursors (item, chunk)
    // Branch-free (the main loop's slices must stay single basic blocks for the scheduler): a
    // copy that does not exist for this wave goes, with an out-of-range offset (zeros), to a
    // dummy slot; a cursor beyond the last item has an empty descriptor.
    auto copy_n = [&](int n, int pb, int wb) {
        if (n < Cfg::kPatchPerWave) {            // compile-time
            const int j = wave + 4 * n;
            const bool real = j < Cfg::kPatchInstr;
            float* dst = real ? sPB + pb * Cfg::kPatchFloats + j * 256 : sDummy;
            blds16(in_rsrc, p_off[n], pch * plane_bytes, dst);
        } else {
            const int j = wave + 4 * (n - Cfg::kPatchPerWave);
            const bool real = j < Cfg::kWInstr;
            float* dst = real ? sWB + wb * Cfg::kWFloats + j * 256 : sDummy;
            blds16(w_rsrc, real ? lane * 16 : kOob, wch * (Cfg::kWFloats * 4) + j * 1024, dst);
        }
    };
    constexpr int kCopies = Cfg::kPatchPerWave + (Cfg::kWInstr + 3) / 4;
    // the same copy with scalar operands only (blds16s): for the issue path between MFMAs
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    const unsigned lds_dummy = lds0 + (unsigned)(2 * Cfg::kBufFloats + 4) * 4;
    const int w_voff = lane * 16;
    auto copy_s = [&](int n, int img) {
        if (n < Cfg::kPatchPerWave) {            // compile-time
            const int j = wave + 4 * n;
            const unsigned dst = j < Cfg::kPatchInstr ? lds0 + (unsigned)(img * Cfg::kPatchFloats + j * 256) * 4 : lds_dummy;
            blds16s(in_rsrc, p_off[n], pch * plane_bytes, dst);
        } else {
            const int j = wave + 4 * (n - Cfg::kPatchPerWave);
            const bool real = j < Cfg::kWInstr;
            const unsigned dst = real ? lds0 + (unsigned)(2 * Cfg::kPatchFloats + img * Cfg::kWFloats + j * 256) * 4 : lds_dummy;
            blds16s(w_rsrc, real ? w_voff : kOob, wch * (Cfg::kWFloats * 4) + j * 1024, dst);
        }
    };

    // the queue: one counter for the launch, or (a.xcd_counters: see conv_bf16_dma.h) one per group of blocks that
    // share an XCD, each walking its own eighth [q_lo, q_hi) of the table
    const bool grouped = a.xcd_counters != nullptr;
    const int vx = grouped ? (int)(blockIdx.x & 7) : 0;
    const int q_lo = grouped ? vx * (a.n_items / 8) + min(vx, a.n_items % 8) : 0;
    const int q_hi = grouped ? q_lo + a.n_items / 8 + (vx < a.n_items % 8 ? 1 : 0) : a.n_items;
    const int q_first = q_lo + (grouped ? ((int)gridDim.x - vx + 7) / 8 : (int)gridDim.x);      // item of ticket 0
    int* const q_counter = grouped ? a.xcd_counters + 16 * vx : a.counter;
    int comp_item = q_lo + (grouped ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    if (comp_item >= q_hi) return;
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

    // lane constants: tile of the lane inside a tile block, weight fragment offset,
    // float offsets of the lane's 4x4 input tile in the patch image (see WinoCfg)
    const int tyl = t >> 3, txl = t & 7;
    const int w_lane = (g * (CB / 2) * 16 + t) * 4;
    int row_off[TB][4], col_off[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int px = 2 * txl + c;
        col_off[c] = px * 8 + (((g >> 1) ^ ((px >> 3) & 1)) * 4) + (g & 1) * 2;
    }
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int py = 2 * (2 * (wave * TB + tb) + tyl) + r;
            row_off[tb][r] = (py * Cfg::kPitch + ((py >> 1) & 1)) * 8;
        }

    // input transform V = B^T d B of the lane's tile in tile block tb from patch image pb,
    // split in the pieces the main loop spreads between its MFMAs
    auto load_d = [&](f32x2_t (&d)[4][4], int tb, int r, int pb) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            d[r][c] = *reinterpret_cast<const f32x2_t*>(sPB + pb * Cfg::kPatchFloats +
                                                          row_off[tb][r] + col_off[c]);
    };
    auto rows_d = [&](f32x2_t (&d)[4][4], int c) {
        const f32x2_t t0 = d[0][c] - d[2][c], t1 = d[1][c] + d[2][c];
        const f32x2_t t2 = d[2][c] - d[1][c], t3 = d[1][c] - d[3][c];
        d[0][c] = t0; d[1][c] = t1; d[2][c] = t2; d[3][c] = t3;
    };
    auto cols_d = [&](const f32x2_t (&d)[4][4], f32x2_t (&v)[16], int i) {
        v[i * 4 + 0] = d[i][0] - d[i][2];
        v[i * 4 + 1] = d[i][1] + d[i][2];
        v[i * 4 + 2] = d[i][2] - d[i][1];
        v[i * 4 + 3] = d[i][1] - d[i][3];
    };

    // ---- prologue: patch(0), W(0) [, patch(1); V(0)] ---------------------------------------------
    pit = wit = comp_item;
    setup_patch(decode(pit));
    setup_w(decode(wit));
#pragma unroll
    for (int n = 0; n < Cfg::kPatchPerWave; ++n) copy_n(n, 0, 0);
    advance(pit, pch, true);
#pragma unroll
    for (int n = Cfg::kPatchPerWave; n < kCopies; ++n) copy_n(n, 0, 0);
    advance(wit, wch, false);
    if constexpr (Cfg::kPipe) {
#pragma unroll
        for (int n = 0; n < Cfg::kPatchPerWave; ++n) copy_n(n, 1, 0);
        advance(pit, pch, true);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    f32x2_t va[TB][16], vb[Cfg::kPipe ? TB : 1][16];
    if constexpr (Cfg::kPipe) {
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            f32x2_t d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) load_d(d, tb, r, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) rows_d(d, c);
#pragma unroll
            for (int i = 0; i < 4; ++i) cols_d(d, va[tb], i);
        }
        __syncthreads();      // patch image 0 is free for patch(2)
    }

    // One step = the MFMAs of chunk k (V from registers, weights from weight image k & 1) and,
    // spread between them: the copies of W(k+1) -> weight image (k+1) & 1 and patch(k+2) ->
    // patch image k & 1, and the input transform of chunk k+1 from patch image (k+1) & 1.
    // PAR = k & 1 (the chunk loop is unrolled by two so that images and V sets are static).
    int k_stamp = 0;      // steps this wave has run (diagnostic stamps)
    auto step = [&](auto par, f32x4 (&acc)[TB][CB][16], auto& vcur, auto& vnext, int comp_ch) {
        constexpr int PAR = decltype(par)::value;
        if (comp_ch == 0 && tid == 0) {
            const int t = q_first + atomicAdd(q_counter, 1);
            s_ctrl[0] = t < q_hi ? t : a.n_items;
        }
        const float* sW = sWB + PAR * Cfg::kWFloats + w_lane;
        f32x2_t d[4][4];
        if constexpr (!Cfg::kPipe) {
            // plain step: copies of chunk k+1 (both images PAR ^ 1), then per tile block the
            // input transform of chunk k and its MFMAs.
            // (a.debug: diagnostic switches of tools/, 0 in production -- 2: no copies; 32:
            //  s_memtime stamps of one wave's first 12 steps into counter_base[32..103])
            const bool stamp = (a.debug & 32) && blockIdx.x == 1 && tid == 0 && k_stamp < 12;
            int* stamps = a.counter_base + 32 + (k_stamp < 12 ? k_stamp : 0) * 6;
            if (stamp) {
                stamps[0] = (int)__builtin_amdgcn_s_memtime();
                if (k_stamp == 0 || k_stamp == 11)    // 100 MHz reference beside it, to calibrate
                    a.counter_base[32 + 72 + (k_stamp != 0)] = (int)__builtin_amdgcn_s_memrealtime();
            }
            // The copies of chunk k+1 (both images PAR ^ 1: free since the barrier that ended the previous step)
            // go out one per point over the first points of the MFMA loop below (so that they have the rest of the
            // step to land), with scalar operands only, in the matrix pipe's shadow -- issued as one burst at the top of the step (round 2) they held this wave's MFMAs back by
            // their issue time, 400-790 cycles of a 4 800-cycle step.
            if (stamp) stamps[1] = (int)__builtin_amdgcn_s_memtime();
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) load_d(d, tb, r, PAR);
#pragma unroll
                for (int c = 0; c < 4; ++c) rows_d(d, c);
#pragma unroll
                for (int i = 0; i < 4; ++i) cols_d(d, vnext[0], i);
                if (stamp) {
                    // make the transform's end observable: the stamp depends on its result
                    stamps[2] = (int)__builtin_amdgcn_s_memtime() + (vnext[0][15][1] == 12345.f);
                }
                // weight fragments two points ahead of the MFMAs that use them (the scheduling barriers that
                // pin the copies would otherwise leave each read's latency in front of its MFMAs)
                constexpr int kWAhead = 2;
                f32x4 wq[kWAhead + 1][CB / 2];
#pragma unroll
                for (int i = 0; i < kWAhead; ++i)
#pragma unroll
                    for (int cp = 0; cp < CB / 2; ++cp)
                        wq[i][cp] = *reinterpret_cast<const f32x4*>(sW + (i * 4 * (CB / 2) + cp) * 64);
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    if (x + kWAhead < 16) {
#pragma unroll
                        for (int cp = 0; cp < CB / 2; ++cp)
                            wq[(x + kWAhead) % (kWAhead + 1)][cp] =
                                *reinterpret_cast<const f32x4*>(sW + ((x + kWAhead) * 4 * (CB / 2) + cp) * 64);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb)
                            acc[tb][cb][x] = mfma16(wq[x % (kWAhead + 1)][cb >> 1][(cb & 1) * 2 + s2], vnext[0][x][s2],
                                                    acc[tb][cb][x]);
                    if (tb == 0 && x < kCopies && !(a.debug & 2)) copy_s(x, PAR ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (stamp) stamps[3] = (int)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0);
            if (stamp) stamps[4] = (int)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            if (stamp) stamps[5] = (int)__builtin_amdgcn_s_memtime();
            ++k_stamp;
            if (comp_ch == 0) q0 = s_ctrl[0];
            advance(pit, pch, true);
            advance(wit, wch, false);
            return;
        } else {
        // weight fragments are read one slice ahead of the MFMAs that use them (the slices are
        // pinned by scheduling barriers, so nothing else prefetches them)
        f32x4 w[2][CB / 2];
#pragma unroll
        for (int cp = 0; cp < CB / 2; ++cp)
            w[0][cp] = *reinterpret_cast<const f32x4*>(sW + cp * 64);
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const int sl = tb * 16 + x;                       // slice number, 0 .. 16 TB - 1
                const int wc = sl & 1, wn = wc ^ 1;
                if (sl + 1 < 16 * TB) {
                    const int xn = (x + 1) & 15;
#pragma unroll
                    for (int cp = 0; cp < CB / 2; ++cp)
                        w[wn][cp] = *reinterpret_cast<const f32x4*>(
                            sW + (xn * 4 * (CB / 2) + cp) * 64);
                }
                // transform of chunk k+1, tile block tb: loads in slices 0..3, rows 5..8,
                // columns 10..13 of this tile block's 16 slices (issued ahead of the MFMAs of
                // the slice, in whose shadow they run)
                if (x <= 3) load_d(d, tb, x, PAR ^ 1);
                if (x >= 5 && x <= 8) rows_d(d, x - 5);
                if (x >= 10 && x <= 13) cols_d(d, vnext[tb], x - 10);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        acc[tb][cb][x] = mfma16(w[wc][cb >> 1][(cb & 1) * 2 + s2],
                                                vcur[tb][x][s2], acc[tb][cb][x]);
                if (sl < kCopies) copy_n(sl, PAR, PAR ^ 1);       // patch(k+2) -> PAR, W(k+1) -> PAR^1
                // One wave per SIMD issues in order: whatever follows a burst of MFMAs waits for
                // the whole burst.  Interleave: after every MFMA (32 cycles in the matrix pipe)
                // two vector instructions and one LDS read of the slice's side work.
#pragma unroll
                for (int i = 0; i < 2 * CB; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        // the copies issued in this step have landed (they are waited for ahead of an epilogue's
        // stores, which share the vmcnt counter); then one barrier per chunk
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_s_barrier();
        if (comp_ch == 0) q0 = s_ctrl[0];
        advance(pit, pch, true);
        advance(wit, wch, false);
    };

    while (comp_item < a.n_items) {
        f32x4 acc[TB][CB][16];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int x = 0; x < 16; ++x) acc[tb][cb][x] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int comp_ch = 0; comp_ch < nchunks; comp_ch += 2) {     // Cin / 8 is even
            if constexpr (Cfg::kPipe) {
                step(std::integral_constant<int, 0>{}, acc, va, vb, comp_ch);
                step(std::integral_constant<int, 1>{}, acc, vb, va, comp_ch + 1);
            } else {      // (V lives only inside a step: `va` is scratch for it)
                step(std::integral_constant<int, 0>{}, acc, vb, va, comp_ch);
                step(std::integral_constant<int, 1>{}, acc, vb, va, comp_ch + 1);
            }
        }

        // ---- epilogue: Y = A^T M A, batch-norm + ReLU, stores (they drain under the next
        //      item's first chunk) ----------------------------------------------------------------
        // (a.debug & 1, tools/: the kernel without its epilogues -- what they cost)
        if (!(a.debug & 1)) {
            // The epilogue's arguments come from the kernarg segment again (scalar loads, through a pointer the
            // compiler cannot see through): kept in SGPRs across the K loop they are spilled to vector lanes and
            // every item pays ~100 v_readlane for them on the vector pipe the MFMAs of the other workgroup need.
            typedef const ConvArgs __attribute__((address_space(4))) KernArgs;
            KernArgs* ep_ = (KernArgs*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ep_));
            KernArgs& e = *ep_;
            const Item it = decode(comp_item);
            float* out = e.out + (size_t)it.frame * e.out_frame_stride;
            const int out_rows = e.H - e.out_y0;
            const long long plane = (long long)out_rows * e.W * 8;
            const bool pool = e.pool_out != nullptr;
            const long long pplane = (long long)(e.H >> 1) * (e.W >> 1) * 8;
            // every pixel of the tile is stored: no per-pixel tests (scalar condition)
            const bool interior = it.ty0 + Cfg::TH <= e.H && it.tx0 + Cfg::TW <= e.W && it.ty0 >= e.out_y0;
            const float relu_floor = e.relu ? 0.0f : -__builtin_inff();
            typedef wino_f32x2 f32x2;
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const int oy0 = it.ty0 + 2 * (2 * (wave * TB + tb) + tyl);
                const int ox0 = it.tx0 + 2 * txl;
                // CB8 maps: the lane's cell offset inside a PAIR of channel planes (plane g >> 1, floats 4 (g & 1) ..
                // of the 32-byte cell), in floats -- a frame's planes stay below 2^31 floats; the pair's base is scalar
                const int w8 = e.W * 8;
                const int cell = (g >> 1) * (int)plane + ((oy0 - e.out_y0) * e.W + ox0) * 8 + 4 * (g & 1);
                const int pcell = (g >> 1) * (int)pplane + ((oy0 >> 1) * (e.W >> 1) + (ox0 >> 1)) * 8 + 4 * (g & 1);
                float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const int c0 = it.ntile * BN + cb * 16 + 4 * g;
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(e.scale + c0);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(e.shift + c0);
                    f32x4 y[4];       // outputs (0,0) (0,1) (1,0) (1,1), 4 channels each
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {      // channel pairs: packed fp32 instructions
                        auto M = [&](int x) { return f32x2{acc[tb][cb][x][2 * hf], acc[tb][cb][x][2 * hf + 1]}; };
                        f32x2 z[4][2];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            z[i][0] = (M(i * 4 + 0) + M(i * 4 + 1)) + M(i * 4 + 2);
                            z[i][1] = pk_sub(pk_sub(M(i * 4 + 1), M(i * 4 + 2)), M(i * 4 + 3));
                        }
                        const f32x2 sc2 = {sc[2 * hf], sc[2 * hf + 1]}, sh2 = {sh[2 * hf], sh[2 * hf + 1]};
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const f32x2 t0 = ((z[0][b] + z[1][b]) + z[2][b]) * sc2 + sh2;
                            const f32x2 t1 = pk_sub(pk_sub(z[1][b], z[2][b]), z[3][b]) * sc2 + sh2;
                            y[0 + b][2 * hf] = fmaxf(t0[0], relu_floor);
                            y[0 + b][2 * hf + 1] = fmaxf(t0[1], relu_floor);
                            y[2 + b][2 * hf] = fmaxf(t1[0], relu_floor);
                            y[2 + b][2 * hf + 1] = fmaxf(t1[1], relu_floor);
                        }
                    }
                    if (!e.out_nhwc) {
                        float* cellp = out + (size_t)((e.out_coff + it.ntile * BN + cb * 16) >> 3) * plane + cell;
                        if (interior) {
                            *reinterpret_cast<f32x4*>(cellp) = y[0];
                            *reinterpret_cast<f32x4*>(cellp + 8) = y[1];
                            *reinterpret_cast<f32x4*>(cellp + w8) = y[2];
                            *reinterpret_cast<f32x4*>(cellp + w8 + 8) = y[3];
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int oy = oy0 + (q >> 1), ox = ox0 + (q & 1);
                                if (oy < e.H && ox < e.W && oy >= e.out_y0)
                                    *reinterpret_cast<f32x4*>(cellp + (q >> 1) * w8 + (q & 1) * 8) = y[q];
                            }
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int oy = oy0 + (q >> 1), ox = ox0 + (q & 1);
                            if (oy < e.H && ox < e.W && oy >= e.out_y0)
                                *reinterpret_cast<f32x4*>(out + ((size_t)(oy - e.out_y0) * e.W + ox) * e.out_ld +
                                                          e.out_coff + c0) = y[q];
                        }
                    }
                    if (e.bneck_w) {
                        const f32x4 bw = *reinterpret_cast<const f32x4*>(e.bneck_w + c0);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            dot[q] += ((y[q][0] * bw[0] + y[q][1] * bw[1]) + y[q][2] * bw[2]) + y[q][3] * bw[3];
                    }
                    // a lane's 2x2 outputs = one window of the following VALID 2x2 max pool
                    if (pool && (interior || (oy0 + 1 < e.H && ox0 + 1 < e.W))) {
                        f32x4 mx;
#pragma unroll
                        for (int k = 0; k < 4; ++k) mx[k] = fmaxf(fmaxf(y[0][k], y[1][k]), fmaxf(y[2][k], y[3][k]));
                        float* dst = e.pool_out + (size_t)it.frame * e.pool_frame_stride +
                                     (size_t)((it.ntile * BN + cb * 16) >> 3) * pplane + pcell;
                        *reinterpret_cast<f32x4*>(dst) = mx;
                    }
                    // keep the channel blocks apart: interleaved, their 4 x 64 accumulator
                    // registers would all be live in VGPRs at once
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (e.bneck_w) {    // the other channels of these pixels live in lanes l ^ 16, ^ 32
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float s = dot[q];
                        s += __shfl_xor(s, 16, 64);
                        s += __shfl_xor(s, 32, 64);
                        const int oy = oy0 + (q >> 1), ox = ox0 + (q & 1);
                        if (g == 0 && oy < e.H && ox < e.W && oy >= e.out_y0)
                            e.bneck_out[(size_t)it.frame * e.bneck_frame_stride +
                                        (size_t)(oy - e.out_y0) * e.W + ox] =
                                fmaxf(s * e.bneck_scale + e.bneck_shift, 0.0f);
                    }
                }
            }
        }
        comp_item = q0;
    }
}

}  // namespace dodt
