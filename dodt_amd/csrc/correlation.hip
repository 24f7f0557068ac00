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
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"
#include "lds_dma.h"

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

// ---- round 4: the whole 128-byte pixels of B in ONE pass ---------------------------------------------------
// The two-pass kernel above reads every cache line of A and of B's neighbourhood twice (a pass takes one 64-byte
// half of a pixel; 64 resident tiles of an XCD stream 15 MB through its 4 MB L2 between the passes) and fetches
// the 24 x 40 patch of a 16 x 32 tile without much help from its neighbours: 307 MB fetched + 56 MB written over
// the fabric for 199 MB of algorithmic bytes (profiles/r3_hbm_summary.md).  This form:
//   * a workgroup owns a 16 x 16 tile, one pixel per lane, all 32 channels: the 24 x 24 neighbourhood of B is
//     staged once, as whole pixels, by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers; pixels outside
//     the image are outside the descriptor's range and arrive as zeros = the reference's zero padding);
//   * pixels are 128 bytes in LDS, unpadded (73.7 KB: two workgroups per CU); the 16-byte quad q of pixel p sits
//     in slot q ^ ((p >> 1) & 7) -- by permuting the SOURCE address of the lane-linear copy -- so that the 16
//     consecutive pixels one lane group of a ds_read_b128 touches fall into 16 different bank columns
//     (8 (p & 1) + (q ^ ((p >> 1) & 7)) takes every value once over any 16 consecutive p);
//   * a displacement column j (two pixels apart) shifts p >> 1 by exactly j, so the lane's eight quad
//     addresses are formed once per j and the five displacement rows are immediates (200 ds_read_b128 per
//     pixel -- the LDS array, 256 B/clk, is what the CU is bound by: ~13 us of the launch);
//   * tiles are walked along a curve of 8 x 8-tile super-blocks, each XCD a contiguous eighth of it (blocks
//     b and b + 8 share an XCD): the 64 tiles an XCD has in flight are one super-block, whose inner halos are
//     L2 hits (208 MB over the fabric for 199 MB algorithmic: profiles/r4_hbm_summary.md);
//   * workgroups are persistent: the next tile's A pixel travels under the arithmetic, two thirds of its
//     neighbourhood under the current tile's output phase (outputs leave as full rows through the first 25.6 KB
//     of the image, 16-byte stores: a lane's 25 floats stored straight from its registers, 100-byte strides, were
//     measured at 64.5 us against 49.9 for one tile per workgroup).
// Same sums in the same order as the two-pass kernel (even / odd channel chains of packed FMAs, channels
// ascending), so the results are bit-identical to it.
constexpr int kSpT = 16;                    // tile edge
constexpr int kSpR = 4;                     // neighbourhood radius: 2 displacements x stride_2 2
constexpr int kSpP = kSpT + 2 * kSpR;       // patch edge: 24 pixels
constexpr int kSpLds = kSpP * kSpP * kC * 4;

// HALVES = 1 (the default): 256 lanes, one pixel per lane, all 32 channels.  HALVES = 2 (DODT_CORR_HALVES=2): 512
// lanes, a wave's lanes 0-31 take channels 0-15 of its 32 pixels, lanes 32-63 channels 16-31, the two partial
// sums meet across the wave -- four waves per SIMD instead of two for the arithmetic phase (a wave alone issues a
// vector instruction every 4 cycles, a SIMD takes one every 2): that phase went from 4.7 to 3.6 us per tile, but
// the copy and output phases of eight waves beside the other workgroup's LDS reads from 3.5 to 8 us: 64.9 us
// against 49.2 for the kernel, not the default.
// (launch bounds: waves per SIMD; two workgroups per CU either way)
template <int HALVES>
__global__ void __launch_bounds__(256 * HALVES, 2 * HALVES)
correlation_sp_kernel(const float* __restrict__ A, const float* __restrict__ B, int H, int W, int d, int pad,
                      int OH, int OW, const int* __restrict__ tile_list, int n_tiles, float* __restrict__ out,
                      int* __restrict__ stamps) {
    constexpr int GW = 5, K = GW * GW, S2 = 2, QH = kC / 4 / HALVES;     // QH: 16-byte quads a lane takes of a pixel
    constexpr int NW = 4 * HALVES;                                      // waves
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    // (tools/, DODT_CORR_STAMPS=1: the phases of workgroup 9's first tiles on the 100 MHz clock, 8 words per tile)
    const bool stamp = stamps != nullptr && blockIdx.x == 9 && tid == 0;
    int n_stamp = 0;
#define CORR_STAMP(i) do { if (stamp && n_stamp < 6) stamps[n_stamp * 8 + (i)] = (int)__builtin_amdgcn_s_memrealtime(); } while (0)
    // Persistent workgroups, 64 per XCD (blocks b and b + 8 share one): XCD x walks positions [x per, (x + 1) per)
    // of the tile curve, workgroup i of it the positions i, i + 64, ... of that range -- so the 64 tiles an XCD
    // has in flight are 64 consecutive positions = one super-block.  The loop ends by the static stride.
    const int per = (n_tiles + 7) / 8;
    const int xcd = blockIdx.x % 8, wg = blockIdx.x / 8, stride = gridDim.x / 8;
    const int first = xcd * per, last = min(first + per, n_tiles);
    const int shift = d - pad;
    // Lane -> pixel (and channel half): every 16-lane group of a ds_read_b128 is one row of 16 consecutive pixels
    // (see above); HALVES = 1: a wave owns four tile rows, HALVES = 2: two, its upper 32 lanes the second half
    // of the channels.
    const int lane = tid & 63, wave = tid >> 6;
    const int half = HALVES == 2 ? lane >> 5 : 0;
    const int l5 = lane & 31;
    const bool grp_a = l5 < 4 || (l5 >= 12 && l5 < 16) || (l5 >= 20 && l5 < 28);
    const int lx = grp_a ? (l5 < 4 ? l5 : l5 < 16 ? l5 - 8 : l5 - 12)
                         : (l5 < 12 ? l5 - 4 : l5 < 20 ? l5 - 8 : l5 - 16);
    const int ly = (HALVES == 2 ? wave * 2 : (tid >> 5) * 2) + (grp_a ? 0 : 1);
    const dodt::i32x4_t rsrc = dodt::make_rsrc(B, (unsigned)((size_t)H * W * kC * 4));

    // the lane's A pixel of a tile, its channels (zeros outside the image)
    auto load_a = [&](int oy0, int ox0, f32x4 (&ah)[QH]) {
        const int ay = oy0 + ly + shift, ax = ox0 + lx + shift;
        const bool in = ay >= 0 && ay < H && ax >= 0 && ax < W;
#pragma unroll
        for (int q = 0; q < QH; ++q) {
            ah[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (in) ah[q] = *reinterpret_cast<const f32x4*>(A + ((size_t)ay * W + ax) * kC + (half * QH + q) * 4);
        }
    };
    // B's neighbourhood of a tile: 576 pixels x 8 quads = 72 wave-level copies of 1 KB (8 pixels each), 72 / NW per wave
    // (`late`: the copies whose destination is the first 25 KB of the neighbourhood, where a tile's outputs are
    //  staged for their row-wise stores; the others may be issued as soon as the arithmetic is done)
    constexpr int kOutCopies = (kSpT * kSpT * K * 4 + 1023) / 1024;      // 25 wave-level copies = 25.6 KB
    auto stage_b = [&](int oy0, int ox0, auto part) {
        constexpr int PART = decltype(part)::value;      // 0: all, 1: early (j >= kOutCopies), 2: late
        // (opaque copy of the lane id: the copies' pixel coordinates are recomputed per tile instead of living
        //  in registers across the arithmetic)
        int slane = lane;
        asm volatile("" : "+v"(slane));
#pragma unroll
        for (int k = 0; k < (kSpP * kSpP * 8 / 64) / NW; ++k) {
            const int j = wave + NW * k;
            // (wave-uniform)
            if (PART == 1 && j < kOutCopies) continue;
            if (PART == 2 && j >= kOutCopies) continue;
            const int s = j * 64 + slane;
            const int p = s >> 3, q = (s & 7) ^ ((p >> 1) & 7);
            const int py = p / kSpP, px = p - py * kSpP;
            const int gy = oy0 + shift - kSpR + py, gx = ox0 + shift - kSpR + px;
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            dodt::blds16(rsrc, ok ? ((gy * W + gx) * kC + q * 4) * 4 : dodt::kOob, 0, smem + j * 256);
        }
    };
    using All = std::integral_constant<int, 0>;
    using Early = std::integral_constant<int, 1>;
    using Late = std::integral_constant<int, 2>;
    int slot = first + wg;
    if (slot >= last) return;
    int tile = tile_list[slot];
    f32x4 ah[QH];
    load_a((tile >> 16) * kSpT, (tile & 0xffff) * kSpT, ah);
    stage_b((tile >> 16) * kSpT, (tile & 0xffff) * kSpT, All{});
    const int p0_lane = ly * kSpP + lx;
    while (true) {
        const int oy0 = (tile >> 16) * kSpT, ox0 = (tile & 0xffff) * kSpT;
        // (opaque per tile: the quad addresses of the displacement columns are formed again for every tile
        //  instead of held in registers across the loop)
        int p0 = p0_lane;
        asm volatile("" : "+v"(p0));
        const char* lane_base = reinterpret_cast<const char*>(smem) + p0 * (kC * 4);
        CORR_STAMP(0);
        __builtin_amdgcn_s_waitcnt(0);      // A in registers, this wave's copies in LDS (and the last tile's stores gone)
        CORR_STAMP(1);
        __syncthreads();
        CORR_STAMP(2);
        // the next tile's A pixel travels under this tile's arithmetic
        const int nslot = slot + stride;
        const bool more = nslot < last;
        const int ntile = more ? tile_list[nslot] : 0;
        f32x4 an[QH];
        if (more) load_a((ntile >> 16) * kSpT, (ntile & 0xffff) * kSpT, an);
        f32x2 res[K];
#pragma unroll
        for (int k = 0; k < K; ++k) res[k] = f32x2{0.f, 0.f};
        // 25 steps (displacement column j, row i), the neighbourhood pixel of step s + 1 read from LDS before the
        // multiply-adds of step s are issued (pinned by scheduling barriers)
        f32x4 bv[2][QH];
        auto read_step = [&](int st, f32x4 (&dst)[QH]) {
            const int j = st / GW, i = st % GW;
            const int sj = ((p0 >> 1) + j) & 7;      // swizzle of the pixels of displacement column j (any row: 24 i = 0 mod 8)
#pragma unroll
            for (int q = 0; q < QH; ++q)
                dst[q] = *reinterpret_cast<const f32x4*>(lane_base + (((half * QH + q) ^ sj) << 4) + S2 * j * (kC * 4) +
                                                         S2 * i * kSpP * (kC * 4));
        };
        read_step(0, bv[0]);
#pragma unroll
        for (int st = 0; st < K; ++st) {
            if (st + 1 < K) read_step(st + 1, bv[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int j = st / GW, i = st % GW;
            // two chains per output (even / odd channels of the lane's half), plain v_fma_f32 (a packed fp32
            // multiply-add is no faster here; this file is built with -fno-slp-vectorize so that they stay plain)
            float s0 = res[i * GW + j][0], s1 = res[i * GW + j][1];
#pragma unroll
            for (int q = 0; q < QH; ++q) {
                s0 = __builtin_fmaf(ah[q][0], bv[st & 1][q][0], s0);
                s1 = __builtin_fmaf(ah[q][1], bv[st & 1][q][1], s1);
                s0 = __builtin_fmaf(ah[q][2], bv[st & 1][q][2], s0);
                s1 = __builtin_fmaf(ah[q][3], bv[st & 1][q][3], s1);
            }
            res[i * GW + j] = f32x2{s0, s1};
            __builtin_amdgcn_sched_barrier(0);
        }
        // HALVES = 2: the pixel's sum = (even + odd chain of channels 0-15) + (even + odd chain of channels 16-31),
        // the other half's value comes across the wave (lane ^ 32)
        float tot[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            tot[k] = res[k][0] + res[k][1];
            if (HALVES == 2) {
                const float other = __shfl_xor(tot[k], 32, 64);
                tot[k] = half == 0 ? tot[k] + other : other + tot[k];
            }
        }
        // (the sums are only used by the guarded stores below; without this use the compiler sinks the
        //  multiply-adds into that branch, behind the barrier, and keeps the loaded quads alive in scratch)
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("" : "+v"(tot[k]));
        CORR_STAMP(3);
        __syncthreads();   // every wave has read its last pixel of this neighbourhood
        CORR_STAMP(4);
        // the next neighbourhood lands: first the part behind the 25.6 KB the outputs are staged in ...
        if (more) stage_b((ntile >> 16) * kSpT, (ntile & 0xffff) * kSpT, Early{});
        if (half == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) smem[(ly * kSpT + lx) * K + k] = tot[k] * (1.0f / (float)kC);   // exact: a power of two
        }
        __syncthreads();
        CORR_STAMP(5);
        {   // rows of the tile are contiguous in the output: 400 floats each, 16-byte stores
            constexpr int kRow = kSpT * K;
            if ((OW * K) % 4 == 0 && ox0 + kSpT <= OW) {
                for (int t = tid; t < kSpT * (kRow / 4); t += 256 * HALVES) {
                    const int row = t / (kRow / 4), off = t - row * (kRow / 4);
                    const int y = oy0 + row;
                    if (y < OH)
                        *reinterpret_cast<f32x4*>(out + ((size_t)y * OW + ox0) * K + off * 4) =
                            *reinterpret_cast<const f32x4*>(smem + row * kRow + off * 4);
                }
            } else {
                for (int t = tid; t < kSpT * kRow; t += 256 * HALVES) {
                    const int row = t / kRow, off = t - row * kRow;
                    const int y = oy0 + row, x = ox0 + off / K;
                    if (y < OH && x < OW) out[((size_t)y * OW + ox0) * K + off] = smem[t];
                }
            }
        }
        CORR_STAMP(6);
        if (!more) break;
        __syncthreads();   // ... then, the staged outputs read by everyone, the part in front
        stage_b((ntile >> 16) * kSpT, (ntile & 0xffff) * kSpT, Late{});
        CORR_STAMP(7);
        ++n_stamp;
        slot = nslot;
        tile = ntile;
#pragma unroll
        for (int q = 0; q < QH; ++q) ah[q] = an[q];
    }
#undef CORR_STAMP
}

// tiles of a (tiles_y x tiles_x) grid along the super-block curve: 8 x 8-tile blocks in raster order, raster
// order inside a block; entry = ty << 16 | tx.  One list per grid shape and device, made on first use.
struct SpTileList {
    int device, tiles_y, tiles_x;
    int* d_list;
};
int sp_tile_list(dodt_ctx* ctx, int tiles_y, int tiles_x, const int** out) {
    static std::mutex mu;
    static std::vector<SpTileList> cache;
    std::lock_guard<std::mutex> lock(mu);
    for (const SpTileList& e : cache)
        if (e.device == ctx->device && e.tiles_y == tiles_y && e.tiles_x == tiles_x) {
            *out = e.d_list;
            return DODT_OK;
        }
    std::vector<int> h;
    h.reserve((size_t)tiles_y * tiles_x);
    for (int by = 0; by < tiles_y; by += 8)
        for (int bx = 0; bx < tiles_x; bx += 8)
            for (int ty = by; ty < std::min(by + 8, tiles_y); ++ty)
                for (int tx = bx; tx < std::min(bx + 8, tiles_x); ++tx) h.push_back(ty << 16 | tx);
    int* d = nullptr;
    DODT_HIP_CHECK(hipMalloc(&d, h.size() * sizeof(int)));
    // (pageable source: the copy has completed on the host's side when this returns)
    DODT_HIP_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
    cache.push_back(SpTileList{ctx->device, tiles_y, tiles_x, d});
    *out = d;
    return DODT_OK;
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
    // the configuration's 5 x 5 grid: whole pixels in one pass (DODT_CORR_TWO_PASS=1: round 3's kernel)
    static const bool two_pass = getenv("DODT_CORR_TWO_PASS") && atoi(getenv("DODT_CORR_TWO_PASS")) != 0;
    if (!two_pass && (size_t)H * W * kC * 4 < (1ull << 31) && H < 65536 * kSpT && W < 65536 * kSpT) {
        const int ty = dodt::ceil_div(OH, kSpT), tx = dodt::ceil_div(OW, kSpT);
        const int* d_list = nullptr;
        if (int rc = sp_tile_list(ctx, ty, tx, &d_list)) return rc;
        static const int halves = getenv("DODT_CORR_HALVES") && atoi(getenv("DODT_CORR_HALVES")) == 2 ? 2 : 1;
        static bool prepared = false;
        if (!prepared) {
            DODT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&correlation_sp_kernel<1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kSpLds));
            DODT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&correlation_sp_kernel<2>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kSpLds));
            prepared = true;
        }
        const int n = ty * tx;
        // two resident workgroups per CU, a multiple of 8 (an equal share per XCD), no more than there are tiles
        static const int grid_env = getenv("DODT_CORR_GRID") ? atoi(getenv("DODT_CORR_GRID")) : 0;   // (tools/: A/B)
        int grid = grid_env > 0 ? grid_env / 8 * 8 : 2 * ctx->num_cus / 8 * 8;
        if (grid > 8 * dodt::ceil_div(n, 8)) grid = 8 * dodt::ceil_div(n, 8);
        static const bool want_stamps = getenv("DODT_CORR_STAMPS") != nullptr;
        static int* d_stamps = nullptr;
        if (want_stamps && !d_stamps) {
            DODT_HIP_CHECK(hipMalloc(&d_stamps, 64 * sizeof(int)));
            DODT_HIP_CHECK(hipMemset(d_stamps, 0, 64 * sizeof(int)));
        }
        if (halves == 2)
            hipLaunchKernelGGL(correlation_sp_kernel<2>, dim3(grid), dim3(512), kSpLds, ctx->stream,
                               d_a, d_b, H, W, max_displacement, pad, OH, OW, d_list, n, d_out, d_stamps);
        else
            hipLaunchKernelGGL(correlation_sp_kernel<1>, dim3(grid), dim3(256), kSpLds, ctx->stream,
                               d_a, d_b, H, W, max_displacement, pad, OH, OW, d_list, n, d_out, d_stamps);
        DODT_LAUNCH_CHECK();
        if (want_stamps) {
            int h[64];
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "[dodt] correlation, workgroup 9, us per phase: wait copies | barrier | arithmetic | barrier | "
                            "early copies + outputs to LDS | stores | barrier + late copies\n");
            for (int t = 0; t < 4; ++t) {
                fprintf(stderr, "[dodt]   tile %d:", t);
                for (int k = 0; k < 7; ++k) fprintf(stderr, " %6.2f", (h[t * 8 + k + 1] - h[t * 8 + k]) / 100.0);
                fprintf(stderr, "   next tile's top +%.2f\n", (h[(t + 1) * 8] - h[t * 8]) / 100.0);
            }
        }
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
