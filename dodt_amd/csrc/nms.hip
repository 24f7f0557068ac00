// Greedy NMS for gfx950: tf.image.non_max_suppression of
// TF 1.3 (non_max_suppression_op.cc), call sites avod/core/models/
// dt_rpn_model.py:587-591 (A anchors, k = 300/1024, thr 0.8) and
// models/dt_avod_model.py:606-613 (P proposals, k = 100, thr 0.01).
//
// Structure (64-wide wave ballots, no 32-lane idioms):
//   1. keys: 64-bit (descending score, ascending index) keys.  n <= 32768: every key's rank is
//      COUNTED (n^2 compares spread over the chip: a (256 keys) x (512-key LDS segment) workgroup
//      per pair, 3-8 us instead of the 60 us of a one-workgroup bitonic sort of 8192 keys) and the
//      boxes are scattered to their rank; above: bitonic sort, the short strides inside one
//      workgroup's LDS (8192 keys = 64 KB per tile).
//   2. boxes are gathered in sorted order with min/max-normalised corners + area.
//   3. rows are processed in chunks (the first just over max_output_size rows -- with the RPN's
//      threshold of 0.8 nearly every candidate survives, so that is where the scan ends --, the
//      others 2048 / 4096) so that the suppression matrix stays small:
//        nms_mask_kernel : one wave per (64-row block, column blocks); lane = column box, the 64
//          row boxes are broadcast from LDS and each `iou > thr` test becomes one __ballot -> the
//          row's 64-bit word; on the diagonal block the lane also keeps its own hits = the
//          block's COLUMN form (which earlier rows of the block suppress this box).
//        nms_scan_kernel : one workgroup walks the chunk 64 rows at a time.  Wave 0 resolves the
//          64x64 diagonal block as a fixed point of  alive[j] = init[j] & !(col[j] & alive)  -- one
//          AND + ballot per iteration, exact after as many iterations as the longest suppression
//          chain (bits 0..t-1 are final after t), against 64 dependent readlane steps for the
//          serial form --, then all 16 waves OR the selected rows into the `removed` bit vector
//          held in LDS; those rows' loads do not depend on the selection and are issued BEFORE
//          the block is resolved.  It stops at max_output_size and raises a flag that turns the
//          remaining chunk launches into no-ops.
// IoU arithmetic is float32, unfused, in TF's operation order.
#include "common.h"

namespace {

constexpr int kTile = 8192;      // keys per LDS sort tile
constexpr int kSortThreads = 1024;
constexpr int kChunkRows = 2048; // candidate rows per mask chunk (4096 while a row is <= 512 words)
constexpr int kColGroup = 1;     // column blocks per wave in the mask kernel
constexpr int kRankMax = 32768;  // up to here the order comes from counted ranks
constexpr int kRankSeg = 512;    // keys per LDS segment of the rank count

struct NmsState {   // lives in device memory next to the scratch arrays
    int count;      // boxes selected so far
    int done;       // 1 once max_out is reached or candidates are exhausted
    int n_eff;      // min(*d_n, n)
    int pad;
};

__device__ __forceinline__ uint32_t ordered_bits(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // monotone increasing in f
}

__global__ void __launch_bounds__(256)
nms_keys_kernel(const float* __restrict__ scores, int n, const int* __restrict__ d_n, int n_pad,
                unsigned long long* __restrict__ keys, NmsState* __restrict__ st,
                unsigned long long* __restrict__ removed, int nb, int* __restrict__ rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_eff = d_n ? min(*d_n, n) : n;
    if (i == 0) { st->count = 0; st->done = (n_eff <= 0); st->n_eff = n_eff; }
    if (i < nb) removed[i] = 0ull;
    if (i >= n_pad) return;
    if (rank) rank[i] = 0;
    unsigned long long k = ~0ull;
    if (i < n_eff) k = ((unsigned long long)(~ordered_bits(scores[i])) << 32) | (uint32_t)i;
    keys[i] = k;
}

__device__ __forceinline__ void cmp_swap(unsigned long long& a, unsigned long long& b, bool up) {
    if ((a > b) == up) { const unsigned long long t = a; a = b; b = t; }
}

// Sorts each kTile-key tile completely (all bitonic stages with k <= tile size).
// (Round 3 rebuilt this with eight keys per thread in registers, wave shuffles for the strides inside a
//  wave and LDS only for strides >= 512 -- 10 barrier steps instead of 91: the same 55-60 us for an
//  8192-key tile.  What bounds a one-workgroup bitonic sort is the 370k 64-bit compare-exchanges on one
//  CU's vector ALUs, not the barriers; a faster NMS #1 needs fewer keys (select the top ~2k by a
//  histogram of the score bits first) or many CUs, not a leaner network.  tools/nms_bench.py times it.)
__global__ void __launch_bounds__(kSortThreads)
nms_sort_local(unsigned long long* __restrict__ keys, int n_pad) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];
    const int tile = min(kTile, n_pad);
    const size_t base = (size_t)blockIdx.x * tile;
    for (int t = threadIdx.x; t < tile; t += kSortThreads) s_keys[t] = keys[base + t];
    __syncthreads();
    for (int k = 2; k <= tile; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < tile / 2; t += kSortThreads) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo + j;
                const bool up = (((base + lo) & (size_t)k) == 0);
                unsigned long long a = s_keys[lo], b = s_keys[hi];
                cmp_swap(a, b, up);
                s_keys[lo] = a;
                s_keys[hi] = b;
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < tile; t += kSortThreads) keys[base + t] = s_keys[t];
}

// One global bitonic step (stride j >= kTile) of stage k.
__global__ void __launch_bounds__(256)
nms_sort_global_step(unsigned long long* __restrict__ keys, int n_pad, int k, int j) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pad / 2) return;
    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    const int hi = lo + j;
    unsigned long long a = keys[lo], b = keys[hi];
    const bool up = ((lo & k) == 0);
    if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
}

// Remaining steps of stage k with stride < kTile, inside LDS.
__global__ void __launch_bounds__(kSortThreads)
nms_sort_local_merge(unsigned long long* __restrict__ keys, int n_pad, int k) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];
    const size_t base = (size_t)blockIdx.x * kTile;
    for (int t = threadIdx.x; t < kTile; t += kSortThreads) s_keys[t] = keys[base + t];
    __syncthreads();
    for (int j = kTile >> 1; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < kTile / 2; t += kSortThreads) {
            const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
            const int hi = lo + j;
            const bool up = (((base + lo) & (size_t)k) == 0);
            unsigned long long a = s_keys[lo], b = s_keys[hi];
            cmp_swap(a, b, up);
            s_keys[lo] = a;
            s_keys[hi] = b;
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < kTile; t += kSortThreads) keys[base + t] = s_keys[t];
}

// rank[i] += number of keys of segment blockIdx.y below key i (keys are distinct: the index is
// their low word, so the ranks of the n_eff valid keys are a permutation of 0..n_eff-1; pad keys
// are ~0 and count for nobody)
__global__ void __launch_bounds__(256)
nms_rank_kernel(const unsigned long long* __restrict__ keys, const NmsState* __restrict__ st, int n_pad,
                int* __restrict__ rank) {
    __shared__ __attribute__((aligned(16))) unsigned long long s_seg[kRankSeg];
    const int n_eff = st->n_eff;
    const int i0 = blockIdx.x * 256, j0 = blockIdx.y * kRankSeg;
    if (i0 >= n_eff || j0 >= n_eff) return;
    for (int t = threadIdx.x; t < kRankSeg; t += 256) s_seg[t] = (j0 + t < n_pad) ? keys[j0 + t] : ~0ull;
    __syncthreads();
    const int i = i0 + threadIdx.x;
    if (i >= n_eff) return;
    const unsigned long long mine = keys[i];
    int cnt = 0;
#pragma unroll 16
    for (int j = 0; j < kRankSeg; ++j) cnt += (s_seg[j] < mine) ? 1 : 0;
    if (cnt) atomicAdd(&rank[i], cnt);
}

// box i -> row rank[i] of the sorted arrays; rows n_eff.. get the empty box
__global__ void __launch_bounds__(256)
nms_scatter_kernel(const float* __restrict__ boxes, const int* __restrict__ rank,
                   const NmsState* __restrict__ st, int n_rows, float* __restrict__ sb,
                   int* __restrict__ sorted_idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    float ymin = 0, xmin = 0, ymax = 0, xmax = 0, area = -1.0f;
    int idx = -1, row = i;
    if (i < st->n_eff) {
        idx = i;
        row = rank[i];
        const float4 b = reinterpret_cast<const float4*>(boxes)[i];
        ymin = fminf(b.x, b.z); ymax = fmaxf(b.x, b.z);
        xmin = fminf(b.y, b.w); xmax = fmaxf(b.y, b.w);
        area = (ymax - ymin) * (xmax - xmin);
    }
    float* o = sb + (size_t)row * 5;
    o[0] = ymin; o[1] = xmin; o[2] = ymax; o[3] = xmax; o[4] = area;
    sorted_idx[row] = idx;
}

struct SBox { float ymin, xmin, ymax, xmax, area; };

__global__ void __launch_bounds__(256)
nms_gather_kernel(const float* __restrict__ boxes, const unsigned long long* __restrict__ keys,
                  const NmsState* __restrict__ st, int n_rows, float* __restrict__ sb,
                  int* __restrict__ sorted_idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    float ymin = 0, xmin = 0, ymax = 0, xmax = 0, area = -1.0f;
    int idx = -1;
    if (i < st->n_eff) {
        idx = (int)(keys[i] & 0xFFFFFFFFull);
        const float4 b = reinterpret_cast<const float4*>(boxes)[idx];
        ymin = fminf(b.x, b.z); ymax = fmaxf(b.x, b.z);
        xmin = fminf(b.y, b.w); xmax = fmaxf(b.y, b.w);
        area = (ymax - ymin) * (xmax - xmin);
    }
    float* o = sb + (size_t)i * 5;
    o[0] = ymin; o[1] = xmin; o[2] = ymax; o[3] = xmax; o[4] = area;
    sorted_idx[i] = idx;
}

__device__ __forceinline__ bool iou_gt(float ymin_i, float xmin_i, float ymax_i, float xmax_i,
                                       float area_i, float ymin_j, float xmin_j, float ymax_j,
                                       float xmax_j, float area_j, float thr) {
    if (area_i <= 0.0f || area_j <= 0.0f) return false;
    const float iy = fmaxf(fminf(ymax_i, ymax_j) - fmaxf(ymin_i, ymin_j), 0.0f);
    const float ix = fmaxf(fminf(xmax_i, xmax_j) - fmaxf(xmin_i, xmin_j), 0.0f);
    const float inter = iy * ix;
    const float iou = inter / ((area_i + area_j) - inter);
    return iou > thr;
}

// mask[(row - row0) * nb + cb] bit l  <=>  sorted box (cb*64 + l) is suppressed by
// sorted box `row` (only l with cb*64 + l > row are set).
__global__ void __launch_bounds__(256)
nms_mask_kernel(const float* __restrict__ sb, const NmsState* __restrict__ st, int row0, int rows,
                int nb, float thr, unsigned long long* __restrict__ mask,
                unsigned long long* __restrict__ diag_cols) {
    if (st->done) return;
    __shared__ float s_rows[4][64][5];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rb = (row0 >> 6) + blockIdx.y * 4 + wave;  // global row block
    const int n_eff = st->n_eff;
    const int row = rb * 64 + lane;
    {
        const float* p = sb + (size_t)row * 5;
        const bool ok = row < n_eff;
#pragma unroll
        for (int k = 0; k < 5; ++k) s_rows[wave][lane][k] = ok ? p[k] : (k == 4 ? -1.0f : 0.0f);
    }
    __syncthreads();
    if (rb * 64 >= n_eff || rb * 64 >= row0 + rows) return;
    const int cb0 = blockIdx.x * kColGroup;
    unsigned long long words[kColGroup];
#pragma unroll
    for (int g = 0; g < kColGroup; ++g) {
        const int cb = cb0 + g;
        unsigned long long mine = 0ull;
        if (cb >= rb && cb < nb) {  // wave-uniform
            const int col = cb * 64 + lane;
            float cy0 = 0, cx0 = 0, cy1 = 0, cx1 = 0, ca = -1.0f;
            if (col < n_eff) {
                const float* p = sb + (size_t)col * 5;
                cy0 = p[0]; cx0 = p[1]; cy1 = p[2]; cx1 = p[3]; ca = p[4];
            }
            unsigned long long hits = 0ull;     // bit r: row r of the block suppresses this column
            for (int r = 0; r < 64; ++r) {
                const float* q = s_rows[wave][r];
                const bool hit = (col > rb * 64 + r) &&
                                 iou_gt(q[0], q[1], q[2], q[3], q[4], cy0, cx0, cy1, cx1, ca, thr);
                const unsigned long long w = __ballot(hit);
                if (lane == r) mine = w;
                hits |= hit ? (1ull << r) : 0ull;
            }
            if (cb == rb) diag_cols[col] = hits;     // col < nb * 64 = the array's size
        }
        words[g] = mine;
    }
    if (row < n_eff) {
        unsigned long long* o = mask + (size_t)(row - row0) * nb + cb0;
#pragma unroll
        for (int g = 0; g < kColGroup; ++g)
            if (cb0 + g < nb) o[g] = words[g];
    }
}

constexpr int kScanDepth = 4;    // row blocks whose words are in flight ahead of the resolution

// Dynamic LDS: removed[nb] | cols[rows] (column form of the chunk's diagonal blocks) | kept[rows / 64].
__global__ void __launch_bounds__(1024)
nms_scan_kernel(const unsigned long long* __restrict__ mask,
                const unsigned long long* __restrict__ diag_cols, const int* __restrict__ sorted_idx,
                NmsState* __restrict__ st, unsigned long long* __restrict__ g_removed, int row0,
                int rows, int nb, int max_out, int* __restrict__ sel_out,
                int* __restrict__ count_out, int last_chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_dyn[];
    unsigned long long* const s_removed = s_dyn;
    unsigned long long* const s_cols = s_dyn + nb;
    unsigned long long* const s_kept = s_cols + rows;
    __shared__ unsigned long long s_sel;
    __shared__ int s_count, s_done;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_eff = st->n_eff;
    if (st->done) {
        if (last_chunk && tid == 0) *count_out = st->count;
        return;
    }
    const int b0 = row0 >> 6;
    const int b1 = min((row0 + rows + 63) >> 6, (n_eff + 63) >> 6);     // <= b0 + 64 blocks
    // Words of block b's rows for the chunk's later column blocks (the ones the scan meets before the chunk
    // ends): lane = column block b + 1 + lane, wave = rows wave, wave + 16, wave + 32, wave + 48.  The loads
    // do not depend on the selection: they go out kScanDepth - 1 blocks ahead and are masked afterwards.
    auto load_block = [&](int b, unsigned long long (&v)[4]) {
        const int c = b + 1 + lane;
        const bool ok = b < b1 && c < b1;
        const unsigned long long* base = mask + (size_t)(b * 64 - row0 + wave) * nb + c;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ok ? base[(size_t)(16 * k) * nb] : 0ull;
    };
    unsigned long long v[kScanDepth][4];
#pragma unroll
    for (int d = 0; d < kScanDepth - 1; ++d) load_block(b0 + d, v[d]);
    for (int c = tid; c < nb; c += 1024) s_removed[c] = g_removed[c];
    for (int r = tid; r < (b1 - b0) * 64; r += 1024) s_cols[r] = diag_cols[b0 * 64 + r];
    if (tid < b1 - b0) s_kept[tid] = 0ull;
    if (tid == 0) { s_count = st->count; s_done = 0; }
    __syncthreads();

    auto block = [&](auto stage, int b) -> bool {
        constexpr int S = decltype(stage)::value;
        load_block(b + kScanDepth - 1, v[(S + kScanDepth - 1) % kScanDepth]);
        if (tid < 64) {  // wave 0 resolves the diagonal 64 x 64 block
            const int row = b * 64 + lane;
            const unsigned long long cols = s_cols[(b - b0) * 64 + lane];   // rows of the block that hit my box
            const int valid = min(64, n_eff - b * 64);
            unsigned long long init = ~s_removed[b];
            if (valid < 64) init &= (1ull << valid) - 1ull;
            const bool mine = (init >> lane) & 1ull;
            // greedy selection = the fixed point of alive[j] = init[j] & no alive earlier row hits j
            unsigned long long alive = __ballot(mine);
            for (;;) {
                const unsigned long long na = __ballot(mine && (cols & alive) == 0ull);
                if (na == alive) break;
                alive = na;
            }
            const int cnt0 = __builtin_amdgcn_readfirstlane(s_count);
            const int room = max_out - cnt0;         // > 0: the scan stops when it reaches 0
            const unsigned long long below = (1ull << lane) - 1ull;
            unsigned long long sel = alive;
            if (__popcll(sel) > room)                // wave-uniform: the first `room` of them
                sel = __ballot(((sel >> lane) & 1ull) && __popcll(sel & below) < room);
            const int cnt = cnt0 + __popcll(sel);
            if ((sel >> lane) & 1ull) sel_out[cnt0 + __popcll(sel & below)] = sorted_idx[row];
            if (lane == 0) {
                s_sel = sel;
                s_kept[b - b0] = sel;
                s_count = cnt;
                if (cnt >= max_out) s_done = 1;
            }
        }
        __syncthreads();
        const unsigned long long sel = s_sel;
        const bool done = s_done != 0;
        if (!done && sel != 0ull) {
            unsigned long long acc = 0ull;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((sel >> (wave + 16 * k)) & 1ull) acc |= v[S][k];
            if (acc) atomicOr(&s_removed[b + 1 + lane], acc);     // (acc != 0 only for columns < b1)
        }
        __syncthreads();
        return done;
    };
    bool stop = false;
    for (int b = b0; b < b1 && !stop; b += kScanDepth) {
        stop = block(std::integral_constant<int, 0>{}, b);
        if (!stop && b + 1 < b1) stop = block(std::integral_constant<int, 1>{}, b + 1);
        if (!stop && b + 2 < b1) stop = block(std::integral_constant<int, 2>{}, b + 2);
        if (!stop && b + 3 < b1) stop = block(std::integral_constant<int, 3>{}, b + 3);
    }
    static_assert(kScanDepth == 4, "the loop above is unrolled by hand");

    // Candidates left and room left: the chunk's selected rows go into `removed` for the column blocks of the
    // later chunks, in one sweep (lane = column, wave = every 16th row, eight loads in flight per thread).
    if (!stop && row0 + rows < n_eff) {
        const int nrows = (b1 - b0) * 64;
        for (int ct = b1; ct < nb; ct += 64) {
            const int c = ct + lane;
            if (c >= nb) continue;
            const unsigned long long* base = mask + (size_t)(b0 * 64 - row0) * nb + c;
            unsigned long long acc = 0ull;
            for (int r0 = wave; r0 < nrows; r0 += 16 * 8) {
                unsigned long long w[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) w[k] = base[(size_t)min(r0 + 16 * k, nrows - 1) * nb];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = r0 + 16 * k;
                    if (r < nrows && ((s_kept[r >> 6] >> (r & 63)) & 1ull)) acc |= w[k];
                }
            }
            if (acc) atomicOr(&s_removed[c], acc);
        }
        __syncthreads();
    }
    for (int c = tid; c < nb; c += 1024) g_removed[c] = s_removed[c];
    if (tid == 0) {
        st->count = s_count;
        const int exhausted = (row0 + rows >= n_eff);
        if (s_done || exhausted) st->done = 1;
        if (last_chunk || s_done || exhausted) *count_out = s_count;
    }
}

int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

extern "C" int dodt_nms(dodt_ctx* ctx, const float* d_boxes, const float* d_scores, int n,
                        const int32_t* d_n, int max_out, float iou_threshold, int32_t* d_sel_out,
                        int32_t* d_count_out) {
    DODT_REQUIRE(ctx && d_sel_out && d_count_out && (n == 0 || (d_boxes && d_scores)),
                 "dodt_nms: NULL argument");
    DODT_REQUIRE(n >= 0 && max_out >= 0, "dodt_nms: negative size");
    DODT_REQUIRE(n <= 400000, "dodt_nms: n = %d exceeds 400000", n);
    if (n == 0 || max_out == 0) {
        DODT_HIP_CHECK(hipMemsetAsync(d_count_out, 0, sizeof(int32_t), ctx->stream));
        return DODT_OK;
    }
    const int n_pad = next_pow2(n < 2 ? 2 : n);
    const int nb = dodt::ceil_div(n, 64);
    const int n_rows = nb * 64;
    const bool ranked = n <= kRankMax;
    // chunks of candidate rows: the first just over max_out (all of them up to 2048 rows: one mask + scan pair
    // then), the others as large as a 32 MB matrix allows, at most 4096 (the scan's 64 in-chunk column blocks)
    const int later_rows = nb <= 1024 ? 2 * kChunkRows : kChunkRows;
    int first_rows = (int)dodt::align_up((size_t)max_out + (size_t)max_out / 8, 64) + 64;
    if (first_rows > later_rows) first_rows = later_rows;
    if (first_rows > n_rows || n_rows <= kChunkRows) first_rows = n_rows;
    // two chunks instead of three where a somewhat longer first one does it (the RPN's 5 760 rows: 1 664 + 4 096)
    if (n_rows - first_rows > later_rows && n_rows - later_rows <= later_rows) first_rows = n_rows - later_rows;
    const int chunk_rows = n_rows < later_rows ? n_rows : later_rows;    // the largest chunk
    // scratch layout
    size_t off = 0;
    const size_t o_state = off; off += 256;
    const size_t o_keys = off; off += dodt::align_up((size_t)n_pad * 8, 256);
    const size_t o_rank = off; off += dodt::align_up((size_t)n_pad * 4, 256);
    const size_t o_sb = off; off += dodt::align_up((size_t)n_rows * 5 * 4, 256);
    const size_t o_idx = off; off += dodt::align_up((size_t)n_rows * 4, 256);
    const size_t o_removed = off; off += dodt::align_up((size_t)nb * 8, 256);
    const size_t o_diag = off; off += dodt::align_up((size_t)n_rows * 8, 256);
    const size_t o_mask = off; off += (size_t)chunk_rows * nb * 8;
    int rc = ctx->nms_ws.reserve(off);
    if (rc) return rc;
    char* ws = reinterpret_cast<char*>(ctx->nms_ws.ptr);
    NmsState* st = reinterpret_cast<NmsState*>(ws + o_state);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws + o_keys);
    int* rank = reinterpret_cast<int*>(ws + o_rank);
    float* sb = reinterpret_cast<float*>(ws + o_sb);
    int* sorted_idx = reinterpret_cast<int*>(ws + o_idx);
    unsigned long long* removed = reinterpret_cast<unsigned long long*>(ws + o_removed);
    unsigned long long* diag_cols = reinterpret_cast<unsigned long long*>(ws + o_diag);
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(ws + o_mask);
    hipStream_t s = ctx->stream;

    const int init_n = n_pad > nb ? n_pad : nb;
    hipLaunchKernelGGL(nms_keys_kernel, dim3(dodt::ceil_div(init_n, 256)), dim3(256), 0, s,
                       d_scores, n, d_n, n_pad, keys, st, removed, nb, ranked ? rank : (int*)nullptr);
    DODT_LAUNCH_CHECK();
    if (ranked) {
        hipLaunchKernelGGL(nms_rank_kernel, dim3(dodt::ceil_div(n, 256), dodt::ceil_div(n, kRankSeg)),
                           dim3(256), 0, s, keys, st, n_pad, rank);
        DODT_LAUNCH_CHECK();
        hipLaunchKernelGGL(nms_scatter_kernel, dim3(dodt::ceil_div(n_rows, 256)), dim3(256), 0, s,
                           d_boxes, rank, st, n_rows, sb, sorted_idx);
        DODT_LAUNCH_CHECK();
    } else {
        const int tile = n_pad < kTile ? n_pad : kTile;
        hipLaunchKernelGGL(nms_sort_local, dim3(n_pad / tile), dim3(kSortThreads), (size_t)tile * 8, s,
                           keys, n_pad);
        DODT_LAUNCH_CHECK();
        for (int k = 2 * kTile; k <= n_pad; k <<= 1) {
            for (int j = k >> 1; j >= kTile; j >>= 1) {
                hipLaunchKernelGGL(nms_sort_global_step, dim3(dodt::ceil_div(n_pad / 2, 256)),
                                   dim3(256), 0, s, keys, n_pad, k, j);
                DODT_LAUNCH_CHECK();
            }
            hipLaunchKernelGGL(nms_sort_local_merge, dim3(n_pad / kTile), dim3(kSortThreads),
                               (size_t)kTile * 8, s, keys, n_pad, k);
            DODT_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(nms_gather_kernel, dim3(dodt::ceil_div(n_rows, 256)), dim3(256), 0, s,
                           d_boxes, keys, st, n_rows, sb, sorted_idx);
        DODT_LAUNCH_CHECK();
    }
    const int col_groups = dodt::ceil_div(nb, kColGroup);
    const size_t scan_lds = ((size_t)nb + chunk_rows + chunk_rows / 64) * 8;      // <= 84 KB at n = 400000
    static const hipError_t lds_attr = hipFuncSetAttribute(
        reinterpret_cast<const void*>(nms_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    DODT_HIP_CHECK(lds_attr);
    for (int row0 = 0; row0 < n_rows;) {
        const int want = row0 == 0 ? first_rows : later_rows;
        const int rows = (n_rows - row0) < want ? (n_rows - row0) : want;
        hipLaunchKernelGGL(nms_mask_kernel, dim3(col_groups, dodt::ceil_div(rows / 64, 4)),
                           dim3(256), 0, s, sb, st, row0, rows, nb, iou_threshold, mask, diag_cols);
        DODT_LAUNCH_CHECK();
        const int last = (row0 + rows >= n_rows);
        hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(1024), scan_lds, s, mask, diag_cols,
                           sorted_idx, st, removed, row0, rows, nb, max_out, d_sel_out,
                           d_count_out, last);
        DODT_LAUNCH_CHECK();
        row0 += rows;
    }
    return DODT_OK;
}
