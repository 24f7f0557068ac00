// Greedy NMS for gfx950: tf.image.non_max_suppression of
// TF 1.3 (non_max_suppression_op.cc), call sites avod/core/models/
// dt_rpn_model.py:587-591 (A anchors, k = 300/1024, thr 0.8) and
// models/dt_avod_model.py:606-613 (P proposals, k = 100, thr 0.01).
//
// Structure (64-wide wave ballots, no 32-lane idioms):
//   1. keys: 64-bit (descending score, ascending index) keys; bitonic sort, the
//      short strides inside one workgroup's LDS (8192 keys = 64 KB per tile).
//   2. boxes are gathered in sorted order with min/max-normalised corners + area.
//   3. rows are processed in chunks of 2048 candidates so that the suppression
//      matrix never exceeds 2048 x ceil(n/64) words:
//        nms_mask_kernel : one wave per (64-row block, 8 column blocks); lane =
//          column box, the 64 row boxes are broadcast from LDS and each
//          `iou > thr` test becomes one __ballot -> the row's 64-bit word.
//        nms_scan_kernel : one workgroup walks the chunk 64 rows at a time: wave 0
//          resolves the 64x64 diagonal block serially in registers (readlane +
//          find-first-set), then all 16 waves OR the selected rows into the
//          `removed` bit vector held in LDS.  It stops at max_output_size and
//          raises a flag that turns the remaining chunk launches into no-ops.
// IoU arithmetic is float32, unfused, in TF's operation order.
#include "common.h"

namespace {

constexpr int kTile = 8192;      // keys per LDS sort tile
constexpr int kSortThreads = 1024;
constexpr int kChunkRows = 2048; // candidate rows per mask chunk
constexpr int kColGroup = 2;     // column blocks per wave in the mask kernel

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
                unsigned long long* __restrict__ removed, int nb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_eff = d_n ? min(*d_n, n) : n;
    if (i == 0) { st->count = 0; st->done = (n_eff <= 0); st->n_eff = n_eff; }
    if (i < nb) removed[i] = 0ull;
    if (i >= n_pad) return;
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
                int nb, float thr, unsigned long long* __restrict__ mask) {
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
            for (int r = 0; r < 64; ++r) {
                const float* q = s_rows[wave][r];
                const bool hit = (col > rb * 64 + r) &&
                                 iou_gt(q[0], q[1], q[2], q[3], q[4], cy0, cx0, cy1, cx1, ca, thr);
                const unsigned long long w = __ballot(hit);
                if (lane == r) mine = w;
            }
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

__global__ void __launch_bounds__(1024)
nms_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ sorted_idx,
                NmsState* __restrict__ st, unsigned long long* __restrict__ g_removed, int row0,
                int rows, int nb, int max_out, int* __restrict__ sel_out,
                int* __restrict__ count_out, int last_chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_removed[];
    __shared__ unsigned long long s_sel;
    __shared__ int s_count, s_done;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_eff = st->n_eff;
    if (st->done) {
        if (last_chunk && tid == 0) *count_out = st->count;
        return;
    }
    for (int c = tid; c < nb; c += 1024) s_removed[c] = g_removed[c];
    if (tid == 0) { s_count = st->count; s_done = 0; }
    __syncthreads();
    const int b0 = row0 >> 6;
    const int b1 = min((row0 + rows + 63) >> 6, (n_eff + 63) >> 6);
    for (int b = b0; b < b1; ++b) {
        if (tid < 64) {  // wave 0 resolves the diagonal 64 x 64 block
            const int row = b * 64 + lane;
            const unsigned long long diag =
                (row < n_eff) ? mask[(size_t)(row - row0) * nb + b] : 0ull;
            const uint32_t dlo = (uint32_t)diag, dhi = (uint32_t)(diag >> 32);
            const int valid = min(64, n_eff - b * 64);
            unsigned long long alive_v = ~s_removed[b];
            if (valid < 64) alive_v &= (1ull << valid) - 1ull;
            // same value in every lane; make that explicit so the loop is scalar
            unsigned long long alive =
                ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(
                     (int)(uint32_t)(alive_v >> 32)) << 32) |
                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)alive_v);
            unsigned long long sel = 0ull;
            const int cnt0 = __builtin_amdgcn_readfirstlane(s_count);
            int cnt = cnt0;
            while (alive != 0ull && cnt < max_out) {  // wave-uniform
                const int r = __ffsll((long long)alive) - 1;
                const unsigned long long d =
                    ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dhi, r) << 32) |
                    (uint32_t)__builtin_amdgcn_readlane((int)dlo, r);
                sel |= 1ull << r;
                alive &= ~(d | (1ull << r));
                ++cnt;
            }
            if ((sel >> lane) & 1ull)
                sel_out[cnt0 + __popcll(sel & ((1ull << lane) - 1ull))] = sorted_idx[row];
            if (lane == 0) {
                s_sel = sel;
                s_count = cnt;
                if (cnt >= max_out) s_done = 1;
            }
        }
        __syncthreads();
        const unsigned long long sel = s_sel;
        const int done = s_done;
        if (!done && sel != 0ull) {
            // OR the selected rows of this block into `removed` for every later column
            // block.  All 64 rows are loaded unconditionally and masked by `sel` (no
            // branch around a load, so the loads of a thread are all in flight at once);
            // the (row group, column) pairs are spread over the whole workgroup and
            // combined with LDS atomics.
            const int ncols = nb - (b + 1);
            if (ncols > 0) {
                const int groups = max(1, min(64, 1024 / ncols));  // row groups per column
                const int c = b + 1 + tid % ncols;
                const int g = tid / ncols;
                if (g < groups) {
                    const unsigned long long* base =
                        mask + (size_t)(b * 64 - row0) * nb + c;
                    unsigned long long acc = 0ull;
                    for (int r0 = g; r0 < 64; r0 += groups * 8) {
                        unsigned long long v[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int r = min(r0 + k * groups, 63);
                            v[k] = base[(size_t)r * nb];
                        }
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int r = r0 + k * groups;
                            if (r < 64 && ((sel >> r) & 1ull)) acc |= v[k];
                        }
                    }
                    if (acc) atomicOr(&s_removed[c], acc);
                }
                // columns beyond one workgroup pass (ncols > 1024)
                for (int cc = b + 1 + 1024 + tid; cc < nb; cc += 1024) {
                    unsigned long long acc = 0ull, rest = sel;
                    while (rest != 0ull) {
                        const int r = __ffsll((long long)rest) - 1;
                        rest &= rest - 1ull;
                        acc |= mask[(size_t)(b * 64 + r - row0) * nb + cc];
                    }
                    s_removed[cc] |= acc;
                }
            }
        }
        __syncthreads();
        if (done) break;
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
    const int chunk_rows = n_rows < kChunkRows ? n_rows : kChunkRows;
    // scratch layout
    size_t off = 0;
    const size_t o_state = off; off += 256;
    const size_t o_keys = off; off += dodt::align_up((size_t)n_pad * 8, 256);
    const size_t o_sb = off; off += dodt::align_up((size_t)n_rows * 5 * 4, 256);
    const size_t o_idx = off; off += dodt::align_up((size_t)n_rows * 4, 256);
    const size_t o_removed = off; off += dodt::align_up((size_t)nb * 8, 256);
    const size_t o_mask = off; off += (size_t)chunk_rows * nb * 8;
    int rc = ctx->nms_ws.reserve(off);
    if (rc) return rc;
    char* ws = reinterpret_cast<char*>(ctx->nms_ws.ptr);
    NmsState* st = reinterpret_cast<NmsState*>(ws + o_state);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws + o_keys);
    float* sb = reinterpret_cast<float*>(ws + o_sb);
    int* sorted_idx = reinterpret_cast<int*>(ws + o_idx);
    unsigned long long* removed = reinterpret_cast<unsigned long long*>(ws + o_removed);
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(ws + o_mask);
    hipStream_t s = ctx->stream;

    const int init_n = n_pad > nb ? n_pad : nb;
    hipLaunchKernelGGL(nms_keys_kernel, dim3(dodt::ceil_div(init_n, 256)), dim3(256), 0, s,
                       d_scores, n, d_n, n_pad, keys, st, removed, nb);
    DODT_LAUNCH_CHECK();
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
    const int col_groups = dodt::ceil_div(nb, kColGroup);
    for (int row0 = 0; row0 < n_rows; row0 += chunk_rows) {
        const int rows = (n_rows - row0) < chunk_rows ? (n_rows - row0) : chunk_rows;
        hipLaunchKernelGGL(nms_mask_kernel, dim3(col_groups, dodt::ceil_div(rows / 64, 4)),
                           dim3(256), 0, s, sb, st, row0, rows, nb, iou_threshold, mask);
        DODT_LAUNCH_CHECK();
        const int last = (row0 + rows >= n_rows);
        hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(1024), (size_t)nb * 8, s, mask,
                           sorted_idx, st, removed, row0, rows, nb, max_out, d_sel_out,
                           d_count_out, last);
        DODT_LAUNCH_CHECK();
    }
    return DODT_OK;
}
