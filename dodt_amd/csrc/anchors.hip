// Empty-anchor filter, anchor projection and the box encoders for gfx950
// (SURVEY.md 8a rows a4, a5, a6, a12, a14).  All of it is tiny, latency-bound
// elementwise work: one lane per anchor / proposal, coalesced row loads.
//
// Reference behaviour (see oracle/anchors.py, oracle/boxes.py):
//   avod/core/anchor_filter.py:64-119, wavedata/.../integral_image_2d.py:39-87
//   avod/core/anchor_projector.py:13-306, anchor_encoder.py:99-150,
//   box_3d_encoder.py:188-322, box_4c_encoder.py:85-165,305-484
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// a4.  The reference builds a summed-area table over the 800x700 occupancy and
// asks it for the number of occupied cells in [x1,x2) x [z1,z2).  The boxes are
// at most ~43 x 43 cells, so the same count comes from popcounts over the
// voxeliser's bit grid (70 KB, L2 resident) with no table to build.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int box_count(const uint32_t* __restrict__ occ, int wpr, int x1,
                                         int z1, int x2, int z2, int thr) {
    int cnt = 0;
    if (x2 <= x1) return 0;
    const int w0 = x1 >> 5, w1 = (x2 - 1) >> 5;
    for (int z = z1; z < z2; ++z) {
        for (int w = w0; w <= w1; ++w) {
            uint32_t word = occ[z * wpr + w];
            const int lo = (w == w0) ? (x1 & 31) : 0;
            const int hi = (w == w1) ? ((x2 - 1) & 31) : 31;
            uint32_t m = (hi == 31 ? 0xFFFFFFFFu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
            cnt += __popc(word & m);
        }
        if (cnt >= thr) return cnt;
    }
    return cnt;
}

// (64-lane workgroups since round 4: a lane walks up to 43 rows x 3 words of L2-resident bits one after the other, and 350
//  workgroups of 256 gave a CU one or two of those latency chains to overlap; block_counts is per 64 anchors)
constexpr int kMaskBlock = 64;
__global__ void __launch_bounds__(kMaskBlock)
anchor_mask_kernel(const uint32_t* __restrict__ occ, int wpr, int nx, int nz,
                   const int4* __restrict__ cells, int n, int thr,
                   uint8_t* __restrict__ mask, int* __restrict__ block_counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int keep = 0;
    if (i < n) {
        int4 b = cells[i];
        // IntegralImage2D.query clamps to the table size again (:71-76)
        const int x1 = min(max(b.x, 0), nx), z1 = min(max(b.y, 0), nz);
        const int x2 = min(max(b.z, 0), nx), z2 = min(max(b.w, 0), nz);
        keep = box_count(occ, wpr, x1, z1, x2, z2, thr) >= thr;
        mask[i] = (uint8_t)keep;
    }
    const int total = __popcll(__ballot(keep));      // one wave per workgroup
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256)
anchor_compact_kernel(const uint8_t* __restrict__ mask, const int* __restrict__ block_counts,
                      int n, int* __restrict__ keep_idx, int* __restrict__ count_out) {
    __shared__ int s_part[256];
    __shared__ int s_wave[4];
    const int tid = threadIdx.x;
    // offset of this block = sum of the counts of the (64-anchor) mask blocks before it
    int part = 0;
    for (int j = tid; j < (int)blockIdx.x * (256 / kMaskBlock); j += 256) part += block_counts[j];
    s_part[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) s_part[tid] += s_part[tid + s];
        __syncthreads();
    }
    const int block_off = s_part[0];
    const int i = blockIdx.x * 256 + tid;
    const int keep = (i < n) ? mask[i] : 0;
    const unsigned long long bal = __ballot(keep);
    const int lane = tid & 63, wave = tid >> 6;
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
    if (keep) keep_idx[block_off + wave_off + before] = i;
    if (blockIdx.x == gridDim.x - 1 && tid == 0)
        *count_out = block_off + s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

// ---------------------------------------------------------------------------
// a5 / a6, numpy branch: float64 in, float32 out, TF box order.
// ---------------------------------------------------------------------------
struct ProjParams64 {
    double x_min, x_max, z_min, z_max;
    double p[12];
    double im_w, im_h;
};

__device__ __forceinline__ void proj_corner64(const double* p, double x, double y, double z,
                                              double& u, double& v) {
    const double un = fma(p[2], z, fma(p[1], y, p[0] * x)) + p[3];
    const double vn = fma(p[6], z, fma(p[5], y, p[4] * x)) + p[7];
    const double w = fma(p[10], z, fma(p[9], y, p[8] * x)) + p[11];
    u = un / w;
    v = vn / w;
}

__global__ void __launch_bounds__(256)
project_f64_kernel(const double* __restrict__ anchors, const int* __restrict__ idx, int n,
                   const int* __restrict__ d_n, const ProjParams64 P,
                   float* __restrict__ bev_norm, float* __restrict__ img_norm,
                   float* __restrict__ anchors_f32) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lim = d_n ? min(*d_n, n) : n;
    if (i >= lim) return;
    const int row = idx ? idx[i] : i;
    const double* a = anchors + (size_t)row * 6;
    const double x = a[0], y = a[1], z = a[2], dx = a[3], dy = a[4], dz = a[5];
    if (anchors_f32) {
        float* o = anchors_f32 + (size_t)i * 6;
        o[0] = (float)x; o[1] = (float)y; o[2] = (float)z;
        o[3] = (float)dx; o[4] = (float)dy; o[5] = (float)dz;
    }
    const double hx = dx / 2.0, hz = dz / 2.0;
    if (bev_norm) {
        const double xr = P.x_max - P.x_min, zr = P.z_max - P.z_min;
        const double x1 = (x - hx) - P.x_min, x2 = (x + hx) - P.x_min;
        const double z1 = (P.z_max - (z + hz)) - P.z_min, z2 = (P.z_max - (z - hz)) - P.z_min;
        float* o = bev_norm + (size_t)i * 4;  // [y1,x1,y2,x2] = [z1,x1,z2,x2]
        o[0] = (float)(z1 / zr); o[1] = (float)(x1 / xr);
        o[2] = (float)(z2 / zr); o[3] = (float)(x2 / xr);
    }
    if (img_norm) {
        double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300;
        // corner order of anchor_projector.py:104-129 (irrelevant for min/max)
        const double xs[2] = {x + hx, x - hx};
        const double ys[2] = {y, y - dy};
        const double zs[2] = {z + hz, z - hz};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double u, v;
            proj_corner64(P.p, xs[(k >> 1) & 1], ys[k >> 2], zs[((k + 1) >> 1) & 1], u, v);
            // np.amin / np.amax propagate NaN (w == 0 with a zero numerator)
            umin = (u < umin || u != u) ? u : umin; umax = (u > umax || u != u) ? u : umax;
            vmin = (v < vmin || v != v) ? v : vmin; vmax = (v > vmax || v != v) ? v : vmax;
        }
        float* o = img_norm + (size_t)i * 4;  // [y1,x1,y2,x2] = [v1,u1,v2,u2]
        o[0] = (float)(vmin / P.im_h); o[1] = (float)(umin / P.im_w);
        o[2] = (float)(vmax / P.im_h); o[3] = (float)(umax / P.im_w);
    }
}

// ---------------------------------------------------------------------------
// TF branch, float32 (no contraction: the library is built -ffp-contract=off)
// ---------------------------------------------------------------------------
struct ProjParams32 {
    float x_min, x_max, z_min, z_max;
    float p[12];
    float im_w, im_h;
};

__device__ __forceinline__ void bev_box32(float x, float z, float dx, float dz,
                                          const ProjParams32& P, float& x1, float& z1,
                                          float& x2, float& z2) {
    const float hx = dx / 2.0f, hz = dz / 2.0f;
    x1 = (x - hx) - P.x_min;
    x2 = (x + hx) - P.x_min;
    z1 = (P.z_max - (z + hz)) - P.z_min;
    z2 = (P.z_max - (z - hz)) - P.z_min;
}

// one anchor (x, y, z, dx, dy, dz) -> its BEV box in metres / normalised for crop_and_resize / its image box (row i of the outputs)
__device__ __forceinline__ void project_f32_elem(int i, float x, float y, float z, float dx, float dy, float dz,
                                                 const ProjParams32& P, float* __restrict__ bev,
                                                 float* __restrict__ bev_norm_tf, float* __restrict__ img_norm_tf) {
    float x1, z1, x2, z2;
    bev_box32(x, z, dx, dz, P, x1, z1, x2, z2);
    if (bev) {
        float* o = bev + (size_t)i * 4;
        o[0] = x1; o[1] = z1; o[2] = x2; o[3] = z2;
    }
    if (bev_norm_tf) {
        const float xr = P.x_max - P.x_min, zr = P.z_max - P.z_min;
        float* o = bev_norm_tf + (size_t)i * 4;
        o[0] = z1 / zr; o[1] = x1 / xr; o[2] = z2 / zr; o[3] = x2 / xr;
    }
    if (img_norm_tf) {
        const float hx = dx / 2.0f, hz = dz / 2.0f;
        const float xs[2] = {x + hx, x - hx};
        const float ys[2] = {y, y - dy};
        const float zs[2] = {z + hz, z - hz};
        float umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float cx = xs[(k >> 1) & 1], cy = ys[k >> 2], cz = zs[((k + 1) >> 1) & 1];
            // tf.matmul row . [x,y,z,1], terms in index order
            const float un = ((P.p[0] * cx + P.p[1] * cy) + P.p[2] * cz) + P.p[3];
            const float vn = ((P.p[4] * cx + P.p[5] * cy) + P.p[6] * cz) + P.p[7];
            const float w = ((P.p[8] * cx + P.p[9] * cy) + P.p[10] * cz) + P.p[11];
            const float u = un / w, v = vn / w;
            umin = fminf(umin, u); umax = fmaxf(umax, u);
            vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
        }
        float* o = img_norm_tf + (size_t)i * 4;
        o[0] = vmin / P.im_h; o[1] = umin / P.im_w; o[2] = vmax / P.im_h; o[3] = umax / P.im_w;
    }
}

__global__ void __launch_bounds__(256)
project_f32_kernel(const float* __restrict__ anchors, int n, const int* __restrict__ d_n,
                   const ProjParams32 P, float* __restrict__ bev, float* __restrict__ bev_norm_tf,
                   float* __restrict__ img_norm_tf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lim = d_n ? min(*d_n, n) : n;
    if (i >= lim) return;
    const float* a = anchors + (size_t)i * 6;
    project_f32_elem(i, a[0], a[1], a[2], a[3], a[4], a[5], P, bev, bev_norm_tf, img_norm_tf);
}

__device__ __forceinline__ void offset_to_anchor_elem(const float* __restrict__ a, const float* __restrict__ t, float o[6]) {
    o[0] = (t[0] * a[3]) + a[0];
    o[1] = (t[1] * a[4]) + a[1];
    o[2] = (t[2] * a[5]) + a[2];
    o[3] = expf(logf(a[3]) + t[3]);
    o[4] = expf(logf(a[4]) + t[4]);
    o[5] = expf(logf(a[5]) + t[5]);
}

__global__ void __launch_bounds__(256)
offset_to_anchor_kernel(const float* __restrict__ anchors, const float* __restrict__ off, int n,
                        const int* __restrict__ d_n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lim = d_n ? min(*d_n, n) : n;
    if (i >= lim) return;
    float o[6];
    offset_to_anchor_elem(anchors + (size_t)i * 6, off + (size_t)i * 6, o);
#pragma unroll
    for (int k = 0; k < 6; ++k) out[(size_t)i * 6 + k] = o[k];
}

__device__ __forceinline__ float softmax_fg_elem(float l0, float l1) {
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    return e1 / (e0 + e1);
}

__global__ void __launch_bounds__(256)
softmax_fg_kernel(const float* __restrict__ logits, int n, const int* __restrict__ d_n,
                  float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lim = d_n ? min(*d_n, n) : n;
    if (i >= lim) return;
    out[i] = softmax_fg_elem(logits[2 * (size_t)i], logits[2 * (size_t)i + 1]);
}

// Round 4: the RPN's three elementwise launches in one (offset_to_anchor -> project_to_bev of the regressed anchor ->
// softmax): the same device functions on the same float values, one launch on a frame's dependent launch chain instead of three.
__global__ void __launch_bounds__(256)
rpn_decode_kernel(const float* __restrict__ anchors, const float* __restrict__ off, const float* __restrict__ logits,
                  int n, const int* __restrict__ d_n, const ProjParams32 P, float* __restrict__ regressed,
                  float* __restrict__ bev_norm_tf, float* __restrict__ scores) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lim = d_n ? min(*d_n, n) : n;
    if (i >= lim) return;
    float o[6];
    offset_to_anchor_elem(anchors + (size_t)i * 6, off + (size_t)i * 6, o);
#pragma unroll
    for (int k = 0; k < 6; ++k) regressed[(size_t)i * 6 + k] = o[k];
    project_f32_elem(i, o[0], o[1], o[2], o[3], o[4], o[5], P, nullptr, bev_norm_tf, nullptr);
    scores[i] = softmax_fg_elem(logits[2 * (size_t)i], logits[2 * (size_t)i + 1]);
}

// ... and the proposals' two: gather the kept rows, project them to both views
__global__ void __launch_bounds__(256)
gather_project_kernel(const float* __restrict__ src, const int* __restrict__ idx, int n, const int* __restrict__ d_n,
                      const ProjParams32 P, float* __restrict__ rows, float* __restrict__ bev_norm_tf,
                      float* __restrict__ img_norm_tf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lim = d_n ? min(*d_n, n) : n;
    if (i >= lim) return;
    const float* a = src + (size_t)idx[i] * 6;
    float o[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { o[k] = a[k]; rows[(size_t)i * 6 + k] = o[k]; }
    project_f32_elem(i, o[0], o[1], o[2], o[3], o[4], o[5], P, nullptr, bev_norm_tf, img_norm_tf);
}

__global__ void __launch_bounds__(256)
gather_rows_kernel(const float* __restrict__ src, int width, const int* __restrict__ idx, int n,
                   const int* __restrict__ d_n, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lim = d_n ? min(*d_n, n) : n;
    const int i = t / width, c = t - i * width;
    if (i >= lim) return;
    out[(size_t)i * width + c] = src[(size_t)idx[i] * width + c];
}

__global__ void __launch_bounds__(256)
max_fg_logit_kernel(const float* __restrict__ logits, int n_cls, int n,
                    const int* __restrict__ d_n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lim = d_n ? min(*d_n, n) : n;
    if (i >= lim) return;
    float m = logits[(size_t)i * n_cls + 1];
    for (int c = 2; c < n_cls; ++c) m = fmaxf(m, logits[(size_t)i * n_cls + c]);
    out[i] = m;
}

// dt_evaluator.py:1166-1212 (box_rep 'box_4ca'): the box decoded from four corners knows its
// heading modulo 90 / 180 degrees only; the regressed angle decides.  float32, unfused, the
// thresholds as float32(k * pi) like numpy's scalar-with-array comparisons.
__device__ __forceinline__ void orientation_correct(float b[7], float ori) {
    const float kPi = (float)M_PI, kTwoPi = (float)(2.0 * M_PI);
    const float q1 = (float)(0.25 * M_PI), q2 = (float)(0.50 * M_PI), q3 = (float)(0.75 * M_PI);
    float diff = b[6] - ori;
    if (diff < -kPi) diff += kTwoPi;
    if (diff > kPi) diff -= kTwoPi;
    const bool pos = q1 < diff && diff < q3;
    const bool neg = -q1 > diff && diff > -q3;
    if (pos || neg) {
        const float l = b[3];
        b[3] = b[4];
        b[4] = l;
    }
    if (pos) b[6] += q2;
    if (neg) b[6] -= q2;
    if (fabsf(diff) >= q3) b[6] += kPi;
    if (b[6] > kPi) b[6] -= kTwoPi;
}

__global__ void __launch_bounds__(256)
pack_detections_kernel(const float* __restrict__ boxes_3d, const float* __restrict__ scores,
                       const float* __restrict__ orientations, const float* __restrict__ corr,
                       const int* __restrict__ sel, const int* __restrict__ d_count, int max_det,
                       float frame_mark, float* __restrict__ rec, int* __restrict__ count_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int cnt = min(*d_count, max_det);
    if (t == 0) *count_out = cnt;
    if (t >= max_det * 17) return;
    const int row = t / 17, col = t - row * 17;
    float v = 0.0f;
    if (row < cnt) {
        const int src = sel[row];
        if (col == 7) v = scores[src];
        else if (col == 16) v = frame_mark;
        else if (col < 7 || (col >= 9 && corr)) {
            float b[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) b[k] = boxes_3d[(size_t)src * 7 + k];
            if (orientations) orientation_correct(b, orientations[src]);
            if (col >= 9) {
                // dt_evaluator.py:1217-1224: the box shifted by (dx, dz, dry) into the next frame
                b[0] += corr[(size_t)src * 3 + 0];
                b[2] += corr[(size_t)src * 3 + 1];
                b[6] += corr[(size_t)src * 3 + 2];
            }
            const int c = col < 7 ? col : col - 9;
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (k == c) v = b[k];
        }
    }
    rec[t] = v;
}

// orientation_encoder.py:20-34: atan2(y, x) of the regressed angle vector [x, y]
__global__ void __launch_bounds__(256)
angle_vector_to_orientation_kernel(const float* __restrict__ vec, int n,
                                   const int* __restrict__ d_n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (d_n) n = min(n, *d_n);
    if (i < n) out[i] = atan2f(vec[2 * i + 1], vec[2 * i]);
}

// calculate_box_3d_info (box_4c_encoder.py:305-366) for one candidate midline
__device__ __forceinline__ void box_info32(float vx, float vz, float mag, const float* px,
                                           const float* pz, float mx, float mz, float& cx,
                                           float& cz, float& len, float& wid, float& ry) {
    const float nx = vx / mag, nz = vz / mag;
    const float ox = -nz, oz = nx;
    float lmin = 3.0e38f, lmax = -3.0e38f, wmin = 3.0e38f, wmax = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float rx = px[k] - mx, rz = pz[k] - mz;
        const float l = rx * nx + rz * nz;
        const float w = rx * ox + rz * oz;
        lmin = fminf(lmin, l); lmax = fmaxf(lmax, l);
        wmin = fminf(wmin, w); wmax = fmaxf(wmax, w);
    }
    len = lmax - lmin;
    wid = wmax - wmin;
    const float wdiff = wmax + wmin;
    ry = -atan2f(vz, vx);
    const float half = (lmin + lmax);
    cx = (mx + (nx * half) / 2.0f) + ox * wdiff;
    cz = (mz + (nz * half) / 2.0f) + oz * wdiff;
}

struct DecodeParams {
    float a, b, c, d;  // ground plane
    float x_min, x_max, z_min, z_max;
};

__device__ __forceinline__ void box_4c_decode_elem(int i, const float* __restrict__ top_anchors,
                                                   const float* __restrict__ offsets, const DecodeParams& P,
                                                   float* __restrict__ boxes_3d, float* __restrict__ pred_anchors,
                                                   float* __restrict__ bev_tf) {
    const float* a = top_anchors + (size_t)i * 6;
    const float half_pi = (float)(M_PI / 2);
    // anchors_to_box_3d(fix_lw=True), tensor branch (box_3d_encoder.py:249-290)
    float bx = a[0], by = a[1], bz = a[2];
    float bl = a[3], bw = a[5], bh = a[4], bry = 0.0f;
    if (bw > bl) { const float t = bl; bl = bw; bw = t; bry = -half_pi; }
    // tf_box_3d_to_anchor (:188-227)
    float ortho = rintf(bry / half_pi) * half_pi;
    float co = fabsf(cosf(ortho)), si = fabsf(sinf(ortho));
    const float dimx = bl * co + bw * si, dimy = bh, dimz = bw * co + bl * si;
    // tf_box_3d_to_box_4c (box_4c_encoder.py:85-165)
    const float hx = dimx / 2, hz = dimz / 2;
    const float diff = bry - ortho;
    const float cd = cosf(diff), sd = sinf(diff);
    const float xs[4] = {hx, hx, -hx, -hx};
    const float zs[4] = {hz, -hz, -hz, hz};
    float px[4], pz[4];
    const float* t = offsets + (size_t)i * 10;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        px[k] = ((cd * xs[k] + sd * zs[k]) + bx) + t[k];
        pz[k] = (((-sd) * xs[k] + cd * zs[k]) + bz) + t[4 + k];
    }
    const float gy = -((P.a * bx + P.c * bz) + P.d) / P.b;
    const float h1 = (gy - by) + t[8];
    const float h2 = ((gy - by) + dimy) + t[9];
    // tf_box_4c_to_box_3d (:369-458)
    const float m12x = (px[0] + px[1]) / 2.0f, m12z = (pz[0] + pz[1]) / 2.0f;
    const float m23x = (px[1] + px[2]) / 2.0f, m23z = (pz[1] + pz[2]) / 2.0f;
    const float m34x = (px[2] + px[3]) / 2.0f, m34z = (pz[2] + pz[3]) / 2.0f;
    const float m14x = (px[0] + px[3]) / 2.0f, m14z = (pz[0] + pz[3]) / 2.0f;
    const float vax = m12x - m34x, vaz = m12z - m34z;
    const float vbx = m14x - m23x, vbz = m14z - m23z;
    const float ma = sqrtf(vax * vax + vaz * vaz), mb = sqrtf(vbx * vbx + vbz * vbz);
    float cax, caz, la, wa, ra, cbx, cbz, lb, wb, rb;
    box_info32(vax, vaz, ma, px, pz, m34x, m34z, cax, caz, la, wa, ra);
    box_info32(vbx, vbz, mb, px, pz, m23x, m23z, cbx, cbz, lb, wb, rb);
    const float fa = (ma > mb) ? 1.0f : 0.0f, fb = 1.0f - fa;
    const float cx = cax * fa + cbx * fb, cz = caz * fa + cbz * fb;
    const float len = la * fa + lb * fb, wid = wa * fa + wb * fb, ry = ra * fa + rb * fb;
    const float gy2 = -((P.a * cx + P.c * cz) + P.d) / P.b;
    const float cy = gy2 - h1, hgt = h2 - h1;
    if (boxes_3d) {
        float* o = boxes_3d + (size_t)i * 7;
        o[0] = cx; o[1] = cy; o[2] = cz; o[3] = len; o[4] = wid; o[5] = hgt; o[6] = ry;
    }
    // tf_box_3d_to_anchor on the prediction, then project_to_bev (metres) + reorder
    ortho = rintf(ry / half_pi) * half_pi;
    co = fabsf(cosf(ortho));
    si = fabsf(sinf(ortho));
    const float adx = len * co + wid * si, adz = wid * co + len * si;
    if (pred_anchors) {
        float* o = pred_anchors + (size_t)i * 6;
        o[0] = cx; o[1] = cy; o[2] = cz; o[3] = adx; o[4] = hgt; o[5] = adz;
    }
    if (bev_tf) {
        const float ahx = adx / 2.0f, ahz = adz / 2.0f;
        const float x1 = (cx - ahx) - P.x_min, x2 = (cx + ahx) - P.x_min;
        const float z1 = (P.z_max - (cz + ahz)) - P.z_min, z2 = (P.z_max - (cz - ahz)) - P.z_min;
        float* o = bev_tf + (size_t)i * 4;
        o[0] = z1; o[1] = x1; o[2] = z2; o[3] = x2;
    }
}

__global__ void __launch_bounds__(256)
box_4c_decode_kernel(const float* __restrict__ top_anchors, const float* __restrict__ offsets,
                     int n, const int* __restrict__ d_n, const DecodeParams P,
                     float* __restrict__ boxes_3d, float* __restrict__ pred_anchors,
                     float* __restrict__ bev_tf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lim = d_n ? min(*d_n, n) : n;
    if (i >= lim) return;
    box_4c_decode_elem(i, top_anchors, offsets, P, boxes_3d, pred_anchors, bev_tf);
}

// Round 4: what follows the stage-2 head per proposal in one launch: box_4c decode, the NMS score (largest foreground
// logit), the record score (softmax) and the orientation of the angle vector -- four launches of a frame's chain before
__global__ void __launch_bounds__(256)
final_decode_kernel(const float* __restrict__ top_anchors, const float* __restrict__ offsets,
                    const float* __restrict__ cls_logits, const float* __restrict__ angle_vec, int n,
                    const int* __restrict__ d_n, const DecodeParams P, float* __restrict__ boxes_3d,
                    float* __restrict__ pred_anchors, float* __restrict__ bev_tf, float* __restrict__ nms_scores,
                    float* __restrict__ det_scores, float* __restrict__ orientations) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lim = d_n ? min(*d_n, n) : n;
    if (i >= lim) return;
    box_4c_decode_elem(i, top_anchors, offsets, P, boxes_3d, pred_anchors, bev_tf);
    const float l0 = cls_logits[2 * (size_t)i], l1 = cls_logits[2 * (size_t)i + 1];
    nms_scores[i] = l1;                                  // max over the one foreground class
    det_scores[i] = softmax_fg_elem(l0, l1);
    if (orientations) orientations[i] = atan2f(angle_vec[2 * i + 1], angle_vec[2 * i]);
}

}  // namespace

extern "C" {

int dodt_anchor_filter(dodt_ctx* ctx, const uint32_t* d_occ_bits, int nx, int nz,
                       const int32_t* d_anchor_cells, int n_anchors, int density_threshold,
                       int32_t* d_keep_idx_out, int32_t* d_count_out) {
    DODT_REQUIRE(ctx && d_occ_bits && d_anchor_cells && d_keep_idx_out && d_count_out,
                 "dodt_anchor_filter: NULL argument");
    DODT_REQUIRE(nx > 0 && nz > 0 && n_anchors >= 0, "dodt_anchor_filter: bad sizes");
    if (n_anchors == 0) {
        DODT_HIP_CHECK(hipMemsetAsync(d_count_out, 0, sizeof(int32_t), ctx->stream));
        return DODT_OK;
    }
    const int blocks = dodt::ceil_div(n_anchors, 256), mask_blocks = dodt::ceil_div(n_anchors, kMaskBlock);
    const size_t mask_bytes = dodt::align_up((size_t)n_anchors, 256);
    int rc = ctx->anchor_ws.reserve(mask_bytes + (size_t)mask_blocks * sizeof(int));
    if (rc) return rc;
    uint8_t* mask = reinterpret_cast<uint8_t*>(ctx->anchor_ws.ptr);
    int* block_counts = reinterpret_cast<int*>(mask + mask_bytes);
    hipLaunchKernelGGL(anchor_mask_kernel, dim3(mask_blocks), dim3(kMaskBlock), 0, ctx->stream, d_occ_bits,
                       dodt::ceil_div(nx, 32), nx, nz,
                       reinterpret_cast<const int4*>(d_anchor_cells), n_anchors,
                       density_threshold, mask, block_counts);
    DODT_LAUNCH_CHECK();
    hipLaunchKernelGGL(anchor_compact_kernel, dim3(blocks), dim3(256), 0, ctx->stream, mask,
                       block_counts, n_anchors, d_keep_idx_out, d_count_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

int dodt_project_anchors_f64(dodt_ctx* ctx, const double* d_anchors, const int32_t* d_idx, int n,
                             const int32_t* d_n, const double bev_extents[4],
                             const double p2[12], double im_w, double im_h,
                             float* d_bev_norm_out, float* d_img_norm_out,
                             float* d_anchors_f32_out) {
    DODT_REQUIRE(ctx && d_anchors && bev_extents && p2, "dodt_project_anchors_f64: NULL argument");
    if (n <= 0) return DODT_OK;
    ProjParams64 P;
    P.x_min = bev_extents[0]; P.x_max = bev_extents[1];
    P.z_min = bev_extents[2]; P.z_max = bev_extents[3];
    for (int k = 0; k < 12; ++k) P.p[k] = p2[k];
    P.im_w = im_w; P.im_h = im_h;
    hipLaunchKernelGGL(project_f64_kernel, dim3(dodt::ceil_div(n, 256)), dim3(256), 0,
                       ctx->stream, d_anchors, d_idx, n, d_n, P, d_bev_norm_out, d_img_norm_out,
                       d_anchors_f32_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

int dodt_project_anchors_f32(dodt_ctx* ctx, const float* d_anchors, int n, const int32_t* d_n,
                             const float bev_extents[4], const float p2[12], float im_w,
                             float im_h, float* d_bev_out, float* d_bev_norm_tf_out,
                             float* d_img_norm_tf_out) {
    DODT_REQUIRE(ctx && d_anchors && bev_extents && p2, "dodt_project_anchors_f32: NULL argument");
    if (n <= 0) return DODT_OK;
    ProjParams32 P;
    P.x_min = bev_extents[0]; P.x_max = bev_extents[1];
    P.z_min = bev_extents[2]; P.z_max = bev_extents[3];
    for (int k = 0; k < 12; ++k) P.p[k] = p2[k];
    P.im_w = im_w; P.im_h = im_h;
    hipLaunchKernelGGL(project_f32_kernel, dim3(dodt::ceil_div(n, 256)), dim3(256), 0,
                       ctx->stream, d_anchors, n, d_n, P, d_bev_out, d_bev_norm_tf_out,
                       d_img_norm_tf_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

int dodt_rpn_decode(dodt_ctx* ctx, const float* d_anchors, const float* d_offsets, const float* d_logits2, int n,
                    const int32_t* d_n, const float bev_extents[4], float* d_regressed_out,
                    float* d_bev_norm_tf_out, float* d_scores_out) {
    DODT_REQUIRE(ctx && d_anchors && d_offsets && d_logits2 && bev_extents && d_regressed_out && d_bev_norm_tf_out &&
                     d_scores_out, "dodt_rpn_decode: NULL argument");
    if (n <= 0) return DODT_OK;
    ProjParams32 P = {};
    P.x_min = bev_extents[0]; P.x_max = bev_extents[1];
    P.z_min = bev_extents[2]; P.z_max = bev_extents[3];
    hipLaunchKernelGGL(rpn_decode_kernel, dim3(dodt::ceil_div(n, 256)), dim3(256), 0, ctx->stream, d_anchors,
                       d_offsets, d_logits2, n, d_n, P, d_regressed_out, d_bev_norm_tf_out, d_scores_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

int dodt_gather_project(dodt_ctx* ctx, const float* d_src, const int32_t* d_idx, int n, const int32_t* d_n,
                        const float bev_extents[4], const float p2[12], float im_w, float im_h,
                        float* d_rows_out, float* d_bev_norm_tf_out, float* d_img_norm_tf_out) {
    DODT_REQUIRE(ctx && d_src && d_idx && bev_extents && p2 && d_rows_out && d_bev_norm_tf_out && d_img_norm_tf_out,
                 "dodt_gather_project: NULL argument");
    if (n <= 0) return DODT_OK;
    ProjParams32 P;
    P.x_min = bev_extents[0]; P.x_max = bev_extents[1];
    P.z_min = bev_extents[2]; P.z_max = bev_extents[3];
    for (int k = 0; k < 12; ++k) P.p[k] = p2[k];
    P.im_w = im_w; P.im_h = im_h;
    hipLaunchKernelGGL(gather_project_kernel, dim3(dodt::ceil_div(n, 256)), dim3(256), 0, ctx->stream, d_src, d_idx,
                       n, d_n, P, d_rows_out, d_bev_norm_tf_out, d_img_norm_tf_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

int dodt_final_decode(dodt_ctx* ctx, const float* d_top_anchors, const float* d_offsets, const float* d_cls_logits2,
                      const float* d_angle_vectors, int n, const int32_t* d_n, const float plane[4],
                      const float bev_extents[4], float* d_boxes_3d_out, float* d_pred_anchors_out,
                      float* d_bev_tf_out, float* d_nms_scores_out, float* d_det_scores_out,
                      float* d_orientations_out) {
    DODT_REQUIRE(ctx && d_top_anchors && d_offsets && d_cls_logits2 && plane && bev_extents && d_nms_scores_out &&
                     d_det_scores_out && (d_angle_vectors != nullptr) == (d_orientations_out != nullptr),
                 "dodt_final_decode: bad argument");
    if (n <= 0) return DODT_OK;
    DecodeParams P;
    P.a = plane[0]; P.b = plane[1]; P.c = plane[2]; P.d = plane[3];
    P.x_min = bev_extents[0]; P.x_max = bev_extents[1];
    P.z_min = bev_extents[2]; P.z_max = bev_extents[3];
    hipLaunchKernelGGL(final_decode_kernel, dim3(dodt::ceil_div(n, 256)), dim3(256), 0, ctx->stream, d_top_anchors,
                       d_offsets, d_cls_logits2, d_angle_vectors, n, d_n, P, d_boxes_3d_out, d_pred_anchors_out,
                       d_bev_tf_out, d_nms_scores_out, d_det_scores_out, d_orientations_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

int dodt_offset_to_anchor(dodt_ctx* ctx, const float* d_anchors, const float* d_offsets, int n,
                          const int32_t* d_n, float* d_out) {
    DODT_REQUIRE(ctx && d_anchors && d_offsets && d_out, "dodt_offset_to_anchor: NULL argument");
    if (n <= 0) return DODT_OK;
    hipLaunchKernelGGL(offset_to_anchor_kernel, dim3(dodt::ceil_div(n, 256)), dim3(256), 0,
                       ctx->stream, d_anchors, d_offsets, n, d_n, d_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

int dodt_softmax_fg(dodt_ctx* ctx, const float* d_logits2, int n, const int32_t* d_n,
                    float* d_scores_out) {
    DODT_REQUIRE(ctx && d_logits2 && d_scores_out, "dodt_softmax_fg: NULL argument");
    if (n <= 0) return DODT_OK;
    hipLaunchKernelGGL(softmax_fg_kernel, dim3(dodt::ceil_div(n, 256)), dim3(256), 0,
                       ctx->stream, d_logits2, n, d_n, d_scores_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

int dodt_gather_rows(dodt_ctx* ctx, const float* d_src, int width, const int32_t* d_idx, int n,
                     const int32_t* d_n, float* d_out) {
    DODT_REQUIRE(ctx && d_src && d_idx && d_out && width > 0, "dodt_gather_rows: bad argument");
    if (n <= 0) return DODT_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(dodt::ceil_div(n * width, 256)), dim3(256), 0,
                       ctx->stream, d_src, width, d_idx, n, d_n, d_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

int dodt_max_fg_logit(dodt_ctx* ctx, const float* d_logits, int n_cls, int n, const int32_t* d_n,
                      float* d_scores_out) {
    DODT_REQUIRE(ctx && d_logits && d_scores_out && n_cls >= 2, "dodt_max_fg_logit: bad argument");
    if (n <= 0) return DODT_OK;
    hipLaunchKernelGGL(max_fg_logit_kernel, dim3(dodt::ceil_div(n, 256)), dim3(256), 0,
                       ctx->stream, d_logits, n_cls, n, d_n, d_scores_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

int dodt_pack_detections(dodt_ctx* ctx, const float* d_boxes_3d, const float* d_scores,
                         const float* d_orientations, const float* d_corr_offsets,
                         const int32_t* d_sel, const int32_t* d_count, int max_det,
                         float frame_mark, float* d_rec_out, int32_t* d_count_out) {
    DODT_REQUIRE(ctx && d_boxes_3d && d_scores && d_sel && d_count && d_rec_out && d_count_out &&
                     max_det > 0,
                 "dodt_pack_detections: bad argument");
    hipLaunchKernelGGL(pack_detections_kernel, dim3(dodt::ceil_div(max_det * 17, 256)), dim3(256),
                       0, ctx->stream, d_boxes_3d, d_scores, d_orientations, d_corr_offsets, d_sel,
                       d_count, max_det, frame_mark, d_rec_out, d_count_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

int dodt_angle_vector_to_orientation(dodt_ctx* ctx, const float* d_angle_vectors, int n,
                                     const int32_t* d_n, float* d_orientations_out) {
    DODT_REQUIRE(ctx && d_angle_vectors && d_orientations_out,
                 "dodt_angle_vector_to_orientation: NULL argument");
    if (n <= 0) return DODT_OK;
    hipLaunchKernelGGL(angle_vector_to_orientation_kernel, dim3(dodt::ceil_div(n, 256)), dim3(256),
                       0, ctx->stream, d_angle_vectors, n, d_n, d_orientations_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

int dodt_box_4c_decode(dodt_ctx* ctx, const float* d_top_anchors, const float* d_offsets, int n,
                       const int32_t* d_n, const float plane[4], const float bev_extents[4],
                       float* d_boxes_3d_out, float* d_pred_anchors_out, float* d_bev_tf_out) {
    DODT_REQUIRE(ctx && d_top_anchors && d_offsets && plane && bev_extents,
                 "dodt_box_4c_decode: NULL argument");
    if (n <= 0) return DODT_OK;
    DecodeParams P;
    P.a = plane[0]; P.b = plane[1]; P.c = plane[2]; P.d = plane[3];
    P.x_min = bev_extents[0]; P.x_max = bev_extents[1];
    P.z_min = bev_extents[2]; P.z_max = bev_extents[3];
    hipLaunchKernelGGL(box_4c_decode_kernel, dim3(dodt::ceil_div(n, 256)), dim3(256), 0,
                       ctx->stream, d_top_anchors, d_offsets, n, d_n, P, d_boxes_3d_out,
                       d_pred_anchors_out, d_bev_tf_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

}  // extern "C"
