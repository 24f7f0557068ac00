// BEV voxel/slice generator for gfx950.
//
// Reference behaviour: avod/core/bev_generators/bev_slices.py:33-150,
// wavedata/.../core/voxel_grid_2d.py:43-160, obj_detection/obj_utils.py:453-500,
// core/calib_utils.py:484-523 (see oracle/points.py for the restatement).
//
// Design (HBM-bound; algorithmic traffic = 16 B/point read + one write of the map):
//   1. hipMemsetAsync the (Z,X,S+1) output to zero -- that IS the one write of
//      the dense map; everything afterwards touches only occupied cells.
//   2. vox_scatter, one lane per point, coalesced 16-B loads: velo->cam in
//      float64, frustum + extents + slice tests, then per member slice an
//      atomicMax of a packed 32-bit key INTO THE OUTPUT WORD ITSELF.  The key is
//      ~((ybin << 25) | point_index): the maximum is the point with the lowest
//      y-bin and, within it, the lowest original index -- the reference's
//      "first row after lexsort(x, z, y)" rule.  0 = empty = 0.0f, so untouched
//      cells need no second pass.  The lane that turns a word non-zero appends
//      the word's offset to a list.  Density counts accumulate in channel S the
//      same way (atomicAdd), the anchor-filter occupancy as bits (atomicOr).
//   3. vox_finalize, one lane per list entry: decode the winning point, redo its
//      transform (bitwise identical), write (height - lo_s) / w as float32 over
//      the key; density words become min(1, ln(n+1)/ln 16).
#include "common.h"

#include <cmath>

namespace {

constexpr int kMaxSlices = 8;
constexpr int kIdxBits = 25;

struct VoxParams {
    int format, n, S, X, Z;
    int cx_min, cy_min, cz_min;
    int occ_words_per_row;
    int origin_off;  // word offset of cell (x=0, z=0), channel 0; -1 if outside
    int has_pre;     // ego-motion warp in the velodyne frame in front of everything else
    double pre_t[3], pre_r[9];
    double m[12];
    double p2[12];
    double im_w, im_h;
    double a, b, c, d, nrm;
    double ext[6];
    double vs;
    double d_hi[kMaxSlices], d_lo[kMaxSlices];  // plane.d - hi_s / - lo_s
    double lo[kMaxSlices];                      // lo_s
    double per_div;
    double dens_d_hi, dens_d_lo, occ_d_hi, occ_d_lo;
    float quirk[kMaxSlices];  // value of the origin cell for a slice with <= 1 member
    float dens_table[16];     // min(1, ln(n+1)/ln 16), n = 0..15
};

// counters layout (uint32): [0] list length, [1] error flags, [2+s] members of slice s
constexpr int kCntList = 0, kCntErr = 1, kCntSlice = 2;

// Point i in the rectified camera frame.  Sum order = k ascending fma chain,
// which is what a BLAS dgemm micro-kernel does for the reference's np.dot.
// warp (velodyne input only): the ego-motion registration of a pair's second frame,
// point_cloud_transform (avod/datasets/kitti/kitti_tracking_dataset.py:324-335): the float32
// xyz become float64, (p + trans) @ matrix, and are stored back into the float32 array the
// points came from -- one rounding to float32 -- before lidar_to_cam_frame reads them.
__device__ __forceinline__ bool load_point(const void* pts, const VoxParams& P, int i,
                                           double& x, double& y, double& z, bool warp = true) {
    if (P.format == DODT_PTS_VELO_XYZI) {
        const float4 v = reinterpret_cast<const float4*>(pts)[i];
        double vx = v.x, vy = v.y, vz = v.z;
        if (warp && P.has_pre) {
            const double ax = vx + P.pre_t[0], ay = vy + P.pre_t[1], az = vz + P.pre_t[2];
            vx = (double)(float)fma(az, P.pre_r[6], fma(ay, P.pre_r[3], ax * P.pre_r[0]));
            vy = (double)(float)fma(az, P.pre_r[7], fma(ay, P.pre_r[4], ax * P.pre_r[1]));
            vz = (double)(float)fma(az, P.pre_r[8], fma(ay, P.pre_r[5], ax * P.pre_r[2]));
        }
        x = fma(P.m[2], vz, fma(P.m[1], vy, P.m[0] * vx)) + P.m[3];
        y = fma(P.m[6], vz, fma(P.m[5], vy, P.m[4] * vx)) + P.m[7];
        z = fma(P.m[10], vz, fma(P.m[9], vy, P.m[8] * vx)) + P.m[11];
        if (!(z > 0.0)) return false;
        const double un = fma(P.p2[2], z, fma(P.p2[1], y, P.p2[0] * x)) + P.p2[3];
        const double vn = fma(P.p2[6], z, fma(P.p2[5], y, P.p2[4] * x)) + P.p2[7];
        const double w = fma(P.p2[10], z, fma(P.p2[9], y, P.p2[8] * x)) + P.p2[11];
        const double u = un / w, vv = vn / w;
        return (u > 0.0) && (u < P.im_w) && (vv > 0.0) && (vv < P.im_h);
    } else {
        const double* p = reinterpret_cast<const double*>(pts);
        x = p[i];
        y = p[(size_t)P.n + i];
        z = p[2 * (size_t)P.n + i];
        return true;
    }
}

__device__ __forceinline__ bool in_extents(const VoxParams& P, double x, double y, double z) {
    return x > P.ext[0] && x < P.ext[1] && y > P.ext[2] && y < P.ext[3] && z > P.ext[4] &&
           z < P.ext[5];
}

__global__ void __launch_bounds__(256)
vox_scatter(const void* __restrict__ pts, const VoxParams P, uint32_t* __restrict__ out,
            uint32_t* __restrict__ occ, uint32_t* __restrict__ list,
            uint32_t* __restrict__ counters, uint32_t list_cap) {
    // Per-workgroup staging of everything that would otherwise hammer ONE global word:
    // slice member counts and the append cursor of the touched-cell list (18k same-address
    // atomics per frame cost ~100 us; one atomic per workgroup and counter costs ~nothing).
    __shared__ uint32_t s_cnt[kMaxSlices];
    __shared__ uint32_t s_n;                       // entries staged by this workgroup
    __shared__ uint32_t s_base;                    // their position in the global list
    __shared__ uint32_t s_items[256 * (kMaxSlices + 1)];
    const int tid = threadIdx.x;
    if (tid < kMaxSlices) s_cnt[tid] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();

    const int i = blockIdx.x * blockDim.x + tid;
    double x, y, z;
    bool ok = i < P.n && load_point(pts, P, i, x, y, z);
    ok = ok && in_extents(P, x, y, z);
    bool occ_ok = ok;
    double occ_dotp = 0.0;
    int occ_xi = 0, occ_zi = 0;
    if (ok) {
        // (plane + [0,0,0,-off]) . [x,y,z,1] < 0 ; d - off is folded on the host
        const double dotp = fma(P.c, z, fma(P.b, y, P.a * x));
        const int xi = (int)floor(x / P.vs) - P.cx_min;
        const int yb = (int)floor(y / P.vs) - P.cy_min;
        const int zi = (int)floor(z / P.vs) - P.cz_min;
        if (xi < 0 || xi >= P.X || zi < 0 || zi >= P.Z || yb < 0 || yb > 126) {
            atomicOr(&counters[kCntErr], 1u);  // reference: ValueError("Extents are smaller ...")
            occ_ok = false;
        } else {
            const int C = P.S + 1;
            const uint32_t base = (uint32_t)(((P.Z - 1 - zi) * P.X + xi) * C);
            const uint32_t key = ~(((uint32_t)yb << kIdxBits) | (uint32_t)i);
            for (int s = 0; s < P.S; ++s) {
                const bool member = ((dotp + P.d_hi[s]) < 0.0) != ((dotp + P.d_lo[s]) < 0.0);
                if (member) {
                    const uint32_t old = atomicMax(&out[base + s], key);
                    if (old == 0u) s_items[atomicAdd(&s_n, 1u)] = base + s;
                    atomicAdd(&s_cnt[s], 1u);
                }
            }
            if (((dotp + P.dens_d_hi) < 0.0) != ((dotp + P.dens_d_lo) < 0.0)) {
                const uint32_t old = atomicAdd(&out[base + P.S], 1u);
                if (old == 0u) s_items[atomicAdd(&s_n, 1u)] = base + P.S;
            }
            occ_dotp = dotp;
            occ_xi = xi;
            occ_zi = zi;
        }
    }
    if (occ != nullptr) {
        // The anchor filter's grid comes from the cloud as read from the file: the reference
        // re-reads it without the ego-motion warp (kitti_tracking_utils.py:98-126 ->
        // tracking_utils.get_lidar_point_cloud) -- reproduced, not fixed
        if (P.has_pre && P.format == DODT_PTS_VELO_XYZI) {
            double ux, uy, uz;
            occ_ok = i < P.n && load_point(pts, P, i, ux, uy, uz, false) && in_extents(P, ux, uy, uz);
            if (occ_ok) {
                occ_dotp = fma(P.c, uz, fma(P.b, uy, P.a * ux));
                occ_xi = (int)floor(ux / P.vs) - P.cx_min;
                occ_zi = (int)floor(uz / P.vs) - P.cz_min;
                const int yb = (int)floor(uy / P.vs) - P.cy_min;
                if (occ_xi < 0 || occ_xi >= P.X || occ_zi < 0 || occ_zi >= P.Z || yb < 0 || yb > 126) {
                    atomicOr(&counters[kCntErr], 1u);
                    occ_ok = false;
                }
            }
        }
        if (occ_ok && (((occ_dotp + P.occ_d_hi) < 0.0) != ((occ_dotp + P.occ_d_lo) < 0.0)))
            atomicOr(&occ[occ_zi * P.occ_words_per_row + (occ_xi >> 5)], 1u << (occ_xi & 31));
    }
    __syncthreads();
    if (tid < P.S && s_cnt[tid]) atomicAdd(&counters[kCntSlice + tid], s_cnt[tid]);
    if (tid == 0) s_base = s_n ? atomicAdd(&counters[kCntList], s_n) : 0u;
    __syncthreads();
    const uint32_t n = s_n, gbase = s_base;
    for (uint32_t k = tid; k < n; k += 256) {
        if (gbase + k < list_cap) list[gbase + k] = s_items[k];
        else atomicOr(&counters[kCntErr], 2u);     // cannot happen with cap = 3 n; reported
    }
}

__global__ void __launch_bounds__(256)
vox_finalize(const void* __restrict__ pts, const VoxParams P, uint32_t* __restrict__ out,
             const uint32_t* __restrict__ list, const uint32_t* __restrict__ counters,
             uint32_t list_cap) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t len = counters[kCntList];
    if (len > list_cap) len = list_cap;
    float* outf = reinterpret_cast<float*>(out);
    const int C = P.S + 1;
    if (e >= len) {
        // tail lanes: slices with <= 1 member become the single point (0,0,0)
        const uint32_t s = e - len;
        if (s < (uint32_t)P.S && counters[kCntSlice + s] <= 1u && P.origin_off >= 0)
            outf[P.origin_off + s] = P.quirk[s];
        return;
    }
    const uint32_t off = list[e];
    const int ch = (int)(off % (uint32_t)C);
    if (ch == P.S) {
        const uint32_t n = out[off];
        outf[off] = (n >= 15u) ? 1.0f : P.dens_table[n];
        return;
    }
    if (counters[kCntSlice + ch] <= 1u) {
        outf[off] = ((int)off == P.origin_off + ch) ? P.quirk[ch] : 0.0f;
        return;
    }
    const uint32_t key = ~out[off];
    const int idx = (int)(key & ((1u << kIdxBits) - 1u));
    double x, y, z;
    (void)load_point(pts, P, idx, x, y, z);
    // dist_to_plane, numpy elementwise order, one rounding per operation
    const double t = __dadd_rn(
        __dadd_rn(__dadd_rn(__dmul_rn(P.a, x), __dmul_rn(P.b, y)), __dmul_rn(P.c, z)), P.d);
    const double h = t / P.nrm;
    outf[off] = (float)(__dsub_rn(h, P.lo[ch]) / P.per_div);
}

}  // namespace

extern "C" int dodt_bev_slices(dodt_ctx* ctx, const void* d_points, int n_points,
                               const dodt_bev_params* bp, float* d_bev_out,
                               uint32_t* d_occ_bits_out) {
    DODT_REQUIRE(ctx && bp && d_bev_out, "dodt_bev_slices: NULL argument");
    DODT_REQUIRE(n_points >= 0 && (n_points == 0 || d_points),
                 "dodt_bev_slices: bad point buffer");
    DODT_REQUIRE(bp->point_format == DODT_PTS_VELO_XYZI || bp->point_format == DODT_PTS_CAM_3XN,
                 "dodt_bev_slices: unknown point_format %d", bp->point_format);
    DODT_REQUIRE(bp->num_slices >= 1 && bp->num_slices <= kMaxSlices,
                 "dodt_bev_slices: num_slices %d not in [1,%d]", bp->num_slices, kMaxSlices);
    DODT_REQUIRE(n_points < (1 << kIdxBits), "dodt_bev_slices: n_points %d >= 2^25", n_points);
    DODT_REQUIRE(bp->voxel_size > 0, "dodt_bev_slices: voxel_size must be positive");

    VoxParams P;
    P.format = bp->point_format;
    P.n = n_points;
    P.S = bp->num_slices;
    const double vs = bp->voxel_size;
    // voxel_grid_2d.py:125-128: min = floor(ext_min / vs), max = ceil(ext_max / vs - 1)
    const double minx = std::floor(bp->extents[0] / vs), maxx = std::ceil(bp->extents[1] / vs - 1);
    const double miny = std::floor(bp->extents[2] / vs), maxy = std::ceil(bp->extents[3] / vs - 1);
    const double minz = std::floor(bp->extents[4] / vs), maxz = std::ceil(bp->extents[5] / vs - 1);
    P.X = (int)(maxx - minx + 1);
    P.Z = (int)(maxz - minz + 1);
    DODT_REQUIRE(P.X > 0 && P.Z > 0, "Extents are the wrong shape");
    DODT_REQUIRE(maxy - miny + 1 <= 127, "dodt_bev_slices: more than 127 y-bins (%g)",
                 maxy - miny + 1);
    DODT_REQUIRE((double)P.X * P.Z * (P.S + 1) < 4.0e9, "dodt_bev_slices: grid too large");
    P.cx_min = (int)minx;
    P.cy_min = (int)miny;
    P.cz_min = (int)minz;
    P.occ_words_per_row = dodt::ceil_div(P.X, 32);
    for (int k = 0; k < 12; ++k) { P.m[k] = bp->velo_to_cam[k]; P.p2[k] = bp->p2[k]; }
    P.has_pre = bp->has_pre_transform != 0;
    DODT_REQUIRE(!P.has_pre || bp->point_format == DODT_PTS_VELO_XYZI,
                 "dodt_bev_slices: the ego-motion pre-transform applies to velodyne points");
    for (int k = 0; k < 3; ++k) P.pre_t[k] = bp->pre_translate[k];
    for (int k = 0; k < 9; ++k) P.pre_r[k] = bp->pre_rotate[k];
    P.im_w = bp->im_w;
    P.im_h = bp->im_h;
    P.a = bp->plane[0]; P.b = bp->plane[1]; P.c = bp->plane[2]; P.d = bp->plane[3];
    P.nrm = std::sqrt(P.a * P.a + P.b * P.b + P.c * P.c);
    DODT_REQUIRE(P.nrm > 0, "dodt_bev_slices: degenerate ground plane");
    for (int k = 0; k < 6; ++k) P.ext[k] = bp->extents[k];
    P.vs = vs;
    // bev_slices.py:30-31,62-63: float64 arithmetic on the float32-rounded scalars
    const double per_div = (bp->height_hi - bp->height_lo) / P.S;
    P.per_div = per_div;
    const double h0 = P.d / P.nrm;  // height of the substitute point (0,0,0)
    for (int s = 0; s < P.S; ++s) {
        const double lo = bp->height_lo + s * per_div;
        const double hi = lo + per_div;
        P.lo[s] = lo;
        P.d_lo[s] = P.d + (-lo);
        P.d_hi[s] = P.d + (-hi);
        P.quirk[s] = (float)((h0 - lo) / per_div);
    }
    P.dens_d_lo = P.d + (-bp->height_lo);
    P.dens_d_hi = P.d + (-bp->height_hi);
    P.occ_d_lo = P.d + (-bp->occ_lo);
    P.occ_d_hi = P.d + (-bp->occ_hi);
    for (int n = 0; n < 16; ++n)
        P.dens_table[n] = (float)std::fmin(1.0, std::log((double)n + 1.0) / std::log(16.0));
    {
        const int ox = 0 - P.cx_min, oz = 0 - P.cz_min;
        P.origin_off = (ox >= 0 && ox < P.X && oz >= 0 && oz < P.Z)
                           ? ((P.Z - 1 - oz) * P.X + ox) * (P.S + 1)
                           : -1;
    }

    const size_t cells = (size_t)P.X * P.Z * (P.S + 1);
    size_t cap = (size_t)3 * n_points;
    if (cap > cells) cap = cells;
    const size_t counters_bytes = 256;
    int rc = ctx->vox_ws.reserve(counters_bytes + (cap + 16) * sizeof(uint32_t));
    if (rc) return rc;
    uint32_t* counters = reinterpret_cast<uint32_t*>(ctx->vox_ws.ptr);
    uint32_t* list = counters + counters_bytes / sizeof(uint32_t);
    uint32_t* out = reinterpret_cast<uint32_t*>(d_bev_out);

    DODT_HIP_CHECK(hipMemsetAsync(out, 0, cells * sizeof(uint32_t), ctx->stream));
    DODT_HIP_CHECK(hipMemsetAsync(counters, 0, counters_bytes, ctx->stream));
    if (d_occ_bits_out)
        DODT_HIP_CHECK(hipMemsetAsync(d_occ_bits_out, 0,
                                      (size_t)P.Z * P.occ_words_per_row * sizeof(uint32_t),
                                      ctx->stream));
    if (n_points > 0) {
        hipLaunchKernelGGL(vox_scatter, dim3(dodt::ceil_div(n_points, 256)), dim3(256), 0,
                           ctx->stream, d_points, P, out, d_occ_bits_out, list, counters,
                           (uint32_t)cap);
        DODT_LAUNCH_CHECK();
    }
    const int fin = (int)cap + P.S;
    hipLaunchKernelGGL(vox_finalize, dim3(dodt::ceil_div(fin, 256)), dim3(256), 0, ctx->stream,
                       d_points, P, out, list, counters, (uint32_t)cap);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

extern "C" int dodt_bev_status(dodt_ctx* ctx, int* flags) {
    DODT_REQUIRE(ctx && flags, "dodt_bev_status: NULL argument");
    *flags = 0;
    if (!ctx->vox_ws.ptr) return DODT_OK;
    uint32_t host[2] = {0, 0};
    DODT_HIP_CHECK(hipMemcpyAsync(host, ctx->vox_ws.ptr, sizeof(host), hipMemcpyDeviceToHost,
                                  ctx->stream));
    DODT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *flags = (int)(host[kCntErr] & 3u);
    return DODT_OK;
}
