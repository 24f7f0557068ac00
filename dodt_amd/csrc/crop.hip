// Bilinear ROI crop for gfx950: tf.image.crop_and_resize
// with box_ind = 0, extrapolation_value = 0 (TF-1.3 crop_and_resize_op.cc), call
// sites avod/core/models/dt_rpn_model.py:418-428 (3x3 on 1-channel maps) and
// models/dt_avod_model.py:253-273 (7x7 on the 32-channel maps).
//
// HBM/L2 gather-bound.  One lane per (box, iy, ix, 4-channel group): with C = 32
// eight adjacent lanes read one pixel's 128 contiguous bytes per tap (float4
// each) and write 128 contiguous output bytes.  Float32, unfused, in TF's order:
// top = tl + (tr - tl) * lx ; bot = bl + (br - bl) * lx ; out = top + (bot - top) * ly.
#include "common.h"

namespace {

template <int VEC>
__global__ void __launch_bounds__(256)
crop_kernel(const float* __restrict__ img, int H, int W, int C, const float* __restrict__ boxes,
            int n, const int* __restrict__ d_n, int ch, int cw, float* __restrict__ out,
            long long out_stride) {
    const int groups = C / VEC;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lim = d_n ? min(*d_n, n) : n;
    const long long total = (long long)lim * ch * cw * groups;
    if (t >= total) return;
    const int g = (int)(t % groups);
    long long r = t / groups;
    const int ix = (int)(r % cw); r /= cw;
    const int iy = (int)(r % ch);
    const int b = (int)(r / ch);
    const float4 bx = reinterpret_cast<const float4*>(boxes)[b];
    const float y1 = bx.x, x1 = bx.y, y2 = bx.z, x2 = bx.w;
    const float hm1 = (float)(H - 1), wm1 = (float)(W - 1);
    const float hs = (ch > 1) ? (y2 - y1) * hm1 / (float)(ch - 1) : 0.0f;
    const float ws = (cw > 1) ? (x2 - x1) * wm1 / (float)(cw - 1) : 0.0f;
    const float in_y = (ch > 1) ? y1 * hm1 + (float)iy * hs : 0.5f * (y1 + y2) * hm1;
    const float in_x = (cw > 1) ? x1 * wm1 + (float)ix * ws : 0.5f * (x1 + x2) * wm1;
    float* o = out + (size_t)b * out_stride + (((size_t)iy * cw + ix) * C + (size_t)g * VEC);
    const bool ok = (in_y >= 0.0f) && (in_y <= hm1) && (in_x >= 0.0f) && (in_x <= wm1);
    if (!ok) {  // also catches NaN coordinates
#pragma unroll
        for (int k = 0; k < VEC; ++k) o[k] = 0.0f;
        return;
    }
    const int ty = (int)floorf(in_y), by = (int)ceilf(in_y);
    const int lx_i = (int)floorf(in_x), rx_i = (int)ceilf(in_x);
    const float ly = in_y - (float)ty, lx = in_x - (float)lx_i;
    const float* ptl = img + ((size_t)ty * W + lx_i) * C + (size_t)g * VEC;
    const float* ptr = img + ((size_t)ty * W + rx_i) * C + (size_t)g * VEC;
    const float* pbl = img + ((size_t)by * W + lx_i) * C + (size_t)g * VEC;
    const float* pbr = img + ((size_t)by * W + rx_i) * C + (size_t)g * VEC;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const float tl = ptl[k], tr = ptr[k], bl = pbl[k], br = pbr[k];
        const float top = tl + (tr - tl) * lx;
        const float bot = bl + (br - bl) * lx;
        o[k] = top + (bot - top) * ly;
    }
}

}  // namespace

extern "C" int dodt_crop_and_resize(dodt_ctx* ctx, const float* d_image, int H, int W, int C,
                                    const float* d_boxes, int n, const int32_t* d_n, int crop_h,
                                    int crop_w, float* d_out) {
    return dodt_crop_and_resize_strided(ctx, d_image, H, W, C, d_boxes, n, d_n, crop_h, crop_w, d_out,
                                        (long long)crop_h * crop_w * C);
}

extern "C" int dodt_crop_and_resize_strided(dodt_ctx* ctx, const float* d_image, int H, int W, int C,
                                            const float* d_boxes, int n, const int32_t* d_n, int crop_h,
                                            int crop_w, float* d_out, long long out_box_stride) {
    DODT_REQUIRE(ctx && d_image && d_out && (n == 0 || d_boxes),
                 "dodt_crop_and_resize: NULL argument");
    DODT_REQUIRE(H > 0 && W > 0 && C > 0 && crop_h > 0 && crop_w > 0 && n >= 0,
                 "dodt_crop_and_resize: bad sizes");
    DODT_REQUIRE(out_box_stride >= (long long)crop_h * crop_w * C,
                 "dodt_crop_and_resize: output stride %lld shorter than a crop", out_box_stride);
    if (n == 0) return DODT_OK;
    // float4 stores need 16-byte aligned crops
    const int vec = (C % 4 == 0 && out_box_stride % 4 == 0) ? 4 : 1;
    const long long total = (long long)n * crop_h * crop_w * (C / vec);
    const int blocks = (int)((total + 255) / 256);
    if (vec == 4)
        hipLaunchKernelGGL(crop_kernel<4>, dim3(blocks), dim3(256), 0, ctx->stream, d_image, H, W,
                           C, d_boxes, n, d_n, crop_h, crop_w, d_out, out_box_stride);
    else
        hipLaunchKernelGGL(crop_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, d_image, H, W,
                           C, d_boxes, n, d_n, crop_h, crop_w, d_out, out_box_stride);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}
