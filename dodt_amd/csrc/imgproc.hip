// Image preprocessing for gfx950:
// ImgFeatureExtractor.preprocess_input, avod/core/feature_extractors/
// img_feature_extractor.py:16-35 = tf.image.resize_images (legacy bilinear,
// align_corners=False: src = dst * in/out) then per-channel mean subtraction.
// HBM-bound, trivial: one lane per output pixel, uint8 in, float32 NHWC out.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
img_preprocess_kernel(const uint8_t* __restrict__ img, int in_h, int in_w, int out_h, int out_w,
                      int out_c, float m0, float m1, float m2, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= out_h * out_w) return;
    const int oy = t / out_w, ox = t - oy * out_w;
    const float sy = (float)in_h / (float)out_h, sx = (float)in_w / (float)out_w;
    const float fy = (float)oy * sy, fx = (float)ox * sx;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min(y0 + 1, in_h - 1), x1 = min(x0 + 1, in_w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const uint8_t* tl = img + ((size_t)y0 * in_w + x0) * 3;
    const uint8_t* tr = img + ((size_t)y0 * in_w + x1) * 3;
    const uint8_t* bl = img + ((size_t)y1 * in_w + x0) * 3;
    const uint8_t* br = img + ((size_t)y1 * in_w + x1) * 3;
    const float mean[3] = {m0, m1, m2};
    float* o = out + (size_t)t * out_c;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = (float)tl[k], b = (float)tr[k], c = (float)bl[k], d = (float)br[k];
        const float top = a + (b - a) * lx;
        const float bot = c + (d - c) * lx;
        o[k] = (top + (bot - top) * ly) - mean[k];
    }
    for (int k = 3; k < out_c; ++k) o[k] = 0.0f;
}

}  // namespace

extern "C" int dodt_img_preprocess(dodt_ctx* ctx, const uint8_t* d_img_u8, int in_h, int in_w,
                                   int out_h, int out_w, int out_c, const float mean_rgb[3],
                                   float* d_out) {
    DODT_REQUIRE(ctx && d_img_u8 && d_out && mean_rgb, "dodt_img_preprocess: NULL argument");
    DODT_REQUIRE(in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0 && out_c >= 3,
                 "dodt_img_preprocess: bad sizes");
    hipLaunchKernelGGL(img_preprocess_kernel, dim3(dodt::ceil_div(out_h * out_w, 256)), dim3(256),
                       0, ctx->stream, d_img_u8, in_h, in_w, out_h, out_w, out_c, mean_rgb[0],
                       mean_rgb[1], mean_rgb[2], d_out);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}
