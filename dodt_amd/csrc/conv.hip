// VGG-pyramid feature extractor for gfx950.
//
// Reference topology: avod/core/feature_extractors/bev_vgg_pyramid.py:57-169 and
// img_vgg_pyramid.py:58-171 (see oracle/extractors.py); 1x1 bottleneck
// avod/core/models/dt_rpn_model.py:298-322.  Every conv is slim.conv2d /
// slim.conv2d_transpose with batch-norm (inference form, no gamma, eps 1e-3) and
// ReLU, no bias.
//
// Data layout in HBM: the network input and the returned feature map are NHWC
// float32 (the reference's layout); every activation in between is channel-blocked
// [frame][C/8][H][W][8] ("CB8", see conv_kernels.h), one buffer per pyramid level.
// The decoder's concat inputs are single buffers whose channel planes the two
// producing kernels write directly (conv_k -> planes 0.., upconv_k -> the planes
// after them), so no concat copy exists.  Both frames of a pair are one batch.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include <array>

#include "common.h"
#include "conv_kernels.h"
#include "conv_variants.h"

namespace {

using dodt::ConvArgs;

using dodt::KernelVariant;
using dodt::Inst;
using dodt::InstSmall;
using dodt::InstWino;
using dodt::InstWino43;
using dodt::InstDeconvDma;
using dodt::tail_only;
using dodt::f32x4;

const std::vector<KernelVariant>& variants() {
    static const std::vector<KernelVariant> v = {
        InstSmall<32, 16, 6>::variant(),
        InstSmall<16, 12, 4>::variant(),
        InstSmall<16, 16, 6>::variant(),
        InstSmall<16, 16, 4>::variant(),
        Inst<32, 16, 4, 1, 32, false>::variant(),
        Inst<16, 16, 4, 1, 32, false>::variant(),
        Inst<16, 8, 4, 1, 64, false>::variant(),
        Inst<16, 12, 4, 1, 32, false>::variant(),
        Inst<8, 8, 4, 1, 32, false>::variant(),
        Inst<8, 8, 4, 1, 64, false>::variant(),
        Inst<8, 4, 2, 2, 128, false>::variant(),
        Inst<8, 4, 4, 1, 64, false>::variant(),
        Inst<4, 4, 2, 2, 128, false>::variant(),
        Inst<4, 4, 4, 1, 64, false>::variant(),
        // quarter-size tiles for tail launches (and tiny layers)
        tail_only(Inst<32, 4, 4, 1, 32, false>::variant()),
        tail_only(Inst<16, 4, 4, 1, 32, false>::variant()),
        tail_only(Inst<8, 4, 4, 1, 32, false>::variant()),
        tail_only(Inst<4, 4, 4, 1, 32, false>::variant()),
        Inst<16, 4, 4, 1, 32, true>::variant(),
        Inst<8, 4, 4, 1, 32, true>::variant(),
        Inst<4, 4, 4, 1, 32, true>::variant(),
        Inst<16, 4, 4, 1, 64, true>::variant(),
        Inst<8, 4, 4, 1, 64, true>::variant(),
        Inst<4, 4, 4, 1, 64, true>::variant(),
        // fp32 Winograd F(2x2,3x3): 16 x 16 px x 64 ch, 32 x 16 px x 32 ch
        InstWino<1, 4>::variant(),
        InstWino<2, 2>::variant(),
        InstWino<1, 2>::variant(),    // 16 x 16 px x 32 ch, 128 accumulators: two workgroups per CU
        InstWino43::variant(),        // F(4x4,3x3): 32 x 16 px x 32 ch, one workgroup per CU
        InstDeconvDma<2>::variant(),  // transposed conv, LDS-DMA staged: 16 x 16 input px x 32 ch
        InstDeconvDma<1>::variant(),  //   ... x 16 ch: twice the items, for layers with few of them
    };
    static const std::vector<KernelVariant> all = [] {
        std::vector<KernelVariant> a = v;
        for (const KernelVariant& b : dodt::bf16_variants()) a.push_back(b);
        for (const KernelVariant& b : dodt::split_variants()) a.push_back(b);
        return a;
    }();
    return all;
}

}  // namespace

// which form the fp32 3x3 stride-1 layers take in this process (read once)
extern "C" int dodt_conv_mode(void) {
    static const int mode = [] {
        const char* e = getenv("DODT_CONV_WINO");
        const int m = e ? atoi(e) : DODT_CONV_MODE_DEFAULT;
        return (m == 0 || m == 1 || m == 2 || m == 4) ? m : DODT_CONV_MODE_DEFAULT;
    }();
    return mode;
}

namespace {

// bf16 LDS-DMA conv kernel: one work queue per group of blocks that share an XCD (DODT_CONV_BF16_XCD=0: one queue)
bool bf16_xcd_queue() {
    static const bool on = !(getenv("DODT_CONV_BF16_XCD") && atoi(getenv("DODT_CONV_BF16_XCD")) == 0);
    return on;
}
// the same for the fp32 Winograd F(2x2) / transposed-conv kernels (DODT_CONV_F32_XCD=0: one queue): both stacks
// 2.90 -> 2.80 ms alone, nothing in the pipeline, half the excess fabric traffic
bool f32_xcd_queue() {
    static const bool on = !(getenv("DODT_CONV_F32_XCD") && atoi(getenv("DODT_CONV_F32_XCD")) == 0);
    return on;
}
bool variant_xcd_queue(const dodt::KernelVariant& v) {
    if (v.dma || (v.deconv_dma && v.bf16)) return bf16_xcd_queue();
    if ((v.wino && v.wino_m != 4) || (v.deconv_dma && !v.bf16)) return f32_xcd_queue();
    return false;
}

// smallest padded pixel count wins; ties go to the larger output tile
int pick_variant(bool deconv, int H, int W, int Cin, int Cout, bool bf16, int parts, int batch,
                 int num_cus) {
    const auto& vs = variants();
    const bool small = Cin < dodt::kCK;
    int best = -1;
    double best_cost = 1e300;
    // fp32 3x3 stride-1 convs with >= 4 chunks of input channels run as Winograd.
    // dodt_conv_mode(): 2 (default) = F(2x2,3x3) with 128 accumulators, two workgroups per CU
    // (wino_kernels.h: both stacks 3.0 ms, rounding noise at the direct kernels' level), 4 =
    // F(4x4,3x3) (wino43_kernel.h: 2.8 ms, but its fp32 accumulation in the transformed domain is
    // 7x noisier, which costs a third of the end-to-end agreement with the exact result: opt-in), 1 = the
    // 256-accumulator F(2x2) variants, one workgroup per CU (no faster than direct), 0 = the direct
    // kernels (4.8 ms).
    static const int wino_mode = dodt_conv_mode();
    if (wino_mode > 0 && !deconv && !bf16 && parts == 1 && Cin >= 32 && Cin % 16 == 0) {
        for (size_t i = 0; i < vs.size(); ++i) {
            if (!vs[i].wino || Cout % vs[i].BN != 0) continue;
            if (vs[i].wino_m == 4) {
                if (wino_mode == 4) return (int)i;
                continue;
            }
            if ((wino_mode == 1) == (vs[i].blocks_per_cu == 2)) continue;
            if (best < 0 || vs[i].BN > vs[best].BN) best = (int)i;
        }
        if (best >= 0) return best;
    }
    // fp32 transposed convs: the LDS-DMA staged kernel (deconv_kernel.h; DODT_CONV_DECONV_DMA=0: the
    // register-staged direct kernel's deconv instantiation)
    static const bool deconv_dma = !(getenv("DODT_CONV_DECONV_DMA") && atoi(getenv("DODT_CONV_DECONV_DMA")) == 0);
    // (bf16 transposed convs take the same kernel on the bf16 MFMA since round 3; DODT_CONV_DECONV_DMA=0
    //  puts both back on the register-staged template)
    if (deconv_dma && deconv && parts == 1 && Cin >= 32 && Cin % 16 == 0) {
        // 32-channel tiles unless that leaves fewer than four items per CU (the CUs are MFMA-bound:
        // what counts is how evenly the items spread)
        const int n32 = dodt::ceil_div(H, 16) * dodt::ceil_div(W, 16) * (Cout / 32) * batch;
        const int want_bn = (Cout % 32 == 0 && n32 >= 4 * num_cus) ? 32 : 16;
        for (size_t i = 0; i < vs.size(); ++i)
            if (vs[i].deconv_dma && vs[i].bf16 == bf16 && vs[i].BN == want_bn && Cout % want_bn == 0) return (int)i;
    }
    // bf16 3x3 stride-1 layers: the LDS-DMA staged kernel (conv_bf16_dma.h: scalar-only copy issue
    // between the MFMAs, accumulators pinned in place; stacks alone as fast as the template's bf16
    // instantiation, frame-pair pipeline 7 % faster: 625 against 585 pairs/s).  DODT_CONV_BF16_DMA=0
    // selects the template.
    static const bool bf16_dma = !(getenv("DODT_CONV_BF16_DMA") && atoi(getenv("DODT_CONV_BF16_DMA")) == 0);
    if (bf16_dma && bf16 && parts == 1 && !deconv && Cin >= 32 && Cin % 32 == 0) {
        // the widest channel tile that still leaves four items per CU (two resident workgroups,
        // two rounds); if none does, the one with the most items (half-height tiles were tried for
        // the small maps: slower, the per-item cost grows faster than the balance improves)
        // the level-1 layers (32 output channels, 2 / 4 chunks): the streaming kernel (DODT_CONV_BF16_STREAM=0: not)
        static const bool stream = !(getenv("DODT_CONV_BF16_STREAM") && atoi(getenv("DODT_CONV_BF16_STREAM")) == 0);
        // (DODT_CONV_BF16_STREAM_LDS=<KB>: the deepest ring within that much LDS per workgroup; the table lists them deepest first)
        static const int stream_lds = getenv("DODT_CONV_BF16_STREAM_LDS") ? atoi(getenv("DODT_CONV_BF16_STREAM_LDS")) * 1024 : 1 << 30;
        if (stream && Cout == 32)
            for (size_t i = 0; i < vs.size(); ++i)
                if (vs[i].stream_nch > 0 && vs[i].stream_nch * 16 == Cin && vs[i].lds_bytes <= stream_lds) return (int)i;
        long best_n = 0;
        static const long per_cu = getenv("DODT_CONV_BF16_ITEMS_PER_CU") ? atol(getenv("DODT_CONV_BF16_ITEMS_PER_CU")) : 4;
        // 8-row tiles (three workgroups per CU): first only where the 16-row tiles gave fewer than 1.6 items per CU -- the
        // 88 x 100 / 45 x 150 maps of the deepest level: conv4_2 34 -> 30 us, stacks 1.065 -> 1.043 ms alone, nothing in
        // the pipeline; at 1.9 items per CU (the image net's level 3, 480 items) they are slower: 26 -> 29 us
        // (DODT_CONV_BF16_MT2=0: never; DODT_CONV_BF16_MT2_BELOW=<items>: another threshold)
        static const bool mt2 = !(getenv("DODT_CONV_BF16_MT2") && atoi(getenv("DODT_CONV_BF16_MT2")) == 0);
        static const long mt2_below = getenv("DODT_CONV_BF16_MT2_BELOW") ? atol(getenv("DODT_CONV_BF16_MT2_BELOW")) : 0;
        long n16 = 0;
        for (size_t i = 0; i < vs.size(); ++i)
            if (vs[i].dma && !vs[i].stream_nch && !vs[i].first2 && vs[i].TH == 16 && Cout % vs[i].BN == 0)
                n16 = std::max(n16, (long)dodt::ceil_div(H, 16) * dodt::ceil_div(W, vs[i].TW) * (Cout / vs[i].BN) * batch);
        // Round 4, late (the epilogue's cost per item fell), measured and NOT the default: a round model instead of the item
        // threshold (DODT_CONV_BF16_MT2_RULE=rounds) -- 16-row tiles take ceil(items / 2 per CU) rounds of an item time, 8-row
        // tiles ceil(items / 3 per CU) rounds of 0.625 of it; 8-row tiles where that is a tenth shorter.  It adds the BEV net's
        // level-2 / level-3 layers (1 144 and 616 items on 512 slots: a mostly empty last round): conv2_2 35 -> 30, conv3_2
        // 33 -> 28, conv3_3 35 -> 29 us, both stacks 0.875 -> 0.849 ms ALONE -- but 0.695 -> 0.71 ms side by side (three
        // workgroups per CU leave the other stack's kernels less room) and 1 013-1 027 -> 1 009 pairs/s in the pipeline.
        static const bool rounds_rule = getenv("DODT_CONV_BF16_MT2_RULE") && !strcmp(getenv("DODT_CONV_BF16_MT2_RULE"), "rounds");
        long n8 = 0;
        for (size_t i = 0; i < vs.size(); ++i)
            if (vs[i].dma && !vs[i].stream_nch && !vs[i].first2 && vs[i].TH == 8 && Cout % vs[i].BN == 0)
                n8 = std::max(n8, (long)dodt::ceil_div(H, 8) * dodt::ceil_div(W, vs[i].TW) * (Cout / vs[i].BN) * batch);
        const double t16 = (double)dodt::ceil_div((int)n16, 2 * num_cus), t8 = 0.625 * dodt::ceil_div((int)n8, 3 * num_cus);
        const bool want_mt2 = mt2 && n8 > 0 && (rounds_rule ? t8 < 0.9 * t16
                                                            : n16 < (mt2_below > 0 ? mt2_below : 8L * num_cus / 5));
        for (size_t i = 0; i < vs.size(); ++i) {
            if (!vs[i].dma || vs[i].stream_nch || vs[i].first2 || Cout % vs[i].BN != 0) continue;
            if ((vs[i].TH == 8) != want_mt2) continue;
            const long n = (long)dodt::ceil_div(H, vs[i].TH) * dodt::ceil_div(W, vs[i].TW) * (Cout / vs[i].BN) * batch;
            const bool enough = n >= per_cu * num_cus, best_enough = best_n >= per_cu * num_cus;
            if (best < 0 || (enough && !best_enough) || (enough && best_enough && vs[i].BN > vs[best].BN) ||
                (!enough && !best_enough && n > best_n)) {
                best = (int)i; best_n = n;
            }
        }
        if (best >= 0) return best;
    }
    for (size_t i = 0; i < vs.size(); ++i) {
        const KernelVariant& v = vs[i];
        if (v.wino || v.dma || v.deconv_dma) continue;
        if (v.deconv != deconv || v.small_cin != small || v.tail_only || v.bf16 != bf16 ||
            v.parts != parts || Cout % v.BN != 0)
            continue;
        static const int max_bn = getenv("DODT_CONV_MAX_BN") ? atoi(getenv("DODT_CONV_MAX_BN")) : 1024;
        if (v.BN > max_bn) continue;
        // the chunk pipeline needs >= 2 chunks (8 channels fp32, 16 channels bf16) per item
        if (small ? (Cin != v.CK || Cout != 32) : (Cin % v.CK != 0 || Cin < 32)) continue;
        const double padded = (double)dodt::ceil_div(H, v.TH) * v.TH * dodt::ceil_div(W, v.TW) * v.TW;
        double cost;
        // which model: measured.  fp32: the padded-pixel rule (the round model's choices run a
        // lone net 9 % faster but the frame-pair pipeline 4 % slower); bf16: the round model
        // (stacks 15 % faster alone, pipeline 4-6 % faster).  DODT_CONV_ROUND_MODEL=0/1 forces.
        static const int force = getenv("DODT_CONV_ROUND_MODEL") ? atoi(getenv("DODT_CONV_ROUND_MODEL")) : -1;
        const bool model = force >= 0 ? force != 0 : bf16;
        if (!model || small) {
            // padded pixels, mild preference for more work per staged byte
            cost = padded * (1.0 + 0.04 / v.MTB + 1.0 / v.BN);
        } else {
            // round model: workgroups sharing a CU share its matrix pipe, so a layer takes
            // ceil(items / CUs) item times; an item = its 32x32 MFMA tiles over the variant's
            // in-tile efficiency (2x2 tiles per wave 1.0, 2 tiles 0.9, 1 tile 0.8)
            const double items = (double)dodt::ceil_div(H, v.TH) * dodt::ceil_div(W, v.TW) * batch *
                                 (Cout / v.BN);
            const int wave_tiles = deconv ? 4 * (v.BN / 32 / v.WN)
                                          : (v.MTB / v.WM) * (v.BN / 32 / v.WN);
            const double eff = wave_tiles >= 4 ? 1.0 : wave_tiles >= 2 ? 0.9 : 0.8;
            const double units = (double)(v.TW * v.TH / 32) * (v.BN / 32);
            cost = std::ceil(items / num_cus) * units / eff;
        }
        if (cost < best_cost) { best_cost = cost; best = (int)i; }
    }
    return best;
}

// ---------------------------------------------------------------------------
// small HBM-bound helpers
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
maxpool2x2_kernel(const float* __restrict__ in, int H, int W, int planes,
                  long long in_frame_stride, float* __restrict__ out, int frames) {
    // CB8 in / CB8 out; one lane per (plane, output pixel, half of the 8 channels);
    // VALID 2x2 stride 2.  Reads the first `planes` planes of the (wider) input.
    const int OH = H / 2, OW = W / 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)frames * planes * OH * OW * 2;
    if (t >= total) return;
    const int g = (int)(t & 1);
    long long r = t >> 1;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH); r /= OH;
    const int pl = (int)(r % planes);
    const int f = (int)(r / planes);
    const float* p = in + (size_t)f * in_frame_stride + (size_t)pl * H * W * 8 +
                     ((size_t)(2 * oy) * W + 2 * ox) * 8 + g * 4;
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 8);
    const float4 c = *reinterpret_cast<const float4*>(p + (size_t)W * 8);
    const float4 d = *reinterpret_cast<const float4*>(p + (size_t)W * 8 + 8);
    float4 o;
    o.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x));
    o.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
    o.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z));
    o.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
    reinterpret_cast<float4*>(out)[t] = o;   // out index == t by construction
}

// 1x1 conv to one channel + batch-norm + ReLU; 8 lanes per pixel, float4 each
__global__ void __launch_bounds__(256)
bottleneck32_kernel(const float* __restrict__ in, long long n_pix, const float* __restrict__ w,
                    float scale, float shift, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long pix = t >> 3;
    const int g = (int)(t & 7);
    float s = 0.0f;
    if (pix < n_pix) {
        const float4 v = reinterpret_cast<const float4*>(in)[pix * 8 + g];
        const float4 k = reinterpret_cast<const float4*>(w)[g];
        s = ((v.x * k.x + v.y * k.y) + v.z * k.z) + v.w * k.w;
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if (pix < n_pix && g == 0) out[pix] = fmaxf(s * scale + shift, 0.0f);
}

__global__ void __launch_bounds__(256)
copy_rows_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n4_per_frame,
                 long long src_frame_stride4, long long dst_frame_stride4, int frames) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4_per_frame * frames) return;
    const int f = (int)(t / n4_per_frame);
    const long long i = t - (long long)f * n4_per_frame;
    dst[f * dst_frame_stride4 + i] = src[f * src_frame_stride4 + i];
}

// tf.image.resize_bilinear (TF 1.3 legacy sampling, align_corners = False: src = dst * in/out)
// of the CB8 conv4_3 map to the NHWC feature map the plain-VGG extractors return
// (bev_vgg.py:102-112, img_vgg.py:104-114), with the 1x1 bottleneck (rpn_model.py:251-267)
// on the upsampled pixel fused in.  C / 4 lanes per output pixel, one float4 of channels each:
// a wave stores 1 KB of contiguous NHWC output per instruction; the source map (9-12 MB)
// stays in L2.  HBM-bound on the output write.  float32, unfused, TF's operation order.
template <int LPP>   // lanes per pixel = C / 4
__global__ void __launch_bounds__(256)
upsample_bilinear_cb8_kernel(const float* __restrict__ in, int IH, int IW, long long in_frame_stride,
                             int OH, int OW, int frames, float* __restrict__ out,
                             const float* __restrict__ bneck_w, float bneck_scale,
                             float bneck_shift, float* __restrict__ bneck_out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long pix = t / LPP;
    const int l = (int)(t % LPP);
    const long long n_pix = (long long)frames * OH * OW;
    const bool ok = pix < n_pix;
    float dot = 0.0f;
    if (ok) {
        const int f = (int)(pix / ((long long)OH * OW));
        const int r = (int)(pix - (long long)f * OH * OW);
        const int oy = r / OW, ox = r - oy * OW;
        const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
        const float fy = (float)oy * sy, fx = (float)ox * sx;
        const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
        const int y1 = min(y0 + 1, IH - 1), x1 = min(x0 + 1, IW - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* base = in + (size_t)f * in_frame_stride + (size_t)(l >> 1) * IH * IW * 8 +
                            (l & 1) * 4;
        const f32x4 tl = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * IW + x0) * 8);
        const f32x4 tr = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * IW + x1) * 8);
        const f32x4 bl = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * IW + x0) * 8);
        const f32x4 br = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * IW + x1) * 8);
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float top = tl[k] + (tr[k] - tl[k]) * lx;
            const float bot = bl[k] + (br[k] - bl[k]) * lx;
            v[k] = top + (bot - top) * ly;
        }
        *reinterpret_cast<f32x4*>(out + (size_t)pix * (LPP * 4) + l * 4) = v;
        if (bneck_out) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(bneck_w + l * 4);
            dot = ((v[0] * w[0] + v[1] * w[1]) + v[2] * w[2]) + v[3] * w[3];
        }
    }
    if (bneck_out) {
#pragma unroll
        for (int m = 1; m < LPP; m <<= 1) dot += __shfl_xor(dot, m, 64);
        if (ok && l == 0) bneck_out[pix] = fmaxf(dot * bneck_scale + bneck_shift, 0.0f);
    }
}

// ---------------------------------------------------------------------------
// extractor object
// ---------------------------------------------------------------------------
struct Buffer {
    int H = 0, W = 0, C = 0;
    float* ptr = nullptr;
    bool bf16 = false;   // CB16 bf16 map (2 bytes per element) instead of CB8 / NHWC fp32
    int parts = 1;       // 2: split mode, [hi map of all frames | lo map of all frames]
    size_t frame_floats() const { return (size_t)H * W * C / (bf16 ? 2 : 1); }   // one part
};

// one kernel launch of a layer: a variant and the work items it walks
struct Launch {
    int variant = -1;
    int n_items = 0;
    int4* d_items = nullptr;
    float* d_w = nullptr;   // weights blocked for this variant's BN
};

struct Layer {
    std::string name;
    bool deconv = false;
    int H = 0, W = 0;  // GEMM grid (conv: output size; deconv: input size)
    int Cin = 0, Cout = 0;
    int src = -1, src_coff = 0;
    int dst = -1, dst_coff = 0;
    int variant = -1;
    Launch main, tail;   // tail.n_items == 0: single launch
    float *d_scale = nullptr, *d_shift = nullptr;
    bool loaded = false;
    int real_cin = 0;  // channels that carry data (conv1_1 of the image net: 3 of 4)
    float* d_first_w = nullptr;   // conv1_1 of a bf16 extractor: hi + lo bf16 MFMA fragments for conv3x3_bf16_first2_kernel
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // dodt_extractor_forward_timed
};

enum Buf { X0, C1A, CAT1, P1, C2A, CAT2, P2, C3A, C3B, CAT3, P3, C4A, C4B, C4C, F3, F2, F1, NBUF };

}  // namespace

struct dodt_extractor {
    dodt_ctx* ctx = nullptr;
    int in_h = 0, in_w = 0, in_c = 0, pad_top = 0, batch = 0;
    int kind = DODT_EXTRACTOR_VGG_PYR;
    int out_h = 0, out_w = 0, out_c = 32;   // the returned feature map
    bool bf16 = false;  // conv path on bf16 MFMA (fp32 accumulate, fp32 BN/ReLU, bf16 maps)
    int parts = 1;      // 2: split mode (hi + lo bf16 maps and weights, three MFMAs per term)
    int H = 0, W = 0;  // padded input size
    Buffer buf[NBUF];
    std::vector<Layer> layers;
    float* d_bneck_w = nullptr;
    int* d_counters = nullptr;  // two work-item counters per layer, zeroed every forward
    float* d_zeros = nullptr;   // a zero page (Winograd kernel: out-of-image pixels)
    float bneck_scale = 1.0f, bneck_shift = 0.0f;
    bool bneck_loaded = false;
    double flops = 0.0;
    bool timed = false;   // this forward records an event pair around every layer
    int first2_variant = -1;   // >= 0: conv1_1 runs folded into conv1_2's launch (bf16 conv path; DODT_CONV_BF16_FIRST2=0: not)
    float* own_x0 = nullptr;   // the extractor's own input buffer while dodt_extractor_set_input points X0 elsewhere
};

namespace {

int find_layer(dodt_extractor* ex, const char* name) {
    for (size_t i = 0; i < ex->layers.size(); ++i)
        if (ex->layers[i].name == name) return (int)i;
    return -1;
}

int buffer_for_layer_output(const Layer& l) { return l.dst; }

int run_launch(dodt_extractor* ex, const Layer& l, const Launch& ln, int which,
               float* override_out, int out_y0, int out_h, float* bneck_out, int pool_dst, const Layer* folded = nullptr) {
    // folded: conv1_1, computed inside this launch of conv1_2 (same items and weights as conv1_2's streaming kernel)
    const KernelVariant& v = variants()[folded ? ex->first2_variant : ln.variant];
    const Buffer& src = ex->buf[folded ? folded->src : l.src];
    const Buffer& dst = ex->buf[l.dst];
    ConvArgs a;
    a.in = src.ptr;
    a.out = override_out ? override_out : dst.ptr;
    a.w = ln.d_w ? ln.d_w : l.main.d_w;
    a.scale = l.d_scale;
    a.shift = l.d_shift;
    a.H = l.H;
    a.W = l.W;
    a.Cin = l.Cin;
    a.Cout = l.Cout;
    a.in_ld = src.C;
    a.in_coff = folded ? folded->src_coff : l.src_coff;
    a.out_ld = dst.C;
    a.out_coff = l.dst_coff;
    a.in_frame_stride = (long long)src.frame_floats();
    a.out_frame_stride = override_out ? (long long)out_h * dst.W * dst.C : (long long)dst.frame_floats();
    a.tiles_x = dodt::ceil_div(l.W, v.TW);
    a.tiles_y = dodt::ceil_div(l.H, v.TH);
    a.relu = 1;
    a.out_y0 = out_y0;
    a.out_nhwc = override_out ? 1 : 0;
    {
        static const int dbg = getenv("DODT_CONV_DEBUG") ? atoi(getenv("DODT_CONV_DEBUG")) : 0;
        a.debug = dbg;
        // DODT_CONV_STAMP_LAYER=<name>: in-kernel step stamps of that layer's launch (diagnostic)
        static const char* stamp_layer = getenv("DODT_CONV_STAMP_LAYER");
        if (stamp_layer && l.name == stamp_layer) a.debug |= 32;     // (+ DODT_CONV_DEBUG=64: the streaming kernel's producer wave)
    }
    a.counter = ex->d_counters + 2 * (&l - ex->layers.data()) + which;
    a.counter_base = ex->d_counters + 64;
    // (eight counters, 64 bytes apart, per layer: words 2048.. of the block)
    a.xcd_counters = variant_xcd_queue(v) ? ex->d_counters + 2048 + 128 * (&l - ex->layers.data()) : nullptr;
    a.items = ln.d_items;
    a.n_items = ln.n_items;
    a.in_part_stride = (long long)src.frame_floats() * ex->batch;
    a.out_part_stride = (long long)dst.frame_floats() * ex->batch;
    a.pool_part_stride = pool_dst >= 0 ? (long long)ex->buf[pool_dst].frame_floats() * ex->batch : 0;
    a.pool_out = pool_dst >= 0 ? ex->buf[pool_dst].ptr : nullptr;
    a.pool_frame_stride = pool_dst >= 0 ? (long long)ex->buf[pool_dst].frame_floats() : 0;
    a.bneck_w = bneck_out ? ex->d_bneck_w : nullptr;
    a.bneck_out = bneck_out;
    a.bneck_scale = ex->bneck_scale;
    a.bneck_shift = ex->bneck_shift;
    a.bneck_frame_stride = (long long)out_h * dst.W;
    a.zeros = ex->d_zeros;
    a.first_w = folded ? folded->d_first_w : nullptr;
    a.first_scale = folded ? folded->d_scale : nullptr;
    a.first_shift = folded ? folded->d_shift : nullptr;
    // persistent workgroups: as many as stay resident, each walks items with that stride
    int grid_x = a.n_items;
    {
        static const int bpc_env = getenv("DODT_CONV_BPC") ? atoi(getenv("DODT_CONV_BPC")) : 0;
        // (the first-layer kernel is persistent since round 4: three workgroups per CU by its registers)
        static const int small_bpc = getenv("DODT_CONV_SMALL_BPC") ? atoi(getenv("DODT_CONV_SMALL_BPC")) : 3;
        const int vb = v.small_cin ? small_bpc : v.blocks_per_cu;
        const int bpc = (bpc_env > 0 && bpc_env < vb) ? bpc_env : vb;
        const int resident = ex->ctx->num_cus * bpc;
        if (grid_x > resident) grid_x = resident;
    }
    dim3 grid(grid_x, 1);
    v.launch(a, grid, ex->ctx->stream);
    DODT_LAUNCH_CHECK();
    if (a.debug & 32) {
        int h[74];
        (void)hipStreamSynchronize(ex->ctx->stream);
        (void)hipMemcpy(h, ex->d_counters + 64 + 32, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[dodt] %s step stamps (cycles): copies | transform | MFMAs | wait copies | barrier | step\n",
                l.name.c_str());
        for (int k = 0; k < 12; ++k)
            fprintf(stderr, "[dodt]   step %2d: %6d %6d %6d %6d %6d   next-top %6d\n", k, h[k * 6 + 1] - h[k * 6],
                    h[k * 6 + 2] - h[k * 6 + 1], h[k * 6 + 3] - h[k * 6 + 2], h[k * 6 + 4] - h[k * 6 + 3],
                    h[k * 6 + 5] - h[k * 6 + 4], k < 11 ? h[(k + 1) * 6] - h[k * 6] : 0);
        if (h[73] != h[72])
            fprintf(stderr, "[dodt]   steps 0..11: %d stamp ticks in %d ticks of the 100 MHz clock (%.0f MHz)\n",
                    h[66] - h[0], h[73] - h[72], 100.0 * (h[66] - h[0]) / (h[73] - h[72]));
    }
    if ((a.debug & 128) && v.dma) {   // diagnostic: when the workgroups of this launch started and ended
        std::vector<int> h(4 * grid_x);
        (void)hipStreamSynchronize(ex->ctx->stream);
        (void)hipMemcpy(h.data(), ex->d_counters + 64 + 256, h.size() * sizeof(int), hipMemcpyDeviceToHost);
        int t0 = h[0], s_max = h[0], f_max = h[1], e_min = h[2], e_max = h[2], items[8] = {};
        double first_sum = 0, run_sum = 0;
        for (int b = 0; b < (int)grid_x; ++b) {
            t0 = std::min(t0, h[4 * b]); s_max = std::max(s_max, h[4 * b]);
            f_max = std::max(f_max, h[4 * b + 1]);
            e_min = std::min(e_min, h[4 * b + 2]); e_max = std::max(e_max, h[4 * b + 2]);
            items[std::min(h[4 * b + 3], 7)]++;
            first_sum += (h[4 * b + 1] - h[4 * b]) / 100.0;
            run_sum += (h[4 * b + 2] - h[4 * b + 1]) / 100.0 / std::max(h[4 * b + 3], 1);
        }
        fprintf(stderr, "[dodt] %-16s %4d wgs, %5d items x %d chunks: last start +%.2f us, last first-step +%.2f, first end +%.2f, "
                "last end +%.2f; mean start->first step %.2f us, mean per item %.2f us; wgs by item count:",
                l.name.c_str(), (int)grid_x, a.n_items, a.Cin / 16, (s_max - t0) / 100.0, (f_max - t0) / 100.0,
                (e_min - t0) / 100.0, (e_max - t0) / 100.0, first_sum / grid_x, run_sum / grid_x);
        for (int k = 0; k < 8; ++k) if (items[k]) fprintf(stderr, " %d:%d", k, items[k]);
        int mt[2] = {0, 0};
        (void)hipMemcpy(mt, ex->d_counters + 64 + 250, sizeof(mt), hipMemcpyDeviceToHost);
        fprintf(stderr, "; shader clock of workgroup 1: %.2f GHz\n", (mt[1] - mt[0]) / ((h[4 + 2] - h[4 + 0]) * 10.0));
    }
    if (a.debug & 8) {   // diagnostic: print the in-kernel clock of this launch
        unsigned long long h[2] = {0, 0};
        (void)hipStreamSynchronize(ex->ctx->stream);
        (void)hipMemcpy(h, ex->d_counters + 96, sizeof(h), hipMemcpyDeviceToHost);
        if (h[1])
            fprintf(stderr, "[dodt] %-16s wg0: %.1f us, shader clock %.3f GHz\n", l.name.c_str(),
                    h[1] / 100.0, (double)h[0] / h[1] * 0.1);
    }
    return DODT_OK;
}

// a variant's waves hold both rows of every 2x2 pooling window (conv_kernels.h kCanPool)
bool variant_can_pool(const KernelVariant& v) {
    if (v.wino) return true;   // a lane's 2x2 outputs are one pooling window
    const int rows_per_mt = 32 / v.TW, mt = v.MTB / v.WM;
    return !v.deconv && !v.small_cin && (rows_per_mt >= 2 || mt % 2 == 0) && v.TH % 2 == 0;
}

bool layer_can_pool(const Layer& l) {
    return variant_can_pool(variants()[l.main.variant]) &&
           (l.tail.n_items == 0 || variant_can_pool(variants()[l.tail.variant]));
}

// pool_dst >= 0: the layer also writes its 2x2 max pool into that buffer
int run_layer(dodt_extractor* ex, const Layer& l, float* override_out, int out_y0, int out_h,
              float* bneck_out = nullptr, int pool_dst = -1, const Layer* folded = nullptr) {
    if (ex->timed && folded) {      // (the folded layer has no time of its own)
        DODT_HIP_CHECK(hipEventRecord(folded->ev0, ex->ctx->stream));
        DODT_HIP_CHECK(hipEventRecord(folded->ev1, ex->ctx->stream));
    }
    if (ex->timed) DODT_HIP_CHECK(hipEventRecord(l.ev0, ex->ctx->stream));
    int rc = run_launch(ex, l, l.main, 0, override_out, out_y0, out_h, bneck_out, pool_dst, folded);
    if (rc == DODT_OK && l.tail.n_items > 0)
        rc = run_launch(ex, l, l.tail, 1, override_out, out_y0, out_h, bneck_out, pool_dst);
    if (rc == DODT_OK && ex->timed) DODT_HIP_CHECK(hipEventRecord(l.ev1, ex->ctx->stream));
    return rc;
}

// the __global__ function a variant launches (what rocprofv3 --kernel-trace lists)
const char* kernel_name(const KernelVariant& v) {
    if (v.wino) return v.wino_m == 4 ? "wino43_f32_kernel" : "wino3x3_f32_kernel";
    if (v.deconv_dma) return "deconv3x3_dma_kernel";
    if (v.first2) return "conv3x3_bf16_first2_kernel";
    if (v.stream_nch) return "conv3x3_bf16_stream_kernel";
    if (v.dma) return "conv3x3_bf16_dma_kernel";
    if (v.small_cin) return "conv3x3_small_cin_kernel";
    return "conv3x3_mfma_kernel";
}

double layer_direct_flops(const dodt_extractor* ex, const Layer& l) {
    return 2.0 * l.H * l.W * (double)l.Cout * 9.0 * l.real_cin * ex->batch;
}

// FLOPs the matrix pipe executes for a layer: the Winograd kernels multiply 36 times per 4x4
// outputs and channel pair (F(4x4,3x3)) or 16 times per 2x2 (F(2x2,3x3)) where the direct form
// needs 144 / 36; split mode issues three bf16 MFMAs per product term
double layer_executed_flops(const dodt_extractor* ex, const Layer& l) {
    const KernelVariant& kv = variants()[l.main.variant];
    const double direct = layer_direct_flops(ex, l);
    if (kv.wino) return direct * (kv.wino_m == 4 ? 36.0 / 144.0 : 16.0 / 36.0);
    return kv.parts == 2 ? 3.0 * direct : direct;
}

// algorithmic HBM bytes of a layer: input map and weights read once, output map (and its pooled
// copy) written once; split mode keeps two bf16 maps (hi + lo) and two bf16 weight sets per tensor
double layer_bytes(const dodt_extractor* ex, const Layer& l) {
    const Buffer& src = ex->buf[l.src];
    const Buffer& dst = ex->buf[l.dst];
    const double in_e = src.bf16 ? 2.0 * src.parts : 4.0;
    const double out_e = dst.bf16 ? 2.0 * dst.parts : 4.0;
    const double w_e = (ex->bf16 && &l != &ex->layers[0]) ? 2.0 * ex->parts : 4.0;
    const double oh = l.deconv ? 2.0 * l.H : l.H, ow = l.deconv ? 2.0 * l.W : l.W;
    double b = ex->batch * ((double)l.H * l.W * l.real_cin * in_e + oh * ow * l.Cout * out_e) +
               9.0 * l.real_cin * l.Cout * w_e;
    // conv1_1 folded into conv1_2's launch: conv1_1's map is neither written nor read
    if (ex->first2_variant >= 0 && &l == &ex->layers[0]) b -= ex->batch * oh * ow * l.Cout * out_e;
    if (ex->first2_variant >= 0 && &l == &ex->layers[1]) b -= ex->batch * (double)l.H * l.W * l.real_cin * in_e;
    if (l.name == "conv1_2" || l.name == "conv2_2" || l.name == "conv3_3")
        b += ex->batch * std::floor(oh / 2) * std::floor(ow / 2) * l.Cout * out_e;
    return b;
}

// Work items of a layer.  The main launch walks big tiles; when their count leaves the
// last round of the persistent grid mostly idle, the surplus tiles are handed to a tail
// launch that cuts each of them into smaller tiles (same TW; fewer rows and/or channels),
// so the tail costs a fraction of a round.  Every output element is still summed by one
// workgroup in one fixed order: results do not depend on how a layer is cut.
void plan_layer(Layer& l, int batch, int num_cus, bool allow_tail, std::vector<int4>& main_items,
                std::vector<int4>& tail_items) {
    const auto& vs = variants();
    const KernelVariant& v = vs[l.variant];
    const int tx = dodt::ceil_div(l.W, v.TW), ty = dodt::ceil_div(l.H, v.TH), nt = l.Cout / v.BN;
    main_items.clear();
    tail_items.clear();
    if (variant_xcd_queue(v)) {
        // XCD-grouped queue (conv_bf16_dma.h): the channel tiles of a pixel tile next to each other
        for (int f = 0; f < batch; ++f)
            for (int y = 0; y < ty; ++y)
                for (int x = 0; x < tx; ++x)
                    for (int n = 0; n < nt; ++n) main_items.push_back(make_int4(f, n, y * v.TH, x * v.TW));
    } else {
        for (int f = 0; f < batch; ++f)
            for (int n = 0; n < nt; ++n)
                for (int y = 0; y < ty; ++y)
                    for (int x = 0; x < tx; ++x) main_items.push_back(make_int4(f, n, y * v.TH, x * v.TW));
    }
    l.main.variant = l.variant;
    l.tail.variant = -1;
    if (v.small_cin || v.wino || v.dma || v.deconv_dma) return;
    static const bool no_tail = getenv("DODT_CONV_NO_TAIL") != nullptr;
    const int n = (int)main_items.size();
    const int G = num_cus * v.blocks_per_cu;
    const int surplus = n % G;
    if (no_tail || !allow_tail || n < G || surplus == 0) return;
    // companion: same TW and kind, dividing the big tile, as small as possible
    int best = -1, best_units = 1 << 30;
    const int big_units = (v.TW * v.TH / 32) * (v.BN / 32);
    for (size_t i = 0; i < vs.size(); ++i) {
        const KernelVariant& c = vs[i];
        if (c.small_cin || c.deconv != v.deconv || c.TW != v.TW || v.TH % c.TH != 0 ||
            v.BN % c.BN != 0)
            continue;
        const int units = (c.TW * c.TH / 32) * (c.BN / 32);
        if (units < big_units && units < best_units) { best = (int)i; best_units = units; }
    }
    if (best < 0) return;
    const KernelVariant& c = vs[best];
    const int ry = v.TH / c.TH, rn = v.BN / c.BN;
    std::vector<int4> small;
    for (int k = n - surplus; k < n; ++k) {
        const int4 b = main_items[k];
        for (int jn = 0; jn < rn; ++jn)
            for (int jy = 0; jy < ry; ++jy)
                if (b.z + jy * c.TH < l.H)
                    small.push_back(make_int4(b.x, b.y * rn + jn, b.z + jy * c.TH, b.w));
    }
    // cost of the tail in rounds of the main launch (one round = blocks_per_cu big tiles per
    // CU).  Measured: small tiles sharing a CU run at the big tiles' rate, a lone small tile
    // per CU at ~0.78 of it.
    const double share = (double)best_units / (v.blocks_per_cu * big_units);
    double cost_small;
    if ((int)small.size() <= num_cus)
        cost_small = share / 0.78;
    else
        cost_small = dodt::ceil_div((int)small.size(), num_cus * c.blocks_per_cu) *
                     c.blocks_per_cu * share;
    cost_small += 0.04;   // second launch
    if (cost_small >= 1.0) return;
    main_items.resize(n - surplus);
    tail_items = small;
    l.tail.variant = best;
}

int run_pool(dodt_extractor* ex, int src, int dst) {
    const Buffer& s = ex->buf[src];
    const Buffer& d = ex->buf[dst];
    const int planes = d.C / 8;
    const long long total = (long long)ex->batch * planes * d.H * d.W * 2;
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       ex->ctx->stream, s.ptr, s.H, s.W, planes, (long long)s.frame_floats(),
                       d.ptr, ex->batch);
    DODT_LAUNCH_CHECK();
    return DODT_OK;
}

}  // namespace

extern "C" {

int dodt_extractor_create(dodt_ctx* ctx, int kind, int in_h, int in_w, int in_c, int pad_top,
                          int batch, dodt_extractor** out) {
    DODT_REQUIRE(ctx && out, "dodt_extractor_create: NULL argument");
    const bool split = (kind & DODT_EXTRACTOR_SPLIT) != 0;
    const bool bf16 = (kind & DODT_EXTRACTOR_BF16) != 0 || split;
    // (the bf16 kernels have no quarter-size instantiations: single launches)
    const bool shared_gpu = (kind & DODT_EXTRACTOR_SHARED_GPU) != 0 || bf16;
    kind &= ~(DODT_EXTRACTOR_SHARED_GPU | DODT_EXTRACTOR_BF16 | DODT_EXTRACTOR_SPLIT);
    DODT_REQUIRE(kind == DODT_EXTRACTOR_VGG_PYR || kind == DODT_EXTRACTOR_VGG,
                 "dodt_extractor_create: unknown kind %d", kind);
    const bool plain = kind == DODT_EXTRACTOR_VGG;
    DODT_REQUIRE(in_h > 0 && in_w > 0 && in_c >= 2 && in_c % 2 == 0 && pad_top >= 0 && batch >= 1,
                 "dodt_extractor_create: bad sizes (in_c must be even)");
    const int H = in_h + pad_top, W = in_w;
    if (plain) {
        DODT_REQUIRE(!bf16, "dodt_extractor_create: the plain VGG extractor is fp32 only");
        DODT_REQUIRE(pad_top == 0 && H >= 16 && W >= 16,
                     "dodt_extractor_create: plain VGG takes no top padding and >= 16x16 inputs");
    } else {
        DODT_REQUIRE(H % 8 == 0 && W % 8 == 0,
                     "dodt_extractor_create: padded input %dx%d must be divisible by 8 "
                     "(three 2x2 pools, three stride-2 upconvs)", H, W);
    }
    for (const KernelVariant& v : variants()) DODT_HIP_CHECK(v.prepare());

    dodt_extractor* ex = new dodt_extractor();
    ex->ctx = ctx;
    ex->kind = kind;
    ex->in_h = in_h; ex->in_w = in_w; ex->in_c = in_c; ex->pad_top = pad_top; ex->batch = batch;
    ex->H = H; ex->W = W;
    ex->bf16 = bf16;
    ex->parts = split ? 2 : 1;
    auto setb = [&](int id, int h, int w, int c) { ex->buf[id].H = h; ex->buf[id].W = w; ex->buf[id].C = c; };
    // VALID 2x2 pools floor odd sizes (plain VGG: 175 -> 87, 795 -> 397 -> 198)
    const int H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, H8 = H4 / 2, W8 = W4 / 2;
    setb(X0, H, W, in_c);
    setb(C1A, H, W, 32); setb(CAT1, H, W, plain ? 32 : 64); setb(P1, H2, W2, 32);
    setb(C2A, H2, W2, 64); setb(CAT2, H2, W2, plain ? 64 : 128); setb(P2, H4, W4, 64);
    setb(C3A, H4, W4, 128); setb(C3B, H4, W4, 128); setb(CAT3, H4, W4, plain ? 128 : 256);
    setb(P3, H8, W8, 128);
    setb(C4A, H8, W8, 256); setb(C4B, H8, W8, 256); setb(C4C, H8, W8, 256);
    if (!plain) { setb(F3, H4, W4, 64); setb(F2, H2, W2, 32); setb(F1, H, W, 32); }
    if (plain) {
        // bev_vgg.py:102-112: resize_bilinear to input_pixel_size / 8 * upsampling_multiplier (4),
        // a float pair that TF casts to int32 (350 x 400 for 700 x 800, 240 x 795 for 480 x 1590)
        ex->out_h = (int)((double)in_h / 8.0 * 4.0);
        ex->out_w = (int)((double)in_w / 8.0 * 4.0);
        ex->out_c = 256;
    } else {
        ex->out_h = in_h; ex->out_w = in_w; ex->out_c = 32;
    }
    for (int i = 0; i < NBUF; ++i) {
        ex->buf[i].bf16 = bf16 && i != X0 && i != F1;
        ex->buf[i].parts = ex->buf[i].bf16 ? ex->parts : 1;
    }
    for (int i = 0; i < NBUF; ++i) {
        if (i == F1 || ex->buf[i].H == 0) continue;  // F1: written into the caller's buffer
        const size_t bytes = ex->buf[i].frame_floats() * batch * ex->buf[i].parts * sizeof(float);
        hipError_t e = hipMalloc(&ex->buf[i].ptr, bytes);
        if (e != hipSuccess) {
            dodt::set_error("dodt_extractor_create: hipMalloc(%zu) failed: %s", bytes,
                            hipGetErrorString(e));
            dodt_extractor_destroy(ex);
            return DODT_ERR_HIP;
        }
    }
    DODT_HIP_CHECK(hipMalloc(&ex->d_counters, 4096 * sizeof(int)));
    DODT_HIP_CHECK(hipMemsetAsync(ex->d_counters, 0, 4096 * sizeof(int), ctx->stream));
    DODT_HIP_CHECK(hipMalloc(&ex->d_zeros, 256));
    DODT_HIP_CHECK(hipMemsetAsync(ex->d_zeros, 0, 256, ctx->stream));
    // the pad rows of X0 stay zero for the life of the extractor
    DODT_HIP_CHECK(hipMemsetAsync(ex->buf[X0].ptr, 0,
                                  ex->buf[X0].frame_floats() * batch * sizeof(float), ctx->stream));

    auto add = [&](const char* name, bool deconv, int h, int w, int cin, int cout, int src,
                   int src_coff, int dst, int dst_coff) {
        Layer l;
        l.name = name; l.deconv = deconv; l.H = h; l.W = w; l.Cin = cin; l.Cout = cout;
        l.src = src; l.src_coff = src_coff; l.dst = dst; l.dst_coff = dst_coff;
        l.variant = pick_variant(deconv, h, w, cin, cout, bf16, ex->parts, batch, ctx->num_cus);
        l.real_cin = cin;
        ex->layers.push_back(l);
    };
    add("conv1_1", false, H, W, in_c, 32, X0, 0, C1A, 0);
    add("conv1_2", false, H, W, 32, 32, C1A, 0, CAT1, 0);
    add("conv2_1", false, H2, W2, 32, 64, P1, 0, C2A, 0);
    add("conv2_2", false, H2, W2, 64, 64, C2A, 0, CAT2, 0);
    add("conv3_1", false, H4, W4, 64, 128, P2, 0, C3A, 0);
    add("conv3_2", false, H4, W4, 128, 128, C3A, 0, C3B, 0);
    add("conv3_3", false, H4, W4, 128, 128, C3B, 0, CAT3, 0);
    add("conv4_1", false, H8, W8, 128, 256, P3, 0, C4A, 0);
    add("conv4_2", false, H8, W8, 256, 256, C4A, 0, C4B, 0);
    add("conv4_3", false, H8, W8, 256, 256, C4B, 0, C4C, 0);
    if (!plain) {
        add("upconv3", true, H8, W8, 256, 128, C4C, 0, CAT3, 128);
        add("pyramid_fusion3", false, H4, W4, 256, 64, CAT3, 0, F3, 0);
        add("upconv2", true, H4, W4, 64, 64, F3, 0, CAT2, 64);
        add("pyramid_fusion2", false, H2, W2, 128, 32, CAT2, 0, F2, 0);
        add("upconv1", true, H2, W2, 32, 32, F2, 0, CAT1, 32);
        add("pyramid_fusion1", false, H, W, 64, 32, CAT1, 0, F1, 0);
    }
    for (const Layer& l : ex->layers) {
        if (l.variant < 0) {
            dodt::set_error("dodt_extractor_create: no kernel variant for layer %s (%dx%d %d->%d)",
                            l.name.c_str(), l.H, l.W, l.Cin, l.Cout);
            dodt_extractor_destroy(ex);
            return DODT_ERR_UNSUPPORTED;
        }
    }
    for (Layer& l : ex->layers) {
        std::vector<int4> mi, ti;
        plan_layer(l, batch, ctx->num_cus, !shared_gpu, mi, ti);
        for (auto pr : {std::make_pair(&l.main, &mi), std::make_pair(&l.tail, &ti)}) {
            pr.first->n_items = (int)pr.second->size();
            if (pr.second->empty()) continue;
            DODT_HIP_CHECK(hipMalloc(&pr.first->d_items, pr.second->size() * sizeof(int4)));
            DODT_HIP_CHECK(hipMemcpyAsync(pr.first->d_items, pr.second->data(),
                                          pr.second->size() * sizeof(int4), hipMemcpyHostToDevice,
                                          ctx->stream));
        }
        DODT_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // the vectors go out of scope
    }
    {
        // bf16 conv path: conv1_1 folded into conv1_2's launch when conv1_2 runs on the streaming kernel (same tiles, same
        // weight blocking) and the input rows are whole 16-byte slots (DODT_CONV_BF16_FIRST2=0: two launches)
        static const bool first2 = !(getenv("DODT_CONV_BF16_FIRST2") && atoi(getenv("DODT_CONV_BF16_FIRST2")) == 0);
        const Layer& c11 = ex->layers[0];
        const Layer& c12 = ex->layers[1];
        const auto& vs = variants();
        if (first2 && bf16 && ex->parts == 1 && vs[c12.variant].stream_nch == 2 && c12.tail.n_items == 0 &&
            c11.Cout == 32 && c11.src_coff == 0 && ex->buf[X0].C == c11.Cin && (W * c11.Cin * 4) % 16 == 0 &&
            variant_can_pool(vs[c12.variant]))
            for (size_t i = 0; i < vs.size() && ex->first2_variant < 0; ++i)
                if (vs[i].first2 == c11.Cin && vs[i].TH == vs[c12.variant].TH && vs[i].TW == vs[c12.variant].TW &&
                    vs[i].lds_bytes <= (getenv("DODT_CONV_BF16_STREAM_LDS") ? atoi(getenv("DODT_CONV_BF16_STREAM_LDS")) * 1024 : 1 << 30))
                    ex->first2_variant = (int)i;
    }
    if (getenv("DODT_DEBUG_PLAN")) {
        if (ex->first2_variant >= 0) fprintf(stderr, "[dodt] conv1_1 runs folded into conv1_2's launch\n");
        for (const Layer& l : ex->layers) {
            const KernelVariant& v = variants()[l.variant];
            fprintf(stderr, "[dodt] %-16s %4dx%-4d %3d->%-3d TW%d TH%d BN%d CK%d lds %d items %d",
                    l.name.c_str(), l.H, l.W, l.Cin, l.Cout, v.TW, v.TH, v.BN, v.CK, v.lds_bytes,
                    l.main.n_items);
            if (l.tail.n_items) {
                const KernelVariant& c = variants()[l.tail.variant];
                fprintf(stderr, " + tail TH%d BN%d items %d", c.TH, c.BN, l.tail.n_items);
            }
            fprintf(stderr, "\n");
        }
    }
    *out = ex;
    return DODT_OK;
}

int dodt_extractor_destroy(dodt_extractor* ex) {
    if (!ex) return DODT_OK;
    if (ex->ctx) (void)hipStreamSynchronize(ex->ctx->stream);
    if (ex->own_x0) ex->buf[X0].ptr = ex->own_x0;      // (never free a caller's input buffer)
    for (int i = 0; i < NBUF; ++i)
        if (ex->buf[i].ptr) (void)hipFree(ex->buf[i].ptr);
    for (Layer& l : ex->layers) {
        for (Launch* ln : {&l.main, &l.tail}) {
            if (ln->d_w) (void)hipFree(ln->d_w);
            if (ln->d_items) (void)hipFree(ln->d_items);
        }
        if (l.d_scale) (void)hipFree(l.d_scale);
        if (l.d_shift) (void)hipFree(l.d_shift);
        if (l.d_first_w) (void)hipFree(l.d_first_w);
    }
    if (ex->d_bneck_w) (void)hipFree(ex->d_bneck_w);
    for (Layer& l : ex->layers) {
        if (l.ev0) (void)hipEventDestroy(l.ev0);
        if (l.ev1) (void)hipEventDestroy(l.ev1);
    }
    if (ex->d_counters) (void)hipFree(ex->d_counters);
    if (ex->d_zeros) (void)hipFree(ex->d_zeros);
    delete ex;
    return DODT_OK;
}

int dodt_extractor_set_layer(dodt_extractor* ex, const char* name, const float* w, int kh, int kw,
                             int c_a, int c_b, const float* beta, const float* mean,
                             const float* var) {
    DODT_REQUIRE(ex && name && w && beta && mean && var, "dodt_extractor_set_layer: NULL argument");
    hipStream_t s = ex->ctx->stream;
    if (std::strcmp(name, "bottleneck") == 0) {
        const int fc = ex->out_c;   // 32 (pyramid) or 256 (plain VGG: rpn_model.py:251-267)
        DODT_REQUIRE(kh == 1 && kw == 1 && c_a == fc && c_b == 1,
                     "bottleneck must be (1,1,%d,1), got (%d,%d,%d,%d)", fc, kh, kw, c_a, c_b);
        if (!ex->d_bneck_w) DODT_HIP_CHECK(hipMalloc(&ex->d_bneck_w, fc * sizeof(float)));
        DODT_HIP_CHECK(hipMemcpyAsync(ex->d_bneck_w, w, fc * sizeof(float), hipMemcpyHostToDevice, s));
        DODT_HIP_CHECK(hipStreamSynchronize(s));
        const float inv = 1.0f / std::sqrt(var[0] + 0.001f);
        ex->bneck_scale = inv;
        ex->bneck_shift = beta[0] - mean[0] * inv;
        ex->bneck_loaded = true;
        return DODT_OK;
    }
    const int li = find_layer(ex, name);
    DODT_REQUIRE(li >= 0, "dodt_extractor_set_layer: unknown layer '%s'", name);
    Layer& l = ex->layers[li];
    DODT_REQUIRE(kh == 3 && kw == 3, "layer %s: kernel must be 3x3", name);
    // TF layouts: conv2d (kh,kw,cin,cout); conv2d_transpose (kh,kw,cout,cin)
    const int cin = l.deconv ? c_b : c_a, cout = l.deconv ? c_a : c_b;
    DODT_REQUIRE(cout == l.Cout && cin <= l.Cin && (cin == l.Cin || li == 0),
                 "layer %s: expected %d->%d channels, got %d->%d", name, l.Cin, l.Cout, cin, cout);
    for (Launch* ln : {&l.main, &l.tail}) {
    if (ln->variant < 0) continue;
    if (ln == &l.tail && variants()[l.tail.variant].BN == variants()[l.main.variant].BN) {
        ln->d_w = nullptr;   // same blocking: share the main launch's copy (set below)
        continue;
    }
    const KernelVariant& v = variants()[ln->variant];
    const int nchunks = l.Cin / v.CK;
    if (v.deconv_dma) {
        // transposed conv, TF layout (kh, kw, Cout, Cin): blocked [n-tile][chunk][tap][g = c / 2][t]
        // [channel block][c & 1]: a lane (t, g) reads 16 bytes = its channel pair for both blocks
        // (bf16: the same bytes hold 16 channels per chunk, [g = c / 4] ... [c & 3] as bf16)
        const int ncb = v.BN / 16;
        const size_t chunk_floats = (size_t)(9 * 8 * v.BN + 255) / 256 * 256;     // whole 1 KB pieces
        std::vector<float> u((size_t)(l.Cout / v.BN) * nchunks * chunk_floats, 0.0f);
        uint16_t* u16 = reinterpret_cast<uint16_t*>(u.data());
        for (int tap = 0; tap < 9; ++tap)
            for (int ci = 0; ci < cin; ++ci)
                for (int co = 0; co < cout; ++co) {
                    const float val = w[((size_t)tap * cout + co) * cin + ci];
                    const int nt = co / v.BN, n = co % v.BN, cb = n / 16, t = n % 16;
                    if (v.bf16) {
                        const int ch = ci / 16, c = ci % 16;
                        u16[(((size_t)nt * nchunks + ch) * chunk_floats) * 2 +
                            ((((size_t)tap * 4 + c / 4) * 16 + t) * ncb + cb) * 4 + (c & 3)] = dodt::float_to_bf16(val);
                    } else {
                        const int ch = ci / 8, c = ci % 8;
                        u[((size_t)nt * nchunks + ch) * chunk_floats +
                          ((((size_t)tap * 4 + c / 2) * 16 + t) * ncb + cb) * 2 + (c & 1)] = val;
                    }
                }
        if (!ln->d_w) DODT_HIP_CHECK(hipMalloc(&ln->d_w, u.size() * sizeof(float)));
        DODT_HIP_CHECK(hipMemcpyAsync(ln->d_w, u.data(), u.size() * sizeof(float),
                                      hipMemcpyHostToDevice, s));
        DODT_HIP_CHECK(hipStreamSynchronize(s));
        continue;
    }
    if (v.wino && v.wino_m == 4) {
        // F(4x4,3x3) filter transform U = G g G^T (6x6 points; float64 on the host, rounded once),
        // blocked [n-tile][chunk][xi / 2][g = c / 2][cb][t][xi & 1][c & 1]: a lane (t, g) of channel
        // block cb reads 16 bytes = its channel pair for two points
        // points 0, +-2/3, +-3/2, infinity (wino43_kernel.h): G[j][k] = p_j^k / prod_{l != j} (p_j - p_l)
        static const std::array<std::array<double, 3>, 6> G = [] {
            const double p[5] = {0.0, 2.0 / 3.0, -2.0 / 3.0, 1.5, -1.5};
            std::array<std::array<double, 3>, 6> g{};
            for (int j = 0; j < 5; ++j) {
                double n = 1.0;
                for (int l = 0; l < 5; ++l)
                    if (l != j) n *= p[j] - p[l];
                g[j] = {1.0 / n, p[j] / n, p[j] * p[j] / n};
            }
            g[5] = {0.0, 0.0, 1.0};
            return g;
        }();
        std::vector<float> u((size_t)36 * l.Cin * l.Cout, 0.0f);
        for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co) {
                double gk[3][3], tmp[6][3];
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx)
                        gk[ky][kx] = w[((size_t)(ky * 3 + kx) * cin + ci) * cout + co];
                for (int i = 0; i < 6; ++i)
                    for (int kx = 0; kx < 3; ++kx)
                        tmp[i][kx] = G[i][0] * gk[0][kx] + G[i][1] * gk[1][kx] + G[i][2] * gk[2][kx];
                const int nt = co / v.BN, n = co % v.BN, ch = ci / 8, c = ci % 8;
                const int cb = n / 16, t = n % 16;
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) {
                        const double val = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                        const int xi = i * 6 + j;
                        u[(((((((size_t)nt * nchunks + ch) * 18 + xi / 2) * 4 + c / 2) * 2 + cb) * 16 + t) * 2 +
                           (xi & 1)) * 2 + (c & 1)] = (float)val;
                    }
            }
        if (!ln->d_w) DODT_HIP_CHECK(hipMalloc(&ln->d_w, u.size() * sizeof(float)));
        DODT_HIP_CHECK(hipMemcpyAsync(ln->d_w, u.data(), u.size() * sizeof(float),
                                      hipMemcpyHostToDevice, s));
        DODT_HIP_CHECK(hipStreamSynchronize(s));
        continue;
    }
    if (v.wino) {
        // Winograd filter transform U = G g G^T (float64 on the host, rounded once to fp32),
        // G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]]; blocked like the direct kernel's
        // weights with the 16 points in place of the 9 taps: [n-tile][chunk][xi][h][n][4]
        static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        std::vector<float> u((size_t)16 * l.Cin * l.Cout, 0.0f);
        for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co) {
                double gk[3][3], tmp[4][3];
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx)
                        gk[ky][kx] = w[((size_t)(ky * 3 + kx) * cin + ci) * cout + co];
                for (int i = 0; i < 4; ++i)
                    for (int kx = 0; kx < 3; ++kx)
                        tmp[i][kx] = G[i][0] * gk[0][kx] + G[i][1] * gk[1][kx] + G[i][2] * gk[2][kx];
                // [n-tile][chunk][xi][g = c/2][cb pair][t][cb & 1][k = c%2]: a lane (t, g) of the
                // kernel reads 16 bytes = its (2g, 2g+1) channel pair for two 16-channel blocks
                const int nt = co / v.BN, n = co % v.BN, ch = ci / 8, c = ci % 8;
                const int cb = n / 16, t = n % 16, cbp = v.BN / 32;
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        const double val = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                        u[((((((size_t)nt * nchunks + ch) * 16 + (i * 4 + j)) * 4 + c / 2) * cbp + cb / 2) * 16 + t) * 4 +
                          (cb % 2) * 2 + c % 2] = (float)val;
                    }
            }
        if (!ln->d_w) DODT_HIP_CHECK(hipMalloc(&ln->d_w, u.size() * sizeof(float)));
        DODT_HIP_CHECK(hipMemcpyAsync(ln->d_w, u.data(), u.size() * sizeof(float),
                                      hipMemcpyHostToDevice, s));
        DODT_HIP_CHECK(hipStreamSynchronize(s));
        continue;
    }
    // fp32 kernels: floats; bf16 MFMA kernels: bf16 pairs packed in the same array (half of it)
    std::vector<float> blocked((size_t)9 * l.Cin * l.Cout, 0.0f);
    uint16_t* blocked16 = reinterpret_cast<uint16_t*>(blocked.data());
    const bool w16 = v.bf16 && !v.small_cin;
    for (int tap = 0; tap < 9; ++tap)
        for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co) {
                const float val = l.deconv ? w[((size_t)tap * cout + co) * cin + ci]
                                           : w[((size_t)tap * cin + ci) * cout + co];
                int n = co % v.BN;
                if (v.bf16) {
                    // MFMA row that delivers channel co (conv_kernels.h group_channel<PERM>):
                    // channel 16a + 8lh + 4b + k  <-  row 8(2a + b) + 4lh + k
                    const int c32 = co % 32, a2 = c32 >> 4, lh = (c32 >> 3) & 1, b2 = (c32 >> 2) & 1;
                    n = (n / 32) * 32 + 8 * (2 * a2 + b2) + 4 * lh + (c32 & 3);
                }
                const int nt = co / v.BN, ch = ci / v.CK, c = ci % v.CK;
                if (v.small_cin) {   // [tap][c][n]
                    blocked[((size_t)tap * v.CK + c) * v.BN + n] = val;
                } else if (!w16) {   // [n_tile][chunk][tap][h = c/4][n][s = c%4]
                    blocked[(((((size_t)nt * nchunks + ch) * 9 + tap) * 2 + c / 4) * v.BN + n) * 4 +
                            c % 4] = val;
                } else {   // [n_tile][chunk16][part][tap][h = c/8][n][j = c%8] bf16
                    const uint16_t hi = dodt::float_to_bf16(val);
                    const size_t base = ((size_t)nt * nchunks + ch) * v.parts;
                    const size_t in = ((size_t)(tap * 2 + c / 8) * v.BN + n) * 8 + c % 8;
                    blocked16[(base + 0) * 9 * 2 * v.BN * 8 + in] = hi;
                    if (v.parts == 2)    // lo = bf16(w - hi): w = hi + lo to 16 mantissa bits
                        blocked16[(base + 1) * 9 * 2 * v.BN * 8 + in] =
                            dodt::float_to_bf16(val - dodt::bf16_to_float(hi));
                }
            }
    if (!ln->d_w) DODT_HIP_CHECK(hipMalloc(&ln->d_w, blocked.size() * sizeof(float)));
    DODT_HIP_CHECK(hipMemcpyAsync(ln->d_w, blocked.data(), blocked.size() * sizeof(float),
                                  hipMemcpyHostToDevice, s));
    DODT_HIP_CHECK(hipStreamSynchronize(s));
    }
    if (li == 0 && ex->bf16 && ex->parts == 1 && l.Cout == 32 && (l.Cin == 6 || l.Cin == 4)) {
        // conv1_1 for conv3x3_bf16_first2_kernel: A fragments [K = 16 step][hi, lo][lane half][32 MFMA rows][8 bf16];
        // a lane half's eight K slots are one tap's six channels + two zeros (Cin 6: taps 2 s + lh) or two taps' four
        // channels (Cin 4: taps 4 s + 2 lh, + 1); w = hi + lo to 16 mantissa bits
        const int steps = l.Cin == 6 ? 5 : 3;
        std::vector<uint16_t> frag((size_t)steps * 2 * 2 * 32 * 8, 0);
        for (int st = 0; st < steps; ++st)
            for (int lh = 0; lh < 2; ++lh)
                for (int row = 0; row < 32; ++row) {
                    // channel of MFMA row 8 (2 a + b) + 4 lh' + k: 16 a + 8 lh' + 4 b + k (group_channel<true>)
                    const int g = row >> 3, lho = (row >> 2) & 1, co = 16 * (g >> 1) + 8 * lho + 4 * (g & 1) + (row & 3);
                    for (int j = 0; j < 8; ++j) {
                        const int tap = l.Cin == 6 ? 2 * st + lh : 4 * st + 2 * lh + (j >> 2);
                        const int ci = l.Cin == 6 ? j : (j & 3);
                        if (tap >= 9 || ci >= cin) continue;
                        const float val = w[((size_t)tap * cin + ci) * cout + co];
                        const uint16_t hi = dodt::float_to_bf16(val);
                        const size_t at = ((((size_t)st * 2 + 0) * 2 + lh) * 32 + row) * 8 + j;
                        frag[at] = hi;
                        frag[at + 2 * 32 * 8] = dodt::float_to_bf16(val - dodt::bf16_to_float(hi));
                    }
                }
        if (!l.d_first_w) DODT_HIP_CHECK(hipMalloc(&l.d_first_w, frag.size() * sizeof(uint16_t)));
        DODT_HIP_CHECK(hipMemcpyAsync(l.d_first_w, frag.data(), frag.size() * sizeof(uint16_t), hipMemcpyHostToDevice, s));
        DODT_HIP_CHECK(hipStreamSynchronize(s));
    }
    std::vector<float> scale(l.Cout), shift(l.Cout);
    for (int co = 0; co < l.Cout; ++co) {
        const float inv = 1.0f / std::sqrt(var[co] + 0.001f);
        scale[co] = inv;
        shift[co] = beta[co] - mean[co] * inv;
    }
    if (!l.d_scale) DODT_HIP_CHECK(hipMalloc(&l.d_scale, l.Cout * sizeof(float)));
    if (!l.d_shift) DODT_HIP_CHECK(hipMalloc(&l.d_shift, l.Cout * sizeof(float)));
    DODT_HIP_CHECK(hipMemcpyAsync(l.d_scale, scale.data(), l.Cout * sizeof(float),
                                  hipMemcpyHostToDevice, s));
    DODT_HIP_CHECK(hipMemcpyAsync(l.d_shift, shift.data(), l.Cout * sizeof(float),
                                  hipMemcpyHostToDevice, s));
    DODT_HIP_CHECK(hipStreamSynchronize(s));
    l.loaded = true;
    l.real_cin = cin;
    return DODT_OK;
}

int dodt_extractor_input(dodt_extractor* ex, float** d_ptr, long long* frame_stride_floats) {
    DODT_REQUIRE(ex && d_ptr && frame_stride_floats, "dodt_extractor_input: NULL argument");
    const Buffer& x0 = ex->buf[X0];
    *d_ptr = x0.ptr + (size_t)ex->pad_top * x0.W * x0.C;
    *frame_stride_floats = (long long)x0.frame_floats();
    return DODT_OK;
}

int dodt_extractor_forward(dodt_extractor* ex, const float* d_in, float* d_feat_out,
                           float* d_bottleneck_out) {
    DODT_REQUIRE(ex && d_feat_out, "dodt_extractor_forward: NULL argument");
    for (const Layer& l : ex->layers)
        DODT_REQUIRE(l.loaded, "dodt_extractor_forward: weights of layer %s not set",
                     l.name.c_str());
    DODT_REQUIRE(!d_bottleneck_out || ex->bneck_loaded,
                 "dodt_extractor_forward: bottleneck weights not set");
    hipStream_t s = ex->ctx->stream;
    const Buffer& x0 = ex->buf[X0];
    float* own_in = x0.ptr + (size_t)ex->pad_top * x0.W * x0.C;
    if (d_in && d_in != own_in) {
        const long long n4 = (long long)ex->in_h * ex->in_w * ex->in_c / 4;
        DODT_REQUIRE(((long long)ex->in_h * ex->in_w * ex->in_c) % 4 == 0, "input not float4-sized");
        hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((n4 * ex->batch + 255) / 256)),
                           dim3(256), 0, s, reinterpret_cast<const float4*>(d_in),
                           reinterpret_cast<float4*>(own_in), n4, n4,
                           (long long)x0.frame_floats() / 4, ex->batch);
        DODT_LAUNCH_CHECK();
    }
    // (the whole block: the launches' counters in words 0..63, the XCD-grouped queues' in words 2048..; the
    //  diagnostic words in between are only read right behind the launch that wrote them)
    DODT_HIP_CHECK(hipMemsetAsync(ex->d_counters, 0, 4096 * sizeof(int), s));
    int rc;
    auto L = [&](const char* n) -> const Layer& { return ex->layers[find_layer(ex, n)]; };
#define RUN(name)                                           \
    if ((rc = run_layer(ex, L(name), nullptr, 0, 0))) return rc;
    // a conv that a 2x2 max pool follows pools in its epilogue when its tiling allows
    static const bool no_fuse = getenv("DODT_CONV_NO_POOL_FUSE") != nullptr;
#define RUN_POOLED(name, src, dst)                                                   \
    if (!no_fuse && layer_can_pool(L(name))) {                                       \
        if ((rc = run_layer(ex, L(name), nullptr, 0, 0, nullptr, dst))) return rc;   \
    } else {                                                                         \
        RUN(name);                                                                   \
        if ((rc = run_pool(ex, src, dst))) return rc;                                \
    }
    if (ex->bf16)   // there is no stand-alone pool kernel for bf16 maps
        for (const char* n : {"conv1_2", "conv2_2", "conv3_3"})
            DODT_REQUIRE(!no_fuse && layer_can_pool(L(n)),
                         "bf16 extractor: layer %s cannot pool in its epilogue", n);
    if (ex->first2_variant >= 0) {
        if ((rc = run_layer(ex, L("conv1_2"), nullptr, 0, 0, nullptr, P1, &L("conv1_1")))) return rc;
    } else {
        RUN("conv1_1"); RUN_POOLED("conv1_2", CAT1, P1);
    }
    RUN("conv2_1"); RUN_POOLED("conv2_2", CAT2, P2);
    RUN("conv3_1"); RUN("conv3_2"); RUN_POOLED("conv3_3", CAT3, P3);
#undef RUN_POOLED
    RUN("conv4_1"); RUN("conv4_2"); RUN("conv4_3");
    if (ex->kind == DODT_EXTRACTOR_VGG) {
        // bev_vgg.py:102-112 / img_vgg.py:104-114: 4x bilinear upsampling of conv4_3, and the
        // 256 -> 1 bottleneck of the RPN (rpn_model.py:251-267) on the upsampled map
        const Buffer& c4 = ex->buf[C4C];
        const long long n_pix = (long long)ex->batch * ex->out_h * ex->out_w;
        constexpr int kLpp = 64;   // 256 channels / 4
        hipLaunchKernelGGL(upsample_bilinear_cb8_kernel<kLpp>,
                           dim3((unsigned)((n_pix * kLpp + 255) / 256)), dim3(256), 0, s, c4.ptr,
                           c4.H, c4.W, (long long)c4.frame_floats(), ex->out_h, ex->out_w,
                           ex->batch, d_feat_out, ex->d_bneck_w, ex->bneck_scale, ex->bneck_shift,
                           d_bottleneck_out);
        DODT_LAUNCH_CHECK();
        return DODT_OK;
    }
    RUN("upconv3"); RUN("pyramid_fusion3");
    RUN("upconv2"); RUN("pyramid_fusion2");
    RUN("upconv1");
#undef RUN
    // last layer writes straight into the caller's buffer, pad rows sliced off
    // ... and, fused into its epilogue, the 1x1 bottleneck (dt_rpn_model.py:298-322)
    const Layer& last = L("pyramid_fusion1");
    const bool fuse = d_bottleneck_out && variants()[last.main.variant].BN == 32 &&
                      (last.tail.n_items == 0 || variants()[last.tail.variant].BN == 32);
    if ((rc = run_layer(ex, last, d_feat_out, ex->pad_top, ex->in_h,
                        fuse ? d_bottleneck_out : nullptr)))
        return rc;
    if (d_bottleneck_out && !fuse) {
        const long long n_pix = (long long)ex->batch * ex->in_h * ex->in_w;
        hipLaunchKernelGGL(bottleneck32_kernel, dim3((unsigned)((n_pix * 8 + 255) / 256)),
                           dim3(256), 0, s, d_feat_out, n_pix, ex->d_bneck_w, ex->bneck_scale,
                           ex->bneck_shift, d_bottleneck_out);
        DODT_LAUNCH_CHECK();
    }
    return DODT_OK;
}

int dodt_extractor_forward_padded(dodt_extractor* ex, const float* d_x0, float* d_feat_out,
                                  float* d_bottleneck_out) {
    DODT_REQUIRE(ex && d_x0 && d_feat_out, "dodt_extractor_forward_padded: NULL argument");
    // the first layer reads the caller's buffer in place of the extractor's own input buffer
    Buffer& x0 = ex->buf[X0];
    float* cur = x0.ptr;
    x0.ptr = const_cast<float*>(d_x0);
    const int rc = dodt_extractor_forward(ex, nullptr, d_feat_out, d_bottleneck_out);
    x0.ptr = cur;
    return rc;
}

int dodt_extractor_set_input(dodt_extractor* ex, const float* d_x0) {
    DODT_REQUIRE(ex, "dodt_extractor_set_input: extractor is NULL");
    if (!ex->own_x0) ex->own_x0 = ex->buf[X0].ptr;
    ex->buf[X0].ptr = d_x0 ? const_cast<float*>(d_x0) : ex->own_x0;
    return DODT_OK;
}

int dodt_extractor_output_shape(const dodt_extractor* ex, int* h, int* w, int* c) {
    DODT_REQUIRE(ex, "dodt_extractor_output_shape: NULL argument");
    if (h) *h = ex->out_h;
    if (w) *w = ex->out_w;
    if (c) *c = ex->out_c;
    return DODT_OK;
}

int dodt_extractor_first_layers_folded(const dodt_extractor* ex) {
    DODT_REQUIRE(ex, "dodt_extractor_first_layers_folded: extractor is NULL");
    return ex->first2_variant >= 0 ? 1 : 0;
}

int dodt_extractor_read_activation(dodt_extractor* ex, const char* name, float* dst, int* h,
                                   int* w, int* c) {
    DODT_REQUIRE(ex && name, "dodt_extractor_read_activation: NULL argument");
    const int li = find_layer(ex, name);
    DODT_REQUIRE(li >= 0, "dodt_extractor_read_activation: unknown layer '%s'", name);
    const Layer& l = ex->layers[li];
    DODT_REQUIRE(!(li == 0 && ex->first2_variant >= 0),
                 "layer %s runs folded into conv1_2's launch: its map is not stored (DODT_CONV_BF16_FIRST2=0 keeps it)", name);
    const Buffer& b = ex->buf[buffer_for_layer_output(l)];
    DODT_REQUIRE(b.ptr != nullptr,
                 "layer %s is written straight into the caller's output buffer", name);
    const int oh = l.deconv ? 2 * l.H : l.H, ow = l.deconv ? 2 * l.W : l.W;
    if (h) *h = oh;
    if (w) *w = ow;
    if (c) *c = l.Cout;
    if (!dst) return DODT_OK;
    // CB8 fp32 (or CB16 bf16) planes of every frame -> dense NHWC fp32 host tensor
    DODT_HIP_CHECK(hipStreamSynchronize(ex->ctx->stream));
    const int pc = b.bf16 ? 16 : 8;                 // channels per plane
    const size_t plane = (size_t)oh * ow * 8;       // floats per plane, both layouts
    const int planes = l.Cout / pc;
    std::vector<float> tmp(plane * planes);
    const uint16_t* tmp16 = reinterpret_cast<const uint16_t*>(tmp.data());
    for (int f = 0; f < ex->batch; ++f) {
        DODT_HIP_CHECK(hipMemcpy(tmp.data(),
                                 b.ptr + (size_t)f * b.frame_floats() + (size_t)(l.dst_coff / pc) * plane,
                                 tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
        float* o = dst + (size_t)f * oh * ow * l.Cout;
        for (int pl = 0; pl < planes; ++pl)
            for (size_t px = 0; px < (size_t)oh * ow; ++px)
                for (int k = 0; k < pc; ++k)
                    o[px * l.Cout + pl * pc + k] =
                        b.bf16 ? dodt::bf16_to_float(tmp16[(pl * plane + px * 8) * 2 + k])
                               : tmp[pl * plane + px * 8 + k];
        if (b.parts == 2) {   // split mode: add the lo map
            DODT_HIP_CHECK(hipMemcpy(tmp.data(),
                                     b.ptr + (size_t)(ex->batch + f) * b.frame_floats() +
                                         (size_t)(l.dst_coff / pc) * plane,
                                     tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (int pl = 0; pl < planes; ++pl)
                for (size_t px = 0; px < (size_t)oh * ow; ++px)
                    for (int k = 0; k < pc; ++k)
                        o[px * l.Cout + pl * pc + k] +=
                            dodt::bf16_to_float(tmp16[(pl * plane + px * 8) * 2 + k]);
        }
    }
    return DODT_OK;
}

double dodt_extractor_bytes(const dodt_extractor* ex) {
    if (!ex) return 0.0;
    double b = 0.0;
    for (const Layer& l : ex->layers) b += layer_bytes(ex, l);
    if (ex->kind == DODT_EXTRACTOR_VGG)   // upsampling: conv4_3 read, the feature map written
        b += (double)ex->batch * ((double)ex->buf[C4C].H * ex->buf[C4C].W * 256 +
                                  (double)ex->out_h * ex->out_w * 256) * 4.0;
    b += (double)ex->batch * ex->out_h * ex->out_w * 4.0;   // bottleneck map
    return b;
}

double dodt_extractor_mfma_flops(const dodt_extractor* ex) {
    if (!ex) return 0.0;
    double f = 0.0;
    for (const Layer& l : ex->layers) f += layer_executed_flops(ex, l);
    return f;
}

int dodt_extractor_layer_count(const dodt_extractor* ex) { return ex ? (int)ex->layers.size() : 0; }

int dodt_extractor_forward_timed(dodt_extractor* ex, const float* d_in, float* d_feat_out,
                                 float* d_bottleneck_out, dodt_layer_info* info, int n_info) {
    DODT_REQUIRE(ex && info && n_info >= (int)ex->layers.size(),
                 "dodt_extractor_forward_timed: info must hold dodt_extractor_layer_count() entries");
    for (Layer& l : ex->layers) {
        if (!l.ev0) DODT_HIP_CHECK(hipEventCreate(&l.ev0));
        if (!l.ev1) DODT_HIP_CHECK(hipEventCreate(&l.ev1));
    }
    ex->timed = true;
    const int rc = dodt_extractor_forward(ex, d_in, d_feat_out, d_bottleneck_out);
    ex->timed = false;
    if (rc) return rc;
    DODT_HIP_CHECK(hipStreamSynchronize(ex->ctx->stream));
    for (size_t i = 0; i < ex->layers.size(); ++i) {
        const Layer& l = ex->layers[i];
        dodt_layer_info& o = info[i];
        memset(&o, 0, sizeof(o));
        snprintf(o.name, sizeof(o.name), "%s", l.name.c_str());
        const bool folded = ex->first2_variant >= 0 && i < 2;      // conv1_1 and conv1_2 are one launch, timed as conv1_2
        snprintf(o.kernel, sizeof(o.kernel), "%s", kernel_name(variants()[folded ? ex->first2_variant : l.main.variant]));
        o.launches = folded && i == 0 ? 0 : l.tail.n_items > 0 ? 2 : 1;
        o.items = l.main.n_items + l.tail.n_items;
        o.flops_direct = layer_direct_flops(ex, l);
        o.flops_executed = layer_executed_flops(ex, l);
        o.bytes = layer_bytes(ex, l);
        DODT_HIP_CHECK(hipEventElapsedTime(&o.ms, l.ev0, l.ev1));
    }
    return DODT_OK;
}

double dodt_extractor_flops(const dodt_extractor* ex) {
    if (!ex) return 0.0;
    double f = 0.0;  // 2*M*N*K per layer; transposed convs counted on input pixels
    for (const Layer& l : ex->layers)
        f += 2.0 * l.H * l.W * (double)l.Cout * 9.0 * l.real_cin * ex->batch;
    return f;
}

}  // extern "C"
