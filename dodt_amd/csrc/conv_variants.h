// Kernel-variant table entries of the conv kernels (shared by conv.hip, which instantiates
// the fp32 kernels, and conv_bf16.hip, which instantiates the bf16 ones).
#pragma once
#include <vector>

#include "conv_kernels.h"
#include "wino_kernels.h"
#include "wino43_kernel.h"
#include "deconv_kernel.h"
#include "conv_bf16_dma.h"

namespace dodt {

struct KernelVariant {
    int TW, MTB, WM, WN, BN, CK;
    bool deconv, small_cin;
    int TH, lds_bytes;
    int blocks_per_cu;  // resident workgroups per CU (registers and LDS permitting)
    void (*launch)(const ConvArgs&, dim3 grid, hipStream_t s);
    hipError_t (*prepare)();
    bool tail_only = false;  // quarter-size tiles: never a layer's main variant
    bool bf16 = false;       // CB16 bf16 activations (first-layer kernels: bf16 OUTPUT)
    int parts = 1;           // 2: split mode, every map is a hi + lo pair of bf16 maps
    bool wino = false;       // Winograd F(2x2,3x3) kernel (wino_kernels.h): 16 weight points, not 9 taps
    bool dma = false;        // bf16 kernel with LDS-DMA staging (conv_bf16_dma.h)
    int wino_m = 2;          // Winograd output block: F(2x2,3x3) or F(4x4,3x3) (wino43_kernel.h)
    bool deconv_dma = false; // transposed conv, LDS-DMA staged (deconv_kernel.h)
    int first2 = 0;          // 6 / 4: conv1_1 (that many input channels) folded into conv1_2 (conv3x3_bf16_first2_kernel)
    int stream_nch = 0;      // > 0: bf16 LDS-DMA kernel with a producer wave and the weights in registers
                             // (conv3x3_bf16_stream_kernel): exactly this many 16-channel chunks, 32 output channels
};

inline KernelVariant tail_only(KernelVariant v) {
    v.tail_only = true;
    return v;
}

template <int TW, int MTB, int WM, int WN, int BN, bool DECONV, bool BF16 = false, int PARTS = 1>
struct Inst {
    using Cfg = ConvCfg<TW, MTB, WM, WN, BN, DECONV, PARTS>;
    static_assert(Cfg::kLdsBytes <= 160 * 1024, "variant does not fit the LDS");
    static void launch(const ConvArgs& a, dim3 grid, hipStream_t s) {
        hipLaunchKernelGGL((conv3x3_mfma_kernel<TW, MTB, WM, WN, BN, DECONV, BF16, PARTS>), grid,
                           dim3(256), Cfg::kLdsBytes, s, a);
    }
    static hipError_t prepare() {
        return hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv3x3_mfma_kernel<TW, MTB, WM, WN, BN, DECONV, BF16, PARTS>),
            hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::kLdsBytes);
    }
    static KernelVariant variant() {
        int per_cu = Cfg::kMinWaves;  // one wave of each workgroup per SIMD
        const int by_lds = (160 * 1024) / Cfg::kLdsBytes;
        if (per_cu > by_lds) per_cu = by_lds;
        KernelVariant v{TW, MTB, WM, WN, BN, BF16 ? 16 : kCK, DECONV, false, Cfg::TH,
                        Cfg::kLdsBytes, per_cu, &launch, &prepare};
        v.bf16 = BF16;
        v.parts = PARTS;
        return v;
    }
};

template <int TB, int CB>
struct InstWino {
    using Cfg = WinoCfg<TB, CB>;
    static_assert(Cfg::kLdsBytes <= 160 * 1024, "variant does not fit the LDS");
    static void launch(const ConvArgs& a, dim3 grid, hipStream_t s) {
        hipLaunchKernelGGL((wino3x3_f32_kernel<TB, CB>), grid, dim3(256), Cfg::kLdsBytes, s, a);
    }
    static hipError_t prepare() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&wino3x3_f32_kernel<TB, CB>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::kLdsBytes);
    }
    static KernelVariant variant() {
        // TW, MTB (unused), WM, WN, BN, CK, deconv, small_cin, TH, lds, blocks_per_cu
        KernelVariant v{Cfg::TW, 0, 4, 1, Cfg::BN, kCK, false, false, Cfg::TH, Cfg::kLdsBytes,
                        Cfg::kPipe ? 1 : 2, &launch, &prepare};
        v.wino = true;
        return v;
    }
};

struct InstWino43 {
    using Cfg = Wino43Cfg;
    static_assert(Cfg::kLdsBytes <= 160 * 1024, "variant does not fit the LDS");
    static void launch(const ConvArgs& a, dim3 grid, hipStream_t s) {
        hipLaunchKernelGGL(wino43_f32_kernel<0>, grid, dim3(256), Cfg::kLdsBytes, s, a);
    }
    static hipError_t prepare() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&wino43_f32_kernel<0>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::kLdsBytes);
    }
    static KernelVariant variant() {
        KernelVariant v{Cfg::TW, 0, 4, 1, Cfg::BN, kCK, false, false, Cfg::TH, Cfg::kLdsBytes, 1,
                        &launch, &prepare};
        v.wino = true;
        v.wino_m = 4;
        return v;
    }
};

template <int CB, bool BF16 = false>
struct InstDeconvDma {
    using Cfg = DeconvCfg<CB>;
    static void launch(const ConvArgs& a, dim3 grid, hipStream_t s) {
        hipLaunchKernelGGL((deconv3x3_dma_kernel<CB, BF16>), grid, dim3(256), Cfg::kLdsBytes, s, a);
    }
    static hipError_t prepare() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&deconv3x3_dma_kernel<CB, BF16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::kLdsBytes);
    }
    static KernelVariant variant() {
        KernelVariant v{Cfg::TW, 0, 4, 1, Cfg::BN, BF16 ? 16 : kCK, true, false, Cfg::TH, Cfg::kLdsBytes, 2,
                        &launch, &prepare};
        v.deconv_dma = true;
        v.bf16 = BF16;
        return v;
    }
};

template <int MT, int NT, int S>
struct InstBf16Dma {
    using Cfg = Bf16DmaCfg<MT, NT, S>;
    static_assert(Cfg::kLdsBytes <= 160 * 1024, "variant does not fit the LDS");
    static void launch(const ConvArgs& a, dim3 grid, hipStream_t s) {
        hipLaunchKernelGGL((conv3x3_bf16_dma_kernel<MT, NT, S>), grid, dim3(256), Cfg::kLdsBytes, s, a);
    }
    static hipError_t prepare() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_dma_kernel<MT, NT, S>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::kLdsBytes);
    }
    static KernelVariant variant() {
        int per_cu = (160 * 1024) / Cfg::kLdsBytes;
        constexpr int kMax = S == 2 ? (MT == 2 ? 3 : 2) : 1;      // the kernel's launch bounds
        if (per_cu > kMax) per_cu = kMax;
        KernelVariant v{Cfg::TW, 4 * MT, 4, 1, Cfg::BN, 16, false, false, Cfg::TH, Cfg::kLdsBytes,
                        per_cu, &launch, &prepare};
        v.bf16 = true;
        v.dma = true;
        return v;
    }
};

template <int MT, int NCH, int S>
struct InstBf16Stream {
    using Cfg = Bf16StreamCfg<MT, NCH, S>;
    static_assert(2 * Cfg::kLdsBytes <= 160 * 1024, "two workgroups per CU");
    static void launch(const ConvArgs& a, dim3 grid, hipStream_t s) {
        hipLaunchKernelGGL((conv3x3_bf16_stream_kernel<MT, NCH, S>), grid, dim3(256), Cfg::kLdsBytes, s, a);
    }
    static hipError_t prepare() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_stream_kernel<MT, NCH, S>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::kLdsBytes);
    }
    static KernelVariant variant() {
        KernelVariant v{Cfg::TW, Cfg::TH, 3, 1, Cfg::BN, 16, false, false, Cfg::TH, Cfg::kLdsBytes, 2, &launch, &prepare};
        v.bf16 = true;
        v.dma = true;
        v.stream_nch = NCH;
        return v;
    }
};

template <int CIN0, int S>
struct InstBf16First2 {
    using Cfg = Bf16First2Cfg<CIN0, S>;
    static_assert(2 * Cfg::kLdsBytes <= 160 * 1024, "two workgroups per CU");
    static void launch(const ConvArgs& a, dim3 grid, hipStream_t s) {
        hipLaunchKernelGGL((conv3x3_bf16_first2_kernel<CIN0, S>), grid, dim3(256), Cfg::kLdsBytes, s, a);
    }
    static hipError_t prepare() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_first2_kernel<CIN0, S>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::kLdsBytes);
    }
    static KernelVariant variant() {
        KernelVariant v{Cfg::TW, Cfg::TH, 3, 1, Cfg::BN, 16, false, false, Cfg::TH, Cfg::kLdsBytes, 2, &launch, &prepare};
        v.bf16 = true;
        v.dma = true;
        v.first2 = CIN0;
        return v;
    }
};

template <int TW, int MTB, int CK, bool OUT16 = false, bool SPLIT = false>
struct InstSmall {
    using Cfg = SmallCfg<TW, MTB, CK>;
    static void launch(const ConvArgs& a, dim3 grid, hipStream_t s) {
        hipLaunchKernelGGL((conv3x3_small_cin_kernel<TW, MTB, CK, OUT16, SPLIT>), grid, dim3(256),
                           Cfg::kLdsBytes, s, a);
    }
    static hipError_t prepare() {
        return hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv3x3_small_cin_kernel<TW, MTB, CK, OUT16, SPLIT>),
            hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::kLdsBytes);
    }
    static KernelVariant variant() {
        KernelVariant v{TW, MTB, 4, 1, 32, CK, false, true, Cfg::TH, Cfg::kLdsBytes, 4, &launch,
                        &prepare};
        v.bf16 = OUT16;
        v.parts = SPLIT ? 2 : 1;
        return v;
    }
};

// conv_bf16.hip, conv_split.hip
std::vector<KernelVariant> bf16_variants();
std::vector<KernelVariant> split_variants();

}  // namespace dodt
