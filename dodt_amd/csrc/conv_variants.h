// Kernel-variant table entries of the conv kernels (shared by conv.hip, which instantiates
// the fp32 kernels, and conv_bf16.hip, which instantiates the bf16 ones).
#pragma once
#include <vector>

#include "conv_kernels.h"

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
};

inline KernelVariant tail_only(KernelVariant v) {
    v.tail_only = true;
    return v;
}

template <int TW, int MTB, int WM, int WN, int BN, bool DECONV, bool BF16 = false>
struct Inst {
    using Cfg = ConvCfg<TW, MTB, WM, WN, BN, DECONV>;
    static void launch(const ConvArgs& a, dim3 grid, hipStream_t s) {
        hipLaunchKernelGGL((conv3x3_mfma_kernel<TW, MTB, WM, WN, BN, DECONV, BF16>), grid,
                           dim3(256), Cfg::kLdsBytes, s, a);
    }
    static hipError_t prepare() {
        return hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv3x3_mfma_kernel<TW, MTB, WM, WN, BN, DECONV, BF16>),
            hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::kLdsBytes);
    }
    static KernelVariant variant() {
        int per_cu = Cfg::kMinWaves;  // one wave of each workgroup per SIMD
        const int by_lds = (160 * 1024) / Cfg::kLdsBytes;
        if (per_cu > by_lds) per_cu = by_lds;
        KernelVariant v{TW, MTB, WM, WN, BN, BF16 ? 16 : kCK, DECONV, false, Cfg::TH,
                        Cfg::kLdsBytes, per_cu, &launch, &prepare};
        v.bf16 = BF16;
        return v;
    }
};

template <int TW, int MTB, int CK, bool OUT16 = false>
struct InstSmall {
    using Cfg = SmallCfg<TW, MTB, CK>;
    static void launch(const ConvArgs& a, dim3 grid, hipStream_t s) {
        hipLaunchKernelGGL((conv3x3_small_cin_kernel<TW, MTB, CK, OUT16>), grid, dim3(256),
                           Cfg::kLdsBytes, s, a);
    }
    static hipError_t prepare() {
        return hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv3x3_small_cin_kernel<TW, MTB, CK, OUT16>),
            hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::kLdsBytes);
    }
    static KernelVariant variant() {
        KernelVariant v{TW, MTB, 4, 1, 32, CK, false, true, Cfg::TH, Cfg::kLdsBytes, 4, &launch,
                        &prepare};
        v.bf16 = OUT16;
        return v;
    }
};

// conv_bf16.hip
std::vector<KernelVariant> bf16_variants();

}  // namespace dodt
