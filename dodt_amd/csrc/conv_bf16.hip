// bf16 instantiations of the conv kernels (separate translation unit: they compile in
// parallel with the fp32 ones).  Same tilings as conv.hip's main variants.
#include "common.h"
#include "conv_variants.h"

namespace dodt {

std::vector<KernelVariant> bf16_variants() {
    return {
        InstSmall<32, 16, 6, true>::variant(),
        InstSmall<16, 12, 4, true>::variant(),
        InstSmall<16, 16, 6, true>::variant(),
        InstSmall<16, 16, 4, true>::variant(),
        Inst<32, 16, 4, 1, 32, false, true>::variant(),
        Inst<16, 16, 4, 1, 32, false, true>::variant(),
        Inst<16, 8, 4, 1, 64, false, true>::variant(),
        Inst<16, 12, 4, 1, 32, false, true>::variant(),
        Inst<8, 8, 4, 1, 32, false, true>::variant(),
        Inst<8, 8, 4, 1, 64, false, true>::variant(),
        Inst<8, 4, 2, 2, 128, false, true>::variant(),
        Inst<8, 4, 4, 1, 64, false, true>::variant(),
        Inst<4, 4, 2, 2, 128, false, true>::variant(),
        Inst<4, 4, 4, 1, 64, false, true>::variant(),
        Inst<16, 4, 4, 1, 32, true, true>::variant(),
        Inst<8, 4, 4, 1, 32, true, true>::variant(),
        Inst<4, 4, 4, 1, 32, true, true>::variant(),
        Inst<16, 4, 4, 1, 64, true, true>::variant(),
        Inst<8, 4, 4, 1, 64, true, true>::variant(),
        Inst<4, 4, 4, 1, 64, true, true>::variant(),
        // round 2: LDS-DMA staged 3x3 stride-1 kernels, 16 x 32 px x 64 / 32 channels
        InstBf16Dma<4, 2, 2>::variant(),
        InstBf16Dma<4, 1, 2>::variant(),
        InstBf16Dma<2, 1, 2>::variant(),      // round 4: 8 x 32 px x 32 channels, three workgroups per CU
        // round 4: producer wave + weights in registers, for the level-1 layers (32 output channels, 2 / 4 chunks)
        InstBf16Stream<2, 2, 8>::variant(),
        InstBf16Stream<2, 4, 6>::variant(),
        // (shallower rings, less LDS: DODT_CONV_BF16_STREAM_LDS=<KB> picks the deepest ring that fits -- tools/)
        InstBf16Stream<2, 2, 5>::variant(),
        InstBf16Stream<2, 2, 3>::variant(),
        InstBf16Stream<2, 4, 4>::variant(),
        InstBf16Stream<2, 4, 3>::variant(),
        // round 4: conv1_1 folded into conv1_2 (never picked by shape: dodt_extractor_forward launches it in conv1_2's place)
        InstBf16First2<6, 4>::variant(),
        InstBf16First2<4, 6>::variant(),
        InstBf16First2<6, 3>::variant(),
        InstBf16First2<4, 4>::variant(),
        InstBf16First2<4, 3>::variant(),
        // round 3: the LDS-DMA staged transposed conv on v_mfma_f32_16x16x16_bf16 (deconv_kernel.h)
        InstDeconvDma<2, true>::variant(),
        InstDeconvDma<1, true>::variant(),
    };
}

}  // namespace dodt
