// Split-mode instantiations of the conv kernels (hi + lo bf16 maps, three bf16 MFMAs per
// product term, fp32 accumulate: fp32-grade results from the bf16 matrix pipe).  Every LDS
// image exists twice, so the 128-channel tiles do not fit; the rest of conv_bf16.hip's list.
#include "common.h"
#include "conv_variants.h"

namespace dodt {

std::vector<KernelVariant> split_variants() {
    return {
        InstSmall<32, 16, 6, true, true>::variant(),
        InstSmall<16, 12, 4, true, true>::variant(),
        InstSmall<16, 16, 6, true, true>::variant(),
        InstSmall<16, 16, 4, true, true>::variant(),
        Inst<32, 16, 4, 1, 32, false, true, 2>::variant(),
        Inst<16, 16, 4, 1, 32, false, true, 2>::variant(),
        Inst<16, 8, 4, 1, 64, false, true, 2>::variant(),
        Inst<16, 12, 4, 1, 32, false, true, 2>::variant(),
        Inst<8, 8, 4, 1, 32, false, true, 2>::variant(),
        Inst<8, 8, 4, 1, 64, false, true, 2>::variant(),
        Inst<8, 4, 4, 1, 64, false, true, 2>::variant(),
        Inst<4, 4, 4, 1, 64, false, true, 2>::variant(),
        Inst<16, 4, 4, 1, 32, true, true, 2>::variant(),
        Inst<8, 4, 4, 1, 32, true, true, 2>::variant(),
        Inst<4, 4, 4, 1, 32, true, true, 2>::variant(),
        Inst<16, 4, 4, 1, 64, true, true, 2>::variant(),
        Inst<8, 4, 4, 1, 64, true, true, 2>::variant(),
        Inst<4, 4, 4, 1, 64, true, true, 2>::variant(),
    };
}

}  // namespace dodt
