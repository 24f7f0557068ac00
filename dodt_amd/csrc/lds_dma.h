// LDS-DMA helpers shared by the Winograd conv kernel and the fully connected kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace dodt {

// 16 bytes per lane global -> LDS through a raw buffer descriptor: the LDS destination is
// lds_base + lane * 16 (wave-uniform base), the source rsrc.base + soffset + voffset; a voffset
// beyond the descriptor's size reads zeros (that is the conv's zero padding: no zero page,
// no per-lane pointer select).
// Issued as inline asm on purpose: hipcc treats the builtin form as an LDS store that every
// later ds_read may alias and drains it (s_waitcnt vmcnt) in front of the next LDS read, which
// would serialise the prefetch of chunk k+1 with the compute of chunk k.  The kernel waits for
// its copies itself (vmcnt(0) ahead of the barrier that publishes the buffer).
typedef int i32x4_t __attribute__((ext_vector_type(4)));
// raw buffer descriptor (V#) over [base, base + bytes): stride 0, 32-bit raw format word
__device__ __forceinline__ i32x4_t make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xffff);
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void blds16(i32x4_t rsrc, int voffset, int soffset, float* lds_base) {
    const unsigned lds_addr = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_base);
    soffset = __builtin_amdgcn_readfirstlane(soffset);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voffset), "s"(rsrc), "s"(soffset), "s"(lds_addr)
        : "memory");
}
// The same with scalar operands supplied by the caller (soffset and the LDS byte address must be
// wave-uniform values the compiler keeps in SGPRs): no v_readfirstlane in the issue path -- a
// vector instruction waits for the MFMAs of the SIMD's other wave, a scalar one does not
// (tools/micro/coissue.hip).
__device__ __forceinline__ void blds16s(i32x4_t rsrc, int voffset, int soffset, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voffset), "s"(rsrc), "s"(soffset), "s"(lds_addr)
        : "memory");
}
constexpr int kOob = (int)0x80000000;   // voffset of an out-of-image pixel

}  // namespace dodt
