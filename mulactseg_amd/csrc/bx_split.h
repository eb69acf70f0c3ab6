// bx_split.h -- the exact three-term bf16 split of f32 values shared by the split-bf16 matrix-core kernels (conv_bx.hip,
// conv_wgrad_bx.hip):
//     x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = x - h - m        (round to nearest even; both subtractions are exact in
//                                                                            f32, and l has at most 8 significand bits left)
// 8 + 8 + 8 significand bits with signed terms: |m| <= 2^-8 |x|, |l| <= 2^-16 |x|.  A product a * b is accumulated in f32 from
// its six partial products of order <= 2 on v_mfma_f32_32x32x16_bf16 (each one exact in f32: 8 x 8 bits):
//     ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm
// What is dropped (am*bl + al*bm + al*bl) is at most (2^-23 + 2^-32) |a*b|, of either sign -- the size of the rounding an f32
// multiply-add commits per product (2^-24), with no bias.  Exact for operands of up to 16 significant bits (integers etc.).
// Valid for 2^-100 < |x| < 2^127 (and 0): beyond that the third term would be subnormal / the first would overflow bf16.
// Numpy restatement: oracle/bx_split.py (tests/test_bx_split_cpu.py).
#pragma once
#include <hip/hip_runtime.h>

// two f32 -> one dword of two bf16 (low half = a, high half = b), round to nearest even
// (-DBX_CVT_BUILTIN: the same instruction through the vector conversion the compiler selects it for -- visible to the scheduler, which
//  an inline-asm statement is not; the measurement builds that interleave the split with MFMAs need it.  0.7-1.3 % slower in the
//  shipped kernels, profiles/r05/p_bx_variants_ab.md.)
static __device__ __forceinline__ unsigned bx_cvt_pk(float a, float b) {
#ifdef BX_CVT_BUILTIN
    typedef __bf16 bx_bf2 __attribute__((ext_vector_type(2)));
    typedef float bx_f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((bx_f2){a, b}, bx_bf2));
#else
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}

// (h, m, l) of two values, packed pairwise: dword = (bf16 of v1) << 16 | bf16 of v0
// (both subtractions of a pair as one v_pk_add_f32 were measured: the packed operands need register pairs, hipcc adds 12 moves for
//  the 16 instructions saved per chunk -- no difference.)
static __device__ __forceinline__ void bx_split2(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
    h = bx_cvt_pk(v0, v1);
    const float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
    m = bx_cvt_pk(r0, r1);
    const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = bx_cvt_pk(q0, q1);          // exact: at most 8 significand bits are left
}
