// bx_split.h -- the exact three-term bf16 split of f32 values shared by the split-bf16 matrix-core kernels (conv_bx.hip,
// conv_wgrad_bx.hip):  x = h + m + l with h = trunc16(x), m = trunc16(x - h), l = x - h - m (8 + 8 + 8 significand bits; both
// subtractions are exact in f32).  A product a * b is then accumulated in f32 from its six partial products of order <= 2 on
// v_mfma_f32_32x32x16_bf16: ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm; what is dropped is below 2^-23 |a*b|.
#pragma once
#include <hip/hip_runtime.h>

// (h, m, l) of two values, packed pairwise: dword = (bf16 of v1) << 16 | bf16 of v0
static __device__ __forceinline__ void bx_split2(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(v0), u1 = __float_as_uint(v1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = v0 - __uint_as_float(u0 & 0xffff0000u), r1 = v1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned s0 = __float_as_uint(r0), s1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(s1, s0, 0x07060302u);
    const float q0 = r0 - __uint_as_float(s0 & 0xffff0000u), q1 = r1 - __uint_as_float(s1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

