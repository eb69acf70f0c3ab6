// common.h -- launch helpers shared by the .hip translation units of libmulactseg_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mulactseg_hip.h"
#include "detmath.h"

// argument-error codes (negative; positive codes are hipError_t values)
#define MAS_ERR_NULL (-1)
#define MAS_ERR_SHAPE (-2)
#define MAS_ERR_CLASSES (-3)
#define MAS_ERR_DTYPE (-4)
#define MAS_ERR_ALIGN (-5)
#define MAS_ERR_RANGE (-6)
#define MAS_ERR_WORKSPACE (-7)

#define MAS_WAVE 64

typedef unsigned long long mas_u64;   // the type HIP's 64-bit atomics are declared on

static inline int mas_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// 16-byte load of logits that are read ONCE (the scans stream 671 MB per pool batch through every CU exactly once): the
// non-temporal policy (global_load_dwordx4 ... nt) leaves the lines out of the way of the L2's replacement and shortens the
// issue -> data round trip under load (MI355X_MICROARCH.md, "nt-weights": -18 %); the ring scan, which is bound by bytes in flight
// divided by that round trip, went from 161 to 154 us per [4,20,1024,2048] batch.  NOT for data another kernel re-reads soon
// (superpixel ids: static maps that stay in the Infinity Cache between batches -- measured 1.5 % slower with nt).
__device__ __forceinline__ float4 mas_load_stream4(const float* p) {
    typedef float mas_v4nt __attribute__((ext_vector_type(4)));
    const mas_v4nt q = __builtin_nontemporal_load(reinterpret_cast<const mas_v4nt*>(p));
    return make_float4(q.x, q.y, q.z, q.w);
}

template <typename IdT>
__device__ __forceinline__ int mas_load_id(const IdT* p, size_t i) {
    return (int)p[i];
}

// Per-pixel softmax(z * invT) over CT register-resident channels; `C` live channels (C == CT when EXACT).
// Operation order is normative (mirrored by oracle/exact.c:softmax_row):
//   m = max_c z_c ; M = m*invT ; e_c = exp_np(fma(z_c, invT, -M)) ; sum = ((e_0+e_1)+e_2)... ; rinv = 1/sum
// On return x[c] = e_c (un-normalised) and the function value is rinv; p_c = e_c * rinv.
template <int CT, bool EXACT>
__device__ __forceinline__ float mas_softmax_regs(float (&x)[CT], int C, float invT) {
    const int Cn = EXACT ? CT : C;
    float m = x[0];
#pragma unroll
    for (int c = 1; c < CT; ++c)
        if (EXACT || c < Cn) m = (x[c] > m) ? x[c] : m;
    const float negM = -(m * invT);
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (EXACT || c < Cn) {
            x[c] = mas_expf_np(mas_fmaf(x[c], invT, negM));
            sum = (c == 0) ? x[c] : (sum + x[c]);
        }
    }
    return 1.0f / sum;
}

// ------------------------------------------------------------------------------------------------
// Packed (2 pixels per lane-op) form of the same arithmetic: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
// evaluate both halves with IEEE semantics, so every element is bit-identical to the scalar functions of
// detmath.h -- the oracle keeps using the scalar form.
// ------------------------------------------------------------------------------------------------
typedef float mas_v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ mas_v2f mas_pk_fma(mas_v2f a, mas_v2f b, mas_v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ mas_v2f mas_splat(float v) { return (mas_v2f){v, v}; }
// {p.x * w.x, p.x * w.y} and {p.y * w.x, p.y * w.y}: one half of the register pair `p` times both halves of `w`, with the half
// selected on SRC0.  Written as instructions because the compiler's own choice for "{s, s} * w" may be the src1 form
// (v_pk_mul_f32 d, w, p op_sel:[0,1]), and on this part a packed-f32 instruction whose VGPR src1 is read with op_sel[1] = 1 returns
// wrong results in lanes 48-63 while a wave of ANOTHER kernel on the same SIMD runs MFMAs with AGPR accumulators (measured:
// profiles/r06/p_pk_opsel_probe.md, tools/pk_opsel_probe.py; tests/test_isa_cpu.py keeps the library free of that form).
__device__ __forceinline__ mas_v2f mas_pk_mul_lo(mas_v2f p, mas_v2f w) {
    mas_v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(p), "v"(w));
    return r;
}
__device__ __forceinline__ mas_v2f mas_pk_mul_hi(mas_v2f p, mas_v2f w) {
    mas_v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(p), "v"(w));
    return r;
}

// single v_max_f32 (no canonicalising pre-ops); operands are never NaN here
__device__ __forceinline__ float mas_vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Two independent pairs at once, written interleaved: a single exp chain is ~17 DEPENDENT packed instructions and
// the compiler does not interleave chains on its own, so one wave would stall on every instruction.  Same
// arithmetic per element as the scalar mas_expf_np of detmath.h (clamp, magic-number rint, Cody-Waite, degree-5
// polynomial, integer exponent add).
__device__ __forceinline__ void mas_expf_np2x2(mas_v2f xa, mas_v2f xb, mas_v2f& ea, mas_v2f& eb) {
    const mas_v2f ca = {mas_vmax(xa.x, -86.0f), mas_vmax(xa.y, -86.0f)};
    const mas_v2f cb = {mas_vmax(xb.x, -86.0f), mas_vmax(xb.y, -86.0f)};
    const mas_v2f magic = mas_splat(12582912.0f);
    const mas_v2f ta = mas_pk_fma(ca, mas_splat(1.44269504088896341f), magic);
    const mas_v2f tb = mas_pk_fma(cb, mas_splat(1.44269504088896341f), magic);
    const mas_v2f na = ta - magic;
    const mas_v2f nb = tb - magic;
    mas_v2f ra = mas_pk_fma(na, mas_splat(-0.693359375f), ca);
    mas_v2f rb = mas_pk_fma(nb, mas_splat(-0.693359375f), cb);
    ra = mas_pk_fma(na, mas_splat(2.12194440e-4f), ra);
    rb = mas_pk_fma(nb, mas_splat(2.12194440e-4f), rb);
    mas_v2f pa = mas_pk_fma(mas_splat(1.9875691500e-4f), ra, mas_splat(1.3981999507e-3f));
    mas_v2f pb = mas_pk_fma(mas_splat(1.9875691500e-4f), rb, mas_splat(1.3981999507e-3f));
    const mas_v2f qa = ra * ra;
    const mas_v2f qb = rb * rb;
    pa = mas_pk_fma(pa, ra, mas_splat(8.3334519073e-3f));
    pb = mas_pk_fma(pb, rb, mas_splat(8.3334519073e-3f));
    pa = mas_pk_fma(pa, ra, mas_splat(4.1665795894e-2f));
    pb = mas_pk_fma(pb, rb, mas_splat(4.1665795894e-2f));
    pa = mas_pk_fma(pa, ra, mas_splat(1.6666665459e-1f));
    pb = mas_pk_fma(pb, rb, mas_splat(1.6666665459e-1f));
    pa = mas_pk_fma(pa, ra, mas_splat(5.0000001201e-1f));
    pb = mas_pk_fma(pb, rb, mas_splat(5.0000001201e-1f));
    const mas_v2f ya = mas_pk_fma(pa, qa, ra) + mas_splat(1.0f);
    const mas_v2f yb = mas_pk_fma(pb, qb, rb) + mas_splat(1.0f);
    ea = (mas_v2f){mas_u2f(mas_f2u(ya.x) + (mas_f2u(ta.x) << 23)), mas_u2f(mas_f2u(ya.y) + (mas_f2u(ta.y) << 23))};
    eb = (mas_v2f){mas_u2f(mas_f2u(yb.x) + (mas_f2u(tb.x) << 23)), mas_u2f(mas_f2u(yb.y) + (mas_f2u(tb.y) << 23))};
}

// softmax of FOUR pixels (two pairs) at once; xa/xb hold raw logits in, e_c out; returns (rinv_a, rinv_b).
template <int CT, bool EXACT, bool KNOWN_MAX = false>
__device__ __forceinline__ void mas_softmax_quad(mas_v2f (&xa)[CT], mas_v2f (&xb)[CT], int C, float invT, mas_v2f& rinv_a,
                                                 mas_v2f& rinv_b, mas_v2f zmax_a = (mas_v2f){0.f, 0.f},
                                                 mas_v2f zmax_b = (mas_v2f){0.f, 0.f}) {
    const int Cn = EXACT ? CT : C;
    const mas_v2f it = mas_splat(invT);
    if (!KNOWN_MAX) {
        float m0 = xa[0].x, m1 = xa[0].y, m2 = xb[0].x, m3 = xb[0].y;
#pragma unroll
        for (int c = 1; c < CT; ++c) {
            if (EXACT || c < Cn) {
                m0 = mas_vmax(m0, xa[c].x);
                m1 = mas_vmax(m1, xa[c].y);
                m2 = mas_vmax(m2, xb[c].x);
                m3 = mas_vmax(m3, xb[c].y);
            }
        }
        zmax_a = (mas_v2f){m0, m1};
        zmax_b = (mas_v2f){m2, m3};
    }
    const mas_v2f negMa = -(zmax_a * it), negMb = -(zmax_b * it);
    mas_v2f sa = mas_splat(0.0f), sb = mas_splat(0.0f);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (EXACT || c < Cn) {
            mas_expf_np2x2(mas_pk_fma(xa[c], it, negMa), mas_pk_fma(xb[c], it, negMb), xa[c], xb[c]);
            sa = (c == 0) ? xa[c] : (sa + xa[c]);
            sb = (c == 0) ? xb[c] : (sb + xb[c]);
        }
    }
    rinv_a = (mas_v2f){1.0f / sa.x, 1.0f / sa.y};
    rinv_b = (mas_v2f){1.0f / sb.x, 1.0f / sb.y};
}

// LDS adds carry workgroup scope, the global fallbacks agent scope: with the same scope on both the compiler
// if-converts "LDS slot or global" into a pointer select and ONE flat_atomic, and a pending FLAT operation forces
// s_waitcnt vmcnt(0) -- which would drain the prefetched row in the middle of the pipeline.
template <typename T>
__device__ __forceinline__ void lds_add(T* p, T v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// BvSB margins of four pixels (mas_bvsb element-wise: exp_np(z2*invT - z1*invT) + 1e-8) on packed pairs, and their
// fixed-point quanta.  mas_fix_unit(v) == mas_fix(v, MAS_SCORE_FRAC) for 0 < v < 2 (a margin lies in [1e-8, 1 + 1e-8]):
// the shift count is e - 110 in [-27, 17], so one 64-bit left shift or one 32-bit right shift suffices.
__device__ __forceinline__ void mas_bvsb_quad(const float (&b1)[4], const float (&b2)[4], float invT, float (&out)[4]) {
    const mas_v2f it = mas_splat(invT);
    const mas_v2f da = ((mas_v2f){b2[0], b2[1]} * it) - ((mas_v2f){b1[0], b1[1]} * it);
    const mas_v2f db = ((mas_v2f){b2[2], b2[3]} * it) - ((mas_v2f){b1[2], b1[3]} * it);
    mas_v2f ea, eb;
    mas_expf_np2x2(da, db, ea, eb);
    ea = ea + mas_splat(1e-8f);
    eb = eb + mas_splat(1e-8f);
    out[0] = ea.x; out[1] = ea.y; out[2] = eb.x; out[3] = eb.y;
}

__device__ __forceinline__ mas_u64 mas_fix_unit(float v) {
    const unsigned b = mas_f2u(v);
    const int sh = (int)(b >> 23) - (150 - MAS_SCORE_FRAC);      // sign bit is 0
    const unsigned m = (b & 0x007fffffu) | 0x00800000u;
    const mas_u64 up = (mas_u64)m << (sh & 31);
    const unsigned dn = m >> ((-sh) & 31);
    return sh >= 0 ? up : (mas_u64)dn;
}
