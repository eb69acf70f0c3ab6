// upsample.hip -- K8 (part): bilinear upsampling, align_corners=False, forward and a DETERMINISTIC gather backward.
//
// Reference: models/segmentation/utils.py:25 (logits x4 to the input size) and deeplabv3.py:116 (ASPP output x4 to the
// low-level feature size) -- F.interpolate(mode='bilinear', align_corners=False).  ATen's backward scatters every output
// gradient into four inputs with float atomics: 1.1 ms for the [4,20,768,768] logit gradient and a result whose last bits
// depend on the atomic order.  Here every INPUT pixel gathers the <= ~(2/scale + 3)^2 outputs that touch it, in a fixed
// order (rows ascending, columns ascending): one read of the gradient (L2 serves the 4x reuse), no atomics, bit-identical
// from run to run.  Index / weight arithmetic is ATen's (area_pixel_compute_source_index): src = max(0, s*(o+0.5)-0.5),
// i0 = (int)src, i1 = i0 + (i0 < n-1), l1 = src - i0, l0 = 1 - l1;  y = l0h*(l0w*v00 + l1w*v01) + l1h*(l0w*v10 + l1w*v11).
#include "common.h"

namespace {
constexpr int kThreads = 256;
constexpr int kMaxTaps = 16;

struct Tap { int i0, i1; float l0, l1; };

__device__ __forceinline__ Tap make_tap(float scale, int o, int n_in) {
    float s = scale * ((float)o + 0.5f) - 0.5f;
    s = s < 0.0f ? 0.0f : s;
    Tap t;
    t.i0 = (int)s;
    t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
    t.l1 = s - (float)t.i0;
    t.l0 = 1.0f - t.l1;
    return t;
}

// grid: (ceil(Wo / (4*256)), Ho, NC); a thread writes four consecutive outputs of one row
__global__ __launch_bounds__(kThreads) void k_upsample_fwd(const float* __restrict__ x, int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                                            float* __restrict__ y) {
    const int ox0 = (blockIdx.x * kThreads + threadIdx.x) * 4;
    if (ox0 >= Wo) return;
    const int oy = blockIdx.y;
    const size_t nc = blockIdx.z;
    const Tap ty = make_tap(sh, oy, Hi);
    const float* r0 = x + (nc * Hi + ty.i0) * Wi;
    const float* r1 = x + (nc * Hi + ty.i1) * Wi;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ox = ox0 + k < Wo ? ox0 + k : Wo - 1;
        const Tap tx = make_tap(sw, ox, Wi);
        v[k] = ty.l0 * (tx.l0 * r0[tx.i0] + tx.l1 * r0[tx.i1]) + ty.l1 * (tx.l0 * r1[tx.i0] + tx.l1 * r1[tx.i1]);
    }
    float* dst = y + (nc * Ho + oy) * Wo + ox0;
    if (ox0 + 3 < Wo && (Wo & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    else
        for (int k = 0; k < 4 && ox0 + k < Wo; ++k) dst[k] = v[k];
}

// first / last output index that can touch input index i (conservative by one on each side; exact test in the loop)
__device__ __forceinline__ void out_range(float inv_scale, int i, int n_out, int& lo, int& hi) {
    lo = (int)floorf(((float)i - 0.5f) * inv_scale - 0.5f) - 1;
    hi = (int)ceilf(((float)i + 1.5f) * inv_scale - 0.5f) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > n_out - 1 ? n_out - 1 : hi;
}

// grid: (ceil(Wi / 64), ceil(Hi / 4), NC), four waves per workgroup, one input row each (input rows are 48 .. 512 wide)
// FAST4 (Wo == 4 Wi, 16-byte aligned rows -- the x4 of the ASPP output, deeplabv3.py:116): an interior column's candidates 4 ix - 4 ..
// 4 ix + 7 carry weight at positions 2 .. 9 only, i.e. output columns 4 ix - 2 .. 4 ix + 5: one 8-byte, one 16-byte and one 8-byte
// load per gradient row instead of 16 tested scalar loads: 187 -> 151 us per training step; the same fmas in the same order.
// (Also unrolling the eight weighted rows of an interior input row: 167 us.  The x4 forward with three loads per input row: 87 = 86 us.)
template <bool FAST4>
__global__ __launch_bounds__(4 * MAS_WAVE) void k_upsample_bwd(const float* __restrict__ gy, int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                                                float* __restrict__ gx) {
    const int ix = blockIdx.x * MAS_WAVE + (threadIdx.x & (MAS_WAVE - 1));
    const int iy = blockIdx.y * 4 + (threadIdx.x / MAS_WAVE);
    if (ix >= Wi || iy >= Hi) return;
    const size_t nc = blockIdx.z;
    int xlo, xhi, ylo, yhi;
    out_range(1.0f / sw, ix, Wo, xlo, xhi);
    out_range(1.0f / sh, iy, Ho, ylo, yhi);
    float wx[kMaxTaps];
#pragma unroll
    for (int k = 0; k < kMaxTaps; ++k) {
        const int ox = xlo + k;
        float w = 0.0f;
        if (ox <= xhi) {
            const Tap t = make_tap(sw, ox, Wi);
            w = (t.i0 == ix ? t.l0 : 0.0f) + (t.i1 == ix ? t.l1 : 0.0f);
        }
        wx[k] = w;
    }
    const float* g = gy + nc * Ho * Wo;
    float acc = 0.0f;
    // (interior column of an exact x4: xlo = 4 ix - 4, the weights at k = 2 .. 9 are all non-zero, every other one is zero)
    const bool fast = FAST4 && ix >= 1 && ix + 2 <= Wi && xlo == 4 * ix - 4;
    for (int oy = ylo; oy <= yhi; ++oy) {
        const Tap t = make_tap(sh, oy, Hi);
        const float wy = (t.i0 == iy ? t.l0 : 0.0f) + (t.i1 == iy ? t.l1 : 0.0f);
        if (wy == 0.0f) continue;
        float row = 0.0f;
        if (fast) {
            const float* q = g + (size_t)oy * Wo + 4 * ix;
            const float2 a = *reinterpret_cast<const float2*>(q - 2);
            const float4 b = *reinterpret_cast<const float4*>(q);
            const float2 c = *reinterpret_cast<const float2*>(q + 4);
            row = mas_fmaf(wx[2], a.x, row); row = mas_fmaf(wx[3], a.y, row);
            row = mas_fmaf(wx[4], b.x, row); row = mas_fmaf(wx[5], b.y, row); row = mas_fmaf(wx[6], b.z, row); row = mas_fmaf(wx[7], b.w, row);
            row = mas_fmaf(wx[8], c.x, row); row = mas_fmaf(wx[9], c.y, row);
        } else {
#pragma unroll
            for (int k = 0; k < kMaxTaps; ++k) {
                const int ox = xlo + k;
                if (ox <= xhi && wx[k] != 0.0f) row = mas_fmaf(wx[k], g[(size_t)oy * Wo + ox], row);
            }
        }
        acc = mas_fmaf(wy, row, acc);
    }
    gx[(nc * Hi + iy) * Wi + ix] = acc;
}

int check(long long NC, int Hi, int Wi, int Ho, int Wo) {
    if (NC <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return MAS_ERR_SHAPE;
    if (NC > 65535 || Ho > 65535 || Hi > 65535) return MAS_ERR_SHAPE;       // grid.y / grid.z limits
    return 0;
}
}  // namespace

extern "C" int mas_upsample_bilinear_fwd(const float* x, int64_t NC, int Hi, int Wi, int Ho, int Wo, float* y, void* stream) {
    if (!x || !y) return MAS_ERR_NULL;
    if (int e = check(NC, Hi, Wi, Ho, Wo)) return e;
    hipLaunchKernelGGL(k_upsample_fwd, dim3((unsigned)((Wo + 4 * kThreads - 1) / (4 * kThreads)), (unsigned)Ho, (unsigned)NC), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), x, Hi, Wi, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo, y);
    return mas_launch_status();
}

/* requires at most 16 candidate output columns per input column: Wo / Wi <= 6 */
extern "C" int mas_upsample_bilinear_bwd(const float* gy, int64_t NC, int Hi, int Wi, int Ho, int Wo, float* gx, void* stream) {
    if (!gy || !gx) return MAS_ERR_NULL;
    if (int e = check(NC, Hi, Wi, Ho, Wo)) return e;
    if ((long long)Wo > 6LL * Wi) return MAS_ERR_RANGE;
    const dim3 grid((unsigned)((Wi + MAS_WAVE - 1) / MAS_WAVE), (unsigned)((Hi + 3) / 4), (unsigned)NC);
    if (Wo == 4 * Wi && ((uintptr_t)gy & 15) == 0)
        hipLaunchKernelGGL(k_upsample_bwd<true>, grid, dim3(4 * MAS_WAVE), 0, static_cast<hipStream_t>(stream), gy, Hi, Wi, Ho, Wo, (float)Hi / (float)Ho,
                           (float)Wi / (float)Wo, gx);
    else
        hipLaunchKernelGGL(k_upsample_bwd<false>, grid, dim3(4 * MAS_WAVE), 0, static_cast<hipStream_t>(stream), gy, Hi, Wi, Ho, Wo, (float)Hi / (float)Ho,
                           (float)Wi / (float)Wo, gx);
    return mas_launch_status();
}
