// stem.hip -- the first convolution of the deep stem: 3 -> C channels, 3x3, stride 2, padding 1, with the inference BatchNorm +
// ReLU that follows it (models/segmentation/backbone/resnet.py:163-171: conv1[0], conv1[1], conv1[2]).
//
// K = 27 is too short for the matrix cores (one 32x32x2 MFMA step per 2 of 27 k values, and the B operand would be a gather),
// and the layer is bound by its output: [4,64,512,1024] = 537 MB written for 100 MB read.  A lane owns four consecutive output
// columns and 16 output channels: per input row it loads two aligned 16-byte vectors and one scalar per input channel (81
// values for the 3 x 3 x 3 window of its four outputs), the weights are wave-uniform scalar operands of packed fmas, the
// epilogue applies scale / shift / ReLU and stores one 16-byte vector per channel.  MIOpen runs this layer as a Winograd
// stride-2 kernel (0.29 ms at [4,3,1024,2048]) plus a separate BatchNorm pass.
#include "common.h"

namespace {
constexpr int kThreads = 256;
constexpr int kCg = 16;          // output channels per workgroup

__global__ __launch_bounds__(kThreads) void k_stem_conv(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int H, int W, int Cout, int Ho, int Wo, int relu,
                                                         float* __restrict__ y) {
    const int ox0 = (blockIdx.x * kThreads + threadIdx.x) * 4;
    if (ox0 >= Wo) return;
    const int oy = blockIdx.y;
    const int groups = Cout / kCg;
    const int n = blockIdx.z / groups, cg = blockIdx.z - n * groups;
    const float* xb = x + (size_t)n * 3 * H * W;
    // in[c][r][0..8]: input columns 2*ox0 - 1 .. 2*ox0 + 7 of rows 2*oy - 1 + r (zero outside the picture)
    float in[3][3][9];
    const int ix0 = 2 * ox0;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = 2 * oy - 1 + r;
            if (iy >= 0 && iy < H) {
                const float* p = xb + ((size_t)c * H + iy) * W + ix0;
                const float4 a = *reinterpret_cast<const float4*>(p);
                const float4 b = (ix0 + 4 < W) ? *reinterpret_cast<const float4*>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                in[c][r][0] = ix0 > 0 ? p[-1] : 0.0f;
                in[c][r][1] = a.x; in[c][r][2] = a.y; in[c][r][3] = a.z; in[c][r][4] = a.w;
                in[c][r][5] = b.x; in[c][r][6] = b.y; in[c][r][7] = b.z; in[c][r][8] = b.w;
            } else {
#pragma unroll
                for (int j = 0; j < 9; ++j) in[c][r][j] = 0.0f;
            }
        }
    const float* wg = w + (size_t)cg * kCg * 27;
    const size_t plane = (size_t)Ho * Wo;
    float* yb = y + ((size_t)n * Cout + cg * kCg) * plane + (size_t)oy * Wo + ox0;
    const float lo = relu ? 0.0f : -INFINITY;
#pragma unroll 4
    for (int m = 0; m < kCg; ++m) {
        // output j (j = 0..3) reads input columns 2j + s (s = 0..2) of the window; accumulation in (c, r, s) order, as a direct sum
        mas_v2f a01 = mas_splat(0.f), a23 = mas_splat(0.f);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) {
                    const mas_v2f wv = mas_splat(wg[m * 27 + (c * 3 + r) * 3 + s2]);
                    a01 = mas_pk_fma((mas_v2f){in[c][r][s2], in[c][r][2 + s2]}, wv, a01);
                    a23 = mas_pk_fma((mas_v2f){in[c][r][4 + s2], in[c][r][6 + s2]}, wv, a23);
                }
        const float sc = scale ? scale[cg * kCg + m] : 1.0f, sh = scale ? shift[cg * kCg + m] : 0.0f;
        float o[4] = {mas_fmaf(a01.x, sc, sh), mas_fmaf(a01.y, sc, sh), mas_fmaf(a23.x, sc, sh), mas_fmaf(a23.y, sc, sh)};
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = o[j] < lo ? lo : o[j];
        *reinterpret_cast<float4*>(yb + (size_t)m * plane) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
}  // namespace

extern "C" int mas_stem_conv_fwd(const float* x, const float* w, int N, int H, int W, int Cout, const float* scale, const float* shift, int relu,
                                 float* y, void* stream) {
    if (!x || !w || !y) return MAS_ERR_NULL;
    if ((scale == nullptr) != (shift == nullptr)) return MAS_ERR_NULL;
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % kCg != 0) return MAS_ERR_SHAPE;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (W % 8 != 0 || (((uintptr_t)x | (uintptr_t)y) & 15) != 0) return MAS_ERR_ALIGN;        // 16-byte vectors in and out
    if ((long long)N * (Cout / kCg) > 65535 || Ho > 65535) return MAS_ERR_SHAPE;
    const dim3 grid((unsigned)((Wo / 4 + kThreads - 1) / kThreads), (unsigned)Ho, (unsigned)(N * (Cout / kCg)));
    hipLaunchKernelGGL(k_stem_conv, grid, dim3(kThreads), 0, static_cast<hipStream_t>(stream), x, w, scale, shift, H, W, Cout, Ho, Wo, relu, y);
    return mas_launch_status();
}
