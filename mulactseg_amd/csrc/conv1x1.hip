// conv1x1.hip -- 1x1 convolution for the small-K layers of the stem / layer1 (K = 64 ... 256 input channels at the highest
// resolutions), optionally fused with the inference BatchNorm + ReLU + residual add that follows it
// (models/segmentation/backbone/resnet.py:143-160).
//
// rocBLAS (through MIOpen's GemmFwd1x1) runs these shapes at 33-78 TFLOP/s and 1.5-3 TB/s: neither bound
// (tools/conv1x1_probe.py).  On gfx950 the packed-f32 VALU rate equals the f32 MFMA rate (157 TFLOP/s), so the kernel is
// plain register tiling: a lane owns four consecutive pixels and MT = 32 output channels (64 packed accumulators), walks
// the K input channels with one coalesced 16-B load each, and takes the weights as wave-uniform (scalar) operands from a
// [K, M] transposed copy.  Output-channel tiles of the same pixel tile are mapped to the same XCD so that their re-reads
// of the input hit that XCD's L2.
#include "common.h"

namespace {
constexpr int kThreads = 256;
constexpr int kMT = 32;

// grid: 8 * ceil(ptiles / 8) * mtiles workgroups; bid = xcd + 8 * (m + mtiles * q), pixel tile = q * 8 + xcd
template <bool EPI>
__global__ __launch_bounds__(kThreads, 2) void k_conv1x1(const float* __restrict__ x, const float* __restrict__ wt, int K, int M, int HW,
                                                          int ptiles, int mtiles, const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ res, int relu, float* __restrict__ y) {
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int r = bid >> 3;
    const int m = r % mtiles;
    const int pt = (r / mtiles) * 8 + xcd;
    if (pt >= ptiles) return;
    const size_t n = blockIdx.y;
    const int p = pt * (kThreads * 4) + threadIdx.x * 4;
    if (p >= HW) return;
    const int m0 = m * kMT;
    const float* xp = x + n * K * HW + p;
    mas_v2f a[kMT], b[kMT];
#pragma unroll
    for (int j = 0; j < kMT; ++j) { a[j] = mas_splat(0.f); b[j] = mas_splat(0.f); }
    // K % 4 == 0 (launcher): four input channels per trip, their loads issued together so that three are in flight while
    // the first is consumed; the weights of a trip (4 x 32 floats) arrive as scalar loads
    for (int k = 0; k < K; k += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(xp + (size_t)(k + u) * HW);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const mas_v2f va = {v[u].x, v[u].y}, vb = {v[u].z, v[u].w};
            const float* wk = wt + (size_t)(k + u) * M + m0;
#pragma unroll
            for (int j = 0; j < kMT; ++j) {
                const mas_v2f w = mas_splat(wk[j]);
                a[j] = mas_pk_fma(va, w, a[j]);
                b[j] = mas_pk_fma(vb, w, b[j]);
            }
        }
    }
    float* yp = y + (n * M + m0) * HW + p;
    const float* rp = EPI && res ? res + (n * M + m0) * HW + p : nullptr;
#pragma unroll
    for (int j = 0; j < kMT; ++j) {
        if (m0 + j < M) {
            float4 o = make_float4(a[j].x, a[j].y, b[j].x, b[j].y);
            if (EPI) {
                const float s = scale[m0 + j], t = shift[m0 + j];
                o.x = o.x * s + t; o.y = o.y * s + t; o.z = o.z * s + t; o.w = o.w * s + t;
                if (rp) {
                    const float4 q = *reinterpret_cast<const float4*>(rp + (size_t)j * HW);
                    o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
                }
                if (relu) { o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f; }
            }
            *reinterpret_cast<float4*>(yp + (size_t)j * HW) = o;
        }
    }
}
}  // namespace

extern "C" int mas_conv1x1_fwd(const float* x, const float* w_t, int N, int K, int M, int HW, const float* scale, const float* shift,
                               const float* residual, int relu, float* y, void* stream) {
    if (!x || !w_t || !y) return MAS_ERR_NULL;
    if ((scale == nullptr) != (shift == nullptr)) return MAS_ERR_NULL;
    if (N <= 0 || N > 65535 || K <= 0 || M <= 0 || HW <= 0 || (HW & 3) != 0 || (M % kMT) != 0 || (K & 3) != 0) return MAS_ERR_SHAPE;
    const int ptiles = (HW + kThreads * 4 - 1) / (kThreads * 4);
    const int mtiles = M / kMT;
    const long long nblk = 8LL * ((ptiles + 7) / 8) * mtiles;
    if (nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (scale)
        hipLaunchKernelGGL((k_conv1x1<true>), dim3((unsigned)nblk, (unsigned)N), dim3(kThreads), 0, st, x, w_t, K, M, HW, ptiles, mtiles, scale, shift,
                           residual, relu, y);
    else
        hipLaunchKernelGGL((k_conv1x1<false>), dim3((unsigned)nblk, (unsigned)N), dim3(kThreads), 0, st, x, w_t, K, M, HW, ptiles, mtiles, nullptr,
                           nullptr, nullptr, 0, y);
    return mas_launch_status();
}
