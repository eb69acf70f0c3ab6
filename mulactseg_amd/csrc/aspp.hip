// aspp.hip -- K7: the depthwise half of the three atrous separable convolutions of ASPP, for gfx950.
//
// Reference: models/segmentation/deeplabv3.py:168-201,216-245 after convert_to_separable_conv (:249-261): each ASPP
// branch d in {6, 12, 18} is  depthwise 3x3 (dilation d, padding d, 2048 channels, no bias) -> pointwise 1x1 2048->256.
// The three depthwise convolutions read the SAME 2048-channel feature map; MIOpen runs them as three kernels (two of
// them its "naive" fallback for dilated depthwise fp32).  Here one workgroup stages one (n, c) plane in LDS and
// produces all three dilations from it: the map is read once instead of three times, and the backward data pass reads
// the three gradient planes once to produce one input-gradient plane.  The pointwise 1x1 stays a GEMM (MFMA, hipBLASLt).
//
//   k_dw3_fwd     y_d[n,c] = sum_{a,b} w_d[c,a,b] * x[n,c, i+(a-1)d, j+(b-1)d]      (zero padding)
//   k_dw3_bwd_x   dx[n,c]  = sum_d sum_{a,b} w_d[c,a,b] * dy_d[n,c, i-(a-1)d, j-(b-1)d]
//   k_dw3_bwd_w   dw_d[c,a,b] = sum_{n,i,j} dy_d[n,c,i,j] * x[n,c, i+(a-1)d, j+(b-1)d]   (fixed-order tree: deterministic)
#include "common.h"

namespace {
constexpr int kThreads = 256;

__device__ __forceinline__ float tap(const float* p, int H, int W, int y, int x) {
    return (y >= 0 && y < H && x >= 0 && x < W) ? p[y * W + x] : 0.0f;
}

// LDS: whole plane (USE_LDS) -- H*W*4 bytes; otherwise taps come from global memory through L1/L2
template <bool USE_LDS>
__global__ __launch_bounds__(kThreads) void k_dw3_fwd(const float* __restrict__ x, const float* __restrict__ w0, const float* __restrict__ w1,
                                                       const float* __restrict__ w2, int C, int H, int W, int d0, int d1, int d2,
                                                       float* __restrict__ y0, float* __restrict__ y1, float* __restrict__ y2) {
    extern __shared__ float s_plane[];
    const size_t plane = (size_t)blockIdx.x * H * W;            // blockIdx.x = n*C + c
    const int c = blockIdx.x % C;
    const float* src = x + plane;
    if (USE_LDS) {
        for (int i = threadIdx.x; i < H * W; i += kThreads) s_plane[i] = src[i];
        __syncthreads();
        src = s_plane;
    }
    float k0[9], k1[9], k2[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { k0[t] = w0[c * 9 + t]; k1[t] = w1[c * 9 + t]; k2[t] = w2[c * 9 + t]; }
    for (int i = threadIdx.x; i < H * W; i += kThreads) {
        const int py = i / W, px = i - py * W;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                a0 = mas_fmaf(k0[a * 3 + b], tap(src, H, W, py + (a - 1) * d0, px + (b - 1) * d0), a0);
                a1 = mas_fmaf(k1[a * 3 + b], tap(src, H, W, py + (a - 1) * d1, px + (b - 1) * d1), a1);
                a2 = mas_fmaf(k2[a * 3 + b], tap(src, H, W, py + (a - 1) * d2, px + (b - 1) * d2), a2);
            }
        y0[plane + i] = a0;
        y1[plane + i] = a1;
        y2[plane + i] = a2;
    }
}

template <bool USE_LDS>
__global__ __launch_bounds__(kThreads) void k_dw3_bwd_x(const float* __restrict__ g0, const float* __restrict__ g1, const float* __restrict__ g2,
                                                         const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ w2,
                                                         int C, int H, int W, int d0, int d1, int d2, float* __restrict__ dx) {
    extern __shared__ float s_planes[];
    const size_t plane = (size_t)blockIdx.x * H * W;
    const int c = blockIdx.x % C;
    const float *p0 = g0 + plane, *p1 = g1 + plane, *p2 = g2 + plane;
    if (USE_LDS) {
        for (int i = threadIdx.x; i < H * W; i += kThreads) {
            s_planes[i] = p0[i];
            s_planes[H * W + i] = p1[i];
            s_planes[2 * H * W + i] = p2[i];
        }
        __syncthreads();
        p0 = s_planes; p1 = s_planes + H * W; p2 = s_planes + 2 * H * W;
    }
    float k0[9], k1[9], k2[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { k0[t] = w0[c * 9 + t]; k1[t] = w1[c * 9 + t]; k2[t] = w2[c * 9 + t]; }
    for (int i = threadIdx.x; i < H * W; i += kThreads) {
        const int py = i / W, px = i - py * W;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                acc = mas_fmaf(k0[a * 3 + b], tap(p0, H, W, py - (a - 1) * d0, px - (b - 1) * d0), acc);
                acc = mas_fmaf(k1[a * 3 + b], tap(p1, H, W, py - (a - 1) * d1, px - (b - 1) * d1), acc);
                acc = mas_fmaf(k2[a * 3 + b], tap(p2, H, W, py - (a - 1) * d2, px - (b - 1) * d2), acc);
            }
        dx[plane + i] = acc;
    }
}

// one workgroup per channel; loops over the batch; 27 partial sums per thread, fixed-order reduction
__global__ __launch_bounds__(kThreads) void k_dw3_bwd_w(const float* __restrict__ x, const float* __restrict__ g0, const float* __restrict__ g1,
                                                         const float* __restrict__ g2, int N, int C, int H, int W, int d0, int d1, int d2,
                                                         float* __restrict__ dw0, float* __restrict__ dw1, float* __restrict__ dw2) {
    __shared__ float s_red[kThreads / MAS_WAVE][27];
    const int c = blockIdx.x;
    float acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = 0.f;
    for (int n = 0; n < N; ++n) {
        const size_t plane = ((size_t)n * C + c) * H * W;
        const float* xp = x + plane;
        for (int i = threadIdx.x; i < H * W; i += kThreads) {
            const int py = i / W, px = i - py * W;
            const float v0 = g0[plane + i], v1 = g1[plane + i], v2 = g2[plane + i];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    acc[a * 3 + b] = mas_fmaf(v0, tap(xp, H, W, py + (a - 1) * d0, px + (b - 1) * d0), acc[a * 3 + b]);
                    acc[9 + a * 3 + b] = mas_fmaf(v1, tap(xp, H, W, py + (a - 1) * d1, px + (b - 1) * d1), acc[9 + a * 3 + b]);
                    acc[18 + a * 3 + b] = mas_fmaf(v2, tap(xp, H, W, py + (a - 1) * d2, px + (b - 1) * d2), acc[18 + a * 3 + b]);
                }
        }
    }
    const int lane = threadIdx.x & (MAS_WAVE - 1), wave = threadIdx.x / MAS_WAVE;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        float v = acc[t];
#pragma unroll
        for (int off = MAS_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, MAS_WAVE);
        if (lane == 0) s_red[wave][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        float v = ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
        float* dst = threadIdx.x < 9 ? dw0 : (threadIdx.x < 18 ? dw1 : dw2);
        dst[c * 9 + (threadIdx.x % 9)] = v;
    }
}

int check(int N, int C, int H, int W, int d0, int d1, int d2) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (long long)N * C > 0x7fffffffLL || (long long)H * W > (1 << 24)) return MAS_ERR_SHAPE;
    if (d0 <= 0 || d1 <= 0 || d2 <= 0) return MAS_ERR_RANGE;
    return 0;
}
}  // namespace

extern "C" int mas_aspp_dw3_fwd(const float* x, const float* w0, const float* w1, const float* w2, int N, int C, int H, int W, int d0,
                                int d1, int d2, float* y0, float* y1, float* y2, void* stream) {
    if (!x || !w0 || !w1 || !w2 || !y0 || !y1 || !y2) return MAS_ERR_NULL;
    if (int e = check(N, C, H, W, d0, d1, d2)) return e;
    const size_t smem = sizeof(float) * (size_t)H * W;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (smem <= 64 * 1024)
        hipLaunchKernelGGL((k_dw3_fwd<true>), dim3((unsigned)(N * C)), dim3(kThreads), smem, st, x, w0, w1, w2, C, H, W, d0, d1, d2, y0, y1, y2);
    else
        hipLaunchKernelGGL((k_dw3_fwd<false>), dim3((unsigned)(N * C)), dim3(kThreads), 0, st, x, w0, w1, w2, C, H, W, d0, d1, d2, y0, y1, y2);
    return mas_launch_status();
}

extern "C" int mas_aspp_dw3_bwd_x(const float* g0, const float* g1, const float* g2, const float* w0, const float* w1, const float* w2,
                                  int N, int C, int H, int W, int d0, int d1, int d2, float* dx, void* stream) {
    if (!g0 || !g1 || !g2 || !w0 || !w1 || !w2 || !dx) return MAS_ERR_NULL;
    if (int e = check(N, C, H, W, d0, d1, d2)) return e;
    const size_t smem = sizeof(float) * 3 * (size_t)H * W;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (smem <= 128 * 1024)
        hipLaunchKernelGGL((k_dw3_bwd_x<true>), dim3((unsigned)(N * C)), dim3(kThreads), smem, st, g0, g1, g2, w0, w1, w2, C, H, W, d0, d1, d2, dx);
    else
        hipLaunchKernelGGL((k_dw3_bwd_x<false>), dim3((unsigned)(N * C)), dim3(kThreads), 0, st, g0, g1, g2, w0, w1, w2, C, H, W, d0, d1, d2, dx);
    return mas_launch_status();
}

extern "C" int mas_aspp_dw3_bwd_w(const float* x, const float* g0, const float* g1, const float* g2, int N, int C, int H, int W, int d0,
                                  int d1, int d2, float* dw0, float* dw1, float* dw2, void* stream) {
    if (!x || !g0 || !g1 || !g2 || !dw0 || !dw1 || !dw2) return MAS_ERR_NULL;
    if (int e = check(N, C, H, W, d0, d1, d2)) return e;
    hipLaunchKernelGGL(k_dw3_bwd_w, dim3((unsigned)C), dim3(kThreads), 0, static_cast<hipStream_t>(stream), x, g0, g1, g2, N, C, H, W, d0,
                       d1, d2, dw0, dw1, dw2);
    return mas_launch_status();
}
