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
#include <cstdlib>

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

// The same sums from a ZERO-PADDED plane in LDS (dilations D, 2D, 3D with D even -- ASPP's (6, 12, 18) / (12, 24, 36)): no bounds test
// per tap, a thread owns four consecutive pixels of a row and reads every tap as two 8-byte LDS reads at compile-time offsets (the
// column shift (b - 1) d is even, the row pitch and both pads are multiples of four floats), 16-byte global loads and stores.
// k_dw3_fwd<true> spends its time in the 27 x (four compares, a select, an address) per pixel: 463 us per pool batch at 2.3 TB/s;
// this form is at the bytes.  Per accumulator the same products in the same order (a padded tap adds k * 0): the same bits.
// BWD: the input gradient -- the three planes are the three gradients, the taps are mirrored, one output plane.
constexpr int kDwPadC(int d) { return (d + 3) / 4 * 4; }                                        // column pad: d rounded up to four floats
// geometry of padded plane j of the backward form (its own dilation (j + 1) D only) / of the one plane of the forward form (3 D)
struct DwPad { int pr, pc, pw, pp, base; };
__device__ __host__ inline DwPad dw_pad_of(int H, int W, int d, int base) {
    DwPad g;
    g.pr = d; g.pc = kDwPadC(d); g.pw = W + 2 * g.pc; g.pp = (H + 2 * d) * g.pw; g.base = base;
    return g;
}
template <int D, bool BWD>
__global__ __launch_bounds__(kThreads) void k_dw3_pad(const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ x2,
                                                       const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ w2,
                                                       int C, int H, int W, float* __restrict__ y0, float* __restrict__ y1, float* __restrict__ y2) {
    extern __shared__ __attribute__((aligned(16))) float s_pad[];
    constexpr int NPL = BWD ? 3 : 1;
    DwPad g[3];
    g[0] = dw_pad_of(H, W, BWD ? D : 3 * D, 0);
    g[1] = dw_pad_of(H, W, 2 * D, g[0].pp);
    g[2] = dw_pad_of(H, W, 3 * D, g[0].pp + g[1].pp);
    const int total = BWD ? g[2].base + g[2].pp : g[0].pp;              // (every plane a multiple of four floats: W % 4 == 0)
    const size_t plane = (size_t)blockIdx.x * H * W;
    const int c = blockIdx.x % C;
    const float* src[3] = {x0 + plane, BWD ? x1 + plane : nullptr, BWD ? x2 + plane : nullptr};
    float4* s4 = reinterpret_cast<float4*>(s_pad);
    for (int i = threadIdx.x; i < total / 4; i += kThreads) s4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int qrow = W / 4, quads = H * qrow;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
        for (int q = threadIdx.x; q < quads; q += kThreads) {
            const int py = q / qrow, px = (q - py * qrow) * 4;
            *reinterpret_cast<float4*>(s_pad + g[pl].base + (py + g[pl].pr) * g[pl].pw + g[pl].pc + px) = *reinterpret_cast<const float4*>(src[pl] + py * W + px);
        }
    __syncthreads();
    float k[3][9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { k[0][t] = w0[c * 9 + t]; k[1][t] = w1[c * 9 + t]; k[2][t] = w2[c * 9 + t]; }
    typedef float f2 __attribute__((ext_vector_type(2)));
    for (int q = threadIdx.x; q < quads; q += kThreads) {
        const int py = q / qrow, px = (q - py * qrow) * 4;
        const float* ctr[3];                                            // 16-byte aligned
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const DwPad& gg = g[BWD ? j : 0];
            ctr[j] = s_pad + gg.base + (py + gg.pr) * gg.pw + gg.pc + px;
        }
        float4 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int d = (j + 1) * D, pw = g[BWD ? j : 0].pw;
                    const int off = BWD ? -((a - 1) * d) * pw - (b - 1) * d : ((a - 1) * d) * pw + (b - 1) * d;     // (even: 8-byte aligned)
                    const float* tp = ctr[j] + off;
                    const f2 lo = *reinterpret_cast<const f2*>(tp), hi = *reinterpret_cast<const f2*>(tp + 2);
                    float4& A = acc[BWD ? 0 : j];
                    const float kk = k[j][a * 3 + b];
                    A.x = mas_fmaf(kk, lo.x, A.x); A.y = mas_fmaf(kk, lo.y, A.y); A.z = mas_fmaf(kk, hi.x, A.z); A.w = mas_fmaf(kk, hi.y, A.w);
                }
        const size_t o = plane + (size_t)py * W + px;
        *reinterpret_cast<float4*>(y0 + o) = acc[0];
        if (!BWD) {
            *reinterpret_cast<float4*>(y1 + o) = acc[1];
            *reinterpret_cast<float4*>(y2 + o) = acc[2];
        }
    }
}

// The weight gradients of the triple from the same padded plane: one workgroup per channel, the images in turn; a thread owns pixel
// quads, loads its quad of the three gradients (16 bytes each) and adds g_d x (x at tap (a, b) of dilation d) into 27 sums -- two
// 8-byte LDS reads and four fmas per tap instead of four bounds-tested scalar taps through L1.  The reduction tree of k_dw3_bwd_w.
template <int D>
__global__ __launch_bounds__(kThreads) void k_dw3_bwd_w_pad(const float* __restrict__ x, const float* __restrict__ g0, const float* __restrict__ g1,
                                                             const float* __restrict__ g2, int N, int C, int H, int W, float* __restrict__ dw0,
                                                             float* __restrict__ dw1, float* __restrict__ dw2) {
    extern __shared__ __attribute__((aligned(16))) float s_pad[];
    __shared__ float s_red[kThreads / MAS_WAVE][27];
    const DwPad gm = dw_pad_of(H, W, 3 * D, 0);
    const int c = blockIdx.x;
    float acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = 0.f;
    float4* s4 = reinterpret_cast<float4*>(s_pad);
    for (int i = threadIdx.x; i < gm.pp / 4; i += kThreads) s4[i] = make_float4(0.f, 0.f, 0.f, 0.f);       // (the halo stays zero)
    const int qrow = W / 4, quads = H * qrow;
    typedef float f2 __attribute__((ext_vector_type(2)));
    for (int n = 0; n < N; ++n) {
        const size_t plane = ((size_t)n * C + c) * H * W;
        __syncthreads();                                    // (the previous image's taps are read / the zero fill is done)
        for (int q = threadIdx.x; q < quads; q += kThreads) {
            const int py = q / qrow, px = (q - py * qrow) * 4;
            *reinterpret_cast<float4*>(s_pad + (py + gm.pr) * gm.pw + gm.pc + px) = *reinterpret_cast<const float4*>(x + plane + py * W + px);
        }
        __syncthreads();
        for (int q = threadIdx.x; q < quads; q += kThreads) {
            const int py = q / qrow, px = (q - py * qrow) * 4;
            const float* ctr = s_pad + (py + gm.pr) * gm.pw + gm.pc + px;
            const size_t o = plane + (size_t)py * W + px;
            const float4 v[3] = {*reinterpret_cast<const float4*>(g0 + o), *reinterpret_cast<const float4*>(g1 + o), *reinterpret_cast<const float4*>(g2 + o)};
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int d = (j + 1) * D;
                        const float* tp = ctr + ((a - 1) * d) * gm.pw + (b - 1) * d;
                        const f2 lo = *reinterpret_cast<const f2*>(tp), hi = *reinterpret_cast<const f2*>(tp + 2);
                        float s_ = acc[9 * j + a * 3 + b];
                        s_ = mas_fmaf(v[j].x, lo.x, s_); s_ = mas_fmaf(v[j].y, lo.y, s_); s_ = mas_fmaf(v[j].z, hi.x, s_); s_ = mas_fmaf(v[j].w, hi.y, s_);
                        acc[9 * j + a * 3 + b] = s_;
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

// does the padded form take this launch?  D = d0 even, (d1, d2) = (2 d0, 3 d0), rows of whole 16-byte groups, LDS for the padded plane(s)
inline size_t dw3_pad_bytes(int H, int W, int D, bool bwd) {
    if (!bwd) return sizeof(float) * (size_t)dw_pad_of(H, W, 3 * D, 0).pp;
    return sizeof(float) * ((size_t)dw_pad_of(H, W, D, 0).pp + dw_pad_of(H, W, 2 * D, 0).pp + dw_pad_of(H, W, 3 * D, 0).pp);
}
inline bool dw3_pad_ok(const void* const* ptrs, int nptr, int H, int W, int d0, int d1, int d2, bool bwd) {
    if ((d0 != 6 && d0 != 12) || d1 != 2 * d0 || d2 != 3 * d0 || W % 4 != 0) return false;
    for (int i = 0; i < nptr; ++i)
        if ((uintptr_t)ptrs[i] % 16 != 0) return false;
    return dw3_pad_bytes(H, W, d0, bwd) <= 80 * 1024;
}
template <int D, bool BWD>
int dw3_pad_launch(const float* x0, const float* x1, const float* x2, const float* w0, const float* w1, const float* w2, int N, int C, int H, int W,
                   float* y0, float* y1, float* y2, hipStream_t st) {
    const size_t smem = dw3_pad_bytes(H, W, D, BWD);
    if (smem > 64 * 1024) {
        static bool raised[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        if (dev < 0 || dev >= 64 || !raised[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dw3_pad<D, BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0 && dev < 64) raised[dev] = true;
        }
    }
    hipLaunchKernelGGL((k_dw3_pad<D, BWD>), dim3((unsigned)(N * C)), dim3(kThreads), smem, st, x0, x1, x2, w0, w1, w2, C, H, W, y0, y1, y2);
    return mas_launch_status();
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


// ------------------------------------------------------------------------------------------------
// Single depthwise 3x3 (stride 1, padding = dilation): the decoder's two separable convolutions
// (models/segmentation/deeplabv3.py:107-112 after convert_to_separable_conv; 304 and 256 channels at 192x192), for which
// MIOpen falls back to its "naive" fp32 kernels (0.48 ms forward, 0.8 ms backward each).  A workgroup stages a
// zero-padded (16 + 2d) x (W + 2d) strip of one (n, c) plane in LDS; the inner loop has no bounds checks.
// FLIP = data gradient (correlation with the flipped kernel).
// ------------------------------------------------------------------------------------------------
constexpr int kStripH = 16;

template <bool FLIP>
__global__ __launch_bounds__(kThreads) void k_dw_strip(const float* __restrict__ x, const float* __restrict__ w, int C, int H, int W, int d,
                                                        int strips, float* __restrict__ y) {
    extern __shared__ float s_tile[];
    const int strip = blockIdx.x % strips;
    const int nc = blockIdx.x / strips;
    const int c = nc % C;
    const size_t plane = (size_t)nc * H * W;
    const int y0 = strip * kStripH;
    const int TW = W + 2 * d, TH = kStripH + 2 * d;
    for (int i = threadIdx.x; i < TH * TW; i += kThreads) {
        const int ty = i / TW, tx = i - ty * TW;
        const int gy = y0 + ty - d, gx = tx - d;
        s_tile[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[plane + (size_t)gy * W + gx] : 0.0f;
    }
    float k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = w[c * 9 + (FLIP ? 8 - t : t)];
    __syncthreads();
    const int rows = (H - y0) < kStripH ? (H - y0) : kStripH;
    for (int i = threadIdx.x; i < rows * W; i += kThreads) {
        const int py = i / W, px = i - py * W;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) acc = mas_fmaf(k[a * 3 + b], s_tile[(py + a * d) * TW + px + b * d], acc);
        y[plane + (size_t)(y0 + py) * W + px] = acc;
    }
}

// Undilated form (the decoder's two separable convolutions, d = 1) without LDS: a lane owns four consecutive columns and walks
// down kRowsPerLane output rows with a sliding window of three input rows in registers -- per output row ONE 16-byte load plus
// the two neighbouring scalars (same cache lines), 36 fmas, one 16-byte store; no integer divisions, no barrier.  The
// accumulation order per output is the strip kernel's (taps row-major), so both give the same bits.
// grid: (ceil(W / 256), ceil(H / kRowsPerLane), N*C), block: one wave per 256 columns x 4 row groups... (64 x 4 threads).
constexpr int kRowsPerLane = 16;

// ALIGNED: W % 4 == 0 and 16-byte aligned tensors.  !ALIGNED (the 193 x 193 decoder plane of the 769 crop: every row starts at another
// 4-byte alignment): the same 16-byte accesses at 4-byte aligned addresses (gfx950 runs them at the aligned rate,
// tools/micro/unaligned.hip); the last group of a row, when it is not whole, goes element by element.  (That plane used to take the
// LDS-strip kernel: 501 vs ~340 us per training step for the four calls.)
typedef float dw_v4u __attribute__((ext_vector_type(4), aligned(4)));

template <bool FLIP, bool ALIGNED>
__global__ __launch_bounds__(kThreads) void k_dw_rows(const float* __restrict__ x, const float* __restrict__ w, int C, int H, int W,
                                                       float* __restrict__ y) {
    const int lane = threadIdx.x & (MAS_WAVE - 1), wave = threadIdx.x / MAS_WAVE;
    const int x0 = (blockIdx.x * MAS_WAVE + lane) * 4;
    const int ys = (blockIdx.y * (kThreads / MAS_WAVE) + wave) * kRowsPerLane;
    if (x0 >= W || ys >= H) return;
    const size_t nc = blockIdx.z;
    const int c = (int)(nc % C);
    const float* xp = x + nc * H * W;
    float* yp = y + nc * H * W;
    float k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = w[c * 9 + (FLIP ? 8 - t : t)];
    // r[j][0..5] = input row (ys - 1 + row index), columns x0 - 1 .. x0 + 4 (zero outside the plane)
    float r[3][6];
    auto load_row = [&](int gy, float (&dst)[6]) {
        if (gy >= 0 && gy < H) {
            const float* p = xp + (size_t)gy * W + x0;
            dst[0] = x0 > 0 ? p[-1] : 0.0f;
            if (ALIGNED) {
                const float4 v = *reinterpret_cast<const float4*>(p);
                dst[1] = v.x; dst[2] = v.y; dst[3] = v.z; dst[4] = v.w;
            } else if (x0 + 4 <= W) {
                const dw_v4u v = *reinterpret_cast<const dw_v4u*>(p);
                dst[1] = v[0]; dst[2] = v[1]; dst[3] = v[2]; dst[4] = v[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[1 + j] = x0 + j < W ? p[j] : 0.0f;
            }
            dst[5] = x0 + 4 < W ? p[4] : 0.0f;
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) dst[j] = 0.0f;
        }
    };
    load_row(ys - 1, r[0]);
    load_row(ys, r[1]);
    const int ye = (ys + kRowsPerLane) < H ? (ys + kRowsPerLane) : H;
    for (int gy = ys; gy < ye; ++gy) {
        load_row(gy + 1, r[2]);
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) acc = mas_fmaf(k[a * 3 + b], r[a][j + b], acc);
            o[j] = acc;
        }
        float* q = yp + (size_t)gy * W + x0;
        if (ALIGNED) {
            *reinterpret_cast<float4*>(q) = make_float4(o[0], o[1], o[2], o[3]);
        } else if (x0 + 4 <= W) {
            *reinterpret_cast<dw_v4u*>(q) = (dw_v4u){o[0], o[1], o[2], o[3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x0 + j < W) q[j] = o[j];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) { r[0][j] = r[1][j]; r[1][j] = r[2][j]; }
    }
}

// dw[c,a,b] = sum_{n,i,j} dy[n,c,i,j] * x[n,c, i+(a-1)d, j+(b-1)d]; one workgroup per (channel, image), nine sums per
// thread, fixed-order wave/workgroup reduction into part[n,c,9]; k_dw_wsum adds the images in index order.
__global__ __launch_bounds__(kThreads) void k_dw_bwd_w(const float* __restrict__ x, const float* __restrict__ g, int C, int H, int W, int d,
                                                        float* __restrict__ part) {
    __shared__ float s_red[kThreads / MAS_WAVE][9];
    const size_t plane = (size_t)blockIdx.x * H * W;           // blockIdx.x = n*C + c
    const float* xp = x + plane;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (int i = threadIdx.x; i < H * W; i += kThreads) {
        const int py = i / W, px = i - py * W;
        const float v = g[plane + i];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[a * 3 + b] = mas_fmaf(v, tap(xp, H, W, py + (a - 1) * d, px + (b - 1) * d), acc[a * 3 + b]);
    }
    const int lane = threadIdx.x & (MAS_WAVE - 1), wave = threadIdx.x / MAS_WAVE;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float v = acc[t];
#pragma unroll
        for (int off = MAS_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, MAS_WAVE);
        if (lane == 0) s_red[wave][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9)
        part[(size_t)blockIdx.x * 9 + threadIdx.x] =
            ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
}

// The undilated form (the decoder's two separable convolutions) the way k_dw_rows walks a plane: a thread owns four consecutive
// columns and a run of rows, keeps three input rows (columns x0 - 1 .. x0 + 4) in registers and adds g[row] x (shifted x rows) into
// its nine sums -- per row two 16-byte loads and two scalars instead of 36 bounds-tested taps through L1 (k_dw_bwd_w: 325 us per
// training step at 2 TB/s).  One workgroup per (n, c) plane as before; the column groups of a row sit in consecutive lanes, the
// plane's rows are dealt to the 256 / ceil(W / 4) row groups in contiguous runs.  Same reduction tree over the threads.
template <bool ALIGNED>
__global__ __launch_bounds__(kThreads) void k_dw_bwd_w_rows(const float* __restrict__ x, const float* __restrict__ g, int C, int H, int W,
                                                             float* __restrict__ part) {
    __shared__ float s_red[kThreads / MAS_WAVE][9];
    const size_t plane = (size_t)blockIdx.x * H * W;           // blockIdx.x = n*C + c
    const float* xp = x + plane;
    const float* gp = g + plane;
    const int qw = (W + 3) / 4, groups = kThreads / qw;         // (qw <= 256: checked by the host)
    const int rg = threadIdx.x / qw, x0 = (threadIdx.x - rg * qw) * 4;
    const int per = (H + groups - 1) / groups;
    const int ys = rg * per, ye = (ys + per) < H ? (ys + per) : H;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    if (rg < groups && ys < H) {
        float r[3][6];
        auto load4 = [&](const float* p, float (&dst)[4]) {
            if (ALIGNED) {
                const float4 v = *reinterpret_cast<const float4*>(p);
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            } else if (x0 + 4 <= W) {
                const dw_v4u v = *reinterpret_cast<const dw_v4u*>(p);
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[j] = x0 + j < W ? p[j] : 0.0f;
            }
        };
        auto load_row = [&](int gy, float (&dst)[6]) {
            if (gy >= 0 && gy < H) {
                const float* p = xp + (size_t)gy * W + x0;
                float m[4];
                load4(p, m);
                dst[0] = x0 > 0 ? p[-1] : 0.0f;
                dst[1] = m[0]; dst[2] = m[1]; dst[3] = m[2]; dst[4] = m[3];
                dst[5] = x0 + 4 < W ? p[4] : 0.0f;
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) dst[j] = 0.0f;
            }
        };
        load_row(ys - 1, r[0]);
        load_row(ys, r[1]);
        for (int gy = ys; gy < ye; ++gy) {
            load_row(gy + 1, r[2]);
            float v[4];
            load4(gp + (size_t)gy * W + x0, v);
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[a * 3 + b] = mas_fmaf(v[j], r[a][j + b], acc[a * 3 + b]);
#pragma unroll
            for (int j = 0; j < 6; ++j) { r[0][j] = r[1][j]; r[1][j] = r[2][j]; }
        }
    }
    const int lane = threadIdx.x & (MAS_WAVE - 1), wave = threadIdx.x / MAS_WAVE;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float v = acc[t];
#pragma unroll
        for (int off = MAS_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, MAS_WAVE);
        if (lane == 0) s_red[wave][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9)
        part[(size_t)blockIdx.x * 9 + threadIdx.x] =
            ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
}

__global__ __launch_bounds__(kThreads) void k_dw_wsum(const float* __restrict__ part, int N, int C9, float* __restrict__ dw) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= C9) return;
    float v = 0.f;
    for (int n = 0; n < N; ++n) v += part[(size_t)n * C9 + i];
    dw[i] = v;
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
    {
        const void* ptrs[4] = {x, y0, y1, y2};
        if (dw3_pad_ok(ptrs, 4, H, W, d0, d1, d2, false))
            return d0 == 6 ? dw3_pad_launch<6, false>(x, nullptr, nullptr, w0, w1, w2, N, C, H, W, y0, y1, y2, st)
                           : dw3_pad_launch<12, false>(x, nullptr, nullptr, w0, w1, w2, N, C, H, W, y0, y1, y2, st);
    }
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
    {
        // (the padded form of the input gradient is SLOWER than the bounds-tested one on the 48 x 48 planes of the training crop --
        //  173 against 139 us: three planes to stage for one to write -- and is kept for measurements only: MAS_DW3_BWD_PAD=1)
        static const bool use_pad = [] { const char* e = getenv("MAS_DW3_BWD_PAD"); return e && e[0] == '1'; }();
        const void* ptrs[4] = {g0, g1, g2, dx};
        if (use_pad && dw3_pad_ok(ptrs, 4, H, W, d0, d1, d2, true))
            return d0 == 6 ? dw3_pad_launch<6, true>(g0, g1, g2, w0, w1, w2, N, C, H, W, dx, nullptr, nullptr, st)
                           : dw3_pad_launch<12, true>(g0, g1, g2, w0, w1, w2, N, C, H, W, dx, nullptr, nullptr, st);
    }
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
    {
        const void* ptrs[4] = {x, g0, g1, g2};
        if (dw3_pad_ok(ptrs, 4, H, W, d0, d1, d2, false) && dw3_pad_bytes(H, W, d0, false) <= 60 * 1024) {
            const size_t smem = dw3_pad_bytes(H, W, d0, false);
            if (d0 == 6) hipLaunchKernelGGL(k_dw3_bwd_w_pad<6>, dim3((unsigned)C), dim3(kThreads), smem, static_cast<hipStream_t>(stream), x, g0, g1, g2, N, C, H, W, dw0, dw1, dw2);
            else hipLaunchKernelGGL(k_dw3_bwd_w_pad<12>, dim3((unsigned)C), dim3(kThreads), smem, static_cast<hipStream_t>(stream), x, g0, g1, g2, N, C, H, W, dw0, dw1, dw2);
            return mas_launch_status();
        }
    }
    hipLaunchKernelGGL(k_dw3_bwd_w, dim3((unsigned)C), dim3(kThreads), 0, static_cast<hipStream_t>(stream), x, g0, g1, g2, N, C, H, W, d0,
                       d1, d2, dw0, dw1, dw2);
    return mas_launch_status();
}

static int dw_launch(bool flip, const float* x, const float* w, int N, int C, int H, int W, int d, float* y, void* stream) {
    if (!x || !w || !y) return MAS_ERR_NULL;
    if (int e = check(N, C, H, W, d, d, d)) return e;
    const int strips = (H + kStripH - 1) / kStripH;
    const size_t smem = sizeof(float) * (size_t)(kStripH + 2 * d) * (W + 2 * d);
    if (smem > 64 * 1024 || (long long)N * C * strips > 0x7fffffffLL) return MAS_ERR_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d == 1 && W >= 16 && (((uintptr_t)x | (uintptr_t)y) & 3) == 0 && (long long)N * C <= 65535) {
        const dim3 grid((unsigned)((W + 4 * MAS_WAVE - 1) / (4 * MAS_WAVE)),
                        (unsigned)((H + kRowsPerLane * (kThreads / MAS_WAVE) - 1) / (kRowsPerLane * (kThreads / MAS_WAVE))), (unsigned)(N * C));
        const bool aligned = W % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
        if (aligned) {
            if (flip) hipLaunchKernelGGL((k_dw_rows<true, true>), grid, dim3(kThreads), 0, st, x, w, C, H, W, y);
            else hipLaunchKernelGGL((k_dw_rows<false, true>), grid, dim3(kThreads), 0, st, x, w, C, H, W, y);
        } else {
            if (flip) hipLaunchKernelGGL((k_dw_rows<true, false>), grid, dim3(kThreads), 0, st, x, w, C, H, W, y);
            else hipLaunchKernelGGL((k_dw_rows<false, false>), grid, dim3(kThreads), 0, st, x, w, C, H, W, y);
        }
        return mas_launch_status();
    }
    if (flip)
        hipLaunchKernelGGL((k_dw_strip<true>), dim3((unsigned)(N * C * strips)), dim3(kThreads), smem, st, x, w, C, H, W, d, strips, y);
    else
        hipLaunchKernelGGL((k_dw_strip<false>), dim3((unsigned)(N * C * strips)), dim3(kThreads), smem, st, x, w, C, H, W, d, strips, y);
    return mas_launch_status();
}

extern "C" int mas_depthwise3x3_fwd(const float* x, const float* w, int N, int C, int H, int W, int dilation, float* y, void* stream) {
    return dw_launch(false, x, w, N, C, H, W, dilation, y, stream);
}

extern "C" int mas_depthwise3x3_bwd_x(const float* dy, const float* w, int N, int C, int H, int W, int dilation, float* dx, void* stream) {
    return dw_launch(true, dy, w, N, C, H, W, dilation, dx, stream);
}

extern "C" int mas_depthwise3x3_bwd_w(const float* x, const float* dy, int N, int C, int H, int W, int dilation, float* partial,
                                      float* dw, void* stream) {
    if (!x || !dy || !partial || !dw) return MAS_ERR_NULL;
    if (int e = check(N, C, H, W, dilation, dilation, dilation)) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dilation == 1 && W >= 16 && W <= 4 * kThreads && (((uintptr_t)x | (uintptr_t)dy) & 3) == 0) {
        const bool aligned = W % 4 == 0 && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0;
        if (aligned) hipLaunchKernelGGL(k_dw_bwd_w_rows<true>, dim3((unsigned)(N * C)), dim3(kThreads), 0, st, x, dy, C, H, W, partial);
        else hipLaunchKernelGGL(k_dw_bwd_w_rows<false>, dim3((unsigned)(N * C)), dim3(kThreads), 0, st, x, dy, C, H, W, partial);
    } else
        hipLaunchKernelGGL(k_dw_bwd_w, dim3((unsigned)(N * C)), dim3(kThreads), 0, st, x, dy, C, H, W, dilation, partial);
    hipLaunchKernelGGL(k_dw_wsum, dim3((unsigned)((C * 9 + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, partial, N, C * 9, dw);
    return mas_launch_status();
}
