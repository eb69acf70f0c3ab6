// head.hip -- K8: the weight-normalised (cosine) classifier of DeepLabHeadV3PlusWN in one pass over the features.
//
// Reference: models/segmentation/deeplabv3.py:121-124 --  feat = F.normalize(point_feature, dim=1);
// out = F.conv2d(feat, F.normalize(proxy, dim=1))  -- i.e. logits[k] = <f, p_k> / max(|f|, 1e-12) with unit proxies p_k.
// PyTorch runs a norm reduction, a clamp, a broadcast division (which materialises the normalised 256-channel map) and a
// 1x1 convolution: four passes over a [N,256,h,w] tensor.  Here one thread owns one pixel, walks the 256 channels once
// (coalesced: neighbouring lanes are neighbouring pixels), keeps |f|^2 and the K <= 32 dot products in registers; the unit
// proxies are wave-uniform operands.  The backward pass for the features is one more pass:
//     dL/df = ( sum_k g_k p_k  -  (sum_k g_k logit_k) * f / n ) / n ,      n = max(|f|, eps)
// (the proxy gradient is a [K x pixels] x [pixels x 256] GEMM and is left to hipBLASLt through torch.einsum).
#include <cstdlib>

#include "common.h"

namespace {
constexpr int kThreads = 256;

template <int K>
__global__ __launch_bounds__(kThreads) void k_cosine_fwd(const float* __restrict__ f, const float* __restrict__ phat, int Ch, int HW,
                                                          float eps, float* __restrict__ logits, float* __restrict__ inv_norm) {
    const int p = blockIdx.x * kThreads + threadIdx.x;
    const size_t n = blockIdx.y;
    if (p >= HW) return;
    const float* fp = f + n * Ch * HW + p;
    float ss = 0.0f, dot[K];
#pragma unroll
    for (int k = 0; k < K; ++k) dot[k] = 0.0f;
    for (int c = 0; c < Ch; ++c) {
        const float v = fp[(size_t)c * HW];
        ss = mas_fmaf(v, v, ss);
#pragma unroll
        for (int k = 0; k < K; ++k) dot[k] = mas_fmaf(v, phat[k * Ch + c], dot[k]);
    }
    float nrm = sqrtf(ss);
    nrm = nrm < eps ? eps : nrm;
    const float inv = 1.0f / nrm;
    inv_norm[n * HW + p] = inv;
#pragma unroll
    for (int k = 0; k < K; ++k) logits[(n * K + k) * HW + p] = dot[k] * inv;
}

// Four consecutive pixels per lane (one 16-B load per channel), packed-f32 accumulators (v_pk_fma_f32 evaluates both halves
// with IEEE semantics: every pixel's sums are the same fma chain, in channel order, as in k_cosine_fwd), four channels of
// loads in flight per trip.  The one-pixel form above spends its time waiting for a single 4-B load per 21 fmas
// (1.3 TB/s at [4,256,256,512]); this one is bound by the 2 x (K + 1) packed fmas per channel.
template <int K, int CF>                         // CF: channels of loads in flight per trip
__global__ __launch_bounds__(kThreads) void k_cosine_fwd4(const float* __restrict__ f, const float* __restrict__ phat, int Ch, int HW,
                                                           float eps, float* __restrict__ logits, float* __restrict__ inv_norm) {
    const int p = (blockIdx.x * kThreads + threadIdx.x) * 4;
    const size_t n = blockIdx.y;
    if (p >= HW) return;
    const float* fp = f + n * Ch * HW + p;
    mas_v2f ssa = mas_splat(0.f), ssb = mas_splat(0.f), da[K], db[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { da[k] = mas_splat(0.f); db[k] = mas_splat(0.f); }
    for (int c = 0; c < Ch; c += CF) {           // Ch % CF == 0 (launcher)
        float4 v[CF];
#pragma unroll
        for (int u = 0; u < CF; ++u) v[u] = *reinterpret_cast<const float4*>(fp + (size_t)(c + u) * HW);
#pragma unroll
        for (int u = 0; u < CF; ++u) {
            const mas_v2f va = {v[u].x, v[u].y}, vb = {v[u].z, v[u].w};
            ssa = mas_pk_fma(va, va, ssa);
            ssb = mas_pk_fma(vb, vb, ssb);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const mas_v2f w = mas_splat(phat[k * Ch + c + u]);
                da[k] = mas_pk_fma(va, w, da[k]);
                db[k] = mas_pk_fma(vb, w, db[k]);
            }
        }
    }
    float inv[4];
    const float ss[4] = {ssa.x, ssa.y, ssb.x, ssb.y};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float nrm = sqrtf(ss[j]);
        nrm = nrm < eps ? eps : nrm;
        inv[j] = 1.0f / nrm;
    }
    *reinterpret_cast<float4*>(inv_norm + n * HW + p) = make_float4(inv[0], inv[1], inv[2], inv[3]);
#pragma unroll
    for (int k = 0; k < K; ++k)
        *reinterpret_cast<float4*>(logits + (n * K + k) * HW + p) = make_float4(da[k].x * inv[0], da[k].y * inv[1], db[k].x * inv[2], db[k].y * inv[3]);
}

template <int K>
__global__ __launch_bounds__(kThreads) void k_cosine_bwd(const float* __restrict__ f, const float* __restrict__ phat,
                                                          const float* __restrict__ logits, const float* __restrict__ inv_norm,
                                                          const float* __restrict__ g, int Ch, int HW, float* __restrict__ df) {
    const int p = blockIdx.x * kThreads + threadIdx.x;
    const size_t n = blockIdx.y;
    if (p >= HW) return;
    float gk[K];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        gk[k] = g[(n * K + k) * HW + p];
        s = mas_fmaf(gk[k], logits[(n * K + k) * HW + p], s);
    }
    const float inv = inv_norm[n * HW + p];
    const float si = s * inv;
    const float* fp = f + n * Ch * HW + p;
    float* dp = df + n * Ch * HW + p;
    for (int c = 0; c < Ch; ++c) {
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) a = mas_fmaf(gk[k], phat[k * Ch + c], a);
        dp[(size_t)c * HW] = (a - si * fp[(size_t)c * HW]) * inv;
    }
}

// Four consecutive pixels per lane, as k_cosine_fwd4: 16-byte loads / stores, packed-f32 fmas (both halves IEEE: per pixel the same
// operations in the same order as k_cosine_bwd -- the same bits).  The one-pixel form is bound by its 21 scalar fmas per 4-byte
// load and store: 175 us per training step at [4,256,192,192] (1.7 TB/s).
template <int K>
__global__ __launch_bounds__(kThreads) void k_cosine_bwd4(const float* __restrict__ f, const float* __restrict__ phat,
                                                           const float* __restrict__ logits, const float* __restrict__ inv_norm,
                                                           const float* __restrict__ g, int Ch, int HW, float* __restrict__ df) {
    const int p = (blockIdx.x * kThreads + threadIdx.x) * 4;
    const size_t n = blockIdx.y;
    if (p >= HW) return;
    mas_v2f ga[K], gb[K];
    mas_v2f sa = mas_splat(0.f), sb = mas_splat(0.f);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float4 gv = *reinterpret_cast<const float4*>(g + (n * K + k) * HW + p);
        const float4 lv = *reinterpret_cast<const float4*>(logits + (n * K + k) * HW + p);
        ga[k] = (mas_v2f){gv.x, gv.y};
        gb[k] = (mas_v2f){gv.z, gv.w};
        sa = mas_pk_fma(ga[k], (mas_v2f){lv.x, lv.y}, sa);
        sb = mas_pk_fma(gb[k], (mas_v2f){lv.z, lv.w}, sb);
    }
    const float4 iv = *reinterpret_cast<const float4*>(inv_norm + n * HW + p);
    const mas_v2f inva = {iv.x, iv.y}, invb = {iv.z, iv.w};
    const mas_v2f sia = sa * inva, sib = sb * invb;
    const float* fp = f + n * Ch * HW + p;
    float* dp = df + n * Ch * HW + p;
    auto one = [&](int c, const float4& fv) {
        mas_v2f aa = mas_splat(0.f), ab = mas_splat(0.f);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const mas_v2f w = mas_splat(phat[k * Ch + c]);
            aa = mas_pk_fma(ga[k], w, aa);
            ab = mas_pk_fma(gb[k], w, ab);
        }
        const mas_v2f ra = (aa - sia * (mas_v2f){fv.x, fv.y}) * inva, rb = (ab - sib * (mas_v2f){fv.z, fv.w}) * invb;
        *reinterpret_cast<float4*>(dp + (size_t)c * HW) = make_float4(ra.x, ra.y, rb.x, rb.y);
    };
    int c = 0;
    for (; c + 4 <= Ch; c += 4) {               // four channels of loads in flight per trip (the loop is otherwise one load per 40 fmas)
        float4 fv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) fv[u] = *reinterpret_cast<const float4*>(fp + (size_t)(c + u) * HW);
#pragma unroll
        for (int u = 0; u < 4; ++u) one(c + u, fv[u]);
    }
    for (; c < Ch; ++c) one(c, *reinterpret_cast<const float4*>(fp + (size_t)c * HW));
}

template <int K>
int launch_fwd(const float* f, const float* phat, int N, int Ch, int HW, float eps, float* logits, float* inv_norm, hipStream_t st) {
    if (HW % 4 == 0 && Ch % 4 == 0 && (((uintptr_t)f | (uintptr_t)logits | (uintptr_t)inv_norm) & 15) == 0) {
        // eight channels of loads in flight per trip: 174 -> 152 us at [4,256,256,512], 122 -> 118 us at [4,256,192,192] (MAS_COSINE_CF=4: four)
        static const bool cf8 = [] { const char* e = getenv("MAS_COSINE_CF"); return !(e && e[0] == '4'); }();
        if (cf8 && Ch % 8 == 0)
            hipLaunchKernelGGL((k_cosine_fwd4<K, 8>), dim3((unsigned)((HW / 4 + kThreads - 1) / kThreads), (unsigned)N), dim3(kThreads), 0, st, f, phat, Ch,
                               HW, eps, logits, inv_norm);
        else
            hipLaunchKernelGGL((k_cosine_fwd4<K, 4>), dim3((unsigned)((HW / 4 + kThreads - 1) / kThreads), (unsigned)N), dim3(kThreads), 0, st, f, phat, Ch,
                               HW, eps, logits, inv_norm);
        return mas_launch_status();
    }
    hipLaunchKernelGGL((k_cosine_fwd<K>), dim3((unsigned)((HW + kThreads - 1) / kThreads), (unsigned)N), dim3(kThreads), 0, st, f, phat, Ch, HW, eps,
                       logits, inv_norm);
    return mas_launch_status();
}
template <int K>
int launch_bwd(const float* f, const float* phat, const float* logits, const float* inv_norm, const float* g, int N, int Ch, int HW,
               float* df, hipStream_t st) {
    if (HW % 4 == 0 && (((uintptr_t)f | (uintptr_t)logits | (uintptr_t)inv_norm | (uintptr_t)g | (uintptr_t)df) & 15) == 0) {
        hipLaunchKernelGGL((k_cosine_bwd4<K>), dim3((unsigned)((HW / 4 + kThreads - 1) / kThreads), (unsigned)N), dim3(kThreads), 0, st, f, phat,
                           logits, inv_norm, g, Ch, HW, df);
        return mas_launch_status();
    }
    hipLaunchKernelGGL((k_cosine_bwd<K>), dim3((unsigned)((HW + kThreads - 1) / kThreads), (unsigned)N), dim3(kThreads), 0, st, f, phat, logits,
                       inv_norm, g, Ch, HW, df);
    return mas_launch_status();
}
int check(int N, int Ch, int K, int HW) {
    if (N <= 0 || N > 65535 || Ch <= 0 || HW <= 0) return MAS_ERR_SHAPE;
    if (K != 19 && K != 20 && K != 21) return MAS_ERR_CLASSES;
    return 0;
}
}  // namespace

extern "C" int mas_cosine_head_fwd(const float* feat, const float* proxy_hat, int N, int Ch, int K, int HW, float eps, float* logits,
                                   float* inv_norm, void* stream) {
    if (!feat || !proxy_hat || !logits || !inv_norm) return MAS_ERR_NULL;
    if (int e = check(N, Ch, K, HW)) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (K) {
        case 19: return launch_fwd<19>(feat, proxy_hat, N, Ch, HW, eps, logits, inv_norm, st);
        case 20: return launch_fwd<20>(feat, proxy_hat, N, Ch, HW, eps, logits, inv_norm, st);
        default: return launch_fwd<21>(feat, proxy_hat, N, Ch, HW, eps, logits, inv_norm, st);
    }
}

extern "C" int mas_cosine_head_bwd(const float* feat, const float* proxy_hat, const float* logits, const float* inv_norm,
                                   const float* dlogits, int N, int Ch, int K, int HW, float* dfeat, void* stream) {
    if (!feat || !proxy_hat || !logits || !inv_norm || !dlogits || !dfeat) return MAS_ERR_NULL;
    if (int e = check(N, Ch, K, HW)) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (K) {
        case 19: return launch_bwd<19>(feat, proxy_hat, logits, inv_norm, dlogits, N, Ch, HW, dfeat, st);
        case 20: return launch_bwd<20>(feat, proxy_hat, logits, inv_norm, dlogits, N, Ch, HW, dfeat, st);
        default: return launch_bwd<21>(feat, proxy_hat, logits, inv_norm, dlogits, N, Ch, HW, dfeat, st);
    }
}

// ---- the 1x1 convolution of the ASPP image-pooling branch on its 1 x 1 map (deeplabv3.py:194-207): a [N,K] x [K,M] product --------
// N = batch (4), K = 2048, M = 256: 2 MFLOP.  Vendor GEMMs take split-K solutions with atomic adds for this shape (run-to-run
// different low bits, which the BatchNorm over N samples that follows amplifies to 1e-4 of the logits); these three kernels add in
// a fixed order.
namespace {
// y[n,m] = sum_k x[n,k] w[m,k]: one wave per output, lane l adds k = l, l + 64, ... then a fixed butterfly
__global__ __launch_bounds__(256) void k_dense_fwd(const float* __restrict__ x, const float* __restrict__ w, int N, int K, int M, float* __restrict__ y) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= N * M) return;
    const int n = o / M, m = o - n * M;
    const float* xr = x + (size_t)n * K;
    const float* wr = w + (size_t)m * K;
    float a = 0.0f;
    for (int k = lane; k < K; k += 64) a = mas_fmaf(xr[k], wr[k], a);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) y[o] = a;
}
// dx[n,k] = sum_m dy[n,m] w[m,k]: one thread per (n, k), m in order
__global__ __launch_bounds__(256) void k_dense_bwd_x(const float* __restrict__ dy, const float* __restrict__ w, int N, int K, int M, float* __restrict__ dx) {
    const int k = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (k >= K) return;
    float a = 0.0f;
    int m = 0;
    for (; m + 8 <= M; m += 8) {                // eight loads in flight in front of the (single, ordered) fma chain: 97 -> ~20 us at 2048 x 256
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(m + u) * K + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) a = mas_fmaf(dy[(size_t)n * M + m + u], wv[u], a);
    }
    for (; m < M; ++m) a = mas_fmaf(dy[(size_t)n * M + m], w[(size_t)m * K + k], a);
    dx[(size_t)n * K + k] = a;
}
// dw[m,k] = sum_n dy[n,m] x[n,k]: one thread per (m, k), n in order
__global__ __launch_bounds__(256) void k_dense_bwd_w(const float* __restrict__ dy, const float* __restrict__ x, int N, int K, int M, float* __restrict__ dw) {
    const int k = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (k >= K) return;
    float a = 0.0f;
    for (int n = 0; n < N; ++n) a = mas_fmaf(dy[(size_t)n * M + m], x[(size_t)n * K + k], a);
    dw[(size_t)m * K + k] = a;
}
int dense_check(int N, int K, int M) {
    return (N <= 0 || K <= 0 || M <= 0 || N > 65535 || M > 65535 || (long long)N * M > 0x3fffffffLL) ? MAS_ERR_SHAPE : 0;
}
}  // namespace

extern "C" int mas_dense_small_fwd(const float* x, const float* w, int N, int K, int M, float* y, void* stream) {
    if (!x || !w || !y) return MAS_ERR_NULL;
    if (int e = dense_check(N, K, M)) return e;
    hipLaunchKernelGGL(k_dense_fwd, dim3((unsigned)((N * M + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, N, K, M, y);
    return mas_launch_status();
}

extern "C" int mas_dense_small_bwd(const float* dy, const float* x, const float* w, int N, int K, int M, float* dx, float* dw, void* stream) {
    if (!dy || !x || !w) return MAS_ERR_NULL;
    if (int e = dense_check(N, K, M)) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dx) hipLaunchKernelGGL(k_dense_bwd_x, dim3((unsigned)((K + 255) / 256), (unsigned)N), dim3(256), 0, st, dy, w, N, K, M, dx);
    if (dw) hipLaunchKernelGGL(k_dense_bwd_w, dim3((unsigned)((K + 255) / 256), (unsigned)M), dim3(256), 0, st, dy, x, N, K, M, dw);
    return mas_launch_status();
}
