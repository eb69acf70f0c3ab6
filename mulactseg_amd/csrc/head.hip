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

template <int K>
int launch_fwd(const float* f, const float* phat, int N, int Ch, int HW, float eps, float* logits, float* inv_norm, hipStream_t st) {
    hipLaunchKernelGGL((k_cosine_fwd<K>), dim3((unsigned)((HW + kThreads - 1) / kThreads), (unsigned)N), dim3(kThreads), 0, st, f, phat, Ch, HW, eps,
                       logits, inv_norm);
    return mas_launch_status();
}
template <int K>
int launch_bwd(const float* f, const float* phat, const float* logits, const float* inv_norm, const float* g, int N, int Ch, int HW,
               float* df, hipStream_t st) {
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
