// bn.hip -- BatchNorm2d fused with the ReLU and the residual add that follow it in every block of the reference model
// (models/segmentation/backbone/resnet.py:119-160 Bottleneck: relu(bn(conv)), relu(bn3(conv3) + identity); the stem, the
// ASPP branches and the decoder are conv -> bn -> relu triples, deeplabv3.py:93-110, 216-245).
//
// MIOpen's BatchNorm kernels already run near HBM speed; what this file removes is the TRAFFIC of the separate ops:
// unfused, a block output is written by BN, re-read and re-written by the add, re-read and re-written by ReLU, and the
// backward repeats that (threshold_backward, grad accumulation).  Fused:
//   forward   k_bn_partial  (sum, sum of squares per (n, c, chunk), double)  ->  k_bn_stats (mean, invstd, running stats,
//             num_batches_tracked)  ->  k_bn_apply  y = relu((x - mean) * invstd * gamma + beta + residual)
//             (+ a ReLU mask of one byte per four outputs, so that the backward does not have to re-read y)
//   backward  k_bn_bwd_partial (sum g, sum g * xhat with g = dy * [y > 0])  ->  k_bn_bwd_stats (dgamma, dbeta, means)
//             ->  k_bn_bwd_apply  dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)),  dres = g
//   inference k_bn_apply with the running statistics.
// All reductions run in a fixed order (per-chunk tree, then chunks in index order, in double): results do not depend on
// the launch or on the run.  Layout NCHW; float4 paths when H*W is a multiple of 4.
#include "common.h"

namespace {
constexpr int kThreads = 256;
constexpr int kChunk = kThreads * 4 * 8;      // elements of one plane handled by one workgroup of the reduction kernels

__device__ __forceinline__ double block_sum(double v, double* s_red) {
#pragma unroll
    for (int off = MAS_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, MAS_WAVE);
    const int lane = threadIdx.x & (MAS_WAVE - 1), wave = threadIdx.x / MAS_WAVE;
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}

// grid (chunks, N, C): part[(c * N + n) * chunks + chunk] = (sum x, sum x^2)
__global__ __launch_bounds__(kThreads) void k_bn_partial(const float* __restrict__ x, int C, int HW, int chunks, double2* __restrict__ part) {
    __shared__ double s_red[kThreads / MAS_WAVE];
    const int chunk = blockIdx.x, n = blockIdx.y, c = blockIdx.z;
    const float* p = x + ((size_t)n * C + c) * HW;
    const int lo = chunk * kChunk, hi = min(HW, lo + kChunk);
    float s = 0.f, q = 0.f;
    if ((HW & 3) == 0) {
        for (int i = lo + threadIdx.x * 4; i < hi; i += kThreads * 4) {
            const float4 v = *reinterpret_cast<const float4*>(p + i);
            s += (v.x + v.y) + (v.z + v.w);
            q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += kThreads) { const float v = p[i]; s += v; q += v * v; }
    }
    const double S = block_sum((double)s, s_red);
    const double Q = block_sum((double)q, s_red);
    if (threadIdx.x == 0) part[((size_t)c * gridDim.y + n) * chunks + chunk] = make_double2(S, Q);
}

// one thread per channel: batch mean / biased variance -> mean, invstd; running statistics (momentum, unbiased variance)
__global__ __launch_bounds__(kThreads) void k_bn_stats(const double2* __restrict__ part, int C, int per_channel, double count, float eps,
                                                        float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        long long* __restrict__ num_batches_tracked) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c == 0 && num_batches_tracked) num_batches_tracked[0] += 1;
    if (c >= C) return;
    double S = 0.0, Q = 0.0;
    for (int i = 0; i < per_channel; ++i) { const double2 v = part[(size_t)c * per_channel + i]; S += v.x; Q += v.y; }
    const double m = S / count;
    double var = Q / count - m * m;
    var = var < 0.0 ? 0.0 : var;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
}

// grid (ceil(HW / (4 * 256)), N * C).  FROM_VAR: `stat2` holds the running variance (inference), else invstd.
template <bool FROM_VAR>
__global__ __launch_bounds__(kThreads) void k_bn_apply(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ stat1, const float* __restrict__ stat2, float eps,
                                                        const float* __restrict__ res, int C, int HW, int relu, float* __restrict__ y,
                                                        unsigned char* __restrict__ mask) {
    const int c = blockIdx.y % C;
    const float mu = stat1[c];
    const float is = FROM_VAR ? 1.0f / sqrtf(stat2[c] + eps) : stat2[c];
    const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    const size_t base = (size_t)blockIdx.y * HW;
    const int i = (blockIdx.x * kThreads + threadIdx.x) * 4;
    if (i >= HW) return;
    auto f = [&](float v, float r) {
        float t = ((v - mu) * is) * g + b;
        t = t + r;
        return (relu && !(t > 0.0f)) ? 0.0f : t;
    };
    if ((HW & 3) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(x + base + i);
        const float4 r = res ? *reinterpret_cast<const float4*>(res + base + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 o = make_float4(f(v.x, r.x), f(v.y, r.y), f(v.z, r.z), f(v.w, r.w));
        *reinterpret_cast<float4*>(y + base + i) = o;
        // ReLU mask for the backward pass: one byte per four elements (1/16 of the bytes of y)
        if (mask) mask[(base + i) >> 2] = (unsigned char)((o.x > 0.f) | ((o.y > 0.f) << 1) | ((o.z > 0.f) << 2) | ((o.w > 0.f) << 3));
    } else {
        for (int k = 0; k < 4 && i + k < HW; ++k) y[base + i + k] = f(x[base + i + k], res ? res[base + i + k] : 0.0f);
    }
}

// part[(c * N + n) * chunks + chunk] = (sum g, sum g * xhat),  g = dy * [y > 0] (relu) or dy
__global__ __launch_bounds__(kThreads) void k_bn_bwd_partial(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                              const unsigned char* __restrict__ mask, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, int C, int HW, int chunks, int relu,
                                                              double2* __restrict__ part) {
    __shared__ double s_red[kThreads / MAS_WAVE];
    const int chunk = blockIdx.x, n = blockIdx.y, c = blockIdx.z;
    const size_t base = ((size_t)n * C + c) * HW;
    const float mu = mean[c], is = invstd[c];
    const int lo = chunk * kChunk, hi = min(HW, lo + kChunk);
    float s = 0.f, q = 0.f;
    auto acc = [&](float g, float xv, float yv) {
        g = (relu && !(yv > 0.0f)) ? 0.0f : g;
        s += g;
        q += g * ((xv - mu) * is);
    };
    if ((HW & 3) == 0) {
        for (int i = lo + threadIdx.x * 4; i < hi; i += kThreads * 4) {
            const float4 g = *reinterpret_cast<const float4*>(dy + base + i);
            const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
            float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (relu && mask) {
                const unsigned m = mask[(base + i) >> 2];
                yv = make_float4((float)(m & 1u), (float)((m >> 1) & 1u), (float)((m >> 2) & 1u), (float)((m >> 3) & 1u));
            } else if (relu) {
                yv = *reinterpret_cast<const float4*>(y + base + i);
            }
            acc(g.x, xv.x, yv.x); acc(g.y, xv.y, yv.y); acc(g.z, xv.z, yv.z); acc(g.w, xv.w, yv.w);
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += kThreads) acc(dy[base + i], x[base + i], relu ? y[base + i] : 1.0f);
    }
    const double S = block_sum((double)s, s_red);
    const double Q = block_sum((double)q, s_red);
    if (threadIdx.x == 0) part[((size_t)c * gridDim.y + n) * chunks + chunk] = make_double2(S, Q);
}

// dbeta = sum g, dgamma = sum g * xhat; coef[c] = (mean g, mean g*xhat)
__global__ __launch_bounds__(kThreads) void k_bn_bwd_stats(const double2* __restrict__ part, int C, int per_channel, double count,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float2* __restrict__ coef) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= C) return;
    double S = 0.0, Q = 0.0;
    for (int i = 0; i < per_channel; ++i) { const double2 v = part[(size_t)c * per_channel + i]; S += v.x; Q += v.y; }
    if (dbeta) dbeta[c] = (float)S;
    if (dgamma) dgamma[c] = (float)Q;
    coef[c] = make_float2((float)(S / count), (float)(Q / count));
}

__global__ __launch_bounds__(kThreads) void k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                            const unsigned char* __restrict__ mask, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float2* __restrict__ coef, int C, int HW,
                                                            int relu, float* __restrict__ dx, float* __restrict__ dres) {
    const int c = blockIdx.y % C;
    const float mu = mean[c], is = invstd[c];
    const float k = (gamma ? gamma[c] : 1.0f) * is;
    const float2 m = coef[c];
    const size_t base = (size_t)blockIdx.y * HW;
    const int i = (blockIdx.x * kThreads + threadIdx.x) * 4;
    if (i >= HW) return;
    auto f = [&](float g, float xv, float yv, float& gout) {
        g = (relu && !(yv > 0.0f)) ? 0.0f : g;
        gout = g;
        return k * ((g - m.x) - ((xv - mu) * is) * m.y);
    };
    if ((HW & 3) == 0) {
        const float4 g = *reinterpret_cast<const float4*>(dy + base + i);
        const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
        float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (relu && mask) {
            const unsigned mk = mask[(base + i) >> 2];
            yv = make_float4((float)(mk & 1u), (float)((mk >> 1) & 1u), (float)((mk >> 2) & 1u), (float)((mk >> 3) & 1u));
        } else if (relu) {
            yv = *reinterpret_cast<const float4*>(y + base + i);
        }
        float4 r, o;
        o.x = f(g.x, xv.x, yv.x, r.x); o.y = f(g.y, xv.y, yv.y, r.y); o.z = f(g.z, xv.z, yv.z, r.z); o.w = f(g.w, xv.w, yv.w, r.w);
        *reinterpret_cast<float4*>(dx + base + i) = o;
        if (dres) *reinterpret_cast<float4*>(dres + base + i) = r;
    } else {
        for (int t = 0; t < 4 && i + t < HW; ++t) {
            float r;
            dx[base + i + t] = f(dy[base + i + t], x[base + i + t], relu ? y[base + i + t] : 1.0f, r);
            if (dres) dres[base + i + t] = r;
        }
    }
}

int check(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0 || N > 65535 || C > 65535 || (long long)N * C > 0x7fffffffLL) return MAS_ERR_SHAPE;
    return 0;
}
inline int chunks_of(int HW) { return (HW + kChunk - 1) / kChunk; }
inline dim3 apply_grid(int N, int C, int HW) { return dim3((unsigned)((HW + kThreads * 4 - 1) / (kThreads * 4)), (unsigned)(N * C)); }
}  // namespace

extern "C" int64_t mas_bn_workspace_bytes(int N, int C, int HW) {
    if (check(N, C, HW)) return -1;
    return (int64_t)sizeof(double2) * C * N * chunks_of(HW) + (int64_t)sizeof(float2) * C;
}

extern "C" int mas_bn_act_train_fwd(const float* x, const float* gamma, const float* beta, const float* residual, int N, int C, int HW,
                                    float eps, float momentum, int relu, float* running_mean, float* running_var,
                                    int64_t* num_batches_tracked, float* save_mean, float* save_invstd, void* workspace, float* y,
                                    uint8_t* relu_mask, void* stream) {
    if (!x || !save_mean || !save_invstd || !workspace || !y) return MAS_ERR_NULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MAS_ERR_NULL;
    if (int e = check(N, C, HW)) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int chunks = chunks_of(HW);
    double2* part = static_cast<double2*>(workspace);
    hipLaunchKernelGGL(k_bn_partial, dim3((unsigned)chunks, (unsigned)N, (unsigned)C), dim3(kThreads), 0, st, x, C, HW, chunks, part);
    hipLaunchKernelGGL(k_bn_stats, dim3((unsigned)((C + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, part, C, N * chunks,
                       (double)N * (double)HW, eps, momentum, save_mean, save_invstd, running_mean, running_var,
                       reinterpret_cast<long long*>(num_batches_tracked));
    hipLaunchKernelGGL((k_bn_apply<false>), apply_grid(N, C, HW), dim3(kThreads), 0, st, x, gamma, beta, save_mean, save_invstd, eps, residual, C,
                       HW, relu, y, (relu && (HW & 3) == 0) ? relu_mask : nullptr);
    return mas_launch_status();
}

extern "C" int mas_bn_act_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, const float* residual, int N, int C, int HW, float eps, int relu, float* y,
                                   void* stream) {
    if (!x || !running_mean || !running_var || !y) return MAS_ERR_NULL;
    if (int e = check(N, C, HW)) return e;
    hipLaunchKernelGGL((k_bn_apply<true>), apply_grid(N, C, HW), dim3(kThreads), 0, static_cast<hipStream_t>(stream), x, gamma, beta,
                       running_mean, running_var, eps, residual, C, HW, relu, y, nullptr);
    return mas_launch_status();
}

extern "C" int mas_bn_act_train_bwd(const float* dy, const float* x, const float* y, const uint8_t* relu_mask, const float* gamma,
                                    const float* save_mean, const float* save_invstd, int N, int C, int HW, int relu, void* workspace,
                                    float* dx, float* dresidual, float* dgamma, float* dbeta, void* stream) {
    if (!dy || !x || !save_mean || !save_invstd || !workspace || !dx) return MAS_ERR_NULL;
    if ((HW & 3) != 0) relu_mask = nullptr;
    if (relu && !y && !relu_mask) return MAS_ERR_NULL;
    if (int e = check(N, C, HW)) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int chunks = chunks_of(HW);
    double2* part = static_cast<double2*>(workspace);
    float2* coef = reinterpret_cast<float2*>(part + (size_t)C * N * chunks);
    hipLaunchKernelGGL(k_bn_bwd_partial, dim3((unsigned)chunks, (unsigned)N, (unsigned)C), dim3(kThreads), 0, st, dy, x, y, relu_mask, save_mean,
                       save_invstd, C, HW, chunks, relu, part);
    hipLaunchKernelGGL(k_bn_bwd_stats, dim3((unsigned)((C + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, part, C, N * chunks,
                       (double)N * (double)HW, dgamma, dbeta, coef);
    hipLaunchKernelGGL(k_bn_bwd_apply, apply_grid(N, C, HW), dim3(kThreads), 0, st, dy, x, y, relu_mask, gamma, save_mean, save_invstd, coef, C, HW,
                       relu, dx, dresidual);
    return mas_launch_status();
}
