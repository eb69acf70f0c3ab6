// optim.hip -- AdamW over ALL parameters of the model in one launch.
//
// Reference: trainer/base.py:64-66 (`optim.AdamW(params=[backbone @ lr, classifier @ cls_lr_scale * lr], weight_decay=...)`), i.e. torch
// 1.11's single-tensor AdamW, per parameter tensor and per step
//     p  <- p * (1 - lr * wd)
//     m  <- m * b1 + g * (1 - b1)
//     v  <- v * b2 + (1 - b2) * (g * g)
//     p  <- p - (lr / (1 - b1^t)) * (m / (sqrt(v) / sqrt(1 - b2^t) + eps))
// in f32, every product and sum rounded separately (no contraction: the file is compiled with -ffp-contract=off).
//
// Why a kernel of its own: the step is HBM-bound bookkeeping -- read p, g, m, v, write p, m, v: 28 bytes per parameter, ~1.1 GB for the
// 40 M parameters of DeepLabv3+/ResNet-50 -- and ATen's multi-tensor form runs it as nine launches of 320-element chunks at ~2.3 TB/s
// (321 us per step, profiles/r05/c_train_768_steady.md).  Here: one launch over a job table (one record per parameter tensor: its
// four pointers, its length, its parameter group, its first block -- static while the gradient buffers keep their addresses; the
// groups' learning rates, which the poly schedule moves every step, travel as kernel arguments), 16-byte accesses, the step count
// and the "skip this step" flag read from device memory (the stream-K give-up word, trainer/base.py:guard_optimizer_step -- no
// host round trip).
#include "common.h"

namespace {
constexpr int kThreads = 256;
constexpr int kPerThread = 8;                       // two float4 per tensor and thread
constexpr int kPerBlock = kThreads * kPerThread;

struct AdamJob {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
    int group;
    unsigned first_block;
};
constexpr int kMaxGroups = 8;

struct AdamArgs {
    const AdamJob* jobs;
    int njobs;
    double beta1, beta2, weight_decay;  // (doubles: 1 - beta, beta^t and 1 - lr wd are formed in double, as Python forms them for torch)
    float eps;
    float lr[kMaxGroups];           // learning rate of every parameter group
    const float* step;              // device scalar: the number of steps taken BEFORE this one (the caller adds 1 afterwards, unless skipped)
    const float* skip;              // NULL, or a device scalar: non-zero -> leave everything untouched
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float decay, float b1, float omb1, float b2, float omb2,
                                         float step_size, float sqrt_bc2, float eps) {
    p = p * decay;
    m = m * b1 + g * omb1;
    v = v * b2 + omb2 * (g * g);                    // (addcmul_ on the GPU: a + value * (t1 * t2), ATen PointwiseOpsKernel.cu)
    const float denom = sqrtf(v) / sqrt_bc2 + eps;
    p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(kThreads) void k_adamw_multi(const AdamArgs a) {
    if (a.skip && *a.skip != 0.0f) return;
    int lo = 0, hi = a.njobs - 1;                   // the last job whose first block is <= this block (wave-uniform search)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.jobs[mid].first_block <= blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const AdamJob jb = a.jobs[lo];
    const double t = (double)*a.step + 1.0;
    const double bc1 = 1.0 - pow(a.beta1, t), bc2 = 1.0 - pow(a.beta2, t);
    const float lr = a.lr[jb.group];
    const float step_size = (float)((double)lr / bc1);
    const float sqrt_bc2 = (float)sqrt(bc2);        // torch: denom = sqrt(v) / math.sqrt(bias_correction2) + eps
    const float decay = (float)(1.0 - (double)lr * a.weight_decay);
    const float b1 = (float)a.beta1, omb1 = (float)(1.0 - a.beta1), b2 = (float)a.beta2, omb2 = (float)(1.0 - a.beta2);
    const long long base = (long long)(blockIdx.x - jb.first_block) * kPerBlock;
#pragma unroll
    for (int r = 0; r < kPerThread / 4; ++r) {
        const long long i = base + ((long long)r * kThreads + threadIdx.x) * 4;
        if (i + 4 <= jb.n && (((uintptr_t)(jb.p + i) | (uintptr_t)(jb.g + i) | (uintptr_t)(jb.m + i) | (uintptr_t)(jb.v + i)) & 15) == 0) {
            float4 p = *reinterpret_cast<const float4*>(jb.p + i);
            const float4 g = *reinterpret_cast<const float4*>(jb.g + i);
            float4 m = *reinterpret_cast<const float4*>(jb.m + i);
            float4 v = *reinterpret_cast<const float4*>(jb.v + i);
            adam_one(p.x, g.x, m.x, v.x, decay, b1, omb1, b2, omb2, step_size, sqrt_bc2, a.eps);
            adam_one(p.y, g.y, m.y, v.y, decay, b1, omb1, b2, omb2, step_size, sqrt_bc2, a.eps);
            adam_one(p.z, g.z, m.z, v.z, decay, b1, omb1, b2, omb2, step_size, sqrt_bc2, a.eps);
            adam_one(p.w, g.w, m.w, v.w, decay, b1, omb1, b2, omb2, step_size, sqrt_bc2, a.eps);
            *reinterpret_cast<float4*>(jb.p + i) = p;
            *reinterpret_cast<float4*>(jb.m + i) = m;
            *reinterpret_cast<float4*>(jb.v + i) = v;
        } else {
            for (long long k = i; k < i + 4 && k < jb.n; ++k) {
                float p = jb.p[k], m = jb.m[k], v = jb.v[k];
                adam_one(p, jb.g[k], m, v, decay, b1, omb1, b2, omb2, step_size, sqrt_bc2, a.eps);
                jb.p[k] = p; jb.m[k] = m; jb.v[k] = v;
            }
        }
    }
}

// step <- step + 1 unless skipped (one thread)
__global__ void k_adamw_count(float* step, const float* skip) {
    if (skip && *skip != 0.0f) return;
    *step += 1.0f;
}
}  // namespace

extern "C" size_t mas_adamw_job_bytes(void) { return sizeof(AdamJob); }

extern "C" int mas_adamw_max_groups(void) { return kMaxGroups; }

extern "C" unsigned mas_adamw_job(void* job_host, float* p, const float* g, float* m, float* v, long long n, int group, unsigned first_block) {
    if (!job_host || !p || !g || !m || !v || n <= 0 || group < 0 || group >= kMaxGroups) return 0;
    const long long blocks = (n + kPerBlock - 1) / kPerBlock;
    if (blocks > 0x7fffffffLL - first_block) return 0;
    AdamJob jb{p, g, m, v, n, group, first_block};
    *static_cast<AdamJob*>(job_host) = jb;
    return (unsigned)blocks;
}

extern "C" int mas_adamw_multi(const void* jobs_dev, int njobs, unsigned nblocks, const float* lr_host, int ngroups, double beta1, double beta2,
                               double eps, double weight_decay, float* step_dev, const float* skip_dev, void* stream) {
    if (!jobs_dev || !step_dev || !lr_host) return MAS_ERR_NULL;
    if (njobs <= 0 || nblocks == 0) return MAS_ERR_SHAPE;
    if (ngroups <= 0 || ngroups > kMaxGroups) return MAS_ERR_RANGE;
    AdamArgs a{};
    a.jobs = static_cast<const AdamJob*>(jobs_dev);
    a.njobs = njobs;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = (float)eps; a.weight_decay = weight_decay;
    for (int i = 0; i < ngroups; ++i) a.lr[i] = lr_host[i];
    a.step = step_dev;
    a.skip = skip_dev;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_adamw_multi, dim3(nblocks), dim3(kThreads), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_adamw_count, dim3(1), dim3(1), 0, st, step_dev, skip_dev);
    return (int)hipGetLastError();
}
