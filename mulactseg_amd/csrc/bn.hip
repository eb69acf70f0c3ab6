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
// the launch or on the run.  Layout NCHW; 16-byte accesses for planes of any length (aligned groups, see groups_of).
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

// ---- 16-byte groups over planes of any length ----------------------------------------------------------------------
// A plane (n, c) starts at element (n*C + c)*HW: with H*W odd (the 769 crop: 385^2, 193^2, 97^2, 49^2) three planes out of
// four are not 16-byte aligned.  Every streaming kernel therefore walks ALIGNED groups of four floats: group g of a plane
// covers the addresses [start + 4g, start + 4g + 4) with start = the plane's first element rounded down to 16 bytes; a
// group that lies completely inside the plane is one float4 access, the (at most two) boundary groups of a plane are
// handled element by element -- their other elements belong to the neighbouring planes, which other workgroups own.
// `a` = misalignment of the plane in elements (0..3) taken from the real address, so tensors with a storage offset work
// too; VEC is false when the tensors of a launch are not congruent modulo 16 bytes (then every group goes element-wise).
struct Groups { const float* start; int a; int n; };

__device__ __forceinline__ Groups groups_of(const float* plane, int HW) {
    Groups g;
    g.a = (int)((reinterpret_cast<uintptr_t>(plane) >> 2) & 3);
    g.start = plane - g.a;
    g.n = (g.a + HW + 3) >> 2;
    return g;
}
__device__ __forceinline__ bool group_full(const Groups& g, int i, int HW) { return 4 * i >= g.a && 4 * i + 4 <= g.a + HW; }
__device__ __forceinline__ bool elem_in(const Groups& g, int i, int k, int HW) { return 4 * i + k >= g.a && 4 * i + k < g.a + HW; }

// NT: the tensor is read for the LAST time by this kernel (or will have left the caches long before its next reader): non-temporal
// load policy, as for the scans' logits (common.h:mas_load_stream4)
template <bool VEC, bool NT = false>
__device__ __forceinline__ float4 load_group(const float* t_start, const Groups& g, int i, int HW, float fill) {
    if (VEC && group_full(g, i, HW)) return NT ? mas_load_stream4(t_start + 4 * i) : *reinterpret_cast<const float4*>(t_start + 4 * i);
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = elem_in(g, i, k, HW) ? t_start[4 * i + k] : fill;
    return make_float4(v[0], v[1], v[2], v[3]);
}
template <bool VEC>
__device__ __forceinline__ void store_group(float* t_start, const Groups& g, int i, int HW, float4 o) {
    if (VEC && group_full(g, i, HW)) { *reinterpret_cast<float4*>(t_start + 4 * i) = o; return; }
    const float v[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (elem_in(g, i, k, HW)) t_start[4 * i + k] = v[k];
}
__device__ __forceinline__ float4 mask_bits(unsigned m) {
    return make_float4((float)(m & 1u), (float)((m >> 1) & 1u), (float)((m >> 2) & 1u), (float)((m >> 3) & 1u));
}
constexpr int kGroupsPerChunk = kChunk / 4;

// grid (chunks, N, C): part[(c * N + n) * chunks + chunk] = (sum x, sum x^2)
// Hand-off of a partial pair to whichever workgroup finishes the channel, WITHOUT device-scope fences: the pair is stored
// write-through at agent scope (sc1), the storing thread drains its stores, then a relaxed agent-scope counter add publishes it; the
// finisher reads the pairs with sc1 loads (served by the coherence point the write-through stores went to).  A __threadfence() per
// workgroup instead -- L2 write-back + invalidate on every XCD, thousands of times per launch, under kernels that stream at HBM speed
// -- made k_bn_bwd_partial 7x slower (10.6 ms per step against 1.45: round 5).  The idiom of csrc/conv_sk.hip.
typedef unsigned v4u_bn __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bn_store_sc1(double2* p, double S, double Q) {
    const double2 v = make_double2(S, Q);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(__builtin_bit_cast(v4u_bn, v)) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ double2 bn_load_sc1(const double2* p) {
    v4u_bn v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return __builtin_bit_cast(double2, v);
}
__device__ __forceinline__ bool bn_last_arriver(unsigned* counter, unsigned per_channel) {
    if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != per_channel - 1) return false;
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // (ready for the next launch on this stream)
    return true;
}

// mean / invstd / running statistics of one channel from its `per_channel` partial pairs, summed in index order (one thread)
template <bool SC1>
__device__ __forceinline__ void bn_channel_stats(const double2* part, int c, int per_channel, double count, float eps, float momentum, float* mean,
                                                 float* invstd, float* running_mean, float* running_var) {
    double S = 0.0, Q = 0.0;
    for (int i = 0; i < per_channel; ++i) {
        const double2 v = SC1 ? bn_load_sc1(part + (size_t)c * per_channel + i) : part[(size_t)c * per_channel + i];
        S += v.x; Q += v.y;
    }
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

// `counters` != NULL (one zeroed word per channel, left zeroed): the workgroup that writes a channel's LAST partial pair turns the
// pairs into that channel's statistics -- the k_bn_stats launch (one of 115 five-microsecond launches per training step) is gone.
// Release / acquire at device scope: partial stored, fence, counter add; the last arriver fences again before it reads the pairs.
struct BnStatOut { float eps, momentum; float* mean; float* invstd; float* running_mean; float* running_var; long long* nbt; unsigned* counters; };

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_bn_partial(const float* __restrict__ x, int C, int HW, int chunks, double2* __restrict__ part,
                                                          const BnStatOut so) {
    __shared__ double s_red[kThreads / MAS_WAVE];
    const int chunk = blockIdx.x, n = blockIdx.y, c = blockIdx.z;
    const Groups g = groups_of(x + ((size_t)n * C + c) * HW, HW);
    const int lo = chunk * kGroupsPerChunk, hi = min(g.n, lo + kGroupsPerChunk);
    float s = 0.f, q = 0.f;
    for (int i = lo + threadIdx.x; i < hi; i += kThreads) {
        const float4 v = load_group<VEC>(g.start, g, i, HW, 0.0f);
        s += (v.x + v.y) + (v.z + v.w);
        q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    const double S = block_sum((double)s, s_red);
    const double Q = block_sum((double)q, s_red);
    if (threadIdx.x != 0) return;
    double2* mine = part + ((size_t)c * gridDim.y + n) * chunks + chunk;
    if (!so.counters) { *mine = make_double2(S, Q); return; }
    bn_store_sc1(mine, S, Q);
    const unsigned per_channel = gridDim.x * gridDim.y;
    if (!bn_last_arriver(&so.counters[c], per_channel)) return;
    if (c == 0 && so.nbt) so.nbt[0] += 1;
    bn_channel_stats<true>(part, c, (int)per_channel, (double)gridDim.y * (double)HW, so.eps, so.momentum, so.mean, so.invstd, so.running_mean,
                           so.running_var);
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

// The same statistics from MANY partials per channel (the epilogue partials of mas_conv_sk_stats: one pair per tile row group, up to
// ~18 000 per channel at 384 x 384): one workgroup per channel, thread t adds entries t, t + 256, ... in index order, then a fixed
// tree over the 256 threads -- a fixed order for a given per_channel, in double.  (k_bn_stats walks them with one thread per
// channel: 10 ms per step at these counts.)
__global__ __launch_bounds__(kThreads) void k_bn_stats_wide(const double2* __restrict__ part, int C, int per_channel, double count, float eps,
                                                             float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var,
                                                             long long* __restrict__ num_batches_tracked) {
    __shared__ double s_red[kThreads / MAS_WAVE];
    const int c = blockIdx.x;
    if (c == 0 && threadIdx.x == 0 && num_batches_tracked) num_batches_tracked[0] += 1;
    double S = 0.0, Q = 0.0;
    for (int i = threadIdx.x; i < per_channel; i += kThreads) { const double2 v = part[(size_t)c * per_channel + i]; S += v.x; Q += v.y; }
    S = block_sum(S, s_red);
    Q = block_sum(Q, s_red);
    if (threadIdx.x != 0) return;
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

// grid (ceil(groups / 256), N * C), one aligned group per thread.  FROM_VAR: `stat2` holds the running variance (inference),
// else invstd.  mask: one byte per group, mask_stride bytes per plane (bit k = output k of the group is positive).
template <bool FROM_VAR, bool VEC>
__global__ __launch_bounds__(kThreads) void k_bn_apply(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ stat1, const float* __restrict__ stat2, float eps,
                                                        const float* __restrict__ res, int C, int HW, int relu, float* __restrict__ y,
                                                        unsigned char* __restrict__ mask, int mask_stride) {
    const int c = blockIdx.y % C;
    const float mu = stat1[c];
    const float is = FROM_VAR ? 1.0f / sqrtf(stat2[c] + eps) : stat2[c];
    const float gm = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    const size_t base = (size_t)blockIdx.y * HW;
    const Groups g = groups_of(x + base, HW);
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= g.n) return;
    auto f = [&](float v, float r) {
        float t = ((v - mu) * is) * gm + b;
        t = t + r;
        return (relu && !(t > 0.0f)) ? 0.0f : t;
    };
    const float4 v = load_group<VEC, true>(g.start, g, i, HW, 0.0f);
    const float4 r = res ? load_group<VEC, true>(res + base - g.a, g, i, HW, 0.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 o = make_float4(f(v.x, r.x), f(v.y, r.y), f(v.z, r.z), f(v.w, r.w));
    store_group<VEC>(y + base - g.a, g, i, HW, o);
    if (mask) mask[(size_t)blockIdx.y * mask_stride + i] = (unsigned char)((o.x > 0.f) | ((o.y > 0.f) << 1) | ((o.z > 0.f) << 2) | ((o.w > 0.f) << 3));
}

// part[(c * N + n) * chunks + chunk] = (sum g, sum g * xhat),  g = dy * [y > 0] (relu) or dy
// Bandwidth on this part is bytes in flight / round trip: a thread that loads one group of each tensor, uses it
// and only then asks for the next keeps 32 bytes in flight (the first form of this kernel: 4.1 TB/s of the 6 the apply kernels
// reach).  Here a thread requests FOUR groups of dy and x (and their mask bytes) before it touches any of them; only groups that
// lie whole inside the plane take that path, the (at most two) boundary groups of the plane are added by one thread, element-wise.
template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_partial(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                              const unsigned char* __restrict__ mask, int mask_stride, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, int C, int HW, int chunks, int relu,
                                                              double2* __restrict__ part, unsigned* __restrict__ counters, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float2* __restrict__ coef) {
    __shared__ double s_red[kThreads / MAS_WAVE];
    const int chunk = blockIdx.x, n = blockIdx.y, c = blockIdx.z;
    const size_t plane = (size_t)n * C + c, base = plane * HW;
    const float mu = mean[c], is = invstd[c];
    const Groups g = groups_of(x + base, HW);
    const int lo = chunk * kGroupsPerChunk, hi = min(g.n, lo + kGroupsPerChunk);
    float s = 0.f, q = 0.f;
    auto acc = [&](float gr, float xv, float yv) {
        gr = (relu && !(yv > 0.0f)) ? 0.0f : gr;
        s += gr;
        q += gr * ((xv - mu) * is);
    };
    const float* dys = dy + base - g.a;
    const float* ys = (relu && !mask) ? y + base - g.a : nullptr;
    const unsigned char* mrow = mask ? mask + plane * mask_stride : nullptr;
    // whole groups of this chunk: [flo, fhi)
    const int first_full = (g.a + 3) >> 2, end_full = (g.a + HW) >> 2;      // groups [first_full, end_full) lie inside the plane
    const int flo = VEC ? max(lo, first_full) : hi, fhi = VEC ? min(hi, end_full) : hi;
    if (VEC) {
        constexpr int U = 4;
        for (int i0 = flo + threadIdx.x; i0 < fhi; i0 += U * kThreads) {
            float4 gr[U], xv[U], yv[U];
            unsigned mk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * kThreads;
                const int j = i < fhi ? i : i0;                      // (a clamped duplicate: loaded, not added)
                gr[u] = *reinterpret_cast<const float4*>(dys + 4 * j);
                xv[u] = *reinterpret_cast<const float4*>(g.start + 4 * j);
                if (mrow) mk[u] = mrow[j];
                else if (ys) yv[u] = *reinterpret_cast<const float4*>(ys + 4 * j);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + u * kThreads < fhi) {
                    float4 yy = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (mrow) yy = mask_bits(mk[u]);
                    else if (ys) yy = yv[u];
                    acc(gr[u].x, xv[u].x, yy.x); acc(gr[u].y, xv[u].y, yy.y); acc(gr[u].z, xv[u].z, yy.z); acc(gr[u].w, xv[u].w, yy.w);
                }
            }
        }
    }
    // what is left: every group of the chunk on the element path (!VEC), or the plane's boundary groups that fall into this chunk
    for (int i = lo + threadIdx.x; i < hi; i += kThreads) {
        if (VEC && i >= flo && i < fhi) continue;
        const float4 gr = load_group<false>(dys, g, i, HW, 0.0f);                   // elements of other planes: gradient 0, x = mean
        const float4 xv = load_group<false>(g.start, g, i, HW, mu);
        float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (relu && mask) yv = mask_bits(mask[plane * mask_stride + i]);
        else if (relu) yv = load_group<false>(y + base - g.a, g, i, HW, 0.0f);
        acc(gr.x, xv.x, yv.x); acc(gr.y, xv.y, yv.y); acc(gr.z, xv.z, yv.z); acc(gr.w, xv.w, yv.w);
    }
    const double S = block_sum((double)s, s_red);
    const double Q = block_sum((double)q, s_red);
    if (threadIdx.x != 0) return;
    double2* mine = part + ((size_t)c * gridDim.y + n) * chunks + chunk;
    if (!counters) { *mine = make_double2(S, Q); return; }
    // (as k_bn_partial: the last workgroup of a channel does k_bn_bwd_stats' work for that channel)
    bn_store_sc1(mine, S, Q);
    const unsigned per_channel = gridDim.x * gridDim.y;
    if (!bn_last_arriver(&counters[c], per_channel)) return;
    double St = 0.0, Qt = 0.0;
    for (unsigned i = 0; i < per_channel; ++i) { const double2 v = bn_load_sc1(part + (size_t)c * per_channel + i); St += v.x; Qt += v.y; }
    const double count = (double)gridDim.y * (double)HW;
    if (dbeta) dbeta[c] = (float)St;
    if (dgamma) dgamma[c] = (float)Qt;
    coef[c] = make_float2((float)(St / count), (float)(Qt / count));
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

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                            const unsigned char* __restrict__ mask, int mask_stride, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float2* __restrict__ coef, int C, int HW,
                                                            int relu, float* __restrict__ dx, float* __restrict__ dres) {
    const int c = blockIdx.y % C;
    const float mu = mean[c], is = invstd[c];
    const float k = (gamma ? gamma[c] : 1.0f) * is;
    const float2 m = coef[c];
    const size_t base = (size_t)blockIdx.y * HW;
    const Groups g = groups_of(x + base, HW);
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= g.n) return;
    auto f = [&](float gr, float xv, float yv, float& gout) {
        gr = (relu && !(yv > 0.0f)) ? 0.0f : gr;
        gout = gr;
        return k * ((gr - m.x) - ((xv - mu) * is) * m.y);
    };
    const float4 gr = load_group<VEC, true>(dy + base - g.a, g, i, HW, 0.0f);
    const float4 xv = load_group<VEC, true>(g.start, g, i, HW, 0.0f);
    float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (relu && mask) yv = mask_bits(mask[(size_t)blockIdx.y * mask_stride + i]);
    else if (relu) yv = load_group<VEC>(y + base - g.a, g, i, HW, 0.0f);
    float4 r, o;
    o.x = f(gr.x, xv.x, yv.x, r.x); o.y = f(gr.y, xv.y, yv.y, r.y); o.z = f(gr.z, xv.z, yv.z, r.z); o.w = f(gr.w, xv.w, yv.w, r.w);
    store_group<VEC>(dx + base - g.a, g, i, HW, o);
    if (dres) store_group<VEC>(dres + base - g.a, g, i, HW, r);
}

int check(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0 || N > 65535 || C > 65535 || (long long)N * C > 0x7fffffffLL) return MAS_ERR_SHAPE;
    return 0;
}
inline int max_groups(int HW) { return (HW + 3) / 4 + 1; }                    // aligned groups a plane can touch (any misalignment)
inline int chunks_of(int HW) { return (max_groups(HW) + kGroupsPerChunk - 1) / kGroupsPerChunk; }
inline dim3 apply_grid(int N, int C, int HW) { return dim3((unsigned)((max_groups(HW) + kThreads - 1) / kThreads), (unsigned)(N * C)); }
// float4 accesses need every tensor of the launch at the same offset modulo 16 bytes (and 4-byte aligned)
inline bool congruent(const void* a, const void* b) { return b == nullptr || (((uintptr_t)a ^ (uintptr_t)b) & 15) == 0; }
}  // namespace

/* bytes of the ReLU mask of mas_bn_act_train_fwd / _bwd: one byte per aligned 16-byte group, max_groups(HW) per plane */
extern "C" int64_t mas_bn_mask_bytes(int N, int C, int HW) {
    if (check(N, C, HW)) return -1;
    return (int64_t)N * C * max_groups(HW);
}

extern "C" int64_t mas_bn_workspace_bytes(int N, int C, int HW) {
    if (check(N, C, HW)) return -1;
    return (int64_t)sizeof(double2) * C * N * chunks_of(HW) + (int64_t)sizeof(float2) * C;
}

extern "C" int mas_bn_act_train_fwd(const float* x, const float* gamma, const float* beta, const float* residual, int N, int C, int HW,
                                    float eps, float momentum, int relu, float* running_mean, float* running_var,
                                    int64_t* num_batches_tracked, float* save_mean, float* save_invstd, void* workspace, float* y,
                                    uint8_t* relu_mask, uint32_t* counters, void* stream) {
    if (!x || !save_mean || !save_invstd || !workspace || !y) return MAS_ERR_NULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MAS_ERR_NULL;
    if (int e = check(N, C, HW)) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int chunks = chunks_of(HW);
    double2* part = static_cast<double2*>(workspace);
    const bool vec = congruent(x, y) && congruent(x, residual) && ((uintptr_t)x & 3) == 0;
    const BnStatOut so{eps, momentum, save_mean, save_invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), counters};
    if (vec) hipLaunchKernelGGL(k_bn_partial<true>, dim3((unsigned)chunks, (unsigned)N, (unsigned)C), dim3(kThreads), 0, st, x, C, HW, chunks, part, so);
    else hipLaunchKernelGGL(k_bn_partial<false>, dim3((unsigned)chunks, (unsigned)N, (unsigned)C), dim3(kThreads), 0, st, x, C, HW, chunks, part, so);
    if (!counters)
        hipLaunchKernelGGL(k_bn_stats, dim3((unsigned)((C + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, part, C, N * chunks,
                           (double)N * (double)HW, eps, momentum, save_mean, save_invstd, running_mean, running_var,
                           reinterpret_cast<long long*>(num_batches_tracked));
    unsigned char* mk = relu ? relu_mask : nullptr;
    if (vec) hipLaunchKernelGGL((k_bn_apply<false, true>), apply_grid(N, C, HW), dim3(kThreads), 0, st, x, gamma, beta, save_mean, save_invstd, eps,
                                residual, C, HW, relu, y, mk, max_groups(HW));
    else hipLaunchKernelGGL((k_bn_apply<false, false>), apply_grid(N, C, HW), dim3(kThreads), 0, st, x, gamma, beta, save_mean, save_invstd, eps,
                            residual, C, HW, relu, y, mk, max_groups(HW));
    return mas_launch_status();
}

/* mas_bn_act_train_fwd with the partial sums already formed by the producer of x (mas_conv_sk_stats: `per_channel` pairs
 * (sum, sum of squares) per channel over disjoint pixel sets, doubles): statistics + apply, no reduction pass over x. */
extern "C" int mas_bn_act_train_fwd_stats(const float* x, const double* partials, int per_channel, const float* gamma, const float* beta,
                                          const float* residual, int N, int C, int HW, float eps, float momentum, int relu, float* running_mean,
                                          float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd, float* y,
                                          uint8_t* relu_mask, void* stream) {
    if (!x || !partials || !save_mean || !save_invstd || !y) return MAS_ERR_NULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MAS_ERR_NULL;
    if (per_channel <= 0) return MAS_ERR_SHAPE;
    if (int e = check(N, C, HW)) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = congruent(x, y) && congruent(x, residual) && ((uintptr_t)x & 3) == 0;
    hipLaunchKernelGGL(k_bn_stats_wide, dim3((unsigned)C), dim3(kThreads), 0, st, reinterpret_cast<const double2*>(partials), C, per_channel,
                       (double)N * (double)HW, eps, momentum, save_mean, save_invstd, running_mean, running_var,
                       reinterpret_cast<long long*>(num_batches_tracked));
    unsigned char* mk = relu ? relu_mask : nullptr;
    if (vec) hipLaunchKernelGGL((k_bn_apply<false, true>), apply_grid(N, C, HW), dim3(kThreads), 0, st, x, gamma, beta, save_mean, save_invstd, eps,
                                residual, C, HW, relu, y, mk, max_groups(HW));
    else hipLaunchKernelGGL((k_bn_apply<false, false>), apply_grid(N, C, HW), dim3(kThreads), 0, st, x, gamma, beta, save_mean, save_invstd, eps,
                            residual, C, HW, relu, y, mk, max_groups(HW));
    return mas_launch_status();
}

extern "C" int mas_bn_act_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, const float* residual, int N, int C, int HW, float eps, int relu, float* y,
                                   void* stream) {
    if (!x || !running_mean || !running_var || !y) return MAS_ERR_NULL;
    if (int e = check(N, C, HW)) return e;
    const bool vec = congruent(x, y) && congruent(x, residual) && ((uintptr_t)x & 3) == 0;
    if (vec) hipLaunchKernelGGL((k_bn_apply<true, true>), apply_grid(N, C, HW), dim3(kThreads), 0, static_cast<hipStream_t>(stream), x, gamma, beta,
                                running_mean, running_var, eps, residual, C, HW, relu, y, nullptr, 0);
    else hipLaunchKernelGGL((k_bn_apply<true, false>), apply_grid(N, C, HW), dim3(kThreads), 0, static_cast<hipStream_t>(stream), x, gamma, beta,
                            running_mean, running_var, eps, residual, C, HW, relu, y, nullptr, 0);
    return mas_launch_status();
}

extern "C" int mas_bn_act_train_bwd(const float* dy, const float* x, const float* y, const uint8_t* relu_mask, const float* gamma,
                                    const float* save_mean, const float* save_invstd, int N, int C, int HW, int relu, void* workspace,
                                    float* dx, float* dresidual, float* dgamma, float* dbeta, uint32_t* counters, void* stream) {
    if (!dy || !x || !save_mean || !save_invstd || !workspace || !dx) return MAS_ERR_NULL;
    if (relu && !y && !relu_mask) return MAS_ERR_NULL;
    if (int e = check(N, C, HW)) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int chunks = chunks_of(HW);
    double2* part = static_cast<double2*>(workspace);
    float2* coef = reinterpret_cast<float2*>(part + (size_t)C * N * chunks);
    const bool vec = congruent(x, dy) && congruent(x, dx) && congruent(x, dresidual) && (relu_mask || congruent(x, y)) && ((uintptr_t)x & 3) == 0;
    const int ms = max_groups(HW);
    if (vec) hipLaunchKernelGGL(k_bn_bwd_partial<true>, dim3((unsigned)chunks, (unsigned)N, (unsigned)C), dim3(kThreads), 0, st, dy, x, y, relu_mask, ms,
                                save_mean, save_invstd, C, HW, chunks, relu, part, counters, dgamma, dbeta, coef);
    else hipLaunchKernelGGL(k_bn_bwd_partial<false>, dim3((unsigned)chunks, (unsigned)N, (unsigned)C), dim3(kThreads), 0, st, dy, x, y, relu_mask, ms,
                            save_mean, save_invstd, C, HW, chunks, relu, part, counters, dgamma, dbeta, coef);
    if (!counters)
        hipLaunchKernelGGL(k_bn_bwd_stats, dim3((unsigned)((C + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, part, C, N * chunks,
                           (double)N * (double)HW, dgamma, dbeta, coef);
    if (vec) hipLaunchKernelGGL(k_bn_bwd_apply<true>, apply_grid(N, C, HW), dim3(kThreads), 0, st, dy, x, y, relu_mask, ms, gamma, save_mean,
                                save_invstd, coef, C, HW, relu, dx, dresidual);
    else hipLaunchKernelGGL(k_bn_bwd_apply<false>, apply_grid(N, C, HW), dim3(kThreads), 0, st, dy, x, y, relu_mask, ms, gamma, save_mean,
                            save_invstd, coef, C, HW, relu, dx, dresidual);
    return mas_launch_status();
}
