// scorer.hip -- acquisition-scorer kernels (K1-K4 of SURVEY.md section 2.3) for gfx950.
//
//   k_class_prob_sum      K2  one streaming read of z[B,C,H,W]; per-thread fixed-point class sums in
//                             registers, wave shuffle reduction, one 64-bit atomic per class per block.
//   k_bvsb_region_accum   K1+K3  one streaming read of z + superpixel ids; per-pixel top-2 in registers,
//                             a per-workgroup open-addressed LDS table id -> (sum, hist[C]) absorbs the
//                             segmented reduction; one global atomic per touched (region, field) at tile end.
//   k_region_finalize     K3 tail + K4 ban (mean, dominant class, ban-ignore).
//
// All kernels are HBM-bound scans (about 2-6 flop/byte); no MFMA on purpose.  Reads are 16 B per lane
// (1 KiB per wave instruction) when rows are 16-B aligned, dword per lane otherwise.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTileW = 256;     // 64 lanes x 4 px
constexpr int kTileH = 16;      // 4 waves x 4 row iterations
constexpr int kLogSlots = 7;
constexpr int kSlots = 1 << kLogSlots;

// ------------------------------------------------------------------------------------------------
// K2
// ------------------------------------------------------------------------------------------------
template <int CT, bool EXACT, bool VEC>
__global__ __launch_bounds__(kThreads) void k_class_prob_sum(const float* __restrict__ z, int C, int HW, float invT,
                                                              mas_u64* __restrict__ prob_sum, int blocks_per_image) {
    const int b = blockIdx.x / blocks_per_image;
    const int j = blockIdx.x - b * blocks_per_image;
    const float* zb = z + (size_t)b * C * HW;
    // per-thread 32-bit accumulators of mas_probq quanta (<= 2^23 each): exact for up to 511 pixels per thread,
    // which the launcher guarantees by its choice of blocks_per_image.  The quanta are accumulated as raw bit
    // patterns of fma(e, R, 2^23); the constant MAS_PROBQ_BIAS is subtracted once per accumulator at the end
    // (modulo 2^32), so the inner loop is one packed fma and one 3-operand add per class and pixel pair.
    unsigned acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0;
    unsigned n_quanta = 0;

    const int chunk_px = kThreads * 4;
    for (int p0 = j * chunk_px; p0 < HW; p0 += blocks_per_image * chunk_px) {
        // 4 pixels per lane as two PAIRS: packed f32 math evaluates both pixels of a pair per instruction
        mas_v2f v[2][CT];
        bool ok[4];
        if (VEC) {
            const int p = p0 + threadIdx.x * 4;
            const bool in = p < HW;      // HW % 4 == 0 on this path
#pragma unroll
            for (int k = 0; k < 4; ++k) ok[k] = in;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                const float* zc = zb + (size_t)c * HW;       // wave-uniform base, 32-bit lane offset
                if ((EXACT || c < C) && in) t = mas_load_stream4(zc + (unsigned)p);
                v[0][c] = (mas_v2f){t.x, t.y};
                v[1][c] = (mas_v2f){t.z, t.w};
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) ok[k] = (p0 + k * kThreads + (int)threadIdx.x) < HW;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                float s[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int p = p0 + k * kThreads + threadIdx.x;
                    s[k] = ((EXACT || c < C) && ok[k]) ? zb[(size_t)c * HW + p] : 0.f;
                }
                v[0][c] = (mas_v2f){s[0], s[1]};
                v[1][c] = (mas_v2f){s[2], s[3]};
            }
        }
        {
            mas_v2f Ra, Rb;
            mas_softmax_quad<CT, EXACT>(v[0], v[1], C, invT, Ra, Rb);
            Ra = Ra * mas_splat(8388608.0f);
            Rb = Rb * mas_splat(8388608.0f);
            Ra = (mas_v2f){ok[0] ? Ra.x : 0.0f, ok[1] ? Ra.y : 0.0f};     // pixels past the end add exactly 0
            Rb = (mas_v2f){ok[2] ? Rb.x : 0.0f, ok[3] ? Rb.y : 0.0f};
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (EXACT || c < C) {
                    const mas_v2f ta = mas_pk_fma(v[0][c], Ra, mas_splat(8388608.0f));      // mas_probq + bias
                    const mas_v2f tb = mas_pk_fma(v[1][c], Rb, mas_splat(8388608.0f));
                    acc[c] += (mas_f2u(ta.x) + mas_f2u(ta.y)) + (mas_f2u(tb.x) + mas_f2u(tb.y));
                }
            }
        }
        n_quanta += 4;
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] -= n_quanta * MAS_PROBQ_BIAS;

    // wave reduction (64 lanes), then 4 waves through LDS, then one atomic per class
    __shared__ mas_u64 s_part[kThreads / MAS_WAVE][CT];
    const int lane = threadIdx.x & (MAS_WAVE - 1);
    const int wave = threadIdx.x / MAS_WAVE;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        mas_u64 a = acc[c];
#pragma unroll
        for (int off = MAS_WAVE / 2; off > 0; off >>= 1) a += __shfl_down(a, off, MAS_WAVE);
        if (lane == 0) s_part[wave][c] = a;
    }
    __syncthreads();
    if (threadIdx.x < CT && (EXACT || (int)threadIdx.x < C)) {
        mas_u64 a = 0;
#pragma unroll
        for (int w = 0; w < kThreads / MAS_WAVE; ++w) a += s_part[w][threadIdx.x];
        if (a) atomicAdd(&prob_sum[(size_t)b * C + threadIdx.x], a);
    }
}

// Ring form of K2 (16-B aligned planes): as in single_pass.hip every wave keeps the NEXT chunk's 20 loads in flight while
// it runs the softmax of the current one (two 80-register buffers, 2 waves/SIMD).  Same arithmetic, same integer sums.
template <int CT, bool EXACT>
__device__ __forceinline__ void k2_issue(float4 (&t)[CT], const float* __restrict__ zb, int C, int HW, unsigned off) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (EXACT || c < C) t[c] = mas_load_stream4(zb + (size_t)c * HW + off);
        else t[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int CT, bool EXACT>
__device__ __forceinline__ void k2_consume(float4 (&t)[CT], bool ok, int C, float invT, unsigned (&acc)[CT]) {
    mas_v2f v[2][CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        v[0][c] = (mas_v2f){t[c].x, t[c].y};
        v[1][c] = (mas_v2f){t[c].z, t[c].w};
    }
    mas_v2f Ra, Rb;
    mas_softmax_quad<CT, EXACT>(v[0], v[1], C, invT, Ra, Rb);
    Ra = Ra * mas_splat(8388608.0f);
    Rb = Rb * mas_splat(8388608.0f);
    Ra = ok ? Ra : mas_splat(0.0f);               // chunks past the end add exactly the bias (removed below)
    Rb = ok ? Rb : mas_splat(0.0f);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (EXACT || c < C) {
            const mas_v2f ta = mas_pk_fma(v[0][c], Ra, mas_splat(8388608.0f));
            const mas_v2f tb = mas_pk_fma(v[1][c], Rb, mas_splat(8388608.0f));
            acc[c] += (mas_f2u(ta.x) + mas_f2u(ta.y)) + (mas_f2u(tb.x) + mas_f2u(tb.y));
        }
    }
}

template <int CT, bool EXACT>
__global__ __launch_bounds__(kThreads, 2) void k_class_prob_sum_ring(const float* __restrict__ z, int C, int HW, float invT,
                                                                      mas_u64* __restrict__ prob_sum, int blocks_per_image, int iters) {
    const int b = blockIdx.x / blocks_per_image;
    const int j = blockIdx.x - b * blocks_per_image;
    const float* zb = z + (size_t)b * C * HW;
    unsigned acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0;
    const int chunk_px = kThreads * 4;
    // chunk `it` of this workgroup starts at pixel (j + it * blocks_per_image) * chunk_px; out-of-range chunks re-read the
    // last valid float4 of the plane (a valid address) and are masked
    auto off_of = [&](int it) -> unsigned {
        const long long p = ((long long)j + (long long)it * blocks_per_image) * chunk_px + threadIdx.x * 4;
        return (unsigned)(p < HW ? p : HW - 4);
    };
    auto ok_of = [&](int it) -> bool {
        return ((long long)j + (long long)it * blocks_per_image) * chunk_px + threadIdx.x * 4 < HW;
    };
    float4 A[CT], Bf[CT];
    k2_issue<CT, EXACT>(A, zb, C, HW, off_of(0));
#pragma unroll 1
    for (int it = 0; it < iters; it += 2) {
        k2_issue<CT, EXACT>(Bf, zb, C, HW, off_of(it + 1));
        __builtin_amdgcn_sched_barrier(0);
        k2_consume<CT, EXACT>(A, ok_of(it), C, invT, acc);
        __builtin_amdgcn_sched_barrier(0);
        k2_issue<CT, EXACT>(A, zb, C, HW, off_of(it + 2));
        __builtin_amdgcn_sched_barrier(0);
        k2_consume<CT, EXACT>(Bf, (it + 1 < iters) && ok_of(it + 1), C, invT, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned n_quanta = (unsigned)(((iters + 1) / 2) * 2) * 4u;
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] -= n_quanta * MAS_PROBQ_BIAS;

    __shared__ mas_u64 s_part[kThreads / MAS_WAVE][CT];
    const int lane = threadIdx.x & (MAS_WAVE - 1);
    const int wave = threadIdx.x / MAS_WAVE;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        mas_u64 a = acc[c];
#pragma unroll
        for (int off = MAS_WAVE / 2; off > 0; off >>= 1) a += __shfl_down(a, off, MAS_WAVE);
        if (lane == 0) s_part[wave][c] = a;
    }
    __syncthreads();
    if (threadIdx.x < CT && (EXACT || (int)threadIdx.x < C)) {
        mas_u64 a = 0;
#pragma unroll
        for (int w = 0; w < kThreads / MAS_WAVE; ++w) a += s_part[w][threadIdx.x];
        if (a) atomicAdd(&prob_sum[(size_t)b * C + threadIdx.x], a);
    }
}

// ------------------------------------------------------------------------------------------------
// K1 + K3
// ------------------------------------------------------------------------------------------------
struct RegionTable {
    int* keys;        // [kSlots]  superpixel id or -1
    mas_u64* sum;     // [kSlots]
    unsigned* hist;   // [kSlots * C]
};

__device__ __forceinline__ int table_slot(int* keys, int id) {
    unsigned h = ((unsigned)id * 2654435769u) >> (32 - kLogSlots);
    for (int probe = 0; probe < kSlots; ++probe) {
        int k = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == id) return (int)h;
        if (k == -1) {
            int old = atomicCAS(&keys[h], -1, id);
            if (old == -1 || old == id) return (int)h;
        }
        h = (h + 1) & (kSlots - 1);
    }
    return -1;   // table full: caller goes straight to global memory
}

template <int CT, bool EXACT, typename IdT, bool VEC>
__global__ __launch_bounds__(kThreads) void k_bvsb_region_accum(const float* __restrict__ z, const IdT* __restrict__ spx,
                                                                 const float* __restrict__ cls_w, int C, int H, int W,
                                                                 int S, float invT, int tiles_x, int tiles_y,
                                                                 mas_u64* __restrict__ score_sum,
                                                                 unsigned* __restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    RegionTable t;
    t.sum = reinterpret_cast<mas_u64*>(smem);
    t.keys = reinterpret_cast<int*>(smem + sizeof(mas_u64) * kSlots);
    float* s_w = reinterpret_cast<float*>(smem + (sizeof(mas_u64) + sizeof(int)) * kSlots);
    t.hist = reinterpret_cast<unsigned*>(smem + (sizeof(mas_u64) + sizeof(int)) * kSlots + sizeof(float) * MAS_MAX_CLASSES);

    for (int i = threadIdx.x; i < kSlots; i += kThreads) { t.keys[i] = -1; t.sum[i] = 0; }
    for (int i = threadIdx.x; i < kSlots * C; i += kThreads) t.hist[i] = 0;
    if (threadIdx.x < MAS_MAX_CLASSES) s_w[threadIdx.x] = (cls_w && (int)threadIdx.x < C) ? cls_w[threadIdx.x] : 1.0f;
    __syncthreads();

    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int HW = H * W;
    const float* zb = z + (size_t)b * C * HW;
    const IdT* sb = spx + (size_t)b * HW;
    mas_u64* gsum = score_sum + (size_t)b * S;
    unsigned* ghist = hist + (size_t)b * S * C;
    const int lane = threadIdx.x & (MAS_WAVE - 1);
    const int wave = threadIdx.x / MAS_WAVE;

    for (int it = 0; it < kTileH / 4; ++it) {
        const int y = ty * kTileH + it * 4 + wave;
        if (y >= H) break;
        int xs[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xs[k] = tx * kTileW + (VEC ? (lane * 4 + k) : (k * MAS_WAVE + lane));
            ok[k] = xs[k] < W;
        }
        float b1[4], b2[4];
        int a1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { b1[k] = -__builtin_inff(); b2[k] = -__builtin_inff(); a1[k] = 0; }
        const size_t row = (size_t)y * W;
        // ids first, so that their latency overlaps the logit loads
        int id[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            id[k] = ok[k] ? mas_load_id(sb, row + xs[k]) : -1;
            if (id[k] >= S) id[k] = -1;
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (EXACT || c < C) {
                float v[4];
                if (VEC) {
                    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok[0]) q = mas_load_stream4(zb + (size_t)c * HW + row + xs[0]);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = ok[k] ? zb[(size_t)c * HW + row + xs[k]] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // top-2 as two medians (single_pass.hip:ring_consume): same values, ties keep the lowest class
                    const bool g1 = v[k] > b1[k];
                    b2[k] = __builtin_amdgcn_fmed3f(b1[k], b2[k], v[k]);
                    a1[k] = g1 ? c : a1[k];
                    b1[k] = __builtin_amdgcn_fmed3f(b1[k], v[k], 3.402823466e+38f);
                }
            }
        }
        mas_u64 q[4];
        float margin[4];
        mas_bvsb_quad(b1, b2, invT, margin);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = margin[k];
            if (cls_w) v = v * s_w[a1[k]];
            q[k] = mas_fix(v, MAS_SCORE_FRAC);
        }
        const bool same_id = (id[0] == id[1]) && (id[1] == id[2]) && (id[2] == id[3]);
        if (same_id) {
            if (id[0] >= 0) {
                const mas_u64 qs = (q[0] + q[1]) + (q[2] + q[3]);
                const int s = table_slot(t.keys, id[0]);
                const bool same_a = (a1[0] == a1[1]) && (a1[1] == a1[2]) && (a1[2] == a1[3]);
                if (s >= 0) {
                    lds_add(&t.sum[s], qs);
                    if (same_a) lds_add(&t.hist[s * C + a1[0]], 4u);
                    else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) lds_add(&t.hist[s * C + a1[k]], 1u);
                    }
                } else {
                    atomicAdd(&gsum[id[0]], qs);
#pragma unroll
                    for (int k = 0; k < 4; ++k) atomicAdd(&ghist[(size_t)id[0] * C + a1[k]], 1u);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (id[k] < 0) continue;
                const int s = table_slot(t.keys, id[k]);
                if (s >= 0) {
                    lds_add(&t.sum[s], q[k]);
                    lds_add(&t.hist[s * C + a1[k]], 1u);
                } else {
                    atomicAdd(&gsum[id[k]], q[k]);
                    atomicAdd(&ghist[(size_t)id[k] * C + a1[k]], 1u);
                }
            }
        }
    }
    __syncthreads();
    // flush the table: one global atomic per touched (region, field)
    for (int i = threadIdx.x; i < kSlots; i += kThreads) {
        const int id = t.keys[i];
        if (id >= 0 && t.sum[i]) atomicAdd(&gsum[id], t.sum[i]);
    }
    for (int i = threadIdx.x; i < kSlots * C; i += kThreads) {
        const unsigned n = t.hist[i];
        if (n) {
            const int s = i / C;
            atomicAdd(&ghist[(size_t)t.keys[s] * C + (i - s * C)], n);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3 tail + K4 ban
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_region_finalize(const mas_u64* __restrict__ score_sum,
                                                               const unsigned* __restrict__ hist, long long n_regions,
                                                               int C, int ban_class, float* __restrict__ score,
                                                               int* __restrict__ dominant, unsigned* __restrict__ count,
                                                               long long* __restrict__ hist_i64) {
    // rows of C histogram words are staged through LDS with flat coalesced loads (see k_region_finalize_weighted)
    extern __shared__ __attribute__((aligned(16))) unsigned char fin_smem[];
    unsigned* s_hist = reinterpret_cast<unsigned*>(fin_smem);
    const int CP = C | 1;
    const long long r0 = (long long)blockIdx.x * kThreads;
    const long long rows = (n_regions - r0) < kThreads ? (n_regions - r0) : kThreads;
    const long long words = rows * C;
    for (long long i = threadIdx.x; i < words; i += kThreads) {
        const int row = (int)(i / C), col = (int)(i - (long long)row * C);
        const unsigned hv = hist[r0 * C + i];
        s_hist[row * CP + col] = hv;
        if (hist_i64) hist_i64[r0 * C + i] = (long long)hv;
    }
    __syncthreads();
    if ((long long)threadIdx.x >= rows) return;
    const long long r = r0 + threadIdx.x;
    const unsigned* h = s_hist + threadIdx.x * CP;
    unsigned long long n = 0;
    unsigned best = 0;
    int arg = 0;
    for (int c = 0; c < C; ++c) {
        const unsigned v = h[c];
        n += v;
        if (v > best) { best = v; arg = c; }     // strict: first maximum wins, empty region -> class 0
    }
    float s = 0.0f;
    if (n) s = mas_fixed_mean(score_sum[r], n, MAS_SCORE_FRAC);
    if (ban_class >= 0 && arg == ban_class) s = 0.0f;
    score[r] = s;
    if (dominant) dominant[r] = arg;
    if (count) count[r] = (unsigned)n;
}

inline size_t accum_smem_bytes(int C) {
    return (sizeof(mas_u64) + sizeof(int)) * kSlots + sizeof(float) * MAS_MAX_CLASSES + sizeof(unsigned) * kSlots * (size_t)C;
}

template <int CT, bool EXACT, typename IdT>
int launch_accum(const float* z, const void* spx, const float* cls_w, int B, int C, int H, int W, int S, float invT,
                 mas_u64* score_sum, unsigned* hist, hipStream_t st) {
    const int tiles_x = (W + kTileW - 1) / kTileW;
    const int tiles_y = (H + kTileH - 1) / kTileH;
    const long long nblk = (long long)B * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
    const bool vec = (W % 4 == 0) && (((uintptr_t)z & 15) == 0);
    const size_t smem = accum_smem_bytes(C);
    const IdT* ids = static_cast<const IdT*>(spx);
    if (vec)
        hipLaunchKernelGGL((k_bvsb_region_accum<CT, EXACT, IdT, true>), dim3((unsigned)nblk), dim3(kThreads), smem, st, z, ids,
                           cls_w, C, H, W, S, invT, tiles_x, tiles_y, score_sum, hist);
    else
        hipLaunchKernelGGL((k_bvsb_region_accum<CT, EXACT, IdT, false>), dim3((unsigned)nblk), dim3(kThreads), smem, st, z, ids,
                           cls_w, C, H, W, S, invT, tiles_x, tiles_y, score_sum, hist);
    return mas_launch_status();
}

template <int CT, bool EXACT>
int dispatch_accum_ids(const float* z, const void* spx, int spx_dtype, const float* cls_w, int B, int C, int H, int W, int S,
                       float invT, mas_u64* score_sum, unsigned* hist, hipStream_t st) {
    switch (spx_dtype) {
        case MAS_ID_I64: return launch_accum<CT, EXACT, long long>(z, spx, cls_w, B, C, H, W, S, invT, score_sum, hist, st);
        case MAS_ID_I32: return launch_accum<CT, EXACT, int>(z, spx, cls_w, B, C, H, W, S, invT, score_sum, hist, st);
        case MAS_ID_U16: return launch_accum<CT, EXACT, unsigned short>(z, spx, cls_w, B, C, H, W, S, invT, score_sum, hist, st);
        default: return MAS_ERR_DTYPE;
    }
}

template <int CT, bool EXACT>
int launch_prob_sum(const float* z, int B, int C, int HW, float invT, mas_u64* prob_sum, hipStream_t st) {
    const int chunk_px = kThreads * 4;
    int bpi = 2048 / B;
    const int max_bpi = (HW + chunk_px - 1) / chunk_px;
    const int min_bpi = (max_bpi + 59) / 60;      // <= 60 iterations x 4 px per thread: 32-bit accumulators stay exact
    if (bpi < min_bpi) bpi = min_bpi;
    if (bpi > max_bpi) bpi = max_bpi;
    if (bpi < 1) bpi = 1;
    const bool vec = (HW % 4 == 0) && (((uintptr_t)z & 15) == 0);
    if constexpr (EXACT) if (vec && HW >= 8 * chunk_px) {
        // one round of workgroups at 2 waves/SIMD: 512 workgroups in all, <= 60 chunks each (32-bit accumulators stay exact)
        int rbpi = 512 / B > 0 ? 512 / B : 1;
        if (rbpi < min_bpi) rbpi = min_bpi;
        if (rbpi > max_bpi) rbpi = max_bpi;
        const int iters = (max_bpi + rbpi - 1) / rbpi;
        hipLaunchKernelGGL((k_class_prob_sum_ring<CT, EXACT>), dim3((unsigned)(B * rbpi)), dim3(kThreads), 0, st, z, C, HW, invT, prob_sum,
                           rbpi, iters);
        return mas_launch_status();
    }
    if (vec)
        hipLaunchKernelGGL((k_class_prob_sum<CT, EXACT, true>), dim3((unsigned)(B * bpi)), dim3(kThreads), 0, st, z, C, HW, invT,
                           prob_sum, bpi);
    else
        hipLaunchKernelGGL((k_class_prob_sum<CT, EXACT, false>), dim3((unsigned)(B * bpi)), dim3(kThreads), 0, st, z, C, HW, invT,
                           prob_sum, bpi);
    return mas_launch_status();
}

}  // namespace

extern "C" int mas_class_prob_sum(const float* z, int B, int C, int H, int W, float invT, uint64_t* prob_sum, void* stream) {
    if (!z || !prob_sum) return MAS_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL / 2) return MAS_ERR_SHAPE;
    if (C < 2 || C > MAS_MAX_CLASSES) return MAS_ERR_CLASSES;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mas_u64* ps = reinterpret_cast<mas_u64*>(prob_sum);
    const int HW = H * W;
    switch (C) {
        case 19: return launch_prob_sum<19, true>(z, B, C, HW, invT, ps, st);
        case 20: return launch_prob_sum<20, true>(z, B, C, HW, invT, ps, st);
        case 21: return launch_prob_sum<21, true>(z, B, C, HW, invT, ps, st);
        default: return launch_prob_sum<MAS_MAX_CLASSES, false>(z, B, C, HW, invT, ps, st);
    }
}

extern "C" int mas_bvsb_region_accum(const float* z, const void* spx, int spx_dtype, const float* cls_w, int B, int C, int H,
                                     int W, int S, float invT, uint64_t* score_sum, uint32_t* hist, void* stream) {
    if (!z || !spx || !score_sum || !hist) return MAS_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || S <= 0 || (long long)H * W > (1LL << 23)) return MAS_ERR_SHAPE;
    if (C < 2 || C > MAS_MAX_CLASSES) return MAS_ERR_CLASSES;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mas_u64* ss = reinterpret_cast<mas_u64*>(score_sum);
    switch (C) {
        case 19: return dispatch_accum_ids<19, true>(z, spx, spx_dtype, cls_w, B, C, H, W, S, invT, ss, hist, st);
        case 20: return dispatch_accum_ids<20, true>(z, spx, spx_dtype, cls_w, B, C, H, W, S, invT, ss, hist, st);
        case 21: return dispatch_accum_ids<21, true>(z, spx, spx_dtype, cls_w, B, C, H, W, S, invT, ss, hist, st);
        default: return dispatch_accum_ids<MAS_MAX_CLASSES, false>(z, spx, spx_dtype, cls_w, B, C, H, W, S, invT, ss, hist, st);
    }
}

extern "C" int mas_region_finalize(const uint64_t* score_sum, const uint32_t* hist, int64_t n_regions, int C, int ban_class,
                                   float* score, int32_t* dominant, uint32_t* count, int64_t* hist_i64, void* stream) {
    if (!score_sum || !hist || !score) return MAS_ERR_NULL;
    if (n_regions <= 0) return MAS_ERR_SHAPE;
    if (C < 1 || C > MAS_MAX_CLASSES) return MAS_ERR_CLASSES;
    const long long nblk = (n_regions + kThreads - 1) / kThreads;
    if (nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
    hipLaunchKernelGGL(k_region_finalize, dim3((unsigned)nblk), dim3(kThreads), sizeof(unsigned) * (size_t)kThreads * (C | 1),
                       static_cast<hipStream_t>(stream),
                       reinterpret_cast<const mas_u64*>(score_sum), hist, (long long)n_regions, C, ban_class, score, dominant,
                       count, reinterpret_cast<long long*>(hist_i64));
    return mas_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Class weights between the two halves of a round, on the device (no host round trip in the round's tail).
// One workgroup per class.  The per-batch integer sums and their f64 terms are independent and computed by all
// threads; the f64 accumulation over batches is strictly sequential IN BATCH ORDER (the reference adds per-batch
// means in loader order, my_bvsb_predclsbal_pwr_banignore.py:42-45), done by thread 0 from LDS.  Every f64
// operation is a single IEEE operation (no contraction, correctly rounded division), so the result equals
// oracle/exact.c:exact_class_weight and engine.class_weight_from_sums bit for bit.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kCwChunk = 1024;

__global__ __launch_bounds__(kThreads) void k_class_weight(const mas_u64* __restrict__ prob_sum, int n_img, int C,
                                                            long long hw, int batch_size, int n_batches, double coeff,
                                                            double* __restrict__ cum, float* __restrict__ cls_w,
                                                            unsigned* __restrict__ w31) {
    __shared__ double s_term[kCwChunk];
    const int c = blockIdx.x;
    double acc = 0.0;
    for (int b0 = 0; b0 < n_batches; b0 += kCwChunk) {
        const int nb = (n_batches - b0) < kCwChunk ? (n_batches - b0) : kCwChunk;
        for (int j = threadIdx.x; j < nb; j += kThreads) {
            const long long i0 = (long long)(b0 + j) * batch_size;
            long long i1 = i0 + batch_size;
            if (i1 > n_img) i1 = n_img;
            mas_u64 s = 0;
            for (long long i = i0; i < i1; ++i) s += prob_sum[i * C + c];
            const long long n = i1 - i0;
            s_term[j] = n > 0 ? ((double)s / 8388608.0) / ((double)n * (double)hw) : -1.0;     // terms are >= 0; -1 marks "no image"
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int j = 0; j < nb; ++j)
                if (s_term[j] >= 0.0) acc += s_term[j];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double m = acc / (double)n_batches;
        const double t = coeff * m + 1.0;
        const float w = (float)(1.0 / (t * t));
        cum[c] = m;
        cls_w[c] = w;
        if (w31) w31[c] = (unsigned)floor((double)w * 2147483648.0);
    }
}

extern "C" int mas_class_weight(const uint64_t* prob_sum, int n_img, int C, int64_t hw, int batch_size, int n_batches,
                                double coeff, double* cum, float* cls_w, uint32_t* w31, void* stream) {
    if (!prob_sum || !cum || !cls_w) return MAS_ERR_NULL;
    if (n_img <= 0 || hw <= 0 || batch_size <= 0 || n_batches <= 0) return MAS_ERR_SHAPE;
    if ((long long)n_batches * batch_size < n_img) return MAS_ERR_RANGE;
    if (C < 1 || C > MAS_MAX_CLASSES) return MAS_ERR_CLASSES;
    hipLaunchKernelGGL(k_class_weight, dim3((unsigned)C), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const mas_u64*>(prob_sum), n_img, C, (long long)hw, batch_size, n_batches, coeff, cum,
                       cls_w, w31);
    return mas_launch_status();
}
