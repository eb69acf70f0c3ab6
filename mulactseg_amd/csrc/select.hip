// select.hip -- K4: ordering of region scores and the budgeted selection walk, on the device.
//
// Reference semantics:
//   selected = sorted(scores, reverse=True)        active_selection/base.py:37
//     -> tuples (score, "img,lbl,spx", id) compared lexicographically, DESCENDING: higher score first;
//        equal scores: later path string first; then larger superpixel id first.
//   walk: cost += multi_hot_cls[img, id].sum() (fair counting) or 1; stop after the region that makes
//        cost > budget                               dataloader/region_active_dataset.py:31-73
//
// Here every region becomes ONE 64-bit key  [ order-preserving score bits : 32 | path rank : 32-b | id : b ]
// (b = bits needed for S-1; path rank = position of the image's path string in ascending order), so
// the reference's tuple order is the descending order of the keys.  0 is the "not in the pool" key.
// The sort itself is rocPRIM's device radix sort (a plain library primitive); keys, cost gather,
// prefix search and decode are the kernels below.  No host synchronisation, caller-owned workspace.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ unsigned order_bits(float f) {
    const unsigned b = mas_f2u(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unorder_bits(unsigned o) {
    return mas_u2f((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

inline int id_bits_for(int S) {
    int b = 1;
    while ((1LL << b) < S) ++b;
    return b;
}

__global__ __launch_bounds__(kThreads) void k_region_keys(const float* __restrict__ score, const unsigned char* __restrict__ valid,
                                                           const int* __restrict__ img_rank, long long n, int S, int id_bits,
                                                           mas_u64* __restrict__ keys) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const long long img = i / S;
    const int id = (int)(i - img * S);
    mas_u64 k = 0;
    if (!valid || valid[i]) {
        const unsigned lo = ((unsigned)img_rank[img] << id_bits) | (unsigned)id;
        k = ((mas_u64)order_bits(score[i]) << 32) | lo;
    }
    keys[i] = k;
}

__global__ __launch_bounds__(kThreads) void k_walk_cost(const mas_u64* __restrict__ keys, long long n,
                                                         const unsigned char* __restrict__ region_cost,
                                                         const int* __restrict__ img_of_rank, int S, int id_bits,
                                                         unsigned* __restrict__ cost) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const mas_u64 k = keys[i];
    unsigned c = 0;
    if (k) {
        c = 1;
        if (region_cost) {
            const unsigned lo = (unsigned)k;
            const int id = (int)(lo & ((1u << id_bits) - 1u));
            const int img = img_of_rank[lo >> id_bits];
            c = (unsigned)region_cost[(long long)img * S + id];
        }
    }
    cost[i] = c;
}

// n_selected = min over i of (i+1) with prefix[i] > budget, else the number of valid keys
__global__ __launch_bounds__(kThreads) void k_walk_find(const mas_u64* __restrict__ keys, const unsigned* __restrict__ prefix,
                                                         long long n, unsigned long long budget,
                                                         unsigned long long* __restrict__ n_selected) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const bool valid = keys[i] != 0;
    const bool prev_valid = (i == 0) ? true : (keys[i - 1] != 0);
    const bool crossed = valid && (unsigned long long)prefix[i] > budget;
    const bool prev_crossed = (i > 0) && ((unsigned long long)prefix[i - 1] > budget);
    if (crossed && !prev_crossed) atomicMin(n_selected, (unsigned long long)(i + 1));   // first crossing
    if (!valid && prev_valid) atomicMin(n_selected, (unsigned long long)i);            // end of the valid prefix
    if (i == n - 1 && valid) atomicMin(n_selected, (unsigned long long)n);
}

__global__ __launch_bounds__(kThreads) void k_walk_emit(const mas_u64* __restrict__ keys, const unsigned long long* __restrict__ n_selected,
                                                         long long max_out, const int* __restrict__ img_of_rank, int id_bits,
                                                         int* __restrict__ sel_img, int* __restrict__ sel_id,
                                                         float* __restrict__ sel_score) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= max_out) return;
    int img = -1, id = -1;
    float s = 0.0f;
    if ((unsigned long long)i < *n_selected) {
        const mas_u64 k = keys[i];
        const unsigned lo = (unsigned)k;
        id = (int)(lo & ((1u << id_bits) - 1u));
        img = img_of_rank[lo >> id_bits];
        s = unorder_bits((unsigned)(k >> 32));
    }
    sel_img[i] = img;
    sel_id[i] = id;
    sel_score[i] = s;
}

__global__ void k_set_u64(unsigned long long* p, unsigned long long v) { *p = v; }

// score[r] <- (dominant[r] == ban_class ? 0 : score[r]) * (cls_w ? cls_w[dominant[r]] : 1)
__global__ __launch_bounds__(kThreads) void k_region_reweight(float* __restrict__ score, const int* __restrict__ dominant,
                                                               long long n, int ban_class, const float* __restrict__ cls_w) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int d = dominant[i];
    float s = score[i];
    if (d == ban_class) s = 0.0f;
    if (cls_w) s = cls_w[d] * s;
    score[i] = s;
}

// counts[c] += #regions with dominant class c (LDS histogram, one 64-bit atomic per class per block)
__global__ __launch_bounds__(kThreads) void k_dominant_hist(const int* __restrict__ dominant, long long n, int C,
                                                             mas_u64* __restrict__ counts) {
    __shared__ unsigned s_cnt[MAS_MAX_CLASSES];
    if (threadIdx.x < MAS_MAX_CLASSES) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const int d = dominant[i];
        if (d >= 0 && d < C) atomicAdd(&s_cnt[d], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < C && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (mas_u64)s_cnt[threadIdx.x]);
}

// min over the non-zero entries and max over all entries, as order-preserving bit patterns
__global__ __launch_bounds__(kThreads) void k_minmax_nonzero(const float* __restrict__ u, long long n, unsigned* __restrict__ mm) {
    unsigned lo = 0xffffffffu, hi = 0u;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const float v = u[i];
        const unsigned o = order_bits(v);
        if (v != 0.0f) lo = o < lo ? o : lo;
        hi = o > hi ? o : hi;
    }
#pragma unroll
    for (int off = MAS_WAVE / 2; off > 0; off >>= 1) {
        const unsigned l2 = __shfl_down(lo, off, MAS_WAVE), h2 = __shfl_down(hi, off, MAS_WAVE);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & (MAS_WAVE - 1)) == 0) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}

__global__ void k_minmax_init(unsigned* mm) { mm[0] = 0xffffffffu; mm[1] = 0u; }

// u <- (u - min_nonzero) / (max - min_nonzero)      (my_bvsb.py:79-81)
__global__ __launch_bounds__(kThreads) void k_minmax_apply(float* __restrict__ u, long long n, const unsigned* __restrict__ mm) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float mn = unorder_bits(mm[0]);
    const float mx = unorder_bits(mm[1]) - mn;
    u[i] = (u[i] - mn) / mx;
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

size_t sort_temp_bytes(long long n) {
    size_t bytes = 0;
    mas_u64* nul = nullptr;
    (void)rocprim::radix_sort_keys_desc(nullptr, bytes, nul, nul, (size_t)n, 0, 64, (hipStream_t)0);
    return bytes;
}
size_t scan_temp_bytes(long long n) {
    size_t bytes = 0;
    unsigned* nul = nullptr;
    (void)rocprim::inclusive_scan(nullptr, bytes, nul, nul, (size_t)n, rocprim::plus<unsigned>(), (hipStream_t)0);
    return bytes;
}

}  // namespace

extern "C" size_t mas_select_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    const size_t t = sort_temp_bytes(n), u = scan_temp_bytes(n);
    return align_up(t > u ? t : u) + 2 * align_up(sizeof(unsigned) * (size_t)n) + 256;
}

extern "C" int mas_region_keys(const float* score, const uint8_t* valid, const int32_t* img_rank, int64_t n_img, int S,
                               uint64_t* keys, void* stream) {
    if (!score || !img_rank || !keys) return MAS_ERR_NULL;
    if (n_img <= 0 || S <= 0) return MAS_ERR_SHAPE;
    const int b = id_bits_for(S);
    if (b > 24 || n_img > (1LL << (32 - b))) return MAS_ERR_RANGE;
    const long long n = (long long)n_img * S;
    const long long nblk = (n + kThreads - 1) / kThreads;
    if (nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
    hipLaunchKernelGGL(k_region_keys, dim3((unsigned)nblk), dim3(kThreads), 0, static_cast<hipStream_t>(stream), score, valid,
                       img_rank, n, S, b, reinterpret_cast<mas_u64*>(keys));
    return mas_launch_status();
}

extern "C" int mas_sort_keys_desc(const uint64_t* keys_in, int64_t n, uint64_t* keys_out, void* workspace, size_t ws_bytes,
                                  void* stream) {
    if (!keys_in || !keys_out || !workspace) return MAS_ERR_NULL;
    if (n <= 0) return MAS_ERR_SHAPE;
    size_t need = sort_temp_bytes(n);
    if (ws_bytes < need) return MAS_ERR_WORKSPACE;
    hipError_t e = rocprim::radix_sort_keys_desc(workspace, need, reinterpret_cast<const mas_u64*>(keys_in),
                                                 reinterpret_cast<mas_u64*>(keys_out), (size_t)n, 0, 64,
                                                 static_cast<hipStream_t>(stream));
    return e == hipSuccess ? mas_launch_status() : (int)e;
}

extern "C" int mas_budget_walk(const uint64_t* sorted_keys, int64_t n, const uint8_t* region_cost, const int32_t* img_of_rank,
                               int S, int64_t budget, int64_t max_out, int64_t* n_selected, int32_t* sel_img, int32_t* sel_id,
                               float* sel_score, void* workspace, size_t ws_bytes, void* stream) {
    if (!sorted_keys || !img_of_rank || !n_selected || !sel_img || !sel_id || !sel_score || !workspace) return MAS_ERR_NULL;
    if (n <= 0 || S <= 0 || max_out <= 0 || budget < 0) return MAS_ERR_SHAPE;
    if (n > (1LL << 24)) return MAS_ERR_RANGE;   // 32-bit prefix sums: n * 255 < 2^32
    if (ws_bytes < mas_select_workspace_bytes(n)) return MAS_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int b = id_bits_for(S);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    size_t temp = scan_temp_bytes(n);
    const size_t t = sort_temp_bytes(n);
    const size_t temp_region = align_up(t > temp ? t : temp);
    unsigned* cost = reinterpret_cast<unsigned*>(ws + temp_region);
    unsigned* prefix = reinterpret_cast<unsigned*>(ws + temp_region + align_up(sizeof(unsigned) * (size_t)n));
    const mas_u64* keys = reinterpret_cast<const mas_u64*>(sorted_keys);
    unsigned long long* nsel = reinterpret_cast<unsigned long long*>(n_selected);
    const long long nblk = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, st, nsel, (unsigned long long)n);
    hipLaunchKernelGGL(k_walk_cost, dim3((unsigned)nblk), dim3(kThreads), 0, st, keys, (long long)n, region_cost, img_of_rank, S, b, cost);
    hipError_t e = rocprim::inclusive_scan(ws, temp, cost, prefix, (size_t)n, rocprim::plus<unsigned>(), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_walk_find, dim3((unsigned)nblk), dim3(kThreads), 0, st, keys, prefix, (long long)n,
                       (unsigned long long)budget, nsel);
    const long long oblk = (max_out + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_walk_emit, dim3((unsigned)oblk), dim3(kThreads), 0, st, keys, nsel, (long long)max_out, img_of_rank, b,
                       sel_img, sel_id, sel_score);
    return mas_launch_status();
}

extern "C" int mas_minmax_normalize(float* scores, int64_t n, uint32_t* scratch2, void* stream) {
    if (!scores || !scratch2) return MAS_ERR_NULL;
    if (n <= 0) return MAS_ERR_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    long long nblk = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(1), 0, st, scratch2);
    hipLaunchKernelGGL(k_minmax_nonzero, dim3((unsigned)(nblk > 1024 ? 1024 : nblk)), dim3(kThreads), 0, st, scores, (long long)n, scratch2);
    hipLaunchKernelGGL(k_minmax_apply, dim3((unsigned)nblk), dim3(kThreads), 0, st, scores, (long long)n, scratch2);
    return mas_launch_status();
}

extern "C" int mas_region_reweight(float* score, const int32_t* dominant, int64_t n, int ban_class, const float* cls_w, void* stream) {
    if (!score || !dominant) return MAS_ERR_NULL;
    if (n <= 0) return MAS_ERR_SHAPE;
    const long long nblk = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_region_reweight, dim3((unsigned)nblk), dim3(kThreads), 0, static_cast<hipStream_t>(stream), score, dominant,
                       (long long)n, ban_class, cls_w);
    return mas_launch_status();
}

extern "C" int mas_dominant_hist(const int32_t* dominant, int64_t n, int C, uint64_t* counts, void* stream) {
    if (!dominant || !counts) return MAS_ERR_NULL;
    if (n <= 0) return MAS_ERR_SHAPE;
    if (C < 1 || C > MAS_MAX_CLASSES) return MAS_ERR_CLASSES;
    long long nblk = (n + kThreads - 1) / kThreads;
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(k_dominant_hist, dim3((unsigned)nblk), dim3(kThreads), 0, static_cast<hipStream_t>(stream), dominant,
                       (long long)n, C, reinterpret_cast<mas_u64*>(counts));
    return mas_launch_status();
}
