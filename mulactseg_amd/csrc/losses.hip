// losses.hip -- stage-1 partial-label losses (K5 merged-positive CE, K6 group/MIL max-pool CE) for gfx950.
//
//   k_target_bits        multi-hot target rows u8[rows, cols] -> one u32 bit mask per superpixel.
//   k_partial_loss_fwd   ONE streaming pass over mask + ids (+ logits of the selected pixels only):
//                        per-pixel softmax in registers, fixed-point CE sums, and the segmented
//                        per-(superpixel, class) max with arg-pixel through a per-workgroup LDS table
//                        of packed (prob bits << 32 | ~pixel) words combined with 64-bit atomic max.
//   k_group_finalize     -log(max + eps) over the (superpixel, class) table.
//   k_loss_values        fixed-point sums -> the reference's mean-normalised f32 losses.
//   k_loss_scales        upstream gradients / (1 + n) -> per-loss scale factors (no host sync).
//   k_partial_loss_bwd   one pass writing dz[N,C,H,W] completely (zeros off-mask), softmax recomputed,
//                        group-loss gradient gathered from the arg-pixel table (no atomics).
//
// Reference semantics: trainer/active_joint_multi_predignore_lossdecomp.py:21-72 (OnehotCEMultihotChoice),
// trainer/active_joint_multi_predignore_mclossablation2.py:22-79 (GroupMultiLabelCE_onlymulti),
// trainer/active_joint_multi_predignore.py:21-128, utils/loss.py:91-141,543-588 (base classes).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTileW = 256;
constexpr int kTileH = 4;
constexpr int kLogSlots = 6;
constexpr int kSlots = 1 << kLogSlots;

// accumulator slots of the u64 `acc` array
enum { ACC_SUM_CE = 0, ACC_SUM_MC = 1, ACC_N_CE = 2, ACC_N_MC = 3, ACC_N_EMPTY = 4, ACC_SUM_GROUP = 5, ACC_N_GROUP = 6 };

__device__ __forceinline__ mas_u64 pack_max(float p, unsigned pix) {
    return ((mas_u64)mas_f2u(p) << 32) | (mas_u64)(0xffffffffu - pix);
}

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
    return -1;
}

__global__ __launch_bounds__(kThreads) void k_target_bits(const unsigned char* __restrict__ tgt, long long n_rows,
                                                           int cols_stored, int cols_used, unsigned* __restrict__ bits) {
    const long long r = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (r >= n_rows) return;
    const unsigned char* t = tgt + r * cols_stored;
    unsigned b = 0;
    for (int c = 0; c < cols_used; ++c) b |= (t[c] ? 1u : 0u) << c;
    bits[r] = b;
}


// ---- logits of one pixel -------------------------------------------------------------------------------------------
// LOWRES: the kernels read the model's quarter-resolution logits zq [C,h,w] and evaluate the final
// F.interpolate(bilinear, align_corners=False) of models/segmentation/utils.py:25 for the pixel in registers -- the
// arithmetic of csrc/upsample.hip (ATen's area_pixel_compute_source_index), so the values equal those of the materialised
// full-resolution tensor bit for bit; that tensor (189 MB per training batch) is then never written or read.
struct LTap { int i0, i1; float l0, l1; };

__device__ __forceinline__ LTap loss_tap(float scale, int o, int n_in) {
    float s = scale * ((float)o + 0.5f) - 0.5f;
    s = s < 0.0f ? 0.0f : s;
    LTap t;
    t.i0 = (int)s;
    t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
    t.l1 = s - (float)t.i0;
    t.l0 = 1.0f - t.l1;
    return t;
}

struct LowRes { int h, w; float sh, sw; };

template <int CT, bool EXACT, bool LOWRES>
__device__ __forceinline__ void load_logits(const float* __restrict__ zb, int C, int HW, size_t pix, int y, int x, const LowRes& lr,
                                            float (&v)[CT], LTap& ty, LTap& tx) {
    if (LOWRES) {
        ty = loss_tap(lr.sh, y, lr.h);
        tx = loss_tap(lr.sw, x, lr.w);
        const int hw = lr.h * lr.w;
        const int o00 = ty.i0 * lr.w + tx.i0, o01 = ty.i0 * lr.w + tx.i1, o10 = ty.i1 * lr.w + tx.i0, o11 = ty.i1 * lr.w + tx.i1;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (EXACT || c < C) {
                const float* q = zb + (size_t)c * hw;
                v[c] = ty.l0 * (tx.l0 * q[o00] + tx.l1 * q[o01]) + ty.l1 * (tx.l0 * q[o10] + tx.l1 * q[o11]);
            } else {
                v[c] = 0.f;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CT; ++c) v[c] = (EXACT || c < C) ? zb[(size_t)c * HW + pix] : 0.f;
    }
}

// signed fixed point (44 fractional bits) of one gradient contribution, round to nearest even; mirrored by oracle/exact.c
#define MAS_GRAD_FRAC 44
__device__ __forceinline__ long long grad_fix(float t) {
    union { double d; mas_u64 u; } s;
    s.u = (mas_u64)(1023 + MAS_GRAD_FRAC) << 52;
    return __double2ll_rn((double)t * s.d);
}

// Selected pixels are sparse (a few % of a crop, whole superpixels at a time).  Both scans therefore run in two phases
// per 16x256 tile: (1) every lane looks at the mask bytes of its pixels and the selected ones are COMPACTED into an LDS
// queue (wave ballot + one LDS counter add per wave); (2) the queue is processed densely, one selected pixel per lane, so
// the softmax / log arithmetic runs with all 64 lanes busy instead of whole waves executing it for a handful of lanes.
// Only the logits of selected pixels are ever loaded.  Sums are integers and the group table is a max, so the
// (non-deterministic) queue order cannot change any result.
constexpr int kTilePx = kTileW * kTileH;

__device__ __forceinline__ void queue_push(bool sel, unsigned short off, unsigned short* queue, int* qcount) {
    const unsigned long long bal = __ballot(sel);
    if (bal == 0) return;
    const int lane = threadIdx.x & (MAS_WAVE - 1);
    int base = 0;
    if (lane == 0) base = atomicAdd(qcount, (int)__popcll(bal));
    base = __shfl(base, 0, MAS_WAVE);
    if (sel) queue[base + (int)__popcll(bal & ((1ull << lane) - 1ull))] = off;
}

// phase 1 of both scans: compact the selected pixels of this tile; BWD additionally zero-fills dz for the whole tile
template <int CT, bool EXACT, bool VEC, bool BWD>
__device__ __forceinline__ void tile_compact(const unsigned char* __restrict__ mb, int C, int H, int W, int HW, int tx, int ty,
                                             unsigned short* queue, int* qcount, float* __restrict__ db) {
    const int lane = threadIdx.x & (MAS_WAVE - 1);
    const int wave = threadIdx.x / MAS_WAVE;
    for (int it = 0; it < kTileH / 4; ++it) {
        const int ry = it * 4 + wave;
        const int y = ty * kTileH + ry;
        if (y >= H) break;
        const size_t row = (size_t)y * W;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rx = VEC ? (lane * 4 + k) : (k * MAS_WAVE + lane);
            const int x = tx * kTileW + rx;
            const bool sel = (x < W) && (mb[row + (x < W ? x : 0)] != 0);
            queue_push(sel, (unsigned short)((ry << 8) | rx), queue, qcount);
        }
        if (BWD) {
            if (VEC) {
                const int x = tx * kTileW + lane * 4;
                if (x < W) {
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        if (EXACT || c < C) *reinterpret_cast<float4*>(db + (size_t)c * HW + row + x) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = tx * kTileW + k * MAS_WAVE + lane;
                    if (x < W) {
#pragma unroll
                        for (int c = 0; c < CT; ++c)
                            if (EXACT || c < C) db[(size_t)c * HW + row + x] = 0.0f;
                    }
                }
            }
        }
    }
}

template <int CT, bool EXACT, typename IdT, bool VEC, bool LOWRES>
__global__ __launch_bounds__(kThreads) void k_partial_loss_fwd(const float* __restrict__ z, const IdT* __restrict__ spx,
                                                                const unsigned char* __restrict__ mask,
                                                                const unsigned* __restrict__ bits, int C, int H, int W, int S,
                                                                float invT, int flags, int tiles_x, int tiles_y,
                                                                mas_u64* __restrict__ gmax, mas_u64* __restrict__ acc, const LowRes lr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mas_u64* t_max = reinterpret_cast<mas_u64*>(smem);                       // [kSlots * C]
    int* t_keys = reinterpret_cast<int*>(smem + sizeof(mas_u64) * kSlots * C);  // [kSlots]
    mas_u64* s_red = reinterpret_cast<mas_u64*>(smem + sizeof(mas_u64) * kSlots * C + sizeof(int) * kSlots);  // [4][5]
    int* qcount = reinterpret_cast<int*>(s_red + 4 * 5);
    unsigned short* queue = reinterpret_cast<unsigned short*>(qcount + 2);     // [kTilePx]

    const bool do_ce = flags & MAS_LOSS_CE;
    const bool do_group = flags & MAS_LOSS_GROUP;
    const bool only_multi = flags & MAS_LOSS_GROUP_ONLY_MULTI;
    const bool tce = flags & MAS_LOSS_TCE;          // `spx` holds class labels, the target of a pixel is its label, no epsilon
    if (threadIdx.x == 0) *qcount = 0;
    if (do_group) {
        for (int i = threadIdx.x; i < kSlots; i += kThreads) t_keys[i] = -1;
        for (int i = threadIdx.x; i < kSlots * C; i += kThreads) t_max[i] = 0;
    }
    __syncthreads();

    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int HW = H * W;
    const float* zb = z + (size_t)n * C * (LOWRES ? lr.h * lr.w : HW);
    const IdT* sb = spx + (size_t)n * HW;
    const unsigned char* mb = mask + (size_t)n * HW;
    const unsigned* bb = bits + (size_t)n * S;
    mas_u64* gm = gmax + (size_t)n * S * C;
    const int lane = threadIdx.x & (MAS_WAVE - 1);
    const int wave = threadIdx.x / MAS_WAVE;

    tile_compact<CT, EXACT, VEC, false>(mb, C, H, W, HW, tx, ty, queue, qcount, nullptr);
    __syncthreads();
    const int nq = *qcount;

    mas_u64 sum_ce = 0, sum_mc = 0;
    unsigned n_ce = 0, n_mc = 0, n_empty = 0;
    for (int i = threadIdx.x; i < nq; i += kThreads) {
        const unsigned off = queue[i];
        const int py = ty * kTileH + (int)(off >> 8), px = tx * kTileW + (int)(off & 255u);
        const size_t pix = (size_t)py * W + px;
        const int id = mas_load_id(sb, pix);
        if (id < 0 || id >= S) continue;
        float v[CT];
        LTap t_y, t_x;
        load_logits<CT, EXACT, LOWRES>(zb, C, HW, pix, py, px, lr, v, t_y, t_x);
        const unsigned Y = tce ? (1u << id) : bb[id];
        const int nb = __popc(Y);
        if (nb == 0) { n_empty += 1; continue; }
        {
            const float rinv = mas_softmax_regs<CT, EXACT>(v, C, invT);
#pragma unroll
            for (int c = 0; c < CT; ++c) v[c] = v[c] * rinv;
        }
        if (do_ce) {
            float pos = 0.0f;
#pragma unroll
            for (int c = 0; c < CT; ++c)
                if (EXACT || c < C) pos = ((Y >> c) & 1u) ? (pos + v[c]) : pos;
            // (temperature CE: no epsilon; a target probability below the smallest normal float -- a logit gap beyond 86 / T, outside the
            //  cosine head's range -- is held there, so neither the logarithm nor the backward's 1 / pos leaves the finite floats)
            const float l = -mas_logf(tce ? (pos < 1.17549435e-38f ? 1.17549435e-38f : pos) : pos + 1e-8f);
            const mas_u64 q = mas_fix(l, MAS_LOSS_FRAC);
            if (nb == 1) { sum_ce += q; n_ce += 1; } else { sum_mc += q; n_mc += 1; }
        }
        if (do_group && (!only_multi || nb > 1)) {
            const int s = table_slot(t_keys, id);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if ((EXACT || c < C) && ((Y >> c) & 1u)) {
                    const mas_u64 w = pack_max(v[c], (unsigned)pix);
                    if (s >= 0) atomicMax(&t_max[s * C + c], w);
                    else atomicMax(&gm[(size_t)id * C + c], w);
                }
            }
        }
    }

    // block reduction of the five scalar accumulators
    mas_u64 r[5] = {sum_ce, sum_mc, (mas_u64)n_ce, (mas_u64)n_mc, (mas_u64)n_empty};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int off = MAS_WAVE / 2; off > 0; off >>= 1) r[i] += __shfl_down(r[i], off, MAS_WAVE);
        if (lane == 0) s_red[wave * 5 + i] = r[i];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        mas_u64 a = 0;
        for (int w = 0; w < kThreads / MAS_WAVE; ++w) a += s_red[w * 5 + threadIdx.x];
        if (a) atomicAdd(&acc[threadIdx.x], a);
    }
    if (do_group) {
        for (int i = threadIdx.x; i < kSlots * C; i += kThreads) {
            const mas_u64 w = t_max[i];
            if (w) {
                const int s = i / C;
                atomicMax(&gm[(size_t)t_keys[s] * C + (i - s * C)], w);
            }
        }
    }
}

// acc[ACC_DONE]: workgroups of k_group_finalize that have added their sums (the fused entry point: the LAST one turns the sums into
// the loss values -- no k_loss_values launch)
constexpr int ACC_DONE = 7;
__device__ __forceinline__ void loss_values_of(const mas_u64* acc, int flags, const float* w, float* out);

__global__ __launch_bounds__(kThreads) void k_group_finalize(const mas_u64* __restrict__ gmax, long long n_entries,
                                                              mas_u64* __restrict__ acc, int flags, const float* __restrict__ w,
                                                              float* __restrict__ values) {
    __shared__ mas_u64 s_red[kThreads / MAS_WAVE][2];
    __shared__ int s_last;
    mas_u64 sum = 0, cnt = 0;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n_entries; i += (long long)gridDim.x * kThreads) {
        const mas_u64 w64 = gmax[i];
        const unsigned pb = (unsigned)(w64 >> 32);
        if (pb) {            // a (superpixel, class) with target bit set whose max prob is non-zero
            const float l = -mas_logf(mas_u2f(pb) + 1e-8f);
            sum += mas_fix(l, MAS_LOSS_FRAC);
            cnt += 1;
        }
    }
    const int lane = threadIdx.x & (MAS_WAVE - 1), wave = threadIdx.x / MAS_WAVE;
#pragma unroll
    for (int off = MAS_WAVE / 2; off > 0; off >>= 1) {
        sum += __shfl_down(sum, off, MAS_WAVE);
        cnt += __shfl_down(cnt, off, MAS_WAVE);
    }
    if (lane == 0) { s_red[wave][0] = sum; s_red[wave][1] = cnt; }
    __syncthreads();
    if (threadIdx.x < 2) {
        mas_u64 a = 0;
        for (int wv = 0; wv < kThreads / MAS_WAVE; ++wv) a += s_red[wv][threadIdx.x];
        if (a) atomicAdd(&acc[threadIdx.x == 0 ? ACC_SUM_GROUP : ACC_N_GROUP], a);
    }
    if (!values) return;
    // the last workgroup to arrive sees every other workgroup's sums: device-scope release (fence + counter add) / acquire
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const mas_u64 done = atomicAdd(&acc[ACC_DONE], (mas_u64)1);
        s_last = done == (mas_u64)(gridDim.x - 1);
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        __threadfence();
        mas_u64 a[8];
        for (int i = 0; i < 7; ++i) a[i] = __hip_atomic_load(&acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        loss_values_of(a, flags, w, values);
    }
}

// the fused forward's first launch: zero the accumulators and the (superpixel, class) table, and (targets != NULL) turn the
// multi-hot target rows into bit masks -- one launch instead of a memset and k_target_bits
__global__ __launch_bounds__(kThreads) void k_loss_prep(mas_u64* __restrict__ zero64, long long n_zero, const unsigned char* __restrict__ tgt,
                                                         long long n_rows, int cols_stored, int cols_used, unsigned* __restrict__ bits) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i < n_zero) zero64[i] = 0;
    if (tgt && i < n_rows) {
        const unsigned char* t = tgt + i * cols_stored;
        unsigned b = 0;
        for (int c = 0; c < cols_used; ++c) b |= (t[c] ? 1u : 0u) << c;
        bits[i] = b;
    }
}

// loss = (sum * 2^-32) / (1 + n) in f64, rounded once to f32; MAS_LOSS_TCE: the plain mean, / n (0 / 0 = NaN as torch's
// CrossEntropyLoss over no valid pixel)
__device__ __forceinline__ float loss_value(mas_u64 sum, mas_u64 n, int one = 1) {
    union { double d; mas_u64 u; } s;
    s.u = (mas_u64)(1023 - MAS_LOSS_FRAC) << 52;
    return (float)(((double)sum * s.d) / (double)(n + one));
}

// losses[0..2] = (ce, mc, group); with weights also losses[3] = (w_ce * ce + w_mc * mc) + w_group * group, every product and sum
// rounded once in f32 -- the operation order of `coeff * ce_loss + coeff_mc * mc_loss + coeff_gm * group_loss`
// (trainer/active_joint_multi_predignore_lossdecomp.py:104), so the value equals the torch expression bit for bit.
__device__ __forceinline__ void loss_values_of(const mas_u64* acc, int flags, const float* w, float* out) {
    if (flags & MAS_LOSS_TCE) {
        out[0] = loss_value(acc[ACC_SUM_CE], acc[ACC_N_CE], 0);
        out[1] = 0.0f;
        out[2] = 0.0f;
        if (w) out[3] = (w[0] * out[0] + w[1] * 0.0f) + w[2] * 0.0f;
        return;
    }
    float ce, mc;
    if (flags & MAS_LOSS_DECOMP) {
        ce = loss_value(acc[ACC_SUM_CE], acc[ACC_N_CE]);
        mc = loss_value(acc[ACC_SUM_MC], acc[ACC_N_MC]);
    } else {
        ce = loss_value(acc[ACC_SUM_CE] + acc[ACC_SUM_MC], acc[ACC_N_CE] + acc[ACC_N_MC]);
        mc = 0.0f;
    }
    const float gr = loss_value(acc[ACC_SUM_GROUP], acc[ACC_N_GROUP]);
    out[0] = ce;
    out[1] = mc;
    out[2] = gr;
    if (w) out[3] = (w[0] * ce + w[1] * mc) + w[2] * gr;
}

__global__ void k_loss_values(const mas_u64* __restrict__ acc, int flags, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    loss_values_of(acc, flags, nullptr, out);
}

__global__ void k_loss_values_weighted(const mas_u64* __restrict__ acc, int flags, const float* __restrict__ w, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    loss_values_of(acc, flags, w, out);
}

// scale_k = upstream_k / (1 + n_k) (f32 division, correctly rounded); with weights: grad[0] is dL/dtotal and
// scale_k = (dL/dtotal * w_k) / (1 + n_k) -- the chain rule through the weighted sum, in the order autograd applies it
__device__ __forceinline__ void loss_scales_of(const mas_u64* acc, const float* grad, const float* w, int flags, float* scale) {
    if (flags & MAS_LOSS_TCE) {
        scale[0] = (w ? grad[0] * w[0] : grad[0]) / (float)acc[ACC_N_CE];
        scale[1] = 0.0f;
        scale[2] = 0.0f;
        return;
    }
    const float g0 = w ? grad[0] * w[0] : grad[0], g1 = w ? grad[0] * w[1] : grad[1], g2 = w ? grad[0] * w[2] : grad[2];
    if (flags & MAS_LOSS_DECOMP) {
        scale[0] = g0 / (float)(acc[ACC_N_CE] + 1);
        scale[1] = g1 / (float)(acc[ACC_N_MC] + 1);
    } else {
        const float sc = g0 / (float)(acc[ACC_N_CE] + acc[ACC_N_MC] + 1);
        scale[0] = sc;
        scale[1] = sc;
    }
    scale[2] = g2 / (float)(acc[ACC_N_GROUP] + 1);
}

__global__ void k_loss_scales_weighted(const mas_u64* __restrict__ acc, const float* __restrict__ grad_total, const float* __restrict__ w,
                                       int flags, float* __restrict__ scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    loss_scales_of(acc, grad_total, w, flags, scale);
}

__global__ void k_loss_scales(const mas_u64* __restrict__ acc, const float* __restrict__ grad_out, int flags,
                              float* __restrict__ scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    loss_scales_of(acc, grad_out, nullptr, flags, scale);
}

// LOWRES: `dz` is the caller-zeroed int64 fixed-point accumulator dzq_fix [N,C,h,w] (MAS_GRAD_FRAC fractional bits): every
// selected pixel adds (gradient of its class-c logit) * (bilinear weight) into the four quarter-resolution elements its
// logits were interpolated from -- integer atomics, so the sums do not depend on the order; nothing is written at full
// resolution.
template <int CT, bool EXACT, typename IdT, bool VEC, bool LOWRES>
__global__ __launch_bounds__(kThreads) void k_partial_loss_bwd(const float* __restrict__ z, const IdT* __restrict__ spx,
                                                                const unsigned char* __restrict__ mask,
                                                                const unsigned* __restrict__ bits,
                                                                const mas_u64* __restrict__ gmax,
                                                                const float* __restrict__ scale, int C, int H, int W, int S,
                                                                float invT, int flags, int tiles_x, int tiles_y,
                                                                float* __restrict__ dz, const LowRes lr, const mas_u64* __restrict__ acc,
                                                                const float* __restrict__ grad, const float* __restrict__ gw) {
    __shared__ int qcount[2];
    __shared__ unsigned short queue[kTilePx];
    // LOWRES: the quarter-resolution footprint of a 4 x 256 tile (<= 4 rows x 72 columns per class for the x4 ratios of the
    // model) is accumulated in LDS first; only its non-zero entries go to memory (16-60x fewer global atomics)
    constexpr bool LACC = LOWRES && EXACT;
    constexpr int kLR = 4, kLC = 72;
    __shared__ mas_u64 s_acc[LACC ? CT * kLR * kLC : 1];
    const bool do_ce = flags & MAS_LOSS_CE;
    const bool do_group = flags & MAS_LOSS_GROUP;
    const bool only_multi = flags & MAS_LOSS_GROUP_ONLY_MULTI;
    if (threadIdx.x == 0) qcount[0] = 0;
    __syncthreads();
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int HW = H * W;
    const int hw_in = LOWRES ? lr.h * lr.w : HW;
    const float* zb = z + (size_t)n * C * hw_in;
    float* db = LOWRES ? nullptr : dz + (size_t)n * C * HW;
    mas_u64* qb = LOWRES ? reinterpret_cast<mas_u64*>(dz) + (size_t)n * C * hw_in : nullptr;
    const IdT* sb = spx + (size_t)n * HW;
    const unsigned char* mb = mask + (size_t)n * HW;
    const unsigned* bb = bits + (size_t)n * S;
    const mas_u64* gm = gmax + (size_t)n * S * C;
    // scale == NULL (the fused entry points): every workgroup forms the three scale factors itself from the accumulators of the
    // forward scan and the upstream gradient (three divisions on uniform values) -- the k_loss_scales launch is gone
    float sc3[3];
    if (scale) { sc3[0] = scale[0]; sc3[1] = scale[1]; sc3[2] = scale[2]; }
    else loss_scales_of(acc, grad, gw, flags, sc3);
    const float a_ce = sc3[0] * invT, a_mc = sc3[1] * invT, g6 = sc3[2] * invT;

    // phase 1: dz = 0 over the whole tile (streaming 16-B stores) while the selected pixels are compacted
    tile_compact<CT, EXACT, VEC, !LOWRES>(mb, C, H, W, HW, tx, ty, queue, qcount, db);
    __syncthreads();        // (waits for this workgroup's zero stores: the gradients below overwrite some of them)
    const int nq = qcount[0];
    int ry0 = 0, rx0 = 0;
    if (LACC) {
        if (nq == 0) return;                              // (uniform) nothing selected in this tile
        ry0 = loss_tap(lr.sh, ty * kTileH, lr.h).i0;
        rx0 = loss_tap(lr.sw, tx * kTileW, lr.w).i0;
        for (int i = threadIdx.x; i < CT * kLR * kLC; i += kThreads) s_acc[i] = 0;
        __syncthreads();
    }
    auto add_q = [&](mas_u64* q, int c, int iy, int ix, long long val) {
        const int r = iy - ry0, x = ix - rx0;
        if (LACC && r < kLR && x < kLC) atomicAdd(&s_acc[(c * kLR + r) * kLC + x], (mas_u64)val);
        else atomicAdd(q + (iy * lr.w + ix), (mas_u64)val);
    };

    // phase 2: gradients of the selected pixels, one pixel per lane
    for (int i = threadIdx.x; i < nq; i += kThreads) {
        const unsigned off = queue[i];
        const int py = ty * kTileH + (int)(off >> 8), px = tx * kTileW + (int)(off & 255u);
        const size_t pix = (size_t)py * W + px;
        const int id = mas_load_id(sb, pix);
        if (id < 0 || id >= S) continue;
        const unsigned Y = (flags & MAS_LOSS_TCE) ? (1u << id) : bb[id];
        const int nb = __popc(Y);
        if (nb == 0) continue;
        float v[CT];
        LTap t_y, t_x;
        load_logits<CT, EXACT, LOWRES>(zb, C, HW, pix, py, px, lr, v, t_y, t_x);
        {
            const float rinv = mas_softmax_regs<CT, EXACT>(v, C, invT);
#pragma unroll
            for (int c = 0; c < CT; ++c) v[c] = v[c] * rinv;
        }
        float coef = 0.0f, pos = 0.0f;
        if (do_ce) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
                if (EXACT || c < C) pos = ((Y >> c) & 1u) ? (pos + v[c]) : pos;
            coef = ((nb == 1) ? a_ce : a_mc) * (1.0f / ((flags & MAS_LOSS_TCE) ? (pos < 1.17549435e-38f ? 1.17549435e-38f : pos) : pos + 1e-8f));
        }
        // group loss: classes of Y whose arg-max pixel is this pixel
        unsigned A = 0;
        float u = 0.0f;
        float t[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) t[c] = 0.0f;
        if (do_group && (!only_multi || nb > 1)) {
            const unsigned key = 0xffffffffu - (unsigned)pix;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if ((EXACT || c < C) && ((Y >> c) & 1u)) {
                    const mas_u64 w = gm[(size_t)id * C + c];
                    if ((unsigned)w == key && (unsigned)(w >> 32) != 0u) {
                        A |= 1u << c;
                        t[c] = -(g6 / (v[c] + 1e-8f));
                        u = u + t[c] * v[c];
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (EXACT || c < C) {
                const float p = v[c];
                const float yj = ((Y >> c) & 1u) ? 1.0f : 0.0f;
                float d = coef * (p * (pos - yj));
                if (A) {
                    if ((A >> c) & 1u) d = d + t[c] * p;
                    d = d - p * u;
                }
                if (LOWRES) {
                    mas_u64* q = qb + (size_t)c * hw_in;
                    add_q(q, c, t_y.i0, t_x.i0, grad_fix(d * (t_y.l0 * t_x.l0)));
                    add_q(q, c, t_y.i0, t_x.i1, grad_fix(d * (t_y.l0 * t_x.l1)));
                    add_q(q, c, t_y.i1, t_x.i0, grad_fix(d * (t_y.l1 * t_x.l0)));
                    add_q(q, c, t_y.i1, t_x.i1, grad_fix(d * (t_y.l1 * t_x.l1)));
                } else {
                    db[(size_t)c * HW + pix] = d;
                }
            }
        }
    }
    if (LACC) {
        __syncthreads();
        for (int i = threadIdx.x; i < CT * kLR * kLC; i += kThreads) {
            const mas_u64 val = s_acc[i];
            if (val) {
                const int c = i / (kLR * kLC), rem = i - c * (kLR * kLC);
                const int r = rem / kLC, x = rem - r * kLC;
                atomicAdd(qb + (size_t)c * hw_in + (size_t)(ry0 + r) * lr.w + (rx0 + x), val);
            }
        }
    }
}

inline size_t fwd_smem_bytes(int C) {
    return sizeof(mas_u64) * kSlots * (size_t)C + sizeof(int) * kSlots + sizeof(mas_u64) * 4 * 5 + sizeof(int) * 2 +
           sizeof(unsigned short) * kTilePx;
}

struct LossArgs {
    const float* z; const void* spx; const unsigned char* mask; const unsigned* bits;
    int N, C, H, W, S; float invT; int flags;
    mas_u64* gmax; mas_u64* acc; const float* scale; float* dz;
    int h = 0, w = 0;           // > 0: `z` is the quarter-resolution tensor [N,C,h,w] (LOWRES kernels)
    const float* grad = nullptr; const float* gw = nullptr;     // backward with scale == NULL: upstream gradient(s) and optional weights
};

template <int CT, bool EXACT, typename IdT>
int launch_loss(const LossArgs& a, bool backward, hipStream_t st) {
    const int tiles_x = (a.W + kTileW - 1) / kTileW;
    const int tiles_y = (a.H + kTileH - 1) / kTileH;
    const long long nblk = (long long)a.N * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
    const IdT* ids = static_cast<const IdT*>(a.spx);
    const bool low = a.h > 0;
    const bool vec = (a.W % 4 == 0) && (low || ((((uintptr_t)a.z & 15) == 0) && (!backward || (((uintptr_t)a.dz & 15) == 0))));
    const dim3 grid((unsigned)nblk), block(kThreads);
    const LowRes lr{a.h, a.w, low ? (float)a.h / (float)a.H : 0.f, low ? (float)a.w / (float)a.W : 0.f};
#define MAS_LAUNCH_LOSS(VECV, LOWV)                                                                                                     \
    do {                                                                                                                                \
        if (!backward)                                                                                                                  \
            hipLaunchKernelGGL((k_partial_loss_fwd<CT, EXACT, IdT, VECV, LOWV>), grid, block, fwd_smem_bytes(a.C), st, a.z, ids, a.mask, \
                               a.bits, a.C, a.H, a.W, a.S, a.invT, a.flags, tiles_x, tiles_y, a.gmax, a.acc, lr);                        \
        else                                                                                                                            \
            hipLaunchKernelGGL((k_partial_loss_bwd<CT, EXACT, IdT, VECV, LOWV>), grid, block, 0, st, a.z, ids, a.mask, a.bits, a.gmax,   \
                               a.scale, a.C, a.H, a.W, a.S, a.invT, a.flags, tiles_x, tiles_y, a.dz, lr, a.acc, a.grad, a.gw);           \
    } while (0)
    if (low) { if (vec) MAS_LAUNCH_LOSS(true, true); else MAS_LAUNCH_LOSS(false, true); }
    else { if (vec) MAS_LAUNCH_LOSS(true, false); else MAS_LAUNCH_LOSS(false, false); }
#undef MAS_LAUNCH_LOSS
    return mas_launch_status();
}

template <int CT, bool EXACT>
int dispatch_loss_ids(const LossArgs& a, int spx_dtype, bool backward, hipStream_t st) {
    switch (spx_dtype) {
        case MAS_ID_I64: return launch_loss<CT, EXACT, long long>(a, backward, st);
        case MAS_ID_I32: return launch_loss<CT, EXACT, int>(a, backward, st);
        case MAS_ID_U16: return launch_loss<CT, EXACT, unsigned short>(a, backward, st);
        default: return MAS_ERR_DTYPE;
    }
}

int dispatch_loss(const LossArgs& a, int spx_dtype, bool backward, hipStream_t st) {
    if (a.N <= 0 || a.H <= 0 || a.W <= 0 || a.S <= 0 || (long long)a.H * a.W > 0x7fffffffLL / 2) return MAS_ERR_SHAPE;
    if (a.C < 2 || a.C > MAS_MAX_CLASSES) return MAS_ERR_CLASSES;
    switch (a.C) {
        case 19: return dispatch_loss_ids<19, true>(a, spx_dtype, backward, st);
        case 20: return dispatch_loss_ids<20, true>(a, spx_dtype, backward, st);
        case 21: return dispatch_loss_ids<21, true>(a, spx_dtype, backward, st);
        default: return dispatch_loss_ids<MAS_MAX_CLASSES, false>(a, spx_dtype, backward, st);
    }
}

}  // namespace

extern "C" int mas_target_bits(const uint8_t* targets, int64_t n_rows, int cols_stored, int cols_used, uint32_t* bits,
                               void* stream) {
    if (!targets || !bits) return MAS_ERR_NULL;
    if (n_rows <= 0 || cols_stored <= 0) return MAS_ERR_SHAPE;
    if (cols_used < 1 || cols_used > cols_stored || cols_used > MAS_MAX_CLASSES) return MAS_ERR_CLASSES;
    const long long nblk = (n_rows + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_target_bits, dim3((unsigned)nblk), dim3(kThreads), 0, static_cast<hipStream_t>(stream), targets,
                       (long long)n_rows, cols_stored, cols_used, bits);
    return mas_launch_status();
}

extern "C" int mas_partial_loss_fwd(const float* z, const void* spx, int spx_dtype, const uint8_t* mask, const uint32_t* bits,
                                    int N, int C, int H, int W, int S, float invT, int flags, uint64_t* gmax, uint64_t* acc,
                                    void* stream) {
    if (!z || !spx || !mask || !bits || !acc) return MAS_ERR_NULL;
    if ((flags & MAS_LOSS_GROUP) && !gmax) return MAS_ERR_NULL;
    LossArgs a{z, spx, mask, bits, N, C, H, W, S, invT, flags, reinterpret_cast<mas_u64*>(gmax),
               reinterpret_cast<mas_u64*>(acc), nullptr, nullptr};
    return dispatch_loss(a, spx_dtype, false, static_cast<hipStream_t>(stream));
}

extern "C" int mas_group_finalize(const uint64_t* gmax, int64_t n_entries, uint64_t* acc, void* stream) {
    if (!gmax || !acc) return MAS_ERR_NULL;
    if (n_entries <= 0) return MAS_ERR_SHAPE;
    long long nblk = (n_entries + kThreads - 1) / kThreads;
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(k_group_finalize, dim3((unsigned)nblk), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const mas_u64*>(gmax), (long long)n_entries, reinterpret_cast<mas_u64*>(acc), 0,
                       static_cast<const float*>(nullptr), static_cast<float*>(nullptr));
    return mas_launch_status();
}

extern "C" int mas_loss_values(const uint64_t* acc, int flags, float* losses, void* stream) {
    if (!acc || !losses) return MAS_ERR_NULL;
    hipLaunchKernelGGL(k_loss_values, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const mas_u64*>(acc), flags, losses);
    return mas_launch_status();
}

extern "C" int mas_loss_values_weighted(const uint64_t* acc, int flags, const float* weights, float* losses4, void* stream) {
    if (!acc || !weights || !losses4) return MAS_ERR_NULL;
    hipLaunchKernelGGL(k_loss_values_weighted, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const mas_u64*>(acc),
                       flags, weights, losses4);
    return mas_launch_status();
}

extern "C" int mas_loss_scales_weighted(const uint64_t* acc, const float* grad_total, const float* weights, int flags, float* scale,
                                        void* stream) {
    if (!acc || !grad_total || !weights || !scale) return MAS_ERR_NULL;
    hipLaunchKernelGGL(k_loss_scales_weighted, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const mas_u64*>(acc),
                       grad_total, weights, flags, scale);
    return mas_launch_status();
}

extern "C" int mas_loss_scales(const uint64_t* acc, const float* grad_out, int flags, float* scale, void* stream) {
    if (!acc || !grad_out || !scale) return MAS_ERR_NULL;
    hipLaunchKernelGGL(k_loss_scales, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const mas_u64*>(acc), grad_out, flags, scale);
    return mas_launch_status();
}

extern "C" int mas_partial_loss_bwd(const float* z, const void* spx, int spx_dtype, const uint8_t* mask, const uint32_t* bits,
                                    const uint64_t* gmax, const float* scale, int N, int C, int H, int W, int S, float invT,
                                    int flags, float* dz, void* stream) {
    if (!z || !spx || !mask || !bits || !scale || !dz) return MAS_ERR_NULL;
    if ((flags & MAS_LOSS_GROUP) && !gmax) return MAS_ERR_NULL;
    LossArgs a{z, spx, mask, bits, N, C, H, W, S, invT, flags,
               const_cast<mas_u64*>(reinterpret_cast<const mas_u64*>(gmax)), nullptr, scale, dz};
    return dispatch_loss(a, spx_dtype, true, static_cast<hipStream_t>(stream));
}

// ---- quarter-resolution forms: the x4 bilinear upsampling of the logits (models/segmentation/utils.py:25) is evaluated per
// selected pixel inside the scans; forward results are bit-identical to mas_partial_loss_fwd on the materialised tensor.
extern "C" int mas_partial_loss_fwd_lowres(const float* zq, int h, int w, const void* spx, int spx_dtype, const uint8_t* mask,
                                           const uint32_t* bits, int N, int C, int H, int W, int S, float invT, int flags,
                                           uint64_t* gmax, uint64_t* acc, void* stream) {
    if (!zq || !spx || !mask || !bits || !acc) return MAS_ERR_NULL;
    if ((flags & MAS_LOSS_GROUP) && !gmax) return MAS_ERR_NULL;
    if (h <= 0 || w <= 0 || h > H || w > W) return MAS_ERR_SHAPE;
    LossArgs a{zq, spx, mask, bits, N, C, H, W, S, invT, flags, reinterpret_cast<mas_u64*>(gmax),
               reinterpret_cast<mas_u64*>(acc), nullptr, nullptr, h, w};
    return dispatch_loss(a, spx_dtype, false, static_cast<hipStream_t>(stream));
}

extern "C" int mas_partial_loss_bwd_lowres(const float* zq, int h, int w, const void* spx, int spx_dtype, const uint8_t* mask,
                                           const uint32_t* bits, const uint64_t* gmax, const float* scale, int N, int C, int H, int W,
                                           int S, float invT, int flags, int64_t* dzq_fix, void* stream) {
    if (!zq || !spx || !mask || !bits || !scale || !dzq_fix) return MAS_ERR_NULL;
    if ((flags & MAS_LOSS_GROUP) && !gmax) return MAS_ERR_NULL;
    if (h <= 0 || w <= 0 || h > H || w > W) return MAS_ERR_SHAPE;
    LossArgs a{zq, spx, mask, bits, N, C, H, W, S, invT, flags, const_cast<mas_u64*>(reinterpret_cast<const mas_u64*>(gmax)), nullptr,
               scale, reinterpret_cast<float*>(dzq_fix), h, w};
    return dispatch_loss(a, spx_dtype, true, static_cast<hipStream_t>(stream));
}

namespace {
__global__ __launch_bounds__(kThreads) void k_fix_to_float(const long long* __restrict__ fix, long long n, int frac, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    union { double d; mas_u64 u; } s;
    s.u = (mas_u64)(1023 - frac) << 52;
    out[i] = (float)((double)fix[i] * s.d);
}
}  // namespace

extern "C" int mas_fix_to_float(const int64_t* fix, int64_t n, int frac_bits, float* out, void* stream) {
    if (!fix || !out) return MAS_ERR_NULL;
    if (n <= 0 || frac_bits < 0 || frac_bits > 62) return MAS_ERR_SHAPE;
    hipLaunchKernelGGL(k_fix_to_float, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long*>(fix), (long long)n, frac_bits, out);
    return mas_launch_status();
}

// ---- fused forms: one call = every launch of a direction ---------------------------------------------------------------------------
// forward : k_loss_prep (zero acc + table, target rows -> bit masks)  ->  forward scan  ->  k_group_finalize whose last workgroup
//           writes the loss values (3 launches; were: memset, k_target_bits, scan, k_group_finalize, k_loss_values[_weighted])
// backward: (LOWRES: memset of the fixed-point accumulator)  ->  backward scan that forms its scale factors itself  ->  (LOWRES:
//           k_fix_to_float)   (1 / 3 launches; were: k_loss_scales[_weighted], scan / memset, k_loss_scales, scan, k_fix_to_float)
// Same kernels, same arithmetic, same bits as the step-by-step entry points above (tests/test_losses_gpu.py compares both with
// oracle/exact.c).  Reference: trainer/active_joint_multi_predignore_lossdecomp.py:21-72, ..._mclossablation2.py:22-79.
namespace {
struct WorkLayout { size_t gmax_off, bits_off, bytes; };
inline WorkLayout work_layout(int N, int S, int C, int flags) {
    WorkLayout L;
    L.gmax_off = 8 * sizeof(mas_u64);
    const size_t g = (flags & MAS_LOSS_GROUP) ? (size_t)N * S * C * sizeof(mas_u64) : 0;
    L.bits_off = L.gmax_off + g;
    L.bytes = L.bits_off + (size_t)N * S * sizeof(unsigned);
    return L;
}
}  // namespace

extern "C" size_t mas_partial_loss_work_bytes(int N, int S, int C, int flags) {
    if (N <= 0 || S <= 0 || C <= 0) return 0;
    return work_layout(N, S, C, flags).bytes;
}

extern "C" int mas_partial_loss_fwd_fused(const float* z, int h, int w, const void* spx, int spx_dtype, const uint8_t* mask,
                                          const uint8_t* targets, int cols_stored, int cols_used, const uint32_t* bits, int N, int C, int H,
                                          int W, int S, float invT, int flags, const float* weights, void* work, size_t work_bytes,
                                          float* losses, void* stream) {
    if (!z || !spx || !mask || !work || (!targets && !bits)) return MAS_ERR_NULL;
    if (N <= 0 || S <= 0 || H <= 0 || W <= 0) return MAS_ERR_SHAPE;
    if (C < 2 || C > MAS_MAX_CLASSES) return MAS_ERR_CLASSES;
    if ((h > 0) != (w > 0) || h > H || w > W) return MAS_ERR_SHAPE;
    if (targets && (cols_stored <= 0 || cols_used < 1 || cols_used > cols_stored || cols_used > MAS_MAX_CLASSES)) return MAS_ERR_CLASSES;
    const WorkLayout L = work_layout(N, S, C, flags);
    if (work_bytes < L.bytes) return MAS_ERR_WORKSPACE;
    if ((uintptr_t)work % 8 != 0) return MAS_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned char* wb = static_cast<unsigned char*>(work);
    mas_u64* acc = reinterpret_cast<mas_u64*>(wb);
    mas_u64* gmax = (flags & MAS_LOSS_GROUP) ? reinterpret_cast<mas_u64*>(wb + L.gmax_off) : nullptr;
    unsigned* wbits = reinterpret_cast<unsigned*>(wb + L.bits_off);
    const long long n_zero = (long long)(L.bits_off / sizeof(mas_u64)), n_rows = (long long)N * S;
    const long long n_prep = n_zero > n_rows ? n_zero : n_rows;
    hipLaunchKernelGGL(k_loss_prep, dim3((unsigned)((n_prep + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, acc, n_zero, targets, n_rows,
                       cols_stored, cols_used, wbits);
    if (int e = mas_launch_status()) return e;
    LossArgs a{z, spx, mask, targets ? wbits : bits, N, C, H, W, S, invT, flags, gmax, acc, nullptr, nullptr, h > 0 ? h : 0, h > 0 ? w : 0};
    if (int e = dispatch_loss(a, spx_dtype, false, st)) return e;
    if (gmax) {
        const long long n_entries = (long long)N * S * C;
        // (few workgroups: every one of them ends in atomics on the same accumulator words -- with 640 of them those serialised at the
        //  L2 for 40 us, measured; 64 x 256 threads x 10 entries each is 4 us)
        long long nblk = (n_entries + 8 * kThreads - 1) / (8 * kThreads);
        if (nblk > 64) nblk = 64;
        hipLaunchKernelGGL(k_group_finalize, dim3((unsigned)nblk), dim3(kThreads), 0, st, gmax, n_entries, acc, flags, weights, losses);
    } else if (losses) {
        if (weights) hipLaunchKernelGGL(k_loss_values_weighted, dim3(1), dim3(64), 0, st, acc, flags, weights, losses);
        else hipLaunchKernelGGL(k_loss_values, dim3(1), dim3(64), 0, st, acc, flags, losses);
    }
    return mas_launch_status();
}

extern "C" int mas_partial_loss_bwd_fused(const float* z, int h, int w, const void* spx, int spx_dtype, const uint8_t* mask, const uint32_t* bits,
                                          const void* work, const float* grad, const float* weights, int N, int C, int H, int W, int S,
                                          float invT, int flags, float* dz, int64_t* dzq_fix, void* stream) {
    if (!z || !spx || !mask || !work || !grad || !dz) return MAS_ERR_NULL;
    if (N <= 0 || S <= 0 || H <= 0 || W <= 0) return MAS_ERR_SHAPE;
    if (C < 2 || C > MAS_MAX_CLASSES) return MAS_ERR_CLASSES;
    const bool low = h > 0;
    if ((h > 0) != (w > 0) || h > H || w > W) return MAS_ERR_SHAPE;
    if (low && !dzq_fix) return MAS_ERR_NULL;
    const WorkLayout L = work_layout(N, S, C, flags);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned char* wb = static_cast<const unsigned char*>(work);
    mas_u64* acc = const_cast<mas_u64*>(reinterpret_cast<const mas_u64*>(wb));
    mas_u64* gmax = (flags & MAS_LOSS_GROUP) ? const_cast<mas_u64*>(reinterpret_cast<const mas_u64*>(wb + L.gmax_off)) : nullptr;
    const unsigned* wbits = bits ? bits : reinterpret_cast<const unsigned*>(wb + L.bits_off);
    const size_t nq = low ? (size_t)N * C * h * w : 0;
    if (low) {
        const hipError_t e = hipMemsetAsync(dzq_fix, 0, nq * sizeof(int64_t), st);
        if (e != hipSuccess) return (int)e;
    }
    LossArgs a{z, spx, mask, wbits, N, C, H, W, S, invT, flags, gmax, acc, nullptr, low ? reinterpret_cast<float*>(dzq_fix) : dz,
               low ? h : 0, low ? w : 0, grad, weights};
    if (int e = dispatch_loss(a, spx_dtype, true, st)) return e;
    if (low) {
        hipLaunchKernelGGL(k_fix_to_float, dim3((unsigned)((nq + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                           reinterpret_cast<const long long*>(dzq_fix), (long long)nq, MAS_GRAD_FRAC, dz);
        return mas_launch_status();
    }
    return 0;
}
