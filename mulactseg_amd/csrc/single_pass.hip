// single_pass.hip -- the MI355X-first form of the acquisition scan: ONE read of the logits per pool image.
//
// The reference needs two passes over the pool (two model forwards per image) because the class weight
// cls_weight[top1] multiplies every pixel's BvSB margin and depends on the class prior of the WHOLE pool
// (active_selection/my_bvsb_predclsbal_pwr_banignore.py:35-47 then :49-72).  But the weight only depends on the
// pixel's arg-max class, so it factors out of the region sum:
//
//     sum_{p in s} bvsb_p * w[top1_p]  =  sum_c  w_c * ( sum_{p in s, top1_p = c} bvsb_p )
//
// k_single_pass accumulates, in one scan, (a) the per-image class-probability sums (K2), (b) per (region, class)
// the fixed-point sum of the UNWEIGHTED margins and (c) the arg-max-class histogram (K3).  After the pool has been
// seen once, k_region_finalize_weighted applies the weights in exact integer arithmetic
// (sum_c class_sum[s,c] * floor(w_c * 2^31), 128-bit accumulate), divides by the pixel count and applies the ban.
// With w = 1 the result is bit-identical to the two-pass kernels; with weights it differs from them only by the
// per-pixel rounding of bvsb*w (~1e-9 relative).  HBM traffic per image: logits + ids once (176 MB instead of 344 MB)
// and, end to end, ONE model forward per pool image instead of two.
//
// Two kernels share the arithmetic: k_single_pass_ring (16-B aligned rows, H >= 64, C in {19, 20, 21}: the Cityscapes /
// bench shapes) keeps the next row's loads in flight while it computes the current row; k_single_pass (any shape) waits
// for a row, then computes it.  MAS_SINGLE_PASS_RING=0 in the environment forces the second one (A/B measurements).
#include <cstdlib>
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTileW = 256;     // 64 lanes x 4 px (16-B loads; two packed pixel pairs per lane)
constexpr int kTileH = 16;      // 4 waves x 4 row iterations
constexpr int kLogSlots = 7;
constexpr int kSlots = 1 << kLogSlots;

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

// LOWRES (K8): `z` is the model's QUARTER-resolution logit tensor [B,C,h,w]; the x4 bilinear upsampling of
// models/segmentation/utils.py:25 is evaluated in registers from an LDS copy of the tile's low-resolution footprint
// (edge-replicated, so the clamped second tap of ATen's index arithmetic is a plain "+1"), in the operation order of
// csrc/upsample.hip -- every pixel's 20 logits, hence every output of the scan, equal those of the materialised tensor bit for
// bit.  The [B,C,H,W] logits (671 MB per pool batch) and the upsampling pass are gone; the scan then has no HBM stream left
// to wait for: it is bound by its VALU work and the latency of its LDS reads (profiles/r02/k_scan_forms_pmc.md).
struct LowSrc { int h, w; float sh, sw; };
constexpr int kLowRows = 7, kLowCols = 68;       // footprint of a 16 x 256 tile for ratios >= 3.8 (checked by the launcher)

__device__ __forceinline__ void low_tap(float scale, int o, int& i0, float& l0, float& l1) {
    float s = scale * ((float)o + 0.5f) - 0.5f;
    s = s < 0.0f ? 0.0f : s;
    i0 = (int)s;
    l1 = s - (float)i0;
    l0 = 1.0f - l1;
}

// X4 (LOWRES, VEC, W == 4 w): a lane's four pixels 4L .. 4L + 3 are one period of the x4 pattern -- pixels 0, 1 share their
// left tap (column L - 1, or 0 at the picture's edge), pixels 2, 3 share theirs (column L): 8 LDS reads per class and row pair
// instead of 16, same operands, same arithmetic (weights still from low_tap: {0.625, 0.875, 0.125, 0.375}, or (1, 0) where the
// source coordinate is clamped at 0).
template <int CT, bool EXACT, typename IdT, bool VEC, bool LOWRES = false, bool X4 = false>
__global__ __launch_bounds__(kThreads) void k_single_pass(const float* __restrict__ z, const IdT* __restrict__ spx, int C, int H,
                                                           int W, int S, float invT, int tiles_x, int tiles_y,
                                                           mas_u64* __restrict__ prob_sum, mas_u64* __restrict__ class_sum,
                                                           unsigned* __restrict__ hist, const LowSrc lr = LowSrc{0, 0, 0.f, 0.f}) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mas_u64* t_sum = reinterpret_cast<mas_u64*>(smem);                                        // [kSlots * C]
    mas_u64* s_part = reinterpret_cast<mas_u64*>(smem + sizeof(mas_u64) * kSlots * C);         // [4][CT]
    unsigned* t_hist = reinterpret_cast<unsigned*>(smem + sizeof(mas_u64) * (kSlots * C + 4 * CT));   // [kSlots * C]
    int* t_keys = reinterpret_cast<int*>(smem + sizeof(mas_u64) * (kSlots * C + 4 * CT) + sizeof(unsigned) * kSlots * C);
    float* s_low = reinterpret_cast<float*>(t_keys + kSlots);                                     // LOWRES: [C][kLowRows][kLowCols]

    for (int i = threadIdx.x; i < kSlots; i += kThreads) t_keys[i] = -1;
    for (int i = threadIdx.x; i < kSlots * C; i += kThreads) { t_sum[i] = 0; t_hist[i] = 0; }

    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int HW = H * W;
    const float* zb = z + (size_t)b * C * (LOWRES ? lr.h * lr.w : HW);
    int low_r0 = 0, low_c0 = 0;
    if (LOWRES) {
        float u0, u1;
        low_tap(lr.sh, ty * kTileH, low_r0, u0, u1);
        low_tap(lr.sw, tx * kTileW, low_c0, u0, u1);
        // one (class, row) line of the footprint per wave and trip: lanes run along the columns, edges replicated
        const int lane_ = threadIdx.x & (MAS_WAVE - 1), wave_ = threadIdx.x / MAS_WAVE;
        for (int cr = wave_; cr < C * kLowRows; cr += kThreads / MAS_WAVE) {
            const int c = cr / kLowRows, r = cr - c * kLowRows;
            const int gy = min(low_r0 + r, lr.h - 1);
            const float* src = zb + ((size_t)c * lr.h + gy) * lr.w;
            for (int x = lane_; x < kLowCols; x += MAS_WAVE) s_low[cr * kLowCols + x] = src[min(low_c0 + x, lr.w - 1)];
        }
    }
    __syncthreads();
    const IdT* sb = spx + (size_t)b * HW;
    mas_u64* gsum = class_sum + (size_t)b * S * C;
    unsigned* ghist = hist + (size_t)b * S * C;
    const int lane = threadIdx.x & (MAS_WAVE - 1);
    const int wave = threadIdx.x / MAS_WAVE;

    unsigned acc[CT];       // mas_probq quanta (raw bit patterns, bias removed at the end): 16 px per thread per tile
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0;
    unsigned n_quanta = 0;
    // LOWRES: the column taps of this lane's four pixels are the same for every row of the tile
    int lx[4];
    mas_v2f l0xa = mas_splat(0.f), l1xa = mas_splat(0.f), l0xb = mas_splat(0.f), l1xb = mas_splat(0.f);
    if (LOWRES) {
        float a0[4], a1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xk = tx * kTileW + (VEC ? (lane * 4 + k) : (k * MAS_WAVE + lane));
            int i0;
            low_tap(lr.sw, xk < W ? xk : W - 1, i0, a0[k], a1[k]);
            lx[k] = i0 - low_c0;
        }
        l0xa = (mas_v2f){a0[0], a0[1]}; l1xa = (mas_v2f){a1[0], a1[1]};
        l0xb = (mas_v2f){a0[2], a0[3]}; l1xb = (mas_v2f){a1[2], a1[3]};
    }

    for (int it = 0; it < kTileH / 4; ++it) {
        const int y = ty * kTileH + it * 4 + wave;
        if (y >= H) break;
        const size_t row = (size_t)y * W;
        int xs[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xs[k] = tx * kTileW + (VEC ? (lane * 4 + k) : (k * MAS_WAVE + lane));
            ok[k] = xs[k] < W;
        }
        // ids first: their HBM latency must overlap the logit loads, not sit exposed behind the softmax
        int id[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            id[k] = ok[k] ? mas_load_id(sb, row + xs[k]) : -1;
            if (id[k] >= S) id[k] = -1;
        }
        // pixel k lives in pair k>>1, half k&1
        mas_v2f v[2][CT];
        float b1[4], b2[4];
        int a1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { b1[k] = -__builtin_inff(); b2[k] = -__builtin_inff(); a1[k] = 0; }
        int ly = 0;
        mas_v2f l0y = mas_splat(0.f), l1y = mas_splat(0.f);
        if (LOWRES) {
            float u0, u1;
            low_tap(lr.sh, y, ly, u0, u1);
            ly -= low_r0;
            l0y = mas_splat(u0);
            l1y = mas_splat(u1);
        }
        if (LOWRES) {
            // y = l0y * (l0x * v00 + l1x * v01) + l1y * (l0x * v10 + l1x * v11), every product and sum rounded once.  The 16 LDS
            // reads of class c + 1 are issued before class c is combined (two register sets), so the combine never waits for its
            // own reads.  Pixels beyond the picture (clamped column taps, finite values) are masked where it matters: their
            // probability quanta through Ra / Rb = 0, their region key through id = -1.
            if constexpr (X4) {
                // four aligned register pairs per class -- (row r0 / r1) x (pixels 0, 1 / pixels 2, 3), each {v[lx], v[lx + 1]} from one
                // ds_read2_b32 -- and the products in the operand-select forms of mas_pk_mul_lo / mas_pk_mul_hi (common.h: the form
                // the compiler picks for "{d, d} * w" is not safe next to another kernel's matrix-core waves)
                mas_v2f raw[2][4];
                auto fetch4 = [&](int c, mas_v2f (&d)[4]) {
                    const float* r0 = s_low + (c * kLowRows + ly) * kLowCols;
                    const float* r1 = r0 + kLowCols;
                    d[0] = (mas_v2f){r0[lx[0]], r0[lx[0] + 1]}; d[1] = (mas_v2f){r1[lx[0]], r1[lx[0] + 1]};
                    d[2] = (mas_v2f){r0[lx[2]], r0[lx[2] + 1]}; d[3] = (mas_v2f){r1[lx[2]], r1[lx[2] + 1]};
                };
                const mas_v2f ly01 = {l0y.x, l1y.x};
                fetch4(0, raw[0]);
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    if (EXACT || c < C) {
                        if (c + 1 < CT && (EXACT || c + 1 < C)) fetch4(c + 1, raw[(c + 1) & 1]);
                        const mas_v2f (&d)[4] = raw[c & 1];
                        v[0][c] = mas_pk_mul_lo(ly01, mas_pk_mul_lo(d[0], l0xa) + mas_pk_mul_hi(d[0], l1xa))
                                + mas_pk_mul_hi(ly01, mas_pk_mul_lo(d[1], l0xa) + mas_pk_mul_hi(d[1], l1xa));
                        v[1][c] = mas_pk_mul_lo(ly01, mas_pk_mul_lo(d[2], l0xb) + mas_pk_mul_hi(d[2], l1xb))
                                + mas_pk_mul_hi(ly01, mas_pk_mul_lo(d[3], l0xb) + mas_pk_mul_hi(d[3], l1xb));
                    }
                }
            } else {
                float raw[2][16];
                auto fetch16 = [&](int c, float (&d)[16]) {
                    const float* r0 = s_low + (c * kLowRows + ly) * kLowCols;
                    const float* r1 = r0 + kLowCols;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        d[4 * k] = r0[lx[k]]; d[4 * k + 1] = r0[lx[k] + 1]; d[4 * k + 2] = r1[lx[k]]; d[4 * k + 3] = r1[lx[k] + 1];
                    }
                };
                fetch16(0, raw[0]);
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    if (EXACT || c < C) {
                        if (c + 1 < CT && (EXACT || c + 1 < C)) fetch16(c + 1, raw[(c + 1) & 1]);
                        const float (&d)[16] = raw[c & 1];
                        const mas_v2f v00a = {d[0], d[4]}, v01a = {d[1], d[5]}, v10a = {d[2], d[6]}, v11a = {d[3], d[7]};
                        const mas_v2f v00b = {d[8], d[12]}, v01b = {d[9], d[13]}, v10b = {d[10], d[14]}, v11b = {d[11], d[15]};
                        v[0][c] = l0y * (l0xa * v00a + l1xa * v01a) + l1y * (l0xa * v10a + l1xa * v11a);
                        v[1][c] = l0y * (l0xb * v00b + l1xb * v01b) + l1y * (l0xb * v10b + l1xb * v11b);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (EXACT || c < C) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                const float* zc = zb + (size_t)c * HW;       // wave-uniform base, 32-bit lane offset
                if (LOWRES) {
                    t = make_float4(v[0][c].x, v[0][c].y, v[1][c].x, v[1][c].y);          // interpolated before this loop
                } else if (VEC) {
                    if (ok[0]) t = mas_load_stream4(zc + (unsigned)(row + xs[0]));
                } else {
                    if (ok[0]) t.x = zc[row + xs[0]];
                    if (ok[1]) t.y = zc[row + xs[1]];
                    if (ok[2]) t.z = zc[row + xs[2]];
                    if (ok[3]) t.w = zc[row + xs[3]];
                }
                v[0][c] = (mas_v2f){t.x, t.y};
                v[1][c] = (mas_v2f){t.z, t.w};
                const float q[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // same update as "if (v > b1) {b2 = b1; b1 = v; a1 = c} else if (v > b2) b2 = v", as two medians (see ring_consume)
                    const bool g1 = q[k] > b1[k];
                    b2[k] = __builtin_amdgcn_fmed3f(b1[k], b2[k], q[k]);
                    a1[k] = g1 ? c : a1[k];
                    b1[k] = __builtin_amdgcn_fmed3f(b1[k], q[k], 3.402823466e+38f);
                }
            } else {
                v[0][c] = mas_splat(0.f);
                v[1][c] = mas_splat(0.f);
            }
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the phases apart: interleaving them doubles the live registers
        // class-probability quanta (K2 arithmetic); the maximum is already known from the top-2 scan
        {
            mas_v2f Ra, Rb;
            mas_softmax_quad<CT, EXACT, true>(v[0], v[1], C, invT, Ra, Rb, (mas_v2f){b1[0], b1[1]}, (mas_v2f){b1[2], b1[3]});
            Ra = Ra * mas_splat(8388608.0f);
            Rb = Rb * mas_splat(8388608.0f);
            Ra = (mas_v2f){ok[0] ? Ra.x : 0.0f, ok[1] ? Ra.y : 0.0f};
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
        __builtin_amdgcn_sched_barrier(0);
        n_quanta += 4;
        // region accumulation of the unweighted margin, keyed by (superpixel, arg-max class)
        mas_u64 q[4];
        int key[4];
        float margin[4];
        mas_bvsb_quad(b1, b2, invT, margin);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            q[k] = mas_fix_unit(margin[k]);
            key[k] = id[k] < 0 ? -1 : id[k] * MAS_MAX_CLASSES + a1[k];
        }
        const bool same = (key[0] == key[1]) && (key[1] == key[2]) && (key[2] == key[3]);
        if (same) {
            if (key[0] >= 0) {
                const int s = table_slot(t_keys, id[0]);
                const mas_u64 qs = (q[0] + q[1]) + (q[2] + q[3]);
                if (s >= 0) {
                    lds_add(&t_sum[s * C + a1[0]], qs);
                    lds_add(&t_hist[s * C + a1[0]], 4u);
                } else {
                    atomicAdd(&gsum[(size_t)id[0] * C + a1[0]], qs);
                    atomicAdd(&ghist[(size_t)id[0] * C + a1[0]], 4u);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (key[k] < 0) continue;
                const int s = table_slot(t_keys, id[k]);
                if (s >= 0) {
                    lds_add(&t_sum[s * C + a1[k]], q[k]);
                    lds_add(&t_hist[s * C + a1[k]], 1u);
                } else {
                    atomicAdd(&gsum[(size_t)id[k] * C + a1[k]], q[k]);
                    atomicAdd(&ghist[(size_t)id[k] * C + a1[k]], 1u);
                }
            }
        }
    }

    // class sums: wave shuffle reduction, 4 waves through LDS, one atomic per class per workgroup
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        mas_u64 a = acc[c] - n_quanta * MAS_PROBQ_BIAS;
#pragma unroll
        for (int off = MAS_WAVE / 2; off > 0; off >>= 1) a += __shfl_down(a, off, MAS_WAVE);
        if (lane == 0) s_part[wave * CT + c] = a;
    }
    __syncthreads();
    if (threadIdx.x < CT && (EXACT || (int)threadIdx.x < C)) {
        mas_u64 a = 0;
#pragma unroll
        for (int w = 0; w < kThreads / MAS_WAVE; ++w) a += s_part[w * CT + threadIdx.x];
        if (a) atomicAdd(&prob_sum[(size_t)b * C + threadIdx.x], a);
    }
    for (int i = threadIdx.x; i < kSlots * C; i += kThreads) {
        const unsigned n = t_hist[i];
        if (n) {
            const int s = i / C;
            const size_t g = (size_t)t_keys[s] * C + (i - s * C);
            atomicAdd(&ghist[g], n);
            atomicAdd(&gsum[g], t_sum[i]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Ring variant (wide, tall images): the same arithmetic, but every wave keeps the NEXT row's loads in flight while it
// computes the current one.  Bandwidth on this part is (bytes in flight) / (round-trip time) and only the VGPR file is
// big enough to hold ~300 KB/CU of outstanding loads, so the kernel runs at 2 waves/SIMD with two
// 84-register row buffers per wave; the class-probability accumulators move to LDS (fire-and-forget ds_add, one
// private word per thread and class) to pay for the second buffer.  A workgroup streams a 256 x 64 px column strip:
// the four waves take interleaved rows, 16 rows each, so the pipeline is filled once per 16 rows.
// ------------------------------------------------------------------------------------------------
constexpr int kRingTileH = 64;
constexpr int kRingRows = kRingTileH / 4;      // rows per wave

// raw superpixel ids of four consecutive pixels, loaded with 16-B (8-B for u16) vector loads; every loaded register
// is consumed by get() -- a dead half would let the allocator recycle a register that still has a load in flight,
// which costs a vmcnt(0) in the middle of the pipeline
template <typename IdT> struct IdQuad;
template <> struct IdQuad<long long> {
    int4 a, b;
    __device__ __forceinline__ void load(const long long* p) {
        a = *reinterpret_cast<const int4*>(p);
        b = *reinterpret_cast<const int4*>(p + 2);
    }
    __device__ __forceinline__ void get(int (&id)[4]) const {     // ids outside int32 are invalid (-1)
        id[0] = a.y == 0 ? a.x : -1;
        id[1] = a.w == 0 ? a.z : -1;
        id[2] = b.y == 0 ? b.x : -1;
        id[3] = b.w == 0 ? b.z : -1;
    }
};
template <> struct IdQuad<int> {
    int4 a;
    __device__ __forceinline__ void load(const int* p) { a = *reinterpret_cast<const int4*>(p); }
    __device__ __forceinline__ void get(int (&id)[4]) const { id[0] = a.x; id[1] = a.y; id[2] = a.z; id[3] = a.w; }
};
template <> struct IdQuad<unsigned short> {
    uint2 a;
    __device__ __forceinline__ void load(const unsigned short* p) { a = *reinterpret_cast<const uint2*>(p); }
    __device__ __forceinline__ void get(int (&id)[4]) const {
        id[0] = (int)(a.x & 0xffffu); id[1] = (int)(a.x >> 16); id[2] = (int)(a.y & 0xffffu); id[3] = (int)(a.y >> 16);
    }
};

template <int CT, typename IdT>
struct RowRegs {
    float4 t[CT];
    IdQuad<IdT> ids;
};

template <int CT, bool EXACT, typename IdT>
__device__ __forceinline__ void ring_issue(RowRegs<CT, IdT>& r, const float* __restrict__ zb, const IdT* __restrict__ sb, int C,
                                           int HW, unsigned off) {
    r.ids.load(sb + off);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (EXACT || c < C) r.t[c] = mas_load_stream4(zb + (size_t)c * HW + off);
        else r.t[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int CT, bool EXACT, typename IdT>
__device__ __forceinline__ void ring_consume(RowRegs<CT, IdT>& r, bool ok, int C, int S, float invT, unsigned* s_acc, int* t_keys,
                                             mas_u64* t_sum, unsigned* t_hist, mas_u64* __restrict__ gsum,
                                             unsigned* __restrict__ ghist, int& last_id, int& last_slot) {
    mas_v2f v[2][CT];
    float b1[4], b2[4];
    int a1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { b1[k] = -__builtin_inff(); b2[k] = -__builtin_inff(); a1[k] = 0; }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (EXACT || c < C) {
            const float4 t = r.t[c];
            v[0][c] = (mas_v2f){t.x, t.y};
            v[1][c] = (mas_v2f){t.z, t.w};
            const float q[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // with b1 >= b2:  b2' = median(b1, b2, q),  b1' = max(b1, q) = median(b1, q, FLT_MAX): two v_med3_f32
                // (the same values as the compare/select form for finite logits; ties keep the lowest class index).
                // FLT_MAX rather than +inf: the compiler folds median(.., +inf) into v_max_f32 and then has to
                // canonicalise every loaded logit first (one more instruction per element)
                const bool g1 = q[k] > b1[k];
                b2[k] = __builtin_amdgcn_fmed3f(b1[k], b2[k], q[k]);
                a1[k] = g1 ? c : a1[k];
                b1[k] = __builtin_amdgcn_fmed3f(b1[k], q[k], 3.402823466e+38f);
            }
        } else {
            v[0][c] = mas_splat(0.f);
            v[1][c] = mas_splat(0.f);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        mas_v2f Ra, Rb;
        mas_softmax_quad<CT, EXACT, true>(v[0], v[1], C, invT, Ra, Rb, (mas_v2f){b1[0], b1[1]}, (mas_v2f){b1[2], b1[3]});
        Ra = Ra * mas_splat(8388608.0f);
        Rb = Rb * mas_splat(8388608.0f);
        Ra = ok ? Ra : mas_splat(0.0f);
        Rb = ok ? Rb : mas_splat(0.0f);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (EXACT || c < C) {
                const mas_v2f ta = mas_pk_fma(v[0][c], Ra, mas_splat(8388608.0f));
                const mas_v2f tb = mas_pk_fma(v[1][c], Rb, mas_splat(8388608.0f));
                lds_add(&s_acc[c * kThreads + threadIdx.x], (mas_f2u(ta.x) + mas_f2u(ta.y)) + (mas_f2u(tb.x) + mas_f2u(tb.y)));
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    mas_u64 q[4];
    int key[4], id[4];
    float margin[4];
    r.ids.get(id);
    mas_bvsb_quad(b1, b2, invT, margin);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        id[k] = (ok && id[k] >= 0 && id[k] < S) ? id[k] : -1;
        q[k] = mas_fix_unit(margin[k]);
        key[k] = id[k] < 0 ? -1 : id[k] * MAS_MAX_CLASSES + a1[k];
    }
    const bool same = (key[0] == key[1]) && (key[1] == key[2]) && (key[2] == key[3]);
    if (same) {
        if (key[0] >= 0) {
            // a lane walks down a column: most of the time it is still inside the superpixel of its previous row
            if (id[0] != last_id) {
                last_slot = table_slot(t_keys, id[0]);
                last_id = id[0];
            }
            const int s = last_slot;
            const mas_u64 qs = (q[0] + q[1]) + (q[2] + q[3]);
            if (s >= 0) {
                lds_add(&t_sum[s * C + a1[0]], qs);
                lds_add(&t_hist[s * C + a1[0]], 4u);
            } else {
                atomicAdd(&gsum[(size_t)id[0] * C + a1[0]], qs);
                atomicAdd(&ghist[(size_t)id[0] * C + a1[0]], 4u);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (key[k] < 0) continue;
            const int s = table_slot(t_keys, id[k]);
            if (s >= 0) {
                lds_add(&t_sum[s * C + a1[k]], q[k]);
                lds_add(&t_hist[s * C + a1[k]], 1u);
            } else {
                atomicAdd(&gsum[(size_t)id[k] * C + a1[k]], q[k]);
                atomicAdd(&ghist[(size_t)id[k] * C + a1[k]], 1u);
            }
        }
    }
}

template <int CT, bool EXACT, typename IdT>
__global__ __launch_bounds__(kThreads, 2) void k_single_pass_ring(const float* __restrict__ z, const IdT* __restrict__ spx, int C,
                                                                   int H, int W, int S, float invT, int tiles_x, int tiles_y,
                                                                   mas_u64* __restrict__ prob_sum,
                                                                   mas_u64* __restrict__ class_sum, unsigned* __restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mas_u64* t_sum = reinterpret_cast<mas_u64*>(smem);                                        // [kSlots * C]
    mas_u64* s_part = reinterpret_cast<mas_u64*>(smem + sizeof(mas_u64) * kSlots * C);         // [4][CT]
    unsigned* t_hist = reinterpret_cast<unsigned*>(smem + sizeof(mas_u64) * (kSlots * C + 4 * CT));   // [kSlots * C]
    int* t_keys = reinterpret_cast<int*>(smem + sizeof(mas_u64) * (kSlots * C + 4 * CT) + sizeof(unsigned) * kSlots * C);
    unsigned* s_acc = reinterpret_cast<unsigned*>(t_keys + kSlots);                            // [CT][kThreads]

    for (int i = threadIdx.x; i < kSlots; i += kThreads) t_keys[i] = -1;
    for (int i = threadIdx.x; i < kSlots * C; i += kThreads) { t_sum[i] = 0; t_hist[i] = 0; }
#pragma unroll
    for (int c = 0; c < CT; ++c) s_acc[c * kThreads + threadIdx.x] = 0;
    __syncthreads();

    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int HW = H * W;
    const float* zb = z + (size_t)b * C * HW;
    const IdT* sb = spx + (size_t)b * HW;
    mas_u64* gsum = class_sum + (size_t)b * S * C;
    unsigned* ghist = hist + (size_t)b * S * C;
    const int lane = threadIdx.x & (MAS_WAVE - 1);
    const int wave = threadIdx.x / MAS_WAVE;

    const int x0 = tx * kTileW + lane * 4;
    const bool okx = x0 < W;                 // W % 4 == 0: all four pixels in or out
    const unsigned xc = okx ? (unsigned)x0 : 0u;
    const int ybase = ty * kRingTileH + wave;
    // out-of-range rows / columns load a valid address (row H-1 / column 0) and are masked in ring_consume
    auto off_of = [&](int it) -> unsigned {
        int y = ybase + it * 4;
        y = y < H ? y : H - 1;
        return (unsigned)y * (unsigned)W + xc;
    };
    auto ok_of = [&](int it) -> bool { return okx && (ybase + it * 4 < H); };

    RowRegs<CT, IdT> A, B;
    int last_id = -2, last_slot = -1;
    ring_issue<CT, EXACT, IdT>(A, zb, sb, C, HW, off_of(0));
#pragma unroll 1
    for (int it = 0; it < kRingRows; it += 2) {
        ring_issue<CT, EXACT, IdT>(B, zb, sb, C, HW, off_of(it + 1));
        __builtin_amdgcn_sched_barrier(0);
        ring_consume<CT, EXACT, IdT>(A, ok_of(it), C, S, invT, s_acc, t_keys, t_sum, t_hist, gsum, ghist, last_id, last_slot);
        __builtin_amdgcn_sched_barrier(0);
        // the last trip re-reads its own second row (an L2 hit, never consumed): the wait counts stay static
        ring_issue<CT, EXACT, IdT>(A, zb, sb, C, HW, off_of(it + 2 < kRingRows ? it + 2 : kRingRows - 1));
        __builtin_amdgcn_sched_barrier(0);
        ring_consume<CT, EXACT, IdT>(B, ok_of(it + 1), C, S, invT, s_acc, t_keys, t_sum, t_hist, gsum, ghist, last_id, last_slot);
        __builtin_amdgcn_sched_barrier(0);
    }

    __syncthreads();
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        mas_u64 a = s_acc[c * kThreads + threadIdx.x] - (unsigned)(kRingRows * 4) * MAS_PROBQ_BIAS;
#pragma unroll
        for (int off = MAS_WAVE / 2; off > 0; off >>= 1) a += __shfl_down(a, off, MAS_WAVE);
        if (lane == 0) s_part[wave * CT + c] = a;
    }
    __syncthreads();
    if (threadIdx.x < CT && (EXACT || (int)threadIdx.x < C)) {
        mas_u64 a = 0;
#pragma unroll
        for (int w = 0; w < kThreads / MAS_WAVE; ++w) a += s_part[w * CT + threadIdx.x];
        if (a) atomicAdd(&prob_sum[(size_t)b * C + threadIdx.x], a);
    }
    for (int i = threadIdx.x; i < kSlots * C; i += kThreads) {
        const unsigned n = t_hist[i];
        if (n) {
            const int s = i / C;
            const size_t g = (size_t)t_keys[s] * C + (i - s * C);
            atomicAdd(&ghist[g], n);
            atomicAdd(&gsum[g], t_sum[i]);
        }
    }
}

// score[r] = floor( (sum_c class_sum[r,c] * W31[c]) >> 31  /  n_r ) * 2^-40 ; dominant class; ban
__global__ __launch_bounds__(kThreads) void k_region_finalize_weighted(const mas_u64* __restrict__ class_sum,
                                                                        const unsigned* __restrict__ hist, long long n_regions,
                                                                        int C, const unsigned* __restrict__ w31, int ban_class,
                                                                        float* __restrict__ score, int* __restrict__ dominant,
                                                                        unsigned* __restrict__ count,
                                                                        long long* __restrict__ hist_i64) {
    // a thread owns a region, but rows of C words are not coalescable per thread: the workgroup's 256 rows are staged
    // through LDS with flat, fully coalesced loads (row stride padded to an odd word count: conflict-free per-thread reads)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int CP = C | 1;
    mas_u64* s_sum = reinterpret_cast<mas_u64*>(smem);                          // [kThreads][CP]
    unsigned* s_hist = reinterpret_cast<unsigned*>(s_sum + (size_t)kThreads * CP);   // [kThreads][CP]
    const long long r0 = (long long)blockIdx.x * kThreads;
    const long long rows = (n_regions - r0) < kThreads ? (n_regions - r0) : kThreads;
    const long long words = rows * C;
    for (long long i = threadIdx.x; i < words; i += kThreads) {
        const int row = (int)(i / C), col = (int)(i - (long long)row * C);
        const unsigned hv = hist[r0 * C + i];
        s_sum[row * CP + col] = class_sum[r0 * C + i];
        s_hist[row * CP + col] = hv;
        if (hist_i64) hist_i64[r0 * C + i] = (long long)hv;
    }
    __syncthreads();
    if ((long long)threadIdx.x >= rows) return;
    const long long r = r0 + threadIdx.x;
    const unsigned* h = s_hist + threadIdx.x * CP;
    const mas_u64* cs = s_sum + threadIdx.x * CP;
    unsigned long long n = 0;
    unsigned best = 0;
    int arg = 0;
    uint64_t hi = 0, lo = 0;
    for (int c = 0; c < C; ++c) {
        const unsigned v = h[c];
        n += v;
        if (v > best) { best = v; arg = c; }
        if (v) mas_mac_u64_u32((uint64_t)cs[c], w31[c], &hi, &lo);
    }
    float sc = 0.0f;
    if (n) sc = mas_fixed_mean(mas_shr31_u128(hi, lo), n, MAS_SCORE_FRAC);
    if (ban_class >= 0 && arg == ban_class) sc = 0.0f;
    score[r] = sc;
    if (dominant) dominant[r] = arg;
    if (count) count[r] = (unsigned)n;
}

inline bool mas_ring_enabled() {
    static const bool on = [] { const char* e = getenv("MAS_SINGLE_PASS_RING"); return !(e && e[0] == '0'); }();
    return on;
}

inline size_t smem_bytes(int C, int CT) {
    return sizeof(mas_u64) * ((size_t)kSlots * C + 4 * CT) + sizeof(unsigned) * (size_t)kSlots * C + sizeof(int) * kSlots;
}

template <int CT, bool EXACT, typename IdT>
int launch(const float* z, const void* spx, int B, int C, int H, int W, int S, float invT, mas_u64* prob_sum, mas_u64* class_sum,
           unsigned* hist, hipStream_t st) {
    const int tiles_x = (W + kTileW - 1) / kTileW;
    const int tiles_y = (H + kTileH - 1) / kTileH;
    const long long nblk = (long long)B * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
    const bool vec = (W % 4 == 0) && (((uintptr_t)z & 15) == 0);
    const size_t smem = smem_bytes(C, CT);
    const IdT* ids = static_cast<const IdT*>(spx);
    // (two row buffers of the generic 32-channel instantiation do not fit 256 VGPRs: it keeps the one-row kernel)
    if constexpr (EXACT) if (vec && H >= kRingTileH && (((uintptr_t)spx & 15) == 0) && mas_ring_enabled()) {
        const int rtiles_y = (H + kRingTileH - 1) / kRingTileH;
        const long long rblk = (long long)B * tiles_x * rtiles_y;
        hipLaunchKernelGGL((k_single_pass_ring<CT, EXACT, IdT>), dim3((unsigned)rblk), dim3(kThreads),
                           smem + sizeof(unsigned) * CT * kThreads, st, z, ids, C, H, W, S, invT, tiles_x, rtiles_y, prob_sum,
                           class_sum, hist);
        return mas_launch_status();
    }
    if (vec)
        hipLaunchKernelGGL((k_single_pass<CT, EXACT, IdT, true>), dim3((unsigned)nblk), dim3(kThreads), smem, st, z, ids, C, H, W, S,
                           invT, tiles_x, tiles_y, prob_sum, class_sum, hist);
    else
        hipLaunchKernelGGL((k_single_pass<CT, EXACT, IdT, false>), dim3((unsigned)nblk), dim3(kThreads), smem, st, z, ids, C, H, W, S,
                           invT, tiles_x, tiles_y, prob_sum, class_sum, hist);
    return mas_launch_status();
}

template <int CT, bool EXACT, typename IdT>
int launch_low(const float* zq, int h, int w, const void* spx, int B, int C, int H, int W, int S, float invT, mas_u64* prob_sum,
               mas_u64* class_sum, unsigned* hist, unsigned flags, hipStream_t st) {
    const int tiles_x = (W + kTileW - 1) / kTileW;
    const int tiles_y = (H + kTileH - 1) / kTileH;
    const long long nblk = (long long)B * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
    const LowSrc lr{h, w, (float)h / (float)H, (float)w / (float)W};
    // the LDS footprint holds kLowRows x kLowCols low-resolution elements per class: enough for a 16 x 256 tile when the
    // ratio is >= ~3.8 in both directions (the model's is 4, or 769 / 193)
    if ((double)kTileH * h / H + 2.0 > kLowRows || (double)kTileW * w / W + 2.0 > kLowCols) return MAS_ERR_RANGE;
    const size_t smem = smem_bytes(C, CT) + sizeof(float) * (size_t)C * kLowRows * kLowCols;
    const bool vec = (W % 4 == 0);
    const IdT* ids = static_cast<const IdT*>(spx);
    // > 64 KB of dynamic LDS needs the attribute on every device's copy of the code object: flags per (kernel, device), the
    // call's status is the launch's status
    static bool attr_vec[64] = {}, attr_scalar[64] = {}, attr_x4[64] = {};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return (int)e;
    const bool cached = dev >= 0 && dev < 64;
    if (vec && W == 4 * w && !(flags & MAS_LOWRES_GENERIC)) {       // the model's own ratio: one period of the x4 pattern per lane
        if (!cached || !attr_x4[dev]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_single_pass<CT, EXACT, IdT, true, true, true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return (int)e;
            if (cached) attr_x4[dev] = true;
        }
        hipLaunchKernelGGL((k_single_pass<CT, EXACT, IdT, true, true, true>), dim3((unsigned)nblk), dim3(kThreads), smem, st, zq, ids, C, H, W,
                           S, invT, tiles_x, tiles_y, prob_sum, class_sum, hist, lr);
    } else if (vec) {
        if (!cached || !attr_vec[dev]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_single_pass<CT, EXACT, IdT, true, true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return (int)e;
            if (cached) attr_vec[dev] = true;
        }
        hipLaunchKernelGGL((k_single_pass<CT, EXACT, IdT, true, true>), dim3((unsigned)nblk), dim3(kThreads), smem, st, zq, ids, C, H, W, S,
                           invT, tiles_x, tiles_y, prob_sum, class_sum, hist, lr);
    } else {
        if (!cached || !attr_scalar[dev]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_single_pass<CT, EXACT, IdT, false, true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return (int)e;
            if (cached) attr_scalar[dev] = true;
        }
        hipLaunchKernelGGL((k_single_pass<CT, EXACT, IdT, false, true>), dim3((unsigned)nblk), dim3(kThreads), smem, st, zq, ids, C, H, W, S,
                           invT, tiles_x, tiles_y, prob_sum, class_sum, hist, lr);
    }
    return mas_launch_status();
}

template <int CT, bool EXACT>
int dispatch_low_ids(const float* zq, int h, int w, const void* spx, int spx_dtype, int B, int C, int H, int W, int S, float invT,
                     mas_u64* prob_sum, mas_u64* class_sum, unsigned* hist, unsigned flags, hipStream_t st) {
    switch (spx_dtype) {
        case MAS_ID_I64: return launch_low<CT, EXACT, long long>(zq, h, w, spx, B, C, H, W, S, invT, prob_sum, class_sum, hist, flags, st);
        case MAS_ID_I32: return launch_low<CT, EXACT, int>(zq, h, w, spx, B, C, H, W, S, invT, prob_sum, class_sum, hist, flags, st);
        case MAS_ID_U16: return launch_low<CT, EXACT, unsigned short>(zq, h, w, spx, B, C, H, W, S, invT, prob_sum, class_sum, hist, flags, st);
        default: return MAS_ERR_DTYPE;
    }
}

template <int CT, bool EXACT>
int dispatch_ids(const float* z, const void* spx, int spx_dtype, int B, int C, int H, int W, int S, float invT, mas_u64* prob_sum,
                 mas_u64* class_sum, unsigned* hist, hipStream_t st) {
    switch (spx_dtype) {
        case MAS_ID_I64: return launch<CT, EXACT, long long>(z, spx, B, C, H, W, S, invT, prob_sum, class_sum, hist, st);
        case MAS_ID_I32: return launch<CT, EXACT, int>(z, spx, B, C, H, W, S, invT, prob_sum, class_sum, hist, st);
        case MAS_ID_U16: return launch<CT, EXACT, unsigned short>(z, spx, B, C, H, W, S, invT, prob_sum, class_sum, hist, st);
        default: return MAS_ERR_DTYPE;
    }
}

}  // namespace

extern "C" int mas_single_pass_accum(const float* z, const void* spx, int spx_dtype, int B, int C, int H, int W, int S, float invT,
                                     uint64_t* prob_sum, uint64_t* class_sum, uint32_t* hist, void* stream) {
    if (!z || !spx || !prob_sum || !class_sum || !hist) return MAS_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || S <= 0 || (long long)H * W > (1LL << 23)) return MAS_ERR_SHAPE;
    if (C < 2 || C > MAS_MAX_CLASSES) return MAS_ERR_CLASSES;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mas_u64* ps = reinterpret_cast<mas_u64*>(prob_sum);
    mas_u64* cs = reinterpret_cast<mas_u64*>(class_sum);
    switch (C) {
        case 19: return dispatch_ids<19, true>(z, spx, spx_dtype, B, C, H, W, S, invT, ps, cs, hist, st);
        case 20: return dispatch_ids<20, true>(z, spx, spx_dtype, B, C, H, W, S, invT, ps, cs, hist, st);
        case 21: return dispatch_ids<21, true>(z, spx, spx_dtype, B, C, H, W, S, invT, ps, cs, hist, st);
        default: return dispatch_ids<MAS_MAX_CLASSES, false>(z, spx, spx_dtype, B, C, H, W, S, invT, ps, cs, hist, st);
    }
}

extern "C" int mas_single_pass_accum_lowres_opt(const float* zq, int h, int w, const void* spx, int spx_dtype, int B, int C, int H, int W, int S,
                                                float invT, uint64_t* prob_sum, uint64_t* class_sum, uint32_t* hist, unsigned flags, void* stream) {
    if (!zq || !spx || !prob_sum || !class_sum || !hist) return MAS_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || S <= 0 || h <= 0 || w <= 0 || h > H || w > W || (long long)H * W > (1LL << 23)) return MAS_ERR_SHAPE;
    if (flags & ~MAS_LOWRES_GENERIC) return MAS_ERR_RANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mas_u64* ps = reinterpret_cast<mas_u64*>(prob_sum);
    mas_u64* cs = reinterpret_cast<mas_u64*>(class_sum);
    switch (C) {                 // (the generic 32-channel footprint would not fit the LDS budget: the Cityscapes / VOC channel counts only)
        case 19: return dispatch_low_ids<19, true>(zq, h, w, spx, spx_dtype, B, C, H, W, S, invT, ps, cs, hist, flags, st);
        case 20: return dispatch_low_ids<20, true>(zq, h, w, spx, spx_dtype, B, C, H, W, S, invT, ps, cs, hist, flags, st);
        case 21: return dispatch_low_ids<21, true>(zq, h, w, spx, spx_dtype, B, C, H, W, S, invT, ps, cs, hist, flags, st);
        default: return MAS_ERR_CLASSES;
    }
}

extern "C" int mas_single_pass_accum_lowres(const float* zq, int h, int w, const void* spx, int spx_dtype, int B, int C, int H, int W, int S,
                                            float invT, uint64_t* prob_sum, uint64_t* class_sum, uint32_t* hist, void* stream) {
    return mas_single_pass_accum_lowres_opt(zq, h, w, spx, spx_dtype, B, C, H, W, S, invT, prob_sum, class_sum, hist, 0u, stream);
}

extern "C" int mas_region_finalize_weighted(const uint64_t* class_sum, const uint32_t* hist, int64_t n_regions, int C,
                                            const uint32_t* w31, int ban_class, float* score, int32_t* dominant, uint32_t* count,
                                            int64_t* hist_i64, void* stream) {
    if (!class_sum || !hist || !w31 || !score) return MAS_ERR_NULL;
    if (n_regions <= 0) return MAS_ERR_SHAPE;
    if (C < 1 || C > MAS_MAX_CLASSES) return MAS_ERR_CLASSES;
    const long long nblk = (n_regions + kThreads - 1) / kThreads;
    if (nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
    const size_t smem = (sizeof(mas_u64) + sizeof(unsigned)) * (size_t)kThreads * (C | 1);
    hipLaunchKernelGGL(k_region_finalize_weighted, dim3((unsigned)nblk), dim3(kThreads), smem, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const mas_u64*>(class_sum), hist, (long long)n_regions, C, w31, ban_class, score, dominant,
                       count, reinterpret_cast<long long*>(hist_i64));
    return mas_launch_status();
}
