// conv_sk.hip -- forward and input gradient of the dense convolutions in TRAINING (1x1 / 3x3, stride 1 / 2 forward, stride 1
// input gradient, any dilation <= 4) as a persistent stream-K implicit GEMM on the f32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32), NCHW in and out, reading the weight in the layout PyTorch stores it ([Cout][Cin][k][k]): nothing
// is re-packed after an optimizer step, and the input gradient is the same kernel with the roles of the weight's two channel
// axes swapped and the taps mirrored.
// Reference: the nn.Conv2d calls (and their autograd backward) of models/segmentation/backbone/resnet.py:129-171 and
// models/segmentation/deeplabv3.py:85-137,168-245 inside trainer/active_joint_multi_predignore_lossdecomp.py:83-116.
//
// GEMM view per picture:  Y[m, p] = sum_{tap, c} A[m, (tap, c)] * X[c, pixel p shifted by tap]
//   M tile 128 (or 64) output channels x N tile 128 output pixels (a TH x TW patch), K walked in chunks of CK channels x taps.
//   One workgroup = 8 waves = one CU: wave w owns rows 32 (w % WM) .. and pixel group w / WM of the tile; waves w, w + 4 share a
//   SIMD.  Two LDS buffers; chunk t + 1 travels global -> registers while chunk t is multiplied and is written to the other
//   buffer three quarters into the multiply phase; one barrier per chunk.  The pipeline runs across tile boundaries.
// Stream-K: the launch has one workgroup per CU; the (tile, chunk) iterations of the whole layer are dealt to them in equal
//   contiguous ranges, so the planes of the deep layers (48 x 48: 144 tiles for 256 CUs) load every CU the same.  A tile whose
//   chunks straddle workgroups is finished by the workgroup that owns its FIRST chunk: the others (for them it is the leading
//   segment of their range, done first) store their accumulators write-through (sc1) into their slot and publish an epoch
//   flag; the finisher polls, acquires, adds the slots in chunk order and runs the epilogue -- a fixed order, so results are
//   run-to-run identical.  No workgroup that is waited for ever waits itself, spins are bounded (error word).
// Epilogue: y = relu?(acc * scale[m] + shift[m] + residual) (all optional): the residual input also serves to accumulate the
//   input gradients of the two consumers of a block input.
#include <type_traits>

#include "common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
constexpr int kSkThreads = 512;
constexpr int kSkBN = 128;                      // pixels per tile
constexpr int kSkSlotFloats = 128 * kSkBN;      // accumulator image of a 128 x 128 tile
constexpr unsigned kSkSpinLimit = 1u << 22;

struct SkP {
    const float* x;             // [N, K, H, W] input of the product (forward: activations; input gradient: dY)
    const float* w;             // weight [Cout][Cin][taps] as PyTorch stores it
    const float* scale;
    const float* shift;
    const float* res;
    float* y;                   // [N, M, Ho, Wo]
    float* slots;               // [P][128 * 128] partial accumulator images
    unsigned* flags;            // [512] epoch flags, then one error word
    unsigned epoch;
    int K, H, W, M, Ho, Wo, stride, dil, pad, relu;
    long long w_rs, w_ks, w_elems;      // weight strides of an output row / a reduction channel, and the tensor's size
    int TH, tw_log2, tiles_x, tiles_y, ptiles, mtiles, nch;
    int PH, PWL, CS, PADL;              // LDS input patch: rows, row length (multiple of 4), channel stride, left halo columns
    long long iters;
    int P;
};

__device__ __forceinline__ void sk_store_sc1(float* p, v4f v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// global -> registers of iteration `it` (its tile and chunk are decoded here: the pipeline crosses tile boundaries)
template <int TAPS, int CK, int WM, bool DGRAD, bool VEC, int NWS, int NXS>
__device__ __forceinline__ void sk_fetch(const SkP& p, long long it, int tid, v4f (&wr)[NWS], v4f (&xr)[NXS]) {
    constexpr int BM = 32 * WM;
    constexpr int KC = TAPS * CK;
    const int tile = (int)(it / p.nch), chunk = (int)(it - (long long)tile * p.nch);
    const int mt = tile % p.mtiles, pt = tile / p.mtiles;
    const int tpi = p.tiles_x * p.tiles_y;
    const int n = pt / tpi, trem = pt - n * tpi;
    const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
    const int TW = 1 << p.tw_log2;
    const int m0 = mt * BM, k0 = chunk * CK;
    // ---- weight: BM rows x KC reduction elements, in memory order -----------------------------------------------------------
    //   forward:        element e -> row e / KC, (channel, tap) = e % KC            address (m0 + row) * rs + k0 * taps + e % KC
    //   input gradient: element e -> channel e / (BM * taps), (row, tap) = rest     address (k0 + ch) * ks + m0 * taps + rest
#pragma unroll
    for (int j = 0; j < NWS; ++j) {
        const int e = (tid + j * kSkThreads) * 4;
        long long off;
        bool ok = e < BM * KC;
        if (!DGRAD) {
            const int row = e / KC, rem = e - row * KC;
            off = (long long)(m0 + row) * p.w_rs + (long long)k0 * TAPS + rem;
            ok = ok && m0 + row < p.M;
        } else {
            const int ch = e / (BM * TAPS), rem = e - ch * (BM * TAPS);
            off = (long long)(k0 + ch) * p.w_ks + (long long)m0 * TAPS + rem;
            ok = ok && k0 + ch < p.K;
        }
        if (VEC) {
            wr[j] = (ok && off + 3 < p.w_elems) ? *reinterpret_cast<const v4f*>(p.w + off) : (v4f){0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) wr[j][i] = (ok && off + i < p.w_elems) ? p.w[off + i] : 0.0f;
        }
    }
    // ---- input patch: CK channels x PH rows x PWL columns, 16-byte groups aligned in memory --------------------------------
    const int oy0 = tyi * p.TH, ox0 = txi * TW;
    const int iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.PADL;
    const int f4r = p.PWL >> 2, f4c = p.PH * f4r;
    const int HW = p.H * p.W;
    const float* xb = p.x + ((size_t)n * p.K + k0) * HW;
#pragma unroll
    for (int j = 0; j < NXS; ++j) {
        const int f = tid + j * kSkThreads;
        const int c = f / f4c, rem = f - c * f4c;
        const int row = rem / f4r, col = (rem - row * f4r) * 4;
        const int iy = iy0 + row, ix = ix0 + col;
        const bool ok = f < CK * f4c && k0 + c < p.K && (unsigned)iy < (unsigned)p.H;
        const float* src = xb + (size_t)c * HW + (long long)iy * p.W + ix;
        if (VEC) {
            xr[j] = (ok && (unsigned)ix < (unsigned)p.W) ? *reinterpret_cast<const v4f*>(src) : (v4f){0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[j][i] = (ok && (unsigned)(ix + i) < (unsigned)p.W) ? src[i] : 0.0f;
        }
    }
}

// registers -> LDS.  Weight image [KC / 8][2][BM][4]: k-step kk = tap * (CK / 2) + c / 2 pairs the channels 2 cp + h of one tap
// (h = lane half of the MFMA), the four k-steps 4 q .. 4 q + 3 of one (half, row) are adjacent: one 16-byte read = four MFMAs.
template <int TAPS, int CK, int WM, bool DGRAD, int NWS, int NXS>
__device__ __forceinline__ void sk_stage(const SkP& p, float* __restrict__ sW, float* __restrict__ sX, int tid, const v4f (&wr)[NWS],
                                         const v4f (&xr)[NXS]) {
    constexpr int BM = 32 * WM;
    constexpr int KC = TAPS * CK;
#pragma unroll
    for (int j = 0; j < NWS; ++j) {
        const int e = (tid + j * kSkThreads) * 4;
        if (e < BM * KC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row, ch, tap;
                if (!DGRAD) {
                    row = (e + i) / KC;
                    const int rem = (e + i) - row * KC;
                    ch = rem / TAPS;
                    tap = rem - ch * TAPS;
                } else {
                    ch = (e + i) / (BM * TAPS);
                    const int rem = (e + i) - ch * (BM * TAPS);
                    row = rem / TAPS;
                    tap = TAPS - 1 - (rem - row * TAPS);            // mirrored taps
                }
                const int kk = tap * (CK / 2) + (ch >> 1), hh = ch & 1;
                sW[(((kk >> 2) * 2 + hh) * BM + row) * 4 + (kk & 3)] = wr[j][i];
            }
        }
    }
    const int f4r = p.PWL >> 2, f4c = p.PH * f4r;
#pragma unroll
    for (int j = 0; j < NXS; ++j) {
        const int f = tid + j * kSkThreads;
        if (f < CK * f4c) {
            const int c = f / f4c, rem = f - c * f4c;
            *reinterpret_cast<v4f*>(sX + c * p.CS + rem * 4) = xr[j];
        }
    }
}

template <int TAPS, int CK, int WM, bool DGRAD, bool VEC, int NXS>
__global__ __launch_bounds__(kSkThreads) void k_conv_sk(const SkP p) {
    constexpr int BM = 32 * WM, BN = kSkBN;
    constexpr int NG = 8 / WM;                  // pixel groups of the 8 waves
    constexpr int TN = BN / NG / 32;            // 32-pixel accumulator tiles per wave: 2 (BM 128) / 1 (BM 64)
    constexpr int KC = TAPS * CK;
    constexpr int NQ = KC / 8;                  // groups of four k-steps per chunk
    constexpr int QSPLIT = (3 * NQ + 3) / 4;
    constexpr int NWS = (BM * KC / 4 + kSkThreads - 1) / kSkThreads;
    static_assert(KC % 8 == 0 && TN >= 1, "tile");
    extern __shared__ __attribute__((aligned(16))) float sk_smem[];
    // [2][sW: KC * BM | sX: CK * CS] then sE [2 tile parities][2][BM]
    const int bufsz = KC * BM + CK * p.CS;
    float* sEbase = sk_smem + 2 * bufsz;
    int tile_parity = 0;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int mtw = wave % WM, ng = wave / WM;
    // logical workgroup index: the workgroups of one XCD (blocks b, b + 8, ...) own one contiguous run of iterations
    const int P = p.P;
    const int g = (P % 8 == 0) ? (int)(blockIdx.x & 7) * (P >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const long long it0 = p.iters * g / P, it1 = p.iters * (g + 1) / P;
    if (it0 >= it1) return;

    const int TW = 1 << p.tw_log2;
    const int aBase = (h * BM + mtw * 32 + l31) * 4;
    int bBase[TN], pl[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        pl[tn] = ng * (BN / NG) + tn * 32 + l31;
        const int ty = pl[tn] >> p.tw_log2, tx = pl[tn] & (TW - 1);
        bBase[tn] = h * p.CS + ty * p.stride * p.PWL + tx * p.stride + p.PADL - p.pad;
    }
    f32x16 acc[TN];
    v4f wr[NWS], xr[NXS];
    const int HWo = p.Ho * p.Wo;

    auto mfma_part = [&](int buf, auto QLc, auto QHc) {
        constexpr int qlo = decltype(QLc)::value, qhi = decltype(QHc)::value;
        const float* sW = sk_smem + buf * bufsz;
        const float* sX = sW + KC * BM;
#pragma unroll
        for (int q = qlo; q < qhi; ++q) {
            const v4f a = *reinterpret_cast<const v4f*>(sW + q * 8 * BM + aBase);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = 4 * q + j, tap = kk / (CK / 2), cp = kk - tap * (CK / 2);
                const int toff = (TAPS == 1) ? 0 : ((tap / 3) * p.PWL + (tap % 3)) * p.dil;
                const float* xrow = sX + 2 * cp * p.CS + toff;
                float b[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) b[tn] = xrow[bBase[tn]];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[tn], acc[tn], 0, 0, 0);
            }
        }
    };

    int buf = 0;
    sk_fetch<TAPS, CK, WM, DGRAD, VEC, NWS, NXS>(p, it0, tid, wr, xr);
    sk_stage<TAPS, CK, WM, DGRAD, NWS, NXS>(p, sk_smem, sk_smem + KC * BM, tid, wr, xr);
    __syncthreads();

    long long it = it0;
    while (it < it1) {
        const int tile = (int)(it / p.nch);
        const int c0 = (int)(it - (long long)tile * p.nch);
        const long long left = it1 - it;
        const int c1 = (left < (long long)(p.nch - c0)) ? c0 + (int)left : p.nch;
        const int mt = tile % p.mtiles, pt = tile / p.mtiles;
        const int m0 = mt * BM;
        // epilogue constants of this tile; the buffer alternates per tile: slower waves may still read the previous tile's
        float* sE = sEbase + tile_parity * 2 * BM;
        tile_parity ^= 1;
        if (tid < BM) {
            const bool real = m0 + tid < p.M;
            sE[tid] = (p.scale && real) ? p.scale[m0 + tid] : 1.0f;
            sE[BM + tid] = (p.scale && real) ? p.shift[m0 + tid] : 0.0f;
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tn][r] = 0.0f;
        for (int c = c0; c < c1; ++c, ++it) {
            const bool more = it + 1 < it1;
            if (more) sk_fetch<TAPS, CK, WM, DGRAD, VEC, NWS, NXS>(p, it + 1, tid, wr, xr);
            mfma_part(buf, std::integral_constant<int, 0>{}, std::integral_constant<int, QSPLIT>{});
            if (more) {
                float* nW = sk_smem + (buf ^ 1) * bufsz;
                sk_stage<TAPS, CK, WM, DGRAD, NWS, NXS>(p, nW, nW + KC * BM, tid, wr, xr);
            }
            mfma_part(buf, std::integral_constant<int, QSPLIT>{}, std::integral_constant<int, NQ>{});
            __syncthreads();
            buf ^= 1;
        }
        // ---- the tile's segment [c0, c1) is in the accumulators ------------------------------------------------------------
        const bool first = c0 == 0, last = c1 == p.nch;
        if (!first) {
            // contributor: accumulator image -> slot g (write-through), then ONE lane publishes the epoch
            float* slot = p.slots + (size_t)g * kSkSlotFloats;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const v4f v = {acc[tn][4 * r4], acc[tn][4 * r4 + 1], acc[tn][4 * r4 + 2], acc[tn][4 * r4 + 3]};
                    sk_store_sc1(slot + ((size_t)(tn * 4 + r4) * kSkThreads + tid) * 4, v);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains
            __syncthreads();
            if (tid == 0) __hip_atomic_store((gu32*)(p.flags + g), p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (!last) {
            // finisher: add the slots of the workgroups that own the rest of this tile, in chunk order
            const long long tile_end = (long long)(tile + 1) * p.nch;
            for (int gg = g + 1; gg < P; ++gg) {
                const long long b0 = p.iters * gg / P;
                if (b0 >= tile_end) break;
                if (wave == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load((gu32*)(p.flags + gg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > kSkSpinLimit) {
                            if (lane == 0) atomicOr(p.flags + 512, 1u);   // give up: the error word says so, nothing hangs
                            break;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                const float* slot = p.slots + (size_t)gg * kSkSlotFloats;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const v4f v = *reinterpret_cast<const v4f*>(slot + ((size_t)(tn * 4 + r4) * kSkThreads + tid) * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[tn][4 * r4 + i] += v[i];
                    }
            }
        }
        // ---- epilogue: accumulator (row = (r & 3) + 8 (r >> 2) + 4 h, column = lane & 31) -> NCHW -----------------------------
        const int tpi = p.tiles_x * p.tiles_y;
        const int n = pt / tpi, trem = pt - n * tpi;
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int oy0 = tyi * p.TH, ox0 = txi * TW;
        float* yb = p.y + ((size_t)n * p.M + m0) * HWo;
        const float* rb = p.res ? p.res + ((size_t)n * p.M + m0) * HWo : nullptr;
        const float lo = p.relu ? 0.0f : -INFINITY;
        const int mlim = p.M - m0;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int oy = oy0 + (pl[tn] >> p.tw_log2), ox = ox0 + (pl[tn] & (TW - 1));
            const bool inside = oy < p.Ho && ox < p.Wo;
            const int po = inside ? oy * p.Wo + ox : 0;
            const int mb = mtw * 32 + 4 * h;
            float rv[16];
            if (rb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    rv[r] = rb[(size_t)(m < mlim ? m : 0) * HWo + po];
                }
            }
            if (inside) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    float v = mas_fmaf(acc[tn][r], sE[m], sE[BM + m]);
                    if (rb) v += rv[r];
                    v = v < lo ? lo : v;
                    if (m < mlim) yb[(size_t)m * HWo + po] = v;
                }
            }
        }
    }
}

struct SkGeom {
    int TAPS, CK, BM, TW, TH, PH, PWL, CS, PADL, nxs;
    size_t smem;
};

inline int sk_ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

inline void sk_geom(int ksize, int stride, int dil, int M, int Wo, SkGeom* g) {
    g->TAPS = ksize * ksize;
    g->CK = ksize == 3 ? 8 : (stride == 2 ? 16 : 64);       // (the stride-2 patch of a 1x1 holds 4x the pixels it uses)
    g->BM = M > 64 ? 128 : 64;
    g->TW = (Wo >= 32 && !(Wo % 32 != 0 && Wo % 16 == 0)) ? 32 : 16;       // 48-wide planes: three exact 16-wide tiles
    g->TH = kSkBN / g->TW;
    const int pad = ksize == 3 ? dil : 0;
    g->PADL = ksize == 3 ? 4 : 0;
    g->PH = (g->TH - 1) * stride + 1 + 2 * pad;
    g->PWL = ((g->TW - 1) * stride + 1 + 2 * g->PADL + 3) & ~3;
    g->CS = g->PH * g->PWL;                          // multiple of 4: 16-byte LDS stores
    const int f4 = g->CK * g->CS / 4;
    g->nxs = (f4 + kSkThreads - 1) / kSkThreads;
    g->smem = sizeof(float) * (2 * ((size_t)g->TAPS * g->CK * g->BM + (size_t)g->CK * g->CS) + 4 * g->BM);
}

int sk_num_cus() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

template <int TAPS, int CK, int WM, bool DGRAD, bool VEC, int NXS>
int sk_launch(const SkP& p, size_t smem, hipStream_t st) {
    auto kern = &k_conv_sk<TAPS, CK, WM, DGRAD, VEC, NXS>;
    if (smem > 64 * 1024) {
        static bool raised[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        if (dev < 0 || dev >= 64 || !raised[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0 && dev < 64) raised[dev] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)p.P), dim3(kSkThreads), smem, st, p);
    return mas_launch_status();
}

// staged 16-byte groups of the input patch per thread: fixed per kernel family (checked against the geometry at launch)
//   3x3 stride 1: CK 8 x (TH + 2 dil) x (TW + 8) floats <= 2 groups;  3x3 stride 2: 3;  1x1: 4
template <int TAPS, int CK, int NXS, bool DGRAD>
int sk_dispatch(const SkP& p, const SkGeom& g, bool vec, hipStream_t st) {
    if (g.nxs > NXS) return MAS_ERR_SHAPE;
    if (g.BM == 128) return vec ? sk_launch<TAPS, CK, 4, DGRAD, true, NXS>(p, g.smem, st) : sk_launch<TAPS, CK, 4, DGRAD, false, NXS>(p, g.smem, st);
    return vec ? sk_launch<TAPS, CK, 2, DGRAD, true, NXS>(p, g.smem, st) : sk_launch<TAPS, CK, 2, DGRAD, false, NXS>(p, g.smem, st);
}
}  // namespace

extern "C" size_t mas_conv_sk_workspace_bytes(void) {
    // slots for up to 512 workgroups + their flags + the error word (padded)
    return (size_t)512 * kSkSlotFloats * sizeof(float) + 4096;
}

extern "C" int mas_conv_sk(const float* x, const float* w, int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil, int dgrad,
                           const float* scale, const float* shift, const float* residual, int relu, float* y, void* workspace,
                           size_t workspace_bytes, unsigned epoch, void* stream) {
    if (!x || !w || !y || !workspace) return MAS_ERR_NULL;
    if ((scale == nullptr) != (shift == nullptr)) return MAS_ERR_NULL;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return MAS_ERR_SHAPE;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || dil < 1 || dil > 4) return MAS_ERR_RANGE;
    if (ksize == 1 && dil != 1) return MAS_ERR_RANGE;
    if (dgrad && stride != 1) return MAS_ERR_RANGE;
    if (stride == 2 && dil != 1) return MAS_ERR_RANGE;
    if (epoch == 0) return MAS_ERR_RANGE;
    if (workspace_bytes < mas_conv_sk_workspace_bytes()) return MAS_ERR_WORKSPACE;
    if ((long long)(Cin > Cout ? Cin : Cout) * H * W > 0x7fffffffLL) return MAS_ERR_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int taps = ksize * ksize;
    SkP p;
    p.x = x; p.w = w; p.scale = scale; p.shift = shift; p.res = residual; p.y = y;
    p.slots = static_cast<float*>(workspace);
    p.flags = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + (size_t)512 * kSkSlotFloats * sizeof(float));
    p.epoch = epoch;
    p.stride = stride; p.dil = dil; p.pad = ksize == 3 ? dil : 0; p.relu = relu;
    p.H = H; p.W = W;
    if (!dgrad) {
        p.K = Cin; p.M = Cout;
        p.w_rs = (long long)Cin * taps; p.w_ks = taps;
        p.Ho = (H - 1) / stride + 1; p.Wo = (W - 1) / stride + 1;
    } else {                        // x = dY [N, Cout, H, W] (stride 1: the same plane), y = dX [N, Cin, H, W]
        p.K = Cout; p.M = Cin;
        p.w_rs = taps; p.w_ks = (long long)Cin * taps;
        p.Ho = H; p.Wo = W;
    }
    p.w_elems = (long long)Cout * Cin * taps;
    SkGeom g;
    sk_geom(ksize, stride, dil, p.M, p.Wo, &g);
    if (g.smem > 160 * 1024) return MAS_ERR_SHAPE;
    p.TH = g.TH; p.tw_log2 = sk_ilog2(g.TW);
    p.tiles_x = (p.Wo + g.TW - 1) / g.TW;
    p.tiles_y = (p.Ho + g.TH - 1) / g.TH;
    p.ptiles = N * p.tiles_x * p.tiles_y;
    p.mtiles = (p.M + g.BM - 1) / g.BM;
    p.nch = (p.K + g.CK - 1) / g.CK;
    p.PH = g.PH; p.PWL = g.PWL; p.CS = g.CS; p.PADL = g.PADL;
    p.iters = (long long)p.ptiles * p.mtiles * p.nch;
    const int cus = sk_num_cus();
    p.P = (int)(p.iters < cus ? p.iters : cus);
    if (p.P > 512) p.P = 512;
    // 16-byte global loads: the weight rows and the planes must keep 16-byte groups whole and aligned
    const bool vec = ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && (W % 4 == 0) && (((long long)Cin * taps) % 4 == 0);
    if (ksize == 3) {
        if (stride == 2) return sk_dispatch<9, 8, 3, false>(p, g, vec, st);
        return dgrad ? sk_dispatch<9, 8, 2, true>(p, g, vec, st) : sk_dispatch<9, 8, 2, false>(p, g, vec, st);
    }
    if (stride == 2) return sk_dispatch<1, 16, 4, false>(p, g, vec, st);
    return dgrad ? sk_dispatch<1, 64, 4, true>(p, g, vec, st) : sk_dispatch<1, 64, 4, false>(p, g, vec, st);
}

/* error word of the last launches on this workspace: non-zero = a bounded spin gave up (host-side read, for tests) */
extern "C" int mas_conv_sk_error(const void* workspace, unsigned* out_host) {
    if (!workspace || !out_host) return MAS_ERR_NULL;
    const char* base = static_cast<const char*>(workspace) + (size_t)512 * kSkSlotFloats * sizeof(float);
    hipError_t e = hipMemcpy(out_host, base + 512 * sizeof(unsigned), sizeof(unsigned), hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : (int)e;
}
