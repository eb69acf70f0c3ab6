// conv_mfma.hip -- dense 2-D convolution (1x1 and 3x3; stride 1 / 2; any dilation) as an implicit GEMM on the f32 matrix
// cores of gfx950 (v_mfma_f32_32x32x2_f32: exact f32, every product rounded once, accumulation in k order), NCHW in and
// out, with the layers that follow a convolution in the network folded into the epilogue:
//     y = relu?( conv(x, w) * scale[m] + shift[m] + residual )          (inference BatchNorm, residual add, ReLU)
// Reference: the convolutions of models/segmentation/backbone/resnet.py:129-160 (Bottleneck: conv1x1 - bn - relu,
// conv3x3 - bn - relu, conv1x1 - bn - (+identity) - relu), the deep stem (:163-171), the ASPP / decoder 1x1 projections
// of models/segmentation/deeplabv3.py:85-137,216-245.
//
// GEMM view per picture:  Y[m, p] = sum_{tap, c} Wt[(tap, c), m] * X[c, pixel p shifted by tap]
//   M = output channels (A operand, from a re-arranged weight [chunk][k-step / 4][lane half][M][4 k-steps]: the four
//       A values a lane needs for four consecutive MFMAs are one 16-byte LDS read),
//   N = output pixels   (B operand: a lane reads ITS pixel of an LDS-resident input patch, tap shifts are address offsets),
//   K = taps * Cin, walked in chunks of CK input channels.
// A workgroup (4 waves) owns BM channels x BN pixels (a TH x TW patch of one output plane); a wave owns 64 x (BN / WN)
// of it as 32x32 MFMA tiles.  The two k values of one MFMA (lane halves) are the channels c, c + 1 of the same tap, so
// per step the operand addresses of both halves differ by a constant and the step offset is wave-uniform.
// Per chunk: global -> registers (issued before the MFMAs of the previous chunk), registers -> LDS after it (one LDS
// buffer, two workgroups per CU overlap each other's staging), every input element is fetched once per workgroup and
// reused for all taps and all BM channels.  Blocks that share an input patch (the M tiles of one pixel tile) sit on one
// XCD so that the re-reads hit its L2.
#include "common.h"

namespace {
constexpr int kThreads = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));      // a native vector (HIP's float4 is a struct: arrays of it stay in scratch)

struct ConvP {
    const float* x;
    const float* wt;
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    int Cin, H, W, Cout, CoutP, Ho, Wo, stride, dil, pad, relu;      // CoutP = Cout rounded up to 64: the M extent of the packed weight
    int tw_log2, TH;                    // output tile: TH rows x (1 << tw_log2) columns, TH << tw_log2 == BN
    int tiles_x, tiles_y, ptiles, mtiles;
    int PH, PW, CS;                     // LDS input patch: rows, columns, channel stride (floats)
};

// global -> registers: the weight tile (NW float4 per thread) and the input patch (NXMAX elements or float4s per thread);
// addresses outside the plane were redirected to offset 0 when the descriptors were built, so every load is unconditional
template <int NW, int NXMAX, bool VEC>
__device__ __forceinline__ void conv_fetch(const float* __restrict__ xc, const float* __restrict__ wc, const int (&woff)[NW],
                                           const int (&goff)[NXMAX], v4f (&wr)[NW], v4f (&xr)[NXMAX]) {
#pragma unroll
    for (int j = 0; j < NW; ++j) wr[j] = *reinterpret_cast<const v4f*>(wc + woff[j]);
#pragma unroll
    for (int j = 0; j < NXMAX; ++j) {
        if (VEC) xr[j] = *reinterpret_cast<const v4f*>(xc + goff[j]);
        else xr[j].x = xc[goff[j]];
    }
}

// registers -> LDS
template <int NW, int NXMAX, bool VEC, int W4>
__device__ __forceinline__ void conv_stage(float* __restrict__ sW, float* __restrict__ sX, int tid, const int (&loff)[NXMAX], unsigned live,
                                           unsigned ok, const v4f (&wr)[NW], const v4f (&xr)[NXMAX]) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        int f = tid + j * kThreads;
        if (f >= W4) f = W4 - 1;
        *reinterpret_cast<v4f*>(sW + 4 * f) = wr[j];
    }
#pragma unroll
    for (int j = 0; j < NXMAX; ++j) {
        if (live & (1u << j)) {
            const bool v = ok & (1u << j);
            if (VEC) *reinterpret_cast<v4f*>(sX + loff[j]) = v ? xr[j] : (v4f){0.f, 0.f, 0.f, 0.f};
            else sX[loff[j]] = v ? xr[j].x : 0.0f;
        }
    }
}

template <int TAPS, int CK, int BM, int BN, int NXMAX, bool VEC, bool RES, bool MPAD>
__global__ __launch_bounds__(kThreads, 2) void k_conv_mfma(const ConvP p) {
    constexpr int WM = BM / 64, WN = 4 / WM, TN = BN / WN / 32;
    constexpr int KC = TAPS * CK;
    constexpr int W4 = KC * BM / 4;
    constexpr int NW = (W4 + kThreads - 1) / kThreads;
    static_assert(BM == 64 || BM == 128, "BM");
    static_assert(TN >= 1 && KC % 8 == 0, "tile");
    extern __shared__ __attribute__((aligned(16))) float conv_smem[];
    float* sW = conv_smem;                  // [KC / 8][2][BM][4]
    float* sE = conv_smem + KC * BM;        // [2][BM]: epilogue scale, shift
    float* sX = sE + 2 * BM;                // [CK][CS]

    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int mt = slot % p.mtiles;
    const int pt = (slot / p.mtiles) * 8 + xcd;
    if (pt >= p.ptiles) return;
    const int tpi = p.tiles_x * p.tiles_y;
    const int n = pt / tpi;
    const int trem = pt - n * tpi;
    const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
    const int TW = 1 << p.tw_log2;
    const int oy0 = tyi * p.TH, ox0 = txi * TW;
    const int m0 = mt * BM;
    const int HW = p.H * p.W;

    // ---- staging descriptors (the same for every chunk; only the base pointers advance) -------------------------------
    int goff[NXMAX], loff[NXMAX];
    unsigned ok = 0, live = 0;
    {
        const int per_c = p.PH * p.PW;
        const int total = VEC ? (CK * per_c) / 4 : CK * per_c;
        const int iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
#pragma unroll
        for (int j = 0; j < NXMAX; ++j) {
            const int e = (tid + j * kThreads) * (VEC ? 4 : 1);
            goff[j] = 0;
            loff[j] = 0;
            if (e < (VEC ? total * 4 : total)) {
                const int c = e / per_c, rem = e - c * per_c;
                const int py = rem / p.PW, px = rem - py * p.PW;
                const int iy = (TAPS == 1) ? (oy0 + py) * p.stride : iy0 + py;
                const int ix = (TAPS == 1) ? (ox0 + px) * p.stride : ix0 + px;
                live |= 1u << j;
                loff[j] = c * p.CS + py * p.PW + px;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                    ok |= 1u << j;
                    goff[j] = c * HW + iy * p.W + ix;
                }
            }
        }
    }
    int woff[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        int f = tid + j * kThreads;
        if (f >= W4) f = W4 - 1;                        // clamped duplicate (same value written twice)
        const int row = f / BM, m = f - row * BM;          // row = (k-step / 4) * 2 + lane half
        woff[j] = (row * p.CoutP + m) * 4;
    }

    // ---- MFMA operand addressing ---------------------------------------------------------------------------------------
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int aBase = (h * BM + wm * 64 + l31) * 4;
    int bBase[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pl = wn * (BN / WN) + tn * 32 + l31;
        const int ty = pl >> p.tw_log2, tx = pl & (TW - 1);
        bBase[tn] = h * p.CS + ((TAPS == 1) ? ty * p.PW + tx : ty * p.stride * p.PW + tx * p.stride);
    }
    f32x16 acc[2][TN];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;

    const float* xb = p.x + (size_t)n * p.Cin * HW;
    const float* wb = p.wt + (size_t)m0 * 4;
    const int nchunks = p.Cin / CK;
    v4f wr[NW];
    v4f xr[NXMAX];

    if (tid < BM) {
        const bool real = !MPAD || m0 + tid < p.Cout;   // rows of the 64-padding carry zero weights and are never stored
        sE[tid] = (p.scale && real) ? p.scale[m0 + tid] : 1.0f;
        sE[BM + tid] = (p.scale && real) ? p.shift[m0 + tid] : 0.0f;
    }
    // epilogue geometry (needed early: the residual values are fetched while the last chunk's MFMAs run)
    const int HWo = p.Ho * p.Wo;
    int po[TN];
    bool inside[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pl = wn * (BN / WN) + tn * 32 + l31;
        const int oy = oy0 + (pl >> p.tw_log2), ox = ox0 + (pl & (TW - 1));
        inside[tn] = oy < p.Ho && ox < p.Wo;
        po[tn] = inside[tn] ? oy * p.Wo + ox : 0;
    }
    const float* rb = RES ? p.res + ((size_t)n * p.Cout + m0) * HWo : nullptr;

    auto mfma_chunk = [&]() {
#pragma unroll
        for (int q = 0; q < KC / 8; ++q) {
            const v4f a0 = *reinterpret_cast<const v4f*>(sW + q * 8 * BM + aBase);
            const v4f a1 = *reinterpret_cast<const v4f*>(sW + q * 8 * BM + aBase + 128);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = 4 * q + j, tap = kk / (CK / 2), cp = kk - tap * (CK / 2);
                const int toff = (TAPS == 1) ? 0 : ((tap / 3) * p.PW + (tap % 3)) * p.dil;
                const float* xrow = sX + 2 * cp * p.CS + toff;
                float b[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) b[tn] = xrow[bBase[tn]];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b[tn], acc[0][tn], 0, 0, 0);
                    acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b[tn], acc[1][tn], 0, 0, 0);
                }
            }
        }
    };
    conv_fetch<NW, NXMAX, VEC>(xb, wb, woff, goff, wr, xr);
    for (int t = 0; t + 1 < nchunks; ++t) {
        conv_stage<NW, NXMAX, VEC, W4>(sW, sX, tid, loff, live, ok, wr, xr);
        __syncthreads();
        conv_fetch<NW, NXMAX, VEC>(xb + (size_t)(t + 1) * CK * HW, wb + (size_t)(t + 1) * KC * p.CoutP, woff, goff, wr, xr);
        mfma_chunk();
        __syncthreads();
    }
    // last chunk (peeled: nothing left to prefetch)
    conv_stage<NW, NXMAX, VEC, W4>(sW, sX, tid, loff, live, ok, wr, xr);
    __syncthreads();
    mfma_chunk();

    // ---- epilogue: accumulator (row = (r & 3) + 8 (r >> 2) + 4 h, column = lane & 31) -> NCHW --------------------------
    // Branch-free per element: scale / shift come from LDS (1 / 0 without a BatchNorm), the 16 residual values of a tile are
    // loaded together (clamped address outside the plane; fetching all 64 behind the last chunk's MFMAs was measured: the
    // registers it pins cost more than the latency it hides), one predicate guards the 16 stores of a tile.
    float* yb = p.y + ((size_t)n * p.Cout + m0) * HWo;
    const float lo = p.relu ? 0.0f : -INFINITY;
    // the residual values of tile i + 1 are requested before tile i is finished (two 16-register sets): half of the memory round
    // trips of the per-tile form are hidden, without pinning all 64 values (measured slower) 
    constexpr int NT = 2 * TN;
    float rv[2][16];
    auto res_load = [&](int i, float (&dst)[16]) {
        const int tn = i >> 1, tm = i & 1;
        const int mb = wm * 64 + tm * 32 + 4 * h;
        const int mlim = MPAD ? p.Cout - m0 : (1 << 30);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            // (the residual is read once: non-temporal, see common.h:mas_load_stream4 -- 26.32 -> 26.20 ms per pool batch; the same policy
            //  on the INPUT patches of layers with one M tile was measured too: 26.30 -> 26.37, not kept)
            dst[r] = __builtin_nontemporal_load(&rb[(size_t)(m < mlim ? m : 0) * HWo + po[tn]]);
        }
    };
    if (RES) res_load(0, rv[0]);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int tn = i >> 1, tm = i & 1;
        if (RES && i + 1 < NT) res_load(i + 1, rv[(i + 1) & 1]);
        const int mb = wm * 64 + tm * 32 + 4 * h;
        const int mlim = MPAD ? p.Cout - m0 : (1 << 30);    // rows >= mlim are padding (MPAD: Cout % 64 != 0)
        float out[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            float v = mas_fmaf(acc[tm][tn][r], sE[m], sE[BM + m]);
            if (RES) v += rv[i & 1][r];
            out[r] = v < lo ? lo : v;
        }
        if (inside[tn]) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m < mlim) yb[(size_t)m * HWo + po[tn]] = out[r];
            }
        }
    }
}

inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

template <int TAPS, int CK, int BM, int BN, int NXMAX, bool VEC, bool RES, bool MPAD>
int launch_res(ConvP p, int N, hipStream_t st) {
    const int TW = 1 << p.tw_log2;
    p.TH = BN / TW;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + p.TH - 1) / p.TH;
    p.ptiles = N * p.tiles_x * p.tiles_y;
    p.mtiles = p.CoutP / BM;
    if (TAPS == 1) {
        p.PH = p.TH;
        p.PW = TW;
    } else {
        p.PH = (p.TH - 1) * p.stride + 2 * p.dil + 1;
        p.PW = (TW - 1) * p.stride + 2 * p.dil + 1;
    }
    p.CS = p.PH * p.PW;
    if (VEC) p.CS = (p.CS + 3) & ~3;
    const int per = CK * p.PH * p.PW;
    if ((VEC ? per / 4 : per) > NXMAX * kThreads) return MAS_ERR_SHAPE;
    const size_t smem = sizeof(float) * ((size_t)TAPS * CK * BM + 2 * BM + (size_t)CK * p.CS);
    if (smem > 80 * 1024) return MAS_ERR_SHAPE;        // two workgroups per CU must fit the 160 KB
    if (smem > 64 * 1024) {
        static bool raised[64] = {};                   // above the default dynamic-LDS limit: raise it once per (instantiation, device)
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        if (dev < 0 || dev >= 64 || !raised[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_mfma<TAPS, CK, BM, BN, NXMAX, VEC, RES, MPAD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0 && dev < 64) raised[dev] = true;
        }
    }
    const long long nblk = 8LL * ((p.ptiles + 7) / 8) * p.mtiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
    hipLaunchKernelGGL((k_conv_mfma<TAPS, CK, BM, BN, NXMAX, VEC, RES, MPAD>), dim3((unsigned)nblk), dim3(kThreads), smem, st, p);
    return mas_launch_status();
}

template <int TAPS, int CK, int BM, int BN, int NXMAX, bool VEC>
int launch(ConvP p, int N, hipStream_t st) {
    if constexpr (BM == 64) {           // output-channel counts that are not a multiple of 64 always take a 64-row M tile
        if (p.Cout % 64 != 0)
            return p.res ? launch_res<TAPS, CK, BM, BN, NXMAX, VEC, true, true>(p, N, st) : launch_res<TAPS, CK, BM, BN, NXMAX, VEC, false, true>(p, N, st);
    }
    if (p.Cout % 64 != 0) return MAS_ERR_SHAPE;
    return p.res ? launch_res<TAPS, CK, BM, BN, NXMAX, VEC, true, false>(p, N, st) : launch_res<TAPS, CK, BM, BN, NXMAX, VEC, false, false>(p, N, st);
}
}  // namespace

extern "C" int mas_conv_chunk(int ksize, int Cin) {
    if (ksize == 3) return Cin % 8 == 0 ? 8 : 0;
    if (ksize == 1) return Cin % 32 == 0 ? 32 : (Cin % 16 == 0 ? 16 : 0);
    return 0;
}

extern "C" int mas_conv_fwd(const float* x, const float* wt, int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil,
                            const float* scale, const float* shift, const float* residual, int relu, float* y, void* stream) {
    if (!x || !wt || !y) return MAS_ERR_NULL;
    if ((scale == nullptr) != (shift == nullptr)) return MAS_ERR_NULL;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return MAS_ERR_SHAPE;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || dil < 1 || dil > 4) return MAS_ERR_RANGE;
    if (ksize == 1 && dil != 1) return MAS_ERR_RANGE;
    if (mas_conv_chunk(ksize, Cin) == 0) return MAS_ERR_SHAPE;
    if ((long long)Cin * H * W > 0x7fffffffLL) return MAS_ERR_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ConvP p;
    p.x = x; p.wt = wt; p.scale = scale; p.shift = shift; p.res = residual; p.y = y;
    p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.CoutP = (Cout + 63) / 64 * 64;
    p.stride = stride; p.dil = dil; p.pad = ksize == 3 ? dil : 0; p.relu = relu;
    p.Ho = (H - 1) / stride + 1;
    p.Wo = (W - 1) / stride + 1;
    // tile width: 32 columns when the plane has them, 16 for the 48 / 49-wide planes of the deep layers
    const int TW = p.Wo >= 32 ? 32 : 16;
    p.tw_log2 = ilog2(TW);
    // Tile choice.  The chip holds 512 workgroups at a time (2 per CU); a launch with fewer than ~400 leaves matrix pipes idle for
    // its whole duration (at exactly 512 = one full round the big tile still wins: measured), so small planes (crop-sized inputs, the deep layers) trade operand reuse for workgroups: 64-row M
    // tiles instead of 128, 128-pixel tiles instead of 256.
    const long long px128 = (long long)N * ((p.Ho + 128 / TW - 1) / (128 / TW)) * ((p.Wo + TW - 1) / TW);
    const long long px256 = (long long)N * ((p.Ho + 256 / TW - 1) / (256 / TW)) * ((p.Wo + TW - 1) / TW);
    const bool big_m = p.CoutP % 128 == 0 && Cout % 64 == 0 && px128 * (p.CoutP / 128) >= 400;
    const bool wide_n = px256 * (p.CoutP / 64) >= 400;         // (for the 64-row tiles) 256-pixel tiles only when there are enough of them
    if (ksize == 1) {
        const bool vec = stride == 1 && (W % 4 == 0) && ((uintptr_t)x % 16 == 0);
        const int ck = mas_conv_chunk(1, Cin);
        // (for one tap the packed weight is the same memory image for every chunk size that divides Cin -- k-steps are plain
        // channel pairs in order.  64-channel chunks, half the barriers per MFMA, were measured: 256 VGPRs, 3 % SLOWER.)
        if (ck == 32) {
            if (big_m) return vec ? launch<1, 32, 128, 128, 4, true>(p, N, st) : launch<1, 32, 128, 128, 16, false>(p, N, st);
            if (!vec) return launch<1, 32, 64, 128, 16, false>(p, N, st);
            return wide_n ? launch<1, 32, 64, 256, 8, true>(p, N, st) : launch<1, 32, 64, 128, 4, true>(p, N, st);
        }
        if (big_m) return vec ? launch<1, 16, 128, 128, 2, true>(p, N, st) : launch<1, 16, 128, 128, 8, false>(p, N, st);
        if (!vec) return launch<1, 16, 64, 128, 8, false>(p, N, st);
        return wide_n ? launch<1, 16, 64, 256, 4, true>(p, N, st) : launch<1, 16, 64, 128, 2, true>(p, N, st);
    }
    if (stride == 2) {
        if (dil != 1) return MAS_ERR_RANGE;
        // (always the 64-row tile: with 20 staging registers the 128-row form needs 256 VGPRs and spills 35 of them inside its loop --
        //  428 vs 446 us for 128 -> 128 at 256 x 512, 394 vs 405 us for 256 -> 256 at 128 x 256)
        return launch<9, 8, 64, 128, 20, false>(p, N, st);
    }
    if (big_m) return launch<9, 8, 128, 128, 9, false>(p, N, st);
    return (dil == 1 && wide_n) ? launch<9, 8, 64, 256, 12, false>(p, N, st) : launch<9, 8, 64, 128, 12, false>(p, N, st);
}
