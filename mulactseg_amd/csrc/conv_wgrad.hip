// conv_wgrad.hip -- weight gradient of the dense convolutions (1x1 / 3x3, stride 1 / 2, dilation 1 / 2 / 4) on the f32
// matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact f32 products, k-ordered accumulation), NCHW operands as autograd
// hands them over -- no NHWC copies:
//     dW[m, c, ky, kx] = sum_{n, oy, ox} dY[n, m, oy, ox] * X[n, c, oy*stride + ky*dil - pad, ox*stride + kx*dil - pad]
// Reference: the backward of every nn.Conv2d of models/segmentation/backbone/resnet.py:129-171 and
// models/segmentation/deeplabv3.py:85-137,168-245 as driven by trainer/active_joint_multi_predignore_lossdecomp.py:83-116
// (loss.backward()).
//
// GEMM view:  M = output channels (A operand = dY), N = (tap, input channel) (B operand = X shifted by the tap), K = output
// pixels of all pictures.  Both operands are pixel-contiguous in NCHW, i.e. K-contiguous: 16-byte global loads on both sides.
//   * K is walked in chunks of KP pixels (a TH x TW patch of one output plane, or KP consecutive pixels for 1x1 / stride 1).
//     Per chunk a workgroup stages dY[BM rows][KP] and the input patch X[BC channels][PH x PWL] that covers the pixels of the
//     chunk under all nine taps -- every input element is fetched once per chunk and reused for all taps and all BM rows.
//   * The two k values of one MFMA (lane halves) are pixels 8j + i and 8j + 4 + i, so the A operand of four consecutive
//     MFMAs is ONE 16-byte LDS read; the B operand is the lane's channel at a tap-shifted pixel: per-lane base + immediate.
//   * One workgroup = 8 waves = one CU (LDS 90-140 KB): waves w and w + 4 share a SIMD and own complementary halves of the
//     n-tiles (taps 0..4 / 5..8), so every SIMD issues the same number of MFMAs; two LDS buffers, chunk t + 1 travels
//     global -> registers while chunk t is multiplied, one barrier per chunk.  The kernel does not depend on a second
//     resident workgroup to hide its staging, so a launch of k * 256 workgroups fills the chip evenly.
//   * split K: the grid is (output tiles) x S pixel ranges; every workgroup writes its partial tile to a workspace
//     [S][Cout][taps][Cin] and k_wgrad_reduce adds the S slices in a fixed order (run-to-run identical, no atomics) into
//     dW[Cout][Cin][k][k].  Workgroups of one pixel range sit on one XCD (they stream the same chunks through its L2).
#include <type_traits>

#include "common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v4fu __attribute__((ext_vector_type(4), aligned(4)));
constexpr int kWgThreads = 512;

template <int I, int N, typename F>
__device__ __forceinline__ void wg_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wg_static_for<I + 1, N>(f);
    }
}

struct WgP {
    const float* x;
    const float* dy;
    float* part;
    int N, Cin, H, W, Cout, Ho, Wo;
    int tiles_x, tiles_y, nch;          // K chunks: pictures x pixel tiles (PATCH) / pictures x ceil(HW / KP) (FLAT)
    int mtiles, ctiles, S;
};

template <int TAPS, int WM, int CG, int TW, int KP, int STRIDE, int DIL, bool FLAT>
struct WgGeom {
    static constexpr int BM = 32 * WM, BC = 32 * CG, NT = TAPS * CG;
    static constexpr int KH = (WM == 2 && !FLAT) ? 2 : 1;       // K halves inside the workgroup (8 waves = WM x 2 tile groups x KH)
    static constexpr int TH = KP / TW;
    static constexpr int PAD = TAPS == 9 ? DIL : 0;
    static constexpr int PH = FLAT ? 1 : (TH - 1) * STRIDE + 1 + 2 * PAD;
    static constexpr int PWL = FLAT ? KP + 4 : (((TW - 1) * STRIDE + 1 + 8) + 3) & ~3;     // patch row: 4 columns of halo on the left, >= 4 on the right
    static constexpr int CS = FLAT ? KP + 4 : PH * PWL + 1;     // channel stride in LDS: odd -> 32 channels on 32 banks
    static constexpr int RS = KP + 4;                           // dY row stride: 4 * odd -> conflict-free 16-byte reads
    static constexpr int ASZ = BM * RS, BSZ = BC * CS;
    static constexpr int NG = FLAT ? 8 / WM : 2;                // n-tile groups of the 8 waves
    static constexpr int NT0 = (NT + NG - 1) / NG;              // n-tiles of a wave (the last group may own fewer)
    static constexpr int RED = KH == 2 ? 4 * NT0 * 1024 : 0;    // floats of the K-half reduction in the epilogue
    static constexpr int FLOATS = 2 * (ASZ + BSZ) > RED ? 2 * (ASZ + BSZ) : RED;
    static constexpr size_t SMEM = sizeof(float) * FLOATS;
};

// FLAT (1x1, stride 1): 8 waves = WM m-tiles x NG n-groups, each wave NTW = NT / NG n-tiles of 32 input channels, both operands
//                       read with ds_read_b128.
// PATCH: 8 waves = WM m-tiles x 2 tile groups (x 2 K halves when WM == 2); B read per pixel with ds_read_b32.
template <int TAPS, int WM, int CG, int TW, int KP, int STRIDE, int DIL, bool VEC, bool FLAT>
__global__ __launch_bounds__(kWgThreads) void k_wgrad(const WgP p) {
    using G = WgGeom<TAPS, WM, CG, TW, KP, STRIDE, DIL, FLAT>;
    constexpr int BM = G::BM, BC = G::BC, NT = G::NT, KH = G::KH, PAD = G::PAD, PH = G::PH, PWL = G::PWL, CS = G::CS, RS = G::RS;
    constexpr int ASZ = G::ASZ, BSZ = G::BSZ;
    constexpr int NG = G::NG, NT0 = G::NT0;
    // 16-byte groups staged per dY row and per input channel: four consecutive elements of the chunk, one 16-byte global load.
    // VEC: the rows of both planes are 16-byte aligned and a group is inside its row or outside.  !VEC (the 769-crop planes 385^2,
    // 193^2, 97^2, 49^2: every row of every plane starts at another alignment): 16-byte loads at 4-byte aligned addresses (same
    // rate on gfx950); the group astride the end of a run reads on into the next row / plane and has the elements past the end
    // zeroed, except where it could leave the tensor (last row of the last picture): there it is loaded element by element.
    // (Element-wise 4-byte loads, the first form: 5.7 vs 3.5 ms per step on the 1x1 layers.)
    constexpr int AL = 0;
    constexpr int ARUN = FLAT ? KP : TW, ARUNS = FLAT ? 1 : G::TH;         // runs of a dY row in one chunk
    constexpr int F4A = ARUNS * (ARUN / 4 + AL);            // groups per dY row
    constexpr int NA = (BM * F4A + kWgThreads - 1) / kWgThreads;
    constexpr int BRUN = FLAT ? KP : PWL, BRUNS = FLAT ? 1 : PH;
    constexpr int F4C = BRUNS * (BRUN / 4 + AL);            // groups per staged input channel
    constexpr int NB = (BC * F4C + kWgThreads - 1) / kWgThreads;
    static_assert(KP % 8 == 0 && (FLAT || (KP % TW == 0 && TW % 8 == 0)), "chunk");
    static_assert(KH == 1 || (KP / 2) % TW == 0, "K halves must split the chunk at a row boundary");
    static_assert(NT % NG == 0 || !FLAT, "FLAT n-tiles must divide");
    extern __shared__ __attribute__((aligned(16))) float wg_smem[];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int mtw = wave % WM;
    const int khw = KH == 2 ? (wave / WM) & 1 : 0;
    const int tg = FLAT ? wave / WM : wave >> 2;            // PATCH: waves w and w + 4 (one SIMD) own complementary tile groups

    // block -> unit (pixel range s, input-channel tile ct) x output-channel tile mt.  The mtiles workgroups of a unit stream
    // the same input patches: a unit sits on ONE XCD (blocks b and b + 8 share an XCD), units are dealt round-robin over the 8.
    const int units = p.S * p.ctiles;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int unit = (slot / p.mtiles) * 8 + xcd;
    if (unit >= units) return;
    const int mt = slot % p.mtiles;
    const int ct = unit % p.ctiles, s = unit / p.ctiles;
    const int m0 = mt * BM, c0 = ct * BC;
    const int q0 = (int)((long long)s * p.nch / p.S), q1 = (int)((long long)(s + 1) * p.nch / p.S);
    const int HW = p.H * p.W, HWo = p.Ho * p.Wo;

    v4f ra[NA], rb[NB];
    int ma[VEC ? 1 : NA], mb[VEC ? 1 : NB];         // !VEC: elements of the group that exist (0..4) | elements it was loaded early << 3

    // !VEC: the group at `src` of which the first nv (<= 0: none, >= 4: all) elements exist; `end`: one past the tensor's last
    // element -- a group that would cross it is loaded so that it ENDS there (una_fix rotates it into place; short rows, e.g. the
    // 1 x 1 map of the ASPP pooling branch, put several rows within three elements of the end).  The loaded values are not touched
    // here: any use behind the load makes hipcc wait for it in the fetch code, which serialises the prefetch.
    auto una_load = [&](const float* src, int nv, const float* end, v4f& out, int& meta) {
        nv = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
        const long long avail = end - src;
        const int early = (nv > 0 && avail < 4) ? 4 - (int)avail : 0;
        meta = nv | (early << 3);
        if (nv > 0) {
            const v4fu u = *reinterpret_cast<const v4fu*>(src - early);
            out = (v4f){u[0], u[1], u[2], u[3]};
        } else {
            out = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto una_fix = [&](v4f v, int meta) -> v4f {
        const int nv = meta & 7, early = meta >> 3;
        if (early) {                            // element i of the group is u[i + early]
            const v4f u = v;
            v[0] = early == 1 ? u[1] : (early == 2 ? u[2] : u[3]);
            v[1] = early == 1 ? u[2] : u[3];
            v[2] = u[3];
        }
        v[1] = nv > 1 ? v[1] : 0.0f;
        v[2] = nv > 2 ? v[2] : 0.0f;
        v[3] = nv > 3 ? v[3] : 0.0f;
        return v;
    };
    const float* const dy_end = p.dy + (size_t)p.N * p.Cout * HWo;
    const float* const x_end = p.x + (size_t)p.N * p.Cin * HW;

    // !VEC, FLAT: a chunk that lies inside its plane with room behind it needs no fix-up of its groups at all (every group is
    // whole and in range): only the last chunk of a plane takes the masked / rotated form (one chunk in 38 at 49 x 49: the fix
    // is ~20 VALU instructions per group in fetch + stage, 4.2 vs 3.5 ms per step on the 1x1 layers of the 769 crop with it)
    bool fix_regs = true;               // the chunk in the staging registers carries fix-up data in ma / mb
    bool fix_next = true;
    // where chunk q lives: scalars that are the same for every slot of the chunk (fetch_begin), then one 16-byte group per slot
    struct FetchCtx {
        const float* dyb;
        const float* xb;
        int k0, oy0, ox0, iy0, ix0;
        bool inner;
    };
    FetchCtx fc;
    auto fetch_begin = [&](int q) {
        fix_next = true;
        if (FLAT) {
            const int cpi = (HW + KP - 1) / KP;
            const int n = q / cpi;
            fc.k0 = (q - n * cpi) * KP;
            fc.dyb = p.dy + (size_t)n * p.Cout * HWo + fc.k0;
            fc.xb = p.x + (size_t)n * p.Cin * HW + fc.k0;
            fc.inner = !VEC && fc.k0 + KP <= HW;           // (FLAT: HW == HWo)
            fix_next = !fc.inner;
        } else {
            const int tpi = p.tiles_x * p.tiles_y;
            const int n = q / tpi, r = q - n * tpi;
            const int tyi = r / p.tiles_x, txi = r - tyi * p.tiles_x;
            fc.oy0 = tyi * G::TH; fc.ox0 = txi * TW;
            fc.iy0 = fc.oy0 * STRIDE - PAD; fc.ix0 = fc.ox0 * STRIDE - 4;
            fc.dyb = p.dy + (size_t)n * p.Cout * HWo;
            fc.xb = p.x + (size_t)n * p.Cin * HW;
        }
    };
    constexpr int GAr = ARUN / 4 + AL, GBr = BRUN / 4 + AL;       // groups per run
    auto fetch_a = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int f = tid + j * kWgThreads;
        if (FLAT) {
            const int m = f / F4A, g = f % F4A;
            const bool rowok = f < BM * F4A && m0 + m < p.Cout;
            const float* src = fc.dyb + (size_t)(m0 + m) * HWo;
            if constexpr (VEC) ra[j] = (rowok && fc.k0 + 4 * g < HW) ? *reinterpret_cast<const v4f*>(src + 4 * g) : (v4f){0.f, 0.f, 0.f, 0.f};
            else if (fc.inner) {
                if (rowok) { const v4fu u = *reinterpret_cast<const v4fu*>(src + 4 * g); ra[j] = (v4f){u[0], u[1], u[2], u[3]}; }
                else ra[j] = (v4f){0.f, 0.f, 0.f, 0.f};
            } else una_load(src + 4 * g, rowok ? HW - fc.k0 - 4 * g : 0, dy_end, ra[j], ma[VEC ? 0 : j]);
        } else {
            const int m = f / F4A, rem = f % F4A;
            const int ty = rem / GAr, g = rem % GAr;
            const int oy = fc.oy0 + ty;
            const bool rowok = f < BM * F4A && m0 + m < p.Cout && oy < p.Ho;
            const float* src = fc.dyb + (size_t)(m0 + m) * HWo + oy * p.Wo + fc.ox0;
            if constexpr (VEC) ra[j] = (rowok && fc.ox0 + 4 * g < p.Wo) ? *reinterpret_cast<const v4f*>(src + 4 * g) : (v4f){0.f, 0.f, 0.f, 0.f};
            else una_load(src + 4 * g, rowok ? p.Wo - fc.ox0 - 4 * g : 0, dy_end, ra[j], ma[VEC ? 0 : j]);
        }
    };
    auto fetch_b = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int f = tid + j * kWgThreads;
        if (FLAT) {
            const int c = f / F4C, g = f % F4C;
            const bool rowok = f < BC * F4C && c0 + c < p.Cin;
            const float* src = fc.xb + (size_t)(c0 + c) * HW;
            if constexpr (VEC) rb[j] = (rowok && fc.k0 + 4 * g < HW) ? *reinterpret_cast<const v4f*>(src + 4 * g) : (v4f){0.f, 0.f, 0.f, 0.f};
            else if (fc.inner) {
                if (rowok) { const v4fu u = *reinterpret_cast<const v4fu*>(src + 4 * g); rb[j] = (v4f){u[0], u[1], u[2], u[3]}; }
                else rb[j] = (v4f){0.f, 0.f, 0.f, 0.f};
            } else una_load(src + 4 * g, rowok ? HW - fc.k0 - 4 * g : 0, x_end, rb[j], mb[VEC ? 0 : j]);
        } else {
            const int c = f / F4C, rem = f % F4C;
            const int row = rem / GBr, g = rem % GBr;
            const int iy = fc.iy0 + row;
            const bool rowok = f < BC * F4C && c0 + c < p.Cin && (unsigned)iy < (unsigned)p.H;
            const float* src = fc.xb + ((long long)(c0 + c) * HW + (long long)iy * p.W + fc.ix0);
            if constexpr (VEC) rb[j] = (rowok && (unsigned)(fc.ix0 + 4 * g) < (unsigned)p.W) ? *reinterpret_cast<const v4f*>(src + 4 * g) : (v4f){0.f, 0.f, 0.f, 0.f};
            else una_load(src + 4 * g, (rowok && fc.ix0 + 4 * g >= 0) ? p.W - fc.ix0 - 4 * g : 0, x_end, rb[j], mb[VEC ? 0 : j]);
        }
    };
    auto fetch = [&](int q) {
        fetch_begin(q);
        wg_static_for<0, NA>([&](auto jc) { fetch_a(jc); });
        wg_static_for<0, NB>([&](auto jc) { fetch_b(jc); });
    };

    auto stage_a = [&](int buf, auto jc) {
        constexpr int j = decltype(jc)::value;
        float* sA = wg_smem + buf * (ASZ + BSZ);
        const int f = tid + j * kWgThreads;
        if (f < BM * F4A) {
            float* run = sA + (f / F4A) * RS + ((f % F4A) / GAr) * ARUN;
            *reinterpret_cast<v4f*>(run + ((f % F4A) % GAr) * 4) = (VEC || !fix_regs) ? ra[j] : una_fix(ra[j], ma[VEC ? 0 : j]);
        }
    };
    auto stage_b = [&](int buf, auto jc) {
        constexpr int j = decltype(jc)::value;
        float* sB = wg_smem + buf * (ASZ + BSZ) + ASZ;
        const int f = tid + j * kWgThreads;
        if (f < BC * F4C) {
            float* run = sB + (f / F4C) * CS + ((f % F4C) / GBr) * BRUN;
            const v4f vb = (VEC || !fix_regs) ? rb[j] : una_fix(rb[j], mb[VEC ? 0 : j]);
            if constexpr (FLAT) {
                *reinterpret_cast<v4f*>(run + ((f % F4C) % GBr) * 4) = vb;
            } else {
                float* dst = run + ((f % F4C) % GBr) * 4;            // channel stride is odd: four 4-byte stores
                dst[0] = vb[0]; dst[1] = vb[1]; dst[2] = vb[2]; dst[3] = vb[3];
            }
        }
    };
    auto stage = [&](int buf) {
        wg_static_for<0, NA>([&](auto jc) { stage_a(buf, jc); });
        wg_static_for<0, NB>([&](auto jc) { stage_b(buf, jc); });
    };

    f32x16 acc[NT0];
#pragma unroll
    for (int t = 0; t < NT0; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // one chunk of MFMAs for the wave's n-tiles [T0, T0 + TN)
    const int aoff = (mtw * 32 + l31) * RS + 4 * h + khw * (KP / 2);
    const int boff = FLAT ? (tg * NT0 * 32 + l31) * CS + 4 * h : l31 * CS + 4 * h * STRIDE + khw * ((KP / 2) / TW) * STRIDE * PWL;
    constexpr int JN = KP / 8 / KH;                 // groups of four MFMA k-steps per chunk and wave
    // The LDS operands of a group of four k-steps are requested one group ahead of the MFMAs that consume them (sched_barrier
    // pins the order): the two waves of a SIMD run the same phase, nothing else hides an LDS round trip.
    auto mfma_chunk = [&](int buf, auto T0c, auto TNc, auto JLc, auto JHc, auto&& hook) {
        constexpr int T0 = decltype(T0c)::value, TN = decltype(TNc)::value, jlo = decltype(JLc)::value, jhi = decltype(JHc)::value;
        const float* sA = wg_smem + buf * (ASZ + BSZ);
        const float* sB = sA + ASZ;
        struct Group {
            v4f a;
            float b[4][TN];
        };
        auto load_group = [&](int jl, Group& o) {
            o.a = *reinterpret_cast<const v4f*>(sA + aoff + 8 * jl);
            if constexpr (FLAT) {
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    const v4f bv = *reinterpret_cast<const v4f*>(sB + boff + (T0 + t) * 32 * CS + 8 * jl);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o.b[i][t] = bv[i];
                }
            } else {
                const int ty = (8 * jl) / TW, txb = (8 * jl) % TW;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        const int cg = (T0 + t) / TAPS, tap = (T0 + t) % TAPS;
                        const int ky = TAPS == 9 ? tap / 3 : 0, kx = TAPS == 9 ? tap % 3 : 0;
                        o.b[i][t] = sB[boff + cg * 32 * CS + (ty * STRIDE + ky * DIL) * PWL + (txb + i) * STRIDE + kx * DIL - PAD + 4];
                    }
            }
        };
        if constexpr (jlo < jhi) {
            Group cur, nxt;
            load_group(jlo, cur);
            wg_static_for<jlo, jhi>([&](auto jlc) {
                constexpr int jl = decltype(jlc)::value;
                if (jl + 1 < jhi) load_group(jl + 1, nxt);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[i], cur.b[i][t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                hook(jlc);                              // staging work that rides in the shadow of this group's MFMAs
                __builtin_amdgcn_sched_barrier(0);
                if (jl + 1 < jhi) cur = nxt;
            });
        }
    };
    auto mfma = [&](int buf, auto jlo, auto jhi, auto&& hook) {
        if constexpr (FLAT) {           // the wave's tile group is an address offset (boff)
            mfma_chunk(buf, std::integral_constant<int, 0>{}, std::integral_constant<int, NT0>{}, jlo, jhi, hook);
        } else {                        // tap subsets differ per group: two instruction streams, chosen per wave
            if (tg == 0) mfma_chunk(buf, std::integral_constant<int, 0>{}, std::integral_constant<int, NT0>{}, jlo, jhi, hook);
            else if constexpr (NT - NT0 > 0) mfma_chunk(buf, std::integral_constant<int, NT0>{}, std::integral_constant<int, NT - NT0>{}, jlo, jhi, hook);
            else wg_static_for<decltype(jlo)::value, decltype(jhi)::value>(hook);     // (a wave without tiles still stages its slots)
        }
    };

    // The registers that carried chunk q + 1 to LDS are refilled with chunk q + 2 right behind that store: a load has a whole
    // iteration to arrive.  As in k_conv_sk, this work rides slot by slot behind the wave's OWN MFMA groups (the first form ran it as
    // one burst after three quarters of the chunk's MFMAs: while one wave of a SIMD issues MFMAs back to back its partner's
    // instructions are starved, so a burst of ~8 LDS stores + 8 loads + their address arithmetic stretched over the partner's whole
    // multiply phase -- tools/sk_phases.py measured that on the forward kernel).  The loop body has no branch: past the last chunk
    // the stores go to the buffer nobody reads any more and the loads re-read the last chunk (an L2 hit).
    int buf = 0;
    fetch(q0);
    fix_regs = fix_next;
    stage(0);
    if (q0 + 1 < q1) {
        fetch(q0 + 1);
        fix_regs = fix_next;
    }
    __syncthreads();
    constexpr int NS = NA + NB;
    for (int q = q0; q < q1; ++q) {
        fetch_begin(q + 2 < q1 ? q + 2 : q1 - 1);
        const bool fix2 = fix_next;
        mfma(buf, std::integral_constant<int, 0>{}, std::integral_constant<int, JN>{}, [&](auto jlc) {
            constexpr int jl = decltype(jlc)::value;
            constexpr int s0 = jl * NS / JN, s1 = (jl + 1) * NS / JN;
            wg_static_for<s0, s1>([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                if constexpr (S < NA) {
                    stage_a(buf ^ 1, std::integral_constant<int, S>{});
                    fetch_a(std::integral_constant<int, S>{});
                } else {
                    stage_b(buf ^ 1, std::integral_constant<int, S - NA>{});
                    fetch_b(std::integral_constant<int, S - NA>{});
                }
            });
        });
        fix_regs = fix2;
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: K halves added through LDS, then the partial tile -> part[s][m][tap][c] ----------------------------------
    if constexpr (KH == 2) {
        float* red = wg_smem;                   // [4 waves][NT0][16][64]: all chunk reads are behind the last barrier
        if (khw == 1) {
#pragma unroll
            for (int t = 0; t < NT0; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((wave & 1) + 2 * tg) * NT0 + t) * 1024 + r * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (khw == 1) return;
#pragma unroll
        for (int t = 0; t < NT0; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += red[(((wave & 1) + 2 * tg) * NT0 + t) * 1024 + r * 64 + lane];
    }
    float* pb = p.part + (size_t)s * p.Cout * TAPS * p.Cin;
    const int tbase = tg * NT0;
#pragma unroll
    for (int t = 0; t < NT0; ++t) {
        const int tt = tbase + t;
        if (tt < NT) {
            const int cg = tt / TAPS, tap = tt % TAPS;
            const int c = c0 + cg * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mtw * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.Cout && c < p.Cin) pb[((size_t)m * TAPS + tap) * p.Cin + c] = acc[t][r];
            }
        }
    }
}

// dW[m][c][tap] = sum_s part[s][m][tap][c]: threads walk the partial layout (coalesced reads), four loads in flight per thread;
// the order of the additions is fixed: ((s = 0, 4, 8, ..) + (1, 5, ..)) + ((2, 6, ..) + (3, 7, ..))
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, int S, int Cout, int Cin, int taps, float* __restrict__ dw) {
    const size_t n = (size_t)Cout * taps * Cin;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
    int s = 0;
    for (; s + 4 <= S; s += 4) {
        const float a = part[(size_t)s * n + i], b = part[(size_t)(s + 1) * n + i], c = part[(size_t)(s + 2) * n + i],
                    d = part[(size_t)(s + 3) * n + i];
        v0 += a; v1 += b; v2 += c; v3 += d;
    }
    if (s < S) v0 += part[(size_t)s * n + i];
    if (s + 1 < S) v1 += part[(size_t)(s + 1) * n + i];
    if (s + 2 < S) v2 += part[(size_t)(s + 2) * n + i];
    const float v = (v0 + v1) + (v2 + v3);
    const int c = (int)(i % Cin);
    const size_t mt = i / Cin;
    const int tap = (int)(mt % taps);
    const size_t m = mt / taps;
    dw[(m * Cin + c) * taps + tap] = v;
}

// S pixel ranges.  One workgroup fills a CU, so the launch should be k * 256 equal workgroups; every workgroup pays about one
// chunk time of prologue / epilogue, and the S partial slices are written and re-read once (bounded by kPartCap).
constexpr size_t kPartCap = (size_t)64 << 20;
inline int wg_pick_split(int tiles, int nch, size_t slice_bytes, double flop_per_chunk) {
    int best = 1;
    double best_score = -1e30;
    const int smax = nch < 1024 ? nch : 1024;
    for (int S = 1; S <= smax; ++S) {
        if (S > 1 && (size_t)S * slice_bytes > kPartCap) break;
        const long long wg = (long long)tiles * S;
        const long long rounds = (wg + 255) / 256;
        const double fill = (double)wg / (double)(rounds * 256);
        const double per = (double)nch / S;                     // chunks per workgroup
        // the slices are written once and read once at ~3 TB/s next to ~100 TFLOP/s of products
        const double extra = (double)S * slice_bytes * 2.0 / 3e12;
        const double work = (double)nch * flop_per_chunk / 1e14 / (fill * per / (per + 1.5));
        const double score = -(work + extra);
        if (score > best_score) {
            best_score = score;
            best = S;
        }
    }
    return best;
}

template <int TAPS, int WM, int CG, int TW, int KP, int STRIDE, int DIL, bool VEC, bool FLAT>
int wg_launch(WgP p, hipStream_t st) {
    using G = WgGeom<TAPS, WM, CG, TW, KP, STRIDE, DIL, FLAT>;
    auto kern = &k_wgrad<TAPS, WM, CG, TW, KP, STRIDE, DIL, VEC, FLAT>;
    static_assert(G::SMEM <= 160 * 1024, "LDS");
    if (G::SMEM > 64 * 1024) {
        static bool raised[64] = {};                    // per device: the attribute belongs to the device's copy of the code object
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        if (dev < 0 || dev >= 64 || !raised[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0 && dev < 64) raised[dev] = true;
        }
    }
    const long long nblk = 8LL * (((long long)p.S * p.ctiles + 7) / 8) * p.mtiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(kWgThreads), G::SMEM, st, p);
    return mas_launch_status();
}

struct WgShape {
    int BM, BC, KP, TW, flat;
};

inline bool wg_shape(int Cout, int ksize, int stride, int dil, int Wo, WgShape* g) {
    g->flat = ksize == 1 && stride == 1;
    g->BM = Cout > 64 ? 128 : 64;
    g->TW = Wo >= 32 && !(Wo % 32 != 0 && Wo % 16 == 0) ? 32 : 16;       // 48-wide planes: three exact 16-wide tiles
    if (g->flat) {
        g->BC = 128;
        g->KP = 64;
    } else if (stride == 2) {
        g->BC = ksize == 3 ? 32 : 64;
        g->KP = 32;
        g->TW = 16;
    } else {
        g->BC = 32;
        g->KP = 64;
        if (dil == 4) g->TW = 16;
    }
    return true;
}
}  // namespace

extern "C" size_t mas_conv_wgrad_workspace_bytes(int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil) {
    (void)dil;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return 0;
    // the split count never exceeds what wg_pick_split's cap admits; one slice always fits
    const size_t slice = sizeof(float) * (size_t)Cout * Cin * ksize * ksize;
    size_t smax = kPartCap / slice;
    if (smax < 1) smax = 1;
    if (smax > 1024) smax = 1024;
    return smax * slice;
}

extern "C" int mas_conv_wgrad_plan(int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil, int* out6) {
    if (!out6) return MAS_ERR_NULL;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return MAS_ERR_SHAPE;
    WgShape g;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    wg_shape(Cout, ksize, stride, dil, Wo, &g);
    const int mtiles = (Cout + g.BM - 1) / g.BM, ctiles = (Cin + g.BC - 1) / g.BC;
    const int nch = g.flat ? N * ((H * W + g.KP - 1) / g.KP) : N * ((Wo + g.TW - 1) / g.TW) * ((Ho + g.KP / g.TW - 1) / (g.KP / g.TW));
    const int S = wg_pick_split(mtiles * ctiles, nch, sizeof(float) * (size_t)Cout * Cin * ksize * ksize,
                                2.0 * mtiles * g.BM * ctiles * g.BC * ksize * ksize * g.KP);
    out6[0] = g.BM; out6[1] = g.BC; out6[2] = g.KP; out6[3] = nch; out6[4] = S; out6[5] = mtiles * ctiles * S;
    return 0;
}

extern "C" int mas_conv_wgrad(const float* x, const float* dy, int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil,
                              float* dw, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || !workspace) return MAS_ERR_NULL;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return MAS_ERR_SHAPE;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return MAS_ERR_RANGE;
    if (ksize == 1 && dil != 1) return MAS_ERR_RANGE;
    if (ksize == 3 && !(dil == 1 || (stride == 1 && (dil == 2 || dil == 4)))) return MAS_ERR_RANGE;
    if ((long long)Cin * H * W > 0x7fffffffLL || (long long)Cout * H * W > 0x7fffffffLL) return MAS_ERR_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    WgP p;
    p.x = x; p.dy = dy; p.part = static_cast<float*>(workspace);
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.Ho = (H - 1) / stride + 1;
    p.Wo = (W - 1) / stride + 1;
    WgShape g;
    wg_shape(Cout, ksize, stride, dil, p.Wo, &g);
    const int taps = ksize * ksize;
    p.mtiles = (Cout + g.BM - 1) / g.BM;
    p.ctiles = (Cin + g.BC - 1) / g.BC;
    if (g.flat) {
        p.tiles_x = p.tiles_y = 0;
        p.nch = N * ((H * W + g.KP - 1) / g.KP);
    } else {
        p.tiles_x = (p.Wo + g.TW - 1) / g.TW;
        p.tiles_y = (p.Ho + g.KP / g.TW - 1) / (g.KP / g.TW);
        p.nch = N * p.tiles_x * p.tiles_y;
    }
    const size_t slice = sizeof(float) * (size_t)Cout * Cin * taps;
    p.S = wg_pick_split(p.mtiles * p.ctiles, p.nch, slice, 2.0 * p.mtiles * g.BM * p.ctiles * g.BC * taps * g.KP);
    while (p.S > 1 && (size_t)p.S * slice > workspace_bytes) --p.S;
    if ((size_t)p.S * slice > workspace_bytes) return MAS_ERR_WORKSPACE;
    const bool al = ((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 16 == 0);
    const bool vec = al && (g.flat ? (H * W) % 4 == 0 : (W % 4 == 0 && p.Wo % 4 == 0));
    // (the unaligned path loads the last partial group of a tensor so that it ends at the tensor's end: four elements at least)
    if (!vec && ((long long)N * Cin * H * W < 4 || (long long)N * Cout * p.Ho * p.Wo < 4)) return MAS_ERR_SHAPE;
    int rc = MAS_ERR_SHAPE;
    const bool big = g.BM == 128;
#define WG_GO(TAPS, CG, TW, KP, STRIDE, DIL, FLAT)                                                                            \
    rc = big ? (vec ? wg_launch<TAPS, 4, CG, TW, KP, STRIDE, DIL, true, FLAT>(p, st) : wg_launch<TAPS, 4, CG, TW, KP, STRIDE, DIL, false, FLAT>(p, st)) \
             : (vec ? wg_launch<TAPS, 2, CG, TW, KP, STRIDE, DIL, true, FLAT>(p, st) : wg_launch<TAPS, 2, CG, TW, KP, STRIDE, DIL, false, FLAT>(p, st))
    if (g.flat) {
        WG_GO(1, 4, 32, 64, 1, 1, true);
    } else if (ksize == 1) {
        WG_GO(1, 2, 16, 32, 2, 1, false);
    } else if (stride == 2) {
        WG_GO(9, 1, 16, 32, 2, 1, false);
    } else if (dil == 1) {
        if (g.TW == 32) { WG_GO(9, 1, 32, 64, 1, 1, false); } else { WG_GO(9, 1, 16, 64, 1, 1, false); }
    } else if (dil == 2) {
        if (g.TW == 32) { WG_GO(9, 1, 32, 64, 1, 2, false); } else { WG_GO(9, 1, 16, 64, 1, 2, false); }
    } else {
        WG_GO(9, 1, 16, 64, 1, 4, false);          // the 32-wide patch of dilation 4 does not fit two LDS buffers
    }
#undef WG_GO
    if (rc != 0) return rc;
    const size_t n = (size_t)Cout * taps * Cin;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p.part, p.S, Cout, Cin, taps, dw);
    return mas_launch_status();
}
