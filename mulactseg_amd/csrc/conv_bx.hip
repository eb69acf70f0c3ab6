// conv_bx.hip -- the dense convolutions of the network (1x1 stride 1 / 2; 3x3 stride 1, dilation 1 / 2; inference with the
// BatchNorm / residual / ReLU epilogue, and in training the bare forward product and -- with the weight image of role 1 -- the
// stride-1 input gradient) as an implicit GEMM on the bf16 matrix cores of gfx950 with f32 OPERANDS AND f32 RESULTS: every f32
// operand is split EXACTLY into three bf16 terms (bx_split.h)
//        x = h + m + l,    h = bf16(x),  m = bf16(x - h),  l = x - h - m        (round to nearest even; 8 + 8 + 8 significand
//                                                                                bits, both subtractions exact in f32)
// and a product a*b is accumulated as the six partial products of order <= 2
//        ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm                          (each exact in f32: 8 x 8 bits)
// on v_mfma_f32_32x32x16_bf16 with f32 accumulation.  What is dropped (am*bl + al*bm + al*bl) is at most 2^-23 |a*b|, of either
// sign -- the size of the rounding the f32 MFMA (v_mfma_f32_32x32x2_f32, csrc/conv_mfma.hip) commits per product -- so the
// result is f32 arithmetic by its error bound, exact on integer data, and six bf16 MFMAs of 16x the f32 rate do the work of
// sixteen f32 ones: 2.67x the f32 matrix peak (157 -> 419 TFLOP/s of f32 convolution).
// Reference: the convolutions of models/segmentation/backbone/resnet.py:129-160 (Bottleneck conv1/2/3 + downsample with
// their BatchNorm / residual / ReLU), the deep stem (:163-171), the 1x1 projections of deeplabv3.py:85-137,216-245 -- the
// model forward of the acquisition round, active_selection/my_bvsb_predclsbal_pwr_banignore.py:35-72.
//
// GEMM view per picture:  Y[m, p] = sum_{tap, c} W[m, (tap, c)] * X[c, pixel p shifted by tap]
//   A = weights, split ONCE per checkpoint by k_bx_pack into ready-made LDS images [M tile][chunk][term][k group][BM][8 bf16]
//       (a k group = 8 consecutive k values = one 16-byte MFMA fragment of a lane),
//   B = activations, f32 NCHW in HBM, split by the VALU between the global load and the LDS store into the image
//       [term][k group][pixel position][8 bf16]: a lane reads ITS pixel's 16-byte fragment, a tap is an address offset.
//   1x1: a chunk is 32 channels (4 k groups), a tile 128 (64) channels x 128 (256) consecutive pixels of the flattened plane;
//   3x3: a chunk is 8 channels x 9 taps (+ one zero-weight tap: a 16-k MFMA step is two taps), a tile 64 channels x 8 x 32
//        pixels over an (8 + 2 dil) x (32 + 2 dil) input patch that is staged once per chunk for all taps.
// A workgroup is 4 waves, a wave owns 64 x 64 of the tile (2 x 2 MFMA tiles: 64 accumulator registers); chunk t + 1 travels
// global -> registers in front of the MFMAs of chunk t and registers -> (split) -> LDS behind them; two workgroups per CU
// cover each other's staging.  Epilogue as conv_mfma.hip: y * scale[m] + shift[m] (+ residual) (ReLU), NCHW stores.
// 3x3 stride 2 (layer2.0.conv2 / layer3.0.conv2): nine shifted 1x1 stride-2 products into one accumulator set -- the chunks run
// over (tap, 32 channels), the tap is a byte offset of the chunk's global loads (no patch in LDS: a tile's input patch at stride
// 2 is four times its output); weight image of role 2 (tap-major chunks).
// Dual form (mas_conv_bx_fwd_dual, 1x1 stride 1): the chunks of a second input with its own weight image follow those of the
// first into the same accumulators -- conv3 + stride-1 downsample of a Bottleneck with both BatchNorm scales folded into the
// weight rows at pack time (row_scale).
#include <type_traits>

#include "common.h"
#include "bx_split.h"

namespace {
constexpr int kThreads = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

constexpr int kBxPPA = 448;                     // 3x3: positions of one term of the B image: the 12 x 36 patch of dilation 2 (10 x 34 at dilation 1)
constexpr int kBxDump = 432;                    //      + 16 positions where the staging tasks beyond the patch put their (unused) stores
constexpr int kBxFlatPP = 608;                  // 3x3, flat tiles (TW = 0): patch capacity -- the FULL rows a run of 256 consecutive pixels touches, + halo
constexpr int kBxFlatPPA = kBxFlatPP + 16;      //      + the dump positions
constexpr int kBxTaps3 = 10;                    // 3x3: nine taps + one zero-weight tap (five 16-k steps per 8-channel chunk)

struct BxP {
    const float* x;
    const v4f* wp;
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    const float* x2;                            // dual form (1x1 stride 1): a second input on the same plane, walked behind the first
    const v4f* wp2;                             //   with its own weight image: y = W1 x + W2 x2 (conv3 + downsample of a Bottleneck)
    int Cin2;                                   //   0: none
    int Cin, H, W, Cout, Ho, Wo, dil, relu;
    int tiles_x, tiles_y, ptiles, mtiles;
    int N;
    double2* stats;                             // training forward (bare, ksplit 1): per-(row, pixel tile x wave column) sums (sum y, sum y^2) of the stored
    int stat_slots;                             //   outputs: [Cout][stat_slots] -- the BatchNorm partial sums, formed in the epilogue (NULL: none)
    const v4f* x3;                              // PRE: the input as a bx3 tensor [N][ceil(Cin/8)][3 terms][H*W][8 bf16] (k_bx3_split / an OUT3 epilogue)
    int t9;                                     // stride-2 form: 1 = the 3x3 convolution as nine shifted 1x1 products (chunks = tap x 32 channels)
    int ksplit;                                 // split-K (bare products only): the chunks of a tile are dealt to `ksplit` workgroups; part 0
    float* part;                                //   stores into y (with the residual), part k > 0 into part + (k - 1) * N * Cout * Ho * Wo;
                                                //   k_bx_reduce then adds the parts into y in index order
#ifdef BX_STAMPS
    unsigned long long* stamps;
#endif
};

// k groups of a chunk's A image / B image
template <int TAPS> struct BxGeo {
    static constexpr int CK = TAPS == 1 ? 32 : 8;
    static constexpr int GA = TAPS == 1 ? 4 : kBxTaps3;
    static constexpr int SLABS = GA / 2;
};

// position (16-byte unit) of pixel p of a 1x1 tile inside one k group of the B image: the four pixels of a thread's 16-byte
// global load go to four rows of 36 units, so that both the staging stores (lanes = consecutive pixel quads) and the MFMA
// fragment reads (lanes = consecutive pixels, serviced in the 16-lane groups of ds_read_b128) are free of bank conflicts
__device__ __forceinline__ int bx_pos1(int p) { return (p >> 7) * 144 + (p & 3) * 36 + ((p & 127) >> 2); }

// ---- BatchNorm partial sums in the epilogue: cross-lane halving ------------------------------------------------------------
// v[0 .. n) in every lane of a 32-lane half -> v[0 .. n / 2): the lanes whose bit `BIT` is clear keep the sums (over the lane pair
// l, partner(l)) of entries 0 .. n/2 - 1, the others those of entries n/2 .. n - 1.  BIT 4: v_permlane16_swap (the odd 16-lane row
// of the first register against the even row of the second: one swap + one add per pair of entries); BIT 3 .. 0: DPP adds
// (row_ror:8, row_half_mirror, quad_perm) + one select.  After the five steps lane l holds the entries whose index (in units of
// the final n) equals l's five bits.
template <int BIT>
__device__ __forceinline__ float bx_lane_pair_sum(float v) {
    constexpr int ctrl = BIT == 3 ? 0x128 : (BIT == 2 ? 0x141 : (BIT == 1 ? 0x4E : 0xB1));      // row_ror:8 | row_half_mirror | quad_perm [2,3,0,1] | [1,0,3,2]
    return v + __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), ctrl, 0xf, 0xf, false));
}
template <int BIT, int N>
__device__ __forceinline__ void bx_halve(float (&v)[N], int lane) {
    typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
    if (BIT == 4) {
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const v2u_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + N / 2]), false, false);
            v[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
        }
    } else {
        const bool up = (lane >> BIT) & 1;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const float lo = bx_lane_pair_sum<BIT>(v[i]), hi = bx_lane_pair_sum<BIT>(v[i + N / 2]);
            v[i] = up ? hi : lo;
        }
    }
}

// ---- weight image --------------------------------------------------------------------------------------------------------
// One job = one weight tensor in one role.  role 0: the forward product (M = Cout rows, K = Cin); role 1: the input gradient
// of the stride-1 convolution, which is the same kernel on dY with the weight's channel axes swapped and the taps mirrored
// (M = Cin rows, K = Cout):  a[m][k = (tap, c)] = w[c][m][8 - tap].
// One thread per (M tile, chunk, k group, row): it reads its eight weights once and writes the three 16-byte units (terms h | m | l) of
// [M tile][chunk][term][k group][row][8]; k = 8 g + j: 1x1 channel = chunk * 32 + k,
// 3x3 tap = g (tap 9: zeros), channel = chunk * 8 + j; rows beyond M and channels beyond K are zeros.
struct BxPackJob {
    const float* w;             // [Cout][Cin][taps] as PyTorch stores it
    const float* row_scale;     // NULL, or [Cout]: the image holds w[m, :, :] * row_scale[m] (role 0: a BatchNorm scale folded into the weight)
    unsigned* out;
    int Cout, Cin, taps, role, BM;
    int t9;                     // role 2: the 3x3 weight as 9 x (Cin / 32) chunks of the 1x1 form, tap-major (the strided 3x3 forward)
    long long units;            // threads of the job (16-byte units of the image / 3)
    unsigned first_block;       // (multi-job launch) the job's first 256-thread block
};

__device__ __forceinline__ void bx_pack_unit(const BxPackJob& jb, long long u) {
    const int taps = jb.taps, BM = jb.BM;
    if (jb.t9) {                                    // [M tile][chunk = tap * (Cin / 32) + cc][term][k group 0..3][row][8]: channel = cc * 32 + 8 g + j
        const int row = (int)(u % BM);
        long long r = u / BM;
        const int g = (int)(r % 4);
        r /= 4;
        const int cpt = jb.Cin / 32, nch = 9 * cpt;
        const int chunk = (int)(r % nch), mt = (int)(r / nch);
        const int tap = chunk / cpt, cc = chunk - tap * cpt;
        const int m = mt * BM + row;
        unsigned o[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v0 = 0.0f, v1 = 0.0f;
            if (m < jb.Cout) {
                const size_t base = ((size_t)m * jb.Cin + cc * 32 + 8 * g + 2 * j) * 9 + tap;
                v0 = jb.w[base];
                v1 = jb.w[base + 9];
                if (jb.row_scale) { v0 *= jb.row_scale[m]; v1 *= jb.row_scale[m]; }
            }
            bx_split2(v0, v1, o[0][j], o[1][j], o[2][j]);
        }
#pragma unroll
        for (int term = 0; term < 3; ++term)
            *reinterpret_cast<uint4*>(jb.out + 4 * (((((long long)mt * nch + chunk) * 3 + term) * 4 + g) * BM + row)) =
                make_uint4(o[term][0], o[term][1], o[term][2], o[term][3]);
        return;
    }
    const int GA = taps == 1 ? 4 : kBxTaps3, CK = taps == 1 ? 32 : 8;
    const int M = jb.role ? jb.Cin : jb.Cout, K = jb.role ? jb.Cout : jb.Cin;
    const int row = (int)(u % BM);
    long long r = u / BM;
    const int g = (int)(r % GA);
    r /= GA;
    const int nch = (K + CK - 1) / CK;              // (a last, partial chunk carries zero weights for the channels that do not exist)
    const int chunk = (int)(r % nch);
    const int mt = (int)(r / nch);
    const int m = mt * BM + row;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float val = 0.0f;
        const int c = chunk * CK + (taps == 1 ? 8 * g + j : j);
        if (m < M && c < K && g < (taps == 1 ? GA : 9)) {
            const int tap = taps == 1 ? 0 : (jb.role ? 8 - g : g);
            const size_t co = jb.role ? c : m, ci = jb.role ? m : c;
            val = jb.w[(co * jb.Cin + ci) * taps + tap];
            if (jb.row_scale) val *= jb.row_scale[co];
        }
        v[j] = val;
    }
    unsigned o[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bx_split2(v[2 * j], v[2 * j + 1], o[0][j], o[1][j], o[2][j]);
#pragma unroll
    for (int term = 0; term < 3; ++term)
        *reinterpret_cast<uint4*>(jb.out + 4 * (((((long long)mt * nch + chunk) * 3 + term) * GA + g) * BM + row)) =
            make_uint4(o[term][0], o[term][1], o[term][2], o[term][3]);
}

__global__ void k_bx_pack(const BxPackJob jb) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < jb.units) bx_pack_unit(jb, u);
}

// all weights of a model in all their roles in ONE launch (after every optimizer step in training)
__global__ void k_bx_pack_multi(const BxPackJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;                     // the last job whose first block is <= this block (wave-uniform search)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const BxPackJob jb = jobs[lo];
    const long long u = (long long)(blockIdx.x - jb.first_block) * blockDim.x + threadIdx.x;
    if (u < jb.units) bx_pack_unit(jb, u);
}

// -DBX_STAMPS: a measurement build (tools/bx_stamps.py) -- wave 0 of every workgroup sums the cycles (s_memtime) it spends in each
// phase of the loop and writes them to the buffer registered with mas_conv_bx_debug_stamps
#ifdef BX_STAMPS
static unsigned long long* g_bx_stamps = nullptr;
#define BX_T(i) do { if (stamp) { const long long now_ = clock64(); acc_t[i] += now_ - last_t; last_t = now_; } } while (0)
#else
#define BX_T(i) do { } while (0)
#endif

// Buffer resources (raw, stride 0): loads / stores at `voffset` beyond the size read zeros / are dropped -- pixels outside the
// plane, channels beyond Cout and the tail of a partial weight load need no predicates, selects or 64-bit address arithmetic;
// the scalar offset (not range-checked) carries what is uniform (channel a of a task, piece j of the weight image).
constexpr unsigned kBxRsrcFlags = 0x00020000;
constexpr int kBxOut = (int)0x80000000u;        // a byte offset beyond every resource used here (sizes are checked < 2^31 by the host)

// ---- the convolution -------------------------------------------------------------------------------------------------------
// What was measured and lost on this kernel (tools/bx_ab.sh: library builds A/B in one GPU session, per-layer table of the pool
// forward): the A image by LDS-DMA into two buffers (no staging registers, no ds_write: 2-20 % SLOWER on the layers with many M
// tiles); two chunks in flight with two register sets (+5 %: the loop is not waiting for memory); three workgroups per CU for
// the 1x1 form (+2 %) and, at 168 registers, for the 3x3 form (+-0.5 %); issue priority raised for the staging phase or for the MFMA
// phase (+-1 %).  In-kernel stamps
// (tools/bx_stamps.py) show why: while the SIMD partner (a wave of the CU's other workgroup) streams MFMAs, a wave gets about one
// instruction issued per MFMA whatever its kind, so the staging + fetch phases cost by their instruction COUNT -- hence the
// buffer-resource forms below (no address arithmetic, no bounds selects).
// V: the stride of the 1x1 form (1 | 2), the dilation of the 3x3 form (1 | 2: the patch geometry is a compile-time constant)
// TW: 3x3 form: the 256 output pixels of a tile are 8 rows x 32 columns (TW = 32) or 16 x 16 (TW = 16: the 48 x 48 planes of
//     layer3 / layer4 at the training crop are 9 such tiles, 12 of the wide ones of which a quarter is padding) or, TW = 0 ("flat"),
//     256 CONSECUTIVE pixels of the plane in row-major order over a patch of the full rows they touch -- the 49 x 49 planes of the
//     769 crop are 10 such tiles against 14 wide / 16 square ones (half of whose pixels are padding); 1x1: 32
// PRE: the activations arrive ALREADY SPLIT (bx3 layout: 16-byte units of 8 channels of one pixel and one term -- exactly a unit of
//      the B image), written by the producer of the tensor; staging is then a 16-byte copy per unit, no VALU work
template <int TAPS, int BM, int BN, int V, bool RES, int TW = 32, bool PRE = false>
__global__ __launch_bounds__(kThreads, 2) void k_conv_bx(const BxP p) {
    constexpr bool S2 = TAPS == 1 && V == 2;
    static_assert(!(PRE && S2), "presplit input: stride 1");
    constexpr int DIL = TAPS == 9 ? V : 1;
    constexpr bool FLAT = TW == 0;
    constexpr int TH = FLAT ? 1 : 256 / (FLAT ? 1 : TW);            // 3x3: rows of a tile
    static_assert(TW == 32 || ((TW == 16 || TW == 0) && TAPS == 9 && !PRE), "tile shape");
    constexpr int PW = TW + 2 * DIL;                                // 3x3: columns of the input patch of a TH x TW tile (flat: p.W + 2 DIL, a run-time value)
    constexpr int PP = FLAT ? kBxFlatPP : (TH + 2 * DIL) * PW;      //      pixels of the patch (flat: its capacity)
    constexpr int PPA = FLAT ? kBxFlatPPA : kBxPPA, DUMP = FLAT ? kBxFlatPP : kBxDump;
    static_assert(PP <= DUMP, "patch");
    constexpr int CK = BxGeo<TAPS>::CK, GA = BxGeo<TAPS>::GA, SLABS = BxGeo<TAPS>::SLABS;
    constexpr int WM = BM / 64, WN = 4 / WM;
    static_assert(WN * 64 == BN, "a wave owns 64 x 64");
    constexpr int AUNITS = 3 * GA * BM;                         // 16-byte units of a chunk's A image
    constexpr int NW = (AUNITS + kThreads - 1) / kThreads;
    constexpr bool WTAIL = AUNITS % kThreads != 0;              // the last weight load of a thread may lie beyond the image
    constexpr int POS1 = PRE ? BN : 144 * (BN / 128);           // 1x1: units per k group of the B image (PRE: pixels in order, no swizzle)
    constexpr int NPRE = TAPS == 1 ? (12 * BN) / kThreads : (3 * PP + kThreads - 1) / kThreads;      // PRE: 16-byte units per thread and chunk
    constexpr int NT = TAPS == 1 ? BN / 128 : (2 * PP + kThreads - 1) / kThreads;       // staging tasks per thread: 1x1 (4 channels x 4 pixels), 3x3 (4 channels x 1 pixel)
    constexpr int NXR = TAPS == 1 ? NT * 4 * (S2 ? 8 : 4) : NT * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char bx_smem[];
    v4f* sA = reinterpret_cast<v4f*>(bx_smem);                                       // [3][GA][BM] units
    float* sE = reinterpret_cast<float*>(bx_smem + (size_t)AUNITS * 16);             // [2][BM]
    unsigned char* sB = bx_smem + (size_t)AUNITS * 16 + 2 * BM * 4;                  // [3][k groups][positions] units
    constexpr int bTerm = TAPS == 1 ? 4 * POS1 * 16 : PPA * 16;                    // bytes of one term of the B image (an immediate of the LDS reads)

    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
#ifdef BX_STAGGER
    // measurement build (tools/build_variant.sh): delay every other workgroup by ~BX_STAGGER x 64 cycles, so that the two workgroups
    // of a CU -- same program, same length -- do not run their staging and their MFMA phases in lockstep
    if ((BX_STAGGER_MODE == 1 && ((bid >> 3) & 1)) || (BX_STAGGER_MODE == 2 && ((bid >> 8) & 1))) __builtin_amdgcn_s_sleep(BX_STAGGER);
#endif
    const int xcd = bid & 7;
    int slot = bid >> 3;
    const int per_part = p.mtiles * ((p.ptiles + 7) >> 3);          // slots of one K part (all of them when ksplit == 1)
    const int kpart = slot / per_part;
    slot -= kpart * per_part;
    const int mt = slot % p.mtiles;
    const int pt = (slot / p.mtiles) * 8 + xcd;
    if (pt >= p.ptiles) return;
    const int HW = p.H * p.W, HWo = p.Ho * p.Wo;
    int n, p0 = 0, oy0 = 0, ox0 = 0;
    int fy0 = 0, fpw = 0, fpp = 0;                                   // flat 3x3: first plane row of the tile, patch row pitch, patch pixels
    if (TAPS == 1 || FLAT) {
        n = pt / p.tiles_x;
        p0 = (pt - n * p.tiles_x) * BN;                          // first pixel of the tile in the flattened output plane
        if (FLAT) {
            const int last = p0 + BN - 1 < HWo - 1 ? p0 + BN - 1 : HWo - 1;
            fy0 = p0 / p.W;
            fpw = p.W + 2 * DIL;
            fpp = (last / p.W - fy0 + 1 + 2 * DIL) * fpw;            // (<= kBxFlatPP: checked by the host)
        }
    } else {
        const int tpi = p.tiles_x * p.tiles_y;
        n = pt / tpi;
        const int trem = pt - n * tpi;
        const int tyi = trem / p.tiles_x;
        oy0 = tyi * TH;
        ox0 = (trem - tyi * p.tiles_x) * TW;
    }
    const int m0 = mt * BM;

    // ---- staging descriptors (the same for every chunk): byte offset inside the chunk's x resource, byte offset in the B image --
    int goff[NT], loff[NT];
    bool s2top[NT], s2left[NT];                 // stride-2 form with taps: the task's output quad lies in output row 0 / starts at output column 0
    int pgoff[PRE ? NPRE : 1], ploff[PRE ? NPRE : 1];
    if (PRE) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int idx = tid + j * kThreads;
            pgoff[j] = kBxOut;
            if (TAPS == 1) {
                const int gt = idx / BN, pp = idx - gt * BN;            // memory order of a chunk: [k group 0..3][term 0..2][pixel]
                const int g = gt / 3, term = gt - 3 * g;
                if (p0 + pp < HWo) pgoff[j] = (gt * HW + p0 + pp) * 16;
                ploff[j] = ((term * 4 + g) * POS1 + pp) * 16;
            } else {
                ploff[j] = (DUMP + (tid & 15)) * 16;
                if (idx < 3 * PP) {
                    const int term = idx / PP, pix = idx - term * PP;
                    const int py = pix / PW, px = pix - py * PW;
                    const int iy = oy0 - DIL + py, ix = ox0 - DIL + px;
                    ploff[j] = (term * PPA + pix) * 16;
                    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) pgoff[j] = (term * HW + iy * p.W + ix) * 16;
                }
            }
        }
    }
    if (PRE) {
        // (no f32 staging descriptors)
    } else if (TAPS == 1) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            // 16 consecutive lanes = 8 pixel quads x the two channel quads of one k group: their 8-byte LDS stores (16-byte pitch,
            // halves 0 / 8) touch 16 different bank pairs; the global loads still run over 128-byte segments
            const int task = tid + j * kThreads;
            const int rest = task >> 4;
            const int pq = (task & 7) | ((rest % (BN / 32)) << 3);                // pixel quad of the tile
            const int q = ((rest / (BN / 32)) << 1) | ((task >> 3) & 1);          // channel quad 0..7 of the chunk
            const int po = p0 + 4 * pq;
            goff[j] = kBxOut;
            s2top[j] = s2left[j] = false;
            if (po < HWo) {
                int pix = po;
                if (S2) {
                    const int oy = po / p.Wo, ox = po - oy * p.Wo;
                    pix = 2 * oy * p.W + 2 * ox;
                    s2top[j] = oy == 0;
                    s2left[j] = ox == 0;
                }
                goff[j] = (q * 4 * HW + pix) * 4;
            }
            loff[j] = ((q >> 1) * POS1 + bx_pos1(4 * pq)) * 16 + (q & 1) * 8;
        }
    } else {
        const int iy0 = FLAT ? fy0 - DIL : oy0 - DIL, ix0 = FLAT ? -DIL : ox0 - DIL;
        const int pp_ = FLAT ? fpp : PP, pw_ = FLAT ? fpw : PW;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int e = tid + j * kThreads;
            goff[j] = kBxOut;
            loff[j] = (DUMP + (tid & 15)) * 16;         // (a task beyond the patch stores zeros there: no branch in the loop)
            if (e < 2 * pp_) {
                const int cq = e >= pp_ ? 1 : 0, pix = e - cq * pp_;
                const int py = pix / pw_, px = pix - py * pw_;
                const int iy = iy0 + py, ix = ix0 + px;
                loff[j] = pix * 16 + cq * 8;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) goff[j] = (cq * 4 * HW + iy * p.W + ix) * 4;
            }
        }
    }
    // (the last weight load of a thread beyond the image re-reads the image's last unit and stores it there again: the same value
    //  from several threads, and no predicate -- an exec-masked store inside the loop made hipcc rotate the loop and copy all 64
    //  accumulator registers at its head)
    const int wtail = (tid + (NW - 1) * kThreads < AUNITS ? tid + (NW - 1) * kThreads : AUNITS - 1) * 16;

    // ---- MFMA operand addressing -----------------------------------------------------------------------------------------------
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int aBase = h * BM + wm * 64 + l31;                       // unit index inside a term's [GA][BM]
    int bBase[2];                                                   // byte offset inside a term of the B image
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        if (TAPS == 1) bBase[tn] = PRE ? (h * POS1 + wn * 64 + tn * 32 + l31) * 16 : (h * POS1 + bx_pos1(wn * 64 + tn * 32 + l31)) * 16;
        else if (FLAT) {
            const int pp = p0 + wn * 64 + tn * 32 + l31;
            const int pc = pp < HWo ? pp : HWo - 1;             // (pixels beyond the plane: any position of the patch; never stored)
            const int y = pc / p.W, x = pc - y * p.W;
            bBase[tn] = ((y - fy0) * fpw + x) * 16;
        } else if (TW == 32) bBase[tn] = ((wn * 2 + tn) * PW + l31) * 16;
        else bBase[tn] = ((wn * 4 + tn * 2 + (l31 >> 4)) * PW + (l31 & 15)) * 16;        // 16 x 16: an MFMA column tile is 2 rows x 16 columns
    }
    int toff[SLABS];                                                // 3x3: the tap of this lane half in every 16-k step
#pragma unroll
    for (int s = 0; s < SLABS; ++s) {
        const int t = 2 * s + h > 8 ? 8 : 2 * s + h;
        toff[s] = TAPS == 1 ? 0 : ((t / 3) * (FLAT ? fpw : PW) + (t % 3)) * DIL * 16;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;

    const float* xb = PRE ? nullptr : p.x + (size_t)n * p.Cin * HW;
    // DUAL (1x1 stride 1 only): the chunks of a second input / weight image follow those of the first; one accumulator set
    constexpr bool DUAL_OK = TAPS == 1 && !S2;
    const int cpt = (p.Cin + CK - 1) / CK;                           // chunks per tap
    const int n1 = (S2 && p.t9) ? 9 * cpt : cpt;
    const int n2 = DUAL_OK ? (p.Cin2 + CK - 1) / CK : 0;
    const int nchunks = n1 + n2;
    const v4f* wb = p.wp + (size_t)mt * n1 * AUNITS;
    const v4f* wb2 = DUAL_OK && p.Cin2 ? p.wp2 + (size_t)mt * n2 * AUNITS : nullptr;
    const float* xb2 = DUAL_OK && p.Cin2 ? p.x2 + (size_t)n * p.Cin2 * HW : nullptr;
    const int hw4 = HW * 4;
    v4f wr[NW];
    float xr[PRE ? 1 : NXR];
    v4f xq[PRE ? NPRE : 1];
    const int cgroups = (p.Cin + 7) >> 3;
    const v4f* x3b = PRE ? p.x3 + (size_t)n * cgroups * 3 * HW : nullptr;

    if (tid < BM) {
        const bool real = m0 + tid < p.Cout;
        sE[tid] = (p.scale && real) ? p.scale[m0 + tid] : 1.0f;
        sE[BM + tid] = (p.shift && real) ? p.shift[m0 + tid] : 0.0f;
    }

    bool pend_kx0 = false;                      // stride-2 form with taps: the chunk in the staging registers belongs to a tap of column kx = 0
    auto fetch = [&](int t) {
        // (the chunk index is wave-uniform; said explicitly, because a resource descriptor the compiler takes for divergent is
        //  applied through a readfirstlane loop around EVERY load -- 11 more instructions per load in the 3x3 form)
        t = __builtin_amdgcn_readfirstlane(t);
        const bool second = DUAL_OK && t >= n1;             // wave-uniform: scalar selects of the bases below
        const v4f* wsrc = second ? wb2 : wb;
        const float* xsrc = second ? xb2 : xb;
        const int cin = second ? p.Cin2 : p.Cin;
        if (second) t -= n1;
        const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(const_cast<v4f*>(wsrc + (size_t)t * AUNITS), 0, AUNITS * 16, kBxRsrcFlags);
        // stride-2 form with taps: chunk t = (tap, channel chunk); the tap (ky, kx) shifts the loads by (ky - 1) rows and (kx - 1)
        // columns.  What falls outside the picture: row -1 (ky 0 on output row 0: the task reads nothing -- zeros) and column -1
        // (kx 0 on output column 0: the task loads from column 0 instead and stage() takes its elements one place further left,
        // a zero first).  Rows / columns beyond the other edges do not occur (H even, W % 8 == 0).
        int tapoff = 0;
        bool ky0 = false, kx0 = false;
        if (S2 && p.t9) {
            const int tap = t / cpt;
            t -= tap * cpt;
            const int ky = tap / 3, kx = tap - 3 * ky;
            tapoff = ((ky - 1) * p.W + (kx - 1)) * 4;
            ky0 = ky == 0;
            kx0 = kx == 0;
        }
        pend_kx0 = kx0;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            if (WTAIL && j == NW - 1) wr[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wres, wtail, 0, 0));
            else wr[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wres, tid * 16, j * kThreads * 16, 0));
        }
        // (the channels of a last, partial chunk that do not exist lie beyond the resource: zeros against zero weights)
        const int cleft = cin - t * CK;
        const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xsrc + (size_t)t * CK * HW), 0, (cleft < CK ? cleft : CK) * hw4, kBxRsrcFlags);
        if (PRE) {
            // (the k groups of a last, partial chunk that do not exist lie beyond the resource: zeros against zero weights)
            constexpr int GPC = TAPS == 1 ? 4 : 1;
            const int gleft = cgroups - t * GPC;
            const __amdgpu_buffer_rsrc_t x3res = __builtin_amdgcn_make_buffer_rsrc(const_cast<v4f*>(x3b + (size_t)t * GPC * 3 * HW), 0,
                                                                                   (gleft < GPC ? gleft : GPC) * 3 * HW * 16, kBxRsrcFlags);
#pragma unroll
            for (int j = 0; j < NPRE; ++j) xq[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(x3res, pgoff[j], 0, 0));
        } else if (TAPS == 1) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (S2) {
                        // input columns c .. c + 3 and c + 3 .. c + 6 (c = 2 ox + kx - 1): the four the quad needs are c, c + 2 | c + 4, c + 6
                        // and nothing is read beyond the last one -- no load runs over the end of a row
                        const int vo_ = (ky0 && s2top[j]) ? kBxOut : goff[j] + tapoff + ((kx0 && s2left[j]) ? 4 : 0);
                        const v4f lo = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xres, vo_, a * hw4, 0));
                        const v4f hi = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xres, vo_, a * hw4 + 12, 0));
                        xr[(j * 4 + a) * 8 + 0] = lo.x; xr[(j * 4 + a) * 8 + 1] = lo.y; xr[(j * 4 + a) * 8 + 2] = lo.z; xr[(j * 4 + a) * 8 + 3] = lo.w;
                        xr[(j * 4 + a) * 8 + 4] = hi.x; xr[(j * 4 + a) * 8 + 5] = hi.y; xr[(j * 4 + a) * 8 + 6] = hi.z; xr[(j * 4 + a) * 8 + 7] = hi.w;
                    } else {
                        const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xres, goff[j], a * hw4, 0));
                        xr[(j * 4 + a) * 4 + 0] = v.x; xr[(j * 4 + a) * 4 + 1] = v.y; xr[(j * 4 + a) * 4 + 2] = v.z; xr[(j * 4 + a) * 4 + 3] = v.w;
                    }
                }
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int a = 0; a < 4; ++a) xr[j * 4 + a] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, goff[j], a * hw4, 0));
        }
    };

    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            if (WTAIL && j == NW - 1) *reinterpret_cast<v4f*>(reinterpret_cast<unsigned char*>(sA) + wtail) = wr[j];
            else sA[tid + j * kThreads] = wr[j];
        }
        if (PRE) {
#pragma unroll
            for (int j = 0; j < NPRE; ++j) *reinterpret_cast<v4f*>(sB + ploff[j]) = xq[j];
        } else if (TAPS == 1) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {                       // pixel i of the quad: channels a = 0..3
                    constexpr int E = S2 ? 8 : 4;
                    // stride 2: of the eight loaded values (columns c .. c + 3, c + 3 .. c + 6) pixel i is column c + 2 i; a task on
                    // the left edge under a kx = 0 tap loaded from column 0 = c + 1: pixel i is one place further left, pixel 0 is zero
                    constexpr int np_[4] = {0, 2, 5, 7}, ep_[4] = {0, 1, 3, 6};
                    const bool edge = S2 && pend_kx0 && s2left[j];
                    auto el = [&](int a) {
                        if (!S2) return xr[(j * 4 + a) * E + i];
                        const float nv = xr[(j * 4 + a) * E + np_[i]], ev = i == 0 ? 0.0f : xr[(j * 4 + a) * E + ep_[i]];
                        return edge ? ev : nv;
                    };
                    unsigned h0, m0_, l0, h1, m1, l1;
                    bx_split2(el(0), el(1), h0, m0_, l0);
                    bx_split2(el(2), el(3), h1, m1, l1);
                    unsigned char* dst = sB + loff[j] + i * 36 * 16;
                    *reinterpret_cast<v2u*>(dst) = (v2u){h0, h1};
                    *reinterpret_cast<v2u*>(dst + bTerm) = (v2u){m0_, m1};
                    *reinterpret_cast<v2u*>(dst + 2 * bTerm) = (v2u){l0, l1};
                }
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                unsigned h0, m0_, l0, h1, m1, l1;
                bx_split2(xr[j * 4 + 0], xr[j * 4 + 1], h0, m0_, l0);
                bx_split2(xr[j * 4 + 2], xr[j * 4 + 3], h1, m1, l1);
                unsigned char* dst = sB + loff[j];
                *reinterpret_cast<v2u*>(dst) = (v2u){h0, h1};
                *reinterpret_cast<v2u*>(dst + bTerm) = (v2u){m0_, m1};
                *reinterpret_cast<v2u*>(dst + 2 * bTerm) = (v2u){l0, l1};
            }
        }
    };

#ifndef BX_PIPE
#define BX_PIPE 0
#endif
    auto mfma_chunk = [&]() {
        if (BX_PIPE) {
            // the operand fragments of 16-k step s + 1 are requested BEFORE the MFMAs of step s (two register sets): no MFMA group
            // but the first of a chunk waits for its own LDS reads
            bf8 fa[2][2][3], fb[2][2][3];
            auto load = [&](int s_, int st_) {
#pragma unroll
                for (int term = 0; term < 3; ++term) {
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        const int off = bBase[tn] + term * bTerm + (TAPS == 1 ? 2 * s_ * POS1 * 16 : toff[s_]);
                        fb[st_][tn][term] = __builtin_bit_cast(bf8, *reinterpret_cast<const v4f*>(sB + off));
                    }
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) fa[st_][tm][term] = __builtin_bit_cast(bf8, sA[(term * GA + 2 * s_) * BM + aBase + tm * 32]);
                }
            };
            load(0, 0);
#pragma unroll
            for (int s = 0; s < SLABS; ++s) {
                if (s + 1 < SLABS) load(s + 1, (s + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);          // (the machine scheduler sinks the reads back to their uses otherwise)
                const int st = s & 1;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
#ifndef BX_PROBE_3OF6       // (measurement build, WRONG results: three of the six products, as a two-term split would issue -- NOTEBOOK 16.2)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][tm][1], fb[st][tn][1], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][tm][0], fb[st][tn][2], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][tm][2], fb[st][tn][0], acc[tm][tn], 0, 0, 0);
#endif
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][tm][0], fb[st][tn][1], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][tm][1], fb[st][tn][0], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][tm][0], fb[st][tn][0], acc[tm][tn], 0, 0, 0);
                    }
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < SLABS; ++s) {
            bf8 b[2][3];
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const int off = bBase[tn] + term * bTerm + (TAPS == 1 ? 2 * s * POS1 * 16 : toff[s]);
                    const v4f q = *reinterpret_cast<const v4f*>(sB + off);
                    b[tn][term] = __builtin_bit_cast(bf8, q);
                }
#ifndef BX_ILV
#define BX_ILV 0
#endif
            if (BX_ILV) {
                // (measurement variant: consecutive MFMAs go to DIFFERENT accumulators -- the same products in the same order per
                //  accumulator, so the same bits -- in case a dependent accumulate chain does not issue back to back)
                bf8 a2[2][3];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int term = 0; term < 3; ++term) a2[tm][term] = __builtin_bit_cast(bf8, sA[(term * GA + 2 * s) * BM + aBase + tm * 32]);
                constexpr int ta[6] = {1, 0, 2, 0, 1, 0}, tb[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int st6 = 0; st6 < 6; ++st6)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[tm][ta[st6]], b[tn][tb[st6]], acc[tm][tn], 0, 0, 0);
                continue;
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                bf8 a[3];
#pragma unroll
                for (int term = 0; term < 3; ++term) {
                    const v4f q = sA[(term * GA + 2 * s) * BM + aBase + tm * 32];
                    a[term] = __builtin_bit_cast(bf8, q);
                }
                // the six products of order <= 2, smallest first
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
#ifndef BX_PROBE_3OF6
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[tn][1], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[tn][2], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[tn][0], acc[tm][tn], 0, 0, 0);
#endif
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[tn][1], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[tn][0], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[tn][0], acc[tm][tn], 0, 0, 0);
                }
            }
        }
    };

#ifdef BX_STAMPS
    const bool stamp = p.stamps != nullptr && tid == 0;
    long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long last_t = stamp ? clock64() : 0;
    const long long first_t = last_t;
#endif
    // One loop, no peeled last iteration: the fetch in the last pass re-reads the last chunk (never used).  With the last pass
    // peeled, hipcc merged its staging code with the loop's and copied every loop-carried register -- 64 accumulators and the
    // staging registers -- at the head of each iteration of the 3x3 form.
    // split-K: part k of `ksplit` walks the chunks [k * nchunks / ksplit, (k + 1) * nchunks / ksplit)
    const int t_lo = __builtin_amdgcn_readfirstlane((int)((long long)kpart * nchunks / p.ksplit));
    const int nloop = __builtin_amdgcn_readfirstlane((int)((long long)(kpart + 1) * nchunks / p.ksplit));      // (a scalar trip count: see fetch)
    fetch(t_lo);
    BX_T(0);
    for (int t = t_lo; t < nloop; ++t) {
        stage();
        BX_T(1);
        __syncthreads();
        BX_T(2);
        fetch(t + 1 < nloop ? t + 1 : t);
        __builtin_amdgcn_sched_barrier(0);          // the loads of chunk t + 1 are issued in FRONT of the MFMAs of chunk t
        BX_T(3);
        mfma_chunk();
        BX_T(4);
        __syncthreads();
        BX_T(5);
    }
    BX_T(6);

    // ---- epilogue: accumulator (row = (r & 3) + 8 (r >> 2) + 4 h, column = lane & 31) -> NCHW ----------------------------------
    // y (and the residual) of this M tile as a buffer resource of min(BM, Cout - m0) rows: a row beyond Cout or a pixel outside
    // the plane is out of range (dropped / zero) -- one 32-bit add per element, no predicates
    const int mrows = p.Cout - m0 < BM ? p.Cout - m0 : BM;
    const int row4 = HWo * 4;
    // (split-K: part 0 stores into y and adds the residual; the others store into their slice of `part` and see an EMPTY residual
    //  resource -- out-of-range reads are zeros, so the RES code needs no branch)
    float* ydst = kpart == 0 ? p.y : p.part + (size_t)(kpart - 1) * p.N * p.Cout * HWo;
    const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(ydst + ((size_t)n * p.Cout + m0) * HWo, 0, mrows * row4, kBxRsrcFlags);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(RES ? p.res + ((size_t)n * p.Cout + m0) * HWo : p.x), 0, (RES && kpart == 0) ? mrows * row4 : 0, kBxRsrcFlags);
    int vo[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        if (TAPS == 1 || FLAT) {
            const int pp = p0 + wn * 64 + tn * 32 + l31;
            vo[tn] = pp < HWo ? pp * 4 : kBxOut;
        } else {
            const int oy = TW == 32 ? oy0 + wn * 2 + tn : oy0 + wn * 4 + tn * 2 + (l31 >> 4);
            const int ox = TW == 32 ? ox0 + l31 : ox0 + (l31 & 15);
            vo[tn] = (oy < p.Ho && ox < p.Wo) ? (oy * p.Wo + ox) * 4 : kBxOut;
        }
    }
    // The epilogue is a large share of the layers with few chunks (K = 64 ... 256: the layer1 / layer2 1x1 layers, every bare
    // product of a training step), and its cost is its instruction count: BARE (no scale / shift, no ReLU: what training asks for)
    // skips the affine part; on a whole M tile (FULL) the row part of a store's address is a scalar offset -- a lane's 16 stores of
    // an MFMA tile then share ONE address register; scale / shift come as four 16-byte LDS reads per tile, the ReLU is one v_max.
    const float lo = p.relu ? 0.0f : -INFINITY;
    float rv[2][16];
    auto res_load = [&](int i, float (&dst)[16]) {
        const int tn = i >> 1, tm = i & 1;
        const int mb = wm * 64 + tm * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r)                          // (read once: non-temporal)
            dst[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, vo[tn] + (mb + (r & 3) + 8 * (r >> 2)) * row4, 0, 2));
    };
    auto epilogue = [&](auto bare_tag, auto full_tag) {
        constexpr bool BARE = decltype(bare_tag)::value, FULL = decltype(full_tag)::value;
        if (RES) res_load(0, rv[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tn = i >> 1, tm = i & 1;
            if (RES && i + 1 < 4) res_load(i + 1, rv[(i + 1) & 1]);
            const int mb = wm * 64 + tm * 32 + 4 * h;
            const int vbase = vo[tn] + mb * row4;               // FULL: the lane's part of the address (kBxOut + mb * row4 stays out of range)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4f sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                if (!BARE) {
                    sc = *reinterpret_cast<const v4f*>(sE + mb + 8 * g);
                    sh = *reinterpret_cast<const v4f*>(sE + BM + mb + 8 * g);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e, mr = e + 8 * g;    // row of the tile: mb + mr
                    float v = acc[tm][tn][r];
                    if (!BARE) v = mas_fmaf(v, sc[e], sh[e]);
                    if (RES) v += rv[i & 1][r];
                    if (!BARE) v = mas_vmax(v, lo);
                    if (FULL) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yres, vbase, mr * row4, 0);
                    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yres, vo[tn] + (mb + mr) * row4, 0, 0);
                }
            }
        }
    };
    const bool bare = p.scale == nullptr && p.shift == nullptr && !p.relu, full = m0 + BM <= p.Cout;     // wave-uniform
#ifdef BX_EPI_PAD
    // measurement build: BX_EPI_PAD dependent VALU instructions per accumulator register pair in front of the bare epilogue -- what
    // BatchNorm sums formed here would cost; the result reaches memory only on a condition that never holds
    if (bare) {
        float pad_s = 0.0f, pad_q = 0.0f;
#pragma unroll
        for (int rep = 0; rep < BX_EPI_PAD; ++rep)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float a0 = vo[0] == kBxOut ? 0.0f : acc[tm][0][r], a1 = vo[1] == kBxOut ? 0.0f : acc[tm][1][r];
                    pad_s += a0 + a1 + (float)rep;
                    pad_q = mas_fmaf(a0, a0, mas_fmaf(a1, a1, pad_q));
                }
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            pad_s += __shfl_xor(pad_s, off, 64);
            pad_q += __shfl_xor(pad_q, off, 64);
        }
        if (pad_s == 1.2345e-30f && pad_q == 5.4321e-30f) p.y[0] = pad_s;
    }
#endif
    if (bare) {
        if (full) epilogue(std::true_type{}, std::true_type{});
        else epilogue(std::true_type{}, std::false_type{});
    } else {
        if (full) epilogue(std::false_type{}, std::true_type{});
        else epilogue(std::false_type{}, std::false_type{});
    }
    // ---- BatchNorm partial sums of the tile (training forward): sum y, sum y^2 over the wave's 64 pixel columns, per row ------------
    // Entry 2 j + stat of a lane = row j = 16 tm + r of its half (stat 0: sum, 1: sum of squares) over its two columns; five halving
    // steps over the 32 lanes leave lane l31 with (sum, sum of squares) of row j = l31.  In f32 (a row of 64 values), stored as a
    // double pair: the accumulation across tiles is k_bn_stats_wide's, in double, in index order.
    if (!RES && p.stats != nullptr) {
        const bool ok0 = vo[0] != kBxOut, ok1 = vo[1] != kBxOut;
        float v[64];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a0 = ok0 ? acc[tm][0][r] : 0.0f, a1 = ok1 ? acc[tm][1][r] : 0.0f;
                v[2 * (16 * tm + r)] = a0 + a1;
                v[2 * (16 * tm + r) + 1] = mas_fmaf(a0, a0, a1 * a1);
            }
        float v32[32], v16[16], v8[8], v4_[4], v2_[2];
        bx_halve<4>(v, lane);
#pragma unroll
        for (int i = 0; i < 32; ++i) v32[i] = v[i];
        bx_halve<3>(v32, lane);
#pragma unroll
        for (int i = 0; i < 16; ++i) v16[i] = v32[i];
        bx_halve<2>(v16, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) v8[i] = v16[i];
        bx_halve<1>(v8, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) v4_[i] = v8[i];
        bx_halve<0>(v4_, lane);
        v2_[0] = v4_[0];
        v2_[1] = v4_[1];
        const int j = l31, tmj = j >> 4, rj = j & 15;
        const int m = m0 + wm * 64 + tmj * 32 + (rj & 3) + 8 * (rj >> 2) + 4 * h;
        if (m < p.Cout) p.stats[(size_t)m * p.stat_slots + pt * WN + wn] = make_double2((double)v2_[0], (double)v2_[1]);
    }
#ifdef BX_STAMPS
    if (stamp) {
        BX_T(7);
        unsigned long long* o = p.stamps + (size_t)bid * 10;
        for (int i = 0; i < 8; ++i) o[i] = (unsigned long long)acc_t[i];
        o[8] = (unsigned long long)(last_t - first_t);
        o[9] = (unsigned long long)first_t;
    }
#endif
}

// flat 3x3 tiles: the patch of 256 consecutive pixels = the full rows they touch + DIL rows above / below, DIL columns left / right
inline bool bx_flat_fits(int H, int W, int dil) {
    (void)H;
    const int rows = (256 + W - 2) / W + 1;                 // rows a run of 256 pixels can touch
    return W >= 8 && (rows + 2 * dil) * (W + 2 * dil) <= kBxFlatPP;
}

// the M tile of a layer's weight image: a pure function of (ksize, Cout), shared by the pack and the launch
inline int bx_bm(int ksize, int Cout) { return (ksize == 1 && Cout % 128 == 0) ? 128 : 64; }

template <int TAPS, int BM, int BN, int V, bool RES, int TW = 32, bool PRE = false>
int bx_launch(BxP p, int N, hipStream_t st) {
    constexpr int GA = BxGeo<TAPS>::GA;
    p.mtiles = (p.Cout + BM - 1) / BM;
    p.N = N;
    if (p.ksplit < 1) p.ksplit = 1;
    size_t bbytes;
    if (TAPS == 1) {
        p.tiles_x = (p.Ho * p.Wo + BN - 1) / BN;
        p.tiles_y = 1;
        bbytes = (size_t)3 * 4 * 144 * (BN / 128) * 16;
    } else if (TW == 0) {
        p.tiles_x = (p.Ho * p.Wo + BN - 1) / BN;
        p.tiles_y = 1;
        if (p.dil != V || !bx_flat_fits(p.H, p.W, V)) return MAS_ERR_RANGE;
        bbytes = (size_t)3 * kBxFlatPPA * 16;
    } else {
        constexpr int TWs = TW == 0 ? 32 : TW;
        p.tiles_x = (p.Wo + TWs - 1) / TWs;
        p.tiles_y = (p.Ho + 256 / TWs - 1) / (256 / TWs);
        if (p.dil != V) return MAS_ERR_RANGE;
        static_assert(kBxPPA >= kBxDump + 16, "dump positions");
        bbytes = (size_t)3 * kBxPPA * 16;
    }
    p.ptiles = N * p.tiles_x * p.tiles_y;
    const size_t smem = (size_t)3 * GA * BM * 16 + 2 * BM * 4 + bbytes;
    if (smem > 80 * 1024) return MAS_ERR_SHAPE;
    if (smem > 64 * 1024) {
        static bool raised[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        if (dev < 0 || dev >= 64 || !raised[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_bx<TAPS, BM, BN, V, RES, TW, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    80 * 1024);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0 && dev < 64) raised[dev] = true;
        }
    }
    const long long nblk = 8LL * ((p.ptiles + 7) / 8) * p.mtiles * p.ksplit;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
    hipLaunchKernelGGL((k_conv_bx<TAPS, BM, BN, V, RES, TW, PRE>), dim3((unsigned)nblk), dim3(kThreads), smem, st, p);
    return mas_launch_status();
}

template <int TAPS, int BM, int BN, int V, int TW = 32, bool PRE = false>
int bx_launch_r(const BxP& p, int N, hipStream_t st) {
    return p.res ? bx_launch<TAPS, BM, BN, V, true, TW, PRE>(p, N, st) : bx_launch<TAPS, BM, BN, V, false, TW, PRE>(p, N, st);
}

// ---- bx3: an activation tensor already split ---------------------------------------------------------------------------------
// x [N,C,H,W] f32  ->  [N][ceil(C/8)][3 terms][H*W][8 bf16]: one 16-byte unit = the (h | m | l) term of 8 consecutive channels of one
// pixel = one unit of the B image of k_conv_bx<..., PRE>.  Channels beyond C are zeros.  One thread per (channel group, 4 pixels).
__global__ __launch_bounds__(256) void k_bx3_split(const float* __restrict__ x, int C, int HW, int groups, unsigned* __restrict__ out) {
    const int n = blockIdx.z, g = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;            // pixel quad
    if (4 * q >= HW) return;
    const float* xb = x + ((size_t)n * C + 8 * g) * HW;
    float v[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const bool real = 8 * g + c < C;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[c][i] = (real && 4 * q + i < HW) ? xb[(size_t)c * HW + 4 * q + i] : 0.0f;
    }
    unsigned* ob = out + ((size_t)(n * groups + g) * 3) * HW * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (4 * q + i >= HW) break;
        unsigned hh[4], mm[4], ll[4];
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) bx_split2(v[2 * c2][i], v[2 * c2 + 1][i], hh[c2], mm[c2], ll[c2]);
        const size_t u = (size_t)(4 * q + i) * 4;
        *reinterpret_cast<uint4*>(ob + u) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
        *reinterpret_cast<uint4*>(ob + (size_t)HW * 4 + u) = make_uint4(mm[0], mm[1], mm[2], mm[3]);
        *reinterpret_cast<uint4*>(ob + (size_t)2 * HW * 4 + u) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
    }
}

// y[i] += part[0][i] + part[1][i] + ... in index order (the parts of a split-K launch; y already holds part 0 + residual)
__global__ __launch_bounds__(256) void k_bx_reduce(float* __restrict__ y, const float* __restrict__ part, int nparts, long long n4, long long stride) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        v4f a = reinterpret_cast<const v4f*>(y)[i];
        for (int k = 0; k < nparts; ++k) {
            const v4f b = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(part + (size_t)k * stride) + i);
            a += b;
        }
        reinterpret_cast<v4f*>(y)[i] = a;
    }
}

// The same pass with the BatchNorm partial sums of the finished y: one workgroup per (plane, chunk of 4096 elements) -- the pass is
// memory-bound, its VALUs idle -- stats[c][n * chunks + chunk] = (sum y, sum y^2), a fixed tree over the 256 threads in double.
constexpr int kBxRedChunk = 4096;
__global__ __launch_bounds__(256) void k_bx_reduce_stats(float* __restrict__ y, const float* __restrict__ part, int nparts, int C, int HW, int chunks,
                                                          long long stride, double2* __restrict__ stats) {
    __shared__ double s_red[2][4];
    const int chunk = blockIdx.x, plane = blockIdx.y;           // plane = n * C + c
    const int n = plane / C, c = plane - n * C;
    const long long base4 = ((long long)plane * HW + (long long)chunk * kBxRedChunk) >> 2;
    const int n4 = (min(HW - chunk * kBxRedChunk, kBxRedChunk)) >> 2;
    float s = 0.0f, q = 0.0f;
    for (int i = threadIdx.x; i < n4; i += 256) {
        v4f a = reinterpret_cast<const v4f*>(y)[base4 + i];
        for (int k = 0; k < nparts; ++k) a += __builtin_nontemporal_load(reinterpret_cast<const v4f*>(part + (size_t)k * stride) + base4 + i);
        reinterpret_cast<v4f*>(y)[base4 + i] = a;
        s += (a.x + a.y) + (a.z + a.w);
        q += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    }
    double S = (double)s, Q = (double)q;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { S += __shfl_down(S, off, 64); Q += __shfl_down(Q, off, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = S; s_red[1][threadIdx.x >> 6] = Q; }
    __syncthreads();
    if (threadIdx.x == 0)
        stats[(size_t)c * (gridDim.y / C) * chunks + (size_t)n * chunks + chunk] =
            make_double2(((s_red[0][0] + s_red[0][1]) + s_red[0][2]) + s_red[0][3], ((s_red[1][0] + s_red[1][1]) + s_red[1][2]) + s_red[1][3]);
}

// ---- the work-splitting plan of a bare stride-1 product (training) --------------------------------------------------------
// A launch has 8 * ceil(pixel tiles / 8) * M tiles workgroups for 2 x 256 resident slots.  The layers of the 48 x 48 planes have
// 144 ... 384 of them (profiles/r04/k_bx_train_table.md: no faster than the persistent f32 kernel there): the chunks of every tile
// are then dealt to `ksplit` workgroups whose partial tiles a second, memory-bound launch adds in index order (run-to-run
// identical; no flags, no waiting -- nothing that needs co-residency).  3x3: the tile shape with fewer padded pixels.
struct BxPlan { int ksplit, tw, wgs; };

// 32: 8 x 32 tiles, 16: 16 x 16 tiles, 1: flat tiles (256 consecutive pixels) -- the shape with the fewest tiles (ties: in that order)
inline int bx_tw(int ksize, int H, int W, int dil) {
    if (ksize != 3 || W < 16) return 32;
    const long long wide = (long long)((H + 7) / 8) * ((W + 31) / 32), square = (long long)((H + 15) / 16) * ((W + 15) / 16);
    const long long flat = bx_flat_fits(H, W, dil) ? ((long long)H * W + 255) / 256 : (1LL << 60);
    // (a tile of another shape has shorter contiguous rows: it must save a tenth of the tiles to pay -- 385 x 385 is 625 square
    //  tiles against 637 wide ones and 9 % slower with them, profiles/r05/k_bx_splitk_769.md)
    if (10 * flat <= 9 * wide && flat <= square) return 1;
    return 10 * square <= 9 * wide ? 16 : 32;
}

inline BxPlan bx_plan(int N, int Cin, int H, int W, int Cout, int ksize, int dil) {
    (void)dil;
    BxPlan pl;
    pl.tw = bx_tw(ksize, H, W, dil);
    const int BM = bx_bm(ksize, Cout);
    const int BN = ksize == 1 ? (BM == 128 ? 128 : 256) : 256;
    const long long ptiles = (ksize == 1 || pl.tw == 1) ? (long long)N * ((H * W + BN - 1) / BN)
                                                        : (long long)N * ((W + pl.tw - 1) / pl.tw) * ((H + 256 / pl.tw - 1) / (256 / pl.tw));
    const long long wg1 = 8 * ((ptiles + 7) / 8) * ((Cout + BM - 1) / BM);
    const int ck = ksize == 1 ? 32 : 8, nch = (Cin + ck - 1) / ck;
    // cost of a plan in microseconds: (workgroups / 512 resident slots, at least one "round") x (chunks per part + the fixed
    // prologue / epilogue of a workgroup) x the time of a chunk + the reduction pass (launch gap + its bytes at ~4 TB/s).  The
    // constants are fitted to tools/bx_splitk_sweep.py (profiles/r05/k_bx_splitk.md): the plan is within 4 % of the best (ksplit,
    // tile) of every layer of the training step.
    const double fixed = ksize == 1 ? 3.0 : 1.2;
    const double chunk_us = ksize == 1 ? 2.0 : 5.0;
    const double out_mb = (double)N * Cout * H * W * 4e-6;
    double best = 1e30;
    pl.ksplit = 1;
    for (int ks = 1; ks <= 8; ++ks) {
        if (ks > 1 && nch / ks < 4) break;
        const long long wgs = wg1 * ks;
        // (workgroups beyond the resident ones start as others end: no whole second round)
        const double rounds = wgs <= 512 ? 1.0 : (double)wgs / 512.0;
        // (a CU that holds one workgroup instead of two runs it faster, but not twice as fast)
        const double alone = wgs <= 256 ? 0.7 : 1.0;
        double cost = rounds * (((nch + ks - 1) / ks) + fixed) * chunk_us * alone;
        if (ks > 1) cost += 4.0 + (ks + 1) * out_mb / 4.0;
        if (cost < best * 0.97) { best = cost; pl.ksplit = ks; }
    }
    pl.wgs = (int)(wg1 * pl.ksplit);
    return pl;
}
}  // namespace

extern "C" int mas_conv_bx_supported(int ksize, int stride, int dil, int Cin, int Cout, int H, int W) {
    if (Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    if ((long long)Cin * H * W > 0x7fffffffLL) return 0;
    // byte sizes of the buffer resources: a chunk of x (32 / 8 channels of one picture), one M tile of y (<= 128 rows)
    if (32LL * H * W * 4 >= 0x7fffffffLL || 128LL * H * W * 4 >= 0x7fffffffLL) return 0;
    if (ksize == 1) {
        if (dil != 1) return 0;
        // (stride 1: any plane -- the product is pixel-local, so a pixel quad that runs over the end of its plane only feeds output
        //  pixels that are never stored; gfx950 runs 16-byte loads at 4-byte aligned addresses at the aligned rate)
        if (stride == 1) return 1;
        if (stride == 2) return H % 2 == 0 && W % 8 == 0;        // output quads stay inside a row and start 16-byte aligned
        return 0;
    }
    if (ksize == 3 && stride == 2)      // nine shifted 1x1 stride-2 products: whole chunks of 32 channels, output quads as the 1x1 stride-2 form
        return dil == 1 && Cin % 32 == 0 && H % 2 == 0 && W % 8 == 0 && W >= 8;
    if (ksize == 3) return stride == 1 && (dil == 1 || dil == 2) && W >= 32;
    return 0;
}

extern "C" long long mas_conv_bx_packed_bytes(int ksize, int Cin, int Cout, int role) {
    if (role == 2) {                    // the forward product of the 3x3 stride-2 convolution: 9 x (Cin / 32) chunks of the 1x1 form
        if (ksize != 3 || Cin <= 0 || Cout <= 0 || Cin % 32 != 0) return 0;
        const int BM = bx_bm(1, Cout);
        return (long long)((Cout + BM - 1) / BM) * 9 * (Cin / 32) * 3 * 4 * BM * 16;
    }
    if ((ksize != 1 && ksize != 3) || Cin <= 0 || Cout <= 0 || (role != 0 && role != 1)) return 0;
    const int M = role ? Cin : Cout, K = role ? Cout : Cin;
    const int BM = bx_bm(ksize, M), ck = ksize == 1 ? 32 : 8, ga = ksize == 1 ? 4 : kBxTaps3;
    return (long long)((M + BM - 1) / BM) * ((K + ck - 1) / ck) * 3 * ga * BM * 16;
}

namespace {
bool bx_fill_job(BxPackJob* jb, const float* w, const float* row_scale, int Cout, int Cin, int ksize, int role, void* wp, unsigned first_block) {
    const long long bytes = mas_conv_bx_packed_bytes(ksize, Cin, Cout, role);
    if (bytes <= 0 || !w || !wp || (uintptr_t)wp % 16 != 0) return false;
    jb->w = w; jb->row_scale = row_scale; jb->out = static_cast<unsigned*>(wp); jb->Cout = Cout; jb->Cin = Cin; jb->taps = ksize * ksize; jb->role = role;
    jb->BM = bx_bm(ksize, role ? Cin : Cout); jb->units = bytes / 48; jb->first_block = first_block;       // threads: one per three units
    jb->t9 = 0;
    if (role == 2) {
        jb->role = 0;
        jb->t9 = 1;
        jb->BM = bx_bm(1, Cout);
    }
    return true;
}
}  // namespace

extern "C" int mas_conv_bx_pack(const float* w, const float* row_scale, int Cout, int Cin, int ksize, int role, void* wp, void* stream) {
    if (!w || !wp) return MAS_ERR_NULL;
    if ((uintptr_t)wp % 16 != 0) return MAS_ERR_ALIGN;
    if (row_scale && role == 1) return MAS_ERR_RANGE;
    BxPackJob jb;
    if (!bx_fill_job(&jb, w, row_scale, Cout, Cin, ksize, role, wp, 0)) return MAS_ERR_SHAPE;
    hipLaunchKernelGGL(k_bx_pack, dim3((unsigned)((jb.units + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), jb);
    return mas_launch_status();
}

extern "C" size_t mas_conv_bx_pack_job_bytes(void) { return sizeof(BxPackJob); }

/* fill one job record (host memory) of a multi-tensor pack; returns the number of 256-thread blocks the job needs (0: rejected) */
extern "C" unsigned mas_conv_bx_pack_job(void* job_host, const float* w, int Cout, int Cin, int ksize, int role, void* wp, unsigned first_block) {
    if (!job_host) return 0;
    BxPackJob* jb = static_cast<BxPackJob*>(job_host);
    if (!bx_fill_job(jb, w, nullptr, Cout, Cin, ksize, role, wp, first_block)) return 0;
    return (unsigned)((jb->units + 255) / 256);
}

/* jobs_dev: `njobs` records (mas_conv_bx_pack_job, in ascending first_block order, copied to the device by the caller), `nblocks` blocks in all */
extern "C" int mas_conv_bx_pack_multi(const void* jobs_dev, int njobs, unsigned nblocks, void* stream) {
    if (!jobs_dev) return MAS_ERR_NULL;
    if (njobs <= 0 || nblocks == 0) return MAS_ERR_SHAPE;
    hipLaunchKernelGGL(k_bx_pack_multi, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const BxPackJob*>(jobs_dev), njobs);
    return mas_launch_status();
}

extern "C" int mas_conv_bx_fwd(const float* x, const void* wp, int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil,
                               const float* scale, const float* shift, const float* residual, int relu, float* y, void* stream) {
    if (!x || !wp || !y) return MAS_ERR_NULL;
    if ((scale == nullptr) != (shift == nullptr)) return MAS_ERR_NULL;
    if (N <= 0) return MAS_ERR_SHAPE;
    if (!mas_conv_bx_supported(ksize, stride, dil, Cin, Cout, H, W)) return MAS_ERR_SHAPE;
    if ((uintptr_t)wp % 16 != 0 || (uintptr_t)x % 4 != 0) return MAS_ERR_ALIGN;
    if (ksize == 1 && stride == 2 && (uintptr_t)x % 16 != 0) return MAS_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    BxP p;
    p.x = x; p.wp = static_cast<const v4f*>(wp); p.scale = scale; p.shift = shift; p.res = residual; p.y = y;
    p.x2 = nullptr; p.wp2 = nullptr; p.Cin2 = 0; p.ksplit = 1; p.part = nullptr; p.x3 = nullptr; p.stats = nullptr; p.stat_slots = 0; p.t9 = 0;
    p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.dil = dil; p.relu = relu;
#ifdef BX_STAMPS
    p.stamps = g_bx_stamps;
#endif
    p.Ho = (H - 1) / stride + 1;
    p.Wo = (W - 1) / stride + 1;
    if (ksize == 3 && stride == 2) {        // (wp: the image of role 2) the stride-2 1x1 kernel with nine taps
        if ((uintptr_t)x % 16 != 0) return MAS_ERR_ALIGN;
        p.t9 = 1;
        return bx_bm(1, Cout) == 128 ? bx_launch_r<1, 128, 128, 2>(p, N, st) : bx_launch_r<1, 64, 256, 2>(p, N, st);
    }
    const int BM = bx_bm(ksize, Cout);
    if (ksize == 1) {
        if (stride == 2) return BM == 128 ? bx_launch_r<1, 128, 128, 2>(p, N, st) : bx_launch_r<1, 64, 256, 2>(p, N, st);
        return BM == 128 ? bx_launch_r<1, 128, 128, 1>(p, N, st) : bx_launch_r<1, 64, 256, 1>(p, N, st);
    }
    return dil == 1 ? bx_launch_r<9, 64, 256, 1>(p, N, st) : bx_launch_r<9, 64, 256, 2>(p, N, st);
}

#ifdef BX_STAMPS
// measurement build only: 10 u64 per workgroup (8 phase sums, total, start) of the NEXT launches; NULL switches it off
extern "C" int mas_conv_bx_debug_stamps(void* buf) {
    g_bx_stamps = static_cast<unsigned long long*>(buf);
    return 0;
}
#endif

extern "C" int mas_conv_bx_fwd_dual(const float* x1, const void* wp1, int Cin1, const float* x2, const void* wp2, int Cin2, int N, int H, int W,
                                    int Cout, const float* shift, int relu, float* y, void* stream) {
    if (!x1 || !wp1 || !x2 || !wp2 || !y) return MAS_ERR_NULL;
    if (N <= 0 || Cin2 <= 0) return MAS_ERR_SHAPE;
    if (!mas_conv_bx_supported(1, 1, 1, Cin1, Cout, H, W) || !mas_conv_bx_supported(1, 1, 1, Cin2, Cout, H, W)) return MAS_ERR_SHAPE;
    if ((uintptr_t)wp1 % 16 != 0 || (uintptr_t)wp2 % 16 != 0 || (uintptr_t)x1 % 4 != 0 || (uintptr_t)x2 % 4 != 0) return MAS_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    BxP p;
    p.x = x1; p.wp = static_cast<const v4f*>(wp1); p.scale = nullptr; p.shift = shift; p.res = nullptr; p.y = y;
    p.x2 = x2; p.wp2 = static_cast<const v4f*>(wp2); p.Cin2 = Cin2; p.ksplit = 1; p.part = nullptr; p.x3 = nullptr; p.stats = nullptr; p.stat_slots = 0; p.t9 = 0;
    p.Cin = Cin1; p.H = H; p.W = W; p.Cout = Cout; p.dil = 1; p.relu = relu;
#ifdef BX_STAMPS
    p.stamps = g_bx_stamps;
#endif
    p.Ho = H; p.Wo = W;
    return bx_bm(1, Cout) == 128 ? bx_launch<1, 128, 128, 1, false>(p, N, st) : bx_launch<1, 64, 256, 1, false>(p, N, st);
}

/* ---- the bare stride-1 product of a training step with the work-splitting plan --------------------------------------------- */
extern "C" int mas_conv_bx_train_plan(int N, int Cin, int H, int W, int Cout, int ksize, int dil, int* out3) {
    if (!out3) return MAS_ERR_NULL;
    if (N <= 0 || !mas_conv_bx_supported(ksize, 1, dil, Cin, Cout, H, W)) return MAS_ERR_SHAPE;
    const BxPlan pl = bx_plan(N, Cin, H, W, Cout, ksize, dil);
    out3[0] = pl.ksplit; out3[1] = pl.tw; out3[2] = pl.wgs;
    return 0;
}

extern "C" size_t mas_conv_bx_train_workspace_bytes(int N, int Cout, int H, int W, int ksplit) {
    if (N <= 0 || Cout <= 0 || H <= 0 || W <= 0 || ksplit <= 1) return 0;
    return (size_t)(ksplit - 1) * N * Cout * H * W * sizeof(float);
}

/* slots per output channel of the BatchNorm partial sums a FORWARD launch can form in its epilogue (pixel tiles x wave columns);
 * 0: this plan cannot (split K: the final values exist only after the reduction pass) */
extern "C" int mas_conv_bx_train_stat_slots(int N, int H, int W, int Cout, int ksize, int dil, int ksplit, int tile_w) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || ksplit < 1) return 0;
    if (ksplit > 1) {                       // split K: the sums come out of the reduction pass, one slot per (picture, 4096-element chunk of a plane)
        const long long HW = (long long)H * W;
        if (HW % 4 != 0 || (long long)N * Cout > 65535) return 0;
        return (int)(N * ((HW + kBxRedChunk - 1) / kBxRedChunk));
    }
    const int BM = bx_bm(ksize, Cout), BN = ksize == 1 ? (BM == 128 ? 128 : 256) : 256;
    long long ptiles;
    if (ksize == 1 || tile_w == 1) ptiles = (long long)N * (((long long)H * W + BN - 1) / BN);
    else if (tile_w == 16 || tile_w == 32) ptiles = (long long)N * ((W + tile_w - 1) / tile_w) * ((H + 256 / tile_w - 1) / (256 / tile_w));
    else return 0;
    (void)dil;
    const long long slots = ptiles * (BN / 64);
    return slots > 0 && slots < 0x7fffffffLL ? (int)slots : 0;
}

extern "C" int mas_conv_bx_train(const float* x, const void* wp, int N, int Cin, int H, int W, int Cout, int ksize, int dil, const float* residual,
                                 float* y, int ksplit, int tile_w, void* workspace, size_t workspace_bytes, double* stats, void* stream) {
    if (!x || !wp || !y) return MAS_ERR_NULL;
    if (N <= 0) return MAS_ERR_SHAPE;
    if (!mas_conv_bx_supported(ksize, 1, dil, Cin, Cout, H, W)) return MAS_ERR_SHAPE;
    if ((uintptr_t)wp % 16 != 0 || (uintptr_t)x % 4 != 0) return MAS_ERR_ALIGN;
    const BxPlan pl = bx_plan(N, Cin, H, W, Cout, ksize, dil);
    if (ksplit <= 0) ksplit = pl.ksplit;
    if (tile_w <= 0) tile_w = pl.tw;
    const int ck = ksize == 1 ? 32 : 8, nch = (Cin + ck - 1) / ck;
    if (ksplit > nch || ksplit > 64) return MAS_ERR_RANGE;
    if (tile_w != 32 && !((tile_w == 16 || (tile_w == 1 && bx_flat_fits(H, W, dil))) && ksize == 3)) return MAS_ERR_RANGE;
    if (stats && (residual || (uintptr_t)stats % 16 != 0)) return MAS_ERR_RANGE;
    const size_t out_elems = (size_t)N * Cout * H * W;
    if (ksplit > 1) {
        if (!workspace) return MAS_ERR_NULL;
        if (workspace_bytes < (size_t)(ksplit - 1) * out_elems * sizeof(float)) return MAS_ERR_WORKSPACE;
        if ((uintptr_t)workspace % 16 != 0 || (uintptr_t)y % 16 != 0 || out_elems % 4 != 0) return MAS_ERR_ALIGN;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    BxP p;
    p.x = x; p.wp = static_cast<const v4f*>(wp); p.scale = nullptr; p.shift = nullptr; p.res = residual; p.y = y;
    p.x2 = nullptr; p.wp2 = nullptr; p.Cin2 = 0; p.ksplit = ksplit; p.part = static_cast<float*>(workspace); p.x3 = nullptr; p.t9 = 0;
    const int stat_slots = stats ? mas_conv_bx_train_stat_slots(N, H, W, Cout, ksize, dil, ksplit, tile_w) : 0;
    if (stats && stat_slots <= 0) return MAS_ERR_RANGE;
    p.stats = ksplit == 1 ? reinterpret_cast<double2*>(stats) : nullptr;       // (split K: formed by the reduction pass below)
    p.stat_slots = stat_slots;
    p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.dil = dil; p.relu = 0;
#ifdef BX_STAMPS
    p.stamps = g_bx_stamps;
#endif
    p.Ho = H; p.Wo = W;
    const int BM = bx_bm(ksize, Cout);
    int rc;
    if (ksize == 1) rc = BM == 128 ? bx_launch_r<1, 128, 128, 1>(p, N, st) : bx_launch_r<1, 64, 256, 1>(p, N, st);
    else if (tile_w == 16) rc = dil == 1 ? bx_launch_r<9, 64, 256, 1, 16>(p, N, st) : bx_launch_r<9, 64, 256, 2, 16>(p, N, st);
    else if (tile_w == 1) rc = dil == 1 ? bx_launch_r<9, 64, 256, 1, 0>(p, N, st) : bx_launch_r<9, 64, 256, 2, 0>(p, N, st);
    else rc = dil == 1 ? bx_launch_r<9, 64, 256, 1>(p, N, st) : bx_launch_r<9, 64, 256, 2>(p, N, st);
    if (rc != 0 || ksplit == 1) return rc;
    if (stats) {
        const int HW = H * W, chunks = (HW + kBxRedChunk - 1) / kBxRedChunk;
        hipLaunchKernelGGL(k_bx_reduce_stats, dim3((unsigned)chunks, (unsigned)(N * Cout)), dim3(256), 0, st, y, static_cast<const float*>(workspace),
                           ksplit - 1, Cout, HW, chunks, (long long)out_elems, reinterpret_cast<double2*>(stats));
        return mas_launch_status();
    }
    const long long n4 = (long long)(out_elems / 4);
    const long long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(k_bx_reduce, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, y, static_cast<const float*>(workspace),
                       ksplit - 1, n4, (long long)out_elems);
    return mas_launch_status();
}

/* ---- presplit activations (bx3) ------------------------------------------------------------------------------------------------ */
extern "C" long long mas_bx3_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return (long long)N * ((C + 7) / 8) * 3 * H * W * 16;
}

extern "C" int mas_bx3_split(const float* x, int N, int C, int H, int W, void* x3, void* stream) {
    if (!x || !x3) return MAS_ERR_NULL;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || N > 65535 || (C + 7) / 8 > 65535) return MAS_ERR_SHAPE;
    if ((uintptr_t)x3 % 16 != 0) return MAS_ERR_ALIGN;
    const int HW = H * W, groups = (C + 7) / 8;
    hipLaunchKernelGGL(k_bx3_split, dim3((unsigned)((HW / 4 + 1 + 255) / 256), (unsigned)groups, (unsigned)N), dim3(256), 0, static_cast<hipStream_t>(stream), x, C,
                       HW, groups, static_cast<unsigned*>(x3));
    return mas_launch_status();
}

/* mas_conv_bx_fwd on an input that is already split (stride 1): x3 = the bx3 form of x [N,Cin,H,W] */
extern "C" int mas_conv_bx_fwd_pre(const void* x3, const void* wp, int N, int Cin, int H, int W, int Cout, int ksize, int dil, const float* scale,
                                   const float* shift, const float* residual, int relu, float* y, void* stream) {
    if (!x3 || !wp || !y) return MAS_ERR_NULL;
    if ((scale == nullptr) != (shift == nullptr)) return MAS_ERR_NULL;
    if (N <= 0) return MAS_ERR_SHAPE;
    if (!mas_conv_bx_supported(ksize, 1, dil, Cin, Cout, H, W) || 12LL * H * W * 16 >= 0x7fffffffLL) return MAS_ERR_SHAPE;
    if ((uintptr_t)wp % 16 != 0 || (uintptr_t)x3 % 16 != 0) return MAS_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    BxP p;
    p.x = nullptr; p.x3 = static_cast<const v4f*>(x3); p.wp = static_cast<const v4f*>(wp); p.scale = scale; p.shift = shift; p.res = residual; p.y = y;
    p.x2 = nullptr; p.wp2 = nullptr; p.Cin2 = 0; p.ksplit = 1; p.part = nullptr; p.stats = nullptr; p.stat_slots = 0; p.t9 = 0;
    p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.dil = dil; p.relu = relu;
#ifdef BX_STAMPS
    p.stamps = g_bx_stamps;
#endif
    p.Ho = H; p.Wo = W;
    const int BM = bx_bm(ksize, Cout);
    if (ksize == 1) return BM == 128 ? bx_launch_r<1, 128, 128, 1, 32, true>(p, N, st) : bx_launch_r<1, 64, 256, 1, 32, true>(p, N, st);
    return dil == 1 ? bx_launch_r<9, 64, 256, 1, 32, true>(p, N, st) : bx_launch_r<9, 64, 256, 2, 32, true>(p, N, st);
}
