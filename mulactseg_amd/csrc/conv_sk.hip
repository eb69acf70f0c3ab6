// conv_sk.hip -- forward and input gradient of the dense convolutions in TRAINING (1x1 / 3x3, stride 1 / 2 forward, stride 1
// input gradient, dilation 1 / 2 / 4) as a persistent stream-K implicit GEMM on the f32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32), NCHW in and out.  The weight is read as a sequence of ready-made LDS images, one per (M tile,
// K chunk), that k_sk_pack writes from PyTorch's [Cout][Cin][k][k] tensor in one small launch per optimizer step and product
// (a workgroup's staging is then a linear 16-byte copy; building the image in the kernel costs 8-way bank-conflicted LDS
// stores for every pixel tile: measured 55-80 TFLOP/s instead of 100+); the input gradient is the same kernel on the image
// with the roles of the weight's two channel axes swapped and the taps mirrored.
// Reference: the nn.Conv2d calls (and their autograd backward) of models/segmentation/backbone/resnet.py:129-171 and
// models/segmentation/deeplabv3.py:85-137,168-245 inside trainer/active_joint_multi_predignore_lossdecomp.py:83-116.
//
// GEMM view per picture:  Y[m, p] = sum_{tap, c} A[m, (tap, c)] * X[c, pixel p shifted by tap]
//   M tile 128 (or 64) output channels x N tile 128 output pixels (a TH x TW patch), K walked in chunks of CK channels x taps.
//   One workgroup = 8 waves = one CU: wave w owns rows 32 (w % WM) .. and pixel group w / WM of the tile; waves w, w + 4 share a
//   SIMD.  Two LDS buffers; chunk t + 2 travels global -> registers (buffer loads at byte offsets computed once per pixel tile)
//   while chunk t is multiplied; every wave stores slot s of its staging registers (chunk t + 1) to the other buffer and refills
//   it behind its OWN MFMA group q (a wave's partner on the SIMD is starved while it multiplies: nothing hides there); one
//   barrier per chunk.  The pipeline runs across tile boundaries.
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
constexpr unsigned kSkSpinLimit = 1u << 22;     // polls (each >= one s_sleep + one agent-scope load, ~1.5 us) before a finisher gives up: seconds

struct SkP {
    const float* x;             // [N, K, H, W] input of the product (forward: activations; input gradient: dY)
    const float* w;             // packed weight: [mtiles][nch][KC / 8][2][BM][4] (k_sk_pack)
    const float* scale;
    const float* shift;
    const float* res;
    float* y;                   // [N, M, Ho, Wo]
    float* slots;               // [P][128 * 128] partial accumulator images
    unsigned* flags;            // [512] epoch flags, then one error word
    double2* stats;             // optional [M][ptiles * (8 / WM)] (sum y, sum y^2) of every tile row: the BatchNorm partials (k_bn_stats reads them)
    const float* zero;          // 64 bytes of zeros (source of the LDS-DMA of padding / out-of-range elements)
    unsigned epoch;
    unsigned long long* stamps; // optional [P][4]: wall-clock stamps of every workgroup (start, pipeline primed, loop done, end): tools only
    int K, H, W, M, Ho, Wo, relu;
    int Wy, HWy, os, oy_off, ox_off;    // where tile pixel (oy, ox) lands in the output plane: row oy * os + oy_off of Wy columns, HWy per channel
    int tiles_x, tiles_y, ptiles, mtiles, nch;
    int iters;                  // all (tile, chunk) iterations of the layer
    int N;                      // pictures
    int img, ck;                // floats of one weight image (taps * CK * BM), channels per chunk
    long long xstep;            // floats from a chunk's first channel to the next chunk's (CK * H * W)
    int P;                      // workgroups (one per CU)
    int rdp, sk_iters;          // whole tiles per workgroup (rounds of P tiles), iterations of the remaining tiles (stream-K part)
    // BatchNorm partials (p.stats): entry (row m, slot, group ng) at stats[m * stats_pitch + slot * NG + ng].  stats_acc: the
    // workgroup-accumulated layout (sk_stats_layout): slot = g / mtiles for the whole tiles of workgroup g (all of ONE M tile when P
    // is a multiple of mtiles: their sums stay in registers across the tiles and are folded and stored once), stats_w + t / mtiles
    // for stream-K tile t.  !stats_acc: one entry per (pixel tile, group), slot = pt.
    int stats_acc, stats_w, stats_pitch;
    int nosplit;                // MAS_SK_NOSPLIT: the remaining tiles go WHOLE to the first workgroups (no hand-off at all)
    unsigned spin_limit;        // polls a finisher waits for one contributor before it gives up
};

// Compile-time geometry of a kernel family: TH x TW output pixels per tile, the LDS input patch PH rows x PWL columns per
// channel (PWL a multiple of 4, PADL columns of halo on the left so that 16-byte groups stay aligned in memory), channel
// stride CS.  With these fixed, every LDS operand address of the multiply loop is one base register + an immediate.
// SUB != 0: one parity class (py, px) = (SUB >> 1, SUB & 1) of the input gradient of a stride-2 3x3 convolution, as a stride-1
// product over the dY plane with (1 + py) x (1 + px) taps at offsets {0, 1} (dX[2i + 1] takes dY[i] through weight row 2 and
// dY[i + 1] through row 0; dX[2i] takes dY[i] through row 1): no padding on the left / top, the patch starts at the tile.
// SUB = 64 / 128 (LW): a 3x3 stride-1 product on a NARROW plane (32 <= W <= 56 / 57 <= W <= 120) walks the plane's pixels in row-major
// order, 128 consecutive pixels per tile (TW = 128), with a patch of LW columns that holds the FULL rows the tile touches (at most
// RMAX of them) plus the halo: the 49 x 49 planes of the 769 crop take 19 tiles instead of 26 4 x 32 ones (a third of every
// second tile lies right of the plane), 97 x 97: 74 instead of 91.  The B-operand base of a lane then depends on the tile (two
// integer divisions per lane and tile); the tap offsets stay immediates (a row below is + PWL whatever the plane's width).
template <int TAPS, int CK_, int TW_, int STRIDE_, int DIL_, int SUB_ = 0>
struct SkG {
    static constexpr int CK = CK_, TW = TW_, STRIDE = STRIDE_, DIL = DIL_;
    static constexpr int LW = SUB_ >= 64 ? SUB_ : 0, SUB = SUB_ < 64 ? SUB_ : 0;
    static constexpr int TR = SUB ? 1 + (SUB >> 1) : (TAPS == 9 ? 3 : 1), TC = SUB ? 1 + (SUB & 1) : (TAPS == 9 ? 3 : 1);
    static_assert(TR * TC == TAPS && (SUB == 0 || (STRIDE == 1 && DIL == 1)), "tap geometry");
    static_assert(LW == 0 || (TAPS == 9 && STRIDE == 1 && TW == 128 && (LW == 64 || LW == 128)), "linear pixel walk");
    static constexpr int TH = kSkBN / TW;
    static constexpr int TWLOG = TW == 128 ? 7 : (TW == 32 ? 5 : 4);
    static constexpr int PAD = (TAPS == 9 && !SUB) ? DIL : 0;
    static constexpr int PADL = (TAPS == 9 && !SUB) ? 4 : 0;
    static constexpr int RMAX = LW == 64 ? 5 : 4;           // rows that 128 consecutive pixels touch when W >= 32 / W >= 57
    static constexpr int PH = LW ? RMAX + 2 * PAD : (SUB ? TH + TR - 1 : (TH - 1) * STRIDE + 1 + 2 * PAD);
    static constexpr int PWL = LW ? LW : (SUB ? (TW + TC - 1 + 3) & ~3 : ((TW - 1) * STRIDE + 1 + 2 * PADL + 3) & ~3);
    static constexpr int CS = PH * PWL;
    static constexpr int NXS = (CK * CS / 4 + kSkThreads - 1) / kSkThreads;
    static constexpr int toff(int tap) { return ((tap / TC) * PWL + (tap % TC)) * DIL; }
};

template <int I, int N, typename F>
__device__ __forceinline__ void sk_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sk_static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void sk_store_sc1(float* p, v4f v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// Position of the prefetch stream in the iteration space; advanced by one chunk at a time (no divisions in the loop).
// Work of workgroup g as a sequence of "virtual" iterations v = 0 .. V - 1:
//   v <  rdp * nch : round r = v / nch, WHOLE tile r * P + g, chunk v % nch.  In a round the 32 workgroups of an XCD hold 32
//                    consecutive tiles (M tile fastest), i.e. a 2-D block of pixel tiles x M tiles, and walk K in step: every weight
//                    image and every input patch they read is shared in the XCD's L2 while it is hot (contiguous tile runs per
//                    workgroup had L2 hit rates of 11-42 % and global-load latencies of ~5 us under load);
//   v >= rdp * nch : the remaining ntiles - rdp * P tiles, their iterations dealt to the workgroups in equal contiguous runs
//                    (stream-K: iteration sk0(g) + (v - rdp * nch) of that space).
// The chunk's addresses travel with the cursor (a step inside a tile is three additions; they are recomputed from the indices only
// when the tile changes): the per-iteration scalar code of the multiply loop is not hidden behind anything -- a wave issues in
// order -- and the full recomputation was ~100 scalar instructions (64-bit multiplies, SGPR spills) per wave and chunk.
struct SkCursor {
    int v, chunk, mt, n, tyi, txi;
    int seg;                    // iterations up to the end of the contiguous piece (sk_decode)
    bool moved;                 // the pixel tile changed since the patch offsets were computed (sk_xoffsets clears it)
    const float* wptr;          // the weight image of (mt, chunk)
    const float* xptr;          // x of (picture n, first channel of the chunk)
    int kleft;                  // channels from the chunk's first to the tensor's last
    long long rem;              // elements from xptr to the end of the tensor
};

// Order of a workgroup's iterations (sk0, sk_len: its run of the stream-K iteration space):
//   1. the LEADING segment of its run, if the run starts inside a tile: it is a contributor there -- done first, its slot is
//      published early;
//   2. its whole tiles;
//   3. the rest of its run, whose last piece may be the first chunks of a tile it FINISHES: by then the contributors of that tile
//      (their step 1) have long published, so the finisher adds their slots without waiting.  (With the run as one block behind
//      the whole tiles, both sides of a split tile finished at the same moment and the hand-off -- slot write, flag, slot read,
//      epilogue -- was a serial tail of 10-19 us per launch: tools/sk_stamps.py.)
// seg = iterations from v to the end of the contiguous piece (a tile's end or a jump of the order).
__device__ __forceinline__ void sk_decode(const SkP& p, int g, int sk0, int sk_len, int v, int& tile, int& chunk, int& seg) {
    const int vdp = p.rdp * p.nch;
    const int c0 = sk0 % p.nch;
    const int lead = (c0 != 0 && sk_len > 0) ? (sk_len < p.nch - c0 ? sk_len : p.nch - c0) : 0;
    int w;
    if (v < lead) {
        w = sk0 + v;
        seg = lead - v;
    } else if (v < lead + vdp) {
        const int u = v - lead;
        const int r = u / p.nch;
        chunk = u - r * p.nch;
        tile = r * p.P + g;
        seg = p.nch - chunk;
        return;
    } else {
        w = sk0 + (v - vdp);
        seg = vdp + sk_len - v;
    }
    const int t = w / p.nch;
    chunk = w - t * p.nch;
    tile = p.rdp * p.P + t;
    seg = seg < p.nch - chunk ? seg : p.nch - chunk;
}

__device__ __forceinline__ void sk_locate(const SkP& p, int g, int sk0, int sk_len, SkCursor& c) {
    int tile;
    sk_decode(p, g, sk0, sk_len, c.v, tile, c.chunk, c.seg);
    c.mt = tile % p.mtiles;
    const int pt = tile / p.mtiles;
    const int tpi = p.tiles_x * p.tiles_y;
    c.n = pt / tpi;
    const int trem = pt - c.n * tpi;
    c.tyi = trem / p.tiles_x;
    c.txi = trem - c.tyi * p.tiles_x;
    c.moved = true;
    const long long HW = (long long)p.H * p.W;
    const int k0 = c.chunk * p.ck;
    c.wptr = p.w + ((size_t)c.mt * p.nch + c.chunk) * (size_t)p.img;
    c.xptr = p.x + ((long long)c.n * p.K + k0) * HW;
    c.kleft = p.K - k0;
    c.rem = ((long long)(p.N - c.n) * p.K - k0) * HW;
}

__device__ __forceinline__ SkCursor sk_cursor(const SkP& p, int g, int sk0, int sk_len, int v) {
    SkCursor c;
    c.v = v;
    sk_locate(p, g, sk0, sk_len, c);
    return c;
}

__device__ __forceinline__ void sk_advance(const SkP& p, int g, int sk0, int sk_len, SkCursor& c) {
    ++c.v;
    if (--c.seg == 0) {
        sk_locate(p, g, sk0, sk_len, c);                    // next piece of the workgroup's list (once per tile)
    } else {
        ++c.chunk;
        c.wptr += p.img;
        c.xptr += p.xstep;
        c.kleft -= p.ck;
        c.rem -= p.xstep;
    }
}

// Per-thread description of its 16-byte groups of the input patch (depends on the thread only): channel, patch row / column
// (packed); -1 = no group.  A group is four consecutive patch columns: one 16-byte load, one 16-byte LDS store; group f = tid +
// 512 j lives at float 4 f of the patch image [CK][PH][PWL] (an affine LDS address: one register + immediates).
// The staging code of a chunk contains NO vector ALU work on aligned planes: while one wave of a SIMD issues its 64 MFMAs back to
// back, an instruction of its partner that needs the vector ALU is issued about once per MFMA (in-kernel stamps: the ~80
// instructions of "stage + refetch" stretched over the partner's whole multiply phase, 5.4k of the 11.2k cycles of an iteration).
// Hence: loads through a buffer resource (scalar base and size per chunk) at per-group byte offsets that are computed once per pixel
// tile -- a group outside the plane gets an offset beyond the resource and comes back as zeros, no exec masking, no selects;
// channels past the tensor's last (K tail) are outside the resource by its size.
// VEC: the rows of the plane are 16-byte aligned (W % 4 == 0, aligned base) and a group is inside the plane's row or outside it.
// !VEC (the 769-crop planes 385 / 193 / 97 / 49: every row starts at another alignment): the same groups, loaded at 4-byte aligned
// addresses (gfx950 runs 16-byte loads there at the aligned rate: tools/micro/unaligned.hip); the one group per row that
// straddles the row's right end reads on into the next row and has its elements past the end zeroed -- when it is written to LDS,
// not behind the load: any use of a loaded value in the fetch code makes hipcc wait for the load there, which serialises the
// prefetch (46.1 ms per step) -- and within three elements of the tensor's end such a group is loaded so that it ENDS at the
// tensor's end and is rotated into place before the LDS store.  (First forms of this path: element-wise 4-byte loads, 4x the
// load instructions, 48.4 ms per step at the 769 crop; aligned groups of memory written to LDS with four predicated 4-byte stores
// at the row's shift, 44.2 ms.)
typedef float v4fu __attribute__((ext_vector_type(4), aligned(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr unsigned kSkNoGroup = 0xfffffff0u;        // byte offset of a group that does not exist: beyond every resource

template <typename G>
struct SkX {
    static constexpr int F4R = G::PWL >> 2;                                   // groups per patch row
    static constexpr int F4C = G::PH * F4R;                                   // per channel
    static constexpr int NXS = (G::CK * F4C + kSkThreads - 1) / kSkThreads;   // per thread
    static_assert(F4C * 4 == G::CS, "patch image: group f at float 4 f");
};

template <int NXS>
struct SkSlots {
    int c[NXS], rc[NXS];
};

template <typename G>
__device__ __forceinline__ void sk_slots(int tid, SkSlots<SkX<G>::NXS>& s) {
    constexpr int f4r = SkX<G>::F4R, f4c = SkX<G>::F4C;
#pragma unroll
    for (int j = 0; j < SkX<G>::NXS; ++j) {
        const int f = tid + j * kSkThreads;
        const int c = f / f4c, rem = f - c * f4c;
        const int row = rem / f4r, col = (rem - row * f4r) * 4;
        const bool live = f < G::CK * f4c;
        s.c[j] = live ? c : -1;
        s.rc[j] = row | (col << 16);
    }
}

// Byte offsets of a thread's patch groups from the (picture, chunk) base of the input: functions of the pixel tile only,
// recomputed when the prefetch stream enters a new one (in the part of an iteration that all waves run).
// !VEC: nv = number of elements of the group inside its row (1..4).
template <int NXS>
struct SkXOff {
    unsigned voff[NXS];
    int nv[NXS];
};

template <typename G, bool VEC>
__device__ __forceinline__ void sk_xoffsets(const SkP& p, SkCursor& cur, const SkSlots<SkX<G>::NXS>& sl, SkXOff<SkX<G>::NXS>& xo) {
    // (linear walk: the patch holds full rows from the first row the tile touches)
    const int iy0 = G::LW ? (cur.txi * kSkBN) / p.W - G::PAD : cur.tyi * G::TH * G::STRIDE - G::PAD;
    const int ix0 = G::LW ? -G::PADL : cur.txi * G::TW * G::STRIDE - G::PADL;
    const int HW = p.H * p.W;
#pragma unroll
    for (int j = 0; j < SkX<G>::NXS; ++j) {
        const int c = sl.c[j];
        const int iy = iy0 + (sl.rc[j] & 0xffff), ix = ix0 + (sl.rc[j] >> 16);        // (ix is a multiple of 4: never astride the left end)
        const bool ok = c >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        xo.voff[j] = ok ? (unsigned)(c * HW + iy * p.W + ix) * 4u : kSkNoGroup;
        if constexpr (!VEC) xo.nv[j] = p.W - ix < 4 ? p.W - ix : 4;
    }
    cur.moved = false;
}

// Where the chunk at the cursor lives: bases and limits that are the same for the whole workgroup.  Computed in code that ALL
// waves run (the top of an iteration), never inside the `wave < 4` phases: hipcc compiles that branch as exec-mask divergence and
// keeps everything assigned inside it in VECTOR registers -- the cursor with its divisions (~250 VALU instructions at a tile
// change) and the 64-bit address arithmetic of every load (~40 per chunk) then run on the VALU, whose issue the partner wave's
// MFMAs hold.  (Turning the branch into a scalar one with readfirstlane made the staging registers conditionally assigned loop
// variables: register moves, spills and a vmcnt(0) at the loop head -- 37.2 vs 33.3 ms per step.)
struct SkPlan {
    __amdgpu_buffer_rsrc_t wres;    // the weight image of (M tile, chunk)
    __amdgpu_buffer_rsrc_t xres;    // x from (picture, first channel of the chunk) to the end of the chunk's channels (or of the tensor)
    unsigned size;              // its size in bytes
    bool slow;                  // !VEC: a group of this chunk may lie within three elements of the tensor's end
};

// valid == false (the cursor has run past the workgroup's last iteration): both resources are empty, every load of the chunk reads
// zeros -- the multiply loop then needs no branch around its loads.
template <typename G, int TAPS, int WM, bool VEC>
__device__ __forceinline__ SkPlan sk_plan_fetch(const SkP& p, SkCursor& cur, const SkSlots<SkX<G>::NXS>& sl, SkXOff<SkX<G>::NXS>& xo,
                                                bool valid = true) {
    constexpr int KC = TAPS * G::CK;
    const int HW = p.H * p.W;
    SkPlan f;
    f.wres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cur.wptr), 0, valid ? KC * 32 * WM * 4 : 0, 0x00020000);
    const float* xb = cur.xptr;
    const int kleft = cur.kleft;
    // the chunk's channels: groups of channels past the tensor's last (K tail) start at or beyond this size and read as zeros
    const unsigned long long chunk_bytes = (unsigned long long)(kleft < G::CK ? kleft : G::CK) * HW * 4ull;
    if constexpr (VEC) {
        f.size = (unsigned)chunk_bytes;             // (aligned groups end inside their row: none crosses the size)
        f.slow = false;
    } else {
        // a group astride the end of a row reads up to three elements of what follows: allowed up to the end of the tensor
        const unsigned long long rem_bytes = (unsigned long long)cur.rem * 4ull;
        const bool full = kleft >= G::CK;
        // (K tail: the size must end with the last channel, so its last row's astride group is loaded early like the tensor's)
        f.slow = !full || rem_bytes < chunk_bytes + 16;
        const unsigned long long sz = full ? (rem_bytes < chunk_bytes + 12 ? rem_bytes : chunk_bytes + 12) : chunk_bytes;
        f.size = (unsigned)sz;
    }
    if (!valid) f.size = 0;
    f.xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)f.size, 0x00020000);
    if (valid && cur.moved) sk_xoffsets<G, VEC>(p, cur, sl, xo);
    return f;
}

// global -> registers of the planned chunk (its tile may differ from the one being multiplied: the pipeline crosses tile
// boundaries).  !VEC: xm[j] = what sk_stage needs to know about group j (elements inside the row | elements loaded early << 3).
template <typename G, int TAPS, int WM, bool VEC, int NWS>
__device__ __forceinline__ void sk_fetch(const SkPlan& f, const SkXOff<SkX<G>::NXS>& xo, int tid, v4f (&wr)[NWS], v4f (&xr)[SkX<G>::NXS],
                                         int (&xm)[VEC ? 1 : SkX<G>::NXS]) {
    constexpr int BM = 32 * WM, NXS = SkX<G>::NXS;
    constexpr int KC = TAPS * G::CK;
    static_assert(BM * KC % 4 == 0, "weight image in 16-byte groups");
#pragma unroll
    for (int j = 0; j < NWS; ++j)       // (a group past the image -- only in a last, partial slot -- is outside the resource: zeros)
        wr[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(f.wres, tid * 16, j * kSkThreads * 16, 0));
    if (VEC || !f.slow) {
#pragma unroll
        for (int j = 0; j < NXS; ++j) {
            xr[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(f.xres, (int)xo.voff[j], 0, 0));
            if constexpr (!VEC && TAPS != 1) xm[j] = xo.nv[j];          // (1x1: see sk_stage)
        }
    } else {
        // (no use of the loaded values here: masks and the rotation are applied by sk_stage)
#pragma unroll
        for (int j = 0; j < NXS; ++j) {
            const unsigned v = xo.voff[j];
            const bool ok = v < f.size;                                         // (false for kSkNoGroup and for the K tail)
            const unsigned avail = ok ? (f.size - v) >> 2 : 4u;                  // elements from the group's first to the end of the resource
            const int nv = xo.nv[j] < (int)avail ? xo.nv[j] : (int)avail;        // (K tail: what follows the last channel is not the row's)
            const unsigned early = avail < 4u ? 4u - avail : 0u;
            if constexpr (!VEC) xm[j] = nv | (int)(early << 3);
            xr[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(f.xres, (int)(ok ? v - 4u * early : kSkNoGroup), 0, 0));
        }
    }
}

// registers -> LDS.  Weight image [KC / 8][2][BM][4] (written by k_sk_pack): k-step kk = tap * (CK / 2) + c / 2 pairs the channels
// 2 cp + h of one tap (h = lane half of the MFMA), the four k-steps 4 q .. 4 q + 3 of one (half, row) are adjacent: one 16-byte
// read = four MFMAs.
template <typename G, int TAPS, int WM, bool VEC, int NWS>
__device__ __forceinline__ void sk_stage(const int (&xm)[VEC ? 1 : SkX<G>::NXS], float* __restrict__ sW, float* __restrict__ sX, int tid,
                                         const v4f (&wr)[NWS], const v4f (&xr)[SkX<G>::NXS], bool slow) {
    // `slow`: the chunk in the registers was fetched on the slow path (a group near the end of the tensor / of a K tail).  A 1x1
    // product needs no zeros past the end of a row -- what lands in the out-of-plane columns of its patch only reaches outputs that
    // the epilogue does not store -- so on unaligned planes its groups are fixed up (rotated, tail zeroed) only for such chunks:
    // the fix is ~17 VALU instructions per group (9.0 vs 8.1 ms per step for the 1x1 layers of the 769 crop with it).
    constexpr int BM = 32 * WM, CK = G::CK, NXS = SkX<G>::NXS;
    constexpr int KC = TAPS * CK;
    constexpr int step = kSkThreads * 4;
    float* wdst = sW + tid * 4;
    float* xdst = sX + tid * 4;
#pragma unroll
    for (int j = 0; j < NWS; ++j)
        if (NWS * step == BM * KC || tid * 4 + j * step < BM * KC) *reinterpret_cast<v4f*>(wdst + j * step) = wr[j];
#pragma unroll
    for (int j = 0; j < NXS; ++j) {
        v4f v = xr[j];
        if constexpr (!VEC) {
            if (TAPS != 1 || slow) {
                const int nv = xm[j] & 7;
                const int early = xm[j] >> 3;
                if (early) {                    // loaded `early` elements early: element i of the group is u[i + early]
                    const v4f u = v;
                    v[0] = early == 1 ? u[1] : (early == 2 ? u[2] : u[3]);
                    v[1] = early == 1 ? u[2] : u[3];
                    v[2] = u[3];
                }
                v[1] = nv > 1 ? v[1] : 0.0f;
                v[2] = nv > 2 ? v[2] : 0.0f;
                v[3] = nv > 3 ? v[3] : 0.0f;
            }
        }
        if (NXS * step == CK * G::CS || tid * 4 + j * step < CK * G::CS) *reinterpret_cast<v4f*>(xdst + j * step) = v;
    }
}

// One slot of the staging registers (S < NWS: weight group S, else patch group S - NWS): its LDS store and its refill, issued
// in the shadow of the wave's own MFMAs (see k_conv_sk).
template <typename G, int TAPS, int WM, bool VEC, int NWS, int S>
__device__ __forceinline__ void sk_stage_slot(const int (&xm)[VEC ? 1 : SkX<G>::NXS], float* __restrict__ sW, float* __restrict__ sX, int tid,
                                              const v4f (&wr)[NWS], const v4f (&xr)[SkX<G>::NXS], bool slow) {
    constexpr int BM = 32 * WM, CK = G::CK, NXS = SkX<G>::NXS;
    constexpr int KC = TAPS * CK;
    constexpr int step = kSkThreads * 4;
    if constexpr (S < NWS) {
        if (NWS * step == BM * KC || tid * 4 + S * step < BM * KC) *reinterpret_cast<v4f*>(sW + tid * 4 + S * step) = wr[S];
    } else {
        constexpr int j = S - NWS;
        v4f v = xr[j];
        if constexpr (!VEC) {
            if (TAPS != 1 || slow) {            // (see sk_stage)
                const int nv = xm[j] & 7;
                const int early = xm[j] >> 3;
                if (early) {
                    const v4f u = v;
                    v[0] = early == 1 ? u[1] : (early == 2 ? u[2] : u[3]);
                    v[1] = early == 1 ? u[2] : u[3];
                    v[2] = u[3];
                }
                v[1] = nv > 1 ? v[1] : 0.0f;
                v[2] = nv > 2 ? v[2] : 0.0f;
                v[3] = nv > 3 ? v[3] : 0.0f;
            }
        }
        if (NXS * step == CK * G::CS || tid * 4 + j * step < CK * G::CS) *reinterpret_cast<v4f*>(sX + tid * 4 + j * step) = v;
    }
}

template <typename G, int TAPS, int WM, bool VEC, int NWS, int S>
__device__ __forceinline__ void sk_fetch_slot(const SkPlan& f, const SkXOff<SkX<G>::NXS>& xo, int tid, v4f (&wr)[NWS], v4f (&xr)[SkX<G>::NXS],
                                              int (&xm)[VEC ? 1 : SkX<G>::NXS]) {
    if constexpr (S < NWS) {
        wr[S] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(f.wres, tid * 16, S * kSkThreads * 16, 0));
    } else {
        constexpr int j = S - NWS;
        if (VEC || !f.slow) {
            xr[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(f.xres, (int)xo.voff[j], 0, 0));
            if constexpr (!VEC && TAPS != 1) xm[j] = xo.nv[j];
        } else {
            const unsigned v = xo.voff[j];
            const bool ok = v < f.size;
            const unsigned avail = ok ? (f.size - v) >> 2 : 4u;
            const int nv = xo.nv[j] < (int)avail ? xo.nv[j] : (int)avail;
            const unsigned early = avail < 4u ? 4u - avail : 0u;
            if constexpr (!VEC) xm[j] = nv | (int)(early << 3);
            xr[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(f.xres, (int)(ok ? v - 4u * early : kSkNoGroup), 0, 0));
        }
    }
}

template <int TAPS, int CK, int WM, int TW, int STRIDE, int DIL, bool VEC, int NB, int SUB = 0>
__global__ __launch_bounds__(kSkThreads) void k_conv_sk(const SkP p) {
    using G = SkG<TAPS, CK, TW, STRIDE, DIL, SUB>;
    constexpr int NXS = SkX<G>::NXS, CS = G::CS, PWL = G::PWL;
    constexpr bool DMA = NB >= 2;
    constexpr int BM = 32 * WM, BN = kSkBN;
    constexpr int NG = 8 / WM;                  // pixel groups of the 8 waves
    constexpr int TN = BN / NG / 32;            // 32-pixel accumulator tiles per wave: 2 (BM 128) / 1 (BM 64)
    constexpr int KC = TAPS * CK;
    constexpr int NQ = KC / 8;                  // groups of four k-steps per chunk
#ifdef SK_PHASED
    constexpr int QSPLIT = NQ;      // waves 0-3 stage after all their MFMA groups,
    constexpr int QEARLY = 0;       // waves 4-7 before theirs
#endif
    constexpr int NWS = (BM * KC / 4 + kSkThreads - 1) / kSkThreads;
    static_assert(KC % 8 == 0 && TN >= 1, "tile");
    extern __shared__ __attribute__((aligned(16))) float sk_smem[];
    // [NBUF][sW: KC * BM | sX: CK * CS (DMA: rounded up to 1 KB pieces)] then sE [2 tile parities][2][BM], then 1 KB DMA dump, then the give-up word
    constexpr int NBUF = DMA ? NB : 2;
    constexpr int XF = DMA ? ((CK * CS + 255) / 256) * 256 : CK * CS;
    constexpr int bufsz = KC * BM + XF;
    float* sEbase = sk_smem + NBUF * bufsz;
    float* sDump = sEbase + 4 * BM;
    int tile_parity = 0;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int mtw = wave % WM, ng = wave / WM;
    // logical workgroup index: the workgroups of one XCD (blocks b, b + 8, ...) own one contiguous run of iterations
    const int P = p.P;
    const int g = (P % 8 == 0) ? (int)(blockIdx.x & 7) * (P >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    // stream-K run of this workgroup; MAS_SK_NOSPLIT: the remaining tiles whole, one each to the first workgroups
    const int sk_rem = p.sk_iters / p.nch;
    const int sk0 = p.nosplit ? (g < sk_rem ? g : sk_rem) * p.nch : (int)((long long)p.sk_iters * g / P);
    const int sk1 = p.nosplit ? (g + 1 < sk_rem ? g + 1 : sk_rem) * p.nch : (int)((long long)p.sk_iters * (g + 1) / P);
    const int vdp = p.rdp * p.nch;
    const int it0 = 0, it1 = vdp + (sk1 - sk0);             // virtual iterations of this workgroup
    if (it0 >= it1) return;
    if (p.stamps && tid == 0) p.stamps[4 * g] = wall_clock64();

    const int aBase = (h * BM + mtw * 32 + l31) * 4;
    int bBase[TN], pl[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        pl[tn] = ng * (BN / NG) + tn * 32 + l31;
        const int ty = pl[tn] >> G::TWLOG, tx = pl[tn] & (TW - 1);
        bBase[tn] = h * CS + ty * STRIDE * PWL + tx * STRIDE + G::PADL - G::PAD;
    }
    f32x16 acc[TN];
    v4f wr[DMA ? 1 : NWS], xr[DMA ? 1 : NXS];
    const int HWo = p.HWy;

    // The LDS operands of a group of four k-steps (one 16-byte A read + 4 * TN B reads) are requested one whole group ahead of
    // the MFMAs that consume them: the two waves of a SIMD run the same phase, so nothing else hides an LDS round trip.
    struct Group {
        v4f a;
        float b[4][TN];
    };
    auto load_group_of = [&](int buf, int q, Group& o) {
        const float* sW = sk_smem + buf * bufsz;
        const float* sX = sW + KC * BM;
        o.a = *reinterpret_cast<const v4f*>(sW + q * 8 * BM + aBase);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = 4 * q + j, tap = kk / (CK / 2), cp = kk - tap * (CK / 2);
            const int toff = G::toff(tap);
            const float* xrow = sX + 2 * cp * CS + toff;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) o.b[j][tn] = xrow[bBase[tn]];
        }
    };
    // `pre`: the operands of group qlo, already requested by the caller (the register-staged loop asks for them right behind the
    // barrier, BEFORE its scalar planning of chunk it + 2: the LDS round trip then runs beside that code instead of behind it)
    auto mfma_part = [&](int buf, auto QLc, auto QHc, auto&& hook, const Group* pre = nullptr) {
        constexpr int qlo = decltype(QLc)::value, qhi = decltype(QHc)::value;
        auto load_group = [&](int q, Group& o) { load_group_of(buf, q, o); };
        if constexpr (qlo >= qhi) return;
        Group cur, nxt;
        if (pre) cur = *pre;
        else load_group(qlo, cur);
        sk_static_for<qlo, qhi>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if (q + 1 < qhi) load_group(q + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);          // the next group's reads stay in front of this group's MFMAs
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[j], cur.b[j][tn], acc[tn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            hook(qc);                                   // staging work that rides in the shadow of this group's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < qhi) cur = nxt;
        });
    };
    auto no_hook = [](auto) {};

    int buf = 0;
    // ---- LDS-DMA pipeline (NB >= 2) ------------------------------------------------------------------------------------------
    // per wave and chunk: PWW pieces of the weight image + PXW pieces of the input patch, every wave the same count (pieces
    // past the end go zero page -> dump area), so one counted s_waitcnt retires exactly one chunk
    constexpr int WPIECES = KC * BM / 256;                      // 1 KB pieces of the weight image
    constexpr int PWW = (WPIECES + 7) / 8;
    constexpr int XPIECES = VEC ? XF / 256 : XF / 64;           // 1 KB (16-byte lanes) or 256-byte (4-byte lanes) pieces
    constexpr int PXW = (XPIECES + 7) / 8;
    constexpr int PT = PWW + PXW;
    constexpr int DEPTH = DMA ? NB - 1 : 1;                     // chunks in flight
    static_assert(!DMA || (DEPTH - 1) * PT <= 63, "vmcnt");
    SkCursor pre = sk_cursor(p, g, sk0, sk1 - sk0, it0);
    int pre_it = it0;
    auto dma_chunk = [&](int target, int nb) {
        if constexpr (DMA) {
            const bool real = target < it1;
            if (real)
                while (pre_it < target) {
                    sk_advance(p, g, sk0, sk1 - sk0, pre);
                    ++pre_it;
                }
            float* bW = sk_smem + nb * bufsz;
            float* bX = bW + KC * BM;
            const float* wb = p.w + ((size_t)pre.mt * p.nch + pre.chunk) * (size_t)(KC * BM);
#pragma unroll
            for (int j = 0; j < PWW; ++j) {
                const int piece = wave + 8 * j;                              // wave-uniform
                const bool ok = real && piece < WPIECES;
                const float* g = ok ? wb + piece * 256 + lane * 4 : p.zero;
                float* l = ok ? bW + piece * 256 : sDump;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
            }
            const int k0 = pre.chunk * CK;
            const int iy0 = pre.tyi * G::TH * STRIDE - G::PAD, ix0 = pre.txi * TW * STRIDE - G::PADL;
            const int HW = p.H * p.W;
            const float* xb = p.x + ((size_t)pre.n * p.K + k0) * HW;
#pragma unroll
            for (int j = 0; j < PXW; ++j) {
                const int piece = wave + 8 * j;
                const bool pok = real && piece < XPIECES;
                if constexpr (VEC) {
                    constexpr int f4r = PWL >> 2, f4c = G::PH * f4r;
                    const int f = piece * 64 + lane;
                    const int c = f / f4c, rem = f - c * f4c;
                    const int row = rem / f4r, col = (rem - row * f4r) * 4;
                    const int iy = iy0 + row, ix = ix0 + col;
                    const bool ok = pok && f < CK * f4c && k0 + c < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    const float* g = ok ? xb + (size_t)c * HW + (long long)iy * p.W + ix : p.zero;
                    float* l = pok ? bX + piece * 256 : sDump;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
                } else {
                    const int f = piece * 64 + lane;                         // element of the patch image [CK][PH][PWL]
                    const int c = f / CS, rem = f - c * CS;
                    const int row = rem / PWL, col = rem - row * PWL;
                    const int iy = iy0 + row, ix = ix0 + col;
                    const bool ok = pok && f < CK * CS && k0 + c < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    const float* g = ok ? xb + (size_t)c * HW + (long long)iy * p.W + ix : p.zero;
                    float* l = pok ? bX + piece * 64 : sDump;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 4, 0, 0);
                }
            }
        }
    };
    // wait until all but the (DEPTH - 1) youngest chunks of this wave have landed, then meet the other waves
    auto dma_publish = [&]() {
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PT) : "memory");
            __syncthreads();
        }
    };
    // ---- register-staged pipeline (NB == 0) -------------------------------------------------------------------------------------
    // Every slot of the staging registers that carried chunk it + 1 to LDS is refilled with chunk it + 2 right behind its store
    // (see `iteration`): a load has a whole iteration to arrive.  (One chunk in flight = 64 KB of staging registers per CU for the
    // 1x1 kernel.  Two register sets spilled 45 VGPRs at the 256-register budget of two waves per SIMD.)
    SkSlots<NXS> slots;
    SkXOff<NXS> xoff;
    int xm[VEC ? 1 : NXS];
    bool slow_regs = false;                 // the chunk in the staging registers was fetched on the slow path (see sk_stage)
    if constexpr (!DMA) {
        sk_slots<G>(tid, slots);
        // (requesting chunks it0 and it0 + 1 back to back -- one memory round trip in front of the first MFMA instead of two -- was
        // measured: no difference, 32.6 vs 32.5 ms per step)
        SkPlan f0 = sk_plan_fetch<G, TAPS, WM, VEC>(p, pre, slots, xoff);
        sk_fetch<G, TAPS, WM, VEC, NWS>(f0, xoff, tid, wr, xr, xm);
        sk_stage<G, TAPS, WM, VEC, NWS>(xm, sk_smem, sk_smem + KC * BM, tid, wr, xr, f0.slow);
        if (it0 + 1 < it1) {
            sk_advance(p, g, sk0, sk1 - sk0, pre);
            f0 = sk_plan_fetch<G, TAPS, WM, VEC>(p, pre, slots, xoff);
            sk_fetch<G, TAPS, WM, VEC, NWS>(f0, xoff, tid, wr, xr, xm);
        }
        slow_regs = f0.slow;
        __syncthreads();
    } else {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) dma_chunk(it0 + d, d);
        dma_publish();                          // chunk it0 has landed everywhere
    }

    if (p.stamps && tid == 0) p.stamps[4 * g + 1] = wall_clock64();
#ifdef SK_PHASE_STAMPS
    unsigned long long t0_phase = __builtin_readcyclecounter();
#endif
    auto iteration = [&](int it) {
        if constexpr (DMA) {
            // the buffer that held chunk it - 1 (all its readers are behind the last barrier) receives chunk it + DEPTH
            int nb = buf + DEPTH;
            if (nb >= NBUF) nb -= NBUF;
            dma_chunk(it + DEPTH, nb);
            mfma_part(buf, std::integral_constant<int, 0>{}, std::integral_constant<int, NQ>{}, no_hook);
            dma_publish();                      // chunk it + 1 has landed; (DEPTH - 1) younger ones stay in flight
            buf = buf + 1 == NBUF ? 0 : buf + 1;
        } else {
#ifndef SK_PHASED
            // In-kernel stamps (tools/sk_phases.py on the -DSK_PHASED build): while one wave of a SIMD issues its MFMAs back to back, its
            // partner on the same SIMD is starved -- ANY instruction of it, scalar or memory, with or without raised priority, is issued
            // about once per two or three MFMAs (an LDS-store + load burst of 16 instructions: 560 cycles when the partner also stages,
            // 3.8-4.5k cycles beside the partner's multiply phase).  Work of one wave cannot hide behind the other wave's MFMAs; it hides
            // behind its OWN: every wave runs the same program, and slot s of the staging registers -- the LDS store of chunk it + 1,
            // then the refill with chunk it + 2 -- rides behind MFMA group q of the wave itself.  The loop body has no branch (a chunk
            // past the end is an empty resource), so hipcc counts the outstanding loads across the back edge: the store of slot s
            // waits for vmcnt(NS - 1), i.e. for a load that is a whole iteration old.
            Group first;
            load_group_of(buf, 0, first);
            __builtin_amdgcn_sched_barrier(0);
            const bool v2 = it + 2 < it1;
            if (v2) sk_advance(p, g, sk0, sk1 - sk0, pre);
            const SkPlan f2 = sk_plan_fetch<G, TAPS, WM, VEC>(p, pre, slots, xoff, v2);
            float* nW = sk_smem + (buf ^ 1) * bufsz;
            float* nX = nW + KC * BM;
            constexpr int NS = NWS + NXS;
            mfma_part(buf, std::integral_constant<int, 0>{}, std::integral_constant<int, NQ>{}, [&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int s0 = q * NS / NQ, s1 = (q + 1) * NS / NQ;
                sk_static_for<s0, s1>([&](auto sc) {
                    constexpr int S = decltype(sc)::value;
                    sk_stage_slot<G, TAPS, WM, VEC, NWS, S>(xm, nW, nX, tid, wr, xr, slow_regs);
                    sk_fetch_slot<G, TAPS, WM, VEC, NWS, S>(f2, xoff, tid, wr, xr, xm);
                });
            }, &first);
            slow_regs = f2.slow;
            __syncthreads();
            buf ^= 1;
#else
            // (the phased form, kept for tools/sk_phases.py: the measurement that led to the interleaved one)
            // The two waves of a SIMD are COMPLEMENTARY: waves 0-3 (the older ones: the matrix pipe serves them first, the
            // younger wave gets almost nothing meanwhile) multiply first and stage + refetch afterwards; waves 4-7 stage + refetch
            // first and multiply afterwards, so each wave's memory phase sits beside its partner's MFMAs.  With both staging at 3/4
            // the older wave ran ahead, staged, finished and idled ~2.8k cycles at the barrier while the younger one staged alone
            // (in-kernel stamps: 11.6k cycles per iteration for 8.2k of MFMAs).
#ifdef SK_PHASE_STAMPS
            unsigned long long t1 = 0, t1b = 0, t2 = 0;
#endif
            // the part every wave runs: where chunk it + 2 lives (scalar registers, see SkPlan)
            SkPlan f2;
            if (it + 2 < it1) {
                sk_advance(p, g, sk0, sk1 - sk0, pre);
                f2 = sk_plan_fetch<G, TAPS, WM, VEC>(p, pre, slots, xoff);
            }
            auto body = [&](auto QS) {
                constexpr int qs = decltype(QS)::value;
                mfma_part(buf, std::integral_constant<int, 0>{}, std::integral_constant<int, qs>{}, no_hook);
#ifdef SK_PHASE_STAMPS
                t1 = __builtin_readcyclecounter();
#endif
                // the memory phase runs at raised priority: beside an older partner that issues MFMAs back to back the younger wave's
                // stores and loads were served only in leftover issue slots (4-5k cycles for ~80 instructions, in-kernel stamps)
                __builtin_amdgcn_s_setprio(3);
#ifdef SK_PHASE_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                t1b = __builtin_readcyclecounter();
#endif
                if (it + 1 < it1) {
                    float* nW = sk_smem + (buf ^ 1) * bufsz;
                    sk_stage<G, TAPS, WM, VEC, NWS>(xm, nW, nW + KC * BM, tid, wr, xr, slow_regs);
#ifdef SK_PHASE_STAMPS
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                    if (it + 2 < it1) {
                        sk_fetch<G, TAPS, WM, VEC, NWS>(f2, xoff, tid, wr, xr, xm);
                        slow_regs = f2.slow;
                    }
                }
                __builtin_amdgcn_s_setprio(0);
#ifdef SK_PHASE_STAMPS
                t2 = __builtin_readcyclecounter();
#endif
                mfma_part(buf, std::integral_constant<int, qs>{}, std::integral_constant<int, NQ>{}, no_hook);
            };
#ifdef SK_BOTH_EARLY
            body(std::integral_constant<int, QEARLY>{});
#else
            if (wave < 4) body(std::integral_constant<int, QSPLIT>{});
            else body(std::integral_constant<int, QEARLY>{});
#endif
#ifdef SK_PHASE_STAMPS
            const unsigned long long t3 = __builtin_readcyclecounter();
            __syncthreads();
            const unsigned long long t4 = __builtin_readcyclecounter();
            if (p.stamps && lane == 0 && g == 7) {
                unsigned long long* ph = p.stamps + 2048 + wave * 8;
                ph[0] += t1 - t0_phase; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += 1; ph[5] += t1b - t1;
            }
            t0_phase = t4;
            buf ^= 1;
            return;
#endif
            __syncthreads();
            buf ^= 1;
#endif
        }
    };

    // the M tiles with one stream-K tile fewer than the others leave their last slot without a writer: workgroup 0 zeroes it
    if (p.stats && p.stats_acc && g == 0) {
        const int sk_tiles = p.sk_iters / p.nch, rem = sk_tiles % p.mtiles;
        if (rem != 0) {
            const int slot = p.stats_w + sk_tiles / p.mtiles;
            for (int e = tid; e < (p.mtiles - rem) * BM * NG; e += kSkThreads) {
                const int m = rem * BM + e / NG;
                if (m < p.M) p.stats[(size_t)m * p.stats_pitch + (size_t)slot * NG + e % NG] = make_double2(0.0, 0.0);
            }
        }
    }
    // BatchNorm partials: per accumulator row of this lane, over the pixels of the tiles finished so far in the current slot
    float rsum[16], rsq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rsum[r] = rsq[r] = 0.0f;
    int run_slot = -1, run_m0 = 0;
    // fold the 32 lanes of a half-wave (a fixed order) and store the entry of (row, run_slot, ng)
    auto stats_flush = [&]() {
        auto fold = [&](float (&a)[16]) -> float {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool up = lane & 1;
                const float send = up ? a[i] : a[i + 8], keep = up ? a[i + 8] : a[i];
                a[i] = keep + __shfl_xor(send, 1, 64);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool up = lane & 2;
                const float send = up ? a[i] : a[i + 4], keep = up ? a[i + 4] : a[i];
                a[i] = keep + __shfl_xor(send, 2, 64);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool up = lane & 4;
                const float send = up ? a[i] : a[i + 2], keep = up ? a[i + 2] : a[i];
                a[i] = keep + __shfl_xor(send, 4, 64);
            }
            {
                const bool up = lane & 8;
                const float send = up ? a[0] : a[1], keep = up ? a[1] : a[0];
                a[0] = keep + __shfl_xor(send, 8, 64);
            }
            return a[0] + __shfl_xor(a[0], 16, 64);
        };
        const float S = fold(rsum), Q = fold(rsq);
        const int r = 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
        const int m = mtw * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
        if (!(lane & 16) && run_m0 + m < p.M)
            p.stats[(size_t)(run_m0 + m) * p.stats_pitch + (size_t)run_slot * NG + ng] = make_double2((double)S, (double)Q);
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) rsum[r2] = rsq[r2] = 0.0f;
    };

    int it = it0;
    while (it < it1) {
        int tile, c0, seg;
        sk_decode(p, g, sk0, sk1 - sk0, it, tile, c0, seg);
        const int c1 = c0 + seg;                                        // (whole tiles: c0 = 0, c1 = nch)
        const int mt = tile % p.mtiles, pt = tile / p.mtiles;
        const int m0 = mt * BM;
        if constexpr (G::LW != 0) {
            // linear walk: where this tile's pixels sit in the full-row patch
            const int pix0 = (pt % p.tiles_x) * kSkBN, y0 = pix0 / p.W, last = p.H * p.W - 1;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int pp = pix0 + pl[tn] < last ? pix0 + pl[tn] : last;
                const int y = pp / p.W, x = pp - y * p.W;
                bBase[tn] = h * CS + (y - y0) * PWL + x + G::PADL - G::PAD;
            }
        }
        // epilogue constants of this tile; the buffer alternates per tile: slower waves may still read the previous tile's
        float* sE = sEbase + tile_parity * 2 * BM;
        tile_parity ^= 1;
        if (tid < BM && (p.scale || p.relu)) {          // (the plain forms of the epilogue never read them)
            const bool real = m0 + tid < p.M;
            sE[tid] = (p.scale && real) ? p.scale[m0 + tid] : 1.0f;
            sE[BM + tid] = (p.scale && real) ? p.shift[m0 + tid] : 0.0f;
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tn][r] = 0.0f;
        for (int c = c0; c < c1; ++c, ++it) iteration(it);
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
            const int tile_end = (tile - p.rdp * P + 1) * p.nch;        // in the iteration space of the stream-K tiles
            int* s_gave_up = reinterpret_cast<int*>(sDump + 256);       // (one word behind the DMA dump area)
            if (tid == 0) *s_gave_up = 0;
            for (int gg = g + 1; gg < P; ++gg) {
                const int b0 = (int)((long long)p.sk_iters * gg / P), b1 = (int)((long long)p.sk_iters * (gg + 1) / P);
                if (b0 >= tile_end) break;
                if (b1 == b0) continue;                     // (a workgroup without stream-K iterations contributes nothing)
                if (wave == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load((gu32*)(p.flags + gg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > p.spin_limit) {
                            // give up: nothing hangs.  The error word says so (the trainer reads it with the loss, mas_conv_sk_error
                            // for everybody else) and the tile is poisoned with NaN below -- a launch that gave up never hands out
                            // plausible numbers.
                            if (lane == 0) {
                                atomicOr(p.flags + 512, 1u);
                                *s_gave_up = 1;
                            }
                            break;
                        }
                    }
                }
                __syncthreads();
                // The slot is read with sc1 loads (they bypass this CU's L1: served by the L2 the write-through stores went to), issued
                // after the barrier the polling wave joined -- no acquire fence: its buffer_inv + wait cost ~1.7 us per contributor at the
                // very end of a finisher's work (MI355X_MICROARCH.md, hand-off table: "sc1 loads may replace the acquire").
                const float* slot = p.slots + (size_t)gg * kSkSlotFloats;
                v4f part[TN * 4];
#pragma unroll
                for (int i = 0; i < TN * 4; ++i)
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(part[i]) : "v"(slot + ((size_t)i * kSkThreads + tid) * 4) : "memory");
#pragma unroll
                for (int i = 0; i < TN * 4; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(part[i])::"memory");
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[tn][4 * r4 + i] += part[tn * 4 + r4][i];
            }
            __syncthreads();
            if (*s_gave_up) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tn][r] = __builtin_nanf("");
            }
        }
        // ---- epilogue: accumulator (row = (r & 3) + 8 (r >> 2) + 4 h, column = lane & 31) -> NCHW -----------------------------
        const int tpi = p.tiles_x * p.tiles_y;
        const int n = pt / tpi, trem = pt - n * tpi;
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int oy0 = tyi * G::TH, ox0 = txi * TW;
        float* yb = p.y + ((size_t)n * p.M + m0) * HWo;
        const float* rb = p.res ? p.res + ((size_t)n * p.M + m0) * HWo : nullptr;
        const float lo = p.relu ? 0.0f : -INFINITY;
        const int mlim = p.M - m0;
        // (all residual values of the tile are requested before the first of them is used: one exposed round trip, not TN)
        const int mb = mtw * 32 + 4 * h;
        int po_[TN];
        bool inside_[TN];
        float rv_[TN][16];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int oy = oy0 + (pl[tn] >> G::TWLOG), ox = ox0 + (pl[tn] & (TW - 1));
            inside_[tn] = oy < p.Ho && ox < p.Wo;
            po_[tn] = inside_[tn] ? (oy * p.os + p.oy_off) * p.Wy + ox * p.os + p.ox_off : 0;
            if (rb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    rv_[tn][r] = __builtin_nontemporal_load(&rb[(size_t)(m < mlim ? m : 0) * HWo + po_[tn]]);     // (read once)
                }
            }
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const bool inside = inside_[tn];
            const int po = po_[tn];
            const float (&rv)[16] = rv_[tn];
            if (inside) {
                // Nothing overlaps the epilogue of a one-workgroup-per-CU kernel, so every instruction in it is exposed: the forms that
                // a training step uses -- the bare product (every forward convolution), the product plus the other consumer's gradient
                // (input gradient of a block input) -- skip the BatchNorm constants in LDS, the ReLU clamp and, on whole M tiles, the
                // row predicates (forward 7.44 -> 7.18 ms, input gradient 7.20 -> 6.93 ms per step for the first of these alone).
                const bool plain = !p.scale && !p.relu, whole = mlim >= BM;
                if (plain && whole) {
                    if (rb) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) yb[(size_t)(mb + (r & 3) + 8 * (r >> 2)) * HWo + po] = acc[tn][r] + rv[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) yb[(size_t)(mb + (r & 3) + 8 * (r >> 2)) * HWo + po] = acc[tn][r];
                    }
                } else if (plain) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mb + (r & 3) + 8 * (r >> 2);
                        if (m < mlim) yb[(size_t)m * HWo + po] = rb ? acc[tn][r] + rv[r] : acc[tn][r];
                    }
                } else {
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
        if (p.stats) {
            // (the host admits statistics only for the bare product: y = acc)
            // BatchNorm partials: the 32 lanes of a half-wave hold the same 16 rows; stats_flush folds them with a halving butterfly
            // (a fixed order: the partials, and the statistics k_bn_stats_wide forms from them in double, are run-to-run identical).
            // Workgroup-accumulated layout: the sums of this workgroup's whole tiles -- one M tile -- stay in registers from tile to
            // tile (32 fmas per tile) and are folded and stored ONCE; the per-tile form paid 2 x 15 cross-lane exchanges and 256
            // scattered 16-byte stores per tile (one chunk per tile in layer1: 90 vs 80 us with / without statistics for 64 -> 256).
            const int slot = p.stats_acc ? (tile < p.rdp * P ? g / p.mtiles : p.stats_w + (tile - p.rdp * P) / p.mtiles) : pt;
            if (slot != run_slot) {
                if (run_slot >= 0) stats_flush();
                run_slot = slot;
                run_m0 = m0;
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int oy = oy0 + (pl[tn] >> G::TWLOG), ox = ox0 + (pl[tn] & (TW - 1));
                if (oy < p.Ho && ox < p.Wo) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        rsum[r] += acc[tn][r];
                        rsq[r] += acc[tn][r] * acc[tn][r];
                    }
                }
            }
        }
        if (p.stamps && tid == 0 && it >= it1) p.stamps[4 * g + 2] = wall_clock64();
    }
    if (p.stats && run_slot >= 0) stats_flush();
    if (p.stamps && tid == 0) p.stamps[4 * g + 3] = wall_clock64();
}

// Element of the weight tensor behind (row, kc, tap) of a packed image.  mode 0: forward; 1: input gradient at stride 1 (channel
// axes swapped, taps mirrored); 2 + SUB: parity class SUB of the input gradient of a stride-2 3x3 convolution -- tap (a, b) of
// its (1 + py) x (1 + px) window reads weight row (py ? (a ? 0 : 2) : 1), column (px ? (b ? 0 : 2) : 1).
__device__ __forceinline__ float sk_pack_elem(const float* __restrict__ w, int Cout, int Cin, int taps, int mode, int row, int kc, int tap) {
    if (mode == 0) return (row < Cout && kc < Cin) ? w[((size_t)row * Cin + kc) * taps + tap] : 0.0f;
    if (mode == 1) return (row < Cin && kc < Cout) ? w[((size_t)kc * Cin + row) * taps + (taps - 1 - tap)] : 0.0f;
    const int py = (mode - 2) >> 1, px = (mode - 2) & 1;
    const int a = tap / (1 + px), b = tap - a * (1 + px);
    const int r = py ? (a ? 0 : 2) : 1, c = px ? (b ? 0 : 2) : 1;
    return (row < Cin && kc < Cout) ? w[((size_t)kc * Cin + row) * 9 + r * 3 + c] : 0.0f;
}

// One thread per element of the packed image [mtiles][nch][KC / 8][2][BM][4]:
//   forward:        (mt, chunk, q, h, r, j) = w[mt * BM + r][chunk * CK + 2 cp + h][tap]
//   input gradient: (mt, chunk, q, h, r, j) = w[chunk * CK + 2 cp + h][mt * BM + r][taps - 1 - tap]      (rows = input channels)
// with kk = 4 q + j, tap = kk / (CK / 2), cp = kk % (CK / 2); zero outside the tensor (M tail, K tail).
__global__ __launch_bounds__(256) void k_sk_pack(const float* __restrict__ w, int Cout, int Cin, int taps, int CK, int BM, int mtiles, int nch,
                                                 int dgrad, float* __restrict__ out) {
    const int KC = taps * CK;
    const size_t total = (size_t)mtiles * nch * KC * BM;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int img = KC * BM;
    const int e = (int)(i % img);
    const int tc = (int)(i / img);
    const int chunk = tc % nch, mt = tc / nch;
    const int j = e & 3, r = (e >> 2) % BM, qh = (e >> 2) / BM;
    const int h = qh & 1, q = qh >> 1;
    const int kk = 4 * q + j, tap = kk / (CK / 2), cp = kk - tap * (CK / 2);
    const int row = mt * BM + r, kc = chunk * CK + 2 * cp + h;
    out[i] = sk_pack_elem(w, Cout, Cin, taps, dgrad, row, kc, tap);
}

// All weights of a network in ONE launch (after an optimizer step): job j packs tensor j into its image; blocks are dealt to the
// jobs through the prefix sums of their block counts.
struct SkPackJob {
    const float* w;
    float* out;
    int Cout, Cin, taps, CK, BM, mtiles, nch, dgrad;
    unsigned long long total;
    unsigned first_block;
};

__global__ __launch_bounds__(256) void k_sk_pack_multi(const SkPackJob* __restrict__ jobs, int njobs) {
    // binary search of the job that owns this block
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const SkPackJob jb = jobs[lo];
    const size_t i = (size_t)(blockIdx.x - jb.first_block) * 256 + threadIdx.x;
    if (i >= jb.total) return;
    const int KC = jb.taps * jb.CK;
    const int img = KC * jb.BM;
    const int e = (int)(i % img);
    const int tc = (int)(i / img);
    const int chunk = tc % jb.nch, mt = tc / jb.nch;
    const int j = e & 3, r = (e >> 2) % jb.BM, qh = (e >> 2) / jb.BM;
    const int h = qh & 1, q = qh >> 1;
    const int kk = 4 * q + j, tap = kk / (jb.CK / 2), cp = kk - tap * (jb.CK / 2);
    const int row = mt * jb.BM + r, kc = chunk * jb.CK + 2 * cp + h;
    jb.out[i] = sk_pack_elem(jb.w, jb.Cout, jb.Cin, jb.taps, jb.dgrad, row, kc, tap);
}

struct SkGeom {
    int TAPS, CK, BM, TW, TH;
};

// mode: 0 forward, 1 input gradient (stride 1), 2 + SUB parity class of the stride-2 3x3 input gradient (1 / 2 / 2 / 4 taps)
inline void sk_geom(int ksize, int stride, int M, int Ho, int Wo, int mode, SkGeom* g) {
    g->TAPS = ksize * ksize;
    g->CK = ksize == 3 ? 8 : (stride == 2 ? 16 : 64);       // (the stride-2 patch of a 1x1 holds 4x the pixels it uses)
    if (mode >= 2) {
        g->TAPS = (1 + ((mode - 2) >> 1)) * (1 + ((mode - 2) & 1));
        g->CK = 64 / g->TAPS;                               // 64 k-rows per chunk in every class
    }
    g->BM = M > 64 ? 128 : 64;
    // 4 x 32 or 8 x 16 pixels per tile: whichever covers the plane with fewer tiles (48 x 48: 18 exact 8 x 16 tiles; the 769-crop
    // planes 385 / 193 / 97: 3-9 % fewer 8 x 16 tiles, 49: 26 4 x 32 tiles against 28)
    const long long t32 = (long long)((Wo + 31) / 32) * ((Ho + 3) / 4), t16 = (long long)((Wo + 15) / 16) * ((Ho + 7) / 8);
    g->TW = t32 <= t16 ? 32 : 16;
    g->TH = kSkBN / g->TW;
}

// Linear pixel walk (SkG::LW) for a 3x3 stride-1 product on an H x W plane: the patch class (64 / 128 columns), or 0 when the plane
// is outside both classes, the patch would not fit the LDS (128 columns at dilation 4) or the walk saves less than a tenth of the
// tiles (48 x 48: 18 tiles either way).
inline int sk_linear_class(int ksize, int stride, int dil, int H, int W) {
    if (ksize != 3 || stride != 1) return 0;
    const int lw = (W >= 32 && W <= 56) ? 64 : ((W >= 57 && W <= 120 && dil <= 2) ? 128 : 0);
    if (!lw) return 0;
    const long long t32 = (long long)((W + 31) / 32) * ((H + 3) / 4), t16 = (long long)((W + 15) / 16) * ((H + 7) / 8);
    const long long t2d = t32 <= t16 ? t32 : t16, tl = ((long long)H * W + kSkBN - 1) / kSkBN;
    return tl * 10 <= t2d * 9 ? lw : 0;
}

// Per-call options (mas_sk_opts, include/mulactseg_hip.h): no process-wide state.  MAS_SK_DMA: LDS-DMA ring instead of the
// register-staged chunks (default: measured 3-5 % faster; the ring is kept, tested, for A/B measurements).
struct SkOpts {
    bool dma, nosplit;
    unsigned spin_limit;
    unsigned long long* stamps;
};
inline SkOpts sk_opts(const mas_sk_opts* o) {
    SkOpts r;
    r.dma = o && (o->flags & MAS_SK_DMA);
    r.nosplit = o && (o->flags & MAS_SK_NOSPLIT);
    r.spin_limit = (o && o->spin_limit) ? o->spin_limit : kSkSpinLimit;
    r.stamps = o ? static_cast<unsigned long long*>(o->stamps) : nullptr;
    return r;
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

template <int TAPS, int CK, int WM, int TW, int STRIDE, int DIL, bool VEC, int NB, int SUB = 0>
int sk_launch(const SkP& p, hipStream_t st) {
    using G = SkG<TAPS, CK, TW, STRIDE, DIL, SUB>;
    constexpr int XF = NB >= 2 ? ((CK * G::CS + 255) / 256) * 256 : CK * G::CS;
    constexpr size_t smem = sizeof(float) * ((NB >= 2 ? NB : 2) * ((size_t)TAPS * CK * 32 * WM + XF) + 4 * 32 * WM + 256 + 4);
    static_assert(smem <= 160 * 1024, "LDS");
    auto kern = &k_conv_sk<TAPS, CK, WM, TW, STRIDE, DIL, VEC, NB, SUB>;
    if (smem > 64 * 1024) {
        static bool raised[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        if (dev < 0 || dev >= 64 || !raised[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0 && dev < 64) raised[dev] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)p.P), dim3(kSkThreads), smem, st, p);
    return mas_launch_status();
}

// Layout of the BatchNorm partials (see SkP::stats_w): entries per row = stats_pitch.  The workgroup-accumulated form needs every
// workgroup's whole tiles in ONE M tile (tile = r * P + g with the M tile fastest: P a multiple of mtiles).
inline void sk_stats_layout(SkP& p, const SkGeom& g) {
    const int NG = g.BM == 128 ? 2 : 4;
    const int ntiles = p.ptiles * p.mtiles;
    const int sk_tiles = ntiles - p.rdp * p.P;
    p.stats_acc = (p.rdp > 0 && p.P % p.mtiles == 0) ? 1 : 0;
    if (p.stats_acc) {
        p.stats_w = p.P / p.mtiles;
        p.stats_pitch = (p.stats_w + (sk_tiles + p.mtiles - 1) / p.mtiles) * NG;
    } else {
        p.stats_w = 0;
        p.stats_pitch = p.ptiles * NG;
    }
}

// tiles, chunks and the deal of the iterations to the workgroups (p.K, p.M, p.Ho, p.Wo set)
inline int sk_plan(SkP& p, const SkGeom& g, int N, int cus, const SkOpts& o) {
    p.N = N;
    p.nosplit = o.nosplit ? 1 : 0;
    p.spin_limit = o.spin_limit;
    p.stamps = o.stamps;
    p.img = g.TAPS * g.CK * g.BM;
    p.ck = g.CK;
    p.xstep = (long long)g.CK * p.H * p.W;
    p.tiles_x = (p.Wo + g.TW - 1) / g.TW;
    p.tiles_y = (p.Ho + g.TH - 1) / g.TH;
    p.ptiles = N * p.tiles_x * p.tiles_y;
    p.mtiles = (p.M + g.BM - 1) / g.BM;
    p.nch = (p.K + g.CK - 1) / g.CK;
    const long long iters = (long long)p.ptiles * p.mtiles * p.nch;
    if (iters > 0x7fffffffLL) return MAS_ERR_SHAPE;
    p.iters = (int)iters;
    p.P = (int)(p.iters < cus ? p.iters : cus);
    if (p.P > 512) p.P = 512;
    const int ntiles = p.ptiles * p.mtiles;
    p.rdp = ntiles / p.P;
    p.sk_iters = (ntiles - p.rdp * p.P) * p.nch;
    sk_stats_layout(p, g);
    return 0;
}

template <int TAPS, int CK, int STRIDE, int DIL, int NB, int SUB = 0>
int sk_dispatch(const SkP& p, const SkGeom& g, bool vec, hipStream_t st) {
#define SK_TW(TW)                                                                                                                 \
    (g.BM == 128 ? (vec ? sk_launch<TAPS, CK, 4, TW, STRIDE, DIL, true, NB, SUB>(p, st) : sk_launch<TAPS, CK, 4, TW, STRIDE, DIL, false, NB, SUB>(p, st)) \
                 : (vec ? sk_launch<TAPS, CK, 2, TW, STRIDE, DIL, true, NB, SUB>(p, st) : sk_launch<TAPS, CK, 2, TW, STRIDE, DIL, false, NB, SUB>(p, st)))
    return g.TW == 32 ? SK_TW(32) : SK_TW(16);
#undef SK_TW
}
}  // namespace

extern "C" size_t mas_conv_sk_workspace_bytes(void) {
    // slots for up to 512 workgroups + their flags + the error word (padded)
    return (size_t)512 * kSkSlotFloats * sizeof(float) + 4096;
}

extern "C" size_t mas_conv_sk_packed_elems(int Cin, int Cout, int ksize, int stride, int dgrad) {
    if (Cin <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || dgrad < 0 || dgrad > 5) return 0;
    if (dgrad >= 2 && (ksize != 3 || stride != 2)) return 0;
    SkGeom g;
    const int M = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
    sk_geom(ksize, stride, M, 4, 32, dgrad, &g);
    return (size_t)((M + g.BM - 1) / g.BM) * ((K + g.CK - 1) / g.CK) * (size_t)g.TAPS * g.CK * g.BM;
}

extern "C" int mas_conv_sk_pack(const float* w, int Cin, int Cout, int ksize, int stride, int dgrad, float* out, void* stream) {
    if (!w || !out) return MAS_ERR_NULL;
    const size_t total = mas_conv_sk_packed_elems(Cin, Cout, ksize, stride, dgrad);
    if (total == 0) return MAS_ERR_SHAPE;
    SkGeom g;
    const int M = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
    sk_geom(ksize, stride, M, 4, 32, dgrad, &g);
    hipLaunchKernelGGL(k_sk_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w, Cout, Cin, g.TAPS,
                       g.CK, g.BM, (M + g.BM - 1) / g.BM, (K + g.CK - 1) / g.CK, dgrad, out);
    return mas_launch_status();
}

extern "C" size_t mas_conv_sk_pack_job_bytes(void) { return sizeof(SkPackJob); }

/* fill one job record (host memory) of a multi-tensor pack; returns the number of 256-thread blocks the job needs */
extern "C" unsigned mas_conv_sk_pack_job(void* job_host, const float* w, int Cin, int Cout, int ksize, int stride, int dgrad, float* out,
                                         unsigned first_block) {
    if (!job_host || !w || !out) return 0;
    const size_t total = mas_conv_sk_packed_elems(Cin, Cout, ksize, stride, dgrad);
    if (total == 0) return 0;
    SkGeom g;
    const int M = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
    sk_geom(ksize, stride, M, 4, 32, dgrad, &g);
    SkPackJob* jb = static_cast<SkPackJob*>(job_host);
    jb->w = w; jb->out = out; jb->Cout = Cout; jb->Cin = Cin; jb->taps = g.TAPS; jb->CK = g.CK; jb->BM = g.BM;
    jb->mtiles = (M + g.BM - 1) / g.BM; jb->nch = (K + g.CK - 1) / g.CK; jb->dgrad = dgrad;
    jb->total = total; jb->first_block = first_block;
    return (unsigned)((total + 255) / 256);
}

/* jobs_dev: `njobs` records (mas_conv_sk_pack_job, copied to the device by the caller), covering `nblocks` blocks in all */
extern "C" int mas_conv_sk_pack_multi(const void* jobs_dev, int njobs, unsigned nblocks, void* stream) {
    if (!jobs_dev) return MAS_ERR_NULL;
    if (njobs <= 0 || nblocks == 0) return MAS_ERR_SHAPE;
    hipLaunchKernelGGL(k_sk_pack_multi, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const SkPackJob*>(jobs_dev), njobs);
    return mas_launch_status();
}

namespace {
int sk_run(const float* x, const float* w, int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil, int dgrad,
           const float* scale, const float* shift, const float* residual, int relu, float* y, double2* stats, void* workspace,
           size_t workspace_bytes, unsigned epoch, const mas_sk_opts* opts, void* stream) {
    const SkOpts o = sk_opts(opts);
    if (!x || !w || !y || !workspace) return MAS_ERR_NULL;
    if (stats && (scale || residual || relu)) return MAS_ERR_RANGE;        // statistics of the bare product only
    if ((scale == nullptr) != (shift == nullptr)) return MAS_ERR_NULL;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return MAS_ERR_SHAPE;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || (dil != 1 && dil != 2 && dil != 4)) return MAS_ERR_RANGE;
    if (ksize == 1 && dil != 1) return MAS_ERR_RANGE;
    if (dgrad && stride != 1) return MAS_ERR_RANGE;
    if (stride == 2 && dil != 1) return MAS_ERR_RANGE;
    if (epoch == 0) return MAS_ERR_RANGE;
    if (workspace_bytes < mas_conv_sk_workspace_bytes()) return MAS_ERR_WORKSPACE;
    if ((long long)(Cin > Cout ? Cin : Cout) * H * W > 0x7fffffffLL) return MAS_ERR_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    SkP p;
    p.x = x; p.w = w; p.scale = scale; p.shift = shift; p.res = residual; p.y = y;
    p.slots = static_cast<float*>(workspace);
    p.flags = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + (size_t)512 * kSkSlotFloats * sizeof(float));
    p.stats = stats;
    p.zero = reinterpret_cast<const float*>(p.flags + 768);     // bytes 3072.. of the tail: zero-filled by the caller, never written
    p.epoch = epoch;
    p.relu = relu;
    p.H = H; p.W = W;
    if (!dgrad) {
        p.K = Cin; p.M = Cout;
        p.Ho = (H - 1) / stride + 1; p.Wo = (W - 1) / stride + 1;
    } else {                        // x = dY [N, Cout, H, W] (stride 1: the same plane), y = dX [N, Cin, H, W]
        p.K = Cout; p.M = Cin;
        p.Ho = H; p.Wo = W;
    }
    const bool dma = o.dma;
    const bool flat = ksize == 1 && stride == 1 && !dma;
    const int lw = dma ? 0 : sk_linear_class(ksize, stride, dil, H, W);        // (input gradient at stride 1: the same plane)
    if (lw) {
        p.Wo = H * W; p.Ho = 1;             // tiles of 128 consecutive pixels; the input plane keeps its rows (p.H, p.W)
    }
    if (flat) {
        // a 1x1 product at stride 1 does not see the plane's rows: the plane is walked as ONE row of H * W pixels in tiles of 128
        // consecutive pixels -- no partial tiles at the right edge of every row (49 x 49: 19 tiles instead of 26), the same
        // kernel for every plane size
        p.W = H * W; p.H = 1;
        p.Wo = p.W; p.Ho = 1;
    }
    p.Wy = p.Wo; p.HWy = p.Ho * p.Wo; p.os = 1; p.oy_off = 0; p.ox_off = 0;
    SkGeom g;
    sk_geom(ksize, stride, p.M, p.Ho, p.Wo, dgrad ? 1 : 0, &g);
    const int cus = sk_num_cus();
    if (flat || lw) { g.TW = 128; g.TH = 1; }
    if (int rc = sk_plan(p, g, N, cus, o)) return rc;
    const int ntiles = p.ptiles * p.mtiles;
    // 16-byte global loads: the weight rows and the planes must keep 16-byte groups whole and aligned
    if ((uintptr_t)w % 16 != 0) return MAS_ERR_ALIGN;
    const bool vec = ((uintptr_t)x % 16 == 0) && (p.W % 4 == 0);   // 16-byte loads of the input patch: whole, aligned groups
    if (!vec && ((long long)N * p.K * H * W < 4 || (p.K % g.CK != 0 && (long long)(p.K % g.CK) * H * W < 4))) return MAS_ERR_SHAPE;   // (the unaligned path loads the last group of the tensor / of a K tail so that it ends there: four elements at least)
    if ((long long)g.CK * H * W * 4 + 16 >= 0xfffffff0LL) return MAS_ERR_SHAPE;          // byte offsets inside a chunk are 32-bit
    // (forward and input gradient are the same kernel: the role lives in the packed weight image)
    if (lw) {
#define SK_LIN(DIL, LW)                                                                                                        \
    (g.BM == 128 ? (vec ? sk_launch<9, 8, 4, 128, 1, DIL, true, 0, LW>(p, st) : sk_launch<9, 8, 4, 128, 1, DIL, false, 0, LW>(p, st)) \
                 : (vec ? sk_launch<9, 8, 2, 128, 1, DIL, true, 0, LW>(p, st) : sk_launch<9, 8, 2, 128, 1, DIL, false, 0, LW>(p, st)))
        if (lw == 64) return dil == 4 ? SK_LIN(4, 64) : (dil == 2 ? SK_LIN(2, 64) : SK_LIN(1, 64));
        return dil == 2 ? SK_LIN(2, 128) : SK_LIN(1, 128);
#undef SK_LIN
    }
    if (ksize == 3) {
        if (stride == 2) return dma ? sk_dispatch<9, 8, 2, 1, 2>(p, g, vec, st) : sk_dispatch<9, 8, 2, 1, 0>(p, g, vec, st);
        if (dil == 4) return dma ? sk_dispatch<9, 8, 1, 4, 2>(p, g, vec, st) : sk_dispatch<9, 8, 1, 4, 0>(p, g, vec, st);
        if (dil == 2) return dma ? sk_dispatch<9, 8, 1, 2, 3>(p, g, vec, st) : sk_dispatch<9, 8, 1, 2, 0>(p, g, vec, st);
        return dma ? sk_dispatch<9, 8, 1, 1, 3>(p, g, vec, st) : sk_dispatch<9, 8, 1, 1, 0>(p, g, vec, st);
    }
    if (stride == 2) return dma ? sk_dispatch<1, 16, 2, 1, 3>(p, g, vec, st) : sk_dispatch<1, 16, 2, 1, 0>(p, g, vec, st);
    if (dma) {
        // the 1x1 image of a 64-channel chunk is two 32-channel images back to back: the DMA kernel walks it in 32-channel
        // chunks (four 32 KB buffers, three chunks in flight)
        SkP q = p;
        q.nch = p.nch * 2;
        q.iters = p.iters * 2;
        if ((long long)p.iters * 2 > 0x7fffffffLL) return MAS_ERR_SHAPE;
        q.P = (int)(q.iters < cus ? q.iters : cus);
        q.rdp = ntiles / q.P;
        q.sk_iters = (ntiles - q.rdp * q.P) * q.nch;
        sk_stats_layout(q, g);
        return sk_dispatch<1, 32, 1, 1, 4>(q, g, vec, st);
    }
    if (g.BM == 128) return vec ? sk_launch<1, 64, 4, 128, 1, 1, true, 0>(p, st) : sk_launch<1, 64, 4, 128, 1, 1, false, 0>(p, st);
    return vec ? sk_launch<1, 64, 2, 128, 1, 1, true, 0>(p, st) : sk_launch<1, 64, 2, 128, 1, 1, false, 0>(p, st);
}
}  // namespace

extern "C" int mas_conv_sk(const float* x, const float* w, int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil, int dgrad,
                           const float* scale, const float* shift, const float* residual, int relu, float* y, void* workspace,
                           size_t workspace_bytes, unsigned epoch, const mas_sk_opts* opts, void* stream) {
    return sk_run(x, w, N, Cin, H, W, Cout, ksize, stride, dil, dgrad, scale, shift, residual, relu, y, nullptr, workspace, workspace_bytes, epoch,
                  opts, stream);
}

/* Entries per output channel of the BatchNorm partials mas_conv_sk_stats writes for this forward product (0: unsupported): the same
 * geometry and deal as the launch itself (sk_plan), so it depends on the device's CU count. */
extern "C" int mas_conv_sk_stats_slots(int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil, unsigned flags) {
    const bool dma = (flags & MAS_SK_DMA) != 0;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return 0;
    SkP p;
    p.K = Cin; p.M = Cout; p.H = H; p.W = W;
    p.Ho = (H - 1) / stride + 1; p.Wo = (W - 1) / stride + 1;
    const bool flat = !dma && ((ksize == 1 && stride == 1) || sk_linear_class(ksize, stride, dil, H, W) != 0);
    if (flat) { p.Wo = H * W; p.Ho = 1; if (ksize == 1) { p.W = H * W; p.H = 1; } }
    SkGeom g;
    sk_geom(ksize, stride, Cout, p.Ho, p.Wo, 0, &g);
    if (flat) { g.TW = 128; g.TH = 1; }
    p.scale = nullptr; p.relu = 0; p.stats = nullptr; p.HWy = p.Ho * p.Wo;
    mas_sk_opts o = {};
    o.flags = flags;
    if (sk_plan(p, g, N, sk_num_cus(), sk_opts(&o)) != 0) return 0;
    if (dma && ksize == 1 && stride == 1) {             // (the LDS-DMA 1x1 kernel walks 32-channel chunks: sk_run re-deals)
        const int ntiles = p.ptiles * p.mtiles;
        const long long iters = (long long)p.iters * 2;
        p.nch *= 2;
        p.P = (int)(iters < sk_num_cus() ? iters : sk_num_cus());
        p.rdp = ntiles / p.P;
        p.sk_iters = (ntiles - p.rdp * p.P) * p.nch;
        sk_stats_layout(p, g);
    }
    return p.stats_pitch;
}

/* mas_conv_sk (forward, bare: no scale / residual / ReLU) that also writes the BatchNorm partial sums of its output:
 * stats [Cout][mas_conv_sk_stats_slots()] pairs of doubles (sum y, sum y^2) over disjoint pixel sets -- the input of
 * mas_bn_act_train_fwd_stats. */
extern "C" int mas_conv_sk_stats(const float* x, const float* w, int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil, float* y,
                                 double* stats, void* workspace, size_t workspace_bytes, unsigned epoch, const mas_sk_opts* opts, void* stream) {
    if (!stats) return MAS_ERR_NULL;
    return sk_run(x, w, N, Cin, H, W, Cout, ksize, stride, dil, 0, nullptr, nullptr, nullptr, 0, y, reinterpret_cast<double2*>(stats), workspace,
                  workspace_bytes, epoch, opts, stream);
}

/* One parity class (sub = 2 py + px) of the input gradient of a 3x3, stride-2, padding-1 convolution: dy [N,Cout,Hd,Wd] with
 * Hd = (H - 1) / 2 + 1, wp = the class image of the weight (mas_conv_sk_pack with dgrad = 2 + sub); writes the pixels
 * (2 i + py, 2 j + px) of dx [N,Cin,H,W] -- the four classes together write every pixel once.  Epilogue as mas_conv_sk. */
extern "C" int mas_conv_sk_dgrad_s2(const float* dy, const float* wp, int N, int Cin, int H, int W, int Cout, int sub, const float* scale,
                                    const float* shift, const float* residual, int relu, float* dx, void* workspace, size_t workspace_bytes,
                                    unsigned epoch, const mas_sk_opts* opts, void* stream) {
    if (!dy || !wp || !dx || !workspace) return MAS_ERR_NULL;
    const SkOpts o = sk_opts(opts);
    if ((scale == nullptr) != (shift == nullptr)) return MAS_ERR_NULL;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return MAS_ERR_SHAPE;
    if (sub < 0 || sub > 3 || epoch == 0) return MAS_ERR_RANGE;
    if (workspace_bytes < mas_conv_sk_workspace_bytes()) return MAS_ERR_WORKSPACE;
    if ((long long)(Cin > Cout ? Cin : Cout) * H * W > 0x7fffffffLL) return MAS_ERR_SHAPE;
    if ((uintptr_t)wp % 16 != 0) return MAS_ERR_ALIGN;
    const int py = sub >> 1, px = sub & 1;
    SkP p;
    p.x = dy; p.w = wp; p.scale = scale; p.shift = shift; p.res = residual; p.y = dx;
    p.slots = static_cast<float*>(workspace);
    p.flags = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + (size_t)512 * kSkSlotFloats * sizeof(float));
    p.stats = nullptr;
    p.zero = reinterpret_cast<const float*>(p.flags + 768);
    p.epoch = epoch;
    p.relu = relu;
    p.K = Cout; p.M = Cin;
    p.H = (H - 1) / 2 + 1; p.W = (W - 1) / 2 + 1;               // the plane the patches are read from: dY
    p.Ho = (H - py + 1) / 2; p.Wo = (W - px + 1) / 2;           // pixels of this class
    p.Wy = W; p.HWy = H * W; p.os = 2; p.oy_off = py; p.ox_off = px;
    if (p.Ho == 0 || p.Wo == 0) return 0;
    SkGeom g;
    sk_geom(3, 2, p.M, p.Ho, p.Wo, 2 + sub, &g);
    if (int rc = sk_plan(p, g, N, sk_num_cus(), o)) return rc;
    const bool vec = ((uintptr_t)dy % 16 == 0) && (p.W % 4 == 0);
    if (!vec && ((long long)N * p.K * p.H * p.W < 4 || (p.K % g.CK != 0 && (long long)(p.K % g.CK) * p.H * p.W < 4))) return MAS_ERR_SHAPE;
    if ((long long)g.CK * p.H * p.W * 4 + 16 >= 0xfffffff0LL) return MAS_ERR_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (sub) {
        case 0: return sk_dispatch<1, 64, 1, 1, 0, 0>(p, g, vec, st);
        case 1: return sk_dispatch<2, 32, 1, 1, 0, 1>(p, g, vec, st);
        case 2: return sk_dispatch<2, 32, 1, 1, 0, 2>(p, g, vec, st);
        default: return sk_dispatch<4, 16, 1, 1, 0, 3>(p, g, vec, st);
    }
}

/* error word of the last launches on this workspace: non-zero = a bounded spin gave up (host-side read, for tests) */
extern "C" int mas_conv_sk_error(const void* workspace, unsigned* out_host) {
    if (!workspace || !out_host) return MAS_ERR_NULL;
    const char* base = static_cast<const char*>(workspace) + (size_t)512 * kSkSlotFloats * sizeof(float);
    hipError_t e = hipMemcpy(out_host, base + 512 * sizeof(unsigned), sizeof(unsigned), hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : (int)e;
}
