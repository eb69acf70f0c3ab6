// conv_wgrad_bx.hip -- weight gradient of the 1x1 stride-1 convolutions on the bf16 matrix cores with f32 operands and f32 results
// (bx_split.h: both operands split exactly into three bf16 terms between the global load and the LDS store, six partial products
// per 16-k step on v_mfma_f32_32x32x16_bf16, f32 accumulation -- the error bound of an f32 product, exact on integer data):
//     dW[m, c] = sum_{n, p} dY[n, m, p] * X[n, c, p]
// Reference: the backward of the 1x1 nn.Conv2d layers of models/segmentation/backbone/resnet.py:129-160 (conv1 / conv3 /
// downsample of every Bottleneck) and models/segmentation/deeplabv3.py:85-137,216-245 (ASPP 1x1, projections, the pointwise halves
// of the separable convolutions) under trainer/active_joint_multi_predignore_lossdecomp.py:83-116 (loss.backward()).
//
// GEMM view: M = output channels (A = dY), N = input channels (B = X), K = the pixels of all pictures.  Both operands are
// pixel-contiguous in NCHW, i.e. K-contiguous, which is what the bf16 MFMA wants (a lane's fragment = 8 consecutive k): a thread
// loads four consecutive pixels of a row, splits them and stores three 8-byte pieces -- no transposition anywhere.
//   * tile 128 x 128 per workgroup of 4 waves (64 x 64 each), K in chunks of 32 pixels (two 16-k steps); chunk t + 1 travels
//     global -> registers in front of the MFMAs of chunk t, registers -> (split) -> LDS behind them; two workgroups per CU.
//     (that is k_wgrad_bx, built with -DWX_PIPE2=0; what runs since round 5 is k_wgrad_bx_p2 below: the same tile, operand images
//     and product order, ONE workgroup per CU, staging interleaved by hand with the MFMAs.)
//   * split K: the grid is (tiles) x S pixel ranges, every workgroup writes its partial tile to workspace [S][Cout][Cin] and
//     k_wgx_reduce adds the slices in a fixed order (no atomics: run-to-run identical).  Workgroups of one range sit on one XCD.
#include <cstdlib>

#include "common.h"
#include "bx_split.h"

namespace {
constexpr int kWxThreads = 256;
constexpr int kWxKP = 32;                   // pixels per chunk
constexpr int kWxGS = 129;                  // 16-byte units per k group: 128 rows + 1 (spreads the staging stores over the banks)
constexpr int kWxImg = 3 * 4 * kWxGS;       // units of one operand image [term][k group][row]
constexpr unsigned kWxRsrcFlags = 0x00020000;
constexpr int kWxOut = (int)0x80000000u;    // a byte offset beyond every resource used here
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

struct WxP {
    const float* x;
    const float* dy;
    float* part;
    int N, Cin, Cout, HW;
    int cpp, nch;                           // chunks per picture, chunks in all
    int mtiles, ctiles, S;
};

__global__ __launch_bounds__(kWxThreads, 2) void k_wgrad_bx(const WxP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wx_smem[];
    v4f* sA = reinterpret_cast<v4f*>(wx_smem);
    v4f* sB = sA + kWxImg;
    const int tid = threadIdx.x;
    const int tiles = p.mtiles * p.ctiles;
    int tile, s;
    if (p.S % 8 == 0) {                     // the workgroups of one pixel range on one XCD: they stream the same chunks through its L2
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        tile = slot % tiles;
        s = (slot / tiles) * 8 + xcd;
    } else {                                // (fewer ranges than XCDs, or an odd count: every XCD must get work first)
        tile = blockIdx.x % tiles;
        s = blockIdx.x / tiles;
    }
    const int mt = tile / p.ctiles, ct = tile - mt * p.ctiles;
    const int m0 = mt * 128, c0 = ct * 128;
    const int HW = p.HW;

    // ---- staging: thread <-> four (row, pixel quad) pairs of each operand ----------------------------------------------------------
    // quad q = tid + 256 j: pixel quad pq = q & 7 (8 lanes = the 128 contiguous bytes of a row), row from q >> 3 with bits 0 and 2
    // swapped, so that the two rows of a 16-lane group are four apart: their 8-byte stores then fall on 16 different bank pairs
    int ga[4], gb[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = tid + j * kWxThreads;
        const int pq = q & 7, rr = q >> 3;
        const int r = (rr & ~5) | ((rr & 1) << 2) | ((rr >> 2) & 1);
        ga[j] = (m0 + r < p.Cout) ? ((m0 + r) * HW + pq * 4) * 4 : kWxOut;
        gb[j] = (c0 + r < p.Cin) ? ((c0 + r) * HW + pq * 4) * 4 : kWxOut;
        lo[j] = ((pq >> 1) * kWxGS + r) * 16 + (pq & 1) * 8;
    }
    // ---- MFMA operand addressing ---------------------------------------------------------------------------------------------------
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int aBase = h * kWxGS + wm * 64 + l31, bBase = h * kWxGS + wn * 64 + l31;
    f32x16 acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;

    const int cq = p.nch / p.S, cr = p.nch - cq * p.S;       // ranges of cq or cq + 1 chunks (the first cr ranges take one more)
    const int c_lo = s * cq + (s < cr ? s : cr), c_hi = c_lo + cq + (s < cr ? 1 : 0);
    v4f ra[4], rb[4];
    int tail = 0;                           // pixels of the chunk in the staging registers that exist (32, fewer in a picture's last chunk)
    auto fetch = [&](int cidx) {
        cidx = __builtin_amdgcn_readfirstlane(cidx);        // wave-uniform (keeps the resource descriptors in scalar registers)
        const int n = cidx / p.cpp, px0 = (cidx - n * p.cpp) * kWxKP;
        tail = HW - px0 < kWxKP ? HW - px0 : kWxKP;
        // the picture's rows from this chunk's first pixel on: every valid (row, quad) lies inside, kWxOut reads zeros
        const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + (size_t)n * p.Cout * HW + px0), 0,
                                                                              (p.Cout * HW - px0) * 4, kWxRsrcFlags);
        const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)n * p.Cin * HW + px0), 0,
                                                                              (p.Cin * HW - px0) * 4, kWxRsrcFlags);
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(ares, ga[j], 0, 0));
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(bres, gb[j], 0, 0));
    };
    auto stage_one = [&](v4f* img, int off, const v4f& v) {
        unsigned h0, m0_, l0, h1, m1, l1;
        bx_split2(v.x, v.y, h0, m0_, l0);
        bx_split2(v.z, v.w, h1, m1, l1);
        unsigned char* dst = reinterpret_cast<unsigned char*>(img) + off;
        *reinterpret_cast<v2u*>(dst) = (v2u){h0, h1};
        *reinterpret_cast<v2u*>(dst + 4 * kWxGS * 16) = (v2u){m0_, m1};
        *reinterpret_cast<v2u*>(dst + 8 * kWxGS * 16) = (v2u){l0, l1};
    };
    auto stage = [&]() {
        // a picture's last, partial chunk (H*W % 32 != 0): the pixels beyond the plane belong to the next row of dY -- zeroed on the
        // dY side (a zero times the finite value on the x side contributes nothing); selects, no branch in the loop
        const int left = tail - (tid & 7) * 4;              // pixels of this thread's quads that exist
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j].x = left > 0 ? ra[j].x : 0.0f;
            ra[j].y = left > 1 ? ra[j].y : 0.0f;
            ra[j].z = left > 2 ? ra[j].z : 0.0f;
            ra[j].w = left > 3 ? ra[j].w : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) stage_one(sA, lo[j], ra[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) stage_one(sB, lo[j], rb[j]);
    };
    auto mfma_chunk = [&]() {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf8 a[2][3], b[2][3];
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[t][term] = __builtin_bit_cast(bf8, sA[(term * 4 + 2 * st) * kWxGS + aBase + t * 32]);
                    b[t][term] = __builtin_bit_cast(bf8, sB[(term * 4 + 2 * st) * kWxGS + bBase + t * 32]);
                }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    f32x16 c = acc[tm][tn];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][2], b[tn][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][0], c, 0, 0, 0);
                    acc[tm][tn] = c;
                }
        }
    };
    // (one loop without a peeled last pass -- its fetch re-reads the last chunk, never used: with the peel hipcc merges the two
    //  staging sequences and copies every loop-carried register at the head of each iteration, see conv_bx.hip)
    const int c_first = __builtin_amdgcn_readfirstlane(c_lo), c_last = __builtin_amdgcn_readfirstlane(c_hi) - 1;
    if (c_first <= c_last) fetch(c_first);
    for (int t = c_first; t <= c_last; ++t) {
        stage();
        __syncthreads();
        fetch(t < c_last ? t + 1 : t);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk();
        __syncthreads();
    }
    // ---- epilogue: the partial tile into slice s of the workspace (rows beyond Cout / columns beyond Cin are out of range) -------------
    // (on a whole M tile the row part of the address is a scalar offset: a lane's 16 stores of an MFMA tile share one address register)
    const __amdgpu_buffer_rsrc_t pres = __builtin_amdgcn_make_buffer_rsrc(p.part + (size_t)s * p.Cout * p.Cin, 0, p.Cout * p.Cin * 4, kWxRsrcFlags);
    const bool full = m0 + 128 <= p.Cout;               // wave-uniform
    const int row4 = p.Cin * 4;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int c = c0 + wn * 64 + tn * 32 + l31;
        const int vb = c < p.Cin ? c * 4 : kWxOut;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int mb = m0 + wm * 64 + tm * 32 + 4 * h;
            const int vbase = vb + mb * row4;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[tm][tn][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), pres, vbase, ((r & 3) + 8 * (r >> 2)) * row4, 0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[tm][tn][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), pres, vb + (mb + (r & 3) + 8 * (r >> 2)) * row4, 0, 0);
                }
            }
        }
    }
}

#ifndef WX_PIPE2
#define WX_PIPE2 1
#endif
#if WX_PIPE2
// The kernel that runs (round 5; -DWX_PIPE2=0 builds the two-workgroup form above for A/B): ONE workgroup per CU, two LDS buffers
// (static arrays), two sets of staging registers; the split + LDS stores of chunk t + 1 are interleaved BY HAND with the MFMAs of
// chunk t in the wave's own instruction stream -- a wave hides about five of its own instructions between two of its own MFMAs,
// but gets one issue slot per MFMA of its SIMD partner (round 4's stamps), which is what kept the two-workgroup form at MfmaUtil
// 0.5: its 300 staging instructions per chunk ran at the partner's MFMA pace.  The chunk's 48 MFMAs are each followed by one of 48
// slices of the next chunk's staging (five VALU instructions, or the three LDS stores of a quad), the order pinned with
// sched_barrier; one barrier per chunk; 280 VGPRs.  Same products in the same order per accumulator.  1x1 weight gradients of a
// training step 2 883 -> 2 718 us (-7 ... -24 % on the layers of up to 512 channels, nothing on the 1024 / 2048-channel layers,
// which re-read their operands from L2 at 5.7 TB/s with 128 x 128 tiles); step -0.3 ms.
// NB: the X side of the tile is NB blocks of 128 input channels (a wave owns 64 x 64 NB): NB = 2 where the input channels fill whole
// pairs of blocks -- dY is then re-read by half as many column tiles and a chunk's 96 MFMAs carry 72 staging slices instead of 48 : 48.
template <int NB>
__global__ __launch_bounds__(kWxThreads, 1) void k_wgrad_bx_p2(const WxP p) {
    constexpr int GSB = 128 * NB + 1, IMGB = 3 * 4 * GSB;        // units per k group / per image of the X operand
    constexpr int NQB = 4 * NB, NTN = 2 * NB;
    __shared__ __attribute__((aligned(16))) v4f wxb0[kWxImg + IMGB], wxb1[kWxImg + IMGB];
    const int tid = threadIdx.x;
    const int tiles = p.mtiles * p.ctiles;
    int tile, s;
    if (p.S % 8 == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        tile = slot % tiles;
        s = (slot / tiles) * 8 + xcd;
    } else {
        tile = blockIdx.x % tiles;
        s = blockIdx.x / tiles;
    }
    const int mt = tile / p.ctiles, ct = tile - mt * p.ctiles;
    const int m0 = mt * 128, c0 = ct * 128 * NB;
    const int HW = p.HW;
    int ga[4], gb[NQB], lo[4], lob[NQB];
#pragma unroll
    for (int j = 0; j < NQB; ++j) {
        const int q = tid + j * kWxThreads;
        const int pq = q & 7, rr = q >> 3;
        const int r = (rr & ~5) | ((rr & 1) << 2) | ((rr >> 2) & 1);
        if (j < 4) {
            ga[j] = (m0 + r < p.Cout) ? ((m0 + r) * HW + pq * 4) * 4 : kWxOut;
            lo[j] = ((pq >> 1) * kWxGS + r) * 16 + (pq & 1) * 8;
        }
        gb[j] = (c0 + r < p.Cin) ? ((c0 + r) * HW + pq * 4) * 4 : kWxOut;
        lob[j] = ((pq >> 1) * GSB + r) * 16 + (pq & 1) * 8;
    }
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int aBase = h * kWxGS + wm * 64 + l31, bBase = h * GSB + wn * 64 * NB + l31;
    f32x16 acc[2][NTN];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;
    const int cq = p.nch / p.S, cr = p.nch - cq * p.S;
    const int c_lo = s * cq + (s < cr ? s : cr), c_hi = c_lo + cq + (s < cr ? 1 : 0);
    v4f ra[2][4], rb[2][NQB];
    int tail[2] = {0, 0};
    auto fetch = [&](int cidx, int rs) {
        cidx = __builtin_amdgcn_readfirstlane(cidx);
        const int n = cidx / p.cpp, px0 = (cidx - n * p.cpp) * kWxKP;
        tail[rs] = HW - px0 < kWxKP ? HW - px0 : kWxKP;
        const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + (size_t)n * p.Cout * HW + px0), 0,
                                                                              (p.Cout * HW - px0) * 4, kWxRsrcFlags);
        const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)n * p.Cin * HW + px0), 0,
                                                                              (p.Cin * HW - px0) * 4, kWxRsrcFlags);
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[rs][j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(ares, ga[j], 0, 0));
#pragma unroll
        for (int j = 0; j < NQB; ++j) rb[rs][j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(bres, gb[j], 0, 0));
    };
    // ---- the staging of one chunk in 48 slices, one per MFMA of the chunk that is being multiplied --------------------------------
    // quad q = 0 .. 3 + 4 NB (four of dY, 4 NB of X), slice 6 q + ph:  ph 0 / 1: h and the first remainders of the pair (x, y) / (z, w);
    // ph 2 / 3: m and the second remainders; ph 4: the two l, the three 8-byte LDS stores; ph 5: the tail selects of the next dY quad.
    // (bx_split2 in pieces: the same operations on the same values.)
    unsigned sh[2], sm[2], sl[2];
    float sr[2][2];
    auto slice = [&](int idx, int rs, int lb) {
        const int q = idx / 6, ph = idx - 6 * q;
        const bool isA = q < 4;
        const int j = isA ? q : q - 4;
        v4f& v = isA ? ra[rs][j] : rb[rs][j];
        if (ph == 0 || ph == 1) {
            const float v0 = ph ? v.z : v.x, v1 = ph ? v.w : v.y;
            const unsigned hh = bx_cvt_pk(v0, v1);
            sh[ph] = hh;
            sr[ph][0] = v0 - __uint_as_float(hh << 16);
            sr[ph][1] = v1 - __uint_as_float(hh & 0xffff0000u);
        } else if (ph == 2 || ph == 3) {
            const int k = ph - 2;
            const unsigned mm = bx_cvt_pk(sr[k][0], sr[k][1]);
            sm[k] = mm;
            sr[k][0] = sr[k][0] - __uint_as_float(mm << 16);
            sr[k][1] = sr[k][1] - __uint_as_float(mm & 0xffff0000u);
        } else if (ph == 4) {
            sl[0] = bx_cvt_pk(sr[0][0], sr[0][1]);
            sl[1] = bx_cvt_pk(sr[1][0], sr[1][1]);
            v4f* img = (lb ? wxb1 : wxb0) + (isA ? 0 : kWxImg);
            unsigned char* dst = reinterpret_cast<unsigned char*>(img) + (isA ? lo[j] : lob[j]);
            const int tstride = (isA ? kWxGS : GSB) * 4 * 16;          // bytes of one term of the image
            *reinterpret_cast<v2u*>(dst) = (v2u){sh[0], sh[1]};
            *reinterpret_cast<v2u*>(dst + tstride) = (v2u){sm[0], sm[1]};
            *reinterpret_cast<v2u*>(dst + 2 * tstride) = (v2u){sl[0], sl[1]};
        } else if (q + 1 < 4) {                  // ph 5: a picture's last, partial chunk -- the pixels beyond the plane are zeroed on the dY side
            const int left = tail[rs] - (tid & 7) * 4;
            v4f& n = ra[rs][q + 1];
            n.x = left > 0 ? n.x : 0.0f; n.y = left > 1 ? n.y : 0.0f; n.z = left > 2 ? n.z : 0.0f; n.w = left > 3 ? n.w : 0.0f;
        }
    };
    auto mask_first = [&](int rs) {             // (the selects of dY quad 0, in front of slice 0)
        const int left = tail[rs] - (tid & 7) * 4;
        v4f& n = ra[rs][0];
        n.x = left > 0 ? n.x : 0.0f; n.y = left > 1 ? n.y : 0.0f; n.z = left > 2 ? n.z : 0.0f; n.w = left > 3 ? n.w : 0.0f;
    };
    auto stage_all = [&](int rs, int lb) {      // (the first chunk of a workgroup: nothing to multiply yet)
        mask_first(rs);
#pragma unroll
        for (int i = 0; i < 6 * (4 + NQB); ++i) slice(i, rs, lb);
    };
    // ---- one step: the 48 MFMAs of the chunk in buffer b, each followed by one staging slice of the next chunk (registers of set
    // 1 - b -> buffer 1 - b); the fragments of the second 16-k step are requested half a step ahead ---------------------------------
    auto frags = [&](int lb, int st, bf8 (&a)[2][3], bf8 (&bb)[NTN][3]) {
        const v4f* sA = lb ? wxb1 : wxb0;
        const v4f* sB = sA + kWxImg;
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int t = 0; t < NTN; ++t) {
                if (t < 2) a[t][term] = __builtin_bit_cast(bf8, sA[(term * 4 + 2 * st) * kWxGS + aBase + t * 32]);
                bb[t][term] = __builtin_bit_cast(bf8, sB[(term * 4 + 2 * st) * GSB + bBase + t * 32]);
            }
    };
    const int c_first = __builtin_amdgcn_readfirstlane(c_lo), c_last = __builtin_amdgcn_readfirstlane(c_hi) - 1;
    if (c_first <= c_last) {
        fetch(c_first, 0);
        fetch(c_first + 1 <= c_last ? c_first + 1 : c_last, 1);
        stage_all(0, 0);
        __syncthreads();
        auto step = [&](int t, int b) {
            fetch(t + 2 <= c_last ? t + 2 : c_last, b);     // (in front of the MFMAs: under them -- after the second -- measured 1 % slower)
            bf8 fa[2][2][3], fb[2][NTN][3];
            frags(b, 0, fa[0], fb[0]);
            mask_first(1 - b);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int ta[6] = {1, 0, 2, 0, 1, 0}, tb[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
                        for (int pr = 0; pr < 6; ++pr) {
                            const int i = ((st * 2 + tm) * NTN + tn) * 6 + pr;
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][tm][ta[pr]], fb[st][tn][tb[pr]], acc[tm][tn], 0, 0, 0);
                            if (i < 6 * (4 + NQB)) slice(i, 1 - b, 1 - b);
                            if (i == (NB == 1 ? 8 : 24 * NB - 8)) frags(b, 1, fa[1], fb[1]);     // (NB 2: late, for the register budget)
                            __builtin_amdgcn_sched_barrier(0);
                        }
            __syncthreads();
        };
        for (int t = c_first; t <= c_last; t += 2) {
            step(t, 0);
            if (t + 1 > c_last) break;
            step(t + 1, 1);
        }
    }
    const __amdgpu_buffer_rsrc_t pres = __builtin_amdgcn_make_buffer_rsrc(p.part + (size_t)s * p.Cout * p.Cin, 0, p.Cout * p.Cin * 4, kWxRsrcFlags);
    const bool full = m0 + 128 <= p.Cout;
    const int row4 = p.Cin * 4;
#pragma unroll
    for (int tn = 0; tn < NTN; ++tn) {
        const int c = c0 + wn * 64 * NB + tn * 32 + l31;
        const int vb = c < p.Cin ? c * 4 : kWxOut;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int mb = m0 + wm * 64 + tm * 32 + 4 * h;
            const int vbase = vb + mb * row4;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[tm][tn][r]), pres, vbase, ((r & 3) + 8 * (r >> 2)) * row4, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[tm][tn][r]), pres, vb + (mb + (r & 3) + 8 * (r >> 2)) * row4, 0, 0);
            }
        }
    }
}
#endif

__global__ __launch_bounds__(256) void k_wgx_reduce(const float* __restrict__ part, int S, size_t n, float* __restrict__ dw) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;          // four chains in a fixed order
    int s = 0;
    for (; s + 4 <= S; s += 4) {
        const float a = part[(size_t)s * n + i], b = part[(size_t)(s + 1) * n + i], c = part[(size_t)(s + 2) * n + i],
                    d = part[(size_t)(s + 3) * n + i];
        v0 += a; v1 += b; v2 += c; v3 += d;
    }
    if (s < S) v0 += part[(size_t)s * n + i];
    if (s + 1 < S) v1 += part[(size_t)(s + 1) * n + i];
    if (s + 2 < S) v2 += part[(size_t)(s + 2) * n + i];
    dw[i] = (v0 + v1) + (v2 + v3);
}

constexpr size_t kWxPartCap = (size_t)64 << 20;
// S pixel ranges: the chip holds 512 workgroups at a time (2 per CU); a workgroup pays about three chunk times of prologue and
// epilogue, the slices are written once and read once
inline int wx_pick_split(int tiles, int nch, size_t slice_bytes, int nb = 1) {
    int best = 1;
    double best_cost = 1e30;
    const int smax = nch < 1024 ? nch : 1024;
    for (int S = 1; S <= smax; ++S) {
        if (S > 1 && (size_t)S * slice_bytes > kWxPartCap) break;
        const long long wg = (long long)tiles * S;
#if WX_PIPE2
        const long long rounds = (wg + 255) / 256;               // (one workgroup per CU)
#else
        const long long rounds = (wg + 511) / 512;
#endif
        const double per = (double)nb * (double)((nch + S - 1) / S);       // (a chunk of a 128 x 256 tile: twice the MFMAs)
        const double cost = (double)rounds * (per + 3.0) * 2.0 + (double)S * (double)slice_bytes * 2.0 / 3e6;      // microseconds
        if (cost < best_cost) {
            best_cost = cost;
            best = S;
        }
    }
    return best;
}
}  // namespace

extern "C" int mas_conv_wgrad_bx_supported(int N, int Cin, int H, int W, int Cout) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    const long long HW = (long long)H * W;
    if ((long long)Cin * HW * 4 >= 0x7fffffffLL || (long long)Cout * HW * 4 >= 0x7fffffffLL || (long long)Cin * Cout * 4 >= 0x7fffffffLL) return 0;
    return 1;
}

extern "C" size_t mas_conv_wgrad_bx_workspace_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    const size_t slice = sizeof(float) * (size_t)Cout * Cin;
    size_t smax = kWxPartCap / slice;
    if (smax < 1) smax = 1;
    if (smax > 1024) smax = 1024;
    return smax * slice;
}

extern "C" int mas_conv_wgrad_bx(const float* x, const float* dy, int N, int Cin, int H, int W, int Cout, float* dw, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || !workspace) return MAS_ERR_NULL;
    if (!mas_conv_wgrad_bx_supported(N, Cin, H, W, Cout)) return MAS_ERR_SHAPE;
    if ((uintptr_t)x % 4 != 0 || (uintptr_t)dy % 4 != 0) return MAS_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    WxP p;
    p.x = x; p.dy = dy; p.part = static_cast<float*>(workspace);
    p.N = N; p.Cin = Cin; p.Cout = Cout; p.HW = H * W;
    p.cpp = (p.HW + kWxKP - 1) / kWxKP;
    p.nch = N * p.cpp;
    p.mtiles = (Cout + 127) / 128;
    p.ctiles = (Cin + 127) / 128;
    // the X side of the tile: 256 input channels where they come in whole pairs of 128-blocks AND the product has at least 64 tiles of
    // 128 x 128 (1024 -> 2048, 512 <-> 2048: -3 ... -5 %; with fewer tiles the wide form loses 3 ... 18 % -- fewer, longer
    // workgroups and more K ranges: gpurun_out/s57)
    int nb = 1;
#if WX_PIPE2
    static const int nb_env = [] { const char* e = getenv("MAS_WGRAD_BX_NB"); return e ? atoi(e) : 0; }();      // (A/B: 1 = always 128 x 128, 2 = wide wherever possible)
    if (p.ctiles % 2 == 0 && nb_env != 1 && (p.mtiles * p.ctiles >= 64 || nb_env == 2)) {
        nb = 2;
        p.ctiles /= 2;
    }
#endif
    const size_t slice = sizeof(float) * (size_t)Cout * Cin;
    p.S = wx_pick_split(p.mtiles * p.ctiles, p.nch, slice, nb);
    while (p.S > 1 && (size_t)p.S * slice > workspace_bytes) --p.S;
    if ((size_t)p.S * slice > workspace_bytes) return MAS_ERR_WORKSPACE;
    const size_t smem = (size_t)2 * kWxImg * 16;
    (void)smem;
    const long long nblk = (long long)p.S * p.mtiles * p.ctiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
#if WX_PIPE2
    if (nb == 2) hipLaunchKernelGGL(k_wgrad_bx_p2<2>, dim3((unsigned)nblk), dim3(kWxThreads), 0, st, p);
    else hipLaunchKernelGGL(k_wgrad_bx_p2<1>, dim3((unsigned)nblk), dim3(kWxThreads), 0, st, p);
#else
    hipLaunchKernelGGL(k_wgrad_bx, dim3((unsigned)nblk), dim3(kWxThreads), smem, st, p);
#endif
    int rc = mas_launch_status();
    if (rc != 0) return rc;
    const size_t n = (size_t)Cout * Cin;
    hipLaunchKernelGGL(k_wgx_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p.part, p.S, n, dw);
    return mas_launch_status();
}
