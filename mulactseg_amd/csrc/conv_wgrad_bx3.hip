// conv_wgrad_bx3.hip -- weight gradient of the 3x3 stride-1 convolutions (dilation 1 | 2, padding = dilation) on the bf16 matrix
// cores with f32 operands and f32 results (bx_split.h: both operands split exactly into three bf16 terms, six partial products per
// 16-k step on v_mfma_f32_32x32x16_bf16, f32 accumulation -- the error bound of an f32 product, exact on integer data):
//     dW[m, c, ty, tx] = sum_{n, y, x} dY[n, m, y, x] * X[n, c, y + (ty - 1) d, x + (tx - 1) d]          (X = 0 outside the plane)
// Reference: the backward of the 3x3 nn.Conv2d layers of models/segmentation/backbone/resnet.py:129-171 (deep stem, conv2 of every
// Bottleneck; layer4 dilated) under trainer/active_joint_multi_predignore_lossdecomp.py:83-116 (loss.backward()).  Rounds 3-4 ran
// these products on the f32 pipe (csrc/conv_wgrad.hip:k_wgrad<9, ...>, 3.6 ms of a 28 ms step): with K = pixels, a tap is a shift of
// the K axis by one pixel, i.e. a 2-byte shift of a K-contiguous bf16 row -- a misaligned MFMA fragment.
//
// What makes it work: gfx950's TRANSPOSING LDS read (ds_read_b64_tr_b16).  The X patch is staged the way the FORWARD 3x3 kernel
// stages it -- [pixel][32 channels], channel-contiguous, split into three terms -- so a tap is a whole-unit address offset; the
// transposing read then hands every lane the 4 consecutive PIXELS (K) of ITS channel (N) that the MFMA's B operand wants.  dY is
// pixel-contiguous in NCHW already (the A operand, as in conv_wgrad_bx.hip).
//
// GEMM view: M = 64 output channels (A = dY), N = 32 input channels (B = X patch), K = pixels in chunks of 4 rows x 16 columns
// (four 16-k steps, one per row); nine (M x N) accumulator tiles, one per tap.  A workgroup is SIX waves = 2 (M halves) x 3 (tap
// rows ty): a wave owns 32 x 32 x 3 taps = 48 accumulator registers, reads the A fragment of a k step once for its three taps and
// the B fragment of every tap with two transposing reads per term.  Chunk t + 1 travels global -> registers in front of the MFMAs
// of chunk t and registers -> (split) -> LDS behind them; 53 / 62 KB of LDS, <= 128 registers: two workgroups (twelve waves) per CU.
// Split K: the grid is (tiles) x S chunk ranges; every workgroup writes its partial tile into slice s of the workspace
// [S][Cout][Cin][9] and k_w3_reduce adds the slices in index order (no atomics: run-to-run identical).
#include "common.h"
#include "bx_split.h"

namespace {
constexpr int kW3Threads = 384;
constexpr int kW3TH = 4, kW3TW = 16;        // output pixels of a chunk
constexpr int kW3BM = 64, kW3BC = 32;
constexpr int kW3AP = 65;                   // 16-byte units per 8-pixel group of the A image: 64 rows + 1 (spreads the staging stores over the banks)
constexpr int kW3PW = 24;                   // pixel pitch of a patch row in the B image: >= 16 + 2 * 2, a multiple of 8 (see w3_bslot)
constexpr unsigned kW3RsrcFlags = 0x00020000;
constexpr int kW3Out = (int)0x80000000u;    // a byte offset beyond every resource used here
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

struct W3P {
    const float* x;
    const float* dy;
    float* part;
    int N, Cin, Cout, H, W;
    int cx, cy, nch;                        // chunk columns / rows per picture, chunks in all
    int mtiles, ctiles, S;
};

// byte offset of the 8-byte piece (pixel slot `slot`, channel quad cq of 8) inside one term of the B image: 64 bytes per pixel,
// the quad position XOR-ed with the slot's low bits -- the staging stores of consecutive pixels then fall on different banks, and a
// transposing read (4 consecutive slots x 8 quads per 32-lane half) still covers 256 contiguous-equivalent bytes without a conflict.
// Offsets that are multiples of 8 slots (a patch row: kW3PW; a 16-k step: 16) leave the XOR term alone, so they are immediates.
__device__ __forceinline__ int w3_bslot(int slot, int cq) { return slot * 64 + ((cq ^ (slot & 7)) << 3); }

// Register budget: 128 (four waves per SIMD), not the 168 that "two workgroups of six waves per CU" would allow on paper: the six
// waves of a workgroup land on the four SIMDs as 2 + 2 + 1 + 1, and a second workgroup may put ITS two pairs on the same SIMDs --
// four waves on one SIMD.  With 147 registers that second workgroup did not fit: one workgroup per CU, every wave of the CU in
// the same phase (SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES = 0.01, MfmaUtil 0.35); at <= 128 any placement fits
// (co-execution 0.10 - 0.15, MfmaUtil 0.49, 352 -> 302 us on the 64 -> 64 layer of the stem).
template <int DIL>
__global__ __attribute__((amdgpu_flat_work_group_size(384, 384), amdgpu_waves_per_eu(4, 4))) void k_wgrad_bx3(const W3P p) {
    constexpr int PR = kW3TH + 2 * DIL, PC = kW3TW + 2 * DIL;          // rows / valid columns of the X patch
    constexpr int PPB = PR * kW3PW;                                     // pixel slots of one term of the B image
    constexpr int TERMB = PPB * 64;                                     // bytes of one term of the B image
    constexpr int TERMA = 8 * kW3AP * 16;                               // bytes of one term of the A image ([8 pixel groups][rows])
    constexpr int NTA = (kW3BM * kW3TH * 4 + kW3Threads - 1) / kW3Threads;      // A tasks per thread: (row, strip row, pixel quad)
    // the patch is staged in pixel quads ALIGNED IN THE PICTURE: columns x0 - 4 .. x0 + 19 (six quads; the taps need x0 - DIL ..
    // x0 + 15 + DIL).  A quad then lies either wholly left of the picture or starts inside it -- no 16-byte load straddles the
    // start of a row's first pixel, where "the element before" belongs to another row (or lies before the tensor)
    constexpr int QX = kW3PW / 4;                                       // 6
    constexpr int NQ = PR * QX * 8;                                     // B tasks of a chunk: (patch row, pixel quad, channel quad) = 288 / 384
    static_assert(NQ <= kW3Threads && PC + (4 - DIL) <= kW3PW, "one B task per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char w3_smem[];
    unsigned char* sA = w3_smem;
    unsigned char* sB = w3_smem + 3 * TERMA;
    const int tid = threadIdx.x;
    const int tiles = p.mtiles * p.ctiles;
    int tile, s;
    if (p.S % 8 == 0) {                     // the workgroups of one chunk range on one XCD: they stream the same pictures through its L2
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        tile = slot % tiles;
        s = (slot / tiles) * 8 + xcd;
    } else {
        tile = blockIdx.x % tiles;
        s = blockIdx.x / tiles;
    }
    const int mt = tile / p.ctiles, ct = tile - mt * p.ctiles;
    const int m0 = mt * kW3BM, c0 = ct * kW3BC;
    const int H = p.H, W = p.W, HW = H * W;

    // ---- staging descriptors (the same for every chunk) ---------------------------------------------------------------------------
    // A: task q = tid + 384 j -> pixel quad pq = q & 3 (4 lanes = the 64 contiguous bytes of a 16-pixel row), strip row and channel
    // row from q >> 2
    int ga[NTA], la[NTA], arow[NTA], acol[NTA];
#pragma unroll
    for (int j = 0; j < NTA; ++j) {
        const int q = tid + j * kW3Threads;
        const int pq = q & 3, rr = q >> 2;                               // rr in [0, 256): strip row r = rr >> 6, channel row m = rr & 63
        const int r = rr >> 6, m = rr & 63;
        const bool real = q < kW3BM * kW3TH * 4 && m0 + m < p.Cout;
        ga[j] = real ? ((m0 + m) * HW + r * W + pq * 4) * 4 : kW3Out;
        la[j] = ((r * 2 + (pq >> 1)) * kW3AP + m) * 16 + (pq & 1) * 8;   // 8-pixel group u = 2 r + (pq >> 1), half pq & 1
        if (q >= kW3BM * kW3TH * 4) la[j] = (8 * kW3AP - 1) * 16;        // (a task beyond the tile: zeros into the pad unit of the last group)
        arow[j] = r;
        acol[j] = pq * 4;
    }
    // B: ONE task per thread = 4 consecutive patch pixels x 4 channels (four 16-byte loads, as the 1x1 forward kernel fetches its
    // operand): thread -> (channel quad cq, patch row py, pixel quad qx); the split pairs channels at the same pixel, so every pixel
    // gives one 8-byte piece [pixel][4 channels] per term.  Threads beyond the NQ tasks store zeros to the unused pad slot.
    // thread -> (channel quad = tid & 7, (patch row, pixel quad) = tid >> 3): the 8 lanes of one pixel quad write the 8 channel
    // quads of a pixel -- 64 contiguous bytes up to the XOR -- so a 16-lane store group covers every bank position twice (2-way).
    // (With the channel quad SLOWEST the 16 lanes of a group wrote pixels 4 apart: (slot & 7) took two values only -- an 8-way bank
    //  conflict on every staging store: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.48 - 0.59, MfmaUtil 0.33.)
    const bool btask = tid < NQ;
    const int bcq = tid & 7, brem = tid >> 3;
    const int bpy = brem / QX, bpx = (brem - bpy * QX) * 4;
    const int bchan = p.Cin - (c0 + 4 * bcq);                            // channels of this task's quad that exist (>= 4: all)
    const bool bvalid = btask && bchan > 0;
    const int gb = ((c0 + 4 * bcq) * HW + bpy * W + bpx) * 4;
    int lb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) lb[i] = btask ? w3_bslot(bpy * kW3PW + bpx + i, bcq) : w3_bslot(PPB - 1, tid & 7);     // (columns 22, 23 of a patch row are never read)
    // ---- MFMA operand addressing ----------------------------------------------------------------------------------------------------
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wm = wave & 1, ty = wave >> 1;                             // M half, tap row
    const int aBase = (h * kW3AP + wm * 32 + l31) * 16;                  // + (term * 8 + 2 r) * AP * 16
    // transposing read: within 16 lanes, lane 4 q + pp supplies the address of (pixel row q of the block, channels 4 pp .. 4 pp + 3);
    // lane i receives channel i of the four pixels.  Group g = lane >> 4: channels 16 (g & 1) + ..., k half h = g >> 1.
    int bBase[3][2];
    {
        const int qq = (lane & 15) >> 2, pp = lane & 3, g = lane >> 4;
        const int cqi = 4 * (g & 1) + pp;
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int slot = 8 * (g >> 1) + 4 * jj + qq + (4 + (tx - 1) * DIL) + ty * DIL * kW3PW;      // + r * kW3PW per k step (a multiple of 8)
                bBase[tx][jj] = w3_bslot(slot, cqi);
            }
    }
    f32x16 acc[3];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tx][r] = 0.0f;

    const int cq_ = p.nch / p.S, cr_ = p.nch - cq_ * p.S;                // ranges of cq or cq + 1 chunks (the first cr ranges take one more)
    const int c_lo = s * cq_ + (s < cr_ ? s : cr_), c_hi = c_lo + cq_ + (s < cr_ ? 1 : 0);
    v4f ra[NTA];
    v4f rb[4];                                                            // channel a of the quad: 4 consecutive pixels
    int rows_left = 0, cols_left = 0, bx_left = 0;                        // of the chunk in the staging registers
    const int cpp = p.cx * p.cy;
    auto fetch = [&](int cidx) {
        cidx = __builtin_amdgcn_readfirstlane(cidx);                      // wave-uniform (keeps the resource descriptors in scalar registers)
        const int n = cidx / cpp, rem = cidx - n * cpp;
        const int tyi = rem / p.cx, txi = rem - tyi * p.cx;
        const int y0 = tyi * kW3TH, x0 = txi * kW3TW;
        rows_left = H - y0;
        cols_left = W - x0;
        // dY of picture n from the chunk's first pixel on: every valid (row, quad) lies inside, kW3Out reads zeros
        const int aoff = y0 * W + x0;
        const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + (size_t)n * p.Cout * HW + aoff), 0,
                                                                              (p.Cout * HW - aoff) * 4, kW3RsrcFlags);
#pragma unroll
        for (int j = 0; j < NTA; ++j) ra[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(ares, ga[j], 0, 0));
        // X of picture n; the patch starts DIL rows / columns before the chunk: a pixel outside the plane reads zeros (kW3Out)
        const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)n * p.Cin * HW), 0, p.Cin * HW * 4, kW3RsrcFlags);
        const int boff = ((y0 - DIL) * W + (x0 - 4)) * 4;
        const int hw4 = HW * 4;
        const int iy = y0 - DIL + bpy;
        bx_left = x0 - 4 + bpx;                                           // input column of the quad's first pixel: -4 (the quad left of the picture), or >= 0
        const int vo = (bvalid && (unsigned)iy < (unsigned)H && (unsigned)bx_left < (unsigned)W) ? gb + boff : kW3Out;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int va = a < bchan ? vo : kW3Out;
            rb[a] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(bres, va, a * hw4, 0));
        }
    };
    auto stage = [&]() {
        // dY beyond the plane's rows / columns (partial chunks at the bottom / right edge) is zeroed: a zero times the finite value on
        // the X side contributes nothing; selects, no branch in the loop
#pragma unroll
        for (int j = 0; j < NTA; ++j) {
            const int left = arow[j] < rows_left ? cols_left - acol[j] : 0;
            const float v0 = left > 0 ? ra[j].x : 0.0f, v1 = left > 1 ? ra[j].y : 0.0f, v2 = left > 2 ? ra[j].z : 0.0f, v3 = left > 3 ? ra[j].w : 0.0f;
            unsigned h0, m0_, l0, h1, m1, l1;
            bx_split2(v0, v1, h0, m0_, l0);
            bx_split2(v2, v3, h1, m1, l1);
            unsigned char* dst = sA + la[j];
            *reinterpret_cast<v2u*>(dst) = (v2u){h0, h1};
            *reinterpret_cast<v2u*>(dst + TERMA) = (v2u){m0_, m1};
            *reinterpret_cast<v2u*>(dst + 2 * TERMA) = (v2u){l0, l1};
        }
        // X: a pixel column >= W is the convolution's zero padding (its 16-byte load ran into the next row): selects
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = (unsigned)(bx_left + i) < (unsigned)W;
            const float v0 = ok ? rb[0][i] : 0.0f, v1 = ok ? rb[1][i] : 0.0f, v2 = ok ? rb[2][i] : 0.0f, v3 = ok ? rb[3][i] : 0.0f;
            unsigned h0, m0_, l0, h1, m1, l1;
            bx_split2(v0, v1, h0, m0_, l0);
            bx_split2(v2, v3, h1, m1, l1);
            unsigned char* dst = sB + lb[i];
            *reinterpret_cast<v2u*>(dst) = (v2u){h0, h1};
            *reinterpret_cast<v2u*>(dst + TERMB) = (v2u){m0_, m1};
            *reinterpret_cast<v2u*>(dst + 2 * TERMB) = (v2u){l0, l1};
        }
    };
    typedef __attribute__((address_space(3))) v4s* lds_v4s;
    auto tr_read = [&](int byte_off) -> v4s {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(sB + byte_off));
    };
    auto mfma_chunk = [&]() {
#pragma unroll
        for (int r = 0; r < kW3TH; ++r) {                                 // one 16-k step per row of the chunk
            bf8 a[3];
#pragma unroll
            for (int term = 0; term < 3; ++term)
                a[term] = __builtin_bit_cast(bf8, *reinterpret_cast<const v4f*>(sA + aBase + (term * 8 + 2 * r) * kW3AP * 16));
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                bf8 b[3];
#pragma unroll
                for (int term = 0; term < 3; ++term) {
                    const v4s lo = tr_read(bBase[tx][0] + term * TERMB + r * kW3PW * 64);
                    const v4s hi = tr_read(bBase[tx][1] + term * TERMB + r * kW3PW * 64);
                    b[term] = __builtin_bit_cast(bf8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
                f32x16 c = acc[tx];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
                acc[tx] = c;
            }
        }
    };
    // the pad unit / pad slot that out-of-tile tasks write to are never read; zero-fill nothing else: every unit that is read is
    // written by a task of every chunk
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
    // ---- epilogue: the wave's three accumulator tiles (taps 3 ty + tx) into slice s: part[s][m][c][tap] -----------------------------
    const __amdgpu_buffer_rsrc_t pres = __builtin_amdgcn_make_buffer_rsrc(p.part + (size_t)s * p.Cout * p.Cin * 9, 0, p.Cout * p.Cin * 9 * 4, kW3RsrcFlags);
    const int c = c0 + l31;
    const int mb = m0 + wm * 32 + 4 * h;
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            const int vo = (c < p.Cin && m < p.Cout) ? ((m * p.Cin + c) * 9 + ty * 3 + tx) * 4 : kW3Out;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[tx][r]), pres, vo, 0, 0);
        }
}

__global__ __launch_bounds__(256) void k_w3_reduce(const float* __restrict__ part, int S, size_t n, float* __restrict__ dw) {
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

constexpr size_t kW3PartCap = (size_t)96 << 20;
// S chunk ranges: the chip holds 512 workgroups at a time (2 per CU); a workgroup pays about two chunk times of prologue and epilogue,
// the slices are written once and read once
inline int w3_pick_split(int tiles, int nch, size_t slice_bytes) {
    int best = 1;
    double best_cost = 1e30;
    const int smax = nch < 2048 ? nch : 2048;
    for (int S = 1; S <= smax; ++S) {
        if (S > 1 && (size_t)S * slice_bytes > kW3PartCap) break;
        const long long wg = (long long)tiles * S;
        const double rounds = wg <= 512 ? 1.0 : (double)wg / 512.0;
        const double per = (double)((nch + S - 1) / S);
        const double cost = rounds * (per + 2.0) * 3.0 + (double)S * (double)slice_bytes * 2.0 / 3e6;      // microseconds
        if (cost < best_cost) {
            best_cost = cost;
            best = S;
        }
    }
    return best;
}
}  // namespace

extern "C" int mas_conv_wgrad_bx3_supported(int N, int Cin, int H, int W, int Cout, int dil) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (dil != 1 && dil != 2)) return 0;
    const long long HW = (long long)H * W;
    if ((long long)Cin * HW * 4 >= 0x7fffffffLL || (long long)Cout * HW * 4 >= 0x7fffffffLL || (long long)Cin * Cout * 36 >= 0x7fffffffLL) return 0;
    return 1;
}

extern "C" size_t mas_conv_wgrad_bx3_workspace_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    const size_t slice = sizeof(float) * 9 * (size_t)Cout * Cin;
    size_t smax = kW3PartCap / slice;
    if (smax < 1) smax = 1;
    if (smax > 2048) smax = 2048;
    return smax * slice;
}

extern "C" int mas_conv_wgrad_bx3(const float* x, const float* dy, int N, int Cin, int H, int W, int Cout, int dil, float* dw, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || !workspace) return MAS_ERR_NULL;
    if (!mas_conv_wgrad_bx3_supported(N, Cin, H, W, Cout, dil)) return MAS_ERR_SHAPE;
    if ((uintptr_t)x % 4 != 0 || (uintptr_t)dy % 4 != 0) return MAS_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    W3P p;
    p.x = x; p.dy = dy; p.part = static_cast<float*>(workspace);
    p.N = N; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
    p.cx = (W + kW3TW - 1) / kW3TW;
    p.cy = (H + kW3TH - 1) / kW3TH;
    p.nch = N * p.cx * p.cy;
    p.mtiles = (Cout + kW3BM - 1) / kW3BM;
    p.ctiles = (Cin + kW3BC - 1) / kW3BC;
    const size_t slice = sizeof(float) * 9 * (size_t)Cout * Cin;
    p.S = w3_pick_split(p.mtiles * p.ctiles, p.nch, slice);
    while (p.S > 1 && (size_t)p.S * slice > workspace_bytes) --p.S;
    if ((size_t)p.S * slice > workspace_bytes) return MAS_ERR_WORKSPACE;
    const int pr = kW3TH + 2 * dil;
    const size_t smem = (size_t)3 * 8 * kW3AP * 16 + (size_t)3 * pr * kW3PW * 64;
    const long long nblk = (long long)p.S * p.mtiles * p.ctiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return MAS_ERR_SHAPE;
    if (dil == 1) hipLaunchKernelGGL(k_wgrad_bx3<1>, dim3((unsigned)nblk), dim3(kW3Threads), smem, st, p);
    else hipLaunchKernelGGL(k_wgrad_bx3<2>, dim3((unsigned)nblk), dim3(kW3Threads), smem, st, p);
    int rc = mas_launch_status();
    if (rc != 0) return rc;
    const size_t n = (size_t)Cout * Cin * 9;
    hipLaunchKernelGGL(k_w3_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p.part, p.S, n, dw);
    return mas_launch_status();
}
