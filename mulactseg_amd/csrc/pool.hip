// pool.hip -- the stem's MaxPool2d(kernel 3, stride 2, padding 1) (models/segmentation/backbone/resnet.py:171,206).
//
// ATen's kernel writes an int64 arg-max index per output (twice the bytes of the output itself: 604 MB for a
// [4,128,512,1024] map) and its backward scatters through those indices.  Here the forward stores a ONE-BYTE window
// offset (0..8, row-major inside the 3x3 window, first maximum wins like ATen's strict '>' scan), and the backward is a
// gather: every input pixel looks at the <= 4 windows that contain it and adds the gradients of those whose recorded
// arg-max is this pixel, in a fixed order -- no atomics, bit-identical from run to run.
#include "common.h"

namespace {
constexpr int kThreads = 256;

// grid (ceil(Wo / (4 * 256)), Ho, N*C): a thread produces four consecutive outputs of one row from columns 8j-1 .. 8j+7
// of three input rows (two aligned 16-B loads and one scalar per row when W % 4 == 0)
__global__ __launch_bounds__(kThreads) void k_maxpool3s2_fwd(const float* __restrict__ x, int H, int W, int Ho, int Wo, float* __restrict__ y,
                                                              unsigned char* __restrict__ arg) {
    const int ox0 = (blockIdx.x * kThreads + threadIdx.x) * 4;
    if (ox0 >= Wo) return;
    const int oy = blockIdx.y;
    const size_t nc = blockIdx.z;
    const float* p = x + nc * H * W;
    const float ninf = -__builtin_inff();
    float best[4] = {ninf, ninf, ninf, ninf};
    int where[4] = {-1, -1, -1, -1};
    const int c0 = ox0 * 2;                          // first even column; the window of output k spans c0 + 2k - 1 .. c0 + 2k + 1
    // 16-byte loads at 4-byte aligned addresses run at the aligned rate on gfx950 (tools/micro/unaligned.hip): rows of any width
    // (the 385-wide plane of the 769 crop) take them; only the group astride the row's end goes element by element
    typedef float mp_v4u __attribute__((ext_vector_type(4), aligned(4)));
    const bool vec = c0 + 7 < W;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int iy = oy * 2 - 1 + a;
        if (iy < 0 || iy >= H) continue;
        const float* row = p + (size_t)iy * W;
        float v[9];                                   // columns c0 - 1 .. c0 + 7
        bool in[9];
        if (vec) {
            const mp_v4u q0 = *reinterpret_cast<const mp_v4u*>(row + c0);
            const mp_v4u q1 = *reinterpret_cast<const mp_v4u*>(row + c0 + 4);
            v[1] = q0[0]; v[2] = q0[1]; v[3] = q0[2]; v[4] = q0[3]; v[5] = q1[0]; v[6] = q1[1]; v[7] = q1[2]; v[8] = q1[3];
            in[0] = c0 > 0;
            v[0] = in[0] ? row[c0 - 1] : ninf;
#pragma unroll
            for (int t = 1; t < 9; ++t) in[t] = true;
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ix = c0 - 1 + t;
                in[t] = ix >= 0 && ix < W;
                v[t] = in[t] ? row[ix] : ninf;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int bcol = 0; bcol < 3; ++bcol) {
                const int t = 2 * k + bcol;
                // ATen: maxidx starts at the first valid element; "val > maxval || isnan(val)" moves it
                // (kept as selects on purpose: with an if-block here and a dynamically indexed store loop below, hipcc 7.2
                //  produced the right maxima but a stale `where`; tests/test_pool_gpu.py checks the gradients against PyTorch)
                const bool upd = in[t] && (where[k] < 0 || v[t] > best[k] || v[t] != v[t]);
                best[k] = upd ? v[t] : best[k];
                where[k] = upd ? (a * 3 + bcol) : where[k];
            }
    }
    const size_t o = (nc * Ho + oy) * Wo + ox0;
    if (ox0 + 3 < Wo) {
        *reinterpret_cast<mp_v4u*>(y + o) = (mp_v4u){best[0], best[1], best[2], best[3]};
        if ((Wo & 3) == 0) {
            *reinterpret_cast<uchar4*>(arg + o) = make_uchar4((unsigned char)where[0], (unsigned char)where[1], (unsigned char)where[2], (unsigned char)where[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) arg[o + k] = (unsigned char)where[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (ox0 + k < Wo) { y[o + k] = best[k]; arg[o + k] = (unsigned char)where[k]; }
    }
}

// grid (ceil(ceil(W/2) / 256), ceil(H/2), N*C): a thread owns the 2x2 input block (2k..2k+1, 2j..2j+1); the windows that
// can select one of its pixels are (k..k+1, j..j+1).  Window (oy,ox) covers rows 2oy-1..2oy+1, so inside it pixel
// (2k,2j) of window (k,j) has offset 4, (2k,2j+1): 5 of (k,j) and 3 of (k,j+1), (2k+1,2j): 7 of (k,j) and 1 of (k+1,j),
// (2k+1,2j+1): 8 of (k,j), 6 of (k,j+1), 2 of (k+1,j), 0 of (k+1,j+1).  Sums run in that order.
__global__ __launch_bounds__(kThreads) void k_maxpool3s2_bwd(const float* __restrict__ g, const unsigned char* __restrict__ arg, int H, int W,
                                                              int Ho, int Wo, float* __restrict__ dx) {
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (2 * j >= W) return;
    const int k = blockIdx.y;
    const size_t nc = blockIdx.z;
    float gv[2][2];
    int av[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dxx = 0; dxx < 2; ++dxx) {
            const int oy = k + dy, ox = j + dxx;
            const bool ok = oy < Ho && ox < Wo;
            const size_t o = (nc * Ho + (ok ? oy : 0)) * Wo + (ok ? ox : 0);
            gv[dy][dxx] = ok ? g[o] : 0.0f;
            av[dy][dxx] = ok ? (int)arg[o] : -1;
        }
    auto pick = [&](int dy, int dxx, int off) { return av[dy][dxx] == off ? gv[dy][dxx] : 0.0f; };
    const float d00 = pick(0, 0, 4);
    const float d01 = pick(0, 0, 5) + pick(0, 1, 3);
    const float d10 = pick(0, 0, 7) + pick(1, 0, 1);
    const float d11 = ((pick(0, 0, 8) + pick(0, 1, 6)) + pick(1, 0, 2)) + pick(1, 1, 0);
    const int iy = 2 * k, ix = 2 * j;
    float* r0 = dx + (nc * H + iy) * W + ix;
    const bool has_x1 = ix + 1 < W, has_y1 = iy + 1 < H;
    if (has_x1 && (W & 1) == 0) {
        *reinterpret_cast<float2*>(r0) = make_float2(d00, d01);
        if (has_y1) *reinterpret_cast<float2*>(r0 + W) = make_float2(d10, d11);
    } else {
        r0[0] = d00;
        if (has_x1) r0[1] = d01;
        if (has_y1) { r0[W] = d10; if (has_x1) r0[W + 1] = d11; }
    }
}
}  // namespace

extern "C" int mas_maxpool3s2_fwd(const float* x, int64_t NC, int H, int W, float* y, uint8_t* arg, void* stream) {
    if (!x || !y || !arg) return MAS_ERR_NULL;
    if (NC <= 0 || NC > 65535 || H <= 0 || W <= 0 || H > 131070) return MAS_ERR_SHAPE;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(k_maxpool3s2_fwd, dim3((unsigned)((Wo + 4 * kThreads - 1) / (4 * kThreads)), (unsigned)Ho, (unsigned)NC), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), x, H, W, Ho, Wo, y, arg);
    return mas_launch_status();
}

extern "C" int mas_maxpool3s2_bwd(const float* dy, const uint8_t* arg, int64_t NC, int H, int W, float* dx, void* stream) {
    if (!dy || !arg || !dx) return MAS_ERR_NULL;
    if (NC <= 0 || NC > 65535 || H <= 0 || W <= 0 || H > 65535) return MAS_ERR_SHAPE;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int Wb = (W + 1) / 2, Hb = (H + 1) / 2;
    hipLaunchKernelGGL(k_maxpool3s2_bwd, dim3((unsigned)((Wb + kThreads - 1) / kThreads), (unsigned)Hb, (unsigned)NC), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), dy, arg, H, W, Ho, Wo, dx);
    return mas_launch_status();
}
