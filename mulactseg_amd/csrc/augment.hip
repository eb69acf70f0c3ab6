// augment.hip -- training-time geometry on the device: random scale (Pillow BILINEAR for the
// picture, NEAREST for label / superpixel maps), pad-if-needed, random crop, horizontal flip, to-float, normalise --
// reference dataloader/transform.py:105-113 with dataloader/ext_transforms.py:172-192, 443-520, 323-341, 384-437.
//
// One output-driven kernel per sample: every output pixel of the crop walks back through flip, crop and padding to a
// pixel (y, x) of the SCALED image and evaluates Pillow's two-pass 8-bit resampling for just that pixel: the vertical
// pass over <= ksize_v rows of the horizontally resampled picture, each of which is <= ksize_h taps of the source row,
// rounded to u8 in between exactly as Pillow's temporary image is.  The fixed-point coefficient tables (22 fractional
// bits) and the nearest-neighbour index tables are computed on the host in double precision, as Pillow computes them
// (dataloader/device_transforms.py); the kernel is integer arithmetic up to the final (v/255 - mean)/std in f32.
// The source picture (6 MB for 1024x2048) and the tables stay in L2; HBM traffic is the 7 MB crop that is written.
#include "common.h"

namespace {
constexpr int kThreads = 256;
constexpr int kPrec = 22;      // Pillow: PRECISION_BITS = 32 - 8 - 2

struct MapArg {
    const void* src;
    void* dst;
    long long pad;
    int in_dtype;       // MAS_ID_I64 / MAS_ID_I32 / MAS_ID_U16 / MAS_MAP_U8
    int out_u8;         // 1: uint8 output, 0: int64 output
};

__device__ __forceinline__ long long load_map(const void* p, int dtype, size_t i) {
    switch (dtype) {
        case MAS_ID_I64: return static_cast<const long long*>(p)[i];
        case MAS_ID_I32: return static_cast<const int*>(p)[i];
        case MAS_ID_U16: return static_cast<const unsigned short*>(p)[i];
        default: return static_cast<const unsigned char*>(p)[i];
    }
}

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ __launch_bounds__(kThreads) void k_train_augment(const unsigned char* __restrict__ img, int H, int W, int th, int tw,
                                                             const int* __restrict__ hb, const int* __restrict__ hk, int hks,
                                                             const int* __restrict__ vb, const int* __restrict__ vk, int vks,
                                                             const int* __restrict__ xidx, const int* __restrict__ yidx, int gap_y,
                                                             int gap_x, int ci, int cj, int flip, int oh, int ow, float m0, float m1,
                                                             float m2, float s0, float s1, float s2, int f0, int f1, int f2,
                                                             MapArg a0, MapArg a1, float* __restrict__ out) {
    const int o = blockIdx.x * kThreads + threadIdx.x;
    if (o >= oh * ow) return;
    const int oy = o / ow, ox = o - oy * ow;
    const int sx = flip ? (ow - 1 - ox) : ox;
    const int y = ci + oy - gap_y, x = cj + sx - gap_x;          // pixel of the scaled image
    const bool inside = (y >= 0 && y < th && x >= 0 && x < tw);
    int r = f0, g = f1, b = f2;
    if (inside) {
        const int x0 = hb[2 * x], xn = hb[2 * x + 1];
        const int y0 = vb[2 * y], yn = vb[2 * y + 1];
        const bool need_h = (tw != W), need_v = (th != H);
        long long ar = 1 << (kPrec - 1), ag = ar, ab = ar;
        const int rows = need_v ? yn : 1;
        for (int t = 0; t < rows; ++t) {
            const int sy = need_v ? (y0 + t) : y;
            const unsigned char* row = img + (size_t)sy * W * 3;
            int hr, hg, hbv;
            if (need_h) {
                int cr = 1 << (kPrec - 1), cg = cr, cb = cr;      // <= 5 taps * 255 * 2^22 fits 32 bits
                for (int u = 0; u < xn; ++u) {
                    const int k = hk[x * hks + u];
                    const unsigned char* px = row + (size_t)(x0 + u) * 3;
                    cr += px[0] * k; cg += px[1] * k; cb += px[2] * k;
                }
                hr = clip8(cr >> kPrec); hg = clip8(cg >> kPrec); hbv = clip8(cb >> kPrec);
            } else {
                const unsigned char* px = row + (size_t)x * 3;
                hr = px[0]; hg = px[1]; hbv = px[2];
            }
            if (need_v) {
                const int k = vk[y * vks + t];
                ar += (long long)hr * k; ag += (long long)hg * k; ab += (long long)hbv * k;
            } else {
                r = hr; g = hg; b = hbv;
            }
        }
        if (need_v) {
            r = clip8((int)(ar >> kPrec)); g = clip8((int)(ag >> kPrec)); b = clip8((int)(ab >> kPrec));
        }
    }
    const size_t plane = (size_t)oh * ow;
    out[o] = ((float)r / 255.0f - m0) / s0;
    out[plane + o] = ((float)g / 255.0f - m1) / s1;
    out[2 * plane + o] = ((float)b / 255.0f - m2) / s2;
    const MapArg* maps[2] = {&a0, &a1};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const MapArg& a = *maps[k];
        if (!a.src) continue;
        const long long v = inside ? load_map(a.src, a.in_dtype, (size_t)yidx[y] * W + xidx[x]) : a.pad;
        if (a.out_u8) static_cast<unsigned char*>(a.dst)[o] = (unsigned char)v;
        else static_cast<long long*>(a.dst)[o] = v;
    }
}
}  // namespace

extern "C" int mas_train_augment(const uint8_t* img, int H, int W, int th, int tw, const int32_t* hbounds, const int32_t* hk, int hks,
                                 const int32_t* vbounds, const int32_t* vk, int vks, const int32_t* xidx, const int32_t* yidx,
                                 int gap_y, int gap_x, int crop_i, int crop_j, int flip, int out_h, int out_w, const float* mean,
                                 const float* std, const uint8_t* fill, const void* map0, int map0_dtype, int64_t pad0, void* out_map0,
                                 int out0_u8, const void* map1, int map1_dtype, int64_t pad1, void* out_map1, int out1_u8,
                                 float* out_img, void* stream) {
    if (!img || !hbounds || !hk || !vbounds || !vk || !xidx || !yidx || !mean || !std || !fill || !out_img) return MAS_ERR_NULL;
    if ((map0 && !out_map0) || (map1 && !out_map1)) return MAS_ERR_NULL;
    if (H <= 0 || W <= 0 || th <= 0 || tw <= 0 || out_h <= 0 || out_w <= 0 || hks <= 0 || vks <= 0 || hks > 9 || vks > 9 ||
        (long long)out_h * out_w > 0x7fffffffLL)
        return MAS_ERR_SHAPE;
    if (gap_y < 0 || gap_x < 0 || crop_i < 0 || crop_j < 0 || crop_i + out_h > th + 2 * gap_y || crop_j + out_w > tw + 2 * gap_x)
        return MAS_ERR_RANGE;
    for (int d : {map0 ? map0_dtype : 0, map1 ? map1_dtype : 0})
        if (d < 0 || d > MAS_MAP_U8) return MAS_ERR_DTYPE;
    MapArg a0{map0, out_map0, (long long)pad0, map0_dtype, out0_u8}, a1{map1, out_map1, (long long)pad1, map1_dtype, out1_u8};
    const int n = out_h * out_w;
    hipLaunchKernelGGL(k_train_augment, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       img, H, W, th, tw, hbounds, hk, hks, vbounds, vk, vks, xidx, yidx, gap_y, gap_x, crop_i, crop_j, flip, out_h, out_w,
                       mean[0], mean[1], mean[2], std[0], std[1], std[2], (int)fill[0], (int)fill[1], (int)fill[2], a0, a1, out_img);
    return mas_launch_status();
}
