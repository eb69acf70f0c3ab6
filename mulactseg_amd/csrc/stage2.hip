// stage2.hip -- K9: stage-2 cosine pseudo-label generation with one-ring propagation, for gfx950.
//
// Reference: trainer/eval_save_cosplbl_prop.py:121-314 (+ ..._includeonehot.py): nested Python loops over the selected
// superpixels, a 2.1 GB full-resolution feature tensor per image (feat_forward upsamples 256 channels to 1024x2048),
// skimage dilation on the CPU per superpixel.  Here:
//   * the per-(superpixel, class) arg-max pixel table comes from mas_partial_loss_fwd (invT = 1, group flags);
//   * features are read from the QUARTER-resolution map and interpolated in registers (align_corners=False, same
//     operation order as F.interpolate) -- the 2.1 GB tensor is never materialised;
//   * k_stage2_assign      nearest prototype of the own superpixel for every valid pixel;
//   * k_stage2_adjacency   3x3-dilation adjacency of superpixels as an S x S bit matrix (one pass over the id map);
//   * k_stage2_propagate   every pixel looks at the valid superpixels adjacent to its own superpixel in ascending id
//                          order; the last one whose prototype similarity passes that prototype's threshold wins --
//                          exactly the result of the reference's sequential overwrite.
// Dot products are sequential fma chains over the channels (normative, mirrored by oracle/exact.c).
#include "common.h"

namespace {
constexpr int kThreads = 256;

struct FeatMap {
    const float* f;
    int Ch, fh, fw, H, W;
};

struct Interp {
    int o00, o01, o10, o11;
    float l0h, l1h, l0w, l1w;
    bool direct;
};

__device__ __forceinline__ Interp make_interp(const FeatMap& m, int y, int x) {
    Interp it;
    it.direct = (m.fh == m.H && m.fw == m.W);
    if (it.direct) {
        it.o00 = y * m.W + x;
        it.o01 = it.o10 = it.o11 = it.o00;
        it.l0h = it.l0w = 1.0f;
        it.l1h = it.l1w = 0.0f;
        return it;
    }
    const float sh = (float)m.fh / (float)m.H, sw = (float)m.fw / (float)m.W;
    float sy = sh * ((float)y + 0.5f) - 0.5f;
    sy = sy < 0.0f ? 0.0f : sy;
    float sx = sw * ((float)x + 0.5f) - 0.5f;
    sx = sx < 0.0f ? 0.0f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < m.fh - 1 ? 1 : 0), x1 = x0 + (x0 < m.fw - 1 ? 1 : 0);
    it.l1h = sy - (float)y0;
    it.l0h = 1.0f - it.l1h;
    it.l1w = sx - (float)x0;
    it.l0w = 1.0f - it.l1w;
    it.o00 = y0 * m.fw + x0; it.o01 = y0 * m.fw + x1;
    it.o10 = y1 * m.fw + x0; it.o11 = y1 * m.fw + x1;
    return it;
}

__device__ __forceinline__ float feat_at(const FeatMap& m, const Interp& it, int k) {
    const float* p = m.f + (size_t)k * m.fh * m.fw;
    if (it.direct) return p[it.o00];
    return it.l0h * (it.l0w * p[it.o00] + it.l1w * p[it.o01]) + it.l1h * (it.l0w * p[it.o10] + it.l1w * p[it.o11]);
}

__global__ __launch_bounds__(kThreads) void k_stage2_gather_protos(FeatMap m, const int* __restrict__ proto_pix, int n_proto,
                                                                    float* __restrict__ P) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (long long)n_proto * m.Ch) return;
    const int j = (int)(i / m.Ch), k = (int)(i - (long long)j * m.Ch);
    const int pix = proto_pix[j];
    const Interp it = make_interp(m, pix / m.W, pix % m.W);
    P[i] = feat_at(m, it, k);
}

// similarities of one pixel to the prototypes [j0, j1): first maximum and "any above its threshold"
__device__ __forceinline__ void proto_scan(const FeatMap& m, const Interp& it, const float* __restrict__ P,
                                           const float* __restrict__ thr, int j0, int j1, float& best, int& arg, bool& ok) {
    for (int j = j0; j < j1; j += 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int n = (j1 - j) < 4 ? (j1 - j) : 4;
        for (int k = 0; k < m.Ch; ++k) {
            const float fx = feat_at(m, it, k);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < n) acc[u] = mas_fmaf(P[(size_t)(j + u) * m.Ch + k], fx, acc[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u < n) {
                if (arg < 0 || acc[u] > best) { best = acc[u]; arg = j + u; }
                if (thr && thr[j + u] < acc[u]) ok = true;
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_stage2_assign(FeatMap m, const long long* __restrict__ spx,
                                                             const unsigned char* __restrict__ mask, int S,
                                                             const int* __restrict__ p_start, const float* __restrict__ P,
                                                             int* __restrict__ nn, float* __restrict__ nn_sim) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= m.H * m.W) return;
    int arg = -1;
    float best = 0.0f;
    if (mask[i]) {
        const long long id = spx[i];
        if (id >= 0 && id < S && p_start[id + 1] > p_start[id]) {
            const Interp it = make_interp(m, i / m.W, i % m.W);
            bool ok = false;
            proto_scan(m, it, P, nullptr, p_start[id], p_start[id + 1], best, arg, ok);
        }
    }
    nn[i] = arg;
    nn_sim[i] = best;
}

__global__ __launch_bounds__(kThreads) void k_stage2_adjacency(const long long* __restrict__ spx, int H, int W, int S,
                                                                const int* __restrict__ p_start, unsigned* __restrict__ adj) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= H * W) return;
    const long long t = spx[i];
    if (t < 0 || t >= S) return;
    const int words = (S + 31) / 32;
    const int y = i / W, x = i % W;
    long long last = -1;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const long long g = spx[(size_t)yy * W + xx];
            if (g < 0 || g >= S || g == last || p_start[g + 1] == p_start[g]) continue;
            last = g;
            const unsigned bit = 1u << (g & 31);
            unsigned* w = &adj[(size_t)t * words + (g >> 5)];
            if (!(*w & bit)) atomicOr(w, bit);
        }
}

__global__ __launch_bounds__(kThreads) void k_stage2_propagate(FeatMap m, const long long* __restrict__ spx, int S,
                                                                const unsigned* __restrict__ adj, const int* __restrict__ p_start,
                                                                const int* __restrict__ p_cls, const float* __restrict__ P,
                                                                const float* __restrict__ thr, const int* __restrict__ nn,
                                                                long long* __restrict__ out) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= m.H * m.W) return;
    int label = 255;
    const long long t = spx[i];
    if (t >= 0 && t < S) {
        const int words = (S + 31) / 32;
        const Interp it = make_interp(m, i / m.W, i % m.W);
        for (int w = 0; w < words; ++w) {
            unsigned bitsw = adj[(size_t)t * words + w];
            while (bitsw) {                       // ascending superpixel id: later neighbours overwrite earlier ones
                const int b = __ffs(bitsw) - 1;
                bitsw &= bitsw - 1;
                const int s = w * 32 + b;
                float best = 0.0f;
                int arg = -1;
                bool ok = false;
                proto_scan(m, it, P, thr, p_start[s], p_start[s + 1], best, arg, ok);
                if (ok) label = p_cls[arg];
            }
        }
    }
    const int own = nn[i];
    if (own >= 0) label = p_cls[own];
    out[i] = label;
}
}  // namespace

static int stage2_check(const float* feat, int Ch, int fh, int fw, int H, int W) {
    if (!feat) return MAS_ERR_NULL;
    if (Ch <= 0 || fh <= 0 || fw <= 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL / 2 || fh > H || fw > W)
        return MAS_ERR_SHAPE;
    return 0;
}

extern "C" int mas_stage2_gather_protos(const float* feat, int Ch, int fh, int fw, int H, int W, const int32_t* proto_pix, int n_proto,
                                        float* P, void* stream) {
    if (int e = stage2_check(feat, Ch, fh, fw, H, W)) return e;
    if (!proto_pix || !P) return MAS_ERR_NULL;
    if (n_proto <= 0) return MAS_ERR_SHAPE;
    const long long n = (long long)n_proto * Ch;
    hipLaunchKernelGGL(k_stage2_gather_protos, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), (FeatMap{feat, Ch, fh, fw, H, W}), proto_pix, n_proto, P);
    return mas_launch_status();
}

extern "C" int mas_stage2_assign(const float* feat, int Ch, int fh, int fw, int H, int W, const int64_t* spx, const uint8_t* mask, int S,
                                 const int32_t* proto_start, const float* P, int32_t* nn_proto, float* nn_sim, void* stream) {
    if (int e = stage2_check(feat, Ch, fh, fw, H, W)) return e;
    if (!spx || !mask || !proto_start || !P || !nn_proto || !nn_sim) return MAS_ERR_NULL;
    hipLaunchKernelGGL(k_stage2_assign, dim3((unsigned)((H * W + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), (FeatMap{feat, Ch, fh, fw, H, W}), reinterpret_cast<const long long*>(spx), mask,
                       S, proto_start, P, nn_proto, nn_sim);
    return mas_launch_status();
}

extern "C" int mas_stage2_adjacency(const int64_t* spx, int H, int W, int S, const int32_t* proto_start, uint32_t* adj, void* stream) {
    if (!spx || !proto_start || !adj) return MAS_ERR_NULL;
    if (H <= 0 || W <= 0 || S <= 0) return MAS_ERR_SHAPE;
    hipLaunchKernelGGL(k_stage2_adjacency, dim3((unsigned)((H * W + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const long long*>(spx), H, W, S, proto_start, adj);
    return mas_launch_status();
}

extern "C" int mas_stage2_propagate(const float* feat, int Ch, int fh, int fw, int H, int W, const int64_t* spx, int S,
                                    const uint32_t* adj, const int32_t* proto_start, const int32_t* proto_cls, const float* P,
                                    const float* thr, const int32_t* nn_proto, int64_t* out, void* stream) {
    if (int e = stage2_check(feat, Ch, fh, fw, H, W)) return e;
    if (!spx || !adj || !proto_start || !proto_cls || !P || !thr || !nn_proto || !out) return MAS_ERR_NULL;
    hipLaunchKernelGGL(k_stage2_propagate, dim3((unsigned)((H * W + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), (FeatMap{feat, Ch, fh, fw, H, W}), reinterpret_cast<const long long*>(spx), S, adj,
                       proto_start, proto_cls, P, thr, nn_proto, reinterpret_cast<long long*>(out));
    return mas_launch_status();
}
