// Developer microbenchmark: achievable HBM read bandwidth for the access patterns of the scorer kernels.
// hipcc -O3 --offload-arch=gfx950 tools/micro/bwtest.hip -o /tmp/bwtest && /tmp/bwtest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_linear(const float4* __restrict__ p, size_t n4, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}

// planes: B images x C planes of HW floats; a block handles chunks of `RUN` float4 per lane-run
template <int C, int UNROLL>
__global__ __launch_bounds__(256) void k_planes(const float* __restrict__ z, int HW, int bpi, float* out) {
    const int b = blockIdx.x / bpi, j = blockIdx.x % bpi;
    const float* zb = z + (size_t)b * C * HW;
    float s = 0.f;
    const int chunk = 256 * 4 * UNROLL;
    for (int p0 = j * chunk; p0 < HW; p0 += bpi * chunk) {
        float4 v[UNROLL][C];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) v[u][c] = *reinterpret_cast<const float4*>(zb + (size_t)c * HW + p0 + u * 1024 + threadIdx.x * 4);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) s += v[u][c].x + v[u][c].y + v[u][c].z + v[u][c].w;
    }
    if (s == 12345.678f) out[0] = s;
}

// same planes, but each block sweeps ONE contiguous range of every plane (consecutive iterations stay in the same pages)
template <int C>
__global__ __launch_bounds__(256) void k_planes_blocked(const float* __restrict__ z, int HW, int bpi, float* out) {
    const int b = blockIdx.x / bpi, j = blockIdx.x % bpi;
    const float* zb = z + (size_t)b * C * HW;
    float s = 0.f;
    const int per = HW / bpi;
    for (int p0 = j * per; p0 < (j + 1) * per; p0 += 1024) {
        float4 v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = *reinterpret_cast<const float4*>(zb + (size_t)c * HW + p0 + threadIdx.x * 4);
#pragma unroll
        for (int c = 0; c < C; ++c) s += v[c].x + v[c].y + v[c].z + v[c].w;
    }
    if (s == 12345.678f) out[0] = s;
}

// wide-short tiles: a wave owns ONE row segment of `SEG` pixels and walks along it (contiguous per plane)
template <int C>
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ z, int H, int W, int seg, float* out) {
    const int segs_x = W / seg, groups_y = H / 4;
    int bid = blockIdx.x;
    const int sx = bid % segs_x; bid /= segs_x;
    const int gy = bid % groups_y; const int b = bid / groups_y;
    const int HW = H * W;
    const float* zb = z + (size_t)b * C * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y = gy * 4 + wave;
    float s = 0.f;
    for (int x = sx * seg; x < (sx + 1) * seg; x += 256) {
        float4 v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = *reinterpret_cast<const float4*>(zb + (size_t)c * HW + (size_t)y * W + x + lane * 4);
#pragma unroll
        for (int c = 0; c < C; ++c) s += v[c].x + v[c].y + v[c].z + v[c].w;
    }
    if (s == 12345.678f) out[0] = s;
}

// tile pattern of K3: 4 waves -> 4 rows, lane -> 4 px, 16 rows per block, 256 px wide
template <int C>
__global__ __launch_bounds__(256) void k_tiles(const float* __restrict__ z, int H, int W, float* out) {
    const int tiles_x = W / 256, tiles_y = H / 16;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int b = bid / tiles_y;
    const int HW = H * W;
    const float* zb = z + (size_t)b * C * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    for (int it = 0; it < 4; ++it) {
        const int y = ty * 16 + it * 4 + wave;
        float4 v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = *reinterpret_cast<const float4*>(zb + (size_t)c * HW + (size_t)y * W + tx * 256 + lane * 4);
#pragma unroll
        for (int c = 0; c < C; ++c) s += v[c].x + v[c].y + v[c].z + v[c].w;
    }
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const int B = 4, C = 20, H = 1024, W = 2048, HW = H * W;
    const size_t n = (size_t)B * C * HW;
    float *z[3], *out;
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&z[i], n * 4)); CK(hipMemset(z[i], 1, n * 4)); }
    CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch(z[i % 3]);
        hipDeviceSynchronize();
        float best = 1e9, tot = 0;
        for (int i = 0; i < 12; ++i) {
            hipEventRecord(e0); launch(z[i % 3]); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; tot += ms;
        }
        printf("%-44s avg %7.1f us  best %7.1f us  -> %.2f TB/s (best %.2f)\n", name, tot / 12 * 1e3, best * 1e3, n * 4 / (tot / 12 * 1e-3) / 1e12, n * 4 / (best * 1e-3) / 1e12);
    };
    for (int g : {1024, 2048, 4096, 8192})
        time(("linear grid=" + std::to_string(g)).c_str(), [&](float* p) { hipLaunchKernelGGL(k_linear, dim3(g), dim3(256), 0, 0, (const float4*)p, n / 4, out); });
    for (int bpi : {64, 128, 256, 512})
        time(("planes C=20 u=1 bpi=" + std::to_string(bpi)).c_str(), [&](float* p) { hipLaunchKernelGGL((k_planes<20, 1>), dim3(B * bpi), dim3(256), 0, 0, p, HW, bpi, out); });
    for (int bpi : {64, 128, 256})
        time(("planes C=20 u=2 bpi=" + std::to_string(bpi)).c_str(), [&](float* p) { hipLaunchKernelGGL((k_planes<20, 2>), dim3(B * bpi), dim3(256), 0, 0, p, HW, bpi, out); });
    for (int bpi : {64, 128, 256, 512})
        time(("planes BLOCKED C=20 bpi=" + std::to_string(bpi)).c_str(), [&](float* p) { hipLaunchKernelGGL((k_planes_blocked<20>), dim3(B * bpi), dim3(256), 0, 0, p, HW, bpi, out); });
    for (int seg : {512, 1024, 2048})
        time(("rows 4 x seg=" + std::to_string(seg)).c_str(), [&](float* p) { hipLaunchKernelGGL((k_rows<20>), dim3(B * (W / seg) * (H / 4)), dim3(256), 0, 0, p, H, W, seg, out); });
    time("tiles 16x256 (K3 pattern)", [&](float* p) { hipLaunchKernelGGL((k_tiles<20>), dim3(B * (W / 256) * (H / 16)), dim3(256), 0, 0, p, H, W, out); });
    return 0;
}
