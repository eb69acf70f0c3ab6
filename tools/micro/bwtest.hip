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


// L2 prefetch experiment.  Column strips of 256 px x 64 rows (the ring kernel's tile): wave w takes rows w, w+4, ...
// MODE 0: real 16-B loads only.  MODE 1: touch loads only (one 4-B load per 128-B line, 160 lines per wave-row: the
// lines land in L2, three VGPRs per row).  MODE 2: real loads of row r plus touches of row r+DIST of the same wave.
// `lds_pad` bytes of dynamic LDS bound the number of resident workgroups per CU.
template <int C, int MODE, int DIST>
__global__ __launch_bounds__(256) void k_strip(const float* __restrict__ z, int H, int W, float* out) {
    extern __shared__ float pad[];
    const int tiles_x = W / 256, tiles_y = H / 64;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int b = bid / tiles_y;
    const int HW = H * W;
    const float* zb = z + (size_t)b * C * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    // touch map: line j = lane + 64 i (i = 0..2) -> plane j / 8, 128-B segment j % 8 of the wave's 1-KB row piece
    size_t toff[3]; bool tok[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = lane + 64 * i;
        tok[i] = j < C * 8;
        const int jj = tok[i] ? j : 0;
        toff[i] = (size_t)(jj / 8) * HW + (jj % 8) * 32;
    }
    for (int it = 0; it < 16; ++it) {
        const int y = ty * 64 + it * 4 + wave;
        const size_t base = (size_t)y * W + tx * 256;
        if (MODE == 1 || MODE == 2) {
            const int yp = (MODE == 2) ? (it + DIST < 16 ? y + DIST * 4 : y) : y;
            const size_t pb = (size_t)yp * W + tx * 256;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (tok[i]) s += zb[pb + toff[i]];
        }
        if (MODE == 0 || MODE == 2) {
            float4 v[C];
#pragma unroll
            for (int c = 0; c < C; ++c) v[c] = *reinterpret_cast<const float4*>(zb + (size_t)c * HW + base + lane * 4);
#pragma unroll
            for (int c = 0; c < C; ++c) s += v[c].x + v[c].y + v[c].z + v[c].w;
        }
    }
    if (s == 12345.678f) out[0] = s + pad[0];
}

// Ring-of-two strip kernel with synthetic per-row arithmetic: WORK dependent FMAs per loaded element (80 elements per
// lane and row), 2 waves/SIMD.  Shows how much of the pure-load rate survives once a wave computes between its loads.
template <int C, int WORK, int WPS>
__global__ __launch_bounds__(256, WPS) void k_ring(const float* __restrict__ z, int H, int W, float* out) {
    const int tiles_x = W / 256, tiles_y = H / 64;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int b = bid / tiles_y;
    const int HW = H * W;
    const float* zb = z + (size_t)b * C * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    auto issue = [&](float4 (&v)[C], int it) {
        const int y = ty * 64 + (it < 16 ? it : 15) * 4 + wave;
        const unsigned off = (unsigned)y * W + tx * 256 + lane * 4;
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = *reinterpret_cast<const float4*>(zb + (size_t)c * HW + off);
    };
    auto consume = [&](float4 (&v)[C]) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float a = v[c].x, b2 = v[c].y, c2 = v[c].z, d = v[c].w;
#pragma unroll
            for (int k = 0; k < WORK; ++k) {
                a = __builtin_fmaf(a, 1.0001f, 0.5f); b2 = __builtin_fmaf(b2, 1.0001f, 0.5f);
                c2 = __builtin_fmaf(c2, 1.0001f, 0.5f); d = __builtin_fmaf(d, 1.0001f, 0.5f);
            }
            s += (a + b2) + (c2 + d);
        }
    };
    float4 A[C], Bv[C];
    issue(A, 0);
#pragma unroll 1
    for (int it = 0; it < 16; it += 2) {
        issue(Bv, it + 1);
        __builtin_amdgcn_sched_barrier(0);
        consume(A);
        __builtin_amdgcn_sched_barrier(0);
        issue(A, it + 2);
        __builtin_amdgcn_sched_barrier(0);
        consume(Bv);
        __builtin_amdgcn_sched_barrier(0);
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
    {
        const int g = B * (W / 256) * (H / 64);
        time("ring2 2w/SIMD work=0", [&](float* p) { hipLaunchKernelGGL((k_ring<20, 0, 2>), dim3(g), dim3(256), 0, 0, p, H, W, out); });
        time("ring2 2w/SIMD work=4  (~400 VALU/row)", [&](float* p) { hipLaunchKernelGGL((k_ring<20, 4, 2>), dim3(g), dim3(256), 0, 0, p, H, W, out); });
        time("ring2 2w/SIMD work=8  (~720 VALU/row)", [&](float* p) { hipLaunchKernelGGL((k_ring<20, 8, 2>), dim3(g), dim3(256), 0, 0, p, H, W, out); });
        time("ring2 2w/SIMD work=12 (~1040 VALU/row)", [&](float* p) { hipLaunchKernelGGL((k_ring<20, 12, 2>), dim3(g), dim3(256), 0, 0, p, H, W, out); });
        time("ring2 2w/SIMD work=16 (~1360 VALU/row)", [&](float* p) { hipLaunchKernelGGL((k_ring<20, 16, 2>), dim3(g), dim3(256), 0, 0, p, H, W, out); });
        time("ring2 2w/SIMD work=20 (~1680 VALU/row)", [&](float* p) { hipLaunchKernelGGL((k_ring<20, 20, 2>), dim3(g), dim3(256), 0, 0, p, H, W, out); });
    }
    for (int pad : {0}) {      // unlimited / 3 / 1 workgroups per CU by LDS
        const int g = B * (W / 256) * (H / 64);
        char nm[96];
        snprintf(nm, sizeof nm, "strip real loads      lds=%dK", pad / 1024);
        time(nm, [&](float* p) { hipLaunchKernelGGL((k_strip<20, 0, 0>), dim3(g), dim3(256), pad, 0, p, H, W, out); });
        snprintf(nm, sizeof nm, "strip touch only      lds=%dK", pad / 1024);
        time(nm, [&](float* p) { hipLaunchKernelGGL((k_strip<20, 1, 0>), dim3(g), dim3(256), pad, 0, p, H, W, out); });
        snprintf(nm, sizeof nm, "strip real+touch d=2  lds=%dK", pad / 1024);
        time(nm, [&](float* p) { hipLaunchKernelGGL((k_strip<20, 2, 2>), dim3(g), dim3(256), pad, 0, p, H, W, out); });
        snprintf(nm, sizeof nm, "strip real+touch d=4  lds=%dK", pad / 1024);
        time(nm, [&](float* p) { hipLaunchKernelGGL((k_strip<20, 2, 4>), dim3(g), dim3(256), pad, 0, p, H, W, out); });
        snprintf(nm, sizeof nm, "strip real+touch d=8  lds=%dK", pad / 1024);
        time(nm, [&](float* p) { hipLaunchKernelGGL((k_strip<20, 2, 8>), dim3(g), dim3(256), pad, 0, p, H, W, out); });
    }
    return 0;
}
