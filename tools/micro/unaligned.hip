// Do 16-byte global loads at 4-byte aligned addresses work on gfx950, and at what rate?   hipcc --offload-arch=gfx950 -O3 unaligned.hip -o unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v4fu __attribute__((ext_vector_type(4), aligned(4)));
__global__ void k_copy(const float* __restrict__ src, float* __restrict__ dst, size_t n4, int shift) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        const v4fu v = *reinterpret_cast<const v4fu*>(src + 4 * i + shift);
        *reinterpret_cast<v4f*>(dst + 4 * i) = (v4f){v[0], v[1], v[2], v[3]};
    }
}
int main() {
    const size_t n4 = 64u << 20;            // 1 GiB of floats read
    float *src, *dst;
    hipMalloc(&src, (n4 * 4 + 16) * 4);
    hipMalloc(&dst, n4 * 4 * 4);
    std::vector<float> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        k_copy<<<256 * 16, 256>>>(src, dst, n4, shift);
        hipEventRecord(a);
        for (int r = 0; r < 5; ++r) k_copy<<<256 * 16, 256>>>(src, dst, n4, shift);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<float> o(64);
        hipMemcpy(o.data(), dst, 64 * 4, hipMemcpyDeviceToHost);
        bool ok = true;
        for (int i = 0; i < 64; ++i) ok = ok && o[i] == (float)(i + shift);
        printf("shift %d: %s, %.1f GB/s (read + write)\n", shift, ok ? "correct" : "WRONG", 5.0 * n4 * 32 / ms / 1e6);
    }
    return 0;
}
