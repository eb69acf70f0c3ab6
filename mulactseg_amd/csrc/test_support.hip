// test_support.hip -- TEST INFRASTRUCTURE, not part of libmulactseg_hip.so: built into tests/libmulactseg_test.so, which only tests/
// load (tests/helpers.py:occupy_cus).  A CU-hogging neighbour kernel for the co-residency tests of the stream-K hand-off
// (tests/test_stream_k_coresidency_gpu.py).
#include "common.h"

// Test-only neighbour kernel: workgroups that hold their CU resources (256 threads, `lds_bytes` of LDS) and wait on the wall
// clock.  Every wave leaves once `ticks` (100 MHz) have passed since ITS OWN start: the grid always drains.
namespace {
__global__ __launch_bounds__(256) void k_test_occupy(unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned occ_smem[];
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        ++n;
    }
    if (sink && n == 0xffffffffu) sink[0] = occ_smem[threadIdx.x];      // (keeps the LDS allocation alive; never taken)
}
}  // namespace

extern "C" int mas_test_occupy(int blocks, int lds_bytes, unsigned long long ticks, void* stream) {
    if (blocks <= 0 || blocks > 65535 || lds_bytes < 0 || lds_bytes > 160 * 1024) return MAS_ERR_RANGE;
    if (ticks > 500000000ull) return MAS_ERR_RANGE;                     // five seconds at most
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_test_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_test_occupy, dim3((unsigned)blocks), dim3(256), (size_t)lds_bytes, static_cast<hipStream_t>(stream), ticks,
                       static_cast<unsigned*>(nullptr));
    return mas_launch_status();
}
