// ds_read_b64_tr_b16 lane semantics on gfx950 (cdna_hip_programming.md T10), checked against what k_wgrad_bx3 assumes:
// within a group of 16 consecutive lanes, lane 4q + p supplies the address of (row q, columns 4p .. 4p + 3) of a 4 x 16 block of
// 16-bit elements, and lane i receives column i of the four rows (row q in element q).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    short* s = reinterpret_cast<short*>(smem);
    for (int i = threadIdx.x; i < 2048; i += 64) s[i] = (short)i;          // element (row, col) of a [64 rows][32 cols] image = row * 32 + col
    __syncthreads();
    const int lane = threadIdx.x;
    const int q = (lane & 15) >> 2, p = lane & 3, g = lane >> 4;
    const int row = q + 8 * (g >> 1), col = 16 * (g & 1) + 4 * p;
    __attribute__((address_space(3))) v4s* ptr = (__attribute__((address_space(3))) v4s*)(smem + (row * 32 + col) * 2);
    const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    out[lane * 4 + 0] = r.x; out[lane * 4 + 1] = r.y; out[lane * 4 + 2] = r.z; out[lane * 4 + 3] = r.w;
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, i = lane & 15;
        for (int e = 0; e < 4; ++e) {
            const int want = (8 * (g >> 1) + e) * 32 + 16 * (g & 1) + i;       // row e of the group's block, column i
            if (h[lane * 4 + e] != want) { if (bad < 8) printf("lane %d elem %d: got %d want %d\n", lane, e, h[lane * 4 + e], want); ++bad; }
        }
    }
    printf("tr_probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "as assumed", bad);
    printf("lane 0: %d %d %d %d   lane 5: %d %d %d %d   lane 21: %d %d %d %d   lane 40: %d %d %d %d\n", h[0], h[1], h[2], h[3], h[20], h[21], h[22], h[23],
           h[84], h[85], h[86], h[87], h[160], h[161], h[162], h[163]);
    return bad != 0;
}
