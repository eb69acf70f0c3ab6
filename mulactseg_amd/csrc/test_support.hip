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

// Packed-f32 operand-select probe (tools/pk_opsel_probe.py): every lane evaluates one packed-f32 instruction form and the same two
// results with scalar instructions, on pseudo-random operands, and counts the results that differ, per lane, in `bad[64]`.
typedef float pk2 __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float smul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sadd(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sfma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int MODE>
__global__ __launch_bounds__(256) void k_test_pk_opsel(int iters, unsigned* bad, pk2 u) {
    const int lane = threadIdx.x & 63;
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        float f[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            s = s * 1664525u + 1013904223u;
            f[k] = __uint_as_float(0x3f800000u | (s >> 9)) - 1.5f;       // [-0.5, 0.5)
        }
        pk2 a = {f[0], f[1]}, b = {f[2], f[3]}, c = {f[4], f[5]}, r, e;
        if (MODE == 0) {            // mul, src1 low half for both results
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
            e = (pk2){smul(a.x, b.x), smul(a.y, b.x)};
        } else if (MODE == 1) {     // mul, src1 high half for both results
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            e = (pk2){smul(a.x, b.y), smul(a.y, b.y)};
        } else if (MODE == 2) {     // mul, src0 low half for both results
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            e = (pk2){smul(a.x, b.x), smul(a.x, b.y)};
        } else if (MODE == 3) {     // mul, no operand select
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            e = (pk2){smul(a.x, b.x), smul(a.y, b.y)};
        } else if (MODE == 4) {     // mul, src0 high half for both results
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
            e = (pk2){smul(a.y, b.x), smul(a.y, b.y)};
        } else if (MODE == 5) {     // add, src1 halves swapped
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
            e = (pk2){sadd(a.x, b.y), sadd(a.y, b.x)};
        } else if (MODE == 6) {     // add, src1 high half for both results
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            e = (pk2){sadd(a.x, b.y), sadd(a.y, b.y)};
        } else if (MODE == 7) {     // fma, src0 high half for both results
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
            e = (pk2){sfma(a.y, b.x, c.x), sfma(a.y, b.y, c.y)};
        } else if (MODE == 8) {     // fma, src1 high half for both results
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
            e = (pk2){sfma(a.x, b.y, c.x), sfma(a.y, b.y, c.y)};
        } else if (MODE == 9) {     // fma, src2 high half for both results
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
            e = (pk2){sfma(a.x, b.x, c.y), sfma(a.y, b.y, c.y)};
        } else if (MODE == 10) {    // fma, src1 in scalar registers, its high half for both results
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(a), "s"(u), "v"(c));
            e = (pk2){sfma(a.x, u.y, c.x), sfma(a.y, u.y, c.y)};
        } else if (MODE == 11) {    // mul, src0 in scalar registers, its high half for both results
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "s"(u), "v"(b));
            e = (pk2){smul(u.y, b.x), smul(u.y, b.y)};
        } else if (MODE == 12) {    // mov: low result from src0's high half, high result from src1's low half
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
            e = (pk2){a.y, b.x};
        } else {                    // mul, src1 high half for the LOW result only (halves swapped)
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
            e = (pk2){smul(a.x, b.y), smul(a.y, b.x)};
        }
        nbad += (__float_as_uint(r.x) != __float_as_uint(e.x)) + (__float_as_uint(r.y) != __float_as_uint(e.y));
    }
    if (nbad) atomicAdd(&bad[lane], nbad);
}

// Neighbours for the probe: workgroups that keep ONE kind of unit busy for `iters` trips -- the matrix cores (KIND 0), the LDS
// (KIND 1: 16-byte reads) or the vector ALUs (KIND 2: packed f32 fma without operand select).
template <int KIND>
__global__ __launch_bounds__(256) void k_test_unit_busy(int iters, float* sink) {
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    __syncthreads();
    float out = 0.f;
    if (KIND == 0) {
        v16f acc0 = {}, acc1 = {};
        v8bf fa, fb;
#pragma unroll
        for (int k = 0; k < 8; ++k) { fa[k] = (__bf16)(0.01f * (float)(threadIdx.x + k)); fb[k] = (__bf16)(0.02f * (float)(k + 1)); }
        for (int it = 0; it < iters; ++it) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc1, 0, 0, 0);
        }
        out = acc0[0] + acc1[3];
    } else if (KIND == 1) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned idx = threadIdx.x;
        for (int it = 0; it < iters; ++it) {
            const float4 v = lds[idx & 255];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            idx = idx * 5u + 1u;
        }
        out = t.x + t.y + t.z + t.w;
    } else if (KIND == 2) {
        pk2 x = {1.0f, 2.0f}, y = {0.999f, 1.001f}, z = {(float)threadIdx.x, 1.f};
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(z) : "v"(y), "v"(x));
        }
        out = x.x + x.y + z.x + z.y;
    } else if (KIND == 3) {             // v_permlane16_swap_b32
        unsigned x = threadIdx.x, y = threadIdx.x * 3u;
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
            x += 1u;
        }
        out = (float)(x ^ y);
    } else if (KIND == 4) {             // v_add_f32_dpp row_shr:1
        float x = (float)threadIdx.x;
        for (int it = 0; it < iters; ++it) {
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
            asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(x));
        }
        out = x;
    } else if (KIND == 5) {             // v_cvt_pk_bf16_f32
        float x = (float)threadIdx.x, y = 1.5f;
        unsigned r = 0;
        for (int it = 0; it < iters; ++it) {
            unsigned t;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(y));
            r ^= t;
            x += 1.0f;
        }
        out = (float)r;
    } else if (KIND == 6) {             // matrix cores at full rate (four independent accumulators) fed by 16-byte LDS reads
        v16f acc[4] = {};
        unsigned idx = threadIdx.x;
        for (int it = 0; it < iters; ++it) {
            const float4 v0 = lds[idx & 255], v1 = lds[(idx + 64) & 255];
            v8bf fa, fb;
            __builtin_memcpy(&fa, &v0, 16);
            __builtin_memcpy(&fb, &v1, 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[k], 0, 0, 0);
            idx = idx * 5u + 1u;
        }
        out = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else if (KIND == 7) {             // v_permlane32_swap_b32
        unsigned x = threadIdx.x, y = threadIdx.x * 3u;
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
            x += 1u;
        }
        out = (float)(x ^ y);
    } else if (KIND == 9) {             // 4-byte global stores (into the 1 MB buffer)
        unsigned idx = blockIdx.x * 256u + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
            sink[idx & 262143u] = (float)it;
            idx += 4099u;
        }
        out = 0.f;
    } else if (KIND == 10) {            // 16-byte LDS writes
        for (int it = 0; it < iters; ++it) {
            lds[(threadIdx.x + it) & 255] = make_float4((float)it, 1.f, 2.f, 3.f);
            asm volatile("" ::: "memory");
        }
        __syncthreads();
        out = lds[threadIdx.x].x;
    } else if (KIND == 11) {            // matrix cores with the accumulators in the AGPR half of the register file
        v8bf fa, fb;
#pragma unroll
        for (int k = 0; k < 8; ++k) { fa[k] = (__bf16)(0.01f * (float)(threadIdx.x + k)); fb[k] = (__bf16)(0.02f * (float)(k + 1)); }
        v16f acc0 = {}, acc1 = {};
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %2, %1" : "+a"(acc0), "+a"(acc1) : "v"(fa), "v"(fb));
        }
        out = acc0[0] + acc1[3];
    } else if (KIND == 12) {            // workgroup barriers
        for (int it = 0; it < iters; ++it) __syncthreads();
        out = 0.f;
    } else {                            // 16-byte global loads (the probe's `sink` is the source: 1 MB, re-read)
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* src = reinterpret_cast<const float4*>(sink);
        unsigned idx = blockIdx.x * 256u + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
            const float4 v = src[idx & 65535u];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            idx += 4099u;
        }
        out = t.x + t.y + t.z + t.w;
    }
    if (sink && out == 123.456f) sink[0] = out;        // (keeps the loop alive; practically never taken)
}
}  // namespace

extern "C" int mas_test_pk_opsel(int mode, int blocks, int iters, unsigned* bad64, void* stream) {
    if (blocks <= 0 || blocks > 65535 || iters <= 0 || iters > (1 << 20) || !bad64) return MAS_ERR_RANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const pk2 u = {0.37f, -0.21f};
#define MAS_PK_CASE(M) case M: hipLaunchKernelGGL(k_test_pk_opsel<M>, dim3((unsigned)blocks), dim3(256), 0, st, iters, bad64, u); break;
    switch (mode) {
        MAS_PK_CASE(0) MAS_PK_CASE(1) MAS_PK_CASE(2) MAS_PK_CASE(3) MAS_PK_CASE(4) MAS_PK_CASE(5) MAS_PK_CASE(6) MAS_PK_CASE(7)
        MAS_PK_CASE(8) MAS_PK_CASE(9) MAS_PK_CASE(10) MAS_PK_CASE(11) MAS_PK_CASE(12) MAS_PK_CASE(13)
        default: return MAS_ERR_RANGE;
    }
#undef MAS_PK_CASE
    return mas_launch_status();
}

extern "C" int mas_test_unit_busy(int kind, int blocks, int iters, float* src, void* stream) {   // src: 1 MB of device memory for kind 8, else unused
    if (blocks <= 0 || blocks > 65535 || iters <= 0 || iters > (1 << 24)) return MAS_ERR_RANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (kind) {
        case 0: hipLaunchKernelGGL(k_test_unit_busy<0>, dim3((unsigned)blocks), dim3(256), 0, st, iters, static_cast<float*>(nullptr)); break;
        case 1: hipLaunchKernelGGL(k_test_unit_busy<1>, dim3((unsigned)blocks), dim3(256), 0, st, iters, static_cast<float*>(nullptr)); break;
        case 2: hipLaunchKernelGGL(k_test_unit_busy<2>, dim3((unsigned)blocks), dim3(256), 0, st, iters, static_cast<float*>(nullptr)); break;
        case 3: hipLaunchKernelGGL(k_test_unit_busy<3>, dim3((unsigned)blocks), dim3(256), 0, st, iters, static_cast<float*>(nullptr)); break;
        case 4: hipLaunchKernelGGL(k_test_unit_busy<4>, dim3((unsigned)blocks), dim3(256), 0, st, iters, static_cast<float*>(nullptr)); break;
        case 5: hipLaunchKernelGGL(k_test_unit_busy<5>, dim3((unsigned)blocks), dim3(256), 0, st, iters, static_cast<float*>(nullptr)); break;
        case 6: hipLaunchKernelGGL(k_test_unit_busy<6>, dim3((unsigned)blocks), dim3(256), 0, st, iters, static_cast<float*>(nullptr)); break;
        case 7: hipLaunchKernelGGL(k_test_unit_busy<7>, dim3((unsigned)blocks), dim3(256), 0, st, iters, static_cast<float*>(nullptr)); break;
        case 9: if (!src) return MAS_ERR_NULL; hipLaunchKernelGGL(k_test_unit_busy<9>, dim3((unsigned)blocks), dim3(256), 0, st, iters, src); break;
        case 10: hipLaunchKernelGGL(k_test_unit_busy<10>, dim3((unsigned)blocks), dim3(256), 0, st, iters, static_cast<float*>(nullptr)); break;
        case 11: hipLaunchKernelGGL(k_test_unit_busy<11>, dim3((unsigned)blocks), dim3(256), 0, st, iters, static_cast<float*>(nullptr)); break;
        case 12: hipLaunchKernelGGL(k_test_unit_busy<12>, dim3((unsigned)blocks), dim3(256), 0, st, iters, static_cast<float*>(nullptr)); break;
        case 8: if (!src) return MAS_ERR_NULL; hipLaunchKernelGGL(k_test_unit_busy<8>, dim3((unsigned)blocks), dim3(256), 0, st, iters, src); break;
        default: return MAS_ERR_RANGE;
    }
    return mas_launch_status();
}

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
