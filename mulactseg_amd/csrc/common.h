// common.h -- launch helpers shared by the .hip translation units of libmulactseg_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mulactseg_hip.h"
#include "detmath.h"

// argument-error codes (negative; positive codes are hipError_t values)
#define MAS_ERR_NULL (-1)
#define MAS_ERR_SHAPE (-2)
#define MAS_ERR_CLASSES (-3)
#define MAS_ERR_DTYPE (-4)
#define MAS_ERR_ALIGN (-5)
#define MAS_ERR_RANGE (-6)
#define MAS_ERR_WORKSPACE (-7)

#define MAS_WAVE 64

typedef unsigned long long mas_u64;   // the type HIP's 64-bit atomics are declared on

static inline int mas_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

template <typename IdT>
__device__ __forceinline__ int mas_load_id(const IdT* p, size_t i) {
    return (int)p[i];
}

// Per-pixel softmax(z * invT) over CT register-resident channels; `C` live channels (C == CT when EXACT).
// Operation order is normative (mirrored by oracle/exact.c:softmax_row):
//   x_c = z_c*invT ; m = max x ; e_c = exp(x_c - m) ; sum = ((e_0+e_1)+e_2)... ; p_c = e_c * (1/sum)
template <int CT, bool EXACT>
__device__ __forceinline__ void mas_softmax_regs(float (&x)[CT], int C, float invT) {
    const int Cn = EXACT ? CT : C;
    float m = x[0] * invT;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (EXACT || c < Cn) {
            x[c] = x[c] * invT;
            m = (x[c] > m) ? x[c] : m;
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (EXACT || c < Cn) {
            x[c] = mas_expf(x[c] - m);
            sum = (c == 0) ? x[c] : (sum + x[c]);
        }
    }
    const float rinv = 1.0f / sum;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (EXACT || c < Cn) x[c] = x[c] * rinv;
    }
}
