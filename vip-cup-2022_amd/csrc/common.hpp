// Shared device/host helpers for libvipcup_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "vipcup_hip.h"

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- host side error plumbing -------------------------------------------------------------
void vip_set_error(const char* fmt, ...);

#define VIP_REQUIRE(cond, code, ...)          \
    do {                                      \
        if (!(cond)) {                        \
            vip_set_error(__VA_ARGS__);       \
            return (code);                    \
        }                                     \
    } while (0)

static inline int vip_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vip_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return VIP_ERR_LAUNCH;
    }
    return VIP_OK;
}

// stride-1 depthwise fast path (dwconv.hip); returns 1 when the shape is not handled there
int vip_dwconv_tiled(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int C, int k,
                     int pt, int pl, int Ho, int Wo, int act, hipStream_t s);

// ---- device helpers -------------------------------------------------------------------------
// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the fp16 output rounding): 1 rcp + 1 exp2 +
// 7 FMAs, branch-free.  libm's erff is a two-branch polynomial (~50 VALU instructions when lanes diverge) and
// made the GELU epilogue of the MLP GEMMs VALU-bound.
__device__ __forceinline__ float vip_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p = p * t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896f);
    return copysignf(fmaf(-p, e, 1.0f), x);
}
__device__ __forceinline__ float vip_sigmoid(float v) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-v * 1.44269504088896f));
}

__device__ __forceinline__ float vip_act(float v, int act) {
    switch (act) {
        case VIP_ACT_RELU: return v > 0.f ? v : 0.f;
        case VIP_ACT_SILU: return v * vip_sigmoid(v);
        case VIP_ACT_GELU: return 0.5f * v * (1.f + vip_erf(v * 0.70710678118654752f));
        case VIP_ACT_SIGMOID: return vip_sigmoid(v);
        default: return v;
    }
}

union U4H8 {
    uint4 u;
    f16x8 h;
    f16 e[8];
};

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
