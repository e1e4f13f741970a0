// Shared device/host helpers for libvipcup_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "vipcup_hip.h"

// Experiments: kernels that were built, verified and measured SLOWER than what they would replace (the depthwise convolution on the
// matrix cores, the fused MBConv front half, the pipelined persistent window attention) are compiled only with
// VIP_BUILD_EXPERIMENTS=1 (build.py); the default library holds what the default step launches.
#ifndef VIP_BUILD_EXPERIMENTS
#define VIP_BUILD_EXPERIMENTS 0
#endif

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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
// partials != NULL: the pooling form (per-workgroup partial sums of the outputs, `parts` rows per image = vip_dwconv_tiled_parts)
int vip_dwconv_tiled(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int k,
                     int pt, int pl, int Ho, int Wo, int act, hipStream_t s, float* partials = nullptr, int parts = 0);
int vip_dwconv_tiled_parts(int B, int H, int W, int C, int k, int Ho, int Wo);
// the same on the packed STRICT storage (dwconv.hip); returns 1 when the shape is not handled there
int vip_dwconv_tiled_h2(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int k, int pt, int pl, int Ho,
                        int Wo, int act, int* status, hipStream_t s);
// attention cores of the packed STRICT storage on the matrix cores (attn_h2.hip); return 1 when the configuration is not built there
int vip_window_attn_h2_mfma(const void* qkv, const void* q_global, const float* bias_table, void* out, int B, int Hp, int Wp, int C, int heads,
                            int ws, int nq, float scale, int* status, hipStream_t s);
int vip_mhsa_h2_mfma(const void* qkv, void* out, int B, int N, int D, int heads, float scale, int* status, hipStream_t s);
// stride-1 7x7 / 5x5 depthwise on the matrix cores (dwconv_mfma.hip); returns 1 when the shape is not handled there
int vip_dwconv_mfma(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int k,
                    int pt, int pl, int Ho, int Wo, int act, hipStream_t s);

// the 14 x 14-window configuration (C = 256, 8 heads) of vip_gcvit_attn_block_f16 (gcvit_block14.hip); arguments validated by the caller
int vip_gcvit_attn_block14(const void* x, const void* q_global, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* wqkv,
                           int ldwq, const float* bqkv, const void* wproj, int ldwp, const float* bproj, const float* table, void* y,
                           int B, int Hp, int Wp, float scale, hipStream_t s);

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

// ---- two-at-a-time activations: written on float2 so that hipcc emits v_pk_fma_f32 / v_pk_mul_f32 (two fp32
// lanes per VALU slot).  GELU uses  gelu(x) = relu(x) - 0.5|x| erfc(|x|/sqrt2)  with
// erfc(z) ~= (1 + a1 z + ... + a6 z^6)^-16  (Abramowitz-Stegun 7.1.28, |err| <= 3e-7; 1/sqrt2 folded into the
// coefficients): ONE transcendental (rcp) instead of rcp + exp2 and no sign handling.  Max |err| 7e-7.
__device__ __forceinline__ f32x2 vip_gelu2(f32x2 x) {
    const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    f32x2 p = ax * 5.3829750e-06f + 4.8890636e-05f;
    p = p * ax + 3.8003575e-05f;
    p = p * ax + 3.2776263e-03f;
    p = p * ax + 2.1141006e-02f;
    p = p * ax + 4.9867347e-02f;
    p = p * ax + 1.0f;
    f32x2 r = {__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};
    r = r * r;
    r = r * r;
    r = r * r;
    r = r * r;
    // relu(x) = 0.5x + 0.5|x|  ->  gelu = 0.5x + h (1 - r),  h = 0.5|x|   (all packed, no v_max / canonicalise)
    const f32x2 h = ax * 0.5f;
    const f32x2 u = h - h * r;
    return x * 0.5f + u;
}
__device__ __forceinline__ f32x2 vip_sigmoid2(f32x2 v) {
    const f32x2 t = v * -1.44269504088896f;
    const f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    const f32x2 d = e + 1.0f;
    return (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
template <int ACT>
__device__ __forceinline__ f32x2 vip_act2(f32x2 v) {
    if constexpr (ACT == VIP_ACT_RELU) return (f32x2){fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)};
    else if constexpr (ACT == VIP_ACT_SILU) return v * vip_sigmoid2(v);
    else if constexpr (ACT == VIP_ACT_GELU) return vip_gelu2(v);
    else if constexpr (ACT == VIP_ACT_SIGMOID) return vip_sigmoid2(v);
    else return v;
}

__device__ __forceinline__ float vip_act(float v, int act) {
    switch (act) {
        case VIP_ACT_RELU: return v > 0.f ? v : 0.f;
        case VIP_ACT_SILU: return v * vip_sigmoid(v);
        case VIP_ACT_GELU: return vip_gelu2((f32x2){v, v}).x;
        case VIP_ACT_SIGMOID: return vip_sigmoid(v);
        default: return v;
    }
}

// ---- STRICT path activations (strict_*.hip): fp32 throughout, no fp16 rounding anywhere.  exp and the reciprocal are the hardware
// instructions (v_exp_f32, v_rcp_f32: ~1 ulp each) and erf is Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7) - 3e-7 absolute on an
// activation, two orders below what the strict tests hold an operator to (2e-5) and four below the north-star 1e-3; libm's expf / erff
// and IEEE division cost ~35-60 VALU instructions per element against ~12 here, and the fp32 GELU epilogues of ConvNeXt / ViT / GCViT
// evaluate 10^10 of them per step.  Keras: "gelu" = 0.5 x (1 + erf(x / sqrt 2)), "swish" = x sigmoid(x).
__device__ __forceinline__ float vip_act_strict(float v, int act) {
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

// ---- packed STRICT storage ("h2", entry points ending in _h2) -----------------------------------------------
// An activation / weight value v is stored as TWO fp16 terms, hi = rn16(v) and lo = rn16(v - hi): v ~= hi + lo to 2^-22 relative for
// |v| >= 2^-3 and to 2^-25 absolute below (lo is then an fp16 subnormal; v_mfma_f32_16x16x32_f16 does not flush subnormal inputs -
// profiles/r04_mfma_f16_subnormal_inputs.log).  Layout: the 8 consecutive channels c0 .. c0+7 (c0 % 8 == 0) of a row are 32 bytes,
// [hi x 8][lo x 8] - 4 bytes per element like fp32, every 16-byte half IS a v_mfma_f32_16x16x32_f16 operand fragment (8 consecutive k of
// one pixel), and a channel slice at a multiple of 8 is contiguous.  A contraction is three MFMAs per fragment pair:
// w x ~= w_lo x_hi + w_hi x_lo + w_hi x_hi (w_lo x_lo <= 2^-22 of the product is dropped).  |v| > 65504 does not fit: producers raise
// the caller's status word (VIP_H2_OVERFLOW) instead of storing garbage silently.
constexpr float VIP_H2_MAX = 65504.f;
// v[8] -> (hi, lo) fragments
__device__ __forceinline__ void h2_split8(const float (&v)[8], U4H8& hi, U4H8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi.e[j] = (f16)v[j];
        lo.e[j] = (f16)(v[j] - (float)hi.e[j]);
    }
}
__device__ __forceinline__ void h2_join8(const U4H8& hi, const U4H8& lo, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)hi.e[j] + (float)lo.e[j];
}
// true when any of the values does not fit the fp16 range (NaN / Inf included)
__device__ __forceinline__ bool h2_overflows8(const float (&v)[8]) {
    const float m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))),
                          fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
    bool bad = !(m <= VIP_H2_MAX);
#pragma unroll
    for (int j = 0; j < 8; ++j) bad |= (v[j] != v[j]);          // fmaxf drops NaNs
    return bad;
}
// four consecutive logical elements i .. i+3 (i % 4 == 0) of a packed tensor whose element 0 sits at `base` (32-byte aligned)
__device__ __forceinline__ f32x4 h2_ld4(const void* base, long i) {
    const char* g = reinterpret_cast<const char*>(base) + (i >> 3) * 32 + ((i >> 2) & 1) * 8;
    const f16x4 h = *reinterpret_cast<const f16x4*>(g), l = *reinterpret_cast<const f16x4*>(g + 16);
    return (f32x4){(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]};
}
__device__ __forceinline__ void h2_st4(void* base, long i, f32x4 v, int* status) {
    char* g = reinterpret_cast<char*>(base) + (i >> 3) * 32 + ((i >> 2) & 1) * 8;
    f16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = (f16)v[j];
        l[j] = (f16)(v[j] - (float)h[j]);
    }
    *reinterpret_cast<f16x4*>(g) = h;
    *reinterpret_cast<f16x4*>(g + 16) = l;
    const float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    if (status && (!(m <= VIP_H2_MAX) || v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3])) *status = VIP_H2_OVERFLOW;
}

// All-reduce (sum) over aligned groups of `lanes` consecutive lanes (8, 16, 32 or 64).  The first four steps are DPP
// adds inside a 16-lane row (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: one VALU instruction each, no LDS
// round trip); only the steps across rows use ds_bpermute (__shfl_xor).  A LayerNorm row is two dependent reductions,
// and with bpermute for every step that chain - not memory - set the kernel's rate.
template <int CTRL>
__device__ __forceinline__ float vip_dpp_add(float v) {
    const int iv = __builtin_bit_cast(int, v);
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float group_allreduce_sum(float v, int lanes) {
    v = vip_dpp_add<0xB1>(v);                 // quad_perm [1,0,3,2]
    v = vip_dpp_add<0x4E>(v);                 // quad_perm [2,3,0,1]
    v = vip_dpp_add<0x141>(v);                // row_half_mirror: 8 lanes
    if (lanes >= 16) v = vip_dpp_add<0x140>(v);   // row_mirror: 16 lanes
    if (lanes >= 32) v += __shfl_xor(v, 16, 64);
    if (lanes >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

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
