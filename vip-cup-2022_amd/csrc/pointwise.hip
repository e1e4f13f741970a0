// HBM-bound NHWC helpers: pooling, global average pool, SE excite + residual + activation,
// LayerNorm, depthwise convolution.  All of them move 16 bytes (8 halfs) per lane per access and
// accumulate in fp32.
#include "common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// pool2d
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool2d_kernel(const f16* __restrict__ x, f16* __restrict__ y, int B, int H,
                                                     int W, int C8, int ldx, int ldy, int k, int stride, int pt,
                                                     int pl, int Ho, int Wo, int mode) {
    const long total = (long)B * Ho * Wo * C8;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % C8);
        long p = idx / C8;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float acc[8];
        // mode 0: zero padding takes part in the max, so start from the first tap's value (0 if padded)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (mode == 0) ? -3.0e38f : 0.f;
        int cnt = 0;
        for (int r = 0; r < k; ++r) {
            const int hi = ho * stride - pt + r;
            for (int s = 0; s < k; ++s) {
                const int wi = wo * stride - pl + s;
                const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
                U4H8 v;
                v.u = make_uint4(0, 0, 0, 0);
                if (ok) {
                    v.u = *reinterpret_cast<const uint4*>(x + ((long)(b * H + hi) * W + wi) * ldx + c8 * 8);
                    ++cnt;
                }
                if (mode == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], (float)v.e[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += (float)v.e[j];
                }
            }
        }
        float div = 1.f;
        if (mode == 1) div = 1.f / (float)(cnt > 0 ? cnt : 1);
        if (mode == 2) div = 1.f / (float)(k * k);
        U4H8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = (f16)(acc[j] * div);
        *reinterpret_cast<uint4*>(y + ((long)(b * Ho + ho) * Wo + wo) * ldy + c8 * 8) = o.u;
    }
}

// ---------------------------------------------------------------------------------------------
// global average pool: block = (image, 64-channel slab); 8 chunk lanes x 32 pixel lanes
// ---------------------------------------------------------------------------------------------
// y_lo_off != 0: the mean is written as TWO fp16 planes, hi = fp16(v) at y[b][c] and lo = fp16(v - hi) y_lo_off halfs further (rows
// of 2 C halfs): a pooled vector feeds a Dense layer directly - its rounding error is not averaged over pixels by anything downstream.
__global__ __launch_bounds__(256) void gap_kernel(const f16* __restrict__ x, f16* __restrict__ y, int HW, int C,
                                                  int ldx, int y_lo_off) {
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c = c0 + cl * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (c < C) {
        const f16* xb = x + (long)b * HW * ldx + c;
        for (int p = pl; p < HW; p += 32) {
            U4H8 v;
            v.u = *reinterpret_cast<const uint4*>(xb + (long)p * ldx);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v.e[j];
        }
    }
    __shared__ float red[32][65];
#pragma unroll
    for (int j = 0; j < 8; ++j) red[pl][cl * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 32; ++p) s += red[p][threadIdx.x];
        const int cc = c0 + threadIdx.x;
        if (cc < C) {
            const float v = s / (float)HW;
            const f16 hi = (f16)v;
            f16* dst = y + (long)b * (y_lo_off ? 2 * C : C) + cc;
            dst[0] = hi;
            if (y_lo_off) dst[y_lo_off] = (f16)(v - (float)hi);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// y = act(x * scale[b,c] + residual)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_add_act_kernel(const f16* __restrict__ x, const f16* __restrict__ sc,
                                                            const f16* __restrict__ res, f16* __restrict__ y,
                                                            f16* __restrict__ y2, long total8, int HW, int C8, int act,
                                                            int act2, int planes) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total8; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % C8);
        const long pix = idx / C8;
        const int b = (int)(pix / HW);
        U4H8 v, s, r, o;
        v.u = *reinterpret_cast<const uint4*>(x + idx * 8);
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)v.e[j];
        if (sc) {
            s.u = *reinterpret_cast<const uint4*>(sc + ((long)b * planes * C8 + c8) * 8);
            float g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = (float)s.e[j];
            if (planes == 2) {   // split gate: hi + lo
                s.u = *reinterpret_cast<const uint4*>(sc + (((long)b * 2 + 1) * C8 + c8) * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] += (float)s.e[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] *= g[j];
        }
        if (res) {
            r.u = *reinterpret_cast<const uint4*>(res + idx * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += (float)r.e[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = (f16)vip_act(f[j], act);
        *reinterpret_cast<uint4*>(y + idx * 8) = o.u;
        if (y2) {   // second output act2(y) from the ROUNDED y: what a separate launch reading y back would compute
            U4H8 o2;
#pragma unroll
            for (int j = 0; j < 8; ++j) o2.e[j] = (f16)vip_act((float)o.e[j], act2);
            *reinterpret_cast<uint4*>(y2 + idx * 8) = o2.u;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over C: LPR lanes per row (power of two, 8..64), CPL 16-byte chunks per lane
// ---------------------------------------------------------------------------------------------
template <int CPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, f16* __restrict__ y,
                                                        int rows, int C, int lpr, float eps) {
    const int rows_per_block = 256 / lpr;
    const int sub = threadIdx.x % lpr;
    const int row = blockIdx.x * rows_per_block + threadIdx.x / lpr;
    const bool row_ok = row < rows;
    const int C8 = C >> 3;
    float v[CPL][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c8 = sub + i * lpr;
        U4H8 t;
        t.u = make_uint4(0, 0, 0, 0);
        if (row_ok && c8 < C8) t.u = *reinterpret_cast<const uint4*>(x + (long)row * C + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[i][j] = (float)t.e[j];
            sum += v[i][j];
        }
    }
    sum = group_allreduce_sum(sum, lpr);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c8 = sub + i * lpr;
        if (c8 < C8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[i][j] - mean;
                sq += d * d;
            }
        }
    }
    sq = group_allreduce_sum(sq, lpr);
    const float rstd = rsqrtf(sq / (float)C + eps);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c8 = sub + i * lpr;
        if (row_ok && c8 < C8) {
            U4H8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c8 * 8 + j;
                o.e[j] = (f16)((v[i][j] - mean) * rstd * gamma[c] + beta[c]);
            }
            *reinterpret_cast<uint4*>(y + (long)row * C + c8 * 8) = o.u;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// depthwise conv: thread = TW consecutive output columns x 8 channels
// ---------------------------------------------------------------------------------------------
template <int K, int S, int TW>
__global__ __launch_bounds__(256) void dwconv_kernel(const f16* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, f16* __restrict__ y, int B,
                                                     int H, int W, int C8, int pt, int pl, int Ho, int Wo, int act) {
    constexpr int NCOL = (TW - 1) * S + K;
    const int WoT = (Wo + TW - 1) / TW;
    const long total = (long)B * Ho * WoT * C8;
    const int C = C8 * 8;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % C8);
        long p = idx / C8;
        const int wt = (int)(p % WoT);
        p /= WoT;
        const int ho = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const int wo0 = wt * TW;
        float acc[TW][8];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int hi = ho * S - pt + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            U4H8 col[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                const int wi = wo0 * S - pl + q;
                col[q].u = make_uint4(0, 0, 0, 0);
                if ((unsigned)wi < (unsigned)W)
                    col[q].u = *reinterpret_cast<const uint4*>(x + ((long)(b * H + hi) * W + wi) * C + c8 * 8);
            }
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const float4 w0 = *reinterpret_cast<const float4*>(w + (long)(r * K + s) * C + c8 * 8);
                const float4 w1 = *reinterpret_cast<const float4*>(w + (long)(r * K + s) * C + c8 * 8 + 4);
                const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int t = 0; t < TW; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[t][j] += (float)col[t * S + s].e[j] * wv[j];
            }
        }
        float bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = bias ? bias[c8 * 8 + j] : 0.f;
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int wo = wo0 + t;
            if (wo >= Wo) break;
            U4H8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.e[j] = (f16)vip_act(acc[t][j] + bv[j], act);
            *reinterpret_cast<uint4*>(y + ((long)(b * Ho + ho) * Wo + wo) * C + c8 * 8) = o.u;
        }
    }
}

inline unsigned grid_for(long total) {
    long g = (total + 255) / 256;
    if (g > 256L * 32) g = 256L * 32;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int vip_pool2d_nhwc_f16(const void* x, void* y, int B, int H, int W, int C, int ldx, int ldy, int k,
                                   int stride, int pt, int pl, int Ho, int Wo, int mode, void* stream) {
    VIP_REQUIRE(x && y, VIP_ERR_BAD_ARG, "vip_pool2d_nhwc_f16: null pointer");
    VIP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && Ho > 0 && Wo > 0 && pt >= 0 && pl >= 0,
                VIP_ERR_BAD_ARG, "vip_pool2d_nhwc_f16: non-positive dimension");
    VIP_REQUIRE(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C, VIP_ERR_ALIGNMENT,
                "vip_pool2d_nhwc_f16: C/ldx/ldy must be multiples of 8");
    VIP_REQUIRE(mode >= 0 && mode <= 2, VIP_ERR_BAD_ARG, "vip_pool2d_nhwc_f16: mode %d", mode);
    const long total = (long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(pool2d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const f16*)x,
                       (f16*)y, B, H, W, C / 8, ldx, ldy, k, stride, pt, pl, Ho, Wo, mode);
    return vip_launch_status("vip_pool2d_nhwc_f16");
}

extern "C" int vip_global_avgpool_f16(const void* x, void* y, int B, int HW, int C, int ldx, void* stream) {
    VIP_REQUIRE(x && y, VIP_ERR_BAD_ARG, "vip_global_avgpool_f16: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0, VIP_ERR_BAD_ARG, "vip_global_avgpool_f16: non-positive dimension");
    VIP_REQUIRE(C % 8 == 0 && ldx % 8 == 0 && ldx >= C, VIP_ERR_ALIGNMENT,
                "vip_global_avgpool_f16: C/ldx must be multiples of 8");
    hipLaunchKernelGGL(gap_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y,
                       HW, C, ldx, 0);
    return vip_launch_status("vip_global_avgpool_f16");
}

extern "C" int vip_global_avgpool_split_f16(const void* x, void* y, int B, int HW, int C, int ldx, void* stream) {
    VIP_REQUIRE(x && y, VIP_ERR_BAD_ARG, "vip_global_avgpool_split_f16: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0, VIP_ERR_BAD_ARG, "vip_global_avgpool_split_f16: non-positive dimension");
    VIP_REQUIRE(C % 8 == 0 && ldx % 8 == 0 && ldx >= C, VIP_ERR_ALIGNMENT,
                "vip_global_avgpool_split_f16: C/ldx must be multiples of 8");
    hipLaunchKernelGGL(gap_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y,
                       HW, C, ldx, C);
    return vip_launch_status("vip_global_avgpool_split_f16");
}

extern "C" int vip_scale_add_act3_f16(const void* x, const void* scale, int scale_planes, const void* residual, void* y,
                                      void* y2, int B, int HW, int C, int act, int act2, void* stream) {
    VIP_REQUIRE(scale_planes == 1 || scale_planes == 2, VIP_ERR_BAD_ARG, "vip_scale_add_act_f16: scale_planes must be 1 or 2");
    VIP_REQUIRE(x && y, VIP_ERR_BAD_ARG, "vip_scale_add_act_f16: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && (unsigned)act <= 4u && (unsigned)act2 <= 4u, VIP_ERR_BAD_ARG,
                "vip_scale_add_act_f16: bad argument");
    VIP_REQUIRE(C % 8 == 0, VIP_ERR_ALIGNMENT, "vip_scale_add_act_f16: C must be a multiple of 8");
    const long total8 = (long)B * HW * (C / 8);
    hipLaunchKernelGGL(scale_add_act_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, (const f16*)x,
                       (const f16*)scale, (const f16*)residual, (f16*)y, (f16*)y2, total8, HW, C / 8, act, act2, scale_planes);
    return vip_launch_status("vip_scale_add_act_f16");
}

extern "C" int vip_scale_add_act2_f16(const void* x, const void* scale, const void* residual, void* y, void* y2, int B,
                                      int HW, int C, int act, int act2, void* stream) {
    return vip_scale_add_act3_f16(x, scale, 1, residual, y, y2, B, HW, C, act, act2, stream);
}

extern "C" int vip_scale_add_act_f16(const void* x, const void* scale, const void* residual, void* y, int B, int HW,
                                     int C, int act, void* stream) {
    return vip_scale_add_act2_f16(x, scale, residual, y, nullptr, B, HW, C, act, VIP_ACT_NONE, stream);
}

extern "C" int vip_layernorm_f16(const void* x, const float* gamma, const float* beta, void* y, int rows, int C,
                                 float eps, void* stream) {
    VIP_REQUIRE(x && y && gamma && beta, VIP_ERR_BAD_ARG, "vip_layernorm_f16: null pointer");
    VIP_REQUIRE(rows > 0 && C > 0, VIP_ERR_BAD_ARG, "vip_layernorm_f16: non-positive dimension");
    VIP_REQUIRE(C % 8 == 0, VIP_ERR_ALIGNMENT, "vip_layernorm_f16: C must be a multiple of 8");
    const int C8 = C / 8;
    int lpr = 8;
    while (lpr < 64 && lpr < C8) lpr <<= 1;
    const int cpl = (C8 + lpr - 1) / lpr;
    VIP_REQUIRE(cpl <= 4, VIP_ERR_UNSUPPORTED, "vip_layernorm_f16: C=%d too large (max 2048)", C);
    const int rpb = 256 / lpr;
    dim3 grid((rows + rpb - 1) / rpb);
    hipStream_t s = (hipStream_t)stream;
    const f16* xi = (const f16*)x;
    f16* yo = (f16*)y;
    switch (cpl) {
        case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, s, xi, gamma, beta, yo, rows, C, lpr, eps); break;
        case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, s, xi, gamma, beta, yo, rows, C, lpr, eps); break;
        case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, dim3(256), 0, s, xi, gamma, beta, yo, rows, C, lpr, eps); break;
        default: hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, s, xi, gamma, beta, yo, rows, C, lpr, eps); break;
    }
    return vip_launch_status("vip_layernorm_f16");
}

#if !VIP_BUILD_EXPERIMENTS
// the matrix-core depthwise experiment (dwconv_mfma.hip) is not in this build: "not handled here"
int vip_dwconv_mfma(const void*, const float*, const float*, void*, int, int, int, int, int, int, int, int, int, int, hipStream_t) { return 1; }
#endif

extern "C" int vip_dwconv2d_nhwc_f16(const void* x, const float* w, const float* bias, void* y, int B, int H, int W,
                                     int C, int k, int stride, int pt, int pl, int Ho, int Wo, int act, void* stream) {
    VIP_REQUIRE(x && w && y, VIP_ERR_BAD_ARG, "vip_dwconv2d_nhwc_f16: null pointer");
    VIP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && pt >= 0 && pl >= 0 && (unsigned)act <= 4u,
                VIP_ERR_BAD_ARG, "vip_dwconv2d_nhwc_f16: bad argument");
    VIP_REQUIRE(C % 8 == 0, VIP_ERR_ALIGNMENT, "vip_dwconv2d_nhwc_f16: C must be a multiple of 8");
    const int C8 = C / 8;
    hipStream_t s = (hipStream_t)stream;
    const f16* xi = (const f16*)x;
    const float* wi = w;
    f16* yo = (f16*)y;
    if (stride == 1) {
        const int sm = vip_dwconv_mfma(x, w, bias, y, B, H, W, C, k, pt, pl, Ho, Wo, act, s);
        if (sm != 1) return sm;
        const int st = vip_dwconv_tiled(x, w, bias, y, B, H, W, C, k, pt, pl, Ho, Wo, act, s);
        if (st != 1) return st;
    }
#define VIP_DW(KK, SS, TW)                                                                                        \
    {                                                                                                             \
        const long total = (long)B * Ho * ((Wo + TW - 1) / TW) * C8;                                              \
        hipLaunchKernelGGL((dwconv_kernel<KK, SS, TW>), dim3(grid_for(total)), dim3(256), 0, s, xi, wi, bias, yo, \
                           B, H, W, C8, pt, pl, Ho, Wo, act);                                                     \
    }
    if (k == 3 && stride == 1) VIP_DW(3, 1, 4)
    else if (k == 3 && stride == 2) VIP_DW(3, 2, 2)
    else if (k == 5 && stride == 1) VIP_DW(5, 1, 4)
    else if (k == 5 && stride == 2) VIP_DW(5, 2, 2)
    else if (k == 7 && stride == 1) VIP_DW(7, 1, 4)
    else {
        vip_set_error("vip_dwconv2d_nhwc_f16: unsupported k=%d stride=%d", k, stride);
        return VIP_ERR_UNSUPPORTED;
    }
#undef VIP_DW
    return vip_launch_status("vip_dwconv2d_nhwc_f16");
}

extern "C" int vip_dwconv2d_pool_parts(int B, int H, int W, int C, int k, int stride, int Ho, int Wo) {
    if (stride != 1 || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return 0;
    return vip_dwconv_tiled_parts(B, H, W, C, k, Ho, Wo);
}

extern "C" int vip_dwconv2d_pool_nhwc_f16(const void* x, const float* w, const float* bias, void* y, float* partials, int parts,
                                          int B, int H, int W, int C, int k, int stride, int pt, int pl, int Ho, int Wo, int act,
                                          void* stream) {
    VIP_REQUIRE(x && w && y && partials, VIP_ERR_BAD_ARG, "vip_dwconv2d_pool_nhwc_f16: null pointer");
    VIP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && pt >= 0 && pl >= 0 && (unsigned)act <= 4u,
                VIP_ERR_BAD_ARG, "vip_dwconv2d_pool_nhwc_f16: bad argument");
    VIP_REQUIRE(C % 8 == 0, VIP_ERR_ALIGNMENT, "vip_dwconv2d_pool_nhwc_f16: C must be a multiple of 8");
    VIP_REQUIRE(stride == 1 && parts > 0 && parts == vip_dwconv_tiled_parts(B, H, W, C, k, Ho, Wo), VIP_ERR_UNSUPPORTED,
                "vip_dwconv2d_pool_nhwc_f16: k=%d stride=%d parts=%d - ask vip_dwconv2d_pool_parts first", k, stride, parts);
    const int st = vip_dwconv_tiled(x, w, bias, y, B, H, W, C, k, pt, pl, Ho, Wo, act, (hipStream_t)stream, partials, parts);
    if (st == 1) {
        vip_set_error("vip_dwconv2d_pool_nhwc_f16: shape not handled by the tile kernel");
        return VIP_ERR_UNSUPPORTED;
    }
    return st;
}

// ---------------------------------------------------------------------------------------------
// classifier head: global average pool + dense, fp32 out.  One block per image.
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void gap_dense_kernel(const f16* __restrict__ x, const float* __restrict__ Wt,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int HW, int C, int ldx, int N) {
    __shared__ float pooled[4096];
    __shared__ float wsum[4];
    const int b = blockIdx.x;
    const f16* xb = x + (long)b * HW * ldx;
    const float inv = 1.f / (float)HW;
    for (int c8 = threadIdx.x; c8 < (C >> 3); c8 += 256) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int p = 0; p < HW; ++p) {
            U4H8 v;
            v.u = *reinterpret_cast<const uint4*>(xb + (long)p * ldx + c8 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v.e[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) pooled[c8 * 8 + j] = acc[j] * inv;
    }
    __syncthreads();
    for (int n = 0; n < N; ++n) {
        float s = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) s += pooled[c] * Wt[(long)n * C + c];
        s = wave_reduce_sum(s);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) out[(long)b * N + n] = wsum[0] + wsum[1] + wsum[2] + wsum[3] + (bias ? bias[n] : 0.f);
        __syncthreads();
    }
}

// Classifier head with a LayerNorm between pool and Dense (tfimm ConvNeXt convnext.py:432-436, kecam HorNet): mean over the pixels,
// LayerNorm over the channels, Dense - all in fp32, one workgroup per image.  The pooled vector is where a rounding error is NOT
// averaged over pixels any more: rounding it (and the LayerNorm output) to fp16 was 60 % of ConvNeXt-T's logit error.
__global__ __launch_bounds__(256) void gap_ln_dense_kernel(const f16* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, const float* __restrict__ Wt,
                                                           const float* __restrict__ bias, float* __restrict__ out, int HW, int C,
                                                           int ldx, int N) {
    __shared__ float pooled[4096];
    __shared__ float red[2][4];
    const int b = blockIdx.x;
    const f16* xb = x + (long)b * HW * ldx;
    const float inv = 1.f / (float)HW;
    float s1 = 0.f;
    for (int c8 = threadIdx.x; c8 < (C >> 3); c8 += 256) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int p = 0; p < HW; ++p) {
            U4H8 v;
            v.u = *reinterpret_cast<const uint4*>(xb + (long)p * ldx + c8 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v.e[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pooled[c8 * 8 + j] = acc[j] * inv;
            s1 += acc[j] * inv;
        }
    }
    s1 = wave_reduce_sum(s1);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s1;
    __syncthreads();
    const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)C;
    float s2 = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float d = pooled[c] - mean;
        s2 += d * d;
    }
    s2 = wave_reduce_sum(s2);
    if ((threadIdx.x & 63) == 0) red[1][threadIdx.x >> 6] = s2;
    __syncthreads();
    const float rstd = rsqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)C + eps);
    for (int c = threadIdx.x; c < C; c += 256) pooled[c] = (pooled[c] - mean) * rstd * gamma[c] + beta[c];
    __syncthreads();
    for (int n = 0; n < N; ++n) {
        float s = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) s += pooled[c] * Wt[(long)n * C + c];
        s = wave_reduce_sum(s);
        if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) out[(long)b * N + n] = red[0][0] + red[0][1] + red[0][2] + red[0][3] + (bias ? bias[n] : 0.f);
        __syncthreads();
    }
}
}  // namespace

extern "C" int vip_gap_ln_dense_f32(const void* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias,
                                    float* out, int B, int HW, int C, int ldx, int N, void* stream) {
    VIP_REQUIRE(x && gamma && beta && W && out, VIP_ERR_BAD_ARG, "vip_gap_ln_dense_f32: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && N > 0 && eps >= 0.f, VIP_ERR_BAD_ARG, "vip_gap_ln_dense_f32: bad dimension or eps");
    VIP_REQUIRE(C % 8 == 0 && ldx % 8 == 0 && ldx >= C, VIP_ERR_ALIGNMENT, "vip_gap_ln_dense_f32: C/ldx must be multiples of 8");
    VIP_REQUIRE(C <= 4096, VIP_ERR_UNSUPPORTED, "vip_gap_ln_dense_f32: C=%d > 4096", C);
    hipLaunchKernelGGL(gap_ln_dense_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const f16*)x, gamma, beta, eps, W, bias, out,
                       HW, C, ldx, N);
    return vip_launch_status("vip_gap_ln_dense_f32");
}

extern "C" int vip_gap_dense_f32(const void* x, const float* W, const float* bias, float* out, int B, int HW, int C,
                                 int ldx, int N, void* stream) {
    VIP_REQUIRE(x && W && out, VIP_ERR_BAD_ARG, "vip_gap_dense_f32: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && N > 0, VIP_ERR_BAD_ARG, "vip_gap_dense_f32: non-positive dimension");
    VIP_REQUIRE(C % 8 == 0 && ldx % 8 == 0 && ldx >= C, VIP_ERR_ALIGNMENT, "vip_gap_dense_f32: C/ldx must be multiples of 8");
    VIP_REQUIRE(C <= 4096, VIP_ERR_UNSUPPORTED, "vip_gap_dense_f32: C=%d > 4096", C);
    hipLaunchKernelGGL(gap_dense_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const f16*)x, W, bias, out, HW,
                       C, ldx, N);
    return vip_launch_status("vip_gap_dense_f32");
}

// ---------------------------------------------------------------------------------------------
// ResNeSt split-attention combine (kecam resnest/resnest.py:57-61): out[b,p,c] = sum_r x[b,p,r*C+c] * s[b,r*C+c]
// ---------------------------------------------------------------------------------------------
namespace {
// y[m, y_off + c] = a[m, a_off + c] * b[m, b_off + c]: channel slices of wider row-major tensors, 8 channels per thread
__global__ __launch_bounds__(256) void mul_kernel(const f16* __restrict__ a, const f16* __restrict__ b, f16* __restrict__ y,
                                                  long total8, int C8, int lda, int ldb, int ldy) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total8; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % C8);
        const long m = idx / C8;
        U4H8 u, v, o;
        u.u = *reinterpret_cast<const uint4*>(a + m * lda + c8 * 8);
        v.u = *reinterpret_cast<const uint4*>(b + m * ldb + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = (f16)((float)u.e[j] * (float)v.e[j]);
        *reinterpret_cast<uint4*>(y + m * ldy + c8 * 8) = o.u;
    }
}

__global__ __launch_bounds__(256) void radix_combine_kernel(const f16* __restrict__ x, const f16* __restrict__ s,
                                                            f16* __restrict__ y, long total8, int HW, int C8, int radix,
                                                            int planes) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total8; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % C8);
        const long pix = idx / C8;
        const int b = (int)(pix / HW);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int r = 0; r < radix; ++r) {
            U4H8 v, w;
            v.u = *reinterpret_cast<const uint4*>(x + (pix * radix * C8 + (long)r * C8 + c8) * 8);
            w.u = *reinterpret_cast<const uint4*>(s + (((long)b * planes * radix + r) * C8 + c8) * 8);
            float g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = (float)w.e[j];
            if (planes == 2) {   // split weights: hi + lo
                w.u = *reinterpret_cast<const uint4*>(s + ((((long)b * 2 + 1) * radix + r) * C8 + c8) * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] += (float)w.e[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v.e[j] * g[j];
        }
        U4H8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = (f16)acc[j];
        *reinterpret_cast<uint4*>(y + idx * 8) = o.u;
    }
}
}  // namespace

extern "C" int vip_radix_combine2_f16(const void* x, const void* scale, int scale_planes, void* y, int B, int HW, int C,
                                      int radix, void* stream) {
    VIP_REQUIRE(x && scale && y, VIP_ERR_BAD_ARG, "vip_radix_combine_f16: null pointer");
    VIP_REQUIRE(scale_planes == 1 || scale_planes == 2, VIP_ERR_BAD_ARG, "vip_radix_combine_f16: scale_planes must be 1 or 2");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && radix > 0, VIP_ERR_BAD_ARG, "vip_radix_combine_f16: non-positive dimension");
    VIP_REQUIRE(C % 8 == 0, VIP_ERR_ALIGNMENT, "vip_radix_combine_f16: C must be a multiple of 8");
    const long total8 = (long)B * HW * (C / 8);
    hipLaunchKernelGGL(radix_combine_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, (const f16*)x,
                       (const f16*)scale, (f16*)y, total8, HW, C / 8, radix, scale_planes);
    return vip_launch_status("vip_radix_combine_f16");
}

extern "C" int vip_radix_combine_f16(const void* x, const void* scale, void* y, int B, int HW, int C, int radix,
                                     void* stream) {
    return vip_radix_combine2_f16(x, scale, 1, y, B, HW, C, radix, stream);
}

extern "C" int vip_mul_f16(const void* a, const void* b, void* y, long rows, int C, int lda, int a_off, int ldb, int b_off,
                           int ldy, int y_off, void* stream) {
    VIP_REQUIRE(a && b && y, VIP_ERR_BAD_ARG, "vip_mul_f16: null pointer");
    VIP_REQUIRE(rows > 0 && C > 0 && a_off >= 0 && b_off >= 0 && y_off >= 0, VIP_ERR_BAD_ARG, "vip_mul_f16: bad size");
    VIP_REQUIRE(C % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldy % 8 == 0 && a_off % 8 == 0 && b_off % 8 == 0 && y_off % 8 == 0,
                VIP_ERR_ALIGNMENT, "vip_mul_f16: C, leading dimensions and offsets must be multiples of 8 halfs");
    VIP_REQUIRE(a_off + C <= lda && b_off + C <= ldb && y_off + C <= ldy, VIP_ERR_BAD_ARG, "vip_mul_f16: slice exceeds its row");
    const long total8 = rows * (C / 8);
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, (const f16*)a + a_off,
                       (const f16*)b + b_off, (f16*)y + y_off, total8, C / 8, lda, ldb, ldy);
    return vip_launch_status("vip_mul_f16");
}

// ---------------------------------------------------------------------------------------------
// scores: what main.py does with a model's logits (:109-114) and with the members' scores (:142-143)
// ---------------------------------------------------------------------------------------------
namespace {
// one thread per image; N (classes) is 1 or a handful
__global__ __launch_bounds__(256) void head_prob_kernel(const float* __restrict__ z, float* __restrict__ p, float* __restrict__ score,
                                                        int B, int N) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* zb = z + (long)b * N;
    if (N == 1) {
        const float v = 1.f / (1.f + __expf(-zb[0]));
        if (p) p[b] = v;
        if (score) score[b] = v;
        return;
    }
    float m = zb[0];
    for (int n = 1; n < N; ++n) m = fmaxf(m, zb[n]);
    float sum = 0.f;
    for (int n = 0; n < N; ++n) sum += __expf(zb[n] - m);
    const float inv = 1.f / sum;
    if (p)
        for (int n = 0; n < N; ++n) p[(long)b * N + n] = __expf(zb[n] - m) * inv;
    if (score) score[b] = 1.f - __expf(zb[0] - m) * inv;        // multi-class -> binary: 1 - p(class 0)
}

__global__ __launch_bounds__(256) void prob_to_score_kernel(const float* __restrict__ p, float* __restrict__ score, int B, int N) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) score[b] = N == 1 ? p[b] : 1.f - p[(long)b * N];
}

__global__ __launch_bounds__(256) void ensemble_mean_kernel(const float* __restrict__ s, float* __restrict__ out, int M, int n, long ld) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int m = 0; m < M; ++m) acc += s[(long)m * ld + i];
    out[i] = acc / (float)M;
}
}  // namespace

extern "C" int vip_head_prob_f32(const float* logits, float* prob, float* score, int B, int N, void* stream) {
    VIP_REQUIRE(logits && (prob || score), VIP_ERR_BAD_ARG, "vip_head_prob_f32: null pointer");
    VIP_REQUIRE(B > 0 && N > 0, VIP_ERR_BAD_ARG, "vip_head_prob_f32: non-positive dimension");
    hipLaunchKernelGGL(head_prob_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, prob, score, B, N);
    return vip_launch_status("vip_head_prob_f32");
}

namespace {
// head activation other than the default pairing: act 0 = linear (logits out), 1 = elementwise sigmoid (any N), 2 = softmax (any N:
// N = 1 gives 1.0, as Keras does)
__global__ __launch_bounds__(256) void head_act_kernel(const float* __restrict__ z, float* __restrict__ p, int B, int N, int act) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* zb = z + (long)b * N;
    float* pb = p + (long)b * N;
    if (act == 0) {
        for (int n = 0; n < N; ++n) pb[n] = zb[n];
    } else if (act == 1) {
        for (int n = 0; n < N; ++n) pb[n] = 1.f / (1.f + __expf(-zb[n]));
    } else {
        float m = zb[0];
        for (int n = 1; n < N; ++n) m = fmaxf(m, zb[n]);
        float sum = 0.f;
        for (int n = 0; n < N; ++n) sum += __expf(zb[n] - m);
        const float inv = 1.f / sum;
        for (int n = 0; n < N; ++n) pb[n] = __expf(zb[n] - m) * inv;
    }
}
}  // namespace

extern "C" int vip_head_act_f32(const float* logits, float* prob, int B, int N, int act, void* stream) {
    VIP_REQUIRE(logits && prob, VIP_ERR_BAD_ARG, "vip_head_act_f32: null pointer");
    VIP_REQUIRE(B > 0 && N > 0 && act >= 0 && act <= 2, VIP_ERR_BAD_ARG, "vip_head_act_f32: bad dimension or activation code");
    hipLaunchKernelGGL(head_act_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, prob, B, N, act);
    return vip_launch_status("vip_head_act_f32");
}

extern "C" int vip_prob_to_score_f32(const float* prob, float* score, int B, int N, void* stream) {
    VIP_REQUIRE(prob && score, VIP_ERR_BAD_ARG, "vip_prob_to_score_f32: null pointer");
    VIP_REQUIRE(B > 0 && N > 0, VIP_ERR_BAD_ARG, "vip_prob_to_score_f32: non-positive dimension");
    hipLaunchKernelGGL(prob_to_score_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, prob, score, B, N);
    return vip_launch_status("vip_prob_to_score_f32");
}

extern "C" int vip_ensemble_mean_f32(const float* scores, float* mean, int M, int n, long ld, void* stream) {
    VIP_REQUIRE(scores && mean, VIP_ERR_BAD_ARG, "vip_ensemble_mean_f32: null pointer");
    VIP_REQUIRE(M > 0 && n > 0 && ld >= n, VIP_ERR_BAD_ARG, "vip_ensemble_mean_f32: bad dimension");
    hipLaunchKernelGGL(ensemble_mean_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, scores, mean, M, n, ld);
    return vip_launch_status("vip_ensemble_mean_f32");
}
