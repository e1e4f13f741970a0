// STRICT precision path, part 2 (see strict_conv.hip / conv_h2.hip): every non-GEMM operator of the four model families with
// fp32 arithmetic - pooling, global average pool, squeeze-excite multiply + residual + activation, LayerNorm, depthwise convolution,
// split-attention combine, channel-slice products, ViT token assembly, classifier heads, and the two attention cores (GCViT window
// attention, ViT MHSA) as plain fp32 VALU kernels.  Every kernel is a template over the STORAGE of its activation tensors:
//   SF32 - fp32 (entry points ending in _s32): 16 bytes = 4 floats per lane per access; all channel counts % 4 == 0;
//   SH2  - the packed (hi, lo) fp16 pairs of common.hpp (entry points ending in _h2): 4 channels = two 8-byte halves of a 32-byte
//          group; channel counts, strides and offsets % 8 == 0; stores raise the caller's status word outside the fp16 range.
// Parameters (LayerNorm gamma / beta, depthwise filters, head matrices, the relative-position table) are fp32 in both.
// Same reference call sites as the fp16 kernels they mirror (pointwise.hip, window_attn.hip, mhsa.hip).
#include "common.hpp"
#include <stdlib.h>

namespace {

inline bool attn_h2_mfma_enabled() {
    static const bool on = !(getenv("VIP_ATTN_H2_MFMA") && atoi(getenv("VIP_ATTN_H2_MFMA")) == 0);
    return on;
}

inline unsigned sgrid(long total) {
    long g = (total + 255) / 256;
    if (g > 256L * 32) g = 256L * 32;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }   // fp32 PARAMETERS

// storage policies: ld / st of 4 consecutive logical elements i .. i+3 (i % 4 == 0) of a tensor whose element 0 is at `b`
struct SF32 {
    static constexpr bool MFMA_ATTN = false;
    static __device__ __forceinline__ f32x4 ld(const void* b, long i) { return *reinterpret_cast<const f32x4*>(static_cast<const float*>(b) + i); }
    static __device__ __forceinline__ void st(void* b, long i, f32x4 v, int*) { *reinterpret_cast<f32x4*>(static_cast<float*>(b) + i) = v; }
};
struct SH2 {
    static constexpr bool MFMA_ATTN = true;
    static __device__ __forceinline__ f32x4 ld(const void* b, long i) { return h2_ld4(b, i); }
    static __device__ __forceinline__ void st(void* b, long i, f32x4 v, int* status) { h2_st4(b, i, v, status); }
};

// ---------------------------------------------------------------------------------------------
// pool2d: mode 0 max with ZERO padding taking part (gcvit feature.py:151-152), 1 average over valid taps (Keras "same"
// AveragePooling2D: resnet_rs_model.py:207-212), 2 average over k*k (zero padded: kecam resnest.py:63-65)
// ---------------------------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(256) void spool2d_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int H, int W, int C4,
                                                      int ldx, int ldy, int k, int stride, int pt, int pl, int Ho, int Wo, int mode,
                                                      int* status) {
    const long total = (long)B * Ho * Wo * C4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        long p = idx / C4;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho);
        const int b = (int)(p / Ho);
        f32x4 acc = (mode == 0) ? (f32x4){-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f} : (f32x4){0.f, 0.f, 0.f, 0.f};
        int cnt = 0;
        for (int r = 0; r < k; ++r) {
            const int hi = ho * stride - pt + r;
            for (int s = 0; s < k; ++s) {
                const int wi = wo * stride - pl + s;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
                    v = S::ld(x, ((long)(b * H + hi) * W + wi) * ldx + c4 * 4);
                    ++cnt;
                }
                if (mode == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], v[j]);
                } else {
                    acc += v;
                }
            }
        }
        // a true division (not a multiply by a rounded reciprocal): what AveragePooling2D computes
        if (mode == 1) acc = acc / (float)(cnt > 0 ? cnt : 1);
        if (mode == 2) acc = acc / (float)(k * k);
        S::st(y, ((long)(b * Ho + ho) * Wo + wo) * ldy + c4 * 4, acc, status);
    }
}

// ---------------------------------------------------------------------------------------------
// global average pool: block = (image, 64-channel slab); 16 float4 lanes x 16 pixel lanes
// ---------------------------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(256) void sgap_kernel(const void* __restrict__ x, void* __restrict__ y, int HW, int C, int ldx, int* status) {
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = c0 + cl * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const long xb = (long)b * HW * ldx + c;
        for (int p = pl; p < HW; p += 16) acc += S::ld(x, xb + (long)p * ldx);
    }
    __shared__ float red[16][65];
#pragma unroll
    for (int j = 0; j < 4; ++j) red[pl][cl * 4 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 16) {                              // thread t: channels c0 + 4 t .. + 3, pixel lanes summed in index order
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += red[p][threadIdx.x * 4 + j];
        const int cc = c0 + threadIdx.x * 4;
        if (cc < C) S::st(y, (long)b * C + cc, s / (float)HW, status);
    }
}

// ---------------------------------------------------------------------------------------------
// y = act(x * scale[b,c] + residual); y2 = act2(y)
// ---------------------------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(256) void sscale_add_act_kernel(const void* __restrict__ x, const void* __restrict__ sc,
                                                             const void* __restrict__ res, void* __restrict__ y, void* __restrict__ y2,
                                                             long total4, int HW, int C4, int act, int act2, int* status) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        const long pix = idx / C4;
        const int b = (int)(pix / HW);
        f32x4 v = S::ld(x, idx * 4);
        if (sc) v = v * S::ld(sc, ((long)b * C4 + c4) * 4);
        if (res) v += S::ld(res, idx * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = vip_act_strict(v[j], act);
        S::st(y, idx * 4, v, status);
        if (y2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = vip_act_strict(v[j], act2);
            S::st(y2, idx * 4, v, status);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over C (Keras LayerNormalization: mean, biased variance of the centred values, (x - mean) * rsqrt(var + eps) * gamma + beta):
// one wave per row, the row is kept in registers (C <= 4096)
// ---------------------------------------------------------------------------------------------
template <typename S, int CPL>
__global__ __launch_bounds__(256) void slayernorm_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, void* __restrict__ y, int rows, int C, float eps,
                                                         int* status) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C4 = C >> 2;
    f32x4 v[CPL];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c4 = lane + i * 64;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c4 < C4) v[i] = S::ld(x, (long)row * C + c4 * 4);
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    sum = wave_reduce_sum(sum);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < C4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                sq += d * d;
            }
        }
    }
    sq = wave_reduce_sum(sq);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < C4) {
            const f32x4 g = ld4(gamma + c4 * 4), bb = ld4(beta + c4 * 4);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + bb[j];
            S::st(y, (long)row * C + c4 * 4, o, status);
        }
    }
}

// The packed storage's own LayerNorm: LPR lanes per row (a power of two, 8..64: narrow rows share a wave), a lane owns CPL 32-byte
// groups (8 channels: one 16-byte load of the hi halves, one of the lo halves), the two row reductions are DPP adds inside 16-lane rows
// (pointwise.hip's scheme).  Same arithmetic as above: fp32 mean, biased variance of the centred values, 1 / sqrt, fp32 gamma / beta.
template <int CPL>
__global__ __launch_bounds__(256) void h2_layernorm_kernel(const char* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, char* __restrict__ y, int rows, int C, int lpr,
                                                           float eps, int* status) {
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
        U4H8 h, l;
        h.u = l.u = make_uint4(0, 0, 0, 0);
        if (row_ok && c8 < C8) {
            const char* src = x + ((long)row * C + c8 * 8) * 4;
            h.u = *reinterpret_cast<const uint4*>(src);
            l.u = *reinterpret_cast<const uint4*>(src + 16);
        }
        h2_join8(h, l, v[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[i][j];
    }
    sum = group_allreduce_sum(sum, lpr);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        if (sub + i * lpr < C8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[i][j] - mean;
                sq += d * d;
            }
        }
    }
    sq = group_allreduce_sum(sq, lpr);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
    bool bad = false;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c8 = sub + i * lpr;
        if (row_ok && c8 < C8) {
            float o[8];
            const f32x4 g0 = ld4(gamma + c8 * 8), g1 = ld4(gamma + c8 * 8 + 4), b0 = ld4(beta + c8 * 8), b1 = ld4(beta + c8 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (v[i][j] - mean) * rstd * g0[j] + b0[j];
                o[4 + j] = (v[i][4 + j] - mean) * rstd * g1[j] + b1[j];
            }
            U4H8 oh, ol;
            h2_split8(o, oh, ol);
            char* dst = y + ((long)row * C + c8 * 8) * 4;
            *reinterpret_cast<uint4*>(dst) = oh.u;
            *reinterpret_cast<uint4*>(dst + 16) = ol.u;
            bad |= h2_overflows8(o);
        }
    }
    if (bad && status) *status = VIP_H2_OVERFLOW;
}

// ---------------------------------------------------------------------------------------------
// depthwise conv: thread = TW consecutive output columns x 4 channels
// ---------------------------------------------------------------------------------------------
template <typename S, int K, int ST, int TW>
__global__ __launch_bounds__(256) void sdwconv_kernel(const void* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      void* __restrict__ y, int B, int H, int W, int C4, int pt, int pl, int Ho, int Wo,
                                                      int act, int* status) {
    constexpr int NCOL = (TW - 1) * ST + K;
    const int WoT = (Wo + TW - 1) / TW;
    const long total = (long)B * Ho * WoT * C4;
    const int C = C4 * 4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        long p = idx / C4;
        const int wt = (int)(p % WoT);
        p /= WoT;
        const int ho = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const int wo0 = wt * TW;
        f32x4 acc[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int hi = ho * ST - pt + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            f32x4 col[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                const int wi = wo0 * ST - pl + q;
                col[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if ((unsigned)wi < (unsigned)W) col[q] = S::ld(x, ((long)(b * H + hi) * W + wi) * C + c4 * 4);
            }
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const f32x4 wv = ld4(w + (long)(r * K + s) * C + c4 * 4);
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[t] += col[t * ST + s] * wv;
            }
        }
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = ld4(bias + c4 * 4);
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int wo = wo0 + t;
            if (wo >= Wo) break;
            f32x4 o = acc[t] + bv;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = vip_act_strict(o[j], act);
            S::st(y, ((long)(b * Ho + ho) * Wo + wo) * C + c4 * 4, o, status);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// channel-slice product, split-attention combine, ViT token assembly (a_off / b_off / y_off: first channel of the slices)
// ---------------------------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(256) void smul_kernel(const void* __restrict__ a, const void* __restrict__ b, void* __restrict__ y,
                                                   long total4, int C4, int lda, int a_off, int ldb, int b_off, int ldy, int y_off,
                                                   int* status) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        const long m = idx / C4;
        S::st(y, m * ldy + y_off + c4 * 4, S::ld(a, m * lda + a_off + c4 * 4) * S::ld(b, m * ldb + b_off + c4 * 4), status);
    }
}

template <typename S>
__global__ __launch_bounds__(256) void sradix_combine_kernel(const void* __restrict__ x, const void* __restrict__ s, void* __restrict__ y,
                                                             long total4, int HW, int C4, int radix, int* status) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        const long pix = idx / C4;
        const int b = (int)(pix / HW);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < radix; ++r)
            acc += S::ld(x, (pix * radix * C4 + (long)r * C4 + c4) * 4) * S::ld(s, (((long)b * radix + r) * C4 + c4) * 4);
        S::st(y, idx * 4, acc, status);
    }
}

template <typename S>
__global__ __launch_bounds__(256) void svit_tokens_kernel(const void* __restrict__ patches, const void* __restrict__ cls,
                                                          const void* __restrict__ pos, void* __restrict__ out, int B, int NP, int D4,
                                                          int* status) {
    const long total = (long)B * (NP + 1) * D4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int d4 = (int)(idx % D4);
        const long t = idx / D4;
        const int n = (int)(t % (NP + 1));
        const int b = (int)(t / (NP + 1));
        const f32x4 v = n == 0 ? S::ld(cls, d4 * 4) : S::ld(patches, (((long)b * NP + n - 1) * D4 + d4) * 4);
        S::st(out, idx * 4, v + S::ld(pos, ((long)n * D4 + d4) * 4), status);
    }
}

// ---------------------------------------------------------------------------------------------
// classifier heads: (GAP | token 0) [-> LayerNorm] -> Dense, one workgroup per image; fp32 result [B][N]
// ---------------------------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(256) void sgap_ln_dense_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, const float* __restrict__ Wt,
                                                            const float* __restrict__ bias, float* __restrict__ out, int HW, int C, int ldx,
                                                            long img_stride, int N) {
    __shared__ float pooled[4096];
    __shared__ float red[2][4];
    const int b = blockIdx.x;
    const long xb = (long)b * img_stride;
    float s1 = 0.f;
    for (int c4 = threadIdx.x; c4 < (C >> 2); c4 += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < HW; ++p) acc += S::ld(x, xb + (long)p * ldx + c4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m = acc[j] / (float)HW;
            pooled[c4 * 4 + j] = m;
            s1 += m;
        }
    }
    __syncthreads();
    if (gamma) {
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
        const float rstd = 1.0f / sqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)C + eps);
        for (int c = threadIdx.x; c < C; c += 256) pooled[c] = (pooled[c] - mean) * rstd * gamma[c] + beta[c];
        __syncthreads();
    }
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

// ---------------------------------------------------------------------------------------------
// attention cores: out = softmax(scale * q k^T (+ relative-position bias)) v, one thread per query token, K / V of the
// (window | image, head) item in LDS as fp32 (broadcast reads), online softmax in registers (libm expf), fp32 throughout.
//   WINDOW: GCViT WindowAttention.call (gcvit/layers/attention.py:52-83) on the feature-map layout: qkv [B,Hp,Wp,nq*C] with channels
//           (q|k|v or k|v, head, hd); q of the global-query blocks from q_global [B, ws*ws, C] (attention.py:62-66, also scaled :69);
//           bias = table[(dy + ws - 1)(2 ws - 1) + dx + ws - 1][head], d = query - key coordinate (attention.py:39-50).
//   !WINDOW: tfimm ViTMultiHeadAttention (vit.py:148-167): qkv [B,N,3D], item = (image, head).
// ---------------------------------------------------------------------------------------------
struct SAttnArgs {
    const void* qkv;
    const void* qg;
    const float* table;
    void* out;
    int B, Hp, Wp, C, heads, ws, nq, N;
    float scale;
    int* status;
};

template <typename S, int HD, bool WINDOW>
__global__ __launch_bounds__(256) void sattn_kernel(SAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* kl = reinterpret_cast<float*>(smem);           // [N][HD]
    float* vl = kl + a.N * HD;                            // [N][HD]
    float* tl = vl + a.N * HD;                            // WINDOW: this head's table column, (2ws-1)^2 floats
    const int tid = threadIdx.x, nthr = blockDim.x;
    int item = blockIdx.x;
    const int head = item % a.heads;
    item /= a.heads;
    const int ld = a.nq * a.C;
    int wx = 0, wy = 0, b;
    if (WINDOW) {
        const int nwx = a.Wp / a.ws, nwy = a.Hp / a.ws;
        wx = item % nwx;
        item /= nwx;
        wy = item % nwy;
        b = item / nwy;
    } else {
        b = item;
    }
    auto tok_off = [&](int t) -> long {                   // element offset of token t's channel 0 in qkv
        if (WINDOW) {
            const int ty = t / a.ws, tx = t - ty * a.ws;
            return (((long)b * a.Hp + wy * a.ws + ty) * a.Wp + wx * a.ws + tx) * ld;
        }
        return ((long)b * a.N + t) * ld;
    };
    const int koff = (a.nq - 2) * a.C + head * HD, voff = (a.nq - 1) * a.C + head * HD;
    for (int i = tid; i < a.N * (HD / 4); i += nthr) {
        const int t = i / (HD / 4), d4 = i - t * (HD / 4);
        const long src = tok_off(t);
        *reinterpret_cast<f32x4*>(kl + t * HD + d4 * 4) = S::ld(a.qkv, src + koff + d4 * 4);
        *reinterpret_cast<f32x4*>(vl + t * HD + d4 * 4) = S::ld(a.qkv, src + voff + d4 * 4);
    }
    if (WINDOW) {
        const int nt = (2 * a.ws - 1) * (2 * a.ws - 1);
        for (int i = tid; i < nt; i += nthr) tl[i] = a.table[(long)i * a.heads + head];
    }
    __syncthreads();
    const int t = tid;
    if (t >= a.N) return;
    float q[HD], o[HD];
    {
        const bool gq = WINDOW && a.qg;
        const void* qb = gq ? a.qg : a.qkv;
        const long qo = gq ? ((long)b * a.N + t) * a.C + head * HD : tok_off(t) + head * HD;
#pragma unroll
        for (int d4 = 0; d4 < HD / 4; ++d4) {
            const f32x4 v = S::ld(qb, qo + d4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) q[d4 * 4 + j] = v[j] * a.scale;
        }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    float m = -3.0e38f, l = 0.f;
    const int qy = WINDOW ? t / a.ws : 0, qx = WINDOW ? t - qy * a.ws : 0;
    const int tw = 2 * a.ws - 1;
    int ky = 0, kx = 0;
    for (int j = 0; j < a.N; ++j) {
        const float* kr = kl + j * HD;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < HD / 4; ++d4) {
            const f32x4 kv = ld4(kr + d4 * 4);
            s0 = fmaf(q[d4 * 4], kv[0], s0);
            s1 = fmaf(q[d4 * 4 + 1], kv[1], s1);
            s2 = fmaf(q[d4 * 4 + 2], kv[2], s2);
            s3 = fmaf(q[d4 * 4 + 3], kv[3], s3);
        }
        float s = (s0 + s1) + (s2 + s3);
        if (WINDOW) {
            s += tl[(qy - ky + a.ws - 1) * tw + (qx - kx + a.ws - 1)];
            if (++kx == a.ws) { kx = 0; ++ky; }
        }
        if (s > m) {                                      // new running maximum: rescale what has been accumulated
            const float al = expf(m - s);
            l *= al;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] *= al;
            m = s;
        }
        const float p = expf(s - m);
        l += p;
        const float* vr = vl + j * HD;
#pragma unroll
        for (int d4 = 0; d4 < HD / 4; ++d4) {
            const f32x4 vv = ld4(vr + d4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[d4 * 4 + e] = fmaf(p, vv[e], o[d4 * 4 + e]);
        }
    }
    const float inv = 1.0f / l;
    long dst;
    if (WINDOW) {
        const int ty = t / a.ws, tx = t - ty * a.ws;
        dst = (((long)b * a.Hp + wy * a.ws + ty) * a.Wp + wx * a.ws + tx) * a.C + head * HD;
    } else {
        dst = ((long)b * a.N + t) * a.C + head * HD;
    }
#pragma unroll
    for (int d4 = 0; d4 < HD / 4; ++d4)
        S::st(a.out, dst + d4 * 4, (f32x4){o[d4 * 4] * inv, o[d4 * 4 + 1] * inv, o[d4 * 4 + 2] * inv, o[d4 * 4 + 3] * inv}, a.status);
}

template <typename S, int HD, bool WINDOW>
int launch_sattn(const SAttnArgs& a, long items, hipStream_t s, const char* what) {
    const int smem = (2 * a.N * HD + (WINDOW ? (2 * a.ws - 1) * (2 * a.ws - 1) : 0)) * 4;
    if (smem > 160 * 1024) {
        vip_set_error("%s: N=%d needs %d bytes of LDS", what, a.N, smem);
        return VIP_ERR_UNSUPPORTED;
    }
    static int attr = 0;
    if (smem > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sattn_kernel<S, HD, WINDOW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr = 160 * 1024;
    }
    const int threads = (a.N + 63) / 64 * 64;
    hipLaunchKernelGGL((sattn_kernel<S, HD, WINDOW>), dim3((unsigned)items), dim3(threads), smem, s, a);
    return vip_launch_status(what);
}

// fp32 rows <-> packed rows
__global__ __launch_bounds__(256) void pack_h2_kernel(const float* __restrict__ x, void* __restrict__ y, long n4, int* status) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) SH2::st(y, i * 4, SF32::ld(x, i * 4), status);
}
__global__ __launch_bounds__(256) void unpack_h2_kernel(const void* __restrict__ x, float* __restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) SF32::st(y, i * 4, SH2::ld(x, i * 4), nullptr);
}

// ---- the two families of entry points share their bodies: A = alignment every channel count / stride / offset must have ----
#define S_ALIGNED(A, ...)                                                     \
    do {                                                                     \
        const int vals_[] = {__VA_ARGS__};                                   \
        for (int v_ : vals_)                                                 \
            if (v_ % (A)) {                                                  \
                vip_set_error("%s: channel counts / strides / offsets must be multiples of %d elements", who, (A)); \
                return VIP_ERR_ALIGNMENT;                                    \
            }                                                                \
    } while (0)

template <typename S, int A>
int pool2d_impl(const char* who, const void* x, void* y, int B, int H, int W, int C, int ldx, int ldy, int k, int stride, int pt, int pl,
                int Ho, int Wo, int mode, int* status, void* stream) {
    VIP_REQUIRE(x && y, VIP_ERR_BAD_ARG, "%s: null pointer", who);
    VIP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && Ho > 0 && Wo > 0 && pt >= 0 && pl >= 0, VIP_ERR_BAD_ARG,
                "%s: non-positive dimension", who);
    VIP_REQUIRE(mode >= 0 && mode <= 2 && ldx >= C && ldy >= C, VIP_ERR_BAD_ARG, "%s: bad mode or stride", who);
    S_ALIGNED(A, C, ldx, ldy);
    const long total = (long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(spool2d_kernel<S>, dim3(sgrid(total)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C / 4, ldx, ldy, k, stride, pt,
                       pl, Ho, Wo, mode, status);
    return vip_launch_status(who);
}

template <typename S, int A>
int gap_impl(const char* who, const void* x, void* y, int B, int HW, int C, int ldx, int* status, void* stream) {
    VIP_REQUIRE(x && y, VIP_ERR_BAD_ARG, "%s: null pointer", who);
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && ldx >= C, VIP_ERR_BAD_ARG, "%s: bad dimension", who);
    S_ALIGNED(A, C, ldx);
    hipLaunchKernelGGL(sgap_kernel<S>, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, y, HW, C, ldx, status);
    return vip_launch_status(who);
}

template <typename S, int A>
int scale_add_act_impl(const char* who, const void* x, const void* scale, const void* residual, void* y, void* y2, int B, int HW, int C,
                       int act, int act2, int* status, void* stream) {
    VIP_REQUIRE(x && y, VIP_ERR_BAD_ARG, "%s: null pointer", who);
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && (unsigned)act <= 4u && (unsigned)act2 <= 4u, VIP_ERR_BAD_ARG, "%s: bad argument", who);
    S_ALIGNED(A, C);
    const long total4 = (long)B * HW * (C / 4);
    hipLaunchKernelGGL(sscale_add_act_kernel<S>, dim3(sgrid(total4)), dim3(256), 0, (hipStream_t)stream, x, scale, residual, y, y2, total4, HW,
                       C / 4, act, act2, status);
    return vip_launch_status(who);
}

template <typename S, int A>
int layernorm_impl(const char* who, const void* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps, int* status,
                   void* stream) {
    VIP_REQUIRE(x && y && gamma && beta, VIP_ERR_BAD_ARG, "%s: null pointer", who);
    VIP_REQUIRE(rows > 0 && C > 0, VIP_ERR_BAD_ARG, "%s: non-positive dimension", who);
    S_ALIGNED(A, C);
    if constexpr (S::MFMA_ATTN) {           // (the packed storage) rows up to 2048 channels: the lanes-per-row kernel
        const int C8 = C / 8;
        int lpr = 8;
        while (lpr < 64 && lpr < C8) lpr <<= 1;
        const int cpl8 = (C8 + lpr - 1) / lpr;
        if (cpl8 <= 4) {
            const int rpb = 256 / lpr;
            dim3 grid8((rows + rpb - 1) / rpb);
            hipStream_t s8 = (hipStream_t)stream;
#define H2_LN(N) hipLaunchKernelGGL(h2_layernorm_kernel<N>, grid8, dim3(256), 0, s8, (const char*)x, gamma, beta, (char*)y, rows, C, lpr, eps, status)
            switch (cpl8) {
                case 1: H2_LN(1); break;
                case 2: H2_LN(2); break;
                case 3: H2_LN(3); break;
                default: H2_LN(4); break;
            }
#undef H2_LN
            return vip_launch_status(who);
        }
    }
    const int cpl = (C / 4 + 63) / 64;
    VIP_REQUIRE(cpl <= 16, VIP_ERR_UNSUPPORTED, "%s: C=%d too large (max 4096)", who, C);
    dim3 grid((rows + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
#define S32_LN(N) hipLaunchKernelGGL((slayernorm_kernel<S, N>), grid, dim3(256), 0, s, x, gamma, beta, y, rows, C, eps, status)
    if (cpl <= 1) S32_LN(1);
    else if (cpl <= 2) S32_LN(2);
    else if (cpl <= 4) S32_LN(4);
    else if (cpl <= 8) S32_LN(8);
    else S32_LN(16);
#undef S32_LN
    return vip_launch_status(who);
}

template <typename S, int A>
int dwconv_impl(const char* who, const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int k, int stride,
                int pt, int pl, int Ho, int Wo, int act, int* status, void* stream) {
    VIP_REQUIRE(x && w && y, VIP_ERR_BAD_ARG, "%s: null pointer", who);
    VIP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && pt >= 0 && pl >= 0 && (unsigned)act <= 4u, VIP_ERR_BAD_ARG,
                "%s: bad argument", who);
    S_ALIGNED(A, C);
    const int C4 = C / 4;
    hipStream_t s = (hipStream_t)stream;
#define S32_DW(KK, SS, TW)                                                                                                          \
    {                                                                                                                               \
        const long total = (long)B * Ho * ((Wo + TW - 1) / TW) * C4;                                                                \
        hipLaunchKernelGGL((sdwconv_kernel<S, KK, SS, TW>), dim3(sgrid(total)), dim3(256), 0, s, x, w, bias, y, B, H, W, C4, pt, pl, \
                           Ho, Wo, act, status);                                                                                    \
    }
    if (k == 3 && stride == 1) S32_DW(3, 1, 4)
    else if (k == 3 && stride == 2) S32_DW(3, 2, 2)
    else if (k == 5 && stride == 1) S32_DW(5, 1, 4)
    else if (k == 5 && stride == 2) S32_DW(5, 2, 2)
    else if (k == 7 && stride == 1) S32_DW(7, 1, 4)
    else {
        vip_set_error("%s: unsupported k=%d stride=%d", who, k, stride);
        return VIP_ERR_UNSUPPORTED;
    }
#undef S32_DW
    return vip_launch_status(who);
}

template <typename S, int A>
int mul_impl(const char* who, const void* a, const void* b, void* y, long rows, int C, int lda, int a_off, int ldb, int b_off, int ldy,
             int y_off, int* status, void* stream) {
    VIP_REQUIRE(a && b && y, VIP_ERR_BAD_ARG, "%s: null pointer", who);
    VIP_REQUIRE(rows > 0 && C > 0 && a_off >= 0 && b_off >= 0 && y_off >= 0, VIP_ERR_BAD_ARG, "%s: bad size", who);
    VIP_REQUIRE(a_off + C <= lda && b_off + C <= ldb && y_off + C <= ldy, VIP_ERR_BAD_ARG, "%s: slice exceeds its row", who);
    S_ALIGNED(A, C, lda, ldb, ldy, a_off, b_off, y_off);
    const long total4 = rows * (C / 4);
    hipLaunchKernelGGL(smul_kernel<S>, dim3(sgrid(total4)), dim3(256), 0, (hipStream_t)stream, a, b, y, total4, C / 4, lda, a_off, ldb, b_off,
                       ldy, y_off, status);
    return vip_launch_status(who);
}

template <typename S, int A>
int radix_combine_impl(const char* who, const void* x, const void* scale, void* y, int B, int HW, int C, int radix, int* status, void* stream) {
    VIP_REQUIRE(x && scale && y, VIP_ERR_BAD_ARG, "%s: null pointer", who);
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && radix > 0, VIP_ERR_BAD_ARG, "%s: non-positive dimension", who);
    S_ALIGNED(A, C);
    const long total4 = (long)B * HW * (C / 4);
    hipLaunchKernelGGL(sradix_combine_kernel<S>, dim3(sgrid(total4)), dim3(256), 0, (hipStream_t)stream, x, scale, y, total4, HW, C / 4, radix,
                       status);
    return vip_launch_status(who);
}

template <typename S, int A>
int vit_tokens_impl(const char* who, const void* patches, const void* cls, const void* pos, void* out, int B, int NP, int D, int* status,
                    void* stream) {
    VIP_REQUIRE(patches && cls && pos && out, VIP_ERR_BAD_ARG, "%s: null pointer", who);
    VIP_REQUIRE(B > 0 && NP > 0 && D > 0, VIP_ERR_BAD_ARG, "%s: non-positive dimension", who);
    S_ALIGNED(A, D);
    const long total = (long)B * (NP + 1) * (D / 4);
    hipLaunchKernelGGL(svit_tokens_kernel<S>, dim3(sgrid(total)), dim3(256), 0, (hipStream_t)stream, patches, cls, pos, out, B, NP, D / 4,
                       status);
    return vip_launch_status(who);
}

template <typename S, int A>
int gap_ln_dense_impl(const char* who, const void* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias,
                      float* out, int B, int HW, int C, int ldx, long img_stride, int N, void* stream) {
    VIP_REQUIRE(x && W && out && (!gamma == !beta), VIP_ERR_BAD_ARG, "%s: null pointer", who);
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && N > 0 && eps >= 0.f && ldx >= C, VIP_ERR_BAD_ARG, "%s: bad dimension or eps", who);
    VIP_REQUIRE(C <= 4096, VIP_ERR_UNSUPPORTED, "%s: C=%d > 4096", who, C);
    VIP_REQUIRE(img_stride % A == 0, VIP_ERR_ALIGNMENT, "%s: image stride must be a multiple of %d elements", who, A);
    S_ALIGNED(A, C, ldx);
    hipLaunchKernelGGL(sgap_ln_dense_kernel<S>, dim3(B), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, W, bias, out, HW, C, ldx,
                       img_stride, N);
    return vip_launch_status(who);
}

template <typename S>
int window_attn_impl(const char* who, const void* qkv, const void* q_global, const float* bias_table, void* out, int B, int Hp, int Wp, int C,
                     int heads, int ws, int nq, float scale, int* status, void* stream) {
    VIP_REQUIRE(qkv && bias_table && out, VIP_ERR_BAD_ARG, "%s: null pointer", who);
    VIP_REQUIRE(B > 0 && Hp > 0 && Wp > 0 && C > 0 && heads > 0 && ws > 0, VIP_ERR_BAD_ARG, "%s: non-positive dimension", who);
    VIP_REQUIRE((nq == 3 && !q_global) || (nq == 2 && q_global), VIP_ERR_BAD_ARG, "%s: nq=3 without q_global or nq=2 with it", who);
    VIP_REQUIRE(Hp % ws == 0 && Wp % ws == 0, VIP_ERR_BAD_ARG, "%s: feature map is not a multiple of the window", who);
    VIP_REQUIRE(C == heads * 32, VIP_ERR_UNSUPPORTED, "%s: head_dim must be 32 (C=%d heads=%d)", who, C, heads);
    VIP_REQUIRE(ws * ws <= 256, VIP_ERR_UNSUPPORTED, "%s: window of %d tokens (max 256)", who, ws * ws);
    SAttnArgs a{qkv, q_global, bias_table, out, B, Hp, Wp, C, heads, ws, nq, ws * ws, scale, status};
    const long items = (long)B * (Hp / ws) * (Wp / ws) * heads;
    VIP_REQUIRE(items < (1L << 31), VIP_ERR_UNSUPPORTED, "%s: too many work items", who);
    if constexpr (S::MFMA_ATTN) {       // packed storage: the matrix-core kernel of attn_h2.hip (VIP_ATTN_H2_MFMA=0: the VALU kernel below)
        if (attn_h2_mfma_enabled()) {
            const int st = vip_window_attn_h2_mfma(qkv, q_global, bias_table, out, B, Hp, Wp, C, heads, ws, nq, scale, status, (hipStream_t)stream);
            if (st != 1) return st;
        }
    }
    return launch_sattn<S, 32, true>(a, items, (hipStream_t)stream, who);
}

template <typename S>
int mhsa_impl(const char* who, const void* qkv, void* out, int B, int N, int D, int heads, float scale, int* status, void* stream) {
    VIP_REQUIRE(qkv && out, VIP_ERR_BAD_ARG, "%s: null pointer", who);
    VIP_REQUIRE(B > 0 && N > 0 && D > 0 && heads > 0, VIP_ERR_BAD_ARG, "%s: non-positive dimension", who);
    VIP_REQUIRE(D == heads * 64, VIP_ERR_UNSUPPORTED, "%s: head_dim must be 64 (D=%d heads=%d)", who, D, heads);
    VIP_REQUIRE(N <= 256, VIP_ERR_UNSUPPORTED, "%s: N=%d tokens (max 256)", who, N);
    if constexpr (S::MFMA_ATTN) {
        if (attn_h2_mfma_enabled()) {
            const int st = vip_mhsa_h2_mfma(qkv, out, B, N, D, heads, scale, status, (hipStream_t)stream);
            if (st != 1) return st;
        }
    }
    SAttnArgs a{qkv, nullptr, nullptr, out, B, 0, 0, D, heads, 0, 3, N, scale, status};
    return launch_sattn<S, 64, false>(a, (long)B * heads, (hipStream_t)stream, who);
}

}  // namespace

// ---- fp32 storage (_s32) ------------------------------------------------------------------------------------------------------
extern "C" int vip_pool2d_nhwc_s32(const float* x, float* y, int B, int H, int W, int C, int ldx, int ldy, int k, int stride, int pt,
                                   int pl, int Ho, int Wo, int mode, void* stream) {
    return pool2d_impl<SF32, 4>("vip_pool2d_nhwc_s32", x, y, B, H, W, C, ldx, ldy, k, stride, pt, pl, Ho, Wo, mode, nullptr, stream);
}
extern "C" int vip_global_avgpool_s32(const float* x, float* y, int B, int HW, int C, int ldx, void* stream) {
    return gap_impl<SF32, 4>("vip_global_avgpool_s32", x, y, B, HW, C, ldx, nullptr, stream);
}
extern "C" int vip_scale_add_act_s32(const float* x, const float* scale, const float* residual, float* y, float* y2, int B, int HW, int C,
                                     int act, int act2, void* stream) {
    return scale_add_act_impl<SF32, 4>("vip_scale_add_act_s32", x, scale, residual, y, y2, B, HW, C, act, act2, nullptr, stream);
}
extern "C" int vip_layernorm_s32(const float* x, const float* gamma, const float* beta, float* y, int rows, int C, float eps, void* stream) {
    return layernorm_impl<SF32, 4>("vip_layernorm_s32", x, gamma, beta, y, rows, C, eps, nullptr, stream);
}
extern "C" int vip_dwconv2d_nhwc_s32(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, int k,
                                     int stride, int pt, int pl, int Ho, int Wo, int act, void* stream) {
    return dwconv_impl<SF32, 4>("vip_dwconv2d_nhwc_s32", x, w, bias, y, B, H, W, C, k, stride, pt, pl, Ho, Wo, act, nullptr, stream);
}
extern "C" int vip_mul_s32(const float* a, const float* b, float* y, long rows, int C, int lda, int a_off, int ldb, int b_off, int ldy,
                           int y_off, void* stream) {
    return mul_impl<SF32, 4>("vip_mul_s32", a, b, y, rows, C, lda, a_off, ldb, b_off, ldy, y_off, nullptr, stream);
}
extern "C" int vip_radix_combine_s32(const float* x, const float* scale, float* y, int B, int HW, int C, int radix, void* stream) {
    return radix_combine_impl<SF32, 4>("vip_radix_combine_s32", x, scale, y, B, HW, C, radix, nullptr, stream);
}
extern "C" int vip_vit_tokens_s32(const float* patches, const float* cls, const float* pos, float* out, int B, int NP, int D, void* stream) {
    return vit_tokens_impl<SF32, 4>("vip_vit_tokens_s32", patches, cls, pos, out, B, NP, D, nullptr, stream);
}
/* (mean over HW rows of x[b]) [-> LayerNorm(gamma, beta, eps) when gamma != NULL] -> Dense(W [N][C], bias) -> out [B][N].
 * x[b] starts img_stride elements after x[b-1]; rows are ldx elements apart (token 0 of [B,N,D]: HW = 1, img_stride = N*D). */
extern "C" int vip_gap_ln_dense_s32(const float* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias,
                                    float* out, int B, int HW, int C, int ldx, long img_stride, int N, void* stream) {
    return gap_ln_dense_impl<SF32, 4>("vip_gap_ln_dense_s32", x, gamma, beta, eps, W, bias, out, B, HW, C, ldx, img_stride, N, stream);
}
extern "C" int vip_window_attn_fwd_s32(const float* qkv, const float* q_global, const float* bias_table, float* out, int B, int Hp, int Wp,
                                       int C, int heads, int ws, int nq, float scale, void* stream) {
    return window_attn_impl<SF32>("vip_window_attn_fwd_s32", qkv, q_global, bias_table, out, B, Hp, Wp, C, heads, ws, nq, scale, nullptr, stream);
}
extern "C" int vip_mhsa_fwd_s32(const float* qkv, float* out, int B, int N, int D, int heads, float scale, void* stream) {
    return mhsa_impl<SF32>("vip_mhsa_fwd_s32", qkv, out, B, N, D, heads, scale, nullptr, stream);
}

// ---- packed (hi, lo) fp16 storage (_h2) ---------------------------------------------------------------------------------------
extern "C" int vip_pack_h2(const float* x, void* y, long n, int* status, void* stream) {
    VIP_REQUIRE(x && y && n > 0 && n % 8 == 0, VIP_ERR_BAD_ARG, "vip_pack_h2: null pointer or n not a positive multiple of 8");
    hipLaunchKernelGGL(pack_h2_kernel, dim3(sgrid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, y, n / 4, status);
    return vip_launch_status("vip_pack_h2");
}
extern "C" int vip_unpack_h2(const void* x, float* y, long n, void* stream) {
    VIP_REQUIRE(x && y && n > 0 && n % 8 == 0, VIP_ERR_BAD_ARG, "vip_unpack_h2: null pointer or n not a positive multiple of 8");
    hipLaunchKernelGGL(unpack_h2_kernel, dim3(sgrid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, y, n / 4);
    return vip_launch_status("vip_unpack_h2");
}
extern "C" int vip_pool2d_nhwc_h2(const void* x, void* y, int B, int H, int W, int C, int ldx, int ldy, int k, int stride, int pt, int pl,
                                  int Ho, int Wo, int mode, int* status, void* stream) {
    return pool2d_impl<SH2, 8>("vip_pool2d_nhwc_h2", x, y, B, H, W, C, ldx, ldy, k, stride, pt, pl, Ho, Wo, mode, status, stream);
}
extern "C" int vip_global_avgpool_h2(const void* x, void* y, int B, int HW, int C, int ldx, int* status, void* stream) {
    return gap_impl<SH2, 8>("vip_global_avgpool_h2", x, y, B, HW, C, ldx, status, stream);
}
extern "C" int vip_scale_add_act_h2(const void* x, const void* scale, const void* residual, void* y, void* y2, int B, int HW, int C, int act,
                                    int act2, int* status, void* stream) {
    return scale_add_act_impl<SH2, 8>("vip_scale_add_act_h2", x, scale, residual, y, y2, B, HW, C, act, act2, status, stream);
}
extern "C" int vip_layernorm_h2(const void* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps, int* status,
                                void* stream) {
    return layernorm_impl<SH2, 8>("vip_layernorm_h2", x, gamma, beta, y, rows, C, eps, status, stream);
}
extern "C" int vip_dwconv2d_nhwc_h2(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int k, int stride,
                                    int pt, int pl, int Ho, int Wo, int act, int* status, void* stream) {
    // stride 1: the register-tiled kernel of dwconv.hip (the fp16 path's scheme; VIP_DW_H2_TILE=0: the plain kernel below).  The LDS-staged
    // kernel takes a quad-major filter and has its own entry point, vip_dwconv2d_s1_h2 (dwconv_lds_h2.hip)
    static const bool tiled = !(getenv("VIP_DW_H2_TILE") && atoi(getenv("VIP_DW_H2_TILE")) == 0);
    if (tiled && stride == 1 && x && w && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && Ho > 0 && Wo > 0 && pt >= 0 && pl >= 0 &&
        (unsigned)act <= 4u && (long)(Ho - 1) - pt < H && (long)(Wo - 1) - pl < W) {
        const int st = vip_dwconv_tiled_h2(x, w, bias, y, B, H, W, C, k, pt, pl, Ho, Wo, act, status, (hipStream_t)stream);
        if (st != 1) return st;
    }
    return dwconv_impl<SH2, 8>("vip_dwconv2d_nhwc_h2", x, w, bias, y, B, H, W, C, k, stride, pt, pl, Ho, Wo, act, status, stream);
}
extern "C" int vip_mul_h2(const void* a, const void* b, void* y, long rows, int C, int lda, int a_off, int ldb, int b_off, int ldy, int y_off,
                          int* status, void* stream) {
    return mul_impl<SH2, 8>("vip_mul_h2", a, b, y, rows, C, lda, a_off, ldb, b_off, ldy, y_off, status, stream);
}
extern "C" int vip_radix_combine_h2(const void* x, const void* scale, void* y, int B, int HW, int C, int radix, int* status, void* stream) {
    return radix_combine_impl<SH2, 8>("vip_radix_combine_h2", x, scale, y, B, HW, C, radix, status, stream);
}
extern "C" int vip_vit_tokens_h2(const void* patches, const void* cls, const void* pos, void* out, int B, int NP, int D, int* status,
                                 void* stream) {
    return vit_tokens_impl<SH2, 8>("vip_vit_tokens_h2", patches, cls, pos, out, B, NP, D, status, stream);
}
extern "C" int vip_gap_ln_dense_h2(const void* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias,
                                   float* out, int B, int HW, int C, int ldx, long img_stride, int N, void* stream) {
    return gap_ln_dense_impl<SH2, 8>("vip_gap_ln_dense_h2", x, gamma, beta, eps, W, bias, out, B, HW, C, ldx, img_stride, N, stream);
}
extern "C" int vip_window_attn_fwd_h2(const void* qkv, const void* q_global, const float* bias_table, void* out, int B, int Hp, int Wp, int C,
                                      int heads, int ws, int nq, float scale, int* status, void* stream) {
    return window_attn_impl<SH2>("vip_window_attn_fwd_h2", qkv, q_global, bias_table, out, B, Hp, Wp, C, heads, ws, nq, scale, status, stream);
}
extern "C" int vip_mhsa_fwd_h2(const void* qkv, void* out, int B, int N, int D, int heads, float scale, int* status, void* stream) {
    return mhsa_impl<SH2>("vip_mhsa_fwd_h2", qkv, out, B, N, D, heads, scale, status, stream);
}
