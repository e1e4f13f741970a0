// STRICT precision path, part 2 (see strict_conv.hip): every non-GEMM operator of the four model families with fp32 storage and
// fp32 arithmetic - pooling, global average pool, squeeze-excite multiply + residual + activation, LayerNorm, depthwise convolution,
// split-attention combine, channel-slice products, ViT token assembly, classifier heads, and the two attention cores (GCViT window
// attention, ViT MHSA) as plain fp32 VALU kernels.  16 bytes = 4 floats per lane per access; all channel counts % 4 == 0.
// Same reference call sites as the fp16 kernels they mirror (pointwise.hip, window_attn.hip, mhsa.hip); the entry points end in _s32.
#include "common.hpp"

namespace {

inline unsigned sgrid(long total) {
    long g = (total + 255) / 256;
    if (g > 256L * 32) g = 256L * 32;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---------------------------------------------------------------------------------------------
// pool2d: mode 0 max with ZERO padding taking part (gcvit feature.py:151-152), 1 average over valid taps (Keras "same"
// AveragePooling2D: resnet_rs_model.py:207-212), 2 average over k*k (zero padded: kecam resnest.py:63-65)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spool2d_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C4,
                                                      int ldx, int ldy, int k, int stride, int pt, int pl, int Ho, int Wo, int mode) {
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
                    v = ld4(x + ((long)(b * H + hi) * W + wi) * ldx + c4 * 4);
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
        st4(y + ((long)(b * Ho + ho) * Wo + wo) * ldy + c4 * 4, acc);
    }
}

// ---------------------------------------------------------------------------------------------
// global average pool: block = (image, 64-channel slab); 16 float4 lanes x 16 pixel lanes
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgap_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C, int ldx) {
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = c0 + cl * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const float* xb = x + (long)b * HW * ldx + c;
        for (int p = pl; p < HW; p += 16) acc += ld4(xb + (long)p * ldx);
    }
    __shared__ float red[16][65];
#pragma unroll
    for (int j = 0; j < 4; ++j) red[pl][cl * 4 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) s += red[p][threadIdx.x];
        const int cc = c0 + threadIdx.x;
        if (cc < C) y[(long)b * C + cc] = s / (float)HW;
    }
}

// ---------------------------------------------------------------------------------------------
// y = act(x * scale[b,c] + residual); y2 = act2(y)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sscale_add_act_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                             const float* __restrict__ res, float* __restrict__ y, float* __restrict__ y2,
                                                             long total4, int HW, int C4, int act, int act2) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        const long pix = idx / C4;
        const int b = (int)(pix / HW);
        f32x4 v = ld4(x + idx * 4);
        if (sc) v = v * ld4(sc + ((long)b * C4 + c4) * 4);
        if (res) v += ld4(res + idx * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = vip_act_strict(v[j], act);
        st4(y + idx * 4, v);
        if (y2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = vip_act_strict(v[j], act2);
            st4(y2 + idx * 4, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over C (Keras LayerNormalization: mean, biased variance of the centred values, (x - mean) * rsqrt(var + eps) * gamma + beta):
// one wave per row, the row is kept in registers (C <= 4096)
// ---------------------------------------------------------------------------------------------
template <int CPL>
__global__ __launch_bounds__(256) void slayernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y, int rows, int C, float eps) {
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
        if (c4 < C4) v[i] = ld4(x + (long)row * C + c4 * 4);
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
            st4(y + (long)row * C + c4 * 4, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// depthwise conv: thread = TW consecutive output columns x 4 channels
// ---------------------------------------------------------------------------------------------
template <int K, int S, int TW>
__global__ __launch_bounds__(256) void sdwconv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ y, int B, int H, int W, int C4, int pt, int pl, int Ho, int Wo,
                                                      int act) {
    constexpr int NCOL = (TW - 1) * S + K;
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
            const int hi = ho * S - pt + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            f32x4 col[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                const int wi = wo0 * S - pl + q;
                col[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if ((unsigned)wi < (unsigned)W) col[q] = ld4(x + ((long)(b * H + hi) * W + wi) * C + c4 * 4);
            }
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const f32x4 wv = ld4(w + (long)(r * K + s) * C + c4 * 4);
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[t] += col[t * S + s] * wv;
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
            st4(y + ((long)(b * Ho + ho) * Wo + wo) * C + c4 * 4, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// channel-slice product, split-attention combine, ViT token assembly
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void smul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                                   long total4, int C4, int lda, int ldb, int ldy) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        const long m = idx / C4;
        st4(y + m * ldy + c4 * 4, ld4(a + m * lda + c4 * 4) * ld4(b + m * ldb + c4 * 4));
    }
}

__global__ __launch_bounds__(256) void sradix_combine_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y,
                                                             long total4, int HW, int C4, int radix) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        const long pix = idx / C4;
        const int b = (int)(pix / HW);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < radix; ++r)
            acc += ld4(x + (pix * radix * C4 + (long)r * C4 + c4) * 4) * ld4(s + (((long)b * radix + r) * C4 + c4) * 4);
        st4(y + idx * 4, acc);
    }
}

__global__ __launch_bounds__(256) void svit_tokens_kernel(const float* __restrict__ patches, const float* __restrict__ cls,
                                                          const float* __restrict__ pos, float* __restrict__ out, int B, int NP, int D4) {
    const long total = (long)B * (NP + 1) * D4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int d4 = (int)(idx % D4);
        const long t = idx / D4;
        const int n = (int)(t % (NP + 1));
        const int b = (int)(t / (NP + 1));
        const f32x4 v = n == 0 ? ld4(cls + d4 * 4) : ld4(patches + (((long)b * NP + n - 1) * D4 + d4) * 4);
        st4(out + idx * 4, v + ld4(pos + ((long)n * D4 + d4) * 4));
    }
}

// ---------------------------------------------------------------------------------------------
// classifier heads: (GAP | token 0) [-> LayerNorm] -> Dense, one workgroup per image
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgap_ln_dense_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, const float* __restrict__ Wt,
                                                            const float* __restrict__ bias, float* __restrict__ out, int HW, int C, int ldx,
                                                            long img_stride, int N) {
    __shared__ float pooled[4096];
    __shared__ float red[2][4];
    const int b = blockIdx.x;
    const float* xb = x + (long)b * img_stride;
    float s1 = 0.f;
    for (int c4 = threadIdx.x; c4 < (C >> 2); c4 += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < HW; ++p) acc += ld4(xb + (long)p * ldx + c4 * 4);
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
// (window | image, head) item in LDS (broadcast reads), online softmax in registers (libm expf), fp32 throughout.
//   WINDOW: GCViT WindowAttention.call (gcvit/layers/attention.py:52-83) on the feature-map layout: qkv [B,Hp,Wp,nq*C] with channels
//           (q|k|v or k|v, head, hd); q of the global-query blocks from q_global [B, ws*ws, C] (attention.py:62-66, also scaled :69);
//           bias = table[(dy + ws - 1)(2 ws - 1) + dx + ws - 1][head], d = query - key coordinate (attention.py:39-50).
//   !WINDOW: tfimm ViTMultiHeadAttention (vit.py:148-167): qkv [B,N,3D], item = (image, head).
// ---------------------------------------------------------------------------------------------
struct SAttnArgs {
    const float* qkv;
    const float* qg;
    const float* table;
    float* out;
    int B, Hp, Wp, C, heads, ws, nq, N;
    float scale;
};

template <int HD, bool WINDOW>
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
    auto tok_off = [&](int t) -> long {                   // float offset of token t's channel 0 in qkv
        if (WINDOW) {
            const int ty = t / a.ws, tx = t - ty * a.ws;
            return (((long)b * a.Hp + wy * a.ws + ty) * a.Wp + wx * a.ws + tx) * ld;
        }
        return ((long)b * a.N + t) * ld;
    };
    const int koff = (a.nq - 2) * a.C + head * HD, voff = (a.nq - 1) * a.C + head * HD;
    for (int i = tid; i < a.N * (HD / 4); i += nthr) {
        const int t = i / (HD / 4), d4 = i - t * (HD / 4);
        const float* src = a.qkv + tok_off(t);
        st4(kl + t * HD + d4 * 4, ld4(src + koff + d4 * 4));
        st4(vl + t * HD + d4 * 4, ld4(src + voff + d4 * 4));
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
        const float* qsrc = (WINDOW && a.qg) ? a.qg + ((long)b * a.N + t) * a.C + head * HD : a.qkv + tok_off(t) + head * HD;
#pragma unroll
        for (int d4 = 0; d4 < HD / 4; ++d4) {
            const f32x4 v = ld4(qsrc + d4 * 4);
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
    float* dst;
    if (WINDOW) {
        const int ty = t / a.ws, tx = t - ty * a.ws;
        dst = a.out + (((long)b * a.Hp + wy * a.ws + ty) * a.Wp + wx * a.ws + tx) * a.C + head * HD;
    } else {
        dst = a.out + ((long)b * a.N + t) * a.C + head * HD;
    }
#pragma unroll
    for (int d4 = 0; d4 < HD / 4; ++d4)
        st4(dst + d4 * 4, (f32x4){o[d4 * 4] * inv, o[d4 * 4 + 1] * inv, o[d4 * 4 + 2] * inv, o[d4 * 4 + 3] * inv});
}

template <int HD, bool WINDOW>
int launch_sattn(const SAttnArgs& a, long items, hipStream_t s, const char* what) {
    const int smem = (2 * a.N * HD + (WINDOW ? (2 * a.ws - 1) * (2 * a.ws - 1) : 0)) * 4;
    if (smem > 160 * 1024) {
        vip_set_error("%s: N=%d needs %d bytes of LDS", what, a.N, smem);
        return VIP_ERR_UNSUPPORTED;
    }
    static int attr = 0;
    if (smem > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sattn_kernel<HD, WINDOW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr = 160 * 1024;
    }
    const int threads = (a.N + 63) / 64 * 64;
    hipLaunchKernelGGL((sattn_kernel<HD, WINDOW>), dim3((unsigned)items), dim3(threads), smem, s, a);
    return vip_launch_status(what);
}

}  // namespace

#define S32_ALIGNED4(...)                                                    \
    do {                                                                     \
        const int vals_[] = {__VA_ARGS__};                                   \
        for (int v_ : vals_)                                                 \
            if (v_ % 4) {                                                    \
                vip_set_error("%s: channel counts / strides must be multiples of 4 floats", __func__); \
                return VIP_ERR_ALIGNMENT;                                    \
            }                                                                \
    } while (0)

extern "C" int vip_pool2d_nhwc_s32(const float* x, float* y, int B, int H, int W, int C, int ldx, int ldy, int k, int stride, int pt,
                                   int pl, int Ho, int Wo, int mode, void* stream) {
    VIP_REQUIRE(x && y, VIP_ERR_BAD_ARG, "vip_pool2d_nhwc_s32: null pointer");
    VIP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && Ho > 0 && Wo > 0 && pt >= 0 && pl >= 0, VIP_ERR_BAD_ARG,
                "vip_pool2d_nhwc_s32: non-positive dimension");
    VIP_REQUIRE(mode >= 0 && mode <= 2 && ldx >= C && ldy >= C, VIP_ERR_BAD_ARG, "vip_pool2d_nhwc_s32: bad mode or stride");
    S32_ALIGNED4(C, ldx, ldy);
    const long total = (long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(spool2d_kernel, dim3(sgrid(total)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C / 4, ldx, ldy, k, stride, pt,
                       pl, Ho, Wo, mode);
    return vip_launch_status("vip_pool2d_nhwc_s32");
}

extern "C" int vip_global_avgpool_s32(const float* x, float* y, int B, int HW, int C, int ldx, void* stream) {
    VIP_REQUIRE(x && y, VIP_ERR_BAD_ARG, "vip_global_avgpool_s32: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && ldx >= C, VIP_ERR_BAD_ARG, "vip_global_avgpool_s32: bad dimension");
    S32_ALIGNED4(C, ldx);
    hipLaunchKernelGGL(sgap_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, y, HW, C, ldx);
    return vip_launch_status("vip_global_avgpool_s32");
}

extern "C" int vip_scale_add_act_s32(const float* x, const float* scale, const float* residual, float* y, float* y2, int B, int HW, int C,
                                     int act, int act2, void* stream) {
    VIP_REQUIRE(x && y, VIP_ERR_BAD_ARG, "vip_scale_add_act_s32: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && (unsigned)act <= 4u && (unsigned)act2 <= 4u, VIP_ERR_BAD_ARG, "vip_scale_add_act_s32: bad argument");
    S32_ALIGNED4(C);
    const long total4 = (long)B * HW * (C / 4);
    hipLaunchKernelGGL(sscale_add_act_kernel, dim3(sgrid(total4)), dim3(256), 0, (hipStream_t)stream, x, scale, residual, y, y2, total4, HW,
                       C / 4, act, act2);
    return vip_launch_status("vip_scale_add_act_s32");
}

extern "C" int vip_layernorm_s32(const float* x, const float* gamma, const float* beta, float* y, int rows, int C, float eps, void* stream) {
    VIP_REQUIRE(x && y && gamma && beta, VIP_ERR_BAD_ARG, "vip_layernorm_s32: null pointer");
    VIP_REQUIRE(rows > 0 && C > 0, VIP_ERR_BAD_ARG, "vip_layernorm_s32: non-positive dimension");
    S32_ALIGNED4(C);
    const int cpl = (C / 4 + 63) / 64;
    VIP_REQUIRE(cpl <= 16, VIP_ERR_UNSUPPORTED, "vip_layernorm_s32: C=%d too large (max 4096)", C);
    dim3 grid((rows + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
#define S32_LN(N) hipLaunchKernelGGL(slayernorm_kernel<N>, grid, dim3(256), 0, s, x, gamma, beta, y, rows, C, eps)
    if (cpl <= 1) S32_LN(1);
    else if (cpl <= 2) S32_LN(2);
    else if (cpl <= 4) S32_LN(4);
    else if (cpl <= 8) S32_LN(8);
    else S32_LN(16);
#undef S32_LN
    return vip_launch_status("vip_layernorm_s32");
}

extern "C" int vip_dwconv2d_nhwc_s32(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, int k,
                                     int stride, int pt, int pl, int Ho, int Wo, int act, void* stream) {
    VIP_REQUIRE(x && w && y, VIP_ERR_BAD_ARG, "vip_dwconv2d_nhwc_s32: null pointer");
    VIP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && pt >= 0 && pl >= 0 && (unsigned)act <= 4u, VIP_ERR_BAD_ARG,
                "vip_dwconv2d_nhwc_s32: bad argument");
    S32_ALIGNED4(C);
    const int C4 = C / 4;
    hipStream_t s = (hipStream_t)stream;
#define S32_DW(KK, SS, TW)                                                                                                       \
    {                                                                                                                            \
        const long total = (long)B * Ho * ((Wo + TW - 1) / TW) * C4;                                                             \
        hipLaunchKernelGGL((sdwconv_kernel<KK, SS, TW>), dim3(sgrid(total)), dim3(256), 0, s, x, w, bias, y, B, H, W, C4, pt, pl, \
                           Ho, Wo, act);                                                                                         \
    }
    if (k == 3 && stride == 1) S32_DW(3, 1, 4)
    else if (k == 3 && stride == 2) S32_DW(3, 2, 2)
    else if (k == 5 && stride == 1) S32_DW(5, 1, 4)
    else if (k == 5 && stride == 2) S32_DW(5, 2, 2)
    else if (k == 7 && stride == 1) S32_DW(7, 1, 4)
    else {
        vip_set_error("vip_dwconv2d_nhwc_s32: unsupported k=%d stride=%d", k, stride);
        return VIP_ERR_UNSUPPORTED;
    }
#undef S32_DW
    return vip_launch_status("vip_dwconv2d_nhwc_s32");
}

extern "C" int vip_mul_s32(const float* a, const float* b, float* y, long rows, int C, int lda, int a_off, int ldb, int b_off, int ldy,
                           int y_off, void* stream) {
    VIP_REQUIRE(a && b && y, VIP_ERR_BAD_ARG, "vip_mul_s32: null pointer");
    VIP_REQUIRE(rows > 0 && C > 0 && a_off >= 0 && b_off >= 0 && y_off >= 0, VIP_ERR_BAD_ARG, "vip_mul_s32: bad size");
    VIP_REQUIRE(a_off + C <= lda && b_off + C <= ldb && y_off + C <= ldy, VIP_ERR_BAD_ARG, "vip_mul_s32: slice exceeds its row");
    S32_ALIGNED4(C, lda, ldb, ldy, a_off, b_off, y_off);
    const long total4 = rows * (C / 4);
    hipLaunchKernelGGL(smul_kernel, dim3(sgrid(total4)), dim3(256), 0, (hipStream_t)stream, a + a_off, b + b_off, y + y_off, total4, C / 4,
                       lda, ldb, ldy);
    return vip_launch_status("vip_mul_s32");
}

extern "C" int vip_radix_combine_s32(const float* x, const float* scale, float* y, int B, int HW, int C, int radix, void* stream) {
    VIP_REQUIRE(x && scale && y, VIP_ERR_BAD_ARG, "vip_radix_combine_s32: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && radix > 0, VIP_ERR_BAD_ARG, "vip_radix_combine_s32: non-positive dimension");
    S32_ALIGNED4(C);
    const long total4 = (long)B * HW * (C / 4);
    hipLaunchKernelGGL(sradix_combine_kernel, dim3(sgrid(total4)), dim3(256), 0, (hipStream_t)stream, x, scale, y, total4, HW, C / 4, radix);
    return vip_launch_status("vip_radix_combine_s32");
}

extern "C" int vip_vit_tokens_s32(const float* patches, const float* cls, const float* pos, float* out, int B, int NP, int D, void* stream) {
    VIP_REQUIRE(patches && cls && pos && out, VIP_ERR_BAD_ARG, "vip_vit_tokens_s32: null pointer");
    VIP_REQUIRE(B > 0 && NP > 0 && D > 0, VIP_ERR_BAD_ARG, "vip_vit_tokens_s32: non-positive dimension");
    S32_ALIGNED4(D);
    const long total = (long)B * (NP + 1) * (D / 4);
    hipLaunchKernelGGL(svit_tokens_kernel, dim3(sgrid(total)), dim3(256), 0, (hipStream_t)stream, patches, cls, pos, out, B, NP, D / 4);
    return vip_launch_status("vip_vit_tokens_s32");
}

/* (mean over HW rows of x[b]) [-> LayerNorm(gamma, beta, eps) when gamma != NULL] -> Dense(W [N][C], bias) -> out [B][N].
 * x[b] starts img_stride floats after x[b-1]; rows are ldx floats apart (token 0 of [B,N,D]: HW = 1, img_stride = N*D). */
extern "C" int vip_gap_ln_dense_s32(const float* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias,
                                    float* out, int B, int HW, int C, int ldx, long img_stride, int N, void* stream) {
    VIP_REQUIRE(x && W && out && (!gamma == !beta), VIP_ERR_BAD_ARG, "vip_gap_ln_dense_s32: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && N > 0 && eps >= 0.f && ldx >= C, VIP_ERR_BAD_ARG, "vip_gap_ln_dense_s32: bad dimension or eps");
    VIP_REQUIRE(C <= 4096, VIP_ERR_UNSUPPORTED, "vip_gap_ln_dense_s32: C=%d > 4096", C);
    VIP_REQUIRE(img_stride % 4 == 0, VIP_ERR_ALIGNMENT, "vip_gap_ln_dense_s32: image stride must be a multiple of 4 floats");
    S32_ALIGNED4(C, ldx);
    hipLaunchKernelGGL(sgap_ln_dense_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, W, bias, out, HW, C, ldx,
                       img_stride, N);
    return vip_launch_status("vip_gap_ln_dense_s32");
}

extern "C" int vip_window_attn_fwd_s32(const float* qkv, const float* q_global, const float* bias_table, float* out, int B, int Hp, int Wp,
                                       int C, int heads, int ws, int nq, float scale, void* stream) {
    VIP_REQUIRE(qkv && bias_table && out, VIP_ERR_BAD_ARG, "vip_window_attn_fwd_s32: null pointer");
    VIP_REQUIRE(B > 0 && Hp > 0 && Wp > 0 && C > 0 && heads > 0 && ws > 0, VIP_ERR_BAD_ARG, "vip_window_attn_fwd_s32: non-positive dimension");
    VIP_REQUIRE((nq == 3 && !q_global) || (nq == 2 && q_global), VIP_ERR_BAD_ARG, "vip_window_attn_fwd_s32: nq=3 without q_global or nq=2 with it");
    VIP_REQUIRE(Hp % ws == 0 && Wp % ws == 0, VIP_ERR_BAD_ARG, "vip_window_attn_fwd_s32: feature map is not a multiple of the window");
    VIP_REQUIRE(C == heads * 32, VIP_ERR_UNSUPPORTED, "vip_window_attn_fwd_s32: head_dim must be 32 (C=%d heads=%d)", C, heads);
    VIP_REQUIRE(ws * ws <= 256, VIP_ERR_UNSUPPORTED, "vip_window_attn_fwd_s32: window of %d tokens (max 256)", ws * ws);
    SAttnArgs a{qkv, q_global, bias_table, out, B, Hp, Wp, C, heads, ws, nq, ws * ws, scale};
    const long items = (long)B * (Hp / ws) * (Wp / ws) * heads;
    VIP_REQUIRE(items < (1L << 31), VIP_ERR_UNSUPPORTED, "vip_window_attn_fwd_s32: too many work items");
    return launch_sattn<32, true>(a, items, (hipStream_t)stream, "vip_window_attn_fwd_s32");
}

extern "C" int vip_mhsa_fwd_s32(const float* qkv, float* out, int B, int N, int D, int heads, float scale, void* stream) {
    VIP_REQUIRE(qkv && out, VIP_ERR_BAD_ARG, "vip_mhsa_fwd_s32: null pointer");
    VIP_REQUIRE(B > 0 && N > 0 && D > 0 && heads > 0, VIP_ERR_BAD_ARG, "vip_mhsa_fwd_s32: non-positive dimension");
    VIP_REQUIRE(D == heads * 64, VIP_ERR_UNSUPPORTED, "vip_mhsa_fwd_s32: head_dim must be 64 (D=%d heads=%d)", D, heads);
    VIP_REQUIRE(N <= 256, VIP_ERR_UNSUPPORTED, "vip_mhsa_fwd_s32: N=%d tokens (max 256)", N);
    SAttnArgs a{qkv, nullptr, nullptr, out, B, 0, 0, D, heads, 0, 3, N, scale};
    return launch_sattn<64, false>(a, (long)B * heads, (hipStream_t)stream, "vip_mhsa_fwd_s32");
}
