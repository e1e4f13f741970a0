// STRICT precision path, part 1: Conv2D / Dense with fp32 storage and fp32 matrix arithmetic.
//
// The reference computes every member in fp32 (main.py:107-109: tf.keras.models.load_model + model.predict, no mixed-precision
// policy anywhere), and BASELINE.json asks for |dz| <= 1e-3 on every member's logit.  The default ("fast") path stores activations
// and weights in fp16; its member-logit error is the fp16 storage floor (7e-4 ... 8e-3).  This file is the
// mode in which the stated tolerance holds: activations and weights stay fp32 in HBM and the contraction runs on the f32-input
// matrix instruction v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulate: bitwise a k-ordered fmaf chain; 64 FLOP/clk/SIMD
// = 157 TFLOP/s on the chip, 1/16 of the fp16 MFMA rate - the price of the tolerance).
//
// One implicit-GEMM kernel covers every Conv2D (+ folded BatchNorm) (+ activation) (+ residual) and every Dense of the four
// model families (same vip_conv_desc as vip_conv2d_nhwc_f16; strides in FLOATS here):
//   models/resnet_rs/resnet_rs_model.py:64-84,97-139,235-280 ; kecam common_layers.py:190-248 ; gcvit/layers/*.py Dense / Conv2D ;
//   tfimm/architectures/{vit,convnext}.py Dense / Conv2D.
//
// Tiling: the WEIGHTS are the MFMA A operand (rows = output channels), the PIXELS the B operand (columns), so a lane ends up with 4
// consecutive output channels of one pixel per accumulator quad = one 16-byte store.  Workgroup = 4 waves, block tile
// (32*WCH*WGC channels) x (32*WPX*WGP pixels) x 32 k; both operand tiles are staged global -> VGPR -> LDS (double buffered, the
// global loads of chunk i+1 are in flight while chunk i is multiplied), LDS rows padded to 36 floats (ds_read_b128 conflict-free).
// K order inside a 32-k chunk: lane half h of k-step (c, j) holds k = 8c + 4h + j for BOTH operands (one ds_read_b128 per operand
// row and 8 k), which is a permutation of the chunk's k - a sum does not care.
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr int SBK = 32;          // k per chunk
constexpr int SLD = 36;          // LDS row stride in floats

struct SConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* res;
    float* y;
    int B, H, W, Ho, Wo;
    int kh, kw, sh, sw, pt, pl;
    int cin_g, cout_g, groups;
    int ldx, cin_off, ldy, cout_off, ldr, res_off, ldw;
    int act_pre, act_post;
    int M, K;                    // pixels, kh*kw*cin_g
};

template <int WGC, int WGP, int WCH, int WPX>
__global__ __launch_bounds__(256, 2) void sconv_kernel(SConvArgs a) {
    constexpr int TCH = 32 * WCH * WGC;       // channels per block
    constexpr int TPX = 32 * WPX * WGP;       // pixels per block
    constexpr int LA = TCH / 32, LB = TPX / 32;   // float4 loads per thread and chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    // [2 stages][TCH + TPX rows][SLD]
    constexpr int STAGE = (TCH + TPX) * SLD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wc = wave % WGC, wp = wave / WGC;
    const int g = blockIdx.z;
    const int ch0 = blockIdx.y * TCH;            // first channel of the block inside the group
    const long px0 = (long)blockIdx.x * TPX;

    // ---- staging roles: float4 kq of rows r0 + 32 i ----
    const int kq = tid & 7, r0 = tid >> 3;
    const float* wrow[LA];
    bool wok[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int n = ch0 + r0 + 32 * i;
        wok[i] = n < a.cout_g;
        wrow[i] = a.w + (long)(g * a.cout_g + (wok[i] ? n : 0)) * a.ldw;
    }
    long xbase[LB];
    int hi0[LB], wi0[LB];
    bool pok[LB];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const long m = px0 + r0 + 32 * i;
        pok[i] = m < a.M;
        const long mm = pok[i] ? m : 0;
        const int b = (int)(mm / HoWo);
        const int rem = (int)(mm - (long)b * HoWo);
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        hi0[i] = ho * a.sh - a.pt;
        wi0[i] = wo * a.sw - a.pl;
        xbase[i] = (long)b * a.H * a.W;
    }
    const int cbase = a.cin_off + g * a.cin_g;
    const bool pointwise = a.kh == 1 && a.kw == 1;

    f32x4 ra[LA], rb[LB];
    auto fetch = [&](int k0) {
        const int k = k0 + 4 * kq;
        const bool kok = k < a.K;
        int c = k, r = 0, s = 0;
        if (!pointwise) {
            const int tap = k / a.cin_g;
            c = k - tap * a.cin_g;
            r = tap / a.kw;
            s = tap - r * a.kw;
        }
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            ra[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (kok && wok[i]) ra[i] = *reinterpret_cast<const f32x4*>(wrow[i] + k);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            rb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int hi = hi0[i] + r, wi = wi0[i] + s;
            if (kok && pok[i] && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W)
                rb[i] = *reinterpret_cast<const f32x4*>(a.x + (xbase[i] + (long)hi * a.W + wi) * a.ldx + cbase + c);
        }
    };
    auto stash = [&](int stage) {
        float* sa = lds + stage * STAGE;
        float* sb = sa + TCH * SLD;
#pragma unroll
        for (int i = 0; i < LA; ++i) *reinterpret_cast<f32x4*>(sa + (r0 + 32 * i) * SLD + 4 * kq) = ra[i];
#pragma unroll
        for (int i = 0; i < LB; ++i) *reinterpret_cast<f32x4*>(sb + (r0 + 32 * i) * SLD + 4 * kq) = rb[i];
    };

    f32x16 acc[WCH][WPX];
#pragma unroll
    for (int i = 0; i < WCH; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nchunks = (a.K + SBK - 1) / SBK;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int ci = 0; ci < nchunks; ++ci) {
        const int cur = ci & 1;
        if (ci + 1 < nchunks) fetch((ci + 1) * SBK);
        const float* sa = lds + cur * STAGE + (wc * WCH * 32 + l31) * SLD + 4 * h;
        const float* sb = lds + cur * STAGE + TCH * SLD + (wp * WPX * 32 + l31) * SLD + 4 * h;
#pragma unroll
        for (int c8 = 0; c8 < SBK / 8; ++c8) {
            f32x4 fa[WCH], fb[WPX];
#pragma unroll
            for (int i = 0; i < WCH; ++i) fa[i] = *reinterpret_cast<const f32x4*>(sa + i * 32 * SLD + 8 * c8);
#pragma unroll
            for (int j = 0; j < WPX; ++j) fb[j] = *reinterpret_cast<const f32x4*>(sb + j * 32 * SLD + 8 * c8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < WCH; ++i)
#pragma unroll
                    for (int j = 0; j < WPX; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
        if (ci + 1 < nchunks) stash(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane = pixel l31 of tile column j; accumulator quad q = channels 8q + 4h .. + 3 of tile row i ----
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        const long m = px0 + (wp * WPX + j) * 32 + l31;
        if (m >= a.M) continue;
        float* yrow = a.y + m * a.ldy + a.cout_off + g * a.cout_g;
        const float* rrow = a.res ? a.res + m * a.ldr + a.res_off + g * a.cout_g : nullptr;
#pragma unroll
        for (int i = 0; i < WCH; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = ch0 + (wc * WCH + i) * 32 + 8 * q + 4 * h;
                if (n >= a.cout_g) continue;
                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + g * a.cout_g + n);
                if (a.act_pre) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = vip_act_strict(v[e], a.act_pre);
                }
                if (rrow) v += *reinterpret_cast<const f32x4*>(rrow + n);
                if (a.act_post) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = vip_act_strict(v[e], a.act_post);
                }
                *reinterpret_cast<f32x4*>(yrow + n) = v;
            }
    }
}

template <int WGC, int WGP, int WCH, int WPX>
int launch_sconv(const SConvArgs& a, hipStream_t s) {
    constexpr int TCH = 32 * WCH * WGC, TPX = 32 * WPX * WGP;
    constexpr int smem = 2 * (TCH + TPX) * SLD * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sconv_kernel<WGC, WGP, WCH, WPX>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr = true;
    }
    const long gx = (a.M + TPX - 1) / TPX;
    const int gy = (a.cout_g + TCH - 1) / TCH;
    if (gx > 2147483647L || gy > 65535 || a.groups > 65535) {
        vip_set_error("vip_conv2d_nhwc_s32: grid too large (%ld x %d x %d)", gx, gy, a.groups);
        return VIP_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((sconv_kernel<WGC, WGP, WCH, WPX>), dim3((unsigned)gx, gy, a.groups), dim3(256), smem, s, a);
    return vip_launch_status("vip_conv2d_nhwc_s32");
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same convolution with every fp32 operand split into THREE bf16 terms (x = b0 + b1 + b2 exactly: 8 + 8 + 8 significant bits) and
// the products kept down to 2^-16 of the leading one: x w ~= b0 c0 + (b0 c1 + b1 c0) + (b1 c1 + b0 c2 + b2 c0) - six
// v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block, f32 accumulate.  What is dropped (b1 c2 + b2 c1 + b2 c2) is <= 3 * 2^-24 of the
// product: f32-quality results (tests/test_gpu_strict.py holds it to the same 2e-5 as the f32-MFMA kernel) at 16 / 6 = 2.7 x the matrix
// rate of v_mfma_f32_32x32x2_f32, and with bf16's f32 exponent range there is nothing to scale and nothing that can overflow.
// Weights arrive pre-split (three bf16 planes [3][Cout][ldw], made once at load time by ops.make_conv_weight); activations are split
// on their way into LDS (v_cvt_pk_bf16_f32 + two exact f32 subtractions per term).  Block tile (32 TCHW x 2) channels x 128 pixels x 32
// k, 4 waves as 2 x 2, one LDS stage of three planes per operand (rows of 64 B + 16 B pad: conflict-free ds_read_b128), the next chunk
// prefetched into registers while the current one is multiplied; weights are the MFMA A operand as in sconv_kernel (a lane ends
// up with 4 consecutive channels of one pixel).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int S6_ROWB = 80;          // LDS bytes per row of a plane: 32 bf16 + 16 B pad

struct SConv6Args {
    SConvArgs c;
    const unsigned short* wp;        // [3][Cout_total][ldwp] bf16
    long plane_stride;               // elements between planes
    int ldwp;
};

__device__ __forceinline__ unsigned s6_pk(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
// 8 consecutive f32 -> three uint4 of 8 bf16 each (planes 0, 1, 2)
__device__ __forceinline__ void s6_split8(const f32x4& lo, const f32x4& hi, uint4& p0, uint4& p1, uint4& p2) {
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    unsigned o0[4], o1[4], o2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        const unsigned q0 = s6_pk(a, b);
        const float ra = a - __builtin_bit_cast(float, q0 << 16), rb = b - __builtin_bit_cast(float, q0 & 0xffff0000u);
        const unsigned q1 = s6_pk(ra, rb);
        const float sa = ra - __builtin_bit_cast(float, q1 << 16), sb = rb - __builtin_bit_cast(float, q1 & 0xffff0000u);
        o0[i] = q0;
        o1[i] = q1;
        o2[i] = s6_pk(sa, sb);
    }
    p0 = make_uint4(o0[0], o0[1], o0[2], o0[3]);
    p1 = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    p2 = make_uint4(o2[0], o2[1], o2[2], o2[3]);
}

template <int TCHW, int NP = 3>      // 16-channel tiles per wave: 4 (block = 128 channels) or 2 (64 channels); NP bf16 terms per operand
__global__ __launch_bounds__(256, NP == 2 ? 3 : 2) void sconv6_kernel(SConv6Args aa) {
    const SConvArgs& a = aa.c;
    constexpr int TCH = 32 * TCHW, TPX = 128;
    constexpr int PLANE_A = TCH * S6_ROWB, PLANE_B = TPX * S6_ROWB;
    __shared__ __attribute__((aligned(16))) char smem[NP * PLANE_A + NP * PLANE_B];
    char* sA = smem;
    char* sB = smem + NP * PLANE_A;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const int wc = wave & 1, wp_ = wave >> 1;
    const int g = blockIdx.z;
    const int ch0 = blockIdx.y * TCH;
    const long px0 = (long)blockIdx.x * TPX;

    // staging roles: 8 consecutive k (index kq, 0..3) of rows r0 + 64 i
    const int kq = tid & 3, r0 = tid >> 2;
    constexpr int LA = (TCH + 63) / 64, LB = TPX / 64;
    long wrow[LA];
    bool wok[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int n = ch0 + r0 + 64 * i;
        wok[i] = (r0 + 64 * i < TCH) && n < a.cout_g;
        wrow[i] = (long)(g * a.cout_g + (wok[i] ? n : 0)) * aa.ldwp;
    }
    long xbase[LB];
    int hi0[LB], wi0[LB];
    bool pok[LB];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const long m = px0 + r0 + 64 * i;
        pok[i] = m < a.M;
        const long mm = pok[i] ? m : 0;
        const int b = (int)(mm / HoWo);
        const int rem = (int)(mm - (long)b * HoWo);
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        hi0[i] = ho * a.sh - a.pt;
        wi0[i] = wo * a.sw - a.pl;
        xbase[i] = (long)b * a.H * a.W;
    }
    const int cbase = a.cin_off + g * a.cin_g;
    const bool pointwise = a.kh == 1 && a.kw == 1;

    uint4 ra[LA][NP];
    f32x4 rb[LB][2];
    auto fetch = [&](int k0) {
        const int k = k0 + 8 * kq;
        const bool kok = k < a.K;            // weight planes are zero-padded to a multiple of 8 k: a group is in or out as a whole
#pragma unroll
        for (int i = 0; i < LA; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                ra[i][p] = make_uint4(0, 0, 0, 0);
                if (kok && wok[i]) ra[i][p] = *reinterpret_cast<const uint4*>(aa.wp + p * aa.plane_stride + wrow[i] + k);
            }
#pragma unroll
        for (int h = 0; h < 2; ++h) {        // the two 4-k halves of the group may belong to different filter taps
            const int kh_ = k + 4 * h;
            const bool hok = kh_ < a.K;
            int c = kh_, r = 0, s = 0;
            if (!pointwise) {
                const int tap = kh_ / a.cin_g;
                c = kh_ - tap * a.cin_g;
                r = tap / a.kw;
                s = tap - r * a.kw;
            }
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                rb[i][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const int hi = hi0[i] + r, wi = wi0[i] + s;
                if (hok && pok[i] && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W)
                    rb[i][h] = *reinterpret_cast<const f32x4*>(a.x + (xbase[i] + (long)hi * a.W + wi) * a.ldx + cbase + c);
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < LA; ++i)
            if (r0 + 64 * i < TCH) {
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<uint4*>(sA + p * PLANE_A + (r0 + 64 * i) * S6_ROWB + kq * 16) = ra[i][p];
            }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            uint4 p0, p1, p2;
            s6_split8(rb[i][0], rb[i][1], p0, p1, p2);
            char* dst = sB + (r0 + 64 * i) * S6_ROWB + kq * 16;
            *reinterpret_cast<uint4*>(dst) = p0;
            *reinterpret_cast<uint4*>(dst + PLANE_B) = p1;
            if constexpr (NP == 3) *reinterpret_cast<uint4*>(dst + 2 * PLANE_B) = p2;
        }
    };

    f32x4 acc[TCHW][4];
#pragma unroll
    for (int i = 0; i < TCHW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunks = (a.K + SBK - 1) / SBK;
    fetch(0);
    for (int ci = 0; ci < nchunks; ++ci) {
        stash();
        __syncthreads();
        if (ci + 1 < nchunks) fetch((ci + 1) * SBK);
        const char* fa = sA + (wc * TCHW * 16 + l15) * S6_ROWB + kg * 16;
        const char* fb = sB + (wp_ * 64 + l15) * S6_ROWB + kg * 16;
        bf16x8 bq[4][NP];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p) bq[j][p] = *reinterpret_cast<const bf16x8*>(fb + j * 16 * S6_ROWB + p * PLANE_B);
#pragma unroll
        for (int i = 0; i < TCHW; ++i) {
            bf16x8 aq[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) aq[p] = *reinterpret_cast<const bf16x8*>(fa + i * 16 * S6_ROWB + p * PLANE_A);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 c = acc[i][j];
                // smallest terms first
                if constexpr (NP == 3) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[2], bq[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[0], bq[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[1], bq[j][1], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[1], bq[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[0], bq[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[0], bq[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
        }
        __syncthreads();
    }

    // epilogue: lane = pixel l15 of pixel tile j, accumulator = channels 16 i + 4 kg .. + 3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long m = px0 + wp_ * 64 + j * 16 + l15;
        if (m >= a.M) continue;
        float* yrow = a.y + m * a.ldy + a.cout_off + g * a.cout_g;
        const float* rrow = a.res ? a.res + m * a.ldr + a.res_off + g * a.cout_g : nullptr;
#pragma unroll
        for (int i = 0; i < TCHW; ++i) {
            const int n = ch0 + (wc * TCHW + i) * 16 + 4 * kg;
            if (n >= a.cout_g) continue;
            f32x4 v = acc[i][j];
            if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + g * a.cout_g + n);
            if (a.act_pre) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = vip_act_strict(v[e], a.act_pre);
            }
            if (rrow) v += *reinterpret_cast<const f32x4*>(rrow + n);
            if (a.act_post) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = vip_act_strict(v[e], a.act_post);
            }
            *reinterpret_cast<f32x4*>(yrow + n) = v;
        }
    }
}

template <int TCHW, int NP = 3>
int launch_sconv6(const SConv6Args& aa, hipStream_t s) {
    constexpr int TCH = 32 * TCHW, TPX = 128;
    const SConvArgs& a = aa.c;
    const long gx = (a.M + TPX - 1) / TPX;
    const int gy = (a.cout_g + TCH - 1) / TCH;
    if (gx > 2147483647L || gy > 65535 || a.groups > 65535) {
        vip_set_error("vip_conv2d_nhwc_s32x: grid too large (%ld x %d x %d)", gx, gy, a.groups);
        return VIP_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((sconv6_kernel<TCHW, NP>), dim3((unsigned)gx, gy, a.groups), dim3(256), 0, s, aa);
    return vip_launch_status("vip_conv2d_nhwc_s32x");
}

}  // namespace

static int sconv_fill(const char* who, const float* x, const void* w, const float* bias, const float* residual, float* y,
                      const vip_conv_desc* d, int ldw, SConvArgs& a) {
    VIP_REQUIRE(x && w && y && d, VIP_ERR_BAD_ARG, "%s: null pointer", who);
    VIP_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->kh > 0 && d->kw > 0 && d->sh > 0 && d->sw > 0 &&
                    d->pt >= 0 && d->pl >= 0 && d->Ho > 0 && d->Wo > 0 && d->groups > 0,
                VIP_ERR_BAD_ARG, "%s: non-positive dimension", who);
    VIP_REQUIRE(d->Cin % d->groups == 0 && d->Cout % d->groups == 0, VIP_ERR_BAD_ARG, "%s: channels not divisible by groups", who);
    VIP_REQUIRE((unsigned)d->act_pre <= 4u && (unsigned)d->act_post <= 4u, VIP_ERR_BAD_ARG, "%s: bad activation code", who);
    const int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups;
    VIP_REQUIRE(cin_g % 4 == 0 && cout_g % 4 == 0 && d->ldx % 4 == 0 && d->ldy % 4 == 0 && ldw % 4 == 0 && d->cin_off % 4 == 0 &&
                    d->cout_off % 4 == 0 && d->ldr % 4 == 0 && d->res_off % 4 == 0,
                VIP_ERR_ALIGNMENT, "%s: channel counts, strides and offsets must be multiples of 4 floats", who);
    VIP_REQUIRE(d->ldx >= d->cin_off + d->Cin && d->ldy >= d->cout_off + d->Cout && ldw >= d->kh * d->kw * cin_g, VIP_ERR_BAD_ARG,
                "%s: a stride is smaller than the channels it spans", who);
    VIP_REQUIRE(!residual || d->ldr >= d->res_off + d->Cout, VIP_ERR_BAD_ARG, "%s: residual stride too small", who);
    // the output must be what the padding implies at most (a caller may ask for fewer rows / columns, never more taps than exist)
    VIP_REQUIRE((long)(d->Ho - 1) * d->sh - d->pt < d->H && (long)(d->Wo - 1) * d->sw - d->pl < d->W, VIP_ERR_BAD_ARG,
                "%s: output size reaches past the input", who);
    a.x = x; a.w = (const float*)w; a.bias = bias; a.res = residual; a.y = y;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo;
    a.kh = d->kh; a.kw = d->kw; a.sh = d->sh; a.sw = d->sw; a.pt = d->pt; a.pl = d->pl;
    a.cin_g = cin_g; a.cout_g = cout_g; a.groups = d->groups;
    a.ldx = d->ldx; a.cin_off = d->cin_off; a.ldy = d->ldy; a.cout_off = d->cout_off; a.ldr = d->ldr; a.res_off = d->res_off; a.ldw = ldw;
    a.act_pre = d->act_pre; a.act_post = d->act_post;
    const long M = (long)d->B * d->Ho * d->Wo;
    VIP_REQUIRE(M < (1L << 31), VIP_ERR_UNSUPPORTED, "%s: more than 2^31 output pixels", who);
    a.M = (int)M;
    a.K = d->kh * d->kw * cin_g;
    return VIP_OK;
}

extern "C" int vip_conv2d_nhwc_s32(const float* x, const float* w, const float* bias, const float* residual, float* y,
                                   const vip_conv_desc* d, void* stream) {
    SConvArgs a;
    const int st = sconv_fill("vip_conv2d_nhwc_s32", x, w, bias, residual, y, d, d ? d->ldw : 0, a);
    if (st != VIP_OK) return st;
    hipStream_t s = (hipStream_t)stream;
    if (a.cout_g > 64) return launch_sconv<2, 2, 2, 2>(a, s);      // 128 channels x 128 pixels
    if (a.cout_g > 32) return launch_sconv<2, 2, 1, 2>(a, s);      //  64 channels x 128 pixels
    return launch_sconv<1, 4, 1, 1>(a, s);                         //  32 channels x 128 pixels
}

/* w_planes: the weights split into three bf16 planes [3][Cout][ldwp] (ldwp % 8 == 0, zero padded; plane p starts p * Cout * ldwp
 * elements in): w = p0 + p1 + p2 exactly.  d->ldw is ignored. */
extern "C" int vip_conv2d_nhwc_s32x(const float* x, const void* w_planes, int ldwp, const float* bias, const float* residual, float* y,
                                    const vip_conv_desc* d, void* stream) {
    SConv6Args aa;
    VIP_REQUIRE(ldwp > 0 && ldwp % 8 == 0, VIP_ERR_ALIGNMENT, "vip_conv2d_nhwc_s32x: ldwp must be a positive multiple of 8");
    const int st = sconv_fill("vip_conv2d_nhwc_s32x", x, w_planes, bias, residual, y, d, ldwp, aa.c);
    if (st != VIP_OK) return st;
    aa.wp = (const unsigned short*)w_planes;
    aa.ldwp = ldwp;
    aa.plane_stride = (long)d->Cout * ldwp;
    hipStream_t s = (hipStream_t)stream;
    // short K = HBM-bound (fp32 tensors): the 64-channel tile (158 VGPRs, 45 KB: three workgroups per CU instead of two) keeps more
    // loads in flight; deeper K stays on the 128-channel tile (half the activation re-reads per MFMA)
    static const int narrow_max_k = getenv("VIP_S6_NARROW_MAXK") ? atoi(getenv("VIP_S6_NARROW_MAXK")) : 192;   // 192.7 -> 184.7 ms of GEMM per strict step
    if (aa.c.cout_g > 64 && aa.c.K > narrow_max_k) return launch_sconv6<4>(aa, s);
    return launch_sconv6<2>(aa, s);
}

/* The same with TWO bf16 terms per operand (planes 0 and 1 of the same w_planes tensor; x w ~= b0 c0 + b0 c1 + b1 c0: three MFMAs per
 * block, 2^-17 of the product dropped - 64 x finer than fp16 storage, not f32 quality: measured member errors in DESIGN.md section 4). */
extern "C" int vip_conv2d_nhwc_s32x2(const float* x, const void* w_planes, int ldwp, const float* bias, const float* residual, float* y,
                                     const vip_conv_desc* d, void* stream) {
    SConv6Args aa;
    VIP_REQUIRE(ldwp > 0 && ldwp % 8 == 0, VIP_ERR_ALIGNMENT, "vip_conv2d_nhwc_s32x2: ldwp must be a positive multiple of 8");
    const int st = sconv_fill("vip_conv2d_nhwc_s32x2", x, w_planes, bias, residual, y, d, ldwp, aa.c);
    if (st != VIP_OK) return st;
    aa.wp = (const unsigned short*)w_planes;
    aa.ldwp = ldwp;
    aa.plane_stride = (long)d->Cout * ldwp;
    hipStream_t s = (hipStream_t)stream;
    if (aa.c.cout_g > 64) return launch_sconv6<4, 2>(aa, s);
    return launch_sconv6<2, 2>(aa, s);
}
