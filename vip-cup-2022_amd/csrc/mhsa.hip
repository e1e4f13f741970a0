// ViT multi-head self-attention core for CDNA4: out = softmax(scale * q k^T) v   (no bias, no mask)
// Reference: models/tfimm/architectures/vit.py:148-167.  qkv [B,N,3D] with channels (q|k|v, head, hd),
// out [B,N,D]; head_dim 64, N <= 224 (ViT/16 @224: N = 197).
//
// Work item = (image, head); one workgroup (4 waves) per item.  K and V (N x 64 halfs each) are staged
// once into LDS (coalesced, zero-padded to 224 rows); the 16-query tiles are dealt round-robin to the
// waves and their Q fragments come straight from global.  Per tile: S^T = K Q^T (2 MFMA 16x16x32 per key
// tile), softmax over keys register-local + two lane exchanges, O^T = V^T P^T with V^T fragments from
// ds_read_b64_tr_b16 (same scheme as window_attn.hip).
//
// LDS images, 128-byte rows: K chunk c of row r at c ^ (r & 7); V 32-byte slot s of row r at
// s ^ ((r >> 1) & 3) — both fragment read patterns are bank-conflict free.
#include "common.hpp"

namespace {

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct MhsaArgs {
    const f16* qkv;
    f16* out;
    int B, N, D, heads;
    float scale_log2e;
};

constexpr int NKT_MAX = 14;  // key tiles of 16 -> N <= 224
constexpr int RPM = NKT_MAX * 16;

__global__ __launch_bounds__(256, 2) void mhsa_kernel(MhsaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* v_lds = smem + RPM * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int item = blockIdx.x;
    const int head = item % a.heads;
    const int b = item / a.heads;
    const int ld = 3 * a.D;
    const f16* base = a.qkv + (long)b * a.N * ld + head * 64;
    const int nkt = (a.N + 15) >> 4;           // key tiles actually holding keys (uniform)
    const int nks = (nkt + 1) >> 1;            // 32-key PV steps
    const int nqt = nkt;

    // ---- Q fragments for this wave's query tiles (<= 4), straight from global ----
    U4H8 qf[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qn = (wave + i * 4) * 16 + l15;
        const bool valid = qn < a.N;
        const f16* src = valid ? base + (long)qn * ld + g * 8 : a.qkv;
        const uint4 v0 = *reinterpret_cast<const uint4*>(src);
        const uint4 v1 = *reinterpret_cast<const uint4*>(src + (valid ? 32 : 0));
        qf[i][0].u = valid ? v0 : make_uint4(0, 0, 0, 0);
        qf[i][1].u = valid ? v1 : make_uint4(0, 0, 0, 0);
    }

    // ---- stage K, V: 2 arrays x 224 rows x 8 chunks of 16 B = 3584 slots / 256 threads = 14 each ----
    {
        uint4 st[14];
#pragma unroll
        for (int it = 0; it < 14; ++it) {
            const int s = tid + it * 256;
            const int arr = s / (RPM * 8);
            const int rem = s - arr * (RPM * 8);
            const int row = rem >> 3, ch = rem & 7;
            const bool valid = row < a.N;
            const f16* src = valid ? base + (long)row * ld + (1 + arr) * a.D + ch * 8 : a.qkv;
            const uint4 v = *reinterpret_cast<const uint4*>(src);
            st[it] = valid ? v : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 14; ++it) {
            const int s = tid + it * 256;
            const int arr = s / (RPM * 8);
            const int rem = s - arr * (RPM * 8);
            const int row = rem >> 3, ch = rem & 7;
            const int pch = (arr == 0) ? (ch ^ (row & 7)) : ((((ch >> 1) ^ ((row >> 1) & 3)) << 1) | (ch & 1));
            *reinterpret_cast<uint4*>((arr == 0 ? k_lds : v_lds) + row * 128 + pch * 16) = st[it];
        }
    }
    __syncthreads();

    const float sc = a.scale_log2e;
    const int tr_q = l15 >> 2, tr_p = l15 & 3;

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qt = wave + i * 4;
        if (qt >= nqt) break;  // wave-uniform
        const int qn = qt * 16 + l15;

        f32x4 acc[NKT_MAX];
        float m = -1.0e30f;
#pragma unroll
        for (int t = 0; t < NKT_MAX; ++t) {
            acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (t < nkt) {  // uniform
                const int row = t * 16 + l15;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    U4H8 kf;
                    kf.u = *reinterpret_cast<const uint4*>(k_lds + row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4));
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf.h, qf[i][ks].h, acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * 16 + 4 * g + r;
                    const float s = (key < a.N) ? acc[t][r] * sc : -1.0e30f;
                    acc[t][r] = s;
                    m = fmaxf(m, s);
                }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float lsum = 0.f;
#pragma unroll
        for (int t = 0; t < NKT_MAX; ++t) {
            if (t < nkt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(acc[t][r] - m);
                    acc[t][r] = p;
                    lsum += p;
                }
            }
        }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);

        // O^T = V^T P^T: 4 head-dim tiles of 16; MFMA k-slot (g, j) carries key 32s + 4g + j (j<4) / 32s+16+4g+(j-4)
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NKT_MAX / 2; ++s) {
            if (s < nks) {  // uniform; tiles beyond nkt hold p = 0 only if computed, so zero them explicitly
                U4H8 pf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pf.e[j] = (f16)acc[2 * s][j];
                    pf.e[4 + j] = (2 * s + 1 < nkt) ? (f16)acc[2 * s + 1][j] : (f16)0.f;
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    union {
                        fp16x4_t t[2];
                        f16x8 v;
                    } vf;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int row = 32 * s + 16 * hh + 4 * g + tr_q;
                        const int slot = dt ^ ((row >> 1) & 3);
                        vf.t[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_t*)(v_lds + row * 128 + slot * 32 + tr_p * 8));
                    }
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf.v, pf.h, o[dt], 0, 0, 0);
                }
            }
        }

        if (qn < a.N) {
            const float inv = 1.f / lsum;
            f16* dst = a.out + ((long)b * a.N + qn) * a.D + head * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (f16)(o[dt][r] * inv);
                *reinterpret_cast<f16x4*>(dst + 16 * dt) = ov;
            }
        }
    }
}

// x[b,0,:] = cls + pos[0];  x[b,1+i,:] = patches[b,i,:] + pos[1+i]      (vit.py:419-426)
__global__ __launch_bounds__(256) void vit_tokens_kernel(const f16* __restrict__ patches, const f16* __restrict__ cls,
                                                         const f16* __restrict__ pos, f16* __restrict__ out, int B,
                                                         int NP, int D8) {
    const long total = (long)B * (NP + 1) * D8;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % D8);
        const long tokb = idx / D8;
        const int tok = (int)(tokb % (NP + 1));
        const int b = (int)(tokb / (NP + 1));
        U4H8 x, p, o;
        // unconditional loads (tok 0 reads patch row 0 and discards it)
        x.u = *reinterpret_cast<const uint4*>(patches + (((long)b * NP + (tok > 0 ? tok - 1 : 0)) * D8 + c8) * 8);
        const uint4 c = *reinterpret_cast<const uint4*>(cls + c8 * 8);
        if (tok == 0) x.u = c;
        p.u = *reinterpret_cast<const uint4*>(pos + ((long)tok * D8 + c8) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = (f16)((float)x.e[j] + (float)p.e[j]);
        *reinterpret_cast<uint4*>(out + idx * 8) = o.u;
    }
}

}  // namespace

extern "C" int vip_mhsa_fwd_f16(const void* qkv, void* out, int B, int N, int D, int heads, float scale,
                                void* stream) {
    VIP_REQUIRE(qkv && out, VIP_ERR_BAD_ARG, "vip_mhsa_fwd_f16: null pointer");
    VIP_REQUIRE(B > 0 && N > 0 && D > 0 && heads > 0, VIP_ERR_BAD_ARG, "vip_mhsa_fwd_f16: non-positive dimension");
    VIP_REQUIRE(D == heads * 64, VIP_ERR_UNSUPPORTED, "vip_mhsa_fwd_f16: head_dim = D/heads must be 64 (D=%d heads=%d)", D, heads);
    VIP_REQUIRE(N <= RPM, VIP_ERR_UNSUPPORTED, "vip_mhsa_fwd_f16: N=%d > %d", N, RPM);
    MhsaArgs a;
    a.qkv = (const f16*)qkv; a.out = (f16*)out; a.B = B; a.N = N; a.D = D; a.heads = heads;
    a.scale_log2e = scale * 1.44269504088896f;
    constexpr int SMEM = 2 * RPM * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(mhsa_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        attr_set = true;
    }
    hipLaunchKernelGGL(mhsa_kernel, dim3(B * heads), dim3(256), SMEM, (hipStream_t)stream, a);
    return vip_launch_status("vip_mhsa_fwd_f16");
}

extern "C" int vip_vit_tokens_f16(const void* patches, const void* cls_token, const void* pos_embed, void* out, int B,
                                  int n_patches, int D, void* stream) {
    VIP_REQUIRE(patches && cls_token && pos_embed && out, VIP_ERR_BAD_ARG, "vip_vit_tokens_f16: null pointer");
    VIP_REQUIRE(B > 0 && n_patches > 0 && D > 0, VIP_ERR_BAD_ARG, "vip_vit_tokens_f16: non-positive dimension");
    VIP_REQUIRE(D % 8 == 0, VIP_ERR_ALIGNMENT, "vip_vit_tokens_f16: D must be a multiple of 8");
    const long total = (long)B * (n_patches + 1) * (D / 8);
    long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(vit_tokens_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const f16*)patches,
                       (const f16*)cls_token, (const f16*)pos_embed, (f16*)out, B, n_patches, D / 8);
    return vip_launch_status("vip_vit_tokens_f16");
}
