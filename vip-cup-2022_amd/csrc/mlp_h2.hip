// Fused two-layer MLP on the packed STRICT storage:  y = W2 . gelu(W1 . LN(x) + b1) + b2 (+ residual),  C = 64 / 96 / 128.
//
// The strict-mode form of mlp_fused.hip (same call sites: the Dense -> GELU -> Dense (+Add) tail of tfimm's ConvNeXtBlock,
// convnext.py:200-229 with the layer scale folded into W2 / b2, and of the GCViT MLP, gcvit/layers/feature.py:20-22).  Unfused, strict
// mode wrote the hidden tensor [M, 4C] at 4 bytes per element and read it back (ConvNeXt stage 0: 7.7 GB per block against 2.9 GB for
// x + residual + y) and ran LayerNorm as its own pass over x; here the hidden tensor never leaves the register file and LayerNorm is a
// prologue on the activation fragments.
//
// Structure (mlp_stream_kernel's): 8 waves x 32 tokens per workgroup; x (hi and lo fragments) and the y accumulators stay in registers
// for the whole hidden loop; per 32-channel hidden slice the workgroup stages W1[32q.., :] and W2[:, 32q..] (packed rows, 24.5 KB at
// C = 96) into a double-buffered LDS image - global -> VGPR one slice ahead, ds_write after the slice's math, one barrier per slice.
// Every contraction is the three-MFMA form of the packed storage (w_lo x_hi + w_hi x_hi + w_hi x_lo); in LDS the hi and lo planes of a
// 32-k step are two 64-byte runs, so a fragment read is the same conflict-free ds_read_b128 pattern as in pw_gemm_kernel.  Weights are
// pre-scaled by a power of two (ops.split_h2_weights), the accumulators start at the scaled bias and are multiplied by 1 / scale.
// GELU runs in packed fp32 (vip_gelu2), the hidden activation is split to (hi, lo) in registers - with the interleaved weight rows of
// frag_channel the pair IS the B operand of the second GEMM.
#include "common.hpp"

namespace {

struct MlpH2Args {
    const char* x;
    const char* w1;
    const float* b1;
    const char* w2;
    const float* b2;
    const char* res;
    char* y;
    const float* ln_g;   // optional LayerNorm over C applied to x first (NULL: none)
    const float* ln_b;
    float ln_eps;
    float os1, os2;      // 1 / weight scale of the two layers
    int M, Hd;
    int ldx, ldy, ldr;   // logical elements per row
    int ldw1, ldw2;      // halfs per packed weight row
    int s1, s2;          // LDS row strides (bytes) of the W1 slice [32 rows] and the W2 slice [C rows]
    int n_tiles;         // tiles of 32 tokens per wave
    int* status;
};

// LDS row j of a 32-row group holds channel (j>>4)*4 + ((j&15)>>2)*8 + (j&3): MFMA tiles 2h, 2h+1 then give a lane the channels
// 8*lq .. 8*lq+7 of the group (mlp_fused.hip)
__device__ __forceinline__ int frag_channel(int j) {
    const int t = (j >> 4) & 1, r = j & 15;
    return (j & ~31) + (r >> 2) * 8 + t * 4 + (r & 3);
}
// position of 16-byte chunk c (0..7: hi0 lo0 hi1 lo1 ..) of a 32-k step inside its 128-byte LDS run: hi plane first, then lo plane
__device__ __forceinline__ int h2_pos(int c) { return ((c & 1) << 2) | (c >> 1); }

// LayerNorm on the fragments of one 16-token tile: lane (l15 = token, lq) holds channels 32 ks + 8 lq + 0..7, a token's C channels sit
// in 4 lanes.  fp32 two-pass statistics, the expression of h2_layernorm_kernel; the normalised row is split to (hi, lo) again.
template <int CK>
__device__ __forceinline__ void ln_fragments_h2(U4H8 (&xh)[CK], U4H8 (&xl)[CK], const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                int lq) {
    constexpr int C = 32 * CK;
    float v[CK][8];
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < CK; ++ks) {
        h2_join8(xh[ks], xl[ks], v[ks]);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[ks][j];
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int ks = 0; ks < CK; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = v[ks][j] - mean;
            sq += d * d;
        }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq / (float)C + eps);
#pragma unroll
    for (int ks = 0; ks < CK; ++ks) {
        const float4 g0 = *reinterpret_cast<const float4*>(g + ks * 32 + lq * 8), g1 = *reinterpret_cast<const float4*>(g + ks * 32 + lq * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(b + ks * 32 + lq * 8), b1 = *reinterpret_cast<const float4*>(b + ks * 32 + lq * 8 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[ks][j] - mean) * rstd * gg[j] + bb[j];
        h2_split8(o, xh[ks], xl[ks]);
    }
}

template <int CK, int NW>
__global__ __launch_bounds__(64 * NW, 2) void mlp_h2_kernel(MlpH2Args a) {
    constexpr int C = 32 * CK, PT = 2, NCT = C / 16, NTHR = 64 * NW;
    constexpr int W_IT = 16 * C / NTHR;             // 16-byte weight chunks staged per thread per slice
    constexpr unsigned OOB = 0xFFFFFFE0u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int stage_bytes = 32 * a.s1 + C * a.s2;   // W1 slice [32][s1] then W2 slice [C][s2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)(4L * a.M * a.ldx), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (unsigned)(4L * a.M * a.ldr) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)(4L * a.M * a.ldy), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.b1, 0, a.b1 ? (unsigned)(a.Hd * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.b2, 0, a.b2 ? (unsigned)(C * 4) : 0u, 0x00020000);

    // staging plan of this thread: chunk idx = tid + NTHR i; the first 8C chunks are the W1 slice (32 rows x C/4 chunks), the rest the W2
    // slice (C rows x 8 chunks).  Plain pointers: one load instruction per chunk whichever matrix it comes from.
    const char* w_ptr[W_IT];   // source of slice 0
    unsigned w_step[W_IT];     // bytes between consecutive slices
    int w_dst[W_IT];           // byte offset inside a stage
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int idx = tid + NTHR * i;
        if (idx < 8 * C) {
            const int row = idx / (C / 4), c = idx - row * (C / 4);
            w_ptr[i] = a.w1 + ((long)frag_channel(row) * a.ldw1) * 2 + c * 16;
            w_step[i] = (unsigned)(32 * a.ldw1 * 2);
            w_dst[i] = row * a.s1 + ((c & ~7) | h2_pos(c & 7)) * 16;
        } else {
            const int j = idx - 8 * C, row = j >> 3, c = j & 7;
            w_ptr[i] = a.w2 + ((long)frag_channel(row) * a.ldw2) * 2 + c * 16;
            w_step[i] = 128u;
            w_dst[i] = 32 * a.s1 + row * a.s2 + h2_pos(c) * 16;
        }
    }
    typedef unsigned wvec4 __attribute__((ext_vector_type(4)));      // a native vector: hip's uint4 struct made this array a stack object
    wvec4 wst[W_IT];
    auto load_w = [&](int q) {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) wst[i] = *reinterpret_cast<const wvec4*>(w_ptr[i] + (size_t)q * w_step[i]);
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) *reinterpret_cast<wvec4*>(__builtin_assume_aligned(smem + buf * stage_bytes + w_dst[i], 16)) = wst[i];
    };

    const int nq = a.Hd >> 5;
    load_w(0);
    store_w(0);
    __syncthreads();
    int buf = 0;
    f32x2 ov_sum = {0.f, 0.f};
    float ov_max = 0.f;

    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int m0 = tile * (NTHR / 64 * 16 * PT) + wave * (16 * PT);
        U4H8 xh[CK][PT], xl[CK][PT];
#pragma unroll
        for (int ks = 0; ks < CK; ++ks)
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int m = m0 + p * 16 + l15;
                const unsigned off = m < a.M ? (unsigned)(m * a.ldx * 4 + (ks * 4 + lq) * 32) : OOB;
                xh[ks][p].u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
                xl[ks][p].u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, off + 16u, 0, 0));
            }
        if (a.ln_g) {   // workgroup-uniform
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                U4H8 ch[CK], cl[CK];
#pragma unroll
                for (int ks = 0; ks < CK; ++ks) {
                    ch[ks] = xh[ks][p];
                    cl[ks] = xl[ks][p];
                }
                ln_fragments_h2<CK>(ch, cl, a.ln_g, a.ln_b, a.ln_eps, lq);
#pragma unroll
                for (int ks = 0; ks < CK; ++ks) {
                    xh[ks][p] = ch[ks];
                    xl[ks][p] = cl[ks];
                }
            }
        }
        // y accumulators start at the scaled b2 (lane: channels 32*hh + 8*lq + 4*t + 0..3 for tile 2*hh + t)
        f32x4 acc2[NCT][PT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const f32x4 bv = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb2, (unsigned)(((ct >> 1) * 32 + lq * 8 + (ct & 1) * 4) * 4), 0, 0));
#pragma unroll
            for (int p = 0; p < PT; ++p) acc2[ct][p] = bv;
        }

#pragma unroll 1
        for (int q = 0; q < nq; ++q) {
            load_w(q + 1 < nq ? q + 1 : 0);          // next slice (slice 0 of the next tile after the last one)
            __builtin_amdgcn_sched_barrier(0);
            const char* w1l = smem + buf * stage_bytes + l15 * a.s1 + lq * 16;
            const char* w2l = smem + buf * stage_bytes + 32 * a.s1 + l15 * a.s2 + lq * 16;
            f32x4 acc1[2][PT];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 bv = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb1, (unsigned)((q * 32 + lq * 8 + t * 4) * 4), 0, 0));
#pragma unroll
                for (int p = 0; p < PT; ++p) acc1[t][p] = bv;
            }
#pragma unroll
            for (int ks = 0; ks < CK; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    U4H8 wh, wl;
                    wh.u = *reinterpret_cast<const uint4*>(w1l + t * 16 * a.s1 + ks * 128);
                    wl.u = *reinterpret_cast<const uint4*>(w1l + t * 16 * a.s1 + ks * 128 + 64);
#pragma unroll
                    for (int p = 0; p < PT; ++p) {
                        acc1[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl.h, xh[ks][p].h, acc1[t][p], 0, 0, 0);
                        acc1[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh.h, xl[ks][p].h, acc1[t][p], 0, 0, 0);
                        acc1[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh.h, xh[ks][p].h, acc1[t][p], 0, 0, 0);
                    }
                }
            // activation, split: the (hi, lo) pair is the B operand of the second GEMM (k = 8*lq + j <-> hidden channel 32q + 8*lq + j)
            U4H8 hh[PT], hl[PT];
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const f32x2 v = vip_gelu2((f32x2){acc1[t][p][j], acc1[t][p][j + 1]} * a.os1);
                        // hi = rn16(v); lo = rn16(v - hi) as one v_fma_mixlo / mixhi_f16 per value (fma(hi, -1, v) is exact: the same
                        // single rounding as converting the fp32 difference).  A hidden value beyond the fp16 range becomes Inf here and
                        // NaN / Inf in y: the range check of the outputs below reports it
                        const unsigned hw = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
                        unsigned lw;
                        const float vx = v.x, vy = v.y;
                        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lw) : "v"(hw), "v"(vx));
                        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lw) : "v"(hw), "v"(vy));
                        reinterpret_cast<unsigned*>(&hh[p])[(t * 4 + j) >> 1] = hw;
                        reinterpret_cast<unsigned*>(&hl[p])[(t * 4 + j) >> 1] = lw;
                    }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                U4H8 wh, wl;
                wh.u = *reinterpret_cast<const uint4*>(w2l + ct * 16 * a.s2);
                wl.u = *reinterpret_cast<const uint4*>(w2l + ct * 16 * a.s2 + 64);
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    acc2[ct][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl.h, hh[p].h, acc2[ct][p], 0, 0, 0);
                    acc2[ct][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh.h, hl[p].h, acc2[ct][p], 0, 0, 0);
                    acc2[ct][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh.h, hh[p].h, acc2[ct][p], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            store_w(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }

        // epilogue: x 1/scale, + residual, split, 16-byte stores (lane: token m, channels 32*hh + 8*lq .. +7)
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m = m0 + p * 16 + l15;
            const bool ok = m < a.M;
#pragma unroll
            for (int hc = 0; hc < CK; ++hc) {
                const unsigned chunk = (unsigned)((hc * 4 + lq) * 32);
                const unsigned roff = ok ? (unsigned)(m * a.ldr * 4) + chunk : OOB;
                U4H8 rh, rl;
                rh.u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rr, roff, 0, 0));
                rl.u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rr, roff + 16u, 0, 0));
                U4H8 oh, ol;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f32x4 av = acc2[2 * hc + (j >> 2)][p];
                    f32x2 v = (f32x2){av[j & 3], av[(j & 3) + 1]} * a.os2;
                    v += (f32x2){(float)rh.e[j], (float)rh.e[j + 1]} + (f32x2){(float)rl.e[j], (float)rl.e[j + 1]};
                    const f16x2 h = __builtin_convertvector(v, f16x2);
                    const f32x2 d = v - __builtin_convertvector(h, f32x2);
                    const f16x2 l = __builtin_convertvector(d, f16x2);
                    oh.e[j] = h.x;
                    oh.e[j + 1] = h.y;
                    ol.e[j] = l.x;
                    ol.e[j + 1] = l.y;
                    ov_sum += v;
                    ov_max = fmaxf(ov_max, fmaxf(fabsf(v.x), fabsf(v.y)));
                }
                typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
                const unsigned yoff = ok ? (unsigned)(m * a.ldy * 4) + chunk : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, oh.u), ry, yoff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ol.u), ry, yoff + 16u, 0, 0);
            }
        }
    }
    float tx = ov_sum.x, ty = ov_sum.y;
    asm volatile("" : "+v"(tx), "+v"(ty));       // two scalars: no half-swapped v_pk_add_f32 for the horizontal sum (tools/isa_lint.py)
    const float ts = tx + ty;
    if (a.status && (!(ov_max <= VIP_H2_MAX) || !(fabsf(ts) <= 3.0e38f))) *a.status = VIP_H2_OVERFLOW;
}

// LDS row stride for `chunks` 16-byte chunks: == 32 (mod 64) bytes, conflict-free for the fragment read pattern under the ds_read_b128
// lane grouping (launch_pw in conv_igemm.hip)
int mlp_h2_stride(int chunks) {
    while ((chunks & 3) != 2) ++chunks;
    return chunks * 16;
}

template <int CK, int NW>
int launch_mlp_h2(MlpH2Args a, hipStream_t s) {
    constexpr int C = 32 * CK;
    a.s1 = mlp_h2_stride(C / 4);
    a.s2 = mlp_h2_stride(8);
    const size_t smem = 2 * ((size_t)32 * a.s1 + (size_t)C * a.s2);
    a.n_tiles = (a.M + 32 * NW - 1) / (32 * NW);
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_h2_kernel<CK, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    // workgroups per CU: what the LDS images allow (two waves per SIMD either way).  Two independent 4-wave workgroups drift apart in
    // phase - one in its MFMA runs while the other evaluates GELU - where one 8-wave workgroup is locked to a barrier per slice
    int per_cu = (int)((160 * 1024) / (smem + 1024));
    if (per_cu > 8 / NW * 2) per_cu = 8 / NW * 2;
    if (per_cu < 1) per_cu = 1;
    const int cap = n_cu * per_cu;
    const int grid = a.n_tiles < cap ? a.n_tiles : cap;
    hipLaunchKernelGGL((mlp_h2_kernel<CK, NW>), dim3(grid), dim3(64 * NW), smem, s, a);
    return vip_launch_status("vip_mlp_fused_h2");
}

}  // namespace

extern "C" int vip_mlp_fused_supported_h2(int M, int C, int hidden, int act) {
    if (act != VIP_ACT_GELU || hidden % 32 != 0 || hidden <= 0 || M < 8192) return 0;
    return C == 64 || C == 96 || C == 128;
}

extern "C" int vip_mlp_fused_h2(const void* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w1, const float* b1,
                                float out_scale1, const void* w2, const float* b2, float out_scale2, const void* residual, void* y, int M, int C,
                                int hidden, int ldx, int ldw1, int ldw2, int ldy, int ldr, int act, int* status, void* stream) {
    VIP_REQUIRE(x && w1 && w2 && y, VIP_ERR_BAD_ARG, "vip_mlp_fused_h2: null pointer");
    VIP_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), VIP_ERR_BAD_ARG,
                "vip_mlp_fused_h2: ln_gamma and ln_beta must both be given or both be NULL");
    VIP_REQUIRE(M > 0 && C > 0 && hidden > 0, VIP_ERR_BAD_ARG, "vip_mlp_fused_h2: non-positive dimension");
    VIP_REQUIRE(out_scale1 > 0.f && out_scale2 > 0.f, VIP_ERR_BAD_ARG, "vip_mlp_fused_h2: out_scale must be positive");
    VIP_REQUIRE(vip_mlp_fused_supported_h2(M, C, hidden, act), VIP_ERR_UNSUPPORTED,
                "vip_mlp_fused_h2: unsupported shape/activation (C=%d hidden=%d act=%d M=%d); use two vip_conv2d_nhwc_h2 calls", C, hidden, act, M);
    VIP_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldw1 % 16 == 0 && ldw2 % 16 == 0 && (!residual || ldr % 8 == 0), VIP_ERR_ALIGNMENT,
                "vip_mlp_fused_h2: activation leading dimensions must be multiples of 8 elements, weight rows of 16 halfs");
    VIP_REQUIRE(ldx >= C && ldy >= C && ldw1 >= 2 * C && ldw2 >= 2 * hidden && (!residual || ldr >= C), VIP_ERR_BAD_ARG,
                "vip_mlp_fused_h2: leading dimension smaller than the row extent");
    VIP_REQUIRE(4L * M * ldx < 0xFFFFFFE0L && 4L * M * ldy < 0xFFFFFFE0L && (!residual || 4L * M * ldr < 0xFFFFFFE0L), VIP_ERR_UNSUPPORTED,
                "vip_mlp_fused_h2: tensor exceeds the 4 GiB buffer-addressing range");
    MlpH2Args a;
    a.x = (const char*)x; a.w1 = (const char*)w1; a.b1 = b1; a.w2 = (const char*)w2; a.b2 = b2;
    a.res = (const char*)residual; a.y = (char*)y;
    a.ln_g = ln_gamma; a.ln_b = ln_beta; a.ln_eps = ln_eps;
    a.os1 = out_scale1; a.os2 = out_scale2;
    a.M = M; a.Hd = hidden; a.ldx = ldx; a.ldy = ldy; a.ldr = ldr; a.ldw1 = ldw1; a.ldw2 = ldw2;
    a.s1 = a.s2 = a.n_tiles = 0;
    a.status = status;
    // waves per workgroup, measured (profiles/r04_mlp_h2_ab.log): C = 64 / 128 prefer two 4-wave workgroups per CU (307 vs 346 us,
    // 239 vs 273 us), C = 96 one 8-wave workgroup (1 843 vs 1 985 us: half the weight traffic out of L2).  VIP_MLP_H2_WAVES overrides.
    static const int nw_env = getenv("VIP_MLP_H2_WAVES") ? atoi(getenv("VIP_MLP_H2_WAVES")) : 0;
    const int nw = nw_env ? nw_env : (C == 96 ? 8 : 4);
    hipStream_t s = (hipStream_t)stream;
    if (nw == 8) {
        if (C == 64) return launch_mlp_h2<2, 8>(a, s);
        if (C == 96) return launch_mlp_h2<3, 8>(a, s);
        return launch_mlp_h2<4, 8>(a, s);
    }
    if (C == 64) return launch_mlp_h2<2, 4>(a, s);
    if (C == 96) return launch_mlp_h2<3, 4>(a, s);
    return launch_mlp_h2<4, 4>(a, s);
}
