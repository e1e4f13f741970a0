// Fused two-layer MLP   y = W2 . act(W1 . x + b1) + b2 (+ residual)   for narrow token widths (C = 64 / 96 / 128 / 192).
//
// Replaces the Dense -> GELU -> Dense (+Add) tail of tfimm's ConvNeXtBlock (convnext.py:200-229, layer-scale gamma
// folded into W2/b2 by the host) and of the GCViT / ViT MLP (gcvit/layers/feature.py:20-22,
// tfimm/layers/transformers.py:192-205) where the hidden tensor [M, 4C] would otherwise be written to HBM by one GEMM
// and read back by the next: at ConvNeXt stage 0 (M = 2.5 M tokens, C = 96) that round trip is 3.9 GB per block
// against 1.4 GB for x + residual + y.
//
// Structure (same operand routing as pw_gemm_kernel in conv_igemm.hip):
//   * BOTH weight matrices live in LDS for the whole kernel (fragment-ordered rows, row stride 32 mod 64 bytes:
//     conflict-free ds_read_b128); 159 KB for C = 96 / hidden 384, so one workgroup per CU
//     (16 waves);
//   * a wave owns 32 tokens: their activation fragments (the MFMA B operand: lane = token, 8 consecutive channels =
//     one 16-byte run of the row) are loaded global -> VGPR once per tile;
//   * the hidden layer is produced 32 channels at a time: H^T = W1 X^T (bias as the MFMA C operand), activation in
//     packed fp32, and - because the weight rows are interleaved so that a lane ends up holding 8 CONSECUTIVE hidden
//     channels of its token - the fp16-packed result already IS the B operand of the second GEMM's k-step; it never
//     leaves the register file;
//   * y accumulates in registers over the 12 hidden slices and goes out with the residual in 16-byte stores.
#include "common.hpp"

namespace {

struct MlpArgs {
    const f16* x;
    const f16* w1;
    const float* b1;
    const f16* w2;
    const float* b2;
    const f16* res;
    f16* y;
    const float* ln_g;   // optional LayerNorm over C applied to x first (NULL: none)
    const float* ln_b;
    float ln_eps;
    int M, Hd;
    int ldx, ldy, ldr, ldw1, ldw2;
    long x_bytes, y_bytes, res_bytes;
    int s1, s2;      // LDS row strides (bytes) of W1 [Hd rows] and W2 [C rows]
    int n_tiles;     // tiles of 512 tokens
};

// LDS row j of a 32-row group holds channel (j>>4)*4 + ((j&15)>>2)*8 + (j&3): MFMA tiles 2h, 2h+1 then give a lane
// the channels 8*lq .. 8*lq+7 of the group
__device__ __forceinline__ int frag_channel(int j) {
    const int t = (j >> 4) & 1, r = j & 15;
    return (j & ~31) + (r >> 2) * 8 + t * 4 + (r & 3);
}


// Optional LayerNorm prologue on the activation fragments of one 16-token tile: lane (l15 = token, lq) holds channels
// 32 ks + 8 lq + 0..7 for ks < CK, so a token's C channels sit in 4 lanes (lq = 0..3): in-lane sums + two cross-row
// exchanges.  Two-pass mean / variance in fp32 and the same expression as layernorm_kernel (pointwise.hip); the
// normalised row is rounded to fp16 exactly where the separate LayerNorm launch rounded it.
template <int CK>
__device__ __forceinline__ void ln_fragments(U4H8 (&xf)[CK], const float* __restrict__ g, const float* __restrict__ b,
                                             float eps, int lq) {
    constexpr int C = 32 * CK;
    float v[CK][8];
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < CK; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[ks][j] = (float)xf[ks].e[j];
            sum += v[ks][j];
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
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[ks].e[j] = (f16)((v[ks][j] - mean) * rstd * gg[j] + bb[j]);
    }
}

template <int CK, int ACT, int PT>
__global__ __launch_bounds__(2048 / PT, 1) void mlp_fused_kernel(MlpArgs a) {
    constexpr int C = 32 * CK, NCT = C / 16, NTHR = 2048 / PT;   // 512 tokens per workgroup: 8 waves x 64 or 16 x 32
    constexpr unsigned OOB = 0xFFFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w1s = smem;
    char* w2s = smem + a.Hd * a.s1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;

    {   // stage both weight matrices and b1 (once per workgroup)
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1, 0, (unsigned)(2L * a.Hd * a.ldw1), 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, (unsigned)(2L * C * a.ldw2), 0x00020000);
        constexpr int cpr1 = CK * 4;
        for (int i = tid; i < a.Hd * cpr1; i += NTHR) {
            const int j = i / cpr1, c = i - j * cpr1;
            const uint4 v = __builtin_bit_cast(
                uint4, __builtin_amdgcn_raw_buffer_load_b128(r1, (unsigned)((frag_channel(j) * a.ldw1 + c * 8) * 2), 0, 0));
            *reinterpret_cast<uint4*>(w1s + j * a.s1 + c * 16) = v;
        }
        const int cpr2 = a.Hd >> 3;
        for (int i = tid; i < C * cpr2; i += NTHR) {
            const int j = i / cpr2, c = i - j * cpr2;
            const uint4 v = __builtin_bit_cast(
                uint4, __builtin_amdgcn_raw_buffer_load_b128(r2, (unsigned)((frag_channel(j) * a.ldw2 + c * 8) * 2), 0, 0));
            *reinterpret_cast<uint4*>(w2s + j * a.s2 + c * 16) = v;
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (unsigned)a.res_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.b1, 0, a.b1 ? (unsigned)(a.Hd * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.b2, 0, a.b2 ? (unsigned)(C * 4) : 0u, 0x00020000);
    const char* w1l = w1s + l15 * a.s1 + lq * 16;
    const char* w2l = w2s + l15 * a.s2 + lq * 16;
    const int nq = a.Hd >> 5;

    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int m0 = tile * 512 + wave * (16 * PT);
        U4H8 xf[CK][PT];
#pragma unroll
        for (int ks = 0; ks < CK; ++ks)
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int m = m0 + p * 16 + l15;
                xf[ks][p].u = __builtin_bit_cast(
                    uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, m < a.M ? (unsigned)((m * a.ldx + ks * 32 + lq * 8) * 2) : OOB, 0, 0));
            }
        if (a.ln_g) {   // workgroup-uniform
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                U4H8 col[CK];
#pragma unroll
                for (int ks = 0; ks < CK; ++ks) col[ks] = xf[ks][p];
                ln_fragments<CK>(col, a.ln_g, a.ln_b, a.ln_eps, lq);
#pragma unroll
                for (int ks = 0; ks < CK; ++ks) xf[ks][p] = col[ks];
            }
        }
        // y accumulators start at b2 (lane: channels 32*hh + 8*lq + 4*t + 0..3 for tile 2*hh + t)
        f32x4 acc2[NCT][PT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const f32x4 bv = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb2, (unsigned)(((ct >> 1) * 32 + lq * 8 + (ct & 1) * 4) * 4), 0, 0));
#pragma unroll
            for (int p = 0; p < PT; ++p) acc2[ct][p] = bv;
        }

        // hidden slice q: H^T[32 x 64 tokens] = W1[32q.., :] . X^T + b1
        auto gemm1 = [&](int q, f32x4 (&acc1)[2][PT]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // b1 from global (L1-resident, 4 distinct addresses per wave): LDS is full to the last KB
                const f32x4 bv = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb1, (unsigned)((q * 32 + lq * 8 + t * 4) * 4), 0, 0));
#pragma unroll
                for (int p = 0; p < PT; ++p) acc1[t][p] = bv;
            }
#pragma unroll
            for (int ks = 0; ks < CK; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    U4H8 wf;
                    wf.u = *reinterpret_cast<const uint4*>(w1l + (q * 32 + t * 16) * a.s1 + ks * 64);
#pragma unroll
                    for (int p = 0; p < PT; ++p)
                        acc1[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf.h, xf[ks][p].h, acc1[t][p], 0, 0, 0);
                }
        };
        // activation; the packed result is the B operand (k = 8*lq + j <-> hidden channel 32q + 8*lq + j)
        auto act_pack = [&](const f32x4 (&acc1)[2][PT], U4H8 (&hf)[PT]) {
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const f32x2 v = vip_act2<ACT>((f32x2){acc1[t][p][j], acc1[t][p][j + 1]});
                        hf[p].e[t * 4 + j] = (f16)v.x;
                        hf[p].e[t * 4 + j + 1] = (f16)v.y;
                    }
        };
        // y^T += W2[:, 32q..32q+31] . H
        auto gemm2 = [&](int q, const U4H8 (&hf)[PT]) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                U4H8 wf;
                wf.u = *reinterpret_cast<const uint4*>(w2l + ct * 16 * a.s2 + q * 64);
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    acc2[ct][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf.h, hf[p].h, acc2[ct][p], 0, 0, 0);
            }
        };
        // (Issuing slice q+1's first-layer MFMAs before slice q's activation - a two-slice software pipeline - was
        // measured 6 % SLOWER: +42 VGPRs and no overlap gained; the second resident wave already fills the gaps.)
        U4H8 hf[PT];
#pragma unroll 1
        for (int q = 0; q < nq; ++q) {
            f32x4 h0[2][PT];
            gemm1(q, h0);
            act_pack(h0, hf);
            gemm2(q, hf);
        }

        // epilogue: + residual, fp16, 16-byte stores (lane: token m, channels 32*hh + 8*lq .. +7)
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m = m0 + p * 16 + l15;
            const bool ok = m < a.M;
#pragma unroll
            for (int hh = 0; hh < CK; ++hh) {
                const int n = hh * 32 + lq * 8;
                U4H8 r;
                r.u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                    rr, ok ? (unsigned)((m * a.ldr + n) * 2) : OOB, 0, 0));
                U4H8 o;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f32x4 av = acc2[2 * hh + (j >> 2)][p];
                    const f32x2 v = (f32x2){av[j & 3], av[(j & 3) + 1]} + (f32x2){(float)r.e[j], (float)r.e[j + 1]};
                    o.e[j] = (f16)v.x;
                    o.e[j + 1] = (f16)v.y;
                }
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o.u), ry,
                    ok ? (unsigned)((m * a.ldy + n) * 2) : OOB, 0, 0);
            }
        }
    }
}

// LDS row stride for `chunks` 16-byte chunks: == 32 (mod 64) bytes, conflict-free for the fragment read pattern under
// the ds_read_b128 lane grouping (see launch_pw in conv_igemm.hip)
static int mlp_stride(int chunks) {
    while ((chunks & 3) != 2) ++chunks;
    return chunks * 16;
}

template <int CK>
int launch_mlp(MlpArgs a, hipStream_t s) {
    constexpr int C = 32 * CK;
    a.s1 = mlp_stride(CK * 4);
    a.s2 = mlp_stride(a.Hd >> 3);
    const size_t smem = (size_t)a.Hd * a.s1 + (size_t)C * a.s2;
    if (smem > 160 * 1024) return 1;
    a.n_tiles = (a.M + 511) / 512;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_kernel<CK, VIP_ACT_GELU, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;
    // 16 waves x 32 tokens (<= 128 VGPRs, four waves per SIMD) measured 2-8 % ahead of 8 waves x 64 tokens
    hipLaunchKernelGGL((mlp_fused_kernel<CK, VIP_ACT_GELU, 2>), dim3(grid), dim3(1024), smem, s, a);
    return vip_launch_status("vip_mlp_fused_f16");
}


// ---- C = 128 / 192: the two weight matrices (4C x C and C x 4C) no longer fit in LDS together, so they are STREAMED:
// per 32-channel hidden slice the workgroup stages W1[32q.., :] and W2[:, 32q..] (24 KB at C = 192) into a
// double-buffered LDS image (global -> VGPR one slice ahead, ds_write after the slice's math, one barrier per slice)
// while x (32 tokens per wave, 8 waves) and the y accumulators stay in registers for all 4C/32 slices.  The weights
// come from L2 (2.3 KB per token at C = 192 against 1.15 KB of HBM traffic); the hidden tensor never exists in memory.
template <int CK, int ACT>
__global__ __launch_bounds__(512, 1) void mlp_stream_kernel(MlpArgs a) {
    constexpr int C = 32 * CK, PT = 2, NCT = C / 16, NTHR = 512;
    constexpr int W_IT = C / 64;                    // 16-byte weight chunks staged per thread per slice (8C / 512)
    constexpr unsigned OOB = 0xFFFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int stage_bytes = 32 * a.s1 + C * a.s2;   // W1 slice [32][s1] then W2 slice [C][s2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (unsigned)a.res_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.b1, 0, a.b1 ? (unsigned)(a.Hd * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.b2, 0, a.b2 ? (unsigned)(C * 4) : 0u, 0x00020000);

    // staging plan of this thread: chunk idx = tid + 512 i; the first 4C chunks are the W1 slice, the rest the W2 slice.
    // Plain pointers (one load instruction per chunk whichever matrix it comes from: a `is_w1 ? load(w1) : load(w2)`
    // would be two predicated loads with a wait each)
    const char* w_ptr[W_IT];   // source of slice 0
    unsigned w_step[W_IT];     // bytes between consecutive slices
    int w_dst[W_IT];           // byte offset inside a stage
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int idx = tid + NTHR * i;
        if (idx < 4 * C) {
            const int row = idx / (C / 8), c = idx - row * (C / 8);
            w_ptr[i] = reinterpret_cast<const char*>(a.w1) + ((long)frag_channel(row) * a.ldw1 + c * 8) * 2;
            w_step[i] = (unsigned)(32 * a.ldw1 * 2);
            w_dst[i] = row * a.s1 + c * 16;
        } else {
            const int j = idx - 4 * C, row = j >> 2, c = j & 3;
            w_ptr[i] = reinterpret_cast<const char*>(a.w2) + ((long)frag_channel(row) * a.ldw2 + c * 8) * 2;
            w_step[i] = 64u;
            w_dst[i] = 32 * a.s1 + row * a.s2 + c * 16;
        }
    }
    uint4 wst[W_IT];
    auto load_w = [&](int q) {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) wst[i] = *reinterpret_cast<const uint4*>(w_ptr[i] + (size_t)q * w_step[i]);
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) *reinterpret_cast<uint4*>(smem + buf * stage_bytes + w_dst[i]) = wst[i];
    };

    const int nq = a.Hd >> 5;
    load_w(0);
    store_w(0);
    __syncthreads();
    int buf = 0;

    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int m0 = tile * (NTHR / 64 * 16 * PT) + wave * (16 * PT);
        U4H8 xf[CK][PT];
#pragma unroll
        for (int ks = 0; ks < CK; ++ks)
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int m = m0 + p * 16 + l15;
                xf[ks][p].u = __builtin_bit_cast(
                    uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, m < a.M ? (unsigned)((m * a.ldx + ks * 32 + lq * 8) * 2) : OOB, 0, 0));
            }
        if (a.ln_g) {   // workgroup-uniform
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                U4H8 col[CK];
#pragma unroll
                for (int ks = 0; ks < CK; ++ks) col[ks] = xf[ks][p];
                ln_fragments<CK>(col, a.ln_g, a.ln_b, a.ln_eps, lq);
#pragma unroll
                for (int ks = 0; ks < CK; ++ks) xf[ks][p] = col[ks];
            }
        }
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
                    U4H8 wf;
                    wf.u = *reinterpret_cast<const uint4*>(w1l + t * 16 * a.s1 + ks * 64);
#pragma unroll
                    for (int p = 0; p < PT; ++p)
                        acc1[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf.h, xf[ks][p].h, acc1[t][p], 0, 0, 0);
                }
            U4H8 hf[PT];
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const f32x2 v = vip_act2<ACT>((f32x2){acc1[t][p][j], acc1[t][p][j + 1]});
                        hf[p].e[t * 4 + j] = (f16)v.x;
                        hf[p].e[t * 4 + j + 1] = (f16)v.y;
                    }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                U4H8 wf;
                wf.u = *reinterpret_cast<const uint4*>(w2l + ct * 16 * a.s2);
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    acc2[ct][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf.h, hf[p].h, acc2[ct][p], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            store_w(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }

#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m = m0 + p * 16 + l15;
            const bool ok = m < a.M;
#pragma unroll
            for (int hh = 0; hh < CK; ++hh) {
                const int n = hh * 32 + lq * 8;
                U4H8 r;
                r.u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                    rr, ok ? (unsigned)((m * a.ldr + n) * 2) : OOB, 0, 0));
                U4H8 o;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f32x4 av = acc2[2 * hh + (j >> 2)][p];
                    const f32x2 v = (f32x2){av[j & 3], av[(j & 3) + 1]} + (f32x2){(float)r.e[j], (float)r.e[j + 1]};
                    o.e[j] = (f16)v.x;
                    o.e[j + 1] = (f16)v.y;
                }
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o.u), ry,
                    ok ? (unsigned)((m * a.ldy + n) * 2) : OOB, 0, 0);
            }
        }
    }
}

template <int CK>
int launch_mlp_stream(MlpArgs a, hipStream_t s) {
    constexpr int C = 32 * CK;
    a.s1 = mlp_stride(C / 8);
    a.s2 = mlp_stride(4);
    const size_t smem = 2 * ((size_t)32 * a.s1 + (size_t)C * a.s2);
    a.n_tiles = (a.M + 255) / 256;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_stream_kernel<CK, VIP_ACT_GELU>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;
    hipLaunchKernelGGL((mlp_stream_kernel<CK, VIP_ACT_GELU>), dim3(grid), dim3(512), smem, s, a);
    return vip_launch_status("vip_mlp_fused_f16(stream)");
}

}  // namespace

extern "C" int vip_mlp_fused_supported(int M, int C, int hidden, int act) {
    if (act != VIP_ACT_GELU || hidden % 32 != 0 || hidden <= 0 || M < 8192) return 0;
    if (C == 192) return 1;   // streamed weights (C = 128 and 256 are wired up too but measured slower than two GEMMs)
    if (C != 64 && C != 96) return 0;
    const long s1 = mlp_stride(C / 8), s2 = mlp_stride(hidden >> 3);      // LDS-resident weights
    return (long)hidden * s1 + (long)C * s2 <= 160 * 1024;
}

extern "C" int vip_mlp_fused_f16(const void* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w1,
                                 const float* b1, const void* w2, const float* b2, const void* residual, void* y, int M,
                                 int C, int hidden, int ldx, int ldw1, int ldw2, int ldy, int ldr, int act, void* stream) {
    VIP_REQUIRE(x && w1 && w2 && y, VIP_ERR_BAD_ARG, "vip_mlp_fused_f16: null pointer");
    VIP_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), VIP_ERR_BAD_ARG,
                "vip_mlp_fused_f16: ln_gamma and ln_beta must both be given or both be NULL");
    VIP_REQUIRE(M > 0 && C > 0 && hidden > 0, VIP_ERR_BAD_ARG, "vip_mlp_fused_f16: non-positive dimension");
    VIP_REQUIRE(vip_mlp_fused_supported(M, C, hidden, act), VIP_ERR_UNSUPPORTED,
                "vip_mlp_fused_f16: unsupported shape/activation (C=%d hidden=%d act=%d M=%d); use two vip_gemm_bias_act_f16 calls",
                C, hidden, act, M);
    VIP_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldw1 % 8 == 0 && ldw2 % 8 == 0 && (!residual || ldr % 8 == 0),
                VIP_ERR_ALIGNMENT, "vip_mlp_fused_f16: leading dimensions must be multiples of 8 halfs");
    VIP_REQUIRE(ldx >= C && ldy >= C && ldw1 >= C && ldw2 >= hidden && (!residual || ldr >= C), VIP_ERR_BAD_ARG,
                "vip_mlp_fused_f16: leading dimension smaller than the row extent");
    MlpArgs a;
    a.x = (const f16*)x; a.w1 = (const f16*)w1; a.b1 = b1; a.w2 = (const f16*)w2; a.b2 = b2;
    a.res = (const f16*)residual; a.y = (f16*)y;
    a.ln_g = ln_gamma; a.ln_b = ln_beta; a.ln_eps = ln_eps;
    a.M = M; a.Hd = hidden; a.ldx = ldx; a.ldy = ldy; a.ldr = ldr; a.ldw1 = ldw1; a.ldw2 = ldw2;
    a.x_bytes = 2L * M * ldx; a.y_bytes = 2L * M * ldy; a.res_bytes = 2L * M * ldr;
    VIP_REQUIRE(a.x_bytes < 0xFFFFFFF0L && a.y_bytes < 0xFFFFFFF0L && a.res_bytes < 0xFFFFFFF0L, VIP_ERR_UNSUPPORTED,
                "vip_mlp_fused_f16: tensor exceeds the 4 GiB buffer-addressing range");
    a.s1 = a.s2 = a.n_tiles = 0;
    int st = 1;
    if (C == 64) st = launch_mlp<2>(a, (hipStream_t)stream);
    else if (C == 96) st = launch_mlp<3>(a, (hipStream_t)stream);
    else if (C == 128) st = launch_mlp_stream<4>(a, (hipStream_t)stream);
    else if (C == 192) st = launch_mlp_stream<6>(a, (hipStream_t)stream);
    else if (C == 256) st = launch_mlp_stream<8>(a, (hipStream_t)stream);
    VIP_REQUIRE(st != 1, VIP_ERR_UNSUPPORTED, "vip_mlp_fused_f16: weights do not fit in LDS");
    return st;
}
