// The GCViT attention half of a block in one launch for the 14 x 14-window level (C = 256, 8 heads of 32; level 2 of GCViT-Tiny:
// 19 of its 31 blocks):
//
//     y = x + proj( window_attention( qkv( LayerNorm(x) ) ) )            (gcvit/layers/block.py:58-79, attention.py:52-83)
//
// Same plan as gcvit_block.hip (weights through LDS, window state in registers, weight rows / columns interleaved so that one MFMA's
// packed result is the next one's operand), re-cut for a window of 196 tokens = 13 MFMA tiles:
//   * ONE workgroup of 8 waves per window; wave w owns token tiles w and w + 8 (five waves own two tiles, three own one): their x^
//     fragments stay in registers (64 VGPRs) for all heads;
//   * per head (loop NOT unrolled): the head's q | k | v weight rows (96 x 256, 51 KB) and its relative-position table are staged in
//     LDS; every wave computes q, k, v for its own tokens - q stays in registers, k and v go to the workgroup's K / V image in the
//     padded order row' = 16 ty + tx of window_attn.hip - then runs the attention core of window_attn.hip<14> on its own query tiles;
//   * the normalised head output O_h (8 halfs per lane and tile) is parked in the rows of y this lane will write anyway (the same 16
//     bytes it later overwrites with the result: a private scratch that costs no memory) - 8 heads of accumulators or outputs do not
//     fit next to the core in 256 registers;
//   * after the last head: proj weight columns of two heads at a time through the same LDS buffer, y^T += Wp O^T for both tiles from
//     the parked fragments, + bias + x, 16-byte stores.
// A block with a global query takes q from q_global [B, 196, C] and computes k, v only.
#include "common.hpp"

namespace {

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct Gb14Args {
    const f16* x;
    const f16* qg;          // [B][196][C] or NULL
    const float* ln_g;
    const float* ln_b;
    float ln_eps;
    const f16* wqkv;        // [nq * C][ldwq], rows: q (if nq == 3), k, v; each [head][32]
    const float* bqkv;      // [nq * C] or NULL
    const f16* wproj;       // [C][ldwp]
    const float* bproj;     // [C] or NULL
    const float* table;     // [27 * 27][heads]
    f16* y;
    int B, Hp, Wp, nWy, nWx, ldwq, ldwp;
    long n_windows, x_bytes;
    float scale_log2e, inv_scale;
};

constexpr int G14_WS = 14, G14_C = 256, G14_HEADS = 8, G14_CK = 8, G14_N = G14_WS * G14_WS;
constexpr int G14_NT = 13;                                      // 16-token tiles of the dense order
constexpr int G14_RP = 224, G14_NKT = 14;                       // key rows of the padded order (14 rows of 16 slots)
constexpr int G14_TW = 48, G14_KSTEP = 48, G14_KCMAX = G14_KSTEP * G14_NKT, G14_NEGSZ = G14_KCMAX + 4, G14_TOFF = G14_NEGSZ + G14_KCMAX;
constexpr int G14_TROWS = 2 * G14_WS - 1, G14_TBF = G14_TOFF + G14_TROWS * G14_TW;   // window_attn.hip WinCfg<14, 16, 4, 4>
constexpr int G14_KVB = 64;
constexpr int G14_WROWB = 2 * G14_C + 32;                       // qkv weight row in LDS (stride = 32 mod 64)
constexpr int G14_PROWB = 2 * 64 + 32;                          // proj weight row of a head PAIR (64 columns)
constexpr int G14_W_OFF = 0;
constexpr int G14_W_BYTES = 96 * G14_WROWB;                     // >= 256 * G14_PROWB
constexpr int G14_KV_OFF = G14_W_OFF + G14_W_BYTES;
constexpr int G14_TB_OFF = G14_KV_OFF + 2 * G14_RP * G14_KVB;
constexpr int G14_SMEM = G14_TB_OFF + (G14_TBF * 4 + 15) / 16 * 16;
static_assert(256 * G14_PROWB <= G14_W_BYTES, "proj pair image fits the weight buffer");

__device__ __forceinline__ int g14_frag32(int j) {              // gcvit_block.hip gb_frag32
    const int t = (j >> 4) & 1, r = j & 15;
    return (r >> 2) * 8 + t * 4 + (r & 3);
}
__device__ __forceinline__ int g14_k_slot(int row, int ch) {
    const int q = (row >> 2) & 3;
    return ch ^ ((0x78 >> (q * 2)) & 3);
}

// LayerNorm of one 16-token tile held as fragments (mlp_fused.hip ln_fragments, C = 256)
__device__ __forceinline__ void g14_layernorm(U4H8 (&xf)[G14_CK], const float* __restrict__ gam, const float* __restrict__ bet, float eps,
                                              int g) {
    float v[G14_CK][8];
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < G14_CK; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[ks][j] = (float)xf[ks].e[j];
            sum += v[ks][j];
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / (float)G14_C;
    float sq = 0.f;
#pragma unroll
    for (int ks = 0; ks < G14_CK; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = v[ks][j] - mean;
            sq += d * d;
        }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq / (float)G14_C + eps);
#pragma unroll
    for (int ks = 0; ks < G14_CK; ++ks) {
        const float4 g0 = *reinterpret_cast<const float4*>(gam + ks * 32 + g * 8), g1 = *reinterpret_cast<const float4*>(gam + ks * 32 + g * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(bet + ks * 32 + g * 8), b1 = *reinterpret_cast<const float4*>(bet + ks * 32 + g * 8 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[ks].e[j] = (f16)((v[ks][j] - mean) * rstd * gg[j] + bb[j]);
    }
}

// One 16-query tile against the workgroup's K / V image (window_attn.hip win_query_tile<14, 16, 4, 4>: S^T = K Q^T with the bias
// table as the C operand, softmax over the 224 key slots, O^T = V^T P^T), normalised and packed for the proj MFMA: k-slot (g, j) =
// head channel 4g + j (j < 4) / 16 + 4g + (j - 4).  (qy, qx): the lane's query in the window, clamped for the padding lanes.
__device__ __forceinline__ U4H8 g14_attn_tile(const U4H8& qfrag, const char* k_lds, const char* v_lds, const float* tb, int qy, int qx,
                                              int l15, int g, float sc) {
    constexpr int NKT = G14_NKT;
    const int lane_term = 4 * g;
    const int tr_q = l15 >> 2, tr_p = l15 & 3;
    const float* tbase = tb + (G14_TOFF + qy * G14_TW + qx + (G14_WS - 1) * (G14_TW + 1) - lane_term - G14_KCMAX);
    f32x4 acc[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int mask = 0;                       // lane groups whose key slot (t, r) is padding (compile-time)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int kp = 16 * t + 4 * gg + r;
                if (((kp & 15) >= G14_WS) || ((kp >> 4) >= G14_WS)) mask |= 1 << gg;
            }
            if (mask == 15) {
                acc[t][r] = -1.0e30f;
            } else {
                const int imm = G14_KCMAX - (G14_KSTEP * t + r);
                const float* bp = (mask == 0) ? tbase : (((mask >> g) & 1) ? tb : tbase);
                acc[t][r] = bp[imm];
            }
        }
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        const int row = t * 16 + l15;
        U4H8 kf;
        kf.u = *reinterpret_cast<const uint4*>(k_lds + row * G14_KVB + g14_k_slot(row, g) * 16);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf.h, qfrag.h, acc[t], 0, 0, 0);
    }
    float m = -1.0e30f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        m = fmaxf(fmaxf(m, acc[t][0]), acc[t][1]);
        m = fmaxf(fmaxf(m, acc[t][2]), acc[t][3]);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const f32x2 nm = {-m * sc, -m * sc};
    f32x2 ls2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const f32x2 e = (f32x2){acc[t][r], acc[t][r + 1]} * sc + nm;
            const f32x2 p = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
            acc[t][r] = p.x;
            acc[t][r + 1] = p.y;
            ls2 += p;
        }
    float lsum = ls2.x + ls2.y;
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);

    f32x4 o[2];
    o[0] = o[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NKT / 2; ++s) {
        U4H8 pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pf.e[j] = (f16)acc[2 * s][j];
            pf.e[4 + j] = (f16)acc[2 * s + 1][j];
        }
        union {
            fp16x4_t t[2];
            f16x8 v;
        } vf[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int row = 32 * s + 16 * hh + 4 * g + tr_q;
                const int half = dt ^ ((row >> 2) & 1);
                vf[dt].t[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) fp16x4_t*)(v_lds + row * G14_KVB + half * 32 + tr_p * 8));
            }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dt].v, pf.h, o[dt], 0, 0, 0);
    }
    const float inv = 1.f / lsum;
    U4H8 of;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        of.e[r] = (f16)(o[0][r] * inv);
        of.e[4 + r] = (f16)(o[1][r] * inv);
    }
    return of;
}

template <bool GLOBALQ>
__global__ __launch_bounds__(512, 2) void gcvit_attn_block14_kernel(Gb14Args a) {
    constexpr int C = G14_C, CK = G14_CK, NTHR = 512, NCT = C / 16;
    constexpr int NQ = GLOBALQ ? 2 : 3;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wbuf = smem + G14_W_OFF;
    char* k_lds = smem + G14_KV_OFF;
    char* v_lds = k_lds + G14_RP * G14_KVB;
    float* tb = reinterpret_cast<float*>(smem + G14_TB_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;

    // the padding slots of the K / V image are never written by a token: finite once and for all
    for (int i = tid; i < 2 * G14_RP * G14_KVB / 16; i += NTHR) *reinterpret_cast<uint4*>(k_lds + i * 16) = make_uint4(0, 0, 0, 0);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbq =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.bqkv, 0, a.bqkv ? (unsigned)(NQ * C * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbp = __builtin_amdgcn_make_buffer_rsrc((void*)a.bproj, 0, a.bproj ? (unsigned)(C * 4) : 0u, 0x00020000);
    const char* wql = wbuf + l15 * G14_WROWB + g * 16;
    const char* wpl = wbuf + l15 * G14_PROWB + g * 16;
    const float sc = a.scale_log2e;
    const int wpi = a.nWy * a.nWx;

    // this lane's tokens: slot s -> tile wave + 8 s, dense index n = 16 tile + l15 = 14 ty + tx
    bool tile_ok[2];
    int qyv[2], qxv[2], krow[2];
    bool tok_ok[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int tile = wave + 8 * s;
        tile_ok[s] = tile < G14_NT;                          // wave-uniform
        const int n = tile * 16 + l15;
        tok_ok[s] = tile_ok[s] && n < G14_N;
        const int nc = n < G14_N ? n : G14_N - 1;
        qyv[s] = nc / G14_WS;
        qxv[s] = nc - qyv[s] * G14_WS;
        krow[s] = qyv[s] * 16 + qxv[s];
    }

    for (long w = blockIdx.x; w < a.n_windows; w += gridDim.x) {          // workgroup-uniform
        const int b = (int)(w / wpi);
        const int wrem = (int)(w - (long)b * wpi);
        const int wy = wrem / a.nWx, wx = wrem - wy * a.nWx;
        unsigned xoff[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const long pix = ((long)b * a.Hp + wy * G14_WS + qyv[s]) * a.Wp + wx * G14_WS + qxv[s];
            xoff[s] = tok_ok[s] ? (unsigned)(pix * C * 2) : OOB;
        }
        U4H8 xf[CK][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ks = 0; ks < CK; ++ks)
                xf[ks][s].u = __builtin_bit_cast(
                    uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[s] == OOB ? OOB : xoff[s] + (ks * 32 + g * 8) * 2, 0, 0));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            U4H8 col[CK];
#pragma unroll
            for (int ks = 0; ks < CK; ++ks) col[ks] = xf[ks][s];
            g14_layernorm(col, a.ln_g, a.ln_b, a.ln_eps, g);
#pragma unroll
            for (int ks = 0; ks < CK; ++ks) xf[ks][s] = col[ks];
        }

#pragma unroll 1
        for (int head = 0; head < G14_HEADS; ++head) {
            // ---- stage this head's q | k | v weight rows (fragment order) and its bias table; the previous head's readers are past
            // the barrier that closed its attention pass ----
            for (int i = tid; i < NQ * 32 * (C / 8); i += NTHR) {
                const int j = i >> 5, c = i & 31;                                 // LDS row (part * 32 + jj), 16-byte chunk
                const int row = (j >> 5) * C + head * 32 + g14_frag32(j & 31);
                *reinterpret_cast<uint4*>(wbuf + j * G14_WROWB + c * 16) = *reinterpret_cast<const uint4*>(a.wqkv + (long)row * a.ldwq + c * 8);
            }
            for (int i = tid; i < G14_TBF; i += NTHR) {
                const int e = i - G14_TOFF;
                const int ry_ = e / G14_TW, rx_ = e - ry_ * G14_TW;
                const bool in_tab = (i >= G14_TOFF) & (rx_ < G14_TROWS);
                const float t = a.table[in_tab ? (ry_ * G14_TROWS + rx_) * G14_HEADS + head : 0];
                tb[i] = in_tab ? t * a.inv_scale : (i < G14_TOFF ? -1.0e30f : 0.f);
            }
            __syncthreads();

            // ---- [q | k | v]^T = W x^T + b for this wave's tokens ----
            U4H8 qf[2];
#pragma unroll
            for (int part = 0; part < NQ; ++part) {
                U4H8 wf[2][CK];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ks = 0; ks < CK; ++ks)
                        wf[t][ks].u = *reinterpret_cast<const uint4*>(wql + (part * 32 + t * 16) * G14_WROWB + ks * 64);
                f32x4 bv[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    bv[t] = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbq, (unsigned)((part * C + head * 32 + g * 8 + t * 4) * 4), 0, 0));
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (!tile_ok[s]) continue;                                    // wave-uniform
                    f32x4 acc[2] = {bv[0], bv[1]};
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int ks = 0; ks < CK; ++ks)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][ks].h, xf[ks][s].h, acc[t], 0, 0, 0);
                    U4H8 pk;                                                      // head channels 8 g .. 8 g + 7 of the lane's token
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        pk.e[i] = (f16)acc[0][i];
                        pk.e[4 + i] = (f16)acc[1][i];
                    }
                    const int row = krow[s];
                    if (!GLOBALQ && part == 0) qf[s] = pk;
                    else if (part == NQ - 2) {
                        if (tok_ok[s]) *reinterpret_cast<uint4*>(k_lds + row * G14_KVB + g14_k_slot(row, g) * 16) = pk.u;
                    } else {
                        if (tok_ok[s]) *reinterpret_cast<uint4*>(v_lds + row * G14_KVB + ((g ^ (((row >> 2) & 1) << 1)) * 16)) = pk.u;
                    }
                }
            }
            if constexpr (GLOBALQ) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int n = qyv[s] * G14_WS + qxv[s];
                    const uint4 v = *reinterpret_cast<const uint4*>(a.qg + ((long)b * G14_N + n) * C + head * 32 + g * 8);
                    qf[s].u = tok_ok[s] ? v : make_uint4(0, 0, 0, 0);
                }
            }
            __syncthreads();

            // ---- attention on this wave's query tiles; the head's output goes to the lane's own 16 bytes of y (scratch) ----
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (!tile_ok[s]) continue;
                const U4H8 of = g14_attn_tile(qf[s], k_lds, v_lds, tb, qyv[s], qxv[s], l15, g, sc);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, of.u),
                                                       ry, xoff[s] == OOB ? OOB : xoff[s] + (head * 32 + g * 8) * 2, 0, 0);
            }
            __syncthreads();
        }

        // ---- proj: y^T = Wp O^T + b over head pairs (64 weight columns at a time through the weight buffer) ----
        f32x4 yacc[NCT][2];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const f32x4 bv = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbp, (unsigned)(((ct >> 1) * 32 + g * 8 + (ct & 1) * 4) * 4), 0, 0));
            yacc[ct][0] = yacc[ct][1] = bv;
        }
#pragma unroll 1
        for (int hp = 0; hp < G14_HEADS / 2; ++hp) {
            for (int i = tid; i < C * 8; i += NTHR) {
                const int j = i >> 3, c = i & 7;                                  // LDS row, chunk = head-in-pair * 4 + lane group
                const int ch = (j & ~31) + g14_frag32(j & 31);
                const f16* src = a.wproj + (long)ch * a.ldwp + (2 * hp + (c >> 2)) * 32 + 4 * (c & 3);
                const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 16);
                *reinterpret_cast<uint4*>(wbuf + j * G14_PROWB + c * 16) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
            // the parked head outputs of this lane (its own stores, completed before the barriers in between; sc0 sc1: past the L1)
            U4H8 of[2][2];
#pragma unroll
            for (int hl = 0; hl < 2; ++hl)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    of[hl][s].u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                ry, xoff[s] == OOB ? OOB : xoff[s] + ((2 * hp + hl) * 32 + g * 8) * 2, 0, 17));
            __syncthreads();
#pragma unroll
            for (int hl = 0; hl < 2; ++hl)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    U4H8 pw;
                    pw.u = *reinterpret_cast<const uint4*>(wpl + ct * 16 * G14_PROWB + hl * 64);
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        if (tile_ok[s]) yacc[ct][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pw.h, of[hl][s].h, yacc[ct][s], 0, 0, 0);
                }
            __syncthreads();
        }

        // ---- epilogue: + x, fp16, 16-byte stores (lane: token, channels 32 hh + 8 g .. + 7) ----
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int hh = 0; hh < CK; ++hh) {
                const unsigned off = xoff[s] == OOB ? OOB : xoff[s] + (hh * 32 + g * 8) * 2;
                U4H8 r;
                r.u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
                U4H8 o;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f32x4 av = yacc[2 * hh + (j >> 2)][s];
                    const f32x2 v = (f32x2){av[j & 3], av[(j & 3) + 1]} + (f32x2){(float)r.e[j], (float)r.e[j + 1]};
                    o.e[j] = (f16)v.x;
                    o.e[j + 1] = (f16)v.y;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o.u),
                                                       ry, off, 0, 0);
            }
    }
}

}  // namespace

// the 14 x 14-window configuration of vip_gcvit_attn_block_f16 (gcvit_block.hip): C = 256, 8 heads
int vip_gcvit_attn_block14(const void* x, const void* q_global, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* wqkv,
                           int ldwq, const float* bqkv, const void* wproj, int ldwp, const float* bproj, const float* table, void* y,
                           int B, int Hp, int Wp, float scale, hipStream_t s) {
    Gb14Args a;
    a.x = (const f16*)x; a.qg = (const f16*)q_global; a.ln_g = ln_gamma; a.ln_b = ln_beta; a.ln_eps = ln_eps;
    a.wqkv = (const f16*)wqkv; a.bqkv = bqkv; a.wproj = (const f16*)wproj; a.bproj = bproj; a.table = table; a.y = (f16*)y;
    a.B = B; a.Hp = Hp; a.Wp = Wp; a.nWy = Hp / G14_WS; a.nWx = Wp / G14_WS; a.ldwq = ldwq; a.ldwp = ldwp;
    a.n_windows = (long)B * a.nWy * a.nWx;
    a.x_bytes = 2L * B * Hp * Wp * G14_C;
    a.scale_log2e = scale * 1.44269504088896f;
    a.inv_scale = 1.f / scale;
    long wgs = a.n_windows < 256 ? a.n_windows : 256;            // one 89 KB, 8-wave workgroup per CU
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gcvit_attn_block14_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, G14_SMEM);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gcvit_attn_block14_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, G14_SMEM);
        attr_set = true;
    }
    if (q_global) hipLaunchKernelGGL(gcvit_attn_block14_kernel<true>, dim3((unsigned)wgs), dim3(512), G14_SMEM, s, a);
    else hipLaunchKernelGGL(gcvit_attn_block14_kernel<false>, dim3((unsigned)wgs), dim3(512), G14_SMEM, s, a);
    return vip_launch_status("vip_gcvit_attn_block_f16(ws 14)");
}
