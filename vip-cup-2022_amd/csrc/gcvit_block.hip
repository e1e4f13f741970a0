// GCViT attention half of a block in ONE launch, for the first two levels (7 x 7 windows; C = 64 with 2 heads of 32, C = 128 with 4):
//
//     y = x + proj( window_attention( qkv( LayerNorm(x) ) ) )            (gcvit/layers/block.py:58-79, attention.py:52-83)
//
// - the fused variant SURVEY.md section 8(d) defines for the north-star path: per window FLOPs 8 N C^2 + 4 N^2 C, bytes 4 N C (x in,
// y out).  It replaces four launches (LayerNorm, qkv Dense, attention core, proj Dense + residual) whose traffic at level 0 is
// x + x^ + x^ + qkv + qkv + att + att + x + y = 15 N C halfs per window against 2 N C here.
//
// One WAVE per window (49 tokens as 4 MFMA tiles in the padded order row' = 8 ty + tx of window_attn.hip), persistent workgroups of
// 4 waves; both weight matrices and the two heads' relative-position tables live in LDS for the whole kernel (44 KB), each wave has a
// private 8 KB K / V image - no workgroup barrier inside the window loop.  Everything else stays in registers:
//   * x fragments (MFMA B operand: lane = token, 8 consecutive channels) -> LayerNorm in place (in-lane sums + two lane exchanges,
//     the expression of layernorm_kernel: the normalised row is rounded to fp16 where the separate launch rounded it);
//   * per head: [q | k | v]^T = W x^T + b with the weight rows interleaved in LDS so that a lane ends up with 8 CONSECUTIVE head
//     channels of its token - the packed fp16 result IS the B operand of S^T = K Q^T (q) or one 16-byte row chunk of the K / V image;
//   * the attention core of window_attn.hip (bias table as the MFMA C operand, register-local softmax, V^T by transposed LDS reads),
//     whose normalised output O^T - head channels 4g..4g+3 and 16+4g..16+4g+3 in a lane - is the B operand of the proj MFMA as it
//     stands, because the proj weight COLUMNS are stored in that order in LDS;
//   * y^T accumulates over the two heads from the proj bias, then + x (re-read: L2), fp16, 16-byte stores.
// A block with a global query (attention.py:60-66) takes q from q_global [B, 49, C] and computes k, v only.
// C = 128 (level 1): TWO waves per window (token tiles 0-1 / 2-3: the same 32 + 64 registers of x^ fragments and accumulators per wave),
// 8 waves = 4 windows per workgroup, the K / V image of a window shared by its two waves (workgroup barriers around it; every
// workgroup walks the same number of window quartets, a wave past the end runs on out-of-range offsets), qkv weights in LDS (108 KB),
// proj weight fragments straight from global (32 KB, L2-resident: LDS is full).
// Measured at B = 256 (tools/bench_gcvit_block.py): C = 64: 124 us local / 110 us global query against 301 / 264 us for the four launches;
// C = 128: 118 / 116 us against 172 / 152 us.  Per window and wave the C = 64 kernel issues 192 MFMAs (3.1 k matrix cycles), ~1 450
// VALU instructions (128 of them v_exp_f32) and ~200 LDS operations behind 145 s_waitcnt: two waves per SIMD do not cover those waits
// (VALU + MFMA issue alone would be ~60 us).  Tried and dropped: the next window's rows requested into the dead x^ registers under
// the last head's attention pass (124 -> 124 us: the x fetch is not what the waves wait for); two waves per window at C = 64 to get
// under 128 VGPRs and four waves per SIMD (158 us: workgroup barriers and 37 spilled registers cost more than the occupancy buys).
#include "common.hpp"

namespace {

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct GbArgs {
    const f16* x;
    const f16* qg;          // [B][49][C] or NULL
    const float* ln_g;
    const float* ln_b;
    float ln_eps;
    const f16* wqkv;        // [nq * C][ldwq], rows: q (if nq == 3), k, v; each [head][32]
    const float* bqkv;      // [nq * C] or NULL
    const f16* wproj;       // [C][ldwp]
    const float* bproj;     // [C] or NULL
    const float* table;     // [(2 ws - 1)^2][heads]
    f16* y;
    int B, Hp, Wp, nWy, nWx, ldwq, ldwp;
    long n_windows, x_bytes;
    float scale_log2e, inv_scale;
};

constexpr int GB_WS = 7;
constexpr int GB_NKT = 4;                                      // 64 key rows (8 x 8 padded grid)
constexpr int GB_TW = 16, GB_KSTEP = 32, GB_KCMAX = GB_KSTEP * GB_NKT, GB_NEGSZ = GB_KCMAX + 4, GB_TOFF = GB_NEGSZ + GB_KCMAX;
constexpr int GB_TROWS = 2 * GB_WS - 1, GB_TBF = GB_TOFF + GB_TROWS * GB_TW;      // floats per head (window_attn.hip WinCfg<7, 8, 3, 1>)
constexpr int GB_KVB = 64;                                     // K / V row: 32 halfs
constexpr int GB_KV_WIN = 2 * 64 * GB_KVB;                     // K + V image of one window (one head at a time)

template <int HEADS>
struct GbCfg {
    static constexpr int C = 32 * HEADS, CK = HEADS;           // channels; 32-wide k-steps of a K = C product
    static constexpr int WPW = HEADS == 2 ? 1 : 2;             // waves per window
    static constexpr int TT = 4 / WPW;                         // 16-token tiles per wave
    static constexpr int NWAVE = 4 * WPW;                      // 4 windows per workgroup pass
    static constexpr bool WP_LDS = HEADS == 2;                 // proj weights in LDS (C = 128: no room, fragments come from global)
    static constexpr int WROWB = 2 * C + 32;                   // weight row in LDS: C halfs + 32 B (stride = 32 mod 64: conflict-free b128)
    static constexpr int WQ_OFF = 0;
    static constexpr int WP_OFF = WQ_OFF + 3 * C * WROWB;
    static constexpr int TB_OFF = WP_OFF + (WP_LDS ? C * WROWB : 0);
    static constexpr int KV_OFF = (TB_OFF + HEADS * GB_TBF * 4 + 15) / 16 * 16;
    static constexpr int SMEM = KV_OFF + 4 * GB_KV_WIN;
};

// LDS row j of a 32-row group holds channel (r >> 2) * 8 + t * 4 + (r & 3), t = j >> 4, r = j & 15: MFMA row tiles 2m, 2m + 1 then
// leave lane group g with channels 8g .. 8g + 7 of the group (mlp_fused.hip)
__device__ __forceinline__ int gb_frag32(int j) {
    const int t = (j >> 4) & 1, r = j & 15;
    return (r >> 2) * 8 + t * 4 + (r & 3);
}
__device__ __forceinline__ int gb_k_slot(int row, int ch) {
    const int q = (row >> 2) & 3;
    return ch ^ ((0x78 >> (q * 2)) & 3);        // window_attn.hip k_slot
}

// LayerNorm of one 16-token tile held as fragments: lane (l15 = token, g) has channels 32 ks + 8 g .. + 7, ks < CK (mlp_fused.hip
// ln_fragments)
template <int CK>
__device__ __forceinline__ void gb_layernorm(U4H8 (&xf)[CK], const float* __restrict__ gam, const float* __restrict__ bet, float eps,
                                             int g) {
    constexpr int GB_C = 32 * CK;
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
    const float mean = sum / (float)GB_C;
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
    const float rstd = rsqrtf(sq / (float)GB_C + eps);
#pragma unroll
    for (int ks = 0; ks < CK; ++ks) {
        const float4 g0 = *reinterpret_cast<const float4*>(gam + ks * 32 + g * 8), g1 = *reinterpret_cast<const float4*>(gam + ks * 32 + g * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(bet + ks * 32 + g * 8), b1 = *reinterpret_cast<const float4*>(bet + ks * 32 + g * 8 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[ks].e[j] = (f16)((v[ks][j] - mean) * rstd * gg[j] + bb[j]);
    }
}

// One 16-query tile against the wave's K / V image: S^T = K Q^T + bias (table values are the C operand), softmax over the keys,
// O^T = V^T P^T, normalised and packed as the proj MFMA's B operand.  The arithmetic of window_attn.hip win_query_tile<7, 8, 3, 1>;
// the query of a lane is (qy, qx) of the PADDED order here (no division).
__device__ __forceinline__ U4H8 gb_attn_tile(const U4H8& qfrag, const char* k_lds, const char* v_lds, const float* tb, int qy, int qx,
                                             int l15, int g, float sc) {
    const int lane_term = (g >> 1) * GB_TW + 4 * (g & 1);
    const int tr_q = l15 >> 2, tr_p = l15 & 3;
    const float* tbase = tb + (GB_TOFF + qy * GB_TW + qx + (GB_WS - 1) * (GB_TW + 1) - lane_term - GB_KCMAX);
    f32x4 acc[GB_NKT];
#pragma unroll
    for (int t = 0; t < GB_NKT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int mask = 0;                       // lane groups whose key slot (t, r) is padding (compile-time)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int kp = 16 * t + 4 * gg + r;
                if (((kp & 7) >= GB_WS) || ((kp >> 3) >= GB_WS)) mask |= 1 << gg;
            }
            if (mask == 15) {
                acc[t][r] = -1.0e30f;
            } else {
                const int imm = GB_KCMAX - (GB_KSTEP * t + r);
                const float* bp = (mask == 0) ? tbase : (((mask >> g) & 1) ? tb : tbase);
                acc[t][r] = bp[imm];
            }
        }
    U4H8 kf[GB_NKT];
#pragma unroll
    for (int t = 0; t < GB_NKT; ++t) {
        const int row = t * 16 + l15;
        kf[t].u = *reinterpret_cast<const uint4*>(k_lds + row * GB_KVB + gb_k_slot(row, g) * 16);
    }
#pragma unroll
    for (int t = 0; t < GB_NKT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t].h, qfrag.h, acc[t], 0, 0, 0);
    float m = -1.0e30f;
#pragma unroll
    for (int t = 0; t < GB_NKT; ++t) {
        m = fmaxf(fmaxf(m, acc[t][0]), acc[t][1]);
        m = fmaxf(fmaxf(m, acc[t][2]), acc[t][3]);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const f32x2 nm = {-m * sc, -m * sc};
    f32x2 ls2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < GB_NKT; ++t)
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
    U4H8 pfa[GB_NKT / 2];
    union VF {
        fp16x4_t t[2];
        f16x8 v;
    } vfa[GB_NKT / 2][2];
#pragma unroll
    for (int s = 0; s < GB_NKT / 2; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pfa[s].e[j] = (f16)acc[2 * s][j];
            pfa[s].e[4 + j] = (f16)acc[2 * s + 1][j];
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int row = 32 * s + 16 * hh + 4 * g + tr_q;
                const int half = dt ^ ((row >> 2) & 1);
                vfa[s][dt].t[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) fp16x4_t*)(v_lds + row * GB_KVB + half * 32 + tr_p * 8));
            }
    }
#pragma unroll
    for (int s = 0; s < GB_NKT / 2; ++s)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfa[s][dt].v, pfa[s].h, o[dt], 0, 0, 0);
    const float inv = 1.f / lsum;
    U4H8 of;                                    // k-slot (g, j): head channel 4g + j (j < 4) / 16 + 4g + (j - 4)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        of.e[r] = (f16)(o[0][r] * inv);
        of.e[4 + r] = (f16)(o[1][r] * inv);
    }
    return of;
}

template <int HEADS, bool GLOBALQ>
__global__ __launch_bounds__(GbCfg<HEADS>::NWAVE * 64, HEADS == 2 ? 2 : 1) void gcvit_attn_block_kernel(GbArgs a) {
    using Cfg = GbCfg<HEADS>;
    constexpr int C = Cfg::C, CK = Cfg::CK, TT = Cfg::TT, WPW = Cfg::WPW, NTHR = Cfg::NWAVE * 64, WROWB = Cfg::WROWB, NCT = C / 16;
    constexpr int NQ = GLOBALQ ? 2 : 3;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wq = smem + Cfg::WQ_OFF;
    char* wp = smem + Cfg::WP_OFF;
    float* tbs = reinterpret_cast<float*>(smem + Cfg::TB_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int slot = wave / WPW, sub = wave % WPW;                 // window slot of the workgroup pass; which token tiles of it
    char* k_lds = smem + Cfg::KV_OFF + slot * GB_KV_WIN;
    char* v_lds = k_lds + 64 * GB_KVB;

    // ---- once per workgroup: weights (fragment-ordered) and the heads' bias tables ----
    constexpr int CPR = C / 8;                                     // 16-byte chunks per weight row
    for (int i = tid; i < NQ * C * CPR; i += NTHR) {
        const int j = i / CPR, c = i - j * CPR;
        const int ch = (j & ~31) + gb_frag32(j & 31);              // rows of [part][head][32] = consecutive groups of 32
        *reinterpret_cast<uint4*>(wq + j * WROWB + c * 16) = *reinterpret_cast<const uint4*>(a.wqkv + (long)ch * a.ldwq + c * 8);
    }
    if constexpr (Cfg::WP_LDS) {
        for (int i = tid; i < C * CPR; i += NTHR) {
            const int j = i / CPR, c = i - j * CPR;                // chunk c = head * 4 + lane group
            const int ch = (j & ~31) + gb_frag32(j & 31);
            const f16* src = a.wproj + (long)ch * a.ldwp + (c >> 2) * 32 + 4 * (c & 3);
            const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 16);
            *reinterpret_cast<uint4*>(wp + j * WROWB + c * 16) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
    for (int i = tid; i < HEADS * GB_TBF; i += NTHR) {
        const int head = i / GB_TBF, ii = i - head * GB_TBF;
        const int e = ii - GB_TOFF;
        const int ry = e / GB_TW, rx = e - ry * GB_TW;
        const bool in_tab = (ii >= GB_TOFF) & (rx < GB_TROWS);
        const float t = a.table[in_tab ? (ry * GB_TROWS + rx) * HEADS + head : 0];
        tbs[i] = in_tab ? t * a.inv_scale : (ii < GB_TOFF ? -1.0e30f : 0.f);
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbq =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.bqkv, 0, a.bqkv ? (unsigned)(NQ * C * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbp = __builtin_amdgcn_make_buffer_rsrc((void*)a.bproj, 0, a.bproj ? (unsigned)(C * 4) : 0u, 0x00020000);
    const char* wql = wq + l15 * WROWB + g * 16;
    const char* wpl = wp + l15 * WROWB + g * 16;
    const float sc = a.scale_log2e;
    const int wpi = a.nWy * a.nWx;
    // proj weight rows of this lane's fragments when they come from global: LDS-image row 16 ct + l15 = channel ...
    int pch[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) pch[ct] = ((16 * ct + l15) & ~31) + gb_frag32((16 * ct + l15) & 31);

    // One wave per window: waves walk their own windows.  Two waves per window: the workgroup walks window quartets in step (barriers
    // inside), a wave whose window is past the end works on out-of-range offsets (loads give zeros, stores are dropped).
    const long w_step = (long)gridDim.x * 4;
    for (long w0 = (long)blockIdx.x * 4; w0 < a.n_windows; w0 += w_step) {
        const long w = w0 + slot;
        const bool live = w < a.n_windows;
        if (WPW == 1 && !live) break;                                         // wave-uniform; no barriers in this form
        const long wc = live ? w : 0;
        const int b = (int)(wc / wpi);
        const int wrem = (int)(wc - (long)b * wpi);
        const int wy = wrem / a.nWx, wx = wrem - wy * a.nWx;
        // tokens of this lane: tile sub * TT + tt, row' = 16 tile + l15 = 8 ty + tx
        unsigned xoff[TT];
        int qyv[TT];
        const int tx = l15 & 7;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int ty = 2 * (sub * TT + tt) + (l15 >> 3);
            const bool valid = live & (tx < GB_WS) & (ty < GB_WS);
            const long pix = ((long)b * a.Hp + wy * GB_WS + ty) * a.Wp + wx * GB_WS + tx;
            xoff[tt] = valid ? (unsigned)(pix * C * 2) : OOB;
            qyv[tt] = ty < GB_WS ? ty : GB_WS - 1;
        }
        U4H8 xf[CK][TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int ks = 0; ks < CK; ++ks)
                xf[ks][tt].u = __builtin_bit_cast(
                    uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[tt] == OOB ? OOB : xoff[tt] + (ks * 32 + g * 8) * 2, 0, 0));
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            U4H8 col[CK];
#pragma unroll
            for (int ks = 0; ks < CK; ++ks) col[ks] = xf[ks][tt];
            gb_layernorm<CK>(col, a.ln_g, a.ln_b, a.ln_eps, g);
#pragma unroll
            for (int ks = 0; ks < CK; ++ks) xf[ks][tt] = col[ks];
        }
        // y^T accumulators from the proj bias: row tile ct, token tile tt; lane: channels 32 (ct >> 1) + 8 g + 4 (ct & 1) + 0..3
        f32x4 yacc[NCT][TT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const f32x4 bv = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbp, (unsigned)(((ct >> 1) * 32 + g * 8 + (ct & 1) * 4) * 4), 0, 0));
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) yacc[ct][tt] = bv;
        }

        constexpr int HEAD_UNROLL = HEADS == 2 ? 2 : 1;         // four unrolled heads spill (every head's weight reads get hoisted)
#pragma unroll HEAD_UNROLL
        for (int head = 0; head < HEADS; ++head) {
            U4H8 qf[TT];
            // ---- [q | k | v]^T = W x^T + b for this wave's token slots ----
#pragma unroll
            for (int part = 0; part < NQ; ++part) {
                const int grp = part * HEADS + head;                          // 32-row group of the LDS image = 32 consecutive outputs
                U4H8 wf[2][CK];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ks = 0; ks < CK; ++ks)
                        wf[t][ks].u = *reinterpret_cast<const uint4*>(wql + (grp * 32 + t * 16) * WROWB + ks * 64);
                f32x4 bv[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    bv[t] = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbq, (unsigned)((grp * 32 + g * 8 + t * 4) * 4), 0, 0));
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    f32x4 acc[2] = {bv[0], bv[1]};
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int ks = 0; ks < CK; ++ks)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][ks].h, xf[ks][tt].h, acc[t], 0, 0, 0);
                    U4H8 pk;                                                  // head channels 8 g .. 8 g + 7 of the lane's token
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        pk.e[i] = (f16)acc[0][i];
                        pk.e[4 + i] = (f16)acc[1][i];
                    }
                    const int row = (sub * TT + tt) * 16 + l15;
                    if (!GLOBALQ && part == 0) qf[tt] = pk;
                    else if (part == NQ - 2) *reinterpret_cast<uint4*>(k_lds + row * GB_KVB + gb_k_slot(row, g) * 16) = pk.u;
                    else *reinterpret_cast<uint4*>(v_lds + row * GB_KVB + ((g ^ (((row >> 2) & 1) << 1)) * 16)) = pk.u;
                }
            }
            if constexpr (GLOBALQ) {
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    const int ty = 2 * (sub * TT + tt) + (l15 >> 3);
                    const bool valid = (tx < GB_WS) & (ty < GB_WS);
                    const f16* src = a.qg + ((long)b * (GB_WS * GB_WS) + (valid ? ty * GB_WS + tx : 0)) * C + head * 32 + g * 8;
                    const uint4 v = *reinterpret_cast<const uint4*>(src);
                    qf[tt].u = valid ? v : make_uint4(0, 0, 0, 0);
                }
            }
            U4H8 pwf[NCT];                                                    // proj weight fragments of this head: row tile ct
            if constexpr (Cfg::WP_LDS) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) pwf[ct].u = *reinterpret_cast<const uint4*>(wpl + ct * 16 * WROWB + head * 64);
            } else {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {                            // k-slot (g, j): head channel 4g + j / 16 + 4g + (j - 4)
                    const f16* src = a.wproj + (long)pch[ct] * a.ldwp + head * 32 + 4 * g;
                    const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 16);
                    pwf[ct].u = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            }
            if constexpr (WPW == 1) {
                // the K / V image is private to the wave: its LDS writes and reads execute in program order, no barrier
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else {
                __syncthreads();
            }

            const float* tb = tbs + head * GB_TBF;
#pragma unroll
            for (int qt = 0; qt < TT; ++qt) {
                const U4H8 of = gb_attn_tile(qf[qt], k_lds, v_lds, tb, qyv[qt], tx < GB_WS ? tx : GB_WS - 1, l15, g, sc);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    yacc[ct][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pwf[ct].h, of.h, yacc[ct][qt], 0, 0, 0);
            }
            // the next head (or window) overwrites the image
            if constexpr (WPW == 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else {
                __syncthreads();
            }
        }

        // ---- epilogue: + x, fp16, 16-byte stores (lane: token, channels 32 hh + 8 g .. + 7) ----
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int hh = 0; hh < CK; ++hh) {
                const unsigned off = xoff[tt] == OOB ? OOB : xoff[tt] + (hh * 32 + g * 8) * 2;
                U4H8 r;
                r.u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
                U4H8 o;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f32x4 av = yacc[2 * hh + (j >> 2)][tt];
                    const f32x2 v = (f32x2){av[j & 3], av[(j & 3) + 1]} + (f32x2){(float)r.e[j], (float)r.e[j + 1]};
                    o.e[j] = (f16)v.x;
                    o.e[j + 1] = (f16)v.y;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o.u),
                                                       ry, off, 0, 0);
            }
    }
}

template <int HEADS>
int launch_gcvit_block(const GbArgs& a, bool global_q, hipStream_t s) {
    using Cfg = GbCfg<HEADS>;
    static_assert(Cfg::SMEM <= 160 * 1024, "LDS budget");
    long wgs = (a.n_windows + 3) / 4;
    const long cap = HEADS == 2 ? 512 : 256;                     // C = 64: two 76 KB workgroups per CU; C = 128: one of 148 KB
    if (wgs > cap) wgs = cap;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gcvit_attn_block_kernel<HEADS, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gcvit_attn_block_kernel<HEADS, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        attr_set = true;
    }
    if (global_q) hipLaunchKernelGGL((gcvit_attn_block_kernel<HEADS, true>), dim3((unsigned)wgs), dim3(Cfg::NWAVE * 64), Cfg::SMEM, s, a);
    else hipLaunchKernelGGL((gcvit_attn_block_kernel<HEADS, false>), dim3((unsigned)wgs), dim3(Cfg::NWAVE * 64), Cfg::SMEM, s, a);
    return vip_launch_status("vip_gcvit_attn_block_f16");
}

}  // namespace

#if !VIP_BUILD_EXPERIMENTS
// gcvit_block14.hip (the 14 x 14-window form: correct, not faster than the four launches) is an experiment-build source
int vip_gcvit_attn_block14(const void*, const void*, const float*, const float*, float, const void*, int, const float*, const void*, int,
                           const float*, const float*, void*, int, int, int, float, hipStream_t) {
    vip_set_error("vip_gcvit_attn_block_f16: the 14 x 14-window form is only in the experiments build (VIP_BUILD_EXPERIMENTS=1)");
    return VIP_ERR_UNSUPPORTED;
}
#endif

extern "C" int vip_gcvit_attn_block_supported(int C, int heads, int ws) {
    if (ws == 14) return VIP_BUILD_EXPERIMENTS && C == 256 && heads == 8;     // gcvit_block14.hip
    return ws == GB_WS && C == 32 * heads && (heads == 2 || heads == 4);
}

/* y = x + proj(window_attention(qkv(LayerNorm(x)))) for 7 x 7 windows, C = 64 / 2 heads or C = 128 / 4 heads
 * (vip_gcvit_attn_block_supported).
 * x, y [B][Hp][Wp][C] f16 (Hp, Wp multiples of 7; y must not alias x: a window reads its x again for the residual after other
 * windows have stored); q_global [B][49][C] f16 or NULL; wqkv [nq*C][ldwq] f16 with nq = 3 (q, k, v) or 2 (k, v; q_global given),
 * bqkv [nq*C] f32 or NULL; wproj [C][ldwp] f16, bproj [C] f32 or NULL; ln_gamma / ln_beta [C] f32;
 * table [(2 ws - 1)^2][heads] f32; scale = head_dim^-0.5. */
extern "C" int vip_gcvit_attn_block_f16(const void* x, const void* q_global, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                        const void* wqkv, int ldwq, const float* bqkv, const void* wproj, int ldwp, const float* bproj,
                                        const float* table, void* y, int B, int Hp, int Wp, int C, int heads, int ws, float scale,
                                        void* stream) {
    VIP_REQUIRE(x && ln_gamma && ln_beta && wqkv && wproj && table && y, VIP_ERR_BAD_ARG, "vip_gcvit_attn_block_f16: null pointer");
    VIP_REQUIRE(x != y, VIP_ERR_BAD_ARG, "vip_gcvit_attn_block_f16: y must not alias x");
    VIP_REQUIRE(vip_gcvit_attn_block_supported(C, heads, ws), VIP_ERR_UNSUPPORTED,
                "vip_gcvit_attn_block_f16: C=%d heads=%d ws=%d (only 64 / 2 / 7, 128 / 4 / 7 and 256 / 8 / 14)", C, heads, ws);
    VIP_REQUIRE(B > 0 && Hp > 0 && Wp > 0 && scale > 0.f, VIP_ERR_BAD_ARG, "vip_gcvit_attn_block_f16: non-positive dimension or scale");
    VIP_REQUIRE(Hp % ws == 0 && Wp % ws == 0, VIP_ERR_BAD_ARG, "vip_gcvit_attn_block_f16: feature map %dx%d not a multiple of the window",
                Hp, Wp);
    VIP_REQUIRE(ldwq >= C && ldwp >= C && ldwq % 8 == 0 && ldwp % 8 == 0, VIP_ERR_ALIGNMENT,
                "vip_gcvit_attn_block_f16: weight row strides must be multiples of 8 halfs and >= C");
    const long bytes = 2L * B * Hp * Wp * C;
    VIP_REQUIRE(bytes < 0xFFFFFFF0L, VIP_ERR_UNSUPPORTED, "vip_gcvit_attn_block_f16: tensor exceeds the 4 GiB buffer-addressing range");
    if (ws == 14)
        return vip_gcvit_attn_block14(x, q_global, ln_gamma, ln_beta, ln_eps, wqkv, ldwq, bqkv, wproj, ldwp, bproj, table, y, B, Hp, Wp, scale,
                                      (hipStream_t)stream);
    GbArgs a;
    a.x = (const f16*)x; a.qg = (const f16*)q_global; a.ln_g = ln_gamma; a.ln_b = ln_beta; a.ln_eps = ln_eps;
    a.wqkv = (const f16*)wqkv; a.bqkv = bqkv; a.wproj = (const f16*)wproj; a.bproj = bproj; a.table = table; a.y = (f16*)y;
    a.B = B; a.Hp = Hp; a.Wp = Wp; a.nWy = Hp / ws; a.nWx = Wp / ws; a.ldwq = ldwq; a.ldwp = ldwp;
    a.n_windows = (long)B * a.nWy * a.nWx;
    a.x_bytes = bytes;
    a.scale_log2e = scale * 1.44269504088896f;
    a.inv_scale = 1.f / scale;
    hipStream_t s = (hipStream_t)stream;
    if (heads == 2) return launch_gcvit_block<2>(a, q_global != nullptr, s);
    return launch_gcvit_block<4>(a, q_global != nullptr, s);
}
