// Attention cores of the packed STRICT path on the matrix cores: out = softmax(scale * q k^T (+ relative-position bias)) v with every
// operand an fp16 (hi, lo) pair (common.hpp) and THREE v_mfma_f32_16x16x32_f16 per fragment pair - S = k_lo q_hi + k_hi q_lo + k_hi q_hi
// (fp32 accumulate), the softmax in fp32 registers, P split into (hi, lo) again for O = v_lo p_hi + v_hi p_lo + v_hi p_hi.
//   WS = 7 / 14: GCViT WindowAttention.call (gcvit/layers/attention.py:52-83) on the feature-map layout qkv [B,Hp,Wp,nq*C] (channels
//                (q|k|v or k|v, head, hd), hd = 32); q of the global-query blocks from q_global [B, ws*ws, C] (:62-66, also scaled :69);
//                bias = table[(dy + ws - 1)(2 ws - 1) + dx + ws - 1][head], d = query - key coordinate (:39-50).
//   WS = 0:      tfimm ViTMultiHeadAttention (vit.py:148-167): qkv [B,N,3D], hd = 64, item = (image, head), N <= 224.
// One 4-wave workgroup per (window | image, head): K and V of the item are staged once into four LDS images (K hi, K lo, V hi, V lo;
// rows of hd halfs, zero padded to whole 32-key steps), the 16-query tiles are dealt round-robin to the waves, Q fragments come
// straight from global (a lane's 8 consecutive channels are one 32-byte (hi, lo) group).  The scheme is mhsa.hip's / window_attn.hip's:
// S^T = K Q^T (keys on accumulator rows, the query on the lane) -> softmax register-local + two lane exchanges -> the probabilities ARE
// the B operand of O^T = V^T P^T, V^T fragments by ds_read_b64_tr_b16.  The table values (divided by the scale) are the MFMAs' C operand:
// the accumulators are initialised with them BEFORE the first MFMA (the form that is correct next to MFMA-heavy co-runners, DESIGN.md
// section 5).  Replaces the one-thread-per-query fp32 VALU kernels of strict_ops.hip for this storage.
#include "common.hpp"

namespace {

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct AttnH2Args {
    const char* qkv;
    const char* qg;
    const float* table;
    char* out;
    int B, Hp, Wp, C, heads, nq, N;
    float scale;
    int* status;
};

template <int HD, int WS>
struct AttnCfg {
    static constexpr int NMAX = WS ? WS * WS : 224;
    static constexpr int RP = (NMAX + 31) / 32 * 32;        // key rows incl. zero padding (whole 32-key PV steps)
    static constexpr int NKT = RP / 16;                     // 16-key tiles
    static constexpr int ROWB = HD * 2;                     // bytes per row of an image
    static constexpr int IMG = RP * ROWB;
    static constexpr int TBL = WS ? (2 * WS - 1) * (2 * WS - 1) : 0;
    static constexpr int SMEM = 4 * IMG + (TBL * 4 + 15) / 16 * 16;
    static constexpr int KS = HD / 32;                      // MFMA k-steps over the head dimension
    static constexpr int DT = HD / 16;                      // 16-channel output tiles
};

// physical 16-byte chunk of logical chunk ch of K row `row` (conflict-free ds_read_b128 for the fragment pattern lane&15 = row,
// lane>>4 = chunk: the read is served in the lane groups {0-3,12-15,20-27}, ... - MI355X_MICROARCH.md, LDS)
template <int HD>
__device__ __forceinline__ int k_chunk(int row, int ch) {
    if constexpr (HD == 64) return ch ^ (row & 7);
    else return ch ^ ((0x78u >> (2 * ((row >> 2) & 3))) & 3);          // pi = {0, 2, 3, 1}
}
// V: 32-byte slots (one 16-channel output tile) swizzled so that the 8 rows a ds_read_b64_tr_b16 half-wave touches hit all banks
template <int HD>
__device__ __forceinline__ int v_chunk(int row, int ch) {
    if constexpr (HD == 64) return ((((ch >> 1) ^ ((row >> 1) & 3)) << 1) | (ch & 1));
    else return ch ^ (((row >> 2) & 1) << 1);
}
template <int HD>
__device__ __forceinline__ int v_slot(int row, int dt) {
    if constexpr (HD == 64) return dt ^ ((row >> 1) & 3);
    else return dt ^ ((row >> 2) & 1);
}

// threads per workgroup: the hd = 64 item (ViT, 197 tokens) needs 115 KB of LDS, one workgroup per CU - with 4 waves that is ONE wave per
// SIMD and nothing to hide a wait behind; 8 waves (two per SIMD, 196 registers each) share the 13 query tiles instead
template <int HD>
constexpr int attn_h2_threads() { return HD == 64 ? 512 : 256; }

template <int HD, int WS>
__global__ __launch_bounds__(attn_h2_threads<HD>(), 2) void attn_h2_kernel(AttnH2Args a) {
    using Cfg = AttnCfg<HD, WS>;
    constexpr int NT = attn_h2_threads<HD>();
    constexpr int RP = Cfg::RP, NKT = Cfg::NKT, ROWB = Cfg::ROWB, KS = Cfg::KS, DT = Cfg::DT, CPR = HD / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_hi = smem;
    char* k_lo = smem + Cfg::IMG;
    char* v_hi = smem + 2 * Cfg::IMG;
    char* v_lo = smem + 3 * Cfg::IMG;
    float* tbl = reinterpret_cast<float*>(smem + 4 * Cfg::IMG);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    int item = blockIdx.x;
    const int head = item % a.heads;
    item /= a.heads;
    const int ld = a.nq * a.C;                              // elements per token row of qkv
    int wx = 0, wy = 0, b;
    if constexpr (WS > 0) {
        const int nwx = a.Wp / WS, nwy = a.Hp / WS;
        wx = item % nwx;
        item /= nwx;
        wy = item % nwy;
        b = item / nwy;
    } else {
        b = item;
    }
    const int N = a.N;
    auto tok_elem = [&](int t) -> long {                    // element index of token t's channel 0 in qkv
        if constexpr (WS > 0) {
            const int ty = t / WS, tx = t - ty * WS;
            return (((long)b * a.Hp + wy * WS + ty) * a.Wp + wx * WS + tx) * ld;
        } else {
            return ((long)b * N + t) * ld;
        }
    };
    const int koff = (a.nq - 2) * a.C + head * HD, voff = (a.nq - 1) * a.C + head * HD;

    // ---- stage K, V: slot = (array, row, 8-channel group) -> one 32-byte (hi, lo) pair from global, two 16-byte LDS writes ----
    {
        constexpr int NSLOT = 2 * RP * CPR;
        constexpr int NIT = (NSLOT + NT - 1) / NT;
        constexpr int BATCH = 7;
        for (int it0 = 0; it0 < NIT; it0 += BATCH) {
            uint4 sh[BATCH], sl[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int s = tid + (it0 + u) * NT;
                const int arr = s / (RP * CPR);
                const int rem = s - arr * (RP * CPR);
                const int row = rem / CPR, cg = rem - row * CPR;
                const bool valid = (it0 + u < NIT) && s < NSLOT && row < N;
                const char* src = valid ? a.qkv + (tok_elem(row) + (arr ? voff : koff) + cg * 8) * 4 : a.qkv;
                const uint4 h = *reinterpret_cast<const uint4*>(src);
                const uint4 l = *reinterpret_cast<const uint4*>(src + 16);
                sh[u] = valid ? h : make_uint4(0, 0, 0, 0);
                sl[u] = valid ? l : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int s = tid + (it0 + u) * NT;
                if (it0 + u < NIT && s < NSLOT) {
                    const int arr = s / (RP * CPR);
                    const int rem = s - arr * (RP * CPR);
                    const int row = rem / CPR, cg = rem - row * CPR;
                    const int pch = arr == 0 ? k_chunk<HD>(row, cg) : v_chunk<HD>(row, cg);
                    *reinterpret_cast<uint4*>((arr == 0 ? k_hi : v_hi) + row * ROWB + pch * 16) = sh[u];
                    *reinterpret_cast<uint4*>((arr == 0 ? k_lo : v_lo) + row * ROWB + pch * 16) = sl[u];
                }
            }
        }
        if constexpr (WS > 0) {                             // this head's table column, pre-divided by the scale (the MFMA C operand)
            const float inv = 1.0f / a.scale;
            for (int i = tid; i < Cfg::TBL; i += NT) tbl[i] = a.table[(long)i * a.heads + head] * inv;
        }
    }
    __syncthreads();

    const float sc = a.scale * 1.44269504088896f;           // scores in log2 units: exp2(s - m) = e^(natural difference)
    const int tr_q = l15 >> 2, tr_p = l15 & 3;
    const int nkt = (N + 15) >> 4;                          // key tiles that hold keys (uniform)
    const int nks = (nkt + 1) >> 1;                         // 32-key PV steps
    bool bad = false;

    for (int qt = wave; qt < nkt; qt += NT / 64) {
        const int qn = qt * 16 + l15;
        const bool qok = qn < N;
        // ---- Q fragments (hi, lo) of this lane's query: channels ks * 32 + g * 8 .. + 7 of the head ----
        U4H8 qh[KS], ql[KS];
        {
            const bool gq = WS > 0 && a.qg != nullptr;
            const char* qb = gq ? a.qg : a.qkv;
            const long qe = gq ? ((long)b * N + (qok ? qn : 0)) * a.C + head * HD : tok_elem(qok ? qn : 0) + head * HD;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const char* src = qb + (qe + ks * 32 + g * 8) * 4;
                const uint4 h = *reinterpret_cast<const uint4*>(src);
                const uint4 l = *reinterpret_cast<const uint4*>(src + 16);
                qh[ks].u = qok ? h : make_uint4(0, 0, 0, 0);
                ql[ks].u = qok ? l : make_uint4(0, 0, 0, 0);
            }
        }
        int qy = 0, qx = 0;
        if constexpr (WS > 0) {
            const int q_ = qok ? qn : 0;
            qy = q_ / WS;
            qx = q_ - qy * WS;
        }

        f32x4 acc[NKT];
        float m = -1.0e30f;
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (t < nkt) {                                  // uniform
                if constexpr (WS > 0) {                     // bias / scale into the accumulators before the MFMAs
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = t * 16 + 4 * g + r;
                        const int kk = key < N ? key : 0;
                        const int ky = kk / WS, kx = kk - ky * WS;
                        acc[t][r] = tbl[(qy - ky + WS - 1) * (2 * WS - 1) + (qx - kx + WS - 1)];
                    }
                }
                const int row = t * 16 + l15;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    U4H8 kh, kl;
                    const int off = row * ROWB + (k_chunk<HD>(row, ks * 4 + g) << 4);
                    kh.u = *reinterpret_cast<const uint4*>(k_hi + off);
                    kl.u = *reinterpret_cast<const uint4*>(k_lo + off);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl.h, qh[ks].h, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh.h, ql[ks].h, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh.h, qh[ks].h, acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * 16 + 4 * g + r;
                    const float s = (key < N) ? acc[t][r] * sc : -1.0e30f;
                    acc[t][r] = s;
                    m = fmaxf(m, s);
                }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float lsum = 0.f;
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
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

        // O^T = V^T P^T; MFMA k-slot (g, j) carries key 32 s + 4 g + j (j < 4) / 32 s + 16 + 4 g + (j - 4)
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NKT / 2; ++s) {
            if (s < nks) {                                  // uniform
                U4H8 ph, pl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p0 = acc[2 * s][j];
                    const float p1 = (2 * s + 1 < nkt) ? acc[2 * s + 1][j] : 0.f;
                    ph.e[j] = (f16)p0;
                    pl.e[j] = (f16)(p0 - (float)ph.e[j]);
                    ph.e[4 + j] = (f16)p1;
                    pl.e[4 + j] = (f16)(p1 - (float)ph.e[4 + j]);
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    union {
                        fp16x4_t t[2];
                        f16x8 v;
                    } vh, vl;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int row = 32 * s + 16 * hh + 4 * g + tr_q;
                        const int off = row * ROWB + v_slot<HD>(row, dt) * 32 + tr_p * 8;
                        vh.t[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(v_hi + off));
                        vl.t[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(v_lo + off));
                    }
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl.v, ph.h, o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh.v, pl.h, o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh.v, ph.h, o[dt], 0, 0, 0);
                }
            }
        }

        if (qok) {
            long oe;                                        // element index of the query's channel 0 of this head in out [.., C]
            if constexpr (WS > 0) oe = (((long)b * a.Hp + wy * WS + qy) * a.Wp + wx * WS + qx) * a.C + head * HD;
            else oe = ((long)b * N + qn) * a.C + head * HD;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f32x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ov[r] = o[dt][r] / lsum;
                    bad |= !(fabsf(ov[r]) <= VIP_H2_MAX);
                }
                h2_st4(a.out, oe + dt * 16 + 4 * g, ov, nullptr);
            }
        }
    }
    if (bad && a.status) *a.status = VIP_H2_OVERFLOW;
}

template <int HD, int WS>
int launch_attn_h2(const AttnH2Args& a, long items, hipStream_t s, const char* who) {
    using Cfg = AttnCfg<HD, WS>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_h2_kernel<HD, WS>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        attr = true;
    }
    hipLaunchKernelGGL((attn_h2_kernel<HD, WS>), dim3((unsigned)items), dim3(attn_h2_threads<HD>()), Cfg::SMEM, s, a);
    return vip_launch_status(who);
}

}  // namespace

// the MFMA forms behind vip_window_attn_fwd_h2 / vip_mhsa_fwd_h2 (strict_ops.hip validates the arguments); return 1 when the
// configuration is not built here (the fp32 VALU kernel takes it)
int vip_window_attn_h2_mfma(const void* qkv, const void* q_global, const float* bias_table, void* out, int B, int Hp, int Wp, int C, int heads,
                            int ws, int nq, float scale, int* status, hipStream_t s) {
    if (ws != 7 && ws != 14) return 1;
    AttnH2Args a{(const char*)qkv, (const char*)q_global, bias_table, (char*)out, B, Hp, Wp, C, heads, nq, ws * ws, scale, status};
    const long items = (long)B * (Hp / ws) * (Wp / ws) * heads;
    return ws == 7 ? launch_attn_h2<32, 7>(a, items, s, "vip_window_attn_fwd_h2(mfma)") : launch_attn_h2<32, 14>(a, items, s, "vip_window_attn_fwd_h2(mfma)");
}

int vip_mhsa_h2_mfma(const void* qkv, void* out, int B, int N, int D, int heads, float scale, int* status, hipStream_t s) {
    if (N > 224) return 1;
    AttnH2Args a{(const char*)qkv, nullptr, nullptr, (char*)out, B, 0, 0, D, heads, 3, N, scale, status};
    return launch_attn_h2<64, 0>(a, (long)B * heads, s, "vip_mhsa_fwd_h2(mfma)");
}
