// GCViT window attention core for CDNA4:  out = softmax(scale * q k^T + rel_pos_bias) v
// (reference: models/gcvit/layers/attention.py:52-83 with window_partition/reverse of window.py:3-15
//  folded into the addressing — qkv and out are plain [B,Hp,Wp,*] feature maps).
//
// Work item = (image, window, head), head_dim = 32.  ws=7: one wave per item, 4 items per workgroup;
// ws=14: four waves share one item (K/V staged once, query tiles split over the waves).
//
// Tokens are re-indexed on the way into LDS as row' = ty*P + tx with P = 8 (ws 7) or 16 (ws 14), so
//   * 49 -> 64 rows (2 MFMA tiles of 32), 196 -> 224 rows (7 tiles) and
//   * the relative-position index (dy+ws-1)*(2ws-1) + (dx+ws-1) becomes, on a [2ws-1][2P] LDS copy of
//     the head's bias table, base(query, lane-half) + a COMPILE-TIME constant per accumulator register:
//     one ds_read_b32 with an immediate offset per score, no index arithmetic; padded key columns are
//     redirected to a block of -1e30 by swapping the base register.
//
// Math: S^T = K Q^T with v_mfma_f32_32x32x16_f16 (keys on rows, queries on the lane), so the softmax
// over keys is register-local plus one lane^32 exchange, the probabilities are already the B operand of
// O^T = V^T P^T, V^T fragments come from ds_read_b64_tr_b16, and the 1/rowsum is lane-local.
#include "common.hpp"

namespace {

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct WinArgs {
    const f16* qkv;
    const f16* qg;
    const float* table;
    f16* out;
    int B, Hp, Wp, C, heads, nq;
    int nWy, nWx;
    int items;
    float scale_log2e;
};

template <int WS, int P, int LOG2P, int NT, int WPI>
struct WinCfg {
    static constexpr int RP = NT * 32;
    static constexpr int IPW = 4 / WPI;
    static constexpr int QK_STRIDE = 80;  // bytes: 64 B of data + 16 B pad (conflict-free ds_read_b128)
    static constexpr int V_STRIDE = 64;   // bytes: unpadded (conflict-free ds_read_b64_tr_b16)
    static constexpr int KCMAX = 2 * (RP - 1);
    static constexpr int NEGSZ = KCMAX + 2;
    static constexpr int TOFF = NEGSZ + KCMAX;
    static constexpr int TROWS = 2 * WS - 1;
    static constexpr int TW = 2 * P;
    static constexpr int TB_FLOATS = TOFF + TROWS * TW;
    static constexpr int Q_OFF = 0;
    static constexpr int K_OFF = RP * QK_STRIDE;
    static constexpr int V_OFF = 2 * RP * QK_STRIDE;
    static constexpr int T_OFF = V_OFF + RP * V_STRIDE;
    static constexpr int ITEM_BYTES = (T_OFF + TB_FLOATS * 4 + 15) / 16 * 16;
    static constexpr int SMEM = ITEM_BYTES * IPW;
};

template <int WS, int P, int LOG2P, int NT, int WPI>
__global__ __launch_bounds__(256) void window_attn_kernel(WinArgs a) {
    using Cfg = WinCfg<WS, P, LOG2P, NT, WPI>;
    constexpr int RP = Cfg::RP;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int slot = wave / WPI;            // item slot inside the workgroup
    const int lt = tid - slot * WPI * 64;   // thread index inside the item's thread group
    int item = blockIdx.x * Cfg::IPW + slot;
    const bool item_ok = item < a.items;
    if (!item_ok) item = a.items - 1;
    const int head = item % a.heads;
    int wq = item / a.heads;
    const int wx = wq % a.nWx;
    wq /= a.nWx;
    const int wy = wq % a.nWy;
    const int b = wq / a.nWy;

    char* base = smem + slot * Cfg::ITEM_BYTES;
    char* q_lds = base + Cfg::Q_OFF;
    char* k_lds = base + Cfg::K_OFF;
    char* v_lds = base + Cfg::V_OFF;
    float* tb = reinterpret_cast<float*>(base + Cfg::T_OFF);

    // ---- stage q / k / v (re-indexed, zero-padded) and the head's bias table -------------------
    const int ldq = a.nq * a.C;
    const long img_pix = (long)b * a.Hp * a.Wp;
    for (int s = lt; s < 3 * RP * 4; s += WPI * 64) {
        const int arr = s / (RP * 4);
        const int rem = s - arr * (RP * 4);
        const int row = rem >> 2, ch = rem & 3;
        const int ty = row >> LOG2P, tx = row & (P - 1);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ty < WS && tx < WS) {
            const long pix = img_pix + (long)(wy * WS + ty) * a.Wp + (wx * WS + tx);
            const f16* src;
            if (arr == 0) {
                src = (a.nq == 3) ? a.qkv + pix * ldq + head * 32 + ch * 8
                                  : a.qg + ((long)b * WS * WS + ty * WS + tx) * a.C + head * 32 + ch * 8;
            } else {
                src = a.qkv + pix * ldq + (a.nq - 3 + arr) * a.C + head * 32 + ch * 8;
            }
            v = *reinterpret_cast<const uint4*>(src);
        }
        char* dst = (arr == 0) ? q_lds + row * Cfg::QK_STRIDE
                               : (arr == 1 ? k_lds + row * Cfg::QK_STRIDE : v_lds + row * Cfg::V_STRIDE);
        *reinterpret_cast<uint4*>(dst + ch * 16) = v;
    }
    for (int i = lt; i < Cfg::TB_FLOATS; i += WPI * 64) {
        float v = -1.0e30f;
        if (i >= Cfg::TOFF) {
            const int e = i - Cfg::TOFF;
            const int ry = e / Cfg::TW, rx = e - ry * Cfg::TW;
            v = (rx < Cfg::TROWS) ? a.table[(long)(ry * Cfg::TROWS + rx) * a.heads + head] * 1.44269504088896f : 0.f;
        }
        tb[i] = v;
    }
    __syncthreads();

    const int l31 = lane & 31, h = lane >> 5;
    const int wi = wave % WPI;
    const float sc = a.scale_log2e;

#pragma unroll 1
    for (int qt = wi; qt < NT; qt += WPI) {
        const int qrow = qt * 32 + l31;
        U4H8 qf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
            qf[t].u = *reinterpret_cast<const uint4*>(q_lds + qrow * Cfg::QK_STRIDE + (16 * t + 8 * h) * 2);

        // S^T tiles: rows = keys kt*32.., cols = this lane's query
        f32x16 acc[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                U4H8 kf;
                kf.u = *reinterpret_cast<const uint4*>(k_lds + (kt * 32 + l31) * Cfg::QK_STRIDE + (16 * t + 8 * h) * 2);
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf.h, qf[t].h, acc[kt], 0, 0, 0);
            }
        }

        // scale + relative-position bias (+ key padding mask), running max
        const int qy = qrow >> LOG2P, qx = qrow & (P - 1);
        const int qyc = qy < WS ? qy : WS - 1, qxc = qx < WS ? qx : WS - 1;
        const int bidx = Cfg::TOFF + qyc * Cfg::TW + qxc + (WS - 1) * (Cfg::TW + 1) - 4 * h - Cfg::KCMAX;
        const float* tbase = tb + bidx;
        const float* tmask = h ? tb : tbase;
        float m = -1.0e30f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                constexpr int dummy = 0;
                (void)dummy;
                const int c = (r & 3) + 8 * (r >> 2);
                const int kp = kt * 32 + c;           // key row' without the 4*h term
                const int kyc = kp >> LOG2P;          // 4*h never carries into ty
                const int kxc = c & (P - 1);
                float s;
                if (kyc >= WS) {
                    s = -1.0e30f;
                } else {
                    const int imm = Cfg::KCMAX - (2 * kp - kxc);
                    const float bias = (kxc + 4 >= WS) ? tmask[imm] : tbase[imm];
                    s = acc[kt][r] * sc + bias;
                }
                acc[kt][r] = s;
                m = fmaxf(m, s);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float lsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(acc[kt][r] - m);
                acc[kt][r] = p;
                lsum += p;
            }
        }
        lsum += __shfl_xor(lsum, 32, 64);

        // O^T = V^T P^T : A = V^T via transposed LDS reads, B = P^T straight from the accumulators
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
        const char* vlane = v_lds + (4 * h + tr_q) * Cfg::V_STRIDE + (16 * tr_g + 4 * tr_p) * 2;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                U4H8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf.e[j] = (f16)acc[kt][8 * s + j];
                const char* vp = vlane + (kt * 32 + 16 * s) * Cfg::V_STRIDE;
                union {
                    fp16x4_t t[2];
                    f16x8 v;
                } vf;
                vf.t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) fp16x4_t*)(vp));
                vf.t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) fp16x4_t*)(vp + 8 * Cfg::V_STRIDE));
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf.v, pf.h, o, 0, 0, 0);
            }
        }

        // store: lane owns query qrow, head-dim d = 8g + 4h + (0..3)
        if (item_ok && qy < WS && qx < WS) {
            const float inv = 1.f / lsum;
            const long pix = img_pix + (long)(wy * WS + qy) * a.Wp + (wx * WS + qx);
            f16* dst = a.out + pix * a.C + head * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = (f16)(o[4 * g + i] * inv);
                *reinterpret_cast<f16x4*>(dst + 8 * g) = ov;
            }
        }
    }
}

template <int WS, int P, int LOG2P, int NT, int WPI>
int launch_win(const WinArgs& a, hipStream_t s) {
    using Cfg = WinCfg<WS, P, LOG2P, NT, WPI>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_kernel<WS, P, LOG2P, NT, WPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        attr_set = true;
    }
    const int wgs = (a.items + Cfg::IPW - 1) / Cfg::IPW;
    hipLaunchKernelGGL((window_attn_kernel<WS, P, LOG2P, NT, WPI>), dim3(wgs), dim3(256), Cfg::SMEM, s, a);
    return vip_launch_status("vip_window_attn_fwd_f16");
}

}  // namespace

extern "C" int vip_window_attn_fwd_f16(const void* qkv, const void* q_global, const float* bias_table, void* out,
                                       int B, int Hp, int Wp, int C, int heads, int ws, int nq, float scale,
                                       void* stream) {
    VIP_REQUIRE(qkv && bias_table && out, VIP_ERR_BAD_ARG, "vip_window_attn_fwd_f16: null pointer");
    VIP_REQUIRE(nq == 3 || (nq == 2 && q_global), VIP_ERR_BAD_ARG,
                "vip_window_attn_fwd_f16: nq must be 3, or 2 with a q_global tensor");
    VIP_REQUIRE(B > 0 && Hp > 0 && Wp > 0 && C > 0 && heads > 0, VIP_ERR_BAD_ARG,
                "vip_window_attn_fwd_f16: non-positive dimension");
    VIP_REQUIRE(C == heads * 32, VIP_ERR_UNSUPPORTED, "vip_window_attn_fwd_f16: head_dim = C/heads must be 32 (C=%d heads=%d)", C, heads);
    VIP_REQUIRE(ws == 7 || ws == 14, VIP_ERR_UNSUPPORTED, "vip_window_attn_fwd_f16: window size %d (only 7, 14)", ws);
    VIP_REQUIRE(Hp % ws == 0 && Wp % ws == 0, VIP_ERR_BAD_ARG,
                "vip_window_attn_fwd_f16: feature map %dx%d not a multiple of the window %d", Hp, Wp, ws);
    WinArgs a;
    a.qkv = (const f16*)qkv; a.qg = (const f16*)q_global; a.table = bias_table; a.out = (f16*)out;
    a.B = B; a.Hp = Hp; a.Wp = Wp; a.C = C; a.heads = heads; a.nq = nq;
    a.nWy = Hp / ws; a.nWx = Wp / ws;
    const long items = (long)B * a.nWy * a.nWx * heads;
    VIP_REQUIRE(items < (1L << 30), VIP_ERR_UNSUPPORTED, "vip_window_attn_fwd_f16: too many windows");
    a.items = (int)items;
    a.scale_log2e = scale * 1.44269504088896f;
    if (ws == 7) return launch_win<7, 8, 3, 2, 1>(a, (hipStream_t)stream);
    return launch_win<14, 16, 4, 7, 4>(a, (hipStream_t)stream);
}
