// REPRODUCER, not product code: the round-2 window-attention kernel (git 56edddf^ : vip-cup-2022_amd/csrc/window_attn.hip) that returns
// wrong tiles when an MFMA-heavy kernel shares its SIMDs, kept with bisecting switches (tools/repro/run_race_repro.sh builds each
// variant into its own library and runs tools/race_matrix.py on it; DESIGN.md section 5 has the findings):
//   RACE_VARIANT 0  as it was: S^T = K Q^T with C = 0, then table value read from LDS + v_pk_add_f32 on the MFMA result
//   RACE_VARIANT 1  64 idle cycles (s_nop) and a scheduling fence between the last QK MFMA and everything that follows
//   RACE_VARIANT 3  two scalar v_add_f32 instead of the packed add
//   RACE_VARIANT 4  ALL table values read into registers and waited for BEFORE the first MFMA (C still 0), packed adds afterwards
//   RACE_VARIANT 5  accumulators pinned (asm barrier) after the MFMAs and again after the adds: no register of a result is reused early
//   RACE_VARIANT 6  s_waitcnt lgkmcnt(0) in front of every packed add (no counted LDS waits in the bias section)
//   RACE_VARIANT 7  s_nop 7 + scheduling fence AFTER every packed add (nothing may touch its source registers for 8 cycles)
//   RACE_VARIANT 8  the two table values of a pair loaded by two ds_read_b32 into a register pair in NATURAL order (no op_sel swap)
#ifndef RACE_VARIANT
#define RACE_VARIANT 0
#endif
// GCViT window attention core for CDNA4:  out = softmax(scale * q k^T + rel_pos_bias) v
// (reference: models/gcvit/layers/attention.py:52-83 with window_partition/reverse of window.py:3-15
//  folded into the addressing — qkv and out are plain [B,Hp,Wp,*] feature maps).
//
// Work item = (image, window, head), head_dim = 32.  ws=7: one wave per item, 4 items per workgroup;
// ws=14: four waves share one item (K/V staged once, the 13 query tiles are dealt round-robin).
// The kernel is HBM-bound by design (intensity N/2 FLOP/B); its job is to keep enough independent
// workgroups resident per CU (<= 128 registers, <= 40 KB LDS) that staging, softmax VALU work and MFMA
// of different workgroups overlap.
//
// Tokens are re-indexed on the way into LDS as row' = ty*P + tx with P = 8 (ws 7) or 16 (ws 14), so
//   * 49 -> 64 key rows, 196 -> 224 key rows (MFMA tiles of 16) and
//   * the relative-position index (dy+ws-1)*(2ws-1) + (dx+ws-1) becomes, on a [2ws-1][2P] LDS copy of
//     the head's bias table, base(query, lane-group) + a COMPILE-TIME constant per accumulator register:
//     one LDS read with an immediate offset per score, no index arithmetic; padded key slots are
//     redirected to a block of -1e30 by swapping the base register.
//
// Math, per 16-query tile: S^T = K Q^T with v_mfma_f32_16x16x32_f16 (keys on rows, queries on the
// lane: the whole head_dim is one MFMA), so the softmax over keys is register-local plus two cross-lane
// exchanges, the probabilities are already the B operand of O^T = V^T P^T, V^T fragments come from
// ds_read_b64_tr_b16, and the 1/rowsum is lane-local.  Q fragments are loaded straight from global.
//
// LDS images (64-byte rows, no padding): K chunk c of row r at slot c ^ pi[(r>>2)&3], pi = {0,2,3,1}
// (conflict-free ds_read_b128 fragment reads); V 32-byte halves swapped on rows with bit 2 set
// (conflict-free transposed reads).
#include "common.hpp"
#include <stdlib.h>

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
    float scale_log2e, inv_scale;
};

template <int WS, int P, int LOG2P, int WPI>
struct WinCfg {
    static constexpr int RP = (WS * P + 31) / 32 * 32;  // key rows (64 / 224)
    static constexpr int NKT = RP / 16;                 // 16-key tiles
    static constexpr int NQT = (WS * WS + 15) / 16;     // 16-query tiles over the DENSE token order (4 / 13)
    static constexpr int IPW = 4 / WPI;
    static constexpr int ROWB = 64;
    // bias-table row stride (floats).  ws 14: 48, not 32 - with a 32-float stride the two token rows a query tile
    // spans land on the same LDS banks (2-way conflicts on 28 % of the table reads); 48 shifts them by 16 banks.
    static constexpr int TW = (P == 16) ? 48 : 2 * P;
    static constexpr int KSTEP = TW * 16 / P;           // table-offset step of one 16-key tile
    static constexpr int KCMAX = KSTEP * NKT;
    static constexpr int NEGSZ = KCMAX + 4;
    static constexpr int TOFF = NEGSZ + KCMAX;
    static constexpr int TROWS = 2 * WS - 1;
    static constexpr int TB_FLOATS = TOFF + TROWS * TW;
    static constexpr int K_OFF = 0;
    static constexpr int V_OFF = RP * ROWB;
    static constexpr int T_OFF = 2 * RP * ROWB;
    static constexpr int ITEM_BYTES = (T_OFF + TB_FLOATS * 4 + 15) / 16 * 16;
    static constexpr int SMEM = ITEM_BYTES * IPW;
    static constexpr int QPW = (NQT + WPI - 1) / WPI;   // query tiles per wave
};

__device__ __forceinline__ int k_slot(int row, int ch) {
    const int q = (row >> 2) & 3;
    return ch ^ ((0x78 >> (q * 2)) & 3);  // pi = {0,2,3,1}
}

// One 16-query tile of one (window, head) item: S^T = K Q^T, bias, softmax, O^T = V^T P^T, store.  Shared by both kernels.
template <int WS, int P, int LOG2P, int WPI>
__device__ __forceinline__ void win_query_tile(const WinArgs& a, const U4H8& qfrag, const char* k_lds, const char* v_lds,
                                               const float* tb, int qt, int l15, int g, bool item_ok, long img_pix, int wy,
                                               int wx, int head) {
    using Cfg = WinCfg<WS, P, LOG2P, WPI>;
    constexpr int NKT = Cfg::NKT;
    const float sc = a.scale_log2e;
    // lane-dependent part of the key term 2k' - kx  (k' = 16t + 4g + r)
    const int lane_term = (P == 16) ? 4 * g : (g >> 1) * Cfg::TW + 4 * (g & 1);
    // LDS byte offsets of this lane's fragment reads
    const int tr_q = l15 >> 2, tr_p = l15 & 3;
    {
        const int qn = qt * 16 + l15;

        // S^T tiles: rows = keys 16t + 4g + r, column = this lane's query.  The relative-position bias (stored as
        // table/scale, padded key slots redirected to the -1e30 block) is added with one packed add per pair (as the MFMA
        // C operand it cost a v_mov per value to assemble), so the softmax needs one packed FMA + one exp2 per score:
        //   p = exp2(scale*log2e * s' - scale*log2e * max s')
        const int qy = qn / WS, qx = qn - qy * WS;
        const int qyc = qy < WS ? qy : WS - 1, qxc = qx;
        const float* tbase = tb + (Cfg::TOFF + qyc * Cfg::TW + qxc + (WS - 1) * (Cfg::TW + 1) - lane_term - Cfg::KCMAX);
        f32x4 acc[NKT];
        auto table_value = [&](int t, int r) -> float {
            // which lane groups g hold a padded key slot in register (t, r)?  (compile-time 4-bit mask)
            int mask = 0;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int kp = 16 * t + 4 * gg + r;
                if (((kp & (P - 1)) >= WS) || ((kp >> LOG2P) >= WS)) mask |= 1 << gg;
            }
            if (mask == 15) return -1.0e30f;
            const int imm = Cfg::KCMAX - (Cfg::KSTEP * t + r);
            const float* bp = (mask == 0) ? tbase : (((mask >> g) & 1) ? tb : tbase);
            return bp[imm];
        };
#if RACE_VARIANT == 4
        f32x4 bvals[NKT];
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) bvals[t][r] = table_value(t, r);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < NKT; ++t) asm volatile("" : "+v"(bvals[t]));
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            const int row = t * 16 + l15;
            U4H8 kf;
            kf.u = *reinterpret_cast<const uint4*>(k_lds + row * Cfg::ROWB + k_slot(row, g) * 16);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf.h, qfrag.h, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
#if RACE_VARIANT == 1
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#endif
#if RACE_VARIANT == 5
#pragma unroll
        for (int t = 0; t < NKT; ++t) asm volatile("" : "+v"(acc[t]));
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
#if RACE_VARIANT == 4
                const f32x2 bv = {bvals[t][r], bvals[t][r + 1]};
#else
                const f32x2 bv = {table_value(t, r), table_value(t, r + 1)};
#endif
#if RACE_VARIANT == 3
                acc[t][r] = __fadd_rn(acc[t][r], bv.x);
                acc[t][r + 1] = __fadd_rn(acc[t][r + 1], bv.y);
                asm volatile("" : "+v"(acc[t][r]), "+v"(acc[t][r + 1]));      // keep the two adds scalar
#else
#if RACE_VARIANT == 6
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#if RACE_VARIANT == 8
                f32x2 bn = bv;
                asm volatile("" : "+v"(bn));                                  // a register pair of its own, in natural order
                const f32x2 sv = (f32x2){acc[t][r], acc[t][r + 1]} + bn;
#else
                const f32x2 sv = (f32x2){acc[t][r], acc[t][r + 1]} + bv;   // v_pk_add_f32
#endif
                acc[t][r] = sv.x;
                acc[t][r + 1] = sv.y;
#if RACE_VARIANT == 7
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#endif
#endif
            }
        }
#if RACE_VARIANT == 5
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NKT; ++t) asm volatile("" : "+v"(acc[t]));
#endif
        float m = -1.0e30f;
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            m = fmaxf(fmaxf(m, acc[t][0]), acc[t][1]);   // v_max3_f32
            m = fmaxf(fmaxf(m, acc[t][2]), acc[t][3]);
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const f32x2 nm = {-m * sc, -m * sc};
        f32x2 ls2 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2 e = (f32x2){acc[t][r], acc[t][r + 1]} * sc + nm;      // v_pk_fma_f32
                const f32x2 p = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
                acc[t][r] = p.x;
                acc[t][r + 1] = p.y;
                ls2 += p;                                                         // v_pk_add_f32
            }
        }
        float lsum = ls2.x + ls2.y;
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);

        // O^T = V^T P^T : A = V^T (two 16-wide head-dim tiles) via transposed LDS reads, B = P^T from the
        // accumulators: MFMA k-slot (g, j) carries key 32s + 4g + j (j < 4) / 32s + 16 + 4g + (j-4).
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
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                union {
                    fp16x4_t t[2];
                    f16x8 v;
                } vf;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int row = 32 * s + 16 * hh + 4 * g + tr_q;
                    const int half = dt ^ ((row >> 2) & 1);  // = dt ^ (g & 1): 32-byte halves swapped on odd row quads
                    vf.t[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                        (__attribute__((address_space(3))) fp16x4_t*)(v_lds + row * Cfg::ROWB + half * 32 + tr_p * 8));
                }
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf.v, pf.h, o[dt], 0, 0, 0);
            }
        }

        // store: lane owns query qn, head-dim d = 16*dt + 4g + (0..3).  Lane pairs (g, g^1) swap one 8-byte half so
        // that every lane holds 8 CONSECUTIVE channels and the four lanes of a token write its whole 64-byte head
        // slice with one 16-byte store each (two 8-byte stores per lane left 32-byte fragments in every line).
        {
            const float inv = 1.f / lsum;
            union { f16x4 h; unsigned u[2]; } o0, o1, snd, rcv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o0.h[r] = (f16)(o[0][r] * inv);
                o1.h[r] = (f16)(o[1][r] * inv);
            }
            const bool odd = g & 1;
            snd.u[0] = odd ? o0.u[0] : o1.u[0];
            snd.u[1] = odd ? o0.u[1] : o1.u[1];
            rcv.u[0] = __shfl_xor(snd.u[0], 16, 64);
            rcv.u[1] = __shfl_xor(snd.u[1], 16, 64);
            if (item_ok && qn < WS * WS) {
                const long pix = img_pix + (long)(wy * WS + qy) * a.Wp + (wx * WS + qx);
                // even g: channels 4g .. 4g+7 = own dt0 half + partner's dt0 half; odd g: 16+4(g-1) .. = partner's dt1 + own dt1
                f16* dst = a.out + pix * a.C + head * 32 + (odd ? 16 + 4 * (g - 1) : 4 * g);
                uint4 v;
                v.x = odd ? rcv.u[0] : o0.u[0];
                v.y = odd ? rcv.u[1] : o0.u[1];
                v.z = odd ? o1.u[0] : rcv.u[0];
                v.w = odd ? o1.u[1] : rcv.u[1];
                *reinterpret_cast<uint4*>(dst) = v;
            }
        }
    }
}

template <int WS, int P, int LOG2P, int WPI>
__global__ __launch_bounds__(256, 2) void window_attn_kernel(WinArgs a) {
    using Cfg = WinCfg<WS, P, LOG2P, WPI>;
    constexpr int RP = Cfg::RP, NKT = Cfg::NKT, NQT = Cfg::NQT, QPW = Cfg::QPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int slot = wave / WPI;
    const int lt = tid - slot * WPI * 64;
    // XCD-aware block order: consecutive logical blocks (the heads of one window: neighbouring 64-byte slices of the
    // same 128-byte lines) run on the same XCD and share its L2, instead of each XCD fetching the line for itself
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int item = bid * Cfg::IPW + slot;
    const bool item_ok = item < a.items;
    if (!item_ok) item = a.items - 1;
    const int head = item % a.heads;
    int wq = item / a.heads;
    const int wx = wq % a.nWx;
    wq /= a.nWx;
    const int wy = wq % a.nWy;
    const int b = wq / a.nWy;

    char* base = smem + slot * Cfg::ITEM_BYTES;
    char* k_lds = base + Cfg::K_OFF;
    char* v_lds = base + Cfg::V_OFF;
    float* tb = reinterpret_cast<float*>(base + Cfg::T_OFF);

    const int ldq = a.nq * a.C;
    const long img_pix = (long)b * a.Hp * a.Wp;
    const int l15 = lane & 15, g = lane >> 4;
    const int wi = wave % WPI;

    // ---- Q fragments of this wave's query tiles, straight from global (issued first: longest latency) ----
    U4H8 qf[QPW];
#pragma unroll
    for (int i = 0; i < QPW; ++i) {
        const int qn = (wi + i * WPI) * 16 + l15;  // dense token index ty*WS + tx
        const int qy = qn / WS, qx = qn - qy * WS;
        const bool valid = qn < WS * WS;
        const long pix = img_pix + (long)(wy * WS + qy) * a.Wp + (wx * WS + qx);
        const f16* src = (a.nq == 2) ? a.qg + ((long)b * WS * WS + qy * WS + qx) * a.C + head * 32 + g * 8
                                     : a.qkv + pix * ldq + head * 32 + g * 8;
        src = valid ? src : a.qkv;
        const uint4 v = *reinterpret_cast<const uint4*>(src);
        qf[i].u = valid ? v : make_uint4(0, 0, 0, 0);
    }

    // ---- stage K / V (re-indexed, zero-padded, swizzled) and the head's bias table.  All global loads
    // are UNCONDITIONAL (masked lanes read a safe address and are zeroed afterwards) and issued back to
    // back before the first LDS write: a load inside an `if` costs a full memory round trip each. ----
    constexpr int NSLOT = 2 * RP * 4, NTHR = WPI * 64, NIT = (NSLOT + NTHR - 1) / NTHR;
    {
        uint4 st[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int s = lt + it * NTHR;
            const int arr = s / (RP * 4);  // 0 = K, 1 = V
            const int rem = s - arr * (RP * 4);
            const int row = rem >> 2, ch = rem & 3;
            const int ty = row >> LOG2P, tx = row & (P - 1);
            const bool valid = (s < NSLOT) & (ty < WS) & (tx < WS);
            const long pix = img_pix + (long)(wy * WS + ty) * a.Wp + (wx * WS + tx);
            const f16* src = a.qkv + pix * ldq + (a.nq - 2 + arr) * a.C + head * 32 + ch * 8;
            src = valid ? src : a.qkv;
            const uint4 v = *reinterpret_cast<const uint4*>(src);
            st[it] = valid ? v : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int s = lt + it * NTHR;
            const int arr = s / (RP * 4);
            const int rem = s - arr * (RP * 4);
            const int row = rem >> 2, ch = rem & 3;
            if (s < NSLOT) {
                const int pch = (arr == 0) ? k_slot(row, ch) : (ch ^ (((row >> 2) & 1) << 1));
                *reinterpret_cast<uint4*>((arr == 0 ? k_lds : v_lds) + row * Cfg::ROWB + pch * 16) = st[it];
            }
        }
    }
    {
        constexpr int TIT = (Cfg::TB_FLOATS + NTHR - 1) / NTHR;
        float tv[TIT];
#pragma unroll
        for (int it = 0; it < TIT; ++it) {
            const int i = lt + it * NTHR;
            const int e = i - Cfg::TOFF;
            const int ry = e / Cfg::TW, rx = e - ry * Cfg::TW;
            const bool in_tab = (i >= Cfg::TOFF) & (i < Cfg::TB_FLOATS) & (rx < Cfg::TROWS);
            const int gi = in_tab ? (ry * Cfg::TROWS + rx) * a.heads + head : 0;
            const float t = a.table[gi];
            tv[it] = in_tab ? t * a.inv_scale : (i < Cfg::TOFF ? -1.0e30f : 0.f);
        }
#pragma unroll
        for (int it = 0; it < TIT; ++it) {
            const int i = lt + it * NTHR;
            if (i < Cfg::TB_FLOATS) tb[i] = tv[it];
        }
    }
    __syncthreads();

#pragma unroll
    for (int i = 0; i < QPW; ++i) {
        const int qt = wi + i * WPI;
        if (qt >= NQT) break;  // wave-uniform
        win_query_tile<WS, P, LOG2P, WPI>(a, qf[i], k_lds, v_lds, tb, qt, l15, g, item_ok, img_pix, wy, wx, head);
    }
}

// ---- ws = 14, persistent + pipelined ----------------------------------------------------------------------------------
// The kernel above is a load phase followed by a compute phase per workgroup; with HBM time (~18 us for the 103 MB of level 2
// at B = 256) about equal to the VALU + MFMA time (~20 us) and every resident workgroup in the same phase, the two add up
// (42 us).  Here a workgroup walks several (window, head) items of ONE head (bias table staged once) and the K / V images of
// item k+1 are written into a second LDS buffer by LDS-DMA (buffer_load_dwordx4 ... lds: no registers, no ds_write pass) while
// item k is being computed, so fetch and math of the same workgroup overlap.  The re-indexing (row' = 16 ty + tx, two
// zero-padded slots per token row) and both swizzles are applied on the SOURCE side: the DMA fills LDS lane-linearly
// (16 rows x 64 B per wave instruction = one token row), lane (tx, physical chunk) fetches the logical chunk that belongs
// there, and padded slots use an out-of-range offset (the buffer unit writes zeros).  The DMA is issued from inline asm:
// hipcc makes every compiler-visible LDS read wait for ALL outstanding LDS-DMA (vmcnt(0)), which would serialise the pipeline;
// the waits are placed by hand (s_waitcnt vmcnt(3): only this wave's newest output stores may still be in flight).
typedef int i32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void win_dma16(unsigned lds_wave_base, unsigned voff, i32x4_t rsrc) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_wave_base), "v"(voff), "s"(rsrc)
                 : "memory");
}

template <int WS, int P, int LOG2P>
__global__ __launch_bounds__(256, 2) void window_attn_pipe_kernel(WinArgs a) {
    using Cfg = WinCfg<WS, P, LOG2P, 4>;
    constexpr int RP = Cfg::RP, NQT = Cfg::NQT, QPW = Cfg::QPW;
    constexpr int KV_BYTES = 2 * RP * Cfg::ROWB;                 // K image then V image
    constexpr int NDMA = 2 * (RP / 16);                          // wave instructions per item (16 rows each): 28
    constexpr int DPW = NDMA / 4;                                // per wave: 7
    static_assert(NDMA % 4 == 0 && P == 16, "one token row per DMA instruction");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tb = reinterpret_cast<float*>(smem + 2 * KV_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    int lb = blockIdx.x;
    const int nblk = gridDim.x;
    {   // XCD-aware order: consecutive logical blocks = the heads of one window (neighbouring 64-byte slices of the same lines)
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = lb & 7, idx = lb >> 3;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int head = lb % a.heads;                               // the same for every item of this block (nblk % heads == 0)
    const int ldq = a.nq * a.C;

    // descriptor over the whole qkv tensor (wave-uniform words, built from kernel arguments only)
    const unsigned long qp = (unsigned long)a.qkv;
    const i32x4_t rq = {(int)(unsigned)qp, (int)((qp >> 32) & 0xffffu), (int)(unsigned)(2L * a.B * a.Hp * a.Wp * ldq), 0x00020000};

    // this lane's part of the DMA plan: instruction j = wave + 4 i  ->  (array = j / 14, token row ty = j % 14), lane = (tx, chunk)
    unsigned rel[DPW];
    unsigned dlds[DPW];
    {
        const int tx = lane >> 2, pc = lane & 3;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int j = wave + 4 * i;
            const int arr = j / (RP / 16), ty = j - arr * (RP / 16);
            const int row = ty * 16 + tx;
            const int ch = arr == 0 ? k_slot(row, pc) : (pc ^ (((row >> 2) & 1) << 1));     // both swizzles are involutions
            const bool valid = (tx < WS) & (ty < WS);
            rel[i] = valid ? (unsigned)(((ty * a.Wp + tx) * ldq + (a.nq - 2 + arr) * a.C + head * 32 + ch * 8) * 2) : OOB;
            dlds[i] = (unsigned)(arr * RP * Cfg::ROWB + ty * 1024);
        }
    }
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;

    struct Item { int b, wy, wx; long img_pix; };
    auto decode = [&](int item) {
        Item it;
        int wq = item / a.heads;
        it.wx = wq % a.nWx;
        wq /= a.nWx;
        it.wy = wq % a.nWy;
        it.b = wq / a.nWy;
        it.img_pix = (long)it.b * a.Hp * a.Wp;
        return it;
    };
    auto issue_dma = [&](const Item& it, int buf) {
        const unsigned base = (unsigned)(((it.img_pix + (long)(it.wy * WS) * a.Wp + it.wx * WS) * ldq) * 2);
#pragma unroll
        for (int i = 0; i < DPW; ++i)
            win_dma16(lds0 + (unsigned)(buf * KV_BYTES) + dlds[i], rel[i] == OOB ? OOB : base + rel[i], rq);
    };
    // Q fragments go global -> VGPR through asm loads: a load hipcc can see would make it place its own s_waitcnt in front of
    // the first use, and - blind to the asm DMA issued in between - that wait would drain the next item's DMA as well.  Masked
    // lanes use an out-of-range offset (the buffer unit returns zeros).  Destinations are named in the wait statements below.
    const unsigned long gp = (unsigned long)(a.nq == 2 ? a.qg : a.qkv);
    const unsigned gbytes = (unsigned)(a.nq == 2 ? 2L * a.B * WS * WS * a.C : 2L * a.B * a.Hp * a.Wp * ldq);
    const i32x4_t rqq = {(int)(unsigned)gp, (int)((gp >> 32) & 0xffffu), (int)gbytes, 0x00020000};
    auto load_q = [&](const Item& it, f16x8 (&q)[QPW]) {
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const int qn = (wave + i * 4) * 16 + l15;            // dense token index ty*WS + tx
            const int qy = qn / WS, qx = qn - qy * WS;
            const long pix = it.img_pix + (long)(it.wy * WS + qy) * a.Wp + (it.wx * WS + qx);
            const long e = (a.nq == 2) ? ((long)it.b * WS * WS + qy * WS + qx) * a.C + head * 32 + g * 8 : pix * ldq + head * 32 + g * 8;
            const unsigned off = qn < WS * WS ? (unsigned)(e * 2) : OOB;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(q[i]) : "v"(off), "s"(rqq) : "memory");
        }
    };

    int item = lb;                                               // block-uniform; nblk <= items, so the first item exists
    Item cur = decode(item);
    static_assert(QPW == 4, "the wait statements name four Q fragments");
    f16x8 qf[QPW], qn[QPW];
    issue_dma(cur, 0);
    load_q(cur, qf);
    {   // the head's bias table, once per workgroup (same image as in the kernel above)
        constexpr int TIT = (Cfg::TB_FLOATS + 255) / 256;
        float tv[TIT];
#pragma unroll
        for (int it = 0; it < TIT; ++it) {
            const int i = tid + it * 256;
            const int e = i - Cfg::TOFF;
            const int ry = e / Cfg::TW, rx = e - ry * Cfg::TW;
            const bool in_tab = (i >= Cfg::TOFF) & (i < Cfg::TB_FLOATS) & (rx < Cfg::TROWS);
            const int gi = in_tab ? (ry * Cfg::TROWS + rx) * a.heads + head : 0;
            const float t = a.table[gi];
            tv[it] = in_tab ? t * a.inv_scale : (i < Cfg::TOFF ? -1.0e30f : 0.f);
        }
#pragma unroll
        for (int it = 0; it < TIT; ++it) {
            const int i = tid + it * 256;
            if (i < Cfg::TB_FLOATS) tb[i] = tv[it];
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3])::"memory");
    __builtin_amdgcn_s_barrier();

    int buf = 0;
    while (true) {
        const int nxt_item = item + nblk;
        const bool has_next = nxt_item < a.items;                // block-uniform
        Item nxt = cur;
        if (has_next) {
            nxt = decode(nxt_item);
            issue_dma(nxt, buf ^ 1);                             // the other buffer was last read before the previous barrier
            load_q(nxt, qn);
        }
        const char* k_lds = smem + buf * KV_BYTES;
        const char* v_lds = k_lds + RP * Cfg::ROWB;
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const int qt = wave + i * 4;
            if (qt >= NQT) break;  // wave-uniform
            U4H8 qv;
            qv.h = qf[i];
            win_query_tile<WS, P, LOG2P, 4>(a, qv, k_lds, v_lds, tb, qt, l15, g, true, cur.img_pix, cur.wy, cur.wx, head);
        }
        if (!has_next) break;
        // everything older than this wave's (at most QPW) newest output stores has retired: the DMA and Q loads of the next item
        asm volatile("s_waitcnt vmcnt(3)" : "+v"(qn[0]), "+v"(qn[1]), "+v"(qn[2]), "+v"(qn[3])::"memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < QPW; ++i) qf[i] = qn[i];
        item = nxt_item;
        cur = nxt;
        buf ^= 1;
    }
}

template <int WS, int P, int LOG2P>
int launch_win_pipe(const WinArgs& a, hipStream_t s) {
    using Cfg = WinCfg<WS, P, LOG2P, 4>;
    constexpr int SMEM = 2 * 2 * Cfg::RP * Cfg::ROWB + (Cfg::TB_FLOATS * 4 + 15) / 16 * 16;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_pipe_kernel<WS, P, LOG2P>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        attr_set = true;
    }
    int wgs = 512;                                              // two 66 KB workgroups per CU
    if (wgs > a.items) wgs = a.items;
    wgs -= wgs % a.heads;                                       // a workgroup keeps one head: items b, b + wgs, ... share it
    hipLaunchKernelGGL((window_attn_pipe_kernel<WS, P, LOG2P>), dim3(wgs), dim3(256), SMEM, s, a);
    return vip_launch_status("vip_window_attn_fwd_f16(pipe)");
}

template <int WS, int P, int LOG2P, int WPI>
int launch_win(const WinArgs& a, hipStream_t s) {
    using Cfg = WinCfg<WS, P, LOG2P, WPI>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_kernel<WS, P, LOG2P, WPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        attr_set = true;
    }
    const int wgs = (a.items + Cfg::IPW - 1) / Cfg::IPW;
    hipLaunchKernelGGL((window_attn_kernel<WS, P, LOG2P, WPI>), dim3(wgs), dim3(256), Cfg::SMEM, s, a);
    return vip_launch_status("vip_window_attn_fwd_f16");
}

}  // namespace

extern "C" int vip_window_attn_fwd_f16(const void* qkv, const void* q_global, const float* bias_table, void* out,
                                       int B, int Hp, int Wp, int C, int heads, int ws, int nq, float scale,
                                       void* stream) {
    VIP_REQUIRE(qkv && bias_table && out, VIP_ERR_BAD_ARG, "vip_window_attn_fwd_f16: null pointer");
    VIP_REQUIRE(nq == 3 || (nq == 2 && q_global), VIP_ERR_BAD_ARG,
                "vip_window_attn_fwd_f16: nq must be 3, or 2 with a q_global tensor");
    VIP_REQUIRE(B > 0 && Hp > 0 && Wp > 0 && C > 0 && heads > 0 && scale > 0.f, VIP_ERR_BAD_ARG,
                "vip_window_attn_fwd_f16: non-positive dimension or scale");
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
    a.inv_scale = 1.f / scale;
    if (ws == 7) return launch_win<7, 8, 3, 1>(a, (hipStream_t)stream);
    // ws 14: the pipelined persistent kernel is opt-in (VIP_ATTN_PIPE=1, read per call): measured 44.6 us against 42.5 us for the
    // one-item kernel at B = 256 - the kernel is VALU-bound (softmax), not fetch-bound, and 4 resident workgroups per CU hide the
    // per-wave MFMA -> VALU -> MFMA chains better than 2 pipelined ones
    const char* pipe_env = getenv("VIP_ATTN_PIPE");
    const int pipe = pipe_env ? atoi(pipe_env) : 0;
    if (pipe && items >= 1024 && items % heads == 0 && 2L * B * Hp * Wp * nq * C < 0x7FFF0000L)
        return launch_win_pipe<14, 16, 4>(a, (hipStream_t)stream);
    return launch_win<14, 16, 4, 4>(a, (hipStream_t)stream);
}
