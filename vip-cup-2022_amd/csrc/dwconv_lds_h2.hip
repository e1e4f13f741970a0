// Depthwise k x k stride-1 convolution on the packed STRICT storage, staged through LDS (k = 3 / 5 / 7).
//
// Replaces the Keras DepthwiseConv2D call sites of tfimm's ConvNeXtBlock (convnext.py:200-229: 7x7), keras_cv_attention_models'
// MBConv / FusedMBConv blocks (efficientnet.py: 3x3 / 5x5 + swish) and GCViT's MBConv-style stem blocks in strict mode.
//
// Why not the register-tiled kernel of dwconv.hip (lanes = channel quads, patch rows straight from global memory): on the packed storage
// a lane's 4 channels are TWO 8-byte runs (hi quad, lo quad) of a pixel and every patch pixel is fetched by ~10 overlapping tiles, so the
// kernel issued 5 vector-memory instructions per output value and the texture-address path, not HBM, set its rate (1.5 TB/s on ConvNeXt's
// 7x7 layers); the (hi, lo) -> fp32 join was repeated per fetch as well.  Here:
//   * a workgroup (4 waves) owns 16 channels (64 contiguous bytes of every pixel) of a block of output pixels; the input patch
//     (block + halo) is read ONCE with 16-byte loads, joined to fp32 and kept in LDS as [channel quad][row][column] float4;
//   * wave = channel quad, lane = a 2 x 4-pixel register tile: the filter taps are wave-uniform and come in as SCALAR loads (SGPR
//     operands of v_pk_fma_f32) - no vector or LDS traffic for the weights; a patch row is 4 + k - 1 ds_read_b128 per lane.  The filter
//     is taken QUAD-MAJOR, [C/4][k*k][4] fp32 (vip_dw_filter_quad_major builds it): a wave's k*k taps are 16 k k contiguous bytes, a dozen
//     lines of the 16 KB scalar cache - with the [k*k][C] layout every tap was its own line and the 7x7 layers missed on each
//     (2 500 stall cycles per patch row measured);
//   * results go back through LDS (split to hi / lo, [quad][row][column] 16-byte slots) so that the global stores are 16-byte runs of a
//     pixel's 64-byte slice, coalesced like the loads;
//   * the NEXT block's patch is fetched global -> VGPR before the current block's math and written to LDS after its stores: loads stay in
//     flight across the compute phase (two workgroups per CU).
// Small maps: a wave packs several sub-regions (e.g. 8 images of a 7 x 7 map, 2 of 14 x 14): the host picks (IMG, LTY, LTX) per shape.
#include <type_traits>
#include "common.hpp"

namespace {

struct DwLdsArgs {
    const char* x;
    const float* w;
    const float* bias;
    char* y;
    int B, H, W, C, pt, pl, Ho, Wo, act;
    int IMG, LTY, LTX;        // lanes of a wave: IMG sub-regions x LTY x LTX register tiles (2 rows x 4 columns of output each)
    int RGY, RGX;             // sub-regions per image
    int PH, PWID, PWP;        // patch rows, columns, padded columns (row stride in 16-byte slots)
    int RH, RW, OWP;          // output rows, columns of a sub-region, padded columns of its LDS image
    int n_cblk;               // 16-channel blocks
    unsigned m_prow, m_ph, m_orow, m_oh;     // ceil(2^32 / d) for the small divisions below
    int n_sub;                // B * RGY * RGX sub-regions
    long n_items;             // ceil(n_sub / IMG) * n_cblk
    int* status;
    float* partials;          // optional [n_sub][C] fp32: per sub-region sums of the outputs (the squeeze-excite pool, vip_se_gate_pooled_h2)
    int dbg;                  // experiments (VIP_DW_LDS_DBG): 1 = no math, 2 = no global stores, 4 = no patch loads
};

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned magic) { return __umulhi(n, magic); }      // n, d < 65536 (d > 1)

// the two fp32 values hi + lo of a packed pair of (hi, lo) halfs: v_fma_mix_f32 (f16 x 1.0f + f16 -> f32, one rounding - the same value as
// two conversions and an add, one instruction instead of three)
__device__ __forceinline__ void h2_join2(unsigned hv, unsigned lv, float& e0, float& e1) {
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(e0) : "v"(hv), "v"(lv));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(e1) : "v"(hv), "v"(lv));
}

constexpr int ST_U = 8;       // staged (pixel, group) items per thread held in registers (the launcher guarantees the patch fits)
constexpr int SO_U = 4;       // stored (pixel, group) items per thread (64 lanes x 8 pixels x 2 groups / 256)

template <int K>
__global__ __launch_bounds__(256, 2) void dwconv_lds_h2_kernel(DwLdsArgs a, const float* __restrict__ wts, const float* __restrict__ bias) {
    constexpr int T = 2, TW = 4, P = T + K - 1, PW = TW + K - 1;
    constexpr unsigned OOB = 0xFFFFFFE0u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // per sub-region of a region group (two groups: the current item's and the next one's): the origin of its patch (row, column, byte
    // offset mod 2^32, valid) and of its output block
    __shared__ int4 meta_in[2][8], meta_out[2][8];

    const int tid = threadIdx.x, lane = tid & 63;
    const int quad = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)(4L * a.B * a.H * a.W * a.C), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)(4L * a.B * a.Ho * a.Wo * a.C), 0x00020000);

    // lane -> (sub-region, tile row, tile column); idle lanes shadow lane 0's addresses and store nothing
    const int per = a.LTY * a.LTX;
    const bool lane_ok = lane < a.IMG * per;
    int im = 0, ty = 0, tx = 0;
    if (lane_ok) {
        im = lane / per;
        const int rem = lane - im * per;
        ty = rem / a.LTX;
        tx = rem - ty * a.LTX;
    }

    // ---- item-independent descriptors: what this thread stages (patch) and stores (output block) for EVERY item ----
    const int n_st = a.IMG * a.PH * a.PWID * 2;      // (sub-region, row, column, 8-channel group) of a patch
    unsigned st_pos[ST_U], st_goff[ST_U], st_lds[ST_U];     // s | g << 3 | ok << 4 | py << 8 | px << 20; byte offset from the origin; LDS byte
    const unsigned qs_in = (unsigned)(a.IMG * a.PH * a.PWP) * 16u;
#pragma unroll
    for (int u = 0; u < ST_U; ++u) {
        const int idx = tid + u * 256;
        const bool ok = idx < n_st;
        const unsigned i = ok ? (unsigned)idx : 0u;
        const unsigned row = fdiv(i, a.m_prow), col2 = i - row * (unsigned)(2 * a.PWID);
        const unsigned g = col2 & 1u, px = col2 >> 1;
        const unsigned sr = fdiv(row, a.m_ph), py = row - sr * (unsigned)a.PH;
        st_pos[u] = sr | (g << 3) | ((ok ? 1u : 0u) << 4) | (py << 8) | (px << 20);
        st_goff[u] = (py * (unsigned)a.W + px) * (unsigned)a.C * 4u + g * 32u;
        st_lds[u] = ((((2 * g) * a.IMG + sr) * a.PH + py) * a.PWP + px + (px >> 2)) * 16u;
    }
    const int n_so = a.IMG * a.RH * a.RW * 2;
    unsigned so_pos[SO_U], so_goff[SO_U], so_lds[SO_U];
    const unsigned qs_out = (unsigned)(a.IMG * a.RH * a.OWP) * 16u;
#pragma unroll
    for (int v = 0; v < SO_U; ++v) {
        const int idx = tid + v * 256;
        const bool ok = idx < n_so;
        const unsigned i = ok ? (unsigned)idx : 0u;
        const unsigned row = fdiv(i, a.m_orow), col2 = i - row * (unsigned)(2 * a.RW);
        const unsigned g = col2 & 1u, ox = col2 >> 1;
        const unsigned sr = fdiv(row, a.m_oh), oy = row - sr * (unsigned)a.RH;
        so_pos[v] = sr | (g << 3) | ((ok ? 1u : 0u) << 4) | (oy << 8) | (ox << 20);
        so_goff[v] = (oy * (unsigned)a.Wo + ox) * (unsigned)a.C * 4u + g * 32u;
        so_lds[v] = ((((2 * g) * a.IMG + sr) * a.RH + oy) * a.OWP + ox + (ox >> 2)) * 16u;
    }

    // a contiguous run of items per workgroup, the 16-channel blocks of one region group back to back: the descriptors of the group are
    // computed once and the two 64-byte halves of a pixel's 128-byte line are read by the same workgroup within microseconds (L2 hits)
    // (workgroups are dealt round-robin to the 8 XCDs: renumbered so that each XCD owns one contiguous band of runs - neighbouring
    // regions, which share halo rows, then meet in ONE L2; PMC before: 2.15 bytes fetched from HBM per byte written)
    int bid = blockIdx.x;
    if (gridDim.x >= 8) {
        const int q = gridDim.x >> 3, r = gridDim.x & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long it0 = a.n_items * bid / gridDim.x, it1 = a.n_items * (bid + 1) / gridDim.x;
    int grp_cur = (int)(it0 / a.n_cblk), cblk = (int)(it0 - (long)grp_cur * a.n_cblk);

    auto group_meta = [&](int group, int slot) {     // threads 0 .. IMG-1 describe the group's sub-regions
        if (tid < a.IMG) {
            const unsigned sr = (unsigned)group * (unsigned)a.IMG + (unsigned)tid;
            int4 mi = make_int4(0, 0, 0, 0), mo = make_int4(0, 0, 0, 0);
            if (sr < (unsigned)a.n_sub) {
                const unsigned per_img = (unsigned)(a.RGY * a.RGX);
                const unsigned b = sr / per_img, r = sr - b * per_img;
                const unsigned rgy = r / (unsigned)a.RGX;
                const int oy0 = (int)rgy * a.RH, ox0 = (int)(r - rgy * a.RGX) * a.RW;
                mi = make_int4(oy0 - a.pt, ox0 - a.pl, (int)(unsigned)((((long)b * a.H + (oy0 - a.pt)) * a.W + (ox0 - a.pl)) * a.C * 4), 1);
                mo = make_int4(oy0, ox0, (int)(unsigned)((((long)b * a.Ho + oy0) * a.Wo + ox0) * a.C * 4), 1);
            }
            meta_in[slot][tid] = mi;
            meta_out[slot][tid] = mo;
        }
    };
    uint4 sh[ST_U], sl[ST_U];
    // global -> VGPR: the patch of (group in `slot`, channel block cb)
    auto stage_load = [&](int slot, int cb, bool live) {
        const unsigned cboff = (unsigned)cb * 64u;
        const bool g1_ok = cb * 16 + 8 < a.C;
#pragma unroll
        for (int u = 0; u < ST_U; ++u) {
            const unsigned pos = st_pos[u];
            const int4 m = meta_in[slot][pos & 7u];
            const int gy = m.x + (int)((pos >> 8) & 0xFFFu), gx = m.y + (int)(pos >> 20);
            const bool ok = live && (pos & 16u) && m.w && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W && (!(pos & 8u) || g1_ok) &&
                            !(a.dbg & 4);
            const unsigned off = ok ? (unsigned)m.z + st_goff[u] + cboff : OOB;
            sh[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
            sl[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, off + 16u, 0, 0));
        }
    };
    // VGPR -> LDS: join (hi, lo) to fp32, [quad][sub-region][row][padded column] float4
    auto stage_store = [&]() {
#pragma unroll
        for (int u = 0; u < ST_U; ++u) {
            if (st_pos[u] & 16u) {
                float4 v0, v1;
                h2_join2(sh[u].x, sl[u].x, v0.x, v0.y);
                h2_join2(sh[u].y, sl[u].y, v0.z, v0.w);
                h2_join2(sh[u].z, sl[u].z, v1.x, v1.y);
                h2_join2(sh[u].w, sl[u].w, v1.z, v1.w);
                *reinterpret_cast<float4*>(smem + st_lds[u]) = v0;
                *reinterpret_cast<float4*>(smem + st_lds[u] + qs_in) = v1;
            }
        }
    };

    float ov_max = 0.f, ov_sum = 0.f;                // fp16-range check of the outputs: max |v| and a NaN-propagating sum
    int cur = 0;
    if (it0 < it1) {
        group_meta(grp_cur, 0);
        __syncthreads();
        stage_load(0, cblk, true);
        stage_store();
    }
    for (long it = it0; it < it1; ++it) {
        int cb_n = cblk + 1, grp_n = grp_cur, slot_n = cur;
        if (cb_n == a.n_cblk) {
            cb_n = 0;
            grp_n = grp_cur + 1;
            slot_n = cur ^ 1;
            if (it + 1 < it1) group_meta(grp_n, slot_n);
        }
        __syncthreads();                             // the patch of `it` is in LDS, the next group's descriptors are visible
        stage_load(slot_n, cb_n, it + 1 < it1);      // in flight over the math below

        const int c0 = cblk * 16 + quad * 4;         // wave-uniform
        const bool quad_ok = c0 < a.C;
        f32x2 acc[T][TW][2];
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < TW; ++j) acc[i][j][0] = acc[i][j][1] = (f32x2){0.f, 0.f};
        if (quad_ok && !(a.dbg & 1)) {
            const char* prow = smem + (size_t)(((quad * a.IMG + im) * a.PH + ty * T) * a.PWP + tx * 5) * 16;
            const float* wq = wts + (long)(c0 >> 2) * (K * K * 4);     // quad-major filter: this wave's k * k taps are 16 k k contiguous bytes
            // one patch row per trip (not unrolled: an unrolled body hoists all k * k scalar tap loads and spills SGPRs); the row feeds
            // output row oy with filter row r = iy - oy, a wave-uniform test
#pragma unroll 1
            for (int iy = 0; iy < P; ++iy) {
                f32x2 xr[PW][2];
#pragma unroll
                for (int q = 0; q < PW; ++q) {
                    const float4 t = *reinterpret_cast<const float4*>(prow + (size_t)iy * a.PWP * 16 + (q + (q >> 2)) * 16);
                    xr[q][0] = (f32x2){t.x, t.y};
                    xr[q][1] = (f32x2){t.z, t.w};
                }
#pragma unroll
                for (int oy = 0; oy < T; ++oy) {
                    const int r = iy - oy;
                    if (r < 0 || r >= K) continue;
#pragma unroll
                    for (int s = 0; s < K; ++s) {
                        const float4 wv = *reinterpret_cast<const float4*>(wq + (r * K + s) * 4);      // scalar load
                        const f32x2 w0 = {wv.x, wv.y}, w1 = {wv.z, wv.w};
#pragma unroll
                        for (int ox = 0; ox < TW; ++ox) {
                            acc[oy][ox][0] = xr[ox + s][0] * w0 + acc[oy][ox][0];
                            acc[oy][ox][1] = xr[ox + s][1] * w1 + acc[oy][ox][1];
                        }
                    }
                }
            }
        }
        __syncthreads();                             // every wave is done with the patch: the output image may overwrite it
        if (quad_ok) {
            const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            char* orow = smem + (size_t)(((quad * a.IMG + im) * a.RH + ty * T) * a.OWP + tx * 5) * 16;
            // pooled form: this lane's sum over its pixels that exist (the tile may overhang the map)
            const int4 mo = meta_out[cur][im];
            float psum[4] = {0.f, 0.f, 0.f, 0.f};
            auto epi = [&](auto atag) {
                constexpr int ACT = decltype(atag)::value;
#pragma unroll
                for (int oy = 0; oy < T; ++oy)
#pragma unroll
                    for (int ox = 0; ox < TW; ++ox) {
                        float v[4] = {acc[oy][ox][0][0] + bv.x, acc[oy][ox][0][1] + bv.y, acc[oy][ox][1][0] + bv.z, acc[oy][ox][1][1] + bv.w};
                        const bool px_ok = lane_ok && mo.w && mo.x + ty * T + oy < a.Ho && mo.y + tx * TW + ox < a.Wo;
                        union {
                            uint4 u;
                            f16x4 q[2];
                        } o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = vip_act_strict(v[e], ACT);
                            o.q[0][e] = (f16)v[e];
                            o.q[1][e] = (f16)(v[e] - (float)o.q[0][e]);      // exact difference, one rounding (v_fma_mixlo_f16 when the compiler folds it)
                            ov_max = fmaxf(ov_max, fabsf(v[e]));
                            ov_sum += v[e];
                            psum[e] += px_ok ? v[e] : 0.f;
                        }
                        *reinterpret_cast<uint4*>(orow + (size_t)oy * a.OWP * 16 + ox * 16) = o.u;
                    }
            };
            switch (a.act) {
                case VIP_ACT_RELU: epi(std::integral_constant<int, VIP_ACT_RELU>{}); break;
                case VIP_ACT_SILU: epi(std::integral_constant<int, VIP_ACT_SILU>{}); break;
                case VIP_ACT_GELU: epi(std::integral_constant<int, VIP_ACT_GELU>{}); break;
                case VIP_ACT_SIGMOID: epi(std::integral_constant<int, VIP_ACT_SIGMOID>{}); break;
                default: epi(std::integral_constant<int, VIP_ACT_NONE>{}); break;
            }
            if (a.partials) {      // workgroup-uniform.  Segmented sum over the lanes of a sub-region, fixed order (bit-reproducible):
                const int seg = lane_ok ? lane - im * per : 0;      // after the step of width d, lane i holds the sum of lanes [i, i + 2d) of its segment
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = __shfl_down(psum[e], d, 64);
                    if (lane_ok && seg + d < per) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) psum[e] += t[e];
                    }
                }
                if (lane_ok && seg == 0 && mo.w)
                    *reinterpret_cast<float4*>(a.partials + ((long)grp_cur * a.IMG + im) * a.C + c0) = make_float4(psum[0], psum[1], psum[2], psum[3]);
            }
        }
        __syncthreads();
        {   // LDS -> global: (sub-region, row, column, group) -> 32 bytes [hi x 8][lo x 8]
            const unsigned cboff = (unsigned)cblk * 64u;
            const bool g1_ok = cblk * 16 + 8 < a.C;
#pragma unroll
            for (int v = 0; v < SO_U; ++v) {
                const unsigned pos = so_pos[v];
                const int4 m = meta_out[cur][pos & 7u];
                const int gy = m.x + (int)((pos >> 8) & 0xFFFu), gx = m.y + (int)(pos >> 20);
                const bool ok = (pos & 16u) && m.w && gy < a.Ho && gx < a.Wo && (!(pos & 8u) || g1_ok) && !(a.dbg & 2);
                const uint4 qa = *reinterpret_cast<const uint4*>(smem + so_lds[v]);
                const uint4 qb = *reinterpret_cast<const uint4*>(smem + so_lds[v] + qs_out);
                const unsigned off = ok ? (unsigned)m.z + so_goff[v] + cboff : OOB;
                typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
                const u32x4 hi = {qa.x, qa.y, qb.x, qb.y}, lo = {qa.z, qa.w, qb.z, qb.w};
                __builtin_amdgcn_raw_buffer_store_b128(hi, ry, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(lo, ry, off + 16u, 0, 0);
            }
        }
        __syncthreads();                             // output image read: the next patch may land
        if (it + 1 < it1) stage_store();
        cblk = cb_n;
        grp_cur = grp_n;
        cur = slot_n;
    }
    if (lane_ok && a.status && (!(ov_max <= VIP_H2_MAX) || ov_sum != ov_sum)) *a.status = VIP_H2_OVERFLOW;
}

unsigned magic_of(int d) { return (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }

// (IMG, LTY, LTX) for a map: the most output pixels per lane slot, halo reads as the tie-break; the patch must fit ST_U staged items per
// thread and half a CU's LDS
bool pick_config(int B, int Ho, int Wo, int K, DwLdsArgs& a) {
    static const int lds_kb = getenv("VIP_DW_LDS_KB") ? atoi(getenv("VIP_DW_LDS_KB")) : 76;
    double best = -1.0;
    for (int lty = 1; lty <= 32; ++lty)
        for (int ltx = 1; ltx * lty <= 64 && ltx <= 16; ++ltx) {
            const int rh = 2 * lty, rw = 4 * ltx;
            const int rgy = (Ho + rh - 1) / rh, rgx = (Wo + rw - 1) / rw;
            const int ph = rh + K - 1, pwid = rw + K - 1, pwp = pwid + (pwid >> 2) + 1;
            const long n_sub = (long)B * rgy * rgx;
            if (n_sub >= (1L << 30) || 2 * pwid >= 65536) continue;
            int img = 64 / (lty * ltx);
            if (img > 8) img = 8;
            // the most sub-regions per wave that the staging registers and half a CU's LDS hold
            while (img > 0 && (img * ph * pwid * 2 > ST_U * 256 || (size_t)4 * img * ph * pwp * 16 > (size_t)lds_kb * 1024)) --img;
            if (img == 0) continue;
            const long groups = (n_sub + img - 1) / img;
            const double eff = (double)B * Ho * Wo / ((double)groups * 512.0);
            const double halo = (double)ph * pwid / ((double)rh * rw);
            const double score = eff / (0.75 + 0.25 * halo);
            if (score > best) {
                best = score;
                a.IMG = img; a.LTY = lty; a.LTX = ltx; a.RGY = rgy; a.RGX = rgx;
                a.PH = ph; a.PWID = pwid; a.PWP = pwp; a.RH = rh; a.RW = rw; a.OWP = rw + (rw >> 2) + 1;
                a.n_sub = (int)n_sub;
            }
        }
    return best > 0.0;
}

template <int K>
int launch_lds(DwLdsArgs a, hipStream_t s) {
    const size_t smem = (size_t)4 * a.IMG * a.PH * a.PWP * 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_lds_h2_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_set = true;
    }
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    long grid = 2L * n_cu;
    if (grid > a.n_items) grid = a.n_items;
    hipLaunchKernelGGL((dwconv_lds_h2_kernel<K>), dim3((unsigned)grid), dim3(256), smem, s, a, a.w, a.bias);
    return vip_launch_status("vip_dwconv2d_nhwc_h2(lds)");
}

__global__ void dw_filter_quad_major_kernel(const float* __restrict__ w, float* __restrict__ wq, int kk, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;           // over [C/4][kk][4]
    if (i >= kk * C) return;
    const int e = i & 3, tap = (i >> 2) % kk, q = (i >> 2) / kk;
    wq[i] = w[(long)tap * C + q * 4 + e];
}

}  // namespace

extern "C" int vip_dw_filter_quad_major(const float* w, float* w_quad, int k, int C, void* stream) {
    VIP_REQUIRE(w && w_quad, VIP_ERR_BAD_ARG, "vip_dw_filter_quad_major: null pointer");
    VIP_REQUIRE(k > 0 && C > 0 && C % 4 == 0, VIP_ERR_BAD_ARG, "vip_dw_filter_quad_major: k > 0, C a positive multiple of 4");
    const int n = k * k * C;
    hipLaunchKernelGGL(dw_filter_quad_major_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, w_quad, k * k, C);
    return vip_launch_status("vip_dw_filter_quad_major");
}

extern "C" int vip_dwconv2d_s1_supported_h2(int B, int H, int W, int C, int k, int Ho, int Wo) {
    if (k != 3 && k != 5 && k != 7) return 0;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || C % 8 != 0) return 0;
    if (4L * B * H * W * C >= 0xFFFFFFE0L || 4L * B * Ho * Wo * C >= 0xFFFFFFE0L) return 0;
    DwLdsArgs a;
    return pick_config(B, Ho, Wo, k, a) ? 1 : 0;
}

static int dwconv_s1_h2_impl(const void* x, const float* w_quad, const float* bias, void* y, float* partials, int B, int H, int W, int C, int k,
                             int pt, int pl, int Ho, int Wo, int act, int* status, void* stream);

extern "C" int vip_dwconv2d_s1_h2(const void* x, const float* w_quad, const float* bias, void* y, int B, int H, int W, int C, int k, int pt, int pl,
                                  int Ho, int Wo, int act, int* status, void* stream) {
    return dwconv_s1_h2_impl(x, w_quad, bias, y, nullptr, B, H, W, C, k, pt, pl, Ho, Wo, act, status, stream);
}

/* The pooling form: also writes partials[B][parts][C] (fp32), the sums of the (activated) outputs over `parts` =
 * vip_dwconv2d_s1_pool_parts_h2 blocks of each image - what vip_se_gate_pooled_h2 finishes the squeeze-excite mean from, so the gate
 * kernel does not read the map again.  Fixed summation order: bit-reproducible. */
extern "C" int vip_dwconv2d_s1_pool_parts_h2(int B, int H, int W, int C, int k, int Ho, int Wo) {
    if (!vip_dwconv2d_s1_supported_h2(B, H, W, C, k, Ho, Wo)) return 0;
    DwLdsArgs a;
    if (!pick_config(B, Ho, Wo, k, a)) return 0;
    return a.RGY * a.RGX;
}

extern "C" int vip_dwconv2d_s1_pool_h2(const void* x, const float* w_quad, const float* bias, void* y, float* partials, int parts, int B, int H,
                                       int W, int C, int k, int pt, int pl, int Ho, int Wo, int act, int* status, void* stream) {
    VIP_REQUIRE(partials, VIP_ERR_BAD_ARG, "vip_dwconv2d_s1_pool_h2: null partials");
    VIP_REQUIRE(parts > 0 && parts == vip_dwconv2d_s1_pool_parts_h2(B, H, W, C, k, Ho, Wo), VIP_ERR_BAD_ARG,
                "vip_dwconv2d_s1_pool_h2: partials sized for %d rows per image, the kernel writes %d", parts,
                vip_dwconv2d_s1_pool_parts_h2(B, H, W, C, k, Ho, Wo));
    return dwconv_s1_h2_impl(x, w_quad, bias, y, partials, B, H, W, C, k, pt, pl, Ho, Wo, act, status, stream);
}

static int dwconv_s1_h2_impl(const void* x, const float* w_quad, const float* bias, void* y, float* partials, int B, int H, int W, int C, int k,
                             int pt, int pl, int Ho, int Wo, int act, int* status, void* stream) {
    VIP_REQUIRE(x && w_quad && y, VIP_ERR_BAD_ARG, "vip_dwconv2d_s1_h2: null pointer");
    VIP_REQUIRE(pt >= 0 && pl >= 0 && (unsigned)act <= 4u, VIP_ERR_BAD_ARG, "vip_dwconv2d_s1_h2: negative padding or unknown activation");
    VIP_REQUIRE(vip_dwconv2d_s1_supported_h2(B, H, W, C, k, Ho, Wo), VIP_ERR_UNSUPPORTED,
                "vip_dwconv2d_s1_h2: k = 3 / 5 / 7, C %% 8 == 0, tensors below 4 GiB (B=%d H=%d W=%d C=%d k=%d); use vip_dwconv2d_nhwc_h2", B, H, W, C, k);
    VIP_REQUIRE((long)(Ho - 1) - pt < H && (long)(Wo - 1) - pl < W, VIP_ERR_BAD_ARG, "vip_dwconv2d_s1_h2: output extent does not match a stride-1 window");
    hipStream_t s = (hipStream_t)stream;
    const float* w = w_quad;
    DwLdsArgs a;
    a.x = (const char*)x; a.w = w; a.bias = bias; a.y = (char*)y;
    a.B = B; a.H = H; a.W = W; a.C = C; a.pt = pt; a.pl = pl; a.Ho = Ho; a.Wo = Wo; a.act = act;
    a.status = status;
    a.partials = partials;
    a.dbg = getenv("VIP_DW_LDS_DBG") ? atoi(getenv("VIP_DW_LDS_DBG")) : 0;
    if (!pick_config(B, Ho, Wo, k, a)) return VIP_ERR_UNSUPPORTED;
    a.n_cblk = (C + 15) / 16;
    a.n_items = (long)((a.n_sub + a.IMG - 1) / a.IMG) * a.n_cblk;
    a.m_prow = magic_of(2 * a.PWID);
    a.m_ph = magic_of(a.PH);
    a.m_orow = magic_of(2 * a.RW);
    a.m_oh = magic_of(a.RH);
    if (k == 3) return launch_lds<3>(a, s);
    if (k == 5) return launch_lds<5>(a, s);
    return launch_lds<7>(a, s);
}
