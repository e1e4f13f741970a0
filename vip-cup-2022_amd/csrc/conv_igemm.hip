// Implicit-GEMM NHWC convolution / dense layer on CDNA4 matrix cores (v_mfma_f32_16x16x32_f16).
//
//   M = B*Ho*Wo output pixels, N = Cout_g, K = kh*kw*Cin_g  (k = (r*kw + s)*Cin_g + c)
//
// Block = 256 threads = 4 waves, tile BM x BN x 64.  Both operands are staged global -> VGPR ->
// LDS in 16-byte chunks (8 halfs of one filter tap), double-buffered so the loads of k-tile t+1
// are in flight while tile t is multiplied (one barrier per k-tile).  Out-of-image taps and the
// M/N/K tails are zero-filled at staging time, so the MFMA loop is branch-free.
//
// Orientation: the WEIGHT fragment is the MFMA "A" operand and the ACTIVATION fragment the "B"
// operand, so each lane ends up holding 4 consecutive output channels of one pixel per MFMA; the
// weight rows a wave feeds to its 4 n-tiles are interleaved (row r of n-tile t = channel
// (t>>1)*32 + (r>>2)*8 + (t&1)*4 + (r&3)) so that one lane owns two runs of 8 consecutive channels of a
// pixel, the epilogue (bias, activation, residual) runs on 16-byte vectors, and the four lanes of a
// pixel write 64 contiguous bytes per store instruction.
//
// LDS image: 128-byte rows (64 halfs), 16-byte chunks XOR-swizzled so that every ds_read_b128 of
// a fragment is bank-conflict free: activation rows use key row&7, weight rows (read in the
// interleaved order above) use key ((row>>3)&3)<<1 | ((row>>1)&1).
//
// Replaces Conv2D+BN+Activation(+Add) of the reference (resnet_rs_model.py:64-84,235-280;
// kecam common_layers.py:190-248; tfimm convnext.py:260-267,320-327) and every Dense layer.
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

// VIP_GEMM_H2 = 1 (conv_h2.hip includes this file a second time): the same kernels for the packed STRICT storage of common.hpp -
// every operand element is an fp16 (hi, lo) pair, 8 channels = [hi x 8][lo x 8] = two 16-byte MFMA fragments.  Seen as halfs, a packed
// tensor is an fp16 tensor with twice the channels, so all staging / addressing / tiling below is shared: ConvArgs carries the
// INPUT side (ldx, cin_off, Cin_g, K, ldw) in halfs (doubled), a 64-half k-chunk is 32 logical k whose eight 16-byte chunks are
// (hi, lo) x 4 k-groups, and LDS images hold them PERMUTED (h2_pos: the four hi chunks first, then the four lo chunks) so that the
// fragment reads are the fp16 kernels' own - "k-step 0" reads the hi plane, "k-step 1" the lo plane.  What differs: three MFMAs per
// fragment pair (w_lo x_hi + w_hi x_hi in k-step 0, w_hi x_lo in k-step 1), direct global fragment loads take chunk 2 lq + plane, and
// the epilogues (fp32 activations, the weights' power-of-two scale undone, packed residual / output, fp16 range check).
#ifndef VIP_GEMM_H2
#define VIP_GEMM_H2 0
#endif

#ifndef VIP_MFMA_PRIO
#define VIP_MFMA_PRIO 1
#endif
#ifndef VIP_MFMA_PRIO_TILE
#define VIP_MFMA_PRIO_TILE 1
#endif

namespace {

constexpr bool H2 = VIP_GEMM_H2;
// LDS position of logical 16-byte chunk c (0..7) of a 64-half row
__device__ __forceinline__ int h2_pos(int c) { return H2 ? (((c & 1) << 2) | (c >> 1)) : c; }
// masked lanes of a packed (32-byte) access: base and base + 16 both stay out of range (spans are checked < 0xFFFFFFE0)
constexpr unsigned OOB2 = 0xFFFFFFE0u;
constexpr int ESZ = H2 ? 4 : 2;          // bytes per output / residual element

struct ConvArgs {
    const f16* x;
    const f16* w;
    const float* bias;
    const f16* res;
    f16* y;
    int H, W, Ho, Wo;
    int Cin_g, Cout_g;
    int kh, kw, sh, sw, pt, pl;
    int ldx, ldy, ldr, ldw;
    int cin_off, cout_off, res_off;
    int M, K;
    long x_span_bytes;  // bytes from x (the tensor base) to the end of the input tensor
    long y_span_bytes, res_span_bytes;
    int bias_elems;
    int act_pre, act_post;
    int m_blocks, n_blocks;
    const f16* w_lo;   // optional low half of the weights, fp16(W32 - fp16(W32)), same layout as w; streaming kernel only
    const f16* gate;   // optional per-image input-channel gate [B][2][K] (hi, lo planes of a squeeze-excite scale, folded into the load); pwk only
    int y_lo_off;      // rows kernel: != 0 -> also write fp16(v - fp16(v)) this many halfs after each output (split gate)
    int gate_hw;       // pixels per image (image index of pixel m = m / gate_hw)
    float out_scale;   // H2: 1 / (power-of-two scale folded into the weights AND the bias), applied to the accumulators
    int* status;       // H2: device word raised to VIP_H2_OVERFLOW when an output does not fit the fp16 range (may be NULL)
};

// ---- H2 epilogue core: 8 consecutive channels of one pixel (two accumulator quads) -> activation -> (+ residual) -> post -> packed
// store.  `off` = byte offset of the channel group in y, `roff` in the residual (OOB2 for masked lanes).
template <int ACT, bool RES, int POST>      // POST: 0 none, 1 ReLU, 2 run-time a.act_post
__device__ __forceinline__ void h2_store8(const ConvArgs& a, const f32x4& q0, const f32x4& q1, unsigned off, unsigned roff,
                                          const __amdgpu_buffer_rsrc_t& rb_res, const __amdgpu_buffer_rsrc_t& rb_y) {
    // two values per VALU slot (v_pk_mul / v_pk_fma / v_pk_add_f32): the GELU / swish epilogues of the many-pixel layers are VALU-bound
    // otherwise.  Activations: the packed forms of common.hpp (hardware exp2 / rcp, erfc polynomial: <= 7e-7 absolute).
    f32x2 p[4] = {{q0[0], q0[1]}, {q0[2], q0[3]}, {q1[0], q1[1]}, {q1[2], q1[3]}};
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = vip_act2<ACT>(p[e] * a.out_scale);
    if constexpr (RES) {
        // residual (hi, lo) pairs joined straight into the accumulators: v_fma_mix_f32 takes the fp16 half as an operand (p += hi; p += lo:
        // 4 instructions per pair against 4 conversions + 2 packed adds)
        const uint4 rh = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rb_res, roff, 0, 0));
        const uint4 rl = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rb_res, roff + 16u, 0, 0));
        const unsigned rhw[4] = {rh.x, rh.y, rh.z, rh.w}, rlw[4] = {rl.x, rl.y, rl.z, rl.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float px = p[e].x, py = p[e].y;
            asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(px) : "v"(rhw[e]));
            asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(py) : "v"(rhw[e]));
            asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(px) : "v"(rlw[e]));
            asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(py) : "v"(rlw[e]));
            p[e] = (f32x2){px, py};
        }
    }
    if constexpr (POST == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = (f32x2){fmaxf(p[e].x, 0.f), fmaxf(p[e].y, 0.f)};
    } else if constexpr (POST == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = (f32x2){vip_act_strict(p[e].x, a.act_post), vip_act_strict(p[e].y, a.act_post)};
    }
    // split: hi = rn16(v) (v_cvt_pk_f16_f32), lo = rn16(v - hi) as ONE v_fma_mixlo / mixhi_f16 per value (fma(hi, -1, v) is exact, so
    // this is the same single rounding as converting the fp32 difference).  Range check on the hi halves: a value beyond the fp16 range
    // (or a NaN) converts to an all-ones exponent, and as unsigned 16-bit integers |NaN| > |Inf| > every finite value - one packed
    // integer max per pair, one test per store
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    unsigned hw[4], lw[4];
    u16x2 mx = {0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hw[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(p[e], f16x2));
        const float px = p[e].x, py = p[e].y;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lw[e]) : "v"(hw[e]), "v"(px));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lw[e]) : "v"(hw[e]), "v"(py));
        mx = __builtin_elementwise_max(mx, __builtin_bit_cast(u16x2, hw[e] & 0x7FFF7FFFu));
    }
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
    const u32x4 oh = {hw[0], hw[1], hw[2], hw[3]}, ol = {lw[0], lw[1], lw[2], lw[3]};
    __builtin_amdgcn_raw_buffer_store_b128(oh, rb_y, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(ol, rb_y, off + 16u, 0, 0);
    if (a.status && off != OOB2 && (mx.x >= 0x7C00 || mx.y >= 0x7C00)) *a.status = VIP_H2_OVERFLOW;
}

__device__ __forceinline__ int swz_x(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int swz_w(int row, int chunk) {
    const int key = (((row >> 3) & 3) << 1) | ((row >> 1) & 1);
    return row * 128 + ((chunk ^ key) << 4);
}

template <int ACT>
__device__ __forceinline__ float act_t(float v) {
    if constexpr (ACT == VIP_ACT_RELU) return v > 0.f ? v : 0.f;
    else if constexpr (ACT == VIP_ACT_SILU) return v * vip_sigmoid(v);
    else if constexpr (ACT == VIP_ACT_GELU) return 0.5f * v * (1.f + vip_erf(v * 0.70710678118654752f));
    else if constexpr (ACT == VIP_ACT_SIGMOID) return vip_sigmoid(v);
    else return v;
}

// Lane owns channels n_first + h*32 + (0..7), h = 0,1, of pixel rows m_base + mt*16.  Branch-free: bias /
// residual / output go through buffer descriptors; masked lanes use an out-of-range offset (loads return
// 0, stores are dropped).
template <int MT, int ACT>
__device__ __forceinline__ void epilogue(const ConvArgs& a, f32x4 (&acc)[MT][4], int m_base, int n_first, int group) {
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const int ch_glob = group * a.Cout_g;
    const __amdgpu_buffer_rsrc_t rb_bias =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)(a.bias_elems * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_res =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (unsigned)a.res_span_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_span_bytes, 0x00020000);
    f32x4 bv[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n_first + h * 32;
        const unsigned off = (n < a.Cout_g) ? (unsigned)((ch_glob + n) * 4) : OOB;
        bv[h][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_bias, off, 0, 0));
        bv[h][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_bias, off, 16, 0));
    }
    if constexpr (H2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m_base + mt * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = n_first + h * 32;
                const bool ok = (m < a.M) & (n < a.Cout_g);
                const unsigned off = ok ? (unsigned)(((long)m * a.ldy + a.cout_off + ch_glob + n) * 4) : OOB2;
                const unsigned roff = (ok && a.res) ? (unsigned)(((long)m * a.ldr + a.res_off + ch_glob + n) * 4) : OOB2;
                // the bias (pre-multiplied by the weights' scale on the host) joins the accumulators before the scale is undone
                const f32x4 q0 = acc[mt][h * 2] + bv[h][0], q1 = acc[mt][h * 2 + 1] + bv[h][1];
                if (a.res) h2_store8<ACT, true, 2>(a, q0, q1, off, roff, rb_res, rb_y);
                else h2_store8<ACT, false, 2>(a, q0, q1, off, roff, rb_res, rb_y);
            }
        }
        return;
    }
    const bool post_relu = a.act_post == VIP_ACT_RELU;
    const bool post_other = a.act_post != VIP_ACT_NONE && !post_relu;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m_base + mt * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = n_first + h * 32;
            const bool ok = (m < a.M) & (n < a.Cout_g);
            U4H8 r;
            r.u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                rb_res, ok ? (unsigned)((m * a.ldr + a.res_off + ch_glob + n) * 2) : OOB, 0, 0));
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {   // pairs: packed fp32 math (v_pk_add/fma/mul_f32)
                const f32x4 av = acc[mt][h * 2 + (j >> 2)], bb = bv[h][j >> 2];
                const f32x2 s = (f32x2){av[j & 3], av[(j & 3) + 1]} + (f32x2){bb[j & 3], bb[(j & 3) + 1]};
                const f32x2 t = vip_act2<ACT>(s) + (f32x2){(float)r.e[j], (float)r.e[j + 1]};
                v[j] = t.x;
                v[j + 1] = t.y;
                if (post_relu) {
                    v[j] = fmaxf(v[j], 0.f);
                    v[j + 1] = fmaxf(v[j + 1], 0.f);
                }
            }
            if (post_other) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = vip_act(v[j], a.act_post);
            }
            U4H8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.e[j] = (f16)v[j];
            __builtin_amdgcn_raw_buffer_store_b128(
                __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o.u), rb_y,
                ok ? (unsigned)((m * a.ldy + a.cout_off + ch_glob + n) * 2) : OOB, 0, 0);
        }
    }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    constexpr int WAVES_N = BN / 64;
    constexpr int WAVES_M = 4 / WAVES_N;
    constexpr int WTM = BM / WAVES_M;  // rows of the block tile owned by one wave
    constexpr int MT = WTM / 16;       // 16-row MFMA tiles per wave
    constexpr int A_IT = BM / 32;      // 16-byte chunks of the activation tile per thread
    constexpr int B_IT = BN / 32;
    constexpr int A_BYTES = BM * 128;
    constexpr int STAGE_BYTES = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int group = blockIdx.z;

    // XCD-aware block remap: blocks b and b+8 share an XCD (and its L2); give every XCD a contiguous
    // run of logical tiles so the n-blocks that re-read one activation tile hit the same L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mb = bid / a.n_blocks;
    const int nb = bid - mb * a.n_blocks;
    const int m0 = mb * BM;
    const int n0 = nb * BN;

    const f16* __restrict__ xg = a.x + a.cin_off + group * a.Cin_g;
    const f16* __restrict__ wg = a.w + (size_t)group * a.Cout_g * a.ldw;

    // ---- per-thread staging state -----------------------------------------------------------
    const int chunk = tid & 7;
    const int row0 = tid >> 3;  // + 32*i
    int hi0[A_IT], wi0[A_IT];
    int pix0[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + row0 + 32 * i;
        if (m < a.M) {
            const int hw = a.Ho * a.Wo;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int ho = rem / a.Wo;
            const int wo = rem - ho * a.Wo;
            hi0[i] = ho * a.sh - a.pt;
            wi0[i] = wo * a.sw - a.pl;
            pix0[i] = b * a.H * a.W;
        } else {
            hi0[i] = -(1 << 28);
            wi0[i] = 0;
            pix0[i] = 0;
        }
    }
    // position of this thread's chunk inside the filter: (tap row r, tap col s, channel cc)
    int cc = chunk * 8, tr = 0, ts = 0;
    while (cc >= a.Cin_g) {
        cc -= a.Cin_g;
        if (++ts == a.kw) { ts = 0; ++tr; }
    }

    uint4 ra[A_IT], rb[B_IT];

    // Buffer descriptors: out-of-range offsets read as zero in hardware, so padding taps, the M/N/K
    // tails and masked rows cost no branch (a "cond ? load : 0" in HIP source compiles to a branch around
    // every load plus a vmcnt(0) per load — 8 dependent memory round trips per k-tile).
    const unsigned x_bytes = (unsigned)min((long)0xFFFFFFF0L, a.x_span_bytes - 2L * (a.cin_off + group * a.Cin_g));
    const unsigned w_bytes = (unsigned)min((long)0xFFFFFFF0L, 2L * a.Cout_g * a.ldw);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;

    auto load_tiles = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int hi = hi0[i] + tr, wi = wi0[i] + ts;
            const bool ok = (tr < a.kh) & ((unsigned)hi < (unsigned)a.H) & ((unsigned)wi < (unsigned)a.W);
            const unsigned off = (unsigned)(((pix0[i] + hi * a.W + wi) * a.ldx + cc) * 2);
            ra[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : OOB, 0, 0));
        }
        const int k = kt * 64 + chunk * 8;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + row0 + 32 * i;
            const bool ok = (n < a.Cout_g) & (k < a.K);
            const unsigned off = (unsigned)((n * a.ldw + k) * 2);
            rb[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? off : OOB, 0, 0));
        }
        // advance the filter position by one k-tile (64 halfs)
        cc += 64;
        while (cc >= a.Cin_g) {
            cc -= a.Cin_g;
            if (++ts == a.kw) { ts = 0; ++tr; }
        }
    };
    auto store_tiles = [&](int buf) {
        char* sa = smem + buf * STAGE_BYTES;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<uint4*>(sa + swz_x(row0 + 32 * i, h2_pos(chunk))) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<uint4*>(sb + swz_w(row0 + 32 * i, h2_pos(chunk))) = rb[i];
    };

    // ---- MFMA fragment addressing -------------------------------------------------------------
    const int wave_m0 = (wave / WAVES_N) * WTM;
    const int wave_n0 = (wave % WAVES_N) * 64;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wrow_base = wave_n0 + (l15 >> 2) * 8 + (l15 & 3);  // + (nt>>1)*32 + (nt&1)*4
    const int xrow_base = wave_m0 + l15;                          // + mt*16

    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (a.K + 63) >> 6;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const char* sa = smem + buf * STAGE_BYTES;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ks * 4 + lq;
            U4H8 wf[4], xf[MT];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt].u = *reinterpret_cast<const uint4*>(sb + swz_w(wrow_base + (nt >> 1) * 32 + (nt & 1) * 4, ch));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[mt].u = *reinterpret_cast<const uint4*>(sa + swz_x(xrow_base + mt * 16, ch));
            if constexpr (H2) {
                // ks = 0: x hi plane against BOTH weight planes; ks = 1: x lo plane against the weights' hi plane only
                U4H8 wl[4];
                if (ks == 0) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) wl[nt].u = *reinterpret_cast<const uint4*>(sb + swz_w(wrow_base + (nt >> 1) * 32 + (nt & 1) * 4, 4 + lq));
                } else {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) wf[nt].u = *reinterpret_cast<const uint4*>(sb + swz_w(wrow_base + (nt >> 1) * 32 + (nt & 1) * 4, lq));
                }
                if (VIP_MFMA_PRIO_TILE) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        if (ks == 0) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt].h, xf[mt].h, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xf[mt].h, acc[mt][nt], 0, 0, 0);
                    }
                if (VIP_MFMA_PRIO_TILE) __builtin_amdgcn_s_setprio(0);
                continue;
            }
            if (VIP_MFMA_PRIO_TILE) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xf[mt].h, acc[mt][nt], 0, 0, 0);
            if (VIP_MFMA_PRIO_TILE) __builtin_amdgcn_s_setprio(0);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue (one instantiation per activation so the per-element code is straight-line) ----
    const int m_base = m0 + wave_m0 + l15;
    const int n_first = n0 + wave_n0 + lq * 8;
    switch (a.act_pre) {
        case VIP_ACT_RELU: epilogue<MT, VIP_ACT_RELU>(a, acc, m_base, n_first, group); break;
        case VIP_ACT_SILU: epilogue<MT, VIP_ACT_SILU>(a, acc, m_base, n_first, group); break;
        case VIP_ACT_GELU: epilogue<MT, VIP_ACT_GELU>(a, acc, m_base, n_first, group); break;
        case VIP_ACT_SIGMOID: epilogue<MT, VIP_ACT_SIGMOID>(a, acc, m_base, n_first, group); break;
        default: epilogue<MT, VIP_ACT_NONE>(a, acc, m_base, n_first, group); break;
    }
}

// ---- pointwise (1x1, stride 1) / dense layers with short K: weights-stationary streaming kernel -------------
// For K <= 256 the layer is a stream: read K halfs per pixel, write N.  The tile kernel above pays an LDS round
// trip + barrier per k-tile for BOTH operands and re-stages the weights for every 64..128 pixels.  Here:
//   * the block's weight slice [nb_ch x K] is staged into LDS ONCE (rows stored in MFMA-fragment order, row stride
//     32 (mod 64) bytes -> conflict-free ds_read_b128) and the block then walks pixel tiles with a
//     grid-stride loop;
//   * the activation fragment of v_mfma_f32_16x16x32_f16 (B operand: lane = pixel, 8 consecutive k) IS a 16-byte
//     run of an NHWC row, so it is loaded global -> VGPR directly (no LDS, no barrier in the loop) and kept in
//     registers while the wave sweeps all output-channel sub-tiles: every activation byte is read once;
//   * waves never synchronise after the weight staging, so one wave's epilogue (VALU + stores) overlaps the
//     others' loads and MFMAs.
// epilogue of the streaming kernel: the bias is already in the accumulators (it was the MFMA C operand), the
// activation runs two values per VALU slot, and residual / post-ReLU code exists only in the variants that use it
template <int PT, int ACT, bool RES, bool POST_RELU>
__device__ __forceinline__ void pw_epilogue(const ConvArgs& a, f32x4 (&acc)[PT][4], int m_base, int n_first,
                                            const __amdgpu_buffer_rsrc_t& rb_res, const __amdgpu_buffer_rsrc_t& rb_y) {
    constexpr unsigned OOB = 0xFFFFFFF0u;
    if constexpr (H2) {
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m = m_base + p * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = n_first + h * 32;
                const bool ok = (m < a.M) & (n < a.Cout_g);
                const unsigned off = ok ? (unsigned)(((long)m * a.ldy + a.cout_off + n) * 4) : OOB2;
                const unsigned roff = (RES && ok) ? (unsigned)(((long)m * a.ldr + a.res_off + n) * 4) : OOB2;
                h2_store8<ACT, RES, POST_RELU ? 1 : 0>(a, acc[p][h * 2], acc[p][h * 2 + 1], off, roff, rb_res, rb_y);
            }
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int m = m_base + p * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = n_first + h * 32;
            const bool ok = (m < a.M) & (n < a.Cout_g);
            U4H8 r;
            if constexpr (RES)
                r.u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                    rb_res, ok ? (unsigned)((m * a.ldr + a.res_off + n) * 2) : OOB, 0, 0));
            U4H8 o;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f32x4 av = acc[p][h * 2 + (j >> 2)];
                f32x2 t = vip_act2<ACT>((f32x2){av[j & 3], av[(j & 3) + 1]});
                if constexpr (RES) t = t + (f32x2){(float)r.e[j], (float)r.e[j + 1]};
                if constexpr (POST_RELU) t = (f32x2){fmaxf(t.x, 0.f), fmaxf(t.y, 0.f)};
                o.e[j] = (f16)t.x;
                o.e[j + 1] = (f16)t.y;
            }
            __builtin_amdgcn_raw_buffer_store_b128(
                __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o.u), rb_y,
                ok ? (unsigned)((m * a.ldy + a.cout_off + n) * 2) : OOB, 0, 0);
        }
    }
}

// mode = act_pre (0..4) without residual, 5 = residual, 6 = residual + post-ReLU (both with act_pre none)
// HILO: the LDS row of a channel holds its fp16 weights followed by their low halves (w_lo); every activation fragment is
// multiplied by both, so the layer computes with ~22-bit weights.  These layers are HBM-bound, the second MFMA is free.
template <int KS, int PT, bool PRE, bool HILO>
__global__ __launch_bounds__(256, 2) void pw_gemm_kernel(ConvArgs a, int nb_ch, int lds_stride, int n_tiles, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int n_chunk0 = blockIdx.y * nb_ch;
    const int nch = min(nb_ch, ((a.Cout_g - n_chunk0) + 63) & ~63);   // channels computed by this block (x64)
    constexpr unsigned OOB = 0xFFFFFFF0u;
    float* bias_lds = reinterpret_cast<float*>(smem + nb_ch * lds_stride);

    {   // stage the weight slice: LDS row j <-> channel n_chunk0 + (j & ~63) + perm(j & 63) (fragment order)
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            (void*)a.w, 0, (unsigned)min((long)0xFFFFFFF0L, 2L * a.Cout_g * a.ldw), 0x00020000);
        const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(
            (void*)a.w_lo, 0, HILO ? (unsigned)min((long)0xFFFFFFF0L, 2L * a.Cout_g * a.ldw) : 0u, 0x00020000);
        const int cpr = KS * 4;                      // 16-byte chunks per (zero-padded) row
        constexpr int HALVES = HILO ? 2 : 1;
        for (int i = tid; i < nch * cpr * HALVES; i += 256) {
            const int j = i / (cpr * HALVES), c2 = i - j * (cpr * HALVES);
            const int c = c2 < cpr ? c2 : c2 - cpr;
            const int t = (j >> 4) & 3, r = j & 15;
            const int ch = n_chunk0 + (j & ~63) + (t >> 1) * 32 + (r >> 2) * 8 + (t & 1) * 4 + (r & 3);
            const bool ok = (ch < a.Cout_g) & (c * 8 < a.K);
            const unsigned off = ok ? (unsigned)((ch * a.ldw + c * 8) * 2) : OOB;
            const uint4 v = __builtin_bit_cast(uint4, c2 < cpr ? __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0)
                                                               : __builtin_amdgcn_raw_buffer_load_b128(rwl, off, 0, 0));
            // H2: logical chunk c of the row -> (64-half k-chunk c >> 3, permuted position of c & 7)
            *reinterpret_cast<uint4*>(smem + j * lds_stride + (H2 ? ((c2 & ~7) | h2_pos(c2 & 7)) : c2) * 16) = v;
        }
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            (void*)a.bias, 0, a.bias ? (unsigned)(a.bias_elems * 4) : 0u, 0x00020000);
        for (int i = tid; i < nch; i += 256) {
            const int ch = n_chunk0 + i;
            bias_lds[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, ch < a.Cout_g ? (unsigned)(ch * 4) : OOB, 0, 0));
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + a.cin_off), 0, (unsigned)min((long)0xFFFFFFF0L, a.x_span_bytes - 2L * a.cin_off), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_res =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (unsigned)a.res_span_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_span_bytes, 0x00020000);
    const char* wl = smem + l15 * lds_stride + lq * 16;

    auto load_x = [&](int tile, U4H8 (&xf)[KS][PT]) {
        const int m0 = tile * (64 * PT) + wave * (16 * PT);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int m = m0 + p * 16 + l15, k = H2 ? (ks >> 1) * 64 + lq * 16 + (ks & 1) * 8 : ks * 32 + lq * 8;
                const bool ok = (m < a.M) & (k < a.K);
                xf[ks][p].u = __builtin_bit_cast(
                    uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)((m * a.ldx + k) * 2) : OOB, 0, 0));
            }
    };

    U4H8 xf[KS][PT], xn[PRE ? KS : 1][PRE ? PT : 1];
    int tile = blockIdx.x;
    if (tile < n_tiles) load_x(tile, xf);
    for (; tile < n_tiles; tile += gridDim.x) {
        const int m0 = tile * (64 * PT) + wave * (16 * PT);
        if constexpr (PRE) {
            // next tile's activations: issued now, consumed after this tile's last epilogue (a tile past the end
            // is all out-of-range offsets: no memory traffic)
            load_x(tile + gridDim.x, xn);
        }
        for (int sub = 0; sub < nch; sub += 64) {
            f32x4 acc[PT][4];
            const char* ws = wl + sub * lds_stride;
            f32x4 bv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                bv[nt] = *reinterpret_cast<const f32x4*>(bias_lds + sub + (nt >> 1) * 32 + lq * 8 + (nt & 1) * 4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                U4H8 wf[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wf[nt].u = *reinterpret_cast<const uint4*>(ws + nt * 16 * lds_stride + ks * 64);
                if constexpr (H2) {
                    // step ks = 2 c + plane of x: the hi plane (even ks) meets the weights' lo (step ks + 1) and hi (step ks) planes, the lo
                    // plane (odd ks) the weights' hi plane (step ks - 1)
                    if ((ks & 1) == 0) {
                        U4H8 wl[4];
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) wl[nt].u = *reinterpret_cast<const uint4*>(ws + nt * 16 * lds_stride + (ks + 1) * 64);
#pragma unroll
                        for (int p = 0; p < PT; ++p)
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) {
                                acc[p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt].h, xf[ks][p].h, ks == 0 ? bv[nt] : acc[p][nt], 0, 0, 0);
                                acc[p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xf[ks][p].h, acc[p][nt], 0, 0, 0);
                            }
                    } else {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) wf[nt].u = *reinterpret_cast<const uint4*>(ws + nt * 16 * lds_stride + (ks - 1) * 64);
#pragma unroll
                        for (int p = 0; p < PT; ++p)
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
                                acc[p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xf[ks][p].h, acc[p][nt], 0, 0, 0);
                    }
                    continue;
                }
#pragma unroll
                for (int p = 0; p < PT; ++p)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xf[ks][p].h, ks == 0 ? bv[nt] : acc[p][nt], 0, 0, 0);
                if constexpr (HILO) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) wf[nt].u = *reinterpret_cast<const uint4*>(ws + nt * 16 * lds_stride + (KS + ks) * 64);
#pragma unroll
                    for (int p = 0; p < PT; ++p)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xf[ks][p].h, acc[p][nt], 0, 0, 0);
                }
            }
            const int m_base = m0 + l15, n_first = n_chunk0 + sub + lq * 8;
            switch (mode) {
                case 1: pw_epilogue<PT, VIP_ACT_RELU, false, false>(a, acc, m_base, n_first, rb_res, rb_y); break;
                case 2: pw_epilogue<PT, VIP_ACT_SILU, false, false>(a, acc, m_base, n_first, rb_res, rb_y); break;
                case 3: pw_epilogue<PT, VIP_ACT_GELU, false, false>(a, acc, m_base, n_first, rb_res, rb_y); break;
                case 4: pw_epilogue<PT, VIP_ACT_SIGMOID, false, false>(a, acc, m_base, n_first, rb_res, rb_y); break;
                case 5: pw_epilogue<PT, VIP_ACT_NONE, true, false>(a, acc, m_base, n_first, rb_res, rb_y); break;
                case 6: pw_epilogue<PT, VIP_ACT_NONE, true, true>(a, acc, m_base, n_first, rb_res, rb_y); break;
                default: pw_epilogue<PT, VIP_ACT_NONE, false, false>(a, acc, m_base, n_first, rb_res, rb_y); break;
            }
        }
        if constexpr (PRE) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int p = 0; p < PT; ++p) xf[ks][p] = xn[ks][p];
        } else {
            if (tile + (int)gridDim.x < n_tiles) load_x(tile + gridDim.x, xf);
        }
    }
}

template <int KS, int PT, bool HILO = false>
int launch_pw(const ConvArgs& a, int mode, hipStream_t s) {
    if constexpr (!HILO) {
        if (a.w_lo) return launch_pw<KS, PT, true>(a, mode, s);
    }
    constexpr bool PRE = KS <= 3;
    // two workgroups per CU.  H2: a packed weight slice is twice as large, so the same budget cuts N into more channel chunks and
    // every chunk re-reads the activations; with >= 512K rows that traffic costs more than the second resident workgroup gains
    // (M=2.5M N=384 K=96: 1.71 -> 1.23 ms), below that the shorter launch prefers the two workgroups (M=160K N=512 K=128:
    // 0.16 vs 0.20 ms) - profiles/r04_ab_pw_h2_lds.log.  VIP_PW_H2_LDS_KB overrides.
    static const int h2_lds_kb = getenv("VIP_PW_H2_LDS_KB") ? atoi(getenv("VIP_PW_H2_LDS_KB")) : 0;
    const int LDS_MAX = (H2 ? (h2_lds_kb ? h2_lds_kb : (a.M >= (1 << 19) ? 156 : 72)) : 72) * 1024;
    // row stride in 16-byte chunks == 2 (mod 4), i.e. 32 (mod 64) bytes: ds_read_b128 is serviced in the lane groups
    // {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... over 64 banks, and with the fragment pattern (lane&15 = row, lane>>4 =
    // chunk) that stride puts each group's 16 chunks on 16 distinct 16-byte slots (an ODD chunk stride does not: 46 %
    // conflict cycles measured)
    int s16 = KS * 4 * (HILO ? 2 : 1);
    while ((s16 & 3) != 2) ++s16;
    const int stride = s16 * 16;
    const int cout64 = (a.Cout_g + 63) & ~63;
    const int max_rows = (LDS_MAX / (stride + 4)) & ~63;
    const int n_chunks = (cout64 + max_rows - 1) / max_rows;
    const int nb_ch = (((cout64 / 64 + n_chunks - 1) / n_chunks)) * 64;
    const int n_tiles = (a.M + 64 * PT - 1) / (64 * PT);
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    int gx = (2 * n_cu + n_chunks - 1) / n_chunks;
    if (gx > n_tiles) gx = n_tiles;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pw_gemm_kernel<KS, PT, PRE, HILO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((pw_gemm_kernel<KS, PT, PRE, HILO>), dim3((unsigned)gx, (unsigned)n_chunks), dim3(256),
                       (size_t)nb_ch * (stride + 4), s, a, nb_ch, stride, n_tiles, mode);
    return vip_launch_status("vip_conv2d_nhwc_f16(pw)");
}

template <int PT>
int launch_pw_k(const ConvArgs& a, int mode, hipStream_t s) {
    const int ks = H2 ? 2 * ((a.K + 63) >> 6) : (a.K + 31) >> 5;     // H2: whole 64-half chunks (hi and lo planes of 32 logical k)
    switch (ks) {
        case 1: return launch_pw<1, PT>(a, mode, s);
        case 2: return launch_pw<2, PT>(a, mode, s);
        case 3: return launch_pw<3, PT>(a, mode, s);
        case 4: return launch_pw<4, PT>(a, mode, s);
        case 5:
        case 6: return launch_pw<6, PT>(a, mode, s);
        default: return launch_pw<8, PT>(a, mode, s);
    }
}

// ---- pointwise (1x1, stride 1) / dense layers, any K: activations direct to registers, weights through LDS ----
// Same operand routing as the streaming kernel, for K too large to keep the weight slice resident: the block's
// [64*NG channels] x 64-k weight chunk is double-buffered in LDS (shared by the 4 waves, one barrier per chunk);
// each wave owns 64 pixels and loads their activation fragments global -> VGPR one chunk ahead.  Against the
// im2col tile kernel this halves LDS traffic (no activation round trip), removes the per-k-tile im2col address
// arithmetic from the VALU (offsets here are linear in k) and gives each wave a 64 x 128 accumulator tile.
// Short-K variant of the kernel below: the activation fragments are fetched global -> VGPR directly (no LDS image,
// one barrier per chunk).  2-5 % faster than the LDS-staged form for K < 768, 5-10 % slower beyond (bench_gemm.py).
// PT = 16-pixel tiles per wave: 4 (256-pixel block tile) by default; 2 / 1 for layers whose 256-pixel grid would leave most of the
// chip idle (7x7 and 13x13 maps with narrow outputs: 98 workgroups for M = 12 544, N = 208) - more, smaller workgroups.
template <int NG, bool GATED, int PT = 4>
__global__ __launch_bounds__(256, 2) void pwk_direct_kernel(ConvArgs a, int mode) {
    constexpr int NB = 64 * NG, ROWB = 160;                  // LDS row: 64 halfs + 32 B pad (stride = 32 mod 64: launch_pw)
    constexpr int STAGE = NB * ROWB;
    constexpr int W_IT = NB / 32;                            // 16-byte weight chunks staged per thread per k-chunk
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mb = bid / a.n_blocks, nb = bid - mb * a.n_blocks;
    const int m0 = mb * (64 * PT) + wave * (16 * PT);
    const int n0 = nb * NB;

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.w, 0, (unsigned)min((long)0xFFFFFFF0L, 2L * a.Cout_g * a.ldw), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + a.cin_off), 0, (unsigned)min((long)0xFFFFFFF0L, a.x_span_bytes - 2L * a.cin_off), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_res =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (unsigned)a.res_span_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_span_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_bias =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)(a.bias_elems * 4) : 0u, 0x00020000);

    // weight staging: thread -> (LDS row j = tid/8 + 32 i, 16-byte chunk c = tid%8); row j holds channel perm(j)
    const int wc = tid & 7;
    unsigned w_off[W_IT];
    int w_lds[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int j = (tid >> 3) + 32 * i;
        const int t = (j >> 4) & 3, r = j & 15;
        const int ch = n0 + (j & ~63) + (t >> 1) * 32 + (r >> 2) * 8 + (t & 1) * 4 + (r & 3);
        w_off[i] = ch < a.Cout_g ? (unsigned)((ch * a.ldw + wc * 8) * 2) : OOB;
        w_lds[i] = j * ROWB + h2_pos(wc) * 16;
    }
    // activation rows of this lane: pixel m0 + 16 p + l15, k offset lq*8 (+ 32 ks + 64 chunk)
    unsigned x_off[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int m = m0 + p * 16 + l15;
        // H2: the lane's k-group is the 32-byte (hi, lo) pair lq of the 64-half chunk; plane `ks` is 16 bytes further
        x_off[p] = m < a.M ? (unsigned)((m * a.ldx + lq * (H2 ? 16 : 8)) * 2) : 0xFFFF0000u;   // + k bytes stays out of range
    }
    // squeeze-excite gate folded into the activation operand: x[m, k] * (hi + lo)[image(m), k] as fma(x, hi, x * lo) in
    // packed fp16 - the product is rounded once, per element; the gate itself carries ~22 bits (see se_gate.hip)
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.gate, 0, GATED ? (unsigned)min((long)0xFFFFFFF0L, (H2 ? 2L : 4L) * ((a.M + a.gate_hw - 1) / a.gate_hw) * a.K) : 0u, 0x00020000);
    // H2: the gate is a packed [B][Cin] tensor (a.K = 2 Cin halfs per image); the lane's 8 channels are the 32-byte pair lq of the chunk
    unsigned g_off[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int m = m0 + p * 16 + l15;
        g_off[p] = (GATED && m < a.M) ? (H2 ? (unsigned)((m / a.gate_hw) * a.K * 2 + lq * 32) : (unsigned)(((m / a.gate_hw) * 2 * a.K + lq * 8) * 2))
                                      : 0xFFFF0000u;
    }
    const int nk = (a.K + 63) >> 6;

    uint4 wst[W_IT];
    auto load_w = [&](int kc) {
        const bool ok = kc * 64 + wc * 8 < a.K;
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            wst[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? w_off[i] + kc * 128 : OOB, 0, 0));
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) *reinterpret_cast<uint4*>(smem + buf * STAGE + w_lds[i]) = wst[i];
    };
    U4H8 xf[2][PT], gq[GATED ? 2 : 1][GATED ? PT : 1];      // gq: the gate planes (hi, lo) of ONE k-step
    auto load_x = [&](int kc, int ks) {
        // weights of the K tail are zero in LDS, but 0 * (Inf/NaN garbage of the next row) is NaN: mask the lanes
        const bool ok = kc * 64 + (H2 ? lq * 16 + ks * 8 : ks * 32 + lq * 8) < a.K;
#pragma unroll
        for (int p = 0; p < PT; ++p)
            xf[ks][p].u = __builtin_bit_cast(
                uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? x_off[p] + kc * 128 + ks * (H2 ? 16 : 64) : OOB, 0, 0));
    };
    // Gate fragments run one k-step ahead of their use in ONE register set (a second prefetched set does not fit next to
    // the accumulators): gate_x(ks) folds them into xf[ks] - fma(x, hi, x * lo) in packed fp16 - and the set is
    // re-requested at once for the following k-step, landing under the MFMAs in between.  The buffer is [B][2][K]
    // halfs, L1/L2-resident.
    auto load_g = [&](int kc, int ks) {
        if constexpr (GATED && H2) {         // both planes of the chunk's gate values, once per chunk (ks unused)
            const bool ok = kc * 64 + lq * 16 < a.K;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                gq[0][p].u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? g_off[p] + kc * 128 : OOB, 0, 0));
                gq[1][p].u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? g_off[p] + kc * 128 + 16 : OOB, 0, 0));
            }
        } else if constexpr (GATED) {
            const bool ok = kc * 64 + ks * 32 + lq * 8 < a.K;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                gq[0][p].u = __builtin_bit_cast(
                    uint4, __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? g_off[p] + kc * 128 + ks * 64 : OOB, 0, 0));
                gq[1][p].u = __builtin_bit_cast(
                    uint4, __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? g_off[p] + kc * 128 + ks * 64 + 2 * a.K : OOB, 0, 0));
            }
        }
    };
    auto gate_x = [&](int ks) {
        if constexpr (GATED && H2) {         // (x_hi + x_lo) * (g_hi + g_lo) in fp32, split again: both planes of the chunk at once
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                float xv[8], gv[8];
                h2_join8(xf[0][p], xf[1][p], xv);
                h2_join8(gq[0][p], gq[1][p], gv);
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] *= gv[j];
                h2_split8(xv, xf[0][p], xf[1][p]);
            }
        } else if constexpr (GATED) {
#pragma unroll
            for (int p = 0; p < PT; ++p)                                              // 4 x (v_pk_mul_f16 + v_pk_fma_f16)
                xf[ks][p].h = __builtin_elementwise_fma(xf[ks][p].h, gq[0][p].h, xf[ks][p].h * gq[1][p].h);
        }
    };

    f32x4 acc[NG][PT][4];
    {   // bias is the C operand of the first MFMA of every accumulator
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = n0 + g * 64 + (nt >> 1) * 32 + lq * 8 + (nt & 1) * 4;
                const f32x4 bv = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_bias, n < a.Cout_g ? (unsigned)(n * 4) : OOB, 0, 0));
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[g][p][nt] = bv;
            }
    }

    load_w(0);
    load_x(0, 0);
    load_x(0, 1);
    load_g(0, 0);
    store_w(0);
    __syncthreads();

    auto compute = [&](int buf, int ks) {
        const char* ws = smem + buf * STAGE + l15 * ROWB + lq * 16 + ks * 64;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            U4H8 wf[4];
            if constexpr (H2) {
                // ks = 0: the x hi fragments against the weights' lo plane (ws + 64) and hi plane (ws); ks = 1: x lo against the hi
                // plane (ws - 64: the lambda's ks * 64 points at the lo plane)
                const char* wh = ws - ks * 64;
                U4H8 wl[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wf[nt].u = *reinterpret_cast<const uint4*>(wh + (g * 64 + nt * 16) * ROWB);
                if (ks == 0) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) wl[nt].u = *reinterpret_cast<const uint4*>(wh + 64 + (g * 64 + nt * 16) * ROWB);
                }
                if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int p = 0; p < PT; ++p) {
                        if (ks == 0) acc[g][p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt].h, xf[0][p].h, acc[g][p][nt], 0, 0, 0);
                        acc[g][p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xf[ks][p].h, acc[g][p][nt], 0, 0, 0);
                    }
                if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
                continue;
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt].u = *reinterpret_cast<const uint4*>(ws + (g * 64 + nt * 16) * ROWB);
            if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    acc[g][p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xf[ks][p].h, acc[g][p][nt], 0, 0, 0);
            if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
        }
    };

    // One activation register set: the fragments of k-step ks are re-loaded for the NEXT chunk as soon as this
    // chunk's MFMAs on them are issued, so each load has the other k-step's 32 MFMAs (plus the second resident
    // wave) to land.  Everything in the loop is unconditional (a chunk past the end is all out-of-range offsets:
    // zeros, no memory traffic): a branch around a prefetch makes hipcc fold the "loads skipped" path into its
    // vmcnt bookkeeping and the next MFMA then waits for loads that were only just issued.
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        load_w(kc + 1);
        __builtin_amdgcn_sched_barrier(0);   // pin the issue points: the scheduler otherwise sinks every load below
        if constexpr (GATED && H2) {         // the MFMAs, right in front of its wait
            gate_x(0);                       // both planes of chunk kc
            __builtin_amdgcn_sched_barrier(0);
            load_g(kc + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (GATED) {
            gate_x(0);
            __builtin_amdgcn_sched_barrier(0);
            load_g(kc, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        compute(buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_x(kc + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (GATED && !H2) {
            gate_x(1);
            __builtin_amdgcn_sched_barrier(0);
            load_g(kc + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        compute(buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        load_x(kc + 1, 1);
        store_w(buf ^ 1);
        __syncthreads();
    }

    // opaque to the optimiser: otherwise the epilogue's address arithmetic is hoisted above the k-loop and its
    // ~20 registers stay live through it (the loop is at the 256-VGPR limit of two waves per SIMD)
    int m_base = m0 + l15, n_lane = n0 + lq * 8;
    asm volatile("" : "+v"(m_base), "+v"(n_lane));
    // (not a loop over g: with the packed-storage epilogue inlined seven times the unroller gives up and the accumulators would be
    // indexed dynamically, i.e. live in scratch)
    auto epi = [&](auto gtag) {
        constexpr int G_ = decltype(gtag)::value;
        const int n_first = n_lane + G_ * 64;
        switch (mode) {
            case 1: pw_epilogue<PT, VIP_ACT_RELU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 2: pw_epilogue<PT, VIP_ACT_SILU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 3: pw_epilogue<PT, VIP_ACT_GELU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 4: pw_epilogue<PT, VIP_ACT_SIGMOID, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 5: pw_epilogue<PT, VIP_ACT_NONE, true, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 6: pw_epilogue<PT, VIP_ACT_NONE, true, true>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            default: pw_epilogue<PT, VIP_ACT_NONE, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
        }
    };
    epi(std::integral_constant<int, 0>{});
    if constexpr (NG > 1) epi(std::integral_constant<int, 1>{});
}


// ---- pointwise / dense layers with SHORT K and WIDE N: activations resident in registers, weights streamed over all of N ----
// pwk_direct_kernel gives every 128-channel slice of the output its own workgroup: with K <= 256 that workgroup runs 2-4 k-chunks, so its
// life is one pipeline fill (the first weight chunk and activation fragments: ~2 us of L2 / HBM latency) and one drain (the epilogue)
// around ~1 us of MFMAs, and each of the N / 128 workgroups of a pixel tile fetches the same activations again (PMC, round 3: matrix
// pipe 30 % busy, half of a wave's life in s_waitcnt).  Here a workgroup keeps the activation fragments of ALL of K in registers
// (KSC 64-half chunks: <= 64 VGPRs at 32 pixels per wave) and walks over every 128-channel tile of N: the weight chunks of
// (tile, chunk) pairs stream through the same double-buffered LDS image back to back - the prefetch of tile t+1's first chunk is issued
// under tile t's last MFMAs, so there is one fill per workgroup instead of one per 128 channels - and the activations are read once.
template <int KSC, int PT>
__global__ __launch_bounds__(256, 2) void pwx_kernel(ConvArgs a, int mode) {
    constexpr int NG = 2, NB = 64 * NG, ROWB = 160;
    constexpr int STAGE = NB * ROWB;
    constexpr int W_IT = NB / 32;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    int bid = blockIdx.x;
    {   // XCD-contiguous pixel tiles (as pwk_direct_kernel)
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = bid * (64 * PT) + wave * (16 * PT);

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.w, 0, (unsigned)min((long)0xFFFFFFF0L, 2L * a.Cout_g * a.ldw), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + a.cin_off), 0, (unsigned)min((long)0xFFFFFFF0L, a.x_span_bytes - 2L * a.cin_off), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_res =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (unsigned)a.res_span_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_span_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_bias =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)(a.bias_elems * 4) : 0u, 0x00020000);

    // weight staging: thread -> (LDS row j = tid/8 + 32 i, 16-byte chunk c = tid%8); row j holds channel n0 + perm(j)
    const int wc = tid & 7;
    int w_ch[W_IT], w_lds[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int j = (tid >> 3) + 32 * i;
        const int t = (j >> 4) & 3, r = j & 15;
        w_ch[i] = (j & ~63) + (t >> 1) * 32 + (r >> 2) * 8 + (t & 1) * 4 + (r & 3);
        w_lds[i] = j * ROWB + h2_pos(wc) * 16;
    }
    uint4 wst[W_IT];
    auto load_w = [&](int n0, int kc) {
        const bool okk = kc * 64 + wc * 8 < a.K;
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            const int ch = n0 + w_ch[i];
            wst[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rw, (okk && ch < a.Cout_g) ? (unsigned)((ch * a.ldw + wc * 8) * 2) + kc * 128 : OOB, 0, 0));
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) *reinterpret_cast<uint4*>(smem + buf * STAGE + w_lds[i]) = wst[i];
    };

    // the activation fragments of the whole K axis (masked beyond K: the weights there are zero in LDS, but 0 * garbage may be NaN)
    U4H8 xf[2 * KSC][PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int m = m0 + p * 16 + l15;
        const unsigned xo = m < a.M ? (unsigned)((m * a.ldx + lq * (H2 ? 16 : 8)) * 2) : 0xFFFF0000u;
#pragma unroll
        for (int kc = 0; kc < KSC; ++kc)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bool ok = kc * 64 + (H2 ? lq * 16 + ks * 8 : ks * 32 + lq * 8) < a.K;
                xf[2 * kc + ks][p].u = __builtin_bit_cast(
                    uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xo + kc * 128 + ks * (H2 ? 16 : 64) : OOB, 0, 0));
            }
    }

    f32x4 acc[NG][PT][4];
    auto init_acc = [&](int n0) {     // bias is the C operand of the first MFMA of every accumulator
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = n0 + g * 64 + (nt >> 1) * 32 + lq * 8 + (nt & 1) * 4;
                const f32x4 bv = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_bias, n < a.Cout_g ? (unsigned)(n * 4) : OOB, 0, 0));
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[g][p][nt] = bv;
            }
    };
    auto compute = [&](int buf, int ks, const U4H8 (&xk)[2][PT]) {
        const char* ws = smem + buf * STAGE + l15 * ROWB + lq * 16 + ks * 64;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            U4H8 wf[4];
            if constexpr (H2) {      // ks = 0: x hi against the weights' lo and hi planes; ks = 1: x lo against the hi plane
                const char* wh = ws - ks * 64;
                U4H8 wl[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wf[nt].u = *reinterpret_cast<const uint4*>(wh + (g * 64 + nt * 16) * ROWB);
                if (ks == 0) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) wl[nt].u = *reinterpret_cast<const uint4*>(wh + 64 + (g * 64 + nt * 16) * ROWB);
                }
                if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int p = 0; p < PT; ++p) {
                        if (ks == 0) acc[g][p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt].h, xk[0][p].h, acc[g][p][nt], 0, 0, 0);
                        acc[g][p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xk[ks][p].h, acc[g][p][nt], 0, 0, 0);
                    }
                if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
                continue;
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt].u = *reinterpret_cast<const uint4*>(ws + (g * 64 + nt * 16) * ROWB);
            if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[g][p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xk[ks][p].h, acc[g][p][nt], 0, 0, 0);
            if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
        }
    };

    const int n_tiles = (a.Cout_g + NB - 1) / NB;
    load_w(0, 0);
    store_w(0);
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (int t = 0; t < n_tiles; ++t) {
        const int n0 = t * NB;
        init_acc(n0);
#pragma unroll
        for (int kc = 0; kc < KSC; ++kc) {
            // next (tile, chunk) pair; past the last tile every offset is out of range (zeros, no traffic)
            if (kc + 1 < KSC) load_w(n0, kc + 1);
            else load_w(n0 + NB, 0);
            __builtin_amdgcn_sched_barrier(0);
            U4H8 xk[2][PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                xk[0][p] = xf[2 * kc][p];
                xk[1][p] = xf[2 * kc + 1][p];
            }
            compute(buf, 0, xk);
            compute(buf, 1, xk);
            __builtin_amdgcn_sched_barrier(0);
            store_w(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        int m_base = m0 + l15, n_lane = n0 + lq * 8;
        asm volatile("" : "+v"(m_base), "+v"(n_lane));
        auto epi = [&](auto gtag) {
            constexpr int G_ = decltype(gtag)::value;
            const int n_first = n_lane + G_ * 64;
            switch (mode) {
                case 1: pw_epilogue<PT, VIP_ACT_RELU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
                case 2: pw_epilogue<PT, VIP_ACT_SILU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
                case 3: pw_epilogue<PT, VIP_ACT_GELU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
                case 4: pw_epilogue<PT, VIP_ACT_SIGMOID, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
                case 5: pw_epilogue<PT, VIP_ACT_NONE, true, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
                case 6: pw_epilogue<PT, VIP_ACT_NONE, true, true>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
                default: pw_epilogue<PT, VIP_ACT_NONE, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            }
        };
        epi(std::integral_constant<int, 0>{});
        epi(std::integral_constant<int, 1>{});
    }
}

template <int KSC, int PT>
int launch_pwx(const ConvArgs& a0, int mode, hipStream_t s) {
    ConvArgs a = a0;
    a.m_blocks = (a.M + 64 * PT - 1) / (64 * PT);
    a.n_blocks = 1;
    hipLaunchKernelGGL((pwx_kernel<KSC, PT>), dim3((unsigned)a.m_blocks), dim3(256), 0, s, a, mode);
    return vip_launch_status("vip_conv2d_nhwc_f16(pwx)");
}

#if VIP_BUILD_EXPERIMENTS
// ---- the direct kernel with the activation fragments requested TWO k-chunks ahead (round 3) ----------------------------------
// EXPERIMENT (VIP_BUILD_EXPERIMENTS=1, selected per call with VIP_PWK_PF2=1) - a NEGATIVE result, kept for the record.
// PMC of pwk_direct_kernel<2, false, 4> on its typical shapes (profiles/r03_pwk_pmc_*.txt): the matrix pipe is busy 30 % of the time,
// a wave spends 47-62 % of its life in s_waitcnt, and a k-chunk takes ~6 000 cycles against 1 024 cycles of MFMAs.  Hypothesis: the
// loop is a chain of memory round trips with too few bytes in flight.  This variant halves the wave tile to 32 pixels x 128 channels
// (64 accumulator registers instead of 128) and spends the registers on a second activation register set: the fragments of chunk
// kc + 2 are requested while chunk kc is multiplied (the k-loop is unrolled by two so that a chunk's register set is a compile-time
// choice), 3 waves per SIMD instead of 2.  Bit-identical to the default kernel and SLOWER on 12 of 16 ensemble shapes (0.81-1.12x,
// sum 680 vs 642 us; profiles/r03_pwk_deep_prefetch_ab.log): the 128-pixel block tile doubles the weight staging and the L2 -> CU
// traffic per output (~7 TB/s chip-wide either way), which costs more than the extra loads in flight buy.
template <int NG>
__global__ __launch_bounds__(256, 3) void pwk_direct2_kernel(ConvArgs a, int mode) {
    constexpr int PT = 2;
    constexpr int NB = 64 * NG, ROWB = 160;
    constexpr int STAGE = NB * ROWB;
    constexpr int W_IT = NB / 32;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mb = bid / a.n_blocks, nb = bid - mb * a.n_blocks;
    const int m0 = mb * (64 * PT) + wave * (16 * PT);
    const int n0 = nb * NB;

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.w, 0, (unsigned)min((long)0xFFFFFFF0L, 2L * a.Cout_g * a.ldw), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + a.cin_off), 0, (unsigned)min((long)0xFFFFFFF0L, a.x_span_bytes - 2L * a.cin_off), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_res =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (unsigned)a.res_span_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_span_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_bias =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)(a.bias_elems * 4) : 0u, 0x00020000);

    const int wc = tid & 7;
    unsigned w_off[W_IT];
    int w_lds[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int j = (tid >> 3) + 32 * i;
        const int t = (j >> 4) & 3, r = j & 15;
        const int ch = n0 + (j & ~63) + (t >> 1) * 32 + (r >> 2) * 8 + (t & 1) * 4 + (r & 3);
        w_off[i] = ch < a.Cout_g ? (unsigned)((ch * a.ldw + wc * 8) * 2) : OOB;
        w_lds[i] = j * ROWB + wc * 16;
    }
    unsigned x_off[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int m = m0 + p * 16 + l15;
        x_off[p] = m < a.M ? (unsigned)((m * a.ldx + lq * 8) * 2) : 0xFFFF0000u;
    }
    const int nk = (a.K + 63) >> 6;

    uint4 wst[W_IT];
    auto load_w = [&](int kc) {
        const bool ok = kc * 64 + wc * 8 < a.K;
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            wst[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? w_off[i] + kc * 128 : OOB, 0, 0));
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) *reinterpret_cast<uint4*>(smem + buf * STAGE + w_lds[i]) = wst[i];
    };
    U4H8 xf[2][2][PT];                                         // [register set = chunk parity][k-step][pixel tile]
    auto load_x = [&](U4H8 (&dst)[PT], int kc, int ks) {
        const bool ok = kc * 64 + ks * 32 + lq * 8 < a.K;       // also false for every chunk past the end of K: zeros, no traffic
#pragma unroll
        for (int p = 0; p < PT; ++p)
            dst[p].u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? x_off[p] + kc * 128 + ks * 64 : OOB, 0, 0));
    };

    f32x4 acc[NG][PT][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + g * 64 + (nt >> 1) * 32 + lq * 8 + (nt & 1) * 4;
            const f32x4 bv = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_bias, n < a.Cout_g ? (unsigned)(n * 4) : OOB, 0, 0));
#pragma unroll
            for (int p = 0; p < PT; ++p) acc[g][p][nt] = bv;
        }

    load_w(0);
    load_x(xf[0][0], 0, 0);
    load_x(xf[0][1], 0, 1);
    load_x(xf[1][0], 1, 0);
    load_x(xf[1][1], 1, 1);
    store_w(0);
    __syncthreads();

    auto compute = [&](int buf, const U4H8 (&xs)[PT], int ks) {
        const char* ws = smem + buf * STAGE + l15 * ROWB + lq * 16 + ks * 64;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            U4H8 wf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt].u = *reinterpret_cast<const uint4*>(ws + (g * 64 + nt * 16) * ROWB);
            if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    acc[g][p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xs[p].h, acc[g][p][nt], 0, 0, 0);
            if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
        }
    };
    // one chunk on register set SET (compile time): its fragments are re-requested for chunk kc + 2 as soon as their MFMAs are issued
#define VIP_PWK2_CHUNK(SET, KC)                               \
    {                                                         \
        const int buf = (KC) & 1;                             \
        load_w((KC) + 1);                                     \
        __builtin_amdgcn_sched_barrier(0);                    \
        compute(buf, xf[SET][0], 0);                          \
        __builtin_amdgcn_sched_barrier(0);                    \
        load_x(xf[SET][0], (KC) + 2, 0);                      \
        __builtin_amdgcn_sched_barrier(0);                    \
        compute(buf, xf[SET][1], 1);                          \
        __builtin_amdgcn_sched_barrier(0);                    \
        load_x(xf[SET][1], (KC) + 2, 1);                      \
        store_w(buf ^ 1);                                     \
        __syncthreads();                                      \
    }
    for (int kc = 0; kc < nk; kc += 2) {
        VIP_PWK2_CHUNK(0, kc)
        if (kc + 1 < nk) VIP_PWK2_CHUNK(1, kc + 1)            // wave-uniform: nk comes from a kernel argument
    }
#undef VIP_PWK2_CHUNK

    int m_base = m0 + l15, n_lane = n0 + lq * 8;
    asm volatile("" : "+v"(m_base), "+v"(n_lane));
    // (not a loop over g: with the packed-storage epilogue inlined seven times the unroller gives up and the accumulators would be
    // indexed dynamically, i.e. live in scratch)
    auto epi = [&](auto gtag) {
        constexpr int G_ = decltype(gtag)::value;
        const int n_first = n_lane + G_ * 64;
        switch (mode) {
            case 1: pw_epilogue<PT, VIP_ACT_RELU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 2: pw_epilogue<PT, VIP_ACT_SILU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 3: pw_epilogue<PT, VIP_ACT_GELU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 4: pw_epilogue<PT, VIP_ACT_SIGMOID, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 5: pw_epilogue<PT, VIP_ACT_NONE, true, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 6: pw_epilogue<PT, VIP_ACT_NONE, true, true>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            default: pw_epilogue<PT, VIP_ACT_NONE, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
        }
    };
    epi(std::integral_constant<int, 0>{});
    if constexpr (NG > 1) epi(std::integral_constant<int, 1>{});
}

#endif  // VIP_BUILD_EXPERIMENTS

template <int NG, int WN, bool IM2COL>
__global__ __launch_bounds__(256 * WN, WN == 1 ? 2 : 1) void pwk_gemm_kernel(ConvArgs a, int mode) {
    // IM2COL: the same kernel for k x k / strided / grouped convolutions - only the activation staging changes (each
    // 16-byte chunk of a k-chunk belongs to one filter tap: the lane's tap and channel come from one division per
    // chunk, the pixel coordinates of its rows are precomputed); group = blockIdx.y.
    if constexpr (IM2COL) {
        const int group = blockIdx.y;
        a.x += group * a.Cin_g;
        a.w += (size_t)group * a.Cout_g * a.ldw;
        if (a.bias) a.bias += group * a.Cout_g;
        a.bias_elems -= group * a.Cout_g;
        a.cout_off += group * a.Cout_g;
        a.res_off += group * a.Cout_g;
        a.x_span_bytes -= 2L * group * a.Cin_g;
    }
    // 4 x WN waves: wave (wm, wn) owns pixels 64 wm .. +63 and channels (64 NG) wn .. of the 256 x (64 NG WN) block tile.
    // WN = 2 halves the L2 traffic of the activations (each activation image feeds two waves) at the same number of
    // resident waves per CU (one 8-wave workgroup instead of two 4-wave ones).
    constexpr int PT = 4, NB = 64 * NG, NBLK = NB * WN, ROWB = 160;   // LDS row: 64 halfs + 32 B pad (32 mod 64: launch_pw)
    constexpr int NTHR = 256 * WN;
    constexpr int STAGE = NBLK * ROWB;
    constexpr int W_IT = NBLK * 8 / NTHR;                    // 16-byte weight chunks staged per thread per k-chunk (4)
    constexpr int X_IT = 256 * 8 / NTHR;                     // 16-byte activation chunks per thread per k-chunk (8 / WN)
    constexpr int XS = 64 * ROWB;                            // one activation image: 64 pixels x 64 k
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 weight stages][4 activation images]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 3, wn = wave >> 2;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mb = bid / a.n_blocks, nb = bid - mb * a.n_blocks;
    const int mblk = mb * 256;                               // first pixel of the block tile
    const int m0 = mblk + wm * 64;                           // first pixel of this wave
    const int nblk = nb * NBLK;
    const int n0 = nblk + wn * NB;

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.w, 0, (unsigned)min((long)0xFFFFFFF0L, 2L * a.Cout_g * a.ldw), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + a.cin_off), 0, (unsigned)min((long)0xFFFFFFF0L, a.x_span_bytes - 2L * a.cin_off), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_res =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (unsigned)a.res_span_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_span_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_bias =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)(a.bias_elems * 4) : 0u, 0x00020000);

    // weight staging: thread -> (LDS row j = tid/8 + (NTHR/8) i, 16-byte chunk c = tid%8); row j holds channel perm(j)
    const int wc = tid & 7;
    unsigned w_off[W_IT];
    int w_lds[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int j = (tid >> 3) + (NTHR / 8) * i;
        const int t = (j >> 4) & 3, r = j & 15;
        const int ch = nblk + (j & ~63) + (t >> 1) * 32 + (r >> 2) * 8 + (t & 1) * 4 + (r & 3);
        w_off[i] = ch < a.Cout_g ? (unsigned)((ch * a.ldw + wc * 8) * 2) : OOB;
        w_lds[i] = j * ROWB + h2_pos(wc) * 16;
    }
    // Activation staging: the block's 256 pixels x 64 k of a chunk are fetched in FULL 128-byte lines (8 consecutive
    // lanes = one pixel row's 64 halfs, 8 rows per wave-instruction), parked in four 64-pixel LDS images and read back
    // as MFMA fragments.  Fetching the fragments directly (16 rows x 64 B per instruction) costs the texture path two
    // half-used lines per row; and with the activation descriptor zero-sized (timing-only build) the K = 3072 GEMM ran
    // 1.84x faster: the activation fetch, not the matrix core, sets this kernel's pace.
    // WN = 1: every wave stages its OWN image (rows 64 wm + lane/8 + 8 i), so rewriting it needs no workgroup barrier -
    // LDS executes one wave's accesses in order; WN = 2: the images are shared, all threads stage all of them.
    constexpr int XROWS = WN == 1 ? 8 : NTHR / 8;            // rows covered by one staging pass of this thread's group
    const int xr = WN == 1 ? wm * 64 + (lane >> 3) : (tid >> 3);
    const int xc = tid & 7;                                  // 16-byte chunk
    char* ximg = smem + 2 * STAGE;
    const unsigned x_base = (unsigned)(((mblk + xr) * a.ldx + xc * 8) * 2);
    const unsigned x_rstep = (unsigned)(XROWS * a.ldx * 2);
    int pix0[IM2COL ? X_IT : 1], hw0[IM2COL ? X_IT : 1];     // image base pixel, packed (hi0, wi0) of this lane's rows
    if constexpr (IM2COL) {
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
            const int m = mblk + xr + XROWS * i;
            int hi0 = -30000, wi0 = 0, pb = 0;
            if (m < a.M) {
                const int hw = a.Ho * a.Wo;
                const int b = m / hw, rem = m - b * hw;
                const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
                hi0 = ho * a.sh - a.pt;
                wi0 = wo * a.sw - a.pl;
                pb = b * a.H * a.W;
            }
            pix0[i] = pb;
            hw0[i] = hi0 * 65536 + (wi0 & 0xFFFF);
        }
    }
    const int nk = (a.K + 63) >> 6;

    uint4 wst[W_IT];
    auto load_w = [&](int kc) {
        const bool ok = kc * 64 + wc * 8 < a.K;
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            wst[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? w_off[i] + kc * 128 : OOB, 0, 0));
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) *reinterpret_cast<uint4*>(smem + buf * STAGE + w_lds[i]) = wst[i];
    };
    U4H8 xst[X_IT];
    auto load_x = [&](int kc) {
        // weights of the K tail are zero in LDS, but 0 * (Inf/NaN garbage of the next row) is NaN: mask the lanes
        const bool kok = kc * 64 + xc * 8 < a.K;
        if constexpr (IM2COL) {
            const int k0 = kc * 64 + xc * 8;
            const int tap = k0 / a.Cin_g, c = k0 - tap * a.Cin_g;
            const int tr = tap / a.kw, ts = tap - tr * a.kw;
#pragma unroll
            for (int i = 0; i < X_IT; ++i) {
                const int hi = (hw0[i] >> 16) + tr, wi = (int)(short)(hw0[i] & 0xFFFF) + ts;
                const bool ok = kok & ((unsigned)hi < (unsigned)a.H) & ((unsigned)wi < (unsigned)a.W);
                const unsigned off = (unsigned)(((pix0[i] + hi * a.W + wi) * a.ldx + c) * 2);
                xst[i].u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : OOB, 0, 0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < X_IT; ++i) {
                const bool ok = kok & (mblk + xr + XROWS * i < a.M);
                xst[i].u = __builtin_bit_cast(
                    uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? x_base + i * x_rstep + kc * 128 : OOB, 0, 0));
            }
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
            *reinterpret_cast<uint4*>(ximg + (xr + XROWS * i) * ROWB + h2_pos(xc) * 16) = xst[i].u;   // image r/64, row r%64
        }
    };
    const char* xs = ximg + wm * XS;

    f32x4 acc[NG][PT][4];
    {   // bias is the C operand of the first MFMA of every accumulator
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = n0 + g * 64 + (nt >> 1) * 32 + lq * 8 + (nt & 1) * 4;
                const f32x4 bv = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_bias, n < a.Cout_g ? (unsigned)(n * 4) : OOB, 0, 0));
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[g][p][nt] = bv;
            }
    }

    load_w(0);
    load_x(0);
    store_w(0);
    store_x();
    __syncthreads();

    auto compute = [&](int buf, int ks) {
        const char* ws = smem + buf * STAGE + l15 * ROWB + lq * 16 + ks * 64;
        const char* xl = xs + l15 * ROWB + lq * 16 + ks * 64;
        U4H8 xf[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) xf[p].u = *reinterpret_cast<const uint4*>(xl + p * 16 * ROWB);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            U4H8 wf[4];
            if constexpr (H2) {     // as in pwk_direct_kernel: xf is the hi (ks = 0) / lo (ks = 1) plane of x
                const char* wh = ws - ks * 64;
                U4H8 wl[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wf[nt].u = *reinterpret_cast<const uint4*>(wh + (g * 64 + nt * 16) * ROWB);
                if (ks == 0) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) wl[nt].u = *reinterpret_cast<const uint4*>(wh + 64 + (g * 64 + nt * 16) * ROWB);
                }
                if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int p = 0; p < PT; ++p) {
                        if (ks == 0) acc[g][p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt].h, xf[p].h, acc[g][p][nt], 0, 0, 0);
                        acc[g][p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xf[p].h, acc[g][p][nt], 0, 0, 0);
                    }
                if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
                continue;
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt].u = *reinterpret_cast<const uint4*>(ws + (g * 64 + nt * 16) * ROWB);
            if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    acc[g][p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xf[p].h, acc[g][p][nt], 0, 0, 0);
            if (VIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
        }
    };

    // The next chunk's weights and activations are requested (global -> VGPR) before this chunk's math and written to
    // LDS after it; the weight image is double-buffered, the activation images are single-buffered (LDS budget) and
    // rewritten between two barriers.  Everything in the loop is
    // unconditional (a chunk past the end is all out-of-range offsets: zeros, no memory traffic): a branch around a
    // prefetch makes hipcc fold the "loads skipped" path into its vmcnt bookkeeping, and without the sched_barriers
    // it sinks every load below the MFMAs, right in front of its wait - both silently serialise the pipeline.
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        load_w(kc + 1);
        load_x(kc + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(buf, 0);
        compute(buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        store_w(buf ^ 1);
        if constexpr (WN > 1) __syncthreads();   // every wave is done reading the (single-buffered, shared) activation images
        store_x();
        __syncthreads();
    }

    // opaque to the optimiser: otherwise the epilogue's address arithmetic is hoisted above the k-loop and its
    // ~20 registers stay live through it (the loop is at the 256-VGPR limit of two waves per SIMD)
    int m_base = m0 + l15, n_lane = n0 + lq * 8;
    asm volatile("" : "+v"(m_base), "+v"(n_lane));
    // (not a loop over g: with the packed-storage epilogue inlined seven times the unroller gives up and the accumulators would be
    // indexed dynamically, i.e. live in scratch)
    auto epi = [&](auto gtag) {
        constexpr int G_ = decltype(gtag)::value;
        const int n_first = n_lane + G_ * 64;
        switch (mode) {
            case 1: pw_epilogue<PT, VIP_ACT_RELU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 2: pw_epilogue<PT, VIP_ACT_SILU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 3: pw_epilogue<PT, VIP_ACT_GELU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 4: pw_epilogue<PT, VIP_ACT_SIGMOID, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 5: pw_epilogue<PT, VIP_ACT_NONE, true, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            case 6: pw_epilogue<PT, VIP_ACT_NONE, true, true>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            default: pw_epilogue<PT, VIP_ACT_NONE, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
        }
    };
    epi(std::integral_constant<int, 0>{});
    if constexpr (NG > 1) epi(std::integral_constant<int, 1>{});
}

template <int NG, int PT>
void launch_pwk_direct_pt(ConvArgs& a, int mode, hipStream_t s) {
    a.m_blocks = (a.M + 64 * PT - 1) / (64 * PT);
    const dim3 grid((unsigned)(a.m_blocks * a.n_blocks));
    if (a.gate) hipLaunchKernelGGL((pwk_direct_kernel<NG, true, PT>), grid, dim3(256), 0, s, a, mode);
    else hipLaunchKernelGGL((pwk_direct_kernel<NG, false, PT>), grid, dim3(256), 0, s, a, mode);
}

template <int NG>
int launch_pwk_direct(const ConvArgs& a0, int mode, hipStream_t s) {
    ConvArgs a = a0;
    a.n_blocks = (a.Cout_g + 64 * NG - 1) / (64 * NG);
    // fewer 256-pixel workgroups than CUs (7x7 / 13x13 maps with narrow outputs): 64-pixel tiles instead - measured 48 -> 29 us on
    // M = 12 544, N = 208, K = 1 248 (98 -> 392 workgroups), 35 -> 28 us on 43 264 x 128 x 768; 128-pixel tiles in the band up to
    // 2 x CUs measured no gain and are not instantiated (tools/bench_pwk_fill.py, profiles/r02_pwk_small_tiles_ab.log)
    static const int fill = getenv("VIP_PWK_FILL") ? atoi(getenv("VIP_PWK_FILL")) : 256;
    const long wg256 = (long)((a.M + 255) / 256) * a.n_blocks;
    if (wg256 < fill) launch_pwk_direct_pt<NG, 1>(a, mode, s);
#if VIP_BUILD_EXPERIMENTS
    else if (!a.gate && getenv("VIP_PWK_PF2") && atoi(getenv("VIP_PWK_PF2"))) {     // the two-chunks-ahead experiment (read per call)
        a.m_blocks = (a.M + 127) / 128;
        hipLaunchKernelGGL((pwk_direct2_kernel<NG>), dim3((unsigned)(a.m_blocks * a.n_blocks)), dim3(256), 0, s, a, mode);
    }
#endif
    else launch_pwk_direct_pt<NG, 4>(a, mode, s);
    return vip_launch_status("vip_conv2d_nhwc_f16(pwk-direct)");
}

template <int NG, int WN>
int launch_pwk(const ConvArgs& a0, int mode, hipStream_t s) {
    ConvArgs a = a0;
    a.m_blocks = (a.M + 255) / 256;
    a.n_blocks = (a.Cout_g + 64 * NG * WN - 1) / (64 * NG * WN);
    constexpr size_t smem = (2 * 64 * NG * WN + 4 * 64) * 160;     // 2 weight stages + 4 activation images
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pwk_gemm_kernel<NG, WN, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    const dim3 grid((unsigned)(a.m_blocks * a.n_blocks));
    hipLaunchKernelGGL((pwk_gemm_kernel<NG, WN, false>), grid, dim3(256 * WN), smem, s, a, mode);
    return vip_launch_status("vip_conv2d_nhwc_f16(pwk)");
}

template <int NG>
int launch_pwk_conv(const ConvArgs& a0, int mode, int groups, hipStream_t s) {
    ConvArgs a = a0;
    a.m_blocks = (a.M + 255) / 256;
    a.n_blocks = (a.Cout_g + 64 * NG - 1) / (64 * NG);
    constexpr size_t smem = (2 * 64 * NG + 4 * 64) * 160;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pwk_gemm_kernel<NG, 1, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    const dim3 grid((unsigned)(a.m_blocks * a.n_blocks), (unsigned)groups);
    hipLaunchKernelGGL((pwk_gemm_kernel<NG, 1, true>), grid, dim3(256), smem, s, a, mode);
    return vip_launch_status("vip_conv2d_nhwc_f16(pwk-im2col)");
}

#include "gemm8p.hpp"

// ---- Dense / 1x1 layers with at most 256 rows (the squeeze-excite and ECA layers: M = batch) ------------------------
// On the tile kernels such a layer is ONE m-block: 2..16 workgroups walk K chunk by chunk behind a barrier each, 30-100
// us of pure latency while the chip idles.  Here a workgroup owns 16 output channels for all rows, both MFMA operands
// are loaded global -> VGPR directly (no LDS staging, no barrier in the loop), K is split over the workgroup's four
// waves whose partial sums meet in LDS once, and (N/16) x (M/64) workgroups run side by side.
// XSPLIT: the rows are [M][2][K] - a hi and a lo fp16 plane (what gap_kernel / this kernel write with y_lo_off) - and every weight
// fragment meets both: the vector keeps ~22 bits through the squeeze-excite / ECA / split-attention chains.
template <bool XSPLIT>
__global__ __launch_bounds__(256) void rows_gemm_kernel(ConvArgs a) {
    // workgroup = (16 channels, 64 rows): blockIdx.y = row quarter; its 4 waves split K and meet in LDS.  (One workgroup
    // per channel slab streaming ALL rows was bound by a single CU's L2 bandwidth: 1 MB of activations per workgroup.)
    constexpr unsigned OOB = 0xFFFFFFF0u;
    __shared__ float red[3][1][16][64];                      // [kq-1][.][acc register][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mw = 0, kq = wave;
    const int l15 = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int m0 = blockIdx.y * 64;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.w, 0, (unsigned)min((long)0xFFFFFFF0L, 2L * a.Cout_g * a.ldw), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + a.cin_off), 0, (unsigned)min((long)0xFFFFFFF0L, a.x_span_bytes - 2L * a.cin_off), 0x00020000);
    // H2: k-step u = 2 c + plane of the 64-half chunk c; the lane's k-group is the 32-byte (hi, lo) pair lq of the chunk
    const unsigned w_off = (n0 + l15 < a.Cout_g) ? (unsigned)(((n0 + l15) * a.ldw + lq * (H2 ? 16 : 8)) * 2) : 0xFFFF0000u;
    unsigned x_off[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int m = m0 + p * 16 + l15;
        x_off[p] = m < a.M ? (unsigned)((m * a.ldx + lq * (H2 ? 16 : 8)) * 2) : 0xFFFF0000u;
    }
    f32x4 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nks = H2 ? 2 * ((a.K + 63) >> 6) : (a.K + 31) >> 5;
    const int per = ((nks + 3) / 4 + 3) & ~3;                // k-steps per K quarter, multiple of the unroll
    const int ks_lo = kq * per, ks_hi = min(nks, ks_lo + per);
    if (m0 < a.M) {
        for (int ks0 = ks_lo; ks0 < ks_hi; ks0 += 4) {
            U4H8 wf[4], xf[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kst = ks0 + u;
                const bool ok = (kst < ks_hi) & ((H2 ? (kst >> 1) * 64 + lq * 16 + (kst & 1) * 8 : kst * 32 + lq * 8) < a.K);
                const unsigned kb = H2 ? (unsigned)((kst >> 1) * 128 + (kst & 1) * 16) : (unsigned)(kst * 64);
                wf[u].u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? w_off + kb : OOB, 0, 0));
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    xf[u][p].u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? x_off[p] + kb : OOB, 0, 0));
            }
            if constexpr (H2) {     // (ks0, per: multiples of 4, so u = 0 / 2 are hi planes and u + 1 their lo planes)
#pragma unroll
                for (int u = 0; u < 4; u += 2)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u + 1].h, xf[u][p].h, acc[p], 0, 0, 0);
                        acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u].h, xf[u + 1][p].h, acc[p], 0, 0, 0);
                        acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u].h, xf[u][p].h, acc[p], 0, 0, 0);
                    }
                continue;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u].h, xf[u][p].h, acc[p], 0, 0, 0);
            if constexpr (XSPLIT) {     // the lo plane: K halfs further in the same row
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = (ks0 + u < ks_hi) & ((ks0 + u) * 32 + lq * 8 < a.K);
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        xf[u][p].u = __builtin_bit_cast(
                            uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? x_off[p] + (ks0 + u) * 64 + 2 * a.K : OOB, 0, 0));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u].h, xf[u][p].h, acc[p], 0, 0, 0);
            }
        }
    }
    if (kq > 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[kq - 1][mw][p * 4 + r][lane] = acc[p][r];
    }
    __syncthreads();
    if (kq > 0 || m0 >= a.M) return;
    // lane: row m0 + 16 p + l15, channels n0 + 4 lq + (0..3)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int m = m0 + p * 16 + l15;
        if (m >= a.M) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + lq * 4 + r;
            v[r] = acc[p][r] + red[0][mw][p * 4 + r][lane] + red[1][mw][p * 4 + r][lane] + red[2][mw][p * 4 + r][lane] +
                   ((a.bias && n < a.Cout_g) ? a.bias[n] : 0.f);
        }
        if constexpr (H2) {     // 4 consecutive channels of a packed row (Cout % 8 == 0, so the quad is whole or absent)
            if (n0 + lq * 4 < a.Cout_g) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = vip_act_strict(v[r] * a.out_scale, a.act_pre);
                h2_st4(a.y, (long)m * a.ldy + a.cout_off + n0 + lq * 4, o, a.status);
            }
            continue;
        }
        f16* dst = a.y + (long)m * a.ldy + a.cout_off + n0 + lq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = vip_act(v[r], a.act_pre);
        if (n0 + lq * 4 + 3 < a.Cout_g) {
            f16x4 o, ol;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = (f16)v[r];
                ol[r] = (f16)(v[r] - (float)o[r]);
            }
            *reinterpret_cast<f16x4*>(dst) = o;
            if (a.y_lo_off) *reinterpret_cast<f16x4*>(dst + a.y_lo_off) = ol;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n0 + lq * 4 + r < a.Cout_g) {
                    const f16 o = (f16)v[r];
                    dst[r] = o;
                    if (a.y_lo_off) dst[a.y_lo_off + r] = (f16)(v[r] - (float)o);
                }
        }
    }
}

template <int BM, int BN>
int launch(const ConvArgs& a0, int groups, hipStream_t s) {
    ConvArgs a = a0;
    a.m_blocks = (a.M + BM - 1) / BM;
    a.n_blocks = (a.Cout_g + BN - 1) / BN;
    const int nk = (a.K + 63) >> 6;
    const size_t smem = (nk == 1 ? 1 : 2) * (BM + BN) * 128;  // a single k-tile needs no second stage
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + BN) * 128);
        attr_set = true;
    }
    dim3 grid((unsigned)(a.m_blocks * a.n_blocks), 1, (unsigned)groups);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN>), grid, dim3(256), smem, s, a);
    return vip_launch_status("vip_conv2d_nhwc_f16");
}

}  // namespace

// vip_conv2d_kernel_name(): the selection below runs with g_dry set, records the kernel it would launch and returns
static thread_local bool g_dry = false;
static thread_local const char* g_pick = "";
#define VIP_PICK(name, call)          \
    do {                              \
        if (g_dry) {                  \
            g_pick = (name);          \
            return VIP_OK;            \
        }                             \
        return (call);                \
    } while (0)

static int conv2d_impl(const void* x, const void* gate, int y_lo_off, const void* w, const float* bias, const void* residual, void* y,
                       const vip_conv_desc* d, void* stream, const void* w_lo = nullptr, bool x_split = false, float out_scale = 1.f,
                       int* status = nullptr) {
    VIP_REQUIRE(x && w && y && d, VIP_ERR_BAD_ARG, "vip_conv2d_nhwc_f16: null pointer");
    VIP_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->kh > 0 && d->kw > 0 &&
                    d->sh > 0 && d->sw > 0 && d->Ho > 0 && d->Wo > 0 && d->groups > 0 && d->pt >= 0 && d->pl >= 0,
                VIP_ERR_BAD_ARG, "vip_conv2d_nhwc_f16: non-positive dimension");
    VIP_REQUIRE(d->Cin % d->groups == 0 && d->Cout % d->groups == 0, VIP_ERR_BAD_ARG,
                "vip_conv2d_nhwc_f16: channels (%d,%d) not divisible by groups %d", d->Cin, d->Cout, d->groups);
    const int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups;
    VIP_REQUIRE(cin_g % 8 == 0 && cout_g % 8 == 0, VIP_ERR_ALIGNMENT,
                "vip_conv2d_nhwc_f16: Cin/groups=%d and Cout/groups=%d must be multiples of 8", cin_g, cout_g);
    VIP_REQUIRE(d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->ldw % 8 == 0 && d->cin_off % 8 == 0 &&
                    d->cout_off % 8 == 0 && (!residual || (d->ldr % 8 == 0 && d->res_off % 8 == 0)),
                VIP_ERR_ALIGNMENT, "vip_conv2d_nhwc_f16: strides/offsets must be multiples of 8 halfs");
    constexpr int IN = H2 ? 2 : 1;       // halfs per input element (H2: an (hi, lo) pair)
    VIP_REQUIRE(d->ldx >= d->cin_off + d->Cin && d->ldy >= d->cout_off + d->Cout && d->ldw >= IN * d->kh * d->kw * cin_g,
                VIP_ERR_BAD_ARG, "vip_conv2d_nhwc_f16: leading dimension smaller than the channel extent");
    VIP_REQUIRE(!H2 || (d->ldw % 16 == 0 && !w_lo && !y_lo_off && !x_split), VIP_ERR_BAD_ARG,
                "vip_conv2d_nhwc_h2: ldw must be a multiple of 16 halfs; no split variants");
    VIP_REQUIRE((unsigned)d->act_pre <= 4u && (unsigned)d->act_post <= 4u, VIP_ERR_BAD_ARG,
                "vip_conv2d_nhwc_f16: unknown activation code");
    // the caller's Ho/Wo must not read past what padding+kernel imply on the top/left; bottom/right
    // overhang is zero-filled, so any Ho/Wo is memory-safe.
    const long M = (long)d->B * d->Ho * d->Wo;
    VIP_REQUIRE(M < (1L << 31) - 256, VIP_ERR_UNSUPPORTED, "vip_conv2d_nhwc_f16: B*Ho*Wo too large");

    ConvArgs a;
    a.x = (const f16*)x; a.w = (const f16*)w; a.bias = bias; a.res = (const f16*)residual; a.y = (f16*)y;
    a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo;
    // the INPUT side in halfs (H2: twice the logical channels - see the head of this file), the output side in logical channels
    a.Cin_g = cin_g * IN; a.Cout_g = cout_g;
    a.kh = d->kh; a.kw = d->kw; a.sh = d->sh; a.sw = d->sw; a.pt = d->pt; a.pl = d->pl;
    a.ldx = d->ldx * IN; a.ldy = d->ldy; a.ldr = d->ldr; a.ldw = d->ldw;
    a.cin_off = d->cin_off * IN; a.cout_off = d->cout_off; a.res_off = d->res_off;
    a.M = (int)M; a.K = d->kh * d->kw * cin_g * IN;
    a.x_span_bytes = 2L * IN * d->B * d->H * d->W * d->ldx;
    a.y_span_bytes = (long)ESZ * M * d->ldy;
    a.res_span_bytes = (long)ESZ * M * d->ldr;
    a.bias_elems = d->Cout;
    a.out_scale = out_scale;
    a.status = status;
    VIP_REQUIRE(a.y_span_bytes < 0xFFFFFFE0L && a.res_span_bytes < 0xFFFFFFE0L, VIP_ERR_UNSUPPORTED,
                "vip_conv2d_nhwc_f16: output or residual tensor exceeds the 4 GiB buffer-addressing range");
    VIP_REQUIRE(a.x_span_bytes < 0xFFFFFFF0L && 2L * d->Cout * d->ldw < 0xFFFFFFF0L, VIP_ERR_UNSUPPORTED,
                "vip_conv2d_nhwc_f16: input or weight tensor exceeds the 4 GiB buffer-addressing range");
    a.act_pre = d->act_pre; a.act_post = d->act_post;
    a.m_blocks = a.n_blocks = 0;
    a.w_lo = (const f16*)w_lo;
    a.gate = (const f16*)gate;
    a.gate_hw = d->Ho * d->Wo;
    a.y_lo_off = y_lo_off;
    hipStream_t s = (hipStream_t)stream;
    // HBM-bound shapes (short K): a smaller M tile -> 24-48 KB LDS and half the accumulators -> 3-5 workgroups per
    // CU in flight instead of 2, which is what hides the load -> MFMA -> store latency chain of a 1-4 k-tile block.
    // (Tried and measured SLOWER on these shapes: a two-deep register prefetch (+40 VGPRs), an LDS-transposed
    // "fully coalesced" epilogue, and a persistent tile loop that prefetches the next tile under the epilogue
    // (+60 VGPRs): all three trade resident workgroups for in-workgroup overlap, and residency wins.)
    const bool short_k = a.K <= 256;
    static const int pw_mode = getenv("VIP_PW") ? atoi(getenv("VIP_PW")) : 3;
    if (pw_mode && d->groups == 1 && d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->pl == 0 &&
        d->Ho == d->H && d->Wo == d->W) {
        // epilogue variants the pointwise kernels carry: act_pre alone, or residual (+ post-ReLU) with no act_pre
        int mode = -1;
        if (!residual && d->act_post == VIP_ACT_NONE) mode = d->act_pre;
        else if (residual && d->act_pre == VIP_ACT_NONE && d->act_post == VIP_ACT_NONE) mode = 5;
        else if (residual && d->act_pre == VIP_ACT_NONE && d->act_post == VIP_ACT_RELU) mode = 6;
        if (w_lo) {     // hi + lo weights: the streaming kernel is the one that carries them (any M)
            VIP_REQUIRE(mode >= 0 && short_k && !gate && !y_lo_off, VIP_ERR_UNSUPPORTED,
                        "vip_conv2d_hilo_nhwc_f16: 1x1 stride-1 ungrouped, K <= 256, (activation) or (residual [+ReLU]) epilogue");
            VIP_PICK("pw_gemm_kernel", launch_pw_k<4>(a, mode, s));
        }
        if (M <= 256 && !residual && !gate && d->act_post == VIP_ACT_NONE && d->ldy % 4 == 0 && d->cout_off % 4 == 0 &&
            a.x_span_bytes < 0xFFFF0000L - 2L * a.K && 2L * cout_g * d->ldw < 0xFFFF0000L - 2L * a.K) {
            if (g_dry) { g_pick = "rows_gemm_kernel"; return VIP_OK; }
            if (x_split) hipLaunchKernelGGL(rows_gemm_kernel<true>, dim3((unsigned)((cout_g + 15) / 16), (unsigned)((M + 63) / 64)), dim3(256), 0, s, a);
            else hipLaunchKernelGGL(rows_gemm_kernel<false>, dim3((unsigned)((cout_g + 15) / 16), (unsigned)((M + 63) / 64)), dim3(256), 0, s, a);
            return vip_launch_status("vip_conv2d_nhwc_f16(rows)");
        }
        VIP_REQUIRE(!y_lo_off && !x_split, VIP_ERR_UNSUPPORTED, "vip_gemm_split_f16: at most 256 rows, N %% 4 == 0");
        if (mode >= 0 && (pw_mode & 1) && short_k && M >= 65536 && !gate) VIP_PICK("pw_gemm_kernel", launch_pw_k<4>(a, mode, s));
        // deep K, wide N: the LDS-DMA kernel (gemm8p.hpp).  VIP_G8P_MINK: smallest K it takes (0 = never).  1024: in isolation it
        // wins from K = 256 up (+6..+27 %), but it owns a CU (128 KB of LDS, 8 waves x 256 VGPRs, persistent) and the ensemble step
        // runs three member streams - with the K = 256-768 layers on it the STEP was 1.5-2 % slower (58.2 ms vs 57.1-57.4,
        // alternating runs on one box, profiles/r02_gemm8p_mink_streams_ab.log) and a single stream gained nothing either.
        const char* g8_env = getenv("VIP_G8P_MINK");     // read per call: the kernel's own tests lower it
        const int g8_min_k = g8_env ? atoi(g8_env) : 1024;
        if (mode >= 0 && g8_min_k > 0 && !gate && a.K >= g8_min_k && gemm8p_eligible(a) && cout_g % 256 == 0 &&
            (long)((M + 255) / 256) * (cout_g / 256) >= 128)
            VIP_PICK("gemm8p_kernel", launch_gemm8p(a, mode, s));
        if (mode >= 0 && (pw_mode & 2) && a.x_span_bytes < 0xFFFFFFF0L) {
            // (a 64 x 256 wave tile at one wave per SIMD - NG = 4 - measured 15-40 % slower than NG = 2 at two)
            static const int xl_min_k = getenv("VIP_PWK_XLK") ? atoi(getenv("VIP_PWK_XLK")) : 768;
            // (gated convolutions: only the direct kernel carries the gate pipeline - their deep-K cases are small launches)
            // short K, wide N: the activation-resident kernel (pwx_kernel) - an EXPERIMENT, off unless VIP_PWX=1 (read per call: its tests
            // set it).  Measured per shape against pwk_direct_kernel (profiles/r04_pwx_vs_pwk_direct.log): a wash on the fp16 storage
            // (sum of ten ensemble shapes 566 vs 564 us) and 7-35 % slower on the packed storage, where K = 256 needs 128 fragment
            // registers, leaves 16 pixels per wave and makes every 64-pixel workgroup stream all of W out of L2.  The K <= 256 layers are
            // not bound by the fill / drain this kernel removes: their GELU / residual epilogues issue 7 VALU instructions per MFMA.
            const char* pwx_env = getenv("VIP_PWX");
            const int pwx_on = pwx_env ? atoi(pwx_env) : 0;
            if (pwx_on && !gate && cout_g >= 256 && a.K <= (H2 ? 512 : 256) && a.K > 64 && M >= 16384 && a.x_span_bytes < 0xFFFF0000L - 2L * a.K) {
                const int ksc = (a.K + 63) >> 6;
                if (H2) {
                    if (ksc <= 4) VIP_PICK("pwx_kernel", (launch_pwx<4, 2>(a, mode, s)));
                    if (ksc <= 6) VIP_PICK("pwx_kernel", (launch_pwx<6, 1>(a, mode, s)));      // (32 pixels per wave spill at 96 fragment registers)
                    VIP_PICK("pwx_kernel", (launch_pwx<8, 1>(a, mode, s)));
                } else {
                    if (ksc <= 2) VIP_PICK("pwx_kernel", (launch_pwx<2, 2>(a, mode, s)));
                    if (ksc <= 3) VIP_PICK("pwx_kernel", (launch_pwx<3, 2>(a, mode, s)));
                    VIP_PICK("pwx_kernel", (launch_pwx<4, 2>(a, mode, s)));
                }
            }
            if ((a.K < xl_min_k || gate) && a.x_span_bytes < 0xFFFF0000L - 2L * a.K)
                VIP_PICK("pwk_direct_kernel", cout_g <= 64 ? launch_pwk_direct<1>(a, mode, s) : launch_pwk_direct<2>(a, mode, s));
            VIP_REQUIRE(!gate, VIP_ERR_UNSUPPORTED, "vip_conv2d_gated_nhwc_f16: input tensor too large (4 GB - 2K)");
            if (cout_g <= 64) VIP_PICK("pwk_gemm_kernel", (launch_pwk<1, 1>(a, mode, s)));
            // 256 x 256 block tiles (8 waves): 5 % on deep-K layers whose N is a multiple of 256; slower whenever the last
            // 256-channel tile is half empty (N = 384: 258 -> 346 us) or K is short
            static const int wn2_min_k = getenv("VIP_PWK_WN2K") ? atoi(getenv("VIP_PWK_WN2K")) : 1024;
            if (cout_g % 256 == 0 && a.K >= wn2_min_k && (long)((M + 255) / 256) * (cout_g / 256) >= 256) VIP_PICK("pwk_gemm_kernel", (launch_pwk<2, 2>(a, mode, s)));
            VIP_PICK("pwk_gemm_kernel", (launch_pwk<2, 1>(a, mode, s)));
        }
    }
    {   // k x k convolutions on the pointwise kernel with im2col staging (64 px x 128 ch wave tiles, half the LDS fragment
        // reads per MFMA of the 64 x 64 tiles below): stems (Cin <= 16: 277 vs 385 us on the EfficientNet stems) and every
        // layer with at least 32 K pixels - measured per shape on the ensemble after the MFMA-priority change: +3..+25 %
        // (ResNeSt's grouped 3x3: 585 -> 726, 721 -> 853 TF), except the 7 x 7-pixel stages (M = 12 544: 637 -> 497 TF),
        // which keep the tile kernel.  VIP_PWK_CONV: 1 = every eligible conv (tests), -1 = stems only.
        int mode = -1;
        if (!residual && d->act_post == VIP_ACT_NONE) mode = d->act_pre;
        else if (residual && d->act_pre == VIP_ACT_NONE && d->act_post == VIP_ACT_NONE) mode = 5;
        else if (residual && d->act_pre == VIP_ACT_NONE && d->act_post == VIP_ACT_RELU) mode = 6;
        static const int im2col_all = getenv("VIP_PWK_CONV") ? atoi(getenv("VIP_PWK_CONV")) : 0;
        if ((cin_g <= 16 || im2col_all > 0 || (im2col_all == 0 && M >= 32768)) && mode >= 0 && !gate && a.x_span_bytes < 0xFFFFFFF0L && d->H < 30000 &&
            d->W < 30000 && d->pt < 16 && d->pl < 16)
            VIP_PICK("pwk_gemm_kernel(im2col)", cout_g <= 64 ? launch_pwk_conv<1>(a, mode, d->groups, s) : launch_pwk_conv<2>(a, mode, d->groups, s));
    }
    VIP_REQUIRE(!w_lo, VIP_ERR_UNSUPPORTED, "vip_conv2d_hilo_nhwc_f16: only 1x1 stride-1 ungrouped convolutions with K <= 256");
    VIP_REQUIRE(!gate, VIP_ERR_UNSUPPORTED,
                "vip_conv2d_gated_nhwc_f16: only 1x1 stride-1 ungrouped convolutions with (activation) or (residual [+ReLU]) "
                "epilogues take a gate; apply vip_scale_add_act_f16 first");
    if (cout_g <= 64) VIP_PICK("conv_igemm_kernel", short_k ? (launch<64, 64>(a, d->groups, s)) : (launch<128, 64>(a, d->groups, s)));
    VIP_PICK("conv_igemm_kernel", short_k ? (launch<64, 128>(a, d->groups, s)) : (launch<128, 128>(a, d->groups, s)));
}

#if VIP_GEMM_H2
/* Conv2D / Dense on the packed STRICT storage (include/vipcup_hip.h): x, residual, y packed [.., C] (4 bytes per element), w packed
 * rows [Cout][ldw halfs] = per 8 k: [hi x 8][lo x 8] of (W * w_scale), bias = b * w_scale (fp32), out_scale = 1 / w_scale. */
extern "C" int vip_conv2d_nhwc_h2(const void* x, const void* w, const float* bias, const void* residual, void* y, const vip_conv_desc* d,
                                  float out_scale, int* status, void* stream) {
    VIP_REQUIRE(out_scale > 0.f, VIP_ERR_BAD_ARG, "vip_conv2d_nhwc_h2: out_scale must be positive");
    return conv2d_impl(x, nullptr, 0, w, bias, residual, y, d, stream, nullptr, false, out_scale, status);
}

/* The same with a squeeze-excite gate folded into the activation operand: gate packed [B][Cin]; the product (x_hi + x_lo)(g_hi + g_lo) is
 * formed in fp32 per element and split again in registers (pwk_direct_kernel) - only 1x1 stride-1 ungrouped convolutions with an
 * (activation) or (residual [+ReLU]) epilogue, cin_off = 0, ldx = Cin; anything else returns VIP_ERR_UNSUPPORTED. */
extern "C" int vip_conv2d_gated_nhwc_h2(const void* x, const void* gate, const void* w, const float* bias, const void* residual, void* y,
                                        const vip_conv_desc* d, float out_scale, int* status, void* stream) {
    VIP_REQUIRE(gate, VIP_ERR_BAD_ARG, "vip_conv2d_gated_nhwc_h2: null gate");
    VIP_REQUIRE(out_scale > 0.f, VIP_ERR_BAD_ARG, "vip_conv2d_gated_nhwc_h2: out_scale must be positive");
    VIP_REQUIRE(d && d->cin_off == 0 && d->ldx == d->Cin, VIP_ERR_UNSUPPORTED,
                "vip_conv2d_gated_nhwc_h2: the gate indexes the whole input channel axis (cin_off = 0, ldx = Cin)");
    return conv2d_impl(x, gate, 0, w, bias, residual, y, d, stream, nullptr, false, out_scale, status);
}

extern "C" int vip_conv2d_kernel_name_h2(const vip_conv_desc* d, int has_residual, char* name, size_t cap) {
    VIP_REQUIRE(d && name && cap > 0, VIP_ERR_BAD_ARG, "vip_conv2d_kernel_name_h2: null pointer");
    static const char dummy[16] = {0};     // non-null stand-ins: nothing is dereferenced or launched in a dry run
    g_dry = true;
    g_pick = "";
    const int st = conv2d_impl(dummy, nullptr, 0, dummy, nullptr, has_residual ? dummy : nullptr, const_cast<char*>(dummy), d, nullptr);
    g_dry = false;
    if (st != VIP_OK) return st;
    snprintf(name, cap, "%s", g_pick);
    return VIP_OK;
}
#else
extern "C" int vip_conv2d_nhwc_f16(const void* x, const void* w, const float* bias, const void* residual, void* y,
                                   const vip_conv_desc* d, void* stream) {
    return conv2d_impl(x, nullptr, 0, w, bias, residual, y, d, stream);
}

extern "C" int vip_conv2d_gated_nhwc_f16(const void* x, const void* gate, const void* w, const float* bias,
                                         const void* residual, void* y, const vip_conv_desc* d, void* stream) {
    VIP_REQUIRE(gate, VIP_ERR_BAD_ARG, "vip_conv2d_gated_nhwc_f16: null gate");
    VIP_REQUIRE(d && d->cin_off == 0 && d->ldx == d->Cin, VIP_ERR_UNSUPPORTED,
                "vip_conv2d_gated_nhwc_f16: the gate indexes the whole input channel axis (cin_off = 0, ldx = Cin)");
    return conv2d_impl(x, gate, 0, w, bias, residual, y, d, stream);
}

extern "C" int vip_conv2d_hilo_nhwc_f16(const void* x, const void* w_hi, const void* w_lo, const float* bias,
                                        const void* residual, void* y, const vip_conv_desc* d, void* stream) {
    VIP_REQUIRE(w_lo, VIP_ERR_BAD_ARG, "vip_conv2d_hilo_nhwc_f16: null w_lo");
    return conv2d_impl(x, nullptr, 0, w_hi, bias, residual, y, d, stream, w_lo);
}

extern "C" int vip_conv2d_kernel_name(const vip_conv_desc* d, int has_residual, int has_gate, int has_w_lo, char* name,
                                      size_t cap) {
    VIP_REQUIRE(d && name && cap > 0, VIP_ERR_BAD_ARG, "vip_conv2d_kernel_name: null pointer");
    static const char dummy[16] = {0};     // non-null stand-ins: nothing is dereferenced or launched in a dry run
    g_dry = true;
    g_pick = "";
    const int st = conv2d_impl(dummy, has_gate ? dummy : nullptr, 0, dummy, nullptr, has_residual ? dummy : nullptr,
                               const_cast<char*>(dummy), d, nullptr, has_w_lo ? dummy : nullptr);
    g_dry = false;
    if (st != VIP_OK) return st;
    snprintf(name, cap, "%s", g_pick);
    return VIP_OK;
}

extern "C" int vip_gemm_bias_act_f16(const void* A, const void* W, const float* bias, const void* residual,
                                     void* C, int M, int N, int K, int lda, int ldw, int ldc, int ldr,
                                     int act_pre, int act_post, void* stream) {
    vip_conv_desc d;
    d.B = M; d.H = 1; d.W = 1; d.Cin = K; d.Cout = N; d.kh = d.kw = 1; d.sh = d.sw = 1; d.pt = d.pl = 0;
    d.Ho = d.Wo = 1; d.groups = 1; d.ldx = lda; d.cin_off = 0; d.ldy = ldc; d.cout_off = 0; d.ldr = ldr;
    d.res_off = 0; d.ldw = ldw; d.act_pre = act_pre; d.act_post = act_post;
    return vip_conv2d_nhwc_f16(A, W, bias, residual, C, &d, stream);
}

extern "C" int vip_gemm_split_f16(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda,
                                  int ldw, int act, void* stream) {
    VIP_REQUIRE(M > 0 && M <= 256, VIP_ERR_UNSUPPORTED, "vip_gemm_split_f16: M=%d (1..256 rows)", M);
    vip_conv_desc d;
    d.B = M; d.H = 1; d.W = 1; d.Cin = K; d.Cout = N; d.kh = d.kw = 1; d.sh = d.sw = 1; d.pt = d.pl = 0;
    d.Ho = d.Wo = 1; d.groups = 1; d.ldx = lda; d.cin_off = 0; d.ldy = 2 * N; d.cout_off = 0; d.ldr = 0;
    d.res_off = 0; d.ldw = ldw; d.act_pre = act; d.act_post = VIP_ACT_NONE;
    return conv2d_impl(A, nullptr, N, W, bias, nullptr, C, &d, stream);
}

extern "C" int vip_gemm_split2_f16(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int ldw, int act,
                                   void* stream) {
    VIP_REQUIRE(M > 0 && M <= 256, VIP_ERR_UNSUPPORTED, "vip_gemm_split2_f16: M=%d (1..256 rows)", M);
    VIP_REQUIRE(K % 8 == 0, VIP_ERR_ALIGNMENT, "vip_gemm_split2_f16: K must be a multiple of 8");
    vip_conv_desc d;
    d.B = M; d.H = 1; d.W = 1; d.Cin = K; d.Cout = N; d.kh = d.kw = 1; d.sh = d.sw = 1; d.pt = d.pl = 0;
    d.Ho = d.Wo = 1; d.groups = 1; d.ldx = 2 * K; d.cin_off = 0; d.ldy = 2 * N; d.cout_off = 0; d.ldr = 0;
    d.res_off = 0; d.ldw = ldw; d.act_pre = act; d.act_post = VIP_ACT_NONE;
    return conv2d_impl(A, nullptr, N, W, bias, nullptr, C, &d, stream, nullptr, true);
}
#endif  // VIP_GEMM_H2
