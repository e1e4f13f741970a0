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

namespace {

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
};

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
            for (int j = 0; j < 8; ++j) {
                v[j] = act_t<ACT>(acc[mt][h * 2 + (j >> 2)][j & 3] + bv[h][j >> 2][j & 3]) + (float)r.e[j];
                if (post_relu) v[j] = fmaxf(v[j], 0.f);
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
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<uint4*>(sa + swz_x(row0 + 32 * i, chunk)) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<uint4*>(sb + swz_w(row0 + 32 * i, chunk)) = rb[i];
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
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt].h, xf[mt].h, acc[mt][nt], 0, 0, 0);
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

template <int BM, int BN>
int launch(const ConvArgs& a0, int groups, hipStream_t s) {
    ConvArgs a = a0;
    a.m_blocks = (a.M + BM - 1) / BM;
    a.n_blocks = (a.Cout_g + BN - 1) / BN;
    const int nk = (a.K + 63) >> 6;
    const size_t smem = (nk == 1 ? 1 : 2) * (BM + BN) * 128;  // a single k-tile needs no second stage
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + BN) * 128);
        attr_set = true;
    }
    dim3 grid((unsigned)(a.m_blocks * a.n_blocks), 1, (unsigned)groups);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN>), grid, dim3(256), smem, s, a);
    return vip_launch_status("vip_conv2d_nhwc_f16");
}

}  // namespace

extern "C" int vip_conv2d_nhwc_f16(const void* x, const void* w, const float* bias, const void* residual,
                                   void* y, const vip_conv_desc* d, void* stream) {
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
    VIP_REQUIRE(d->ldx >= d->cin_off + d->Cin && d->ldy >= d->cout_off + d->Cout && d->ldw >= d->kh * d->kw * cin_g,
                VIP_ERR_BAD_ARG, "vip_conv2d_nhwc_f16: leading dimension smaller than the channel extent");
    VIP_REQUIRE((unsigned)d->act_pre <= 4u && (unsigned)d->act_post <= 4u, VIP_ERR_BAD_ARG,
                "vip_conv2d_nhwc_f16: unknown activation code");
    // the caller's Ho/Wo must not read past what padding+kernel imply on the top/left; bottom/right
    // overhang is zero-filled, so any Ho/Wo is memory-safe.
    const long M = (long)d->B * d->Ho * d->Wo;
    VIP_REQUIRE(M < (1L << 31) - 256, VIP_ERR_UNSUPPORTED, "vip_conv2d_nhwc_f16: B*Ho*Wo too large");

    ConvArgs a;
    a.x = (const f16*)x; a.w = (const f16*)w; a.bias = bias; a.res = (const f16*)residual; a.y = (f16*)y;
    a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo;
    a.Cin_g = cin_g; a.Cout_g = cout_g;
    a.kh = d->kh; a.kw = d->kw; a.sh = d->sh; a.sw = d->sw; a.pt = d->pt; a.pl = d->pl;
    a.ldx = d->ldx; a.ldy = d->ldy; a.ldr = d->ldr; a.ldw = d->ldw;
    a.cin_off = d->cin_off; a.cout_off = d->cout_off; a.res_off = d->res_off;
    a.M = (int)M; a.K = d->kh * d->kw * cin_g;
    a.x_span_bytes = 2L * d->B * d->H * d->W * d->ldx;
    a.y_span_bytes = 2L * M * d->ldy;
    a.res_span_bytes = 2L * M * d->ldr;
    a.bias_elems = d->Cout;
    VIP_REQUIRE(a.y_span_bytes < 0xFFFFFFF0L && a.res_span_bytes < 0xFFFFFFF0L, VIP_ERR_UNSUPPORTED,
                "vip_conv2d_nhwc_f16: output or residual tensor exceeds the 4 GiB buffer-addressing range");
    VIP_REQUIRE(a.x_span_bytes < 0xFFFFFFF0L && 2L * d->Cout * d->ldw < 0xFFFFFFF0L, VIP_ERR_UNSUPPORTED,
                "vip_conv2d_nhwc_f16: input or weight tensor exceeds the 4 GiB buffer-addressing range");
    a.act_pre = d->act_pre; a.act_post = d->act_post;
    a.m_blocks = a.n_blocks = 0;
    hipStream_t s = (hipStream_t)stream;
    // HBM-bound shapes (short K): a smaller M tile -> 24-48 KB LDS and half the accumulators -> 3-5 workgroups per
    // CU in flight instead of 2, which is what hides the load -> MFMA -> store latency chain of a 1-4 k-tile block.
    // (Tried and measured SLOWER on these shapes: a two-deep register prefetch (+40 VGPRs), an LDS-transposed
    // "fully coalesced" epilogue, and a persistent tile loop that prefetches the next tile under the epilogue
    // (+60 VGPRs): all three trade resident workgroups for in-workgroup overlap, and residency wins.)
    const bool short_k = a.K <= 256;
    if (cout_g <= 64) return short_k ? launch<64, 64>(a, d->groups, s) : launch<128, 64>(a, d->groups, s);
    return short_k ? launch<64, 128>(a, d->groups, s) : launch<128, 128>(a, d->groups, s);
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
