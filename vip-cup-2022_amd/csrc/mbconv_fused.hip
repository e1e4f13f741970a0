// MBConv front half in one launch:   z = act( dwconv_kxk,s( zero_pad( act( W_e . x + b_e ) ) ) + b_d )
//
// Replaces the expand Conv2D(1x1)+BN+swish followed by DepthwiseConv2D(k, strides)+BN+swish of kecam's inverted residual block
// (efficientnet_v2.py:63-66 and :85-90).  As two launches the EXPANDED tensor (4-6x the block's input channels, at the block's
// INPUT resolution) is written by the 1x1 convolution and read back by the depthwise one: 2 x 737 MB for EfficientNet-B4's first
// stride-2 block at batch 256 against 123 MB of input and 184 MB of output.  Here a workgroup computes the expanded activations
// of one spatial tile (with its k - 1 halo) for a slab of NS channels on the matrix cores straight into LDS - rounded to fp16
// exactly where the 1x1 kernel rounds them - and runs the depthwise filter out of LDS; the expanded tensor never exists in HBM.
//   phase A  Y[pixel][NS] = act(W_e X^T + b_e): A operand = weight rows (LDS, optional two-term hi + lo weights as in
//            vip_conv2d_hilo_nhwc_f16), B operand = the pixel's channels, a 16-byte run of its NHWC row loaded global -> VGPR;
//            halo pixels outside the image become ZERO (the depthwise conv pads its input, not the block's);
//   phase B  depthwise k x k from the LDS tile, fp32 filters (LDS), two horizontally adjacent outputs x 8 channels per lane.
// Workgroups are persistent over spatial tiles of one channel slab (weights staged once).  The halo is recomputed per tile
// ((10/8)^2 = 1.56x the 1x1 FLOPs for k = 3, 2.25x for k = 5 - K is at most 128 here, the layer is nowhere near MFMA-bound).
//
// MEASURED (B = 256, tools/bench_mbconv.py, profiles/r02_mbconv_fused_vs_two_launches.log): 0.41-0.77x the speed of the two launches
// (EfficientNet-B4 stage 2: 317 vs 246 us, its first stride-2 block 861 vs 611 us) - so the host calls it only with
// VIP_MBCONV_FUSED=1.  The traffic does drop as planned, but the two separate kernels are HBM-bound and evaluate their swish
// activations (v_exp + v_rcp per element, quarter-rate) under memory time, while here 10.5 k swish evaluations per 8 x 8 x 64 tile
// (6.4 k of them on the halo-extended expand output) are ~1 800 VALU cycles per tile and CU before any depthwise FMA: the fused
// kernel's floor is ~1.5x the two launches at best, and as written (no cross-tile prefetch) it does not reach it.
#include "common.hpp"

namespace {

struct MbArgs {
    const f16* x;
    const f16* we;      // [Ce][ldw] expand weights (fp16, BN folded)
    const f16* we_lo;   // optional low parts (two-term weights) or NULL
    const float* be;    // [Ce] or NULL
    const float* wd;    // [k][k][Ce] fp32 depthwise filter
    const float* bd;    // [Ce] or NULL
    f16* z;
    int B, H, W, Cin, Ce, ldw, pt, pl, Ho, Wo, act_e, act_d;
    int tiles_x, tiles_y, n_tiles;
    long x_bytes;
};

template <int KS, int S, int NS>
struct MbCfg {
    static constexpr int TOY = S == 1 ? 8 : 4, TOX = 8;
    static constexpr int TIY = (TOY - 1) * S + KS, TIX = (TOX - 1) * S + KS;
    static constexpr int NPIX = TIY * TIX, MT = (NPIX + 15) / 16;
    static constexpr int YSTR = NS * 2 + 16;              // bytes per pixel row of the LDS tile (16-byte pad: bank spread)
    static constexpr int NT = NS / 16;
    static constexpr int ITEMS = (TOY * TOX / 2) * (NS / 8);
    static_assert(ITEMS <= 256, "one depthwise item per thread");
};

template <int KS, int S, int NS, bool HILO>
__global__ __launch_bounds__(256, 4) void mbconv_expand_dw_kernel(MbArgs a, int kp, int wstr) {
    using Cf = MbCfg<KS, S, NS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: expand weights hi [NS][wstr] (+ lo), depthwise filter fp32 [KS*KS][NS], biases [2][NS], tile [MT*16][YSTR]
    unsigned char* whi = smem;
    unsigned char* wlo = whi + NS * wstr;
    float* wdl = reinterpret_cast<float*>(wlo + (HILO ? NS * wstr : 0));
    float* bel = wdl + KS * KS * NS;
    float* bdl = bel + NS;
    unsigned char* yt = reinterpret_cast<unsigned char*>(bdl + NS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, Q = lane >> 4;
    const int c0 = blockIdx.y * NS;
    const int cpr = kp >> 3;                                  // 16-byte chunks per staged weight row
    for (int i = tid; i < NS * cpr; i += 256) {
        const int n = i / cpr, c = i - n * cpr;
        const bool in = c * 8 < a.Cin;                        // zero-padded K: the activation run may hold a neighbour's channels
        uint4 v = {0u, 0u, 0u, 0u};
        if (in) v = *reinterpret_cast<const uint4*>(a.we + (long)(c0 + n) * a.ldw + c * 8);
        *reinterpret_cast<uint4*>(whi + n * wstr + c * 16) = v;
        if (HILO) {
            uint4 u = {0u, 0u, 0u, 0u};
            if (in) u = *reinterpret_cast<const uint4*>(a.we_lo + (long)(c0 + n) * a.ldw + c * 8);
            *reinterpret_cast<uint4*>(wlo + n * wstr + c * 16) = u;
        }
    }
    for (int i = tid; i < KS * KS * NS; i += 256) wdl[i] = a.wd[(long)(i / NS) * a.Ce + c0 + (i % NS)];
    for (int i = tid; i < NS; i += 256) {
        bel[i] = a.be ? a.be[c0 + i] : 0.f;
        bdl[i] = a.bd ? a.bd[c0 + i] : 0.f;
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)a.x_bytes, 0x00020000);
    const int ksteps = kp >> 5;
    const int tiles_img = a.tiles_x * a.tiles_y;

    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int b = tile / tiles_img, rem = tile - b * tiles_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int oy0 = ty * Cf::TOY, ox0 = tx * Cf::TOX;
        const int iy0 = oy0 * S - a.pt, ix0 = ox0 * S - a.pl;

        // ---- phase A: expanded activations of the (halo-extended) tile -> LDS
        for (int mt = wave; mt < Cf::MT; mt += 4) {
            const int p = mt * 16 + l15;
            const int py = p / Cf::TIX, px = p - py * Cf::TIX;
            const int gy = iy0 + py, gx = ix0 + px;
            const bool inimg = (p < Cf::NPIX) & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W);
            const unsigned xoff = inimg ? (unsigned)(((((long)b * a.H + gy) * a.W + gx) * a.Cin + Q * 8) * 2) : 0xFFFFFFF0u;
            f32x4 acc[Cf::NT];
#pragma unroll
            for (int nt = 0; nt < Cf::NT; ++nt) acc[nt] = *reinterpret_cast<const f32x4*>(bel + nt * 16 + 4 * Q);
            for (int ks = 0; ks < ksteps; ++ks) {
                U4H8 xf;
                xf.u = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, inimg ? xoff + ks * 64 : 0xFFFFFFF0u, 0, 0));
#pragma unroll
                for (int nt = 0; nt < Cf::NT; ++nt) {
                    U4H8 wf;
                    wf.u = *reinterpret_cast<const uint4*>(whi + (nt * 16 + l15) * wstr + ks * 64 + Q * 16);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf.h, xf.h, acc[nt], 0, 0, 0);
                    if (HILO) {
                        wf.u = *reinterpret_cast<const uint4*>(wlo + (nt * 16 + l15) * wstr + ks * 64 + Q * 16);
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf.h, xf.h, acc[nt], 0, 0, 0);
                    }
                }
            }
            // lane: pixel p = l15 of the m-tile, channels nt*16 + 4Q + (0..3)
#pragma unroll
            for (int nt = 0; nt < Cf::NT; ++nt) {
                f16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = inimg ? (f16)vip_act(acc[nt][r], a.act_e) : (f16)0.f;
                *reinterpret_cast<f16x4*>(yt + p * Cf::YSTR + (nt * 16 + 4 * Q) * 2) = o;
            }
        }
        __syncthreads();

        // ---- phase B: depthwise from the tile; item = (two adjacent outputs of one row, 8 channels)
        if (tid < Cf::ITEMS) {
            constexpr int NCH = NS / 8;
            const int ch = tid % NCH, pp = tid / NCH;
            const int oy = pp / (Cf::TOX / 2), ox = (pp - oy * (Cf::TOX / 2)) * 2;
            f32x2 acc[2][4];
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[o][e] = (f32x2){bdl[ch * 8 + 2 * e], bdl[ch * 8 + 2 * e + 1]};
#pragma unroll 1                                       // one filter row at a time: unrolled, hipcc hoists every LDS read (450 VGPRs for k = 5)
            for (int r = 0; r < KS; ++r) {
                f32x2 xr[KS + S][4];
#pragma unroll
                for (int q = 0; q < KS + S; ++q) {
                    U4H8 v;
                    v.u = *reinterpret_cast<const uint4*>(yt + ((oy * S + r) * Cf::TIX + ox * S + q) * Cf::YSTR + ch * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[q][e] = (f32x2){(float)v.e[2 * e], (float)v.e[2 * e + 1]};
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const float4 w0 = *reinterpret_cast<const float4*>(wdl + (r * KS + s) * NS + ch * 8);
                    const float4 w1 = *reinterpret_cast<const float4*>(wdl + (r * KS + s) * NS + ch * 8 + 4);
                    const f32x2 wv[4] = {{w0.x, w0.y}, {w0.z, w0.w}, {w1.x, w1.y}, {w1.z, w1.w}};
#pragma unroll
                    for (int o = 0; o < 2; ++o)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[o][e] = xr[o * S + s][e] * wv[e] + acc[o][e];
                }
            }
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int gy = oy0 + oy, gx = ox0 + ox + o;
                if (gy < a.Ho && gx < a.Wo) {
                    U4H8 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ov.e[2 * e] = (f16)vip_act(acc[o][e].x, a.act_d);
                        ov.e[2 * e + 1] = (f16)vip_act(acc[o][e].y, a.act_d);
                    }
                    *reinterpret_cast<uint4*>(a.z + (((long)b * a.Ho + gy) * a.Wo + gx) * a.Ce + c0 + ch * 8) = ov.u;
                }
            }
        }
        __syncthreads();
    }
}

int mb_cu_count() {
    static const int n = [] {
        int dev = 0;
        hipDeviceProp_t pr;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
                   ? pr.multiProcessorCount : 256;
    }();
    return n;
}

template <int KS, int S, int NS, bool HILO>
int launch_mb(MbArgs a, hipStream_t s) {
    using Cf = MbCfg<KS, S, NS>;
    const int kp = (a.Cin + 31) / 32 * 32;
    int chunks = kp / 8;
    while ((chunks & 3) != 2) ++chunks;                      // row stride == 32 (mod 64) bytes: conflict-free ds_read_b128 fragments
    const int wstr = chunks * 16;
    const size_t smem = (size_t)NS * wstr * (HILO ? 2 : 1) + (size_t)(KS * KS * NS + 2 * NS) * 4 + (size_t)Cf::MT * 16 * Cf::YSTR;
    a.tiles_x = (a.Wo + Cf::TOX - 1) / Cf::TOX;
    a.tiles_y = (a.Ho + Cf::TOY - 1) / Cf::TOY;
    a.n_tiles = a.B * a.tiles_x * a.tiles_y;
    const int nslab = a.Ce / NS;
    int per_slab = (4 * mb_cu_count() + nslab - 1) / nslab;  // ~4 resident workgroups per CU over all slabs
    if (per_slab > a.n_tiles) per_slab = a.n_tiles;
    if (per_slab < 1) per_slab = 1;
    auto kern = mbconv_expand_dw_kernel<KS, S, NS, HILO>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_done = true;
    }
    if (smem > 96 * 1024) return 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)per_slab, (unsigned)nslab), dim3(256), smem, s, a, kp, wstr);
    return vip_launch_status("vip_mbconv_expand_dw_f16");
}

template <int KS, int S>
int dispatch_ns(const MbArgs& a, bool hilo, hipStream_t s) {
#define VIP_MB(NS_)                                                     \
    if (a.Ce % NS_ == 0)                                                \
        return hilo ? launch_mb<KS, S, NS_, true>(a, s) : launch_mb<KS, S, NS_, false>(a, s);
    VIP_MB(64) VIP_MB(48) VIP_MB(32)
#undef VIP_MB
    return 1;
}

}  // namespace

extern "C" int vip_mbconv_expand_dw_supported(int Cin, int Ce, int k, int stride) {
    return (Cin % 8 == 0 && Cin <= 128 && (Ce % 64 == 0 || Ce % 48 == 0 || Ce % 32 == 0) && (k == 3 || k == 5) && (stride == 1 || stride == 2))
               ? 1 : 0;
}

extern "C" int vip_mbconv_expand_dw_f16(const void* x, const void* we, const void* we_lo, const float* be, const float* wd,
                                        const float* bd, void* z, int B, int H, int W, int Cin, int Ce, int ldw, int k, int stride,
                                        int pt, int pl, int Ho, int Wo, int act_e, int act_d, void* stream) {
    VIP_REQUIRE(x && we && wd && z, VIP_ERR_BAD_ARG, "vip_mbconv_expand_dw_f16: null pointer");
    VIP_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && pt >= 0 && pl >= 0 && (unsigned)act_e <= 4u && (unsigned)act_d <= 4u,
                VIP_ERR_BAD_ARG, "vip_mbconv_expand_dw_f16: bad argument");
    VIP_REQUIRE(vip_mbconv_expand_dw_supported(Cin, Ce, k, stride), VIP_ERR_UNSUPPORTED,
                "vip_mbconv_expand_dw_f16: Cin=%d (<= 128, %% 8), Ce=%d (%% 32), k=%d (3, 5), stride=%d (1, 2)", Cin, Ce, k, stride);
    VIP_REQUIRE(ldw % 8 == 0 && ldw >= Cin, VIP_ERR_ALIGNMENT, "vip_mbconv_expand_dw_f16: ldw must be a multiple of 8 and >= Cin");
    VIP_REQUIRE(2L * B * H * W * Cin < 0xFFFFFFF0L, VIP_ERR_UNSUPPORTED, "vip_mbconv_expand_dw_f16: input larger than 4 GiB");
    VIP_REQUIRE((Ho - 1) * stride + k - pt <= H + k && (Wo - 1) * stride + k - pl <= W + k, VIP_ERR_BAD_ARG,
                "vip_mbconv_expand_dw_f16: output size does not match the padding");
    MbArgs a;
    a.x = (const f16*)x; a.we = (const f16*)we; a.we_lo = (const f16*)we_lo; a.be = be; a.wd = wd; a.bd = bd; a.z = (f16*)z;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Ce = Ce; a.ldw = ldw; a.pt = pt; a.pl = pl; a.Ho = Ho; a.Wo = Wo;
    a.act_e = act_e; a.act_d = act_d; a.x_bytes = 2L * B * H * W * Cin;
    hipStream_t s = (hipStream_t)stream;
    int st = 1;
    if (k == 3 && stride == 1) st = dispatch_ns<3, 1>(a, we_lo != nullptr, s);
    else if (k == 3 && stride == 2) st = dispatch_ns<3, 2>(a, we_lo != nullptr, s);
    else if (k == 5 && stride == 1) st = dispatch_ns<5, 1>(a, we_lo != nullptr, s);
    else if (k == 5 && stride == 2) st = dispatch_ns<5, 2>(a, we_lo != nullptr, s);
    if (st == 1) {
        vip_set_error("vip_mbconv_expand_dw_f16: shape needs more than 96 KiB of LDS");
        return VIP_ERR_UNSUPPORTED;
    }
    return st;
}
