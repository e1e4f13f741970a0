// GPU half of the input pipeline (dataset/dataset.py:22-39):
//   quantised DCT coefficients -> dequantise + 8x8 ISLOW IDCT -> component planes (u8)
//   -> "fancy" chroma upsampling + YCbCr->RGB -> RGB u8   [== tf.image.decode_jpeg(channels=3)]
//   -> cast f32 -> bicubic resize (TF legacy kernel, half-pixel centres) -> /255 -> f16 NHWC (C padded)
//   (+ the TTA flips / gray of dataset/augment.py:115-120,142-146).
//
// The integer stages restate libjpeg(-turbo)'s jidctint.c (jpeg_idct_islow), jdsample.c
// (h2v1/h2v2/h1v2_fancy_upsample) and jdcolor.c (ycc_rgb_convert) bit for bit: same constants, same
// rounding, same edge replication.  All of it is HBM/latency-bound byte work: 16-byte coefficient loads,
// no MFMA.
#include "common.hpp"

namespace {

// ---- jidctint.c constants (CONST_BITS = 13, PASS1_BITS = 2) ----
constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270,
              F_0_899976223 = 7373, F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137,
              F_1_961570560 = 16069, F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 1-D pass; SHIFT_IN: the even-part DC terms are shifted by CONST_BITS, the outputs descaled by OUT_SHIFT
template <int OUT_SHIFT>
__device__ __forceinline__ void idct_1d(const int (&in)[8], int (&out)[8]) {
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * F_0_541196100;
    int tmp2 = z1 + z3 * (-F_1_847759065);
    int tmp3 = z1 + z2 * F_0_765366865;
    z2 = in[0];
    z3 = in[4];
    int tmp0 = (z2 + z3) << CONST_BITS;
    int tmp1 = (z2 - z3) << CONST_BITS;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7];
    tmp1 = in[5];
    tmp2 = in[3];
    tmp3 = in[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * F_1_175875602;
    tmp0 *= F_0_298631336;
    tmp1 *= F_2_053119869;
    tmp2 *= F_3_072711026;
    tmp3 *= F_1_501321110;
    z1 *= -F_0_899976223;
    z2 *= -F_2_562915447;
    z3 *= -F_1_961570560;
    z4 *= -F_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    out[0] = descale(tmp10 + tmp3, OUT_SHIFT);
    out[7] = descale(tmp10 - tmp3, OUT_SHIFT);
    out[1] = descale(tmp11 + tmp2, OUT_SHIFT);
    out[6] = descale(tmp11 - tmp2, OUT_SHIFT);
    out[2] = descale(tmp12 + tmp1, OUT_SHIFT);
    out[5] = descale(tmp12 - tmp1, OUT_SHIFT);
    out[3] = descale(tmp13 + tmp0, OUT_SHIFT);
    out[4] = descale(tmp13 - tmp0, OUT_SHIFT);
}

// idct_range_limit[x & RANGE_MASK] of jdmaster.c prepare_range_limit_table
__device__ __forceinline__ int idct_range_limit(int x) {
    x &= 1023;
    if (x < 128) return x + 128;
    if (x < 512) return 255;
    if (x < 896) return 0;
    return x - 896;
}

// one thread = one 8x8 block; grid.y = image, grid.x covers the image's blocks (all components)
__global__ __launch_bounds__(64) void jpeg_idct_kernel(const int16_t* __restrict__ coef,
                                                       const vip_jpeg_desc* __restrict__ desc,
                                                       uint8_t* __restrict__ planes) {
    const vip_jpeg_desc& d = desc[blockIdx.y];
    int blk = blockIdx.x * 64 + threadIdx.x;
    int c = 0;
    for (; c < d.ncomp; ++c) {
        const int nb = d.blocks_w[c] * d.blocks_h[c];
        if (blk < nb) break;
        blk -= nb;
    }
    if (c >= d.ncomp) return;
    const int bw = d.blocks_w[c];
    const int brow = blk / bw, bcol = blk - brow * bw;
    const int16_t* src = coef + d.coef_off[c] + (long)blk * 64;
    int ws[8][8];  // [row][col]
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint4 v = *reinterpret_cast<const uint4*>(src + r * 8);
        const int16_t* s = reinterpret_cast<const int16_t*>(&v);
#pragma unroll
        for (int k = 0; k < 8; ++k) ws[r][k] = (int)s[k] * (int)d.qt[c][r * 8 + k];
    }
    // pass 1: columns
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = ws[r][col];
        idct_1d<CONST_BITS - PASS1_BITS>(in, out);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r][col] = out[r];
    }
    // pass 2: rows, range-limit, store
    uint8_t* dst = planes + d.coef_off[c] + ((long)brow * 8) * (bw * 8) + bcol * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int out[8];
        idct_1d<CONST_BITS + PASS1_BITS + 3>(ws[r], out);
        uint2 pk;
        pk.x = idct_range_limit(out[0]) | (idct_range_limit(out[1]) << 8) | (idct_range_limit(out[2]) << 16) |
               (idct_range_limit(out[3]) << 24);
        pk.y = idct_range_limit(out[4]) | (idct_range_limit(out[5]) << 8) | (idct_range_limit(out[6]) << 16) |
               (idct_range_limit(out[7]) << 24);
        *reinterpret_cast<uint2*>(dst + (long)r * (bw * 8)) = pk;
    }
}

// fancy-upsampled chroma sample at output pixel (y, x); plane holds the real (un-padded) region dw x dh
__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ pl, int stride, int dw, int dh, int hs, int vs,
                                         int y, int x) {
    // hs/vs = luma/chroma ratio (1 or 2)
    if (hs == 1 && vs == 1) return pl[(long)y * stride + x];
    // jdsample.c jinit_upsampler: h2v1 / h2v2 planes no wider than 2 samples get plain replication, not the fancy filter
    if (hs == 2 && dw <= 2) return pl[(long)(vs == 2 ? y >> 1 : y) * stride + (x >> 1)];
    if (hs == 2 && vs == 1) {  // h2v1_fancy_upsample
        const uint8_t* row = pl + (long)y * stride;
        const int i = x >> 1;
        const int v = row[i];
        if (x & 1) return (i == dw - 1) ? v : (v * 3 + row[i + 1] + 2) >> 2;
        return (i == 0) ? v : (v * 3 + row[i - 1] + 1) >> 2;
    }
    const int iy = y >> 1;
    int ny = (y & 1) ? iy + 1 : iy - 1;  // the nearer neighbour row; edges replicate (jdmainct.c context rows)
    ny = ny < 0 ? 0 : (ny > dh - 1 ? dh - 1 : ny);
    const uint8_t* r0 = pl + (long)iy * stride;
    const uint8_t* r1 = pl + (long)ny * stride;
    if (hs == 1) {  // h1v2_fancy_upsample
        return (r0[x] * 3 + r1[x] + ((y & 1) ? 2 : 1)) >> 2;
    }
    // h2v2_fancy_upsample
    const int i = x >> 1;
    const int cur = r0[i] * 3 + r1[i];
    if (x & 1) {
        if (i == dw - 1) return (cur * 4 + 7) >> 4;
        return (cur * 3 + (r0[i + 1] * 3 + r1[i + 1]) + 7) >> 4;
    }
    if (i == 0) return (cur * 4 + 8) >> 4;
    return (cur * 3 + (r0[i - 1] * 3 + r1[i - 1]) + 8) >> 4;
}

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one thread = one output pixel; grid.z = image
__global__ __launch_bounds__(256) void jpeg_color_kernel(const uint8_t* __restrict__ planes,
                                                         const vip_jpeg_desc* __restrict__ desc,
                                                         uint8_t* __restrict__ rgb, int maxH, int maxW) {
    const vip_jpeg_desc& d = desc[blockIdx.z];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= d.width || y >= d.height) return;
    const uint8_t* py = planes + d.coef_off[0];
    const int Y = py[(long)y * (d.blocks_w[0] * 8) + x];
    int R = Y, G = Y, B = Y;
    if (d.ncomp == 3) {
        const int hs = d.hsamp[0] / d.hsamp[1], vs = d.vsamp[0] / d.vsamp[1];
        const int dw = (d.width * d.hsamp[1] + d.hsamp[0] - 1) / d.hsamp[0];
        const int dh = (d.height * d.vsamp[1] + d.vsamp[0] - 1) / d.vsamp[0];
        const int c1 = chroma_at(planes + d.coef_off[1], d.blocks_w[1] * 8, dw, dh, hs, vs, y, x);
        const int c2 = chroma_at(planes + d.coef_off[2], d.blocks_w[2] * 8, dw, dh, hs, vs, y, x);
        if (d.rgb_coded) {                 // the planes are R, G, B already (Adobe transform 0): null colour conversion
            G = c1;
            B = c2;
        } else {
            const int cb = c1 - 128, cr = c2 - 128;
            // jdcolor.c build_ycc_rgb_table: SCALEBITS = 16, ONE_HALF = 32768
            R = clamp255(Y + ((91881 * cr + 32768) >> 16));
            B = clamp255(Y + ((116130 * cb + 32768) >> 16));
            G = clamp255(Y + ((-22554 * cb + 32768 + (-46802) * cr) >> 16));
        }
    }
    uint8_t* o = rgb + (((long)blockIdx.z * maxH + y) * maxW + x) * 3;
    o[0] = (uint8_t)R;
    o[1] = (uint8_t)G;
    o[2] = (uint8_t)B;
}

// ---- tf.image.resize(method="bicubic") legacy kernel: weights and clamped indices for one output coordinate
struct Taps {
    int idx[4];
    float w[4];
};
__device__ __forceinline__ Taps bicubic_taps(int out_loc, int in_size, float scale, const float* __restrict__ table) {
    Taps t;
    const float in_loc_f = __fsub_rn(__fmul_rn(__fadd_rn((float)out_loc, 0.5f), scale), 0.5f);
    const float fl = floorf(in_loc_f);
    const int in_loc = (int)fl;
    const float delta = __fsub_rn(in_loc_f, fl);
    const int offset = (int)lrintf(__fmul_rn(delta, 1024.f));
    const float w[4] = {table[offset * 2 + 1], table[offset * 2], table[(1024 - offset) * 2], table[(1024 - offset) * 2 + 1]};
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int want = in_loc - 1 + k;
        const int got = want < 0 ? 0 : (want > in_size - 1 ? in_size - 1 : want);
        t.idx[k] = got;
        t.w[k] = (got == want) ? w[k] : 0.f;
        sum = __fadd_rn(sum, t.w[k]);
    }
    if (fabsf(sum) >= 1000.f * 1.17549435e-38f) {
        const float inv = __fdiv_rn(1.f, sum);
#pragma unroll
        for (int k = 0; k < 4; ++k) t.w[k] = __fmul_rn(t.w[k], inv);
    }
    return t;
}

// T = f16 (fast path) or float (STRICT path: the fp32 values tf.image.resize / 255 produces, not rounded)
template <typename T>
__global__ __launch_bounds__(256) void resize_norm_kernel(const uint8_t* __restrict__ rgb, const int* __restrict__ sizes,
                                                          const float* __restrict__ table, T* __restrict__ out,
                                                          int maxH, int maxW, int outH, int outW, int c_out) {
    const int n = blockIdx.z;
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= outW || oy >= outH) return;
    const int h = sizes[2 * n], w = sizes[2 * n + 1];
    T* o = out + (((long)n * outH + oy) * outW + ox) * c_out;
    auto store = [&](const float (&res)[3]) {
        if (c_out == 8 && sizeof(T) == 2) {          // one 16-byte store per pixel
            U4H8 v;
            v.u = make_uint4(0, 0, 0, 0);
            v.e[0] = (f16)res[0]; v.e[1] = (f16)res[1]; v.e[2] = (f16)res[2];
            *reinterpret_cast<uint4*>(o) = v.u;
        } else {
            for (int c = 0; c < c_out; ++c) o[c] = (T)(c < 3 ? res[c] : 0.f);
        }
    };
    if (h == outH && w == outW) {
        // same size (every 200x200 image at a 200x200 member): the legacy bicubic kernel's taps at scale 1 are exactly
        // (0, 1, 0, 0) in both axes, so the general path below returns p / 255 bit for bit - read 3 bytes instead of 48
        const uint8_t* p = rgb + ((long)n * maxH * maxW + (long)oy * maxW + ox) * 3;
        const float res[3] = {__fdiv_rn((float)p[0], 255.f), __fdiv_rn((float)p[1], 255.f), __fdiv_rn((float)p[2], 255.f)};
        store(res);
        return;
    }
    const Taps ty = bicubic_taps(oy, h, __fdiv_rn((float)h, (float)outH), table);
    const Taps tx = bicubic_taps(ox, w, __fdiv_rn((float)w, (float)outW), table);
    const uint8_t* img = rgb + (long)n * maxH * maxW * 3;
    float res[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float rowv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint8_t* p = img + ((long)ty.idx[r] * maxW) * 3 + c;
            float v = __fmul_rn((float)p[tx.idx[0] * 3], tx.w[0]);
            v = __fadd_rn(v, __fmul_rn((float)p[tx.idx[1] * 3], tx.w[1]));
            v = __fadd_rn(v, __fmul_rn((float)p[tx.idx[2] * 3], tx.w[2]));
            v = __fadd_rn(v, __fmul_rn((float)p[tx.idx[3] * 3], tx.w[3]));
            rowv[r] = v;
        }
        float v = __fmul_rn(rowv[0], ty.w[0]);
        v = __fadd_rn(v, __fmul_rn(rowv[1], ty.w[1]));
        v = __fadd_rn(v, __fmul_rn(rowv[2], ty.w[2]));
        v = __fadd_rn(v, __fmul_rn(rowv[3], ty.w[3]));
        res[c] = __fdiv_rn(v, 255.f);
    }
    store(res);
}

// TTA: flags bit0 hflip, bit1 vflip, bit2 gray (tf.image.rgb_to_grayscale weights 0.2989/0.5870/0.1140)
template <typename T>
__global__ __launch_bounds__(256) void tta_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                  const int* __restrict__ flags, int H, int W, int C) {
    const int b = blockIdx.z;
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= W || oy >= H) return;
    const int f = flags[b];
    const int sx = (f & 1) ? W - 1 - ox : ox;
    const int sy = (f & 2) ? H - 1 - oy : oy;
    const T* s = x + (((long)b * H + sy) * W + sx) * C;
    T* o = y + (((long)b * H + oy) * W + ox) * C;
    if (f & 4) {
        const float g = 0.2989f * (float)s[0] + 0.5870f * (float)s[1] + 0.1140f * (float)s[2];
        o[0] = o[1] = o[2] = (T)g;
        for (int c = 3; c < C; ++c) o[c] = s[c];
    } else {
        for (int c = 0; c < C; ++c) o[c] = s[c];
    }
}

}  // namespace

extern "C" int vip_jpeg_idct_rgb_u8(const int16_t* coef, const vip_jpeg_desc* desc, int n, int max_blocks,
                                    uint8_t* planes_ws, uint8_t* rgb_u8, int maxH, int maxW, void* stream) {
    VIP_REQUIRE(coef && desc && planes_ws && rgb_u8, VIP_ERR_BAD_ARG, "vip_jpeg_idct_rgb_u8: null pointer");
    VIP_REQUIRE(n > 0 && max_blocks > 0 && maxH > 0 && maxW > 0, VIP_ERR_BAD_ARG, "vip_jpeg_idct_rgb_u8: bad size");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((max_blocks + 63) / 64, n), dim3(64), 0, s, coef, desc, planes_ws);
    int st = vip_launch_status("vip_jpeg_idct_rgb_u8(idct)");
    if (st != VIP_OK) return st;
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((maxW + 63) / 64, (maxH + 3) / 4, n), dim3(256), 0, s,
                       (const uint8_t*)planes_ws, desc, rgb_u8, maxH, maxW);
    return vip_launch_status("vip_jpeg_idct_rgb_u8(color)");
}

extern "C" int vip_bicubic_table_f32(float* table_h) {
    // tensorflow/core/kernels/image/resize_bicubic_op.cc InitCoeffsTable(a = -0.5): evaluated in double,
    // stored as float
    VIP_REQUIRE(table_h, VIP_ERR_BAD_ARG, "vip_bicubic_table_f32: null pointer");
    const double a = -0.5;
    for (int i = 0; i <= 1024; ++i) {
        double x = i * 1.0 / 1024;
        table_h[i * 2] = (float)(((a + 2) * x - (a + 3)) * x * x + 1);
        x += 1.0;
        table_h[i * 2 + 1] = (float)(((a * x - 5 * a) * x + 8 * a) * x - 4 * a);
    }
    return VIP_OK;
}

extern "C" int vip_resize_bicubic_norm_f16(const uint8_t* rgb_u8, const int32_t* sizes_hw, const float* table,
                                           int n, int maxH, int maxW, void* out, int outH, int outW, int c_out,
                                           void* stream) {
    VIP_REQUIRE(rgb_u8 && sizes_hw && table && out, VIP_ERR_BAD_ARG, "vip_resize_bicubic_norm_f16: null pointer");
    VIP_REQUIRE(n > 0 && maxH > 0 && maxW > 0 && outH > 0 && outW > 0 && c_out >= 3, VIP_ERR_BAD_ARG,
                "vip_resize_bicubic_norm_f16: bad size");
    hipLaunchKernelGGL(resize_norm_kernel<f16>, dim3((outW + 63) / 64, (outH + 3) / 4, n), dim3(256), 0, (hipStream_t)stream,
                       rgb_u8, sizes_hw, table, (f16*)out, maxH, maxW, outH, outW, c_out);
    return vip_launch_status("vip_resize_bicubic_norm_f16");
}

extern "C" int vip_resize_bicubic_norm_s32(const uint8_t* rgb_u8, const int32_t* sizes_hw, const float* table,
                                           int n, int maxH, int maxW, float* out, int outH, int outW, int c_out,
                                           void* stream) {
    VIP_REQUIRE(rgb_u8 && sizes_hw && table && out, VIP_ERR_BAD_ARG, "vip_resize_bicubic_norm_s32: null pointer");
    VIP_REQUIRE(n > 0 && maxH > 0 && maxW > 0 && outH > 0 && outW > 0 && c_out >= 3, VIP_ERR_BAD_ARG,
                "vip_resize_bicubic_norm_s32: bad size");
    hipLaunchKernelGGL(resize_norm_kernel<float>, dim3((outW + 63) / 64, (outH + 3) / 4, n), dim3(256), 0, (hipStream_t)stream,
                       rgb_u8, sizes_hw, table, out, maxH, maxW, outH, outW, c_out);
    return vip_launch_status("vip_resize_bicubic_norm_s32");
}

extern "C" int vip_tta_augment_f16(const void* x, void* y, const int32_t* flags, int B, int H, int W, int C,
                                   void* stream) {
    VIP_REQUIRE(x && y && flags, VIP_ERR_BAD_ARG, "vip_tta_augment_f16: null pointer");
    VIP_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 3, VIP_ERR_BAD_ARG, "vip_tta_augment_f16: bad size");
    VIP_REQUIRE(x != y, VIP_ERR_BAD_ARG, "vip_tta_augment_f16: in-place flips are not supported");
    hipLaunchKernelGGL(tta_kernel<f16>, dim3((W + 63) / 64, (H + 3) / 4, B), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)x, (f16*)y, flags, H, W, C);
    return vip_launch_status("vip_tta_augment_f16");
}

extern "C" int vip_tta_augment_s32(const float* x, float* y, const int32_t* flags, int B, int H, int W, int C, void* stream) {
    VIP_REQUIRE(x && y && flags, VIP_ERR_BAD_ARG, "vip_tta_augment_s32: null pointer");
    VIP_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 3, VIP_ERR_BAD_ARG, "vip_tta_augment_s32: bad size");
    VIP_REQUIRE(x != y, VIP_ERR_BAD_ARG, "vip_tta_augment_s32: in-place flips are not supported");
    hipLaunchKernelGGL(tta_kernel<float>, dim3((W + 63) / 64, (H + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, x, y, flags, H, W, C);
    return vip_launch_status("vip_tta_augment_s32");
}
