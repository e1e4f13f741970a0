// Depthwise convolution, stride 1, register-tiled: each thread owns a T x TW patch of output pixels for 8
// channels and walks the (4+K-1) input rows ONCE (one 16-byte load and one fp16->fp32 conversion per input
// chunk, 6.25 loads per output chunk for K=7 instead of 17.5), with the K*K*C filter pre-converted to fp32 in
// LDS (every lane of a channel chunk reads the same address: broadcast, conflict-free).
// The op is VALU-bound on CDNA4 (49 FMAs per output element for the ConvNeXt 7x7 and no matrix-core mapping
// for a per-channel filter), so the point of the structure is to spend the VALU on FMAs, not on conversions.
// Replaces tf.keras.layers.DepthwiseConv2D (tfimm convnext.py:192-198; gcvit feature.py:93,133; kecam
// efficientnet_v2.py:85) for stride 1; strided cases stay on the simple kernel of pointwise.hip.
#include "common.hpp"

namespace {

static int VIP_DW_BLOCKS_PER_CU = 8;

typedef float f32x2 __attribute__((ext_vector_type(2)));


template <int T, int TW, int ACT, bool POOL>
__device__ __forceinline__ void dw_store(f32x2 (&acc)[T][TW][4], const float (&bv)[8], f16* __restrict__ y, int b, int oy0,
                                         int ox0, int Ho, int Wo, int C, int c0, f32x2 (&ps)[4]) {
#pragma unroll
    for (int oy = 0; oy < T; ++oy) {
        const int gy = oy0 + oy;
        if (gy >= Ho) break;
#pragma unroll
        for (int ox = 0; ox < TW; ++ox) {
            const int gx = ox0 + ox;
            if (gx >= Wo) break;
            U4H8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 v = vip_act2<ACT>(acc[oy][ox][e] + (f32x2){bv[2 * e], bv[2 * e + 1]});
                o.e[2 * e] = (f16)v.x;
                o.e[2 * e + 1] = (f16)v.y;
                if constexpr (POOL) ps[e] += v;        // the activated fp32 values, before the fp16 rounding of the store
            }
            *reinterpret_cast<uint4*>(y + (((long)b * Ho + gy) * Wo + gx) * C + c0) = o.u;
        }
    }
}

// POOL: the kernel also leaves per-workgroup partial sums of its (activated, fp32) outputs for the squeeze-excite pool that follows
// a depthwise convolution in MBConv / GCViT FeatExtract blocks (efficientnet_v2.py:85-97, gcvit feature.py:46-70), so that the gate
// kernel does not read the whole map again: tile groups are then image-aligned (`gpi` groups per image, the last one partly idle) and
// group g of image b writes partials[(b * gpi + g) * C + c] - a fixed summation order, no atomics: bit-reproducible.
template <int K, int T, int TW, bool WHOLE, bool POOL>
__global__ __launch_bounds__(256, 2) void dwconv_tile_kernel(const f16* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, f16* __restrict__ y,
                                                             int B, int H, int W, int C, int pt, int pl, int Ho,
                                                             int Wo, int act, int cb_chunks, int tiles_x, int tiles_y,
                                                             long n_tiles, long x_bytes, float* __restrict__ partials, int gpi) {
    constexpr int P = T + K - 1;         // input patch rows
    constexpr int PW = TW + K - 1;       // input patch columns (TW output columns per thread)
    extern __shared__ __attribute__((aligned(16))) float wlds[];  // [K*K][cb_chunks*8] fp32

    const int c8_0 = blockIdx.y * cb_chunks;               // first channel chunk of this block
    const int C8 = C >> 3;
    const int nch = min(cb_chunks, C8 - c8_0);              // chunks this block really has
    // stage the filter slice as fp32
    for (int i = threadIdx.x; i < K * K * nch * 8; i += 256) {
        const int tap = i / (nch * 8), c = i - tap * (nch * 8);
        wlds[tap * (cb_chunks * 8) + c] = w[(long)tap * C + c8_0 * 8 + c];
    }
    __syncthreads();

    const int tiles_per_block = 256 / cb_chunks;
    int lc = threadIdx.x % cb_chunks;                       // chunk inside the block (fastest: coalesced rows)
    const int lt = threadIdx.x / cb_chunks;
    const bool lane_ok = lt < tiles_per_block && lc < nch;
    if constexpr (POOL) {
        // every lane reaches the barriers of the reduction below: a lane without work runs a tile below the image (all stores and
        // pooled sums skipped) on a channel chunk that exists
        if (lc >= nch) lc = nch - 1;
    } else {
        if (!lane_ok) return;
    }
    const int c0 = (c8_0 + lc) * 8;
    const float* wl = wlds + lc * 8;
    // buffer descriptor over the whole input: out-of-image taps use an out-of-range offset and read as zero in
    // hardware (a `cond ? load : 0` in source makes hipcc predicate + serialise every load, see conv_igemm.hip)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)x_bytes, 0x00020000);

    // The block walks tile groups with a stride (the filter staging above is paid once per block).  XCD-aware: blocks
    // b, b+8, ... share an XCD and its L2, so each XCD gets one CONTIGUOUS band of tile groups - vertically adjacent
    // tiles re-read K-1 of their T+K-1 input rows, and with a plain grid stride those neighbours sit on other XCDs and
    // every XCD pulls the halo rows over the fabric again (PMC: 2.7x the algorithmic read bytes).
    const long n_groups = POOL ? (long)B * gpi : (n_tiles + tiles_per_block - 1) / tiles_per_block;
    long g_lo = 0, g_hi = n_groups, g_step = gridDim.x, g_first = blockIdx.x;
    if (gridDim.x >= 8) {
        const int xcd = blockIdx.x & 7;
        const long chunk = (n_groups + 7) / 8;
        g_lo = xcd * chunk;
        g_hi = min(n_groups, g_lo + chunk);
        g_step = (gridDim.x + 7 - xcd) >> 3;           // blocks that share this XCD id
        g_first = g_lo + (blockIdx.x >> 3);
    }
    for (long grp = g_first; grp < g_hi; grp += g_step) {
    int tx, ty, b;
    if constexpr (POOL) {
        b = (int)(grp / gpi);
        const int ti = (int)(grp - (long)b * gpi) * tiles_per_block + lt;
        const bool work = lane_ok && ti < tiles_x * tiles_y;
        tx = work ? ti % tiles_x : 0;
        ty = work ? ti / tiles_x : tiles_y;                  // first output row Ho or beyond: nothing stored, nothing pooled
    } else {
        const long tile = grp * tiles_per_block + lt;
        if (tile >= n_tiles) break;
        tx = (int)(tile % tiles_x);
        ty = (int)((tile / tiles_x) % tiles_y);
        b = (int)(tile / ((long)tiles_x * tiles_y));
    }
    const int oy0 = ty * T, ox0 = tx * TW;

    f32x2 acc[T][TW][4];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] = (f32x2){0.f, 0.f};

    const unsigned xoff0 = (unsigned)((((long)b * H * W) * C + c0) * 2);
    auto load_row = [&](int iy, uint4 (&raw)[PW]) {
        const int gy = oy0 - pt + iy;
        const bool row_ok = (unsigned)gy < (unsigned)H;
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int gx = ox0 - pl + q;
            const bool ok = row_ok & ((unsigned)gx < (unsigned)W);
            const unsigned off = ok ? xoff0 + (unsigned)((gy * W + gx) * C * 2) : 0xFFFFFFF0u;
            raw[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
    };
    // multiply-accumulate one converted input row into the output rows it feeds
    auto mac_row = [&](int iy, const uint4 (&rawrow)[PW]) {
        f32x2 xr[PW][4];
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            U4H8 v;
            v.u = rawrow[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xr[q][e][0] = (float)v.e[2 * e];
                xr[q][e][1] = (float)v.e[2 * e + 1];
            }
        }
#pragma unroll
        for (int oy = 0; oy < T; ++oy) {
            const int r = iy - oy;                 // wave-uniform
            if (r < 0 || r >= K) continue;
            // weights of tap (r, s+1) are fetched from LDS before the FMAs of tap (r, s)
            const float* wrow = wl + (r * K) * (cb_chunks * 8);
            float4 w0 = *reinterpret_cast<const float4*>(wrow);
            float4 w1 = *reinterpret_cast<const float4*>(wrow + 4);
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const int sn = s + 1 < K ? s + 1 : s;
                const float4 n0 = *reinterpret_cast<const float4*>(wrow + sn * (cb_chunks * 8));
                const float4 n1 = *reinterpret_cast<const float4*>(wrow + sn * (cb_chunks * 8) + 4);
                const f32x2 wv[4] = {{w0.x, w0.y}, {w0.z, w0.w}, {w1.x, w1.y}, {w1.z, w1.w}};
#pragma unroll
                for (int ox = 0; ox < TW; ++ox)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[oy][ox][e] = xr[ox + s][e] * wv[e] + acc[oy][ox][e];
                w0 = n0;
                w1 = n1;
            }
        }
    };
    if constexpr (WHOLE) {
        // small filters: the whole (T+K-1) x (TW+K-1) patch is requested up front - a row's ~100 VALU instructions
        // cannot cover a memory round trip, P x PW loads in flight per lane can
        uint4 raw[P][PW];
#pragma unroll
        for (int iy = 0; iy < P; ++iy) load_row(iy, raw[iy]);
#pragma unroll
        for (int iy = 0; iy < P; ++iy) mac_row(iy, raw[iy]);
    } else {
        uint4 raw[PW];
        load_row(0, raw);
        // NOT unrolled (unrolling makes hipcc hoist every load and spill); the NEXT row's loads are issued before the
        // current row is converted and multiplied, so their latency hides under ~400 VALU instructions
#pragma unroll 1
        for (int iy = 0; iy < P; ++iy) {
            uint4 nxt[PW];
            load_row(iy + 1 < P ? iy + 1 : iy, nxt);
            mac_row(iy, raw);
#pragma unroll
            for (int q = 0; q < PW; ++q) raw[q] = nxt[q];
        }
    }

    float bv[8];     // loaded per tile (L1-resident): keeping it across the tile loop costs 8 VGPRs of a full budget
    {
        const float4 b0 = bias ? *reinterpret_cast<const float4*>(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 b1 = bias ? *reinterpret_cast<const float4*>(bias + c0 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
        bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
    f32x2 ps[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    switch (act) {   // one straight-line, packed-math epilogue per activation
        case VIP_ACT_RELU: dw_store<T, TW, VIP_ACT_RELU, POOL>(acc, bv, y, b, oy0, ox0, Ho, Wo, C, c0, ps); break;
        case VIP_ACT_SILU: dw_store<T, TW, VIP_ACT_SILU, POOL>(acc, bv, y, b, oy0, ox0, Ho, Wo, C, c0, ps); break;
        case VIP_ACT_GELU: dw_store<T, TW, VIP_ACT_GELU, POOL>(acc, bv, y, b, oy0, ox0, Ho, Wo, C, c0, ps); break;
        case VIP_ACT_SIGMOID: dw_store<T, TW, VIP_ACT_SIGMOID, POOL>(acc, bv, y, b, oy0, ox0, Ho, Wo, C, c0, ps); break;
        default: dw_store<T, TW, VIP_ACT_NONE, POOL>(acc, bv, y, b, oy0, ox0, Ho, Wo, C, c0, ps); break;
    }
    if constexpr (POOL) {
        // tiles of the group -> one row of partial sums: [tile][channel] through LDS, then one thread per channel adds the tiles
        // in index order
        float* red = wlds + K * K * cb_chunks * 8;           // [tiles_per_block][cb_chunks * 8]
        __syncthreads();                                     // the previous group's readers are done
        if (lane_ok) {
            float* r = red + lt * (cb_chunks * 8) + lc * 8;
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x2*>(r + 2 * e) = ps[e];
        }
        __syncthreads();
        if ((int)threadIdx.x < nch * 8) {
            float sum = 0.f;
            for (int t = 0; t < tiles_per_block; ++t) sum += red[t * (cb_chunks * 8) + threadIdx.x];
            partials[grp * C + c8_0 * 8 + threadIdx.x] = sum;
        }
    }
    }   // tile loop
}

// ---- the same register-tiled scheme on the packed STRICT storage (common.hpp: 8 channels = [hi x 8][lo x 8], 4 bytes per element) ----
// A thread owns a T x TW patch of output pixels for FOUR channels (half a 32-byte group: 8 + 8 bytes per pixel) - with eight, the raw
// row, its prefetched successor, the joined fp32 row and the accumulators do not fit 256 registers.  Values are joined to fp32 once
// per input element (hi + lo), the filter is fp32 in LDS as above, accumulation and activation are fp32 (vip_act_strict), the result
// is split again on the store and checked against the fp16 range.  Replaces the same DepthwiseConv2D call sites (strict mode).
template <int K, int T, int TW, bool WHOLE>
__global__ __launch_bounds__(256, 2) void dwconv_tile_h2_kernel(const char* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, char* __restrict__ y, int B, int H, int W,
                                                                int C, int pt, int pl, int Ho, int Wo, int act, int cb_chunks, int tiles_x,
                                                                int tiles_y, long n_tiles, long x_bytes, int* status) {
    constexpr int P = T + K - 1;
    constexpr int PW = TW + K - 1;
    extern __shared__ __attribute__((aligned(16))) float wlds[];  // [K*K][cb_chunks*4] fp32

    const int c4_0 = blockIdx.y * cb_chunks;               // first 4-channel chunk of this block
    const int C4 = C >> 2;
    const int nch = min(cb_chunks, C4 - c4_0);
    for (int i = threadIdx.x; i < K * K * nch * 4; i += 256) {
        const int tap = i / (nch * 4), c = i - tap * (nch * 4);
        wlds[tap * (cb_chunks * 4) + c] = w[(long)tap * C + c4_0 * 4 + c];
    }
    __syncthreads();

    const int tiles_per_block = 256 / cb_chunks;
    const int lc = threadIdx.x % cb_chunks;
    const int lt = threadIdx.x / cb_chunks;
    if (!(lt < tiles_per_block && lc < nch)) return;
    const int c0 = (c4_0 + lc) * 4;
    const unsigned coff = (unsigned)((c0 >> 3) * 32 + ((c0 >> 2) & 1) * 8);      // byte offset of the hi quad inside a pixel; lo: + 16
    const float* wl = wlds + lc * 4;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)x_bytes, 0x00020000);

    const long n_groups = (n_tiles + tiles_per_block - 1) / tiles_per_block;
    long g_lo = 0, g_hi = n_groups, g_step = gridDim.x, g_first = blockIdx.x;
    if (gridDim.x >= 8) {                                   // one contiguous band of tile groups per XCD (halo rows stay in its L2)
        const int xcd = blockIdx.x & 7;
        const long chunk = (n_groups + 7) / 8;
        g_lo = xcd * chunk;
        g_hi = min(n_groups, g_lo + chunk);
        g_step = (gridDim.x + 7 - xcd) >> 3;
        g_first = g_lo + (blockIdx.x >> 3);
    }
    bool bad = false;
    for (long grp = g_first; grp < g_hi; grp += g_step) {
        const long tile = grp * tiles_per_block + lt;
        if (tile >= n_tiles) break;
        const int tx = (int)(tile % tiles_x);
        const int ty = (int)((tile / tiles_x) % tiles_y);
        const int b = (int)(tile / ((long)tiles_x * tiles_y));
        const int oy0 = ty * T, ox0 = tx * TW;

        f32x2 acc[T][TW][2];
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < TW; ++j) acc[i][j][0] = acc[i][j][1] = (f32x2){0.f, 0.f};

        const unsigned xoff0 = (unsigned)(((long)b * H * W) * C * 4) + coff;
        auto load_row = [&](int iy, uint2 (&rh)[PW], uint2 (&rl)[PW]) {
            const int gy = oy0 - pt + iy;
            const bool row_ok = (unsigned)gy < (unsigned)H;
#pragma unroll
            for (int q = 0; q < PW; ++q) {
                const int gx = ox0 - pl + q;
                const bool ok = row_ok & ((unsigned)gx < (unsigned)W);
                const unsigned off = ok ? xoff0 + (unsigned)((gy * W + gx) * C * 4) : 0xFFFFFFE0u;
                rh[q] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0));
                rl[q] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rx, off + 16u, 0, 0));
            }
        };
        auto mac_row = [&](int iy, const uint2 (&rh)[PW], const uint2 (&rl)[PW]) {
            f32x2 xr[PW][2];
#pragma unroll
            for (int q = 0; q < PW; ++q) {
                const f16x4 h = __builtin_bit_cast(f16x4, rh[q]), l = __builtin_bit_cast(f16x4, rl[q]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    xr[q][e][0] = (float)h[2 * e] + (float)l[2 * e];
                    xr[q][e][1] = (float)h[2 * e + 1] + (float)l[2 * e + 1];
                }
            }
#pragma unroll
            for (int oy = 0; oy < T; ++oy) {
                const int r = iy - oy;                 // wave-uniform
                if (r < 0 || r >= K) continue;
                const float* wrow = wl + (r * K) * (cb_chunks * 4);
                float4 w0 = *reinterpret_cast<const float4*>(wrow);
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const int sn = s + 1 < K ? s + 1 : s;
                    const float4 n0 = *reinterpret_cast<const float4*>(wrow + sn * (cb_chunks * 4));
                    const f32x2 wv[2] = {{w0.x, w0.y}, {w0.z, w0.w}};
#pragma unroll
                    for (int ox = 0; ox < TW; ++ox)
#pragma unroll
                        for (int e = 0; e < 2; ++e) acc[oy][ox][e] = xr[ox + s][e] * wv[e] + acc[oy][ox][e];
                    w0 = n0;
                }
            }
        };
        if constexpr (WHOLE) {
            uint2 rh[P][PW], rl[P][PW];
#pragma unroll
            for (int iy = 0; iy < P; ++iy) load_row(iy, rh[iy], rl[iy]);
#pragma unroll
            for (int iy = 0; iy < P; ++iy) mac_row(iy, rh[iy], rl[iy]);
        } else {
            uint2 rh[PW], rl[PW];
            load_row(0, rh, rl);
#pragma unroll 1
            for (int iy = 0; iy < P; ++iy) {
                uint2 nh[PW], nl[PW];
                load_row(iy + 1 < P ? iy + 1 : iy, nh, nl);
                mac_row(iy, rh, rl);
#pragma unroll
                for (int q = 0; q < PW; ++q) {
                    rh[q] = nh[q];
                    rl[q] = nl[q];
                }
            }
        }

        const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int oy = 0; oy < T; ++oy) {
            const int gy = oy0 + oy;
            if (gy >= Ho) break;
#pragma unroll
            for (int ox = 0; ox < TW; ++ox) {
                const int gx = ox0 + ox;
                if (gx >= Wo) break;
                float v[4] = {acc[oy][ox][0][0] + bv.x, acc[oy][ox][0][1] + bv.y, acc[oy][ox][1][0] + bv.z, acc[oy][ox][1][1] + bv.w};
                f16x4 oh, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = vip_act_strict(v[e], act);
                    oh[e] = (f16)v[e];
                    ol[e] = (f16)(v[e] - (float)oh[e]);
                    bad |= !(fabsf(v[e]) <= VIP_H2_MAX);
                }
                char* dst = y + (((long)b * Ho + gy) * Wo + gx) * C * 4 + coff;
                *reinterpret_cast<f16x4*>(dst) = oh;
                *reinterpret_cast<f16x4*>(dst + 16) = ol;
            }
        }
    }
    if (bad && status) *status = VIP_H2_OVERFLOW;
}

template <int K, int T, int TW, bool WHOLE>
int launch_tile_h2(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int pt, int pl, int Ho, int Wo,
                   int act, int* status, hipStream_t s) {
    const int C4 = C / 4;
    int cb = C4 < 32 ? C4 : 32;                                  // 4-channel chunks per block: <= 128 channels
    if (C4 % 24 == 0 && C4 % 32 != 0) cb = 24;                    // ConvNeXt widths 96/192/384/768
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + T - 1) / T;
    const long n_tiles = (long)B * tiles_x * tiles_y;
    const int tiles_per_block = 256 / cb;
    const int gyc = (C4 + cb - 1) / cb;
    long gx = (n_tiles + tiles_per_block - 1) / tiles_per_block;
    const long gx_cap = (256L * VIP_DW_BLOCKS_PER_CU + gyc - 1) / gyc;
    if (gx > gx_cap) gx = gx_cap;
    const size_t smem = (size_t)K * K * cb * 4 * sizeof(float);
    hipLaunchKernelGGL((dwconv_tile_h2_kernel<K, T, TW, WHOLE>), dim3((unsigned)gx, (unsigned)gyc), dim3(256), smem, s, (const char*)x, w,
                       bias, (char*)y, B, H, W, C, pt, pl, Ho, Wo, act, cb, tiles_x, tiles_y, n_tiles, 4L * B * H * W * C, status);
    return vip_launch_status("vip_dwconv2d_nhwc_h2(tile)");
}

template <int K, int T, int TW, bool WHOLE>
int launch_tile(const f16* x, const float* w, const float* bias, f16* y, int B, int H, int W, int C, int pt, int pl,
                int Ho, int Wo, int act, hipStream_t s, float* partials = nullptr, int parts = 0) {
    const int C8 = C / 8;
    // channel chunks per block: a divisor-friendly width <= 16 chunks (128 channels) that wastes few lanes
    int cb = C8 < 16 ? C8 : 16;
    if (C8 % 12 == 0 && C8 % 16 != 0) cb = 12;                   // ConvNeXt widths 96/192/384/768
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + T - 1) / T;
    const long n_tiles = (long)B * tiles_x * tiles_y;
    const int tiles_per_block = 256 / cb;
    const int gyc = (C8 + cb - 1) / cb;
    const int gpi = (tiles_x * tiles_y + tiles_per_block - 1) / tiles_per_block;       // image-aligned tile groups (pooling form)
    long gx = partials ? (long)B * gpi : (n_tiles + tiles_per_block - 1) / tiles_per_block;
    const long gx_cap = (256L * VIP_DW_BLOCKS_PER_CU + gyc - 1) / gyc;     // ~resident blocks of the whole chip
    if (gx > gx_cap) gx = gx_cap;
    const size_t smem = (size_t)K * K * cb * 8 * sizeof(float);
    if (partials) {
        if (parts != gpi) {
            vip_set_error("vip_dwconv2d_pool_nhwc_f16: partials sized for %d rows per image, the kernel writes %d", parts, gpi);
            return VIP_ERR_BAD_ARG;
        }
        hipLaunchKernelGGL((dwconv_tile_kernel<K, T, TW, WHOLE, true>), dim3((unsigned)gx, (unsigned)gyc), dim3(256),
                           smem + (size_t)tiles_per_block * cb * 8 * sizeof(float), s, x, w, bias, y, B, H, W, C, pt, pl, Ho, Wo, act,
                           cb, tiles_x, tiles_y, n_tiles, 2L * B * H * W * C, partials, gpi);
        return vip_launch_status("vip_dwconv2d_pool_nhwc_f16(tile)");
    }
    hipLaunchKernelGGL((dwconv_tile_kernel<K, T, TW, WHOLE, false>), dim3((unsigned)gx, (unsigned)gyc), dim3(256), smem, s, x, w, bias,
                       y, B, H, W, C, pt, pl, Ho, Wo, act, cb, tiles_x, tiles_y, n_tiles, 2L * B * H * W * C, (float*)nullptr, 0);
    return vip_launch_status("vip_dwconv2d_nhwc_f16(tile)");
}

// rows of partial sums per image the pooling form writes for this shape (the launcher's own tiling)
template <int T, int TW>
int tile_parts(int Ho, int Wo, int C) {
    const int C8 = C / 8;
    int cb = C8 < 16 ? C8 : 16;
    if (C8 % 12 == 0 && C8 % 16 != 0) cb = 12;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + T - 1) / T;
    const int tiles_per_block = 256 / cb;
    const int gpi = (tiles_x * tiles_y + tiles_per_block - 1) / tiles_per_block;
    // image-aligned groups idle the tail of each image's last group: below ~85 % lane use the depthwise kernel loses more than the
    // gate kernel saves (7x7 maps: 8 tiles in 16 slots; measured on EfficientNetV2-T, +0.2 ms per 256 images) - plain calls there
    if (tiles_x * tiles_y * 100 < gpi * tiles_per_block * 85) return 0;
    return gpi;
}

}  // namespace

// stride-1 fast path; returns 1 if the shape is not handled here
int vip_dwconv_tiled(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int k,
                     int pt, int pl, int Ho, int Wo, int act, hipStream_t s, float* partials, int parts) {
    const f16* xi = (const f16*)x;
    const float* wi = w;
    f16* yo = (f16*)y;
    const long gx = ((long)B * ((Wo + 1) / 2) * ((Ho + 1) / 2) + 15) / 16;
    if (gx >= (1L << 31) || 2L * B * H * W * C >= 0xFFFFFFF0L) return 1;
    // tile widths chosen so that accumulators + one fp32 patch row stay well under 256 VGPRs (no scratch)
    // register tiles (rows x cols per thread) picked by measurement on the ensemble's layer shapes (tools/bench_dw.py):
    // smaller tiles -> fewer VGPRs -> more resident waves, which beats the extra halo loads for k = 3 / 5
#define VIP_GO(KK, TT, WW, WH) return launch_tile<KK, TT, WW, WH>(xi, wi, bias, yo, B, H, W, C, pt, pl, Ho, Wo, act, s, partials, parts)
    // measured on the ensemble's layer shapes (tools/bench_dw.py): 3x3 wants the whole 4x6 patch in flight (+15-25 %
    // over row-at-a-time); 5x5 / 7x7 whole-patch variants spill, and their rows carry enough FMAs to cover a load
    if (k == 3) VIP_GO(3, 2, 4, true);
    if (k == 5) VIP_GO(5, 2, 2, false);
    if (k == 7) VIP_GO(7, 2, 4, false);
#undef VIP_GO
    return 1;
}

// stride-1 fast path of vip_dwconv2d_nhwc_h2 (strict_ops.hip); returns 1 if the shape is not handled here
int vip_dwconv_tiled_h2(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int k, int pt, int pl, int Ho,
                        int Wo, int act, int* status, hipStream_t s) {
    if (4L * B * H * W * C >= 0xFFFFFFE0L || C % 8 != 0) return 1;
#define VIP_GO(KK, TT, WW, WH) return launch_tile_h2<KK, TT, WW, WH>(x, w, bias, y, B, H, W, C, pt, pl, Ho, Wo, act, status, s)
    if (k == 3) VIP_GO(3, 2, 4, true);
    if (k == 5) VIP_GO(5, 2, 2, false);
    if (k == 7) VIP_GO(7, 2, 4, false);
#undef VIP_GO
    return 1;
}

// partial-sum rows per image of the pooling form (same register tiles as above), 0 if the shape is not handled by the tile kernel
int vip_dwconv_tiled_parts(int B, int H, int W, int C, int k, int Ho, int Wo) {
    const long gx = ((long)B * ((Wo + 1) / 2) * ((Ho + 1) / 2) + 15) / 16;
    if (gx >= (1L << 31) || 2L * B * H * W * C >= 0xFFFFFFF0L || C % 8 != 0) return 0;
    if (k == 3) return tile_parts<2, 4>(Ho, Wo, C);
    if (k == 5) return tile_parts<2, 2>(Ho, Wo, C);
    if (k == 7) return tile_parts<2, 4>(Ho, Wo, C);
    return 0;
}
