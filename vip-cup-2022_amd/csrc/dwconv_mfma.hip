// Depthwise 7x7 / 5x5 convolution (stride 1) on the matrix cores.
//
// A per-channel filter has no GEMM shape across channels, but along one image axis it has: for one channel and one
// filter row dy,   out[y, xo] += sum_xi in[y + dy, xi] * T_dy[xi, xo],   T_dy[xi, xo] = w[dy][xi - xo]   (a banded Toeplitz
// matrix).  A 16 x 16 output tile of one channel is therefore   D[16 y][16 xo] = A[16 y][K] . B[K][16 xo]   with
// k = (dy, xi): A is the channel's input patch - row y + dy, 24 consecutive columns, i.e. plain 16-byte runs of a
// channel-major LDS image - and B is built ONCE per wave from the channel's k*k weights and stays in registers while the
// wave walks over tiles.  K = 7 * 24 = 168 -> 6 steps of v_mfma_f32_16x16x32_f16 (5x5: 4 steps), 3.9x the useful MACs, and
// twice that because B is carried as hi + lo fp16 (the filters are fp32 at the boundary: a rounded depthwise filter is a
// per-channel gain error, DESIGN.md section 4 ii).  12 MFMAs = 192 SIMD cycles per 256 outputs against ~400 cycles of
// packed-fp32 FMAs plus 160 conversions on the VALU kernel (dwconv.hip): the op becomes HBM-bound.
//
// The price is two transpositions through LDS: NHWC global (channels contiguous) -> channel-major planes on the way in
// (four pixels x eight channels per lane: 16-byte loads, 16-bit interleaves in registers, 8-byte LDS writes), and the
// accumulators (one channel per MFMA) -> NHWC pixels on the way out (four channels per wave packed to 8 bytes, 16-byte
// stores).  Workgroup = 4 waves = one slab of 16 channels (4 per wave: 192 VGPRs of B); workgroups are persistent on a
// (slab, tile band) and the slabs of one tile sit next to each other on one XCD, so the 32-byte slices they read of every
// 128-byte line meet in that XCD's L2.
// Replaces tf.keras.layers.DepthwiseConv2D of tfimm convnext.py:192-198 (7x7) and kecam efficientnet_v2.py:85 (5x5).
//
// MEASURED (B = 256, profiles/r02_dwconv_mfma_*.log): correct (tests/test_gpu_ops.py::test_dwconv_matrix_cores) but NOT faster than the
// VALU kernel - 603 vs 644 us on ConvNeXt stage 0 (99 x 99 x 96), 336 vs 341 (stage 1), 200 vs 139 (stage 2), 108 vs 80 (stage 3) - so it
// is opt-in (VIP_DW_MFMA=1).  Switching phases off one at a time shows why: loop control + output staging 173 us, transposition 67,
// A reads + MFMAs 107, DMA 200, stores ~100 - the phases ADD UP instead of overlapping.  The 192 VGPRs of B fragments leave room for 4
// channels per wave and 2 waves per SIMD, so a workgroup covers only 16 channels = 32-byte slices of each pixel (PMC: 3.1x the
// algorithmic read bytes out of L2), and per tile a wave issues ~900 instructions around its 48 MFMAs.  What it would take: B operands
// that are not per-lane register images (none exists for a per-channel Toeplitz matrix), or a channel-planar activation layout upstream.
#include "common.hpp"
#include <stdlib.h>

namespace {

#ifndef VIP_DW_SKIP
#define VIP_DW_SKIP 0      // timing experiments only: 1 no DMA, 2 no transposition, 4 no MFMA, 8 no output stores
#endif
#ifndef VIP_DW_WAIT
#define VIP_DW_WAIT (Cf::DPW + 4)
#endif

template <int KS>
struct DwM {
    static constexpr int TY = 16, TX = 16, CS = 16;
    static constexpr int PR = TY + KS - 1;            // patch rows
    static constexpr int ROWP = 24;                   // halfs per patch row (TX + KS - 1 <= 22 used)
    static constexpr int NG = KS * 3;                 // 8-column groups of K: (dy, g)
    static constexpr int KSTEPS = (NG + 3) / 4;       // MFMA k-steps of 32
    static constexpr int PLANE = PR * ROWP;           // halfs per channel plane
    static constexpr int OUTP = 20;                   // halfs per staged output pixel: 16 channels + 8 bytes (bank spread)
    static constexpr int XQ = ROWP / 4;               // 4-pixel groups per patch row
    static constexpr int ITEMS = PR * XQ * (CS / 8);  // (row, 4-pixel group, channel octet) transposition items per tile
    static constexpr int DPW = (4 * ITEMS + 255) / 256;   // LDS-DMA instructions per wave and tile (64 x 16 bytes each)
    static constexpr int STAGE_BYTES = DPW * 4 * 64 * 16; // one staging buffer: slot (t, item) = pixel t of the item, 16 bytes
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES + CS * PLANE * 2 + TY * TX * OUTP * 2 + KS * KS * CS * 4;
};

struct DwMArgs {
    const f16* x;
    const float* w;
    const float* bias;
    f16* y;
    int B, H, W, C, pt, pl, Ho, Wo, act;
    int tiles_x, tiles_y, nslab, ntw;     // ntw: workgroups per (XCD, slab)
    int n_tiles;
    long x_bytes, y_bytes;
};

typedef int dw_i32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA from inline asm (M0 = this wave's 1 KiB destination, lane L lands at +16 L): hipcc drains every outstanding
// LDS-DMA in front of each LDS read it can see, which would serialise the prefetch; waits are placed by hand.
__device__ __forceinline__ void dw_dma16(unsigned lds_wave_base, unsigned voff, dw_i32x4 rsrc) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_wave_base), "v"(voff), "s"(rsrc)
                 : "memory");
}

template <int KS>
__global__ __launch_bounds__(256, 2) void dwconv_mfma_kernel(DwMArgs a) {
    using Cf = DwM<KS>;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* stage = smem;                                                 // [2][STAGE_BYTES]
    f16* planes = reinterpret_cast<f16*>(smem + 2 * Cf::STAGE_BYTES);            // [CS][PR][ROWP]
    f16* outst = planes + Cf::CS * Cf::PLANE;                                    // [TY*TX][OUTP]
    float* wst = reinterpret_cast<float*>(outst + Cf::TY * Cf::TX * Cf::OUTP);   // [KS*KS][CS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, Q = lane >> 4;
    // block -> (XCD, slab, band walker): blocks b, b+8, ... share an XCD; inside it the slabs of one walker are adjacent
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int slab = j % a.nslab, tw = j / a.nslab;
    const int c0 = slab * Cf::CS;

    for (int i = tid; i < KS * KS * Cf::CS; i += 256) wst[i] = a.w[(long)(i / Cf::CS) * a.C + c0 + (i % Cf::CS)];
    __syncthreads();

    // B fragments: lane (xo = l15, Q), step s, element e: k-group gi = 4 s + Q = (dy, g), xi = 8 g + e, tap dx = xi - xo
    f16x8 bhi[4][Cf::KSTEPS], blo[4][Cf::KSTEPS];
    // A fragment of step s: row l15 + dy, columns 8 g ... = byte l15 * 48 + 16 (4 s + Q) of the plane, because a patch row is
    // exactly three groups long: one base register and immediates.  The groups past NG (last step only) meet zero B columns but
    // must still read finite data: they are pointed at the plane's start.
    const unsigned abase = (unsigned)(l15 * Cf::ROWP * 2 + 16 * Q);
    const unsigned alast = (4 * (Cf::KSTEPS - 1) + Q < Cf::NG) ? abase + 64u * (Cf::KSTEPS - 1) : 0u;
    static_assert(Cf::ROWP == 24, "three 8-column groups per patch row");
#pragma unroll
    for (int s = 0; s < Cf::KSTEPS; ++s) {
        const int gi = 4 * s + Q;
        const int dy = gi / 3, g = gi - dy * 3;
        const bool live = gi < Cf::NG;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int dx = g * 8 + e - l15;
                const bool ok = live && dx >= 0 && dx < KS;
                const float v = ok ? wst[(dy * KS + dx) * Cf::CS + wave * 4 + ci] : 0.f;
                const _Float16 hi = (_Float16)v;
                bhi[ci][s][e] = hi;
                blo[ci][s][e] = (_Float16)(v - (float)hi);
            }
        }
    }
    float bv[4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) bv[ci] = a.bias ? a.bias[c0 + wave * 4 + ci] : 0.f;

    // this lane's DMA slots: instruction i of wave w fills slots (4 i + w) * 64 + lane; slot S = pixel t = S / ITEMS of item S % ITEMS.
    // Decoded again for every tile (a dozen integer instructions per DMA): kept across the tile loop, the five descriptors were
    // the registers that spilled - and a scratch reload waits with vmcnt(0), which drains the look-ahead DMA.
    const unsigned long xp = (unsigned long)a.x, yp = (unsigned long)a.y;
    const dw_i32x4 rx = {(int)(unsigned)xp, (int)((xp >> 32) & 0xffffu), (int)(unsigned)a.x_bytes, 0x00020000};
    const dw_i32x4 ry4 = {(int)(unsigned)yp, (int)((yp >> 32) & 0xffffu), (int)(unsigned)a.y_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;

    const int band = (a.n_tiles + 7) / 8;
    const int t_hi = min(a.n_tiles, (xcd + 1) * band);
    const int tiles_img = a.tiles_x * a.tiles_y;
    const unsigned char* my_planes = reinterpret_cast<const unsigned char*>(planes) + (wave * 4) * Cf::PLANE * 2;

    struct Pos {
        int b, oy0, ox0;
        bool live;
    };
    auto decode = [&](int tile) {                     // wave-uniform 32-bit arithmetic (64-bit divisions here spilled and drained vmcnt)
        Pos p;
        p.live = tile < t_hi;
        const unsigned t = p.live ? (unsigned)tile : 0u;
        const unsigned b = t / (unsigned)tiles_img, rem = t - b * (unsigned)tiles_img;
        const unsigned ty = rem / (unsigned)a.tiles_x;
        p.b = (int)b;
        p.oy0 = (int)ty * Cf::TY;
        p.ox0 = (int)(rem - ty * (unsigned)a.tiles_x) * Cf::TX;
        return p;
    };
    auto issue_dma = [&](const Pos& p, int buf) {     // all-OOB (zeros) past the band: the instruction count per tile never changes
        const bool live = p.live;
        const int gy0 = p.oy0 - a.pt, gx0 = p.ox0 - a.pl;
        const unsigned img_off = (unsigned)((((long)p.b * a.H * a.W) * a.C + c0) * 2);
        int ln = lane;
        asm volatile("" : "+v"(ln));                  // opaque: keeps the decode below inside the loop
#pragma unroll
        for (int i = 0; i < Cf::DPW; ++i) {
            const int S = (4 * i + wave) * 64 + ln;
            const int t = S / Cf::ITEMS, it = S - t * Cf::ITEMS;
            const int oct = it & 1, rq = it >> 1;
            const int row = rq / Cf::XQ, xq = rq - row * Cf::XQ;
            const int gy = gy0 + row, gx = gx0 + xq * 4 + t;
            const bool ok = live & (t < 4) & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W);
            const unsigned off = ok ? img_off + (unsigned)(((gy * a.W + gx) * a.C + oct * 8) * 2) : OOB;
            if (!(VIP_DW_SKIP & 1)) dw_dma16(lds0 + (unsigned)(buf * Cf::STAGE_BYTES + (4 * i + wave) * 1024), off, rx);
        }
    };
    // Output stores from inline asm as well: the hand-counted vmcnt waits need EXACTLY two per tile and wave (hipcc merged the
    // two identical placeholder stores of the prologue into one, and the first tile was read before its DMA had landed).
    auto store16 = [&](const uint4& v, unsigned off) {
        const dw_i32x4 d = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" ::"v"(d), "v"(off), "s"(ry4) : "memory");
    };
    auto dummy_stores = [&]() {                       // two dropped stores: the vmcnt bookkeeping of the first tiles = steady state
        const uint4 z = {0u, 0u, 0u, 0u};
        store16(z, OOB);
        store16(z, OOB);
    };

    int tile = xcd * band + tw;
    Pos cur = decode(tile), nx1 = decode(tile + a.ntw);
    issue_dma(cur, 0);
    dummy_stores();
    issue_dma(nx1, 1);
    dummy_stores();

    int buf = 0;
    for (; tile < t_hi; tile += a.ntw, buf ^= 1) {
        const int b = cur.b, oy0 = cur.oy0, ox0 = cur.ox0;
        const Pos nx2 = decode(tile + 2 * a.ntw);

        // in order behind this tile's DMA: [2 stores][next tile's DMA][2 stores] - everything older has landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VIP_DW_WAIT) : "memory");
        __syncthreads();

        // ---- 1. staging (pixels) -> channel-major planes
        const unsigned char* sb = stage + buf * Cf::STAGE_BYTES;
#pragma unroll
        for (int h = 0; h < (Cf::ITEMS + 255) / 256; ++h) {
            const int it = tid + 256 * h;
            if (it < Cf::ITEMS && !(VIP_DW_SKIP & 2)) {
                const int oct = it & 1, rq = it >> 1;
                const int row = rq / Cf::XQ, xq = rq - row * Cf::XQ;
                uint4 r[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) r[t] = *reinterpret_cast<const uint4*>(sb + (t * Cf::ITEMS + it) * 16);
                const unsigned* rr = reinterpret_cast<const unsigned*>(r);      // rr[4 t + d]: channels 2d, 2d+1 of pixel t
                f16* dst = planes + (oct * 8) * Cf::PLANE + row * Cf::ROWP + xq * 4;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint2 lo, hi;
                    lo.x = (rr[d] & 0xFFFFu) | (rr[4 + d] << 16);
                    lo.y = (rr[8 + d] & 0xFFFFu) | (rr[12 + d] << 16);
                    hi.x = (rr[d] >> 16) | (rr[4 + d] & 0xFFFF0000u);
                    hi.y = (rr[8 + d] >> 16) | (rr[12 + d] & 0xFFFF0000u);
                    *reinterpret_cast<uint2*>(dst + (2 * d) * Cf::PLANE) = lo;
                    *reinterpret_cast<uint2*>(dst + (2 * d + 1) * Cf::PLANE) = hi;
                }
            }
        }
        __syncthreads();
        issue_dma(nx2, buf);                          // the staging buffer just emptied: two tiles of lead

        // ---- 2. D[y][xo] = A . (Bhi + Blo) for this wave's four channels
        f32x4 acc[4];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) acc[ci] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < ((VIP_DW_SKIP & 4) ? 0 : Cf::KSTEPS); ++s) {
            U4H8 af[4];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
                af[ci].u = *reinterpret_cast<const uint4*>(my_planes + ci * Cf::PLANE * 2 + (s + 1 < Cf::KSTEPS ? abase + 64u * s : alast));
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) acc[ci] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ci].h, bhi[ci][s], acc[ci], 0, 0, 0);
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) acc[ci] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ci].h, blo[ci][s], acc[ci], 0, 0, 0);
        }

        // ---- 3. accumulators -> staged NHWC pixels: lane holds (y = 4 Q + r, xo = l15) of its wave's 4 channels
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f16x4 o;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) o[ci] = (f16)vip_act(acc[ci][r] + bv[ci], a.act);
            *reinterpret_cast<f16x4*>(outst + ((4 * Q + r) * Cf::TX + l15) * Cf::OUTP + wave * 4) = o;
        }
        __syncthreads();

        // ---- 4. staged pixels -> global: exactly two 16-byte stores per lane, out-of-image pixels dropped by the buffer unit
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int it = tid + 256 * h;
            const int oct = it & 1, p = it >> 1;
            const int gy = oy0 + (p >> 4), gx = ox0 + (p & 15);
            const uint2* src = reinterpret_cast<const uint2*>(outst + p * Cf::OUTP + oct * 8);
            const uint2 v0 = src[0], v1 = src[1];
            const uint4 v = {v0.x, v0.y, v1.x, v1.y};
            const unsigned off = (gy < a.Ho && gx < a.Wo) ? (unsigned)(((((long)b * a.Ho + gy) * a.Wo + gx) * a.C + c0 + oct * 8) * 2) : OOB;
            if (!(VIP_DW_SKIP & 8)) store16(v, off);
        }
        cur = nx1;
        nx1 = nx2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead DMA of tiles past the band must not outlive the LDS allocation
}

int dw_cu_count() {
    static const int n = [] {
        int dev = 0;
        hipDeviceProp_t pr;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
                   ? pr.multiProcessorCount : 256;
    }();
    return n;
}

template <int KS>
int launch_dw_mfma(const f16* x, const float* w, const float* bias, f16* y, int B, int H, int W, int C, int pt, int pl, int Ho,
                   int Wo, int act, hipStream_t s) {
    using Cf = DwM<KS>;
    DwMArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.B = B; a.H = H; a.W = W; a.C = C; a.pt = pt; a.pl = pl; a.Ho = Ho; a.Wo = Wo; a.act = act;
    a.tiles_x = (Wo + Cf::TX - 1) / Cf::TX;
    a.tiles_y = (Ho + Cf::TY - 1) / Cf::TY;
    a.n_tiles = B * a.tiles_x * a.tiles_y;
    a.nslab = C / Cf::CS;
    a.x_bytes = 2L * B * H * W * C;
    a.y_bytes = 2L * B * Ho * Wo * C;
    const int per_xcd = 2 * dw_cu_count() / 8;                // resident workgroups of one XCD (2 per CU)
    const int band = (a.n_tiles + 7) / 8;
    int ntw = per_xcd / a.nslab;
    if (ntw < 1) ntw = 1;
    if (ntw > band) ntw = band;
    a.ntw = ntw;
    static bool attr_done[8] = {};
    if (!attr_done[KS]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_mfma_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  Cf::LDS_BYTES);
        attr_done[KS] = true;
    }
    hipLaunchKernelGGL(dwconv_mfma_kernel<KS>, dim3((unsigned)(8 * a.nslab * ntw)), dim3(256), Cf::LDS_BYTES, s, a);
    return vip_launch_status("vip_dwconv2d_nhwc_f16(mfma)");
}

}  // namespace

// stride-1 7x7 / 5x5 with C % 16 == 0 on the matrix cores; returns 1 if the shape is not handled here.  OPT-IN (VIP_DW_MFMA=1, read
// per call): measured at 0.94x (ConvNeXt stage 0) ... 1.4x (stage 2) the time of the VALU kernel - see the header and DESIGN.md.
int vip_dwconv_mfma(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int k, int pt, int pl,
                    int Ho, int Wo, int act, hipStream_t s) {
    const char* e = getenv("VIP_DW_MFMA");
    const int on = e ? atoi(e) : 0;
    if (!on || C % 16 != 0 || 2L * B * H * W * C >= 0x80000000L || 2L * B * Ho * Wo * C >= 0x80000000L) return 1;
    if (k != 7 && k != 5) return 1;
    // the 24-column patch rows must cover the taps of 16 outputs: pl + (k - 1 - pl) = k - 1 <= 8 always; rows likewise
    if (k == 7) return launch_dw_mfma<7>((const f16*)x, w, bias, (f16*)y, B, H, W, C, pt, pl, Ho, Wo, act, s);
    return launch_dw_mfma<5>((const f16*)x, w, bias, (f16*)y, B, H, W, C, pt, pl, Ho, Wo, act, s);
}
