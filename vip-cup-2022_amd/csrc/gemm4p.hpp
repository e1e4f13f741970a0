// Mid-size pointwise GEMM for gfx950: 128 pixels x 128 channels x 64 k tiles, 4 waves, TWO workgroups per CU, LDS-DMA staging.
// Included by conv_igemm.hip after gemm8p.hpp (shares its helpers).  For 1x1 stride-1 ungrouped convolutions / Dense layers with
// K % 64 == 0, N % 128 == 0 that gemm8p does not take (N % 256 != 0, or too few 256 x 256 tiles to fill the chip).
//
// Why: on these layers (K = 128 ... 1536, a handful of k-tiles per output tile) pwk_direct_kernel is latency-bound - it requests the
// next 64-k chunk only one chunk ahead (0.4 us of MFMA work against ~1.5 us of memory latency), with weights staged through registers and
// activations fetched as fragments (16 rows x 64 B per instruction: two half-used lines per row).  Here both operands arrive by
// `buffer_load_dwordx4 ... lds` in full 128-byte lines, 1.5 k-tiles ahead, with no registers in flight; and because a workgroup needs
// only 64 KiB of LDS and ~140 VGPRs, two of them share a CU, so the fill and the store tail of one run under the MFMAs of the other
// (what the persistent 128 KiB gemm8p workgroup has to arrange by hand).
//
// Geometry: waves 2 (pixels) x 2 (channels); wave (wm, wn): pixels 64 wm .. +63 (4 tiles), channels 64 wn .. +63 (4 tiles, interleaved
// rows as everywhere: a lane owns 8 consecutive channels)  ->  acc[4][4].
// LDS: 2 buffers x 4 half-tile slots (A0, B0, B1, A1) of 64 rows x 128 B = 64 KiB.  A_g row j: wave-column wn = j >> 5, channel
// 64 wn + perm(32 g + (j & 31));  B_h row j: wave-row wm = j >> 5, pixel 64 wm + 32 h + (j & 31).  Swizzle and DMA mapping as in gemm8p
// (key = (row >> 1) & 7 on the source side; a wave instruction = 8 rows).
// Schedule, k-tile t in buffer t & 1, two phases of 16 MFMAs:
//   P0: read B0, B1, A0(t) | DMA A1(t+1)                       | MFMA (A0 x B)         | barrier
//   P1: read A1(t)         | DMA A0, B0, B1(t+2) | vmcnt(6)    | MFMA (A1 x B)         | barrier
//   A1(t+1) goes into the other buffer (last read in P1(t-1)); A0 / B0 / B1(t+2) into this buffer (last read in P0(t)).  The wait in P1
//   leaves the three half-tiles just issued in flight, so all of tile t+1 has landed; it is read one barrier later.
#pragma once

constexpr int G4_SLOT = 64 * 128;                  // bytes per half-tile slot

__global__ __launch_bounds__(256, 2) void gemm4p_kernel(ConvArgs a, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    int bid = blockIdx.x;
    {   // XCD-aware remap (bijective for any grid size): consecutive tiles of one XCD share an L2
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mb = bid / a.n_blocks, nb = bid - mb * a.n_blocks;
    const int mblk = mb * 128, nblk = nb * 128;

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)(2L * a.Cout_g * a.ldw), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + a.cin_off), 0,
                                                                        (unsigned)(a.x_span_bytes - 2L * a.cin_off), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_res =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (unsigned)a.res_span_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_span_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_bias =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)(a.bias_elems * 4) : 0u, 0x00020000);

    // ---- DMA plan: this thread moves rows j = 16 wave + 8 i + (lane >> 3), i = 0, 1, of every 64-row half-tile -----------------
    const int pc = lane & 7;
    unsigned vA[2][2], vB[2][2];                   // [half][i]
    int dma_lds[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rg = wave * 2 + i;               // 8-row group 0..7
        const int j = rg * 8 + (lane >> 3);
        const int c = pc ^ ((j >> 1) & 7);
        dma_lds[i] = rg * 1024;
        const int jj = j & 31;                     // row inside the wave-column's / wave-row's 32 rows
#pragma unroll
        for (int g = 0; g < 2; ++g) {              // A_g: channel tiles 2g, 2g+1 of wave-column j >> 5
            const int rr = 32 * g + jj, t = (rr >> 4) & 3, r = rr & 15;
            const int ch = nblk + (j >> 5) * 64 + (t >> 1) * 32 + (r >> 2) * 8 + (t & 1) * 4 + (r & 3);
            vA[g][i] = ch < a.Cout_g ? (unsigned)((ch * a.ldw + c * 8) * 2) : G8_OOB;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int px = mblk + (j >> 5) * 64 + 32 * h + jj;
            vB[h][i] = px < a.M ? (unsigned)((px * a.ldx + c * 8) * 2) : G8_OOB;
        }
    }
    const int nk = a.K >> 6;
    auto stageA = [&](int g, char* slot, int kt) {
        const unsigned ko = kt < nk ? (unsigned)kt * 128u : G8_OOB;
        g8_dma(rw, slot + dma_lds[0], vA[g][0] + ko);
        g8_dma(rw, slot + dma_lds[1], vA[g][1] + ko);
    };
    auto stageB = [&](int h, char* slot, int kt) {
        const unsigned ko = kt < nk ? (unsigned)kt * 128u : G8_OOB;
        g8_dma(rx, slot + dma_lds[0], vB[h][0] + ko);
        g8_dma(rx, slot + dma_lds[1], vB[h][1] + ko);
    };

    // ---- fragment read offsets: this wave's 32 rows of a slot start at (wn or wm) * 4096 --------------------------------------
    const int key = (l15 >> 1) & 7;
    const int fo0 = l15 * 128 + ((lq ^ key) << 4), fo1 = fo0 ^ 64;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    const unsigned a_base0 = lds0 + (unsigned)(wn * 4096 + fo0), a_base1 = lds0 + (unsigned)(wn * 4096 + fo1);
    const unsigned b_base0 = lds0 + (unsigned)(wm * 4096 + fo0), b_base1 = lds0 + (unsigned)(wm * 4096 + fo1);

    f32x4 acc[4][4];
    {   // bias is the C operand of the first MFMA of every accumulator
        const int n0 = nblk + wn * 64;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + (nt >> 1) * 32 + lq * 8 + (nt & 1) * 4;
            const f32x4 bv = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_bias, n < a.Cout_g ? (unsigned)(n * 4) : OOB, 0, 0));
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[p][nt] = bv;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the bias loads are the only ordinary loads: retire them before the DMA ring starts

    // ---- prologue: half-tiles A0, B0, B1, A1 of k-tile 0 and A0, B0, B1 of k-tile 1 (slot order inside a buffer: A0 B0 B1 A1) --------
    stageA(0, smem + 0 * G4_SLOT, 0);
    stageB(0, smem + 1 * G4_SLOT, 0);
    stageB(1, smem + 2 * G4_SLOT, 0);
    stageA(1, smem + 3 * G4_SLOT, 0);
    stageA(0, smem + 4 * G4_SLOT, 1);
    stageB(0, smem + 5 * G4_SLOT, 1);
    stageB(1, smem + 6 * G4_SLOT, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // all of k-tile 0
    __builtin_amdgcn_s_barrier();

    U4H8 af[2][2], bf[4][2];                                // [tile][k-step]
#define G4_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define G4_WAIT_A() \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0].h), "+v"(af[0][1].h), "+v"(af[1][0].h), "+v"(af[1][1].h)::"memory")
#define G4_WAIT_AB()                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                     \
                 : "+v"(af[0][0].h), "+v"(af[0][1].h), "+v"(af[1][0].h), "+v"(af[1][1].h), "+v"(bf[0][0].h), "+v"(bf[0][1].h), \
                   "+v"(bf[1][0].h), "+v"(bf[1][1].h), "+v"(bf[2][0].h), "+v"(bf[2][1].h), "+v"(bf[3][0].h), "+v"(bf[3][1].h)::"memory")
#define G4_MFMA(NT0)                                                                                                    \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                            \
                _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                           \
                    acc[p][NT0 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt][ks].h, bf[p][ks].h,                \
                                                                              acc[p][NT0 + nt], 0, 0, 0);              \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
    } while (0)

    for (int t = 0; t < nk; ++t) {
        const int b = t & 1;
        char* cur = smem + b * 4 * G4_SLOT;
        char* nxt = smem + (b ^ 1) * 4 * G4_SLOT;
        const unsigned bo = (unsigned)(b * 4 * G4_SLOT);
        const unsigned a0 = a_base0 + bo, a1 = a_base1 + bo, b0 = b_base0 + bo, b1 = b_base1 + bo;
        // ---- P0: (A0 x B0, B1) ----
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            G4_DSR(bf[pp][0].h, b0, 1 * G4_SLOT + pp * 2048);
            G4_DSR(bf[pp][1].h, b1, 1 * G4_SLOT + pp * 2048);
            G4_DSR(bf[2 + pp][0].h, b0, 2 * G4_SLOT + pp * 2048);
            G4_DSR(bf[2 + pp][1].h, b1, 2 * G4_SLOT + pp * 2048);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            G4_DSR(af[nt][0].h, a0, 0 * G4_SLOT + nt * 2048);
            G4_DSR(af[nt][1].h, a1, 0 * G4_SLOT + nt * 2048);
        }
        stageA(1, nxt + 3 * G4_SLOT, t + 1);
        G4_WAIT_AB();
        G4_MFMA(0);
        __builtin_amdgcn_s_barrier();
        // ---- P1: (A1 x B0, B1) ----
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            G4_DSR(af[nt][0].h, a0, 3 * G4_SLOT + nt * 2048);
            G4_DSR(af[nt][1].h, a1, 3 * G4_SLOT + nt * 2048);
        }
        stageA(0, cur + 0 * G4_SLOT, t + 2);
        stageB(0, cur + 1 * G4_SLOT, t + 2);
        stageB(1, cur + 2 * G4_SLOT, t + 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        G4_WAIT_A();
        G4_MFMA(2);
        __builtin_amdgcn_s_barrier();
    }
#undef G4_DSR
#undef G4_WAIT_A
#undef G4_WAIT_AB
#undef G4_MFMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the ring's last (out-of-range) half-tiles

    int m_base = mblk + wm * 64 + l15, n_first = nblk + wn * 64 + lq * 8;
    asm volatile("" : "+v"(m_base), "+v"(n_first));
    switch (mode) {
        case 1: pw_epilogue<4, VIP_ACT_RELU, false, false>(a, acc, m_base, n_first, rb_res, rb_y); break;
        case 2: pw_epilogue<4, VIP_ACT_SILU, false, false>(a, acc, m_base, n_first, rb_res, rb_y); break;
        case 3: pw_epilogue<4, VIP_ACT_GELU, false, false>(a, acc, m_base, n_first, rb_res, rb_y); break;
        case 4: pw_epilogue<4, VIP_ACT_SIGMOID, false, false>(a, acc, m_base, n_first, rb_res, rb_y); break;
        case 5: pw_epilogue<4, VIP_ACT_NONE, true, false>(a, acc, m_base, n_first, rb_res, rb_y); break;
        case 6: pw_epilogue<4, VIP_ACT_NONE, true, true>(a, acc, m_base, n_first, rb_res, rb_y); break;
        default: pw_epilogue<4, VIP_ACT_NONE, false, false>(a, acc, m_base, n_first, rb_res, rb_y); break;
    }
}

inline int launch_gemm4p(const ConvArgs& a0, int mode, hipStream_t s) {
    ConvArgs a = a0;
    a.m_blocks = (a.M + 127) / 128;
    a.n_blocks = (a.Cout_g + 127) / 128;
    constexpr size_t smem = 8 * G4_SLOT;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm4p_kernel, dim3((unsigned)(a.m_blocks * a.n_blocks)), dim3(256), smem, s, a, mode);
    return vip_launch_status("vip_conv2d_nhwc_f16(gemm4p)");
}
