// Deep-K pointwise GEMM for gfx950: 256 pixels x 256 channels x 64 k block tiles, 8 waves, LDS-DMA staging with counted
// vmcnt across raw barriers, four MFMA phases per k-tile.  Included by conv_igemm.hip (inside its anonymous namespace, after
// ConvArgs / pw_epilogue); used for 1x1 stride-1 ungrouped convolutions / Dense layers with K % 64 == 0, K >= 384.
//
// Why a separate kernel: in pwk_gemm_kernel a wave visits the MFMA pipe, the LDS pipe (fragment reads + staging writes) and
// the vector-memory pipe (register staging) one after the other per 64-k chunk - three pipes of about equal load, ~30 % of
// the MFMA peak whatever the variant.  Here nothing is staged through registers: `buffer_load_dwordx4 ... lds` writes the
// next tiles straight into LDS (no VGPRs, no ds_write pass), three half-tiles stay in flight across the barriers
// (s_waitcnt vmcnt(6) once per k-tile, never 0 in the loop), and each phase issues the NEXT fragments' ds_reads and one
// half-tile of DMA before its 16 MFMAs, so the three pipes run side by side.
//
// Tile geometry (per workgroup, 512 threads = 8 waves as 4 (pixels) x 2 (channels)):
//   wave (wm, wn): pixels 64 wm .. +63, channels 128 wn .. +127  ->  acc[2 channel groups][4 pixel tiles][4 channel tiles]
//   weights are the MFMA A operand (rows = channels, interleaved as in the other pointwise kernels so that a lane owns 8
//   consecutive channels), activations the B operand (lane = pixel).
// LDS: 2 buffers x 4 half-tile slots (A0, B0, B1, A1) of 128 rows x 128 B = 128 KiB.  Slot A_g row j: wave-column wn = j >> 6,
//   channel 128 wn + 64 g + perm(j & 63);  slot B_h row j: wave-row wm = j >> 5, pixel 64 wm + 32 h + (j & 31).
//   16-byte chunk c of row j sits at physical chunk c ^ key(j), key(j) = (j >> 1) & 7.  ds_read_b128 is served in four groups of 16
//   lanes that are NOT consecutive (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...): a group reads rows
//   {0-3, 12-15} at chunk c and rows {4-11} at chunk c ^ 1; with this key those 16 accesses cover all 64 banks exactly once.  The DMA writes LDS lane-linearly (base + 16 lane), so the
//   swizzle is applied to the SOURCE address: lane (row j, physical chunk p) fetches logical chunk p ^ key(j).
// Schedule (tile t lives in buffer t & 1; phase p of tile t):
//   p0: read B0,A0 frags | DMA A1(t+1) | MFMA (A0,B0)        p1: read B1 | DMA A0(t+2) | MFMA (A0,B1)
//   p2: read A1          | DMA B0(t+2) | MFMA (A1,B1)        p3: -       | DMA B1(t+2) | vmcnt(6) | MFMA (A1,B0)
//   A slot is rewritten one phase after its last read (the phase-end barrier orders it); the wait of p3 leaves the three newest
//   half-tiles in flight, so everything tile t+1 needs has landed, and it is read one barrier later.
#pragma once

constexpr int G8_SLOT = 128 * 128;                 // bytes per half-tile slot
constexpr unsigned G8_OOB = 0x80000000u;           // out-of-range voffset that survives "+ k offset" without wrapping

template <int KIND>   // 0 = A0, 1 = B0, 2 = B1, 3 = A1 : slot order inside a buffer
__device__ __forceinline__ char* g8_slot(char* smem, int buf) { return smem + (buf * 4 + KIND) * G8_SLOT; }

__device__ __forceinline__ void g8_dma(const __amdgpu_buffer_rsrc_t& r, char* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}

template <bool TWO_BARRIERS, bool PIPE>
__global__ __launch_bounds__(512, 1) void gemm8p_kernel(ConvArgs a, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    // Block -> tile: each XCD (private 4 MiB L2; workgroup ids are dealt round-robin, id = 8 idx + xcd) gets one contiguous band of
    // the (m-block major, n-block minor) tile list, so the 256-pixel activation panel of an m-block is fetched by one XCD only.
    // (Measured: m-minor order and an XCD grid that splits the n-blocks so that each XCD keeps 1/2 of the weights are within +-3 %
    // of this order on every ensemble shape - the kernel is not L2-capacity bound.)
    int mb, nb;
    {
        int bid = blockIdx.x;
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        mb = bid / a.n_blocks;
        nb = bid - mb * a.n_blocks;
    }
    const int mblk = mb * 256, nblk = nb * 256;

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)(2L * a.Cout_g * a.ldw), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + a.cin_off), 0,
                                                                        (unsigned)(a.x_span_bytes - 2L * a.cin_off), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_res =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (unsigned)a.res_span_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_span_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_bias =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)(a.bias_elems * 4) : 0u, 0x00020000);

    // ---- DMA source offsets: this thread moves rows j = 16 wave + 8 i + (lane >> 3), i = 0, 1, of every half-tile -------------
    const int pc = lane & 7;
    unsigned vA[2][2], vB[2][2];                   // [half g / h][i]
    int dma_lds[2];                                // wave-uniform LDS offset of the 8-row group inside a slot
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rg = wave * 2 + i;               // 8-row group 0..15
        const int j = rg * 8 + (lane >> 3);
        const int c = pc ^ ((j >> 1) & 7);
        dma_lds[i] = rg * 1024;
        {   // weights: slot row j -> channel
            const int jj = j & 63, t = (jj >> 4) & 3, r = jj & 15;
            const int ch0 = nblk + (j >> 6) * 128 + (t >> 1) * 32 + (r >> 2) * 8 + (t & 1) * 4 + (r & 3);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ch = ch0 + g * 64;
                vA[g][i] = ch < a.Cout_g ? (unsigned)((ch * a.ldw + c * 8) * 2) : G8_OOB;
            }
        }
        {   // activations: slot row j -> pixel
            const int px0 = mblk + (j >> 5) * 64 + (j & 31);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int px = px0 + h * 32;
                vB[h][i] = px < a.M ? (unsigned)((px * a.ldx + c * 8) * 2) : G8_OOB;
            }
        }
    }
    const int nk = a.K >> 6;
    // stage<KIND>(buffer, k-tile): two DMA instructions per thread; a k-tile past the end reads past the row (never consumed)
    // or out of range (zeros) - issued unconditionally so that the vmcnt bookkeeping is the same in every iteration
    auto stageA = [&](int g, char* slot, int kt) {
        const unsigned ko = (unsigned)kt * 128u;
        g8_dma(rw, slot + dma_lds[0], vA[g][0] + ko);
        g8_dma(rw, slot + dma_lds[1], vA[g][1] + ko);
    };
    auto stageB = [&](int h, char* slot, int kt) {
        const unsigned ko = (unsigned)kt * 128u;
        g8_dma(rx, slot + dma_lds[0], vB[h][0] + ko);
        g8_dma(rx, slot + dma_lds[1], vB[h][1] + ko);
    };

    // ---- fragment read offsets ---------------------------------------------------------------------------------------------
    const int key = (l15 >> 1) & 7;
    const int fo0 = l15 * 128 + ((lq ^ key) << 4);          // k-step 0: logical chunk lq;  k-step 1: chunk 4 + lq = fo0 ^ 64
    const int fo1 = fo0 ^ 64;
    const int a_off = wn * 8192, b_off = wm * 4096;         // this wave's 64 rows of an A slot / 32 rows of a B slot

    f32x4 acc[2][4][4];
    {   // bias is the C operand of the first MFMA of every accumulator
        const int n0 = nblk + wn * 128;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = n0 + g * 64 + (nt >> 1) * 32 + lq * 8 + (nt & 1) * 4;
                const f32x4 bv = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_bias, n < a.Cout_g ? (unsigned)(n * 4) : OOB, 0, 0));
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[g][p][nt] = bv;
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the bias loads are the only ordinary loads: retire them before the DMA ring starts

    // ---- prologue: half-tiles 0..6 = A0,B0,B1,A1 of tile 0 and A0,B0,B1 of tile 1 ---------------------------------------------
    stageA(0, g8_slot<0>(smem, 0), 0);
    stageB(0, g8_slot<1>(smem, 0), 0);
    stageB(1, g8_slot<2>(smem, 0), 0);
    stageA(1, g8_slot<3>(smem, 0), 0);
    stageA(0, g8_slot<0>(smem, 1), 1);
    stageB(0, g8_slot<1>(smem, 1), 1);
    stageB(1, g8_slot<2>(smem, 1), 1);
    if constexpr (PIPE) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // A0, B0, B1 of tile 0 have landed
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                    // all of tile 0
    __builtin_amdgcn_s_barrier();

    // Fragment reads are inline-asm ds_read_b128: hipcc's waitcnt pass makes every compiler-visible LDS read wait for ALL
    // outstanding LDS-DMA (vmcnt(0) in front of each read group - it cannot tell which slot a read touches), which would drain
    // the ring four times per k-tile.  The asm reads are ordered by hand: s_waitcnt lgkmcnt(0) + sched_barrier before the MFMAs.
    U4H8 af[4][2], af2[PIPE ? 4 : 1][2], b0[2][2], b1[2][2];   // [tile][k-step]; af2: the second weight-fragment set of the pipelined schedule
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    const unsigned a_base0 = lds0 + (unsigned)(a_off + fo0), a_base1 = lds0 + (unsigned)(a_off + fo1);
    const unsigned b_base0 = lds0 + (unsigned)(b_off + fo0), b_base1 = lds0 + (unsigned)(b_off + fo1);
#define G8_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define G8_READ_A(KIND)                                                                          \
    do {                                                                                         \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                       \
            G8_DSR(af[nt][0].h, a0, (KIND) * G8_SLOT + nt * 2048);                               \
            G8_DSR(af[nt][1].h, a1, (KIND) * G8_SLOT + nt * 2048);                               \
        }                                                                                        \
    } while (0)
#define G8_READ_B(BF, KIND)                                                                      \
    do {                                                                                         \
        _Pragma("unroll") for (int pp = 0; pp < 2; ++pp) {                                       \
            G8_DSR(BF[pp][0].h, b0a, (KIND) * G8_SLOT + pp * 2048);                              \
            G8_DSR(BF[pp][1].h, b1a, (KIND) * G8_SLOT + pp * 2048);                              \
        }                                                                                        \
    } while (0)
#define G8_LDS_WAIT()                                         \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                    \
    } while (0)
#define G8_MFMA2(G, P0, AF, BF)                                                                                              \
    do {                                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                            \
                _Pragma("unroll") for (int pp = 0; pp < 2; ++pp)                                                        \
                    acc[G][P0 + pp][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AF[nt][ks].h, BF[pp][ks].h,            \
                                                                                 acc[G][P0 + pp][nt], 0, 0, 0);        \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
    } while (0)

#define G8_MFMA(G, P0, BF) G8_MFMA2(G, P0, af, BF)
    if constexpr (PIPE) {
        // ---- software-pipelined schedule: the fragments phase p+1 needs are read (into registers no MFMA of phase p touches)
        // BEFORE phase p's MFMAs, so the LDS pipe works under the matrix pipe instead of between its bursts.  Register roles per
        // k-tile: A0 -> af, A1 -> af2, B0 -> BP, B1 -> BQ with BP / BQ swapping every tile (the B0 of tile t+1 is read into the set
        // that held B1 of tile t).  Reads: p0 B1(t), p1 A1(t), p2 A0(t+1), p3 B0(t+1); DMA as before (p0 A1(t+1), p1 A0(t+2), p2
        // B0(t+2), p3 B1(t+2)) but with s_waitcnt vmcnt(8) in EVERY phase: four half-tiles stay in flight and each one is read five
        // phases after it was issued, one barrier after the wait that retired it.  A slot is rewritten three phases after its
        // last read.  The wait for a phase's reads sits at the END of the phase and names their destinations, so no asm load is in
        // flight across the loop's back edge (hipcc may copy loop-carried registers there).
#define G8_RA(AF, BASE0, BASE1, KIND)                                                            \
    do {                                                                                         \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                       \
            G8_DSR(AF[nt][0].h, BASE0, (KIND) * G8_SLOT + nt * 2048);                            \
            G8_DSR(AF[nt][1].h, BASE1, (KIND) * G8_SLOT + nt * 2048);                            \
        }                                                                                        \
    } while (0)
#define G8_RB(BF, BASE0, BASE1, KIND)                                                            \
    do {                                                                                         \
        _Pragma("unroll") for (int pp = 0; pp < 2; ++pp) {                                       \
            G8_DSR(BF[pp][0].h, BASE0, (KIND) * G8_SLOT + pp * 2048);                            \
            G8_DSR(BF[pp][1].h, BASE1, (KIND) * G8_SLOT + pp * 2048);                            \
        }                                                                                        \
    } while (0)
#define G8_WAIT_A(AF)                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                     \
                 : "+v"(AF[0][0].h), "+v"(AF[0][1].h), "+v"(AF[1][0].h), "+v"(AF[1][1].h), "+v"(AF[2][0].h), "+v"(AF[2][1].h), \
                   "+v"(AF[3][0].h), "+v"(AF[3][1].h)::"memory")
#define G8_WAIT_B(BF) \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(BF[0][0].h), "+v"(BF[0][1].h), "+v"(BF[1][0].h), "+v"(BF[1][1].h)::"memory")
#define G8_PHASE_MID()                                    \
    do {                                                  \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  \
        __builtin_amdgcn_sched_barrier(0);                \
    } while (0)
#define G8_TILE(B, BP, BQ, TT)                                                                         \
    do {                                                                                               \
        char* cur = smem + (B) * 4 * G8_SLOT;                                                          \
        char* nxt = smem + ((B) ^ 1) * 4 * G8_SLOT;                                                    \
        const unsigned co = (unsigned)((B) * 4 * G8_SLOT), no = (unsigned)(((B) ^ 1) * 4 * G8_SLOT);   \
        /* p0: (A0, B0) */                                                                             \
        G8_RB(BQ, b_base0 + co, b_base1 + co, 2);                                                      \
        stageA(1, nxt + 3 * G8_SLOT, (TT) + 1);                                                        \
        G8_PHASE_MID();                                                                                \
        G8_MFMA2(0, 0, af, BP);                                                                        \
        G8_WAIT_B(BQ);                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                  \
        /* p1: (A0, B1) */                                                                             \
        G8_RA(af2, a_base0 + co, a_base1 + co, 3);                                                     \
        stageA(0, cur + 0 * G8_SLOT, (TT) + 2);                                                        \
        G8_PHASE_MID();                                                                                \
        G8_MFMA2(0, 2, af, BQ);                                                                        \
        G8_WAIT_A(af2);                                                                                \
        __builtin_amdgcn_s_barrier();                                                                  \
        /* p2: (A1, B1) */                                                                             \
        G8_RA(af, a_base0 + no, a_base1 + no, 0);                                                      \
        stageB(0, cur + 1 * G8_SLOT, (TT) + 2);                                                        \
        G8_PHASE_MID();                                                                                \
        G8_MFMA2(1, 2, af2, BQ);                                                                       \
        G8_WAIT_A(af);                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                  \
        /* p3: (A1, B0) */                                                                             \
        G8_RB(BQ, b_base0 + no, b_base1 + no, 1);                                                      \
        stageB(1, cur + 2 * G8_SLOT, (TT) + 2);                                                        \
        G8_PHASE_MID();                                                                                \
        G8_MFMA2(1, 0, af2, BP);                                                                       \
        G8_WAIT_B(BQ);                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                  \
    } while (0)
        G8_RA(af, a_base0, a_base1, 0);
        G8_RB(b0, b_base0, b_base1, 1);
        G8_WAIT_A(af);
        G8_WAIT_B(b0);
        for (int t = 0; t < nk; t += 2) {          // nk is even (K % 128 == 0)
            G8_TILE(0, b0, b1, t);
            G8_TILE(1, b1, b0, t + 1);
        }
#undef G8_TILE
#undef G8_PHASE_MID
#undef G8_WAIT_A
#undef G8_WAIT_B
#undef G8_RA
#undef G8_RB
    } else
    for (int t = 0; t < nk; ++t) {
        const int b = t & 1;
        char* cur = smem + b * 4 * G8_SLOT;
        char* nxt = smem + (b ^ 1) * 4 * G8_SLOT;
        const unsigned bo = (unsigned)(b * 4 * G8_SLOT);
        const unsigned a0 = a_base0 + bo, a1 = a_base1 + bo, b0a = b_base0 + bo, b1a = b_base1 + bo;
        // ---- phase 0: (A0, B0) ----
        G8_READ_B(b0, 1);
        G8_READ_A(0);
        stageA(1, nxt + 3 * G8_SLOT, t + 1);
        if constexpr (TWO_BARRIERS) __builtin_amdgcn_s_barrier();
        G8_LDS_WAIT();
        G8_MFMA(0, 0, b0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 1: (A0, B1) ----
        G8_READ_B(b1, 2);
        stageA(0, cur + 0 * G8_SLOT, t + 2);
        if constexpr (TWO_BARRIERS) __builtin_amdgcn_s_barrier();
        G8_LDS_WAIT();
        G8_MFMA(0, 2, b1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: (A1, B1) ----
        G8_READ_A(3);
        stageB(0, cur + 1 * G8_SLOT, t + 2);
        if constexpr (TWO_BARRIERS) __builtin_amdgcn_s_barrier();
        G8_LDS_WAIT();
        G8_MFMA(1, 2, b1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: (A1, B0) ----
        stageB(1, cur + 2 * G8_SLOT, t + 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if constexpr (TWO_BARRIERS) __builtin_amdgcn_s_barrier();
        G8_MFMA(1, 0, b0);
        __builtin_amdgcn_s_barrier();
    }
#undef G8_DSR
#undef G8_READ_A
#undef G8_READ_B
#undef G8_LDS_WAIT
#undef G8_MFMA
#undef G8_MFMA2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the ring's last (unused) half-tiles: nothing may land after the workgroup ends

    int m_base = mblk + wm * 64 + l15, n_lane = nblk + wn * 128 + lq * 8;
    asm volatile("" : "+v"(m_base), "+v"(n_lane));
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int n_first = n_lane + g * 64;
        switch (mode) {
            case 1: pw_epilogue<4, VIP_ACT_RELU, false, false>(a, acc[g], m_base, n_first, rb_res, rb_y); break;
            case 2: pw_epilogue<4, VIP_ACT_SILU, false, false>(a, acc[g], m_base, n_first, rb_res, rb_y); break;
            case 3: pw_epilogue<4, VIP_ACT_GELU, false, false>(a, acc[g], m_base, n_first, rb_res, rb_y); break;
            case 4: pw_epilogue<4, VIP_ACT_SIGMOID, false, false>(a, acc[g], m_base, n_first, rb_res, rb_y); break;
            case 5: pw_epilogue<4, VIP_ACT_NONE, true, false>(a, acc[g], m_base, n_first, rb_res, rb_y); break;
            case 6: pw_epilogue<4, VIP_ACT_NONE, true, true>(a, acc[g], m_base, n_first, rb_res, rb_y); break;
            default: pw_epilogue<4, VIP_ACT_NONE, false, false>(a, acc[g], m_base, n_first, rb_res, rb_y); break;
        }
    }
}

// shapes this kernel takes (host side of the dispatch): 1x1 stride-1 ungrouped, K a multiple of 64, tensors below 2 GiB (the
// out-of-range sentinel must survive "+ k offset")
inline bool gemm8p_eligible(const ConvArgs& a) {
    return a.K % 64 == 0 && a.K >= 128 && a.x_span_bytes < 0x7FFF0000L && 2L * a.Cout_g * a.ldw < 0x7FFF0000L;
}

inline int launch_gemm8p(const ConvArgs& a0, int mode, hipStream_t s) {
    ConvArgs a = a0;
    a.m_blocks = (a.M + 255) / 256;
    a.n_blocks = (a.Cout_g + 255) / 256;
    constexpr size_t smem = 8 * G8_SLOT;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8p_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8p_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8p_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    // VIP_G8P_VARIANT: 2 = software-pipelined fragment reads (default where K % 128 == 0), 1 = one barrier per phase, 0 = two
    static const int variant = getenv("VIP_G8P_VARIANT") ? atoi(getenv("VIP_G8P_VARIANT")) : 2;
    const dim3 grid((unsigned)(a.m_blocks * a.n_blocks));
    if (variant == 2 && a.K % 128 == 0) hipLaunchKernelGGL((gemm8p_kernel<false, true>), grid, dim3(512), smem, s, a, mode);
    else if (variant >= 1) hipLaunchKernelGGL((gemm8p_kernel<false, false>), grid, dim3(512), smem, s, a, mode);
    else hipLaunchKernelGGL((gemm8p_kernel<true, false>), grid, dim3(512), smem, s, a, mode);
    return vip_launch_status("vip_conv2d_nhwc_f16(gemm8p)");
}
