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
// Basic schedule (k-tile t lives in buffer t & 1; phase p of tile t; used when K % 128 != 0):
//   p0: read B0,A0 frags | DMA A1(t+1) | MFMA (A0,B0)        p1: read B1 | DMA A0(t+2) | MFMA (A0,B1)
//   p2: read A1          | DMA B0(t+2) | MFMA (A1,B1)        p3: -       | DMA B1(t+2) | vmcnt(6) | MFMA (A1,B0)
//   A slot is rewritten one phase after its last read (the phase-end barrier orders it); the wait of p3 leaves the three newest
//   half-tiles in flight, so everything tile t+1 needs has landed, and it is read one barrier later.
// Pipelined schedule (K % 128 == 0): see the comment at the loop.
// Persistent: a workgroup walks output tiles b, b + grid, ...; the DMA prologue of the next tile is issued BEFORE the epilogue of
// the current one (the accumulators are in registers, LDS is free), so the fill latency and the store tail of consecutive tiles
// overlap - with one 128 KiB workgroup per CU nothing else could hide them (measured on the ensemble shapes: time = (3.7 + K/64)
// k-tile times per output tile before, i.e. 24-38 % of a K = 768 / 384 tile was fill + drain).
#pragma once

constexpr int G8_SLOT = 128 * 128;                 // bytes per half-tile slot
// Out-of-range voffset for rows / k-tiles that do not exist (the buffer descriptor returns zeros, no traffic).  A row sentinel plus
// an in-range k offset stays >= 2^31; row sentinel + K SENTINEL would wrap to 0 and fetch real data from offset 0, so the two are
// combined with g8_voff() (saturating), never added.
constexpr unsigned G8_OOB = 0x80000000u;
__device__ __forceinline__ unsigned g8_voff(unsigned row_off, unsigned k_off) {
    return ((row_off | k_off) & G8_OOB) ? G8_OOB : row_off + k_off;
}

template <int KIND>   // 0 = A0, 1 = B0, 2 = B1, 3 = A1 : slot order inside a buffer
__device__ __forceinline__ char* g8_slot(char* smem, int buf) { return smem + (buf * 4 + KIND) * G8_SLOT; }

__device__ __forceinline__ void g8_dma(const __amdgpu_buffer_rsrc_t& r, char* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}

template <bool PIPE>
__global__ __launch_bounds__(512, 1) void gemm8p_kernel(ConvArgs a, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    // Workgroup -> tile sequence.  Hardware deals workgroup ids round-robin to the 8 XCDs (private 4 MiB L2 each): id = 8 idx + xcd.
    // Logical id = xcd * (grid / 8) + idx, tile = logical + round * grid: in every round an XCD works on one contiguous run of the
    // (m-block major, n-block minor) tile list, so the 256-pixel activation panel of an m-block is fetched by one XCD only.
    // (Measured: m-minor order and an XCD grid that splits the n-blocks so that each XCD keeps half of the weights are within
    // +-3 % of this order on every ensemble shape - the kernel is not L2-capacity bound.)
    const int n_tiles = a.m_blocks * a.n_blocks;
    int tile;
    {
        const int bid = blockIdx.x, nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_step = gridDim.x;

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)(2L * a.Cout_g * a.ldw), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + a.cin_off), 0,
                                                                        (unsigned)(a.x_span_bytes - 2L * a.cin_off), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_res =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (unsigned)a.res_span_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (unsigned)a.y_span_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_bias =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)(a.bias_elems * 4) : 0u, 0x00020000);

    // ---- DMA plan: this thread moves rows j = 16 wave + 8 i + (lane >> 3), i = 0, 1, of every half-tile --------------------------
    const int pc = lane & 7;
    unsigned vA[2][2], vB[2][2];                   // [half g / h][i]: source byte offsets of the current tile
    int dma_lds[2];                                // wave-uniform LDS offset of the 8-row group inside a slot
    int rowA[2], rowB[2], chk[2];                  // tile-independent parts: channel / pixel of row j inside the tile, source chunk
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rg = wave * 2 + i;               // 8-row group 0..15
        const int j = rg * 8 + (lane >> 3);
        // physical chunk pc of LDS row j holds logical POSITION pc ^ key(j); H2: position L is source chunk ((L & 3) << 1) | (L >> 2)
        // (the inverse of h2_pos: the four hi chunks of the 64-half row first, then the four lo chunks)
        const int lpos = pc ^ ((j >> 1) & 7);
        chk[i] = H2 ? (((lpos & 3) << 1) | (lpos >> 2)) : lpos;
        dma_lds[i] = rg * 1024;
        const int jj = j & 63, t = (jj >> 4) & 3, r = jj & 15;
        rowA[i] = (j >> 6) * 128 + (t >> 1) * 32 + (r >> 2) * 8 + (t & 1) * 4 + (r & 3);     // + 64 g
        rowB[i] = (j >> 5) * 64 + (j & 31);                                                  // + 32 h
    }
    int mblk = 0, nblk = 0;
    auto setup_tile = [&](int tl) {
        const int mb = tl / a.n_blocks, nb = tl - mb * a.n_blocks;
        mblk = mb * 256;
        nblk = nb * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ch = nblk + rowA[i] + g * 64;
                vA[g][i] = ch < a.Cout_g ? (unsigned)((ch * a.ldw + chk[i] * 8) * 2) : G8_OOB;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int px = mblk + rowB[i] + h * 32;
                vB[h][i] = px < a.M ? (unsigned)((px * a.ldx + chk[i] * 8) * 2) : G8_OOB;
            }
        }
    };
    const int nk = a.K >> 6;
    // stage(slot, k-tile): two DMA instructions per thread.  The ring runs two k-tiles past the end of K; those stages are issued
    // all the same (the vmcnt bookkeeping is identical in every iteration) but with an out-of-range offset: zeros, no memory traffic
    // (in range they fetched 2/nk extra bytes per tile: PMC read 1.35x the algorithmic bytes on the K = 384 layers).
    auto stageA = [&](int g, char* slot, int kt) {
        const unsigned ko = kt < nk ? (unsigned)kt * 128u : G8_OOB;
        g8_dma(rw, slot + dma_lds[0], g8_voff(vA[g][0], ko));
        g8_dma(rw, slot + dma_lds[1], g8_voff(vA[g][1], ko));
    };
    auto stageB = [&](int h, char* slot, int kt) {
        const unsigned ko = kt < nk ? (unsigned)kt * 128u : G8_OOB;
        g8_dma(rx, slot + dma_lds[0], g8_voff(vB[h][0], ko));
        g8_dma(rx, slot + dma_lds[1], g8_voff(vB[h][1], ko));
    };
    // prologue: half-tiles 0..6 = A0,B0,B1,A1 of k-tile 0 and A0,B0,B1 of k-tile 1
    auto issue_prologue = [&]() {
        stageA(0, g8_slot<0>(smem, 0), 0);
        stageB(0, g8_slot<1>(smem, 0), 0);
        stageB(1, g8_slot<2>(smem, 0), 0);
        stageA(1, g8_slot<3>(smem, 0), 0);
        stageA(0, g8_slot<0>(smem, 1), 1);
        stageB(0, g8_slot<1>(smem, 1), 1);
        stageB(1, g8_slot<2>(smem, 1), 1);
    };

    // ---- fragment read offsets ---------------------------------------------------------------------------------------------
    const int key = (l15 >> 1) & 7;
    const int fo0 = l15 * 128 + ((lq ^ key) << 4);          // k-step 0: logical chunk lq;  k-step 1: chunk 4 + lq = fo0 ^ 64
    const int fo1 = fo0 ^ 64;
    const int a_off = wn * 8192, b_off = wm * 4096;         // this wave's 64 rows of an A slot / 32 rows of a B slot
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    const unsigned a_base0 = lds0 + (unsigned)(a_off + fo0), a_base1 = lds0 + (unsigned)(a_off + fo1);
    const unsigned b_base0 = lds0 + (unsigned)(b_off + fo0), b_base1 = lds0 + (unsigned)(b_off + fo1);

    // Fragment reads are inline-asm ds_read_b128: hipcc's waitcnt pass makes every compiler-visible LDS read wait for ALL
    // outstanding LDS-DMA (vmcnt(0) in front of each read group - it cannot tell which slot a read touches), which would drain
    // the ring four times per k-tile.  The asm reads are ordered by hand (s_waitcnt lgkmcnt(0) statements naming their
    // destinations; sched_barrier keeps the MFMAs below the waits they depend on).
    U4H8 af[4][2], af2[PIPE ? 4 : 1][2], b0[2][2], b1[2][2];   // [tile][k-step]; af2: second weight-fragment set (pipelined schedule)
#define G8_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
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
#define G8_MFMA2(G, P0, AF, BF)                                                                                         \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        if constexpr (H2) {     /* [0] = hi plane, [1] = lo plane: w_lo x_hi + w_hi x_hi + w_hi x_lo */                         \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                            \
                _Pragma("unroll") for (int pp = 0; pp < 2; ++pp) {                                                      \
                    acc[G][P0 + pp][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AF[nt][1].h, BF[pp][0].h,              \
                                                                                 acc[G][P0 + pp][nt], 0, 0, 0);        \
                    acc[G][P0 + pp][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AF[nt][0].h, BF[pp][0].h,              \
                                                                                 acc[G][P0 + pp][nt], 0, 0, 0);        \
                }                                                                                                       \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                            \
                _Pragma("unroll") for (int pp = 0; pp < 2; ++pp)                                                        \
                    acc[G][P0 + pp][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AF[nt][0].h, BF[pp][1].h,              \
                                                                                 acc[G][P0 + pp][nt], 0, 0, 0);        \
        } else {                                                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                            \
                _Pragma("unroll") for (int pp = 0; pp < 2; ++pp)                                                        \
                    acc[G][P0 + pp][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AF[nt][ks].h, BF[pp][ks].h,            \
                                                                                 acc[G][P0 + pp][nt], 0, 0, 0);        \
        }                                                                                                               \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
    } while (0)

    setup_tile(tile);
    issue_prologue();
    while (true) {
        f32x4 acc[2][4][4];
        {   // bias is the C operand of the first MFMA of every accumulator.  These are ordinary loads: hipcc waits vmcnt(0) before
            // their first use, which also retires the prologue DMA issued ahead of them - wanted here, the tile starts right after.
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        if constexpr (PIPE) {
            // ---- software-pipelined schedule: the fragments phase p+1 needs are read (into registers no MFMA of phase p touches)
            // BEFORE phase p's MFMAs, so the LDS pipe works under the matrix pipe instead of between its bursts.  Register roles per
            // k-tile: A0 -> af, A1 -> af2, B0 -> BP, B1 -> BQ with BP / BQ swapping every tile (the B0 of tile t+1 is read into the
            // set that held B1 of tile t).  Reads: p0 B1(t), p1 A1(t), p2 A0(t+1), p3 B0(t+1); DMA: p0 A1(t+1), p1 A0(t+2), p2
            // B0(t+2), p3 B1(t+2) with s_waitcnt vmcnt(8) in EVERY phase: four half-tiles stay in flight and each one is read five
            // phases after it was issued, one barrier after the wait that retired it.  A slot is rewritten three phases after its
            // last read.  The wait for a phase's reads sits at the END of the phase and names their destinations, so no asm load is
            // in flight across the loop's back edge (hipcc may copy loop-carried registers there).
#define G8_TILE(B, BP, BQ, TT)                                                                         \
    do {                                                                                               \
        char* cur = smem + (B) * 4 * G8_SLOT;                                                          \
        char* nxt = smem + ((B) ^ 1) * 4 * G8_SLOT;                                                    \
        const unsigned co = (unsigned)((B) * 4 * G8_SLOT), no = (unsigned)(((B) ^ 1) * 4 * G8_SLOT);   \
        /* p0: (A0, B0) */                                                                             \
        G8_RB(BQ, b_base0 + co, b_base1 + co, 2);                                                      \
        stageA(1, nxt + 3 * G8_SLOT, (TT) + 1);                                                        \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                               \
        G8_MFMA2(0, 0, af, BP);                                                                        \
        G8_WAIT_B(BQ);                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                  \
        /* p1: (A0, B1) */                                                                             \
        G8_RA(af2, a_base0 + co, a_base1 + co, 3);                                                     \
        stageA(0, cur + 0 * G8_SLOT, (TT) + 2);                                                        \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                               \
        G8_MFMA2(0, 2, af, BQ);                                                                        \
        G8_WAIT_A(af2);                                                                                \
        __builtin_amdgcn_s_barrier();                                                                  \
        /* p2: (A1, B1) */                                                                             \
        G8_RA(af, a_base0 + no, a_base1 + no, 0);                                                      \
        stageB(0, cur + 1 * G8_SLOT, (TT) + 2);                                                        \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                               \
        G8_MFMA2(1, 2, af2, BQ);                                                                       \
        G8_WAIT_A(af);                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                  \
        /* p3: (A1, B0) */                                                                             \
        G8_RB(BQ, b_base0 + no, b_base1 + no, 1);                                                      \
        stageB(1, cur + 2 * G8_SLOT, (TT) + 2);                                                        \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                               \
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
        } else {
            for (int t = 0; t < nk; ++t) {
                const int b = t & 1;
                char* cur = smem + b * 4 * G8_SLOT;
                char* nxt = smem + (b ^ 1) * 4 * G8_SLOT;
                const unsigned bo = (unsigned)(b * 4 * G8_SLOT);
                // ---- phase 0: (A0, B0) ----
                G8_RB(b0, b_base0 + bo, b_base1 + bo, 1);
                G8_RA(af, a_base0 + bo, a_base1 + bo, 0);
                stageA(1, nxt + 3 * G8_SLOT, t + 1);
                G8_WAIT_B(b0);
                G8_WAIT_A(af);
                G8_MFMA2(0, 0, af, b0);
                __builtin_amdgcn_s_barrier();
                // ---- phase 1: (A0, B1) ----
                G8_RB(b1, b_base0 + bo, b_base1 + bo, 2);
                stageA(0, cur + 0 * G8_SLOT, t + 2);
                G8_WAIT_B(b1);
                G8_MFMA2(0, 2, af, b1);
                __builtin_amdgcn_s_barrier();
                // ---- phase 2: (A1, B1) ----
                G8_RA(af, a_base0 + bo, a_base1 + bo, 3);
                stageB(0, cur + 1 * G8_SLOT, t + 2);
                G8_WAIT_A(af);
                G8_MFMA2(1, 2, af, b1);
                __builtin_amdgcn_s_barrier();
                // ---- phase 3: (A1, B0) ----
                stageB(1, cur + 2 * G8_SLOT, t + 2);
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                G8_MFMA2(1, 0, af, b0);
                __builtin_amdgcn_s_barrier();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the ring's last (unused) half-tiles: nothing may land after this point

        // this tile's output coordinates; then the next tile's fill goes out ahead of the store tail
        int m_base = mblk + wm * 64 + l15, n_lane = nblk + wn * 128 + lq * 8;
        asm volatile("" : "+v"(m_base), "+v"(n_lane));
        tile += tile_step;
        const bool more = tile < n_tiles;                   // workgroup-uniform
        if (more) {
            setup_tile(tile);
            issue_prologue();                               // every wave is past the k-loop's last barrier: no fragment read is pending
        }
        // (not a loop over g: with the packed-storage epilogue inlined seven times the unroller gives up and the accumulators would be
        // indexed dynamically, i.e. live in scratch)
        auto epi = [&](auto gtag) {
            constexpr int G_ = decltype(gtag)::value;
            const int n_first = n_lane + G_ * 64;
            switch (mode) {
                case 1: pw_epilogue<4, VIP_ACT_RELU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
                case 2: pw_epilogue<4, VIP_ACT_SILU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
                case 3: pw_epilogue<4, VIP_ACT_GELU, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
                case 4: pw_epilogue<4, VIP_ACT_SIGMOID, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
                case 5: pw_epilogue<4, VIP_ACT_NONE, true, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
                case 6: pw_epilogue<4, VIP_ACT_NONE, true, true>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
                default: pw_epilogue<4, VIP_ACT_NONE, false, false>(a, acc[G_], m_base, n_first, rb_res, rb_y); break;
            }
        };
        epi(std::integral_constant<int, 0>{});
        if constexpr (2 > 1) epi(std::integral_constant<int, 1>{});
        if (!more) break;
    }
#undef G8_DSR
#undef G8_RA
#undef G8_RB
#undef G8_WAIT_A
#undef G8_WAIT_B
#undef G8_MFMA2
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
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8p_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8p_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    // one 128 KiB workgroup per CU; VIP_G8P_PERSIST=0 launches one workgroup per tile instead (no fill / drain overlap)
    static const int persist = getenv("VIP_G8P_PERSIST") ? atoi(getenv("VIP_G8P_PERSIST")) : 1;
    static const int cus = [] {
        hipDeviceProp_t pr;
        int dev = 0;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
                   ? pr.multiProcessorCount : 256;
    }();
    const int tiles = a.m_blocks * a.n_blocks;
    const dim3 grid((unsigned)(persist ? (tiles < cus ? tiles : cus) : tiles));
    if (a.K % 128 == 0) hipLaunchKernelGGL((gemm8p_kernel<true>), grid, dim3(512), smem, s, a, mode);
    else hipLaunchKernelGGL((gemm8p_kernel<false>), grid, dim3(512), smem, s, a, mode);
    return vip_launch_status("vip_conv2d_nhwc_f16(gemm8p)");
}
