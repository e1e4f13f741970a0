// Minimal reproducer for the round-3 window-attention failure ("wrong tiles whenever an MFMA-heavy kernel shares the SIMDs"):
// does the distance a VALU / LDS-return consumer must keep from an MFMA's result depend on what ELSE occupies the SIMD's matrix pipe?
//
// A victim wave executes, from inline asm with fixed registers (nothing for the compiler to schedule or pad):
//   RAW : v[40:43] = 0 ; long drain ; v_mfma_f32_16x16x32_f16 v[40:43], A, B, 0 ; s_nop K ; v_mov r, v40        expect 32 (A = B = 1)
//   WAW : v[40:43] = 0 ; long drain ; v_mfma ... v[40:43] ; s_nop K ; ds_read_b32 v40, lds(= 7.0) ; s_waitcnt lgkmcnt(0) ;
//         s_nop 15 x 4 ; v_mov r, v40                                                                             expect 7
//   CHAIN: v_mfma v[40:43] (C = 0) ; s_nop K ; v_pk_add_f32 v[44:45], v[40:41], one    - the old kernel's own pair         expect 33
// for K = 0 .. 15, alone and next to a kernel that keeps 4 waves per SIMD issuing independent MFMAs back to back.  Every mismatch is
// counted per (test, K).  The ISA requires software wait states between an XDL write and a VALU read / write of the same VGPR
// (LLVM GCNHazardRecognizer::checkMAIVALUHazards); hipcc pads its own code accordingly - the question is whether the REQUIRED count
// is a constant, as that table assumes, when the matrix pipe is contended.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_hazard.hip -o tools/micro/mfma_hazard && tools/micro/mfma_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                      \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

// 16 independent accumulator chains per wave, nothing but MFMAs: the co-runner that corrupted 39 of 40 attention launches in round 3
__global__ __launch_bounds__(256) void mfma_hog(float* sink, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(1.0f + 0.001f * (threadIdx.x & 7));
        b[i] = (_Float16)(0.5f);
    }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    if (s == 12345.678f) sink[0] = s;
}

template <int TEST, int K>
__global__ __launch_bounds__(256) void victim(unsigned* bad, int reps) {
    __shared__ float lds[66];
    if (threadIdx.x < 66) lds[threadIdx.x] = (TEST >= 7 && TEST != 10 && TEST != 11) ? 7.0f + 2.0f * (float)(threadIdx.x & 1) : 7.0f;
    __syncthreads();
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = b[i] = (_Float16)1.0f;
    const unsigned ldsaddr = (unsigned)(unsigned long)(__attribute__((address_space(3))) float*)lds + 4u * (threadIdx.x & 63);
    unsigned nbad = 0;
    for (int r = 0; r < reps; ++r) {
        float got;
        if constexpr (TEST == 0) {
            asm volatile(
                "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"
                "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                "v_mfma_f32_16x16x32_f16 v[40:43], %1, %2, 0\n"
                "s_nop %3\n"
                "v_mov_b32 %0, v40\n"
                "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                : "=v"(got) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43");
            if (got != 32.0f) ++nbad;
        } else if constexpr (TEST == 1) {
            asm volatile(
                "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"
                "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                "v_mfma_f32_16x16x32_f16 v[40:43], %1, %2, 0\n"
                "s_nop %3\n"
                "ds_read_b32 v40, %4\n"
                "s_waitcnt lgkmcnt(0)\n"
                "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                "v_mov_b32 %0, v40\n"
                : "=v"(got) : "v"(a), "v"(b), "n"(K), "v"(ldsaddr) : "v40", "v41", "v42", "v43");
            if (got != 7.0f) ++nbad;
        } else if constexpr (TEST == 12 || TEST == 13) {
            // the kernel's inner pattern, 16 times back to back, no MFMA in the victim: address VALU -> ds_read2_b32 INTO the address
            // register pair -> wait -> v_pk_add_f32 accumulating the returned pair, half-swapped (12) or in natural order (13)
#define HZ_STEP_SW "v_add_u32 v14, 0, %1\n ds_read2_b32 v[14:15], v14 offset1:1\n s_waitcnt lgkmcnt(0)\n v_pk_add_f32 v[40:41], v[40:41], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]\n"
#define HZ_STEP_NO "v_add_u32 v14, 0, %1\n ds_read2_b32 v[14:15], v14 offset1:1\n s_waitcnt lgkmcnt(0)\n v_pk_add_f32 v[40:41], v[40:41], v[14:15]\n"
            const float odd = 7.0f + 2.0f * (float)((threadIdx.x + 1) & 1), even = 7.0f + 2.0f * (float)(threadIdx.x & 1);
            if constexpr (TEST == 12)
                asm volatile("v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n s_nop %2\n"
                             HZ_STEP_SW HZ_STEP_SW HZ_STEP_SW HZ_STEP_SW HZ_STEP_SW HZ_STEP_SW HZ_STEP_SW HZ_STEP_SW
                             HZ_STEP_SW HZ_STEP_SW HZ_STEP_SW HZ_STEP_SW HZ_STEP_SW HZ_STEP_SW HZ_STEP_SW HZ_STEP_SW
                             "s_nop 15\n v_mov_b32 %0, v40\n"
                             : "=v"(got) : "v"(ldsaddr), "n"(K) : "v14", "v15", "v40", "v41");
            else
                asm volatile("v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n s_nop %2\n"
                             HZ_STEP_NO HZ_STEP_NO HZ_STEP_NO HZ_STEP_NO HZ_STEP_NO HZ_STEP_NO HZ_STEP_NO HZ_STEP_NO
                             HZ_STEP_NO HZ_STEP_NO HZ_STEP_NO HZ_STEP_NO HZ_STEP_NO HZ_STEP_NO HZ_STEP_NO HZ_STEP_NO
                             "s_nop 15\n v_mov_b32 %0, v40\n"
                             : "=v"(got) : "v"(ldsaddr), "n"(K) : "v14", "v15", "v40", "v41");
            if (got != 16.0f * (TEST == 12 ? odd : even)) ++nbad;
        } else if constexpr (TEST == 10 || TEST == 11) {
            // THE PAIR the failing ISA shows everywhere (tools/repro variant 1, ws 14 kernel):
            //     v_pk_add_f32 v[114:115], v[24:25], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]
            //     v_add_u32_e32 v14, 0x8e40, v60          <- plain VALU WRITE to a source register of the packed add just issued
            // TEST 10: v_pk_add_f32 reading v[50:51], then s_nop K, then v_mov_b32 v50 / v51 <- garbage.  TEST 11: the same with two scalar
            // v_add_f32 (control).  No MFMA in the victim at all: the only matrix work on the SIMD is the co-runner's.
            if constexpr (TEST == 10)
                asm volatile(
                    "v_mov_b32 v40, 32.0\n v_mov_b32 v41, 32.0\n v_mov_b32 v50, 7.0\n v_mov_b32 v51, 7.0\n"
                    "s_nop 15\n s_nop 15\n"
                    "v_pk_add_f32 v[48:49], v[40:41], v[50:51] op_sel:[0,1] op_sel_hi:[1,0]\n"
                    "s_nop %1\n"
                    "v_mov_b32 v50, 0x8e40\n"
                    "v_mov_b32 v51, 0x8e44\n"
                    "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                    "v_add_f32 %0, v48, v49\n"
                    : "=v"(got) : "n"(K) : "v40", "v41", "v48", "v49", "v50", "v51");
            else
                asm volatile(
                    "v_mov_b32 v40, 32.0\n v_mov_b32 v41, 32.0\n v_mov_b32 v50, 7.0\n v_mov_b32 v51, 7.0\n"
                    "s_nop 15\n s_nop 15\n"
                    "v_add_f32 v48, v40, v51\n"
                    "v_add_f32 v49, v41, v50\n"
                    "s_nop %1\n"
                    "v_mov_b32 v50, 0x8e40\n"
                    "v_mov_b32 v51, 0x8e44\n"
                    "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                    "v_add_f32 %0, v48, v49\n"
                    : "=v"(got) : "n"(K) : "v40", "v41", "v48", "v49", "v50", "v51");
            if (got != 78.0f) ++nbad;
        } else if constexpr (TEST >= 7 && TEST <= 9) {
            // what the bisect of the real kernel points at (tools/repro: scalar adds or table values read BEFORE the MFMAs cure it, 64 idle
            // cycles after the MFMAs do not): v_pk_add_f32 with a HALF-SWAPPED source (op_sel:[0,1] op_sel_hi:[1,0] - hipcc's way of
            // using the two floats one ds_read2_b32 returned in reversed order) on a register pair that has just come back from LDS.
            //   TEST 7: MFMA result + fresh LDS pair, half-swapped      TEST 8: constant + fresh LDS pair, half-swapped, an unrelated MFMA
            //   in flight      TEST 9: TEST 7 without the swap.  lds[j] = 7 + 2 (j & 1): lo result = 32 + lds[lane + 1] (swapped) / lds[lane].
            const float odd = 7.0f + 2.0f * (float)((threadIdx.x + 1) & 1), even = 7.0f + 2.0f * (float)(threadIdx.x & 1);
            if constexpr (TEST == 7)
                asm volatile(
                    "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"
                    "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                    "v_mfma_f32_16x16x32_f16 v[40:43], %1, %2, 0\n"
                    "v_mfma_f32_16x16x32_f16 v[44:47], %1, %2, 0\n"
                    "s_nop 15\n s_nop 15\n"
                    "ds_read2_b32 v[50:51], %4 offset1:1\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "s_nop %3\n"
                    "v_pk_add_f32 v[48:49], v[40:41], v[50:51] op_sel:[0,1] op_sel_hi:[1,0]\n"
                    "s_nop 15\n s_nop 15\n"
                    "v_mov_b32 %0, v48\n"
                    : "=v"(got) : "v"(a), "v"(b), "n"(K), "v"(ldsaddr)
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
            else if constexpr (TEST == 8)
                asm volatile(
                    "v_mov_b32 v40, 32.0\n v_mov_b32 v41, 32.0\n"
                    "s_nop 15\n s_nop 15\n"
                    "v_mfma_f32_16x16x32_f16 v[44:47], %1, %2, 0\n"
                    "ds_read2_b32 v[50:51], %4 offset1:1\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "s_nop %3\n"
                    "v_pk_add_f32 v[48:49], v[40:41], v[50:51] op_sel:[0,1] op_sel_hi:[1,0]\n"
                    "s_nop 15\n s_nop 15\n"
                    "v_mov_b32 %0, v48\n"
                    : "=v"(got) : "v"(a), "v"(b), "n"(K), "v"(ldsaddr)
                    : "v40", "v41", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
            else
                asm volatile(
                    "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"
                    "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                    "v_mfma_f32_16x16x32_f16 v[40:43], %1, %2, 0\n"
                    "v_mfma_f32_16x16x32_f16 v[44:47], %1, %2, 0\n"
                    "s_nop 15\n s_nop 15\n"
                    "ds_read2_b32 v[50:51], %4 offset1:1\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "s_nop %3\n"
                    "v_pk_add_f32 v[48:49], v[40:41], v[50:51]\n"
                    "s_nop 15\n s_nop 15\n"
                    "v_mov_b32 %0, v48\n"
                    : "=v"(got) : "v"(a), "v"(b), "n"(K), "v"(ldsaddr)
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
            if (got != 32.0f + (TEST == 9 ? even : odd)) ++nbad;
        } else if constexpr (TEST == 5 || TEST == 6) {
            // WAR on an MFMA SOURCE operand: three independent MFMAs back to back share their B operand v[60:63] (the round-2 kernel's
            // last QK MFMAs, old14.s lines 839-841, share the Q fragment v[14:17]), and the very next instruction OVERWRITES v60 -
            // TEST 5 with a VALU write (there: v_add_u32 v14, ...), TEST 6 with an LDS load (there: ds_read2_b32 v[14:15]) - after
            // s_nop K.  hipcc emits K = none: its hazard table has no entry "XDL read SrcA/B -> VALU / LDS write".  B = 1 -> 32.
            if constexpr (TEST == 5)
                asm volatile(
                    "v_mov_b32 v60, %1\n v_mov_b32 v61, %1\n v_mov_b32 v62, %1\n v_mov_b32 v63, %1\n"
                    "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                    "v_mfma_f32_16x16x32_f16 v[40:43], %2, v[60:63], 0\n"
                    "v_mfma_f32_16x16x32_f16 v[44:47], %2, v[60:63], 0\n"
                    "v_mfma_f32_16x16x32_f16 v[48:51], %2, v[60:63], 0\n"
                    "s_nop %3\n"
                    "v_mov_b32 v60, 0x4d004d00\n"
                    "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                    "v_mov_b32 %0, v48\n"
                    : "=v"(got) : "v"(0x3c003c00u), "v"(a), "n"(K), "v"(ldsaddr)
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v60", "v61", "v62", "v63");
            else
                asm volatile(
                    "v_mov_b32 v60, %1\n v_mov_b32 v61, %1\n v_mov_b32 v62, %1\n v_mov_b32 v63, %1\n"
                    "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                    "v_mfma_f32_16x16x32_f16 v[40:43], %2, v[60:63], 0\n"
                    "v_mfma_f32_16x16x32_f16 v[44:47], %2, v[60:63], 0\n"
                    "v_mfma_f32_16x16x32_f16 v[48:51], %2, v[60:63], 0\n"
                    "s_nop %3\n"
                    "ds_read_b32 v60, %4\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                    "v_mov_b32 %0, v48\n"
                    : "=v"(got) : "v"(0x3c003c00u), "v"(a), "n"(K), "v"(ldsaddr)
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v60", "v61", "v62", "v63");
            if (got != 32.0f) ++nbad;
        } else if constexpr (TEST == 3 || TEST == 4) {
            // the round-2 window-attention kernel's own instruction sequence (hipcc's output, /tmp/race/old14.s lines 812-820 of the
            // ws 14 kernel): two independent MFMAs, the first one's result consumed by v_pk_add_f32 after
            // [ds_read2_b32, s_mov, s_mov, v_mfma, v_lshrrev, s_waitcnt, s_nop K].  TEST 3: in place (vDst = SrcA, as compiled);
            // TEST 4: vDst in registers of its own.  A = B = 1 -> 32; table value 7 -> expect 39.
            if constexpr (TEST == 3)
                asm volatile(
                    "v_mov_b32 v40, %1\n v_mov_b32 v41, %1\n v_mov_b32 v42, %1\n v_mov_b32 v43, %1\n"
                    "v_mov_b32 v44, %1\n v_mov_b32 v45, %1\n v_mov_b32 v46, %1\n v_mov_b32 v47, %1\n"
                    "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                    "v_mfma_f32_16x16x32_f16 v[40:43], v[40:43], %2, 0\n"
                    "ds_read2_b32 v[50:51], %4 offset1:1\n"
                    "s_mov_b32 s20, 0xf149f2ca\n"
                    "s_mov_b32 s21, s20\n"
                    "v_mfma_f32_16x16x32_f16 v[44:47], v[44:47], %2, 0\n"
                    "v_lshrrev_b32 v52, 2, v52\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "s_nop %3\n"
                    "v_pk_add_f32 v[48:49], v[40:41], v[50:51]\n"
                    "s_nop 15\n s_nop 15\n"
                    "v_mov_b32 %0, v48\n"
                    : "=v"(got) : "v"(0x3c003c00u), "v"(b), "n"(K), "v"(ldsaddr)
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "s20", "s21");
            else
                asm volatile(
                    "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"
                    "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"
                    "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                    "v_mfma_f32_16x16x32_f16 v[40:43], %1, %2, 0\n"
                    "ds_read2_b32 v[50:51], %4 offset1:1\n"
                    "s_mov_b32 s20, 0xf149f2ca\n"
                    "s_mov_b32 s21, s20\n"
                    "v_mfma_f32_16x16x32_f16 v[44:47], %1, %2, 0\n"
                    "v_lshrrev_b32 v52, 2, v52\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "s_nop %3\n"
                    "v_pk_add_f32 v[48:49], v[40:41], v[50:51]\n"
                    "s_nop 15\n s_nop 15\n"
                    "v_mov_b32 %0, v48\n"
                    : "=v"(got) : "v"(a), "v"(b), "n"(K), "v"(ldsaddr)
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "s20", "s21");
            if (got != 39.0f) ++nbad;
        } else {
            asm volatile(
                "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"
                "v_mov_b32 v46, 1.0\n v_mov_b32 v47, 1.0\n"
                "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                "v_mfma_f32_16x16x32_f16 v[40:43], %1, %2, 0\n"
                "s_nop %3\n"
                "v_pk_add_f32 v[44:45], v[40:41], v[46:47]\n"
                "s_nop 15\n s_nop 15\n"
                "v_mov_b32 %0, v44\n"
                : "=v"(got) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
            if (got != 33.0f) ++nbad;
        }
    }
    if (nbad) atomicAdd(&bad[TEST * 16 + K], nbad);
}

template <int TEST, int K>
void launch_one(unsigned* bad, int reps, hipStream_t s) {
    hipLaunchKernelGGL((victim<TEST, K>), dim3(1024), dim3(256), 0, s, bad, reps);
}

template <int TEST>
void launch_all(unsigned* bad, int reps, hipStream_t s) {
    launch_one<TEST, 0>(bad, reps, s);
    launch_one<TEST, 1>(bad, reps, s);
    launch_one<TEST, 2>(bad, reps, s);
    launch_one<TEST, 3>(bad, reps, s);
    launch_one<TEST, 4>(bad, reps, s);
    launch_one<TEST, 5>(bad, reps, s);
    launch_one<TEST, 6>(bad, reps, s);
    launch_one<TEST, 7>(bad, reps, s);
    launch_one<TEST, 8>(bad, reps, s);
    launch_one<TEST, 10>(bad, reps, s);
    launch_one<TEST, 12>(bad, reps, s);
    launch_one<TEST, 15>(bad, reps, s);
}

int main() {
    unsigned* bad;
    float* sink;
    CHECK(hipMalloc(&bad, 224 * 4));
    CHECK(hipMalloc(&sink, 64));
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa));
    CHECK(hipStreamCreate(&sb));
    const int reps = 200;
    const char* names[14] = {"RAW  mfma -> s_nop K -> v_mov (VALU read of the result)          ",
                            "WAW  mfma -> s_nop K -> ds_read into its vDst -> (long wait) -> read",
                            "PAIR mfma (C = 0) -> s_nop K -> v_pk_add_f32 on the result           ",
                            "OLD  the round-2 kernel's sequence, vDst = SrcA (hipcc emitted K = 1)",
                            "OLD' the same with vDst in registers of its own                      ",
                            "WARv 3 mfma sharing SrcB -> s_nop K -> VALU write to SrcB's register  ",
                            "WARl 3 mfma sharing SrcB -> s_nop K -> ds_read into SrcB's register   ",
                            "SWAP mfma result + fresh ds_read2 pair, v_pk_add_f32 half-swapped     ",
                            "SWAPc constant + fresh ds_read2 pair, half-swapped, mfma in flight    ",
                            "NOSW mfma result + fresh ds_read2 pair, v_pk_add_f32 plain            ",
                            "PKWAR v_pk_add_f32 -> s_nop K -> plain VALU write to its SOURCE pair   ",
                            "SCWAR two v_add_f32 -> s_nop K -> plain VALU write to their sources    ",
                            "LOOPs 16 x [addr; ds_read2 into addr pair; wait; v_pk_add_f32 SWAPPED]",
                            "LOOPn 16 x [addr; ds_read2 into addr pair; wait; v_pk_add_f32 plain]  "};
    const int ks[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 15};
    for (int pass = 0; pass < 2; ++pass) {
        CHECK(hipMemset(bad, 0, 224 * 4));
        if (pass == 1)      // 4096 blocks x 4 waves of back-to-back MFMAs: several waves per SIMD for the whole victim run
            hipLaunchKernelGGL(mfma_hog, dim3(4096), dim3(256), 0, sb, sink, 60000);
        launch_all<0>(bad, reps, sa);
        launch_all<1>(bad, reps, sa);
        launch_all<2>(bad, reps, sa);
        launch_all<3>(bad, reps, sa);
        launch_all<4>(bad, reps, sa);
        launch_all<5>(bad, reps, sa);
        launch_all<6>(bad, reps, sa);
        launch_all<7>(bad, reps, sa);
        launch_all<8>(bad, reps, sa);
        launch_all<9>(bad, reps, sa);
        launch_all<10>(bad, reps, sa);
        launch_all<11>(bad, reps, sa);
        launch_all<12>(bad, reps, sa);
        launch_all<13>(bad, reps, sa);
        CHECK(hipStreamSynchronize(sa));
        std::vector<unsigned> h(224);
        CHECK(hipMemcpy(h.data(), bad, 224 * 4, hipMemcpyDeviceToHost));
        CHECK(hipDeviceSynchronize());
        printf("== %s (mismatches of %ld lane-results per cell)\n", pass ? "NEXT TO the MFMA-saturating kernel" : "alone", 1024L * 256 * reps);
        for (int t = 0; t < 14; ++t) {
            printf("%s :", names[t]);
            for (int i = 0; i < 12; ++i) printf(" K=%d:%u", ks[i], h[t * 16 + ks[i]]);
            printf("\n");
        }
    }
    return 0;
}
