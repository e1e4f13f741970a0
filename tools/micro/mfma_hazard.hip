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
    __shared__ float lds[64];
    lds[threadIdx.x & 63] = 7.0f;
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
    CHECK(hipMalloc(&bad, 112 * 4));
    CHECK(hipMalloc(&sink, 64));
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa));
    CHECK(hipStreamCreate(&sb));
    const int reps = 200;
    const char* names[7] = {"RAW  mfma -> s_nop K -> v_mov (VALU read of the result)          ",
                            "WAW  mfma -> s_nop K -> ds_read into its vDst -> (long wait) -> read",
                            "PAIR mfma (C = 0) -> s_nop K -> v_pk_add_f32 on the result           ",
                            "OLD  the round-2 kernel's sequence, vDst = SrcA (hipcc emitted K = 1)",
                            "OLD' the same with vDst in registers of its own                      ",
                            "WARv 3 mfma sharing SrcB -> s_nop K -> VALU write to SrcB's register  ",
                            "WARl 3 mfma sharing SrcB -> s_nop K -> ds_read into SrcB's register   "};
    const int ks[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 15};
    for (int pass = 0; pass < 2; ++pass) {
        CHECK(hipMemset(bad, 0, 112 * 4));
        if (pass == 1)      // 4096 blocks x 4 waves of back-to-back MFMAs: several waves per SIMD for the whole victim run
            hipLaunchKernelGGL(mfma_hog, dim3(4096), dim3(256), 0, sb, sink, 60000);
        launch_all<0>(bad, reps, sa);
        launch_all<1>(bad, reps, sa);
        launch_all<2>(bad, reps, sa);
        launch_all<3>(bad, reps, sa);
        launch_all<4>(bad, reps, sa);
        launch_all<5>(bad, reps, sa);
        launch_all<6>(bad, reps, sa);
        CHECK(hipStreamSynchronize(sa));
        std::vector<unsigned> h(112);
        CHECK(hipMemcpy(h.data(), bad, 112 * 4, hipMemcpyDeviceToHost));
        CHECK(hipDeviceSynchronize());
        printf("== %s (mismatches of %ld lane-results per cell)\n", pass ? "NEXT TO the MFMA-saturating kernel" : "alone", 1024L * 256 * reps);
        for (int t = 0; t < 7; ++t) {
            printf("%s :", names[t]);
            for (int i = 0; i < 12; ++i) printf(" K=%d:%u", ks[i], h[t * 16 + ks[i]]);
            printf("\n");
        }
    }
    return 0;
}
