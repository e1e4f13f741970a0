// Issue rate of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 operands, unit scales) against v_mfma_f32_16x16x32_f16 on this box: the
// premise of DESIGN.md section 8.1 (the two small terms of a packed-storage product at twice the fp16 matrix rate).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_fp8_rate tools/micro/mfma_fp8_rate.hip && tools/micro/mfma_fp8_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void fp8_kernel(float* out, int iters) {
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x3C3C3C3C - threadIdx.x; }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 0, 0, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 0, 0, 0, 127, 0, 127);
        c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 0, 0, 0, 127, 0, 127);
        c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 0, 0, 0, 127, 0, 127);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(256) void f16_kernel(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.5f + 0.001f * threadIdx.x); b[i] = (_Float16)(0.25f); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4, iters = 20000;      // 4 workgroups x 4 waves per CU = 4 waves per SIMD
    for (int rep = 0; rep < 3; ++rep) {
        for (int which = 0; which < 2; ++which) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(f16_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(fp8_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double k = which == 0 ? 32.0 : 128.0;
            const double flop = 2.0 * 16 * 16 * k * 4.0 * iters * (blocks * 4.0);
            printf("%s: %.2f ms, %.0f TFLOP/s, %.1f ns per instruction per SIMD\n", which == 0 ? "v_mfma_f32_16x16x32_f16        " : "v_mfma_scale_f32_16x16x128 fp8",
                   ms, flop / ms / 1e9, ms * 1e6 / (4.0 * iters * 4.0));
        }
    }
    return 0;
}
