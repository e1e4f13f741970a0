// Does v_mfma_f32_16x16x32_f16 flush SUBNORMAL f16 inputs?  (decides the lo-term format of the packed strict storage)
// Also: v_cvt_f16_f32 producing subnormals, v_pk_mul_f16 on subnormals, and the (hi, lo) split round trip of small values.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_denorm.hip -o /tmp/mfma_denorm && /tmp/mfma_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(float a_val, float b_val, float* out) {
    // A[row][k] = a_val for all, B[k][col] = b_val: every D element = 32 * a * b
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (f16)a_val;
        b[i] = (f16)b_val;
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = c[0];
        out[1] = (float)a[0];                        // what the f16 conversion kept
        f16 h = (f16)a_val;
        f16 m = h * (f16)0.5f;                       // VALU f16 multiply on a (possibly) subnormal
        out[2] = (float)m;
    }
}

__global__ void split_rt(const float* x, float* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const f16 hi = (f16)v;
    const f16 lo = (f16)(v - (float)hi);
    y[i] = (float)hi + (float)lo;
}

int main() {
    float* d;
    hipMalloc(&d, 64);
    const float tests[][2] = {{1.0f, 1.0f}, {ldexpf(1.f, -15), 1024.f}, {ldexpf(1.f, -20), 1024.f}, {ldexpf(1.f, -24), 16384.f},
                              {ldexpf(3.f, -24), 16384.f}, {1024.f, ldexpf(1.f, -20)}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, t[0], t[1], d);
        float h[3];
        hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%.6e b=%.6e  mfma=%.9e expect=%.9e  f16(a)=%.6e  f16(a)*0.5=%.6e  %s\n", t[0], t[1], h[0], 32.0 * t[0] * t[1], h[1], h[2],
               fabs(h[0] - 32.0 * t[0] * t[1]) <= 1e-6 * fabs(32.0 * t[0] * t[1]) ? "KEPT" : "FLUSHED/ROUNDED");
    }
    const int n = 8;
    float hx[n] = {0.1f, 0.01f, 0.001f, 1e-4f, 1e-5f, 3.3e-6f, 1.2345678f, 1000.123f}, hy[n];
    float *dx, *dy;
    hipMalloc(&dx, n * 4);
    hipMalloc(&dy, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split_rt, dim3(1), dim3(64), 0, 0, dx, dy, n);
    hipMemcpy(hy, dy, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("split x=%.9e -> %.9e  rel err %.3e  abs err %.3e\n", hx[i], hy[i], fabs(hy[i] - hx[i]) / fabs(hx[i]), fabs(hy[i] - hx[i]));
    return 0;
}
