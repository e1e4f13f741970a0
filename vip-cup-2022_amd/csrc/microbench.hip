// On-box peak probes (bench.py roofline denominators) and the workspace query.
#include "common.hpp"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy16_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // four independent 16-byte loads in flight per lane
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(src + i);
        const u32x4 b = __builtin_nontemporal_load(src + i + stride);
        const u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride);
        const u32x4 d = __builtin_nontemporal_load(src + i + 3 * stride);
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + stride);
        __builtin_nontemporal_store(c, dst + i + 2 * stride);
        __builtin_nontemporal_store(d, dst + i + 3 * stride);
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// variant 1: what MI355X_MICROARCH.md quotes 6.29 TB/s for - a plain float4 copy, one 16-byte element per thread, no loop
__global__ __launch_bounds__(256) void copy16_flat_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
// variant 2: each workgroup streams ONE contiguous 64 KiB span (16 x 16 bytes per lane, all loads issued before the first store)
__global__ __launch_bounds__(256) void copy16_span_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t base = (size_t)blockIdx.x * 4096 + threadIdx.x;
    u32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const size_t i = base + (size_t)u * 256;
        v[u] = i < n16 ? src[i] : (u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) dst[i] = v[u];
    }
}

constexpr int MFMA_PER_ROUND = 16;

__global__ __launch_bounds__(256) void mfma_peak_kernel(float* sink, int iters) {
    // operands are lane-dependent non-trivial values (an all-zero operand lets the clock run higher than real data does)
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (f16)(0.001f * (float)((threadIdx.x * 7 + i * 13) % 97) - 0.04f);
        b[i] = (f16)(0.002f * (float)((threadIdx.x * 11 + i * 5) % 89) - 0.08f);
    }
    f32x4 acc[MFMA_PER_ROUND];
#pragma unroll
    for (int j = 0; j < MFMA_PER_ROUND; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < MFMA_PER_ROUND; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < MFMA_PER_ROUND; ++j) t += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (t == 12345.678f) sink[0] = t;   // keeps the accumulators live; never true in practice
}

}  // namespace

extern "C" int vip_microbench_copy(const void* src, void* dst, size_t bytes, void* stream) {
    VIP_REQUIRE(src && dst && bytes >= 16 && bytes % 16 == 0, VIP_ERR_BAD_ARG, "vip_microbench_copy: bytes must be a positive multiple of 16");
    VIP_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, VIP_ERR_ALIGNMENT, "vip_microbench_copy: 16-byte aligned pointers");
    hipLaunchKernelGGL(copy16_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, bytes / 16);
    return vip_launch_status("vip_microbench_copy");
}

/* the other copy probes (bench.py takes the best of the three as the box's achievable HBM rate): 0 = vip_microbench_copy's grid-stride
 * non-temporal kernel, 1 = flat float4 copy (one element per thread: the form MI355X_MICROARCH.md's 6.29 TB/s is quoted for),
 * 2 = one contiguous 64 KiB span per workgroup */
extern "C" int vip_microbench_copy_variant(const void* src, void* dst, size_t bytes, int variant, void* stream) {
    if (variant == 0) return vip_microbench_copy(src, dst, bytes, stream);
    VIP_REQUIRE(src && dst && bytes >= 16 && bytes % 16 == 0, VIP_ERR_BAD_ARG, "vip_microbench_copy_variant: bytes must be a positive multiple of 16");
    VIP_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, VIP_ERR_ALIGNMENT, "vip_microbench_copy_variant: 16-byte aligned pointers");
    VIP_REQUIRE(variant == 1 || variant == 2, VIP_ERR_BAD_ARG, "vip_microbench_copy_variant: variant 0, 1 or 2");
    const size_t n16 = bytes / 16;
    if (variant == 1) {
        const size_t blocks = (n16 + 255) / 256;
        VIP_REQUIRE(blocks < (1ull << 31), VIP_ERR_UNSUPPORTED, "vip_microbench_copy_variant: buffer too large");
        hipLaunchKernelGGL(copy16_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, n16);
    } else {
        const size_t blocks = (n16 + 4095) / 4096;
        VIP_REQUIRE(blocks < (1ull << 31), VIP_ERR_UNSUPPORTED, "vip_microbench_copy_variant: buffer too large");
        hipLaunchKernelGGL(copy16_span_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, n16);
    }
    return vip_launch_status("vip_microbench_copy_variant");
}

extern "C" int vip_microbench_mfma_f16(void* sink, int iters, double* flops_h, void* stream) {
    VIP_REQUIRE(sink && iters > 0, VIP_ERR_BAD_ARG, "vip_microbench_mfma_f16: null sink or iters <= 0");
    const int blocks = 256 * 4;     // 256 CUs x 4 workgroups of 4 waves: 4 waves per SIMD
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)sink, iters);
    if (flops_h) *flops_h = (double)blocks * 4.0 * (double)iters * MFMA_PER_ROUND * (2.0 * 16 * 16 * 32);
    return vip_launch_status("vip_microbench_mfma_f16");
}

extern "C" size_t vip_workspace_bytes(int op, const int64_t* dims, int ndims) {
    // every operator works in place on its operands (tiles live in LDS / registers); the only scratch buffer in the ABI
    // is the JPEG component-plane buffer: one byte per coefficient
    if (op == VIP_OP_JPEG_IDCT_RGB && dims && ndims >= 1 && dims[0] > 0) return (size_t)dims[0];
    return 0;
}
