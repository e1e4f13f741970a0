// Squeeze-excite gate in ONE launch:  gate[b, :] = act2( W2 . act1( W1 . mean_hw(x[b]) + b1 ) + b2 )
//
// Replaces GlobalAveragePooling2D -> Conv1x1/Dense(+act) -> Conv1x1/Dense(+sigmoid) of kecam `se_module`
// (common_layers.py:311-332), ResNet-RS `SE` (resnet_rs_model.py:145-183), GCViT `SE` (gcvit/layers/feature.py:46-70) and
// the ResNeSt split-attention gate (resnest.py:44-57).  As three launches (pool, Dense with M = batch, Dense with M =
// batch) the chain is pure launch latency: the two GEMMs have 256 rows and run on a handful of workgroups while the
// rest of the chip idles, 50-120 us per block for ~30 us of memory time.  Here one workgroup per image pools its
// feature map (fp32 partial sums through LDS), then runs the two tiny matrix-vector products out of L2 with the pooled
// vector in LDS.  The pooled and hidden vectors stay fp32.  The gate is the one value of a squeeze-excite block whose
// rounding error is COHERENT over a whole channel map (it survives every later spatial average instead of shrinking
// with sqrt(pixels)): with fp16 gates EfficientNet-B4's logit error is 1.0e-2 rms, of which 0.8e-2 is this rounding.
// With `split` the gate is therefore written as two fp16 planes, hi = fp16(g) and lo = fp16(g - hi), [B][2][Cout];
// consumers apply x * hi + x * lo.
#include "common.hpp"

namespace {

struct SeArgs {
    const f16* x;
    const f16* w1;
    const float* b1;
    const f16* w2;
    const float* b2;
    f16* gate;
    int HW, C, ldx, Cr, ldw1, Co, ldw2, ldg;
    int act1, act2, split;
    const float* part;   // pooled form: [B][parts][C] fp32 partial sums of x over pixels (vip_dwconv2d_pool_nhwc_f16), x unused
    int parts;
    float s1, s2;        // H2S: 1 / (power-of-two scale folded into w1 / w2 and their biases)
    int* status;         // H2S: raised when a gate value leaves the fp16 range
};

// H2S: the packed STRICT storage (common.hpp) - x [B][HW][C] and the gate [B][Co] are (hi, lo) fp16 pairs (8 channels = 32 bytes),
// the weight rows are packed pairs of W * scale (ops.split_h2_weights): every value is joined to fp32 once, the arithmetic is fp32 and
// the activations are the strict ones (vip_act_strict).

constexpr int SE_THREADS = 512;

template <bool H2S>
__global__ __launch_bounds__(SE_THREADS) void se_gate_kernel(SeArgs a) {
    constexpr int ES = H2S ? 2 : 1;                      // halfs per element
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C8 = a.C >> 3;
    const int cw = C8 < SE_THREADS ? C8 : SE_THREADS;    // channel-chunk lanes
    const int G = a.part ? 1 : SE_THREADS / cw;          // pixel groups (pooled form: the sums arrive as one row)
    float* part = sm;                                    // [G][C]
    float* mean = sm + G * a.C;                          // [C]
    float* hid = mean + a.C;                             // [Cr]
    const f16* xb = a.x + (long)blockIdx.x * a.HW * a.ldx * ES;
    // 8 consecutive elements starting at element index e (e % 8 == 0) of a row-major fp16 / packed array -> fp32
    auto ld8 = [](const f16* base, long e, float (&out)[8]) {
        U4H8 h;
        h.u = *reinterpret_cast<const uint4*>(base + e * ES);
        if constexpr (H2S) {
            U4H8 l;
            l.u = *reinterpret_cast<const uint4*>(base + e * ES + 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = (float)h.e[j] + (float)l.e[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = (float)h.e[j];
        }
    };

    // ---- 1. pool: thread = (channel chunk cl (+ k*cw), pixel group pg); 16-byte loads, fp32 sums
    const int cl = tid % cw, pg = tid / cw;
    if (a.part) {
        const float* pb = a.part + (long)blockIdx.x * a.parts * a.C;
        for (int c = tid; c < a.C; c += SE_THREADS) {
            float s = 0.f;
            for (int g = 0; g < a.parts; ++g) s += pb[(long)g * a.C + c];
            part[c] = s;                                     // G = 1 row of the reduction below
        }
    } else if (pg < G) {
        for (int c8 = cl; c8 < C8; c8 += cw) {
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int p = pg;
            // eight independent 16-byte loads in flight per lane: one workgroup per CU has to cover the memory latency
            // by itself (512 lanes x 8 x 16 B = 64 KB outstanding)
            if constexpr (H2S) {
                for (; p + 3 * G < a.HW; p += 4 * G) {       // four (hi, lo) pairs = eight 16-byte loads in flight per lane
                    float v[4][8];
#pragma unroll
                    for (int u = 0; u < 4; ++u) ld8(xb, (long)(p + u * G) * a.ldx + c8 * 8, v[u]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) s[j] += (v[0][j] + v[1][j]) + (v[2][j] + v[3][j]);
                }
            } else {
            for (; p + 7 * G < a.HW; p += 8 * G) {
                U4H8 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u].u = *reinterpret_cast<const uint4*>(xb + (long)(p + u * G) * a.ldx + c8 * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    s[j] += (((float)v[0].e[j] + (float)v[1].e[j]) + ((float)v[2].e[j] + (float)v[3].e[j])) +
                            (((float)v[4].e[j] + (float)v[5].e[j]) + ((float)v[6].e[j] + (float)v[7].e[j]));
            }
            }
            for (; p < a.HW; p += G) {
                float v[8];
                ld8(xb, (long)p * a.ldx + c8 * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += v[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) part[pg * a.C + c8 * 8 + j] = s[j];
        }
    }
    __syncthreads();
    const float inv = 1.f / (float)a.HW;
    for (int c = tid; c < a.C; c += SE_THREADS) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += part[g * a.C + c];
        mean[c] = s * inv;
    }
    __syncthreads();

    // ---- 2. hidden = act1(W1 . mean + b1): one output per wave at a time, lanes split the C axis in 16-byte chunks
    // (four outputs per wave in flight: a single row at a time is one dependent L2 round trip after another)
    for (int r0 = wave * 4; r0 < a.Cr; r0 += SE_THREADS / 64 * 4) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c8 = lane; c8 < C8; c8 += 64) {
            float w[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + u < a.Cr ? r0 + u : a.Cr - 1;
                ld8(a.w1 + (long)r * a.ldw1, c8 * 8, w[u]);               // (ldw in halfs in both storages)
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float m = mean[c8 * 8 + j];
#pragma unroll
                for (int u = 0; u < 4; ++u) s[u] += w[u][j] * m;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float t = wave_reduce_sum(s[u]);
            if (lane == 0 && r0 + u < a.Cr) {
                const float pre = t + (a.b1 ? a.b1[r0 + u] : 0.f);
                hid[r0 + u] = H2S ? vip_act_strict(pre * a.s1, a.act1) : vip_act(pre, a.act1);
            }
        }
    }
    __syncthreads();

    // ---- 3. gate = act2(W2 . hidden + b2): one output channel per thread (rows of W2 are Cr halfs, contiguous)
    const int R8 = a.Cr >> 3;
    for (int c = tid; c < a.Co; c += SE_THREADS) {
        float s = a.b2 ? a.b2[c] : 0.f;
#pragma unroll 4
        for (int r8 = 0; r8 < R8; ++r8) {
            float w[8];
            ld8(a.w2 + (long)c * a.ldw2, r8 * 8, w);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += w[j] * hid[r8 * 8 + j];
        }
        if constexpr (H2S) {                                 // packed [B][Co]: element c = group c / 8, slot c % 8
            const float g = vip_act_strict(s * a.s2, a.act2);
            const f16 hi = (f16)g;
            f16* grp = a.gate + ((long)blockIdx.x * a.Co + (c & ~7)) * 2 + (c & 7);
            grp[0] = hi;
            grp[8] = (f16)(g - (float)hi);
            if (a.status && !(fabsf(g) <= VIP_H2_MAX)) *a.status = VIP_H2_OVERFLOW;
            continue;
        }
        const float g = vip_act(s, a.act2);
        const f16 hi = (f16)g;
        a.gate[(long)blockIdx.x * a.ldg + c] = hi;
        if (a.split) a.gate[(long)blockIdx.x * a.ldg + a.Co + c] = (f16)(g - (float)hi);
    }
}

}  // namespace

extern "C" int vip_se_gate_f16(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* gate,
                               int B, int HW, int C, int ldx, int Cr, int ldw1, int Cout, int ldw2, int act1, int act2,
                               int split, void* stream) {
    VIP_REQUIRE(x && w1 && w2 && gate, VIP_ERR_BAD_ARG, "vip_se_gate_f16: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && Cr > 0 && Cout > 0, VIP_ERR_BAD_ARG, "vip_se_gate_f16: non-positive dimension");
    VIP_REQUIRE((unsigned)act1 <= 4u && (unsigned)act2 <= 4u, VIP_ERR_BAD_ARG, "vip_se_gate_f16: unknown activation code");
    VIP_REQUIRE(C % 8 == 0 && Cr % 8 == 0 && ldx % 8 == 0 && ldw1 % 8 == 0 && ldw2 % 8 == 0, VIP_ERR_ALIGNMENT,
                "vip_se_gate_f16: C, Cr and the leading dimensions must be multiples of 8 halfs");
    VIP_REQUIRE(ldx >= C && ldw1 >= C && ldw2 >= Cr, VIP_ERR_BAD_ARG, "vip_se_gate_f16: leading dimension too small");
    const int C8 = C / 8, cw = C8 < SE_THREADS ? C8 : SE_THREADS, G = SE_THREADS / cw;
    const size_t smem = ((size_t)G * C + C + Cr) * sizeof(float);
    VIP_REQUIRE(smem <= 64 * 1024, VIP_ERR_UNSUPPORTED, "vip_se_gate_f16: C=%d too wide", C);
    SeArgs a;
    a.x = (const f16*)x; a.w1 = (const f16*)w1; a.b1 = b1; a.w2 = (const f16*)w2; a.b2 = b2; a.gate = (f16*)gate;
    a.HW = HW; a.C = C; a.ldx = ldx; a.Cr = Cr; a.ldw1 = ldw1; a.Co = Cout; a.ldw2 = ldw2; a.ldg = split ? 2 * Cout : Cout;
    a.act1 = act1; a.act2 = act2; a.split = split ? 1 : 0;
    a.part = nullptr; a.parts = 0;
    a.s1 = a.s2 = 1.f; a.status = nullptr;
    hipLaunchKernelGGL(se_gate_kernel<false>, dim3(B), dim3(SE_THREADS), smem, (hipStream_t)stream, a);
    return vip_launch_status("vip_se_gate_f16");
}

/* The same chain on the packed STRICT storage: x [B][HW][ldx] and gate [B][Cout] packed (hi, lo) pairs (counts in ELEMENTS), w1 / w2
 * packed rows of W * scale (ldw1 / ldw2 in halfs, >= 2 C / 2 Cr), b1 / b2 = bias * scale, s1 / s2 = 1 / scale. */
extern "C" int vip_se_gate_h2(const void* x, const void* w1, const float* b1, float s1, const void* w2, const float* b2, float s2,
                              void* gate, int B, int HW, int C, int ldx, int Cr, int ldw1, int Cout, int ldw2, int act1, int act2,
                              int* status, void* stream) {
    VIP_REQUIRE(x && w1 && w2 && gate, VIP_ERR_BAD_ARG, "vip_se_gate_h2: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && Cr > 0 && Cout > 0 && s1 > 0.f && s2 > 0.f, VIP_ERR_BAD_ARG, "vip_se_gate_h2: non-positive dimension or scale");
    VIP_REQUIRE((unsigned)act1 <= 4u && (unsigned)act2 <= 4u, VIP_ERR_BAD_ARG, "vip_se_gate_h2: unknown activation code");
    VIP_REQUIRE(C % 8 == 0 && Cr % 8 == 0 && Cout % 8 == 0 && ldx % 8 == 0 && ldw1 % 16 == 0 && ldw2 % 16 == 0, VIP_ERR_ALIGNMENT,
                "vip_se_gate_h2: C, Cr, Cout, ldx must be multiples of 8 elements, ldw1 / ldw2 of 16 halfs");
    VIP_REQUIRE(ldx >= C && ldw1 >= 2 * C && ldw2 >= 2 * Cr, VIP_ERR_BAD_ARG, "vip_se_gate_h2: leading dimension too small");
    const int C8 = C / 8, cw = C8 < SE_THREADS ? C8 : SE_THREADS, G = SE_THREADS / cw;
    const size_t smem = ((size_t)G * C + C + Cr) * sizeof(float);
    VIP_REQUIRE(smem <= 64 * 1024, VIP_ERR_UNSUPPORTED, "vip_se_gate_h2: C=%d too wide", C);
    SeArgs a;
    a.x = (const f16*)x; a.w1 = (const f16*)w1; a.b1 = b1; a.w2 = (const f16*)w2; a.b2 = b2; a.gate = (f16*)gate;
    a.HW = HW; a.C = C; a.ldx = ldx; a.Cr = Cr; a.ldw1 = ldw1; a.Co = Cout; a.ldw2 = ldw2; a.ldg = Cout;
    a.act1 = act1; a.act2 = act2; a.split = 0;
    a.part = nullptr; a.parts = 0;
    a.s1 = s1; a.s2 = s2; a.status = status;
    hipLaunchKernelGGL(se_gate_kernel<true>, dim3(B), dim3(SE_THREADS), smem, (hipStream_t)stream, a);
    return vip_launch_status("vip_se_gate_h2");
}

/* vip_se_gate_h2 from pooled partial sums (vip_dwconv2d_s1_pool_h2): partials [B][parts][C] fp32, gate packed [B][Cout]. */
extern "C" int vip_se_gate_pooled_h2(const float* partials, int parts, const void* w1, const float* b1, float s1, const void* w2, const float* b2,
                                     float s2, void* gate, int B, int HW, int C, int Cr, int ldw1, int Cout, int ldw2, int act1, int act2,
                                     int* status, void* stream) {
    VIP_REQUIRE(partials && w1 && w2 && gate, VIP_ERR_BAD_ARG, "vip_se_gate_pooled_h2: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && Cr > 0 && Cout > 0 && parts > 0 && s1 > 0.f && s2 > 0.f, VIP_ERR_BAD_ARG,
                "vip_se_gate_pooled_h2: non-positive dimension or scale");
    VIP_REQUIRE((unsigned)act1 <= 4u && (unsigned)act2 <= 4u, VIP_ERR_BAD_ARG, "vip_se_gate_pooled_h2: unknown activation code");
    VIP_REQUIRE(C % 8 == 0 && Cr % 8 == 0 && Cout % 8 == 0 && ldw1 % 16 == 0 && ldw2 % 16 == 0, VIP_ERR_ALIGNMENT,
                "vip_se_gate_pooled_h2: C, Cr, Cout must be multiples of 8 elements, ldw1 / ldw2 of 16 halfs");
    VIP_REQUIRE(ldw1 >= 2 * C && ldw2 >= 2 * Cr, VIP_ERR_BAD_ARG, "vip_se_gate_pooled_h2: leading dimension too small");
    const size_t smem = ((size_t)C + C + Cr) * sizeof(float);
    VIP_REQUIRE(smem <= 64 * 1024, VIP_ERR_UNSUPPORTED, "vip_se_gate_pooled_h2: C=%d too wide", C);
    SeArgs a;
    a.x = nullptr; a.w1 = (const f16*)w1; a.b1 = b1; a.w2 = (const f16*)w2; a.b2 = b2; a.gate = (f16*)gate;
    a.HW = HW; a.C = C; a.ldx = C; a.Cr = Cr; a.ldw1 = ldw1; a.Co = Cout; a.ldw2 = ldw2; a.ldg = Cout;
    a.act1 = act1; a.act2 = act2; a.split = 0;
    a.part = partials; a.parts = parts;
    a.s1 = s1; a.s2 = s2; a.status = status;
    hipLaunchKernelGGL(se_gate_kernel<true>, dim3(B), dim3(SE_THREADS), smem, (hipStream_t)stream, a);
    return vip_launch_status("vip_se_gate_pooled_h2");
}

extern "C" int vip_se_gate_pooled_f16(const float* partials, int parts, const void* w1, const float* b1, const void* w2,
                                      const float* b2, void* gate, int B, int HW, int C, int Cr, int ldw1, int Cout, int ldw2,
                                      int act1, int act2, int split, void* stream) {
    VIP_REQUIRE(partials && w1 && w2 && gate, VIP_ERR_BAD_ARG, "vip_se_gate_pooled_f16: null pointer");
    VIP_REQUIRE(B > 0 && HW > 0 && C > 0 && Cr > 0 && Cout > 0 && parts > 0, VIP_ERR_BAD_ARG,
                "vip_se_gate_pooled_f16: non-positive dimension");
    VIP_REQUIRE((unsigned)act1 <= 4u && (unsigned)act2 <= 4u, VIP_ERR_BAD_ARG, "vip_se_gate_pooled_f16: unknown activation code");
    VIP_REQUIRE(C % 8 == 0 && Cr % 8 == 0 && ldw1 % 8 == 0 && ldw2 % 8 == 0, VIP_ERR_ALIGNMENT,
                "vip_se_gate_pooled_f16: C, Cr and the leading dimensions must be multiples of 8 halfs");
    VIP_REQUIRE(ldw1 >= C && ldw2 >= Cr, VIP_ERR_BAD_ARG, "vip_se_gate_pooled_f16: leading dimension too small");
    const size_t smem = ((size_t)C + C + Cr) * sizeof(float);
    VIP_REQUIRE(smem <= 64 * 1024, VIP_ERR_UNSUPPORTED, "vip_se_gate_pooled_f16: C=%d too wide", C);
    SeArgs a;
    a.x = nullptr; a.w1 = (const f16*)w1; a.b1 = b1; a.w2 = (const f16*)w2; a.b2 = b2; a.gate = (f16*)gate;
    a.HW = HW; a.C = C; a.ldx = C; a.Cr = Cr; a.ldw1 = ldw1; a.Co = Cout; a.ldw2 = ldw2; a.ldg = split ? 2 * Cout : Cout;
    a.act1 = act1; a.act2 = act2; a.split = split ? 1 : 0;
    a.part = partials; a.parts = parts;
    a.s1 = a.s2 = 1.f; a.status = nullptr;
    hipLaunchKernelGGL(se_gate_kernel<false>, dim3(B), dim3(SE_THREADS), smem, (hipStream_t)stream, a);
    return vip_launch_status("vip_se_gate_pooled_f16");
}
