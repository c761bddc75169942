// Microbenchmark: fp32 MFMA (32x32x2) issue rate with and without interleaved u8->f32 converts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ in, float* __restrict__ out, int iters)
{
    f32x16 a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0};
    unsigned w0 = in[threadIdx.x], w1 = in[threadIdx.x + 256], w2 = in[threadIdx.x + 512], w3 = in[threadIdx.x + 768];
    for (int it = 0; it < iters; it++) {
        if (VARIANT == 4) {
            // operands as f32 in LDS: 16 ds_read_b128 per 64 MFMAs, zero VALU
            __shared__ float lds[4096 + 64];
            if (it == 0) { for (int q = threadIdx.x; q < 4096; q += 256) lds[q] = (float)(q & 1); __syncthreads(); }
            const float* base = lds + (threadIdx.x & 63) * 36 + ((it & 3) << 2);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 fa0 = *reinterpret_cast<const float4*>(base + j * 8);
                const float4 fa1 = *reinterpret_cast<const float4*>(base + j * 8 + 1152);
                const float4 fb0 = *reinterpret_cast<const float4*>(base + j * 8 + 576);
                const float4 fb1 = *reinterpret_cast<const float4*>(base + j * 8 + 1728);
                const float a0[4] = {fa0.x, fa0.y, fa0.z, fa0.w}, a1[4] = {fa1.x, fa1.y, fa1.z, fa1.w};
                const float b0[4] = {fb0.x, fb0.y, fb0.z, fb0.w}, b1[4] = {fb1.x, fb1.y, fb1.z, fb1.w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    a00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], a00, 0, 0, 0);
                    a01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], a01, 0, 0, 0);
                    a10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], a10, 0, 0, 0);
                    a11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], a11, 0, 0, 0);
                }
            }
        } else if (VARIANT == 0) {
            float fa0 = __uint_as_float(w0), fa1 = __uint_as_float(w1), fb0 = __uint_as_float(w2), fb1 = __uint_as_float(w3);
#pragma unroll
            for (int t = 0; t < 16; t++) {
                a00 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, a00, 0, 0, 0);
                a01 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb1, a01, 0, 0, 0);
                a10 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb0, a10, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, a11, 0, 0, 0);
            }
        } else if (VARIANT == 3) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int q = 0; q < 4; q++) {
#define HALF(HW)                                                                                   \
    {                                                                                                  \
        const f32x2 fa0 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, HW), fa1 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, HW); \
        const f32x2 fb0 = __builtin_amdgcn_cvt_pk_f32_fp8(w2, HW), fb1 = __builtin_amdgcn_cvt_pk_f32_fp8(w3, HW); \
        _Pragma("unroll") for (int e = 0; e < 2; e++) {                                              \
            a00 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[e], fb0[e], a00, 0, 0, 0);                  \
            a01 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[e], fb1[e], a01, 0, 0, 0);                  \
            a10 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[e], fb0[e], a10, 0, 0, 0);                  \
            a11 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[e], fb1[e], a11, 0, 0, 0);                  \
        }                                                                                              \
    }
                HALF(false) HALF(true)
                w0 = w0 * 3 + 1; w1 = w1 * 5 + 1; w2 ^= w0; w3 ^= w1;
            }
        } else if (VARIANT == 5) {
            // operands as fp4 (e2m1) nibbles: v_cvt_scalef32_pk_f32_fp4 expands one byte = two values per instruction
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int q = 0; q < 2; q++) {
#define QUART(SEL)                                                                                     \
    {                                                                                                  \
        const f32x2 fa0 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w0, 1.0f, SEL), fa1 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w1, 1.0f, SEL); \
        const f32x2 fb0 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w2, 1.0f, SEL), fb1 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w3, 1.0f, SEL); \
        _Pragma("unroll") for (int e = 0; e < 2; e++) {                                              \
            a00 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[e], fb0[e], a00, 0, 0, 0);                  \
            a01 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[e], fb1[e], a01, 0, 0, 0);                  \
            a10 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[e], fb0[e], a10, 0, 0, 0);                  \
            a11 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[e], fb1[e], a11, 0, 0, 0);                  \
        }                                                                                              \
    }
                QUART(0) QUART(1) QUART(2) QUART(3)
                w0 = w0 * 3 + 1; w1 = w1 * 5 + 1; w2 ^= w0; w3 ^= w1;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    float fa0 = (float)((w0 >> (8 * b)) & 255), fa1 = (float)((w1 >> (8 * b)) & 255);
                    float fb0 = (float)((w2 >> (8 * b)) & 255), fb1 = (float)((w3 >> (8 * b)) & 255);
                    a00 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, a00, 0, 0, 0);
                    a01 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb1, a01, 0, 0, 0);
                    a10 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb0, a10, 0, 0, 0);
                    a11 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, a11, 0, 0, 0);
                    if (VARIANT == 2) __builtin_amdgcn_sched_barrier(0);
                }
                w0 = w0 * 3 + 1; w1 = w1 * 5 + 1; w2 ^= w0; w3 ^= w1;
            }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += a00[r] + a01[r] + a10[r] + a11[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    unsigned* in; float* out;
    hipMalloc(&in, 4096); hipMemset(in, 0x38, 4096);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 3; variant < 6; variant++)
        for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu++) {
            const int blocks = 256 * wg_per_cu;
            hipMalloc(&out, (size_t)blocks * 256 * 4);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                if (variant == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                if (variant == 4) hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                if (variant == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                if (variant == 5) hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                if (variant == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
            printf("variant %d  waves/SIMD %d  %.3f ms  %.1f TFLOP/s\n", variant, wg_per_cu, ms, flops / ms / 1e9);
            hipFree(out);
        }
    return 0;
}
