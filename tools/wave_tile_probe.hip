// Microbenchmark (round 4): how much of the fp32 matrix pipe do the operand conversions cost at LARGER wave tiles?
// gram_kernel's waves own 64 x 64 (four 32 x 32 accumulators): per K pair-step 4 v_cvt_pk_f32_fp8 feed 8 MFMAs (0.5 conversions per
// MFMA; tools/cvt_probe.hip: 0.905 of the pipe).  A 128 x 64 wave tile (eight accumulators, 128 accumulator registers) needs 6
// conversions per 16 MFMAs (0.375), 128 x 128 (sixteen accumulators, 256 registers: one wave per SIMD) 8 per 32 (0.25).
// No LDS, no loads: registers only -- the ceiling each shape could reach, at the occupancy its registers allow.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, C, 0, 0, 0)
#define CVT(W, HI) ((HI) ? __builtin_amdgcn_cvt_pk_f32_fp8((int)(W), true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)(W), false))

template <int NA, int NB, int OCC>
__global__ __launch_bounds__(256, OCC) void probe(const unsigned* __restrict__ in, float* __restrict__ out, int iters)
{
    f32x16 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; i++)
#pragma unroll
        for (int j = 0; j < NB; j++) acc[i][j] = f32x16{0};
    unsigned wa[NA], wb[NB];
#pragma unroll
    for (int i = 0; i < NA; i++) wa[i] = in[threadIdx.x + 64 * i];
#pragma unroll
    for (int j = 0; j < NB; j++) wb[j] = in[threadIdx.x + 64 * (NA + j) + 7];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
#pragma unroll
            for (int hw = 0; hw < 2; hw++) {
                f32x2 fa[NA], fb[NB];
#pragma unroll
                for (int i = 0; i < NA; i++) fa[i] = CVT(wa[i], hw);
#pragma unroll
                for (int j = 0; j < NB; j++) fb[j] = CVT(wb[j], hw);
#pragma unroll
                for (int e = 0; e < 2; e++)
#pragma unroll
                    for (int i = 0; i < NA; i++)
#pragma unroll
                        for (int j = 0; j < NB; j++) MFMA(fa[i][e], fb[j][e], acc[i][j]);
            }
#pragma unroll
            for (int i = 0; i < NA; i++) wa[i] = wa[i] * 3 + 1;
#pragma unroll
            for (int j = 0; j < NB; j++) wb[j] ^= wa[0];
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NA; i++)
#pragma unroll
        for (int j = 0; j < NB; j++)
            for (int r = 0; r < 16; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NA, int NB, int OCC>
static void run(const unsigned* in, int wg_per_cu)
{
    const int iters = 1000 * 4 / (NA * NB);
    const int blocks = 256 * wg_per_cu;
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NA, NB, OCC>), dim3(blocks), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double mfmas = (double)blocks * 4 * iters * 8 * 2 * 2 * NA * NB;       // per wave: 8 q x 2 halves x 2 e x NA x NB
    const double tf = mfmas * 4096.0 / best / 1e9;
    printf("wave tile %3d x %3d  conversions per MFMA %.3f  waves/SIMD %d  %.3f ms  %.1f TFLOP/s = %.3f of 157.3\n", 32 * NA, 32 * NB,
           (double)(NA + NB) / (2.0 * NA * NB), wg_per_cu, best, tf, tf / 157.3);
    hipFree(out);
}

int main()
{
    unsigned* in; hipMalloc(&in, 16384);
    { unsigned h[4096]; for (int i = 0; i < 4096; i++) h[i] = 0x38400038u ^ (i * 2654435761u & 0x00404000u); hipMemcpy(in, h, 16384, hipMemcpyHostToDevice); }
    for (int w = 1; w <= 4; w *= 2) run<2, 2, 4>(in, w);
    for (int w = 1; w <= 2; w *= 2) run<4, 2, 2>(in, w);
    for (int w = 1; w <= 2; w *= 2) run<2, 4, 2>(in, w);
    run<4, 4, 1>(in, 1);
    run<1, 1, 4>(in, 4);
    run<2, 1, 4>(in, 4);
    return 0;
}
