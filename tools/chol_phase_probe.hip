// Micro-probe: where the 64x64 tile routine (tile_chol_inv_blk, k_solve.hip) spends its time, phase by phase.
//   hipcc -O3 --offload-arch=gfx950 -o gpurun_out/chol_phase tools/chol_phase_probe.hip && gpurun_out/chol_phase
#include "../gauss_amd/csrc/k_solve.hip"
#include <cstdio>
#include <vector>
#include <cmath>

using namespace gauss;

#define STAMP(k) do { __syncthreads(); if (tid == 0) ts[k] = wall_clock64(); } while (0)

__device__ int chol_timed(double* __restrict__ D, double* __restrict__ X, int tid, int* s_flag, long long* ts)
{
    __shared__ double s_rinv[NB];
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    STAMP(0);
    if (tid == 0) *s_flag = 0;
    for (int e = tid; e < NB * NB; e += 256) X[(e >> 6) * LDT + (e & 63)] = 0.0;
    STAMP(1);
    int bad = 0;
    if (wave == 0) chol_block_column<0>(D, s_rinv, lane, bad);
    STAMP(2);
    chol_trailing<0>(D, wave, lane);
    STAMP(3);
    if (wave == 0) chol_block_column<1>(D, s_rinv, lane, bad);
    STAMP(4);
    chol_trailing<1>(D, wave, lane);
    STAMP(5);
    if (wave == 0) chol_block_column<2>(D, s_rinv, lane, bad);
    STAMP(6);
    chol_trailing<2>(D, wave, lane);
    STAMP(7);
    if (wave == 0) { chol_block_column<3>(D, s_rinv, lane, bad); if (bad && lane == 0) *s_flag = 1; }
    STAMP(8);
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        if ((c >> 4) > (r >> 4)) D[r * LDT + c] = 0.0;
    }
    {
        const int c = lane & 15, b = 16 * wave;
        double sacc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) sacc[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const double xj = sacc[j] * s_rinv[b + j];
            sacc[j] = xj;
#pragma unroll
            for (int i = j + 1; i < 16; i++) sacc[i] = fma(-D[(b + i) * LDT + b + j], xj, sacc[i]);
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; i++) X[(b + i) * LDT + b + c] = sacc[i];
        }
    }
    STAMP(9);
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int dist = 1; dist < 4; dist++) {
        const int i = dist + wave, j = wave;
        if (i < 4) {
            f64x4 s = f64x4{0.0, 0.0, 0.0, 0.0};
            for (int m = j; m < i; m++)
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4)
                    s = __builtin_amdgcn_mfma_f64_16x16x4f64(D[(16 * i + lr) * LDT + 16 * m + k0 + lk],
                                                             X[(16 * m + k0 + lk) * LDT + 16 * j + lr], s, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) X[(16 * i + lk + 4 * r) * LDT + 16 * j + lr] = s[r];
            WAVE_LDS_SYNC();
            f64x4 o = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += 4)
                o = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[(16 * i + lr) * LDT + 16 * i + k0 + lk],
                                                         X[(16 * i + k0 + lk) * LDT + 16 * j + lr], o, 0, 0, 0);
            WAVE_LDS_SYNC();
#pragma unroll
            for (int r = 0; r < 4; r++) X[(16 * i + lk + 4 * r) * LDT + 16 * j + lr] = o[r];
        }
        STAMP(9 + dist);
    }
    return *s_flag;
}

__global__ __launch_bounds__(256) void probe(const double* A, long long* out, int reps)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TD = smem;
    double* TX = TD + NB * LDT;
    __shared__ int s_flag;
    __shared__ long long ts[16];
    const int tid = threadIdx.x;
    for (int r = 0; r < reps; r++) {
        for (int e = tid; e < NB * NB; e += 256) TD[(e >> 6) * LDT + (e & 63)] = A[e];
        __syncthreads();
        chol_timed(TD, TX, tid, &s_flag, ts);
        __syncthreads();
    }
    if (tid < 13) out[tid] = ts[tid];
}

int main()
{
    const int n = NB;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) A[i * n + j] = (i == j ? 1.1 : 0.0) + 0.5 * std::exp(-std::fabs(i - j) / 7.0);
    double* dA; long long* dc;
    hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dc, 8 * 16);
    hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    const size_t sh = (size_t)2 * NB * LDT * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), sh, 0, dA, dc, 10);
    hipDeviceSynchronize();
    long long c[16];
    hipMemcpy(c, dc, 8 * 13, hipMemcpyDeviceToHost);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    const char* names[12] = {"clear X", "block column 0", "trailing 0", "block column 1", "trailing 1", "block column 2",
                             "trailing 2", "block column 3", "clear upper + diagonal inverses", "off-diagonal dist 1",
                             "off-diagonal dist 2", "off-diagonal dist 3"};
    for (int k = 0; k < 12; k++) printf("%-34s %6.2f us\n", names[k], (c[k + 1] - c[k]) * 1e3 / khz);
    printf("%-34s %6.2f us\n", "total", (c[12] - c[0]) * 1e3 / khz);
    return 0;
}
