// Micro-probe: cycles of gauss::tile_chol_inv (64x64 fp64 Cholesky + inverse in LDS) on one workgroup.
//   hipcc -O3 --offload-arch=gfx950 -o gpurun_out/chol_probe tools/chol_probe.hip && gpurun_out/chol_probe
#include "../gauss_amd/csrc/k_solve.hip"
#include <cstdio>
#include <vector>
#include <cmath>

using namespace gauss;


// ---- experimental variants (probe only) ----
template <int MODE>
__device__ int chol_variant(double* __restrict__ D, double* __restrict__ X, int tid, int* s_flag)
{
    __shared__ double s_lc[NB];
    __shared__ double s_m[NB];
    const int lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *s_flag = 0;
    for (int e = tid; e < NB * NB; e += 256) X[(e >> 6) * LDT + (e & 63)] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0;
    __syncthreads();
    int bad = 0;
    if (MODE == 3) {
        // single wave, no workgroup barriers: lane = column, rows swept sequentially in batches of 16
        if (wave == 0) {
            const int c = lane;
            for (int j = 0; j < NB; j++) {
                const double piv = D[j * LDT + j];
                if (!(piv > 0.0)) bad = 1;
                const double a = D[c * LDT + j];
                const double t = X[j * LDT + c];
                double r = __builtin_amdgcn_rsq(piv);
                const double h = 0.5 * piv;
                r = fma(r, fma(-h * r, r, 0.5), r);
                r = fma(r, fma(-h * r, r, 0.5), r);
                double d = piv * r;
                d = fma(fma(-d, d, piv), 0.5 * r, d);
                const double l = (c == j) ? d : ((c > j) ? a * r : 0.0);
                const double x = (c <= j) ? t * r : 0.0;
                WAVE_LDS_SYNC();
                D[c * LDT + j] = l;
                X[j * LDT + c] = x;
                s_lc[c] = l;
                WAVE_LDS_SYNC();
                const double m = (c > j) ? l : x;
                double* const base = (c > j) ? D : X;
                for (int i0 = j + 1; i0 < NB; i0 += 16) {
                    double li[16], v[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const int i = (i0 + u < NB) ? i0 + u : NB - 1;
                        li[u] = s_lc[i];
                        v[u] = base[i * LDT + c];
                    }
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const int i = i0 + u;
                        if (i < NB && c <= i) base[i * LDT + c] = fma(-li[u], m, v[u]);
                    }
                }
                WAVE_LDS_SYNC();
            }
            if (bad) *s_flag = 1;
        }
        __syncthreads();
        return *s_flag;
    }
    for (int j = 0; j < NB; j++) {
        if (wave == 0 && MODE != 2) {
            const int c = lane;
            const double piv = D[j * LDT + j];
            if (!(piv > 0.0)) bad = 1;
            const double a = D[c * LDT + j];
            const double t = X[j * LDT + c];
            double r = __builtin_amdgcn_rsq(piv);
            const double h = 0.5 * piv;
            r = fma(r, fma(-h * r, r, 0.5), r);
            r = fma(r, fma(-h * r, r, 0.5), r);
            double d = piv * r;
            d = fma(fma(-d, d, piv), 0.5 * r, d);
            const double l = (c == j) ? d : ((c > j) ? a * r : 0.0);
            const double x = (c <= j) ? t * r : 0.0;
            D[c * LDT + j] = l;
            X[j * LDT + c] = x;
            s_lc[c] = l;
            s_m[c] = (c > j) ? l : x;
        }
        __syncthreads();
        if (MODE != 1) {
            const int c = lane;
            const double m = s_m[c];
            double* const base = (c > j) ? D : X;
            const int i0 = j + 1 + wave;
            double li[16], v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int i = (i0 + 4 * u < NB) ? i0 + 4 * u : NB - 1;
                li[u] = s_lc[i];
                v[u] = base[i * LDT + c];
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int i = i0 + 4 * u;
                if (i < NB && c <= i) base[i * LDT + c] = fma(-li[u], m, v[u]);
            }
        }
        __syncthreads();
    }
    if (bad) *s_flag = 1;
    __syncthreads();
    return *s_flag;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe_variant(const double* A, long long* cycles, int reps)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TD = smem;
    double* TX = TD + NB * LDT;
    __shared__ int s_flag;
    const int tid = threadIdx.x;
    long long best = 1LL << 60;
    for (int r = 0; r < reps; r++) {
        for (int e = tid; e < NB * NB; e += 256) TD[(e >> 6) * LDT + (e & 63)] = A[e];
        __syncthreads();
        const long long t0 = wall_clock64();
        chol_variant<MODE>(TD, TX, tid, &s_flag);
        const long long t1 = wall_clock64();
        if (t1 - t0 < best) best = t1 - t0;
        __syncthreads();
    }
    if (tid == 0) cycles[0] = best;
}

// variant 4: rows kept in registers (wave w owns rows w, w+4, ...; lane = column), one barrier per column
__device__ __forceinline__ double rl64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ int chol_regs(double* __restrict__ D, double* __restrict__ X, int tid, int* s_flag)
{
    __shared__ double s_m[2][NB];
    const int lane = tid & 63, wave = tid >> 6;
    const int c = lane;
    if (tid == 0) *s_flag = 0;
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = D[(wave + 4 * u) * LDT + c];
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 256) { D[(e >> 6) * LDT + (e & 63)] = 0.0; X[(e >> 6) * LDT + (e & 63)] = 0.0; }
    __syncthreads();
    int bad = 0;
    for (int j = 0; j < NB; j++) {
        const int uo = j >> 2;
        if (wave == (j & 3)) {
            double row = 0.0;
#pragma unroll
            for (int u = 0; u < 16; u++) row = (u == uo) ? v[u] : row;
            const double piv = rl64(row, j);
            if (!(piv > 0.0)) bad = 1;
            double r = __builtin_amdgcn_rsq(piv);
            const double h = 0.5 * piv;
            r = fma(r, fma(-h * r, r, 0.5), r);
            r = fma(r, fma(-h * r, r, 0.5), r);
            double d = piv * r;
            d = fma(fma(-d, d, piv), 0.5 * r, d);
            const double xr = (c == j) ? r : row * r;          // c < j: X[j][c]; c > j: L[c][j]; c == j: 1 / L[j][j]
            s_m[j & 1][c] = xr;
            if (c <= j) X[j * LDT + c] = xr;
            if (c == j) D[j * LDT + j] = d;
        }
        __syncthreads();
        const double m = s_m[j & 1][c];
        const double r = s_m[j & 1][j];
        const int ustart = (j >= wave) ? ((j - wave) >> 2) + 1 : 0;      // rows wave + 4u > j
        double lcol = 0.0;                                            // lane u collects L[wave + 4u][j]
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (u >= ustart) {
                const double li = rl64(v[u], j) * r;                    // L[i][j], uniform
                lcol = (lane == u) ? li : lcol;
                v[u] = fma(-li, m, (c == j) ? 0.0 : v[u]);
            }
        }
        if (lane >= ustart && lane < 16) D[(wave + 4 * lane) * LDT + j] = lcol;
    }
    if (bad) *s_flag = 1;
    __syncthreads();
    return *s_flag;
}

__global__ __launch_bounds__(256) void probe_regs(const double* A, double* L, double* Xo, long long* cycles, int reps)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TD = smem;
    double* TX = TD + NB * LDT;
    __shared__ int s_flag;
    const int tid = threadIdx.x;
    long long best = 1LL << 60;
    for (int r = 0; r < reps; r++) {
        for (int e = tid; e < NB * NB; e += 256) TD[(e >> 6) * LDT + (e & 63)] = A[e];
        __syncthreads();
        const long long t0 = wall_clock64();
        const int fail = chol_regs(TD, TX, tid, &s_flag);
        const long long t1 = wall_clock64();
        if (t1 - t0 < best) best = t1 - t0;
        if (fail && tid == 0) cycles[1] = 1;
        __syncthreads();
    }
    for (int e = tid; e < NB * NB; e += 256) { L[e] = TD[(e >> 6) * LDT + (e & 63)]; Xo[e] = TX[(e >> 6) * LDT + (e & 63)]; }
    if (tid == 0) cycles[0] = best;
}

template <bool BLK>
__global__ __launch_bounds__(256) void probe_kernel_t(const double* A, double* L, double* Xo, long long* cycles, int reps)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TD = smem;
    double* TX = TD + NB * LDT;
    __shared__ int s_flag;
    const int tid = threadIdx.x;
    long long best = 1LL << 60;
    for (int r = 0; r < reps; r++) {
        for (int e = tid; e < NB * NB; e += 256) TD[(e >> 6) * LDT + (e & 63)] = A[e];
        __syncthreads();
        const long long t0 = wall_clock64();
        const int fail = BLK ? tile_chol_inv_blk(TD, TX, tid, &s_flag) : tile_chol_inv(TD, TX, tid, &s_flag);
        const long long t1 = wall_clock64();
        if (t1 - t0 < best) best = t1 - t0;
        if (fail && tid == 0) cycles[1] = 1;
        __syncthreads();
    }
    for (int e = tid; e < NB * NB; e += 256) { L[e] = TD[(e >> 6) * LDT + (e & 63)]; Xo[e] = TX[(e >> 6) * LDT + (e & 63)]; }
    if (tid == 0) cycles[0] = best;
}

__global__ __launch_bounds__(256) void probe_kernel(const double* A, double* L, double* Xo, long long* cycles, int reps)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TD = smem;
    double* TX = TD + NB * LDT;
    __shared__ int s_flag;
    const int tid = threadIdx.x;
    long long best = 1LL << 60;
    for (int r = 0; r < reps; r++) {
        for (int e = tid; e < NB * NB; e += 256) TD[(e >> 6) * LDT + (e & 63)] = A[e];
        __syncthreads();
        const long long t0 = wall_clock64();
        const int fail = tile_chol_inv(TD, TX, tid, &s_flag);
        const long long t1 = wall_clock64();
        if (t1 - t0 < best) best = t1 - t0;
        if (fail && tid == 0) cycles[1] = 1;
        __syncthreads();
    }
    for (int e = tid; e < NB * NB; e += 256) { L[e] = TD[(e >> 6) * LDT + (e & 63)]; Xo[e] = TX[(e >> 6) * LDT + (e & 63)]; }
    if (tid == 0) cycles[0] = best;
}

int main()
{
    const int n = NB;
    std::vector<double> A(n * n), L(n * n), X(n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) A[i * n + j] = (i == j ? 1.1 : 0.0) + 0.5 * std::exp(-std::fabs(i - j) / 7.0);
    double *dA, *dL, *dX; long long* dc;
    hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dL, sizeof(double) * n * n); hipMalloc(&dX, sizeof(double) * n * n);
    hipMalloc(&dc, 16); hipMemset(dc, 0, 16);
    hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    const size_t sh = (size_t)2 * NB * LDT * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), sh, 0, dA, dL, dX, dc, 20);
    hipDeviceSynchronize();
    long long c[2];
    hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    hipMemcpy(L.data(), dL, sizeof(double) * n * n, hipMemcpyDeviceToHost);
    hipMemcpy(X.data(), dX, sizeof(double) * n * n, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;                       // |L L^T - A|, |X L - I|
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0, t = 0;
            for (int k = 0; k < n; k++) { s += L[i * n + k] * L[j * n + k]; t += X[i * n + k] * L[k * n + j]; }
            e1 = std::fmax(e1, std::fabs(s - A[i * n + j]));
            e2 = std::fmax(e2, std::fabs(t - (i == j ? 1.0 : 0.0)));
        }
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    printf("tile_chol_inv: %lld wall-clock ticks at %d kHz = %.2f us (min of 20), fail=%lld, |LL^T-A|=%.2e |XL-I|=%.2e\n",
           c[0], khz, khz ? c[0] * 1e3 / khz : 0.0, c[1], e1, e2);
    for (int blk = 0; blk < 2; blk++) {
        auto kern = blk ? probe_kernel_t<true> : probe_kernel_t<false>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipMemset(dc, 0, 16);
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), sh, 0, dA, dL, dX, dc, 20);
        hipDeviceSynchronize();
        hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        hipMemcpy(L.data(), dL, sizeof(double) * n * n, hipMemcpyDeviceToHost);
        hipMemcpy(X.data(), dX, sizeof(double) * n * n, hipMemcpyDeviceToHost);
        double f1 = 0, f2 = 0, up = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                double s2 = 0, t2 = 0;
                for (int k = 0; k < n; k++) { s2 += L[i * n + k] * L[j * n + k]; t2 += X[i * n + k] * L[k * n + j]; }
                f1 = std::fmax(f1, std::fabs(s2 - A[i * n + j]));
                f2 = std::fmax(f2, std::fabs(t2 - (i == j ? 1.0 : 0.0)));
                if (j > i) up = std::fmax(up, std::fmax(std::fabs(L[i * n + j]), std::fabs(X[i * n + j])));
            }
        printf("%s  %.2f us  fail=%lld |LL^T-A|=%.2e |XL-I|=%.2e upper=%.1e\n",
               blk ? "tile_chol_inv_blk (MFMA-blocked, product)" : "tile_chol_inv (rank-1 sweep, round 1)    ",
               khz ? c[0] * 1e3 / khz : 0.0, c[1], f1, f2, up);
    }
    auto run = [&](auto kern, const char* what) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipMemset(dc, 0, 16);
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), sh, 0, dA, dc, 20);
        hipDeviceSynchronize();
        hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        printf("  variant %-40s %.2f us\n", what, khz ? c[0] * 1e3 / khz : 0.0);
    };
    {
        hipFuncSetAttribute(reinterpret_cast<const void*>(probe_regs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipMemset(dc, 0, 16);
        hipLaunchKernelGGL(probe_regs, dim3(1), dim3(256), sh, 0, dA, dL, dX, dc, 20);
        hipDeviceSynchronize();
        hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        hipMemcpy(L.data(), dL, sizeof(double) * n * n, hipMemcpyDeviceToHost);
        hipMemcpy(X.data(), dX, sizeof(double) * n * n, hipMemcpyDeviceToHost);
        double f1 = 0, f2 = 0, up = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                double s2 = 0, t2 = 0;
                for (int k = 0; k < n; k++) { s2 += L[i * n + k] * L[j * n + k]; t2 += X[i * n + k] * L[k * n + j]; }
                f1 = std::fmax(f1, std::fabs(s2 - A[i * n + j]));
                f2 = std::fmax(f2, std::fabs(t2 - (i == j ? 1.0 : 0.0)));
                if (j > i) up = std::fmax(up, std::fmax(std::fabs(L[i * n + j]), std::fabs(X[i * n + j])));
            }
        printf("  variant 4: rows in registers               %.2f us  fail=%lld |LL^T-A|=%.2e |XL-I|=%.2e upper=%.1e\n",
               khz ? c[0] * 1e3 / khz : 0.0, c[1], f1, f2, up);
    }
    run(probe_variant<0>, "0: current copy");
    run(probe_variant<1>, "1: no phase 2 (barriers + pivots)");
    run(probe_variant<2>, "2: no phase 1 (barriers + sweeps)");
    run(probe_variant<3>, "3: single wave, no barriers");
    return 0;
}
