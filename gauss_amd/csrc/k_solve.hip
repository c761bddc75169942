// K5 blocked Cholesky (fp64), K6/K7 forward solve + imputation finalize.
//
// Replaces  MakePosDef + InvMat + per-SNP MpMatMat  of run_dist / run_distmix
// (dist.cpp:181-202, distmix.cpp:203-228, util.cpp:262-264,298-318):
//     B11 = L L^T                       (B11 already carries lambda on its diagonal)
//     v_u = L^-1 b21_u^T ,  y = L^-1 Z1
//     z_u = v_u . y   ( = b21_u B11^-1 Z1 )          dist.cpp:193-194
//     info_u = v_u . v_u ( = b21_u B11^-1 b12_u )    dist.cpp:197-198
//     out_z = z_u / sqrt(info_u), out_info = |info_u| dist.cpp:200-202
// In fp64 this agrees with the reference's full-pivot-LU inverse to ~1e-12 relative.
//
// MakePosDef (util.cpp:302-318) only acts when the smallest eigenvalue of B11 is below
// min_abs_eig.  That condition is tested exactly, on the GPU, by factoring the shifted matrix
// A[1] = B11 - min_abs_eig*I alongside A[0] = B11: the shifted factorisation succeeds iff every
// eigenvalue exceeds min_abs_eig (then MakePosDef is the identity map).  If it fails, status[1]
// is raised and the host driver reruns the window through the Jacobi eigen-clamp path.
#include "gauss_internal.h"

namespace gauss {

constexpr int LDB = NB + 1;   // padded LDS leading dimension (doubles)

// ---- 64x64 tile helpers; 256 threads; LDS tiles are [NB][LDB] doubles -----------------------

// C -= A * B^T   (all 64x64).  Thread t owns rows r0 = (t>>4)*4.. +3, cols c0 = (t&15)*4.. +3.
__device__ __forceinline__ void tile_gemm_nt_sub(double* __restrict__ C, const double* __restrict__ A,
                                                 const double* __restrict__ B, int tid)
{
    const int r0 = (tid >> 4) << 2, c0 = (tid & 15) << 2;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
    for (int k = 0; k < NB; k++) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = A[(r0 + i) * LDB + k];
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = B[(c0 + j) * LDB + k];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = fma(a[i], b[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) C[(r0 + i) * LDB + c0 + j] -= acc[i][j];
}

// C = A * B^T  (64x64), B lower/any.
__device__ __forceinline__ void tile_gemm_nt_set(double* __restrict__ C, const double* __restrict__ A,
                                                 const double* __restrict__ B, int tid)
{
    const int r0 = (tid >> 4) << 2, c0 = (tid & 15) << 2;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
    for (int k = 0; k < NB; k++) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = A[(r0 + i) * LDB + k];
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = B[(c0 + j) * LDB + k];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = fma(a[i], b[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) C[(r0 + i) * LDB + c0 + j] = acc[i][j];
}

__device__ __forceinline__ void tile_load(double* __restrict__ T, const double* __restrict__ g, int ld, int tid)
{
    for (int e = tid; e < NB * NB; e += 256) T[(e >> 6) * LDB + (e & 63)] = g[(size_t)(e >> 6) * ld + (e & 63)];
}
__device__ __forceinline__ void tile_store(double* __restrict__ g, int ld, const double* __restrict__ T, int tid)
{
    for (int e = tid; e < NB * NB; e += 256) g[(size_t)(e >> 6) * ld + (e & 63)] = T[(e >> 6) * LDB + (e & 63)];
}

// In-place Cholesky of the 64x64 LDS tile D (lower triangle result, upper zeroed), followed by
// the inverse of the factor into X.  Returns (block-uniform) 1 if a pivot was not positive.
// The factorisation is a dependency chain of 64 columns: it is run by ONE wave (lane = row), which
// needs no workgroup barriers -- LDS operations of a wave complete in order -- so a column costs
// one dot-product sweep instead of two barriers.  Left-looking:
//     s_i = a_ij - sum_{k<j} l_ik l_jk ;  l_jj = sqrt(s_j) ;  l_ij = s_i / l_jj
// then X = L^-1 by forward substitution, lane = column of X.
#define WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ int tile_chol_inv(double* __restrict__ D, double* __restrict__ X, int tid,
                             double* s_diag, int* s_flag)
{
    (void)s_diag;
    if (tid == 0) *s_flag = 0;
    __syncthreads();
    if (tid < NB) {
        const int i = tid;
        int bad = 0;
        for (int j = 0; j < NB; j++) {
            double s0 = D[i * LDB + j], s1 = 0.0;
            int k = 0;
            for (; k + 1 < j; k += 2) {
                s0 = fma(-D[i * LDB + k], D[j * LDB + k], s0);
                s1 = fma(-D[i * LDB + k + 1], D[j * LDB + k + 1], s1);
            }
            if (k < j) s0 = fma(-D[i * LDB + k], D[j * LDB + k], s0);
            const double s = s0 + s1;
            const double piv = __shfl(s, j);
            if (!(piv > 0.0)) bad = 1;
            const double d = sqrt(piv);
            const double l = (i == j) ? d : ((i > j) ? s / d : 0.0);
            WAVE_LDS_SYNC();                       // every lane has finished reading column/row data
            D[i * LDB + j] = l;
            WAVE_LDS_SYNC();                       // column j visible to the whole wave
        }
        // X = L^-1 : lane c owns column c
        const int c = tid;
        for (int r = 0; r < NB; r++) {
            double s0 = (r == c) ? 1.0 : 0.0, s1 = 0.0;
            int k = 0;
            for (; k + 1 < r; k += 2) {
                s0 = fma(-D[r * LDB + k], X[k * LDB + c], s0);
                s1 = fma(-D[r * LDB + k + 1], X[(k + 1) * LDB + c], s1);
            }
            if (k < r) s0 = fma(-D[r * LDB + k], X[k * LDB + c], s0);
            const double x = (r >= c) ? (s0 + s1) / D[r * LDB + r] : 0.0;
            X[r * LDB + c] = x;                    // only this lane ever reads column c of X
        }
        if (bad) *s_flag = 1;
    }
    __syncthreads();
    return *s_flag;
}

// ------------------------------------------------------------------------------------------
// K5: one step of the right-looking blocked Cholesky with one-column look-ahead.
// Launch `step` = s does, for every problem and both matrices:
//   * every block (bi,bj), s <= bj <= bi:   A_bibj -= L_{bi,s-1} L_{bj,s-1}^T      (s > 0)
//   * blocks of column s additionally finish their column: each recomputes the updated diagonal
//     block, factors it (redundantly -- no inter-workgroup traffic inside a launch), and
//     L_{bi,s} = A_{bi,s} Linv_ss^T.  L overwrites the lower triangle of A; the diagonal
//     block of column s is written by the (s,s) workgroup only, and nobody reads it in the same
//     launch from A: column workgroups read A_ss *before* ... see note below.
// Note on the in-launch hazard: workgroup (s,s) must not overwrite A_ss while workgroups
// (bi,s) still read it, so L blocks are written to the separate array Lm (row-major like A)
// and A is only ever updated in place by the block's own workgroup.
// grid.x enumerates (bi,bj) pairs of the remaining trailing matrix, grid.y = problem*2+matrix.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void factor_step_kernel(const Prob* __restrict__ probs, int step)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* T0 = smem;                 // the block being updated (kept until the end)
    double* T1 = T0 + NB * LDB;        // L_{bi,s-1}, then the diagonal block / L_ss, then the result
    double* T2 = T1 + NB * LDB;        // L_{bj,s-1}, then Linv_ss
    double* s_diag = T2 + NB * LDB;    // [NB]
    __shared__ int s_flag;

    const Prob& pb = probs[blockIdx.y >> 1];
    const int mat = blockIdx.y & 1;
    if (pb.ld_only || pb.npanel == 0) return;
    const int nb = pb.nblk;
    const int rem = nb - step;
    if (rem <= 0) return;
    // decode blockIdx.x -> (bi, bj) in the remaining lower triangle (row-major enumeration)
    const int t = blockIdx.x;
    if (t >= rem * (rem + 1) / 2) return;
    int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while (bi * (bi + 1) / 2 > t) bi--;
    while ((bi + 1) * (bi + 2) / 2 <= t) bi++;
    int bj = t - bi * (bi + 1) / 2;
    bi += step; bj += step;

    const int tid = threadIdx.x;
    const int ld = pb.Mld;
    double* A = pb.A + (size_t)mat * ld * ld;
    double* Lm = pb.A + (size_t)(2 + mat) * ld * ld;
    double* Ablk = A + (size_t)bi * NB * ld + (size_t)bj * NB;

    tile_load(T0, Ablk, ld, tid);
    if (step > 0) {
        tile_load(T1, Lm + (size_t)bi * NB * ld + (size_t)(step - 1) * NB, ld, tid);
        tile_load(T2, Lm + (size_t)bj * NB * ld + (size_t)(step - 1) * NB, ld, tid);
        __syncthreads();
        tile_gemm_nt_sub(T0, T1, T2, tid);
    }
    __syncthreads();
    if (bj != step) {                  // plain trailing block: write back and finish
        tile_store(Ablk, ld, T0, tid);
        return;
    }
    // column `step`: T1 <- updated diagonal block A_ss
    if (bi == step) {
        for (int e = tid; e < NB * LDB; e += 256) T1[e] = T0[e];
    } else {
        tile_load(T1, A + (size_t)step * NB * ld + (size_t)step * NB, ld, tid);
        if (step > 0) {
            __syncthreads();
            tile_gemm_nt_sub(T1, T2, T2, tid);      // T2 = L_{step,step-1} because bj == step
        }
    }
    __syncthreads();
    const int fail = tile_chol_inv(T1, T2, tid, s_diag, &s_flag);   // T1 = L_ss, T2 = Linv_ss
    if (bi == step) {
        tile_store(Lm + (size_t)step * NB * ld + (size_t)step * NB, ld, T1, tid);
        double* Li = pb.Linv + ((size_t)mat * nb + step) * NB * NB;
        for (int e = tid; e < NB * NB; e += 256) Li[e] = T2[(e >> 6) * LDB + (e & 63)];
        if (fail && tid == 0) pb.status[mat] = 1;
    } else {
        tile_gemm_nt_set(T1, T0, T2, tid);          // L_{bi,s} = A_{bi,s} * Linv_ss^T
        __syncthreads();
        tile_store(Lm + (size_t)bi * NB * ld + (size_t)step * NB, ld, T1, tid);
    }
}

static const size_t FACTOR_SMEM = ((size_t)3 * NB * LDB + NB) * sizeof(double);

void launch_factor_step(const Prob* d_probs, int n_prob, int step, int max_nblk, hipStream_t s)
{
    const int rem = max_nblk - step;
    if (rem <= 0 || n_prob <= 0) return;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(factor_step_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)FACTOR_SMEM);
        attr_set = true;
    }
    hipLaunchKernelGGL(factor_step_kernel, dim3(rem * (rem + 1) / 2, n_prob * 2), dim3(256), FACTOR_SMEM, s, d_probs, step);
}

// ------------------------------------------------------------------------------------------
// K6/K7: forward substitution for one panel of NR = 32 right-hand sides (31 unmeasured SNPs'
// b21 rows + the z1 column), left-looking over the NB-blocks of L, then z / info.
// One workgroup per panel; panels are independent (no inter-workgroup traffic).
//   V_k = Linv_kk * (B_k - sum_{j<k} L_kj V_j)
// V blocks are kept in the problem's V scratch ([panel][Mld][NR]) for reuse by later blocks.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void solve_kernel(const Prob* __restrict__ probs,
                                                    const int2* __restrict__ panelmap)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TL = smem;                       // [NB][LDB]  L_kj or Linv_kk
    double* TV = TL + NB * LDB;              // [NB][NR+1] V_j
    double* TX = TV + NB * (NR + 1);         // [NB][NR+1] running rhs / result
    double* red = TX + NB * (NR + 1);        // [2][256] reduction scratch
    constexpr int LV = NR + 1;

    const int2 pm = panelmap[blockIdx.x];
    const Prob& pb = probs[pm.x];
    const int panel = pm.y;
    const int tid = threadIdx.x;
    const int ld = pb.Mld, nb = pb.nblk;
    const double* Lm = pb.A + (size_t)2 * ld * ld;            // factor of A[0]
    const double* Linv = pb.Linv;                             // matrix 0
    double* V = pb.V + (size_t)panel * ld * NR;
    const int u0 = panel * NRU;

    // thread -> (row pair, col quad) of a 64 x 32 tile: rows r0, r0+1; cols c0..c0+3
    const int r0 = (tid >> 3) << 1, c0 = (tid & 7) << 2;
    // reduction ownership: column cc = tid & 31, row group rg = tid >> 5 (8 rows each)
    const int cc = tid & 31, rg = tid >> 5;
    double zsum = 0.0, isum = 0.0;

    for (int kb = 0; kb < nb; kb++) {
        // TX <- B block: rhs c < 31: B21[u0+c][kb*64 + r]; c == 31: z1 (zero padded)
        for (int e = tid; e < NB * NR; e += 256) {
            const int c = e >> 6, r = e & 63;          // r fastest: coalesced along a B21 row
            const int k = kb * NB + r;
            double v = 0.0;
            if (c < NRU) { const int u = u0 + c; if (u < pb.U) v = pb.B21[(size_t)u * ld + k]; }
            else if (k < pb.M) v = pb.z1[k];
            TX[r * LV + c] = v;
        }
        double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int jb = 0; jb < kb; jb++) {
            __syncthreads();
            tile_load(TL, Lm + (size_t)kb * NB * ld + (size_t)jb * NB, ld, tid);
            for (int e = tid; e < NB * NR; e += 256) TV[(e >> 5) * LV + (e & 31)] = V[(size_t)(jb * NB + (e >> 5)) * NR + (e & 31)];
            __syncthreads();
            for (int k = 0; k < NB; k++) {
                const double a0 = TL[r0 * LDB + k], a1 = TL[(r0 + 1) * LDB + k];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const double b = TV[k * LV + c0 + j];
                    acc[0][j] = fma(a0, b, acc[0][j]);
                    acc[1][j] = fma(a1, b, acc[1][j]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            TX[r0 * LV + c0 + j] -= acc[0][j];
            TX[(r0 + 1) * LV + c0 + j] -= acc[1][j];
        }
        // TL <- Linv_kk ; V_k = Linv_kk * TX
        {
            const double* Li = Linv + (size_t)kb * NB * NB;
            for (int e = tid; e < NB * NB; e += 256) TL[(e >> 6) * LDB + (e & 63)] = Li[e];
        }
        __syncthreads();
        double v[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int k = 0; k <= r0 + 1; k++) {
            const double a0 = TL[r0 * LDB + k], a1 = TL[(r0 + 1) * LDB + k];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const double b = TX[k * LV + c0 + j];
                v[0][j] = fma(a0, b, v[0][j]);
                v[1][j] = fma(a1, b, v[1][j]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            TV[r0 * LV + c0 + j] = v[0][j];
            TV[(r0 + 1) * LV + c0 + j] = v[1][j];
            V[(size_t)(kb * NB + r0) * NR + c0 + j] = v[0][j];
            V[(size_t)(kb * NB + r0 + 1) * NR + c0 + j] = v[1][j];
        }
        __syncthreads();
        // accumulate z and info for column cc over this block's rows rg*8 .. rg*8+7
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const double x = TV[(rg * 8 + r) * LV + cc];
            const double y = TV[(rg * 8 + r) * LV + NRU];
            zsum = fma(x, y, zsum);
            isum = fma(x, x, isum);
        }
    }
    __syncthreads();
    red[tid] = zsum;
    red[256 + tid] = isum;
    __syncthreads();
    if (tid < NRU) {
        double z = 0.0, info = 0.0;
        for (int g = 0; g < 8; g++) { z += red[g * 32 + tid]; info += red[256 + g * 32 + tid]; }
        const int u = u0 + tid;
        if (u < pb.U) {
            info = fabs(info);                         // dist.cpp:198
            pb.out_z[u] = z / sqrt(info);              // dist.cpp:200
            pb.out_info[u] = info;                     // dist.cpp:202
        }
    }
}

void launch_solve(const Prob* d_probs, const int2* d_panelmap, int n_panels, hipStream_t s)
{
    if (n_panels <= 0) return;
    const size_t sh = ((size_t)NB * LDB + 2 * (size_t)NB * (NR + 1) + 512) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(solve_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        attr_set = true;
    }
    hipLaunchKernelGGL(solve_kernel, dim3(n_panels), dim3(256), sh, s, d_probs, d_panelmap);
}

}  // namespace gauss
