// K5 blocked Cholesky (fp64), K6/K7 inverse factor + imputation product -- fp64 matrix cores.
//
// Replaces  MakePosDef + InvMat + per-SNP MpMatMat  of run_dist / run_distmix
// (dist.cpp:181-202, distmix.cpp:203-228, util.cpp:262-264,298-318):
//     B11 = L L^T                       (B11 already carries lambda on its diagonal)
//     [X | y] = L^-1 [I | Z1]           (rows ride in the factorisation's launches)
//     w_u = X b21_u^T                   (one product W = B21 X^T, impute_gemm_kernel)
//     z_u = w_u . y   ( = b21_u B11^-1 Z1 )          dist.cpp:193-194
//     info_u = w_u . w_u ( = b21_u B11^-1 b12_u )    dist.cpp:197-198
//     out_z = z_u / sqrt(info_u), out_info = |info_u| dist.cpp:200-202
// In fp64 this agrees with the reference's full-pivot-LU inverse to ~1e-13 relative.  (The stand-alone solve_kernel
// of the clamp path substitutes the window's own right-hand sides instead: v_u = L^-1 b21_u^T.)
//
// MakePosDef (util.cpp:302-318) only acts when the smallest eigenvalue of B11 is below
// min_abs_eig.  That condition is tested exactly, on the GPU, by factoring the shifted matrix
// A[1] = B11 - min_abs_eig*I alongside A[0] = B11: the shifted factorisation succeeds iff every
// eigenvalue exceeds min_abs_eig (then MakePosDef is the identity map).  If it fails, status[1]
// is raised and the host driver reruns the window through the Jacobi eigen-clamp path.
//
// All block products run on v_mfma_f64_16x16x4_f64 (one f64 A and one f64 B value per lane; result
// rows (lane>>4) + 4*reg, column lane&15).  A workgroup is 4 waves; wave w owns rows 16w..16w+15
// of a 64-row block.
//
// The factorisation kernels of this file have small-footprint twins in k_solve_lite.hip (same arithmetic, same bits,
// 23 KB of LDS and <= 96 registers) that run BESIDE the Gram kernel on jobs large enough to hide them; what the two
// families share lives in k_solve_common.h.  These are the forms with the chip to themselves: single windows, the clamp
// path's redo, GAUSS_CHAIN_ASIDE=0.
#include "gauss_internal.h"
#include "k_solve_common.h"
#include <algorithm>
#include <cstdlib>

namespace gauss {

// acc[n] += sign * A(rows 16w.., K=64) * B^T, A and B both [row][k] tiles with leading dimension LDT.
template <int NT, bool NEG>
__device__ __forceinline__ void mfma_nt(f64x4 (&acc)[NT], const double* __restrict__ A, const double* __restrict__ B,
                                        int wave, int lane)
{
    const double* ap = A + (16 * wave + (lane & 15)) * LDT + (lane >> 4);
    const double* bp = B + (lane & 15) * LDT + (lane >> 4);
#pragma unroll 8
    for (int k0 = 0; k0 < NB; k0 += 4) {
        double a = ap[k0];
        if (NEG) a = -a;
#pragma unroll
        for (int n = 0; n < NT; n++) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[n * 16 * LDT + k0], acc[n], 0, 0, 0);
    }
}

// acc[n] += sign * A(rows 16w.., K=64, [row][k], LDT) * V ([k][col], leading dimension LDV)
template <int NT, bool NEG>
__device__ __forceinline__ void mfma_nn(f64x4 (&acc)[NT], const double* __restrict__ A, const double* __restrict__ V,
                                        int wave, int lane)
{
    const double* ap = A + (16 * wave + (lane & 15)) * LDT + (lane >> 4);
    const double* vp = V + (lane >> 4) * LDV + (lane & 15);
#pragma unroll 8
    for (int k0 = 0; k0 < NB; k0 += 4) {
        double a = ap[k0];
        if (NEG) a = -a;
#pragma unroll
        for (int n = 0; n < NT; n++) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, vp[k0 * LDV + n * 16], acc[n], 0, 0, 0);
    }
}

template <typename P>
__device__ __forceinline__ void tile_load(double* __restrict__ T, P g, int ld, int tid)
{
    for (int e = tid; e < NB * NB; e += 256) T[(e >> 6) * LDT + (e & 63)] = g[(size_t)(e >> 6) * ld + (e & 63)];
}
template <typename P>
__device__ __forceinline__ void tile_store(P g, int ld, const double* __restrict__ T, int tid)
{
    for (int e = tid; e < NB * NB; e += 256) g[(size_t)(e >> 6) * ld + (e & 63)] = T[(e >> 6) * LDT + (e & 63)];
}

// Register-staged tile transfer: fetch() issues the global loads of a 64 x 64 tile (8 x 16 bytes per
// thread) and returns at once; commit() stores them into an LDS image later, so the loads fly while
// the matrix cores work on the previous tile.
struct TileRegs { f64x2 v[8]; };

template <typename P>
__device__ __forceinline__ void tile_fetch(TileRegs& t, P g, int ld, int tid)
{
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int e = tid + 256 * q;                 // 2048 pairs: row e / 32, columns 2 (e % 32)
        const auto p = g + (size_t)(e >> 5) * ld + ((e & 31) << 1);
        t.v[q] = f64x2{p[0], p[1]};
    }
}
template <int LD>
__device__ __forceinline__ void tile_commit(double* __restrict__ T, const TileRegs& t, int tid)
{
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int e = tid + 256 * q;
        *reinterpret_cast<f64x2*>(T + (e >> 5) * LD + ((e & 31) << 1)) = t.v[q];
    }
}

// In-place Cholesky of the 64x64 LDS tile D (lower triangle result, upper zeroed) together with the
// inverse of the factor in X (both [row][col] with leading dimension LDT).  Returns (block-uniform) 1
// if a pivot was not positive.  Right-looking, all four waves, two barriers per column j:
//   phase 1 (wave 0, lane c):   d = sqrt(D[j][j]);  L[c][j] = D[c][j] / d  (c > j),  L[j][j] = d;
//                               X[j][c] = T[j][c] / d  (c <= j)   -- row j of L^-1, T starts as I --
//                               M[c] = L[c][j] for c > j,  X[j][c] for c <= j
//   phase 2 (rows i > j dealt round-robin to the waves, lane = column c <= i):
//                               c >  j:  D[i][c] -= L[i][j] * L[c][j]     (trailing update of the Cholesky)
//                               c <= j:  X[i][c] -= L[i][j] * X[j][c]     (forward substitution of L X = I)
// i.e. one fused rank-1 sweep per column; every LDS access of phase 2 is row-contiguous.  The single-wave
// left-looking routine this replaces took 75 us per tile (a chain of ~8000 dependent LDS reads).
__device__ int tile_chol_inv(double* __restrict__ D, double* __restrict__ X, int tid, int* s_flag)
{
    __shared__ double s_lc[NB];      // column j of L
    __shared__ double s_m[NB];       // the row multiplier M of phase 2
    const int lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *s_flag = 0;
    for (int e = tid; e < NB * NB; e += 256) X[(e >> 6) * LDT + (e & 63)] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0;
    __syncthreads();
    int bad = 0;
    for (int j = 0; j < NB; j++) {
        if (wave == 0) {
            const int c = lane;
            const double piv = D[j * LDT + j];
            if (!(piv > 0.0)) bad = 1;
            const double a = D[c * LDT + j];
            const double t = X[j * LDT + c];
            // 1/sqrt(piv) by v_rsq_f64 + two Newton steps (full double accuracy) instead of an fp64 sqrt and two
            // fp64 divisions on the 64-step critical path; a non-positive pivot gives NaN and is flagged above
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
        {
            const int c = lane;
            const double m = s_m[c];
            double* const base = (c > j) ? D : X;
            // rows j+1+wave, +4, ... in batches of four: the LDS reads of a batch are issued before its first
            // dependent op; the LDS pipe is the bottleneck of this routine, so no row beyond NB is touched
            for (int i0 = j + 1 + wave; i0 < NB; i0 += 16) {
                double li[4], v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + 4 * u;
                    if (i < NB) { li[u] = s_lc[i]; v[u] = base[i * LDT + c]; }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + 4 * u;
                    if (i < NB && c <= i) base[i * LDT + c] = fma(-li[u], m, v[u]);
                }
            }
        }
        __syncthreads();
    }
    if (bad) *s_flag = 1;            // only wave 0 ever sets bad
    __syncthreads();
    return *s_flag;
}

// ------------------------------------------------------------------------------------------
// MFMA-blocked form of the same routine (block columns of 16), the one the factor kernels use.
//   for K = 0..3:   wave 0 holds block column K in registers, lane = tile row r (rows >= 16K matter), and runs
//                   the 16 pivot steps of that column on the whole 64-row panel at once: pivot by v_readlane,
//                   1/sqrt by v_rsq_f64 + Newton, column scale, and the rank-1 update of the panel's remaining
//                   columns with the multipliers L[16K+c][j] fetched by v_readlane (no LDS round trip, no
//                   workgroup barrier inside a block column; the rows below the diagonal block come out as the
//                   finished panel, so no triangular solve / inverse is needed on this critical path);
//                   then all four waves apply the rank-16 update of the trailing blocks on the fp64 matrix cores.
//   X = L^-1:       the four diagonal 16x16 blocks are inverted in parallel (wave w: block w, lane = column of X,
//                   forward substitution with wave-uniform L entries), the off-diagonal blocks follow by distance
//                   from the diagonal:  X_ij = -X_ii * sum_{m=j}^{i-1} L_im X_mj   (two small MFMA products).
// ------------------------------------------------------------------------------------------
// wave 0: the 16 pivot steps of block column K on the 64-row panel (lane = row).  `bad` is wave-uniform.
template <int K>
__device__ __forceinline__ void chol_block_column(double* __restrict__ D, double* __restrict__ s_rinv, int lane, int& bad)
{
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
        const f64x2 v = *reinterpret_cast<const f64x2*>(D + lane * LDT + 16 * K + c);
        a[c] = v[0]; a[c + 1] = v[1];
    }
    chol_pivots<K>(a, s_rinv, lane, bad);
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
        f64x2 v;
        v[0] = (lane >= 16 * K + c) ? a[c] : 0.0;
        v[1] = (lane >= 16 * K + c + 1) ? a[c + 1] : 0.0;
        *reinterpret_cast<f64x2*>(D + lane * LDT + 16 * K + c) = v;
    }
}

// D_ij -= L_iK L_jK^T for one 16 x 16 block (i >= j > K), one wave
template <int K>
__device__ __forceinline__ void chol_trailing_block(double* __restrict__ D, int i, int j, int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    f64x4 acc;
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] = D[(16 * i + lk + 4 * r) * LDT + 16 * j + lr];
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-D[(16 * i + lr) * LDT + 16 * K + k0 + lk],
                                                   D[(16 * j + lr) * LDT + 16 * K + k0 + lk], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) D[(16 * i + lk + 4 * r) * LDT + 16 * j + lr] = acc[r];
}

template <int K>
__device__ __forceinline__ void chol_trailing(double* __restrict__ D, int wave, int lane)
{
    int t = 0;
#pragma unroll
    for (int j = K + 1; j < 4; j++)
#pragma unroll
        for (int i = j; i < 4; i++) {
            if ((t & 3) == wave) chol_trailing_block<K>(D, i, j, lane);
            t++;
        }
}

__device__ int tile_chol_inv_blk(double* __restrict__ D, double* __restrict__ X, int tid, int* s_flag)
{
    __shared__ double s_rinv[NB];
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) *s_flag = 0;
    for (int e = tid; e < NB * NB; e += 256) X[(e >> 6) * LDT + (e & 63)] = 0.0;
    __syncthreads();
    int bad = 0;
    if (wave == 0) chol_block_column<0>(D, s_rinv, lane, bad);
    __syncthreads();
    chol_trailing<0>(D, wave, lane);
    __syncthreads();
    if (wave == 0) chol_block_column<1>(D, s_rinv, lane, bad);
    __syncthreads();
    chol_trailing<1>(D, wave, lane);
    __syncthreads();
    if (wave == 0) chol_block_column<2>(D, s_rinv, lane, bad);
    __syncthreads();
    chol_trailing<2>(D, wave, lane);
    __syncthreads();
    if (wave == 0) { chol_block_column<3>(D, s_rinv, lane, bad); if (bad && lane == 0) *s_flag = 1; }
    __syncthreads();
    // the strict upper blocks of D were never touched by the factorisation: clear them (L is lower triangular)
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        if ((c >> 4) > (r >> 4)) D[r * LDT + c] = 0.0;
    }
    // ---- X_ww = L_ww^-1: lane c (< 16) solves L_ww x = e_c by forward substitution; the L entries are wave-uniform
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
            for (int i = 0; i < 16; i++) X[(b + i) * LDT + b + c] = sacc[i];     // exact zeros above the diagonal
        }
    }
    __syncthreads();
    // ---- off-diagonal blocks by distance from the diagonal
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int dist = 1; dist < 4; dist++) {
        const int i = dist + wave, j = wave;                    // block (i, j): one per wave, 4 - dist of them
        if (i < 4) {
            f64x4 s = f64x4{0.0, 0.0, 0.0, 0.0};
            for (int m = j; m < i; m++)
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4)
                    s = __builtin_amdgcn_mfma_f64_16x16x4f64(D[(16 * i + lr) * LDT + 16 * m + k0 + lk],
                                                             X[(16 * m + k0 + lk) * LDT + 16 * j + lr], s, 0, 0, 0);
            // park S in the (still empty) X_ij block, then X_ij = -X_ii S
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
        __syncthreads();
    }
    return *s_flag;
}

// Layout of the factor workspace of one problem: pb.A = [A0 | A1 | L0 | L1 | W0], each Mld x Mld
// row-major; pb.Linv = [2][nblk][NB x NB] inverses of the diagonal blocks of L.

// ------------------------------------------------------------------------------------------
// K5: right-looking blocked Cholesky (block 64) of both matrices of every problem.
// Working copies: matrix 0 is factored in W0 = pb.A + 4 ld^2 (the epilogue writes B11 there as well; A[0]
// itself stays intact for QCAT right-hand sides, B11 export and the clamp path), matrix 1 = B11 - eps I is
// factored in place in A[1] (nobody needs it afterwards).  L goes to A[2], A[3], the inverses of the diagonal
// blocks to pb.Linv.  Per block column s:
//   panel(s):   L[k][s] = W[k][s] * Linv_ss^T                      for k > s        (one product per workgroup)
//   update(s):  W[k][j] -= L[k][s] * L[j][s]^T                      for s < j <= k   (one product per workgroup)
//               and the workgroup of tile (s+1, s+1) -- dispatched first -- goes on to factor that tile and
//               to invert the factor, so that panel(s+1) finds L_dd^-1 ready.
// Every step is the same short dependency chain (one product, one product + 64x64 Cholesky) whatever s is;
// the left-looking variant this replaces chained s products per workgroup.
// ------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------
// K6/K7, two ways to z / info of a window:
//   * fused (default, every batched job): the rows that ride in the factorisation's launches solve
//     L [X | y] = [I | z1]  -- M + 1 right-hand sides in panels of NR columns (ride_pre / ride_fin below) -- and
//     impute_gemm_kernel forms  W = B21 X^T  as a plain product with the z / info sums in its epilogue.
//   * solve_kernel (single-window redo of the clamp path, GAUSS_FUSED_SOLVE=0): forward substitution of the window's
//     own right-hand sides, one panel of NR - 1 b21 rows + the z1 column per workgroup, left-looking over the
//     64-blocks of L:  V_k = Linv_kk * (B_k - sum_{j<k} L_kj V_j),  then z / info from the V blocks.
// Both keep their V blocks in the problem's V scratch ([panel][Mld][NR]).
// ------------------------------------------------------------------------------------------
constexpr int SOLVE_NG = 256 / NR;             // row groups of the z / info reduction
constexpr int SOLVE_RG = NB / SOLVE_NG;        // rows per group
static const size_t SOLVE_SMEM = ((size_t)NB * LDT + (size_t)NB * LDV + 768) * sizeof(double);

struct SolveSums { double z, info, v; };       // per-thread partial sums: column tid % NR, rows of group tid / NR

// acc -= sum_{j = j0, j0 + jstep, ... < kb} L[kb][j] V[j]   (the products of block row kb; TL / TV: the workgroup's LDS tiles)
__device__ __forceinline__ void solve_products(const Prob& pb, int panel, int kb, int j0, int jstep, f64x4 (&acc)[SOLVE_NT],
                                               double* __restrict__ TL, double* __restrict__ TV, int tid)
{
    constexpr int NT = SOLVE_NT;
    const int lane = tid & 63, wave = tid >> 6;
    const int ld = pb.Mld;
    const auto Lm = pb.A + (size_t)2 * ld * ld;               // factor of A[0]
    const auto V = pb.V + (size_t)panel * ld * NR;
    TileRegs rl, rv[NR / 64];
    if (j0 < kb) {
        tile_fetch(rl, Lm + (size_t)kb * NB * ld + (size_t)j0 * NB, ld, tid);
#pragma unroll
        for (int h = 0; h < NR / 64; h++) tile_fetch(rv[h], V + (size_t)j0 * NB * NR + 64 * h, NR, tid);
    }
    for (int jb = j0; jb < kb; jb += jstep) {
        __syncthreads();                                  // previous tiles are no longer being read
        tile_commit<LDT>(TL, rl, tid);
#pragma unroll
        for (int h = 0; h < NR / 64; h++) tile_commit<LDV>(TV + 64 * h, rv[h], tid);
        __syncthreads();
        if (jb + jstep < kb) {                            // next tiles fly during the product
            tile_fetch(rl, Lm + (size_t)kb * NB * ld + (size_t)(jb + jstep) * NB, ld, tid);
#pragma unroll
            for (int h = 0; h < NR / 64; h++) tile_fetch(rv[h], V + (size_t)(jb + jstep) * NB * NR + 64 * h, NR, tid);
        }
        mfma_nn<NT, true>(acc, TL, TV, wave, lane);       // acc = - sum_j L_kj V_j
    }
}

// The right-hand-side block and Linv_kk of block row kb, requested into registers (16 + 16 doubles per thread) so that a
// caller can overlap them with other loads; solve_tail commits them to LDS.
struct SolveRhs { double b[(NB * NR) / 256]; TileRegs li; };

__device__ __forceinline__ void solve_rhs_fetch(const Prob& pb, int panel, int kb, SolveRhs& q, int tid)
{
    const int ld = pb.Mld;
    const int u0 = panel * NRU;
    const bool qcat = pb.kind == WIN_QCAT;
    const int n_predm = pb.n_predm;
    // QCAT right-hand sides (qcat.cpp:216-243): first the B11 columns of the tested measured SNPs
    // (rows n_head .. of the symmetric A[0], which the factorisation leaves intact), then the B21 rows
    const auto Brow = pb.A + (size_t)pb.n_head * ld;
    // column c < NRU: B21[u0+c][kb*64 + r]; column NRU: z1 (zero padded)
#pragma unroll
    for (int i = 0; i < (NB * NR) / 256; i++) {
        const int e = tid + 256 * i;
        const int c = e >> 6, r = e & 63;                 // r fastest: coalesced along a B21 row
        const int k = kb * NB + r;
        double v = 0.0;
        if (c < NRU) {
            const int u = u0 + c;
            if (qcat && u < n_predm) v = Brow[(size_t)u * ld + k];
            else if (u - (qcat ? n_predm : 0) < pb.U) v = pb.B21[(size_t)(u - (qcat ? n_predm : 0)) * ld + k];
        }
        else if (k < pb.M) v = pb.z1[k];
        q.b[i] = v;
    }
    tile_fetch(q.li, pb.Linv + (size_t)kb * NB * NB, NB, tid);
}

// X = B_kb + acc;  V_kb = Linv_kk X  (stored);  z / info sums of the block's rows
__device__ __forceinline__ void solve_tail(const Prob& pb, int panel, int kb, f64x4 (&acc)[SOLVE_NT], const SolveRhs& q,
                                           double* __restrict__ TL, double* __restrict__ TV, SolveSums& sums, int tid)
{
    constexpr int NT = SOLVE_NT;
    const int lane = tid & 63, wave = tid >> 6;
    const int ld = pb.Mld;
    const auto V = pb.V + (size_t)panel * ld * NR;
    const int cc = tid % NR, rg = tid / NR;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (NB * NR) / 256; i++) {
        const int e = tid + 256 * i;
        TV[(e & 63) * LDV + (e >> 6)] = q.b[i];
    }
    tile_commit<LDT>(TL, q.li, tid);
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) TV[acc_row(wave, lane, r) * LDV + acc_col(lane, n)] += acc[n][r];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
    mfma_nn<NT, false>(acc, TL, TV, wave, lane);          // V_k = Linv_kk * X
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = acc_row(wave, lane, r), col = acc_col(lane, n);
            TV[row * LDV + col] = acc[n][r];
            V[(size_t)(kb * NB + row) * NR + col] = acc[n][r];
        }
    __syncthreads();
    // accumulate z and info for column cc over this block's rows RG rg .. RG rg + RG - 1
#pragma unroll
    for (int r = 0; r < SOLVE_RG; r++) {
        const double x = TV[(rg * SOLVE_RG + r) * LDV + cc];
        const double y = TV[(rg * SOLVE_RG + r) * LDV + NRU];
        sums.z = fma(x, y, sums.z);
        sums.info = fma(x, x, sums.info);
        sums.v += x;
    }
}

// block row kb of one panel, all of it in this workgroup; the products are summed in SOLVE_SPLIT interleaved classes
// (j = g, g + SOLVE_SPLIT, ...) that are then added in class order
__device__ __forceinline__ void solve_row(const Prob& pb, int panel, int kb, double* __restrict__ TL, double* __restrict__ TV,
                                          SolveSums& sums, int tid)
{
    f64x4 acc[SOLVE_NT];
#pragma unroll
    for (int n = 0; n < SOLVE_NT; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int g = 0; g < SOLVE_SPLIT && g < kb; g++) {
        f64x4 part[SOLVE_NT];
#pragma unroll
        for (int n = 0; n < SOLVE_NT; n++) part[n] = f64x4{0.0, 0.0, 0.0, 0.0};
        solve_products(pb, panel, kb, g, SOLVE_SPLIT, part, TL, TV, tid);
#pragma unroll
        for (int n = 0; n < SOLVE_NT; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[n][r] += part[n][r];
    }
    SolveRhs q;
    solve_rhs_fetch(pb, panel, kb, q, tid);
    solve_tail(pb, panel, kb, acc, q, TL, TV, sums, tid);
}

// after the last block row: combine the per-thread sums and write z / info (or the QCAT correlation)
__device__ __forceinline__ void solve_finish(const Prob& pb, int panel, double* __restrict__ red, const SolveSums& sums, int tid)
{
    constexpr int NG = SOLVE_NG;
    const int u0 = panel * NRU;
    const bool qcat = pb.kind == WIN_QCAT;
    __syncthreads();
    red[tid] = sums.z;
    red[256 + tid] = sums.info;
    red[512 + tid] = sums.v;
    __syncthreads();
    if (tid < NRU) {
        double z = 0.0, info = 0.0;
        for (int g = 0; g < NG; g++) { z += red[g * NR + tid]; info += red[256 + g * NR + tid]; }
        const int u = u0 + tid;
        if (qcat) {
            // r = CalCor(Linv z1, Linv b)  (util.cpp:72-101; qcat.cpp:221,239), vectors of length M
            double sv = 0.0, sy = 0.0, syy = 0.0;
            for (int g = 0; g < NG; g++) {
                sv += red[512 + g * NR + tid];
                sy += red[512 + g * NR + NRU];
                syy += red[256 + g * NR + NRU];
            }
            if (u < pb.n_rhs) {
                const double n = (double)pb.M;
                const double mx = sy / n, mv = sv / n;
                const double cxx = syy - n * mx * mx;
                const double cvv = info - n * mv * mv;
                const double cxv = z - n * mx * mv;
                pb.out_z[u] = cxv / sqrt(cxx * cvv);
                pb.out_info[u] = cvv;
            }
        } else if (u < pb.U) {
            info = fabs(info);                         // dist.cpp:198
            pb.out_z[u] = z / sqrt(info);              // dist.cpp:200
            pb.out_info[u] = info;                     // dist.cpp:202
        }
    }
}

// ---- rows of the inverse riding in the factorisation's launches ------------------------------------------------
// Block row r of [X | y] for one panel p:  V_r = Linv_rr (B_r - sum_{j = first}^{r-1} L_rj V_j).  Row r needs row
// r - 1, so the rows form a chain next to the factorisation's own; to keep a link of that chain short the sum is
// taken in two launches:
//   pre(r)  in update(r - 1):  the EARLY products j <= r - 2 (they need rows <= r - 2 only), summed in SOLVE_SPLIT
//           interleaved classes (j = g, g + SOLVE_SPLIT, ...): rows with at least `split` early products give one
//           workgroup to each class, shorter rows run the classes one after the other in one workgroup and park their
//           sum (added up in class order, as fin does) -- the same bits either way;
//   fin(r)  in update(r):      the parked sums in class order, then the LAST product j = r - 1 accumulated on top, the
//           right-hand side, Linv_rr, and the store of V_r -- one product and one triangular block: ~12 us.
// update(s) therefore carries fin(s) and pre(s + 1); both read only what earlier launches wrote (L rows <= s + 1 from
// panel(<= s), Linv_ss from update(s - 1), V rows <= s - 1 and pre(s)'s sums from update(s - 1)).  Part is double
// buffered by row parity (pre(s + 1) writes while fin(s) reads).  solve_last_kernel runs fin of the batch's last row.
// acc -= sum over j = j0, j0 + jstep, ... <= jlast of L[r][j] V[j], skipping j below the panel's first row
__device__ __forceinline__ void ride_products(const Prob& pb, int panel, int r, int j0, int jstep, int jlast, f64x4 (&acc)[SOLVE_NT],
                                              double* __restrict__ TL, double* __restrict__ TV, int tid)
{
    constexpr int NT = SOLVE_NT;
    const int first = inv_first_row(pb, panel);
    if (j0 < first) j0 += (first - j0 + jstep - 1) / jstep * jstep;
    const int lane = tid & 63, wave = tid >> 6;
    const int ld = pb.Mld;
    const auto Lm = pb.A + (size_t)2 * ld * ld;               // factor of A[0]
    const auto V = pb.V + (size_t)panel * ld * NR;
    TileRegs rl, rv;
    if (j0 <= jlast) {
        tile_fetch(rl, Lm + (size_t)r * NB * ld + (size_t)j0 * NB, ld, tid);
        tile_fetch(rv, V + (size_t)j0 * NB * NR, NR, tid);
    }
    for (int jb = j0; jb <= jlast; jb += jstep) {
        __syncthreads();                                  // previous tiles are no longer being read
        tile_commit<LDT>(TL, rl, tid);
        tile_commit<LDV>(TV, rv, tid);
        __syncthreads();
        if (jb + jstep <= jlast) {                        // next tiles fly during the product
            tile_fetch(rl, Lm + (size_t)r * NB * ld + (size_t)(jb + jstep) * NB, ld, tid);
            tile_fetch(rv, V + (size_t)(jb + jstep) * NB * NR, NR, tid);
        }
        mfma_nn<NT, true>(acc, TL, TV, wave, lane);
    }
}

__device__ __forceinline__ void ride_pre(const Prob& pb, int panel, int r, int g, int split, double* __restrict__ smem, int tid)
{
    const int n_early = r - 1 - inv_first_row(pb, panel);     // products j = first .. r - 2
    if (n_early < 1) return;
    const bool cut = split > 0 && n_early >= split;
    if (!cut && g != 0) return;
    double* TL = smem;
    double* TV = TL + NB * LDT;
    f64x4 acc[SOLVE_NT];
#pragma unroll
    for (int n = 0; n < SOLVE_NT; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
    if (cut) ride_products(pb, panel, r, g, SOLVE_SPLIT, r - 2, acc, TL, TV, tid);
    else {
        for (int gg = 0; gg < SOLVE_SPLIT; gg++) {
            f64x4 part[SOLVE_NT];
#pragma unroll
            for (int n = 0; n < SOLVE_NT; n++) part[n] = f64x4{0.0, 0.0, 0.0, 0.0};
            ride_products(pb, panel, r, gg, SOLVE_SPLIT, r - 2, part, TL, TV, tid);
#pragma unroll
            for (int n = 0; n < SOLVE_NT; n++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[n][q] += part[n][q];
        }
    }
    const auto P = ride_part(pb, panel, r, g);
#pragma unroll
    for (int n = 0; n < SOLVE_NT; n++)
#pragma unroll
        for (int q = 0; q < 4; q++) P[(size_t)(n * 4 + q) * 256 + tid] = acc[n][q];      // register layout, coalesced
}

__device__ __forceinline__ void ride_fin(const Prob& pb, int panel, int r, int split, double* __restrict__ smem, int tid)
{
    const int first = inv_first_row(pb, panel);
    if (r < first) return;
    double* TL = smem;
    double* TV = TL + NB * LDT;
    const int lane = tid & 63, wave = tid >> 6;
    const int n_early = r - 1 - first;
    const int np = n_early < 1 ? 0 : ((split > 0 && n_early >= split) ? SOLVE_SPLIT : 1);
    // Loads are requested well ahead of their use, but in two rounds so that the workgroup stays within 256 registers
    // (two workgroups per CU): first the right-hand side with Linv_rr and the first two parked sums, then the other two
    // together with the last product's two tiles.
    const bool has_last = r - 1 >= first;
    TileRegs li;
    tile_fetch(li, pb.Linv + (size_t)r * NB * NB, NB, tid);
    f64x4 acc[SOLVE_NT], part[2][SOLVE_NT];
#pragma unroll
    for (int n = 0; n < SOLVE_NT; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
    auto load2 = [&](int g0) {
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (g0 + g < np) {
                const auto P = ride_part(pb, panel, r, g0 + g);
#pragma unroll
                for (int n = 0; n < SOLVE_NT; n++)
#pragma unroll
                    for (int c = 0; c < 4; c++) part[g][n][c] = P[(size_t)(n * 4 + c) * 256 + tid];
            }
        }
    };
    auto add2 = [&](int g0) {
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (g0 + g < np) {
#pragma unroll
                for (int n = 0; n < SOLVE_NT; n++)
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[n][c] += part[g][n][c];
            }
        }
    };
    load2(0);
    TileRegs rl, rv;
    add2(0);
    load2(2);
    if (has_last) {
        const int ld = pb.Mld;
        tile_fetch(rl, pb.A + (size_t)2 * ld * ld + (size_t)r * NB * ld + (size_t)(r - 1) * NB, ld, tid);
        tile_fetch(rv, pb.V + (size_t)panel * ld * NR + (size_t)(r - 1) * NB * NR, NR, tid);
    }
    add2(2);
    if (has_last) {
        tile_commit<LDT>(TL, rl, tid);
        tile_commit<LDV>(TV, rv, tid);
        __syncthreads();
        mfma_nn<SOLVE_NT, true>(acc, TL, TV, wave, lane);
    }
    // X = B_r + acc with B = [I | z1] (column g = 64 panel + c is e_g for g < M and z1 for g == M);  V_r = Linv_rr X
    const auto V = pb.V + (size_t)panel * pb.Mld * NR;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (NB * NR) / 256; i++) {
        const int e = tid + 256 * i;
        const int c = e >> 6, rr = e & 63;
        const int k = r * NB + rr, g = panel * NR + c;
        TV[rr * LDV + c] = (g < pb.M) ? ((g == k) ? 1.0 : 0.0) : ((g == pb.M && k < pb.M) ? pb.z1[k] : 0.0);
    }
    tile_commit<LDT>(TL, li, tid);
    __syncthreads();
#pragma unroll
    for (int n = 0; n < SOLVE_NT; n++)
#pragma unroll
        for (int c = 0; c < 4; c++) TV[acc_row(wave, lane, c) * LDV + acc_col(lane, n)] += acc[n][c];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < SOLVE_NT; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
    mfma_nn<SOLVE_NT, false>(acc, TL, TV, wave, lane);
#pragma unroll
    for (int n = 0; n < SOLVE_NT; n++)
#pragma unroll
        for (int c = 0; c < 4; c++) V[(size_t)(r * NB + acc_row(wave, lane, c)) * NR + acc_col(lane, n)] = acc[n][c];
}

// ------------------------------------------------------------------------------------------
// Certificate that the shifted factorisation is not needed.  MakePosDef (util.cpp:302-318) / CountPC
// (util.cpp:355-388) act only if lambda_min(B11) < eps, and B11 = R + lambda I with R the LD matrix in exact
// arithmetic plus rounding noise E, |E|_2 <= M * 1e-15.  For the pooled estimator R is a Gram matrix of
// standardised rows: PSD.  For the weighted estimator (util.cpp:103-124), in covariance scale
//     C = sum_p wf_p m_p (centred Gram of population p)  +  [ sum_p w_p mu_p mu_p^T - mubar mubar^T ],
// the first term is PSD for w_p >= 0, and by Cauchy-Schwarz the bracket is >= -(W - 1)_+ sum_p w_p mu_p mu_p^T
// with W = sum_p w_p (the PGC2 weights are un-normalised, W = 1.061).  In correlation scale therefore
//     lambda_min(B11) >= lambda - (W - 1)_+ * sum_p w_p sum_i (mu_p(i) / sd_i)^2 - M * 1e-10.
// If that bound exceeds eps the predicate is decided (status[3] = 1) and matrix 1 = B11 - eps I is never
// factored; otherwise (negative weights, degenerate rows, tiny lambda) the exact test runs as before.
// The m^2 scaling of the reference's covariance (quirk Q1) makes the correction term ~1e-3 on real panels.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shift_cert_kernel(const Prob* __restrict__ probs)
{
    __shared__ double red[256];
    __shared__ int s_ok;
    const Prob& pb = probs[blockIdx.x];
    if (pb.ld_only || pb.npanel == 0) return;
    const int tid = threadIdx.x;
    const int P = pb.P;
    if (tid == 0) s_ok = 1;
    __syncthreads();
    double wtot = 0.0;
    int ok = 1;
    if (pb.mode != 0)
        for (int p = 0; p < P; p++) {
            const double w = pb.pop_w[p];
            if (!(w >= 0.0) || !(pb.pop_md[p] >= 2.0)) ok = 0;
            wtot += w;
        }
    else wtot = 1.0;
    double acc = 0.0;
    for (int i = tid; i < pb.M; i += 256) {
        const double sd = pb.rt_sd[i];
        if (!(sd > 0.0) || !(sd < 1e300)) ok = 0;
        if (pb.mode != 0) {
            const double inv = 1.0 / (sd * sd);
            for (int p = 0; p < P; p++) {
                const double mu = pb.rt_mu[(size_t)i * P + p];
                acc = fma(pb.pop_w[p] * mu, mu * inv, acc);
            }
        }
    }
    if (!ok) s_ok = 0;
    red[tid] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) red[tid] += red[tid + h];
        __syncthreads();
    }
    if (tid == 0) {
        const double excess = wtot > 1.0 ? wtot - 1.0 : 0.0;
        const double bound = pb.lambda - excess * red[0] - 1e-10 * (double)pb.M;
        pb.status[3] = (s_ok && bound > pb.eps) ? 1 : 0;       // NaN compares false
    }
}

__global__ __launch_bounds__(256) void factor_init_kernel(const Prob* __restrict__ probs)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TD = smem;
    double* TX = TD + NB * LDT;
    __shared__ int s_flag;
    const Prob& pb = probs[blockIdx.x >> 1];
    const int mat = blockIdx.x & 1;
    if (pb.ld_only || pb.npanel == 0) return;
    if (mat == 1 && pb.status[3]) return;              // certified: lambda_min(B11) > eps
    const int tid = threadIdx.x, ld = pb.Mld;
    const auto W = factor_work(pb, mat);
    const auto Lm = pb.A + (size_t)(2 + mat) * ld * ld;
    tile_load(TD, W, ld, tid);
    const int fail = tile_chol_inv_blk(TD, TX, tid, &s_flag);
    tile_store(Lm, ld, TD, tid);
    const auto Li = pb.Linv + (size_t)mat * pb.nblk * NB * NB;
    for (int e = tid; e < NB * NB; e += 256) Li[e] = TX[(e >> 6) * LDT + (e & 63)];
    if (fail && tid == 0) pb.status[mat] = 1;
}

// panel(s): grid.x = max_nblk - 1 - s (block row k = s + 1 + x), grid.y = problem * 2 + matrix
__global__ __launch_bounds__(256) void factor_panel_kernel(const Prob* __restrict__ probs, int s, int T, int split, int n_comb)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TA = smem;                 // W[k][s]
    double* TB = TA + NB * LDT;        // Linv_ss
    const Prob& pb = probs[blockIdx.y >> 1];
    const int mat = blockIdx.y & 1;
    if (pb.ld_only || pb.npanel == 0) return;
    (void)split;
    if (mat == 1 && pb.status[3]) return;              // certified: lambda_min(B11) > eps
    const int nb = pb.nblk;
    const int k = s + 1 + ((int)blockIdx.x - n_comb);
    if (k >= nb) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = pb.Mld;
    const auto W = factor_work(pb, mat);
    const auto Lm = pb.A + (size_t)(2 + mat) * ld * ld;
    TileRegs ra;
    tile_fetch(ra, W + (size_t)k * NB * ld + (size_t)s * NB, ld, tid);
    {
        const auto Li = pb.Linv + ((size_t)mat * nb + s) * NB * NB;
        for (int e = tid; e < NB * NB; e += 256) TB[(e >> 6) * LDT + (e & 63)] = Li[e];
    }
    tile_commit<LDT>(TA, ra, tid);
    __syncthreads();
    f64x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
    mfma_nt<4, false>(acc, TA, TB, wave, lane);              // L[k][s] = W[k][s] * Linv_ss^T
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            Lm[(size_t)(k * NB + acc_row(wave, lane, r)) * ld + s * NB + acc_col(lane, n)] = acc[n][r];
}

// update(s): grid.x = T (T + 1) / 2 with T = max_nblk - 1 - s; x = 0 is tile (s+1, s+1)
__global__ __launch_bounds__(256, 2) void factor_update_kernel(const Prob* __restrict__ probs, int s, int T, int n_tri, int split, int n_ride,
                                                               int own_panel)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TA = smem;                 // L[k][s]            | D (next diagonal tile)
    double* TB = TA + NB * LDT;        // L[j][s]            | its inverse
    __shared__ int s_flag;
    const Prob& pb = probs[blockIdx.y >> 1];
    const int mat = blockIdx.y & 1;
    if (pb.ld_only || pb.npanel == 0) return;
    // launch order: x = 0 the next diagonal tile (the longest workgroup: tile Cholesky), x = 1 .. n_ride the rows of the
    // inverse that ride in this launch (chains of dependent products), then the one-product trailing tiles, which fill
    // in behind them -- with the riding rows last, every launch ended on a 15-20 us tail of theirs
    if ((int)blockIdx.x >= 1 && (int)blockIdx.x <= n_ride) {
        // riding rows of the inverse: per panel one workgroup for fin(s) and SOLVE_SPLIT for pre(s + 1)
        const int idx = (int)blockIdx.x - 1;
        const int panel = idx / (SOLVE_SPLIT + 1), g = idx % (SOLVE_SPLIT + 1);
        if (mat != 0 || panel >= pb.npi) return;
        if (g == SOLVE_SPLIT) { if (s < pb.nblk) ride_fin(pb, panel, s, split, smem, threadIdx.x); }
        else if (s + 1 < pb.nblk) ride_pre(pb, panel, s + 1, g, split, smem, threadIdx.x);
        return;
    }
    if (mat == 1 && pb.status[3]) return;              // certified: lambda_min(B11) > eps
    const int nb = pb.nblk;
    // x -> (jj, kk), 0 <= jj <= kk < T, column-major over the lower triangle: x = 0 is (0, 0)
    int jj = 0, rem = blockIdx.x == 0 ? 0 : (int)blockIdx.x - n_ride;
    if (rem >= n_tri) return;
    while (rem >= T - jj) { rem -= T - jj; jj++; }
    const int kk = jj + rem;
    const int j = s + 1 + jj, k = s + 1 + kk;
    if (k >= nb) return;
    const bool next_diag = (jj == 0 && kk == 0);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = pb.Mld;
    const auto W = factor_work(pb, mat);
    const auto Lm = pb.A + (size_t)(2 + mat) * ld * ld;
    TileRegs ra, rb;
    f64x4 acc[4];
    if (own_panel) {
        // Small jobs: no panel launch.  This workgroup forms the two tiles of block column s it needs itself,
        // L[k][s] = W[k][s] Linv_ss^T and L[j][s] = W[j][s] Linv_ss^T (the products panel(s) does, same arithmetic,
        // done again by every workgroup that needs them); the workgroups of the first trailing column (j = s + 1)
        // store theirs for the rows of the inverse and for later steps' readers.  One launch and one round trip of
        // tiles between CUs less per block step; the extra products are noise on a latency-bound job.
        TileRegs rli;
        tile_fetch(ra, W + (size_t)k * NB * ld + (size_t)s * NB, ld, tid);
        if (k != j) tile_fetch(rb, W + (size_t)j * NB * ld + (size_t)s * NB, ld, tid);
        tile_fetch(rli, pb.Linv + ((size_t)mat * nb + s) * NB * NB, NB, tid);
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                acc[n][r] = W[(size_t)(k * NB + acc_row(wave, lane, r)) * ld + j * NB + acc_col(lane, n)];
        tile_commit<LDT>(TA, ra, tid);
        tile_commit<LDT>(TB, rli, tid);
        __syncthreads();
        f64x4 lk[4], lj[4];
#pragma unroll
        for (int n = 0; n < 4; n++) { lk[n] = f64x4{0.0, 0.0, 0.0, 0.0}; lj[n] = f64x4{0.0, 0.0, 0.0, 0.0}; }
        mfma_nt<4, false>(lk, TA, TB, wave, lane);
        if (k != j) {
            __syncthreads();
            tile_commit<LDT>(TA, rb, tid);
            __syncthreads();
            mfma_nt<4, false>(lj, TA, TB, wave, lane);
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                TA[acc_row(wave, lane, r) * LDT + acc_col(lane, n)] = lk[n][r];
                if (k != j) TB[acc_row(wave, lane, r) * LDT + acc_col(lane, n)] = lj[n][r];
                if (jj == 0) Lm[(size_t)(k * NB + acc_row(wave, lane, r)) * ld + s * NB + acc_col(lane, n)] = lk[n][r];
            }
    } else {
        tile_fetch(ra, Lm + (size_t)k * NB * ld + (size_t)s * NB, ld, tid);
        if (k != j) tile_fetch(rb, Lm + (size_t)j * NB * ld + (size_t)s * NB, ld, tid);
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                acc[n][r] = W[(size_t)(k * NB + acc_row(wave, lane, r)) * ld + j * NB + acc_col(lane, n)];
        tile_commit<LDT>(TA, ra, tid);
        if (k != j) tile_commit<LDT>(TB, rb, tid);
    }
    __syncthreads();
    mfma_nt<4, true>(acc, TA, (k != j) ? TB : TA, wave, lane);       // W[k][j] -= L[k][s] L[j][s]^T
    if (!next_diag) {
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                W[(size_t)(k * NB + acc_row(wave, lane, r)) * ld + j * NB + acc_col(lane, n)] = acc[n][r];
        return;
    }
    __syncthreads();                                                  // everyone is done reading TA
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) TA[acc_row(wave, lane, r) * LDT + acc_col(lane, n)] = acc[n][r];
    __syncthreads();
    const int fail = tile_chol_inv_blk(TA, TB, tid, &s_flag);         // TA = L_dd, TB = its inverse
    tile_store(Lm + (size_t)k * NB * ld + (size_t)k * NB, ld, TA, tid);
    const auto Li = pb.Linv + ((size_t)mat * nb + k) * NB * NB;
    for (int e = tid; e < NB * NB; e += 256) Li[e] = TB[(e >> 6) * LDT + (e & 63)];
    if (fail && tid == 0) pb.status[mat] = 1;
}

static const size_t FACTOR_SMEM = (size_t)2 * NB * LDT * sizeof(double);

// step 0 factors the first diagonal block; step s >= 1 builds block column s-1 and updates the trailing matrix.
// max_npanel > 0: update(c) also carries fin(c) and pre(c + 1) of the inverse's rows for every panel of [I | z1]
// (ride_pre / ride_fin above); 0: factorisation only.
void launch_factor_step(const Prob* d_probs, int n_prob, int step, int max_nblk, int max_npanel, int split, int own_panel, hipStream_t st)
{
    if (n_prob <= 0 || step >= max_nblk) return;
    const size_t upd_smem = std::max(FACTOR_SMEM, SOLVE_SMEM);
    static DeviceOnce attr_once;
    attr_once.run([&] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(factor_init_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FACTOR_SMEM);
        hipFuncSetAttribute(reinterpret_cast<const void*>(factor_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)upd_smem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(factor_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)upd_smem);
    });
    if (step == 0) {
        hipLaunchKernelGGL(factor_init_kernel, dim3(n_prob * 2), dim3(256), FACTOR_SMEM, st, d_probs);
        return;
    }
    const int s = step - 1;
    const int T = max_nblk - 1 - s;
    if (T <= 0) return;
    const int n_tri = T * (T + 1) / 2;
    const bool fuse = max_npanel > 0;
    const int n_comb = 0, n_ride = fuse ? max_npanel * (SOLVE_SPLIT + 1) : 0;
    if (!own_panel)
        hipLaunchKernelGGL(factor_panel_kernel, dim3(T + n_comb, n_prob * 2), dim3(256), FACTOR_SMEM, st,
                           d_probs, s, T, split, n_comb);
    hipLaunchKernelGGL(factor_update_kernel, dim3(n_tri + n_ride, n_prob * 2), dim3(256),
                       upd_smem, st, d_probs, s, T, n_tri, split, n_ride, own_panel);
}

__global__ __launch_bounds__(256) void solve_kernel(const Prob* __restrict__ probs,
                                                    const int2* __restrict__ panelmap)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TL = smem;                       // [64][LDT]   L_kj, then Linv_kk
    double* TV = TL + NB * LDT;              // [64][LDV]   V_j, then the rhs block X, then V_k
    double* red = TV + NB * LDV;             // [3][256]
    const int2 pm = panelmap[blockIdx.x];
    const Prob& pb = probs[pm.x];
    const int tid = threadIdx.x;
    SolveSums sums{0.0, 0.0, 0.0};
    for (int kb = 0; kb < pb.nblk; kb++) solve_row(pb, pm.y, kb, TL, TV, sums, tid);
    solve_finish(pb, pm.y, red, sums, tid);
}

// What the update launches could not carry: fin of the last block row of the windows that are as tall as the batch's
// tallest one (there is no update(max_nblk - 1)).
__global__ __launch_bounds__(256, 2) void solve_last_kernel(const Prob* __restrict__ probs, const int2* __restrict__ panelmap,
                                                         int s_last, int split)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int2 pm = panelmap[blockIdx.x];
    const Prob& pb = probs[pm.x];
    const int last = pb.nblk - 1, tid = threadIdx.x, panel = pm.y;
    if (last == s_last) ride_fin(pb, panel, s_last, split, smem, tid);     // pre(s_last) rode in update(s_last - 1)
}

// ------------------------------------------------------------------------------------------
// K7: W = R X^T with the imputation sums in the epilogue (fused path).
//   R   right-hand sides, one per row: the window's B21 rows (QCAT: first the B11 columns of the tested measured SNPs)
//   X   = L^-1, block (kb, p) at pb.V[p][64 kb ..][64], lower block triangle; y = L^-1 z1 is column M of [X | y]
//   w_u = X r_u  (= L^-1 b21_u^T);   z_u = w_u . y,  info_u = w_u . w_u,  v_u = sum(w_u)
// One workgroup = UT right-hand sides x 128 rows of X (k), NW waves of 64 (k) x 2 UT / NW (4 x 2 v_mfma_f64_16x16x4_f64 tiles a
// wave in both shipped forms: UT = 128 with 8 waves, UT = 64 with 4); K runs over the columns
// j < 64 (block + 1) of X in stages of 16, double buffered in LDS, the next stage's tiles in flight in registers.
// Nothing of W is stored: each workgroup reduces its tile to three sums per right-hand side in a fixed order and
// parks them in pb.Gsum[k block][rhs][3]; impute_finish_kernel adds the k blocks in order.  A right-hand side's sums
// depend on its own column of the tile only, so UT is free: large jobs take 128 (throughput), small jobs 64 -- twice
// the workgroups, each with half the matrix work per stage of its (latency-bound) K loop -- and the result does not
// depend on the choice or on what else the job holds.
// ------------------------------------------------------------------------------------------
constexpr int GK = 16;                 // K per stage
constexpr int GLD = GK + 2;            // LDS row stride in doubles: 144 B puts the 16 rows of a b128 fragment read on distinct banks
constexpr int GT = 128;                // tile edge (rows of X, and right-hand sides)
static const size_t GEMM_SMEM = (size_t)(2 * 2 * GT * GLD + GT) * sizeof(double);

template <int LQ> struct GemmRegsT { f64x2 v[LQ]; };       // 2 LQ doubles of a tile row's 16-column stage

// NW waves per workgroup: 4 (wave tile 64 k x UT / 2) or 8 (64 k x UT / 4: half the accumulators a wave, four waves a SIMD instead
// of two on the same 74 KB of LDS).  Round 5: the 128-wide form with eight waves takes 1.07 ms per 36-window step against 1.13
// with four (the same sums in the same order: same bits); the 64-wide form of small jobs keeps four.
template <int UT, int NW>
__global__ __launch_bounds__(64 * NW, 2) void impute_gemm_kernel(const Prob* __restrict__ probs, const int2* __restrict__ gmap)
{
    constexpr int NU = UT / (8 * NW);                           // 16-column tiles of a wave along u
    constexpr int TPR = NW / 2;                                 // threads per tile row of a stage (64 NW threads, 128 rows)
    constexpr int LQ = 8 / TPR;                                 // f64x2 loads per thread and operand
    typedef GemmRegsT<LQ> GemmRegs;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int2 gm = gmap[blockIdx.x];
    const Prob& pb = probs[gm.x];
    const int kblock = gm.y & 255, upanel = gm.y >> 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 1, wu = wave >> 1;
    const int ld = pb.Mld, nb = pb.nblk;
    const int k0 = kblock * GT, u0 = upanel * UT;
    double* ys = smem + 2 * 2 * GT * GLD;

    // ---- operand rows of this thread: row tid / TPR of both tiles, columns 2 LQ (tid % TPR) .. of a stage
    const int trow = tid / TPR, tcol = (tid % TPR) * 2 * LQ;
    const int xk = k0 + trow;                                   // row of X
    const bool x_live = xk < ld;
    const int u = u0 + trow;                                    // right-hand side
    const bool qcat = pb.kind == WIN_QCAT;
    const int n_predm = qcat ? pb.n_predm : 0;
    const bool r_live = trow < UT && u < pb.n_rhs;
    // QCAT right-hand sides (qcat.cpp:216-243): the B11 columns of the tested measured SNPs (rows n_head .. of the
    // symmetric A[0], which the factorisation leaves intact), then the B21 rows
    const auto rrow = !r_live ? pb.B21 : (u < n_predm ? pb.A + (size_t)(pb.n_head + u) * ld : pb.B21 + (size_t)(u - n_predm) * ld);
    auto fetch = [&](GemmRegs& rx, GemmRegs& rr, int j0) {
        const auto xp = pb.V + ((size_t)(j0 >> 6) * ld + xk) * NR + (j0 & 63) + tcol;
        const auto rp = rrow + j0 + tcol;
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            rx.v[q] = x_live ? f64x2{xp[2 * q], xp[2 * q + 1]} : f64x2{0.0, 0.0};
            rr.v[q] = r_live ? f64x2{rp[2 * q], rp[2 * q + 1]} : f64x2{0.0, 0.0};
        }
    };
    auto commit = [&](const GemmRegs& rx, const GemmRegs& rr, int buf) {
        double* xa = smem + (size_t)buf * 2 * GT * GLD + trow * GLD + tcol;
        double* rb = xa + GT * GLD;
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            *reinterpret_cast<f64x2*>(xa + 2 * q) = rx.v[q];
            *reinterpret_cast<f64x2*>(rb + 2 * q) = rr.v[q];
        }
    };

    // columns of X this workgroup needs: blocks 0 .. (last block row it holds), and nothing from column M on (the
    // right-hand sides are zero there).  A wave stops at its own block row, and inside that diagonal block X is lower
    // triangular: its row tile i (16 rows) has nothing right of column 16 (i + 1), and no work at all if it starts
    // at or below row M (padding) -- lim[i] = stages row tile i takes part in.
    const int kb_hi = min(2 * kblock + 1, nb - 1);
    const int n_stage = min((kb_hi + 1) * (NB / GK), (pb.M + GK - 1) / GK);
    const int my_kb = 2 * kblock + wk;
    int lim[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        lim[i] = (my_kb < nb && my_kb * NB + 16 * i < pb.M) ? min(my_kb * (NB / GK) + i + 1, n_stage) : 0;
    const int my_stages = max(max(lim[0], lim[1]), max(lim[2], lim[3]));

    f64x4 acc[4][NU];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int n = 0; n < NU; n++) acc[i][n] = f64x4{0.0, 0.0, 0.0, 0.0};

    GemmRegs rx, rr;
    fetch(rx, rr, 0);
    if (tid < GT) {
        const int k = k0 + tid;
        ys[tid] = k < ld ? pb.V[((size_t)(pb.M / NR) * ld + k) * NR + (pb.M % NR)] : 0.0;
    }
    commit(rx, rr, 0);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int st = 0; st < n_stage; st++) {
        const int buf = st & 1;
        if (st + 1 < n_stage) fetch(rx, rr, (st + 1) * GK);
        if (st < my_stages) {
            const double* xa = smem + (size_t)buf * 2 * GT * GLD + (64 * wk + fr) * GLD + 4 * fg;
            const double* rb = smem + (size_t)buf * 2 * GT * GLD + GT * GLD + (16 * NU * wu + fr) * GLD + 4 * fg;
            f64x2 b[NU][2];
#pragma unroll
            for (int n = 0; n < NU; n++) {
                b[n][0] = *reinterpret_cast<const f64x2*>(rb + 16 * n * GLD);
                b[n][1] = *reinterpret_cast<const f64x2*>(rb + 16 * n * GLD + 2);
            }
            // (Round 5 tried to take the X fragment reads out from in front of their MFMAs -- all four row tiles' fragments requested at
            // the top of the stage, or tile i + 1's before tile i's sixteen MFMAs; the ISA then waits with lgkmcnt(5), (4), ... instead
            // of four exposed LDS round trips a stage -- and the product got SLOWER, 1.22-1.24 ms against 1.12: with two workgroups per
            // CU the other workgroup's wave fills exactly those gaps, and the early reads only lengthen the fragments' lifetimes.)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (st < lim[i]) {                              // wave-uniform
                    const f64x2 a0 = *reinterpret_cast<const f64x2*>(xa + 16 * i * GLD);
                    const f64x2 a1 = *reinterpret_cast<const f64x2*>(xa + 16 * i * GLD + 2);
#pragma unroll
                    for (int ks = 0; ks < 4; ks++)
#pragma unroll
                        for (int n = 0; n < NU; n++)
                            acc[i][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(ks < 2 ? a0[ks & 1] : a1[ks & 1], b[n][ks >> 1][ks & 1],
                                                                             acc[i][n], 0, 0, 0);
                }
            }
        }
        if (st + 1 < n_stage) commit(rx, rr, buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: per lane the 16 rows (k) it holds of each of its 4 columns (u), in a fixed order
    double pz[NU], pi[NU], pv[NU];
#pragma unroll
    for (int n = 0; n < NU; n++) { pz[n] = 0.0; pi[n] = 0.0; pv[n] = 0.0; }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double y = ys[64 * wk + 16 * i + fg + 4 * r];
#pragma unroll
            for (int n = 0; n < NU; n++) {
                const double w = acc[i][n][r];
                pz[n] = fma(w, y, pz[n]);
                pi[n] = fma(w, w, pi[n]);
                pv[n] += w;
            }
        }
    double* red = smem;                                         // [wave][n][lane][3], the tiles are no longer read
#pragma unroll
    for (int n = 0; n < NU; n++) {
        double* q = red + ((size_t)(wave * 4 + n) * 64 + lane) * 3;
        q[0] = pz[n]; q[1] = pi[n]; q[2] = pv[n];
    }
    __syncthreads();
    if (tid < UT) {
        const int tu = tid / (16 * NU), n = (tid >> 4) % NU, c = tid & 15;
        double z = 0.0, info = 0.0, v = 0.0;
#pragma unroll
        for (int w = 0; w < 2; w++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const double* q = red + ((size_t)((w + 2 * tu) * 4 + n) * 64 + 16 * g + c) * 3;
                z += q[0]; info += q[1]; v += q[2];
            }
        const auto o = pb.Gsum + ((size_t)kblock * pb.Up128 + u0 + tid) * 3;
        o[0] = z; o[1] = info; o[2] = v;
    }
}

// z / info of every right-hand side from the k blocks' sums (block order), and the QCAT correlation
__global__ __launch_bounds__(256) void impute_finish_kernel(const Prob* __restrict__ probs, const int2* __restrict__ fmap)
{
    __shared__ double s_y[2];
    const int2 fm = fmap[blockIdx.x];
    const Prob& pb = probs[fm.x];
    const int tid = threadIdx.x;
    const int u = fm.y * 256 + tid;
    const bool qcat = pb.kind == WIN_QCAT;
    if (qcat) {
        // sums of y = L^-1 z1 over its Mld entries (the padding rows hold zeros), in index order
        if (tid == 0) {
            double sy = 0.0, syy = 0.0;
            const auto yp = pb.V + (size_t)(pb.M / NR) * pb.Mld * NR + (pb.M % NR);
            for (int k = 0; k < pb.Mld; k++) { const double y = yp[(size_t)k * NR]; sy += y; syy = fma(y, y, syy); }
            s_y[0] = sy; s_y[1] = syy;
        }
        __syncthreads();
    }
    if (u >= pb.n_rhs) return;
    const int nkb = (pb.Mld + GT - 1) / GT;
    double z = 0.0, info = 0.0, sv = 0.0;
    for (int kb = 0; kb < nkb; kb++) {
        const auto q = pb.Gsum + ((size_t)kb * pb.Up128 + u) * 3;
        z += q[0]; info += q[1]; sv += q[2];
    }
    if (qcat) {
        // r = CalCor(Linv z1, Linv b)  (util.cpp:72-101; qcat.cpp:221,239), vectors of length M
        const double n = (double)pb.M;
        const double mx = s_y[0] / n, mv = sv / n;
        const double cxx = s_y[1] - n * mx * mx;
        const double cvv = info - n * mv * mv;
        const double cxv = z - n * mx * mv;
        pb.out_z[u] = cxv / sqrt(cxx * cvv);
        pb.out_info[u] = cvv;
    } else {
        info = fabs(info);                         // dist.cpp:198
        pb.out_z[u] = z / sqrt(info);              // dist.cpp:200
        pb.out_info[u] = info;                     // dist.cpp:202
    }
}

void launch_impute_gemm(const Prob* d_probs, const int2* d_gmap, int n_tiles, int u_tile, const int2* d_fmap, int n_chunks, hipStream_t s)
{
    if (n_tiles <= 0) return;
    static DeviceOnce attr_once;
    attr_once.run([&] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(impute_gemm_kernel<128, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_SMEM);
        hipFuncSetAttribute(reinterpret_cast<const void*>(impute_gemm_kernel<64, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_SMEM);
    });
    if (u_tile == 64) hipLaunchKernelGGL((impute_gemm_kernel<64, 4>), dim3(n_tiles), dim3(256), GEMM_SMEM, s, d_probs, d_gmap);
    else hipLaunchKernelGGL((impute_gemm_kernel<128, 8>), dim3(n_tiles), dim3(512), GEMM_SMEM, s, d_probs, d_gmap);
    hipLaunchKernelGGL(impute_finish_kernel, dim3(n_chunks), dim3(256), 0, s, d_probs, d_fmap);
}

// The certificate only needs the row tables: it is launched right after row_stats, ahead of the Gram kernel, so that
// its ~30 us never sit on the latency-bound factorisation chain.  status[3] must not be cleared afterwards.
void launch_shift_cert(const Prob* d_probs, int n_prob, hipStream_t st)
{
    const bool no_cert = getenv("GAUSS_NO_SHIFT_CERT") != nullptr;     // fallback: never trust the bound, always factor B11 - eps I too (read per call)
    if (n_prob > 0 && !no_cert) hipLaunchKernelGGL(shift_cert_kernel, dim3(n_prob), dim3(256), 0, st, d_probs);
}

void launch_solve(const Prob* d_probs, const int2* d_panelmap, int n_panels, hipStream_t s)
{
    if (n_panels <= 0) return;
    static DeviceOnce attr_once;
    attr_once.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SOLVE_SMEM); });
    hipLaunchKernelGGL(solve_kernel, dim3(n_panels), dim3(256), SOLVE_SMEM, s, d_probs, d_panelmap);
}

void launch_solve_last(const Prob* d_probs, const int2* d_panelmap, int n_panels, int max_nblk, int split, hipStream_t s)
{
    if (n_panels <= 0 || max_nblk < 1) return;
    static DeviceOnce attr_once;
    attr_once.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(solve_last_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SOLVE_SMEM); });
    hipLaunchKernelGGL(solve_last_kernel, dim3(n_panels), dim3(256), SOLVE_SMEM, s, d_probs, d_panelmap, max_nblk - 1, split);
}

}  // namespace gauss
