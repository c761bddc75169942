// K5 blocked Cholesky (fp64), K6/K7 forward solve + imputation finalize -- fp64 matrix cores.
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
//
// All block products run on v_mfma_f64_16x16x4_f64 (one f64 A and one f64 B value per lane; result
// rows (lane>>4) + 4*reg, column lane&15).  A workgroup is 4 waves; wave w owns rows 16w..16w+15
// of a 64-row block.
#include "gauss_internal.h"

namespace gauss {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int LDT = NB + 2;    // LDS leading dimension of a 64 x 64 [row][k] tile: 66 doubles = 528 B;
                               // 528 mod 256 = 16 puts the 32 lanes of a ds_read_b64 group on distinct banks
constexpr int LDV = NR + 2;     // LDS leading dimension of the 64 x NR [k][col] tile of the solve

#define WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

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

// accumulator tile element (reg r of tile n) -> (row, col) inside the 64 x (16 NT) block
__device__ __forceinline__ int acc_row(int wave, int lane, int r) { return 16 * wave + (lane >> 4) + 4 * r; }
__device__ __forceinline__ int acc_col(int lane, int n) { return 16 * n + (lane & 15); }

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
typedef double f64x2 __attribute__((ext_vector_type(2)));
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
__device__ __forceinline__ GP(double) factor_work(const Prob& pb, int mat)
{
    const size_t ld2 = (size_t)pb.Mld * pb.Mld;
    return mat == 0 ? pb.A + 4 * ld2 : pb.A + ld2;
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
    const int tid = threadIdx.x, ld = pb.Mld;
    const auto W = factor_work(pb, mat);
    const auto Lm = pb.A + (size_t)(2 + mat) * ld * ld;
    tile_load(TD, W, ld, tid);
    const int fail = tile_chol_inv(TD, TX, tid, &s_flag);
    tile_store(Lm, ld, TD, tid);
    const auto Li = pb.Linv + (size_t)mat * pb.nblk * NB * NB;
    for (int e = tid; e < NB * NB; e += 256) Li[e] = TX[(e >> 6) * LDT + (e & 63)];
    if (fail && tid == 0) pb.status[mat] = 1;
}

// panel(s): grid.x = max_nblk - 1 - s (block row k = s + 1 + x), grid.y = problem * 2 + matrix
__global__ __launch_bounds__(256) void factor_panel_kernel(const Prob* __restrict__ probs, int s)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TA = smem;                 // W[k][s]
    double* TB = TA + NB * LDT;        // Linv_ss
    const Prob& pb = probs[blockIdx.y >> 1];
    const int mat = blockIdx.y & 1;
    if (pb.ld_only || pb.npanel == 0) return;
    const int nb = pb.nblk;
    const int k = s + 1 + blockIdx.x;
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
__global__ __launch_bounds__(256) void factor_update_kernel(const Prob* __restrict__ probs, int s, int T)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TA = smem;                 // L[k][s]            | D (next diagonal tile)
    double* TB = TA + NB * LDT;        // L[j][s]            | its inverse
    __shared__ int s_flag;
    const Prob& pb = probs[blockIdx.y >> 1];
    const int mat = blockIdx.y & 1;
    if (pb.ld_only || pb.npanel == 0) return;
    const int nb = pb.nblk;
    // x -> (jj, kk), 0 <= jj <= kk < T, column-major over the lower triangle: x = 0 is (0, 0)
    int jj = 0, rem = blockIdx.x;
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
    tile_fetch(ra, Lm + (size_t)k * NB * ld + (size_t)s * NB, ld, tid);
    if (k != j) tile_fetch(rb, Lm + (size_t)j * NB * ld + (size_t)s * NB, ld, tid);
    f64x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            acc[n][r] = W[(size_t)(k * NB + acc_row(wave, lane, r)) * ld + j * NB + acc_col(lane, n)];
    tile_commit<LDT>(TA, ra, tid);
    if (k != j) tile_commit<LDT>(TB, rb, tid);
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
    const int fail = tile_chol_inv(TA, TB, tid, &s_flag);             // TA = L_dd, TB = its inverse
    tile_store(Lm + (size_t)k * NB * ld + (size_t)k * NB, ld, TA, tid);
    const auto Li = pb.Linv + ((size_t)mat * nb + k) * NB * NB;
    for (int e = tid; e < NB * NB; e += 256) Li[e] = TB[(e >> 6) * LDT + (e & 63)];
    if (fail && tid == 0) pb.status[mat] = 1;
}

static const size_t FACTOR_SMEM = (size_t)2 * NB * LDT * sizeof(double);

// step 0 factors the first diagonal block; step s >= 1 builds block column s-1 and updates the trailing matrix
void launch_factor_step(const Prob* d_probs, int n_prob, int step, int max_nblk, hipStream_t st)
{
    if (n_prob <= 0 || step >= max_nblk) return;
    static std::atomic<unsigned long long> attr_set{0};
    if (first_use_on_device(attr_set)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(factor_init_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FACTOR_SMEM);
        hipFuncSetAttribute(reinterpret_cast<const void*>(factor_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FACTOR_SMEM);
        hipFuncSetAttribute(reinterpret_cast<const void*>(factor_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FACTOR_SMEM);
    }
    if (step == 0) {
        hipLaunchKernelGGL(factor_init_kernel, dim3(n_prob * 2), dim3(256), FACTOR_SMEM, st, d_probs);
        return;
    }
    const int s = step - 1;
    const int T = max_nblk - 1 - s;
    if (T <= 0) return;
    hipLaunchKernelGGL(factor_panel_kernel, dim3(T, n_prob * 2), dim3(256), FACTOR_SMEM, st, d_probs, s);
    hipLaunchKernelGGL(factor_update_kernel, dim3(T * (T + 1) / 2, n_prob * 2), dim3(256), FACTOR_SMEM, st, d_probs, s, T);
}

// ------------------------------------------------------------------------------------------
// K6/K7: forward substitution for one panel of NR right-hand sides (NR - 1 unmeasured SNPs'
// b21 rows + the z1 column), left-looking over the 64-blocks of L, then z / info.
// One workgroup per panel; panels are independent (no inter-workgroup traffic).
//   V_k = Linv_kk * (B_k - sum_{j<k} L_kj V_j)
// V blocks live in the problem's V scratch ([panel][Mld][NR]) for reuse by later blocks.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void solve_kernel(const Prob* __restrict__ probs,
                                                    const int2* __restrict__ panelmap)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* TL = smem;                       // [64][LDT]   L_kj, then Linv_kk
    double* TV = TL + NB * LDT;              // [64][LDV]   V_j, then the rhs block X, then V_k
    double* red = TV + NB * LDV;             // [3][256]
    constexpr int NT = NR / 16;              // accumulator tiles (16 columns each) per wave
    constexpr int NG = 256 / NR;             // row groups of the z / info reduction
    constexpr int RG = NB / NG;              // rows per group

    const int2 pm = panelmap[blockIdx.x];
    const Prob& pb = probs[pm.x];
    const int panel = pm.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = pb.Mld, nb = pb.nblk;
    const auto Lm = pb.A + (size_t)2 * ld * ld;               // factor of A[0]
    const auto Linv = pb.Linv;                                // matrix 0
    const auto V = pb.V + (size_t)panel * ld * NR;
    const int u0 = panel * NRU;
    const bool qcat = pb.kind == WIN_QCAT;
    const int n_predm = pb.n_predm;
    // QCAT right-hand sides (qcat.cpp:216-243): first the B11 columns of the tested measured SNPs
    // (rows n_head .. of the symmetric A[0], which the factorisation leaves intact), then the B21 rows
    const auto Brow = pb.A + (size_t)pb.n_head * ld;

    const int cc = tid % NR, rg = tid / NR;                   // reduction: column cc, rows RG rg ..
    double zsum = 0.0, isum = 0.0, vsum = 0.0;

    for (int kb = 0; kb < nb; kb++) {
        f64x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
        TileRegs rl, rv[NR / 64];
        if (kb > 0) {
            tile_fetch(rl, Lm + (size_t)kb * NB * ld, ld, tid);
#pragma unroll
            for (int h = 0; h < NR / 64; h++) tile_fetch(rv[h], V + 64 * h, NR, tid);
        }
        for (int jb = 0; jb < kb; jb++) {
            __syncthreads();                                  // previous tiles are no longer being read
            tile_commit<LDT>(TL, rl, tid);
#pragma unroll
            for (int h = 0; h < NR / 64; h++) tile_commit<LDV>(TV + 64 * h, rv[h], tid);
            __syncthreads();
            if (jb + 1 < kb) {                                // next tiles fly during the product
                tile_fetch(rl, Lm + (size_t)kb * NB * ld + (size_t)(jb + 1) * NB, ld, tid);
#pragma unroll
                for (int h = 0; h < NR / 64; h++) tile_fetch(rv[h], V + (size_t)(jb + 1) * NB * NR + 64 * h, NR, tid);
            }
            mfma_nn<NT, true>(acc, TL, TV, wave, lane);       // acc = - sum_j L_kj V_j
        }
        __syncthreads();
        // TV <- rhs block: column c < NRU: B21[u0+c][kb*64 + r]; column NRU: z1 (zero padded)
        for (int e = tid; e < NB * NR; e += 256) {
            const int c = e >> 6, r = e & 63;                 // r fastest: coalesced along a B21 row
            const int k = kb * NB + r;
            double v = 0.0;
            if (c < NRU) {
                const int u = u0 + c;
                if (qcat && u < n_predm) v = Brow[(size_t)u * ld + k];
                else if (u - (qcat ? n_predm : 0) < pb.U) v = pb.B21[(size_t)(u - (qcat ? n_predm : 0)) * ld + k];
            }
            else if (k < pb.M) v = pb.z1[k];
            TV[r * LDV + c] = v;
        }
        {
            const auto Li = Linv + (size_t)kb * NB * NB;
            for (int e = tid; e < NB * NB; e += 256) TL[(e >> 6) * LDT + (e & 63)] = Li[e];
        }
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
        for (int r = 0; r < RG; r++) {
            const double x = TV[(rg * RG + r) * LDV + cc];
            const double y = TV[(rg * RG + r) * LDV + NRU];
            zsum = fma(x, y, zsum);
            isum = fma(x, x, isum);
            vsum += x;
        }
    }
    __syncthreads();
    red[tid] = zsum;
    red[256 + tid] = isum;
    red[512 + tid] = vsum;
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

void launch_solve(const Prob* d_probs, const int2* d_panelmap, int n_panels, hipStream_t s)
{
    if (n_panels <= 0) return;
    const size_t sh = ((size_t)NB * LDT + (size_t)NB * LDV + 768) * sizeof(double);
    static std::atomic<unsigned long long> attr_set{0};
    if (first_use_on_device(attr_set))
        hipFuncSetAttribute(reinterpret_cast<const void*>(solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(solve_kernel, dim3(n_panels), dim3(256), sh, s, d_probs, d_panelmap);
}

}  // namespace gauss
