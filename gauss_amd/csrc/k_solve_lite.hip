// The factorisation chain (K5 + the riding rows of the inverse, k_solve.hip) rebuilt to run BESIDE the Gram kernel.
//
// gram_kernel<float> keeps four workgroups on every CU: 4 x 104 of the 512 vector registers of a SIMD lane, 4 x 32 KB of the
// 160 KB of LDS, four of the wave slots of a SIMD.  What is left on EVERY CU at ALL times is 96 registers per lane, 32 KB of
// LDS and four wave slots per SIMD -- room for one more workgroup of 256 threads, if it fits.  The kernels of k_solve.hip do
// not (two or three 64 x 64 fp64 tiles in LDS = 68-74 KB, up to 256 registers): queued beside a Gram launch they wait until the
// Gram grid has drained (tools/experiments/README.md, round 2 "tail_stream").  The kernels below do (23 KB of LDS, <= 96
// registers), and tools/coresident_probe.hip measured what such a workgroup gets: it is placed at once, and with s_setprio 3 a
// chain of dependent launches runs at 2.6 x its stand-alone latency while the Gram kernel loses nothing measurable.  The
// chain is latency-bound (18 dependent block steps for the tallest window of chr22, whatever the batch), so latency hidden
// under a 30 ms Gram launch is free: gauss_job_run sends B11's Gram items, B11's epilogue tiles and then this chain ahead,
// and the chain runs on the context's chain queue under the Gram launch of B21's items (gauss_run.cpp:job_queue_run).
//
// Same arithmetic as k_solve.hip, operation for operation -- the two families give the same bits (tests/test_gpu_parity.py)
// -- only the staging differs:
//   * 64 x 64 x 64 products take their operands in K slabs of 16 (two [64][16] slabs = 18 KB instead of two tiles = 68 KB);
//     every accumulator still sums k = 0 .. 63 in ascending order through one MFMA chain;
//   * the tile factorisation keeps the ten lower 16 x 16 blocks of the tile (23 KB) and nothing else: the inverse of the
//     factor is built per wave in registers (wave w owns block column w: every product it needs reads blocks of its own
//     column, which it holds in accumulator layout, and one diagonal block from LDS), where tile_chol_inv_blk parks the
//     blocks in a second LDS tile;
//   * always panel + update launches (the variant in which an update workgroup forms its own panel tiles needs three
//     accumulator sets at once); same bits.
#include "gauss_internal.h"
#include "k_solve_common.h"
#include <algorithm>
#include <cstdlib>

namespace gauss {

constexpr int LK = 16;                 // K slab
constexpr int LDA = LK + 2;            // leading dimension of a [64 rows][16 k] slab (144 B rows: b128-aligned)
constexpr int LDS_V = NR + 2;          // leading dimension of a [16 k][NR columns] slab
constexpr int LDB = 18;                // leading dimension of a 16 x 16 block of the tile factorisation
constexpr int LITE_DBLK = 10 * 16 * LDB;                       // doubles: the ten lower blocks
static const size_t LITE_SMEM = (size_t)(LITE_DBLK + NB) * sizeof(double);      // + s_rinv; the slabs (2 x 64 x 18) alias the blocks
static_assert(2 * NB * LDA <= LITE_DBLK && NB * LDA + LK * LDS_V <= LITE_DBLK, "product slabs alias the block area");
static_assert(NR == 64 && NB == 64, "slab thread maps assume 64-wide tiles");

#define LITE_PRIO() __builtin_amdgcn_s_setprio(3)

struct SlabA { f64x2 v[2]; };          // [64][16] slab: row tid / 4, columns 4 (tid % 4) .. + 3
template <typename P>
__device__ __forceinline__ void slab_a_fetch(SlabA& t, P g, int ld, int s, int tid)
{
    const auto p = g + (size_t)(tid >> 2) * ld + LK * s + 4 * (tid & 3);
    t.v[0] = f64x2{p[0], p[1]};
    t.v[1] = f64x2{p[2], p[3]};
}
__device__ __forceinline__ void slab_a_commit(double* __restrict__ S, const SlabA& t, int tid)
{
    double* q = S + (tid >> 2) * LDA + 4 * (tid & 3);
    *reinterpret_cast<f64x2*>(q) = t.v[0];
    *reinterpret_cast<f64x2*>(q + 2) = t.v[1];
}
struct SlabV { f64x2 v[2]; };          // [16][64] slab: row tid / 16, columns 4 (tid % 16) .. + 3
template <typename P>
__device__ __forceinline__ void slab_v_fetch(SlabV& t, P g, int s, int tid)           // g: a 64 x NR block of V, ld NR
{
    const auto p = g + (size_t)(LK * s + (tid >> 4)) * NR + 4 * (tid & 15);
    t.v[0] = f64x2{p[0], p[1]};
    t.v[1] = f64x2{p[2], p[3]};
}
__device__ __forceinline__ void slab_v_commit(double* __restrict__ S, const SlabV& t, int tid)
{
    double* q = S + (tid >> 4) * LDS_V + 4 * (tid & 15);
    *reinterpret_cast<f64x2*>(q) = t.v[0];
    *reinterpret_cast<f64x2*>(q + 2) = t.v[1];
}

// acc[n] += sign * A B^T, A and B 64 x 64 [row][k] tiles in global memory (mfma_nt of k_solve.hip)
template <bool NEG, typename PA, typename PB>
__device__ __forceinline__ void lite_nt(f64x4 (&acc)[4], PA gA, int lda, PB gB, int ldb, bool same, double* __restrict__ SA,
                                        double* __restrict__ SB, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    SlabA ra, rb;
    slab_a_fetch(ra, gA, lda, 0, tid);
    if (!same) slab_a_fetch(rb, gB, ldb, 0, tid);
    const double* ap = SA + (16 * wave + (lane & 15)) * LDA + (lane >> 4);
    const double* bp = (same ? SA : SB) + (lane & 15) * LDA + (lane >> 4);
    for (int s = 0; s < NB / LK; s++) {
        __syncthreads();                                  // the previous slab is no longer being read
        slab_a_commit(SA, ra, tid);
        if (!same) slab_a_commit(SB, rb, tid);
        __syncthreads();
        if (s + 1 < NB / LK) {
            slab_a_fetch(ra, gA, lda, s + 1, tid);
            if (!same) slab_a_fetch(rb, gB, ldb, s + 1, tid);
        }
#pragma unroll
        for (int k0 = 0; k0 < LK; k0 += 4) {
            double a = ap[k0];
            if (NEG) a = -a;
#pragma unroll
            for (int n = 0; n < 4; n++) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[n * 16 * LDA + k0], acc[n], 0, 0, 0);
        }
    }
}

// acc[n] += sign * A V, A a 64 x 64 [row][k] tile, V a 64 x NR [k][col] block, both in global memory (mfma_nn of k_solve.hip)
template <bool NEG, typename PA, typename PV>
__device__ __forceinline__ void lite_nn(f64x4 (&acc)[SOLVE_NT], PA gA, int lda, PV gV, double* __restrict__ SA, double* __restrict__ SV,
                                        int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    SlabA ra;
    SlabV rv;
    slab_a_fetch(ra, gA, lda, 0, tid);
    slab_v_fetch(rv, gV, 0, tid);
    const double* ap = SA + (16 * wave + (lane & 15)) * LDA + (lane >> 4);
    const double* vp = SV + (lane >> 4) * LDS_V + (lane & 15);
    for (int s = 0; s < NB / LK; s++) {
        __syncthreads();
        slab_a_commit(SA, ra, tid);
        slab_v_commit(SV, rv, tid);
        __syncthreads();
        if (s + 1 < NB / LK) {
            slab_a_fetch(ra, gA, lda, s + 1, tid);
            slab_v_fetch(rv, gV, s + 1, tid);
        }
#pragma unroll
        for (int k0 = 0; k0 < LK; k0 += 4) {
            double a = ap[k0];
            if (NEG) a = -a;
#pragma unroll
            for (int n = 0; n < SOLVE_NT; n++) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, vp[k0 * LDS_V + n * 16], acc[n], 0, 0, 0);
        }
    }
}

// ---- tile factorisation + inverse of the factor on the ten lower 16 x 16 blocks (tile_chol_inv_blk of k_solve.hip) ----
__device__ __forceinline__ int blk_of(int i, int j) { return (i * (i + 1) / 2 + j) * 16 * LDB; }      // i >= j

template <int K>
__device__ __forceinline__ void lite_block_column(double* __restrict__ Dblk, double* __restrict__ s_rinv, int lane, int& bad)
{
    const int bi = lane >> 4;
    const bool live = bi >= K;                       // lanes above the block column have no block: they compute junk on a copy
    double* row = Dblk + blk_of(live ? bi : K, K) + (lane & 15) * LDB;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
        const f64x2 v = *reinterpret_cast<const f64x2*>(row + c);
        a[c] = v[0]; a[c + 1] = v[1];
    }
    chol_pivots<K>(a, s_rinv, lane, bad);
    if (live) {
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            f64x2 v;
            v[0] = (lane >= 16 * K + c) ? a[c] : 0.0;
            v[1] = (lane >= 16 * K + c + 1) ? a[c + 1] : 0.0;
            *reinterpret_cast<f64x2*>(row + c) = v;
        }
    }
}

template <int K>
__device__ __forceinline__ void lite_trailing(double* __restrict__ Dblk, int wave, int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    int t = 0;
#pragma unroll
    for (int j = K + 1; j < 4; j++)
#pragma unroll
        for (int i = j; i < 4; i++) {
            if ((t & 3) == wave) {
                double* C = Dblk + blk_of(i, j);
                const double* Li = Dblk + blk_of(i, K);
                const double* Lj = Dblk + blk_of(j, K);
                f64x4 acc;
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = C[(lk + 4 * r) * LDB + lr];
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Li[lr * LDB + k0 + lk], Lj[lr * LDB + k0 + lk], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) C[(lk + 4 * r) * LDB + lr] = acc[r];
            }
            t++;
        }
}

// tile: the symmetric 64 x 64 tile in accumulator layout (acc_row / acc_col).  Writes the factor L (zeros above the diagonal)
// to Lout (leading dimension ldl) and L^-1 to Xout (64 x 64, contiguous).  Returns (block-uniform) 1 if a pivot was not positive.
template <typename PL, typename PX>
__device__ __forceinline__ int lite_tile_chol_inv(const f64x4 (&tile)[4], double* __restrict__ smem, PL Lout, int ldl, PX Xout, int tid,
                                                  int* s_flag)
{
    double* Dblk = smem;
    double* s_rinv = smem + LITE_DBLK;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    __syncthreads();                                 // the product slabs that alias the blocks are no longer being read
    if (tid == 0) *s_flag = 0;
#pragma unroll
    for (int n = 0; n < 4; n++)
        if (n <= wave) {
#pragma unroll
            for (int r = 0; r < 4; r++) Dblk[blk_of(wave, n) + (lk + 4 * r) * LDB + lr] = tile[n][r];
        }
    __syncthreads();
    int bad = 0;
    if (wave == 0) lite_block_column<0>(Dblk, s_rinv, lane, bad);
    __syncthreads();
    lite_trailing<0>(Dblk, wave, lane);
    __syncthreads();
    if (wave == 0) lite_block_column<1>(Dblk, s_rinv, lane, bad);
    __syncthreads();
    lite_trailing<1>(Dblk, wave, lane);
    __syncthreads();
    if (wave == 0) lite_block_column<2>(Dblk, s_rinv, lane, bad);
    __syncthreads();
    lite_trailing<2>(Dblk, wave, lane);
    __syncthreads();
    if (wave == 0) { lite_block_column<3>(Dblk, s_rinv, lane, bad); if (bad && lane == 0) *s_flag = 1; }
    __syncthreads();
    // the factor leaves for global memory before the inverse overwrites the diagonal blocks
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e >> 6, c = e & 63, i = r >> 4, j = c >> 4;
        Lout[(size_t)r * ldl + c] = (j <= i) ? Dblk[blk_of(i, j) + (r & 15) * LDB + (c & 15)] : 0.0;
    }
    __syncthreads();
    // ---- X_ww = L_ww^-1, in place: lane c (< 16) solves L_ww x = e_c by forward substitution; the L entries are wave-uniform
    double* Dww = Dblk + blk_of(wave, wave);
    {
        const int c = lane & 15, b = 16 * wave;
        double sacc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) sacc[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const double xj = sacc[j] * s_rinv[b + j];
            sacc[j] = xj;
            // one column's reads at a time: left alone the compiler merges the reads of neighbouring columns into b128 loads
            // and gathers the whole block up front (240 registers)
            typedef __attribute__((address_space(3))) const double* lds_cptr;
            lds_cptr Dj = (lds_cptr)(Dww + j);
            asm volatile("" : "+v"(Dj) :: "memory");
#pragma unroll
            for (int i = j + 1; i < 16; i++) sacc[i] = fma(-Dj[i * LDB], xj, sacc[i]);
            // ... and the column's arithmetic tied to this point of the instruction stream (the values pass through an empty
            // volatile asm), or all 120 reads are issued ahead of the first multiply-add
            asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]), "+v"(sacc[2]), "+v"(sacc[3]), "+v"(sacc[4]), "+v"(sacc[5]), "+v"(sacc[6]), "+v"(sacc[7]));
            asm volatile("" : "+v"(sacc[8]), "+v"(sacc[9]), "+v"(sacc[10]), "+v"(sacc[11]), "+v"(sacc[12]), "+v"(sacc[13]), "+v"(sacc[14]), "+v"(sacc[15]));
        }
        WAVE_LDS_SYNC();                             // this wave's reads of L_ww are done (nobody else touches the block)
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; i++) Dww[i * LDB + c] = sacc[i];               // exact zeros above the diagonal
        }
    }
    __syncthreads();
    // ---- block column `wave` of X: X_ww (LDS), then by distance from the diagonal  X_iw = -X_ii * sum_{m = w}^{i-1} L_im X_mw,
    // the X_mw of earlier distances in this wave's registers (accumulator layout = the MFMA's B operand layout)
    f64x4 xs[3];
#pragma unroll
    for (int dist = 1; dist < 4; dist++) {
        const int i = dist + wave;
        if (i < 4) {
            f64x4 s = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int mo = 0; mo < dist; mo++) {
                const double* Lim = Dblk + blk_of(i, wave + mo);
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4) {
                    const double b = (mo == 0) ? Dww[(k0 + lk) * LDB + lr] : xs[mo > 0 ? mo - 1 : 0][k0 >> 2];
                    s = __builtin_amdgcn_mfma_f64_16x16x4f64(Lim[lr * LDB + k0 + lk], b, s, 0, 0, 0);
                }
            }
            const double* Xii = Dblk + blk_of(i, i);
            f64x4 o = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += 4)
                o = __builtin_amdgcn_mfma_f64_16x16x4f64(-Xii[lr * LDB + k0 + lk], s[k0 >> 2], o, 0, 0, 0);
            xs[dist - 1] = o;
        }
    }
    // block column `wave` of X to global memory: zeros above the diagonal block, the diagonal block, the blocks below
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = 0.0;
            if (i == wave) v = Dww[(lk + 4 * r) * LDB + lr];
            else if (i > wave) v = (i - wave == 1) ? xs[0][r] : ((i - wave == 2) ? xs[1][r] : xs[2][r]);
            Xout[(size_t)(16 * i + lk + 4 * r) * NB + 16 * wave + lr] = v;
        }
    __syncthreads();
    return *s_flag;
}

// ---- rows of the inverse (ride_pre / ride_fin of k_solve.hip) --------------------------------------------------------------
__device__ __forceinline__ void lite_ride_products(const Prob& pb, int panel, int r, int j0, int jstep, int jlast, f64x4 (&acc)[SOLVE_NT],
                                                   double* __restrict__ SA, double* __restrict__ SV, int tid)
{
    const int first = inv_first_row(pb, panel);
    if (j0 < first) j0 += (first - j0 + jstep - 1) / jstep * jstep;
    const int ld = pb.Mld;
    const auto Lm = pb.A + (size_t)2 * ld * ld;               // factor of A[0]
    const auto V = pb.V + (size_t)panel * ld * NR;
    for (int jb = j0; jb <= jlast; jb += jstep)
        lite_nn<true>(acc, Lm + (size_t)r * NB * ld + (size_t)jb * NB, ld, V + (size_t)jb * NB * NR, SA, SV, tid);
}

__device__ __forceinline__ void lite_ride_pre(const Prob& pb, int panel, int r, int g, int split, double* __restrict__ smem, int tid)
{
    const int n_early = r - 1 - inv_first_row(pb, panel);     // products j = first .. r - 2
    if (n_early < 1) return;
    const bool cut = split > 0 && n_early >= split;
    if (!cut && g != 0) return;
    double* SA = smem;
    double* SV = smem + NB * LDA;
    f64x4 acc[SOLVE_NT];
#pragma unroll
    for (int n = 0; n < SOLVE_NT; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
    if (cut) lite_ride_products(pb, panel, r, g, SOLVE_SPLIT, r - 2, acc, SA, SV, tid);
    else {
        for (int gg = 0; gg < SOLVE_SPLIT; gg++) {
            f64x4 part[SOLVE_NT];
#pragma unroll
            for (int n = 0; n < SOLVE_NT; n++) part[n] = f64x4{0.0, 0.0, 0.0, 0.0};
            lite_ride_products(pb, panel, r, gg, SOLVE_SPLIT, r - 2, part, SA, SV, tid);
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
        for (int q = 0; q < 4; q++) P[(size_t)(n * 4 + q) * 256 + tid] = acc[n][q];
}

__device__ __forceinline__ void lite_ride_fin(const Prob& pb, int panel, int r, int split, double* __restrict__ smem, int tid)
{
    const int first = inv_first_row(pb, panel);
    if (r < first) return;
    double* SA = smem;
    double* SV = smem + NB * LDA;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15;
    const int n_early = r - 1 - first;
    const int np = n_early < 1 ? 0 : ((split > 0 && n_early >= split) ? SOLVE_SPLIT : 1);
    const int ld = pb.Mld;
    f64x4 acc[SOLVE_NT];
#pragma unroll
    for (int n = 0; n < SOLVE_NT; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int g = 0; g < np; g++) {                             // the parked sums in class order
        const auto P = ride_part(pb, panel, r, g);
#pragma unroll
        for (int n = 0; n < SOLVE_NT; n++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[n][c] += P[(size_t)(n * 4 + c) * 256 + tid];
    }
    const auto V = pb.V + (size_t)panel * ld * NR;
    if (r - 1 >= first)                                        // the last product j = r - 1 on top
        lite_nn<true>(acc, pb.A + (size_t)2 * ld * ld + (size_t)r * NB * ld + (size_t)(r - 1) * NB, ld, V + (size_t)(r - 1) * NB * NR, SA, SV, tid);
    // X = B_r + acc with B = [I | z1] (column g = 64 panel + c is e_g for g < M and z1 for g == M), parked in V_r's own place
    // in global memory (nobody else reads block row r of this panel during this launch; a workgroup's own global writes are
    // visible to it after the barrier), then  V_r = Linv_rr X  as one more slab-staged product
    const auto Vr = V + (size_t)r * NB * NR;
    {
        // the z1 column (g == M) is column zc of this panel, if it lies in it: at most one of a thread's four columns
        const int zc = pb.M - panel * NR;
        const bool has_z = zc >= 0 && zc < NR && (zc & 15) == lr;
        double zv[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int k = r * NB + acc_row(wave, lane, c);
            zv[c] = (has_z && k < pb.M) ? pb.z1[k] : 0.0;
        }
#pragma unroll
        for (int n = 0; n < SOLVE_NT; n++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int row = acc_row(wave, lane, c), col = acc_col(lane, n);
                const int k = r * NB + row, g = panel * NR + col;
                const double b = (g < pb.M) ? ((g == k) ? 1.0 : 0.0) : ((has_z && n == (zc >> 4)) ? zv[c] : 0.0);
                Vr[(size_t)row * NR + col] = b + acc[n][c];
            }
    }
    __syncthreads();
    f64x4 out[SOLVE_NT];
#pragma unroll
    for (int n = 0; n < SOLVE_NT; n++) out[n] = f64x4{0.0, 0.0, 0.0, 0.0};
    lite_nn<false>(out, pb.Linv + (size_t)r * NB * NB, NB, Vr, SA, SV, tid);      // every slab of X has left global memory when it returns
#pragma unroll
    for (int n = 0; n < SOLVE_NT; n++)
#pragma unroll
        for (int c = 0; c < 4; c++) Vr[(size_t)acc_row(wave, lane, c) * NR + acc_col(lane, n)] = out[n][c];
}

// ---- the launches (same grids as factor_init / factor_panel / factor_update / solve_last of k_solve.hip) ----------------
__global__ __launch_bounds__(256, 5) void factor_init_lite_kernel(const Prob* __restrict__ probs)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int s_flag;
    LITE_PRIO();
    const Prob& pb = probs[blockIdx.x >> 1];
    const int mat = blockIdx.x & 1;
    if (pb.ld_only || pb.npanel == 0) return;
    if (mat == 1 && pb.status[3]) return;              // certified: lambda_min(B11) > eps
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ld = pb.Mld;
    const auto W = factor_work(pb, mat);
    f64x4 tile[4];
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) tile[n][r] = W[(size_t)acc_row(wave, lane, r) * ld + acc_col(lane, n)];
    const int fail = lite_tile_chol_inv(tile, smem, pb.A + (size_t)(2 + mat) * ld * ld, ld, pb.Linv + (size_t)mat * pb.nblk * NB * NB, tid, &s_flag);
    if (fail && tid == 0) pb.status[mat] = 1;
}

// panel(s): grid.x = max_nblk - 1 - s (block row k = s + 1 + x), grid.y = problem * 2 + matrix
__global__ __launch_bounds__(256, 5) void factor_panel_lite_kernel(const Prob* __restrict__ probs, int s)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    LITE_PRIO();
    const Prob& pb = probs[blockIdx.y >> 1];
    const int mat = blockIdx.y & 1;
    if (pb.ld_only || pb.npanel == 0) return;
    if (mat == 1 && pb.status[3]) return;
    const int nb = pb.nblk;
    const int k = s + 1 + (int)blockIdx.x;
    if (k >= nb) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = pb.Mld;
    const auto W = factor_work(pb, mat);
    const auto Lm = pb.A + (size_t)(2 + mat) * ld * ld;
    f64x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
    // L[k][s] = W[k][s] * Linv_ss^T
    lite_nt<false>(acc, W + (size_t)k * NB * ld + (size_t)s * NB, ld, pb.Linv + ((size_t)mat * nb + s) * NB * NB, NB, false,
                   smem, smem + NB * LDA, tid);
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            Lm[(size_t)(k * NB + acc_row(wave, lane, r)) * ld + s * NB + acc_col(lane, n)] = acc[n][r];
}

// update(s): grid.x = n_tri + n_ride; x = 0 is tile (s+1, s+1), x = 1 .. n_ride the riding rows, then the trailing tiles
__global__ __launch_bounds__(256, 5) void factor_update_lite_kernel(const Prob* __restrict__ probs, int s, int T, int n_tri, int split, int n_ride)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int s_flag;
    LITE_PRIO();
    const Prob& pb = probs[blockIdx.y >> 1];
    const int mat = blockIdx.y & 1;
    if (pb.ld_only || pb.npanel == 0) return;
    if ((int)blockIdx.x >= 1 && (int)blockIdx.x <= n_ride) {
        const int idx = (int)blockIdx.x - 1;
        const int panel = idx / (SOLVE_SPLIT + 1), g = idx % (SOLVE_SPLIT + 1);
        if (mat != 0 || panel >= pb.npi) return;
        if (g == SOLVE_SPLIT) { if (s < pb.nblk) lite_ride_fin(pb, panel, s, split, smem, threadIdx.x); }
        else if (s + 1 < pb.nblk) lite_ride_pre(pb, panel, s + 1, g, split, smem, threadIdx.x);
        return;
    }
    if (mat == 1 && pb.status[3]) return;
    const int nb = pb.nblk;
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
    f64x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            acc[n][r] = W[(size_t)(k * NB + acc_row(wave, lane, r)) * ld + j * NB + acc_col(lane, n)];
    // W[k][j] -= L[k][s] L[j][s]^T
    lite_nt<true>(acc, Lm + (size_t)k * NB * ld + (size_t)s * NB, ld, Lm + (size_t)j * NB * ld + (size_t)s * NB, ld, k == j,
                  smem, smem + NB * LDA, tid);
    if (!next_diag) {
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                W[(size_t)(k * NB + acc_row(wave, lane, r)) * ld + j * NB + acc_col(lane, n)] = acc[n][r];
        return;
    }
    const int fail = lite_tile_chol_inv(acc, smem, Lm + (size_t)k * NB * ld + (size_t)k * NB, ld, pb.Linv + ((size_t)mat * nb + k) * NB * NB, tid,
                                        &s_flag);
    if (fail && tid == 0) pb.status[mat] = 1;
}

__global__ __launch_bounds__(256, 5) void solve_last_lite_kernel(const Prob* __restrict__ probs, const int2* __restrict__ panelmap, int s_last, int split)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    LITE_PRIO();
    const int2 pm = panelmap[blockIdx.x];
    const Prob& pb = probs[pm.x];
    if (pb.nblk - 1 == s_last) lite_ride_fin(pb, pm.y, s_last, split, smem, threadIdx.x);
}

// The launches of launch_factor_step(step) in their small-footprint form (always panel + update; max_npanel > 0: the rows of
// the inverse ride along), and of launch_solve_last.
void launch_factor_step_lite(const Prob* d_probs, int n_prob, int step, int max_nblk, int max_npanel, int split, hipStream_t st)
{
    if (n_prob <= 0 || step >= max_nblk) return;
    if (step == 0) {
        hipLaunchKernelGGL(factor_init_lite_kernel, dim3(n_prob * 2), dim3(256), LITE_SMEM, st, d_probs);
        return;
    }
    const int s = step - 1;
    const int T = max_nblk - 1 - s;
    if (T <= 0) return;
    const int n_tri = T * (T + 1) / 2;
    const int n_ride = max_npanel > 0 ? max_npanel * (SOLVE_SPLIT + 1) : 0;
    hipLaunchKernelGGL(factor_panel_lite_kernel, dim3(T, n_prob * 2), dim3(256), LITE_SMEM, st, d_probs, s);
    hipLaunchKernelGGL(factor_update_lite_kernel, dim3(n_tri + n_ride, n_prob * 2), dim3(256), LITE_SMEM, st, d_probs, s, T, n_tri, split, n_ride);
}

void launch_solve_last_lite(const Prob* d_probs, const int2* d_panelmap, int n_panels, int max_nblk, int split, hipStream_t s)
{
    if (n_panels <= 0 || max_nblk < 1) return;
    hipLaunchKernelGGL(solve_last_lite_kernel, dim3(n_panels), dim3(256), LITE_SMEM, s, d_probs, d_panelmap, max_nblk - 1, split);
}

}  // namespace gauss
