// Pieces shared by the factorisation / inverse kernels of k_solve.hip and their small-footprint twins in k_solve_lite.hip:
// the two families must produce the same bits, so everything that is arithmetic or data layout in global memory lives here.
#pragma once
#include "gauss_internal.h"

namespace gauss {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int LDT = NB + 2;    // LDS leading dimension of a 64 x 64 [row][k] tile: 66 doubles = 528 B;
                               // 528 mod 256 = 16 puts the 32 lanes of a ds_read_b64 group on distinct banks
constexpr int LDV = NR + 2;     // LDS leading dimension of the 64 x NR [k][col] tile of the solve

#define WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

constexpr int SOLVE_NT = NR / 16;              // accumulator tiles (16 columns each) per wave
constexpr int SOLVE_SPLIT = 4;                 // interleaved classes of a row's products (solve_row, ride_pre)

#if defined(__HIPCC__)
// accumulator tile element (reg r of tile n) -> (row, col) inside the 64 x (16 NT) block
__device__ __forceinline__ int acc_row(int wave, int lane, int r) { return 16 * wave + (lane >> 4) + 4 * r; }
__device__ __forceinline__ int acc_col(int lane, int n) { return 16 * n + (lane & 15); }

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// The 16 pivot steps of block column K of a 64 x 64 tile factorisation, one wave, lane = tile row, a[c] = the lane's entry
// of column 16 K + c: pivot by v_readlane, 1 / sqrt by v_rsq_f64 + Newton, column scale, rank-1 update of the block
// column's remaining columns with the multipliers fetched by v_readlane.  `bad` is wave-uniform.  Lanes above the block
// column (rows < 16 K) compute junk nobody reads.
template <int K>
__device__ __forceinline__ void chol_pivots(double (&a)[16], double* __restrict__ s_rinv, int lane, int& bad)
{
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int jj = 16 * K + j;
        const double piv = readlane_f64(a[j], jj);
        if (!(piv > 0.0)) bad = 1;
        double r = __builtin_amdgcn_rsq(piv);
        const double h = 0.5 * piv;
        r = fma(r, fma(-h * r, r, 0.5), r);
        r = fma(r, fma(-h * r, r, 0.5), r);
        double d = piv * r;
        d = fma(fma(-d, d, piv), 0.5 * r, d);
        const double l = (lane == jj) ? d : a[j] * r;        // rows above jj hold junk that is never stored
        a[j] = l;
        if (lane == jj) s_rinv[jj] = r;                      // 1 / L[jj][jj]
#pragma unroll
        for (int c = j + 1; c < 16; c++) {
            const double lc = readlane_f64(l, 16 * K + c);   // L[16K + c][jj], wave-uniform
            a[c] = fma(-l, lc, a[c]);
        }
    }
}

// Layout of the factor workspace of one problem: pb.A = [A0 | A1 | L0 | L1 | W0], each Mld x Mld row-major; matrix 0 is
// factored in W0 (A0 stays intact), matrix 1 = B11 - eps I in place in A1.
__device__ __forceinline__ GP(double) factor_work(const Prob& pb, int mat)
{
    const size_t ld2 = (size_t)pb.Mld * pb.Mld;
    return mat == 0 ? pb.A + 4 * ld2 : pb.A + ld2;
}

// Fused path: column g < M of the right-hand side [I | z1] is e_g, so block (kb, panel) of X is a structural zero for
// kb < panel and is never computed, stored or read; the panel that holds column M (z1) is dense.
__device__ __forceinline__ int inv_first_row(const Prob& pb, int panel) { return panel == pb.M / NR ? 0 : panel; }

// parked partial sums of a riding row (ride_pre -> ride_fin), double buffered by row parity
__device__ __forceinline__ GP(double) ride_part(const Prob& pb, int panel, int r, int g)
{
    return pb.Part + (((size_t)(r & 1) * pb.npi + panel) * SOLVE_SPLIT + g) * (NB * NR);
}
#endif

}  // namespace gauss
