// Pieces shared by the Gram kernels (k_gram.hip: e4m3 / int8 operands, k_gram_fp6.hip: e2m3 operands): how a wave's
// exact partial sums leave for the partial slabs.
#pragma once
#include "gauss_internal.h"

namespace gauss {

__device__ __forceinline__ float slab_bits(float v) { return v; }
__device__ __forceinline__ float slab_bits(int v) { return __int_as_float(v); }
// 16-bit slabs: the exact sum as an unsigned integer; rows 2k and 2k + 1 of a column share one dword
__device__ __forceinline__ uint32_t slab_u(float v) { return (uint32_t)v; }
__device__ __forceinline__ uint32_t slab_u(int v) { return (uint32_t)v; }
__device__ __forceinline__ float slab_pair(uint32_t lo, uint32_t hi) { return __uint_as_float((lo & 0xFFFFu) | (hi << 16)); }

// End of a K segment: the wave's four 32 x 32 accumulators go to that segment's slab.
// C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// obase / obase16: the lane's first entry in a slab of f32 (int32) values / of uint16 row pairs.
template <int NA, int NB, bool SK10, typename OUT, typename ACC>
__device__ __forceinline__ void flush_acc(OUT out, bool slab16, int obase, int obase16, const ACC& acc00, const ACC& acc01,
                                          const ACC& acc10, const ACC& acc11)
{
    if (slab16) {
        // accumulator registers r and r + 1 (r even) are rows 2k and 2k + 1 of the same column: one dword
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int o = obase16 + (((r & 3) + 8 * (r >> 2)) >> 1) * TILE;
            out[o] = slab_pair(slab_u(acc00[r]), slab_u(acc00[r + 1]));
            if (NB > 1) out[o + 32] = slab_pair(slab_u(acc01[r]), slab_u(acc01[r + 1]));
            if (NA > 1) {
                if (!SK10) out[o + 16 * TILE] = slab_pair(slab_u(acc10[r]), slab_u(acc10[r + 1]));
                if (NB > 1) out[o + 16 * TILE + 32] = slab_pair(slab_u(acc11[r]), slab_u(acc11[r + 1]));
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = obase + ((r & 3) + 8 * (r >> 2)) * TILE;
            out[o] = slab_bits(acc00[r]);
            if (NB > 1) out[o + 32] = slab_bits(acc01[r]);
            if (NA > 1) {
                if (!SK10) out[o + 32 * TILE] = slab_bits(acc10[r]);
                if (NB > 1) out[o + 32 * TILE + 32] = slab_bits(acc11[r]);
            }
        }
    }
}

}  // namespace gauss
