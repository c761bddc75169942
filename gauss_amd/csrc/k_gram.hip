// K2/K3: the LD GEMM.  Exact integer Gram partials  S_xy = sum_n x_i[n] * x_j[n]  on the
// fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the `sumxy` accumulation of CalCor / CalWgtCov (util.cpp:62, util.cpp:114), which
// the reference re-runs for every SNP pair (dist.cpp:174,189; distmix.cpp:195,213;
// computeLD.cpp:111; gene.cpp:309,581).  Here all pairs of a 128 x 128 tile are formed at once.
//
// Exactness: operands are genotype codes (0..15 admitted, 0..2 in practice); a segment (the run of samples whose
// products one accumulator sums before it is flushed) never spans more than 8192 samples of ONE population -- the
// planner cuts longer populations, at SEG_MAX = 2048 for jobs of fewer than four windows (more work items) and at
// 4096 otherwise (gauss_plan.cpp:seg_max_for; any cap up to 8192 is admissible) -- so every partial sum is an integer
// <= 15 * 15 * 8192 < 2^24 and the
// f32 MFMA accumulation (bitwise a k-ordered fmaf chain) is exact in any summation order.  The k order inside a
// chunk is therefore permuted freely to make the LDS reads wide.
//
// Work item = (tile pair, run of K segments).  Workgroup = 256 threads = 4 waves, each wave owns a
// 64 x 64 sub-tile = 2 x 2 MFMA tiles of 32 x 32 (four independent accumulators keep the MFMA pipe
// issuing back to back).  K is walked in chunks of KC = 64 bytes staged by LDS-DMA
// (global_load_lds_dwordx4) into a ring of NS chunk images (NS = 2: one chunk in flight while the previous
// one is multiplied; the ring depth is a template parameter, deeper rings brought nothing).
//
// LDS image: [128 rows][64 bytes] per operand, unpadded; the bank conflicts of the ds_read_b128 fragment
// reads are avoided by an XOR swizzle applied to the DMA source address and to the read address.
#include "gauss_internal.h"
#include "k_gram_common.h"
#include <cstdlib>
#include <type_traits>

#ifndef GAUSS_GRAM_EDGE16
#define GAUSS_GRAM_EDGE16 1
#endif

namespace gauss {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int LROW = 64;                   // LDS row stride in bytes (unpadded; conflict-free through the XOR swizzle)
constexpr int LTILE = TILE * LROW;         // bytes per operand tile image

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One K chunk (64 samples) of a wave's 64 x 64 sub-tile.  NA / NB = number of live 32-row halves
// of the wave's A / B rows (rows past the problem's real row count are zero padding: their
// products are never read, so their MFMAs are not issued at all).
// Operand bytes are genotype codes in OCP e4m3 (0 -> 0x00, 1 -> 0x38, 2 -> 0x40, written by
// pack_stats_kernel): v_cvt_pk_f32_fp8 expands TWO of them per VALU instruction, exactly.  VALU
// issue is not free next to the fp32 MFMA (each VALU op costs about 6 cycles of matrix-pipe time on
// gfx950, measured), so halving the converts is worth ~5 % of the kernel.
// SK10: the wave sits on the diagonal of a diagonal tile (its 64 x 64 block is symmetric): the lower-left 32 x 32
// sub-block mirrors the upper-right one, no reader of the slab looks at it, its MFMAs are not issued.
template <int NA, int NB, bool SK10>
__device__ __forceinline__ void chunk_mfma(const uint8_t* __restrict__ la, const uint8_t* __restrict__ lb,
                                           const int (&aoff)[2], const int (&boff)[2], f32x16& acc00, f32x16& acc01,
                                           f32x16& acc10, f32x16& acc11, int nl)
{
#define GAUSS_MFMA_PAIR(AW0, AW1, BW0, BW1, HI)                                                       \
    {                                                                                                  \
        const f32x2 fa0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)(AW0), HI);                             \
        const f32x2 fb0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)(BW0), HI);                             \
        f32x2 fa1 = fa0, fb1 = fb0;                                                                    \
        if (NA > 1) fa1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)(AW1), HI);                             \
        if (NB > 1) fb1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)(BW1), HI);                             \
        _Pragma("unroll") for (int e = 0; e < 2; e++) {                                              \
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[e], fb0[e], acc00, 0, 0, 0);              \
            if (NB > 1) acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[e], fb1[e], acc01, 0, 0, 0);  \
            if (NA > 1) {                                                                              \
                if (!SK10) acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[e], fb0[e], acc10, 0, 0, 0); \
                if (NB > 1) acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[e], fb1[e], acc11, 0, 0, 0); \
            }                                                                                          \
        }                                                                                              \
    }
#pragma unroll
    for (int g = 0; g < 2; g++) {
        if (4 * g >= nl) break;            // units 4 g .. of the chunk are padding zeros (wave-uniform; nl = live units of 8 samples)
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(la + aoff[g]);
        const u32x4 b0 = *reinterpret_cast<const u32x4*>(lb + boff[g]);
        u32x4 a1 = a0, b1 = b0;
        if (NA > 1) a1 = *reinterpret_cast<const u32x4*>(la + aoff[g] + 32 * LROW);
        if (NB > 1) b1 = *reinterpret_cast<const u32x4*>(lb + boff[g] + 32 * LROW);
        const uint32_t aw0[4] = {a0.x, a0.y, a0.z, a0.w};
        const uint32_t aw1[4] = {a1.x, a1.y, a1.z, a1.w};
        const uint32_t bw0[4] = {b0.x, b0.y, b0.z, b0.w};
        const uint32_t bw1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (4 * g + q >= nl) break;    // dword q of both lane halves = unit 4 g + q (gauss_internal.h: chunk_unit_layout)
            GAUSS_MFMA_PAIR(aw0[q], aw1[q], bw0[q], bw1[q], false)
            GAUSS_MFMA_PAIR(aw0[q], aw1[q], bw0[q], bw1[q], true)
        }
    }
#undef GAUSS_MFMA_PAIR
}

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// The same chunk on the int8 matrix cores (v_mfma_i32_32x32x32_i8): operands are the raw genotype
// codes, 16 bytes per lane and MFMA, sums are exact int32.  The 16 bytes a lane reads for A and for B
// cover the same k positions, so the products line up whatever the instruction's internal k order is.
template <int NA, int NB, bool SK10>
__device__ __forceinline__ void chunk_mfma(const uint8_t* __restrict__ la, const uint8_t* __restrict__ lb,
                                           const int (&aoff)[2], const int (&boff)[2], i32x16& acc00, i32x16& acc01,
                                           i32x16& acc10, i32x16& acc11, int nl)
{
#pragma unroll
    for (int g = 0; g < 2; g++) {
        if (4 * g >= nl) break;            // (the instruction takes a lane's 16 bytes whole: only a dead group can be skipped)
        const i32x4 a0 = *reinterpret_cast<const i32x4*>(la + aoff[g]);
        const i32x4 b0 = *reinterpret_cast<const i32x4*>(lb + boff[g]);
        acc00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc00, 0, 0, 0);
        if (NB > 1) {
            const i32x4 b1 = *reinterpret_cast<const i32x4*>(lb + boff[g] + 32 * LROW);
            acc01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc01, 0, 0, 0);
            if (NA > 1) {
                const i32x4 a1 = *reinterpret_cast<const i32x4*>(la + aoff[g] + 32 * LROW);
                if (!SK10) acc10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc11, 0, 0, 0);
            }
        } else if (NA > 1) {
            const i32x4 a1 = *reinterpret_cast<const i32x4*>(la + aoff[g] + 32 * LROW);
            acc10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc10, 0, 0, 0);
        }
    }
}

// ---- 16-column edge (f32 path) -------------------------------------------------------------------------------------
// A tile's live B rows (= output columns: a window's measured SNPs) are rounded up to 32 by the 32 x 32 MFMA; on the real
// chr22 windows that rounding is 2 % of the issued flops.  A wave whose last live 32-column half holds at most 16 live
// columns -- 1 or 3 live groups of 16 among its 64 columns -- multiplies all its columns with v_mfma_f32_16x16x4_f32
// instead: NC groups of 16 columns against the wave's 2 NA groups of 16 A rows, the same matrix-pipe time per flop and no
// work on the dead group.  Operand layout of that instruction (tools/mfma16_layout_probe.hip):
// lane l supplies A[row l % 16][k = l / 16] and B[k = l / 16][column l % 16] and holds D[row 4 (l / 16) + reg][column l % 16].
// Lane (r, q) therefore reads piece q (16 samples) of row r of the chunk image -- one ds_read_b128 per 16-row group covers
// the chunk's 64 samples across the four lane groups -- and MFMA j of the chunk consumes byte j of every lane's 16.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NA, int NC>
__device__ __forceinline__ void chunk_mfma_edge(const uint8_t* __restrict__ la, const uint8_t* __restrict__ lb, int eoffa,
                                                int eoffb, f32x4 (&acce)[4][NC])
{
    // (the B words are converted again for every A group: keeping the converted pairs would take the kernel past the 104
    // registers that leave room for a fifth, small-footprint workgroup per CU -- k_solve_lite.hip)
    uint32_t b[NC][4];
#pragma unroll
    for (int nc = 0; nc < NC; nc++) {
        const u32x4 bw = *reinterpret_cast<const u32x4*>(lb + eoffb + nc * 16 * LROW);
        b[nc][0] = bw.x; b[nc][1] = bw.y; b[nc][2] = bw.z; b[nc][3] = bw.w;
    }
#pragma unroll
    for (int ga = 0; ga < 2 * NA; ga++) {
        const u32x4 aw = *reinterpret_cast<const u32x4*>(la + eoffa + ga * 16 * LROW);      // the next 16 rows: same swizzle
        const uint32_t a[4] = {aw.x, aw.y, aw.z, aw.w};
        if (NC > 1) {
            // the B words pass through an empty asm per A group, or the compiler recognises the conversions of one group in the
            // next (common subexpressions) and keeps all 8 NC converted pairs alive: 48 registers at NC = 3
#pragma unroll
            for (int nc = 0; nc < NC; nc++) asm volatile("" : "+v"(b[nc][0]), "+v"(b[nc][1]), "+v"(b[nc][2]), "+v"(b[nc][3]));
        }
#pragma unroll
        for (int w = 0; w < 4; w++) {
#pragma unroll
            for (int hi = 0; hi < 2; hi++) {
                const f32x2 fa = hi ? __builtin_amdgcn_cvt_pk_f32_fp8((int)a[w], true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)a[w], false);
#pragma unroll
                for (int nc = 0; nc < NC; nc++) {
                    const f32x2 fb = hi ? __builtin_amdgcn_cvt_pk_f32_fp8((int)b[nc][w], true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)b[nc][w], false);
#pragma unroll
                    for (int e = 0; e < 2; e++) acce[ga][nc] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], fb[e], acce[ga][nc], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);       // one word's conversions at a time (left alone the compiler converts everything first)
        }
    }
}

// the edge accumulators of a segment to its slab: rows erow0 + 16 ga + reg (erow0 = wr * 64 + 4 (lane / 16)), columns
// ecol0 + 16 nc (ecol0 = wc * 64 + lane % 16)
template <int NA, int NC, typename OUT>
__device__ __forceinline__ void flush_edge(OUT out, bool slab16, int erow0, int ecol0, const f32x4 (&acce)[4][NC])
{
#pragma unroll
    for (int ga = 0; ga < 2 * NA; ga++)
#pragma unroll
        for (int nc = 0; nc < NC; nc++) {
            const int row = erow0 + 16 * ga, col = ecol0 + 16 * nc;
            if (slab16) {
                out[(row >> 1) * TILE + col] = slab_pair(slab_u(acce[ga][nc][0]), slab_u(acce[ga][nc][1]));
                out[((row >> 1) + 1) * TILE + col] = slab_pair(slab_u(acce[ga][nc][2]), slab_u(acce[ga][nc][3]));
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) out[(row + r) * TILE + col] = slab_bits(acce[ga][nc][r]);
            }
        }
}

// One work item for a wave with NA x NB live 32-row halves (NA = 0: staging and barriers only).
// The K loop runs over the whole run of segments without draining the prefetch pipeline; at each
// segment end the accumulators are flushed to that segment's slab and cleared.
// s_waitcnt vmcnt(N) + workgroup barrier.  Vector memory operations complete in issue order, so "at most N
// outstanding" means every DMA group but the newest N / 4 has landed in LDS (stores issued later, e.g. a slab
// flush, only make the wait longer, never shorter).  lgkmcnt(0): this wave's LDS reads of the image about to be
// overwritten are done.  The "memory" clobber keeps the compiler from moving LDS traffic across it.
template <int N>
__device__ __forceinline__ void wait_dma_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
template <int NS>
__device__ __forceinline__ void wait_dma_barrier_dyn(int groups_newer)
{
    if (NS <= 2 || groups_newer <= 0) wait_dma_barrier<0>();
    else if (groups_newer == 1) wait_dma_barrier<4>();
    else wait_dma_barrier<8>();
}

// NC > 0 (f32 only, then NB = 0): the wave's live B rows are NC groups of 16 with the last one at most half a 32-row half --
// 1 or 3 groups -- and all of them go through chunk_mfma_edge.
template <int NA, int NB, typename ACC, int NS, bool SK10 = false, int NC = 0>
__device__ __forceinline__ void run_item(const Item& it, uint8_t* lds, int wr, int wc)
{
    const int Kp = it.Kp;
    const auto Ag = it.a;
    const auto Bg = it.b;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;

    // Staging by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes land contiguously in LDS, no
    // VGPR round trip, no ds_write).  One wave-instruction carries 16 rows x 64 bytes of a tile chunk;
    // each wave issues two per operand.  The image has unpadded 64-byte rows; bank conflicts of the
    // ds_read_b128 fragment reads are avoided by an XOR swizzle applied to the SOURCE address here and
    // to the read address below (the same involution): piece c of row r sits at slot c ^ ((r >> 2) & 3).
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int srow = lane >> 2;                                   // row inside the 16-row group
    const int spiece = (lane & 3) ^ ((lane >> 4) & 3);            // swizzled source piece
    size_t gsrc[2];
#pragma unroll
    for (int j = 0; j < 2; j++) gsrc[j] = (size_t)(16 * (wave * 2 + j) + srow) * Kp + 16 * spiece;
    typedef __attribute__((address_space(3))) uint8_t* lds_ptr;
    typedef __attribute__((address_space(1))) const uint8_t* glb_ptr;

    ACC acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};

    const int k0 = it.k0;
    const int nseg = it.nseg;
    // descriptor fields as values: the asm waits in the K loop clobber memory, a field read through `it` would be
    // fetched again every chunk
    const auto seg_k1 = it.seg_k1;
    const auto live_tab = it.chunk_live;
    const int klast = uniform_load<int>(seg_k1, nseg - 1);
    auto out = it.slab;
    const int obase = (wr * 64 + 4 * lh) * TILE + wc * 64 + li;
    const int obase16 = (wr * 32 + 2 * lh) * TILE + wc * 64 + li;         // row pair (wr * 64 + 4 lh) / 2

#define GAUSS_STAGE(BUF, KOFF)                                                                                   \
    {                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; j++) {                                                        \
            const int t = wave * 2 + j;                                                                          \
            __builtin_amdgcn_global_load_lds((glb_ptr)Ag + gsrc[j] + (KOFF), (lds_ptr)(lds + (BUF) * 2 * LTILE + t * 1024), 16, 0, 0);          \
            __builtin_amdgcn_global_load_lds((glb_ptr)Bg + gsrc[j] + (KOFF), (lds_ptr)(lds + (BUF) * 2 * LTILE + LTILE + t * 1024), 16, 0, 0);  \
        }                                                                                                        \
    }
    // ring of NS chunk images: up to NS - 1 chunks are in flight beyond the one being multiplied
    int cur = 0;
    int kpre = k0;                                                // next chunk to request
    int ahead = 0;                                                // chunks requested and not yet consumed
#pragma unroll
    for (int d = 0; d < NS - 1; d++)
        if (kpre < klast) { GAUSS_STAGE(d, kpre) kpre += KC; ahead++; }
    wait_dma_barrier_dyn<NS>(ahead - 1);                          // the first chunk has landed (all waves)
    ahead--;

    // fragment read offsets: logical piece 2g + lh of rows wr*64 + li (+32 has the same swizzle)
    const int sw = (li >> 2) & 3;
    int aoff[2], boff[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        aoff[g] = (wr * 64 + li) * LROW + 16 * ((2 * g + lh) ^ sw);
        boff[g] = (wc * 64 + li) * LROW + 16 * ((2 * g + lh) ^ sw);
    }

    // 16-column edge: fragment offsets in the (row lane % 16, piece lane / 16) layout, and where its results go
    constexpr bool BE = NC > 0;
    f32x4 acce[4][BE ? NC : 1];
    int eoffa = 0, eoffb = 0, erow0 = 0, ecol0 = 0;
    if (BE) {
        const int r16 = lane & 15, q16 = lane >> 4, sw16 = (r16 >> 2) & 3;
#pragma unroll
        for (int ga = 0; ga < 4; ga++)
#pragma unroll
            for (int nc = 0; nc < (BE ? NC : 1); nc++) acce[ga][nc] = f32x4{0.f, 0.f, 0.f, 0.f};
        eoffa = (wr * 64 + r16) * LROW + 16 * (q16 ^ sw16);
        eoffb = (wc * 64 + r16) * LROW + 16 * (q16 ^ sw16);
        erow0 = wr * 64 + 4 * q16;
        ecol0 = wc * 64 + r16;
    }

    int k = k0;
    uint32_t lbits = uniform_load<uint32_t>(live_tab, k0 >> 9);         // live units per chunk, 8 chunks per dword
    int nl_next = chunk_live_units(lbits, k0 >> 6);
    for (int seg = 0; seg < nseg; seg++) {
        const int kend = uniform_load<int>(seg_k1, seg);
        for (; k < kend; k += KC) {
            // the image consumed in the previous iteration is free: request the chunk NS - 1 ahead into it
            // (the prefetch runs across segment ends)
            if (kpre < klast) {
                int nb_ = cur + NS - 1;
                if (nb_ >= NS) nb_ -= NS;
                GAUSS_STAGE(nb_, kpre)
                kpre += KC;
                ahead++;
            }
            const uint8_t* la = lds + cur * 2 * LTILE;
            const uint8_t* lb = la + LTILE;
            const int nl = nl_next;
            if (k + KC < klast) {
                const int c1 = (k + KC) >> 6;
                if ((c1 & 7) == 0) lbits = uniform_load<uint32_t>(live_tab, c1 >> 3);        // scalar load, once per 8 chunks
                nl_next = chunk_live_units(lbits, c1);
            }
            if (NA > 0 && NB > 0) chunk_mfma<NA, (NB > 0 ? NB : 1), SK10>(la, lb, aoff, boff, acc00, acc01, acc10, acc11, nl);
            if (BE) chunk_mfma_edge<(NA > 0 ? NA : 1), (BE ? NC : 1)>(la, lb, eoffa, eoffb, acce);
            wait_dma_barrier_dyn<NS>(ahead - 1);         // next chunk landed; everyone is done reading this one
            if (ahead > 0) ahead--;
            cur = (cur + 1 == NS) ? 0 : cur + 1;
        }
#undef GAUSS_STAGE
        // end of a segment: flush its exact partial sums, start the next segment from zero.
        // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        if (NA > 0 && NB > 0) {
            flush_acc<NA, (NB > 0 ? NB : 1), SK10>(out, (it.flags & 2) != 0, obase, obase16, acc00, acc01, acc10, acc11);
#pragma unroll
            for (int r = 0; r < 16; r++) { acc00[r] = 0; acc01[r] = 0; acc10[r] = 0; acc11[r] = 0; }
        }
        if (BE) {
            // (the store addresses are formed here, not ahead of the K loop: twelve 64-bit pointers held across it would spill)
            int er = erow0, ec = ecol0;
            asm volatile("" : "+v"(er), "+v"(ec));
            flush_edge<(NA > 0 ? NA : 1), (BE ? NC : 1)>(out, (it.flags & 2) != 0, er, ec, acce);
#pragma unroll
            for (int ga = 0; ga < 4; ga++)
#pragma unroll
                for (int nc = 0; nc < (BE ? NC : 1); nc++) acce[ga][nc] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        out += (it.flags & 2) ? TILE * TILE / 2 : TILE * TILE;
    }
}

// 104 registers at most: four workgroups per CU then leave 96 per lane for the small-footprint kernels that run beside this one
// (k_solve_lite.hip); the common paths need 104, the 16-column edge path would take 107 if left alone.
template <typename ACC, int NS>
__device__ __forceinline__ void gram_item(const Item& it, uint8_t* lds)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // live 32-row halves of this wave's A rows / B rows (wave-uniform)
    int na = (it.rows_a - wr * 64 + 31) / 32;
    int nb = (it.rows_b - wc * 64 + 31) / 32;
    na = na < 0 ? 0 : (na > 2 ? 2 : na);
    nb = nb < 0 ? 0 : (nb > 2 ? 2 : nb);
    // diagonal tile: the lower-left 64 x 64 quadrant mirrors the upper-right one and is never read
    const bool diag = (it.flags & 1) != 0;
    if (diag && wr == 1 && wc == 0) na = 0;
    // f32 path: a last half of at most 16 live B rows goes to the 16-column edge routine (GAUSS_GRAM_EDGE16 compiled in)
    if (std::is_same<ACC, f32x16>::value && GAUSS_GRAM_EDGE16 && na > 0) {
        int nb16 = (it.rows_b - wc * 64 + 15) / 16;
        nb16 = nb16 < 0 ? 0 : (nb16 > 4 ? 4 : nb16);
        if (nb16 & 1) {
            if (nb16 == 3) { if (na == 2) run_item<2, 0, ACC, NS, false, 3>(it, lds, wr, wc); else run_item<1, 0, ACC, NS, false, 3>(it, lds, wr, wc); }
            else { if (na == 2) run_item<2, 0, ACC, NS, false, 1>(it, lds, wr, wc); else run_item<1, 0, ACC, NS, false, 1>(it, lds, wr, wc); }
            return;
        }
    }
    if (na == 0 || nb == 0) run_item<0, 0, ACC, NS>(it, lds, wr, wc);
    else if (na == 2 && nb == 2 && diag && wr == wc) run_item<2, 2, ACC, NS, true>(it, lds, wr, wc);
    else if (na == 2 && nb == 2) run_item<2, 2, ACC, NS>(it, lds, wr, wc);
    else if (na == 2) run_item<2, 1, ACC, NS>(it, lds, wr, wc);
    else if (nb == 2) run_item<1, 2, ACC, NS>(it, lds, wr, wc);
    else run_item<1, 1, ACC, NS>(it, lds, wr, wc);
}

// 104 registers at most: four workgroups per CU then leave 96 per lane for the small-footprint kernels that run beside this one
// (k_solve_lite.hip); the common paths need 104, the 16-column edge path would take 107 if left alone.
//
// `b11_done` (may be null): items with flag bit 4 -- B11's tile pairs of a job whose factorisation chain runs beside this launch
// (gauss_run.cpp:job_queue_run) -- and items with flag bit 5 -- B21's tile pairs of the windows whose epilogue tiles fill this launch's
// last round -- count themselves off there (counters [0] and [8]) when their slabs are out, so that the chain queue can start on B11 while
// this SAME launch goes on with B21's items: no second launch, no drained chip between the two.  The hand-off follows the
// producer recipe for data another kernel reads (MI355X_MICROARCH.md, inter-workgroup visibility): every storing wave waits for
// its stores, the workgroup meets at a barrier, one lane releases at agent scope (writes the XCD L2's dirty lines back), waits
// for that, and only then adds to the counter; the reader is a LATER KERNEL on the chain queue (its launch is the acquire),
// started by wait_count_kernel below.
template <typename ACC, int NS, int OCC>
__global__ __launch_bounds__(256, OCC) __attribute__((amdgpu_num_vgpr(52))) void gram_kernel(const Item* __restrict__ items,
                                                                                             unsigned long long* __restrict__ b11_done)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[NS * 2 * LTILE];

    const Item& it = items[blockIdx.x];
    const int counted = it.flags & 48;                                // scalar: read before the K loop's memory clobbers
    gram_item<ACC, NS>(it, lds);
    if (counted && b11_done) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's slab stores have reached the L2
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // buffer_wbl2 sc1: visible to the other XCDs
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the compiler may drop the fence's own wait: guide, compiler hazard)
            // counter 0: B11's items (the chain queue waits for them); counter 1, a cache line on: the early windows' B21 items
            __hip_atomic_fetch_add(b11_done + ((counted & 32) ? 8 : 0), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Head of the chain queue in a merged launch: ONE wave that returns when `*count >= target` (the counted items of this run are
// done; the counter only grows, run r waits for (r + 1) x items, so a stale value of the run before can never satisfy it).  A relaxed agent-scope
// poll (an sc1 load: served by the L2 / fabric, never by this CU's L1) every ~3 us; the kernels queued behind it start through an
// ordinary launch, which is their acquire.  Bounded: after `timeout_ticks` of the 100 MHz wall clock it gives up and raises the
// job-wide failure flag (`status[0 .. n_status)`) instead of holding the queue forever (a Gram launch that never ran: the host
// reports the run as failed).
__global__ __launch_bounds__(64) void wait_count_kernel(const unsigned long long* __restrict__ count, unsigned long long target,
                                                        unsigned long long timeout_ticks, int* __restrict__ status, int n_status)
{
    if (threadIdx.x != 0) return;
    __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
        if (wall_clock64() - t0 > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(127);
    }
    for (int i = 0; i < n_status; i++) status[i] = 1;                 // the job-wide failure flag (gauss_job_fetch reports the run as failed)
}

// the other half of the context's queue probe (gauss_ctx.cpp: queues_side_by_side): one lane adds 1 to a counter
__global__ __launch_bounds__(64) void count_up_kernel(unsigned long long* __restrict__ count)
{
    if (threadIdx.x == 0) __hip_atomic_fetch_add(count, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

void launch_count_up(unsigned long long* d_count, hipStream_t s)
{
    hipLaunchKernelGGL(count_up_kernel, dim3(1), dim3(64), 0, s, d_count);
}

// wait_count_kernel with an explicit bound (no two-second floor, no test hook): the queue probe waits a few milliseconds at most
void launch_wait_count_for(const unsigned long long* d_count, unsigned long long target, int* d_status, hipStream_t s, double bound_us)
{
    hipLaunchKernelGGL(wait_count_kernel, dim3(1), dim3(64), 0, s, d_count, target, (unsigned long long)(bound_us * 100.0), d_status, 1);
}

void launch_wait_count(const unsigned long long* d_count, unsigned long long target, int* d_status, int n_status, hipStream_t s,
                       double bound_us)
{
    // two seconds of wall_clock64 ticks (100 MHz), or the caller's larger bound; GAUSS_WAIT_COUNT_TIMEOUT_US overrides; a NEGATIVE value is the tests' way into
    // the give-up path: the wait is for a count that never comes and ends after that many microseconds
    unsigned long long ticks = bound_us > 2e6 ? (unsigned long long)(bound_us * 100.0) : 200000000ull;     // (the caller's bound for very large jobs)
    if (const char* e = getenv("GAUSS_WAIT_COUNT_TIMEOUT_US")) {
        const long long us = atoll(e);
        if (us < 0) target = ~0ull;
        ticks = (unsigned long long)(us < 0 ? -us : us) * 100ull;
    }
    hipLaunchKernelGGL(wait_count_kernel, dim3(1), dim3(64), 0, s, d_count, target, ticks, d_status, n_status);
}

void launch_gram(const Item* d_items, int n_items, int dtype_i8, hipStream_t s, unsigned long long* d_b11_done)
{
    if (n_items <= 0) return;
    // two chunk images (32 KB), four workgroups per CU for both paths.  Deeper rings were measured for the int8
    // kernel (NS 3 / 4 with 3 / 2 workgroups per CU: 5.28 / 5.38 ms against 5.16 ms): it is not waiting for its
    // loads but for LDS bandwidth (1 KB of fragment reads per MFMA with 64 x 64 wave tiles)
#ifndef GAUSS_I8_NS
#define GAUSS_I8_NS 2
#define GAUSS_I8_OCC 4
#endif
    if (dtype_i8) hipLaunchKernelGGL((gram_kernel<i32x16, GAUSS_I8_NS, GAUSS_I8_OCC>), dim3(n_items), dim3(256), 0, s, d_items, d_b11_done);
    else hipLaunchKernelGGL((gram_kernel<f32x16, 2, 4>), dim3(n_items), dim3(256), 0, s, d_items, d_b11_done);
}

}  // namespace gauss
