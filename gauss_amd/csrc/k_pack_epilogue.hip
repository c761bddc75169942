// K1 pack_stats, K1b row_stats, K4 LD epilogue (fp64, reference operation order).
//
// This translation unit is compiled with -ffp-contract=off: the fp64 tails below follow the
// reference's CalCor / CalWgtCov tails operation by operation (util.cpp:66-68, 117-123;
// distmix.cpp:181-182,195-196), and a fused multiply-add would round differently from the
// reference's separately rounded x86-64 multiply and add.
#include "gauss_internal.h"

namespace gauss {

// ------------------------------------------------------------------------------------------
// K1: raw genotype bytes (ASCII digits or 0..2) -> packed operand rows (one e4m3 byte per code) + per-population
// integer sums.  Replaces the per-pair recomputation of sumx / sumxsq inside CalCor
// (util.cpp:58-61) and the allele count of MakeSnpVec (gauss.cpp:581-583).
// One workgroup per SNP row; a thread handles 16 packed bytes at a time.
// Packed layout: population p occupies bytes [pop_pk_off[p], pop_pk_off[p]+m_p), zero padded
// up to the next multiple of KC, so a K chunk never straddles two populations and zero padding
// contributes nothing to any sum.
// ------------------------------------------------------------------------------------------
struct __attribute__((packed)) U32u { uint32_t v; };

// (Round 5 measured two ways of making this kernel faster, neither of which moved it: several rows per workgroup -- tables staged
// once, fewer starts -- took 1.53 ms against 1.07 for one row each, the launch's load balance and latency hiding suffer; a third
// fewer vector instructions -- the 2-bit decode by spreading, a branch-free e4m3 encoding, quad instead of wave reductions: kept,
// they are simpler -- left it at 1.04-1.07 ms.  It moves 0.9 GB in + 3.3 GB out in that time, 4.0 TB/s, where torch's copy_
// kernel reaches 4.6 TB/s on a 1 : 1 stream and its 1 : 4 expanding copy 3.9 (tools/hbm_write_probe.py: fill_ alone 6.5): the
// kernel sits at what a write-heavy mixed stream gets out of this memory system.)
__global__ __launch_bounds__(256) void pack_stats_kernel(const Prob* __restrict__ probs,
                                                         const int2* __restrict__ rowmap)
{
    __shared__ int s_sx[64];
    __shared__ int s_sxx[64];
    // word -> source block tables staged in LDS: looked up once per 16 samples, and a chain of dependent global
    // loads (word table -> offsets -> data) would set the pace of the whole kernel
    constexpr int PACK_LDS_WORDS = 8192;
    __shared__ uint8_t s_word[PACK_LDS_WORDS];
    __shared__ int s_pk[65], s_src[65], s_len[64];
    __shared__ double s_cov[64], s_mm[64], s_wmu[64];
    const int2 rm = rowmap[blockIdx.x];
    const Prob& pb = probs[rm.x];
    const int r = rm.y;
    const uint8_t* src;
    int prow;                       // row inside its part: measured rows [0, M), unmeasured rows [0, U)
    const bool um = r >= pb.M;
    int code = 0;
    if (!um) {
        const long long srow = pb.rows_m ? pb.rows_m[r] : r;            // row lists gather from a resident store
        src = pb.raw_m + (size_t)srow * pb.ld_raw; prow = r;
    } else {
        // row block b of the unmeasured rows re-reads raw row (r - M) % U_raw under coding code_blk[b]
        const int ur = r - pb.M;
        const int blk = ur / pb.U_raw;
        const int rr = ur - blk * pb.U_raw;
        const long long srow = pb.rows_u ? pb.rows_u[rr] : rr;
        src = pb.raw_u + (size_t)srow * pb.ld_raw;
        prow = ur;
        code = pb.code_blk[blk];
    }
    uint4* dst = reinterpret_cast<uint4*>((um ? pb.packed_u : pb.packed) + (size_t)prow * pb.Kp);

    if (threadIdx.x < 64) { s_sx[threadIdx.x] = 0; s_sxx[threadIdx.x] = 0; }
    const int nwords = pb.Kp >> 4;                 // a multiple of 4 (Kp is a multiple of 64)
    const bool fmt2 = pb.geno_fmt != 0;
    const int n_tab = fmt2 ? pb.n_run : pb.P;      // source blocks ("runs"): populations, or 2-bit blocks
    const bool tab_lds = nwords <= PACK_LDS_WORDS;
    {
        const auto wt = fmt2 ? pb.word_run : pb.word_pop;
        if (tab_lds)
            for (int w = threadIdx.x; w < nwords; w += 256) s_word[w] = wt[w];
        if (threadIdx.x < n_tab) {
            const int r = threadIdx.x;
            s_pk[r] = fmt2 ? pb.run_pk_off[r] : pb.pop_pk_off[r];
            s_src[r] = fmt2 ? pb.run_src[r] : pb.pop_raw_off[r];
            s_len[r] = fmt2 ? 0 : pb.pop_raw_off[r + 1] - pb.pop_raw_off[r];
        }
    }
    __syncthreads();

    // one packed word (16 samples) of the row: recode, per-population sums, operand encoding, store
    auto finish_word = [&](int w, bool live, int p, uint32_t (&v)[4]) {
        int sx = 0, sxx = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t x = fmt2 ? v[q] : (v[q] & 0x0F0F0F0Fu);         // '0'..'9' and 0..15 both decode as byte & 0x0F (2-bit sources: 0..3 already)
            if (code) {
                // ConvertGenotypesToDominant / ToRecessive (gauss.cpp:1196-1250): only codes 0..2 are mapped
                const uint32_t ge4 = ((x >> 2) | (x >> 3)) & 0x01010101u;
                const uint32_t is3 = x & (x >> 1) & 0x01010101u & ~ge4;
                const uint32_t keep = (ge4 | is3) * 0xFFu;
                const uint32_t rec = (code == 1) ? ((x | (x >> 1)) & 0x01010101u) : ((x >> 1) & 0x01010101u);
                x = (x & keep) | (rec & ~keep);
            }
            sx = __builtin_amdgcn_udot4(x, 0x01010101u, sx, false);
            sxx = __builtin_amdgcn_udot4(x, x, sxx, false);
            // operand encoding for the Gram kernel: the code as an OCP e4m3 byte (exact for 0..15)
            uint32_t c;
            if (pb.gram_i8) {
                c = x;                                   // i8 MFMA path: operands are the raw codes
            } else if (fmt2 || (x & 0x08080808u) == 0) {
                // codes 0..7 -- all a 2-bit source can hold, and the rule for byte sources -- through v_perm_b32 as an eight-entry
                // byte table: selector byte t picks byte t of {0x4E4C4A48 : 0x44403800} = the e4m3 codes of 0..7 (0x00, 0x38, 0x40, 0x44,
                // 0x48, 0x4A, 0x4C, 0x4E).  One instruction a dword: this kernel is bound by its vector instructions (86 % of the VALU
                // issue slots in round 4's counters, 80 % after round 5's first cuts), not by HBM
                c = __builtin_amdgcn_perm(0x4E4C4A48u, 0x44403800u, x);
            } else {
                c = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t t = (x >> (8 * b)) & 0xFu;
                    // e4m3: 0, 1=0x38, 2=0x40, 3=0x44, 4..7 = 0x48 + 2(t-4), 8..15 = 0x50 + (t-8)
                    const uint32_t e = t == 0 ? 0u : t == 1 ? 0x38u : t < 4 ? 0x40u + 4u * (t - 2) : t < 8 ? 0x48u + 2u * (t - 4) : 0x50u + (t - 8);
                    c |= e << (8 * b);
                }
            }
            v[q] = c;
        }
        // chunk_unit_layout (gauss_internal.h): the eight dwords of a 32-sample group leave transposed -- this word (even w:
        // sample dwords 0..3 of the group, odd w: 4..7) keeps two of its dwords and trades the other two with the neighbouring
        // lane, which holds the group's other word: even words end up with sample dwords 0, 2, 4, 6, odd ones with 1, 3, 5, 7.
        // (Groups start on even words: blocks are padded to 64 samples.  Lane pairs are always in the loop together.)
        {
            const bool odd = (w & 1) != 0;
            const uint32_t s0 = odd ? v[0] : v[1], s1 = odd ? v[2] : v[3];
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, true);    // quad_perm [1, 0, 3, 2]: lane ^ 1
            const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, true);
            if (live) dst[w] = odd ? make_uint4(r0, r1, v[1], v[3]) : make_uint4(v[0], v[2], r0, r1);
        }
        // per-population sums: a lane quad holds four consecutive words = one 64-sample chunk, and a chunk never straddles two
        // populations (blocks are padded to 64 samples; lane pairs and quads are always in the loop together): two DPP steps add
        // the quad up, its first lane adds to the population's LDS counters.  (Round 4 reduced across the whole wave when it sat
        // inside one population: twelve ds_bpermute + their adds per word, a fifth of the kernel's vector instructions.)
        sx += __builtin_amdgcn_update_dpp(0, sx, 0xB1, 0xF, 0xF, true);        // quad_perm [1, 0, 3, 2]
        sxx += __builtin_amdgcn_update_dpp(0, sxx, 0xB1, 0xF, 0xF, true);
        sx += __builtin_amdgcn_update_dpp(0, sx, 0x4E, 0xF, 0xF, true);        // quad_perm [2, 3, 0, 1]
        sxx += __builtin_amdgcn_update_dpp(0, sxx, 0x4E, 0xF, 0xF, true);
        if ((threadIdx.x & 3) == 0 && sx) { atomicAdd(&s_sx[p], sx); atomicAdd(&s_sxx[p], sxx); }
    };

    const int nloop = (nwords + 255) & ~255;
    if (fmt2 && tab_lds) {
        // 2-bit packed source: 16 samples = 4 bytes; blocks are 16-byte aligned and zero padded to 64 samples, i.e. they
        // have exactly the packed operand layout at a quarter of the bytes.  A lane's load is only 4 bytes, so several
        // words per lane are requested before the first is used (the row is 8 KB: bytes in flight, not bandwidth,
        // set the pace of a one-load-at-a-time loop).
#ifndef GAUSS_PACK_BATCH
#define GAUSS_PACK_BATCH 4
#endif
        constexpr int PB = GAUSS_PACK_BATCH;
        for (int w0 = threadIdx.x; w0 < nloop; w0 += PB * 256) {
            uint32_t bits[PB];
            int ww[PB], pp[PB];
            bool lv[PB];
#pragma unroll
            for (int j = 0; j < PB; j++) {
                const int wj = w0 + 256 * j;
                lv[j] = wj < nwords;
                ww[j] = lv[j] ? wj : nwords - 1;
                const int run = s_word[ww[j]];
                pp[j] = (pb.P == 1) ? 0 : run;
                const int o = (ww[j] << 4) - s_pk[run];
                bits[j] = lv[j] ? *reinterpret_cast<const uint32_t*>(src + s_src[run] + (o >> 2)) : 0u;
            }
#pragma unroll
            for (int j = 0; j < PB; j++) {
                if (w0 + 256 * j >= nloop) break;              // whole waves stay together: nloop is a multiple of 256
                uint32_t v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t b = (bits[j] >> (8 * q)) & 0xFFu;            // four 2-bit codes -> one code per byte, by spreading
                    b = (b | (b << 12)) & 0x000F000Fu;
                    v[q] = (b | (b << 6)) & 0x03030303u;
                }
                finish_word(ww[j], lv[j], pp[j], v);
            }
        }
    } else {
        for (int w0 = threadIdx.x; w0 < nloop; w0 += 256) {
            const bool live = w0 < nwords;                // keep whole waves in the loop for the cross-lane sums
            const int w = live ? w0 : nwords - 1;
            const int run = tab_lds ? s_word[w] : (fmt2 ? pb.word_run[w] : pb.word_pop[w]);
            const int p = (pb.P == 1) ? 0 : run;          // pooled statistics see one pseudo-population
            const int o = (w << 4) - s_pk[run];
            uint32_t v[4] = {0u, 0u, 0u, 0u};
            if (fmt2) {
                const uint32_t bits = live ? *reinterpret_cast<const uint32_t*>(src + s_src[run] + (o >> 2)) : 0u;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t b = (bits >> (8 * q)) & 0xFFu;
                    b = (b | (b << 12)) & 0x000F000Fu;
                    v[q] = (b | (b << 6)) & 0x03030303u;
                }
            } else {
                int valid = s_len[run] - o;
                valid = valid < 0 ? 0 : (valid > 16 ? 16 : valid);
                if (!live) valid = 0;
                const uint8_t* s = src + s_src[run] + o;
                if (valid == 16) {
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] = reinterpret_cast<const U32u*>(s + 4 * q)->v;
                } else {
                    for (int b = 0; b < valid; b++) v[b >> 2] |= (uint32_t)s[b] << (8 * (b & 3));
                }
            }
            finish_word(w, live, p, v);
        }
    }
    __syncthreads();
    const int P = pb.P;
    if (threadIdx.x < P) {
        (um ? pb.sx_u : pb.sx)[(size_t)prow * P + threadIdx.x] = s_sx[threadIdx.x];
        (um ? pb.sxx_u : pb.sxx)[(size_t)prow * P + threadIdx.x] = s_sxx[threadIdx.x];
    }
    // ---- the row's fp64 tables (K1b: what row_stats_kernel did in a launch of its own; the sums are in LDS right here) ----
    //  weighted (CalWgtCov(x,x), distmix.cpp:180-187 / computeLD.cpp:100-103):
    //     rt_mu[p]  = sumx_p / m_p            rt_wmu[p] = w_p * (sumx_p / m_p)
    //     rt_wm     = sum_p w_p * mu_p        rt_sd     = sqrt(CalWgtCov(x,x))
    //  pooled (CalCor, util.cpp:66-67):
    //     rt_wm = sumx (all samples)          rt_sd = sqrt(n*sumxsq - sumx*sumx)
    // Every population's terms are formed by a lane of its own -- the same expressions, operation for operation -- and one
    // lane adds them in population order, as the reference's loop does (util.cpp:113-122): the same bits.
    const auto rt_wm = um ? pb.rt_wm_u : pb.rt_wm;
    const auto rt_sd = um ? pb.rt_sd_u : pb.rt_sd;
    if (pb.mode == 0) {
        if (threadIdx.x == 0) {
            double sumx = 0, sumxsq = 0;
            int num_samples = 0;
            for (int p = 0; p < P; p++) {
                sumx += (double)s_sx[p];
                sumxsq += (double)s_sxx[p];
                num_samples += pb.pop_raw_off[p + 1] - pb.pop_raw_off[p];
            }
            rt_wm[prow] = sumx;
            rt_sd[prow] = sqrt((num_samples) * sumxsq - sumx * sumx);
        }
        return;
    }
    if (threadIdx.x < P) {
        const int p = threadIdx.x;
        const int m = pb.pop_raw_off[p + 1] - pb.pop_raw_off[p];
        const double wgt_val = pb.pop_w[p];
        const double sumx = (double)s_sx[p];
        const double sumxy = (double)s_sxx[p];
        const double factor = ((double)m) / (m - 1);
        s_cov[p] = wgt_val * factor * (m * sumxy - sumx * sumx);          // util.cpp:118
        const double mu = sumx / m;
        const double wmu = wgt_val * mu;
        s_mm[p] = wmu * mu;                                               // util.cpp:119
        s_wmu[p] = wmu;
        (um ? pb.rt_mu_u : pb.rt_mu)[(size_t)prow * P + p] = mu;
        (um ? pb.rt_wmu_u : pb.rt_wmu)[(size_t)prow * P + p] = wmu;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double wsumcov = 0, wsum_mi_mj = 0, wsum_mi = 0;
        for (int p = 0; p < P; p++) { wsumcov += s_cov[p]; wsum_mi_mj += s_mm[p]; wsum_mi += s_wmu[p]; }
        rt_wm[prow] = wsum_mi;
        rt_sd[prow] = sqrt(wsumcov + wsum_mi_mj - wsum_mi * wsum_mi);     // util.cpp:123; distmix.cpp:181-182
    }
}

void launch_pack_stats(const Prob* d_probs, const int2* d_rowmap, int n_rows, hipStream_t s)
{
    if (n_rows > 0) hipLaunchKernelGGL(pack_stats_kernel, dim3(n_rows), dim3(256), 0, s, d_probs, d_rowmap);
}

// ------------------------------------------------------------------------------------------
// Correlation of packed rows (ri, rj) from the exact per-segment Gram partials of one tile.
// `x` is the first argument of CalCor / CalWgtCov (row ri), `y` the second (row rj).
// ------------------------------------------------------------------------------------------
// A partial slab holds exact integers, as f32 (f32 MFMA path) or int32 (i8 MFMA path).
__device__ __forceinline__ double slab_val(float v, int is_int) { return is_int ? (double)__float_as_int(v) : (double)v; }

// ri: row of the first argument inside its part -- a measured row, or (ri_unmeasured) an unmeasured one; rj: measured row.
__device__ __forceinline__ double cor_entry(const Prob& pb, const float* __restrict__ tile_slab,
                                            int off, int ri, int rj, bool ri_unmeasured = false)
{
    const int P = pb.P;
    const auto rt_wm_i = ri_unmeasured ? pb.rt_wm_u : pb.rt_wm;
    const auto rt_sd_i = ri_unmeasured ? pb.rt_sd_u : pb.rt_sd;
    const size_t seg_stride = (size_t)TILE * TILE;
    if (pb.mode == 0) {
        double sumxy = 0;
        for (int s = 0; s < pb.nseg; s++) sumxy += slab_val(tile_slab[s * seg_stride + off], pb.gram_i8);
        const int num_samples = pb.N;
        const double numer = num_samples * sumxy - rt_wm_i[ri] * pb.rt_wm[rj];    // util.cpp:66
        const double denor = rt_sd_i[ri] * pb.rt_sd[rj];                          // util.cpp:67
        return numer / denor;                                                     // util.cpp:68
    }
    double wsumcov = 0, wsum_mi_mj = 0;
    const int* sxi = (ri_unmeasured ? pb.sx_u : pb.sx) + (size_t)ri * P;
    const int* sxj = pb.sx + (size_t)rj * P;
    const double* wmui = (ri_unmeasured ? pb.rt_wmu_u : pb.rt_wmu) + (size_t)ri * P;
    const double* muj = pb.rt_mu + (size_t)rj * P;
    for (int p = 0; p < P; p++) {
        double sumxy = 0;
        for (int s = pb.pop_seg0[p]; s < pb.pop_seg0[p + 1]; s++)
            sumxy += slab_val(tile_slab[s * seg_stride + off], pb.gram_i8);
        const double sumx = (double)sxi[p], sumy = (double)sxj[p];
        // pop_wf = wgt_val * factor, factor = (double)m/(m-1) (util.cpp:117), pop_md = (double)m
        wsumcov += pb.pop_wf[p] * (pb.pop_md[p] * sumxy - sumx * sumy);            // util.cpp:118
        wsum_mi_mj += wmui[p] * muj[p];                                            // util.cpp:119
    }
    const double cov = wsumcov + wsum_mi_mj - rt_wm_i[ri] * pb.rt_wm[rj];         // util.cpp:123
    return cov / (rt_sd_i[ri] * pb.rt_sd[rj]);                                    // distmix.cpp:196
}

// ------------------------------------------------------------------------------------------
// K4: LD epilogue.  One workgroup per tile pair.  Turns the exact integer Gram partials into
// fp64 correlations and scatters them into
//   - A[0] = B11 (diag 1+lambda), A[1] = B11 - eps*I, both row-major Mld x Mld, identity padded
//   - B21 (U x Mld row-major), zero padded
// or, for LD-only problems, into out_ld (S x S, diag = pb.diag).
// HBM-bound: it reads every partial once (nseg floats per entry) and writes one double.
// Thread t owns column jj = t & 127 and rows ii = (t >> 7) + 2r: a wave reads 64 consecutive
// floats of a partial row (coalesced).  The per-population row/column tables of the tile live
// in LDS; rows are handled four at a time to keep several partial loads in flight.
// ------------------------------------------------------------------------------------------
constexpr int EPI_MAXP = 32;     // LDS table capacity (33KG has 29 populations); beyond: global tables

struct EpiOut { double v0, v1; };

// Partial slabs hold exact integers: as f32 (f32 MFMA path) or as int32 bit patterns (i8 MFMA path).  The sum
// of a population's partials is accumulated as int32 -- exact below 2^31, i.e. for populations of up to
// 9.5 M samples even with the largest accepted codes (15 * 15 * m; the planner enforces the bound) -- and
// widened to fp64 once per population.
template <bool ISINT> struct SlabAcc {
    int v;
    __device__ __forceinline__ void zero() { v = 0; }
    __device__ __forceinline__ void add(float x) { v += ISINT ? __float_as_int(x) : (int)x; }
    __device__ __forceinline__ double get() const { return (double)v; }
};

// The 4 x 4 block of one partial slab that a thread owns, in flight as raw dwords: four 16-byte loads for f32 / int32
// slabs (rows rg + 32 q), two for 16-bit slabs (rows 2 rg, 2 rg + 1, 64 + 2 rg, 65 + 2 rg: a dword carries a row pair).
typedef uint32_t u32x4e __attribute__((ext_vector_type(4)));
template <bool ISINT, bool S16> struct SlabBlock {
    u32x4e raw[S16 ? 2 : 4];
    template <typename PT>
    __device__ __forceinline__ void load(PT tile_slab, int s, const int (&rows)[4], int c0)
    {
        if (S16) {
            const auto base = (GP(const uint32_t))tile_slab + (size_t)s * (TILE * TILE / 2);
            raw[0] = *(GP(const u32x4e))(base + (rows[0] >> 1) * TILE + c0);
            raw[1] = *(GP(const u32x4e))(base + (rows[2] >> 1) * TILE + c0);
        } else {
            const auto base = (GP(const uint32_t))tile_slab + (size_t)s * (TILE * TILE);
#pragma unroll
            for (int q = 0; q < 4; q++) raw[q] = *(GP(const u32x4e))(base + rows[q] * TILE + c0);
        }
    }
    __device__ __forceinline__ int get(int q, int c) const
    {
        if (S16) { const uint32_t d = raw[q >> 1][c]; return (int)((q & 1) ? (d >> 16) : (d & 0xFFFFu)); }
        return ISINT ? (int)raw[q][c] : (int)__uint_as_float(raw[q][c]);
    }
};

// One entry (ri <= rj, window-relative measured rows) of a measured x measured tile goes to its places: LD export, or
// A[0] = B11 (diag 1 + lambda), A[1] = B11 - eps I and the factorisation's working copy, identity padded to Mld.
__device__ __forceinline__ void epi_store_sym(const Prob& pb, int ri, int rj, double cv, bool need_a1)
{
    const int Mld = pb.Mld;
    if (ri > rj || ri < 0) return;                        // mirror handles the lower part; ri < 0: not this window's SNP
    if (pb.ld_only) {
        if (ri >= pb.M || rj >= pb.M) return;
        const double v = (ri == rj) ? pb.diag : cv;
        pb.out_ld[(size_t)ri * pb.M + rj] = v;
        pb.out_ld[(size_t)rj * pb.M + ri] = v;
        return;
    }
    if (ri >= Mld || rj >= Mld) return;
    double v0, v1;
    if (ri >= pb.M || rj >= pb.M) { v0 = v1 = (ri == rj) ? 1.0 : 0.0; }
    else if (ri == rj) { v0 = 1.0 + pb.lambda; v1 = v0 - pb.eps; }   // dist.cpp:172
    else { v0 = v1 = cv; }                                            // dist.cpp:174-177
    const auto A0 = pb.A;
    const auto A1 = pb.A + (size_t)Mld * Mld;
    A0[(size_t)ri * Mld + rj] = v0; A0[(size_t)rj * Mld + ri] = v0;
    // the shifted twin is only ever read by its own factorisation, which the certificate (shift_cert_kernel,
    // ahead of the Gram kernel on this stream) has already ruled out for most windows
    if (need_a1) { A1[(size_t)ri * Mld + rj] = v1; A1[(size_t)rj * Mld + ri] = v1; }
    if (pb.npanel > 0) {                                  // working copy of B11 for the in-place factorisation
        const auto W0 = pb.A + (size_t)4 * Mld * Mld;
        W0[(size_t)ri * Mld + rj] = v0; W0[(size_t)rj * Mld + ri] = v0;
    }
}

// Job-wide tiles end where the chromosome's measured SNPs end, not where this window's do: the identity padding of B11
// between M and Mld (rows of no SNP; the tiles of an unshared window write it as part of their own padding) is written
// by the workgroup (nthreads threads) of the window's LAST diagonal tile.
__device__ __forceinline__ void epi_pad_identity(const Prob& pb, int ti, int tj, bool need_a1, int tid, int nthreads)
{
    const int Mld = pb.Mld;
    if (!(ti == tj && !pb.ld_only && pb.g0 + pb.M - 1 >= ti * TILE && pb.g0 + pb.M - 1 < (ti + 1) * TILE)) return;
    for (int e = tid; e < (Mld - pb.M) * Mld; e += nthreads) {
        const int r = pb.M + e / Mld, c = e % Mld;
        const double v = (r == c) ? 1.0 : 0.0;
        pb.A[(size_t)r * Mld + c] = v; pb.A[(size_t)c * Mld + r] = v;
        if (need_a1) { pb.A[(size_t)Mld * Mld + (size_t)r * Mld + c] = v; pb.A[(size_t)Mld * Mld + (size_t)c * Mld + r] = v; }
        if (pb.npanel > 0) { pb.A[(size_t)4 * Mld * Mld + (size_t)r * Mld + c] = v; pb.A[(size_t)4 * Mld * Mld + (size_t)c * Mld + r] = v; }
    }
}

template <bool ISINT, bool S16>
__device__ __forceinline__ void epilogue_tile(const Prob& pb, int pair_in, int lds_pop_cap, char* esm)
{
    // A B11 tile pair of a job whose windows share their measured rows lies on JOB-WIDE row tiles (gpair_*, slab_g); this
    // window sees it shifted by g0, its first measured SNP: tile row r of job-wide tile gi is the window's measured row
    // gi * TILE + r - g0, which may fall outside [0, M) (another window's SNP: skipped below).  The row tables are
    // reached through the window's `X` pointers, which point into the job-wide arrays at g0, so the same (possibly
    // negative) window-relative index addresses them.
    const bool gb11 = (pair_in & TILE_GB11) != 0;
    const int pair = pair_in & ~TILE_GB11;
    const int ti = gb11 ? pb.gpair_ti[pair] : pb.pair_ti[pair], tj = gb11 ? pb.gpair_tj[pair] : pb.pair_tj[pair];
    const int P = pb.P, nseg = pb.nseg;
    // the pair's slabs; 16-bit slabs take half the floats (the planner only selects them when P fits the LDS tables)
    const auto tile_slab = (gb11 ? pb.slab_g : pb.slab) + (size_t)pair * nseg * (S16 ? TILE * TILE / 2 : TILE * TILE);
    const int mt = pb.Mp / TILE;        // number of measured row tiles
    const bool sym = gb11 || (ti < mt); // measured x measured tile (ti <= tj)
    const int Mld = pb.Mld;
    // status[3]: lambda_min(B11) > eps is certified.  Only B11's tiles read it (B21's tiles may run while the chain queue,
    // which makes the certificate in a merged launch, is still busy: gauss_run.cpp:job_queue_run)
    const bool need_a1 = !sym || !(pb.npanel > 0 && pb.status[3] != 0);
    const int tid = threadIdx.x;
    const bool weighted = pb.mode != 0;
    const bool lds_tables = weighted && P <= lds_pop_cap;
    // first row of the tile inside its part (measured rows / unmeasured rows), and the row arrays of that part
    const int row0_i = gb11 ? ti * TILE - pb.g0 : (sym ? ti * TILE : (ti - mt) * TILE);
    const int row0_j = gb11 ? tj * TILE - pb.g0 : tj * TILE;
    const auto i_wmu = sym ? pb.rt_wmu : pb.rt_wmu_u;
    const auto i_sx = sym ? pb.sx : pb.sx_u;
    const auto i_wm = sym ? pb.rt_wm : pb.rt_wm_u;
    const auto i_sd = sym ? pb.rt_sd : pb.rt_sd_u;

    double* s_wmui = reinterpret_cast<double*>(esm);          // [P][128]  w_p * mu_p(row i)
    double* s_muj = s_wmui + P * TILE;                        // [P][128]  mu_p(col j)
    int* s_sxi = reinterpret_cast<int*>(s_muj + P * TILE);    // [P][128]
    int* s_sxj = s_sxi + P * TILE;                            // [P][128]
    if (lds_tables) {
        for (int idx = tid; idx < P * TILE; idx += 1024) {
            const int r = idx / P, p = idx % P;               // consecutive threads: consecutive p of a row
            s_wmui[p * TILE + r] = i_wmu[(ptrdiff_t)(row0_i + r) * P + p];
            s_sxi[p * TILE + r] = i_sx[(ptrdiff_t)(row0_i + r) * P + p];
            s_muj[p * TILE + r] = pb.rt_mu[(ptrdiff_t)(row0_j + r) * P + p];
            s_sxj[p * TILE + r] = pb.sx[(ptrdiff_t)(row0_j + r) * P + p];
        }
        __syncthreads();
    }

    // Thread t owns the 4 consecutive columns 4 (t & 31) .. +3 and the 4 rows (t >> 5) + 32 q: every partial row
    // is read with one 16-byte load per thread (a wave covers two 512-byte rows), 16 entries per thread.
    const int c0 = (tid & 31) * 4, rg = tid >> 5;
    int rows[4];
#pragma unroll
    for (int q = 0; q < 4; q++) rows[q] = S16 ? (2 * rg + (q & 1) + 64 * (q >> 1)) : (rg + 32 * q);
    double cov[4][4];
    double sd_j[4], wm_j[4];
#pragma unroll
    for (int c = 0; c < 4; c++) { sd_j[c] = pb.rt_sd[row0_j + c0 + c]; wm_j[c] = pb.rt_wm[row0_j + c0 + c]; }
    const int num_samples = pb.N;

    if (!weighted) {
        // CalCor tail (util.cpp:66-68): r = (n*sxy - sx*sy) / (sqrt(..x..) * sqrt(..y..))
        SlabAcc<ISINT> acc[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[q][c].zero();
        SlabBlock<ISINT, S16> nxt;
        nxt.load(tile_slab, 0, rows, c0);
        for (int s = 0; s < nseg; s++) {
            const SlabBlock<ISINT, S16> cur = nxt;
            if (s + 1 < nseg) nxt.load(tile_slab, s + 1, rows, c0);          // next partial flies while this one is added
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[q][c].v += cur.get(q, c);
        }
        double sumxy[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int c = 0; c < 4; c++) sumxy[q][c] = acc[q][c].get();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ri = row0_i + rows[q];
            const double wm_i = i_wm[ri], sd_i = i_sd[ri];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const double numer = num_samples * sumxy[q][c] - wm_i * wm_j[c];
                const double denor = sd_i * sd_j[c];
                cov[q][c] = numer / denor;
            }
        }
    } else if (lds_tables) {
        // CalWgtCov (util.cpp:103-124) in the reference's population order
        double wsumcov[4][4] = {};
        SlabBlock<ISINT, S16> nxt;
        nxt.load(tile_slab, 0, rows, c0);
        for (int p = 0; p < P; p++) {
            SlabAcc<ISINT> acc[4][4];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[q][c].zero();
            for (int s = pb.pop_seg0[p]; s < pb.pop_seg0[p + 1]; s++) {
                const SlabBlock<ISINT, S16> cur = nxt;
                // segments are stored in population order: the next partial (same or next population) flies during
                // the fp64 tail
                if (s + 1 < nseg) nxt.load(tile_slab, s + 1, rows, c0);
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[q][c].v += cur.get(q, c);
            }
            double sumxy[4][4];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int c = 0; c < 4; c++) sumxy[q][c] = acc[q][c].get();
            const double md = pb.pop_md[p], wf = pb.pop_wf[p];
            double sumy[4];
#pragma unroll
            for (int c = 0; c < 4; c++) sumy[c] = (double)s_sxj[p * TILE + c0 + c];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double sumx = (double)s_sxi[p * TILE + rows[q]];
#pragma unroll
                for (int c = 0; c < 4; c++) wsumcov[q][c] += wf * (md * sumxy[q][c] - sumx * sumy[c]);      // util.cpp:118
            }
        }
        double wmm[4][4] = {};
        for (int p = 0; p < P; p++) {
            double mu_y[4];
#pragma unroll
            for (int c = 0; c < 4; c++) mu_y[c] = s_muj[p * TILE + c0 + c];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double wmu_x = s_wmui[p * TILE + rows[q]];
#pragma unroll
                for (int c = 0; c < 4; c++) wmm[q][c] += wmu_x * mu_y[c];                                    // util.cpp:119
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ri = row0_i + rows[q];
            const double wm_i = i_wm[ri], sd_i = i_sd[ri];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const double cv = wsumcov[q][c] + wmm[q][c] - wm_i * wm_j[c];                                // util.cpp:123
                cov[q][c] = cv / (sd_i * sd_j[c]);                                                           // distmix.cpp:196
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int c = 0; c < 4; c++)
                cov[q][c] = cor_entry(pb, (const float*)tile_slab, rows[q] * TILE + c0 + c, row0_i + rows[q], row0_j + c0 + c, !sym);
    }

#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int ri = row0_i + rows[q];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int rj = row0_j + c0 + c;
            if (sym) epi_store_sym(pb, ri, rj, cov[q][c], need_a1);
            else {
                const int u = ri;                                 // unmeasured row (x), measured col (y)
                if (u >= pb.U || rj >= pb.M) continue;
                pb.B21[(size_t)u * Mld + rj] = cov[q][c];         // dist.cpp:188-191
            }
        }
    }
    if (gb11) epi_pad_identity(pb, ti, tj, need_a1, tid, 1024);
}

template <bool ISINT>
__global__ __launch_bounds__(1024) void epilogue_kernel(const Prob* __restrict__ probs,
                                                        const int2* __restrict__ tilemap, int lds_pop_cap)
{
    extern __shared__ __attribute__((aligned(16))) char esm[];
    const int2 tm = tilemap[blockIdx.x];
    const Prob& pb = probs[tm.x];
    if (pb.slab16) epilogue_tile<ISINT, true>(pb, tm.y, lds_pop_cap, esm);
    else epilogue_tile<ISINT, false>(pb, tm.y, lds_pop_cap, esm);
}

void launch_epilogue(const Prob* d_probs, const int2* d_tilemap, int n_tiles, int max_pop, int dtype_i8, hipStream_t s)
{
    if (n_tiles <= 0) return;
    static DeviceOnce attr_once;
    attr_once.run([&] {
        const int maxb = (int)((size_t)EPI_MAXP * TILE * (2 * sizeof(double) + 2 * sizeof(int)));
        hipFuncSetAttribute(reinterpret_cast<const void*>(epilogue_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
        hipFuncSetAttribute(reinterpret_cast<const void*>(epilogue_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
    });
    const int cap = max_pop > EPI_MAXP ? EPI_MAXP : (max_pop < 1 ? 1 : max_pop);   // P > cap: tables stay in global memory
    const size_t smem = (size_t)cap * TILE * (2 * sizeof(double) + 2 * sizeof(int));
    if (dtype_i8) hipLaunchKernelGGL(epilogue_kernel<true>, dim3(n_tiles), dim3(1024), smem, s, d_probs, d_tilemap, cap);
    else hipLaunchKernelGGL(epilogue_kernel<false>, dim3(n_tiles), dim3(1024), smem, s, d_probs, d_tilemap, cap);
}

// ------------------------------------------------------------------------------------------
// B11's epilogue tiles in a small-footprint form: 256 threads, 18 KB of LDS, <= 96 registers, so that a workgroup fits into
// what gram_kernel's four workgroups per CU leave free and the tiles run BESIDE the Gram launch of B21's items, ahead of
// the factorisation chain on the chain queue (gauss_run.cpp:job_queue_run, k_solve_lite.hip).  Entry for entry the arithmetic of
// epilogue_tile -- integer sum of a population's partials, util.cpp:118 / :119 in population order, util.cpp:123,
// distmix.cpp:196 (pooled: util.cpp:66-68) -- hence the same bits.  What differs is the staging: the tile is walked in
// 2 column halves x 4 passes of 32 rows (thread = one row pair x 4 columns), and the per-population tables come through
// LDS in chunks of 16 populations (64 columns + 32 rows at a time) instead of all at once (78 KB at 26 populations).
// ------------------------------------------------------------------------------------------
constexpr int ELP = 16;                                    // populations per table chunk
struct EpiLiteTables {
    double muj[ELP][64];                                   // mu_p(col j)
    double wmui[ELP][32];                                  // w_p * mu_p(row i)
    int sxj[ELP][64];
    int sxi[ELP][32];
};

template <bool ISINT, bool S16>
__device__ __forceinline__ void epilogue_b11_lite_tile(const Prob& pb, int pair_in, EpiLiteTables& tb)
{
    const bool gb11 = (pair_in & TILE_GB11) != 0;
    const int pair = pair_in & ~TILE_GB11;
    const int ti = gb11 ? pb.gpair_ti[pair] : pb.pair_ti[pair], tj = gb11 ? pb.gpair_tj[pair] : pb.pair_tj[pair];
    const int P = pb.P, nseg = pb.nseg;
    const auto tile_slab = (gb11 ? pb.slab_g : pb.slab) + (size_t)pair * nseg * (S16 ? TILE * TILE / 2 : TILE * TILE);
    const bool need_a1 = !(pb.npanel > 0 && pb.status[3] != 0);
    const int tid = threadIdx.x;
    const bool weighted = pb.mode != 0;
    const int row0_i = gb11 ? ti * TILE - pb.g0 : ti * TILE;
    const int row0_j = gb11 ? tj * TILE - pb.g0 : tj * TILE;
    const int cg = tid & 15, rp = tid >> 4;
    const int num_samples = pb.N;
    // the thread's two rows x four columns of one partial slab, as raw dwords (16-bit slabs: one dword = a row pair)
    auto load = [&](u32x4e (&raw)[2], int s, int q, int c0) {
        if (S16) raw[0] = *(GP(const u32x4e))((GP(const uint32_t))tile_slab + (size_t)s * (TILE * TILE / 2) + q * TILE + c0);
        else {
            const auto base = (GP(const uint32_t))tile_slab + (size_t)s * (TILE * TILE);
            raw[0] = *(GP(const u32x4e))(base + (2 * q) * TILE + c0);
            raw[1] = *(GP(const u32x4e))(base + (2 * q + 1) * TILE + c0);
        }
    };
    auto get = [&](const u32x4e (&raw)[2], int h, int c) -> int {
        if (S16) { const uint32_t d = raw[0][c]; return (int)(h ? (d >> 16) : (d & 0xFFFFu)); }
        return ISINT ? (int)raw[h][c] : (int)__uint_as_float(raw[h][c]);
    };
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        const int c0 = 64 * half + 4 * cg;
        double sd_j[4], wm_j[4];
#pragma unroll
        for (int c = 0; c < 4; c++) { sd_j[c] = pb.rt_sd[row0_j + c0 + c]; wm_j[c] = pb.rt_wm[row0_j + c0 + c]; }
#pragma unroll 1
        for (int pass = 0; pass < 4; pass++) {
            const int q = rp + 16 * pass;                  // row pair: tile rows 2 q, 2 q + 1
            double cov[2][4];
            if (!weighted) {
                int acc[2][4] = {};
                u32x4e nxt[2];
                load(nxt, 0, q, c0);
                for (int sgm = 0; sgm < nseg; sgm++) {
                    u32x4e cur[2] = {nxt[0], nxt[1]};
                    if (sgm + 1 < nseg) load(nxt, sgm + 1, q, c0);
#pragma unroll
                    for (int h = 0; h < 2; h++)
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[h][c] += get(cur, h, c);
                }
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int ri = row0_i + 2 * q + h;
                    const double wm_i = pb.rt_wm[ri], sd_i = pb.rt_sd[ri];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const double numer = num_samples * (double)acc[h][c] - wm_i * wm_j[c];
                        const double denor = sd_i * sd_j[c];
                        cov[h][c] = numer / denor;
                    }
                }
            } else {
                double wsumcov[2][4] = {}, wmm[2][4] = {};
                u32x4e nxt[2];
                load(nxt, 0, q, c0);
                for (int p0 = 0; p0 < P; p0 += ELP) {
                    const int pn = min(ELP, P - p0);
                    __syncthreads();                       // the previous chunk's tables are no longer being read
                    for (int idx = tid; idx < pn * 64; idx += 256) {
                        const int r = idx / pn, p = idx % pn;             // consecutive threads: consecutive p of a row
                        tb.muj[p][r] = pb.rt_mu[(ptrdiff_t)(row0_j + 64 * half + r) * P + p0 + p];
                        tb.sxj[p][r] = pb.sx[(ptrdiff_t)(row0_j + 64 * half + r) * P + p0 + p];
                    }
                    for (int idx = tid; idx < pn * 32; idx += 256) {
                        const int r = idx / pn, p = idx % pn;
                        tb.wmui[p][r] = pb.rt_wmu[(ptrdiff_t)(row0_i + 32 * pass + r) * P + p0 + p];
                        tb.sxi[p][r] = pb.sx[(ptrdiff_t)(row0_i + 32 * pass + r) * P + p0 + p];
                    }
                    __syncthreads();
                    for (int p = 0; p < pn; p++) {
                        int acc[2][4] = {};
                        for (int sgm = pb.pop_seg0[p0 + p]; sgm < pb.pop_seg0[p0 + p + 1]; sgm++) {
                            u32x4e cur[2] = {nxt[0], nxt[1]};
                            if (sgm + 1 < nseg) load(nxt, sgm + 1, q, c0);          // segments are stored in population order
#pragma unroll
                            for (int h = 0; h < 2; h++)
#pragma unroll
                                for (int c = 0; c < 4; c++) acc[h][c] += get(cur, h, c);
                        }
                        const double md = pb.pop_md[p0 + p], wf = pb.pop_wf[p0 + p];
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const int lr = 2 * rp + h;                                 // row inside the pass
                            const double sumx = (double)tb.sxi[p][lr];
                            const double wmu_x = tb.wmui[p][lr];
#pragma unroll
                            for (int c = 0; c < 4; c++) {
                                const double sumy = (double)tb.sxj[p][4 * cg + c];
                                wsumcov[h][c] += wf * (md * (double)acc[h][c] - sumx * sumy);      // util.cpp:118
                                wmm[h][c] += wmu_x * tb.muj[p][4 * cg + c];                         // util.cpp:119
                            }
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int ri = row0_i + 2 * q + h;
                    const double wm_i = pb.rt_wm[ri], sd_i = pb.rt_sd[ri];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const double cv = wsumcov[h][c] + wmm[h][c] - wm_i * wm_j[c];               // util.cpp:123
                        cov[h][c] = cv / (sd_i * sd_j[c]);                                          // distmix.cpp:196
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int c = 0; c < 4; c++) epi_store_sym(pb, row0_i + 2 * q + h, row0_j + c0 + c, cov[h][c], need_a1);
        }
    }
    if (gb11) epi_pad_identity(pb, ti, tj, need_a1, tid, 256);
}

template <bool ISINT>
__global__ __launch_bounds__(256, 5) void epilogue_b11_lite_kernel(const Prob* __restrict__ probs, const int2* __restrict__ tilemap)
{
    __shared__ EpiLiteTables tb;
    __builtin_amdgcn_s_setprio(3);
    const int2 tm = tilemap[blockIdx.x];
    const Prob& pb = probs[tm.x];
    if (pb.slab16) epilogue_b11_lite_tile<ISINT, true>(pb, tm.y, tb);
    else epilogue_b11_lite_tile<ISINT, false>(pb, tm.y, tb);
}

// measured x measured tiles only (the first n_tiles_b11 entries of a job's tile map)
void launch_epilogue_b11_lite(const Prob* d_probs, const int2* d_tilemap, int n_tiles, int dtype_i8, hipStream_t s)
{
    if (n_tiles <= 0) return;
    if (dtype_i8) hipLaunchKernelGGL(epilogue_b11_lite_kernel<true>, dim3(n_tiles), dim3(256), 0, s, d_probs, d_tilemap);
    else hipLaunchKernelGGL(epilogue_b11_lite_kernel<false>, dim3(n_tiles), dim3(256), 0, s, d_probs, d_tilemap);
}

// Gene batches (gene.cpp:305-315, 571-586): block g is n_g x n_g with pb.diag on the diagonal.
__global__ __launch_bounds__(256) void gene_epilogue_kernel(const Prob* __restrict__ probs, int prob)
{
    const Prob& pb = probs[prob];
    const int g = blockIdx.x;
    const int r0 = pb.gene_off[g];
    const int n = pb.gene_off[g + 1] - r0;
    double* out = pb.out_ld + pb.gene_out_off[g];
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int a = e / n, b = e % n;
        if (a > b) continue;
        double v;
        if (a == b) v = pb.diag;
        else {
            const int ri = r0 + a, rj = r0 + b;
            const int ti = ri / TILE, tj = rj / TILE;
            const int pair = pb.pair_lut[ti * pb.nT + tj];
            const float* tile_slab = pb.slab + (size_t)pair * pb.nseg * TILE * TILE;
            v = cor_entry(pb, tile_slab, (ri % TILE) * TILE + (rj % TILE), ri, rj);
        }
        out[(size_t)a * n + b] = v;
        out[(size_t)b * n + a] = v;
    }
}

void launch_gene_epilogue(const Prob* d_probs, int prob, int n_gene, hipStream_t s)
{
    if (n_gene > 0) hipLaunchKernelGGL(gene_epilogue_kernel, dim3(n_gene), dim3(256), 0, s, d_probs, prob);
}

// Exact integer counts (integer parity hook): out[i][j] = sum over all segments of the slab.
__global__ __launch_bounds__(256) void counts_kernel(const Prob* __restrict__ probs, int prob,
                                                     long long* __restrict__ out)
{
    const Prob& pb = probs[prob];
    const int pair = blockIdx.x;
    const int ti = pb.pair_ti[pair], tj = pb.pair_tj[pair];
    const float* tile_slab = pb.slab + (size_t)pair * pb.nseg * TILE * TILE;
    for (int e = threadIdx.x; e < TILE * TILE; e += 256) {
        const int ri = ti * TILE + e / TILE, rj = tj * TILE + e % TILE;
        if (ri >= pb.M || rj >= pb.M) continue;
        if (ti == tj && ri > rj) continue;      // diagonal tile: the Gram kernel skips the mirrored quadrant
        long long s = 0;
        for (int g = 0; g < pb.nseg; g++) s += (long long)slab_val(tile_slab[(size_t)g * TILE * TILE + e], pb.gram_i8);
        out[(size_t)ri * pb.M + rj] = s;
        out[(size_t)rj * pb.M + ri] = s;
    }
}

// Per-population Pearson correlation of every SNP pair i < j (prep_zmix5, zmix.cpp:158-176; CalCor on one
// population's strings, util.cpp:153-169):  r_p = (m*Sxy_p - Sx_p*Sy_p) / (sqrt(m*Sxx_p - Sx_p^2) * sqrt(m*Syy_p - Sy_p^2)).
// out is [P][S(S-1)/2]: population-major, pairs in the reference's row order (i ascending, then j).
__global__ __launch_bounds__(256) void pop_cor_kernel(const Prob* __restrict__ probs, int prob, double* __restrict__ out)
{
    const Prob& pb = probs[prob];
    const int pair = blockIdx.x;
    const int ti = pb.pair_ti[pair], tj = pb.pair_tj[pair];
    const int P = pb.P, S = pb.M;
    const float* tile_slab = pb.slab + (size_t)pair * pb.nseg * TILE * TILE;
    const long long npairs = (long long)S * (S - 1) / 2;
    for (int e = threadIdx.x; e < TILE * TILE; e += 256) {
        const int ri = ti * TILE + e / TILE, rj = tj * TILE + e % TILE;
        if (ri >= S || rj >= S || ri >= rj) continue;       // i < j only; diagonal tiles hold the upper part
        const long long row = (long long)ri * S - (long long)ri * (ri + 1) / 2 + (rj - ri - 1);
        const int* sxi = pb.sx + (size_t)ri * P;
        const int* sxj = pb.sx + (size_t)rj * P;
        const int* sxxi = pb.sxx + (size_t)ri * P;
        const int* sxxj = pb.sxx + (size_t)rj * P;
        for (int p = 0; p < P; p++) {
            double sumxy = 0;
            for (int g = pb.pop_seg0[p]; g < pb.pop_seg0[p + 1]; g++)
                sumxy += slab_val(tile_slab[(size_t)g * TILE * TILE + e], pb.gram_i8);
            const int n = pb.pop_raw_off[p + 1] - pb.pop_raw_off[p];
            const double sumx = (double)sxi[p], sumy = (double)sxj[p];
            const double sumxsq = (double)sxxi[p], sumysq = (double)sxxj[p];
            const double numer = n * sumxy - sumx * sumy;                                            // util.cpp:165
            const double denor = sqrt((n) * sumxsq - sumx * sumx) * sqrt((n) * sumysq - sumy * sumy);  // util.cpp:166
            out[(size_t)p * npairs + row] = numer / denor;
        }
    }
}


// Pearson correlation of LISTED SNP pairs inside each population -- or inside each GROUP of populations pooled (the
// super-populations of prep_zmix5_sup, CalCorSup zmix.cpp:1221-1246) -- the selectors prep_zmix / prep_zmix2 / prep_zmix3 /
// prep_zmix4 (zmix.cpp:201-1076) differ from prep_zmix5 only in which pairs they list.  Sums over a group are exact
// integers (per-population partial Grams, per-population sum x and sum x^2 from the pack kernel); the tail is CalCor's
// (util.cpp:165-167): numer = n * Sxy - Sx * Sy, denor = sqrt(n * Sxx - Sx^2) * sqrt(n * Syy - Sy^2).
// pairs[k] = (i, j), i < j, rows of the problem; out is [n_group][n_pairs].
__global__ __launch_bounds__(256) void pair_cor_kernel(const Prob* __restrict__ probs, int prob, const int2* __restrict__ pairs,
                                                       long long n_pairs, const int* __restrict__ pop_group, int n_group,
                                                       double* __restrict__ out)
{
    const Prob& pb = probs[prob];
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_pairs) return;
    const int ri = pairs[k].x, rj = pairs[k].y;
    const int ti = ri / TILE, tj = rj / TILE;
    const int pair = pb.pair_lut[ti * pb.nT + tj];
    const float* tile_slab = pb.slab + (size_t)pair * pb.nseg * TILE * TILE;
    const int e = (ri % TILE) * TILE + (rj % TILE);
    const int P = pb.P;
    const int* sxi = pb.sx + (size_t)ri * P;
    const int* sxj = pb.sx + (size_t)rj * P;
    const int* sxxi = pb.sxx + (size_t)ri * P;
    const int* sxxj = pb.sxx + (size_t)rj * P;
    for (int g = 0; g < n_group; g++) {
        double n = 0, sumxy = 0, sumx = 0, sumy = 0, sumxsq = 0, sumysq = 0;          // exact integers in fp64
        for (int p = 0; p < P; p++) {
            if ((pop_group ? pop_group[p] : p) != g) continue;
            for (int s = pb.pop_seg0[p]; s < pb.pop_seg0[p + 1]; s++)
                sumxy += slab_val(tile_slab[(size_t)s * TILE * TILE + e], pb.gram_i8);
            n += (double)(pb.pop_raw_off[p + 1] - pb.pop_raw_off[p]);
            sumx += (double)sxi[p]; sumy += (double)sxj[p];
            sumxsq += (double)sxxi[p]; sumysq += (double)sxxj[p];
        }
        const double numer = n * sumxy - sumx * sumy;
        const double denor = sqrt(n * sumxsq - sumx * sumx) * sqrt(n * sumysq - sumy * sumy);
        out[(size_t)g * n_pairs + k] = numer / denor;
    }
}

void launch_pair_cor(const Prob* d_probs, int prob, const int2* d_pairs, long long n_pairs, const int* d_pop_group, int n_group,
                     double* d_out, hipStream_t s)
{
    if (n_pairs > 0)
        hipLaunchKernelGGL(pair_cor_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, d_probs, prob, d_pairs, n_pairs,
                           d_pop_group, n_group, d_out);
}

void launch_pop_cor(const Prob* d_probs, int prob, int npair, double* d_out, hipStream_t s)
{
    if (npair > 0) hipLaunchKernelGGL(pop_cor_kernel, dim3(npair), dim3(256), 0, s, d_probs, prob, d_out);
}

void launch_counts(const Prob* d_probs, int prob, int npair, long long* d_out, hipStream_t s)
{
    if (npair > 0) hipLaunchKernelGGL(counts_kernel, dim3(npair), dim3(256), 0, s, d_probs, prob, d_out);
}

}  // namespace gauss
