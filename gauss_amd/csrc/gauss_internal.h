// Internal declarations shared by the HIP translation units of libgauss_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>

namespace gauss {

constexpr int TILE = 128;      // Gram output tile edge per workgroup (4 waves x 64x64)
constexpr int KC = 64;         // packed K chunk in bytes (= samples); pops are padded to it
// chunk_unit_layout: inside every 32-sample group of a packed row the eight dwords (4 samples each) are stored transposed --
// sample dword d = 2 q + h sits at position 4 h + q -- so that the two lane halves of the 32 x 32 x 2 MFMA (half h reads bytes
// 16 h .. 16 h + 15 of the group) meet samples 8 q .. 8 q + 7 of the group in their dword q: a population whose last chunk
// holds r live samples needs only the first ceil(r / 8) dword steps of it (Item::chunk_live).  Sums over k do not depend on
// the order, every row is stored the same way, and nothing but the Gram kernels reads packed rows.
constexpr int SEG_MAX = 2048;  // K segment cap for small jobs (< 4 windows); larger jobs use 4096 (seg_max_for); any cap up to 8192
                               // keeps a partial sum below 2^24 (15 * 15 * 8192), the exactness bound of the f32 accumulators
constexpr int NB = 64;         // fp64 factor / solve block edge
constexpr int NR = 64;         // right-hand sides per solve panel (63 SNPs + the z1 column); the solve kernel is
                               // written for any multiple of 64 -- 128 was measured slower (2.23 vs 1.80 ms: half as
                               // many workgroups, each more than twice as long)
constexpr int NRU = NR - 1;
constexpr int WIN_QCAT = 1;          // gauss_window_desc.kind == GAUSS_WIN_QCAT
constexpr int WIN_LD = 2;            // gauss_window_desc.kind == GAUSS_WIN_LD
constexpr int TILE_GB11 = 1 << 30;   // tilemap.y flag: the entry is a job-wide B11 pair (Prob::gpair_*), seen from this window

// Pointers stored inside a Prob are loaded from memory, so the compiler could not infer their
// address space and would emit flat_* accesses.  Everything a Prob points to is device global
// memory: in the device pass the members are declared address_space(1) (same size and layout).
#if defined(__HIP_DEVICE_COMPILE__)
#define GP(T) T __attribute__((address_space(1)))*
#else
#define GP(T) T*
#endif

// Device-visible description of one window ("problem").  Built on the host by the planner,
// uploaded once per job.  All pointers are device pointers.
struct Prob {
    int mode, P;            // P = populations as seen by the kernels (1 pseudo-pop when pooled)
    int M, U;               // measured / unmeasured SNP rows
    int Mp, Up, Sp;         // row counts padded to TILE; Sp = Mp + Up
    int N;                  // samples (sum of population sizes)
    int Kp;                 // packed row length in bytes (multiple of KC)
    int nseg, npair, nT;    // K segments, tile pairs, row tiles (Sp / TILE)
    int Mld, nblk;          // solve leading dimension (M padded to NB) and block count
    int npanel;             // panels of the stand-alone solve (ceil(n_rhs / NRU)), 0 for LD-only problems
    int npi;                // panels of the fused path's right-hand sides [I | z1]: ceil((M + 1) / NR) (k_solve.hip)
    int Up128;              // n_rhs rounded up to 128: row count of a k block of Gsum
    int ld_only;            // 1: write out_ld (S x S) instead of B11/B21
    int kind;               // 0 imputation (z, info), 1 QCAT (correlation of whitened vectors)
    int n_head, n_predm;    // QCAT: measured rows before / inside the prediction window
    int n_rhs;              // right-hand sides of the solve: U (imputation) or n_predm + U (QCAT)
    int U_raw;              // rows of raw_u; U = U_raw * (number of codings)
    int code_blk[3];        // coding of B21 row block b: 0 additive, 1 dominant, 2 recessive (gauss.cpp:1196-1250)
    int n_run;              // 2-bit sources: number of source blocks (selected populations)
    int geno_fmt;           // 0: one byte per genotype (ASCII digit or small integer); 1: 2-bit packed blocks
    int gram_i8;            // 1: operands are raw codes and slabs hold int32 (i8 MFMA path); 0: e4m3 codes, f32 slabs
    int slab16;             // 1: partial slabs hold exact uint16 sums, two rows per dword (2-bit sources: codes <= 3 and
                            // segments <= 7168 samples keep a partial below 2^16); halves the slab traffic
    double lambda, eps, diag;
    long long ld_raw;
    GP(const uint8_t) raw_m;   // [M x ld_raw]
    GP(const uint8_t) raw_u;   // [U x ld_raw]
    // Row arrays come in two parts: the measured rows [0, Mp) behind `X`, the unmeasured rows [Mp, Sp) behind `X_u`
    // (indexed from 0).  Ordinarily X_u = X + Mp rows of one array.  In a job whose windows share their measured SNPs
    // (gauss_plan.cpp: shared measured rows) X points INTO the job-wide arrays of the chromosome's measured SNPs, at this
    // window's first one (g0): rows the windows have in common are packed once, and the tile pairs of B11 -- formed on
    // job-wide row tiles, `gpair_*` / `slab_g` -- are multiplied once for all the windows they lie in.
    GP(uint8_t) packed;        // [Mp x Kp]
    GP(uint8_t) packed_u;      // [Up x Kp]
    GP(int) sx;                // [.. x P] per-population sum x
    GP(int) sxx;               // [.. x P] per-population sum x^2
    GP(int) sx_u;
    GP(int) sxx_u;
    GP(const int) pop_raw_off; // [P+1]
    GP(const int) pop_pk_off;  // [P+1] packed column offsets (multiples of KC)
    GP(const double) pop_w;    // [P]
    GP(const double) pop_wf;   // [P] wgt_val * ((double)m / (m - 1))   (util.cpp:117-118)
    GP(const double) pop_md;   // [P] (double)m
    GP(const int) seg_pop;     // [nseg]
    GP(const int) seg_k0;      // [nseg] packed byte range
    GP(const int) seg_k1;
    GP(const int) pop_seg0;    // [P+1] segment range per population
    GP(const int) pair_ti;     // [npair]
    GP(const int) pair_tj;
    GP(const int) pair_lut;    // [nT x nT] -> pair index or -1
    GP(const uint8_t) word_pop;// [Kp/16] population of each packed 16-byte word
    GP(const int) rows_m;      // store row of each measured / unmeasured matrix row, or null (contiguous)
    GP(const int) rows_u;
    GP(const uint8_t) word_run;// 2-bit sources: [Kp/16] source block ("run") of each packed word
    GP(const int) run_pk_off;  // [n_run+1] packed column range of each run
    GP(const int) run_src;     // [n_run] byte offset of each run inside a source row
    GP(float) slab;            // [npair*nseg][TILE*TILE] exact integer partial Grams
    GP(double) rt_sd;          // [..] weighted: sqrt(self cov); pooled: sqrt(n*Sxx - Sx^2)
    GP(double) rt_wm;          // [..] weighted: sum_p w_p mu_p ; pooled: Sx (as double)
    GP(double) rt_mu;          // [.. x P] Sx_p / m_p
    GP(double) rt_wmu;         // [.. x P] w_p * mu_p
    GP(double) rt_sd_u;
    GP(double) rt_wm_u;
    GP(double) rt_mu_u;
    GP(double) rt_wmu_u;
    int g0;                    // shared measured rows: job-wide index of this window's first measured SNP (else 0)
    int n_gpair;               // shared measured rows: job-wide B11 tile pairs (0: B11's pairs are pair_ti / pair_tj / slab)
    GP(const int) gpair_ti;    // [n_gpair] job-wide row tiles of the pair (gi <= gj)
    GP(const int) gpair_tj;
    GP(float) slab_g;          // [n_gpair * nseg][TILE*TILE] partial Grams of the job-wide B11 pairs
    GP(const double) z1;       // [M]
    GP(double) A;              // [5][Mld x Mld] row-major: B11, B11 - eps*I, their factors L0, L1, working copy W0 of B11
    GP(double) Linv;           // [2][nblk][NB x NB] inverses of the diagonal Cholesky blocks
    GP(double) B21;            // [Upad x Mld] row-major (Upad = npanel*NRU rounded)
    GP(double) V;              // [max(npanel, npi)][Mld][NR]: fused path: [X | y] = L^-1 [I | z1] by panels of NR columns
    GP(double) Gsum;           // [ceil(Mld / 128)][Up128][3] z / info / v sums of every k block (impute_gemm_kernel)
    GP(double) Part;           // [2][npi][SOLVE_SPLIT][NB x NR] parked sums of a row's early products, by row parity
    GP(double) out_z;          // [U]
    GP(double) out_info;       // [U]
    GP(int) status;            // [4]: [0] fail flag matrix 0, [1] fail flag matrix 1, [2] nonfinite
    GP(double) out_ld;         // [S x S] for ld_only
    GP(const int) gene_off;    // gene batches: [n_gene+1], else null
    int n_gene;
    GP(long long) gene_out_off;// [n_gene] offsets into out_ld
};

// One unit of Gram work: a 128 x 128 tile pair times a run of consecutive K segments.  The kernel
// streams the whole run without draining its load pipeline and flushes the accumulators into one
// partial slab per segment.  Self-contained (64 bytes, fetched with scalar loads): no pointer
// chasing through the Prob before the first operand load.
struct Item {
    GP(const uint8_t) a;     // packed rows of tile ti (row 0, column 0)
    GP(const uint8_t) b;     // packed rows of tile tj
    GP(float) slab;          // slab of (pair, first segment of the run); later segments follow
    GP(const int) seg_k1;    // end column of each segment of the run
    GP(const uint32_t) chunk_live; // nibble c (word c / 8, bits 4 (c % 8) ..): 0 = all 64 samples of K chunk c are live; n = 1..7:
                             // only the chunk's first n UNITS of 8 samples are (the rest is the zero padding that ends a population
                             // block).  The pack kernel stores every 32-sample group dword-transposed (chunk_unit_layout below), which
                             // puts unit u of a chunk into dword u % 4 of BOTH lane halves of group u / 4 -- the MFMAs of a dead
                             // unit are not issued.  Dwords, read through the scalar cache (uniform_load)
    int Kp;                  // packed row stride
    int k0;                  // first column of the run
    int nseg;                // segments in the run
    int rows_a, rows_b;      // live rows of the two tiles (the rest is zero padding)
    int flags;               // bit 0: ti == tj (diagonal tile); bit 1: slabs are uint16 pairs (Prob::slab16); bit 4: a B11 item of a
                             // job built for a merged launch: counts itself off in the launch's `b11_done[0]` when one is passed
                             // (k_gram.hip); bit 5: a B21 item of an "early" window, counted in `b11_done[8]`
};
static_assert(sizeof(Item) == 64, "work items are fetched as one 64-byte descriptor");

template <typename T> using gptr = T __attribute__((address_space(1)))*;
template <typename T> __device__ __forceinline__ gptr<T> G(T* p) { return (gptr<T>)p; }
template <typename T> __device__ __forceinline__ gptr<T> G(gptr<T> p) { return p; }

// Wave-uniform table reads inside the Gram kernels' K loops (segment ends, half-chunk flags).  The tables are reached
// through pointers that were themselves loaded from memory, so the compiler cannot prove them read-only and emits VECTOR
// loads -- and a vector load's s_waitcnt vmcnt(0) also waits for every LDS-DMA group issued before it, i.e. for the
// operand chunks that were just requested for LATER iterations (round 2's half-chunk flag was a global_load_ubyte in the
// loop: every chunk's MFMAs waited for the next chunk's DMA).  Reading them through the constant address space makes
// them scalar loads (lgkmcnt), which leave the DMA queue alone.
#if defined(__HIPCC__)
template <typename T>
__device__ __forceinline__ T uniform_load(GP(const T) p, int idx)
{
    return ((const T __attribute__((address_space(4)))*)p)[idx];
}
// live units (1..8) of K chunk `chunk` from the nibble table
__device__ __forceinline__ int chunk_live_units(uint32_t word, int chunk)
{
    const int n = (int)((word >> (4 * (chunk & 7))) & 15u);
    return n ? n : 8;
}
#endif

// hipFuncSetAttribute is per device: `set` runs once for every device a launcher is used on (a process may drive
// several GPUs), and a second thread that arrives on the same device meanwhile (another context of bench --streams, a
// farm thread) waits until the attributes are in place instead of launching with the default dynamic-LDS limit.
struct DeviceOnce {
    std::atomic<unsigned long long> done{0};
    std::mutex mu;
    template <typename F> void run(F&& set)
    {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned long long bit = 1ull << (d & 63);
        if (done.load(std::memory_order_acquire) & bit) return;
        std::lock_guard<std::mutex> lock(mu);
        if (done.load(std::memory_order_relaxed) & bit) return;
        set();
        done.fetch_or(bit, std::memory_order_release);
    }
};

// ---- launchers (host functions defined in the .hip files) ----
void launch_pack_stats(const Prob* d_probs, const int2* d_rowmap, int n_rows, hipStream_t s);
// d_b11_done (may be null): items with flag bit 4 count themselves off there when their slabs are out (k_gram.hip)
void launch_gram(const Item* d_items, int n_items, int dtype_i8, hipStream_t s, unsigned long long* d_b11_done = nullptr);
// one wave that returns when *d_count >= target (bounded; on timeout d_status[0 .. n_status) = 1)
// bound_us: give-up bound in microseconds (at least two seconds are always granted)
void launch_wait_count(const unsigned long long* d_count, unsigned long long target, int* d_status, int n_status, hipStream_t s,
                       double bound_us = 0.0);
void launch_wait_count_for(const unsigned long long* d_count, unsigned long long target, int* d_status, hipStream_t s, double bound_us);
void launch_count_up(unsigned long long* d_count, hipStream_t s);
void launch_epilogue(const Prob* d_probs, const int2* d_tilemap, int n_tiles, int max_pop, int dtype_i8, hipStream_t s);
void launch_epilogue_b11_lite(const Prob* d_probs, const int2* d_tilemap, int n_tiles, int dtype_i8, hipStream_t s);
void launch_pop_cor(const Prob* d_probs, int prob, int npair, double* d_out, hipStream_t s);
void launch_pair_cor(const Prob* d_probs, int prob, const int2* d_pairs, long long n_pairs, const int* d_pop_group, int n_group,
                     double* d_out, hipStream_t s);
void launch_gene_epilogue(const Prob* d_probs, int prob, int n_gene, hipStream_t s);
void launch_factor_step(const Prob* d_probs, int n_prob, int step, int max_nblk, int max_npanel, int split, int own_panel, hipStream_t s);
// small-footprint twins (k_solve_lite.hip): same bits, built to run beside the Gram kernel
void launch_factor_step_lite(const Prob* d_probs, int n_prob, int step, int max_nblk, int max_npanel, int split, hipStream_t s);
void launch_solve_last_lite(const Prob* d_probs, const int2* d_panelmap, int n_panels, int max_nblk, int split, hipStream_t s);
void launch_shift_cert(const Prob* d_probs, int n_prob, hipStream_t s);
void launch_solve(const Prob* d_probs, const int2* d_panelmap, int n_panels, hipStream_t s);
void launch_impute_gemm(const Prob* d_probs, const int2* d_gmap, int n_tiles, int u_tile, const int2* d_fmap, int n_chunks, hipStream_t s);
void launch_solve_last(const Prob* d_probs, const int2* d_panelmap, int n_panels, int max_nblk, int split, hipStream_t s);
void launch_counts(const Prob* d_probs, int prob, int npair, long long* d_out, hipStream_t s);
void launch_pack2bit(const uint8_t* d_in, long long ld_in, uint8_t* d_out, long long ld_out, int n_snp,
                     const int* d_pop_off, const int* d_blk_off, int n_pop, hipStream_t s);
void launch_h2d_copy(void* d_dst, const void* pinned_src, size_t bytes, hipStream_t s);
// Matrix exports (raw LD export / want_mats): `n` device matrices [rows x pitch] are written as compact [rows x width] doubles into the
// job's pinned export mirror by ONE kernel (the stores cross PCIe), instead of a pitched copy per matrix and a compaction on the host.
struct ExportD {
    const double* src;      // device matrix, row pitch `pitch` doubles
    double* dst;            // compact [rows x width] in the pinned mirror (device-visible host memory)
    int rows, width, pitch, pad_;
};
void launch_export_rows(const ExportD* d_exports, int n, long long max_elems, hipStream_t s);
void launch_synth(uint8_t* d_out, int n_snp, long long ld, const int* d_pop_off, int n_pop,
                  int n_samples, const float* d_thr, const float* d_rho, uint64_t seed,
                  hipStream_t s);

}  // namespace gauss
