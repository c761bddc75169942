/*
 * gauss_hip.h -- C ABI of libgauss_hip.so: the MI355X (gfx950) replacement for the numeric
 * hot path of statsleelab/gauss (LD matrix + DIST/DISTMIX Z-score imputation + JEPEG LD step).
 *
 * Boundary: the Rcpp entry points computeLD()/dist()/distmix()/jepeg()/
 * jepegmix() keep their signatures (R/RcppExports.R:28,57,74,88,102).  Inside the drivers, the
 * code between "genotype strings are in memory" (ReadGenotype, src/gauss.cpp:720-785) and
 * "z / info / cormat doubles" is replaced by one call into this library:
 *
 *   reference code being replaced                          entry point here
 *   -----------------------------------------------------  ---------------------------------
 *   computeLD.cpp:95-116  (CalWgtCov pair loops)           gauss_ld(mode=WEIGHTED, diag=1.0)
 *   dist.cpp:129-227      run_dist                         gauss_impute_window(mode=POOLED)
 *   distmix.cpp:138-253   run_distmix                      gauss_impute_window(mode=WEIGHTED)
 *   qcat.cpp:134-262      run_qcat                         gauss_impute_window(kind=QCAT, POOLED)
 *   qcatmix.cpp:145-286   run_qcatmix                      gauss_impute_window(kind=QCAT, WEIGHTED)
 *   gene.cpp:305-315      CorG via CalCor   (jepeg)        gauss_gene_ld_batch(mode=POOLED)
 *   gene.cpp:571-586      CorG via CalWgtCov (jepegmix)    gauss_gene_ld_batch(mode=WEIGHTED)
 *   util.cpp:49-70,103-124 (sumxy accumulation)            gauss_gram_counts (integer parity)
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary.  Every call returns 0 on success
 *     or a negative GAUSS_E_* code; gauss_last_error() gives a thread-local message.  The
 *     Rcpp driver turns non-zero into Rcpp::stop(msg) (reference convention, dist.cpp:150).
 *   - the caller owns every input and output buffer; the library keeps nothing past return
 *     except inside handles it created (contexts, jobs).
 *   - genotype matrices are SNP-major: one SNP = one row of n_samples bytes, row stride `ld`
 *     bytes.  A byte is either the ASCII digit the reference stores ('0','1','2';
 *     Snp::genotype_vec_, src/snp.h:109) or the raw count 0..2; both decode as (byte & 0x0F).
 *     A row is the concatenation, in panel order, of the per-population strings of the
 *     selected populations: population p occupies columns [pop_off[p], pop_off[p+1]).
 *   - host-pointer calls are blocking (R's .Call semantics); the *_dev / job API is
 *     asynchronous on the context's streams and is what the multi-window farm uses.
 */
#ifndef GAUSS_HIP_H
#define GAUSS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAUSS_MODE_POOLED   0  /* CalCor, util.cpp:49-70   (dist, jepeg)                 */
#define GAUSS_MODE_WEIGHTED 1  /* CalWgtCov, util.cpp:103-124 (distmix, computeLD, jepegmix) */

#define GAUSS_OK            0
#define GAUSS_E_INVALID    -1  /* bad argument                                   */
#define GAUSS_E_DEVICE     -2  /* HIP runtime error                              */
#define GAUSS_E_NOMEM      -3  /* host or device allocation failed               */
#define GAUSS_E_RANGE      -4  /* exact-integer Gram range would be exceeded     */

/* per-window status bits returned in out_status */
#define GAUSS_ST_CLAMPED    1  /* MakePosDef rebuilt B11 (some eigenvalue < min_abs_eig, util.cpp:310) */
#define GAUSS_ST_NONFINITE  2  /* B11 not finite / not factorisable: outputs are NaN like the reference's */

typedef struct gauss_ctx gauss_ctx;
typedef struct gauss_job gauss_job;

/* One imputation window (one R call of dist()/distmix()).  geno_m/geno_u may be host or
 * device pointers (see gauss_job_create's `on_device`). */
typedef struct gauss_window_desc {
    int mode;                 /* GAUSS_MODE_*                                                      */
    int n_pop;                /* P: selected populations (pooled mode may pass 1)                  */
    const int32_t* pop_off;   /* [P+1] column ranges; pop_off[P] = n_samples                       */
    const double* pop_wgt;    /* [P] weights in the same order (ignored for POOLED; may be NULL)   */
    int n_measured;           /* M: type-1 SNPs of the extended window (dist.cpp:137-138)          */
    int n_unmeasured;         /* U: type-0 SNPs of the prediction window (dist.cpp:135-136); 0 = LD only */
    const uint8_t* geno_m;    /* [M x ld] measured genotypes                                       */
    const uint8_t* geno_u;    /* [U x ld] unmeasured genotypes                                     */
    int64_t ld;               /* row stride in bytes (>= n_samples)                                */
    const double* z1;         /* [M] measured z-scores (host pointer, always)                      */
    double lambda;            /* ridge added to diag(B11): Arguments::lambda = 0.1 (gauss.cpp:20)  */
    double min_abs_eig;       /* MakePosDef floor: 1e-5 (gauss.cpp:21)                             */
    double* out_z;            /* [U] imputed z / sqrt(info)    (dist.cpp:200)  host pointer        */
    double* out_info;         /* [U] info = |b21 B11^-1 b12|   (dist.cpp:198)  host pointer        */
    int32_t* out_status;      /* [1] GAUSS_ST_* bits                           host pointer        */
    double* out_b11;          /* optional [M x M] B11 incl. lambda (symmetric)  host pointer / NULL */
    double* out_b21;          /* optional [U x M] row-major                     host pointer / NULL */
    /* ---- QCAT / QCATMIX windows (run_qcat qcat.cpp:134-262, run_qcatmix qcatmix.cpp:145-286) ----
     * kind = GAUSS_WIN_QCAT tests, for the n_pred_measured measured SNPs of the prediction window
     * (rows n_head_measured .. of geno_m) and then the U unmeasured ones, the Pearson correlation
     * between L^-1 Z1 and L^-1 b (b = their row of B11 / B21; L = Cholesky factor of B11). */
    int kind;                 /* GAUSS_WIN_IMPUTE (0, default), GAUSS_WIN_QCAT or GAUSS_WIN_LD        */
    int n_head_measured;      /* measured SNPs with bp < start_bp (qcat.cpp:146-147)               */
    int n_pred_measured;      /* measured SNPs inside the prediction window (qcat.cpp:148-149)     */
    double eig_cutoff;        /* CountPC cutoff: Arguments::eig_cutoff = 0.01 (gauss.cpp:22)       */
    double* out_r;            /* [n_pred_measured + U] correlations, measured first   host pointer */
    int32_t* out_num_eig;     /* [1] CountPC(B11, eig_cutoff)                          host pointer */
    /* ---- raw LD export (prep_qcat prep_qcat.cpp:104-132, prep_recessive_impute prep_qcatmix.cpp:136-221) ----
     * kind = GAUSS_WIN_LD stops after the LD step: out_b11 = B11 (diagonal 1 + lambda; pass lambda = 0 for
     * the reference's 1.0) and out_b21 = the LD of the geno_u rows against the measured rows; no z1 needed.
     * u_codings (any kind, 0 = additive only) makes the device derive several codings of every geno_u row
     * (gauss.cpp:1196-1250): B21 then holds one block of n_unmeasured rows per selected coding, in the
     * order additive, dominant (0/1/2 -> 0/1/1), recessive (0/1/2 -> 0/0/1); out_b21 is
     * [n_codings * U x M].  Codes outside 0..2 are left unchanged, as in the reference. */
    int u_codings;            /* bit mask of GAUSS_CODE_*                                          */
    /* ---- packed panel rows -------------------------------------------------
     * geno_format = GAUSS_GENO_2BIT: a genotype row is a 2-bit stream (sample s of a population block at bits
     * 2*(s%4) of byte s/4, codes 0..2), each selected population's block starting on a 16-byte boundary and
     * zero padded to a multiple of 64 samples.  pop_src_off[q] is the byte offset of selected population q
     * inside a row (NULL: the blocks follow each other in pop_off order); ld is the row stride in bytes.
     * rows_m / rows_u (either format): when non-NULL, geno_m / geno_u point at a row store (e.g. a whole
     * chromosome resident in HBM, see gauss_store_upload) and matrix row r is store row rows_x[r]. */
    int geno_format;          /* GAUSS_GENO_U8 (0, default) or GAUSS_GENO_2BIT                      */
    const int32_t* rows_m;    /* [M] store row of each measured SNP, or NULL          host pointer */
    const int32_t* rows_u;    /* [U] store row of each geno_u SNP, or NULL            host pointer */
    const int32_t* pop_src_off; /* [P] 2-bit format: byte offset of each population block, or NULL */
} gauss_window_desc;

#define GAUSS_GENO_U8   0
#define GAUSS_GENO_2BIT 1

#define GAUSS_WIN_IMPUTE 0
#define GAUSS_WIN_QCAT   1
#define GAUSS_WIN_LD     2

#define GAUSS_CODE_ADDITIVE  1
#define GAUSS_CODE_DOMINANT  2
#define GAUSS_CODE_RECESSIVE 4

/* ---- pinned host buffers ---------------------------------------------------------------------
 * Page-locked host memory for the genotype matrices a driver marshals (pack_genotypes): the
 * upload of a window is then one DMA at PCIe speed instead of a staged copy of pageable memory (a full-size
 * window is 101 MB of genotype bytes).  Plain pointers; any host pointer remains valid input. */
int gauss_pinned_alloc(gauss_ctx* ctx, int64_t bytes, void** out_host_ptr);
int gauss_pinned_free(gauss_ctx* ctx, void* host_ptr);

/* ---- resident row store ---------------------------------------------------------------------
 * Copies `bytes` of genotype rows to the context's GPU and returns the device pointer to use as
 * geno_m / geno_u (with gauss_job_create's on_device = 1 and rows_m / rows_u).  288 GB of HBM hold a
 * whole 2-bit panel (33 k samples x 10 M SNPs = 82 GB), so a panel is uploaded once, not per window. */
int gauss_store_upload(gauss_ctx* ctx, const void* host_rows, int64_t bytes, void** out_device_ptr);
/* The rows are `bytes` bytes of an open file from file_offset on (a packed panel's genotype section): read with pread
 * straight into the pinned staging buffers -- no page of a mapping of the file is touched (a memcpy out of a fresh mapping
 * costs a page fault per 4 KB).  Blocking, like gauss_store_upload. */
int gauss_store_upload_fd(gauss_ctx* ctx, int fd, int64_t file_offset, int64_t bytes, void** out_device_ptr);
int gauss_store_free(gauss_ctx* ctx, void* device_ptr);
/* The same store filled piece by piece: gauss_store_alloc reserves `bytes` on the GPU (contents undefined until filled) and
 * gauss_store_fill copies host_rows[offset .. offset + len) to the same offsets of the store, returning when they have landed.
 * The copy travels through the pinned double buffers on a queue of its own, so whatever the context's other queues are
 * computing keeps running: a driver that walks a chromosome fills the rows of batch k + 1 while batch k computes
 * (gauss_host_impute_chromosome on a panel that is not resident yet).  A job may only name rows that have been filled. */
int gauss_store_alloc(gauss_ctx* ctx, int64_t bytes, void** out_device_ptr);
int gauss_store_fill(gauss_ctx* ctx, void* device_ptr, const void* host_rows, int64_t offset, int64_t len);
/* gauss_store_fill from a file: store bytes [offset, offset + len) come from file bytes [file_offset + offset, ...). */
int gauss_store_fill_fd(gauss_ctx* ctx, void* device_ptr, int fd, int64_t file_offset, int64_t offset, int64_t len);
/* The same upload without waiting for it: returns at once with the device pointer; a library thread streams the rows
 * IN ORDER (pinned double buffers, a stream of its own).  host_rows must stay valid until the upload is complete.
 * gauss_store_wait(ctx, ptr, n) makes the context's main stream wait until the first n bytes have landed -- a job queued
 * afterwards that reads rows below that mark starts while the rest of the panel is still crossing PCIe -- and blocks the
 * host only until those bytes have been queued; n <= 0: the whole store, and the host waits for completion.  Stores
 * made by gauss_store_upload need no wait (it is a no-op on them); a pointer that is not (or no longer) a row store of this
 * context is an error (a store freed by another call is not waited for, nor read). */
int gauss_store_upload_async(gauss_ctx* ctx, const void* host_rows, int64_t bytes, void** out_device_ptr);
/* The rows are `bytes` bytes of an open file starting at file_offset (a packed panel's genotype section): read with
 * pread straight into the pinned staging buffers -- no page of the caller's address space is touched, so the upload
 * does not contend with the threads that parse the study beside it.  fd must stay open until the upload is complete. */
int gauss_store_upload_fd_async(gauss_ctx* ctx, int fd, int64_t file_offset, int64_t bytes, void** out_device_ptr);
int gauss_store_wait(gauss_ctx* ctx, const void* device_ptr, int64_t bytes_needed);

/* ---- context ------------------------------------------------------------------------------- */
int gauss_hip_init(int device, gauss_ctx** out_ctx);
/* Number of HIP devices visible to the process (no context is created; a farm rank picks
 * LOCAL_RANK modulo this), and the device index a context was created on. */
int gauss_hip_device_count(int* out_n);
int gauss_hip_device_of(const gauss_ctx* ctx);
/* Lifetime rule.  A context may be destroyed while jobs (gauss_job_create) and row stores (gauss_store_upload) made on
 * it are still alive: gauss_hip_destroy waits for their queued work, frees every device / pinned block they hold and
 * leaves the job handles behind as empty shells.  After that
 *   - gauss_job_destroy(job) is still required (it frees the shell) and is safe in any order with the context;
 *   - every other call on such a job returns GAUSS_E_INVALID ("the job's context has been destroyed");
 *   - device pointers returned by gauss_store_upload are dead; gauss_store_free on them must not be called.
 * So the destructors of an Rcpp driver's RAII wrappers may run in whatever order an Rcpp::stop unwinds them
 * (RcppExports.cpp:66,81: exceptions leave the driver through the generated try / catch).  Destroying a context
 * must not race with calls that use it from other threads.  (Round 2 had no such rule: a job destroyed after its
 * context wrote into the freed context -- the heap corruption behind the one process abort of that round.) */
void gauss_hip_destroy(gauss_ctx* ctx);
/* A number that identifies the context for the life of the process and is never reused (a pointer may be: a new
 * context can land at a freed context's address).  Caches keyed per context key on this. */
uint64_t gauss_hip_context_id(const gauss_ctx* ctx);
/* fn(ctx, id, user) is called at the start of every gauss_hip_destroy, while the context is still whole: whoever
 * caches device memory per context (libgauss_host's resident panels) releases it there.  Process-wide, idempotent. */
int gauss_hip_add_destroy_hook(void (*fn)(gauss_ctx* ctx, uint64_t id, void* user), void* user);
/* Freed job workspaces are kept per context for reuse (hipFree would stall a pipeline that retires job k while job
 * k + 1 runs), up to a third of the device's memory.  Every allocation of the library flushes this cache and retries
 * before it reports GAUSS_E_NOMEM; gauss_hip_trim_cache gives the memory back at once (it waits for the context's
 * streams first), e.g. before another context or another library needs the device. */
int gauss_hip_trim_cache(gauss_ctx* ctx, int64_t* out_bytes_freed);
/* How the runs of this context's jobs were queued, since the context was made:
 *   out4[0] runs queued in the merged form (ONE Gram launch whose B11 items count themselves off for a waiting kernel at the
 *           head of the chain queue);
 *   out4[1] runs of jobs built for that form that were queued as two launches joined by an event instead, because the
 *           context could not be sure of a hardware queue per stream (the runtime shares queues once a priority class holds
 *           more streams than GPU_MAX_HW_QUEUES -- a second context on the device is enough);
 *   out4[2] merged runs whose waiting kernel gave up at its bound; gauss_job_fetch ran each of them again in the
 *           two-launch form inside the same call;
 *   out4[3] of those, the ones whose second form failed too (gauss_job_fetch returned GAUSS_E_DEVICE).
 * One call of the reference is one window or an error (dist.cpp:30-126): a give-up is never handed to the caller as results. */
int gauss_hip_counters(gauss_ctx* ctx, int64_t* out4);
/* The same four counts for ONE job (its own runs only): what a caller with several jobs -- or several calls -- in flight on a
 * context sums over ITS jobs (the context's counters are shared by everything that runs on it). */
int gauss_job_counters(gauss_job* job, int64_t* out4);
/* What the merged form's condition rests on, for diagnostics: out4[0] / out4[1] the library's live high- / low-priority streams on
 * the context's device (all contexts of the process), out4[2] the hardware queues the runtime makes per priority class
 * (GPU_MAX_HW_QUEUES, default 4; 0: unknown), out4[3] = 1 if gauss_hip_init SAW the context's chain and low-priority queues run a
 * kernel beside a later kernel of its main queue (a probe of a few microseconds: three different hardware queues), else 0.  A run
 * is queued merged only while out4[3] = 1 and both stream counts are within out4[2]. */
int gauss_hip_queues(gauss_ctx* ctx, int32_t* out4);
const char* gauss_last_error(void);
const char* gauss_hip_version(void);
/* Hash of the sources this library was built from (gauss_amd/build.py:source_hash): profiles/<tag>_provenance.json record it,
 * and bench.py only quotes a profile-derived figure (roofline.traffic) when it matches. */
const char* gauss_hip_source_hash(void);

/* Arithmetic of the LD Gram kernel for jobs created afterwards on this context.
 *   GAUSS_GRAM_F32 (default): v_mfma_f32_32x32x2_f32 on e4m3-coded operands, exact f32 integer sums.
 *   GAUSS_GRAM_I8           : v_mfma_i32_32x32x32_i8 on the raw codes, exact int32 sums.
 * Both produce bit-identical LD, z and info (the partial sums are the same integers). */
#define GAUSS_GRAM_F32 0
#define GAUSS_GRAM_I8  1
int gauss_hip_set_gram_dtype(gauss_ctx* ctx, int dtype);

/* ---- blocking host-pointer calls (what the Rcpp drivers bind) ------------------------------ */

/* LD matrix among S SNPs.  mode WEIGHTED + diag 1.0 = computeLD.cpp:95-116;
 * mode POOLED + diag 1+lambda = CorG of gene.cpp:305-315.  out_cor: S x S doubles (symmetric,
 * so column-major == row-major; fits an Rcpp::NumericMatrix directly). */
int gauss_ld(gauss_ctx* ctx, int mode, const uint8_t* geno, int n_snp, int64_t ld,
             const int32_t* pop_off, const double* pop_wgt, int n_pop, double diag,
             double* out_cor);

/* One window: run_dist (POOLED) / run_distmix (WEIGHTED).  Blocking. */
int gauss_impute_window(gauss_ctx* ctx, const gauss_window_desc* win);

/* LD blocks of many genes in one launch: gene g owns SNP rows [gene_off[g], gene_off[g+1]).
 * out_blocks receives the n_g x n_g blocks back to back (block g at sum_{h<g} n_h^2), each
 * with `diag` on its diagonal. */
int gauss_gene_ld_batch(gauss_ctx* ctx, int mode, const uint8_t* geno, int n_snp, int64_t ld,
                        const int32_t* pop_off, const double* pop_wgt, int n_pop,
                        const int32_t* gene_off, int n_gene, double diag, double* out_blocks);

/* gauss_ld / gauss_gene_ld_batch over rows of a row store instead of a contiguous byte matrix: `store` holds rows of
 * `ld` bytes in geno_format (GAUSS_GENO_U8 or GAUSS_GENO_2BIT, with pop_src_off as in gauss_window_desc), SNP s is store
 * row rows[s] (rows NULL: row s).  on_device != 0: `store` is a device pointer, e.g. a packed panel made resident
 * with gauss_store_upload -- nothing but the row indices travels to the GPU and only the LD blocks come back. */
int gauss_ld_rows(gauss_ctx* ctx, int mode, const uint8_t* store, int64_t ld, int geno_format, const int32_t* rows, int n_snp,
                  const int32_t* pop_off, const int32_t* pop_src_off, const double* pop_wgt, int n_pop, double diag,
                  int on_device, double* out_cor);
int gauss_gene_ld_batch_rows(gauss_ctx* ctx, int mode, const uint8_t* store, int64_t ld, int geno_format, const int32_t* rows,
                             int n_snp, const int32_t* pop_off, const int32_t* pop_src_off, const double* pop_wgt, int n_pop,
                             const int32_t* gene_off, int n_gene, double diag, int on_device, double* out_blocks);

/* Per-population Pearson correlations of every SNP pair (prep_zmix5, zmix.cpp:158-176: CalCor on each
 * population's genotype strings, util.cpp:153-169).  out is [n_pop][n_snp*(n_snp-1)/2], population-major,
 * pairs in the reference's row order (i ascending, then j > i): column k+1 of the reference's column-major
 * NumericMatrix data_mat is out + k * npairs.  A population in which a SNP is monomorphic gives NaN there,
 * as in the reference. */
int gauss_ld_per_pop(gauss_ctx* ctx, const uint8_t* geno, int n_snp, int64_t ld,
                     const int32_t* pop_off, int n_pop, double* out);

/* The same correlations for LISTED pairs only, per population or per GROUP of populations pooled: what the other
 * prep_zmix selectors need (zmix.cpp:201-1076 -- prep_zmix, prep_zmix2, prep_zmix3, prep_zmix4 list pairs by interval /
 * offset / steps; prep_zmix5_sup pools the populations of a super-population, CalCorSup zmix.cpp:1221-1246).  pair k is
 * (pair_i[k], pair_j[k]) with 0 <= i < j < n_snp; pop_group[p] in 0..n_group-1 names the group of population p (NULL:
 * every population its own group, n_group ignored).  Only the tile pairs the listed pairs touch are multiplied.
 * out is [n_group][n_pairs]. */
int gauss_ld_per_pop_pairs(gauss_ctx* ctx, const uint8_t* geno, int n_snp, int64_t ld, const int32_t* pop_off, int n_pop,
                           const int32_t* pop_group, int n_group, const int32_t* pair_i, const int32_t* pair_j, int64_t n_pairs,
                           double* out);

/* Exact co-occurrence counts sum_n x_i[n] x_j[n] over all columns -- the integer the reference
 * accumulates as `sumxy` (util.cpp:62,114).  out: S x S int64, row-major.  Integer parity hook. */
int gauss_gram_counts(gauss_ctx* ctx, const uint8_t* geno, int n_snp, int n_samples, int64_t ld,
                      int64_t* out_counts);

/* ---- asynchronous multi-window jobs (farm harness, bench) ----------------------------------- */

/* Build a job of n_win windows.  If on_device != 0, geno_m/geno_u are device pointers that stay
 * resident for the life of the job; otherwise they are host pointers and are uploaded once here.
 * All windows of a job run through the same batched launches. */
int gauss_job_create(gauss_ctx* ctx, const gauss_window_desc* wins, int n_win, int on_device,
                     gauss_job** out_job);
/* Enqueue one full pass (pack -> Gram -> LD epilogue -> factorisation + inverse factor -> closing product) on the
 * context's streams; returns at once.  Up to TWO runs of a job may be in flight: a second gauss_job_run before the first
 * has been fetched is allowed (the job keeps two result mirrors), a third is refused (GAUSS_E_INVALID). */
int gauss_job_run(gauss_job* job);
/* Wait for the OLDEST run that has not been fetched and copy its z / info / status (and optional b11/b21) to the host
 * pointers of each window descriptor.  run, fetch, run, fetch ... behaves as before; run, run, fetch, run, fetch ...
 * keeps the GPU busy while the host handles the previous run's results. */
int gauss_job_fetch(gauss_job* job);
void gauss_job_destroy(gauss_job* job);
/* Device time from the moment `first` started to run until `last` had delivered its results (HIP events on the
 * context's stream; both jobs must have run, on the same context).  With a pipeline of jobs this is the span the
 * GPU was busy with them, whatever the host did meanwhile. */
int gauss_job_span_ms(gauss_job* first, gauss_job* last, double* out_ms);

/* Profiling hooks for bench.py: HIP events are recorded around every launch of the Gram kernel
 * on the job's own stream while enabled. */
int gauss_job_profile(gauss_job* job, int enable);
/* kernel 0 = gram (LD GEMM), 1 = pack/stats, 2 = LD epilogue, 3 = factor, 4 = solve.
 * Returns accumulated milliseconds and launch count since profiling was enabled. */
int gauss_job_profile_get(gauss_job* job, int kernel, double* out_ms, int64_t* out_launches);
/* Algorithmic work of one run of the job (SURVEY.md section 8d definitions). */
int gauss_job_work(gauss_job* job, double* out_ld_flops, double* out_solve_flops,
                   double* out_bytes, int64_t* out_imputed_snps);

/* Launch geometry of the job: out[0] = Gram work items, out[1] = MFMA flops actually issued per run
 * (padding included, skipped padding halves excluded), out[2] = bytes of partial-Gram slabs,
 * out[3] = device workspace bytes.  Diagnostic only. */
int gauss_job_stats(gauss_job* job, double* out4);

/* Device-side conversion of one-byte genotype rows [n_snp x ld_in] into GAUSS_GENO_2BIT rows [n_snp x ld_out]
 * (population blocks consecutive, 16-byte aligned, zero padded to 64 samples; ld_out >= their total and a
 * multiple of 16).  Both pointers are device pointers.  Used to build a resident row store from rows that are
 * already in HBM (bench plumbing; a packed panel file is uploaded as it is). */
int gauss_pack2bit_device(gauss_ctx* ctx, const uint8_t* d_in, int64_t ld_in, uint8_t* d_out, int64_t ld_out,
                          int n_snp, const int32_t* pop_off, int n_pop);

/* Synthetic genotype generator on the device (bench plumbing; mirrors gauss_amd/synth.py's
 * model with a counter-based RNG).  Writes n_snp rows of n_samples bytes {0,1,2} at d_out with
 * row stride ld.  thr is a host array [n_snp x n_pop] of per-population latent thresholds,
 * rho a host array [n_snp] of AR(1) coefficients. */
int gauss_synth_device(gauss_ctx* ctx, uint8_t* d_out, int n_snp, int64_t ld,
                       const int32_t* pop_off, int n_pop, const float* thr, const float* rho,
                       uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* GAUSS_HIP_H */
