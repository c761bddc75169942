/*
 * gauss_host.h -- C ABI of libgauss_host.so: the host side of the hot path, i.e. the
 * reference entry points with their reference argument lists, in plain C.
 *
 *   reference (Rcpp, src/RcppExports.cpp:335-340)          here
 *   -----------------------------------------------------  ------------------------------
 *   computeLD(chr,start_bp,end_bp,pop_wgt_df,input_file,    gauss_host_computeLD
 *             reference_index_file,reference_data_file,
 *             reference_pop_desc_file,af1_cutoff)           computeLD.cpp:26-166
 *   dist(chr,start_bp,end_bp,wing_size,study_pop,...)       gauss_host_dist      dist.cpp:30-126
 *   distmix(chr,start_bp,end_bp,wing_size,pop_wgt_df,...)   gauss_host_distmix   distmix.cpp:30-135
 *   jepeg(study_pop,input_file,annotation_file,...)         gauss_host_jepeg     jepeg.cpp:28-153
 *   jepegmix(pop_wgt_df,input_file,annotation_file,...)     gauss_host_jepegmix  jepegmix.cpp:26-161
 *   qcat(chr,start_bp,end_bp,wing_size,study_pop,...)       gauss_host_qcat      qcat.cpp:30-132
 *   qcatmix(chr,start_bp,end_bp,wing_size,pop_wgt_df,...)   gauss_host_qcatmix   qcatmix.cpp:30-140
 *   prep_qcat(chr,start_bp,end_bp,wing_size,study_pop,...)  gauss_host_prep_qcat prep_qcat.cpp:16-205
 *   prep_recessive_impute(chr,...,pop_wgt_df,...)           gauss_host_prep_recessive_impute
 *                                                           prep_qcatmix.cpp:36-316
 *   prep_zmix5(input_file,...,percentile,interval)          gauss_host_prep_zmix5 zmix.cpp:44-190
 *
 * Same argument meaning, same defaults (af1_cutoff NaN = R's NULL -> 0.01, dist.cpp:53-57), same
 * row order (std::map order on (chr,bp,a1,a2), gauss.h:72-99), same column names and types as
 * the reference's DataFrames, same error texts (returned through gauss_host_last_error instead of
 * Rcpp::stop).  A pop_wgt_df is passed as two parallel arrays (names, weights).
 *
 * The host data layer (text + BGZF readers, allele matching, AF filters, window partition:
 * src/gauss.cpp:121-190, 293-399, 431-518, 543-604, 631-693, 720-785, 951-1117, 1275-1439) is
 * restated here in C++; the numeric part is delegated to libgauss_hip.so (include/gauss_hip.h).
 * The *_prepare functions run the host part alone (no GPU needed) and expose what would be handed
 * to the GPU; the farm harness and the CPU tests use them.
 */
#ifndef GAUSS_HOST_H
#define GAUSS_HOST_H

#include <stdint.h>

#include "gauss_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gauss_table gauss_table;     /* a result DataFrame (column store)            */
typedef struct gauss_prepared gauss_prepared; /* one window/gene set after the host data layer */

#define GAUSS_COL_STR 0
#define GAUSS_COL_INT 1
#define GAUSS_COL_DBL 2

#define GAUSS_KIND_COMPUTELD 0
#define GAUSS_KIND_DIST      1
#define GAUSS_KIND_DISTMIX   2
#define GAUSS_KIND_JEPEG     3
#define GAUSS_KIND_JEPEGMIX  4
#define GAUSS_KIND_QCAT      5
#define GAUSS_KIND_QCATMIX   6
#define GAUSS_KIND_PREP_QCAT 7
#define GAUSS_KIND_PREP_RECESSIVE 8

const char* gauss_host_last_error(void);

/* ---- result tables --------------------------------------------------------------------------- */
int gauss_table_nrow(const gauss_table* t);
int gauss_table_ncol(const gauss_table* t);
const char* gauss_table_colname(const gauss_table* t, int col);
int gauss_table_coltype(const gauss_table* t, int col);
const char* gauss_table_str(const gauss_table* t, int col, int row);
const int32_t* gauss_table_int(const gauss_table* t, int col);
const double* gauss_table_dbl(const gauss_table* t, int col);
/* A whole string column at once: NUL-separated values in one buffer (n values, *bytes long in total). */
const char* gauss_table_strcol(const gauss_table* t, int c, int64_t* bytes);
/* Named numeric members of a result List (prep_qcat's z_vec / cor_mat1 / cor_mat2, prep_recessive_impute's
 * zvec / cormat / cormat_add / cormat_dom / cormat_rec): nrow x ncol, COLUMN-major like an R NumericMatrix
 * (vectors have ncol = 1). */
int gauss_table_n_named(const gauss_table* t);
const char* gauss_table_named_name(const gauss_table* t, int k);
const double* gauss_table_named(const gauss_table* t, int k, int* nrow, int* ncol);
/* computeLD's `cormat` (n x n, symmetric); NULL for the other tables */
const double* gauss_table_matrix(const gauss_table* t, int* n);
void gauss_table_free(gauss_table* t);

/* ---- the five entry points (blocking; ctx from gauss_hip_init) -------------------------------- */
int gauss_host_computeLD(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp,
                         const char* const* pop_names, const double* pop_wgts, int n_pop_wgt,
                         const char* input_file, const char* reference_index_file,
                         const char* reference_data_file, const char* reference_pop_desc_file,
                         double af1_cutoff, gauss_table** out);
int gauss_host_dist(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size,
                    const char* study_pop, const char* input_file, const char* reference_index_file,
                    const char* reference_data_file, const char* reference_pop_desc_file,
                    double af1_cutoff, gauss_table** out);
int gauss_host_distmix(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size,
                       const char* const* pop_names, const double* pop_wgts, int n_pop_wgt,
                       const char* input_file, const char* reference_index_file,
                       const char* reference_data_file, const char* reference_pop_desc_file,
                       double af1_cutoff, gauss_table** out);
int gauss_host_jepeg(gauss_ctx* ctx, const char* study_pop, const char* input_file,
                     const char* annotation_file, const char* reference_index_file,
                     const char* reference_data_file, const char* reference_pop_desc_file,
                     double af1_cutoff, gauss_table** out);
int gauss_host_jepegmix(gauss_ctx* ctx, const char* const* pop_names, const double* pop_wgts,
                        int n_pop_wgt, const char* input_file, const char* annotation_file,
                        const char* reference_index_file, const char* reference_data_file,
                        const char* reference_pop_desc_file, double af1_cutoff, gauss_table** out);

/* jepeg() / jepegmix() on several GPUs (BASELINE.json configs[4], SURVEY.md section 8e: "genes: contiguous gene ranges or one
 * gene-batch per GPU").  Genes are independent (jepeg.cpp:114-131 builds and tests one Gene at a time; jepegmix.cpp:119-140;
 * grouping at gauss.cpp:1383-1439), so rank `rank` of `world` runs the host data layer on the same files as every other rank,
 * derives the same plan -- contiguous gene ranges of equal cost, a gene costing its SNP pairs n (n + 1) plus a constant for its
 * k x k tail -- and computes CorG (GPU) and the tails of ITS range only: no communication.  The tables of ranks 0 .. world-1,
 * concatenated in rank order, are gauss_host_jepeg(mix)'s table row for row, bit for bit.  The result carries the named matrix
 * "gene_range" [1 x 3] = first gene, one past the last gene, genes in all.  kind = GAUSS_KIND_JEPEG (study_pop) or
 * GAUSS_KIND_JEPEGMIX (pop_names / pop_wgts); rank 0 of world 1 is the reference's call. */
int gauss_host_jepeg_rank(gauss_ctx* ctx, int kind, const char* study_pop, const char* const* pop_names, const double* pop_wgts,
                          int n_pop_wgt, const char* input_file, const char* annotation_file, const char* reference_index_file,
                          const char* reference_data_file, const char* reference_pop_desc_file, double af1_cutoff,
                          int rank, int world, gauss_table** out);
/* The loop over calls above that: n_calls independent jepeg() / jepegmix() calls -- the reference's user runs one per chromosome
 * file set (jepeg.cpp:28-34 takes one input / annotation / panel per call) -- dealt WHOLE to the ranks, longest first by the size of
 * the annotation file onto the least loaded rank (every rank sees the same files and derives the same deal; no communication).  A
 * call is host-bound (3.4 ms, of which 0.6 ms GPU), so whole calls per rank is the split that scales; the gene split above divides
 * only the GPU batch and the tails.  out[c] = call c's table on its owner, NULL on the other ranks; owner_out[c] (may be NULL) = the
 * rank that ran it.  reference_index_files may be NULL when every data file is a packed panel.  A call that fails leaves out[c] NULL,
 * the others still run, and the function returns -1 with the first failure's message. */
int gauss_host_jepeg_genome(gauss_ctx* ctx, int kind, int n_calls, const char* study_pop, const char* const* pop_names,
                            const double* pop_wgts, int n_pop_wgt, const char* const* input_files, const char* const* annotation_files,
                            const char* const* reference_index_files, const char* const* reference_data_files,
                            const char* reference_pop_desc_file, double af1_cutoff, int rank, int world, gauss_table** out,
                            int32_t* owner_out);
/* The two halves of gauss_host_jepeg_rank for a harness that computes CorG itself (the CPU tests put the oracle there): the plan
 * -- first[r] = rank r's first gene, first[world] = genes in all -- and the gene table of genes [g0, g1) from their CorG blocks
 * (concatenated n_g x n_g, row-major, diagonal 1 + lambda = 1.1).  No GPU needed. */
int gauss_prepared_jepeg_plan(const gauss_prepared* p, int world, int32_t* first);
int gauss_prepared_jepeg_finish(const gauss_prepared* p, int g0, int g1, const double* blocks, gauss_table** out);

/* QCAT / QCATMIX: same feeder as dist / distmix, the window core is
 * run_qcat (qcat.cpp:134-262) / run_qcatmix (qcatmix.cpp:144-297).  af1_cutoff NaN -> 0.05 for qcat
 * (qcat.cpp:53-57), 0.01 for qcatmix (qcatmix.cpp:61-65).  Output columns: rsid chr bp a1 a2
 * af1ref|af1mix z qcat_m qcat_t qcat_chisq qcat_pval type. */
int gauss_host_qcat(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size,
                    const char* study_pop, const char* input_file, const char* reference_index_file,
                    const char* reference_data_file, const char* reference_pop_desc_file,
                    double af1_cutoff, gauss_table** out);
int gauss_host_qcatmix(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size,
                       const char* const* pop_names, const double* pop_wgts, int n_pop_wgt,
                       const char* input_file, const char* reference_index_file,
                       const char* reference_data_file, const char* reference_pop_desc_file,
                       double af1_cutoff, gauss_table** out);

/* Raw LD export.  prep_qcat (prep_qcat.cpp:16-205): snplist = every SNP of the
 * extended window after the AF filter (rsid chr bp a1 a2 af1ref z type); z_vec [M]; cor_mat1 [M x M] pooled
 * LD among the measured SNPs, unit diagonal; cor_mat2 [A x M] LD of ALL prediction-window SNPs (measured and
 * unmeasured) against them.  prep_recessive_impute (prep_qcatmix.cpp:36-316): SNPs are first re-oriented to
 * the minor allele (UpdateSnpToMinorAllele, gauss.cpp:1137-1184); snplist = the prediction-window SNPs
 * (rsid chr bp a1 a2 af1mix z type); zvec [M]; cormat [M x M] weighted LD; cormat_add / cormat_dom /
 * cormat_rec [A x M] with the prediction-window SNPs additive / dominant / recessive coded
 * (gauss.cpp:1196-1250, recoded on the device). */
int gauss_host_prep_qcat(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size,
                         const char* study_pop, const char* input_file, const char* reference_index_file,
                         const char* reference_data_file, const char* reference_pop_desc_file,
                         double af1_cutoff, gauss_table** out);
int gauss_host_prep_recessive_impute(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size,
                                     const char* const* pop_names, const double* pop_wgts, int n_pop_wgt,
                                     const char* input_file, const char* reference_index_file,
                                     const char* reference_data_file, const char* reference_pop_desc_file,
                                     double af1_cutoff, gauss_table** out);

/* prep_zmix5 (zmix.cpp:44-190): every `interval`-th measured SNP, keep those whose
 * across-population allele-frequency variance var(AF)/(mean(1-mean)) exceeds its `percentile` quantile
 * (R stats::quantile type 7, restated), and tabulate for every pair of them z_i*z_j and the Pearson
 * correlation of their genotypes inside each of the panel's populations (on the GPU).  Result: named
 * matrix "data_mat", [n_pairs x (1 + n_pop)] column-major like the reference's NumericMatrix; the table
 * lists the selected SNPs (rsid chr bp a1 a2 z norm_var).  percentile NaN -> 0.99, interval <= 0 -> 1. */
int gauss_host_prep_zmix5(gauss_ctx* ctx, const char* input_file, const char* reference_index_file,
                          const char* reference_data_file, const char* reference_pop_desc_file,
                          double percentile, int interval, gauss_table** out);
/* The other pair selectors of the family (R/RcppExports.R:231-311, zmix.cpp:201-1076); same files, same output layout
 * (named matrix "data_mat": one row per listed SNP pair, z_i * z_j then one correlation column per population), which
 * pairs they list differs:
 *   prep_zmix      every interval-th measured SNP (<= 0: 1), all pairs                       zmix.cpp:940-1076
 *   prep_zmix2     pairs (i, i + offset), i = 0, interval, 2 interval, ... (<= 0: 1000, 3)   zmix.cpp:651-760
 *   prep_zmix3     every interval-th SNP with its next `steps` neighbours (<= 0: 1000, 5)    zmix.cpp:511-650
 *   prep_zmix4     for h = 0 .. interval-1: pairs (i, i + offset), i = h, h + interval, ...; data_mat has a leading
 *                  column h (<= 0: 1000, 3)                                                  zmix.cpp:363-510
 *   prep_zmix5_sup prep_zmix5's SNPs; correlations pooled per super-population, columns in order of first
 *                  appearance in the description file (CalCorSup)                            zmix.cpp:201-361, 1221-1246
 * The table lists the SNPs that occur in some pair (rsid chr bp a1 a2 z [norm_var]); named matrix "pairs" [n_pairs x 2]
 * gives each pair as rows of that table; the table's messages name the correlation columns (populations / groups). */
int gauss_host_prep_zmix(gauss_ctx* ctx, const char* input_file, const char* reference_index_file, const char* reference_data_file,
                         const char* reference_pop_desc_file, int interval, gauss_table** out);
int gauss_host_prep_zmix2(gauss_ctx* ctx, const char* input_file, const char* reference_index_file, const char* reference_data_file,
                          const char* reference_pop_desc_file, int interval, int offset, gauss_table** out);
int gauss_host_prep_zmix3(gauss_ctx* ctx, const char* input_file, const char* reference_index_file, const char* reference_data_file,
                          const char* reference_pop_desc_file, int interval, int steps, gauss_table** out);
int gauss_host_prep_zmix4(gauss_ctx* ctx, const char* input_file, const char* reference_index_file, const char* reference_data_file,
                          const char* reference_pop_desc_file, int interval, int offset, gauss_table** out);
int gauss_host_prep_zmix5_sup(gauss_ctx* ctx, const char* input_file, const char* reference_index_file, const char* reference_data_file,
                              const char* reference_pop_desc_file, double percentile, int interval, gauss_table** out);

/* ---- packed panel -------------------------------------------------
 * Converts the reference's BGZF text panel (index + data + population description) into one mmap-able
 * file: SNP table, per-population allele frequencies and allele counts, and 2-bit genotype rows
 * (gauss_amd/csrc/host/packed_panel.h).  Passing that file as reference_data_file to any entry point
 * above selects the packed feeder (reference_index_file is then ignored): no inflate, no text parsing,
 * windows entered by binary search, each SNP row read once -- results are identical to the text path.
 * Returns the number of SNPs packed or -1. */
int64_t gauss_host_pack_panel(const char* reference_index_file, const char* reference_data_file,
                              const char* reference_pop_desc_file, const char* out_file);
/* The packed-panel cache ("auto-pack on first use").  The packed form of a text panel lives in a cache directory
 * ($GAUSS_PANEL_CACHE, else ".gauss_panel_cache" beside the data file, else /tmp/gauss_panel_cache_<uid>) under a name that
 * carries path, size and mtime of all three panel files.  Returns 0 and the packed panel's path in out_path (the data file
 * itself when it is packed already; *snps_packed_now > 0 when this call made it), 1 when there is no cached panel and
 * create == 0, -1 on error.  Concurrent callers (one rank per GPU) are serialised by a lock file; the panel appears by
 * rename.  Policy of the entry points, env GAUSS_AUTO_PACK: 0 = never use the cache; 1 = pack on first use everywhere;
 * unset = the one-window entry points use a cached panel when one exists, gauss_host_impute_chromosome makes it. */
int gauss_host_panel_cache(const char* reference_index_file, const char* reference_data_file,
                           const char* reference_pop_desc_file, int create, char* out_path, int out_len,
                           int64_t* snps_packed_now);
/* For a prepared window that reads a packed panel: the panel's genotype section (host pointer into the
 * mmap), its size and row stride, so a harness can upload it once with gauss_store_upload and run its
 * windows with on_device = 1.  base = NULL for byte-matrix windows. */
int gauss_prepared_packed_store(const gauss_prepared* p, const uint8_t** base, int64_t* bytes, int64_t* row_bytes);

/* ---- resident panels and whole-chromosome runs (the farm's native side) --------------------------------
 * gauss_host_panel_resident uploads the genotype section of a packed panel to the context's GPU once (pinned
 * double-buffered hipMemcpyAsync; later calls for the same file and context are no-ops) -- 288 GB of HBM hold
 * the whole 33KG panel.  gauss_host_panel_evict frees it (packed_file NULL: every panel of the context). */
int gauss_host_panel_resident(gauss_ctx* ctx, const char* packed_file, int64_t* bytes_uploaded);
int gauss_host_panel_evict(gauss_ctx* ctx, const char* packed_file);
/* Device pointer of row 0 of a resident panel (error if it is not resident on this context), and, for a prepared
 * object that reads a packed panel, the panel rows of its measured / unmeasured SNPs and the byte offset of each
 * selected population block inside a row: what gauss_ld_rows / gauss_gene_ld_batch_rows / a window descriptor take. */
int gauss_host_panel_device_rows(gauss_ctx* ctx, const char* packed_file, const void** out_device_ptr);
int gauss_prepared_store_rows(const gauss_prepared* p, const int32_t** rows_m, const int32_t** rows_u,
                              const int32_t** pop_src_off, int* n_pop_selected);

typedef struct gauss_chrom_stats {
    int32_t n_windows, n_windows_mine, n_skipped, n_failed, n_batches;
    int32_t n_merged_giveups;      /* batches whose merged Gram launch gave up waiting and were run again in the two-launch form
                                      inside their fetch (gauss_hip_counters; 0 in a healthy run)                              */
    int64_t imputed;               /* unmeasured SNPs imputed by this rank                                  */
    int64_t panel_bytes_uploaded;  /* 0 when the panel was already resident                                 */
    double t_total, t_plan, t_panel_upload;
    double t_feeder_wait;          /* main thread waiting for the data layer of the next batch              */
    double t_job_create;           /* planning + queuing the batches (the GPU keeps running meanwhile)      */
    double t_gpu_wait;             /* main thread waiting for a batch's results                             */
    double t_tables;               /* building and concatenating the result tables                          */
    double gpu_span_ms;            /* device time from the first batch's start to the last batch's results  */
    double t_tables_tail;          /* the part of t_tables behind the LAST batch's results (nothing overlaps it) */
} gauss_chrom_stats;

/* dist / distmix / qcat / qcatmix over every window [start_bp + k*window_size, ...] of [start_bp, end_bp] -- the
 * caller-level loop the reference leaves to the R user (one R call = one window, dist.cpp:30-126) -- as one native
 * call per rank.  Windows are sharded over `world` ranks by LPT on their LD flops (the same plan on every rank,
 * no communication); this rank's windows run as a pipeline of n_batches jobs (<= 0: chosen here): host threads
 * prepare batch b+1 and build the tables of batch b-1 while the GPU works on batch b.  reference_data_file is a packed
 * panel (reference_index_file may then be NULL) or the reference's BGZF text panel (gauss.cpp:293-399, 720-785), whose
 * packed form is made on first use and kept in the panel cache (gauss_host_panel_cache); it is made resident on first use.  The result holds the reference's output table of every
 * window of this rank in window order, an extra int column "window", a named matrix "windows"
 * [n_windows x 6: start_bp end_bp owner status measured unmeasured; status 0 done, 1 skipped by the ">10" guards
 * (dist.cpp:145-151), 2 failed, -1 another rank's] and one message per failed window (gauss_table_message):
 * a window that fails never takes the others with it. */
/* The planner's cost of a window with n_measured / n_unmeasured SNPs, per sample: the fp32 matrix flops the Gram kernel issues
 * for it (128-row tiles, 32 / 16 granular edges, B11's tile triangle) plus 8 % on B21's share for the per-entry tails.  What
 * gauss_host_impute_chromosome balances the ranks on; gauss_amd/farm.py:piece_cost is its Python twin. */
double gauss_host_plan_cost(int n_measured, int n_unmeasured);
int gauss_host_impute_chromosome(gauss_ctx* ctx, int kind, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size,
                                 int64_t window_size, const char* study_pop, const char* const* pop_names,
                                 const double* pop_wgts, int n_pop_wgt, const char* input_file, const char* reference_index_file,
                                 const char* reference_data_file, const char* reference_pop_desc_file, double af1_cutoff,
                                 int rank, int world, int n_batches, gauss_table** out, gauss_chrom_stats* stats);
/* The loop over chromosomes above that loop over windows: gauss_host_impute_chromosome for chromosome chr[c] over
 * [start_bp[c], end_bp[c]], c = 0 .. n_chrom-1, the same study / panel files and arguments for all of them (the reference's user
 * passes the same files and another `chr` per call, dist.cpp:30-60), with `depth` calls in flight on the context at any time
 * (<= 0: 2; host threads of the library): while one call's host part runs -- plan, data layer, job tables, result tables: ~2.5 ms
 * that nothing overlaps when a rank's share of a chromosome is a few windows -- the other call's batches keep the GPU busy.
 * out[c] / stats[c] (stats may be NULL) receive chromosome c's table and statistics exactly as gauss_host_impute_chromosome
 * returns them (same bits); a chromosome that fails leaves out[c] = NULL, the others still complete, and the call returns -1
 * with the first failure's message.  The caller frees every table. */
int gauss_host_impute_genome(gauss_ctx* ctx, int kind, int n_chrom, const int32_t* chr, const int64_t* start_bp, const int64_t* end_bp,
                             int64_t wing_size, int64_t window_size, const char* study_pop, const char* const* pop_names,
                             const double* pop_wgts, int n_pop_wgt, const char* input_file, const char* reference_index_file,
                             const char* reference_data_file, const char* reference_pop_desc_file, double af1_cutoff,
                             int rank, int world, int depth, gauss_table** out, gauss_chrom_stats* stats);
/* A test / debugging view (no GPU): ONE window as gauss_host_impute_chromosome builds it on a sorted packed panel -- by a merge of
 * the study's rows and the panel's SNP table, both ordered by position, instead of the reference's per-SNP objects in a std::map
 * (gauss.cpp:121-190, 293-399, 543-693: what gauss_host_prepare restates).  Columns as gauss_prepared_snps (fpos = panel row; a
 * wing's unmeasured SNPs, which nothing reads, are not listed), named matrices "rows_m", "rows_u", "z1" and "counts" =
 * [measured, unmeasured, n_head, n_predm]; the ">10" guard's text, if the window trips it, is message 0. */
int gauss_host_chrom_window_view(int kind, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size, const char* study_pop,
                                 const char* const* pop_names, const double* pop_wgts, int n_pop_wgt, const char* input_file,
                                 const char* packed_file, const char* reference_pop_desc_file, double af1_cutoff, gauss_table** out);
/* The host part of jepeg()/jepegmix() for ONE gene, given the CorG block the GPU produced (diagonal 1 + lambda):
 * Gene::RunJepeg bookkeeping + CalJepegPval from W on (gene.cpp:88-185, 317-550).  corg [n x n], has / wgt [n x 6]
 * row-major (category present, category weight).  top_categ / top_snp are indices (-1 when df = 0).  No GPU needed. */
int gauss_host_jepeg_gene_tail(int n, const double* corg, const double* z, const double* info, const int32_t* has,
                               const double* wgt, double* chisq, int32_t* df, double* jepeg_pval, int32_t* top_categ,
                               double* top_categ_pval, int32_t* top_snp, double* top_snp_pval);
int gauss_table_n_messages(const gauss_table* t);
const char* gauss_table_message(const gauss_table* t, int k);
/* A whole string column as one fixed-width, NUL-padded byte matrix [nrow x *width]. */
const char* gauss_table_strcol_fixed(const gauss_table* t, int c, int* width);

/* Re-block a BGZF text file line by line (reader + writer round trip); returns lines copied or -1. */
int64_t gauss_host_bgzf_copy(const char* in_path, const char* out_path);

/* Host threads used to inflate and split panel lines inside one prepare call (default 4). */
void gauss_host_set_threads(int n);

/* ---- host data layer only (no GPU) ----------------------------------------------------------- */
/* Runs everything up to and including ReadGenotype + the measured/unmeasured partition.
 * kind = GAUSS_KIND_*; arguments that a kind does not take are ignored (pass 0/NULL). */
int gauss_host_prepare(int kind, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size,
                       const char* study_pop, const char* const* pop_names, const double* pop_wgts,
                       int n_pop_wgt, const char* input_file, const char* annotation_file,
                       const char* reference_index_file, const char* reference_data_file,
                       const char* reference_pop_desc_file, double af1_cutoff,
                       gauss_prepared** out);
/* SNP list after the AF filter (snp_vec of the reference), columns:
 * rsid chr bp a1 a2 af1 z info type fpos geneid
 * dist / distmix on a PACKED panel: wing SNPs of the panel that no study SNP shares a position with are not listed
 * (the reference enters them as type 0, filters them, and then neither imputes nor prints them: dist.cpp:91-93,132-140;
 * the text feeder lists them).  The window's result table is the same either way. */
const gauss_table* gauss_prepared_snps(const gauss_prepared* p);
int gauss_prepared_counts(const gauss_prepared* p, int* n_measured, int* n_unmeasured,
                          int* n_samples, int* n_pop, int* n_gene);
/* row indices (into the SNP list) of the measured / unmeasured SNPs, in matrix row order */
const int32_t* gauss_prepared_measured_rows(const gauss_prepared* p);
const int32_t* gauss_prepared_unmeasured_rows(const gauss_prepared* p);
/* genotype matrices exactly as they go to gauss_impute_window / gauss_ld (ASCII digits) */
const uint8_t* gauss_prepared_geno_m(const gauss_prepared* p, int64_t* ld);
const uint8_t* gauss_prepared_geno_u(const gauss_prepared* p, int64_t* ld);
const int32_t* gauss_prepared_pop_off(const gauss_prepared* p);
const double* gauss_prepared_pop_wgt(const gauss_prepared* p);
const double* gauss_prepared_z1(const gauss_prepared* p);
const int32_t* gauss_prepared_gene_off(const gauss_prepared* p);
/* QCAT windows: number of measured SNPs left of the prediction window and inside it (qcat.cpp:147-150) */
int gauss_prepared_qcat_counts(const gauss_prepared* p, int* n_head_measured, int* n_pred_measured);
/* Fill a window descriptor for gauss_job_create from a prepared dist/distmix/qcat/qcatmix window; the outputs
 * point into the prepared object and are written back into its SNP list by gauss_prepared_finish. */
int gauss_prepared_window_desc(gauss_prepared* p, gauss_window_desc* out);
/* After the GPU results are in: build the reference's output DataFrame (dist.cpp:91-124). */
int gauss_prepared_finish(gauss_prepared* p, gauss_table** out);
void gauss_prepared_free(gauss_prepared* p);

#ifdef __cplusplus
}
#endif
#endif /* GAUSS_HOST_H */
