// libgauss_hip.so -- the C ABI of include/gauss_hip.h: jobs, the blocking window call, the LD entry points.
#include "gauss_job.h"

// entry points that need the job's context
#define JOB_ALIVE(job)                                                                                          \
    do {                                                                                                        \
        if (!(job)) return fail(GAUSS_E_INVALID, "job is NULL");                                                \
        if (!(job)->ctx) return fail(GAUSS_E_INVALID, "the job's context has been destroyed (gauss_hip_destroy)"); \
    } while (0)

static WinSpec spec_from_desc(const gauss_window_desc& d)
{
    WinSpec w;
    w.mode = d.mode; w.n_pop = d.n_pop; w.pop_off = d.pop_off; w.pop_wgt = d.pop_wgt;
    w.M = d.n_measured; w.U = d.n_unmeasured; w.geno_m = d.geno_m; w.geno_u = d.geno_u; w.ld = d.ld;
    w.z1 = d.z1; w.lambda = d.lambda; w.eps = d.min_abs_eig; w.diag = 1.0; w.ld_only = 0;
    w.gene_off = nullptr; w.n_gene = 0;
    w.kind = d.kind; w.n_head = d.n_head_measured; w.n_predm = d.n_pred_measured; w.eig_cutoff = d.eig_cutoff;
    w.u_codings = d.u_codings;
    w.geno_fmt = d.geno_format; w.rows_m = d.rows_m; w.rows_u = d.rows_u; w.pop_src_off = d.pop_src_off;
    w.out_b11 = d.out_b11; w.out_b21 = d.out_b21;
    return w;
}
extern "C" {

int gauss_job_create(gauss_ctx* ctx, const gauss_window_desc* wins, int n_win, int on_device, gauss_job** out_job)
{
    if (!ctx || !wins || n_win < 1 || !out_job) return fail(GAUSS_E_INVALID, "bad arguments to gauss_job_create");
    std::vector<WinSpec> specs;
    for (int i = 0; i < n_win; i++) {
        if (wins[i].n_unmeasured < 1 && wins[i].kind != GAUSS_WIN_LD &&
            !(wins[i].kind == GAUSS_WIN_QCAT && wins[i].n_pred_measured > 0))
            return fail(GAUSS_E_INVALID, "window %d has no unmeasured SNPs", i);
        specs.push_back(spec_from_desc(wins[i]));
    }
    gauss_job* job = nullptr;
    int rc = job_build(ctx, specs, on_device, &job);
    if (rc) return rc;
    for (int i = 0; i < n_win; i++) {
        Plan& pl = job->plans[i];
        pl.out_z = wins[i].out_z; pl.out_info = wins[i].out_info; pl.out_status = wins[i].out_status;
        pl.out_b11 = wins[i].out_b11; pl.out_b21 = wins[i].out_b21;
        pl.out_r = wins[i].out_r; pl.out_num_eig = wins[i].out_num_eig;
    }
    *out_job = job;
    return GAUSS_OK;
}

int gauss_job_run(gauss_job* job) { JOB_ALIVE(job); return job_run(job, true); }
int gauss_job_fetch(gauss_job* job) { JOB_ALIVE(job); return job_fetch(job); }
void gauss_job_destroy(gauss_job* job) { job_free(job); }

int gauss_job_span_ms(gauss_job* first, gauss_job* last, double* out_ms)
{
    JOB_ALIVE(first); JOB_ALIVE(last);
    if (!out_ms || !first->ran || !last->ran) return fail(GAUSS_E_INVALID, "gauss_job_span_ms: both jobs must have run");
    HIPCHK(hipEventSynchronize(last->done));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, first->begin, last->done));
    *out_ms = (double)ms;
    return GAUSS_OK;
}

int gauss_job_profile(gauss_job* job, int enable)
{
    JOB_ALIVE(job);
    if (job->prof && !enable) prof_collect(job);
    job->prof = enable != 0;
    if (enable) { for (int k = 0; k < 5; k++) { job->prof_ms[k] = 0; job->prof_n[k] = 0; } }
    return GAUSS_OK;
}

int gauss_job_profile_get(gauss_job* job, int kernel, double* out_ms, int64_t* out_launches)
{
    JOB_ALIVE(job);
    if (kernel < 0 || kernel > 4) return fail(GAUSS_E_INVALID, "bad arguments");
    hipStreamSynchronize(job->ctx->stream);
    prof_collect(job);
    if (out_ms) *out_ms = job->prof_ms[kernel];
    if (out_launches) *out_launches = job->prof_n[kernel];
    return GAUSS_OK;
}

int gauss_job_work(gauss_job* job, double* out_ld_flops, double* out_solve_flops, double* out_bytes, int64_t* out_imputed)
{
    if (!job) return fail(GAUSS_E_INVALID, "job is NULL");
    double ldf = 0, sf = 0, by = 0;
    int64_t imp = 0;
    for (const Plan& pl : job->plans) {
        const double M = pl.p.M, U = pl.p.U, N = pl.p.N;
        ldf += N * M * (M + 1) + 2.0 * N * U * M;            // symmetric half of B11 + B21
        sf += M * M * M / 3.0 + 2.0 * U * M * M + 4.0 * U * M;
        by += (M + U) * N + (M * M + U * M) * 8.0;
        imp += pl.p.U;
    }
    if (out_ld_flops) *out_ld_flops = ldf;
    if (out_solve_flops) *out_solve_flops = sf;
    if (out_bytes) *out_bytes = by;
    if (out_imputed) *out_imputed = imp;
    return GAUSS_OK;
}

int gauss_job_counters(gauss_job* job, int64_t* out4)
{
    if (!job || !out4) return fail(GAUSS_E_INVALID, "bad arguments to gauss_job_counters");
    out4[0] = job->n_merged; out4[1] = job->n_demoted; out4[2] = job->n_giveups; out4[3] = job->n_rerun_failed;
    return GAUSS_OK;
}

int gauss_job_stats(gauss_job* job, double* out4)
{
    if (!job || !out4) return fail(GAUSS_E_INVALID, "bad arguments");
    double flops = 0, slab = 0;
    const bool shm = job->gplan != nullptr;
    const bool edge16 = true;                  // (k_gram.hip: the 16-column edge routine is compiled in)
    auto add = [&](const Plan& pl, bool skip_b11) {
        const Prob& p = pl.p;
        const int mt = p.Mp / TILE;
        auto rows = [&](int t) {
            if (!pl.tile_live.empty()) return pl.tile_live[(size_t)t];
            int left = (t < mt) ? p.M - t * TILE : p.U - (t - mt) * TILE; return left > TILE ? TILE : left;
        };
        auto halves = [](int r, int w) { int n = (r - w * 64 + 31) / 32; return n < 0 ? 0 : (n > 2 ? 2 : n); };
        // samples per row the kernel multiplies: every zero-padded block (population, or 2-bit source block) rounded up to
        // the units the K loop can skip -- 8 samples on the f32 path (Item::chunk_live), 32 on the int8 path (whole groups);
        // the 16-column edge routine takes every chunk whole
        double k_main = 0;
        {
            const int gran = job->gram_i8 ? 32 : 8;
            const std::vector<int>& blk = (p.geno_fmt == GAUSS_GENO_2BIT && !pl.run_pk_off.empty()) ? pl.run_pk_off : pl.pop_pk_off;
            const bool runs = &blk == &pl.run_pk_off;
            for (size_t q = 0; q + 1 < blk.size(); q++) {
                // live samples of the block: its real size where known (populations), else its padded size
                int live = blk[q + 1] - blk[q];
                if (!runs && q + 1 < pl.pop_raw_off.size()) live = pl.pop_raw_off[q + 1] - pl.pop_raw_off[q];
                else if (runs && pl.run_len_known(q)) live = pl.run_len[q];
                k_main += (double)((live + gran - 1) / gran * gran);
            }
        }
        int pairs = 0;
        for (int pr = 0; pr < p.npair; pr++) {
            const int ti = pl.pair_ti[pr], tj = pl.pair_tj[pr];
            if (skip_b11 && ti < mt) continue;               // multiplied once, on the job-wide tiles
            pairs++;
            double tiles32 = 0, tiles_edge = 0;
            for (int wr = 0; wr < 2; wr++)
                for (int wc = 0; wc < 2; wc++) {
                    if (ti == tj && wr == 1 && wc == 0) continue;
                    const int na = halves(rows(ti), wr);
                    // f32 path: a wave whose last live 32-column half holds at most 16 live columns multiplies 16-column groups
                    // (k_gram.hip, chunk_mfma_edge): 1 or 3 of them
                    int nb16 = (rows(tj) - wc * 64 + 15) / 16;
                    nb16 = nb16 < 0 ? 0 : (nb16 > 4 ? 4 : nb16);
                    if (edge16 && !job->gram_i8 && na > 0 && (nb16 & 1)) { tiles_edge += na * nb16 * 0.5; continue; }
                    double t32 = na * halves(rows(tj), wc);
                    if (ti == tj && wr == wc && t32 == 4) t32 = 3;      // mirrored 32 x 32 sub-block of a diagonal quadrant
                    tiles32 += t32;
                }
            flops += 32.0 * 32.0 * 2.0 * (tiles32 * k_main + tiles_edge * p.Kp);
        }
        slab += (double)pairs * p.nseg * TILE * TILE * (p.slab16 ? sizeof(uint16_t) : sizeof(float));
    };
    for (const Plan& pl : job->plans) add(pl, shm);
    if (shm) add(*job->gplan, false);
    out4[0] = job->n_items; out4[1] = flops; out4[2] = slab; out4[3] = (double)job->ws_bytes;
    return GAUSS_OK;
}

int gauss_impute_window(gauss_ctx* ctx, const gauss_window_desc* win)
{
    if (!ctx || !win) return fail(GAUSS_E_INVALID, "bad arguments to gauss_impute_window");
    // Streamed form (default): upload and compute overlap (job_run_streamed).  It covers the windows the drivers make --
    // contiguous host matrices, additive coding, something to solve; the clamp path re-reads the job's buffers and works
    // on either form.  GAUSS_STREAM_WINDOW=0 (or GAUSS_FUSED_SOLVE=0): upload everything, then run.
    const bool fused = env_int("GAUSS_FUSED_SOLVE", 1) != 0;           // read per run: the tests drive both forms
    bool streamed = env_int("GAUSS_STREAM_WINDOW", 1) != 0 && fused && !win->rows_m && !win->rows_u && win->n_unmeasured >= 1 &&
                    win->n_measured >= 1 && win->kind != GAUSS_WIN_LD && (win->u_codings & ~GAUSS_CODE_ADDITIVE) == 0 &&
                    win->geno_m && win->geno_u && win->pop_off && win->n_pop >= 1 && win->n_pop <= 64;
    // bytes of a source row (what plan_problem will find; anything odd is left to the unstreamed path and its messages)
    size_t row_bytes = 0;
    if (streamed) {
        const int N = win->pop_off[win->n_pop];
        if (win->geno_format == GAUSS_GENO_U8) {
            streamed = N >= 1 && win->ld >= N;
            row_bytes = (size_t)std::max(N, 0);
        } else if (win->geno_format == GAUSS_GENO_2BIT && win->ld % 16 == 0) {
            long long end = 0;
            for (int q = 0; q < win->n_pop && streamed; q++) {
                const long long blk = (long long)rup((size_t)std::max(win->pop_off[q + 1] - win->pop_off[q], 0), 64) / 4;
                const long long off = win->pop_src_off ? win->pop_src_off[q] : end;
                if (off < 0 || off % 16 || off + blk > win->ld) streamed = false;
                if (!win->pop_src_off) end = off + blk;
                row_bytes = std::max(row_bytes, (size_t)(off + blk));
            }
        } else streamed = false;
    }
    gauss_job* job = nullptr;
    int rc;
    const bool trace = trace_on("stream");
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (trace) fprintf(stderr, "[stream] %s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    if (streamed) {
        std::lock_guard<std::mutex> one(ctx->stream_mu);
        HIPCHK(hipSetDevice(ctx->device));
        StreamSetup su;
        rc = stream_start_copies(ctx, *win, row_bytes, su);
        if (rc) return rc;
        // On every path out of here the worker has finished its task (`su` lives on this stack) AND the copies it queued
        // have left the caller's matrices: from pinned memory they are true asynchronous DMAs, and the caller may free
        // the matrices the moment this call returns.  A successful fetch has waited for them already (`landed`).
        struct Waiter {
            gauss_ctx* c; bool landed = false;
            ~Waiter() { c->worker->wait(); if (!landed) (void)hipStreamSynchronize(c->copy); }
        } waiter{ctx};
        lap("copies started");
        std::vector<WinSpec> specs{spec_from_desc(*win)};
        rc = job_build(ctx, specs, 0, &job, &su);
        lap("job built");
        if (rc) return rc;
        Plan& pl = job->plans[0];
        pl.out_z = win->out_z; pl.out_info = win->out_info; pl.out_status = win->out_status;
        pl.out_b11 = win->out_b11; pl.out_b21 = win->out_b21; pl.out_r = win->out_r; pl.out_num_eig = win->out_num_eig;
        rc = job_run_streamed(job, su);
        lap("run queued");
        if (!rc) rc = job_fetch(job);
        lap("fetched");
        if (!rc) waiter.landed = true;             // the results were computed from every chunk: all copies are complete
        else {
            // nothing may still be writing into the landing buffer or reading it when the next call reuses it
            ctx->worker->wait();
            for (hipStream_t q : {ctx->copy, ctx->aux, ctx->chain, ctx->stream}) (void)hipStreamSynchronize(q);
        }
        job_free(job);
        lap("freed");
        return rc;
    }
    rc = gauss_job_create(ctx, win, 1, 0, &job);
    if (rc) return rc;
    rc = job_run(job, true);
    if (!rc) rc = job_fetch(job);
    job_free(job);
    return rc;
}

struct RowSource {                 // where the rows of an LD-only call come from (default: a contiguous host byte matrix)
    int geno_fmt = GAUSS_GENO_U8;
    const int32_t* rows = nullptr;
    const int32_t* pop_src_off = nullptr;
    int on_device = 0;
};

static int ld_common(gauss_ctx* ctx, int mode, const uint8_t* geno, int n_snp, int64_t ld,
                     const int32_t* pop_off, const double* pop_wgt, int n_pop, double diag,
                     const int32_t* gene_off, int n_gene, double* out, int64_t* out_counts, int n_samples,
                     const RowSource& src = RowSource())
{
    if (!ctx || !geno || (!out && !out_counts)) return fail(GAUSS_E_INVALID, "bad arguments");
    WinSpec w;
    int32_t off1[2] = {0, n_samples};
    w.mode = mode; w.n_pop = out_counts ? 1 : n_pop; w.pop_off = out_counts ? off1 : pop_off; w.pop_wgt = pop_wgt;
    w.M = n_snp; w.U = 0; w.geno_m = geno; w.geno_u = nullptr; w.ld = ld; w.z1 = nullptr;
    w.lambda = 0; w.eps = 0; w.diag = diag; w.ld_only = 1; w.gene_off = gene_off; w.n_gene = n_gene;
    w.geno_fmt = src.geno_fmt; w.rows_m = src.rows; w.pop_src_off = src.pop_src_off;
    gauss_job* job = nullptr;
    std::vector<WinSpec> specs{w};
    int rc = job_build(ctx, specs, src.on_device, &job);
    if (rc) return rc;
    std::unique_ptr<gauss_job, void (*)(gauss_job*)> guard(job, job_free);
    if (!out_counts) job->plans[0].out_ld_user = out;
    rc = job_run(job, false);
    if (rc) return rc;
    if (out_counts) {
        DevBuf d_cnt;
        const size_t bytes = sizeof(long long) * (size_t)n_snp * n_snp;
        if (d_cnt.alloc(ctx, bytes) != hipSuccess) return fail(GAUSS_E_NOMEM, "hipMalloc(%zu bytes of counts) failed", bytes);
        launch_counts(job->d_probs, 0, job->plans[0].p.npair, d_cnt.as<long long>(), ctx->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpy(out_counts, d_cnt.p, bytes, hipMemcpyDeviceToHost));
    }
    return job_fetch(job);
}

int gauss_ld(gauss_ctx* ctx, int mode, const uint8_t* geno, int n_snp, int64_t ld, const int32_t* pop_off,
             const double* pop_wgt, int n_pop, double diag, double* out_cor)
{
    return ld_common(ctx, mode, geno, n_snp, ld, pop_off, pop_wgt, n_pop, diag, nullptr, 0, out_cor, nullptr, 0);
}

int gauss_gene_ld_batch(gauss_ctx* ctx, int mode, const uint8_t* geno, int n_snp, int64_t ld,
                        const int32_t* pop_off, const double* pop_wgt, int n_pop,
                        const int32_t* gene_off, int n_gene, double diag, double* out_blocks)
{
    if (!gene_off || n_gene < 1) return fail(GAUSS_E_INVALID, "gene_off is NULL or n_gene < 1");
    return ld_common(ctx, mode, geno, n_snp, ld, pop_off, pop_wgt, n_pop, diag, gene_off, n_gene, out_blocks, nullptr, 0);
}

int gauss_ld_rows(gauss_ctx* ctx, int mode, const uint8_t* store, int64_t ld, int geno_format, const int32_t* rows, int n_snp,
                  const int32_t* pop_off, const int32_t* pop_src_off, const double* pop_wgt, int n_pop, double diag,
                  int on_device, double* out_cor)
{
    RowSource src;
    src.geno_fmt = geno_format; src.rows = rows; src.pop_src_off = pop_src_off; src.on_device = on_device;
    return ld_common(ctx, mode, store, n_snp, ld, pop_off, pop_wgt, n_pop, diag, nullptr, 0, out_cor, nullptr, 0, src);
}

int gauss_gene_ld_batch_rows(gauss_ctx* ctx, int mode, const uint8_t* store, int64_t ld, int geno_format, const int32_t* rows,
                             int n_snp, const int32_t* pop_off, const int32_t* pop_src_off, const double* pop_wgt, int n_pop,
                             const int32_t* gene_off, int n_gene, double diag, int on_device, double* out_blocks)
{
    if (!gene_off || n_gene < 1) return fail(GAUSS_E_INVALID, "gene_off is NULL or n_gene < 1");
    RowSource src;
    src.geno_fmt = geno_format; src.rows = rows; src.pop_src_off = pop_src_off; src.on_device = on_device;
    return ld_common(ctx, mode, store, n_snp, ld, pop_off, pop_wgt, n_pop, diag, gene_off, n_gene, out_blocks, nullptr, 0, src);
}

int gauss_ld_per_pop(gauss_ctx* ctx, const uint8_t* geno, int n_snp, int64_t ld, const int32_t* pop_off, int n_pop,
                     double* out)
{
    if (!ctx || !geno || !pop_off || !out) return fail(GAUSS_E_INVALID, "bad arguments to gauss_ld_per_pop");
    if (n_snp < 2) return fail(GAUSS_E_INVALID, "need at least two SNPs");
    // the weighted layout keeps one exact Gram partial per population: all that is needed here
    std::vector<double> ones((size_t)std::max(n_pop, 1), 1.0);
    WinSpec w;
    w.mode = GAUSS_MODE_WEIGHTED; w.n_pop = n_pop; w.pop_off = pop_off; w.pop_wgt = ones.data();
    w.M = n_snp; w.U = 0; w.geno_m = geno; w.geno_u = nullptr; w.ld = ld; w.z1 = nullptr;
    w.lambda = 0; w.eps = 0; w.diag = 1.0; w.ld_only = 1; w.gene_off = nullptr; w.n_gene = 0;
    gauss_job* job = nullptr;
    std::vector<WinSpec> specs{w};
    int rc = job_build(ctx, specs, 0, &job);
    if (rc) return rc;
    std::unique_ptr<gauss_job, void (*)(gauss_job*)> guard(job, job_free);
    rc = job_run(job, false);
    if (rc) return rc;
    const size_t npairs = (size_t)n_snp * (n_snp - 1) / 2;
    const size_t bytes = sizeof(double) * npairs * (size_t)n_pop;
    DevBuf d_out;
    if (d_out.alloc(ctx, bytes) != hipSuccess) return fail(GAUSS_E_NOMEM, "hipMalloc(%zu bytes per-population LD) failed", bytes);
    launch_pop_cor(job->d_probs, 0, job->plans[0].p.npair, d_out.as<double>(), ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(out, d_out.p, bytes, hipMemcpyDeviceToHost));
    return GAUSS_OK;
}

int gauss_ld_per_pop_pairs(gauss_ctx* ctx, const uint8_t* geno, int n_snp, int64_t ld, const int32_t* pop_off, int n_pop,
                           const int32_t* pop_group, int n_group, const int32_t* pair_i, const int32_t* pair_j, int64_t n_pairs,
                           double* out)
{
    if (!ctx || !geno || !pop_off || !out || !pair_i || !pair_j) return fail(GAUSS_E_INVALID, "bad arguments to gauss_ld_per_pop_pairs");
    if (n_snp < 2 || n_pairs < 1) return fail(GAUSS_E_INVALID, "need at least two SNPs and one pair");
    if (!pop_group) n_group = n_pop;
    if (n_group < 1) return fail(GAUSS_E_INVALID, "n_group < 1");
    if (pop_group)
        for (int p = 0; p < n_pop; p++)
            if (pop_group[p] < 0 || pop_group[p] >= n_group) return fail(GAUSS_E_INVALID, "pop_group[%d] = %d is outside 0..%d", p, pop_group[p], n_group - 1);
    std::vector<double> ones((size_t)std::max(n_pop, 1), 1.0);
    WinSpec w;
    w.mode = GAUSS_MODE_WEIGHTED; w.n_pop = n_pop; w.pop_off = pop_off; w.pop_wgt = ones.data();
    w.M = n_snp; w.U = 0; w.geno_m = geno; w.geno_u = nullptr; w.ld = ld; w.z1 = nullptr;
    w.lambda = 0; w.eps = 0; w.diag = 1.0; w.ld_only = 1; w.gene_off = nullptr; w.n_gene = 0;
    w.pair_i = pair_i; w.pair_j = pair_j; w.n_pairs = n_pairs;
    gauss_job* job = nullptr;
    std::vector<WinSpec> specs{w};
    int rc = job_build(ctx, specs, 0, &job);
    if (rc) return rc;
    std::unique_ptr<gauss_job, void (*)(gauss_job*)> guard(job, job_free);
    // pack + Gram only: the LD epilogue has nothing to write for a pair list
    hipStream_t st = ctx->stream;
    launch_pack_stats(job->d_probs, job->d_rowmap, job->n_rows, st);
    launch_gram(job->d_items, job->n_items, job->gram_i8, st);
    HIPCHK(hipGetLastError());
    std::vector<int2> pairs((size_t)n_pairs);
    for (int64_t k = 0; k < n_pairs; k++) pairs[(size_t)k] = make_int2(pair_i[k], pair_j[k]);
    DevBuf d_pairs, d_grp, d_out;
    const size_t out_bytes = sizeof(double) * (size_t)n_pairs * (size_t)n_group;
    if (d_pairs.alloc(ctx, sizeof(int2) * pairs.size()) != hipSuccess || d_grp.alloc(ctx, sizeof(int) * (size_t)std::max(n_pop, 1)) != hipSuccess ||
        d_out.alloc(ctx, out_bytes) != hipSuccess)
        return fail(GAUSS_E_NOMEM, "hipMalloc(%zu bytes of pair correlations) failed", out_bytes);
    HIPCHK(hipMemcpyAsync(d_pairs.p, pairs.data(), sizeof(int2) * pairs.size(), hipMemcpyHostToDevice, st));
    if (pop_group) HIPCHK(hipMemcpyAsync(d_grp.p, pop_group, sizeof(int) * (size_t)n_pop, hipMemcpyHostToDevice, st));
    launch_pair_cor(job->d_probs, 0, d_pairs.as<int2>(), n_pairs, pop_group ? d_grp.as<int>() : nullptr, n_group, d_out.as<double>(), st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(out, d_out.p, out_bytes, hipMemcpyDeviceToHost));
    return GAUSS_OK;
}

int gauss_gram_counts(gauss_ctx* ctx, const uint8_t* geno, int n_snp, int n_samples, int64_t ld, int64_t* out_counts)
{
    if (!out_counts) return fail(GAUSS_E_INVALID, "out_counts is NULL");
    return ld_common(ctx, GAUSS_MODE_POOLED, geno, n_snp, ld, nullptr, nullptr, 1, 1.0, nullptr, 0, nullptr,
                     out_counts, n_samples);
}

int gauss_pack2bit_device(gauss_ctx* ctx, const uint8_t* d_in, int64_t ld_in, uint8_t* d_out, int64_t ld_out,
                          int n_snp, const int32_t* pop_off, int n_pop)
{
    if (!ctx || !d_in || !d_out || !pop_off || n_snp < 1 || n_pop < 1) return fail(GAUSS_E_INVALID, "bad arguments");
    std::vector<int> blk(n_pop + 1, 0);
    for (int q = 0; q < n_pop; q++) blk[q + 1] = blk[q] + (int)rup((size_t)(pop_off[q + 1] - pop_off[q]), 64) / 4;
    if (ld_out % 16 || ld_out < blk[n_pop]) return fail(GAUSS_E_INVALID, "ld_out must be a multiple of 16 and >= %d", blk[n_pop]);
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf tab;
    HIPCHK(tab.alloc(sizeof(int) * 2 * (n_pop + 1)));
    int* d_tab = tab.as<int>();
    HIPCHK(hipMemcpy(d_tab, pop_off, sizeof(int) * (n_pop + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_tab + n_pop + 1, blk.data(), sizeof(int) * (n_pop + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemsetAsync(d_out, 0, (size_t)n_snp * ld_out, ctx->stream));
    launch_pack2bit(d_in, ld_in, d_out, ld_out, n_snp, d_tab, d_tab + n_pop + 1, n_pop, ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GAUSS_OK;
}

int gauss_synth_device(gauss_ctx* ctx, uint8_t* d_out, int n_snp, int64_t ld, const int32_t* pop_off, int n_pop,
                       const float* thr, const float* rho, uint64_t seed)
{
    if (!ctx || !d_out || !pop_off || !thr || !rho || n_snp < 1 || n_pop < 1) return fail(GAUSS_E_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    const int N = pop_off[n_pop];
    DevBuf b_off, b_thr, b_rho;
    HIPCHK(b_off.alloc(sizeof(int) * (n_pop + 1)));
    HIPCHK(b_thr.alloc(sizeof(float) * (size_t)n_snp * n_pop));
    HIPCHK(b_rho.alloc(sizeof(float) * n_snp));
    HIPCHK(hipMemcpy(b_off.p, pop_off, sizeof(int) * (n_pop + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b_thr.p, thr, sizeof(float) * (size_t)n_snp * n_pop, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b_rho.p, rho, sizeof(float) * n_snp, hipMemcpyHostToDevice));
    launch_synth(d_out, n_snp, ld, b_off.as<int>(), n_pop, N, b_thr.as<float>(), b_rho.as<float>(), seed, ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GAUSS_OK;
}

}  // extern "C"
