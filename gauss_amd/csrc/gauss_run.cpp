// libgauss_hip.so -- a run of a job: which kernel goes on which queue, how the results come back, the rare repair paths.
#include "gauss_job.h"

// ------------------------------------------------------------------------------------------
// profiling helpers
// ------------------------------------------------------------------------------------------
static void prof_begin(gauss_job* job, int kernel, hipStream_t st, int launches = 1)
{
    if (!job->prof) return;
    ProfSlot s;
    s.kernel = kernel;
    s.launches = launches;
    s.run = job->prof_run;
    hipEventCreate(&s.a);
    hipEventCreate(&s.b);
    hipEventRecord(s.a, st);
    job->slots.push_back(s);
}
static void prof_end(gauss_job* job, hipStream_t st)
{
    if (!job->prof) return;
    hipEventRecord(job->slots.back().b, st);
}
// Collects the stage timers of the runs before `run_end` (default: all).  gauss_job_fetch passes the run it has just
// fetched: with two runs in flight the later run's events are still pending, and waiting for them here would make
// the fetch of run k block until run k + 1 has finished -- the host's share of a step would no longer overlap GPU work.
void prof_collect(gauss_job* job, unsigned run_end)
{
    std::vector<ProfSlot> keep;
    for (ProfSlot& s : job->slots) {
        if (run_end != ~0u && (int)(s.run - run_end) >= 0) { keep.push_back(s); continue; }
        hipEventSynchronize(s.b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, s.a, s.b);
        job->prof_ms[s.kernel] += ms;
        job->prof_n[s.kernel] += s.launches;
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
    }
    job->slots.swap(keep);
}


// ------------------------------------------------------------------------------------------
// the result mirrors travel with the run, so that gauss_job_fetch waits for THIS job only (an event), not for
// whatever else has been queued on the stream since (the next job of a pipeline)
// Export chunks [c0, c1) into the pinned mirror by kernel (its stores cross PCIe at ~56 GB/s), an event behind each.
// What was tried to hide the link time under the Gram kernel of a 32-window computeLD() batch, and measured (4.85-5.0 ms a step as it
// is: kernels 3.3, link 1.3, last host copy + status 0.3): the early half of the windows as a Gram launch of its own with its
// exports on the chain queue beside the second launch -- 4.75-4.8 ms (the export kernel's workgroups cost that launch 0.3-0.5 ms,
// about what they hide; with 64 workgroups instead of 1 000 the launch keeps its speed and the exports starve at 17 GB/s);
// one launch with the early items counted off (the imputation runs' form) -- the epilogue kernel's 1 024-thread workgroups do not
// fit beside the Gram kernel's and ran when the launch was over; hipMemcpyAsync from a compacted device image -- the runtime
// copies device -> pinned host with a blit KERNEL here (__amd_rocclr_copyBuffer), same contention.  None kept.
static int queue_exports(gauss_job* job, size_t c0, size_t c1, hipStream_t st)
{
    for (size_t c = c0; c < c1 && c < job->exp_chunks.size(); c++) {
        const int x0 = job->exp_chunks[c].first, x1 = job->exp_chunks[c].second;
        long long mx = 0;
        for (int x = x0; x < x1; x++) mx = std::max(mx, (long long)job->exports[(size_t)x].rows * job->exports[(size_t)x].width);
        launch_export_rows(job->d_exports + x0, x1 - x0, mx, st);
        HIPCHK(hipEventRecord(job->exp_ev[c], st));
    }
    return GAUSS_OK;
}

static int job_queue_results(gauss_job* job, int par, hipStream_t st)
{
    HIPCHK(hipGetLastError());
    const bool job_trace = trace_on("job");
    const auto t0 = std::chrono::steady_clock::now();
    // By kernel into the pinned mirrors, not by hipMemcpyAsync: beside a background upload (a chromosome's first call) one run in
    // eight made the caller wait 10-18 ms for a DMA engine here, and every run of a 36-window job 5 ms (round 4).
    // Both mirrors have room for the 16-byte word the copy rounds up to (pin_res in job_build; the status block is 16 (n + 1) bytes);
    // the event below is a system-scope release, so the host reads what the kernel wrote.
    // matrix exports first, chunk by chunk: gauss_job_fetch starts copying chunk c out while the later chunks still cross the link
    { const int rc = queue_exports(job, 0, job->exp_chunks.size(), st); if (rc) return rc; }
    if (job->n_results) launch_h2d_copy(job->h_res2[par], job->d_results, rup(sizeof(double) * job->n_results, 16), st);
    launch_h2d_copy(job->h_st2[par], job->d_status, sizeof(int) * (4 * job->n + 4), st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(job->done2[par], st));
    if (job_trace) fprintf(stderr, "[job] run: result copies queued in %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    job->done = job->done2[par];
    return GAUSS_OK;
}

// One pass of the job on the context's queues, with the cross-queue events and the result mirrors of parity `par`.
// allow_merged = false: the two-launch form whatever the job was built for (the re-run after a give-up, job_fetch).
static int job_queue_run(gauss_job* job, bool solve, int par, bool allow_merged)
{
    gauss_ctx* ctx = job->ctx;
    hipStream_t st = ctx->stream;
    const gauss_job::RunEvents& ev = job->rev[par];      // this run's cross-queue events (the parity's own set)
    job->queue_touched = true;
    std::lock_guard<std::mutex> run_lock(ctx->run_mu);         // every queue sees this context's runs in the same order (gauss_job.h)
    // ONE Gram launch whose B11 items count themselves off for a spinning kernel at the head of the chain queue is only sound
    // while that kernel cannot sit in front of work it waits for, i.e. while every priority stream of the library on this
    // device owns its hardware queue (gauss_ctx.cpp).  The decision and the queuing are one step: a context whose streams
    // would start sharing queues waits (exclusively) until no run is being queued and the spinning kernels have drained.
    std::shared_lock<std::shared_mutex> qlock(queue_registry_mutex());
    const bool merged = job->merged && allow_merged && solve && (job->force_merged || (ctx->queues_probed_distinct && queues_exclusive(ctx->device)));
    if (job->merged && solve) { (merged ? ctx->n_runs_merged : ctx->n_runs_demoted)++; (merged ? job->n_merged : job->n_demoted)++; }
    const auto t_run0 = std::chrono::steady_clock::now();
    HIPCHK(hipEventRecord(job->begin, st));
    HIPCHK(hipMemsetAsync(job->d_status, 0, sizeof(int) * (4 * job->n + 4), st));
    if (trace_on("job")) fprintf(stderr, "[job] run (%s): begin mark + status zeroing queued in %.2f ms\n", merged ? "merged" : (job->chain_aside ? "two launches" : "one queue"),
                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run0).count());
    // fused tail: the factorisation chain needs B11 only and the closing product is the first reader of B21, so B21's
    // tiles of the epilogue (85 % of them) go to the side stream and run beside the chain
    const bool fused = env_int("GAUSS_FUSED_SOLVE", 1) != 0;           // read per run: the tests drive both forms
    hipStream_t side = (solve && fused && job->n_panels > 0 && job->n_tiles > job->n_tiles_b11) ? ctx->side : nullptr;
    prof_begin(job, 1, st);
    launch_pack_stats(job->d_probs, job->d_rowmap, job->n_rows, st);
    // the certificate needs the row tables only and is read by B11's epilogue tiles and the chain: in a merged launch it
    // moves to the head of the chain queue, beside the Gram kernel's start (23 us off the main queue's critical path)
    const bool cert_on_chain = solve && job->chain_aside && merged;
    if (solve && job->n_panels > 0 && !cert_on_chain) launch_shift_cert(job->d_probs, job->n, st);
    prof_end(job, st);
    if (solve && job->chain_aside) {
        // Chain beside the Gram kernel.
        //   main:   Gram(B11's items, then B21's: one launch or two) -> B21's epilogue tiles -> closing product;
        //   chain:  from the moment B11's items have finished: B11's epilogue tiles, then the whole
        //           factorisation with the riding rows of the inverse, all in small-footprint form (k_pack_epilogue.hip
        //           epilogue_b11_lite_kernel, k_solve_lite.hip): their workgroups fit into what the Gram kernel's four workgroups
        //           per CU leave free, so the ~19 dependent block steps run UNDER the Gram launch instead of behind it.
        // Same arithmetic, same bits as the path below.
        hipStream_t ch = ctx->chain;
        if (merged) {
            // ONE launch: B11's items first (they count themselves off in d_b11_done), B21's items behind them in the same grid.
            // The chain queue joins the main queue right BEFORE the launch (operands, row tables are complete)
            // and then waits for the count of this run: the counter only grows, run r is complete at (r + 1) x n_items_b11.
            HIPCHK(hipEventRecord(ev.gram, st));
            prof_begin(job, 0, st, 1);
            launch_gram(job->d_items, job->n_items, job->gram_i8, st, job->d_b11_done);
            job->merged_runs++;                            // counted per LAUNCH, not per completed call: a later error must not shift the target
            prof_end(job, st);
            HIPCHK(hipStreamWaitEvent(ch, ev.gram, 0));
            if (cert_on_chain) launch_shift_cert(job->d_probs, job->n, ch);
            launch_wait_count(job->d_b11_done, job->merged_runs * (unsigned long long)job->n_items_b11, job->d_status + 4 * job->n, 1, ch, job->wait_bound_us);
        } else {
            // two launches joined by an event: B11's items, then B21's (no kernel waits for another queue's progress)
            prof_begin(job, 0, st, 2);
            launch_gram(job->d_items, job->n_items_b11, job->gram_i8, st);
            HIPCHK(hipEventRecord(ev.gram, st));
            launch_gram(job->d_items + job->n_items_b11, job->n_items - job->n_items_b11, job->gram_i8, st);
            prof_end(job, st);
            HIPCHK(hipStreamWaitEvent(ch, ev.gram, 0));
        }
        prof_begin(job, 2, ch);
        launch_epilogue_b11_lite(job->d_probs, job->d_tilemap, job->n_tiles_b11, job->gram_i8, ch);
        prof_end(job, ch);
        for (int i = 0; i < job->n; i++) {
            Plan& pl = job->plans[i];
            if (pl.out_b11 && pl.p.npanel > 0)
                HIPCHK(hipMemcpyAsync(pl.d_b11_copy, pl.p.A, sizeof(double) * pl.p.Mld * pl.p.Mld, hipMemcpyDeviceToDevice, ch));
        }
        prof_begin(job, 3, ch);
        for (int s = 0; s < job->max_nblk; s++)
            launch_factor_step_lite(job->d_probs, job->n, s, job->max_nblk, job->max_npanel, job->solve_split, ch);
        launch_solve_last_lite(job->d_probs, job->d_panelmap, job->n_panels, job->max_nblk, job->solve_split, ch);
        prof_end(job, ch);
        HIPCHK(hipEventRecord(ev.side, ch));
        // (B21's tiles read neither B11 nor the certificate -- epilogue_tile looks at status[3] for B11's tiles only -- so they
        // need not wait for the chain queue; the closing product does)
        if (merged && job->n_tiles_b21_early > 0) {
            // Early epilogue: the tiles of the windows whose B21 items are done before the launch's last round go to the LOW-priority
            // queue behind a wait for their count.  The hardware hands a lower-priority queue's workgroups out when the Gram grid
            // has none left to dispatch: they run in the slots the launch's last round leaves idle (measured: 0.16 ms of the
            // 36-window step, 0.09 ms of an 8-rank share's).  The late windows' tiles follow the launch on the main queue.
            hipStream_t lo = ctx->side;
            HIPCHK(hipStreamWaitEvent(lo, ev.gram, 0));
            launch_wait_count(job->d_b11_done + 8, job->merged_runs * (unsigned long long)job->n_items_b21_early, job->d_status + 4 * job->n, 1, lo, job->wait_bound_us);
            launch_epilogue(job->d_probs, job->d_tilemap + job->n_tiles_b11, job->n_tiles_b21_early, job->max_pop, job->gram_i8, lo);
            HIPCHK(hipEventRecord(ev.epi, lo));
            prof_begin(job, 2, st);
            launch_epilogue(job->d_probs, job->d_tilemap + job->n_tiles_b11 + job->n_tiles_b21_early,
                            job->n_tiles - job->n_tiles_b11 - job->n_tiles_b21_early, job->max_pop, job->gram_i8, st);
            prof_end(job, st);
            HIPCHK(hipStreamWaitEvent(st, ev.epi, 0));
        } else {
            prof_begin(job, 2, st);
            launch_epilogue(job->d_probs, job->d_tilemap + job->n_tiles_b11, job->n_tiles - job->n_tiles_b11, job->max_pop, job->gram_i8, st);
            prof_end(job, st);
        }
        HIPCHK(hipStreamWaitEvent(st, ev.side, 0));
        prof_begin(job, 4, st);
        launch_impute_gemm(job->d_probs, job->d_gemmmap, job->n_gemm, job->gemm_ut, job->d_finmap, job->n_fin, st);
        prof_end(job, st);
        return job_queue_results(job, par, st);
    }
    prof_begin(job, 0, st, 1);
    launch_gram(job->d_items, job->n_items, job->gram_i8, st);
    prof_end(job, st);
    if (side) {
        HIPCHK(hipEventRecord(ev.gram, st));
        HIPCHK(hipStreamWaitEvent(side, ev.gram, 0));
        launch_epilogue(job->d_probs, job->d_tilemap + job->n_tiles_b11, job->n_tiles - job->n_tiles_b11, job->max_pop, job->gram_i8, side);
        HIPCHK(hipEventRecord(ev.side, side));
    }
    prof_begin(job, 2, st);
    launch_epilogue(job->d_probs, job->d_tilemap, side ? job->n_tiles_b11 : job->n_tiles, job->max_pop, job->gram_i8, st);
    for (int i = 0; i < job->n; i++)
        if (job->plans[i].p.n_gene) launch_gene_epilogue(job->d_probs, i, job->plans[i].p.n_gene, st);
    prof_end(job, st);
    if (solve && job->n_panels > 0) {
        for (int i = 0; i < job->n; i++) {
            Plan& pl = job->plans[i];
            if (pl.out_b11 && pl.p.npanel > 0)
                HIPCHK(hipMemcpyAsync(pl.d_b11_copy, pl.p.A, sizeof(double) * pl.p.Mld * pl.p.Mld, hipMemcpyDeviceToDevice, st));
        }
        // fused (default): the rows of [X | y] = L^-1 [I | z1] ride in the factorisation's update launches and the
        // closing product forms z / info (k_solve.hip); the stage timers read "factor" = factorisation + riding rows,
        // "solve" = closing row + product + finish
        prof_begin(job, 3, st);
        for (int s = 0; s < job->max_nblk; s++)
            launch_factor_step(job->d_probs, job->n, s, job->max_nblk, fused ? job->max_npanel : 0, job->solve_split,
                               job->own_panel, st);
        prof_end(job, st);
        prof_begin(job, 4, st);
        if (fused) {
            launch_solve_last(job->d_probs, job->d_panelmap, job->n_panels, job->max_nblk, job->solve_split, st);
            if (side) HIPCHK(hipStreamWaitEvent(st, ev.side, 0));
            launch_impute_gemm(job->d_probs, job->d_gemmmap, job->n_gemm, job->gemm_ut, job->d_finmap, job->n_fin, st);
        } else launch_solve(job->d_probs, job->d_dpanelmap, job->n_dpanels, st);
        prof_end(job, st);
    }
    return job_queue_results(job, par, st);
}

int job_run(gauss_job* job, bool solve)
{
    HIPCHK(hipSetDevice(job->ctx->device));
    if (job->run_seq - job->fetch_seq >= 2u)
        return fail(GAUSS_E_INVALID, "gauss_job_run: two runs of this job are in flight already; fetch one first");
    job->prof_run = job->run_seq;
    job->ran_solve = solve;
    const int rc = job_queue_run(job, solve, (int)(job->run_seq & 1u), true);
    if (rc) return rc;
    job->run_seq++;
    job->ran = true;
    return GAUSS_OK;
}


// One window whose genotype rows are still in HOST memory (the blocking call the Rcpp drivers bind).  Four queues:
//   copy   the rows travel chunk by chunk -- measured rows first, then the unmeasured rows a few row tiles at a time --
//          into the context's landing buffer; issued by the context's copy worker (stream_start_copies), which starts
//          BEFORE the window is planned: a copy from pageable memory (an Rcpp driver's std::vector) returns only when
//          the runtime has staged the bytes, and the PCIe link must wait neither for the planner nor for launches
//   aux    pack + row tables of a chunk as soon as it has landed (and the certificate after the measured rows)
//   main   the Gram launches, one per chunk, back to back: they are what bounds the compute side
//   chain  B11's epilogue tiles and the whole factorisation chain (it needs B11 only, i.e. the first Gram launch):
//          latency-bound launches that slip in between the chunks' Gram launches
// then B21's epilogue tiles, the closing product and the results on the main stream.  Same kernels on the same data as
// job_run: the same bits.
static int ctx_stream_init(gauss_ctx* ctx)
{
    if (ctx->copy) return GAUSS_OK;
    // (the fourth high-priority stream of a context: alone on its device it still gets a hardware queue of its own)
    int rc = ctx_stream_create(ctx, &ctx->aux, STREAM_HIGH);
    if (!rc && !ctx->chain) rc = ctx_stream_create(ctx, &ctx->chain, STREAM_HIGH);
    if (rc) return rc;
    HIPCHK(hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
    ctx->worker = new CopyWorker(ctx->device);
    return GAUSS_OK;
}

// Chunk table, landing buffer and events of a streamed window; then the copy worker is set going.
int stream_start_copies(gauss_ctx* ctx, const gauss_window_desc& win, size_t row_bytes, StreamSetup& su)
{
    int rc = ctx_stream_init(ctx);
    if (rc) return rc;
    su.M = win.n_measured; su.U = win.n_unmeasured; su.row_bytes = row_bytes;
    const bool linear = (size_t)win.ld <= row_bytes + row_bytes / 8 + 64;        // same rule as job_build: one linear copy per chunk
    su.ldraw = linear ? win.ld : (long long)rup(row_bytes, 16);
    // Chunks of `ct` row tiles (the last `lt` tiles may form a closing chunk of their own).  Measured on a mean chr22
    // window (M = 736, U = 2526, 105 MB of genotype bytes, 20 row tiles; tools/window_trace.py, medians of interleaved
    // calls): 4 to 8 tiles per chunk 2.77-2.83 ms, 3 tiles 3.4 ms (every chunk pays under-filled launches and two event
    // hops of ~50 us), a closing chunk of 1 or 2 tiles +0.06-0.1 ms; upload-then-run 3.83 ms
    const int ct = 6;
    const int n_ut = (su.U + TILE - 1) / TILE;
    su.tile_group.assign((size_t)std::max(n_ut, 1), 1);
    su.first_tile = {0, 0};
    for (int t = 0, g = 1; t < n_ut; g++) {
        const int sz = std::min(ct, n_ut - t);
        for (int k = 0; k < sz; k++) su.tile_group[(size_t)(t + k)] = g;
        t += sz;
        su.first_tile.push_back(t);
    }
    const int ng = su.n_groups();
    const size_t need = (size_t)(su.M + su.U) * (size_t)su.ldraw + 256;
    if (ctx->landing_bytes < need) {
        if (ctx->landing) { HIPCHK(hipStreamSynchronize(ctx->copy)); HIPCHK(hipFree(ctx->landing)); ctx->landing = nullptr; ctx->landing_bytes = 0; }
        const size_t want = need + need / 4;
        void* d = nullptr;
        hipError_t e = ctx_malloc_retry(ctx, &d, want);
        if (e != hipSuccess) return fail(GAUSS_E_NOMEM, "hipMalloc(%zu bytes landing buffer) failed: %s", want, hipGetErrorString(e));
        ctx->landing = (uint8_t*)d; ctx->landing_bytes = want;
    }
    su.d_m = ctx->landing;
    su.d_u = ctx->landing + rup((size_t)su.M * (size_t)su.ldraw + 64, 256);
    while ((int)ctx->ev_pool.size() < ng) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ev_pool.push_back(e);
    }
    su.ev.assign(ctx->ev_pool.begin(), ctx->ev_pool.begin() + ng);
    hipStream_t cs = ctx->copy;
    const uint8_t* hm = win.geno_m;
    const uint8_t* hu = win.geno_u;
    const long long user_ld = win.ld;
    StreamSetup* sp = &su;
    ctx->worker->submit([sp, cs, hm, hu, user_ld, ng]() {
        StreamSetup& su = *sp;
        for (int g = 0; g < ng; g++) {
            const int r0 = g == 0 ? 0 : su.first_tile[(size_t)g] * TILE;
            const int r1 = g == 0 ? su.M : std::min(su.U, su.first_tile[(size_t)g + 1] * TILE);
            const int nrows = r1 - r0;
            uint8_t* dst = (g == 0 ? su.d_m : su.d_u) + (size_t)r0 * su.ldraw;
            const uint8_t* src = (g == 0 ? hm : hu) + (size_t)r0 * user_ld;
            hipError_t e = hipSuccess;
            if (nrows > 0) {
                if (su.ldraw == user_ld)
                    e = hipMemcpyAsync(dst, src, (size_t)(nrows - 1) * user_ld + su.row_bytes, hipMemcpyHostToDevice, cs);
                else
                    e = hipMemcpy2DAsync(dst, (size_t)su.ldraw, src, (size_t)user_ld, su.row_bytes, (size_t)nrows, hipMemcpyHostToDevice, cs);
            }
            if (e == hipSuccess) e = hipEventRecord(su.ev[(size_t)g], cs);
            std::lock_guard<std::mutex> lock(su.mu);
            if (e != hipSuccess) {
                su.rc = GAUSS_E_DEVICE;
                su.err = std::string("streamed window: copy of a chunk failed: ") + hipGetErrorString(e);
                su.recorded = ng;
                su.cv.notify_all();
                return;
            }
            su.recorded = g + 1;
            su.cv.notify_all();
        }
    });
    return GAUSS_OK;
}

int job_run_streamed(gauss_job* job, StreamSetup& su)
{
    gauss_ctx* ctx = job->ctx;
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t ax = ctx->aux;
    hipStream_t ch = ctx->chain;
    Plan& pl = job->plans[0];
    const Prob& p = pl.p;
    const size_t ng = job->sgroups.size();
    const gauss_job::RunEvents& ev = job->rev[job->run_seq & 1u];
    job->queue_touched = true;
    HIPCHK(hipEventRecord(job->begin, st));
    HIPCHK(hipMemsetAsync(job->d_status, 0, sizeof(int) * (4 * job->n + 4), st));
    // what job_build queued on the main stream (zeroing, tables) comes before anything on the other queues
    HIPCHK(hipEventRecord(ev.pack, st));
    HIPCHK(hipStreamWaitEvent(ax, ev.pack, 0));
    for (size_t g = 0; g < ng; g++) {
        const gauss_job::StreamGroup& sg = job->sgroups[g];
        {
            // a stream can only wait for an event that HAS been recorded: take chunk g up once the worker has queued
            // "chunk g has landed" behind its copy
            std::unique_lock<std::mutex> lock(su.mu);
            su.cv.wait(lock, [&] { return su.recorded > (int)g; });
            if (su.rc) return fail(su.rc, "%s", su.err.c_str());
        }
        // pack + row tables on a stream of their own: the pack kernel shares the chip with the previous chunk's Gram
        // launch (measured: 2.86 ms per call against 3.07 with pack in front of the Gram launch on the main stream)
        hipStream_t ps = ax;
        HIPCHK(hipStreamWaitEvent(ps, su.ev[g], 0));
        launch_pack_stats(job->d_probs, job->d_rowmap + sg.row0, sg.n_rows, ps);
        if (g == 0) launch_shift_cert(job->d_probs, job->n, ps);
        if (ps != st) {
            HIPCHK(hipEventRecord(job->sevp[g], ps));
            HIPCHK(hipStreamWaitEvent(st, job->sevp[g], 0));
        }
        launch_gram(job->d_items + sg.item0, sg.n_items, job->gram_i8, st);
        if (g == 0) {
            if (ch != st) {
                HIPCHK(hipEventRecord(ev.gram, st));
                HIPCHK(hipStreamWaitEvent(ch, ev.gram, 0));
            }
            launch_epilogue(job->d_probs, job->d_tilemap, job->n_tiles_b11, job->max_pop, job->gram_i8, ch);
            if (pl.out_b11)
                HIPCHK(hipMemcpyAsync(pl.d_b11_copy, p.A, sizeof(double) * p.Mld * p.Mld, hipMemcpyDeviceToDevice, ch));
            for (int s = 0; s < job->max_nblk; s++)
                launch_factor_step(job->d_probs, job->n, s, job->max_nblk, job->max_npanel, job->solve_split, job->own_panel, ch);
            launch_solve_last(job->d_probs, job->d_panelmap, job->n_panels, job->max_nblk, job->solve_split, ch);
            if (ch != st) HIPCHK(hipEventRecord(ev.side, ch));
        }
    }
    launch_epilogue(job->d_probs, job->d_tilemap + job->n_tiles_b11, job->n_tiles - job->n_tiles_b11, job->max_pop, job->gram_i8, st);
    if (ch != st) HIPCHK(hipStreamWaitEvent(st, ev.side, 0));
    launch_impute_gemm(job->d_probs, job->d_gemmmap, job->n_gemm, job->gemm_ut, job->d_finmap, job->n_fin, st);
    const int rc = job_queue_results(job, (int)(job->run_seq & 1u), st);
    if (rc) return rc;
    job->run_seq++;
    job->ran = true;
    job->ran_solve = true;
    return GAUSS_OK;
}

// Rare path: MakePosDef would have modified B11 (util.cpp:310-317).  Rebuild B11 from the
// epilogue, clamp its spectrum on the device (Jacobi), refactor and re-solve this window alone.
static int job_clamp_window(gauss_job* job, int i, int* status_bits)
{
    hipStream_t st = job->ctx->stream;
    Plan& pl = job->plans[i];
    Prob& p = pl.p;
    // Re-run the epilogue for this problem only to restore A[0] (the factorisation overwrote it)
    std::vector<int2> tm;
    tm = job->win_tiles[(size_t)i];                   // this window's epilogue tiles (job-wide B11 pairs included)
    DevBuf d_tm, d_work, d_pm;
    HIPCHK(d_tm.alloc(job->ctx, sizeof(int2) * tm.size()));
    HIPCHK(hipMemcpyAsync(d_tm.p, tm.data(), sizeof(int2) * tm.size(), hipMemcpyHostToDevice, st));
    launch_epilogue(job->d_probs, d_tm.as<int2>(), (int)tm.size(), job->max_pop, job->gram_i8, st);
    HIPCHK(hipGetLastError());
    const size_t n = (size_t)p.Mld;
    HIPCHK(d_work.alloc(job->ctx, sizeof(double) * (2 * n * n + 4 * n)));
    HIPCHK(hipMemsetAsync(p.status, 0, sizeof(int) * 4, st));
    launch_jacobi_clamp(job->d_probs, i, p, d_work.as<double>(), true, st);
    HIPCHK(hipGetLastError());
    // refactor (both matrices are factored again; only matrix 0 is used) and solve this window
    std::vector<int2> pm;
    for (int pn = 0; pn < p.npanel; pn++) pm.push_back(make_int2(i, pn));
    HIPCHK(d_pm.alloc(job->ctx, sizeof(int2) * pm.size()));
    HIPCHK(hipMemcpyAsync(d_pm.p, pm.data(), sizeof(int2) * pm.size(), hipMemcpyHostToDevice, st));
    if (pl.out_b11) HIPCHK(hipMemcpyAsync(pl.d_b11_copy, p.A, sizeof(double) * n * n, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(p.A + 4 * n * n, p.A, sizeof(double) * n * n, hipMemcpyDeviceToDevice, st));   // W0 = clamped B11
    for (int s = 0; s < p.nblk; s++) {
        // launch over all problems would redo the others; use a single-problem launch instead
        launch_factor_step(job->d_probs + i, 1, s, p.nblk, 0, 0, 0, st);
    }
    launch_solve(job->d_probs, d_pm.as<int2>(), (int)pm.size(), st);
    HIPCHK(hipGetLastError());
    int h_status[4];
    HIPCHK(hipMemcpyAsync(h_status, p.status, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(job->h_results + pl.res_off, job->d_results + pl.res_off, sizeof(double) * 2 * p.n_rhs, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *status_bits = (h_status[2] || h_status[0]) ? GAUSS_ST_NONFINITE : GAUSS_ST_CLAMPED;
    return GAUSS_OK;
}

// Device matrix [rows x pitch] -> host [rows x width] doubles.  One linear copy into a staging buffer and a
// row-wise compaction on the host: a pitched device-to-host copy of a few thousand rows is many times slower.
static int fetch_matrix(double* dst, const double* d_src, int rows, int width, int pitch)
{
    if (rows <= 0 || width <= 0) return GAUSS_OK;
    if (pitch == width) { HIPCHK(hipMemcpy(dst, d_src, sizeof(double) * (size_t)rows * width, hipMemcpyDeviceToHost)); return GAUSS_OK; }
    std::vector<double> tmp((size_t)(rows - 1) * pitch + width);
    HIPCHK(hipMemcpy(tmp.data(), d_src, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
    for (int r = 0; r < rows; r++) memcpy(dst + (size_t)r * width, tmp.data() + (size_t)r * pitch, sizeof(double) * width);
    return GAUSS_OK;
}

// CountPC (util.cpp:355-388) when the smallest eigenvalue of B11 is below the cutoff: eigenvalues by the
// device Jacobi sweep, counted on the host (the matrix itself is left alone).
static int job_count_small_eigs(gauss_job* job, int i, int* num_eig)
{
    hipStream_t st = job->ctx->stream;
    Plan& pl = job->plans[i];
    Prob& p = pl.p;
    const size_t n = (size_t)p.Mld;
    DevBuf d_work;
    HIPCHK(d_work.alloc(job->ctx, sizeof(double) * (2 * n * n + 4 * n)));
    launch_jacobi_clamp(job->d_probs, i, p, d_work.as<double>(), false, st);
    HIPCHK(hipGetLastError());
    std::vector<double> delta(n);
    HIPCHK(hipMemcpyAsync(delta.data(), d_work.as<double>() + 2 * n * n, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int small = 0;
    for (size_t k = 0; k < n; k++) if (delta[k] > 0.0) small++;
    *num_eig = p.M - small;
    return GAUSS_OK;
}

// The exported matrices of chunks [c0, c1) from the pinned mirror to the caller's memory, each chunk as soon as its event is complete.
static int export_copy_out(gauss_job* job, size_t c0, size_t c1, std::atomic<size_t>* next)
{
    for (;;) {
        const size_t c = next ? next->fetch_add(1) : c0++;
        if (c >= c1) return GAUSS_OK;
        const auto t0 = std::chrono::steady_clock::now();
        if (hipEventSynchronize(job->exp_ev[c]) != hipSuccess) return GAUSS_E_DEVICE;
        const auto t1 = std::chrono::steady_clock::now();
        size_t bytes = 0;
        for (int x = job->exp_chunks[c].first; x < job->exp_chunks[c].second; x++) {
            const gauss_job::Export& ex = job->exports[(size_t)x];
            memcpy(ex.user, job->h_export + ex.off, sizeof(double) * (size_t)ex.rows * ex.width);
            bytes += sizeof(double) * (size_t)ex.rows * ex.width;
        }
        if (trace_on("job")) {
            const auto t2 = std::chrono::steady_clock::now();
            fprintf(stderr, "[job] export chunk %zu: waited %.3f ms, copied %.2f MB in %.3f ms\n", c, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    bytes / 1e6, std::chrono::duration<double, std::milli>(t2 - t1).count());
        }
    }
}

static int export_fetch_all(gauss_job* job)
{
    const size_t nc = job->exp_chunks.size();
    if (!nc) return GAUSS_OK;
    size_t bytes = 0;
    for (const gauss_job::Export& ex : job->exports) bytes += sizeof(double) * (size_t)ex.rows * ex.width;
    // one host thread copies ~10 GB/s; the link delivers ~50: large exports take a few helpers (the caller's thread is one of them)
    const int helpers = bytes >= ((size_t)16 << 20) ? (int)std::min<size_t>(3, nc - 1) : 0;
    std::atomic<size_t> next{0};
    std::atomic<int> rc_all{0};
    std::vector<std::thread> th;
    const int device = job->ctx->device;
    for (int t = 0; t < helpers; t++)
        th.emplace_back([&, device]() { (void)hipSetDevice(device); const int rc = export_copy_out(job, 0, nc, &next); if (rc) rc_all = rc; });
    const int rc = export_copy_out(job, 0, nc, &next);
    for (std::thread& t : th) t.join();
    if (rc || rc_all) return fail(GAUSS_E_DEVICE, "waiting for a matrix export failed");
    return GAUSS_OK;
}

int job_fetch(gauss_job* job)
{
    if (!job->ran || job->fetch_seq == job->run_seq) return fail(GAUSS_E_INVALID, "gauss_job_fetch: no run of this job is waiting to be fetched");
    hipStream_t st = job->ctx->stream;
    HIPCHK(hipSetDevice(job->ctx->device));
    const int par = (int)(job->fetch_seq & 1u);
    job->h_results = job->h_res2[par];
    job->h_status = job->h_st2[par];
    // the exported matrices leave the pinned mirror chunk by chunk while the rest of the run is still on the queue (with a later run
    // of the job in flight that run rewrites the mirror with the same values: it is left to finish first, as for any device read)
    const bool exporting = !job->exports.empty();
    if (exporting) {
        if (job->run_seq - job->fetch_seq > 1u) HIPCHK(hipEventSynchronize(job->done));
        const int rc = export_fetch_all(job);
        if (rc) return rc;
    }
    HIPCHK(hipEventSynchronize(job->done2[par]));
    bool rerun = false;
    if (job->h_status[4 * job->n] != 0) {
        rerun = true;
        // A waiting kernel of the merged launch gave up (k_gram.hip: wait_count_kernel raises this job-wide flag after its
        // bound): what the chain computed from then on is not valid.  The run is queued once more in the two-launch form,
        // which has no kernel that waits for another queue, into the same mirrors -- the job's inputs do not change between
        // runs -- after everything of this job has left the queues (a later run of the job that is in flight included: its
        // results sit in the other parity's mirrors and stay valid).
        gauss_ctx* ctx = job->ctx;
        ctx->n_merged_giveups++;
        job->n_giveups++;
        for (hipStream_t q : {ctx->stream, ctx->chain, ctx->side}) if (q) HIPCHK(hipStreamSynchronize(q));
        // the stage timers of the run being replaced are void (its chain ran on nothing): dropped, so that the re-run's timers do not
        // add to them; and the run number the timers are recorded under goes back to the newest queued run afterwards (a later run of
        // the job may be in flight: its slots keep their number)
        {
            std::vector<ProfSlot> keep;
            for (ProfSlot& ps : job->slots) {
                if (ps.run == job->fetch_seq) { hipEventDestroy(ps.a); hipEventDestroy(ps.b); }
                else keep.push_back(ps);
            }
            job->slots.swap(keep);
        }
        const unsigned prof_run_was = job->prof_run;
        job->prof_run = job->fetch_seq;
        int rc = job_queue_run(job, job->ran_solve, par, false);
        job->prof_run = prof_run_was;
        if (!rc && hipEventSynchronize(job->done2[par]) != hipSuccess) rc = fail(GAUSS_E_DEVICE, "waiting for the re-run failed");
        if (rc || job->h_status[4 * job->n] != 0) {
            ctx->n_rerun_failed++;
            job->n_rerun_failed++;
            job->fetch_seq++;
            const std::string why = rc ? g_err : std::string("its failure flag is set again");
            return fail(GAUSS_E_DEVICE, "the chain queue gave up waiting for B11's tile pairs of this run (merged Gram launch) and the re-run in the two-launch form failed: %s", why.c_str());
        }
        if (job->run_seq - job->fetch_seq > 1u) job->done = job->done2[par ^ 1];      // (the later run's event stays the newest: it was synchronised above)
    }
    if (rerun && exporting) { const int rc = export_fetch_all(job); if (rc) return rc; }      // what the failed merged run exported is void
    // With a later run of the job already queued, anything that reads the job's DEVICE buffers (matrix exports, the
    // clamp path, the eigenvalue count) first lets that run finish: the job's inputs do not change between runs, so
    // what it leaves on the device is what the fetched run left.
    if (job->run_seq - job->fetch_seq > 1u) {
        bool device_reads = false;
        for (int i = 0; i < job->n && !device_reads; i++) {
            const Plan& pl = job->plans[i];
            device_reads = pl.out_b11 || pl.out_b21 || (pl.out_ld_user && pl.out_ld_count) ||
                           job->h_status[4 * i + 0] || job->h_status[4 * i + 1];
        }
        if (device_reads) HIPCHK(hipEventSynchronize(job->done));
    }
    job->fetch_seq++;
    for (int i = 0; i < job->n; i++) {
        Plan& pl = job->plans[i];
        const Prob& p = pl.p;
        int bits = 0;
        if (p.kind == GAUSS_WIN_LD) {
            // raw LD export: B11 sits unfactored in A[0] (diagonal 1 + lambda), B21 in its buffer
            if (pl.out_b11 && !exporting)
                { int rc2 = fetch_matrix(pl.out_b11, p.A, p.M, p.M, p.Mld); if (rc2) return rc2; }
            if (pl.out_b21 && p.U > 0 && !exporting)
                { int rc2 = fetch_matrix(pl.out_b21, p.B21, p.U, p.M, p.Mld); if (rc2) return rc2; }
            if (pl.out_status) *pl.out_status = 0;
            continue;
        }
        if (p.kind == GAUSS_WIN_QCAT) {
            // QCAT never repairs B11 (MakePosDef is commented out, qcat.cpp:206); CountPC only counts
            int num_eig = p.M;
            if (job->h_status[4 * i + 0]) bits = GAUSS_ST_NONFINITE;          // B11 has no Cholesky factor
            // (no factor: B11 is indefinite -- weights summing far above 1 -- or not finite.  The reference still counts: CountPC runs
            // before the factorisation, qcat.cpp:203, and an eigenvalue below the cutoff, negative ones included, is not counted)
            if (job->h_status[4 * i + 0] || job->h_status[4 * i + 1]) { int rc = job_count_small_eigs(job, i, &num_eig); if (rc) return rc; }
            if (bits & GAUSS_ST_NONFINITE)
                for (int u = 0; u < 2 * p.n_rhs; u++) job->h_results[pl.res_off + u] = NAN;
            if (pl.out_r) memcpy(pl.out_r, job->h_results + pl.res_off, sizeof(double) * p.n_rhs);
            if (pl.out_num_eig) *pl.out_num_eig = num_eig;
            if (pl.out_status) *pl.out_status = bits;
            if (pl.out_b11 && !exporting)
                { int rc2 = fetch_matrix(pl.out_b11, pl.d_b11_copy, p.M, p.M, p.Mld); if (rc2) return rc2; }
            if (pl.out_b21 && p.U > 0 && !exporting)
                { int rc2 = fetch_matrix(pl.out_b21, p.B21, p.U, p.M, p.Mld); if (rc2) return rc2; }
            continue;
        }
        bool clamped = false;
        if (p.npanel > 0 && (job->h_status[4 * i + 0] || job->h_status[4 * i + 1])) {
            int rc = job_clamp_window(job, i, &bits);
            if (rc) return rc;
            clamped = true;                    // B11's exported copy predates the clamp: fetched again below
        }
        if (p.npanel > 0) {
            if (bits & GAUSS_ST_NONFINITE) {
                // the reference's eigen-solver / LU propagate non-finite values to every output
                for (int u = 0; u < 2 * p.U; u++) job->h_results[pl.res_off + u] = NAN;
            }
            if (pl.out_z) memcpy(pl.out_z, job->h_results + pl.res_off, sizeof(double) * p.U);
            if (pl.out_info) memcpy(pl.out_info, job->h_results + pl.res_off + p.U, sizeof(double) * p.U);
            if (pl.out_b11 && (!exporting || clamped))
                { int rc2 = fetch_matrix(pl.out_b11, pl.d_b11_copy, p.M, p.M, p.Mld); if (rc2) return rc2; }
            if (pl.out_b21 && p.U > 0 && !exporting)
                { int rc2 = fetch_matrix(pl.out_b21, p.B21, p.U, p.M, p.Mld); if (rc2) return rc2; }
        }
        if (pl.out_status) *pl.out_status = bits;
        if (pl.out_ld_user && pl.out_ld_count)
            HIPCHK(hipMemcpy(pl.out_ld_user, p.out_ld, sizeof(double) * pl.out_ld_count, hipMemcpyDeviceToHost));
    }
    if (job->prof) prof_collect(job, job->fetch_seq);      // the slots of the run just fetched (fetch_seq already counts it)
    if (job->fetch_seq == job->run_seq) job->queue_touched = false;      // every run has delivered: nothing of this job is left on a queue
    return GAUSS_OK;
}

// Everything a job holds on its context: waits for its queued runs, then gives the blocks back and destroys the events.
// Called by job_free, and by gauss_hip_destroy for the jobs that outlive their context (the context is still whole then).
void job_release(gauss_job* job)
{
    gauss_ctx* ctx = job->ctx;
    if (!ctx) return;
    hipSetDevice(ctx->device);
    // runs that were queued and never fetched: their result copies target this job's pinned block
    if (job->run_seq != job->fetch_seq && job->done) (void)hipEventSynchronize(job->done);
    // Anything else of this job that no completed fetch covers -- a run whose queuing failed half way (kernels are queued,
    // run_seq was not advanced), a job that was built and never run (its workspace is being zeroed), a streamed window that
    // failed with row copies in flight: the queues drain before the blocks go back to the cache, where the next job would
    // zero and reuse them on another queue.
    if (job->queue_touched) {
        for (hipStream_t q : {ctx->stream, ctx->chain, ctx->side, job->zero_queue}) if (q) (void)hipStreamSynchronize(q);
        if (!job->sgroups.empty() && ctx->aux) (void)hipStreamSynchronize(ctx->aux);
        job->queue_touched = false;
    }
    for (ProfSlot& s : job->slots) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
    job->slots.clear();
    ctx_dev_release(ctx, job->d_ws);
    ctx_dev_release(ctx, job->d_tab);
    ctx_pin_release(ctx, job->h_pin);
    job->d_ws = nullptr; job->d_tab = nullptr; job->h_pin = nullptr;
    if (job->begin) hipEventDestroy(job->begin);
    if (job->zeroed) { hipEventDestroy(job->zeroed); job->zeroed = nullptr; }
    for (int k = 0; k < 2; k++) if (job->done2[k]) hipEventDestroy(job->done2[k]);
    for (int k = 0; k < 2; k++)
        for (hipEvent_t* e : {&job->rev[k].gram, &job->rev[k].side, &job->rev[k].pack, &job->rev[k].rows, &job->rev[k].epi})
            if (*e) { hipEventDestroy(*e); *e = nullptr; }
    for (hipEvent_t e : job->sevp) if (e) hipEventDestroy(e);
    job->sevp.clear();
    for (hipEvent_t e : job->exp_ev) if (e) hipEventDestroy(e);
    job->exp_ev.clear();
    job->begin = job->done = nullptr;
    job->done2[0] = job->done2[1] = nullptr;
    { std::lock_guard<std::mutex> lock(ctx->mu); ctx->jobs.erase(job); }
    job->ctx = nullptr;                       // from here on the handle is an orphan: only gauss_job_destroy accepts it
}

void job_free(gauss_job* job)
{
    if (!job) return;
    job_release(job);
    delete job;
}
