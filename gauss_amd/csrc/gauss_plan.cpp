// libgauss_hip.so -- the planner: one window -> Plan (segments, tile pairs, tables), a batch of windows -> a job.
//
// A job is a batch of independent windows that share every launch: one pack, one Gram, one
// epilogue, nblk factor steps and one solve launch serve all windows of the job, so the chip is
// filled by work items of many windows at once (the per-window matrices are too small to fill
// 256 CUs on their own) and the latency-bound factor steps are amortised over the batch.
#include "gauss_job.h"

// K segment length: short segments give a single window enough work items to fill 256 CUs;
// a batch of windows has plenty of items already, and longer segments mean fewer partial slabs
// for the epilogue to read back.  Either way a partial sum stays an exact f32 integer
// (15 * 15 * 8192 < 2^24).
static int seg_max_for(const std::vector<WinSpec>& specs)
{
    // 4096 for batched jobs (8192 until the chain moved beside the Gram kernel: with B11's and B21's items in launches of their
    // own the shorter items balance each launch's last round better -- 5 / 9 / 18 / 36 windows: step 4.86 -> 4.75, 8.97 -> 8.93,
    // 19.47 -> 19.29, 41.00 -> 40.96 ms -- for 19 % more work items)
    if (specs.size() >= 4) return 4096;
    // One to three windows: a work item per (tile pair, segment), so the segment length decides how many workgroups the launch
    // has.  A window with few tile pairs -- computeLD()'s one 3 Mb window: 529 SNPs, 15 pairs -- left three quarters of the chip's
    // 1 024 workgroup slots empty at 2 048 samples a segment; segments are cut so that the job has ~1 300 items, down to 384 samples
    // (below that the epilogue's reads of the partial slabs cost what the Gram launch gains).  Measured on that window, Gram
    // launch / LD epilogue in us: 2048: 196 / 59 (46 TFLOP/s); 1024: 154 / 67; 768: 140 / 69; 512: 127 / 74; 384: 120 / 81
    // (75 TFLOP/s); 256: 116 / 92.  Any cut is exact (integer partial sums); listed-pair and gene jobs keep 2 048.
    double pairs = 0, kp = 0;
    for (const WinSpec& w : specs) {
        if (w.gene_off || w.pair_i || w.M < 1 || w.n_pop < 1 || w.n_pop > 64 || !w.pop_off) return SEG_MAX;      // (plan_problem reports what is wrong)
        int nc = 0;
        for (int c = 0; c < 3; c++) nc += (w.u_codings >> c) & 1;
        const double mt = (w.M + TILE - 1) / TILE, ut = ((double)w.U * std::max(nc, 1) + TILE - 1) / TILE;
        pairs += mt * (mt + 1) / 2 + ut * mt;
        kp = std::max(kp, (double)w.pop_off[w.n_pop] + 32.0 * w.n_pop);
    }
    const double want = pairs * kp / 1300.0;
    return (int)std::min<double>(SEG_MAX, std::max<double>(384.0, std::floor(want / KC) * KC));
}
// Consecutive segments are chained into one work item until the run reaches this many samples
// (a fresh item costs a pipeline fill: descriptor, first operand tiles, barrier).  0 = no chaining.
static int group_target_for(size_t n_windows)
{
    // 2048 rather than 4096: same kernel time on the bench workload, 13 % less fabric read traffic (the tiles
    // co-resident items share stay in the XCD's L2 more often; tools/group_pmc.sh)
    return n_windows >= 4 ? 2048 : 0;
}
int plan_problem(const WinSpec& w, Plan& pl, int seg_max, int group_target)
{
    if (w.mode != GAUSS_MODE_POOLED && w.mode != GAUSS_MODE_WEIGHTED) return fail(GAUSS_E_INVALID, "bad mode %d", w.mode);
    if (w.n_pop < 1 || w.n_pop > 64) return fail(GAUSS_E_INVALID, "n_pop must be in 1..64 (got %d)", w.n_pop);
    if (!w.pop_off) return fail(GAUSS_E_INVALID, "pop_off is NULL");
    if (w.M < 1) return fail(GAUSS_E_INVALID, "need at least one measured SNP row (got %d)", w.M);
    if (w.U < 0) return fail(GAUSS_E_INVALID, "negative n_unmeasured");
    if (w.mode == GAUSS_MODE_WEIGHTED && !w.pop_wgt) return fail(GAUSS_E_INVALID, "pop_wgt is NULL in weighted mode");
    if (!w.geno_m || (w.U > 0 && !w.geno_u)) return fail(GAUSS_E_INVALID, "genotype pointer is NULL");
    for (int p = 0; p < w.n_pop; p++)
        if (w.pop_off[p + 1] < w.pop_off[p]) return fail(GAUSS_E_INVALID, "pop_off must be non-decreasing");
    if (w.pop_off[0] != 0) return fail(GAUSS_E_INVALID, "pop_off[0] must be 0");
    const int N = w.pop_off[w.n_pop];
    if (N < 1) return fail(GAUSS_E_INVALID, "no samples");
    // sums of code products are kept as exact integers: 15 * 15 * n must stay below 2^31
    if ((long long)N * 225 >= (1LL << 31)) return fail(GAUSS_E_RANGE, "%d samples exceed the exact-integer range (9.5 M)", N);
    if (w.geno_fmt != GAUSS_GENO_U8 && w.geno_fmt != GAUSS_GENO_2BIT) return fail(GAUSS_E_INVALID, "bad geno_format %d", w.geno_fmt);
    if (w.geno_fmt == GAUSS_GENO_U8 && w.ld < N) return fail(GAUSS_E_INVALID, "ld (%lld) < n_samples (%d)", w.ld, N);
    if (w.geno_fmt == GAUSS_GENO_2BIT) {
        if (w.ld % 16) return fail(GAUSS_E_INVALID, "2-bit rows need a stride that is a multiple of 16 bytes (got %lld)", w.ld);
        long long end = 0;
        for (int q = 0; q < w.n_pop; q++) {
            const long long blk = (long long)rup((size_t)(w.pop_off[q + 1] - w.pop_off[q]), 64) / 4;
            const long long off = w.pop_src_off ? w.pop_src_off[q] : end;
            if (off < 0 || off % 16) return fail(GAUSS_E_INVALID, "pop_src_off[%d] = %lld is not a multiple of 16", q, off);
            if (off + blk > w.ld) return fail(GAUSS_E_INVALID, "population block %d ends past the row stride", q);
            if (!w.pop_src_off) end = off + blk;
        }
    }
    if (w.kind != GAUSS_WIN_IMPUTE && w.kind != GAUSS_WIN_QCAT && w.kind != GAUSS_WIN_LD)
        return fail(GAUSS_E_INVALID, "bad window kind %d", w.kind);
    if (!w.ld_only && w.kind != GAUSS_WIN_LD && (w.U > 0 || w.kind == GAUSS_WIN_QCAT) && !w.z1) return fail(GAUSS_E_INVALID, "z1 is NULL");
    if (w.u_codings & ~(GAUSS_CODE_ADDITIVE | GAUSS_CODE_DOMINANT | GAUSS_CODE_RECESSIVE))
        return fail(GAUSS_E_INVALID, "bad u_codings mask %d", w.u_codings);
    if (w.kind == GAUSS_WIN_QCAT && (w.n_head < 0 || w.n_predm < 0 || w.n_head + w.n_predm > w.M))
        return fail(GAUSS_E_INVALID, "QCAT: n_head_measured + n_pred_measured exceeds n_measured");

    Prob& p = pl.p;
    memset(&p, 0, sizeof(p));
    p.mode = w.mode;
    p.M = w.M; p.N = N;
    {
        int nc = 0;
        for (int c = 0; c < 3; c++) if (w.u_codings & (1 << c)) p.code_blk[nc++] = c;
        if (nc == 0) { p.code_blk[0] = 0; nc = 1; }
        p.U_raw = std::max(w.U, 1);
        p.U = w.U * nc;                          // one block of U rows per coding
    }
    p.lambda = w.lambda; p.diag = w.diag;
    p.ld_only = w.ld_only;
    p.kind = w.kind; p.n_head = w.n_head; p.n_predm = w.n_predm;
    // the shifted factorisation tests lambda_min against MakePosDef's floor (imputation, util.cpp:310)
    // or against CountPC's cutoff (QCAT, util.cpp:379)
    p.eps = (w.kind == GAUSS_WIN_QCAT) ? w.eig_cutoff : w.eps;
    p.n_rhs = (w.kind == GAUSS_WIN_QCAT) ? w.n_predm + p.U : p.U;
    if (w.mode == GAUSS_MODE_POOLED) {
        // CalCor pools every selected population (util.cpp:53-64): one pseudo-population
        p.P = 1;
        pl.pop_raw_off = {0, N};
        pl.pop_w = {1.0};
    } else {
        p.P = w.n_pop;
        pl.pop_raw_off.assign(w.pop_off, w.pop_off + w.n_pop + 1);
        pl.pop_w.assign(w.pop_wgt, w.pop_wgt + w.n_pop);
    }
    const int P = p.P;
    for (int q = 0; q < P; q++) {
        const int m = pl.pop_raw_off[q + 1] - pl.pop_raw_off[q];
        const double factor = ((double)m) / (m - 1);           // util.cpp:117 (inf for m == 1, like the reference)
        pl.pop_wf.push_back(pl.pop_w[q] * factor);             // util.cpp:118: wgt_val*factor*(...) groups left to right
        pl.pop_md.push_back((double)m);
    }
    pl.pop_pk_off.assign(P + 1, 0);
    for (int q = 0; q < P; q++) {
        const int m = pl.pop_raw_off[q + 1] - pl.pop_raw_off[q];
        pl.pop_pk_off[q + 1] = pl.pop_pk_off[q] + (int)rup((size_t)m, KC);
    }
    p.geno_fmt = w.geno_fmt;
    pl.row_bytes = (size_t)N;
    if (w.geno_fmt == GAUSS_GENO_2BIT) {
        // every selected population is one source block ("run") padded to 64 samples, in the source row and in
        // the packed operand row alike; pooled statistics still see one pseudo-population spanning all runs
        pl.run_pk_off.assign(w.n_pop + 1, 0);
        long long end = 0;
        pl.row_bytes = 0;
        for (int q = 0; q < w.n_pop; q++) {
            const int blk = (int)rup((size_t)(w.pop_off[q + 1] - w.pop_off[q]), 64);
            pl.run_pk_off[q + 1] = pl.run_pk_off[q] + blk;
            const long long off = w.pop_src_off ? w.pop_src_off[q] : end;
            pl.run_src.push_back((int)off);
            pl.run_len.push_back(w.pop_off[q + 1] - w.pop_off[q]);
            if (!w.pop_src_off) end = off + blk / 4;
            pl.row_bytes = std::max(pl.row_bytes, (size_t)(off + blk / 4));
        }
        if (P == 1) pl.pop_pk_off[1] = pl.run_pk_off[w.n_pop];
        p.n_run = w.n_pop;
        pl.word_run.assign(pl.run_pk_off[w.n_pop] / 16, 0);
        for (int q = 0; q < w.n_pop; q++)
            for (int b = pl.run_pk_off[q] / 16; b < pl.run_pk_off[q + 1] / 16; b++) pl.word_run[b] = (uint8_t)q;
    }
    if (w.rows_m) pl.rows_m.assign(w.rows_m, w.rows_m + w.M);
    if (w.rows_u && w.U > 0) pl.rows_u.assign(w.rows_u, w.rows_u + w.U);
    p.Kp = pl.pop_pk_off[P];
    // 16-bit partial slabs: 2-bit sources carry codes 0..3 (recoding only lowers them), so a segment of at most 7168
    // samples sums to <= 9 * 7168 < 2^16; the fast epilogue reads them (windows with LDS-resident population tables);
    // LD-only calls and gene batches keep f32 / int32 slabs
    p.slab16 = (w.geno_fmt == GAUSS_GENO_2BIT && !w.ld_only && !w.gene_off && P <= 32) ? 1 : 0;
    if (p.slab16) seg_max = std::min(seg_max, 7168);
    pl.word_pop.assign(p.Kp / 16, 0);
    // K chunks that end a zero-padded block (a population, or a 2-bit source block) with fewer than 64 live samples: how many
    // units of 8 samples are live (Item::chunk_live; 0 = the whole chunk)
    pl.chunk_live.assign((p.Kp / KC + 7) / 8 + 1, 0u);
    auto set_live = [&](int chunk, int samples) {
        const int units = (samples + 7) / 8;
        if (units >= 1 && units < 8) pl.chunk_live[chunk >> 3] |= (uint32_t)units << (4 * (chunk & 7));
    };
    if (w.geno_fmt == GAUSS_GENO_2BIT) {
        for (int q = 0; q < w.n_pop; q++) {
            const int m = w.pop_off[q + 1] - w.pop_off[q], rem = m % KC;
            if (m > 0 && rem) set_live(pl.run_pk_off[q + 1] / KC - 1, rem);
        }
    } else {
        for (int q = 0; q < P; q++) {
            const int m = pl.pop_raw_off[q + 1] - pl.pop_raw_off[q], rem = m % KC;
            if (m > 0 && rem) set_live(pl.pop_pk_off[q + 1] / KC - 1, rem);
        }
    }
    pl.pop_seg0.assign(P + 1, 0);
    for (int q = 0; q < P; q++) {
        for (int b = pl.pop_pk_off[q] / 16; b < pl.pop_pk_off[q + 1] / 16; b++) pl.word_pop[b] = (uint8_t)q;
        const int chunks = (pl.pop_pk_off[q + 1] - pl.pop_pk_off[q]) / KC;
        pl.pop_seg0[q] = (int)pl.seg_pop.size();
        if (chunks > 0) {
            const int max_chunks = seg_max / KC;
            const int ns = (chunks + max_chunks - 1) / max_chunks;
            const int per = (chunks + ns - 1) / ns;
            for (int c = 0; c < chunks; c += per) {
                const int c1 = std::min(chunks, c + per);
                pl.seg_pop.push_back(q);
                pl.seg_k0.push_back(pl.pop_pk_off[q] + c * KC);
                pl.seg_k1.push_back(pl.pop_pk_off[q] + c1 * KC);
            }
        }
    }
    pl.pop_seg0[P] = (int)pl.seg_pop.size();
    p.nseg = (int)pl.seg_pop.size();
    for (int s0 = 0; s0 < p.nseg;) {
        int s1 = s0 + 1;
        int len = pl.seg_k1[s0] - pl.seg_k0[s0];
        while (s1 < p.nseg && len < group_target) { len += pl.seg_k1[s1] - pl.seg_k0[s1]; s1++; }
        pl.groups.push_back(std::make_pair(s0, s1));
        s0 = s1;
    }
    for (int s0 = 0; s0 < p.nseg; s0++) pl.fine.push_back(std::make_pair(s0, s0 + 1));

    p.Mp = (int)rup((size_t)w.M, TILE);
    p.Up = (int)rup((size_t)p.U, TILE);
    p.Sp = p.Mp + p.Up;
    p.nT = p.Sp / TILE;
    const int mt = p.Mp / TILE;
    pl.pair_lut.assign((size_t)p.nT * p.nT, -1);
    auto add_pair = [&](int ti, int tj) {
        if (pl.pair_lut[(size_t)ti * p.nT + tj] >= 0) return;
        const int id = (int)pl.pair_ti.size();
        pl.pair_ti.push_back(ti); pl.pair_tj.push_back(tj);
        pl.pair_lut[(size_t)ti * p.nT + tj] = id;
        pl.pair_lut[(size_t)tj * p.nT + ti] = id;
    };
    if (w.gene_off) {
        // LD is only needed inside genes (gene.cpp:305-315): tile pairs touched by some gene
        pl.gene_off.assign(w.gene_off, w.gene_off + w.n_gene + 1);
        long long off = 0;
        for (int g = 0; g < w.n_gene; g++) {
            const int r0 = pl.gene_off[g], r1 = pl.gene_off[g + 1];
            if (r0 < 0 || r1 < r0 || r1 > w.M) return fail(GAUSS_E_INVALID, "gene_off out of range at gene %d", g);
            pl.gene_out_off.push_back(off);
            off += (long long)(r1 - r0) * (r1 - r0);
            if (r1 > r0)
                for (int ti = r0 / TILE; ti <= (r1 - 1) / TILE; ti++)
                    for (int tj = ti; tj <= (r1 - 1) / TILE; tj++) add_pair(ti, tj);
        }
        pl.out_ld_count = (size_t)off;
        p.n_gene = w.n_gene;
    } else if (w.pair_i) {
        // listed pairs (prep_zmix selectors): the tile pairs they touch
        for (int64_t k = 0; k < w.n_pairs; k++) {
            const int i = w.pair_i[k], j = w.pair_j[k];
            if (i < 0 || j <= i || j >= w.M) return fail(GAUSS_E_INVALID, "pair %lld = (%d, %d) is not i < j < n_snp", (long long)k, i, j);
            add_pair(i / TILE, j / TILE);
        }
        pl.out_ld_count = 0;
    } else {
        for (int ti = 0; ti < mt; ti++)
            for (int tj = ti; tj < mt; tj++) add_pair(ti, tj);          // B11 (upper tiles)
        for (int tu = mt; tu < p.nT; tu++)
            for (int tj = 0; tj < mt; tj++) add_pair(tu, tj);           // B21
        if (w.ld_only) pl.out_ld_count = (size_t)w.M * w.M;
    }
    p.npair = (int)pl.pair_ti.size();
    p.Mld = (int)rup((size_t)w.M, NB);
    p.nblk = p.Mld / NB;
    p.npanel = (w.ld_only || w.kind == GAUSS_WIN_LD) ? 0 : (p.n_rhs + NRU - 1) / NRU;
    p.npi = p.npanel > 0 ? (w.M + 1 + NR - 1) / NR : 0;
    p.Up128 = (int)rup((size_t)std::max(p.n_rhs, 1), 128);
    pl.U_user = w.U;
    if (w.z1) pl.z1.assign(w.z1, w.z1 + w.M);
    pl.h_geno_m = w.geno_m; pl.h_geno_u = w.geno_u; pl.user_ld = w.ld;
    pl.out_b11 = w.out_b11; pl.out_b21 = w.out_b21;
    return GAUSS_OK;
}

// Arena layout helper
struct Arena {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off = rup(off + bytes, 256); return o; }
};

template <typename T>
static size_t put(std::vector<char>& blob, Arena& a, const std::vector<T>& v)
{
    const size_t bytes = std::max<size_t>(v.size() * sizeof(T), 1);
    const size_t o = a.take(bytes);
    if (blob.size() < a.off) blob.resize(a.off);
    if (!v.empty()) memcpy(blob.data() + o, v.data(), v.size() * sizeof(T));
    return o;
}
// streamed: one window on contiguous host matrices whose upload is left to job_run_streamed (chunk by chunk on the
// copy stream, overlapped with the pack / Gram launches of the rows that have landed)
int job_build(gauss_ctx* ctx, const std::vector<WinSpec>& specs, int on_device, gauss_job** out, const StreamSetup* stream)
{
    const bool streamed = stream != nullptr;
    const auto tb0 = std::chrono::steady_clock::now();
    gauss_job* job = new gauss_job();
    // every early return below (bad arguments, a failed HIP call) releases the job and what it owns
    std::unique_ptr<gauss_job, void (*)(gauss_job*)> guard(job, job_free);
    job->ctx = ctx;
    job->n = (int)specs.size();
    job->on_device = on_device;
    job->gram_i8 = ctx->gram_i8;
    job->plans.resize(job->n);
    const int seg_max = seg_max_for(specs);
    for (int i = 0; i < job->n; i++) {
        int rc = plan_problem(specs[i], job->plans[i], seg_max, group_target_for(specs.size()));
        if (rc) return rc;
        job->plans[i].p.gram_i8 = job->gram_i8;
    }
    // Matrix exports (gauss_job.h): which matrices the caller wants back, and where each lands in the pinned mirror
    size_t exp_doubles = 0;
    if (!streamed) {
        for (int i = 0; i < job->n; i++) {
            const Plan& pl = job->plans[i];
            const Prob& p = pl.p;
            const bool has = p.kind == GAUSS_WIN_LD || p.npanel > 0;          // (the windows gauss_job_fetch exports matrices for)
            if (!has) continue;
            if (pl.out_b11) { job->exports.push_back(gauss_job::Export{i, 0, exp_doubles, p.M, p.M, p.Mld, pl.out_b11}); exp_doubles += (size_t)p.M * p.M; }
            if (pl.out_b21 && p.U > 0) { job->exports.push_back(gauss_job::Export{i, 1, exp_doubles, p.U, p.M, p.Mld, pl.out_b21}); exp_doubles += (size_t)p.U * p.M; }
            exp_doubles = rup(exp_doubles, 32);                                // every window's block starts on a 256-byte boundary
        }
        if (exp_doubles * sizeof(double) > ((size_t)256 << 20)) { job->exports.clear(); exp_doubles = 0; }      // pinned budget: per matrix instead
    }
    const auto tb1 = std::chrono::steady_clock::now();
    // Shared measured rows: every window reads its measured SNPs from the same resident store under the same populations.
    // The job keeps ONE list of measured rows, made of CLUSTERS: a window whose rows continue a run of the current cluster
    // (the next window of a chromosome: half its measured SNPs are the previous window's) joins it at that offset -- when
    // that costs no more B11 tile pairs than tiles of its own would -- and every other window starts a new cluster on a
    // tile boundary (padding rows in between: never packed, all zero), where it costs exactly what its own tiles would.
    // So sharing can only save work: the scattered windows of a multi-GPU share keep their own tiles, two neighbours that
    // landed on the same rank share theirs (round 3 had one all-or-nothing list whose tiles started where the LIST started:
    // a share's scattered windows paid an extra row tile each and sharing was switched off for them).
    if (on_device && !streamed && job->n >= 2 && env_int("GAUSS_SHARE_MEASURED", 1) != 0) {
        const Plan& a = job->plans[0];
        bool ok = true;
        std::vector<int32_t> gl;
        std::vector<int> g0((size_t)job->n, 0);
        std::set<std::pair<int, int>> have;                     // job-wide B11 tile pairs so far
        size_t cl0 = 0, own_pairs = 0;                          // start of the current cluster in gl
        for (int i = 0; i < job->n && ok; i++) {
            const Plan& b = job->plans[i];
            ok = !b.rows_m.empty() && !b.p.ld_only && !b.p.n_gene && b.p.P <= 32 && b.h_geno_m == a.h_geno_m && b.user_ld == a.user_ld &&
                 b.p.mode == a.p.mode && b.p.P == a.p.P && b.p.geno_fmt == a.p.geno_fmt && b.p.slab16 == a.p.slab16 && b.p.Kp == a.p.Kp &&
                 b.pop_raw_off == a.pop_raw_off && b.pop_pk_off == a.pop_pk_off && b.pop_w == a.pop_w && b.seg_k0 == a.seg_k0 &&
                 b.seg_k1 == a.seg_k1 && b.seg_pop == a.seg_pop && b.run_src == a.run_src && b.run_pk_off == a.run_pk_off &&
                 b.groups == a.groups;
            if (!ok) break;
            const size_t M = b.rows_m.size(), mt = (M + TILE - 1) / TILE;
            own_pairs += mt * (mt + 1) / 2;
            // does the window continue a run of the current cluster?  (the cluster is ascending where that matters: a window
            // only joins through a binary search for its first row and an element-wise comparison of the overlap)
            size_t pos = gl.size();
            bool join = false;
            if (gl.size() > cl0 && std::is_sorted(gl.begin() + (ptrdiff_t)cl0, gl.end())) {
                const int32_t first = b.rows_m[0];
                const size_t p0 = (size_t)(std::lower_bound(gl.begin() + (ptrdiff_t)cl0, gl.end(), first) - gl.begin());
                if (p0 < gl.size() && gl[p0] == first) {
                    join = true;
                    for (size_t k = 0; k < M && join; k++) {
                        if (k > 0 && b.rows_m[k] <= b.rows_m[k - 1]) join = false;
                        else if (p0 + k < gl.size() && gl[p0 + k] != b.rows_m[k]) join = false;
                    }
                    if (join) {
                        // tile pairs the window would ADD as part of the cluster, against tiles of its own
                        const int lo = (int)(p0 / TILE), hi = (int)((p0 + M - 1) / TILE);
                        size_t add = 0;
                        for (int ti = lo; ti <= hi; ti++)
                            for (int tj = ti; tj <= hi; tj++) add += have.count(std::make_pair(ti, tj)) ? 0 : 1;
                        join = add <= mt * (mt + 1) / 2;
                        pos = p0;
                    }
                }
            }
            if (!join) {
                gl.resize(rup(gl.size(), TILE), -1);               // a new cluster on a tile boundary (padding rows: never packed)
                cl0 = pos = gl.size();
            }
            g0[(size_t)i] = (int)pos;
            for (size_t k = 0; k < M; k++)
                if (pos + k >= gl.size()) gl.push_back(b.rows_m[k]);
            const int lo = (int)(pos / TILE), hi = (int)((pos + M - 1) / TILE);
            for (int ti = lo; ti <= hi; ti++)
                for (int tj = ti; tj <= hi; tj++) have.insert(std::make_pair(ti, tj));
        }
        // worth the extra descriptor only if something IS shared (GAUSS_SHARE_MEASURED=2: always, for the tests)
        if (ok && (long long)gl.size() < (1 << 24) && (have.size() < own_pairs || env_int("GAUSS_SHARE_MEASURED", 1) == 2)) {
            job->gplan.reset(new Plan(a));
            job->g0 = g0;
            Plan& g = *job->gplan;
            g.rows_m = gl;
            g.rows_u.clear(); g.z1.clear(); g.gene_off.clear(); g.gene_out_off.clear();
            Prob& q = g.p;
            q.M = (int)gl.size(); q.U = 0; q.U_raw = 1; q.n_rhs = 0;
            q.Mp = (int)rup((size_t)q.M, TILE); q.Up = 0; q.Sp = q.Mp; q.nT = q.Mp / TILE;
            q.Mld = 0; q.nblk = 0; q.npanel = 0; q.npi = 0; q.kind = 0; q.ld_only = 0; q.n_head = q.n_predm = 0;
            g.U_user = 0; g.h_geno_u = nullptr;
            // job-wide B11 pairs: the tile pairs some window lies in
            g.pair_ti.clear(); g.pair_tj.clear(); g.pair_lut.assign((size_t)q.nT * q.nT, -1);
            for (int i = 0; i < job->n; i++) {
                const int lo = g0[(size_t)i] / TILE, hi = (g0[(size_t)i] + job->plans[i].p.M - 1) / TILE;
                for (int ti = lo; ti <= hi; ti++)
                    for (int tj = ti; tj <= hi; tj++)
                        if (g.pair_lut[(size_t)ti * q.nT + tj] < 0) {
                            g.pair_lut[(size_t)ti * q.nT + tj] = g.pair_lut[(size_t)tj * q.nT + ti] = (int)g.pair_ti.size();
                            g.pair_ti.push_back(ti); g.pair_tj.push_back(tj);
                        }
            }
            q.npair = (int)g.pair_ti.size();
            // live rows per job-wide row tile: up to the last real row in it (a cluster's last tile ends in padding rows, whose
            // dead 32-row halves the Gram kernel skips exactly as it does in a window's own last tile)
            g.tile_live.assign((size_t)q.nT, 0);
            for (size_t r = 0; r < gl.size(); r++)
                if (gl[r] >= 0) g.tile_live[r / TILE] = (int)(r % TILE) + 1;
        }
    }
    const bool shm = job->gplan != nullptr;
    const int n_prob = job->n + (shm ? 1 : 0);             // descriptors on the device: the windows, then the job-wide rows
    auto plan_of = [&](int i) -> Plan& { return i < job->n ? job->plans[i] : *job->gplan; };

    // Row lists are resolved by the pack kernel: an index beyond the store would be an out-of-bounds read on the
    // GPU.  Host stores cannot be checked (only a pointer is known), stores made by gauss_store_upload can.
    std::map<const void*, size_t> stores;
    { std::lock_guard<std::mutex> lock(ctx->mu); stores = ctx->stores; }
    for (int i = 0; i < job->n; i++) {
        const Plan& pl = job->plans[i];
        for (int side = 0; side < 2; side++) {
            const std::vector<int32_t>& rows = side ? pl.rows_u : pl.rows_m;
            const uint8_t* base = side ? pl.h_geno_u : pl.h_geno_m;
            if (rows.empty()) continue;
            long long mx = -1;
            for (int32_t r : rows) {
                if (r < 0) { return fail(GAUSS_E_INVALID, "window %d: negative row index %d", i, (int)r); }
                mx = std::max<long long>(mx, r);
            }
            if (!on_device) continue;
            // the store that contains `base` (a window may point into the middle of an uploaded store)
            auto it = stores.upper_bound(base);
            if (it == stores.begin()) continue;                      // not one of ours: caller's responsibility
            --it;
            const uint8_t* s0 = (const uint8_t*)it->first;
            if (base >= s0 + it->second) continue;
            const size_t need = (size_t)(base - s0) + (size_t)mx * (size_t)pl.user_ld + pl.row_bytes;
            if (need > it->second) {
                return fail(GAUSS_E_INVALID, "window %d: row index %lld reaches past the end of the row store (%zu bytes)", i, mx, it->second);
            }
        }
    }
    HIPCHK(hipSetDevice(ctx->device));

    // ---- table arena (host mirrored) ----
    Arena ta;
    std::vector<char>& blob = job->h_tab;
    blob.reserve((size_t)n_prob * ((size_t)320 << 10) + ((size_t)64 << 10));      // (~0.3 MB a window: grown in place, not copied over and over)
    const auto tb2 = std::chrono::steady_clock::now();
    struct TabOff { size_t raw_off, pk_off, w, wf, md, seg_pop, k0, k1, seg0, ti, tj, lut, wp, z1, goff, gout, wr, rpk, rsrc, rm, ru, ch; };
    std::vector<TabOff> to((size_t)n_prob);
    for (int i = 0; i < n_prob; i++) {
        Plan& pl = plan_of(i);
        to[i].raw_off = put(blob, ta, pl.pop_raw_off);
        to[i].pk_off = put(blob, ta, pl.pop_pk_off);
        to[i].w = put(blob, ta, pl.pop_w);
        to[i].wf = put(blob, ta, pl.pop_wf);
        to[i].md = put(blob, ta, pl.pop_md);
        to[i].seg_pop = put(blob, ta, pl.seg_pop);
        to[i].k0 = put(blob, ta, pl.seg_k0);
        to[i].k1 = put(blob, ta, pl.seg_k1);
        to[i].seg0 = put(blob, ta, pl.pop_seg0);
        to[i].ti = put(blob, ta, pl.pair_ti);
        to[i].tj = put(blob, ta, pl.pair_tj);
        to[i].lut = put(blob, ta, pl.pair_lut);
        to[i].wp = put(blob, ta, pl.word_pop);
        to[i].z1 = put(blob, ta, pl.z1);
        to[i].goff = put(blob, ta, pl.gene_off);
        to[i].gout = put(blob, ta, pl.gene_out_off);
        to[i].wr = put(blob, ta, pl.word_run);
        to[i].rpk = put(blob, ta, pl.run_pk_off);
        to[i].rsrc = put(blob, ta, pl.run_src);
        to[i].rm = put(blob, ta, pl.rows_m);
        to[i].ru = put(blob, ta, pl.rows_u);
        to[i].ch = put(blob, ta, pl.chunk_live);
    }
    // work lists
    struct ItemH { int prob, pair, group, len, b11, ord = 0; };     // b11: an item of B11 (job-wide pairs, or a window's own measured x measured pairs); ord: launch-order key (below)
    std::vector<ItemH> items;
    std::vector<char> late_window;                         // early epilogue: windows whose B21 items end the merged launch
    std::vector<int2> rowmap, tilemap, tilemap_b21, panelmap, dpanelmap, gemmmap, finmap;
    job->max_nblk = 0;
    {
        // tiles of the closing product at 128 right-hand sides each: a small job (an 8-rank share: ~570) cannot fill the
        // chip's 512 workgroup slots with them and is bound by the tiles' K loops, so it takes 64 (k_solve.hip)
        size_t t128 = 0;
        for (int i = 0; i < job->n; i++) {
            const Prob& p = job->plans[i].p;
            if (p.npanel > 0) t128 += (size_t)(p.Up128 / 128) * ((p.Mld + 127) / 128);
        }
        job->gemm_ut = t128 < (size_t)GEMM_SMALL_TILES ? 64 : 128;
    }
    job->win_tiles.assign((size_t)job->n, std::vector<int2>());
    if (shm) {
        // the job-wide measured rows are packed once, and B11's job-wide tile pairs multiplied once
        const Plan& g = *job->gplan;
        for (int pr = 0; pr < g.p.npair; pr++)
            for (size_t k = 0; k < g.groups.size(); k++)
                items.push_back(ItemH{job->n, pr, (int)k, g.seg_k1[g.groups[k].second - 1] - g.seg_k0[g.groups[k].first], 1});
        for (int r = 0; r < g.p.M; r++)
            if (g.rows_m[(size_t)r] >= 0) rowmap.push_back(make_int2(job->n, r));      // (padding rows between clusters stay zero)
    }
    // Every `fine_every`-th tile pair is cut into one work item per K segment (a population: 64 ... 3 600 samples)
    // instead of runs of >= 2048 samples: sorted by length they end up last and fill the launch's final round, in which
    // the 1 024 workgroup slots otherwise finish up to one 0.75 ms item apart.  Same segments, same slabs: same bits.
    // Measured on the bench job (Gram kernel): none 37.43 ms; every 16th / 8th / 4th / 2nd pair 37.11 / 37.15 / 37.11 /
    // 37.10; every pair 37.08 ms with twice the work items.
    const int fine_every = job->n >= 4 ? 16 : 0;
    int pair_no = 0;
    for (int i = 0; i < job->n; i++) {
        const Prob& p = job->plans[i].p;
        const int mt_i = p.Mp / TILE;
        for (int pr = 0; pr < p.npair; pr++) {
            const int b11 = job->plans[i].pair_ti[pr] < mt_i ? 1 : 0;
            if (shm && b11) continue;                                    // a B11 pair: done on the job-wide tiles
            if (fine_every > 0 && ++pair_no % fine_every == 0 && job->plans[i].groups.size() < job->plans[i].fine.size()) {
                for (size_t g = 0; g < job->plans[i].fine.size(); g++) {
                    const std::pair<int, int>& gr = job->plans[i].fine[g];
                    items.push_back(ItemH{i, pr, -1 - (int)g, job->plans[i].seg_k1[gr.second - 1] - job->plans[i].seg_k0[gr.first], b11});
                }
                continue;
            }
            for (size_t g = 0; g < job->plans[i].groups.size(); g++) {
                const std::pair<int, int>& gr = job->plans[i].groups[g];
                items.push_back(ItemH{i, pr, (int)g, job->plans[i].seg_k1[gr.second - 1] - job->plans[i].seg_k0[gr.first], b11});
            }
        }
        for (int r = shm ? p.M : 0; r < p.M + p.U; r++) rowmap.push_back(make_int2(i, r));
        if (shm) {
            // this window's view of the job-wide B11 pairs it lies in
            const Plan& g = *job->gplan;
            const int lo = job->g0[(size_t)i] / TILE, hi = (job->g0[(size_t)i] + p.M - 1) / TILE;
            for (int ti = lo; ti <= hi; ti++)
                for (int tj = ti; tj <= hi; tj++) {
                    const int2 e = make_int2(i, g.pair_lut[(size_t)ti * g.p.nT + tj] | TILE_GB11);
                    tilemap.push_back(e); job->win_tiles[(size_t)i].push_back(e);
                }
        }
        if (!p.n_gene)
            for (int pr = 0; pr < p.npair; pr++) {
                const bool b21 = !p.ld_only && job->plans[i].pair_ti[pr] >= p.Mp / TILE;      // a tile of U rows x M columns
                if (shm && !b21) continue;
                (b21 ? tilemap_b21 : tilemap).push_back(make_int2(i, pr));
                job->win_tiles[(size_t)i].push_back(make_int2(i, pr));
            }
        for (int pn = 0; pn < p.npi; pn++) panelmap.push_back(make_int2(i, pn));
        for (int pn = 0; pn < p.npanel; pn++) dpanelmap.push_back(make_int2(i, pn));
        if (p.npanel > 0) {
            job->max_nblk = std::max(job->max_nblk, p.nblk); job->max_npanel = std::max(job->max_npanel, p.npi);
            for (int up = 0; up < p.Up128 / job->gemm_ut; up++)
                for (int kb = 0; kb < (p.Mld + 127) / 128; kb++) gemmmap.push_back(make_int2(i, (up << 8) | kb));
            for (int c = 0; c < (p.n_rhs + 255) / 256; c++) finmap.push_back(make_int2(i, c));
        }
        if (p.mode != 0) job->max_pop = std::max(job->max_pop, p.P);
    }
    // The launch order: longest segments first (the tail of the launch is then made of short items), inside up to three classes --
    // B11's items, the early windows' B21 items, the late windows' (below).  One stable counting pass over (class, length): lengths
    // are whole K chunks and a job has a handful of distinct ones (three stable sorts through the plans' tables were 0.3 ms of the
    // 0.53 ms a three-window job took to plan; a rank of eight waits for exactly that before its GPU starts).
    // The order key: the item's K length scaled by how much of the tile its slowest wave multiplies (a wave issues na x nb 32 x 32
    // blocks per K step, 4 at most; the waves of a workgroup meet at a barrier every chunk, so the fullest wave sets the item's
    // pace): full tiles start first, edge tiles (a window's last 17 rows are a quarter of a tile's work) end the launch.  Measured:
    // one 529-SNP window (1.3 rounds of the chip's workgroup slots) Gram 115 -> 109 us; the 36-window job 36.62 -> 36.52 ms; an
    // 8-rank share unchanged.  The order changes no bit (items are independent).
    {
        auto live_rows = [&](const Plan& pl, int t) {
            const Prob& p = pl.p;
            const int mt = p.Mp / TILE;
            if (!pl.tile_live.empty()) return pl.tile_live[(size_t)t];
            const int left = (t < mt) ? p.M - t * TILE : p.U - (t - mt) * TILE;
            return left > TILE ? TILE : left;
        };
        auto halves = [](int r, int w) { const int n = (r - w * 64 + 31) / 32; return n < 0 ? 0 : (n > 2 ? 2 : n); };
        for (ItemH& h : items) {
            h.ord = h.len;
            if (job->gram_i8) continue;          // (the int8 kernel is bound by operand delivery: it keeps items of one K range together -- 5.40 against 5.53 ms)
            const Plan& pl = plan_of(h.prob);
            const int ra = live_rows(pl, pl.pair_ti[h.pair]), rb = live_rows(pl, pl.pair_tj[h.pair]);
            int mx = 0;
            for (int wr = 0; wr < 2; wr++)
                for (int wc = 0; wc < 2; wc++) mx = std::max(mx, halves(ra, wr) * halves(rb, wc));
            h.ord = std::max(KC, (h.len * std::max(mx, 1) / 4 + KC - 1) / KC * KC);
        }
    }
    auto is_b11 = [&](const ItemH& h) { return h.b11 != 0; };
    std::vector<uint8_t> item_class(items.size(), 0);
    auto order_items = [&]() {
        int L = 0;
        bool whole = true;
        for (const ItemH& h : items) { L = std::max(L, h.ord / KC); whole = whole && h.ord % KC == 0 && h.ord >= 0; }
        if (!whole || (size_t)L > 4 * items.size() + 1024) {            // (never on a plan of this file: lengths are multiples of KC)
            std::vector<size_t> idx(items.size());
            for (size_t n = 0; n < idx.size(); n++) idx[n] = n;
            std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
                return item_class[a] != item_class[b] ? item_class[a] < item_class[b] : items[a].ord > items[b].ord; });
            std::vector<ItemH> out(items.size());
            for (size_t n = 0; n < idx.size(); n++) out[n] = items[idx[n]];
            items.swap(out);
            return;
        }
        const size_t nb = (size_t)3 * (size_t)(L + 1);
        std::vector<uint32_t> start(nb + 1, 0u);
        auto bucket = [&](size_t n) { return (size_t)item_class[n] * (size_t)(L + 1) + (size_t)(L - items[n].ord / KC); };
        for (size_t n = 0; n < items.size(); n++) start[bucket(n) + 1]++;
        for (size_t b = 0; b < nb; b++) start[b + 1] += start[b];
        std::vector<ItemH> out(items.size());
        for (size_t n = 0; n < items.size(); n++) out[start[bucket(n)]++] = items[n];
        items.swap(out);
    };
    {
        // Chain beside the Gram kernel (k_solve_lite.hip): B11's items become a launch of their own, B11's epilogue tiles and
        // the factorisation chain follow it on the chain queue, and the chain's latency hides under the Gram launch of
        // B21's items.  Worth it when that launch is long enough to cover B11's small-footprint epilogue tiles (~0.5 ms) and
        // the chain, which runs ~3 x slower beside the Gram kernel than alone (~100 us per block step: two launches);
        // GAUSS_CHAIN_ASIDE = 0 never, 2 always (tests), 1 (default) by this estimate.
        const int mode = env_int("GAUSS_CHAIN_ASIDE", 1);
        bool genes = false;
        double b21_len = 0.0;
        for (const ItemH& h : items) if (!is_b11(h)) b21_len += (double)h.len;
        for (int i = 0; i < job->n; i++) genes = genes || job->plans[i].p.n_gene > 0;
        // (the int8 Gram kernel is ~8 x faster: an 8-rank share's B21 launch, 0.5 ms, no longer covers its chain)
        const double t_b21 = b21_len * 2.0 * TILE * TILE / (job->gram_i8 ? 960e12 : 120e12), t_chain = 0.5e-3 + 100e-6 * job->max_nblk;
        {
            double all_len = 0.0;
            for (const ItemH& h : items) all_len += (double)h.len;
            job->wait_bound_us = 50.0 * 1e6 * all_len * 2.0 * TILE * TILE / (job->gram_i8 ? 960e12 : 120e12);      // a whole-genome job must not trip a fixed bound
        }
        job->chain_aside = !streamed && mode != 0 && !panelmap.empty() && !tilemap_b21.empty() && !genes && job->ctx->chain &&
                           job->ctx->side && (mode == 2 || t_b21 >= 1.2 * t_chain);
        if (job->chain_aside)
            for (size_t n = 0; n < items.size(); n++) {
                item_class[n] = is_b11(items[n]) ? 0 : 1;
                job->n_items_b11 += is_b11(items[n]) ? 1 : 0;
            }
        // Merged launch (round 4): B11's items and B21's items are ONE launch, B11's first; they count themselves off and the
        // chain queue starts when the count is complete (k_gram.hip: wait_count_kernel) -- the chip is never drained between
        // the two halves (two launches: 37.1 ms, one: 36.6 on the 36-window job).  GAUSS_CHAIN_MERGED=0: two launches + event.
        // (f32 only: the int8 Gram kernel is operand-delivery bound and the chain's memory traffic beside ALL of it costs more than
        // the second launch's start-up -- measured 8.70 ms per step merged against 8.35 ms as two launches; =2 forces it for both)
        {
            const int mm = env_int("GAUSS_CHAIN_MERGED", 1);
            job->merged = job->chain_aside && mm != 0 && (!job->gram_i8 || mm == 2);
            // =2 (tests): merged whatever the queue registry says -- job_queue_run otherwise takes the two-launch form for a run
            // whose context cannot be sure of a hardware queue per stream (gauss_ctx.cpp)
            job->force_merged = job->merged && mm == 2;
        }
        // Early epilogue: the smallest windows of the job that together hold about a third of B21's Gram work are "late" -- their
        // items end the launch -- and every other window is "early" (GAUSS_EPI_EARLY=0: off.  Measured by the late share,
        // 36-window step / slowest 8-rank share: off 40.04 / 5.45 ms; 6 % 39.89 / 5.51; 12 % 39.76 / 5.51; 25 % 39.75 / 5.45; 35 %
        // 39.67 / 5.40; 50 % 39.64 / 5.47 -- a late part that is too small opens the gate only when the launch is all but over and
        // the two epilogue launches cost their event hops).  One window: nothing to split.
        late_window.assign((size_t)job->n + 1, 0);
        if (job->merged && job->n >= 2 && job->ctx->side && env_int("GAUSS_EPI_EARLY", 1) != 0) {
            std::vector<double> w21((size_t)job->n, 0.0);
            double tot = 0;
            for (const ItemH& h : items) if (!is_b11(h)) { w21[(size_t)h.prob] += h.len; tot += h.len; }
            const double want = tot * 0.35;
            double acc = 0;
            int n_late = 0;
            // (the windows with the least B21 work: a share of four windows gives up its smallest one, not whichever comes last)
            std::vector<int> by_work((size_t)job->n);
            for (int i = 0; i < job->n; i++) by_work[(size_t)i] = i;
            std::stable_sort(by_work.begin(), by_work.end(), [&](int a, int b) { return w21[(size_t)a] < w21[(size_t)b]; });
            for (int k = 0; k + 1 < job->n && acc < want; k++) { late_window[(size_t)by_work[(size_t)k]] = 1; acc += w21[(size_t)by_work[(size_t)k]]; n_late++; }
            if (n_late > 0 && acc < tot) {
                for (size_t n = 0; n < items.size(); n++) {
                    if (is_b11(items[n])) continue;
                    if (late_window[(size_t)items[n].prob]) item_class[n] = 2;
                    else job->n_items_b21_early++;
                }
            } else late_window.assign((size_t)job->n + 1, 0);
        }
        order_items();
    }
    std::vector<int> sgroup_of_item;
    if (streamed) {
        // streamed window: B11's items (measured rows only) first, then B21's items by chunk of `ct` unmeasured row
        // tiles; each group is one Gram launch that starts as soon as its rows have landed
        const Plan& pl0 = job->plans[0];
        const int mt = pl0.p.Mp / TILE;
        const std::vector<int>& tile_group = stream->tile_group;
        const std::vector<int>& first_tile = stream->first_tile;
        const int ngrp = (int)first_tile.size() - 1;
        auto grp = [&](const ItemH& h) { const int ti = pl0.pair_ti[h.pair]; return ti < mt ? 0 : tile_group[(size_t)(ti - mt)]; };
        std::stable_sort(items.begin(), items.end(), [&](const ItemH& a, const ItemH& b) { return grp(a) < grp(b); });
        job->sgroups.assign((size_t)ngrp, gauss_job::StreamGroup{0, 0, 0, 0, 0, 0});
        for (size_t n = 0; n < items.size(); n++) {
            gauss_job::StreamGroup& g = job->sgroups[(size_t)grp(items[n])];
            if (g.n_items == 0) g.item0 = (int)n;
            g.n_items++;
        }
        job->sgroups[0].row0 = 0; job->sgroups[0].n_rows = pl0.p.M;
        for (int g = 1; g < ngrp; g++) {
            const int u0 = first_tile[(size_t)g] * TILE, u1 = std::min(pl0.p.U, first_tile[(size_t)g + 1] * TILE);
            job->sgroups[g].row0 = pl0.p.M + u0; job->sgroups[g].n_rows = std::max(0, u1 - u0);
            // B21's epilogue tiles are listed in pair order (row tile, then column tile): a chunk's tiles are contiguous
            job->sgroups[g].tile0 = first_tile[(size_t)g] * mt;
            job->sgroups[g].n_tiles = (first_tile[(size_t)g + 1] - first_tile[(size_t)g]) * mt;
        }
    }
    // XCD-aware launch order.  Workgroup b runs on XCD b % 8 (each XCD has its own 4 MiB L2).  Neighbours in
    // the sorted list share operand tiles (same window, same K range, adjacent tile pairs), so the list is
    // cut into super-blocks of xcd_block items and super-block j is queued on XCD j % 8: the items that are
    // resident together on one XCD then read the same tiles at about the same K position.  Measured on the
    // bench workload (rocprofv3 FETCH_SIZE): 22.0 GB -> 16.0 GB per launch at 36, same kernel time; larger
    // blocks start to cost time (load balance).
    {
        // Small jobs (the 4-5 windows an 8-rank run leaves per GPU, ~7 000 items) take blocks of 8: a block of 36 is
        // 4 % of an XCD's share there, and the XCD that gets one more than the others finishes last (Gram kernel of
        // the 8-rank shares 5.10 -> 5.03 ms; 36 windows: 38.5 ms either way, but 36 reads 27 % less from the fabric).
        const int xcd_block = items.size() >= 20000 ? 36 : 8;
        // (a job whose B11 items are a launch of their own interleaves each launch's list by itself)
        auto interleave = [&](size_t i0, size_t i1) {
            if (xcd_block <= 0 || i1 - i0 <= (size_t)8 * xcd_block) return;
            std::vector<std::vector<ItemH>> q(8);
            for (size_t i = i0; i < i1; i++) q[((i - i0) / xcd_block) % 8].push_back(items[i]);
            size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n = i0;
            while (n < i1)
                for (int x = 0; x < 8; x++)
                    if (pos[x] < q[x].size()) items[n++] = q[x][pos[x]++];
        };
        if (!streamed) {
            interleave(0, (size_t)job->n_items_b11);
            if (job->n_items_b21_early > 0) {
                interleave((size_t)job->n_items_b11, (size_t)(job->n_items_b11 + job->n_items_b21_early));
                interleave((size_t)(job->n_items_b11 + job->n_items_b21_early), items.size());
            } else interleave((size_t)job->n_items_b11, items.size());
        }
    }
    const size_t o_items = ta.take(sizeof(Item) * std::max<size_t>(items.size(), 1));
    const size_t o_rowmap = put(blob, ta, rowmap);
    job->n_tiles_b11 = (int)tilemap.size();
    if (job->n_items_b21_early > 0) {
        std::stable_partition(tilemap_b21.begin(), tilemap_b21.end(), [&](const int2& a) { return !late_window[(size_t)a.x]; });
        for (const int2& t : tilemap_b21) job->n_tiles_b21_early += late_window[(size_t)t.x] ? 0 : 1;
    }
    tilemap.insert(tilemap.end(), tilemap_b21.begin(), tilemap_b21.end());
    const size_t o_tilemap = put(blob, ta, tilemap);
    // the product's tiles with the longest K loop (highest k block) first
    std::stable_sort(gemmmap.begin(), gemmmap.end(), [](const int2& a, const int2& b) { return (a.y & 255) > (b.y & 255); });
    const size_t o_panelmap = put(blob, ta, panelmap);
    const size_t o_dpanelmap = put(blob, ta, dpanelmap);
    const size_t o_gemmmap = put(blob, ta, gemmmap);
    const size_t o_finmap = put(blob, ta, finmap);
    const size_t o_probs = ta.take(sizeof(Prob) * (size_t)n_prob);
    const size_t o_exports = ta.take(sizeof(ExportD) * std::max<size_t>(job->exports.size(), 1));
    blob.resize(ta.off);
    job->n_items = (int)items.size();
    job->n_rows = (int)rowmap.size();
    job->n_tiles = (int)tilemap.size();
    job->n_panels = (int)panelmap.size();
    job->n_dpanels = (int)dpanelmap.size();
    job->n_gemm = (int)gemmmap.size();
    job->n_fin = (int)finmap.size();

    // ---- workspace arena ----
    Arena wa;         // zeroed once per job: operand padding, B21 padding and the solve matrices rely on it
    Arena wslab;      // partial slabs: every entry a reader keeps is written by the Gram kernel first, so no zeroing
    struct WsOff { size_t raw_m, raw_u, packed, sx, sxx, slab, sd, wm, mu, wmu, A, Linv, B21, V, ld, b11c, gsum, part; long long ldraw; };
    std::vector<WsOff> wo(job->n);
    size_t res = 0;
    {
        // The early products of a row of the inverse (k_solve.hip, ride_pre) are a chain of dependent tile products:
        // rows with at least `solve_split` of them give one workgroup to each of their SOLVE_SPLIT classes, shorter rows
        // run the classes in one workgroup; 0 = never cut.  Either form sums in the same order.  Cutting from two
        // products on is what keeps every riding workgroup shorter than the diagonal tile's (36 windows, factorisation
        // with riding rows: never 1.72 ms, >= 8 1.70, >= 4 1.49, >= 2 1.46 before the pre / fin form, 1.31 with it; factorisation alone 1.10).
        // few windows: every launch of the factorisation is a latency-bound link of a chain -- drop the panel launches
        // (k_solve.hip, factor_update_kernel own_panel); same bits either way.  Factorisation + riding rows, with /
        // without panel launches: 5 windows 0.65 / 0.58 ms, 9 windows 0.82 / 0.78, 18 windows 0.87 / 0.84, 36 windows
        // 1.31 / 1.42 (the repeated panel products start to cost workgroup slots)
        int n_solve = 0;
        for (int i = 0; i < job->n; i++) n_solve += job->plans[i].p.npanel > 0 ? 1 : 0;
        job->own_panel = (n_solve > 0 && n_solve <= OWN_PANEL_MAX_WINDOWS) ? 1 : 0;
        job->solve_split = job->n_panels > 0 ? SOLVE_SPLIT_MIN : 0;
    }
    for (int i = 0; i < job->n; i++) {
        Plan& pl = job->plans[i];
        Prob& p = pl.p;
        WsOff& w = wo[i];
        if (streamed) {
            w.ldraw = stream->ldraw;                               // the rows land in the context's landing buffer
        } else if (!on_device) {
            // contiguous host matrices whose stride is close to the row length keep their stride on the device:
            // the upload is then ONE linear copy (a pitched copy of 3 000 rows runs at a fraction of that rate)
            const bool linear = pl.rows_m.empty() && pl.rows_u.empty() && (size_t)pl.user_ld <= pl.row_bytes + pl.row_bytes / 8 + 64;
            w.ldraw = linear ? pl.user_ld : (long long)rup(pl.row_bytes, 16);
            w.raw_m = wa.take((size_t)p.M * w.ldraw + 64);
            w.raw_u = wa.take((size_t)std::max(pl.U_user, 1) * w.ldraw + 64);
        } else {
            w.ldraw = pl.user_ld;
        }
        w.packed = wa.take((size_t)p.Sp * p.Kp);
        w.sx = wa.take((size_t)p.Sp * p.P * sizeof(int));
        w.sxx = wa.take((size_t)p.Sp * p.P * sizeof(int));
        w.slab = wslab.take((size_t)p.npair * p.nseg * TILE * TILE * (p.slab16 ? sizeof(uint16_t) : sizeof(float)));
        w.sd = wa.take((size_t)p.Sp * sizeof(double));
        w.wm = wa.take((size_t)p.Sp * sizeof(double));
        w.mu = wa.take((size_t)p.Sp * p.P * sizeof(double));
        w.wmu = wa.take((size_t)p.Sp * p.P * sizeof(double));
        if (!p.ld_only) {
            // the LD epilogue writes B11 (and its shifted twin) and B21 for every window that is not a plain
            // gauss_ld / gene batch; the factor and solve scratch only exists when there is something to solve
            w.A = wa.take((size_t)(p.npanel > 0 ? 5 : 2) * p.Mld * p.Mld * sizeof(double));   // A0 A1 [L0 L1 W0]
            w.B21 = wa.take((size_t)std::max(p.U, 1) * p.Mld * sizeof(double));
        }
        if (p.npanel > 0) {
            w.Linv = wa.take((size_t)2 * p.nblk * NB * NB * sizeof(double));
            w.V = wa.take((size_t)std::max(p.npanel, p.npi) * p.Mld * NR * sizeof(double));
            w.gsum = wa.take((size_t)((p.Mld + 127) / 128) * p.Up128 * 3 * sizeof(double));
            w.part = wa.take((size_t)2 * p.npi * 4 * NB * NR * sizeof(double));      // double buffered by row parity
            w.b11c = wa.take((size_t)p.Mld * p.Mld * sizeof(double));
        }
        w.ld = wa.take(std::max<size_t>(pl.out_ld_count, 1) * sizeof(double));
        pl.res_off = res;
        res += 2 * (size_t)p.n_rhs;
    }
    // job-wide measured rows (shared measured rows): one more tile of rows than Mp, because a window's last row tile
    // starts wherever the window starts and may reach past the chromosome's last measured SNP (zero rows there)
    struct GOff { size_t packed, sx, sxx, sd, wm, mu, wmu, slab; } go = {0, 0, 0, 0, 0, 0, 0, 0};
    if (shm) {
        const Prob& q = job->gplan->p;
        const size_t rows = (size_t)q.Mp + TILE;
        go.packed = wa.take(rows * q.Kp);
        go.sx = wa.take(rows * q.P * sizeof(int));
        go.sxx = wa.take(rows * q.P * sizeof(int));
        go.sd = wa.take(rows * sizeof(double));
        go.wm = wa.take(rows * sizeof(double));
        go.mu = wa.take(rows * q.P * sizeof(double));
        go.wmu = wa.take(rows * q.P * sizeof(double));
        go.slab = wslab.take((size_t)q.npair * q.nseg * TILE * TILE * (q.slab16 ? sizeof(uint16_t) : sizeof(float)));
    }
    const size_t o_status = wa.take(sizeof(int) * (4 * job->n + 4));    // [n][4], then 4 job-wide ints ([4 n]: the chain queue timed out)
    const size_t o_count = wa.take(128);                                 // counters of the merged Gram launch, a cache line each: [0] B11's items, [8] the early windows' B21 items (zeroed once; they only grow)
    const size_t o_results = wa.take(sizeof(double) * std::max<size_t>(res, 1));
    job->n_results = res;
    const size_t slab_base = rup(wa.off, 4096);
    for (WsOff& w : wo) w.slab += slab_base;
    go.slab += slab_base;
    job->ws_bytes = slab_base + wslab.off;
    job->tab_bytes = rup(blob.size(), 16);                               // (copied in 16-byte words, below)

    const bool job_trace = trace_on("job");
    const auto tj0 = std::chrono::steady_clock::now();
    hipError_t e = ctx_dev_alloc(ctx, job->ws_bytes, (void**)&job->d_ws);
    if (e != hipSuccess) { job->d_ws = nullptr; return fail(GAUSS_E_NOMEM, "hipMalloc(%zu bytes workspace) failed: %s", wa.off, hipGetErrorString(e)); }
    e = ctx_dev_alloc(ctx, job->tab_bytes, (void**)&job->d_tab);
    if (e != hipSuccess) { job->d_tab = nullptr; return fail(GAUSS_E_NOMEM, "hipMalloc(%zu bytes tables) failed", blob.size()); }
    // one pinned block for the table image and the result mirrors: the table upload is then a true asynchronous
    // DMA and a job over resident rows is created without waiting for the stream (another job may be running on it)
    const size_t pin_tab = rup(job->tab_bytes, 256), pin_res = rup(sizeof(double) * std::max<size_t>(res, 1), 256);
    const size_t pin_st = rup(sizeof(int) * (4 * job->n + 4), 256);
    const size_t pin_exp = rup(sizeof(double) * exp_doubles, 256);
    e = ctx_pin_alloc(ctx, pin_tab + 2 * (pin_res + pin_st) + pin_exp, (void**)&job->h_pin);
    if (e != hipSuccess) { job->h_pin = nullptr; return fail(GAUSS_E_NOMEM, "hipHostMalloc(%zu bytes) failed", pin_tab + 2 * pin_res); }
    for (int k = 0; k < 2; k++) {
        job->h_res2[k] = (double*)(job->h_pin + pin_tab + k * (pin_res + pin_st));
        job->h_st2[k] = (int*)(job->h_pin + pin_tab + k * (pin_res + pin_st) + pin_res);
        HIPCHK(hipEventCreate(&job->done2[k]));
    }
    job->h_results = job->h_res2[0];
    job->h_status = job->h_st2[0];
    job->done = job->done2[0];
    if (!job->exports.empty()) {
        job->h_export = (double*)(job->h_pin + pin_tab + 2 * (pin_res + pin_st));
        // chunks of whole exports, >= 2 MB each (at most ~32): one kernel launch and one event per chunk
        const size_t per = std::max<size_t>((size_t)1 << 18, (exp_doubles + 31) / 32);       // doubles
        size_t acc = 0;
        int first = 0;
        const int nx = (int)job->exports.size();
        for (int x = 0; x < nx; x++) {
            acc += (size_t)job->exports[(size_t)x].rows * job->exports[(size_t)x].width;
            if (acc >= per || x + 1 == nx) { job->exp_chunks.push_back(std::make_pair(first, x + 1)); first = x + 1; acc = 0; }
        }
        job->exp_ev.assign(job->exp_chunks.size(), nullptr);
        for (hipEvent_t& ev : job->exp_ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const auto tj1 = std::chrono::steady_clock::now();
    HIPCHK(hipEventCreate(&job->begin));
    for (int k = 0; k < 2; k++)
        for (hipEvent_t* e : {&job->rev[k].gram, &job->rev[k].side, &job->rev[k].pack, &job->rev[k].rows, &job->rev[k].epi})
            HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));

    hipStream_t st = ctx->stream;
    // zero once: operand padding, B21 padding and the solve matrices rely on it.  On the (otherwise idle) upload queue, not on the
    // main queue: a pipeline creates the job of batch b + 1 while batch b computes, and 1-2 GB of zeroes at the head of the next
    // batch were a 0.3-0.4 ms gap between the batches of a chromosome (gauss_host_impute_chromosome: 36.4 ms of GPU span for
    // 35.0 ms of batches); beside the previous batch's Gram launch they cost nothing.  The main queue waits for the event IN ORDER,
    // i.e. behind whatever it is computing now.  While a background upload is using that queue the zeroes stay where they were.
    hipStream_t zs = st;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        if (ctx->upload && ctx->uploads.empty()) zs = ctx->upload;
    }
    job->queue_touched = true;                 // from here on job_release must let the queues drain before the blocks go back
    job->zero_queue = zs;
    HIPCHK(hipMemsetAsync(job->d_ws, 0, slab_base, zs));
    if (zs != st) {
        HIPCHK(hipEventCreateWithFlags(&job->zeroed, hipEventDisableTiming));
        HIPCHK(hipEventRecord(job->zeroed, zs));
        HIPCHK(hipStreamWaitEvent(st, job->zeroed, 0));
    }
    if (streamed) {
        job->sevp.resize(job->sgroups.size(), nullptr);
        for (hipEvent_t& e : job->sevp) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }

    job->d_status = (int*)(job->d_ws + o_status);
    job->d_b11_done = (unsigned long long*)(job->d_ws + o_count);
    job->d_results = (double*)(job->d_ws + o_results);
    std::deque<std::vector<uint8_t>> stage;      // host gather buffers, alive until the copies have drained
    for (int i = 0; i < job->n; i++) {
        Plan& pl = job->plans[i];
        Prob& p = pl.p;
        const WsOff& w = wo[i];
        char* T = job->d_tab;
        char* W = job->d_ws;
        p.ld_raw = w.ldraw;
        if (streamed) { p.raw_m = stream->d_m; p.raw_u = stream->d_u; }
        else if (on_device) { p.raw_m = pl.h_geno_m; p.raw_u = pl.h_geno_u; }
        else { p.raw_m = (const uint8_t*)(W + w.raw_m); p.raw_u = (const uint8_t*)(W + w.raw_u); }
        p.packed = (uint8_t*)(W + w.packed);
        p.sx = (int*)(W + w.sx); p.sxx = (int*)(W + w.sxx);
        p.pop_raw_off = (const int*)(T + to[i].raw_off);
        p.pop_pk_off = (const int*)(T + to[i].pk_off);
        p.pop_w = (const double*)(T + to[i].w);
        p.pop_wf = (const double*)(T + to[i].wf);
        p.pop_md = (const double*)(T + to[i].md);
        p.seg_pop = (const int*)(T + to[i].seg_pop);
        p.seg_k0 = (const int*)(T + to[i].k0);
        p.seg_k1 = (const int*)(T + to[i].k1);
        p.pop_seg0 = (const int*)(T + to[i].seg0);
        p.pair_ti = (const int*)(T + to[i].ti);
        p.pair_tj = (const int*)(T + to[i].tj);
        p.pair_lut = (const int*)(T + to[i].lut);
        p.word_pop = (const uint8_t*)(T + to[i].wp);
        p.word_run = (const uint8_t*)(T + to[i].wr);
        // row lists are resolved on the device only for a resident store; host rows are gathered while staging
        p.rows_m = (on_device && !pl.rows_m.empty()) ? (const int*)(T + to[i].rm) : nullptr;
        p.rows_u = (on_device && !pl.rows_u.empty()) ? (const int*)(T + to[i].ru) : nullptr;
        p.run_pk_off = (const int*)(T + to[i].rpk);
        p.run_src = (const int*)(T + to[i].rsrc);
        p.slab = (float*)(W + w.slab);
        p.rt_sd = (double*)(W + w.sd); p.rt_wm = (double*)(W + w.wm);
        p.rt_mu = (double*)(W + w.mu); p.rt_wmu = (double*)(W + w.wmu);
        p.z1 = (const double*)(T + to[i].z1);
        if (!p.ld_only) { p.A = (double*)(W + w.A); p.B21 = (double*)(W + w.B21); }
        if (p.npanel > 0) {
            p.Linv = (double*)(W + w.Linv); p.V = (double*)(W + w.V); p.Gsum = (double*)(W + w.gsum);
            p.Part = (double*)(W + w.part);
            pl.d_b11_copy = (double*)(W + w.b11c);
        }
        p.out_z = job->d_results + pl.res_off;
        p.out_info = job->d_results + pl.res_off + p.n_rhs;
        p.status = job->d_status + 4 * i;
        p.out_ld = (double*)(W + w.ld);
        p.gene_off = p.n_gene ? (const int*)(T + to[i].goff) : nullptr;
        p.gene_out_off = p.n_gene ? (long long*)(T + to[i].gout) : nullptr;
        // the unmeasured part of every row array follows the measured part ...
        p.packed_u = p.packed + (size_t)p.Mp * p.Kp;
        p.sx_u = p.sx + (size_t)p.Mp * p.P; p.sxx_u = p.sxx + (size_t)p.Mp * p.P;
        p.rt_sd_u = p.rt_sd + p.Mp; p.rt_wm_u = p.rt_wm + p.Mp;
        p.rt_mu_u = p.rt_mu + (size_t)p.Mp * p.P; p.rt_wmu_u = p.rt_wmu + (size_t)p.Mp * p.P;
        p.g0 = 0; p.n_gpair = 0; p.gpair_ti = p.pair_ti; p.gpair_tj = p.pair_tj; p.slab_g = p.slab;
        if (shm) {
            // ... unless the measured rows are the job-wide ones: this window's run starts at g0
            const Prob& q = job->gplan->p;
            const size_t g0 = (size_t)job->g0[(size_t)i];
            p.g0 = (int)g0; p.n_gpair = q.npair;
            p.packed = (uint8_t*)(W + go.packed) + g0 * q.Kp;
            p.sx = (int*)(W + go.sx) + g0 * q.P; p.sxx = (int*)(W + go.sxx) + g0 * q.P;
            p.rt_sd = (double*)(W + go.sd) + g0; p.rt_wm = (double*)(W + go.wm) + g0;
            p.rt_mu = (double*)(W + go.mu) + g0 * q.P; p.rt_wmu = (double*)(W + go.wmu) + g0 * q.P;
            p.gpair_ti = (const int*)(T + to[(size_t)job->n].ti); p.gpair_tj = (const int*)(T + to[(size_t)job->n].tj);
            p.slab_g = (float*)(W + go.slab);
        }
        memcpy(blob.data() + o_probs + sizeof(Prob) * i, &p, sizeof(Prob));
        if (!on_device && !streamed) {
            auto upload = [&](size_t dst_off, const uint8_t* src, const std::vector<int32_t>& rows, int nrows) -> int {
                if (nrows <= 0) return GAUSS_OK;
                if (rows.empty()) {
                    if (w.ldraw == pl.user_ld)      // same stride: one linear copy (the last row stops at its data)
                        HIPCHK(hipMemcpyAsync(W + dst_off, src, (size_t)(nrows - 1) * pl.user_ld + pl.row_bytes,
                                              hipMemcpyHostToDevice, st));
                    else
                        HIPCHK(hipMemcpy2DAsync(W + dst_off, (size_t)w.ldraw, src, (size_t)pl.user_ld, pl.row_bytes,
                                                (size_t)nrows, hipMemcpyHostToDevice, st));
                    return GAUSS_OK;
                }
                stage.emplace_back((size_t)nrows * w.ldraw);               // gather the listed store rows
                std::vector<uint8_t>& buf = stage.back();
                for (int r = 0; r < nrows; r++)
                    memcpy(buf.data() + (size_t)r * w.ldraw, src + (size_t)rows[r] * pl.user_ld, pl.row_bytes);
                HIPCHK(hipMemcpyAsync(W + dst_off, buf.data(), buf.size(), hipMemcpyHostToDevice, st));
                return GAUSS_OK;
            };
            int rc = upload(w.raw_m, pl.h_geno_m, pl.rows_m, p.M);
            if (!rc) rc = upload(w.raw_u, pl.h_geno_u, pl.rows_u, pl.U_user);
            if (rc) return rc;
        }
    }
    if (shm) {
        // descriptor n: the job-wide measured rows (pack_stats / row_stats work on it; nothing else is launched for it)
        Plan& g = *job->gplan;
        Prob& q = g.p;
        const size_t n = (size_t)job->n;
        char* T = job->d_tab;
        char* W = job->d_ws;
        q.ld_raw = g.user_ld;
        q.raw_m = g.h_geno_m; q.raw_u = nullptr;
        q.packed = (uint8_t*)(W + go.packed); q.packed_u = q.packed;
        q.sx = (int*)(W + go.sx); q.sxx = (int*)(W + go.sxx); q.sx_u = q.sx; q.sxx_u = q.sxx;
        q.rt_sd = (double*)(W + go.sd); q.rt_wm = (double*)(W + go.wm); q.rt_sd_u = q.rt_sd; q.rt_wm_u = q.rt_wm;
        q.rt_mu = (double*)(W + go.mu); q.rt_wmu = (double*)(W + go.wmu); q.rt_mu_u = q.rt_mu; q.rt_wmu_u = q.rt_wmu;
        q.pop_raw_off = (const int*)(T + to[n].raw_off); q.pop_pk_off = (const int*)(T + to[n].pk_off);
        q.pop_w = (const double*)(T + to[n].w); q.pop_wf = (const double*)(T + to[n].wf); q.pop_md = (const double*)(T + to[n].md);
        q.seg_pop = (const int*)(T + to[n].seg_pop); q.seg_k0 = (const int*)(T + to[n].k0); q.seg_k1 = (const int*)(T + to[n].k1);
        q.pop_seg0 = (const int*)(T + to[n].seg0);
        q.pair_ti = (const int*)(T + to[n].ti); q.pair_tj = (const int*)(T + to[n].tj); q.pair_lut = (const int*)(T + to[n].lut);
        q.word_pop = (const uint8_t*)(T + to[n].wp); q.word_run = (const uint8_t*)(T + to[n].wr);
        q.rows_m = (const int*)(T + to[n].rm); q.rows_u = nullptr;
        q.run_pk_off = (const int*)(T + to[n].rpk); q.run_src = (const int*)(T + to[n].rsrc);
        q.slab = (float*)(W + go.slab); q.slab_g = q.slab; q.gpair_ti = q.pair_ti; q.gpair_tj = q.pair_tj; q.g0 = 0; q.n_gpair = q.npair;
        q.z1 = nullptr; q.A = nullptr; q.B21 = nullptr; q.Linv = nullptr; q.V = nullptr; q.Gsum = nullptr; q.Part = nullptr;
        q.out_z = q.out_info = nullptr; q.out_ld = nullptr; q.status = job->d_status;      // never written for this descriptor
        q.gene_off = nullptr; q.gene_out_off = nullptr; q.n_gene = 0;
        memcpy(blob.data() + o_probs + sizeof(Prob) * n, &q, sizeof(Prob));
    }
    // device work items: every pointer is resolved here so the kernel starts loading operands at once
    for (size_t n = 0; n < items.size(); n++) {
        const ItemH& h = items[n];
        const Plan& pl = plan_of(h.prob);
        const Prob& p = pl.p;
        const std::pair<int, int>& gr = h.group >= 0 ? pl.groups[h.group] : pl.fine[(size_t)(-1 - h.group)];
        const int ti = pl.pair_ti[h.pair], tj = pl.pair_tj[h.pair];
        const int mt = p.Mp / TILE;
        auto rows = [&](int t) {
            if (!pl.tile_live.empty()) return pl.tile_live[(size_t)t];
            int left = (t < mt) ? p.M - t * TILE : p.U - (t - mt) * TILE; return left > TILE ? TILE : left;
        };
        // a row tile of the measured part (for a window that shares its measured rows: inside the job-wide array, from
        // wherever the window starts) or of the unmeasured part
        auto tile_rows = [&](int t) { return t < mt ? p.packed + (size_t)t * TILE * p.Kp : p.packed_u + (size_t)(t - mt) * TILE * p.Kp; };
        Item it;
        it.a = tile_rows(ti);
        it.b = tile_rows(tj);
        it.slab = p.slab + ((size_t)h.pair * p.nseg + gr.first) * (p.slab16 ? TILE * TILE / 2 : TILE * TILE);
        it.seg_k1 = p.seg_k1 + gr.first;
        it.chunk_live = (const uint32_t*)(job->d_tab + to[h.prob].ch);
        it.Kp = p.Kp; it.k0 = pl.seg_k0[gr.first]; it.nseg = gr.second - gr.first;
        it.rows_a = rows(ti); it.rows_b = rows(tj); it.flags = (ti == tj ? 1 : 0) | (p.slab16 ? 2 : 0);
        if (job->merged && (int)n < job->n_items_b11) it.flags |= 16;         // counts itself off in b11_done[0]
        else if (job->merged && (int)n < job->n_items_b11 + job->n_items_b21_early) it.flags |= 32;      // ... in b11_done[8] (the early windows' B21 items)
        memcpy(blob.data() + o_items + sizeof(Item) * n, &it, sizeof(Item));
    }
    for (size_t x = 0; x < job->exports.size(); x++) {
        const gauss_job::Export& ex = job->exports[x];
        const Plan& pl = job->plans[(size_t)ex.plan];
        ExportD d;
        d.src = ex.which ? pl.p.B21 : (pl.p.kind == GAUSS_WIN_LD ? pl.p.A : pl.d_b11_copy);
        d.dst = job->h_export + ex.off;
        d.rows = ex.rows; d.width = ex.width; d.pitch = ex.pitch; d.pad_ = 0;
        memcpy(blob.data() + o_exports + sizeof(ExportD) * x, &d, sizeof(ExportD));
    }
    const auto tj2 = std::chrono::steady_clock::now();
    memcpy(job->h_pin, blob.data(), blob.size());
    // The table image crosses PCIe by kernel, not by hipMemcpyAsync: while a background upload keeps the DMA engines busy
    // (gauss_store_upload_async: a chromosome's first call) the runtime made the CALLER wait for an engine -- one job creation in
    // five stalled for 11-16 ms on these few MB (GAUSS_JOB_TRACE=1, round 4), the GPU idle meanwhile.
    launch_h2d_copy(job->d_tab, job->h_pin, job->tab_bytes, st);
    HIPCHK(hipGetLastError());
    if (job_trace) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[job] %d windows: plan %.2f ms (the windows' plans %.2f, shared rows + row checks %.2f, tables + work items %.2f), allocations %.2f ms (workspace %.1f MB, tables %.2f MB, pinned %.2f MB), events + zeroing + tables %.2f ms, table copy queued in %.2f ms\n",
                job->n, ms(tb0, tj0), ms(tb0, tb1), ms(tb1, tb2), ms(tb2, tj0), ms(tj0, tj1), job->ws_bytes / 1e6, job->tab_bytes / 1e6, (pin_tab + 2 * (pin_res + pin_st)) / 1e6, ms(tj1, tj2),
                ms(tj2, std::chrono::steady_clock::now()));
    }
    job->d_probs = (Prob*)(job->d_tab + o_probs);
    job->d_items = (Item*)(job->d_tab + o_items);
    job->d_rowmap = (int2*)(job->d_tab + o_rowmap);
    job->d_tilemap = (int2*)(job->d_tab + o_tilemap);
    job->d_panelmap = (int2*)(job->d_tab + o_panelmap);
    job->d_dpanelmap = (int2*)(job->d_tab + o_dpanelmap);
    job->d_gemmmap = (int2*)(job->d_tab + o_gemmmap);
    job->d_finmap = (int2*)(job->d_tab + o_finmap);
    job->d_exports = (ExportD*)(job->d_tab + o_exports);
    if (!on_device && !streamed) HIPCHK(hipStreamSynchronize(st));   // uploads from pageable user memory are complete
    std::vector<char>().swap(job->h_tab);
    { std::lock_guard<std::mutex> lock(ctx->mu); ctx->jobs.insert(job); }
    *out = guard.release();
    return GAUSS_OK;
}
