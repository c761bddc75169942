// libgauss_host.so -- a whole chromosome (and a genome) as one native call: resident panels, the driver's own window,
// gauss_host_impute_chromosome / gauss_host_impute_genome (the loop over windows the reference leaves to its R user,
// docs/articles/dist_example.md:144-153).
#include "host_internal.h"

// ------------------------------------------------------------------------------------------
// Resident panels: the genotype section of a packed panel file, uploaded once per (context, file) and kept in HBM
// (288 GB hold the whole 33KG panel, 82 GB as 2-bit rows).  Windows then name their rows by index.
// ------------------------------------------------------------------------------------------
struct ResidentPanel {
    std::shared_ptr<PackedPanel> pk;       // keeps the mapping (and so the file identity) alive
    void* dev = nullptr;
    int64_t bytes = 0;
};
// Keyed by the context's id, not its address: ids are never reused, so a context created at the address of a destroyed
// one cannot inherit a stale entry (whose device pointer may even belong to another GPU).  The destroy hook drops a
// context's entries while the context is still whole; the device memory itself goes with the context's stores.
static std::mutex g_res_mu;
static std::map<std::pair<uint64_t, std::string>, ResidentPanel> g_resident;

static void resident_ctx_destroyed(gauss_ctx* ctx, uint64_t id, void*)
{
    std::lock_guard<std::mutex> lock(g_res_mu);
    for (auto it = g_resident.begin(); it != g_resident.end();) {
        if (it->first.first == id) { gauss_store_free(ctx, it->second.dev); it = g_resident.erase(it); }
        else ++it;
    }
}

static std::string file_key(const std::string& path)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return path;
    char key[64];
    snprintf(key, sizeof(key), "|%lld|%lld.%ld", (long long)st.st_size, (long long)st.st_mtim.tv_sec, (long)st.st_mtim.tv_nsec);
    return path + key;
}

// returns the device pointer of the panel's row 0 (uploading the section on first use); *uploaded = bytes moved now
// async: the upload is only STARTED (gauss_store_upload_fd_async); whoever reads rows calls gauss_store_wait for the ones it
// needs first (panel_ready for all of them) -- a cheap no-op once the upload has been retired
int panel_make_resident(gauss_ctx* ctx, const std::string& path, void** dev, int64_t* uploaded, bool async)
{
    if (uploaded) *uploaded = 0;
    const std::pair<uint64_t, std::string> key(gauss_hip_context_id(ctx), file_key(path));
    std::lock_guard<std::mutex> lock(g_res_mu);
    auto it = g_resident.find(key);
    if (it != g_resident.end()) { *dev = it->second.dev; return 0; }
    gauss_hip_add_destroy_hook(resident_ctx_destroyed, nullptr);
    std::string err;
    ResidentPanel rp;
    rp.pk = open_packed_shared(path, err);
    if (!rp.pk) return herr("%s", err.c_str());
    rp.bytes = rp.pk->n_snp() * rp.pk->row_bytes();
    if (rp.bytes <= 0) return herr("packed panel '%s' holds no SNPs", path.c_str());
    if ((async ? gauss_store_upload_fd_async(ctx, rp.pk->fd(), rp.pk->geno_file_offset(), rp.bytes, &rp.dev)
               : gauss_store_upload_fd(ctx, rp.pk->fd(), rp.pk->geno_file_offset(), rp.bytes, &rp.dev)) != 0)
        return herr("%s", gauss_last_error());
    if (uploaded) *uploaded = rp.bytes;
    *dev = rp.dev;
    g_resident[key] = rp;
    return 0;
}

// the entry of a panel that is (being made) resident on this context
static bool panel_entry(gauss_ctx* ctx, const std::string& path, ResidentPanel& out)
{
    const std::pair<uint64_t, std::string> key(gauss_hip_context_id(ctx), file_key(path));
    std::lock_guard<std::mutex> lock(g_res_mu);
    auto it = g_resident.find(key);
    if (it == g_resident.end()) return false;
    out = it->second;
    return true;
}

// Is the panel in HBM on this context?  `wait`: and have all its rows landed -- a background upload that another call (or
// another thread of this one) started is waited for, so that whoever gets the pointer may read any row (a failed upload:
// the entry is dropped and the answer is no).  The chromosome driver asks without waiting: its batches wait for the rows they
// name, one by one.
bool panel_is_resident(gauss_ctx* ctx, const std::string& path, void** dev, bool wait)
{
    ResidentPanel rp;
    if (!panel_entry(ctx, path, rp)) return false;
    if (wait && gauss_store_wait(ctx, rp.dev, 0) != 0) {
        const std::string keep = gauss_last_error();
        gauss_host_panel_evict(ctx, path.c_str());
        herr("%s", keep.c_str());
        return false;
    }
    *dev = rp.dev;
    return true;
}


extern "C" {

int gauss_host_panel_resident(gauss_ctx* ctx, const char* packed_file, int64_t* bytes_uploaded)
{
    if (!ctx || !packed_file) return herr("bad arguments");
    if (!PackedPanel::is_packed(packed_file)) return herr("'%s' is not a packed panel", packed_file);
    void* dev = nullptr;
    if (panel_make_resident(ctx, packed_file, &dev, bytes_uploaded) != 0) return -1;
    // (an upload that another call started in the background: resident means every row has landed)
    if (gauss_store_wait(ctx, dev, 0) != 0) return herr("%s", gauss_last_error());
    return 0;
}

int gauss_host_panel_device_rows(gauss_ctx* ctx, const char* packed_file, const void** out_device_ptr)
{
    if (!ctx || !packed_file || !out_device_ptr) return herr("bad arguments");
    void* dev = nullptr;
    if (!panel_is_resident(ctx, packed_file, &dev)) return herr("packed panel '%s' is not resident on this context", packed_file);
    *out_device_ptr = dev;
    return 0;
}

int gauss_prepared_store_rows(const gauss_prepared* p, const int32_t** rows_m, const int32_t** rows_u,
                              const int32_t** pop_src_off, int* n_pop_selected)
{
    if (!p) return herr("bad arguments");
    if (!p->args.pk) return herr("not a packed-panel window");
    if (rows_m) *rows_m = p->store_rows_m.data();
    if (rows_u) *rows_u = p->store_rows_u.data();
    if (pop_src_off) *pop_src_off = p->pop_src_off.data();
    if (n_pop_selected) *n_pop_selected = (int)p->pop_src_off.size();
    return 0;
}

int gauss_host_panel_evict(gauss_ctx* ctx, const char* packed_file)
{
    if (!ctx) return herr("ctx is NULL");
    std::lock_guard<std::mutex> lock(g_res_mu);
    for (auto it = g_resident.begin(); it != g_resident.end();) {
        const size_t pl = packed_file ? strlen(packed_file) : 0;      // keys are "<path>|<size>|<mtime>"
        if (it->first.first == gauss_hip_context_id(ctx) && (!packed_file || (it->first.second.compare(0, pl, packed_file) == 0 &&
                                                         (it->first.second.size() == pl || it->first.second[pl] == '|')))) {
            gauss_store_free(ctx, it->second.dev);
            it = g_resident.erase(it);
        } else ++it;
    }
    return 0;
}

// Planner cost of a window (the C++ twin of gauss_amd/farm.py:piece_cost; tests/test_farm.py compares them): the flops the Gram
// kernel ISSUES per sample -- 128-row tiles with the kernel's 32 / 16 granular edges, B11's tile triangle with the mirrored parts of
// its diagonal tiles skipped -- plus 8 % on B21's share for what follows it per entry (B21's epilogue tiles, the closing
// product).  The 8-rank emulation on MI355X (round 4) showed the ranks' Gram times following their issued flops to +-2 % while
// their algorithmic flops differed by 4.5 %; the factorisation chain runs under the Gram kernel and costs a rank no time.
static double units32(int rows, bool edge16)
{
    double total = 0;
    for (int t0 = 0; t0 < rows; t0 += 128) {
        const int r = std::min(128, rows - t0);
        for (int w = 0; w < 2; w++) {
            const int left = r - 64 * w;
            if (left <= 0) continue;
            const int n16 = std::min(4, (left + 15) / 16);
            total += (edge16 && (n16 & 1)) ? n16 * 0.5 : std::min(2, (left + 31) / 32);
        }
    }
    return total;
}
static double b11_units(int m)
{
    const int nt = (m + 127) / 128;
    double total = 0;
    for (int ti = 0; ti < nt; ti++) {
        const int ri = std::min(128, m - 128 * ti);
        for (int tj = ti; tj < nt; tj++) {
            const int rj = std::min(128, m - 128 * tj);
            for (int wr = 0; wr < 2; wr++)
                for (int wc = 0; wc < 2; wc++) {
                    if (ti == tj && wr == 1 && wc == 0) continue;
                    const int a = std::min(2, std::max(0, (ri - 64 * wr + 31) / 32)), left = rj - 64 * wc;
                    if (left <= 0 || a == 0) continue;
                    const int n16 = std::min(4, (left + 15) / 16);
                    if (n16 & 1) { total += a * n16 * 0.5; continue; }
                    const int t32 = a * std::min(2, (left + 31) / 32);
                    total += (ti == tj && wr == wc && t32 == 4) ? 3 : t32;
                }
        }
    }
    return total;
}
static double issued_cost_per_sample(int m, int u)
{
    return 2048.0 * b11_units(m) + 2.0 * 32.0 * units32(m, true) * 1.08 * (double)u;
}
double gauss_host_plan_cost(int n_measured, int n_unmeasured) { return issued_cost_per_sample(n_measured, n_unmeasured); }

}  // extern "C"

// ------------------------------------------------------------------------------------------
// The chromosome driver's window: prepare()'s data layer as ONE merge of two sorted arrays.
//
// prepare() restates the reference literally: every SNP of the extended window becomes an object with five strings in a std::map
// keyed by (chr, bp, a1, a2) (ReadInputZ, ReadReferenceIndex, MakeSnpVec: gauss.cpp:121-190, 293-399, 543-693) -- 0.65-1.0 ms
// for a window of 3 000 SNPs, paid again by every window of a chromosome, and with one rank of eight holding four windows it is
// what the GPU waits for (docs/HISTORY.md section 9e item 10).  A window of a SORTED packed panel needs none of it.  The study's rows
// (cached, ordered by position) and the panel's SNP table (ordered by position) ascend together; a position is settled where
// the two walks meet:
//
//   panel entry alone at its position, no study row there ......... type 0 (an unmeasured SNP)
//   one panel entry, one study row, same alleles .................. type 1, the study's z
//   one panel entry, one study row, alleles swapped ............... type 1, -z (the panel's order is adopted, gauss.cpp:362-372)
//   one panel entry, one study row, other alleles ................. the panel entry is type 0; the study SNP has no panel row
//   study rows without a panel entry .............................. type 2: no panel row, the AF filter drops them (gauss.cpp:574)
//   anything else (a site the panel or the study lists more than once, equal alleles): the position's rows go through the
//   very map code of prepare() -- a map of that one position, whose entries only ever meet entries of their own position
//   (merge_index_entry looks up (chr, bp, a1, a2) and (chr, bp, a2, a1)) -- and come out in the map's order.
//
// Every string of the output (rsid, a1, a2) is the panel's: a type-1 SNP takes the panel's rsid and, matched or swapped, has
// the panel's alleles.  So a SNP is a panel row number, z, info, af and a type -- 48 bytes, no allocation -- and the tables
// read the strings out of the panel's pool when they are built.  Type-0 SNPs of the wings are not entered (the partition reads
// type 0 inside the prediction window only, dist.cpp:132-140 / qcat.cpp:140-152, and the tables are cut to it).
// What a window costs now: docs/HISTORY.md section 9e item 11.  tests/test_feeder.py holds it against prepare() on random
// studies with multi-allelic, duplicated, swapped and study-only sites, for all four kinds.
// ------------------------------------------------------------------------------------------
// 0, or -1 with the message prepare() would have given every window (the caller then lets prepare() give it)
int chrom_setup(ChromSetup& cs, int kind, int chr, int64_t wing_size, const char* study_pop, const char* const* pop_names,
                       const double* pop_wgts, int n_pop_wgt, const char* input_file, const std::string& packed_path,
                       const char* desc_file, double af1_cutoff, const std::shared_ptr<PackedPanel>& pk,
                       const std::shared_ptr<const GwasCache>& gw)
{
    cs.kind = kind;
    cs.mix = (kind == GAUSS_KIND_DISTMIX || kind == GAUSS_KIND_QCATMIX || kind == GAUSS_KIND_COMPUTELD);
    cs.measured_only = (kind == GAUSS_KIND_COMPUTELD);
    cs.qcat = (kind == GAUSS_KIND_QCAT || kind == GAUSS_KIND_QCATMIX);
    Args& a = cs.a;
    a.chr = chr; a.wing_size = wing_size;
    if (study_pop) a.study_pop = study_pop;
    a.input_file = input_file; a.reference_data_file = packed_path; a.reference_pop_desc_file = desc_file;
    a.pk = pk;
    a.drop_wing_unmeasured = (kind == GAUSS_KIND_DIST || kind == GAUSS_KIND_DISTMIX);
    a.af1_cutoff = std::isnan(af1_cutoff) ? (kind == GAUSS_KIND_QCAT ? 0.05 : 0.01) : af1_cutoff;
    if (cs.mix) {
        if (!pop_names || !pop_wgts || n_pop_wgt < 1) return herr("pop_wgt_df is empty");
        set_pop_wgt_map(a, pop_names, pop_wgts, n_pop_wgt);
    } else if (!study_pop) return herr("study_pop is NULL");
    if (read_ref_desc(a)) return -1;
    if (pk->n_pop() != a.num_pops) return herr("packed panel has %d populations, the description file %d", pk->n_pop(), a.num_pops);
    for (int k = 0; k < a.num_pops; k++)
        if (a.ref_pop_vec[k] != pk->pop(k).name || a.ref_pop_size_vec[k] != (int)pk->pop(k).size)
            return herr("packed panel population %d is %s (%u samples), the description file says %s (%d)", k,
                        pk->pop(k).name, pk->pop(k).size, a.ref_pop_vec[k].c_str(), a.ref_pop_size_vec[k]);
    if (cs.mix) init_pop_flag_wgt_vec(a);
    else if (init_pop_flag_vec(a)) return -1;
    cs.pop_off.assign(1, 0);
    double num_subj = 0;
    for (int k = 0; k < a.num_pops; k++)
        if (a.pop_flag_vec[k]) {
            cs.sel.push_back(k);
            cs.pop_off.push_back(cs.pop_off.back() + a.ref_pop_size_vec[k]);
            cs.pop_src_off.push_back((int32_t)pk->pop(k).byte_off);
            num_subj += a.ref_pop_size_vec[k];
        }
    cs.two_subj = 2 * num_subj;
    if (cs.mix) cs.pop_wgt = a.pop_wgt_vec;
    else cs.pop_wgt.assign(cs.pop_off.size() - 1, 1.0);
    cs.gw = gw;
    return 0;
}

int lean_window_build(LeanWindow& w, const ChromSetup& cs, long long start_bp, long long end_bp)
{
    const Args& a = cs.a;
    const PackedPanel& pk = *a.pk;
    const GwasCache& gw = *cs.gw;
    w.cs = &cs; w.start_bp = start_bp; w.end_bp = end_bp;
    const long long lo = start_bp - a.wing_size, hi = end_bp + a.wing_size;
    auto before = [&](uint32_t x, long long bp) { const GwasRow& r = gw.rows[x]; return r.chr < a.chr || (r.chr == a.chr && r.bp < bp); };
    size_t q = (size_t)(std::lower_bound(gw.by_pos.begin(), gw.by_pos.end(), lo, before) - gw.by_pos.begin());
    const size_t q1 = (size_t)(std::lower_bound(gw.by_pos.begin(), gw.by_pos.end(), hi + 1, before) - gw.by_pos.begin());
    int64_t i = pk.lower_bound(a.chr, lo);
    const int64_t i1 = pk.lower_bound(a.chr, hi + 1);
    w.v.reserve((size_t)std::max<int64_t>(i1 - i, 0));
    const double cutoff = a.af1_cutoff;
    // MakeSnpVec / MakeSnpVecMix on the panel's tabulated counts and frequencies (MakeSnpVecPacked above), then the list
    auto keep = [&](int64_t row, long long bp, int type, double z, double info) {
        if (type == 0 && (cs.measured_only || bp < start_bp || bp > end_bp)) return;       // a wing's unmeasured SNP (computeLD: any): nothing reads it
        double af = 0;
        if (!cs.mix) {
            double allele_counter = 0;                                  // gauss.cpp:574-591 (integer-valued sums)
            const int32_t* c = pk.cnt(row);
            for (int k : cs.sel) allele_counter += (double)c[k];
            af = allele_counter / cs.two_subj;
            af = std::ceil(af * 100000.0) / 100000.0;
        } else {
            const double* f = pk.af(row);                               // gauss.cpp:676-682
            int j = 0;
            for (int k : cs.sel) af += f[k] * a.pop_wgt_vec[j++];
        }
        if (!((af > cutoff) && (af < (1 - cutoff)))) return;
        w.v.push_back(LeanSnp{row, bp, z, info, af, type, 0, 0.0, 0.0});
    };
    std::unique_ptr<Args> range;                                        // merge_index_entry's window filter, for the odd positions
    while (i < i1) {
        const PkSnp& s = pk.snp(i);
        const long long bp = s.bp;
        int64_t ie = i + 1;
        while (ie < i1 && pk.snp(ie).bp == bp) ie++;
        while (q < q1 && gw.rows[gw.by_pos[q]].bp < bp) q++;            // study-only positions
        size_t qe = q;
        while (qe < q1 && gw.rows[gw.by_pos[qe]].bp == bp) qe++;
        if (ie - i == 1 && qe == q) { keep(i, bp, 0, 0.0, -1.0); i = ie; continue; }    // the rule: nothing of the study here (no string is read)
        const char *pa1 = pk.str(s.a1), *pa2 = pk.str(s.a2);
        if (ie - i == 1 && qe - q == 1 && strcmp(pa1, pa2) != 0) {
            {
                const GwasRow& r = gw.rows[gw.by_pos[q]];
                if (r.a1 == pa1 && r.a2 == pa2) keep(i, bp, 1, r.z, 1.0);
                else if (r.a1 == pa2 && r.a2 == pa1) keep(i, bp, 1, r.z * (-1), 1.0);
                else keep(i, bp, 0, 0.0, -1.0);
            }
        } else {
            // the position as prepare() handles it: ReadInputZ's rows (a key listed twice ends with its later row), then the
            // panel's entries in panel order
            SnpMap m;
            for (size_t k = q; k < qe; k++) {
                const GwasRow& r = gw.rows[gw.by_pos[k]];
                SnpPtr sp = m.make();
                sp->rsid = r.rsid; sp->chr = r.chr; sp->bp = r.bp; sp->a1 = r.a1; sp->a2 = r.a2; sp->z = r.z;
                sp->info = 1.0; sp->type = 2;
                m.try_emplace(MapKey{r.chr, r.bp, r.a1, r.a2}).first->second = std::move(sp);
            }
            if (!range) { range.reset(new Args()); range->chr = 0; range->start_bp = lo; range->end_bp = hi; range->wing_size = 0; }
            for (int64_t j = i; j < ie; j++) {
                const PkSnp& sj = pk.snp(j);
                if (m.empty()) {
                    if (a.drop_wing_unmeasured && ie - i == 1 && (bp < start_bp || bp > end_bp)) continue;
                    SnpPtr sp = m.make();                              // gauss.cpp:373-385
                    sp->rsid = pk.str(sj.rsid); sp->chr = sj.chr; sp->bp = sj.bp; sp->a1 = pk.str(sj.a1); sp->a2 = pk.str(sj.a2); sp->type = 0; sp->fpos = j;
                    m.emplace(MapKey{sj.chr, sj.bp, sp->a1, sp->a2}, std::move(sp));
                } else if (merge_index_entry(m, *range, false, pk.str(sj.rsid), sj.chr, sj.bp, pk.str(sj.a1), pk.str(sj.a2), j)) return -1;
            }
            for (auto& kv : m) {
                const Snp& sn = *kv.second;
                if (sn.fpos < 0 || sn.fpos >= pk.n_snp()) continue;     // a study SNP without a panel row
                keep(sn.fpos, bp, sn.type, sn.z, sn.info);
            }
        }
        i = ie; q = qe;
    }
    // the partition: dist.cpp:132-140, qcat.cpp:140-152
    for (size_t r = 0; r < w.v.size(); r++) {
        const LeanSnp& sn = w.v[r];
        if (sn.type == 0) { w.unmeasured.push_back((int32_t)r); w.store_rows_u.push_back((int32_t)sn.row); }   // (inside the prediction window: keep())
        else if (sn.type == 1) {
            w.measured.push_back((int32_t)r); w.store_rows_m.push_back((int32_t)sn.row); w.z1.push_back(sn.z);
            if (sn.bp < start_bp) w.n_head++;
            else if (sn.bp <= end_bp) w.n_predm++;
        }
    }
    return 0;
}

// gauss_prepared_window_desc for the four window kinds (same guards, same texts); geno_m / geno_u are set by the caller
int lean_window_desc(LeanWindow& w, gauss_window_desc* d)
{
    const ChromSetup& cs = *w.cs;
    const Args& a = cs.a;
    const LeanWindow& src = lean_src(w);
    const int M = (int)src.measured.size(), U = (int)src.unmeasured.size();
    if (cs.qcat) {
        if (cs.kind == GAUSS_KIND_QCAT && M <= a.min_num_measured_snp)
            return herr("Not enough number of SNPs loaded - QCAT not performed (measured %d, unmeasured %d)", M, U);
        if (cs.kind == GAUSS_KIND_QCATMIX && (M <= a.min_num_measured_snp || U <= a.min_num_unmeasured_snp))
            return herr("Not enough number of SNPs loaded - QCAT performed (measured %d, unmeasured %d)", M, U);
    } else if (M <= a.min_num_measured_snp || U <= a.min_num_unmeasured_snp)
        return herr("Not enough number of SNPs loaded - %s not performed (measured %d, unmeasured %d)",
                    cs.kind == GAUSS_KIND_DIST ? "DIST" : "DISTMIX", M, U);
    memset(d, 0, sizeof(*d));
    d->mode = cs.mix ? GAUSS_MODE_WEIGHTED : GAUSS_MODE_POOLED;
    d->n_pop = (int)cs.pop_off.size() - 1;
    d->pop_off = cs.pop_off.data(); d->pop_wgt = cs.pop_wgt.data();
    d->n_measured = M; d->n_unmeasured = U;
    d->geno_format = GAUSS_GENO_2BIT;
    d->ld = a.pk->row_bytes();
    d->rows_m = src.store_rows_m.data(); d->rows_u = src.store_rows_u.data();
    d->pop_src_off = cs.pop_src_off.data();
    d->z1 = src.z1.data(); d->lambda = a.lambda; d->min_abs_eig = a.min_abs_eig;
    d->out_status = &w.status;
    if (cs.qcat) {
        w.out_r.assign((size_t)src.n_predm + U, 0.0);
        w.num_eig = M;
        d->kind = GAUSS_WIN_QCAT;
        d->n_head_measured = src.n_head; d->n_pred_measured = src.n_predm; d->eig_cutoff = a.eig_cutoff;
        d->out_r = w.out_r.data(); d->out_num_eig = &w.num_eig;
        if (src.n_predm + U < 1) return herr("QCAT window has no SNP to test");
        return 0;
    }
    w.out_z.assign(U, 0.0); w.out_info.assign(U, 0.0);
    d->out_z = w.out_z.data(); d->out_info = w.out_info.data();
    return 0;
}

// gauss_prepared_finish + dist_output / qcat_output in two steps.  Everything a table holds that does not wait for the GPU -- the
// strings, positions, frequencies, the measured SNPs' z and p-values -- is built while the window's batch computes
// (lean_table_prebuild, before the driver waits for the batch); what the results change is filled in after (lean_window_finish):
// the last batch's tables are otherwise the tail of the call that nothing overlaps.
void lean_table_prebuild(LeanWindow& w)
{
    if (w.pre) return;
    const ChromSetup& cs = *w.cs;
    const PackedPanel& pk = *cs.a.pk;
    std::unique_ptr<gauss_table> t(new gauss_table());
    Column rsid{"rsid", GAUSS_COL_STR, {}, {}, {}}, chr{"chr", GAUSS_COL_INT, {}, {}, {}}, bp{"bp", GAUSS_COL_INT, {}, {}, {}};
    Column a1{"a1", GAUSS_COL_STR, {}, {}, {}}, a2{"a2", GAUSS_COL_STR, {}, {}, {}};
    Column af{cs.mix ? "af1mix" : "af1ref", GAUSS_COL_DBL, {}, {}, {}}, z{"z", GAUSS_COL_DBL, {}, {}, {}};
    Column pval{"pval", GAUSS_COL_DBL, {}, {}, {}}, info{"info", GAUSS_COL_DBL, {}, {}, {}}, type{"type", GAUSS_COL_INT, {}, {}, {}};
    Column qm{"qcat_m", GAUSS_COL_INT, {}, {}, {}}, qt{"qcat_t", GAUSS_COL_DBL, {}, {}, {}};
    Column qc{"qcat_chisq", GAUSS_COL_DBL, {}, {}, {}}, qp{"qcat_pval", GAUSS_COL_DBL, {}, {}, {}};
    size_t n_out = 0;
    w.out_row.assign(w.v.size(), -1);
    for (size_t r = 0; r < w.v.size(); r++) {
        const int ibp = (int)w.v[r].bp;                                           // dist.cpp:92, qcat.cpp:95
        if (ibp >= w.start_bp && ibp <= w.end_bp) w.out_row[r] = (int32_t)n_out++;
    }
    for (Column* c : {&rsid, &a1, &a2}) c->s.reserve(n_out);
    for (Column* c : {&chr, &bp, &type}) c->i.reserve(n_out);
    for (Column* c : {&af, &z}) c->d.reserve(n_out);
    if (cs.qcat) { qm.i.reserve(n_out); for (Column* c : {&qt, &qc, &qp}) c->d.reserve(n_out); }
    else for (Column* c : {&pval, &info}) c->d.reserve(n_out);
    for (size_t r = 0; r < w.v.size(); r++) {
        if (w.out_row[r] < 0) continue;
        const LeanSnp& sn = w.v[r];
        const PkSnp& ps = pk.snp(sn.row);
        rsid.s.emplace_back(pk.str(ps.rsid)); chr.i.push_back(ps.chr); bp.i.push_back((int)sn.bp);
        a1.s.emplace_back(pk.str(ps.a1)); a2.s.emplace_back(pk.str(ps.a2));
        af.d.push_back(sn.af); z.d.push_back(sn.z); type.i.push_back(sn.type);
        if (cs.qcat) {
            qm.i.push_back(sn.qcat_m); qt.d.push_back(sn.qcat_t); qc.d.push_back(sn.qcat_chisq);
            qp.d.push_back(pchisq_upper(sn.qcat_chisq, 1));                       // qcat.cpp:107
        } else {
            pval.d.push_back(2 * pnorm_upper(fabs(sn.z)));                        // dist.cpp:101
            info.d.push_back(sn.info);
        }
    }
    t->cols.reserve(12);
    if (cs.qcat) for (Column* c : {&rsid, &chr, &bp, &a1, &a2, &af, &z, &qm, &qt, &qc, &qp, &type}) t->cols.push_back(std::move(*c));
    else for (Column* c : {&rsid, &chr, &bp, &a1, &a2, &af, &z, &pval, &info, &type}) t->cols.push_back(std::move(*c));
    w.pre = std::move(t);
}

std::unique_ptr<LeanWindow> lean_window_clone(const LeanWindow& w)
{
    std::unique_ptr<LeanWindow> c(new LeanWindow());
    c->cs = w.cs; c->start_bp = w.start_bp; c->end_bp = w.end_bp;
    c->v = w.v; c->measured = w.measured; c->unmeasured = w.unmeasured;
    c->store_rows_m = w.store_rows_m; c->store_rows_u = w.store_rows_u; c->z1 = w.z1;
    c->n_head = w.n_head; c->n_predm = w.n_predm;
    return c;
}

int lean_table_count(LeanWindow& w)
{
    if (w.n_out >= 0) return w.n_out;
    if (w.base && w.base->n_out >= 0) { w.n_out = w.base->n_out; return w.n_out; }        // (counted once, when the window was cached)
    size_t n_out = 0;
    w.out_row.assign(w.v.size(), -1);
    for (size_t r = 0; r < w.v.size(); r++) {
        const int ibp = (int)w.v[r].bp;                                           // dist.cpp:92, qcat.cpp:95
        if (ibp >= w.start_bp && ibp <= w.end_bp) w.out_row[r] = (int32_t)n_out++;
    }
    w.n_out = (int)n_out;
    return w.n_out;
}

// column order of the reference's tables (dist.cpp:112-124, qcat.cpp:118-131), as lean_table_prebuild lays them out
enum { LC_RSID = 0, LC_CHR, LC_BP, LC_A1, LC_A2, LC_AF, LC_Z, LC_7, LC_8, LC_9, LC_10, LC_11 };

void lean_table_prebuild_into(LeanWindow& w, gauss_table& all, size_t off)
{
    const ChromSetup& cs = *w.cs;
    const PackedPanel& pk = *cs.a.pk;
    lean_table_count(w);
    w.tab_off = off;
    const LeanWindow& src = lean_src(w);
    std::vector<Column>& c = all.cols;
    for (size_t r = 0; r < src.v.size(); r++) {
        if (src.out_row[r] < 0) continue;
        const size_t o = off + (size_t)src.out_row[r];
        const LeanSnp& sn = src.v[r];
        const PkSnp& ps = pk.snp(sn.row);
        c[LC_RSID].s[o] = pk.str(ps.rsid); c[LC_CHR].i[o] = ps.chr; c[LC_BP].i[o] = (int)sn.bp;
        c[LC_A1].s[o] = pk.str(ps.a1); c[LC_A2].s[o] = pk.str(ps.a2);
        c[LC_AF].d[o] = sn.af; c[LC_Z].d[o] = sn.z;
        if (cs.qcat) {
            c[7].i[o] = sn.qcat_m; c[8].d[o] = sn.qcat_t; c[9].d[o] = sn.qcat_chisq;
            c[10].d[o] = pchisq_upper(sn.qcat_chisq, 1);                          // qcat.cpp:107
            c[11].i[o] = sn.type;
        } else {
            c[7].d[o] = 2 * pnorm_upper(fabs(sn.z));                              // dist.cpp:101
            c[8].d[o] = sn.info;
            c[9].i[o] = sn.type;
        }
    }
}

void lean_window_finish_into(LeanWindow& w, gauss_table& all)
{
    const ChromSetup& cs = *w.cs;
    const LeanWindow& src = lean_src(w);
    std::vector<Column>& c = all.cols;
    const size_t off = w.tab_off;
    if (cs.qcat) {
        const int m = w.num_eig;
        for (size_t k = 0; k < w.out_r.size(); k++) {                            // qcat.cpp:216-243
            const size_t vi = (size_t)((k < (size_t)src.n_predm) ? src.measured[(size_t)src.n_head + k] : src.unmeasured[k - (size_t)src.n_predm]);
            const double r = w.out_r[k];
            const double qt = std::sqrt((double)(m - 3)) * r, qc = (m - 3) * r * r;
            const int32_t row = src.out_row[vi];
            if (row < 0) continue;
            const size_t o = off + (size_t)row;
            c[7].i[o] = m; c[8].d[o] = qt; c[9].d[o] = qc;
            c[10].d[o] = pchisq_upper(qc, 1);                                     // qcat.cpp:107
        }
        return;
    }
    for (size_t i = 0; i < src.unmeasured.size() && i < w.out_z.size(); i++) {    // dist.cpp:200-202
        const int32_t row = src.out_row[(size_t)src.unmeasured[i]];
        if (row < 0) continue;
        const size_t o = off + (size_t)row;
        const double zz = w.out_z[i];
        c[LC_Z].d[o] = zz; c[8].d[o] = w.out_info[i];
        c[7].d[o] = 2 * pnorm_upper(fabs(zz));                                    // dist.cpp:101
    }
}

gauss_table* lean_window_finish(LeanWindow& w)
{
    const ChromSetup& cs = *w.cs;
    lean_table_prebuild(w);
    gauss_table& t = *w.pre;
    if (cs.qcat) {
        Column &qm = t.cols[7], &qt = t.cols[8], &qc = t.cols[9], &qp = t.cols[10];
        const int m = w.num_eig;
        for (size_t k = 0; k < w.out_r.size(); k++) {                            // qcat.cpp:216-243
            const size_t vi = (size_t)((k < (size_t)w.n_predm) ? w.measured[(size_t)w.n_head + k] : w.unmeasured[k - (size_t)w.n_predm]);
            LeanSnp& sn = w.v[vi];
            const double r = w.out_r[k];
            sn.qcat_m = m;
            sn.qcat_t = std::sqrt((double)(m - 3)) * r;
            sn.qcat_chisq = (m - 3) * r * r;
            const int32_t row = w.out_row[vi];
            if (row < 0) continue;
            qm.i[(size_t)row] = sn.qcat_m; qt.d[(size_t)row] = sn.qcat_t; qc.d[(size_t)row] = sn.qcat_chisq;
            qp.d[(size_t)row] = pchisq_upper(sn.qcat_chisq, 1);                   // qcat.cpp:107
        }
    } else {
        Column &z = t.cols[6], &pval = t.cols[7], &info = t.cols[8];
        for (size_t i = 0; i < w.unmeasured.size() && i < w.out_z.size(); i++) {  // dist.cpp:200-202
            LeanSnp& sn = w.v[(size_t)w.unmeasured[i]];
            sn.z = w.out_z[i];
            sn.info = w.out_info[i];
            const int32_t row = w.out_row[(size_t)w.unmeasured[i]];
            if (row < 0) continue;
            z.d[(size_t)row] = sn.z; info.d[(size_t)row] = sn.info;
            pval.d[(size_t)row] = 2 * pnorm_upper(fabs(sn.z));                    // dist.cpp:101
        }
    }
    return w.pre.release();
}

extern "C" {

// ------------------------------------------------------------------------------------------
// The chromosome driver's plan: windows, their costs, their owners (identical on every rank: no communication)
// ------------------------------------------------------------------------------------------
struct ChromWin { int64_t s, e; double cost; int owner, status, M, U; std::string why; };

// The windows of [start_bp, end_bp] with their predicted sizes: measured SNPs from the study (rows of the extended window), panel
// SNPs of the prediction window from the packed index, unmeasured = the difference.  One chromosome: counts are ranges of the
// study's (chr, bp)-ordered index (no copy, no sort: that copy and sort were most of the plan's 0.13 ms).
static void plan_windows(std::vector<ChromWin>& wins, const GwasCache& gw, const PackedPanel& pk, int chr, int64_t start_bp, int64_t end_bp,
                         int64_t wing_size, int64_t window_size)
{
    std::vector<long long> gbp;                                   // a call over every chromosome: positions of all rows, sorted
    if (chr <= 0) {
        for (const GwasRow& r : gw.rows) gbp.push_back(r.bp);
        std::sort(gbp.begin(), gbp.end());
    }
    auto before = [&](uint32_t x, long long bp) { const GwasRow& r = gw.rows[x]; return r.chr < chr || (r.chr == chr && r.bp < bp); };
    auto rows_in = [&](long long lo, long long hi) -> double {    // study rows with lo <= bp <= hi
        if (chr <= 0) return (double)(std::upper_bound(gbp.begin(), gbp.end(), hi) - std::lower_bound(gbp.begin(), gbp.end(), lo));
        return (double)(std::lower_bound(gw.by_pos.begin(), gw.by_pos.end(), hi + 1, before) - std::lower_bound(gw.by_pos.begin(), gw.by_pos.end(), lo, before));
    };
    for (int64_t s0 = start_bp; s0 <= end_bp; s0 += window_size) {
        ChromWin w;
        w.s = s0; w.e = std::min(end_bp, s0 + window_size - 1);
        const double m = rows_in((long long)(w.s - wing_size), (long long)(w.e + wing_size));
        double u = 0;
        if (pk.header().sorted && chr > 0) {
            const double in_panel = (double)(pk.lower_bound(chr, w.e + 1) - pk.lower_bound(chr, w.s));
            u = std::max(0.0, in_panel - rows_in((long long)w.s, (long long)w.e));
        }
        w.cost = issued_cost_per_sample((int)m, (int)u) + 1.0;   // what the Gram kernel issues for the window, per sample (above)
        w.owner = 0; w.status = -1; w.M = (int)m; w.U = (int)u;
        wins.push_back(w);
    }
}

static void plan_owners(std::vector<ChromWin>& wins, const PackedPanel& pk_, int world)
{
    const PackedPanel* pk = &pk_;
    {   // Whole windows by longest processing time first (ties by index), then a local search -- the idea of
        // farm.level_windows without the cuts, on a simpler cost model: a window costs its LD flops per sample (the
        // Python planner also prices B11's factorisation and the per-SNP tail, so the two plans may pick different
        // owners; each is used consistently by every rank of its own driver).  A rank's load is the cost of its windows
        // plus the factorisation chain of its tallest one (the chain is latency bound on the few windows a rank
        // holds: DESIGN.md section 8; farm.CHAIN_STEP_COST, here per sample).
        std::vector<int> order(wins.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return wins[a].cost > wins[b].cost; });
        std::vector<double> load((size_t)world, 0.0);
        for (int i : order) {
            int r = 0;
            for (int k = 1; k < world; k++) if (load[k] < load[r]) r = k;
            wins[i].owner = r; load[r] += wins[i].cost;
        }
        double n_samples = 0;
        for (uint32_t q = 0; q < pk->header().n_pop; q++) n_samples += pk->pop((int)q).size;
        const double chain = 2.0e8 / std::max(1.0, n_samples);       // (a tie-breaker since the chain runs under the Gram kernel: farm.CHAIN_STEP_COST)
        // Local search of moves (a window of the fullest rank goes to another rank) and trades.  Per rank: the summed cost
        // and the block counts of its windows (the chain term needs the tallest one, also "the tallest without window
        // x"), so a candidate costs O(1); candidates are counted and the search stops at a fixed budget -- the same
        // on every rank, which must all arrive at the same plan (no wall-clock limits) -- so a chromosome cut into
        // thousands of windows plans in milliseconds too (it used to be cubic in the window count).
        auto nblk = [&](int i) { return (wins[i].M + 63) / 64; };
        std::vector<double> rcost((size_t)world, 0.0);
        std::vector<std::multiset<int>> rtall((size_t)world);
        std::vector<std::vector<int>> rwin((size_t)world);
        auto rebuild = [&]() {
            for (int r = 0; r < world; r++) { rcost[r] = 0; rtall[r].clear(); rwin[r].clear(); }
            for (size_t i = 0; i < wins.size(); i++) {
                const int r = wins[i].owner;
                rcost[r] += wins[i].cost; rtall[r].insert(nblk((int)i)); rwin[r].push_back((int)i);
            }
        };
        auto load_of = [&](int r, int drop, int add) {            // rank r's load without window `drop`, with window `add`
            double c = rcost[r];
            int tall = 0;
            if (drop >= 0) {
                c -= wins[drop].cost;
                auto it = rtall[r].end();
                if (!rtall[r].empty()) {
                    --it;                                          // the tallest; if that is `drop` itself, the next one
                    if (*it == nblk(drop)) { if (it != rtall[r].begin()) { --it; tall = *it; } }
                    else tall = *it;
                }
            } else if (!rtall[r].empty()) tall = *rtall[r].rbegin();
            if (add >= 0) { c += wins[add].cost; tall = std::max(tall, nblk(add)); }
            return c + chain * tall;
        };
        rebuild();
        long long budget = 4000000;                                // candidate evaluations
        for (size_t it = 0; world > 1 && it < 4 * wins.size() && budget > 0; it++) {
            int hi = 0; double top = -1;
            for (int r = 0; r < world; r++) { const double l = load_of(r, -1, -1); if (l > top) { top = l; hi = r; } }
            double best = top * (1 - 1e-9); int ba = -1, bb = -1, br = -1;
            for (int a : rwin[hi]) {
                for (int r = 0; r < world && budget > 0; r++) {
                    if (r == hi) continue;
                    budget -= 1 + (long long)rwin[r].size();
                    double m = std::max(load_of(hi, a, -1), load_of(r, -1, a));          // move a to r
                    if (m < best) { best = m; ba = a; bb = -1; br = r; }
                    for (int b2 : rwin[r]) {                                              // trade a for b2
                        m = std::max(load_of(hi, a, b2), load_of(r, b2, a));
                        if (m < best) { best = m; ba = a; bb = b2; br = r; }
                    }
                }
            }
            if (ba < 0) break;
            wins[ba].owner = br;
            if (bb >= 0) wins[bb].owner = hi;
            rebuild();
        }
    }
}

// ------------------------------------------------------------------------------------------
// A whole chromosome: the caller-level loop over windows that the reference leaves to the R user
// (docs/articles/dist_example.md:144-153 calls one window), as ONE native call per rank.
//
//   windows   [start + k*window_size, ...] over [start_bp, end_bp]; sharded over `world` ranks by LPT on their LD
//             flops (measured count from the GWAS file, panel count from the packed index; the same list on every
//             rank, no communication)
//   pipeline  this rank's windows are cut into batches; host threads run the data layer of batch b+1 (window
//             membership, allele matching, AF filter: gauss_host_prepare) and build the tables of batch b-1 while
//             the GPU works on batch b: jobs are created and queued without waiting for the stream, results come
//             back through a per-job event.  The panel's rows are resident in HBM (uploaded once per context and
//             file through pinned double buffers), so a window carries only row indices to the device.
//   failures  a window that fails its guards (dist.cpp:145-151) or its data layer is reported in the "windows"
//             matrix and skipped; a batch whose job fails is retried window by window, so one bad window never
//             takes the rank's other windows with it.
// Result: the reference's output table for every window of this rank, concatenated in window order, plus an int
// column "window"; named matrix "windows" [n_windows x 6]: start_bp end_bp owner status measured unmeasured
// (status 0 done, 1 skipped by the ">10" guards, 2 failed, -1 another rank's).
// ------------------------------------------------------------------------------------------
// A test / debugging view of ONE window as the chromosome driver builds it (no GPU involved): the SNP list with the columns of
// gauss_prepared_snps (fpos = panel row) and the named matrices "rows_m", "rows_u", "z1", "counts" = [M, U, n_head, n_predm].
// tests/test_feeder.py holds it against gauss_host_prepare on the same arguments.
int gauss_host_chrom_window_view(int kind, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size, const char* study_pop,
                                 const char* const* pop_names, const double* pop_wgts, int n_pop_wgt, const char* input_file,
                                 const char* packed_file, const char* reference_pop_desc_file, double af1_cutoff, gauss_table** out)
{
    if (!out || !input_file || !packed_file || !reference_pop_desc_file) return herr("bad arguments");
    if (kind != GAUSS_KIND_DIST && kind != GAUSS_KIND_DISTMIX && kind != GAUSS_KIND_QCAT && kind != GAUSS_KIND_QCATMIX)
        return herr("gauss_host_chrom_window_view: kind must be dist, distmix, qcat or qcatmix");
    std::string err;
    std::shared_ptr<PackedPanel> pk = open_packed_shared(packed_file, err);
    if (!pk) return herr("%s", err.c_str());
    if (chr <= 0 || !pk->header().sorted) return herr("the chromosome driver builds its own windows on a sorted packed panel and one chromosome only");
    std::shared_ptr<const GwasCache> gw = load_gwas_cached(input_file, err);
    if (!gw) return herr("%s", err.c_str());
    const double t0 = now_s();
    ChromSetup cs;
    if (chrom_setup(cs, kind, chr, wing_size, study_pop, pop_names, pop_wgts, n_pop_wgt, input_file, packed_file, reference_pop_desc_file,
                    af1_cutoff, pk, gw)) return -1;
    const double t1 = now_s();
    LeanWindow w;
    if (lean_window_build(w, cs, start_bp, end_bp)) return -1;
    if (host_trace("prep"))
        fprintf(stderr, "[window] setup %.3f ms (once per call in the driver), build %.3f ms (list %zu, measured %zu, unmeasured %zu)\n",
                (t1 - t0) * 1e3, (now_s() - t1) * 1e3, w.v.size(), w.measured.size(), w.unmeasured.size());
    std::unique_ptr<gauss_table> t(new gauss_table());
    Column& rsid = t->add("rsid", GAUSS_COL_STR);
    for (const LeanSnp& sn : w.v) rsid.s.emplace_back(pk->str(pk->snp(sn.row).rsid));
    Column& cchr = t->add("chr", GAUSS_COL_INT);
    for (const LeanSnp& sn : w.v) cchr.i.push_back(pk->snp(sn.row).chr);
    Column& bp = t->add("bp", GAUSS_COL_INT);
    for (const LeanSnp& sn : w.v) bp.i.push_back((int)sn.bp);
    Column& a1 = t->add("a1", GAUSS_COL_STR);
    for (const LeanSnp& sn : w.v) a1.s.emplace_back(pk->str(pk->snp(sn.row).a1));
    Column& a2 = t->add("a2", GAUSS_COL_STR);
    for (const LeanSnp& sn : w.v) a2.s.emplace_back(pk->str(pk->snp(sn.row).a2));
    Column& af = t->add(cs.mix ? "af1mix" : "af1ref", GAUSS_COL_DBL);
    for (const LeanSnp& sn : w.v) af.d.push_back(sn.af);
    Column& z = t->add("z", GAUSS_COL_DBL);
    for (const LeanSnp& sn : w.v) z.d.push_back(sn.z);
    Column& info = t->add("info", GAUSS_COL_DBL);
    for (const LeanSnp& sn : w.v) info.d.push_back(sn.info);
    Column& type = t->add("type", GAUSS_COL_INT);
    for (const LeanSnp& sn : w.v) type.i.push_back(sn.type);
    Column& fpos = t->add("fpos", GAUSS_COL_DBL);
    for (const LeanSnp& sn : w.v) fpos.d.push_back((double)sn.row);
    auto named = [&](const char* name, const std::vector<double>& v) {
        NamedMat nm;
        nm.name = name; nm.nrow = (int)v.size(); nm.ncol = 1; nm.d = v;
        t->named.push_back(std::move(nm));
    };
    named("rows_m", std::vector<double>(w.store_rows_m.begin(), w.store_rows_m.end()));
    named("rows_u", std::vector<double>(w.store_rows_u.begin(), w.store_rows_u.end()));
    named("z1", w.z1);
    named("counts", std::vector<double>{(double)w.measured.size(), (double)w.unmeasured.size(), (double)w.n_head, (double)w.n_predm});
    gauss_window_desc d;
    if (lean_window_desc(w, &d)) t->messages.push_back(gauss_host_last_error());     // the guard's text, as the driver would report it
    *out = t.release();
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// The window cache.  A built window -- which panel rows are its measured / unmeasured SNPs, their z, their filtered frequencies --
// is a function of the panel, the study file and the call's arguments alone; like the parsed study (load_gwas_cached) and the
// opened panel (open_packed_shared) it is kept across calls, keyed by their identity: a session that imputes a study again (another
// kind on the same windows, the same call after a failure further on, a benchmark's warm calls) finds its windows built.  What
// a hit saves is the merge, 0.12-0.15 ms a window -- with one rank of eight holding four or five windows and two host threads, 0.35 ms
// that the GPU waited for.  A hit hands out a window that READS the cached lists (nothing is copied; the call's results and its slice
// of the table are its own) and is settled on the calling thread, no host thread is started for it; at most 128 windows are kept
// (~0.3 MB each), the cache is dropped whole when full.  GAUSS_WINDOW_CACHE=0: every window is built by the call that needs it.
// ------------------------------------------------------------------------------------------
struct LeanCacheEntry { std::shared_ptr<PackedPanel> pk; std::shared_ptr<const GwasCache> gw; std::shared_ptr<const LeanWindow> w; };
static std::mutex g_lw_mu;
static std::map<std::string, std::shared_ptr<LeanCacheEntry>> g_lw_cache;

static std::string lean_cache_base(const ChromSetup& cs, const PackedPanel* pk)
{
    std::string k;
    char buf[160];
    uint64_t cut;
    memcpy(&cut, &cs.a.af1_cutoff, 8);
    snprintf(buf, sizeof(buf), "%p|%p|%d|%d|%lld|%llx|", (const void*)pk, (const void*)cs.gw.get(), cs.kind, cs.a.chr, (long long)cs.a.wing_size,
             (unsigned long long)cut);
    k = buf;
    k += cs.a.study_pop; k += '|';
    for (size_t j = 0; j < cs.sel.size(); j++) {
        uint64_t wb = 0;
        if (j < cs.pop_wgt.size()) memcpy(&wb, &cs.pop_wgt[j], 8);
        snprintf(buf, sizeof(buf), "%d:%llx,", cs.sel[j], (unsigned long long)wb);
        k += buf;
    }
    return k;
}
// A hit: a window that owns nothing but the call's own state and reads the lists through `base` (host_internal.h: lean_src)
static std::unique_ptr<LeanWindow> lean_cache_get(const std::string& base, long long s, long long e)
{
    const std::string k = base + "|" + std::to_string(s) + "-" + std::to_string(e);
    std::shared_ptr<LeanCacheEntry> hit;
    { std::lock_guard<std::mutex> lock(g_lw_mu); auto it = g_lw_cache.find(k); if (it != g_lw_cache.end()) hit = it->second; }
    if (!hit) return nullptr;
    std::unique_ptr<LeanWindow> w(new LeanWindow());
    w->base = hit->w;
    w->start_bp = hit->w->start_bp; w->end_bp = hit->w->end_bp;
    w->n_head = hit->w->n_head; w->n_predm = hit->w->n_predm;
    return w;
}
static void lean_cache_put(const std::string& base, const LeanWindow& w, const std::shared_ptr<PackedPanel>& pk, const std::shared_ptr<const GwasCache>& gw)
{
    std::shared_ptr<LeanCacheEntry> e = std::make_shared<LeanCacheEntry>();
    e->pk = pk; e->gw = gw;                       // the key holds their addresses: they stay alive (and unique) while the entry does
    std::unique_ptr<LeanWindow> c = lean_window_clone(w);
    c->cs = nullptr;
    lean_table_count(*c);                         // which of its SNPs the table lists: once, here
    e->w = std::shared_ptr<const LeanWindow>(c.release());
    const std::string k = base + "|" + std::to_string(w.start_bp) + "-" + std::to_string(w.end_bp);
    std::lock_guard<std::mutex> lock(g_lw_mu);
    if (g_lw_cache.size() >= 128) g_lw_cache.clear();
    g_lw_cache[k] = std::move(e);
}

extern "C" {

static thread_local int tl_calls_in_flight = 1;       // > 1: this thread's call is one of several the genome driver keeps in flight

int gauss_host_impute_chromosome(gauss_ctx* ctx, int kind, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size,
                                 int64_t window_size, const char* study_pop, const char* const* pop_names,
                                 const double* pop_wgts, int n_pop_wgt, const char* input_file, const char* reference_index_file,
                                 const char* reference_data_file_in, const char* reference_pop_desc_file, double af1_cutoff,
                                 int rank, int world, int n_batches, gauss_table** out, gauss_chrom_stats* stats)
{
    if (!ctx || !out || !input_file || !reference_data_file_in || !reference_pop_desc_file) return herr("bad arguments");
    // the reference's own panel format is accepted: its packed form is made on first use and kept in the panel cache
    std::string packed_path;
    double t_autopack = 0;
    if (!PackedPanel::is_packed(reference_data_file_in)) {
        if (auto_pack_mode() == 0 || !reference_index_file)
            return herr("gauss_host_impute_chromosome needs a packed panel (gauss_host_pack_panel), or the text panel's index file "
                        "with GAUSS_AUTO_PACK not 0");
        const double t0 = now_s();
        std::string err;
        if (resolve_packed_panel(reference_index_file, reference_data_file_in, reference_pop_desc_file, true, packed_path, err) != 0)
            return herr("%s", err.c_str());
        t_autopack = now_s() - t0;
    } else packed_path = reference_data_file_in;
    const char* reference_data_file = packed_path.c_str();
    if (kind != GAUSS_KIND_DIST && kind != GAUSS_KIND_DISTMIX && kind != GAUSS_KIND_QCAT && kind != GAUSS_KIND_QCATMIX)
        return herr("gauss_host_impute_chromosome: kind must be dist, distmix, qcat or qcatmix");
    if (window_size < 1 || end_bp < start_bp || world < 1 || rank < 0 || rank >= world) return herr("bad window / rank arguments");
    const double t_begin = now_s() - t_autopack;
    const bool chrom_trace = host_trace("chrom");
    gauss_chrom_stats st;
    memset(&st, 0, sizeof(st));

    // ---- plan: windows, costs, owners (identical on every rank) ----
    std::string err;
    std::shared_ptr<PackedPanel> pk = open_packed_shared(reference_data_file, err);
    if (!pk) return herr("%s", err.c_str());
    const double t_opened = now_s();
    // First use of the panel: its rows start travelling NOW, before the study file is even parsed (2 ms for a chromosome's study) --
    // the upload below then finds the store under way.  (Only the default, background form; an error shows up at that later call.)
    void* dev_probe0 = nullptr;
    const bool first_use = !panel_is_resident(ctx, reference_data_file, &dev_probe0, false);
    const bool async_upload = env_flag("GAUSS_CHROM_ASYNC_UPLOAD", true);      // =0: in one go before the first batch
    int64_t early_uploaded = 0;
    if (first_use && async_upload && panel_make_resident(ctx, reference_data_file, &dev_probe0, &early_uploaded, true) != 0) early_uploaded = 0;
    // Whatever way this call ends, nobody may find the panel "resident" while rows are still on their way: every exit that does
    // not reach the wait at the end of the call (a study file that cannot be read, bad arguments to the planner, a failed batch)
    // waits for the background upload here -- or, if that failed, drops the half-made store.
    struct UploadGuard {
        gauss_ctx* ctx; const char* path; void* dev = nullptr; bool settled = false;
        ~UploadGuard()
        {
            if (!dev || settled) return;
            if (gauss_store_wait(ctx, dev, 0) != 0) {
                const std::string keep = gauss_host_last_error();
                gauss_host_panel_evict(ctx, path);
                herr("%s", keep.c_str());
            }
        }
    } upload_guard{ctx, reference_data_file};
    if (early_uploaded > 0) upload_guard.dev = dev_probe0;
    std::shared_ptr<const GwasCache> gw = load_gwas_cached(input_file, err);
    if (!gw) return herr("%s", err.c_str());
    const double t_study = now_s();
    std::vector<ChromWin> wins;
    plan_windows(wins, *gw, *pk, chr, start_bp, end_bp, wing_size, window_size);
    plan_owners(wins, *pk, world);
    std::vector<int> mine;
    for (size_t i = 0; i < wins.size(); i++) if (wins[i].owner == rank) mine.push_back((int)i);
    st.n_windows = (int)wins.size();
    st.n_windows_mine = (int)mine.size();
    // First use of the panel on this context: its rows travel to HBM while the batches compute (below), in panel order, and a
    // batch starts when the rows it names have landed -- so the early batches are smaller then (six batches, the first three
    // 0.3 / 0.5 / 0.8 of a share): the GPU starts on the first fifth of the rows and stays busy behind the upload.
    const bool auto_batches = n_batches < 1;
    // A share of a few windows whose call is one of SEVERAL in flight on this context (gauss_host_impute_genome: the other thread's
    // call keeps the GPU busy while this one's data layer runs) goes as ONE batch: a job of one or two windows has nothing to
    // hide its factorisation chain under, and the pipeline's overlap comes from the neighbouring call.
    if (n_batches < 1 && tl_calls_in_flight > 1 && mine.size() <= 8) n_batches = 1;
    // (A LEAD batch of one window -- so that the GPU starts after ONE window's data layer instead of a 0.3-share batch's -- was
    // measured in round 5 and is not built: the GPU started 0.8 ms earlier and the chromosome's span grew by 0.9 ms, a job of one
    // window has nothing to hide its factorisation chain under; 8-rank shares 7.4-8.4 ms against 7.8-8.2.  docs/HISTORY.md section 9e item 10.)
    const size_t lead = 0;
    const size_t n_rest = mine.size() - lead;
    // How many batches: a batch boundary costs ~0.35 ms of GPU span (the chip drains and fills), a batch in front of the GPU one
    // window's data layer and job tables (~0.2 ms a window since the driver builds its own windows).  Measured on the chr22 study,
    // warm (tools/e2e_rank_trace.py, ms per call with 1 / 2 / 3 / 4 batches): 4-5 windows (a rank of eight) 5.9 / 6.3 / - / -;
    // 9 windows 11.6 / 11.0 / 11.3 / 11.1; 18 windows 20.8 / 19.7 / 19.6 / 19.9; 36 windows - / 39.4 / 38.1 / 37.5 (six: 37.8).
    if (n_batches < 1)
        n_batches = (int)lead + (n_rest >= 28 ? (first_use && async_upload ? 6 : 4) : (n_rest >= 20 ? 3 : (n_rest >= 9 ? 2 : 1)));
    n_batches = std::max(1, std::min<int>(n_batches, std::max<size_t>(mine.size(), 1)));
    // contiguous batches by cost.  The first batch is the one nothing overlaps with on the way in (its data layer)
    // and the last one on the way out (its tables), so with four or more batches those two get 0.3 of a share.
    std::vector<std::vector<int>> batches((size_t)n_batches);
    {
        const int nb = n_batches - (int)lead;                      // batches behind the lead window
        std::vector<double> share((size_t)std::max(nb, 1), 1.0);
        if (nb >= 4) { share.front() = 0.3; share.back() = 0.3; }    // measured: 0.5 / 0.5 46.3 ms, 0.3 / 0.3 45.3 ms per chromosome
        if (auto_batches && first_use && nb == 6) { share[1] = 0.5; share[2] = 0.8; }
        double ssum = 0;
        for (double v : share) ssum += v;
        double total = 0;
        for (size_t q = lead; q < mine.size(); q++) total += wins[mine[q]].cost;
        double acc = 0, edge = share[0] / ssum;
        int b = 0;
        if (lead) batches[0].push_back(mine[0]);
        for (size_t q = lead; q < mine.size(); q++) {
            const int i = mine[q];
            while (b + 1 < nb && total > 0 && acc / total >= edge - 1e-12 && !batches[(size_t)b + lead].empty()) { b++; edge += share[b] / ssum; }
            batches[(size_t)b + lead].push_back(i);
            acc += wins[i].cost;
        }
    }
    st.n_batches = n_batches;
    st.t_plan = now_s() - t_begin;
    if (chrom_trace)
        fprintf(stderr, "[chrom] plan %.2f ms: panel opened %.2f, study file %.2f, windows + owners + batches %.2f\n", st.t_plan * 1e3,
                (t_opened - t_begin) * 1e3, (t_study - t_opened) * 1e3, (now_s() - t_study) * 1e3);

    // ---- feeder thread: the data layer, batch by batch ----
    int rc_upload = 0;
    struct Slot { std::unique_ptr<gauss_prepared> p; std::unique_ptr<LeanWindow> lw; gauss_window_desc d; bool ok = false; gauss_table* tab = nullptr; };
    // The windows of a sorted packed panel are built by the merge above (LeanWindow); an unsorted panel or a call over every
    // chromosome goes window by window through prepare(), and so does a call whose arguments prepare() would refuse (it then
    // gives every window its message).
    ChromSetup cs;
    const bool lean = chr > 0 && pk->header().sorted && !env_flag("GAUSS_HOST_FULL_MAP", false) &&
                      chrom_setup(cs, kind, chr, wing_size, study_pop, pop_names, pop_wgts, n_pop_wgt, input_file, packed_path,
                                  reference_pop_desc_file, af1_cutoff, pk, gw) == 0;
    const bool use_window_cache = lean && env_flag("GAUSS_WINDOW_CACHE", true);
    const std::string cache_base = use_window_cache ? lean_cache_base(cs, pk.get()) : std::string();
    std::vector<std::vector<Slot>> slots((size_t)n_batches);
    for (int b = 0; b < n_batches; b++) slots[b].resize(batches[b].size());
    std::mutex mu;
    std::condition_variable cv;
    const unsigned hw = std::thread::hardware_concurrency();
    // 8 rather than 16: a warm chromosome takes the same 45 ms, a process's first call 88 instead of 110 ms (the first
    // hipMalloc of the workspaces and the cold allocator share the host with these threads; tools/cold_probe.sh)
    // (the ranks of THIS node share its cores: torchrun exports LOCAL_WORLD_SIZE; `world` may span nodes)
    const char* lws = getenv("LOCAL_WORLD_SIZE");
    const unsigned local_ranks = (unsigned)std::max(1, lws ? atoi(lws) : 1);
    int nthreads = (int)std::max(1u, std::min(8u, (hw ? hw : 4u) / local_ranks));
    {
        // first use of the panel on this context: the upload's copy threads (page faults on the mapping, or preads) run
        // beside the data layer, and more than four data-layer threads slow BOTH down (cold call 73-84 ms with 4, 103-124
        // with 8, 115-132 with 16; the data layer of a chromosome is 36 x 2 ms, well hidden either way)
        if (first_use) nthreads = std::min(nthreads, 4);
    }
    // the result tables are built after the upload has finished: they keep the full count (a first call built its tables on the
    // four threads meant for the time of the upload: 20 ms of tables instead of 11)
    int nthreads_tables = (int)std::max(1u, std::min(8u, (hw ? hw : 4u) / local_ranks));
    // One pool over ALL windows in batch order (not one fork-join per batch: a batch of five windows would leave
    // eleven of sixteen threads idle); a batch is ready when its last window is.
    std::vector<std::pair<int, int>> order;                       // (batch, slot)
    for (int b = 0; b < n_batches; b++)
        for (int k = 0; k < (int)batches[b].size(); k++) order.push_back(std::make_pair(b, k));
    std::vector<int> left((size_t)n_batches);                     // windows of batch b still in the data layer (under mu)
    for (int b = 0; b < n_batches; b++) left[b] = (int)batches[b].size();
    auto batch_ready = [&](int b) { return left[b] == 0; };
    // one window through the data layer.  hit_only: take it from the window cache or leave it (returns false) -- the calling thread
    // settles the hits before any host thread exists
    auto process = [&](int q, bool hit_only) -> bool {
        const int b = order[q].first, k = order[q].second;
        ChromWin& w = wins[batches[b][k]];
        Slot& sl = slots[b][k];
        gauss_prepared* p = nullptr;
        if (lean) {
            std::unique_ptr<LeanWindow> lw = use_window_cache ? lean_cache_get(cache_base, w.s, w.e) : nullptr;
            if (!lw && hit_only) return false;
            bool built = true;
            if (lw) lw->cs = &cs;
            else {
                lw.reset(new LeanWindow());
                built = lean_window_build(*lw, cs, w.s, w.e) == 0;
                if (built && use_window_cache) lean_cache_put(cache_base, *lw, pk, gw);
            }
            if (!built) { w.status = 2; w.why = gauss_host_last_error(); }
            else {
                w.M = (int)lean_src(*lw).measured.size(); w.U = (int)lean_src(*lw).unmeasured.size();
                if (lean_window_desc(*lw, &sl.d)) { w.status = 1; w.why = gauss_host_last_error(); }      // the ">10" guards (dist.cpp:145-151)
                else { sl.lw = std::move(lw); sl.ok = true; }
            }
        } else if (hit_only) return false;
        else if (gauss_host_prepare(kind, chr, w.s, w.e, wing_size, study_pop, pop_names, pop_wgts, n_pop_wgt, input_file, nullptr,
                                    "(packed)", reference_data_file, reference_pop_desc_file, af1_cutoff, &p)) {
            w.status = 2; w.why = gauss_host_last_error();
        } else {
            sl.p.reset(p);
            w.M = (int)p->measured.size(); w.U = (int)p->unmeasured.size();
            if (gauss_prepared_window_desc(p, &sl.d)) {       // the ">10" guards (dist.cpp:145-151)
                w.status = 1; w.why = gauss_host_last_error();
                sl.p.reset();
            } else {
                sl.ok = true;                                 // geno_m / geno_u: the resident panel, set when the batch is queued
            }
        }
        bool last;
        { std::lock_guard<std::mutex> lock(mu); last = (--left[b] == 0); }
        if (last) cv.notify_all();
        if (last && chrom_trace) fprintf(stderr, "[chrom] data layer of batch %d done at %.2f ms\n", b, (now_s() - t_begin) * 1e3);
        return true;
    };
    std::vector<int> todo;                                      // the windows no cache entry settled
    for (int q = 0; q < (int)order.size(); q++)
        if (!(use_window_cache && process(q, true))) todo.push_back(q);
    std::thread feeder;
    if (!todo.empty()) feeder = std::thread([&]() { parallel_for((int)todo.size(), nthreads, [&](int i) { process(todo[(size_t)i], false); }); });

    // ---- the panel's rows in HBM ----
    // First use of this panel on this context: the upload was only STARTED above (gauss_store_upload_fd_async: staged through two
    // pinned buffers, 846 MB of a chromosome in ~20 ms) and every batch waits for the rows it names (gauss_store_wait) -- the rows
    // travel in panel order, the batches follow the chromosome, so batch 0 starts when the first fifth of the rows has landed.
    // Measured on the chr22 panel as the bench's first call (round 4, tools/cold_trace.sh): in one go before the first batch
    // (GAUSS_CHROM_ASYNC_UPLOAD=0) 63-64 ms, beside the batches 46-55 ms against 40.5 warm.  That only pays since the data layer
    // no longer fights the upload for the host (windows' SNP maps in pooled blocks, above: cold data layer 44 -> 11 ms; rounds 2
    // and 3 measured the asynchronous form slower, 105-155 ms, for that reason).  (Round 4 also measured the store reserved and
    // filled piece by piece one batch ahead of the GPU, and in two pieces: 75-88 ms, never ahead of the other forms, and gone.)
    void* d_rows = nullptr;
    const int64_t panel_row_bytes = pk->row_bytes();
    {
        const double tu = now_s();
        if ((!mine.empty() || early_uploaded > 0) && panel_make_resident(ctx, reference_data_file, &d_rows, &st.panel_bytes_uploaded, async_upload)) rc_upload = -1;
        st.panel_bytes_uploaded += early_uploaded;             // (started before the study file was parsed, above)
        if (!rc_upload && d_rows) upload_guard.dev = d_rows;   // (made by this call or by another one that is still uploading)
        st.t_panel_upload = now_s() - tu;
    }

    // ---- GPU pipeline ----
    std::vector<gauss_job*> jobs((size_t)n_batches, nullptr);
    std::vector<std::vector<int>> live((size_t)n_batches);       // slots of batch b that are in its job
    int rc_fatal = rc_upload;
    bool upload_failed = rc_upload != 0;
    // the result table grows batch by batch (batches are contiguous in window order), so that only the last batch's
    // rows are appended after the GPU has finished
    std::unique_ptr<gauss_table> all(new gauss_table());
    Column win_col{"window", GAUSS_COL_INT, {}, {}, {}};
    bool first = true, first_reserve = true;
    auto append_batch = [&](int b) {
        for (size_t k = 0; k < slots[b].size(); k++) {
            Slot& sl = slots[b][k];
            if (!sl.tab) continue;
            const int nr = sl.tab->nrow();
            if (first) {
                const size_t guess = (size_t)nr * (mine.size() + 1);
                for (const Column& c : sl.tab->cols) {
                    Column& nc = all->add(c.name.c_str(), c.type);
                    if (c.type == GAUSS_COL_STR) nc.s.reserve(guess);
                    else if (c.type == GAUSS_COL_INT) nc.i.reserve(guess);
                    else nc.d.reserve(guess);
                }
                win_col.i.reserve(guess);
                first = false;
            }
            for (size_t c = 0; c < sl.tab->cols.size(); c++) {
                Column& src = sl.tab->cols[c];
                Column& dst = all->cols[c];
                if (src.type == GAUSS_COL_STR) for (std::string& v : src.s) dst.s.push_back(std::move(v));
                else if (src.type == GAUSS_COL_INT) dst.i.insert(dst.i.end(), src.i.begin(), src.i.end());
                else dst.d.insert(dst.d.end(), src.d.begin(), src.d.end());
            }
            win_col.i.insert(win_col.i.end(), (size_t)nr, (int32_t)batches[b][k]);
            if (wins[batches[b][k]].status == 0) st.imputed += wins[batches[b][k]].U;
            delete sl.tab;
            sl.tab = nullptr;
        }
    };
    // The call's ONE table is laid out while the GPU works: a lean window's rows are a slice of `all`'s columns, written in place --
    // strings, positions, frequencies, the measured SNPs' z and p-values before the batch is waited for, the unmeasured SNPs' z /
    // info / pval after (lean_table_prebuild_into / lean_window_finish_into).  No table per window, no append: what follows the
    // GPU's last result is one pass over the unmeasured SNPs (a rank of eight: 0.26 -> 0.1 ms).
    auto ensure_columns = [&]() {
        if (!first) return;
        const bool mix = (kind == GAUSS_KIND_DISTMIX || kind == GAUSS_KIND_QCATMIX);
        const bool qc = (kind == GAUSS_KIND_QCAT || kind == GAUSS_KIND_QCATMIX);
        all->add("rsid", GAUSS_COL_STR); all->add("chr", GAUSS_COL_INT); all->add("bp", GAUSS_COL_INT);
        all->add("a1", GAUSS_COL_STR); all->add("a2", GAUSS_COL_STR); all->add(mix ? "af1mix" : "af1ref", GAUSS_COL_DBL);
        all->add("z", GAUSS_COL_DBL);
        if (qc) { all->add("qcat_m", GAUSS_COL_INT); all->add("qcat_t", GAUSS_COL_DBL); all->add("qcat_chisq", GAUSS_COL_DBL); all->add("qcat_pval", GAUSS_COL_DBL); }
        else { all->add("pval", GAUSS_COL_DBL); all->add("info", GAUSS_COL_DBL); }
        all->add("type", GAUSS_COL_INT);
        first = false;
    };
    size_t all_rows = 0;
    std::vector<std::pair<size_t, size_t>> dead;                  // slices of windows that failed after their rows were laid out
    auto resize_all = [&](size_t n) {
        for (Column& c : all->cols) {
            if (c.type == GAUSS_COL_STR) c.s.resize(n);
            else if (c.type == GAUSS_COL_INT) c.i.resize(n);
            else c.d.resize(n);
        }
        win_col.i.resize(n);
    };
    auto retire = [&](int b) {
        // results of batch b -> the table (host threads; the GPU is on batch b+1 meanwhile)
        // (what the table holds that the results do not change is built BEFORE the wait: the last batch has no batch b+1 to hide under)
        double tp = now_s();
        const bool in_place = lean;
        if (in_place) {
            ensure_columns();
            size_t off = all_rows;
            for (size_t k = 0; k < slots[b].size(); k++) {
                Slot& sl = slots[b][k];
                if (!sl.ok || !sl.lw) continue;
                sl.lw->tab_off = off;
                off += (size_t)lean_table_count(*sl.lw);
            }
            if (first_reserve && off > 0) {                        // room for the whole share (batches are of similar size)
                const size_t guess = off * (size_t)n_batches + 64;
                for (Column& c : all->cols) {
                    if (c.type == GAUSS_COL_STR) c.s.reserve(guess);
                    else if (c.type == GAUSS_COL_INT) c.i.reserve(guess);
                    else c.d.reserve(guess);
                }
                win_col.i.reserve(guess);
                first_reserve = false;
            }
            resize_all(off);
            parallel_for((int)slots[b].size(), nthreads_tables, [&](int k) {
                Slot& sl = slots[b][k];
                if (!sl.ok || !sl.lw) return;
                lean_table_prebuild_into(*sl.lw, *all, sl.lw->tab_off);
                std::fill(win_col.i.begin() + (ptrdiff_t)sl.lw->tab_off, win_col.i.begin() + (ptrdiff_t)(sl.lw->tab_off + (size_t)sl.lw->n_out), (int32_t)batches[b][k]);
            });
            all_rows = off;
            if (b == n_batches - 1) {
                // every row of the table is laid out now: the fixed-width images of the string columns that the binding reads
                // (gauss_table_strcol_fixed) are made here, under the last batch's GPU time, not in the caller's time after the call
                parallel_for((int)all->cols.size(), nthreads_tables, [&](int c) { if (all->cols[(size_t)c].type == GAUSS_COL_STR) build_fixed_image(all->cols[(size_t)c]); });
            }
        } else
            parallel_for((int)slots[b].size(), nthreads_tables, [&](int k) { if (slots[b][k].ok && slots[b][k].lw) lean_table_prebuild(*slots[b][k].lw); });
        st.t_tables += now_s() - tp;
        double tw = now_s();
        int rc = jobs[b] ? gauss_job_fetch(jobs[b]) : 0;
        st.t_gpu_wait += now_s() - tw;
        if (rc != 0 && jobs[b]) {
            // the batch failed as a whole: run its windows one by one so that only the culprit is lost
            const std::string why = gauss_last_error();
            for (int k : live[b]) {
                Slot& sl = slots[b][k];
                gauss_job* one = nullptr;
                if (gauss_job_create(ctx, &sl.d, 1, 1, &one) != 0 || gauss_job_run(one) != 0 || gauss_job_fetch(one) != 0) {
                    ChromWin& w = wins[batches[b][k]];
                    w.status = 2; w.why = std::string(gauss_last_error()) + " (batch error: " + why + ")";
                    sl.ok = false;
                    if (in_place && sl.lw && sl.lw->n_out > 0) dead.emplace_back(sl.lw->tab_off, sl.lw->tab_off + (size_t)sl.lw->n_out);
                }
                if (one) gauss_job_destroy(one);
            }
        }
        double tt = now_s();
        struct rusage ru0;
        getrusage(RUSAGE_SELF, &ru0);
        parallel_for((int)slots[b].size(), nthreads_tables, [&](int k) {
            Slot& sl = slots[b][k];
            if (!sl.ok) return;
            gauss_table* t = nullptr;
            if (sl.lw && in_place) { lean_window_finish_into(*sl.lw, *all); wins[batches[b][k]].status = 0; }
            else if (sl.lw) { sl.tab = lean_window_finish(*sl.lw); wins[batches[b][k]].status = 0; }
            else if (gauss_prepared_finish(sl.p.get(), &t) == 0) { sl.tab = t; wins[batches[b][k]].status = 0; }
            else { wins[batches[b][k]].status = 2; wins[batches[b][k]].why = gauss_host_last_error(); }
            sl.p.reset();
        });
        const double t_fin = now_s();
        if (in_place) {
            for (size_t k = 0; k < slots[b].size(); k++) {
                Slot& sl = slots[b][k];
                if (sl.ok && sl.lw && wins[batches[b][k]].status == 0) st.imputed += wins[batches[b][k]].U;
                sl.lw.reset();
            }
        } else append_batch(b);
        st.t_tables += now_s() - tt;
        if (b == n_batches - 1) st.t_tables_tail = now_s() - tt;
        if (chrom_trace) {
            struct rusage ru1;
            getrusage(RUSAGE_SELF, &ru1);
            fprintf(stderr, "[chrom] retire batch %d: tables %.2f ms (finish %.2f, append %.2f), %ld minor faults, %d windows\n", b, (now_s() - tt) * 1e3,
                    (t_fin - tt) * 1e3, (now_s() - t_fin) * 1e3, ru1.ru_minflt - ru0.ru_minflt, (int)slots[b].size());
        }
    };
    for (int b = 0; b < n_batches && !rc_fatal; b++) {
        double tw = now_s();
        { std::unique_lock<std::mutex> lock(mu); cv.wait(lock, [&]() { return batch_ready(b); }); }
        st.t_feeder_wait += now_s() - tw;
        std::vector<gauss_window_desc> descs;
        for (size_t k = 0; k < slots[b].size(); k++)
            if (slots[b][k].ok) {
                slots[b][k].d.geno_m = slots[b][k].d.geno_u = (const uint8_t*)d_rows;   // rows_m / rows_u are panel row indices already
                descs.push_back(slots[b][k].d); live[b].push_back((int)k);
            }
        if (!descs.empty()) {
            {
                // the highest panel row this batch reads; its job may start once the upload has passed it
                int64_t top = -1;
                for (int k : live[b]) {
                    const Slot& sq = slots[b][(size_t)k];
                    for (int32_t r : sq.lw ? lean_src(*sq.lw).store_rows_m : sq.p->store_rows_m) top = std::max<int64_t>(top, r);
                    for (int32_t r : sq.lw ? lean_src(*sq.lw).store_rows_u : sq.p->store_rows_u) top = std::max<int64_t>(top, r);
                }
                const double tu = now_s();
                if (gauss_store_wait(ctx, d_rows, (top + 1) * panel_row_bytes) != 0) { herr("%s", gauss_last_error()); rc_fatal = -1; upload_failed = true; break; }
                st.t_panel_upload += now_s() - tu;
            }
            double tc = now_s();
            double t_created = 0;
            if (gauss_job_create(ctx, descs.data(), (int)descs.size(), 1, &jobs[b]) != 0 || ((t_created = now_s()), gauss_job_run(jobs[b])) != 0) {
                // could not even queue the batch: fall back to single windows at retire time
                if (jobs[b]) { gauss_job_destroy(jobs[b]); jobs[b] = nullptr; }
                for (int k : live[b]) {
                    Slot& sl = slots[b][k];
                    gauss_job* one = nullptr;
                    if (gauss_job_create(ctx, &sl.d, 1, 1, &one) != 0 || gauss_job_run(one) != 0 || gauss_job_fetch(one) != 0) {
                        ChromWin& w = wins[batches[b][k]];
                        w.status = 2; w.why = gauss_last_error();
                        sl.ok = false;
                    }
                    if (one) gauss_job_destroy(one);
                }
            }
            st.t_job_create += now_s() - tc;
            if (chrom_trace) fprintf(stderr, "[chrom] batch %d queued at %.2f ms (job create %.2f ms, run %.2f ms)\n", b, (now_s() - t_begin) * 1e3, (t_created - tc) * 1e3, (now_s() - t_created) * 1e3);
        }
        if (b > 0) retire(b - 1);
    }
    if (!rc_fatal && n_batches > 0) retire(n_batches - 1);
    if (feeder.joinable()) feeder.join();
    {
        gauss_job *jf = nullptr, *jl = nullptr;
        for (gauss_job* j : jobs) if (j) { if (!jf) jf = j; jl = j; }
        if (jf && gauss_job_span_ms(jf, jl, &st.gpu_span_ms) != 0) st.gpu_span_ms = 0.0;
        if (jf && chrom_trace)
            for (size_t b = 0; b < jobs.size(); b++) {
                double to_end = 0, own = 0;
                if (jobs[b] && gauss_job_span_ms(jf, jobs[b], &to_end) == 0 && gauss_job_span_ms(jobs[b], jobs[b], &own) == 0)
                    fprintf(stderr, "[chrom] batch %zu on the GPU: starts %.2f ms after the first batch, runs %.2f ms\n", b, to_end - own, own);
            }
    }
    // (the call's own jobs, not the context's counters: another call may be in flight on this context)
    for (gauss_job* j : jobs) {
        int64_t c4[4] = {0, 0, 0, 0};
        if (j && gauss_job_counters(j, c4) == 0) st.n_merged_giveups += (int32_t)c4[2];
    }
    for (gauss_job* j : jobs) if (j) gauss_job_destroy(j);
    // whoever finds this panel resident later (another study, an LD call on rows this chromosome never touched) must
    // find all of it: the upload is complete before the call returns
    if (d_rows && !upload_failed && gauss_store_wait(ctx, d_rows, 0) != 0) { if (!rc_fatal) herr("%s", gauss_last_error()); rc_fatal = -1; upload_failed = true; }
    upload_guard.settled = true;                                // waited for (or failed and evicted below)
    if (rc_fatal) {
        // tables of windows that retired before the failure
        for (auto& bs : slots) for (Slot& sl : bs) { delete sl.tab; sl.tab = nullptr; }
        if (upload_failed) {
            // a failed background upload must not poison the context: without this every later call on this panel would find
            // the half-made store "resident" and fail on the stored error until gauss_host_panel_evict.  The message survives.
            const std::string keep = gauss_host_last_error();
            gauss_host_panel_evict(ctx, reference_data_file);
            herr("%s", keep.c_str());
        }
        return -1;
    }

    // ---- one table, window order (batches were appended as they retired) ----
    double tt = now_s();
    ensure_columns();      // (no window produced rows: still the reference's column set)
    if (!dead.empty()) {
        // a window failed after its rows had been laid out (its batch failed as a whole and its own re-run failed too): close the gaps
        std::sort(dead.begin(), dead.end());
        std::vector<char> keep(all_rows, 1);
        for (auto& d : dead) for (size_t r = d.first; r < d.second && r < all_rows; r++) keep[r] = 0;
        auto compact = [&](Column& c) {
            size_t o = 0;
            for (size_t r = 0; r < all_rows; r++) {
                if (!keep[r]) continue;
                if (c.type == GAUSS_COL_STR) { if (o != r) c.s[o] = std::move(c.s[r]); }
                else if (c.type == GAUSS_COL_INT) c.i[o] = c.i[r];
                else c.d[o] = c.d[r];
                o++;
            }
            if (c.type == GAUSS_COL_STR) c.s.resize(o); else if (c.type == GAUSS_COL_INT) c.i.resize(o); else c.d.resize(o);
        };
        for (Column& c : all->cols) compact(c);
        compact(win_col);
    }
    all->cols.push_back(win_col);
    {
        NamedMat nm;
        nm.name = "windows"; nm.nrow = (int)wins.size(); nm.ncol = 6;
        nm.d.assign((size_t)nm.nrow * 6, 0.0);
        for (int i = 0; i < nm.nrow; i++) {
            const ChromWin& w = wins[i];
            const double v[6] = {(double)w.s, (double)w.e, (double)w.owner, (double)w.status, (double)w.M, (double)w.U};
            for (int c = 0; c < 6; c++) nm.d[(size_t)c * nm.nrow + i] = v[c];
            if (w.status == 1) st.n_skipped++;
            if (w.status == 2) { st.n_failed++; if (all->messages.size() < 64) all->messages.push_back("window " + std::to_string(i) + ": " + w.why); }
        }
        all->named.push_back(nm);
    }
    st.t_tables += now_s() - tt;
    st.t_total = now_s() - t_begin;
    if (stats) *stats = st;
    *out = all.release();
    return 0;
}

// ------------------------------------------------------------------------------------------
// A genome: the loop over chromosomes above the loop over windows.  One rank's share of ONE chromosome is a few
// milliseconds of GPU work behind ~2.5 ms of host work that nothing overlaps (plan, the first batch's data layer and job, the
// last batch's tables): at 8 ranks the host part is a third of the call (bench.py end_to_end.emulated_world8: 7.8-8.6 ms per
// rank for 5.3 ms of GPU span).  Here `depth` host threads (default 2) take the chromosomes in order, each running
// gauss_host_impute_chromosome on the SAME context: while one call's data layer runs, the other call's batches keep the GPU busy
// (jobs of different calls simply follow each other on the context's queues; the library's job and row-store calls are
// thread-safe per context).  Measured on the chr22 study, rank r of 8 (tools/e2e_pipeline_probe.py): 8.0-8.3 ms per call one
// after the other, 5.7 ms with two in flight, for 5.3-5.4 ms of GPU span -- the host latency is hidden.
// Tables and stats come back per chromosome; a chromosome that fails leaves out[c] = NULL and its message in the returned error
// (the first failure's), the others still complete.
// ------------------------------------------------------------------------------------------
int gauss_host_impute_genome(gauss_ctx* ctx, int kind, int n_chrom, const int32_t* chr, const int64_t* start_bp, const int64_t* end_bp,
                             int64_t wing_size, int64_t window_size, const char* study_pop, const char* const* pop_names,
                             const double* pop_wgts, int n_pop_wgt, const char* input_file, const char* reference_index_file,
                             const char* reference_data_file, const char* reference_pop_desc_file, double af1_cutoff,
                             int rank, int world, int depth, gauss_table** out, gauss_chrom_stats* stats)
{
    if (!ctx || n_chrom < 1 || !chr || !start_bp || !end_bp || !out) return herr("bad arguments to gauss_host_impute_genome");
    depth = std::max(1, std::min(depth <= 0 ? 2 : depth, std::min(n_chrom, 4)));
    for (int c = 0; c < n_chrom; c++) out[c] = nullptr;
    std::atomic<int> next{0};
    std::mutex emu;
    std::string first_err;
    int n_failed = 0;
    auto work = [&]() {
        tl_calls_in_flight = depth;
        for (int c = next.fetch_add(1); c < n_chrom; c = next.fetch_add(1)) {
            gauss_chrom_stats st;
            memset(&st, 0, sizeof(st));
            const int rc = gauss_host_impute_chromosome(ctx, kind, chr[c], start_bp[c], end_bp[c], wing_size, window_size, study_pop, pop_names,
                                                        pop_wgts, n_pop_wgt, input_file, reference_index_file, reference_data_file,
                                                        reference_pop_desc_file, af1_cutoff, rank, world, 0, &out[c], &st);
            if (stats) stats[c] = st;
            if (rc != 0) {
                std::lock_guard<std::mutex> lock(emu);
                n_failed++;
                if (first_err.empty()) first_err = "chromosome " + std::to_string(chr[c]) + " (call " + std::to_string(c) + "): " + gauss_host_last_error();
                out[c] = nullptr;
            }
        }
        tl_calls_in_flight = 1;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < depth; t++) th.emplace_back(work);
    work();
    for (std::thread& x : th) x.join();
    if (n_failed) return herr("%s%s", first_err.c_str(), n_failed > 1 ? " (and more)" : "");
    return 0;
}


}  // extern "C"
