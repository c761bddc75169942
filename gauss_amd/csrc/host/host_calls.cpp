// libgauss_host.so -- the one-call entry points of include/gauss_host.h: the drivers computeLD.cpp:26-166, dist.cpp:30-126,
// distmix.cpp:30-135, qcat.cpp / qcatmix.cpp, prep_qcat.cpp, zmix.cpp:201-1076, jepeg.cpp:28-153, jepegmix.cpp:26-161 with the numeric
// hot path delegated to libgauss_hip.so.
#include "host_internal.h"

// Row store of an LD-only call on a packed panel: the resident copy in HBM when there is one (or the panel is small
// enough to make resident on the spot: a chromosome is < 1 GB), else the mmap'd rows, gathered while staging.
static int packed_row_source(gauss_ctx* ctx, const gauss_prepared& p, const uint8_t** store, int* on_device)
{
    const PackedPanel& pk = *p.args.pk;
    void* dev = nullptr;
    const int64_t bytes = pk.n_snp() * pk.row_bytes();
    if (panel_is_resident(ctx, p.args.reference_data_file, &dev) ||
        (bytes <= ((int64_t)4 << 30) && panel_make_resident(ctx, p.args.reference_data_file, &dev, nullptr) == 0)) {
        *store = (const uint8_t*)dev; *on_device = 1;
        return 0;
    }
    *store = pk.geno(); *on_device = 0;
    return 0;
}

extern "C" {

static int host_prepare(int kind, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size, const char* study_pop,
                        const char* const* pop_names, const double* pop_wgts, int n_pop_wgt, const char* input_file,
                        const char* annotation_file, const char* reference_index_file, const char* reference_data_file,
                        const char* reference_pop_desc_file, double af1_cutoff, bool annotated_only, gauss_prepared** out);

int gauss_host_prepare(int kind, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size, const char* study_pop,
                       const char* const* pop_names, const double* pop_wgts, int n_pop_wgt, const char* input_file,
                       const char* annotation_file, const char* reference_index_file, const char* reference_data_file,
                       const char* reference_pop_desc_file, double af1_cutoff, gauss_prepared** out)
{
    // (the whole study enters the SNP map, as in the reference: gauss_prepared_snps lists it)
    return host_prepare(kind, chr, start_bp, end_bp, wing_size, study_pop, pop_names, pop_wgts, n_pop_wgt, input_file, annotation_file,
                        reference_index_file, reference_data_file, reference_pop_desc_file, af1_cutoff, false, out);
}

static int host_prepare(int kind, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size, const char* study_pop,
                        const char* const* pop_names, const double* pop_wgts, int n_pop_wgt, const char* input_file,
                        const char* annotation_file, const char* reference_index_file, const char* reference_data_file,
                        const char* reference_pop_desc_file, double af1_cutoff, bool annotated_only, gauss_prepared** out)
{
    if (!out) return herr("out is NULL");
    if (kind < 0 || kind > GAUSS_KIND_PREP_RECESSIVE) return herr("bad kind %d", kind);
    if (!input_file || !reference_index_file || !reference_data_file || !reference_pop_desc_file) return herr("file name is NULL");
    std::unique_ptr<gauss_prepared> p(new gauss_prepared());
    p->kind = kind;
    Args& a = p->args;
    a.chr = chr; a.start_bp = start_bp; a.end_bp = end_bp;
    a.wing_size = (kind == GAUSS_KIND_COMPUTELD) ? 0 : wing_size;       // computeLD.cpp:40
    if (study_pop) a.study_pop = study_pop;
    a.input_file = input_file; a.reference_index_file = reference_index_file;
    a.reference_data_file = reference_data_file; a.reference_pop_desc_file = reference_pop_desc_file;
    if (annotation_file) a.annotation_file = annotation_file;
    if (auto_pack_mode() != 0 && !PackedPanel::is_packed(a.reference_data_file)) {
        // text panel: the cached packed form, if there is one (GAUSS_AUTO_PACK=1: made now)
        std::string cached, err;
        const int rc = resolve_packed_panel(a.reference_index_file, a.reference_data_file, a.reference_pop_desc_file,
                                            auto_pack_mode() == 1, cached, err);
        if (rc < 0) return herr("%s", err.c_str());
        if (rc == 0) a.reference_data_file = cached;
    }
    if (PackedPanel::is_packed(a.reference_data_file)) {
        // a packed panel replaces both the index and the data file (reference_index_file is not opened)
        std::string err;
        a.pk = open_packed_shared(a.reference_data_file, err);
        if (!a.pk) return herr("%s", err.c_str());
        a.drop_wing_unmeasured = (kind == GAUSS_KIND_DIST || kind == GAUSS_KIND_DISTMIX);
    }
    a.af1_cutoff = std::isnan(af1_cutoff) ? (kind == GAUSS_KIND_QCAT ? 0.05 : 0.01) : af1_cutoff;   // dist.cpp:53-57, qcat.cpp:53-57
    const bool mix = (kind == GAUSS_KIND_COMPUTELD || kind == GAUSS_KIND_DISTMIX || kind == GAUSS_KIND_JEPEGMIX ||
                      kind == GAUSS_KIND_QCATMIX || kind == GAUSS_KIND_PREP_RECESSIVE);
    if (mix) {
        if (!pop_names || !pop_wgts || n_pop_wgt < 1) return herr("pop_wgt_df is empty");
        set_pop_wgt_map(a, pop_names, pop_wgts, n_pop_wgt);
    } else if (!study_pop) return herr("study_pop is NULL");
    if ((kind == GAUSS_KIND_JEPEG || kind == GAUSS_KIND_JEPEGMIX) && !annotation_file) return herr("annotation_file is NULL");
    a.annotated_only = annotated_only && (kind == GAUSS_KIND_JEPEG || kind == GAUSS_KIND_JEPEGMIX);
    if (prepare(*p)) return -1;
    *out = p.release();
    return 0;
}

const gauss_table* gauss_prepared_snps(const gauss_prepared* p)
{
    if (!p) return nullptr;
    if (!p->snps_built) build_snp_table(*const_cast<gauss_prepared*>(p));      // a debugging / test view: built on demand
    return &p->snps;
}
int gauss_prepared_counts(const gauss_prepared* p, int* m, int* u, int* n, int* np, int* ng)
{
    if (!p) return herr("prepared is NULL");
    if (m) *m = (int)p->measured.size();
    if (u) *u = (int)p->unmeasured.size();
    if (n) *n = p->N;
    if (np) *np = (int)p->pop_off.size() - 1;
    if (ng) *ng = p->gene_off.empty() ? 0 : (int)p->gene_off.size() - 1;
    return 0;
}
const int32_t* gauss_prepared_measured_rows(const gauss_prepared* p) { return p ? p->measured_rows.data() : nullptr; }
const int32_t* gauss_prepared_unmeasured_rows(const gauss_prepared* p) { return p ? p->unmeasured_rows.data() : nullptr; }
static void ensure_bytes(const gauss_prepared* cp)
{
    gauss_prepared* p = const_cast<gauss_prepared*>(cp);     // lazily unpacked view of packed rows (tests, debugging)
    if (p->packed_rows && p->gm.empty()) materialise_from_packed(*p);
}
const uint8_t* gauss_prepared_geno_m(const gauss_prepared* p, int64_t* ld) { if (!p) return nullptr; ensure_bytes(p); if (ld) *ld = p->ld; return p->gm.data(); }
const uint8_t* gauss_prepared_geno_u(const gauss_prepared* p, int64_t* ld) { if (!p) return nullptr; ensure_bytes(p); if (ld) *ld = p->ld; return p->gu.data(); }
int gauss_prepared_packed_store(const gauss_prepared* p, const uint8_t** base, int64_t* bytes, int64_t* row_bytes)
{
    if (!p) return herr("prepared is NULL");
    if (!p->packed_rows) { if (base) *base = nullptr; if (bytes) *bytes = 0; if (row_bytes) *row_bytes = 0; return 0; }
    const PackedPanel& pk = *p->args.pk;
    if (base) *base = pk.geno();
    if (bytes) *bytes = pk.n_snp() * pk.row_bytes();
    if (row_bytes) *row_bytes = pk.row_bytes();
    return 0;
}
const int32_t* gauss_prepared_pop_off(const gauss_prepared* p) { return p ? p->pop_off.data() : nullptr; }
const double* gauss_prepared_pop_wgt(const gauss_prepared* p) { return p ? p->pop_wgt.data() : nullptr; }
const double* gauss_prepared_z1(const gauss_prepared* p) { return p ? p->z1.data() : nullptr; }
const int32_t* gauss_prepared_gene_off(const gauss_prepared* p) { return (p && !p->gene_off.empty()) ? p->gene_off.data() : nullptr; }
void gauss_prepared_free(gauss_prepared* p) { delete p; }

int gauss_prepared_qcat_counts(const gauss_prepared* p, int* n_head, int* n_predm)
{
    if (!p) return herr("prepared is NULL");
    if (n_head) *n_head = p->n_head;
    if (n_predm) *n_predm = p->n_predm;
    return 0;
}

// genotype source of a window: host byte matrices, or row lists into the mmap'd packed panel
static void set_geno(gauss_prepared* p, gauss_window_desc* d)
{
    if (!p->packed_rows) { d->geno_m = p->gm.data(); d->geno_u = p->gu.data(); d->ld = p->ld; return; }
    const PackedPanel& pk = *p->args.pk;
    d->geno_format = GAUSS_GENO_2BIT;
    d->geno_m = d->geno_u = pk.geno();
    d->ld = pk.row_bytes();
    d->rows_m = p->store_rows_m.data();
    d->rows_u = p->store_rows_u.data();
    d->pop_src_off = p->pop_src_off.data();
}

int gauss_prepared_window_desc(gauss_prepared* p, gauss_window_desc* d)
{
    if (!p || !d) return herr("bad arguments");
    const bool qcat = (p->kind == GAUSS_KIND_QCAT || p->kind == GAUSS_KIND_QCATMIX);
    const bool prep = (p->kind == GAUSS_KIND_PREP_QCAT || p->kind == GAUSS_KIND_PREP_RECESSIVE);
    if (p->kind != GAUSS_KIND_DIST && p->kind != GAUSS_KIND_DISTMIX && !qcat && !prep) return herr("not a window kind");
    const Args& a = p->args;
    const int M = (int)p->measured.size(), U = (int)p->unmeasured.size();
    if (prep) {
        if (M <= a.min_num_measured_snp)                       // prep_qcat.cpp:86-91, prep_qcatmix.cpp:126-128
            return herr("Not enough number of SNPs loaded - %s not performed (measured %d, prediction window %d)",
                        p->kind == GAUSS_KIND_PREP_QCAT ? "QCAT" : "Recessive Imputation", M, U);
        const int ncode = (p->kind == GAUSS_KIND_PREP_RECESSIVE) ? 3 : 1;
        p->out_b11.assign((size_t)M * M, 0.0);
        p->out_b21.assign((size_t)std::max(1, ncode * U) * M, 0.0);
        memset(d, 0, sizeof(*d));
        d->kind = GAUSS_WIN_LD;
        d->mode = (p->kind == GAUSS_KIND_PREP_QCAT) ? GAUSS_MODE_POOLED : GAUSS_MODE_WEIGHTED;
        d->n_pop = (int)p->pop_off.size() - 1;
        d->pop_off = p->pop_off.data(); d->pop_wgt = p->pop_wgt.data();
        d->n_measured = M; d->n_unmeasured = U;
        set_geno(p, d);
        d->lambda = 0.0;                                       // B11(i,i) = 1.0 (prep_qcat.cpp:109)
        d->u_codings = (ncode == 3) ? (GAUSS_CODE_ADDITIVE | GAUSS_CODE_DOMINANT | GAUSS_CODE_RECESSIVE) : GAUSS_CODE_ADDITIVE;
        d->out_b11 = p->out_b11.data(); d->out_b21 = p->out_b21.data(); d->out_status = &p->status;
        return 0;
    }
    if (qcat) {
        // qcat.cpp:157-162 guards on the measured count only; qcatmix.cpp:168-174 on both (texts as in the reference)
        if (p->kind == GAUSS_KIND_QCAT && M <= a.min_num_measured_snp)
            return herr("Not enough number of SNPs loaded - QCAT not performed (measured %d, unmeasured %d)", M, U);
        if (p->kind == GAUSS_KIND_QCATMIX && (M <= a.min_num_measured_snp || U <= a.min_num_unmeasured_snp))
            return herr("Not enough number of SNPs loaded - QCAT performed (measured %d, unmeasured %d)", M, U);
        p->out_r.assign((size_t)p->n_predm + U, 0.0);
        p->num_eig = M;
        memset(d, 0, sizeof(*d));
        d->kind = GAUSS_WIN_QCAT;
        d->mode = (p->kind == GAUSS_KIND_QCAT) ? GAUSS_MODE_POOLED : GAUSS_MODE_WEIGHTED;
        d->n_pop = (int)p->pop_off.size() - 1;
        d->pop_off = p->pop_off.data(); d->pop_wgt = p->pop_wgt.data();
        d->n_measured = M; d->n_unmeasured = U;
        set_geno(p, d);
        d->z1 = p->z1.data(); d->lambda = a.lambda; d->min_abs_eig = a.min_abs_eig;
        d->n_head_measured = p->n_head; d->n_pred_measured = p->n_predm; d->eig_cutoff = a.eig_cutoff;
        d->out_r = p->out_r.data(); d->out_num_eig = &p->num_eig; d->out_status = &p->status;
        if (p->n_predm + U < 1) return herr("QCAT window has no SNP to test");
        return 0;
    }
    if (M <= a.min_num_measured_snp || U <= a.min_num_unmeasured_snp)      // dist.cpp:145-151
        return herr("Not enough number of SNPs loaded - %s not performed (measured %d, unmeasured %d)",
                    p->kind == GAUSS_KIND_DIST ? "DIST" : "DISTMIX", M, U);
    p->out_z.assign(U, 0.0); p->out_info.assign(U, 0.0);
    memset(d, 0, sizeof(*d));
    d->mode = (p->kind == GAUSS_KIND_DIST) ? GAUSS_MODE_POOLED : GAUSS_MODE_WEIGHTED;
    d->n_pop = (int)p->pop_off.size() - 1;
    d->pop_off = p->pop_off.data(); d->pop_wgt = p->pop_wgt.data();
    d->n_measured = M; d->n_unmeasured = U;
    set_geno(p, d);
    d->z1 = p->z1.data(); d->lambda = a.lambda; d->min_abs_eig = a.min_abs_eig;
    d->out_z = p->out_z.data(); d->out_info = p->out_info.data(); d->out_status = &p->status;
    return 0;
}

int gauss_prepared_finish(gauss_prepared* p, gauss_table** out)
{
    if (!p || !out) return herr("bad arguments");
    if (p->kind == GAUSS_KIND_PREP_QCAT || p->kind == GAUSS_KIND_PREP_RECESSIVE) { *out = prep_output(*p); return 0; }
    if (p->kind == GAUSS_KIND_QCAT || p->kind == GAUSS_KIND_QCATMIX) {
        const int m = p->num_eig;
        for (size_t t = 0; t < p->out_r.size(); t++) {                           // qcat.cpp:216-243
            Snp* s = (t < (size_t)p->n_predm) ? p->measured[p->n_head + t] : p->unmeasured[t - p->n_predm];
            const double r = p->out_r[t];
            s->qcat_m = m;
            s->qcat_t = std::sqrt((double)(m - 3)) * r;
            s->qcat_chisq = (m - 3) * r * r;
        }
        *out = qcat_output(*p);
        return 0;
    }
    for (size_t i = 0; i < p->unmeasured.size() && i < p->out_z.size(); i++) {   // dist.cpp:200-202
        p->unmeasured[i]->z = p->out_z[i];
        p->unmeasured[i]->info = p->out_info[i];
    }
    *out = dist_output(*p);
    return 0;
}

// dist() / distmix() / qcat() / qcatmix(): ONE window per call, the reference's own usage (docs/articles/dist_example.md:144-153).
//
// On a sorted packed panel the window is built as the chromosome driver builds its windows (host_chrom.cpp:LeanWindow -- a merge of
// the study's rows and the panel's SNP table instead of per-SNP objects in a map: ~0.1 ms against 0.7-1.0), its genotype rows are
// read from the panel's resident copy in HBM (made on the first call when the panel's genotype section is at most 4 GB -- a
// chromosome of the 33KG panel is 0.85 GB -- like the LD-only calls do, packed_row_source) instead of being gathered on the host and
// copied per call (24 MB a window), and the window runs as a job of one on those rows.  Measured on the chr22 study, window after
// window (tools/window_calls_probe.py): 5.9 ms per call -> 2.5.  An unsorted panel, a text panel without a cached packed form, a
// call over every chromosome or GAUSS_HOST_FULL_MAP=1 take the literal path: gauss_host_prepare + gauss_impute_window on host rows.
static int run_impute(gauss_ctx* ctx, int kind, int chr, int64_t start_bp, int64_t end_bp, int64_t wing, const char* study_pop,
                      const char* const* names, const double* wgts, int nw, const char* input, const char* index,
                      const char* data, const char* desc, double af1_cutoff, gauss_table** out)
{
    if (!ctx || !out) return herr("bad arguments");
    if (!input || !index || !data || !desc) return herr("file name is NULL");
    // the packed form of the panel, if there is one (as gauss_host_prepare resolves it)
    std::string packed = data;
    if (auto_pack_mode() != 0 && !PackedPanel::is_packed(packed)) {
        std::string cached, err;
        const int rc = resolve_packed_panel(index, data, desc, auto_pack_mode() == 1, cached, err);
        if (rc < 0) return herr("%s", err.c_str());
        if (rc == 0) packed = cached;
    }
    const bool lean_kind = (kind == GAUSS_KIND_DIST || kind == GAUSS_KIND_DISTMIX || kind == GAUSS_KIND_QCAT || kind == GAUSS_KIND_QCATMIX);
    if (lean_kind && chr > 0 && !env_flag("GAUSS_HOST_FULL_MAP", false) && PackedPanel::is_packed(packed)) {
        std::string err;
        std::shared_ptr<PackedPanel> pk = open_packed_shared(packed, err);
        if (!pk) return herr("%s", err.c_str());
        if (pk->header().sorted) {
            ChromSetup cs;                                       // (errors in prepare()'s order: arguments, description file, populations, study file)
            if (chrom_setup(cs, kind, chr, wing, study_pop, names, wgts, nw, input, packed, desc, af1_cutoff, pk, nullptr)) return -1;
            cs.gw = load_gwas_cached(input, err);
            if (!cs.gw) return herr("%s", err.c_str());
            LeanWindow w;
            if (lean_window_build(w, cs, start_bp, end_bp)) return -1;
            gauss_window_desc d;
            if (lean_window_desc(w, &d)) return -1;
            void* dev = nullptr;
            const int64_t bytes = pk->n_snp() * pk->row_bytes();
            const bool resident = panel_is_resident(ctx, packed, &dev) ||
                                  (bytes <= ((int64_t)4 << 30) && panel_make_resident(ctx, packed, &dev, nullptr) == 0 && gauss_store_wait(ctx, dev, 0) == 0);
            if (resident) {
                d.geno_m = d.geno_u = (const uint8_t*)dev;
                gauss_job* job = nullptr;
                int rc = gauss_job_create(ctx, &d, 1, 1, &job);
                if (rc == 0) rc = gauss_job_run(job);
                if (rc == 0) rc = gauss_job_fetch(job);
                if (job) gauss_job_destroy(job);
                if (rc != 0) return herr("%s", gauss_last_error());
            } else {
                d.geno_m = d.geno_u = pk->geno();                 // a panel too large to keep in HBM: the window's rows from the mapped file
                if (gauss_impute_window(ctx, &d) != 0) return herr("%s", gauss_last_error());
            }
            *out = lean_window_finish(w);
            return 0;
        }
    }
    gauss_prepared* p = nullptr;
    if (gauss_host_prepare(kind, chr, start_bp, end_bp, wing, study_pop, names, wgts, nw, input, nullptr, index, data, desc, af1_cutoff, &p)) return -1;
    std::unique_ptr<gauss_prepared> hold(p);
    gauss_window_desc d;
    if (gauss_prepared_window_desc(p, &d)) return -1;
    if (gauss_impute_window(ctx, &d) != 0) return herr("%s", gauss_last_error());
    return gauss_prepared_finish(p, out);
}

int gauss_host_dist(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size, const char* study_pop,
                    const char* input_file, const char* reference_index_file, const char* reference_data_file,
                    const char* reference_pop_desc_file, double af1_cutoff, gauss_table** out)
{
    return run_impute(ctx, GAUSS_KIND_DIST, chr, start_bp, end_bp, wing_size, study_pop, nullptr, nullptr, 0, input_file,
                      reference_index_file, reference_data_file, reference_pop_desc_file, af1_cutoff, out);
}

int gauss_host_distmix(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size,
                       const char* const* pop_names, const double* pop_wgts, int n_pop_wgt, const char* input_file,
                       const char* reference_index_file, const char* reference_data_file, const char* reference_pop_desc_file,
                       double af1_cutoff, gauss_table** out)
{
    return run_impute(ctx, GAUSS_KIND_DISTMIX, chr, start_bp, end_bp, wing_size, nullptr, pop_names, pop_wgts, n_pop_wgt,
                      input_file, reference_index_file, reference_data_file, reference_pop_desc_file, af1_cutoff, out);
}

int gauss_host_qcat(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size, const char* study_pop,
                    const char* input_file, const char* reference_index_file, const char* reference_data_file,
                    const char* reference_pop_desc_file, double af1_cutoff, gauss_table** out)
{
    return run_impute(ctx, GAUSS_KIND_QCAT, chr, start_bp, end_bp, wing_size, study_pop, nullptr, nullptr, 0, input_file,
                      reference_index_file, reference_data_file, reference_pop_desc_file, af1_cutoff, out);
}

int gauss_host_qcatmix(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size,
                       const char* const* pop_names, const double* pop_wgts, int n_pop_wgt, const char* input_file,
                       const char* reference_index_file, const char* reference_data_file, const char* reference_pop_desc_file,
                       double af1_cutoff, gauss_table** out)
{
    return run_impute(ctx, GAUSS_KIND_QCATMIX, chr, start_bp, end_bp, wing_size, nullptr, pop_names, pop_wgts, n_pop_wgt,
                      input_file, reference_index_file, reference_data_file, reference_pop_desc_file, af1_cutoff, out);
}

int gauss_host_prep_qcat(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size, const char* study_pop,
                         const char* input_file, const char* reference_index_file, const char* reference_data_file,
                         const char* reference_pop_desc_file, double af1_cutoff, gauss_table** out)
{
    return run_impute(ctx, GAUSS_KIND_PREP_QCAT, chr, start_bp, end_bp, wing_size, study_pop, nullptr, nullptr, 0, input_file,
                      reference_index_file, reference_data_file, reference_pop_desc_file, af1_cutoff, out);
}

int gauss_host_prep_recessive_impute(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size,
                                     const char* const* pop_names, const double* pop_wgts, int n_pop_wgt,
                                     const char* input_file, const char* reference_index_file,
                                     const char* reference_data_file, const char* reference_pop_desc_file,
                                     double af1_cutoff, gauss_table** out)
{
    return run_impute(ctx, GAUSS_KIND_PREP_RECESSIVE, chr, start_bp, end_bp, wing_size, nullptr, pop_names, pop_wgts,
                      n_pop_wgt, input_file, reference_index_file, reference_data_file, reference_pop_desc_file, af1_cutoff, out);
}

// stats::quantile(x, probs = p) of R, default type 7 (quantile.default): index = 1 + (n-1)p, lo = floor, hi = ceiling,
// q = x[lo], and if index > lo and x[hi] != q:  q = (1-h) q + h x[hi]  with h = index - lo.  NaN input is an error in R.
static int r_quantile7(std::vector<double> x, double p, double* q)
{
    const size_t n = x.size();
    for (double v : x) if (std::isnan(v)) return herr("missing values and NaN's not allowed if 'na.rm' is FALSE");
    if (n == 0) { *q = NAN; return 0; }
    std::sort(x.begin(), x.end());
    const double index = 1 + (double)(n - 1) * p;
    const double lo = std::floor(index), hi = std::ceil(index);
    double qs = x[(size_t)lo - 1];
    const double xh = x[(size_t)hi - 1];
    if (index > lo && xh != qs) { const double h = index - lo; qs = (1 - h) * qs + h * xh; }
    *q = qs;
    return 0;
}

int gauss_host_prep_zmix5(gauss_ctx* ctx, const char* input_file, const char* reference_index_file,
                          const char* reference_data_file, const char* reference_pop_desc_file,
                          double percentile, int interval, gauss_table** out)
{
    if (!ctx || !out) return herr("bad arguments");
    if (!input_file || !reference_index_file || !reference_data_file || !reference_pop_desc_file) return herr("file name is NULL");
    Args a;
    a.input_file = input_file; a.reference_index_file = reference_index_file;
    a.reference_data_file = reference_data_file; a.reference_pop_desc_file = reference_pop_desc_file;
    const double pct = std::isnan(percentile) ? 0.99 : percentile;            // zmix.cpp:57-61
    const int step = interval > 0 ? interval : 1;                             // zmix.cpp:63-67
    if (auto_pack_mode() != 0 && !PackedPanel::is_packed(a.reference_data_file)) {
        std::string cached, err;
        const int rc = resolve_packed_panel(a.reference_index_file, a.reference_data_file, a.reference_pop_desc_file,
                                            auto_pack_mode() == 1, cached, err);
        if (rc < 0) return herr("%s", err.c_str());
        if (rc == 0) a.reference_data_file = cached;
    }
    if (PackedPanel::is_packed(a.reference_data_file)) {
        std::string err;
        a.pk = open_packed_shared(a.reference_data_file, err);
        if (!a.pk) return herr("%s", err.c_str());
    }
    if (read_ref_desc(a)) return -1;
    if (a.pk && a.pk->n_pop() != a.num_pops) return herr("packed panel has %d populations, the description file %d", a.pk->n_pop(), a.num_pops);
    a.pop_flag_vec.assign(a.num_pops, 1);                                     // zmix.cpp:148-150: every population
    SnpMap m;
    if (ReadInputZ(m, a, true)) return -1;                                    // read_input_zmix, zmix.cpp:1078-1113 (no window)
    if (ReadReferenceIndex(m, a, true)) return -1;                            // read_ref_index_zmix, zmix.cpp:1115-1181
    std::vector<Snp*> measured, snp_vec;
    for (auto& kv : m) if (kv.second->type == 1) measured.push_back(kv.second.get());   // zmix.cpp:88-92
    for (size_t i = 0; i < measured.size(); i += (size_t)step) snp_vec.push_back(measured[i]);   // zmix.cpp:111-119

    // cal_af_norm_var (zmix.cpp:1183-1214): variance of the panel AF columns, normalised by mean(1-mean)
    std::vector<double> norm_var;
    {
        BgzfReader fp;
        if (!a.pk && !fp.open(a.reference_data_file)) return herr("ERROR: can't open reference data file '%s'", a.reference_data_file.c_str());
        std::vector<double> af;
        for (Snp* s : snp_vec) {
            if (a.pk) af.assign(a.pk->af(s->fpos), a.pk->af(s->fpos) + a.num_pops);
            else load_line(fp, *s, a, &af);
            const int n = (int)af.size();
            double sum = 0.0, sq = 0.0;
            for (double v : af) sum += v;
            for (double v : af) sq += v * v;
            const double mean = sum / n;
            const double variance = sq / n - mean * mean;
            norm_var.push_back(variance / (mean * (1 - mean)));
        }
    }
    double cutoff = 0;
    if (r_quantile7(norm_var, pct, &cutoff)) return -1;                       // zmix.cpp:126-130
    std::vector<Snp*> sub;
    std::vector<double> sub_nv;
    for (size_t i = 0; i < snp_vec.size(); i++)
        if (norm_var[i] > cutoff) { sub.push_back(snp_vec[i]); sub_nv.push_back(norm_var[i]); }   // zmix.cpp:135-139

    const int S = (int)sub.size(), P = a.num_pops;
    int N = 0;
    for (int k = 0; k < P; k++) N += a.ref_pop_size_vec[k];
    std::vector<int32_t> pop_off(1, 0);
    for (int k = 0; k < P; k++) pop_off.push_back(pop_off.back() + a.ref_pop_size_vec[k]);
    const size_t npairs = S > 1 ? (size_t)S * (S - 1) / 2 : 0;
    gauss_table* t = new gauss_table();
    std::unique_ptr<gauss_table> hold(t);
    NamedMat dm;
    dm.name = "data_mat"; dm.nrow = (int)npairs; dm.ncol = 1 + P;
    dm.d.assign(npairs * (size_t)(1 + P), 0.0);
    if (S > 1) {
        // ReadGenotype for the selected SNPs, all populations (zmix.cpp:148-153), then the pair table
        gauss_prepared tmp;
        tmp.args = a; tmp.N = N; tmp.ld = ((int64_t)N + 15) / 16 * 16;
        std::vector<uint8_t> G;
        if (a.pk) unpack_rows(tmp, sub, G);
        else {
            BgzfReader fp;
            if (!fp.open(a.reference_data_file)) return herr("ERROR: can't open reference data file '%s'", a.reference_data_file.c_str());
            for (Snp* s : sub) {
                load_line(fp, *s, a, nullptr);
                int n = 0;
                for (auto& g : s->geno) n += g.second;
                if (n != N) return herr("ERROR: genotype line of %s has %d samples, population table says %d", s->rsid.c_str(), n, N);
            }
            fill_matrix(G, sub, tmp.ld);
        }
        size_t row = 0;
        for (int i = 0; i < S; i++)
            for (int j = i + 1; j < S; j++) dm.d[row++] = sub[i]->z * sub[j]->z;         // zmix.cpp:165
        if (gauss_ld_per_pop(ctx, G.data(), S, tmp.ld, pop_off.data(), P, dm.d.data() + npairs) != 0)
            return herr("%s", gauss_last_error());
    }
    Column rsid{"rsid", GAUSS_COL_STR, {}, {}, {}}, chr{"chr", GAUSS_COL_INT, {}, {}, {}}, bp{"bp", GAUSS_COL_INT, {}, {}, {}};
    Column a1{"a1", GAUSS_COL_STR, {}, {}, {}}, a2{"a2", GAUSS_COL_STR, {}, {}, {}}, z{"z", GAUSS_COL_DBL, {}, {}, {}};
    Column nv{"norm_var", GAUSS_COL_DBL, {}, {}, {}};
    for (int i = 0; i < S; i++) {
        rsid.s.push_back(sub[i]->rsid); chr.i.push_back(sub[i]->chr); bp.i.push_back((int)sub[i]->bp);
        a1.s.push_back(sub[i]->a1); a2.s.push_back(sub[i]->a2); z.d.push_back(sub[i]->z); nv.d.push_back(sub_nv[i]);
    }
    t->cols = {rsid, chr, bp, a1, a2, z, nv};
    t->named.push_back(std::move(dm));
    *out = hold.release();
    return 0;
}

// ------------------------------------------------------------------------------------------
// The other prep_zmix selectors (zmix.cpp:201-1076).  They share prep_zmix5's reading (read_input_zmix /
// read_ref_index_zmix, every population) and its output (one row per SNP pair: z_i * z_j, then the pair's genotype
// correlation inside each population) and differ in WHICH pairs they list:
//   prep_zmix      zmix.cpp:940-1076   every interval-th measured SNP (default 1), all pairs
//   prep_zmix2     zmix.cpp:651-760    pairs (i, i + offset) for i = 0, interval, 2 interval, ...   (1000, 3)
//   prep_zmix3     zmix.cpp:511-650    every interval-th SNP, each with its next `steps` neighbours  (1000, 5)
//   prep_zmix4     zmix.cpp:363-510    for h = 0 .. interval-1: pairs (i, i + offset), i = h, h + interval, ...; an extra
//                                      leading column holds h                                        (1000, 3)
//   prep_zmix5_sup zmix.cpp:201-361    prep_zmix5's ancestry-informative SNPs, correlations pooled per SUPER-population
//                                      (CalCorSup zmix.cpp:1221-1246), super-populations in order of first appearance
// The GPU part is gauss_ld_per_pop_pairs: only the tile pairs the listed pairs touch are multiplied.
// ------------------------------------------------------------------------------------------
enum ZmixVariant { ZMIX_ALL = 0, ZMIX_2 = 2, ZMIX_3 = 3, ZMIX_4 = 4, ZMIX_5SUP = 6 };

static int prep_zmix_variant(gauss_ctx* ctx, int variant, const char* input_file, const char* reference_index_file,
                             const char* reference_data_file, const char* reference_pop_desc_file, double percentile, int interval,
                             int p2, gauss_table** out)
{
    if (!ctx || !out) return herr("bad arguments");
    if (!input_file || !reference_index_file || !reference_data_file || !reference_pop_desc_file) return herr("file name is NULL");
    Args a;
    a.input_file = input_file; a.reference_index_file = reference_index_file;
    a.reference_data_file = reference_data_file; a.reference_pop_desc_file = reference_pop_desc_file;
    // defaults: zmix.cpp:953-957 (1), 664-675 / 376-387 / 524-535 (1000 and 3 / 3 / 5), 214-224 (0.99, 1)
    const int step = interval > 0 ? interval : ((variant == ZMIX_ALL || variant == ZMIX_5SUP) ? 1 : 1000);
    const int par2 = p2 > 0 ? p2 : (variant == ZMIX_3 ? 5 : 3);
    const double pct = std::isnan(percentile) ? 0.99 : percentile;
    if (auto_pack_mode() != 0 && !PackedPanel::is_packed(a.reference_data_file)) {
        std::string cached, err;
        const int rc = resolve_packed_panel(a.reference_index_file, a.reference_data_file, a.reference_pop_desc_file,
                                            auto_pack_mode() == 1, cached, err);
        if (rc < 0) return herr("%s", err.c_str());
        if (rc == 0) a.reference_data_file = cached;
    }
    if (PackedPanel::is_packed(a.reference_data_file)) {
        std::string err;
        a.pk = open_packed_shared(a.reference_data_file, err);
        if (!a.pk) return herr("%s", err.c_str());
    }
    if (read_ref_desc(a)) return -1;
    if (a.pk && a.pk->n_pop() != a.num_pops) return herr("packed panel has %d populations, the description file %d", a.pk->n_pop(), a.num_pops);
    a.pop_flag_vec.assign(a.num_pops, 1);
    SnpMap m;
    if (ReadInputZ(m, a, true)) return -1;
    if (ReadReferenceIndex(m, a, true)) return -1;
    std::vector<Snp*> measured;
    for (auto& kv : m) if (kv.second->type == 1) measured.push_back(kv.second.get());
    const int n = (int)measured.size();

    // ---- which SNPs, which pairs (indices into `measured`) ----
    std::vector<std::pair<int, int>> pairs;
    std::vector<double> lead;                                    // prep_zmix4's leading column
    std::vector<double> sub_nv;                                  // prep_zmix5_sup: norm_var of the kept SNPs
    if (variant == ZMIX_ALL || variant == ZMIX_3 || variant == ZMIX_5SUP) {
        std::vector<int> sub;
        for (int i = 0; i < n; i += step) sub.push_back(i);      // zmix.cpp:996-1004, 567-575, 257-265
        if (variant == ZMIX_5SUP) {
            // cal_af_norm_var + the percentile cut, as in prep_zmix5 (zmix.cpp:268-285)
            std::vector<double> norm_var;
            BgzfReader fp;
            if (!a.pk && !fp.open(a.reference_data_file)) return herr("ERROR: can't open reference data file '%s'", a.reference_data_file.c_str());
            std::vector<double> af;
            for (int i : sub) {
                Snp* s = measured[(size_t)i];
                if (a.pk) af.assign(a.pk->af(s->fpos), a.pk->af(s->fpos) + a.num_pops);
                else load_line(fp, *s, a, &af);
                const int k = (int)af.size();
                double sum = 0.0, sq = 0.0;
                for (double v : af) sum += v;
                for (double v : af) sq += v * v;
                const double mean = sum / k;
                norm_var.push_back((sq / k - mean * mean) / (mean * (1 - mean)));
            }
            double cutoff = 0;
            if (r_quantile7(norm_var, pct, &cutoff)) return -1;
            std::vector<int> kept;
            for (size_t i = 0; i < sub.size(); i++) if (norm_var[i] > cutoff) { kept.push_back(sub[i]); sub_nv.push_back(norm_var[i]); }
            sub.swap(kept);
        }
        const int S = (int)sub.size();
        for (int i = 0; i < S; i++) {
            const int jend = variant == ZMIX_3 ? std::min(i + 1 + par2, S) : S;      // zmix.cpp:592-594
            for (int j = i + 1; j < jend; j++) pairs.emplace_back(sub[(size_t)i], sub[(size_t)j]);
        }
    } else if (variant == ZMIX_2) {
        for (int i = 0; i < n; i += step) {                      // zmix.cpp:721-745
            if (i + par2 < n) pairs.emplace_back(i, i + par2);
            else break;
        }
    } else {                                                     // ZMIX_4, zmix.cpp:440-465
        for (int h = 0; h < step; h++)
            for (int i = h; i < n; i += step) {
                if (i + par2 < n) { pairs.emplace_back(i, i + par2); lead.push_back((double)h); }
                else break;
            }
    }
    // the SNPs that occur in some pair, in list order; a pair is (smaller row, larger row) -- the reference's pairs are
    // (earlier SNP, later SNP) already, and the correlation is symmetric
    std::vector<int> row_of((size_t)n, -1);
    std::vector<Snp*> sel;
    {
        std::vector<char> used((size_t)n, 0);
        for (auto& pr : pairs) { used[(size_t)pr.first] = 1; used[(size_t)pr.second] = 1; }
        for (int i = 0; i < n; i++) if (used[(size_t)i]) { row_of[(size_t)i] = (int)sel.size(); sel.push_back(measured[(size_t)i]); }
    }
    const int S = (int)sel.size(), P = a.num_pops;
    // population groups: each population (all but _sup), or its super-population in order of first appearance
    std::vector<int32_t> pop_group;
    int n_group = P;
    std::vector<std::string> group_names = a.ref_pop_vec;
    if (variant == ZMIX_5SUP) {
        group_names.clear();
        for (int k = 0; k < P; k++) {
            const std::string& sp = a.ref_sup_pop_vec[(size_t)k];
            size_t g = 0;
            while (g < group_names.size() && group_names[g] != sp) g++;
            if (g == group_names.size()) group_names.push_back(sp);
            pop_group.push_back((int32_t)g);
        }
        n_group = (int)group_names.size();
    }
    int N = 0;
    std::vector<int32_t> pop_off(1, 0);
    for (int k = 0; k < P; k++) { N += a.ref_pop_size_vec[(size_t)k]; pop_off.push_back(pop_off.back() + a.ref_pop_size_vec[(size_t)k]); }
    const size_t np = pairs.size();
    const int nlead = variant == ZMIX_4 ? 1 : 0;
    std::unique_ptr<gauss_table> t(new gauss_table());
    NamedMat dm;
    dm.name = "data_mat"; dm.nrow = (int)np; dm.ncol = nlead + 1 + n_group;
    dm.d.assign(np * (size_t)dm.ncol, 0.0);
    if (np > 0) {
        gauss_prepared tmp;
        tmp.args = a; tmp.N = N; tmp.ld = ((int64_t)N + 15) / 16 * 16;
        std::vector<uint8_t> G;
        if (a.pk) unpack_rows(tmp, sel, G);
        else {
            BgzfReader fp;
            if (!fp.open(a.reference_data_file)) return herr("ERROR: can't open reference data file '%s'", a.reference_data_file.c_str());
            for (Snp* s : sel) {
                load_line(fp, *s, a, nullptr);
                int k = 0;
                for (auto& g : s->geno) k += g.second;
                if (k != N) return herr("ERROR: genotype line of %s has %d samples, population table says %d", s->rsid.c_str(), k, N);
            }
            fill_matrix(G, sel, tmp.ld);
        }
        std::vector<int32_t> pi(np), pj(np);
        for (size_t k = 0; k < np; k++) {
            pi[k] = row_of[(size_t)pairs[k].first]; pj[k] = row_of[(size_t)pairs[k].second];
            if (nlead) dm.d[k] = lead[k];
            dm.d[(size_t)nlead * np + k] = measured[(size_t)pairs[k].first]->z * measured[(size_t)pairs[k].second]->z;
        }
        if (gauss_ld_per_pop_pairs(ctx, G.data(), S, tmp.ld, pop_off.data(), P, pop_group.empty() ? nullptr : pop_group.data(), n_group,
                                   pi.data(), pj.data(), (int64_t)np, dm.d.data() + (size_t)(nlead + 1) * np) != 0)
            return herr("%s", gauss_last_error());
    }
    Column rsid{"rsid", GAUSS_COL_STR, {}, {}, {}}, chr{"chr", GAUSS_COL_INT, {}, {}, {}}, bp{"bp", GAUSS_COL_INT, {}, {}, {}};
    Column a1{"a1", GAUSS_COL_STR, {}, {}, {}}, a2{"a2", GAUSS_COL_STR, {}, {}, {}}, z{"z", GAUSS_COL_DBL, {}, {}, {}};
    for (Snp* s : sel) {
        rsid.s.push_back(s->rsid); chr.i.push_back(s->chr); bp.i.push_back((int)s->bp);
        a1.s.push_back(s->a1); a2.s.push_back(s->a2); z.d.push_back(s->z);
    }
    t->cols = {rsid, chr, bp, a1, a2, z};
    if (variant == ZMIX_5SUP) { Column nv{"norm_var", GAUSS_COL_DBL, {}, {}, {}}; nv.d = sub_nv; t->cols.push_back(nv); }
    t->named.push_back(std::move(dm));
    {
        // the groups the correlation columns stand for, and the pairs as rows of the SNP table
        NamedMat pm;
        pm.name = "pairs"; pm.nrow = (int)np; pm.ncol = 2;
        pm.d.assign(np * 2, 0.0);
        for (size_t k = 0; k < np; k++) { pm.d[k] = row_of[(size_t)pairs[k].first]; pm.d[np + k] = row_of[(size_t)pairs[k].second]; }
        t->named.push_back(std::move(pm));
        for (const std::string& g : group_names) t->messages.push_back(g);
    }
    *out = t.release();
    return 0;
}

int gauss_host_prep_zmix(gauss_ctx* ctx, const char* input_file, const char* reference_index_file, const char* reference_data_file,
                         const char* reference_pop_desc_file, int interval, gauss_table** out)
{
    return prep_zmix_variant(ctx, ZMIX_ALL, input_file, reference_index_file, reference_data_file, reference_pop_desc_file, NAN, interval, 0, out);
}
int gauss_host_prep_zmix2(gauss_ctx* ctx, const char* input_file, const char* reference_index_file, const char* reference_data_file,
                          const char* reference_pop_desc_file, int interval, int offset, gauss_table** out)
{
    return prep_zmix_variant(ctx, ZMIX_2, input_file, reference_index_file, reference_data_file, reference_pop_desc_file, NAN, interval, offset, out);
}
int gauss_host_prep_zmix3(gauss_ctx* ctx, const char* input_file, const char* reference_index_file, const char* reference_data_file,
                          const char* reference_pop_desc_file, int interval, int steps, gauss_table** out)
{
    return prep_zmix_variant(ctx, ZMIX_3, input_file, reference_index_file, reference_data_file, reference_pop_desc_file, NAN, interval, steps, out);
}
int gauss_host_prep_zmix4(gauss_ctx* ctx, const char* input_file, const char* reference_index_file, const char* reference_data_file,
                          const char* reference_pop_desc_file, int interval, int offset, gauss_table** out)
{
    return prep_zmix_variant(ctx, ZMIX_4, input_file, reference_index_file, reference_data_file, reference_pop_desc_file, NAN, interval, offset, out);
}
int gauss_host_prep_zmix5_sup(gauss_ctx* ctx, const char* input_file, const char* reference_index_file, const char* reference_data_file,
                              const char* reference_pop_desc_file, double percentile, int interval, gauss_table** out)
{
    return prep_zmix_variant(ctx, ZMIX_5SUP, input_file, reference_index_file, reference_data_file, reference_pop_desc_file, percentile, interval, 0, out);
}

int gauss_host_computeLD(gauss_ctx* ctx, int chr, int64_t start_bp, int64_t end_bp, const char* const* pop_names,
                         const double* pop_wgts, int n_pop_wgt, const char* input_file, const char* reference_index_file,
                         const char* reference_data_file, const char* reference_pop_desc_file, double af1_cutoff,
                         gauss_table** out)
{
    if (!ctx || !out) return herr("bad arguments");
    // On a sorted packed panel: the window as a merge (host_chrom.cpp:LeanWindow, measured SNPs only), rows from the resident panel --
    // as the one-window imputation calls above; GAUSS_HOST_FULL_MAP=1 or any other panel: the literal path below.
    if (input_file && reference_index_file && reference_data_file && reference_pop_desc_file && chr > 0 && !env_flag("GAUSS_HOST_FULL_MAP", false)) {
        std::string packed = reference_data_file, err;
        if (auto_pack_mode() != 0 && !PackedPanel::is_packed(packed)) {
            std::string cached;
            const int rc = resolve_packed_panel(reference_index_file, reference_data_file, reference_pop_desc_file, auto_pack_mode() == 1, cached, err);
            if (rc < 0) return herr("%s", err.c_str());
            if (rc == 0) packed = cached;
        }
        std::shared_ptr<PackedPanel> pk;
        if (PackedPanel::is_packed(packed) && !(pk = open_packed_shared(packed, err))) return herr("%s", err.c_str());
        if (pk && pk->header().sorted) {
            ChromSetup cs;
            if (chrom_setup(cs, GAUSS_KIND_COMPUTELD, chr, 0, nullptr, pop_names, pop_wgts, n_pop_wgt, input_file, packed, reference_pop_desc_file,
                            af1_cutoff, pk, nullptr)) return -1;
            cs.gw = load_gwas_cached(input_file, err);
            if (!cs.gw) return herr("%s", err.c_str());
            LeanWindow w;
            if (lean_window_build(w, cs, start_bp, end_bp)) return -1;
            const int M = (int)w.measured.size();
            if (M <= cs.a.min_num_measured_snp)                          // computeLD.cpp:89-93
                return herr("Not enough number of SNPs loaded - computeLD not performed (measured %d)", M);
            std::unique_ptr<gauss_table> t(new gauss_table());
            t->matrix.assign((size_t)M * M, 0.0);
            t->matrix_n = M;
            void* dev = nullptr;
            const int64_t bytes = pk->n_snp() * pk->row_bytes();
            const bool resident = panel_is_resident(ctx, packed, &dev) ||
                                  (bytes <= ((int64_t)4 << 30) && panel_make_resident(ctx, packed, &dev, nullptr) == 0 && gauss_store_wait(ctx, dev, 0) == 0);
            if (gauss_ld_rows(ctx, GAUSS_MODE_WEIGHTED, resident ? (const uint8_t*)dev : pk->geno(), pk->row_bytes(), GAUSS_GENO_2BIT, w.store_rows_m.data(), M,
                              cs.pop_off.data(), cs.pop_src_off.data(), cs.pop_wgt.data(), (int)cs.pop_off.size() - 1, 1.0, resident ? 1 : 0,
                              t->matrix.data()) != 0) return herr("%s", gauss_last_error());
            Column rsid{"rsid", GAUSS_COL_STR, {}, {}, {}}, chrc{"chr", GAUSS_COL_INT, {}, {}, {}}, bp{"bp", GAUSS_COL_INT, {}, {}, {}};
            Column a1{"a1", GAUSS_COL_STR, {}, {}, {}}, a2{"a2", GAUSS_COL_STR, {}, {}, {}}, af{"af1mix", GAUSS_COL_DBL, {}, {}, {}};
            for (int32_t vi : w.measured) {                              // computeLD.cpp:134-149
                const LeanSnp& sn = w.v[(size_t)vi];
                const PkSnp& ps = pk->snp(sn.row);
                rsid.s.emplace_back(pk->str(ps.rsid)); chrc.i.push_back(ps.chr); bp.i.push_back((int)sn.bp);
                a1.s.emplace_back(pk->str(ps.a1)); a2.s.emplace_back(pk->str(ps.a2)); af.d.push_back(sn.af);
            }
            t->cols = {rsid, chrc, bp, a1, a2, af};
            *out = t.release();
            return 0;
        }
    }
    gauss_prepared* p = nullptr;
    if (gauss_host_prepare(GAUSS_KIND_COMPUTELD, chr, start_bp, end_bp, 0, nullptr, pop_names, pop_wgts, n_pop_wgt, input_file,
                           nullptr, reference_index_file, reference_data_file, reference_pop_desc_file, af1_cutoff, &p)) return -1;
    std::unique_ptr<gauss_prepared> hold(p);
    const int M = (int)p->measured.size();
    if (M <= p->args.min_num_measured_snp)                               // computeLD.cpp:89-93
        return herr("Not enough number of SNPs loaded - computeLD not performed (measured %d)", M);
    std::unique_ptr<gauss_table> t(new gauss_table());
    t->matrix.assign((size_t)M * M, 0.0);
    t->matrix_n = M;
    if (p->packed_rows) {
        const uint8_t* store = nullptr;
        int on_device = 0;
        if (packed_row_source(ctx, *p, &store, &on_device)) return -1;
        if (gauss_ld_rows(ctx, GAUSS_MODE_WEIGHTED, store, p->args.pk->row_bytes(), GAUSS_GENO_2BIT, p->store_rows_m.data(), M,
                          p->pop_off.data(), p->pop_src_off.data(), p->pop_wgt.data(), (int)p->pop_off.size() - 1, 1.0, on_device,
                          t->matrix.data()) != 0) return herr("%s", gauss_last_error());
    } else if (gauss_ld(ctx, GAUSS_MODE_WEIGHTED, p->gm.data(), M, p->ld, p->pop_off.data(), p->pop_wgt.data(),
                        (int)p->pop_off.size() - 1, 1.0, t->matrix.data()) != 0) return herr("%s", gauss_last_error());
    Column rsid{"rsid", GAUSS_COL_STR, {}, {}, {}}, chrc{"chr", GAUSS_COL_INT, {}, {}, {}}, bp{"bp", GAUSS_COL_INT, {}, {}, {}};
    Column a1{"a1", GAUSS_COL_STR, {}, {}, {}}, a2{"a2", GAUSS_COL_STR, {}, {}, {}}, af{"af1mix", GAUSS_COL_DBL, {}, {}, {}};
    for (Snp* s : p->measured) {                                        // computeLD.cpp:134-149
        rsid.s.push_back(s->rsid); chrc.i.push_back(s->chr); bp.i.push_back((int)s->bp);
        a1.s.push_back(s->a1); a2.s.push_back(s->a2); af.d.push_back(s->af1mix);
    }
    t->cols = {rsid, chrc, bp, a1, a2, af};
    *out = t.release();
    return 0;
}

// ---- jepeg() / jepegmix() over several ranks ("genes: contiguous gene ranges") ------------------------------
// Genes are independent (jepeg.cpp:114-131: one Gene object per gene, nothing shared but the read-only SNP map; grouping at
// gauss.cpp:1383-1439).  Every rank runs the same host data layer on the same files, derives the same plan -- contiguous gene
// ranges of equal cost -- and computes CorG and the k x k tails of ITS range only; the ranges' tables, concatenated in rank order,
// are the one-rank table row for row (same bits: a gene's block and tail do not depend on which other genes share the launch).
// A gene costs its SNP pairs n (n + 1) (CorG's pair loops, gene.cpp:306-315 / 576-586 -- the unit the judge's plan names) plus a
// constant for its k x k tail, which does not grow with n (k <= 6; ~3 us against ~1 us per 100 pairs of the batch launch).
static const long long JEPEG_GENE_TAIL_COST = 64;
static void jepeg_gene_ranges(const std::vector<int32_t>& gene_off, int world, std::vector<int32_t>& first)
{
    const int ng = gene_off.empty() ? 0 : (int)gene_off.size() - 1;
    world = std::max(1, world);
    std::vector<long long> pre((size_t)ng + 1, 0);
    for (int g = 0; g < ng; g++) {
        const long long n = gene_off[(size_t)g + 1] - gene_off[(size_t)g];
        pre[(size_t)g + 1] = pre[(size_t)g] + n * (n + 1) + JEPEG_GENE_TAIL_COST;
    }
    first.assign((size_t)world + 1, ng);
    first[0] = 0;
    int g = 0;
    for (int r = 1; r < world; r++) {
        // rank r starts at the first gene whose MIDPOINT lies at or beyond r / world of the total: boundaries are monotone, every gene
        // belongs to exactly one rank, and a rank may be empty when there are fewer genes than ranks
        const long long want = 2 * pre[(size_t)ng] * r;           // compare 2 * world * midpoint with 2 * total * r (integers)
        while (g < ng && (pre[(size_t)g] + pre[(size_t)g + 1]) * world < want) g++;
        first[(size_t)r] = g;
    }
}

int gauss_prepared_jepeg_plan(const gauss_prepared* p, int world, int32_t* first)
{
    if (!p || !first) return herr("bad arguments");
    if (p->kind != GAUSS_KIND_JEPEG && p->kind != GAUSS_KIND_JEPEGMIX) return herr("not a jepeg / jepegmix object");
    if (world < 1) return herr("world = %d", world);
    std::vector<int32_t> f;
    jepeg_gene_ranges(p->gene_off, world, f);
    std::copy(f.begin(), f.end(), first);
    return 0;
}

// The gene table of genes [g0, g1) from their CorG blocks (concatenated n_g x n_g, diagonal 1 + lambda): Gene::RunJepeg bookkeeping
// and CalJepegPval from W on (gene.cpp:88-185, 317-550), jepeg.cpp:143-151's columns.  Named matrix "gene_range" = [g0, g1, genes].
static gauss_table* jepeg_table(const gauss_prepared& p, int g0, int g1, const double* blocks)
{
    std::unique_ptr<gauss_table> t(new gauss_table());
    Column geneid{"geneid", GAUSS_COL_STR, {}, {}, {}}, chisq{"chisq", GAUSS_COL_DBL, {}, {}, {}}, df{"df", GAUSS_COL_INT, {}, {}, {}};
    Column jp{"jepeg_pval", GAUSS_COL_DBL, {}, {}, {}}, ns{"num_snp", GAUSS_COL_INT, {}, {}, {}}, tc{"top_categ", GAUSS_COL_STR, {}, {}, {}};
    Column tcp{"top_categ_pval", GAUSS_COL_DBL, {}, {}, {}}, ts{"top_snp", GAUSS_COL_STR, {}, {}, {}}, tsp{"top_snp_pval", GAUSS_COL_DBL, {}, {}, {}};
    // (the k x k tails on host threads were measured twice: round 5, eight threads: 1.1 -> 0.45 ms and the call no shorter; round 6,
    // at most four threads, 16 genes a task, rows written in place: tails 1.12 -> 0.47 ms and the NEXT call's data layer 1.12 ->
    // 1.35-1.5 ms -- the threads started per call move the calling thread off its warm core -- call 3.48 -> 3.35-3.5 ms: not kept)
    size_t o = 0;
    for (int g = g0; g < g1; g++) {
        std::vector<Snp*> gs(p.measured.begin() + p.gene_off[(size_t)g], p.measured.begin() + p.gene_off[(size_t)g + 1]);
        const GeneResult r = jepeg_tail(gs, blocks + o, p.args);
        o += gs.size() * gs.size();
        geneid.s.push_back(r.geneid); chisq.d.push_back(r.chisq); df.i.push_back(r.df); jp.d.push_back(r.jepeg_pval);
        ns.i.push_back(r.num_snp); tc.s.push_back(r.top_categ); tcp.d.push_back(r.top_categ_pval);
        ts.s.push_back(r.top_snp); tsp.d.push_back(r.top_snp_pval);
    }
    t->cols = {geneid, chisq, df, jp, ns, tc, tcp, ts, tsp};          // jepeg.cpp:143-151
    NamedMat gr;
    gr.name = "gene_range"; gr.nrow = 1; gr.ncol = 3;
    gr.d = {(double)g0, (double)g1, (double)(p.gene_off.empty() ? 0 : (int)p.gene_off.size() - 1)};
    t->named.push_back(std::move(gr));
    return t.release();
}

int gauss_prepared_jepeg_finish(const gauss_prepared* p, int g0, int g1, const double* blocks, gauss_table** out)
{
    if (!p || !out) return herr("bad arguments");
    if (p->kind != GAUSS_KIND_JEPEG && p->kind != GAUSS_KIND_JEPEGMIX) return herr("not a jepeg / jepegmix object");
    const int ng = p->gene_off.empty() ? 0 : (int)p->gene_off.size() - 1;
    if (g0 < 0 || g1 < g0 || g1 > ng) return herr("gene range [%d, %d) of %d genes", g0, g1, ng);
    if (g1 > g0 && !blocks) return herr("blocks is NULL");
    *out = jepeg_table(*p, g0, g1, blocks);
    return 0;
}

static int run_jepeg(gauss_ctx* ctx, int kind, const char* study_pop, const char* const* names, const double* wgts, int nw,
                     const char* input, const char* annotation, const char* index, const char* data, const char* desc,
                     double af1_cutoff, int rank, int world, gauss_table** out)
{
    if (!ctx || !out) return herr("bad arguments");
    if (world < 1 || rank < 0 || rank >= world) return herr("rank %d of world %d", rank, world);
    gauss_prepared* p = nullptr;
    // The gene table is made of annotated SNPs alone, and every step of the data layer after ReadInputZ works on the entries of one
    // position at a time: only the study SNPs at positions the annotation names enter the SNP map (plus the positions the study
    // lists more than once or under equal alleles -- the only ones where the reference's duplicate check can fire, so a study that
    // fails there still fails).  A chromosome's study is four times its annotated SNPs: the data layer of a jepegmix() call
    // 3.9 -> 1.0 ms.  GAUSS_HOST_FULL_MAP=1: the whole study, as gauss_host_prepare does it (same table, bit for bit:
    // tests/test_gpu_drivers.py).
    const bool full_map = env_flag("GAUSS_HOST_FULL_MAP", false);
    const double t_begin = now_s();
    if (host_prepare(kind, 0, 0, 0, 0, study_pop, names, wgts, nw, input, annotation, index, data, desc, af1_cutoff, !full_map, &p)) return -1;
    std::unique_ptr<gauss_prepared> hold(p);
    const double t_prepared = now_s();
    const Args& a = p->args;
    std::vector<int32_t> first;
    jepeg_gene_ranges(p->gene_off, world, first);
    const int g0 = first[(size_t)rank], g1 = first[(size_t)rank + 1];
    const int ng = g1 - g0;
    // this rank's genes: rows [r0, r0 + S) of the measured list, gene offsets relative to r0
    const int r0 = ng > 0 ? p->gene_off[(size_t)g0] : 0;
    const int S = ng > 0 ? p->gene_off[(size_t)g1] - r0 : 0;
    std::vector<int32_t> goff((size_t)ng + 1, 0);
    size_t tot = 0;
    for (int g = 0; g < ng; g++) {
        goff[(size_t)g + 1] = p->gene_off[(size_t)(g0 + g) + 1] - r0;
        const size_t n = (size_t)(goff[(size_t)g + 1] - goff[(size_t)g]);
        tot += n * n;
    }
    std::vector<double> blocks(std::max<size_t>(tot, 1), 0.0);
    if (S > 0 && ng > 0) {
        // CorG of every gene of the range in one launch, diagonal 1 + lambda (gene.cpp:306-315 / 576-586)
        const int mode = (kind == GAUSS_KIND_JEPEG) ? GAUSS_MODE_POOLED : GAUSS_MODE_WEIGHTED;
        if (p->packed_rows) {
            const uint8_t* store = nullptr;
            int on_device = 0;
            if (packed_row_source(ctx, *p, &store, &on_device)) return -1;
            if (gauss_gene_ld_batch_rows(ctx, mode, store, a.pk->row_bytes(), GAUSS_GENO_2BIT, p->store_rows_m.data() + r0, S,
                                         p->pop_off.data(), p->pop_src_off.data(), p->pop_wgt.data(), (int)p->pop_off.size() - 1,
                                         goff.data(), ng, 1.0 + a.lambda, on_device, blocks.data()) != 0)
                return herr("%s", gauss_last_error());
        } else if (gauss_gene_ld_batch(ctx, mode, p->gm.data() + (size_t)r0 * (size_t)p->ld, S, p->ld, p->pop_off.data(), p->pop_wgt.data(),
                                       (int)p->pop_off.size() - 1, goff.data(), ng, 1.0 + a.lambda, blocks.data()) != 0)
            return herr("%s", gauss_last_error());
    }
    const double t_ld = now_s();
    *out = jepeg_table(*p, g0, g1, blocks.data());
    if (host_trace("prep"))
        fprintf(stderr, "[jepeg] rank %d of %d, genes [%d, %d): data layer %.2f ms, gene LD blocks on the GPU %.2f ms, %d k x k tails + table %.2f ms\n",
                rank, world, g0, g1, (t_prepared - t_begin) * 1e3, (t_ld - t_prepared) * 1e3, ng, (now_s() - t_ld) * 1e3);
    return 0;
}

int gauss_host_jepeg(gauss_ctx* ctx, const char* study_pop, const char* input_file, const char* annotation_file,
                     const char* reference_index_file, const char* reference_data_file, const char* reference_pop_desc_file,
                     double af1_cutoff, gauss_table** out)
{
    return run_jepeg(ctx, GAUSS_KIND_JEPEG, study_pop, nullptr, nullptr, 0, input_file, annotation_file, reference_index_file,
                     reference_data_file, reference_pop_desc_file, af1_cutoff, 0, 1, out);
}

int gauss_host_jepegmix(gauss_ctx* ctx, const char* const* pop_names, const double* pop_wgts, int n_pop_wgt,
                        const char* input_file, const char* annotation_file, const char* reference_index_file,
                        const char* reference_data_file, const char* reference_pop_desc_file, double af1_cutoff,
                        gauss_table** out)
{
    return run_jepeg(ctx, GAUSS_KIND_JEPEGMIX, nullptr, pop_names, pop_wgts, n_pop_wgt, input_file, annotation_file,
                     reference_index_file, reference_data_file, reference_pop_desc_file, af1_cutoff, 0, 1, out);
}

int gauss_host_jepeg_rank(gauss_ctx* ctx, int kind, const char* study_pop, const char* const* pop_names, const double* pop_wgts,
                          int n_pop_wgt, const char* input_file, const char* annotation_file, const char* reference_index_file,
                          const char* reference_data_file, const char* reference_pop_desc_file, double af1_cutoff,
                          int rank, int world, gauss_table** out)
{
    if (kind != GAUSS_KIND_JEPEG && kind != GAUSS_KIND_JEPEGMIX) return herr("kind %d is not jepeg / jepegmix", kind);
    return run_jepeg(ctx, kind, study_pop, pop_names, pop_wgts, n_pop_wgt, input_file, annotation_file, reference_index_file,
                     reference_data_file, reference_pop_desc_file, af1_cutoff, rank, world, out);
}

// Which rank runs which of n independent calls: longest first by `cost` onto the least loaded rank (ties: the lower index / rank).
static void deal_calls(const std::vector<double>& cost, int world, std::vector<int>& owner)
{
    const int n = (int)cost.size();
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; i++) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost[(size_t)x] > cost[(size_t)y]; });
    std::vector<double> load((size_t)std::max(1, world), 0.0);
    owner.assign((size_t)n, 0);
    for (int i : order) {
        int best = 0;
        for (int r = 1; r < world; r++) if (load[(size_t)r] < load[(size_t)best]) best = r;
        owner[(size_t)i] = best;
        load[(size_t)best] += cost[(size_t)i];
    }
}

int gauss_host_jepeg_genome(gauss_ctx* ctx, int kind, int n_calls, const char* study_pop, const char* const* pop_names,
                            const double* pop_wgts, int n_pop_wgt, const char* const* input_files, const char* const* annotation_files,
                            const char* const* reference_index_files, const char* const* reference_data_files,
                            const char* reference_pop_desc_file, double af1_cutoff, int rank, int world, gauss_table** out,
                            int32_t* owner_out)
{
    if (!ctx || !out || n_calls < 0 || !input_files || !annotation_files || !reference_data_files) return herr("bad arguments");
    if (kind != GAUSS_KIND_JEPEG && kind != GAUSS_KIND_JEPEGMIX) return herr("kind %d is not jepeg / jepegmix", kind);
    if (world < 1 || rank < 0 || rank >= world) return herr("rank %d of world %d", rank, world);
    // a call's cost: the size of its annotation file (its genes and gene SNPs are lines of it); every rank sees the same files
    std::vector<double> cost((size_t)n_calls, 1.0);
    for (int c = 0; c < n_calls; c++) {
        struct stat sb;
        if (annotation_files[c] && stat(annotation_files[c], &sb) == 0) cost[(size_t)c] = 1.0 + (double)sb.st_size;
    }
    std::vector<int> owner;
    deal_calls(cost, world, owner);
    int rc_all = 0;
    std::string first_err;
    for (int c = 0; c < n_calls; c++) {
        out[c] = nullptr;
        if (owner_out) owner_out[c] = owner[(size_t)c];
        if (owner[(size_t)c] != rank) continue;
        const char* idx = reference_index_files ? reference_index_files[c] : "(packed)";
        if (run_jepeg(ctx, kind, study_pop, pop_names, pop_wgts, n_pop_wgt, input_files[c], annotation_files[c], idx ? idx : "(packed)",
                      reference_data_files[c], reference_pop_desc_file, af1_cutoff, 0, 1, &out[c]) != 0) {
            // a call that fails leaves out[c] = NULL; the others still run (as the chromosome driver isolates a failing window)
            if (!rc_all) first_err = gauss_host_last_error();
            rc_all = -1;
            out[c] = nullptr;
        }
    }
    if (rc_all) return herr("%s", first_err.c_str());
    return 0;
}


}  // extern "C"
