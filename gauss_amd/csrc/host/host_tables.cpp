// libgauss_host.so -- result tables (the reference's output lists: dist.cpp:91-124, qcat.cpp:94-131, prep_qcat.cpp:135-204),
// the JEPEG k x k tail (gene.cpp:88-185, 317-550) and the table accessors of the C ABI (include/gauss_host.h).
#include "host_internal.h"

thread_local std::string g_err;

int herr(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

// ------------------------------------------------------------------------------------------
// small dense helpers for the JEPEG k x k tail (k <= 6): gene.cpp:317-550
// ------------------------------------------------------------------------------------------
double pnorm_upper(double x) { return 0.5 * erfc(x / 1.4142135623730951); }   // R::pnorm5(x,0,1,0,0)

double pchisq_upper(double x, int df)                                            // R::pchisq(x,df,0,0)
{
    if (df <= 0) return NAN;
    if (!(x > 0.0)) return (x != x) ? NAN : 1.0;
    const double h = 0.5 * x;
    if ((df & 1) == 0) {
        double term = 1.0, sum = 1.0;
        for (int k = 1; k < df / 2; k++) { term *= h / k; sum += term; }
        return exp(-h) * sum;
    }
    double q = erfc(sqrt(h));
    if (df > 1) {
        double term = sqrt(h) / 0.886226925452758, sum = term;
        for (int k = 2; k <= (df - 1) / 2; k++) { term *= h / (k - 0.5); sum += term; }
        q += exp(-h) * sum;
    }
    return q;
}

// cyclic Jacobi for symmetric k x k (k <= 6); V columns are eigenvectors
static void jacobi_small(int n, double* A, double* V, double* d)
{
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) d[i] = A[i * n + i];
}

static void make_pos_def_small(int n, double* M, double min_abs_eig)      // util.cpp:302-318
{
    double A[36], V[36], d[6];
    memcpy(A, M, sizeof(double) * n * n);
    jacobi_small(n, A, V, d);
    double mn = d[0];
    for (int i = 1; i < n; i++) mn = std::min(mn, d[i]);
    if (!(mn < min_abs_eig)) return;
    for (int i = 0; i < n; i++) if (d[i] < min_abs_eig) d[i] = min_abs_eig;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += V[i * n + k] * d[k] * V[j * n + k];
            M[i * n + j] = s;
        }
}

static void inv_small(int n, const double* Min, double* inv)               // util.cpp:298-300 (full pivoting)
{
    double A[36];
    int rp[6], cp[6];
    memcpy(A, Min, sizeof(double) * n * n);
    for (int k = 0; k < n; k++) {
        int pi = k, pj = k; double best = -1;
        for (int i = k; i < n; i++) for (int j = k; j < n; j++) if (fabs(A[i * n + j]) > best) { best = fabs(A[i * n + j]); pi = i; pj = j; }
        rp[k] = pi; cp[k] = pj;
        if (pi != k) for (int j = 0; j < n; j++) std::swap(A[k * n + j], A[pi * n + j]);
        if (pj != k) for (int i = 0; i < n; i++) std::swap(A[i * n + k], A[i * n + pj]);
        for (int i = k + 1; i < n; i++) A[i * n + k] /= A[k * n + k];
        for (int i = k + 1; i < n; i++) for (int j = k + 1; j < n; j++) A[i * n + j] -= A[i * n + k] * A[k * n + j];
    }
    for (int c = 0; c < n; c++) {
        double col[6];
        for (int i = 0; i < n; i++) col[i] = (i == c);
        for (int k = 0; k < n; k++) if (rp[k] != k) std::swap(col[k], col[rp[k]]);
        for (int k = 0; k < n; k++) for (int i = k + 1; i < n; i++) col[i] -= A[i * n + k] * col[k];
        for (int k = n - 1; k >= 0; k--) { col[k] /= A[k * n + k]; for (int i = 0; i < k; i++) col[i] -= A[i * n + k] * col[k]; }
        for (int k = n - 1; k >= 0; k--) if (cp[k] != k) std::swap(col[k], col[cp[k]]);
        for (int i = 0; i < n; i++) inv[i * n + c] = col[i];
    }
}


static const char* categ_name(int c)       // Categ::GetName, gene.cpp:17-33
{
    static const char* nm[6] = {"PFS", "TFB", "STR", "TAR", "CIS", "TRN"};
    return (c >= 0 && c < 6) ? nm[c] : "";
}

// Gene::RunJepeg + CalJepegPval tail (gene.cpp:88-185, 317-550) given CorG (n x n row-major)
GeneResult jepeg_tail(const std::vector<Snp*>& gs, const double* CorG, const Args& a)
{
    GeneResult r;
    const int n = (int)gs.size();
    r.num_snp = n;
    int count[6] = {0, 0, 0, 0, 0, 0};
    for (Snp* s : gs) for (auto& kv : s->categ) if (kv.first >= 0 && kv.first < 6) count[kv.first]++;
    int cat[6], k = 0;
    for (int c = 0; c < 6; c++) if (count[c]) cat[k++] = c;
    if (n == 0 || k == 0) return r;
    std::vector<double> W((size_t)k * n), WC((size_t)k * n);
    for (int s = 0; s < n; s++)
        for (int i = 0; i < k; i++) {
            auto it = gs[s]->categ.find(cat[i]);
            const double w = (it != gs[s]->categ.end()) ? it->second : 0.0;     // Snp::GetCategWgt
            W[(size_t)i * n + s] = w * sqrt(gs[s]->info);                       // gene.cpp:871
        }
    double WWt[36], CovU[36], CorU[36], U[6], pv[6]; bool rmv[6];
    for (int i = 0; i < k; i++) for (int j = 0; j < k; j++) {
        double s = 0; for (int t = 0; t < n; t++) s += W[(size_t)i * n + t] * W[(size_t)j * n + t];
        WWt[i * k + j] = s;
    }
    for (int i = 0; i < k; i++) for (int s = 0; s < n; s++) {
        double v = 0; for (int t = 0; t < n; t++) v += W[(size_t)i * n + t] * CorG[(size_t)t * n + s];
        WC[(size_t)i * n + s] = v;
    }
    for (int i = 0; i < k; i++) for (int j = 0; j < k; j++) {
        double s = 0; for (int t = 0; t < n; t++) s += WC[(size_t)i * n + t] * W[(size_t)j * n + t];
        CovU[i * k + j] = s;
    }
    for (int i = 0; i < k; i++) for (int j = i; j < k; j++) {          // CnvrtCovToCor, util.cpp:284-296
        const double c = CovU[i * k + j] / (sqrt(CovU[i * k + i]) * sqrt(CovU[j * k + j]));
        CorU[i * k + j] = c; CorU[j * k + i] = c;
    }
    for (int i = 0; i < k; i++) {
        double s = 0; for (int t = 0; t < n; t++) s += W[(size_t)i * n + t] * gs[t]->z;
        U[i] = s;
        pv[i] = 2 * pnorm_upper(fabs(U[i] / sqrt(CovU[i * k + i])));      // gene.cpp:372-377
        rmv[i] = false;
    }
    for (int j = k - 1; j > 0; j--)                                      // gene.cpp:391-399
        for (int i = 0; i < j; i++) if (fabs(CorU[i * k + j]) > a.categ_cor_cutoff) { rmv[j] = true; break; }
    for (int i = 0; i < k; i++) if (CovU[i * k + i] < WWt[i * k + i] / a.denorm_norm_w) rmv[i] = true;   // gene.cpp:408-414
    int df = 0;
    for (int i = 0; i < k; i++) if (!rmv[i]) df++;
    r.df = df;
    if (!df) return r;
    double X[6], CovX[36], Inv[36];
    int ii = 0;
    for (int i = 0; i < k; i++) if (!rmv[i]) X[ii++] = U[i];
    int nn = 0;
    for (int i = 0; i < k; i++) { if (rmv[i]) continue; int mm = 0; for (int j = 0; j < k; j++) { if (rmv[j]) continue; CovX[nn * df + mm] = CovU[i * k + j]; mm++; } nn++; }
    make_pos_def_small(df, CovX, a.min_abs_eig);                        // gene.cpp:493
    inv_small(df, CovX, Inv);                                           // gene.cpp:494
    double cs = 0;
    for (int c = 0; c < df; c++) { double t = 0; for (int q = 0; q < df; q++) t += X[q] * Inv[q * df + c]; cs += t * X[c]; }
    r.chisq = cs;
    r.jepeg_pval = pchisq_upper(cs, df);                                // gene.cpp:509
    int top = 0;                                                        // GetTopCateg, gene.cpp:880-891
    for (int i = 0; i < k; i++) if ((pv[top] > pv[i]) & !rmv[i]) top = i;
    r.top_categ = categ_name(cat[top]);
    r.top_categ_pval = pv[top];
    int ts = 0;                                                         // GetTopSNP, gene.cpp:894-904
    for (int i = 0; i < n; i++) if (fabs(gs[ts]->z) < fabs(gs[i]->z)) ts = i;
    r.top_snp = gs[ts]->rsid;
    r.top_snp_pval = 2 * pnorm_upper(fabs(gs[ts]->z));
    r.geneid = gs[0]->geneid;                                           // gene.cpp:524 (only when df > 0)
    return r;
}

// ------------------------------------------------------------------------------------------
// outputs
// ------------------------------------------------------------------------------------------
gauss_table* dist_output(gauss_prepared& p)     // dist.cpp:91-124 / distmix.cpp:100-133
{
    const Args& a = p.args;
    const bool mix = p.kind == GAUSS_KIND_DISTMIX;
    gauss_table* t = new gauss_table();
    Column rsid{"rsid", GAUSS_COL_STR, {}, {}, {}}, chr{"chr", GAUSS_COL_INT, {}, {}, {}}, bp{"bp", GAUSS_COL_INT, {}, {}, {}};
    Column a1{"a1", GAUSS_COL_STR, {}, {}, {}}, a2{"a2", GAUSS_COL_STR, {}, {}, {}};
    Column af{mix ? "af1mix" : "af1ref", GAUSS_COL_DBL, {}, {}, {}}, z{"z", GAUSS_COL_DBL, {}, {}, {}};
    Column pval{"pval", GAUSS_COL_DBL, {}, {}, {}}, info{"info", GAUSS_COL_DBL, {}, {}, {}}, type{"type", GAUSS_COL_INT, {}, {}, {}};
    size_t n_out = 0;
    for (Snp* s : p.snp_vec) { const int ibp = (int)s->bp; n_out += (ibp >= a.start_bp && ibp <= a.end_bp) ? 1 : 0; }
    for (Column* c : {&rsid, &a1, &a2}) c->s.reserve(n_out);
    for (Column* c : {&chr, &bp, &type}) c->i.reserve(n_out);
    for (Column* c : {&af, &z, &pval, &info}) c->d.reserve(n_out);
    for (Snp* s : p.snp_vec) {
        const int ibp = (int)s->bp;                               // dist.cpp:92
        if (ibp >= a.start_bp && ibp <= a.end_bp) {
            rsid.s.push_back(s->rsid); chr.i.push_back(s->chr); bp.i.push_back(ibp);
            a1.s.push_back(s->a1); a2.s.push_back(s->a2);
            af.d.push_back(mix ? s->af1mix : s->af1ref);
            z.d.push_back(s->z);
            pval.d.push_back(2 * pnorm_upper(fabs(s->z)));        // dist.cpp:101
            info.d.push_back(s->info); type.i.push_back(s->type);
        }
    }
    t->cols.reserve(10);                                          // (moved, not copied: a chromosome's tables are 92 000 rows)
    for (Column* c : {&rsid, &chr, &bp, &a1, &a2, &af, &z, &pval, &info, &type}) t->cols.push_back(std::move(*c));
    return t;
}

gauss_table* qcat_output(gauss_prepared& p)     // qcat.cpp:94-131 / qcatmix.cpp:102-139
{
    const Args& a = p.args;
    const bool mix = p.kind == GAUSS_KIND_QCATMIX;
    gauss_table* t = new gauss_table();
    Column rsid{"rsid", GAUSS_COL_STR, {}, {}, {}}, chr{"chr", GAUSS_COL_INT, {}, {}, {}}, bp{"bp", GAUSS_COL_INT, {}, {}, {}};
    Column a1{"a1", GAUSS_COL_STR, {}, {}, {}}, a2{"a2", GAUSS_COL_STR, {}, {}, {}};
    Column af{mix ? "af1mix" : "af1ref", GAUSS_COL_DBL, {}, {}, {}}, z{"z", GAUSS_COL_DBL, {}, {}, {}};
    Column qm{"qcat_m", GAUSS_COL_INT, {}, {}, {}}, qt{"qcat_t", GAUSS_COL_DBL, {}, {}, {}};
    Column qc{"qcat_chisq", GAUSS_COL_DBL, {}, {}, {}}, qp{"qcat_pval", GAUSS_COL_DBL, {}, {}, {}};
    Column type{"type", GAUSS_COL_INT, {}, {}, {}};
    for (Snp* s : p.snp_vec) {
        const int ibp = (int)s->bp;                               // qcat.cpp:95
        if (ibp >= a.start_bp && ibp <= a.end_bp) {
            rsid.s.push_back(s->rsid); chr.i.push_back(s->chr); bp.i.push_back(ibp);
            a1.s.push_back(s->a1); a2.s.push_back(s->a2);
            af.d.push_back(mix ? s->af1mix : s->af1ref);
            z.d.push_back(s->z);
            qm.i.push_back(s->qcat_m); qt.d.push_back(s->qcat_t); qc.d.push_back(s->qcat_chisq);
            qp.d.push_back(pchisq_upper(s->qcat_chisq, 1));       // qcat.cpp:107
            type.i.push_back(s->type);
        }
    }
    t->cols.reserve(12);
    for (Column* c : {&rsid, &chr, &bp, &a1, &a2, &af, &z, &qm, &qt, &qc, &qp, &type}) t->cols.push_back(std::move(*c));
    return t;
}

static void add_named(gauss_table* t, const char* name, int nrow, int ncol, const double* row_major)
{
    NamedMat m;
    m.name = name; m.nrow = nrow; m.ncol = ncol;
    m.d.resize((size_t)nrow * ncol);
    for (int r = 0; r < nrow; r++)
        for (int c = 0; c < ncol; c++) m.d[(size_t)c * nrow + r] = row_major[(size_t)r * ncol + c];
    t->named.push_back(std::move(m));
}

gauss_table* prep_output(gauss_prepared& p)     // prep_qcat.cpp:135-204 / prep_qcatmix.cpp:262-315
{
    const bool rec = p.kind == GAUSS_KIND_PREP_RECESSIVE;
    const int M = (int)p.measured.size(), U = (int)p.unmeasured.size();
    gauss_table* t = new gauss_table();
    Column rsid{"rsid", GAUSS_COL_STR, {}, {}, {}}, chr{"chr", GAUSS_COL_INT, {}, {}, {}}, bp{"bp", GAUSS_COL_INT, {}, {}, {}};
    Column a1{"a1", GAUSS_COL_STR, {}, {}, {}}, a2{"a2", GAUSS_COL_STR, {}, {}, {}};
    Column af{rec ? "af1mix" : "af1ref", GAUSS_COL_DBL, {}, {}, {}}, z{"z", GAUSS_COL_DBL, {}, {}, {}};
    Column type{"type", GAUSS_COL_INT, {}, {}, {}};
    // prep_qcat lists the whole extended window (prep_qcat.cpp:146-155), prep_recessive_impute only the
    // prediction window (prep_qcatmix.cpp:267-276)
    const std::vector<Snp*>& rows = rec ? p.unmeasured : p.snp_vec;
    for (Snp* s : rows) {
        rsid.s.push_back(s->rsid); chr.i.push_back(s->chr); bp.i.push_back((int)s->bp);
        a1.s.push_back(s->a1); a2.s.push_back(s->a2);
        af.d.push_back(rec ? s->af1mix : s->af1ref);
        z.d.push_back(s->z); type.i.push_back(s->type);
    }
    t->cols = {rsid, chr, bp, a1, a2, af, z, type};
    add_named(t, rec ? "zvec" : "z_vec", M, 1, p.z1.data());
    add_named(t, rec ? "cormat" : "cor_mat1", M, M, p.out_b11.data());
    if (!rec) add_named(t, "cor_mat2", U, M, p.out_b21.data());
    else {
        add_named(t, "cormat_add", U, M, p.out_b21.data());
        add_named(t, "cormat_dom", U, M, p.out_b21.data() + (size_t)U * M);
        add_named(t, "cormat_rec", U, M, p.out_b21.data() + (size_t)2 * U * M);
    }
    return t;
}

extern "C" {

const char* gauss_host_last_error(void) { return g_err.c_str(); }

int gauss_table_nrow(const gauss_table* t) { return t ? t->nrow() : 0; }
int gauss_table_ncol(const gauss_table* t) { return t ? (int)t->cols.size() : 0; }
const char* gauss_table_colname(const gauss_table* t, int c) { return (t && c >= 0 && c < (int)t->cols.size()) ? t->cols[c].name.c_str() : nullptr; }
int gauss_table_coltype(const gauss_table* t, int c) { return (t && c >= 0 && c < (int)t->cols.size()) ? t->cols[c].type : -1; }
const char* gauss_table_str(const gauss_table* t, int c, int r)
{
    if (!t || c < 0 || c >= (int)t->cols.size() || t->cols[c].type != GAUSS_COL_STR || r < 0 || r >= (int)t->cols[c].s.size()) return nullptr;
    return t->cols[c].s[r].c_str();
}
const int32_t* gauss_table_int(const gauss_table* t, int c) { return (t && c >= 0 && c < (int)t->cols.size() && t->cols[c].type == GAUSS_COL_INT) ? t->cols[c].i.data() : nullptr; }
const double* gauss_table_dbl(const gauss_table* t, int c) { return (t && c >= 0 && c < (int)t->cols.size() && t->cols[c].type == GAUSS_COL_DBL) ? t->cols[c].d.data() : nullptr; }
const double* gauss_table_matrix(const gauss_table* t, int* n) { if (!t || t->matrix.empty()) { if (n) *n = 0; return nullptr; } if (n) *n = t->matrix_n; return t->matrix.data(); }
const char* gauss_table_strcol(const gauss_table* t, int c, int64_t* bytes)
{
    if (!t || c < 0 || c >= (int)t->cols.size() || t->cols[c].type != GAUSS_COL_STR) return nullptr;
    const Column& col = t->cols[c];
    if (col.joined.empty() && !col.s.empty()) {
        size_t n = 0;
        for (const std::string& v : col.s) n += v.size() + 1;
        col.joined.reserve(n);
        for (const std::string& v : col.s) { col.joined += v; col.joined.push_back('\0'); }
    }
    if (bytes) *bytes = (int64_t)col.joined.size();
    return col.joined.data();
}
void gauss_table_free(gauss_table* t) { delete t; }
int gauss_table_n_named(const gauss_table* t) { return t ? (int)t->named.size() : 0; }
const char* gauss_table_named_name(const gauss_table* t, int k) { return (t && k >= 0 && k < (int)t->named.size()) ? t->named[k].name.c_str() : nullptr; }
const double* gauss_table_named(const gauss_table* t, int k, int* nrow, int* ncol)
{
    if (!t || k < 0 || k >= (int)t->named.size()) return nullptr;
    if (nrow) *nrow = t->named[k].nrow;
    if (ncol) *ncol = t->named[k].ncol;
    return t->named[k].d.data();
}

// The JEPEG k x k tail of one gene on the host, as run_jepeg calls it after the GPU has produced CorG: exposed so that
// it can be checked on its own (no GPU involved).  corg is n x n (symmetric), has / wgt are n x 6 row-major.
int gauss_host_jepeg_gene_tail(int n, const double* corg, const double* z, const double* info, const int32_t* has,
                               const double* wgt, double* chisq, int32_t* df, double* jepeg_pval, int32_t* top_categ,
                               double* top_categ_pval, int32_t* top_snp, double* top_snp_pval)
{
    if (n < 0 || (n > 0 && (!corg || !z || !info || !has || !wgt))) return herr("bad arguments");
    Args a;
    std::vector<std::unique_ptr<Snp>> own;
    std::vector<Snp*> gs;
    for (int s = 0; s < n; s++) {
        own.emplace_back(new Snp());
        Snp& sn = *own.back();
        sn.rsid = std::to_string(s); sn.z = z[s]; sn.info = info[s]; sn.geneid = "G";
        for (int c = 0; c < 6; c++) if (has[(size_t)s * 6 + c]) sn.categ[c] = wgt[(size_t)s * 6 + c];
        gs.push_back(&sn);
    }
    const GeneResult r = jepeg_tail(gs, corg, a);
    if (chisq) *chisq = r.chisq;
    if (df) *df = r.df;
    if (jepeg_pval) *jepeg_pval = r.jepeg_pval;
    if (top_categ) { *top_categ = -1; for (int c = 0; c < 6; c++) if (r.top_categ == categ_name(c)) *top_categ = c; }
    if (top_categ_pval) *top_categ_pval = r.top_categ_pval;
    if (top_snp) *top_snp = (r.top_snp == ".") ? -1 : atoi(r.top_snp.c_str());
    if (top_snp_pval) *top_snp_pval = r.top_snp_pval;
    return 0;
}

int gauss_table_n_messages(const gauss_table* t) { return t ? (int)t->messages.size() : 0; }
const char* gauss_table_message(const gauss_table* t, int k) { return (t && k >= 0 && k < (int)t->messages.size()) ? t->messages[k].c_str() : nullptr; }

}  // extern "C"
void build_fixed_image(const Column& col)
{
    size_t w = 1;
    for (const std::string& v : col.s) w = std::max(w, v.size());
    if (col.fixed.size() == w * col.s.size() && col.fixed_w == (int)w && !col.s.empty()) return;       // made already (rows are not edited afterwards)
    col.fixed.assign(w * col.s.size(), '\0');
    for (size_t r = 0; r < col.s.size(); r++) memcpy(&col.fixed[r * w], col.s[r].data(), col.s[r].size());
    col.fixed_w = (int)w;
}
extern "C" {
// A whole string column as one fixed-width, NUL-padded byte matrix [nrow x *width] (numpy dtype "S<width>"):
// 90 000 rows come across the boundary as one buffer instead of 90 000 Python strings.
const char* gauss_table_strcol_fixed(const gauss_table* t, int c, int* width)
{
    if (!t || c < 0 || c >= (int)t->cols.size() || t->cols[c].type != GAUSS_COL_STR) return nullptr;
    const Column& col = t->cols[c];
    build_fixed_image(col);
    if (width) *width = col.fixed_w;
    return col.fixed.data();
}

}  // extern "C"
