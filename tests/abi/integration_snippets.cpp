// Compile check (g++ -fsyntax-only) of the driver-side code shown in INTEGRATION.md against the real headers:
// a mock of the few reference types the snippets touch (Snp, Arguments, Rcpp::stop) stands in for Rcpp, which
// is absent from the build image.  Keeps the documented bindings in step with include/gauss_hip.h.
#include <cmath>
#include <cstring>
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#include "gauss_hip.h"

namespace Rcpp { inline void stop(const std::string& m) { throw std::runtime_error(m); } }

struct Snp {                               // the accessors of src/snp.h used below
    std::vector<std::string> geno;
    double z = 0, info = 0, qt = 0, qc = 0;
    int qm = 0;
    std::vector<std::string>& GetGenotypeVec() { return geno; }
    double GetZ() { return z; }
    void SetZ(double v) { z = v; }
    void SetInfo(double v) { info = v; }
    void SetQcatM(int v) { qm = v; }
    void SetQcatT(double v) { qt = v; }
    void SetQcatChisq(double v) { qc = v; }
};
struct Arguments { double lambda = 0.1, min_abs_eig = 1e-5, eig_cutoff = 0.01; std::vector<double> pop_wgt_vec; };

// ---- INTEGRATION.md section 2 ----
inline gauss_ctx* gauss_hip_ctx() {
  static gauss_ctx* ctx = nullptr;
  if (!ctx && gauss_hip_init(0, &ctx) != GAUSS_OK) Rcpp::stop(gauss_last_error());
  return ctx;
}
inline void pack_genotypes(std::deque<Snp*>& snps, std::vector<uint8_t>& G, int64_t& ld,
                           std::vector<int32_t>& pop_off) {
  std::vector<std::string>& first = snps.front()->GetGenotypeVec();
  pop_off.assign(1, 0);
  for (auto& s : first) pop_off.push_back(pop_off.back() + (int32_t)s.size());
  ld = pop_off.back();
  G.resize((size_t)snps.size() * ld);
  for (size_t r = 0; r < snps.size(); r++) {
    uint8_t* dst = &G[r * ld];
    for (auto& s : snps[r]->GetGenotypeVec()) { memcpy(dst, s.data(), s.size()); dst += s.size(); }
  }
}

// ---- section 3: run_dist / run_distmix ----
void run_dist_body(std::deque<Snp*>& sliding_window_measured, std::deque<Snp*>& sliding_window_unmeasured, Arguments& args) {
  const int num_measured = (int)sliding_window_measured.size(), num_unmeasured = (int)sliding_window_unmeasured.size();
  std::vector<uint8_t> Gm, Gu; std::vector<int32_t> pop_off, pop_off_u; int64_t ld, ld_u;
  pack_genotypes(sliding_window_measured, Gm, ld, pop_off);
  pack_genotypes(sliding_window_unmeasured, Gu, ld_u, pop_off_u);
  std::vector<double> z1(num_measured), z(num_unmeasured), info(num_unmeasured);
  for (int i = 0; i < num_measured; i++) z1[i] = sliding_window_measured[i]->GetZ();
  int32_t status = 0;
  gauss_window_desc w = {};
  w.mode = GAUSS_MODE_POOLED;
  w.n_pop = (int)pop_off.size() - 1;  w.pop_off = pop_off.data();
  w.pop_wgt = NULL;
  w.n_measured = num_measured;  w.n_unmeasured = num_unmeasured;
  w.geno_m = Gm.data();  w.geno_u = Gu.data();  w.ld = ld;
  w.z1 = z1.data();  w.lambda = args.lambda;  w.min_abs_eig = args.min_abs_eig;
  w.out_z = z.data();  w.out_info = info.data();  w.out_status = &status;
  if (gauss_impute_window(gauss_hip_ctx(), &w) != GAUSS_OK) Rcpp::stop(gauss_last_error());
  for (int i = 0; i < num_unmeasured; i++) {
    sliding_window_unmeasured[i]->SetZ(z[i]);
    sliding_window_unmeasured[i]->SetInfo(info[i]);
  }
}

// ---- section 4: computeLD ----
void computeLD_body(std::deque<Snp*>& rows, Arguments& args, double* Cor_Mat /* &Cor_Mat[0] */) {
  const int num_measured = (int)rows.size();
  std::vector<uint8_t> G; std::vector<int32_t> pop_off; int64_t ld;
  pack_genotypes(rows, G, ld, pop_off);
  if (gauss_ld(gauss_hip_ctx(), GAUSS_MODE_WEIGHTED, G.data(), num_measured, ld, pop_off.data(),
               args.pop_wgt_vec.data(), (int)pop_off.size() - 1, /*diag=*/1.0, Cor_Mat) != GAUSS_OK)
    Rcpp::stop(gauss_last_error());
}

// ---- section 5: jepeg / jepegmix ----
void jepeg_body(std::deque<Snp*>& rows, std::vector<int32_t>& gene_off, Arguments& args, std::vector<double>& blocks) {
  std::vector<uint8_t> G; std::vector<int32_t> pop_off; int64_t ld;
  pack_genotypes(rows, G, ld, pop_off);
  size_t total = 0; for (size_t g = 0; g + 1 < gene_off.size(); g++) { size_t n = gene_off[g+1]-gene_off[g]; total += n*n; }
  blocks.resize(total);
  if (gauss_gene_ld_batch(gauss_hip_ctx(), GAUSS_MODE_POOLED, G.data(),
                          (int)rows.size(), ld, pop_off.data(), /*pop_wgt*/NULL, (int)pop_off.size()-1,
                          gene_off.data(), (int)gene_off.size()-1, 1.0 + args.lambda, blocks.data()) != GAUSS_OK)
    Rcpp::stop(gauss_last_error());
}

// ---- section 5b: run_qcat / run_qcatmix ----
void run_qcat_body(std::deque<Snp*>& sliding_window_measured_ext, std::deque<Snp*>& sliding_window_unmeasured_pred,
                   int num_measured_headwing, int num_measured_pred, Arguments& args) {
  const int num_measured_ext = (int)sliding_window_measured_ext.size(), num_unmeasured_pred = (int)sliding_window_unmeasured_pred.size();
  std::vector<uint8_t> Gm, Gu; std::vector<int32_t> pop_off, pop_off_u; int64_t ld, ld_u;
  pack_genotypes(sliding_window_measured_ext, Gm, ld, pop_off);
  pack_genotypes(sliding_window_unmeasured_pred, Gu, ld_u, pop_off_u);
  std::vector<double> z1(num_measured_ext), r(num_measured_pred + num_unmeasured_pred);
  for (int i = 0; i < num_measured_ext; i++) z1[i] = sliding_window_measured_ext[i]->GetZ();
  int32_t status = 0, num_eig = 0;
  gauss_window_desc w = {};
  w.kind = GAUSS_WIN_QCAT;
  w.mode = GAUSS_MODE_POOLED;
  w.n_pop = (int)pop_off.size() - 1;  w.pop_off = pop_off.data();
  w.n_measured = num_measured_ext;  w.n_unmeasured = num_unmeasured_pred;
  w.geno_m = Gm.data();  w.geno_u = Gu.data();  w.ld = ld;
  w.z1 = z1.data();  w.lambda = args.lambda;  w.eig_cutoff = args.eig_cutoff;
  w.n_head_measured = num_measured_headwing;  w.n_pred_measured = num_measured_pred;
  w.out_r = r.data();  w.out_num_eig = &num_eig;  w.out_status = &status;
  if (gauss_impute_window(gauss_hip_ctx(), &w) != GAUSS_OK) Rcpp::stop(gauss_last_error());
  for (int t = 0; t < (int)r.size(); t++) {
    Snp* s = t < num_measured_pred ? sliding_window_measured_ext[t + num_measured_headwing]
                                   : sliding_window_unmeasured_pred[t - num_measured_pred];
    s->SetQcatM(num_eig);
    s->SetQcatT(std::sqrt(num_eig - 3) * r[t]);
    s->SetQcatChisq((num_eig - 3) * r[t] * r[t]);
  }
}

// ---- section 5c: prep_qcat / prep_recessive_impute ----
void prep_recessive_body(std::deque<Snp*>& ext_window_measured, std::deque<Snp*>& pred_window_all, Arguments& args,
                         double* cormat /* &cormat[0] */) {
  const int num_measured_ext = (int)ext_window_measured.size(), num_all_pred = (int)pred_window_all.size();
  std::vector<uint8_t> Gm, Gu; std::vector<int32_t> pop_off, pop_off_u; int64_t ld, ld_u;
  int32_t status = 0;
  pack_genotypes(ext_window_measured, Gm, ld, pop_off);
  pack_genotypes(pred_window_all, Gu, ld_u, pop_off_u);
  std::vector<double> b21((size_t)3 * num_all_pred * num_measured_ext);
  gauss_window_desc w = {};
  w.kind = GAUSS_WIN_LD;  w.mode = GAUSS_MODE_WEIGHTED;  w.lambda = 0.0;
  w.u_codings = GAUSS_CODE_ADDITIVE | GAUSS_CODE_DOMINANT | GAUSS_CODE_RECESSIVE;
  w.n_pop = (int)pop_off.size() - 1;  w.pop_off = pop_off.data();  w.pop_wgt = args.pop_wgt_vec.data();
  w.n_measured = num_measured_ext;  w.n_unmeasured = num_all_pred;
  w.geno_m = Gm.data();  w.geno_u = Gu.data();  w.ld = ld;
  w.out_b11 = cormat;  w.out_b21 = b21.data();  w.out_status = &status;
  if (gauss_impute_window(gauss_hip_ctx(), &w) != GAUSS_OK) Rcpp::stop(gauss_last_error());
}

// ---- section 6: prep_zmix5 pair table ----
void prep_zmix5_body(std::deque<Snp*>& snp_subvec, double* data_mat_col1 /* &data_mat(0, 1) */) {
  std::vector<uint8_t> G; std::vector<int32_t> pop_off; int64_t ld;
  pack_genotypes(snp_subvec, G, ld, pop_off);
  if (gauss_ld_per_pop(gauss_hip_ctx(), G.data(), (int)snp_subvec.size(), ld, pop_off.data(), (int)pop_off.size() - 1,
                       data_mat_col1) != GAUSS_OK)
    Rcpp::stop(gauss_last_error());
}

// ---- section 5d: packed rows + resident store ----
void packed_window_body(gauss_ctx* ctx, const uint8_t* mapped_rows, int64_t row_bytes, int64_t n_rows,
                        std::vector<int32_t>& rows_m, std::vector<int32_t>& rows_u, std::vector<int32_t>& pop_off,
                        std::vector<int32_t>& pop_src_off, std::vector<double>& z1, std::vector<double>& z,
                        std::vector<double>& info) {
  void* dev = nullptr;
  if (gauss_store_upload(ctx, mapped_rows, n_rows * row_bytes, &dev) != GAUSS_OK) Rcpp::stop(gauss_last_error());
  int32_t status = 0;
  gauss_window_desc w = {};
  w.mode = GAUSS_MODE_POOLED;  w.n_pop = (int)pop_off.size() - 1;  w.pop_off = pop_off.data();
  w.n_measured = (int)rows_m.size();  w.n_unmeasured = (int)rows_u.size();
  w.geno_format = GAUSS_GENO_2BIT;  w.geno_m = w.geno_u = (const uint8_t*)dev;  w.ld = row_bytes;
  w.rows_m = rows_m.data();  w.rows_u = rows_u.data();  w.pop_src_off = pop_src_off.data();
  w.z1 = z1.data();  w.lambda = 0.1;  w.min_abs_eig = 1e-5;
  w.out_z = z.data();  w.out_info = info.data();  w.out_status = &status;
  gauss_job* job = nullptr;
  if (gauss_job_create(ctx, &w, 1, /*on_device=*/1, &job) != GAUSS_OK) Rcpp::stop(gauss_last_error());
  if (gauss_job_run(job) != GAUSS_OK || gauss_job_fetch(job) != GAUSS_OK) Rcpp::stop(gauss_last_error());
  gauss_job_destroy(job);
  gauss_store_free(ctx, dev);
}


// ---- section 3b: handles that survive Rcpp::stop ----
struct HipContext {
  gauss_ctx* h = nullptr;
  explicit HipContext(int device = 0) { if (gauss_hip_init(device, &h) != GAUSS_OK) Rcpp::stop(gauss_last_error()); }
  ~HipContext() { gauss_hip_destroy(h); }
  HipContext(const HipContext&) = delete;  HipContext& operator=(const HipContext&) = delete;
};
struct HipJob {
  gauss_job* h = nullptr;
  HipJob(gauss_ctx* ctx, const std::vector<gauss_window_desc>& wins) {
    if (gauss_job_create(ctx, wins.data(), (int)wins.size(), 0, &h) != GAUSS_OK) Rcpp::stop(gauss_last_error());
  }
  ~HipJob() { gauss_job_destroy(h); }
  void run()   { if (gauss_job_run(h)   != GAUSS_OK) Rcpp::stop(gauss_last_error()); }
  void fetch() { if (gauss_job_fetch(h) != GAUSS_OK) Rcpp::stop(gauss_last_error()); }
  HipJob(const HipJob&) = delete;  HipJob& operator=(const HipJob&) = delete;
};
void impute_windows(std::vector<gauss_window_desc>& wins) {
  HipContext ctx;
  HipJob job(ctx.h, wins);
  job.run();
  job.fetch();
}

// ---- section 5e: a whole chromosome per call (include/gauss_host.h) ----
#include "gauss_host.h"
struct ChromTable { std::vector<std::string> rsid; std::vector<int> bp, window; std::vector<double> z, info; };
ChromTable distmix_chromosome_body(int chr, long long start_bp, long long end_bp, long long wing_size, long long window_size,
                                   const std::vector<std::string>& names, const std::vector<double>& wgts,
                                   const std::string& input_file, const std::string& packed_panel, const std::string& desc) {
  std::vector<const char*> np;
  for (auto& s : names) np.push_back(s.c_str());
  gauss_table* t = nullptr;
  gauss_chrom_stats st;
  if (gauss_host_impute_chromosome(gauss_hip_ctx(), GAUSS_KIND_DISTMIX, chr, start_bp, end_bp, wing_size, window_size, NULL,
                                   np.data(), wgts.data(), (int)np.size(), input_file.c_str(), /*index: only for a text panel*/NULL, packed_panel.c_str(), desc.c_str(),
                                   NAN, /*rank*/0, /*world*/1, /*n_batches*/0, &t, &st) != 0)
    Rcpp::stop(gauss_host_last_error());
  ChromTable out;
  const int n = gauss_table_nrow(t);
  for (int c = 0; c < gauss_table_ncol(t); c++) {
    const std::string name = gauss_table_colname(t, c);
    if (name == "rsid") for (int r = 0; r < n; r++) out.rsid.push_back(gauss_table_str(t, c, r));
    else if (name == "bp") out.bp.assign(gauss_table_int(t, c), gauss_table_int(t, c) + n);
    else if (name == "window") out.window.assign(gauss_table_int(t, c), gauss_table_int(t, c) + n);
    else if (name == "z") out.z.assign(gauss_table_dbl(t, c), gauss_table_dbl(t, c) + n);
    else if (name == "info") out.info.assign(gauss_table_dbl(t, c), gauss_table_dbl(t, c) + n);
  }
  for (int k = 0; k < gauss_table_n_messages(t); k++) (void)gauss_table_message(t, k);     // one text per failed window
  gauss_table_free(t);
  return out;
}

// ---- section 5f: a genome per call (two chromosome calls in flight on the context) ----
std::vector<long long> distmix_genome_body(const std::vector<int>& chr, const std::vector<long long>& start_bp, const std::vector<long long>& end_bp,
                                           long long wing_size, long long window_size, const std::vector<std::string>& names,
                                           const std::vector<double>& wgts, const std::string& input_file, const std::string& packed_panel,
                                           const std::string& desc, int rank, int world) {
  const int n = (int)chr.size();
  std::vector<int32_t> c(chr.begin(), chr.end());
  std::vector<int64_t> lo(start_bp.begin(), start_bp.end()), hi(end_bp.begin(), end_bp.end());
  std::vector<const char*> np;
  for (auto& s : names) np.push_back(s.c_str());
  std::vector<gauss_table*> t((size_t)n, nullptr);
  std::vector<gauss_chrom_stats> st((size_t)n);
  const int rc = gauss_host_impute_genome(gauss_hip_ctx(), GAUSS_KIND_DISTMIX, n, c.data(), lo.data(), hi.data(), wing_size, window_size, NULL,
                                          np.data(), wgts.data(), (int)np.size(), input_file.c_str(), NULL, packed_panel.c_str(), desc.c_str(), NAN,
                                          rank, world, /*depth*/2, t.data(), st.data());
  std::vector<long long> imputed;
  for (int k = 0; k < n; k++) {
    imputed.push_back(t[(size_t)k] ? (long long)st[(size_t)k].imputed : -1);       // -1: this chromosome's call failed, the others are complete
    if (st[(size_t)k].n_merged_giveups) (void)0;                                    // batches repaired inside their fetch (normally 0)
    gauss_table_free(t[(size_t)k]);
  }
  int64_t counters[4];
  (void)gauss_hip_counters(gauss_hip_ctx(), counters);                              // merged / demoted / give-ups / failed re-runs
  if (rc != 0) (void)gauss_host_last_error();
  return imputed;
}

// ---- section 5g: jepeg() / jepegmix() on several GPUs (gene ranges of one call; whole calls dealt to ranks) ----
int jepegmix_rank_body(const std::vector<std::string>& names, const std::vector<double>& wgts, const std::string& input_file,
                       const std::string& annotation_file, const std::string& index_file, const std::string& data_file,
                       const std::string& desc, int rank, int world) {
  std::vector<const char*> np;
  for (auto& s : names) np.push_back(s.c_str());
  gauss_table* t = nullptr;
  if (gauss_host_jepeg_rank(gauss_hip_ctx(), GAUSS_KIND_JEPEGMIX, NULL, np.data(), wgts.data(), (int)np.size(), input_file.c_str(),
                            annotation_file.c_str(), index_file.c_str(), data_file.c_str(), desc.c_str(), NAN, rank, world, &t) != 0)
    Rcpp::stop(gauss_host_last_error());
  int nr = 0, nc = 0;
  const double* range = nullptr;
  for (int k = 0; k < gauss_table_n_named(t); k++)
    if (std::string(gauss_table_named_name(t, k)) == "gene_range") range = gauss_table_named(t, k, &nr, &nc);     // [first, one past last, genes in all]
  const int genes_mine = range ? (int)(range[1] - range[0]) : 0;
  gauss_table_free(t);
  return genes_mine;
}

int jepegmix_genome_body(const std::vector<std::string>& names, const std::vector<double>& wgts, const std::vector<std::string>& inputs,
                         const std::vector<std::string>& annots, const std::vector<std::string>& panels, const std::string& desc,
                         int rank, int world) {
  std::vector<const char*> np, in, an, pa;
  for (auto& s : names) np.push_back(s.c_str());
  for (auto& s : inputs) in.push_back(s.c_str());
  for (auto& s : annots) an.push_back(s.c_str());
  for (auto& s : panels) pa.push_back(s.c_str());
  const int n = (int)in.size();
  std::vector<gauss_table*> out((size_t)n, nullptr);
  std::vector<int32_t> owner((size_t)n, -1);
  const int rc = gauss_host_jepeg_genome(gauss_hip_ctx(), GAUSS_KIND_JEPEGMIX, n, NULL, np.data(), wgts.data(), (int)np.size(), in.data(), an.data(),
                                         /*index files: packed panels need none*/NULL, pa.data(), desc.c_str(), NAN, rank, world, out.data(), owner.data());
  int mine = 0;
  for (int c = 0; c < n; c++) { if (out[(size_t)c]) mine++; gauss_table_free(out[(size_t)c]); }
  if (rc != 0) (void)gauss_host_last_error();                     // the first failing call's message; the other calls are complete
  return mine;
}
