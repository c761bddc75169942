// libgauss_host.so -- what its translation units share: the data model of the reference's feeder (src/snp.h:14-110,
// src/gauss.h:18-99), result tables, the arena a window's SNP map lives in, the cached file images, and the prototypes of
// the routines that cross files.  Nothing here is exported: the library is built with -fvisibility=hidden and only the C ABI
// of include/gauss_host.h is visible.
//   host_feeder.cpp   the reference's readers and filters (ReadInputZ, ReadReferenceIndex, MakeSnpVec, ReadAnnotation, ...), the
//                     packed-panel cache, prepare()
//   host_tables.cpp   result tables, the JEPEG k x k tail, the table accessors of the C ABI
//   host_calls.cpp    the one-window / one-call entry points (computeLD, dist, distmix, qcat, prep_*, jepeg, jepegmix)
//   host_chrom.cpp    resident panels, the chromosome driver's own window, gauss_host_impute_chromosome / _genome
#pragma once
#pragma GCC visibility push(default)
#include "../../../include/gauss_host.h"
#pragma GCC visibility pop

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/file.h>
#include <fcntl.h>
#include <unistd.h>
#include <cerrno>

#include <mutex>

#include "bgzf_io.h"
#include "packed_panel.h"

using gauss_host::BgzfReader;
using gauss_host::PackedPanel;
using gauss_host::PkSnp;

// ------------------------------------------------------------------------------------------
extern thread_local std::string g_err;          // host_tables.cpp; gauss_host_last_error() returns it
int herr(const char* fmt, ...);

// Diagnostics on stderr: GAUSS_TRACE=chrom,prep (any subset, or "all"; libgauss_hip reads job, upload, stream from the same variable)
inline bool host_trace(const char* what)
{
    const char* e = getenv("GAUSS_TRACE");
    return e && *e && (strstr(e, "all") != nullptr || strstr(e, what) != nullptr);
}

// ------------------------------------------------------------------------------------------
// tables
// ------------------------------------------------------------------------------------------
struct Column {
    std::string name;
    int type;
    std::vector<std::string> s;
    mutable std::string joined;      // lazily built NUL-separated image of s (gauss_table_strcol)
    mutable std::vector<char> fixed; // lazily built fixed-width image of s (gauss_table_strcol_fixed)
    mutable int fixed_w = 0;
    std::vector<int32_t> i;
    std::vector<double> d;
};

void build_fixed_image(const Column& col);      // Column::fixed / fixed_w (host_tables.cpp)

struct NamedMat {
    std::string name;
    int nrow = 0, ncol = 0;
    std::vector<double> d;        // column-major (R NumericMatrix layout)
};

struct gauss_table {
    std::vector<Column> cols;
    std::vector<double> matrix;
    int matrix_n = 0;
    std::vector<NamedMat> named;
    std::vector<std::string> messages;   // per-window failure texts of a chromosome run
    int nrow() const
    {
        if (cols.empty()) return 0;
        const Column& c = cols[0];
        return (int)(c.type == GAUSS_COL_STR ? c.s.size() : c.type == GAUSS_COL_INT ? c.i.size() : c.d.size());
    }
    Column& add(const char* name, int type) { Column c; c.name = name; c.type = type; cols.push_back(std::move(c)); return cols.back(); }
};

// ------------------------------------------------------------------------------------------
// data model (src/snp.h:14-110, src/gauss.h:18-99)
// ------------------------------------------------------------------------------------------
struct Snp {
    std::string rsid = ".";
    int chr = -1;
    long long bp = -1;
    std::string a1 = ".", a2 = ".";
    double af1mix = -1.0, af1ref = -1.0;
    double z = 0.0, info = -1.0;
    int qcat_m = 0;                // snp.cpp:26-28
    double qcat_t = 0.0, qcat_chisq = 0.0;
    int type = -1;                 // 0 panel only, 1 GWAS and panel, 2 GWAS only (snp.h:61)
    long long fpos = -1;
    std::string geneid = ".";
    std::map<int, double> categ;   // Snp::categ_map_
    bool flip_geno = false;        // UpdateSnpToMinorAllele (gauss.cpp:1137-1184): genotype d -> 2 - d
    std::string line;              // cached panel data line (read once instead of twice)
    bool have_line = false;
    std::vector<std::pair<const char*, int>> geno;   // selected populations' genotype strings (into `line`)
};

struct MapKey {
    int chr; long long bp; std::string a1, a2;
    bool operator<(const MapKey& r) const     // gauss.h:77-91
    {
        if (chr == r.chr) {
            if (bp == r.bp) {
                if (a1 == r.a1) return a2 < r.a2;
                return a1 < r.a1;
            }
            return bp < r.bp;
        }
        return chr < r.chr;
    }
};
// A window's SNP map lives in blocks of its own instead of the C library's heap.  A 100 000-SNP chromosome enters ~126 000 Snp
// objects and as many map nodes (~480 B a SNP, ~60 MB over the 36 windows); on the FIRST call of a process every page of that is
// touched for the first time -- ~14 000 minor faults, 40 ms of kernel time next to 50 ms of user time for the whole data layer,
// spread over the worker threads' fresh malloc arenas (measured, docs/HISTORY.md section 9e item 9).  Blocks are 2 MB, 2 MB-aligned,
// advised as huge pages and populated in one call (one fault or one batched population instead of 512 traps); a window frees
// nothing one by one -- its blocks go back to a process-wide list when the window is closed, so later calls touch no new page.
struct BlockPool {
    std::mutex mu;
    std::vector<void*> idle;
    size_t keep;
    size_t block;
    BlockPool(size_t block_bytes, size_t keep_bytes) : keep(keep_bytes / block_bytes), block(block_bytes) {}
    void* get()
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            if (!idle.empty()) { void* b = idle.back(); idle.pop_back(); return b; }
        }
        char* raw = (char*)mmap(nullptr, 2 * block, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (raw == (char*)MAP_FAILED) throw std::bad_alloc();
        char* b = (char*)(((uintptr_t)raw + block - 1) & ~(uintptr_t)(block - 1));
        if (b > raw) munmap(raw, (size_t)(b - raw));
        if (b + block < raw + 2 * block) munmap(b + block, (size_t)(raw + 2 * block - (b + block)));
#ifdef MADV_HUGEPAGE
        if (block >= ((size_t)2 << 20)) madvise(b, block, MADV_HUGEPAGE);   // advice only: without huge pages the block is 512 small ones
#endif
#ifdef MADV_POPULATE_WRITE
        madvise(b, block, MADV_POPULATE_WRITE);                         // Linux 5.14+; an older kernel faults the pages in on first use
#endif
        return b;
    }
    void put(void* b)
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            if (idle.size() < keep) { idle.push_back(b); return; }
        }
        munmap(b, block);
    }
};
// Two sizes: a window's FIRST block is small (128 KB: a fine-grained run -- thousands of windows of a few hundred SNPs -- must not
// hold 2 MB apiece), everything after it comes in 2 MB blocks.  GAUSS_HOST_ARENA_KEEP_MB: idle memory kept for the next call
// (default 256 MB in large blocks + 32 MB in small ones); GAUSS_HOST_ARENA_BLOCK_KB (tests): the large block's size, so that a
// small window spans several.  Never destroyed: windows may outlive static destructors.
struct BlockPools {
    BlockPool* small_;
    BlockPool* large_;
    BlockPools()
    {
        size_t large = (size_t)2 << 20;
        if (const char* b = getenv("GAUSS_HOST_ARENA_BLOCK_KB")) {
            size_t kb = 64;
            while (kb < (size_t)std::max(64, atoi(b)) && kb < 2048) kb *= 2;
            large = kb << 10;
        }
        const char* e = getenv("GAUSS_HOST_ARENA_KEEP_MB");
        const size_t keep = (size_t)(e ? std::max(0, atoi(e)) : 256) << 20;
        large_ = new BlockPool(large, keep);
        small_ = new BlockPool(std::min<size_t>(large, (size_t)128 << 10), keep / 8);
    }
};
BlockPools& block_pools();      // host_feeder.cpp (one per process, never destroyed)

struct Arena {
    std::vector<std::pair<void*, BlockPool*>> blocks;
    std::vector<void*> big;
    char* cur = nullptr;
    size_t left = 0;
    Arena() = default;
    Arena(const Arena&) = delete;
    Arena& operator=(const Arena&) = delete;
    void* alloc(size_t n)
    {
        n = (n + 15) & ~(size_t)15;
        if (n > left) {
            BlockPools& bp = block_pools();
            BlockPool* from = blocks.empty() ? bp.small_ : bp.large_;
            if (n > from->block / 8) { void* q = ::operator new(n); big.push_back(q); return q; }
            cur = (char*)from->get();
            blocks.emplace_back(cur, from);
            left = from->block;
        }
        void* q = cur;
        cur += n; left -= n;
        return q;
    }
    ~Arena()
    {
        for (auto& b : blocks) b.second->put(b.first);
        for (void* q : big) ::operator delete(q);
    }
};

template <class T>
struct ArenaAlloc {
    typedef T value_type;
    Arena* a;
    explicit ArenaAlloc(Arena* arena) : a(arena) {}
    template <class U> ArenaAlloc(const ArenaAlloc<U>& o) : a(o.a) {}
    T* allocate(size_t n) { return (T*)a->alloc(n * sizeof(T)); }
    void deallocate(T*, size_t) {}                                      // the blocks go back as a whole
    template <class U> bool operator==(const ArenaAlloc<U>& o) const { return a == o.a; }
    template <class U> bool operator!=(const ArenaAlloc<U>& o) const { return a != o.a; }
};

struct SnpDestroy { void operator()(Snp* s) const { s->~Snp(); } };     // storage is the arena's
typedef std::unique_ptr<Snp, SnpDestroy> SnpPtr;
struct SnpMapArena { Arena arena; };
typedef std::map<MapKey, SnpPtr, std::less<MapKey>, ArenaAlloc<std::pair<const MapKey, SnpPtr>>> SnpMapBase;
struct SnpMap : private SnpMapArena, public SnpMapBase {               // the arena is built before the map and outlives it
    SnpMap() : SnpMapBase(std::less<MapKey>(), ArenaAlloc<std::pair<const MapKey, SnpPtr>>(&arena)) {}
    SnpMap(const SnpMap&) = delete;
    SnpMap& operator=(const SnpMap&) = delete;
    SnpPtr make() { return SnpPtr(new (arena.alloc(sizeof(Snp))) Snp()); }
};

struct Args {                       // Arguments, gauss.h:18-69 with the defaults of gauss.cpp:18-35
    int chr = 0;
    long long start_bp = 0, end_bp = 0, wing_size = 0;
    std::string study_pop, input_file, reference_index_file, reference_data_file, reference_pop_desc_file, annotation_file;
    std::vector<std::string> ref_pop_vec, ref_sup_pop_vec;
    std::vector<int> ref_pop_size_vec;
    double lambda = 0.1, min_abs_eig = 1e-5, eig_cutoff = 0.01;
    std::vector<int> pop_flag_vec;
    std::vector<double> pop_wgt_vec;
    std::map<std::string, double> pop_wgt_map;
    int num_pops = 0, num_samples = 0;
    double af1_cutoff = 0.01;
    int min_num_measured_snp = 10, min_num_unmeasured_snp = 10;
    std::shared_ptr<PackedPanel> pk;   // set when reference_data_file is a packed panel (packed_panel.h)
    // dist / distmix on a packed panel: a panel SNP that no study SNP shares its position with and that lies in a wing
    // would enter the map as type 0 (gauss.cpp:373-385), pass the AF filter and then be dropped by the partition
    // (dist.cpp:132-140 imputes type-0 SNPs of the prediction window only; the output is cut to it, dist.cpp:91-93) --
    // nothing ever reads it, so it is not entered at all (about half of an extended window's panel SNPs)
    bool drop_wing_unmeasured = false;
    // jepeg / jepegmix drivers: only study SNPs at positions the annotation names (and the study's odd positions, above) enter the SNP
    // map -- every step after ReadInputZ touches entries of ONE position at a time, the gene table is made of annotated SNPs alone,
    // and a chromosome's study is four times the annotated SNPs (host_calls.cpp:run_jepeg; GAUSS_HOST_FULL_MAP=1: the whole study)
    bool annotated_only = false;
    int total_num_categ = 6;
    double categ_cor_cutoff = 0.8;
    int denorm_norm_w = 3;
};

// whitespace tokeniser with the semantics of `istringstream >> a >> b ...` for well-formed lines
struct Tok {
    const char* p; const char* e;
    explicit Tok(const std::string& s) : p(s.data()), e(s.data() + s.size()) {}
    // whitespace as std::isspace in the "C" locale (what operator>> skips), from a table: panel lines are ~33 kB
    // of genotype digits and this scan runs over every byte of them
    static const bool* ws_table()
    {
        static const struct T { bool t[256]; T() { for (int c = 0; c < 256; c++) t[c] = (c == ' ' || (c >= 9 && c <= 13)); } } tab;
        return tab.t;
    }
    bool next(const char*& b, int& n)
    {
        const bool* ws = ws_table();
        while (p < e && ws[(unsigned char)*p]) p++;
        if (p >= e) return false;
        b = p;
        // long tokens (a population's genotype string) end at a blank in practice: let memchr find it, then
        // make sure no other white-space character came first
        const char* q = (const char*)memchr(p, ' ', (size_t)(e - p));
        const char* lim = q ? q : e;
        const char* r = p;
        while (r < lim && !ws[(unsigned char)*r]) {
            // skip ahead in blocks of 8 digits while there is room
            if (lim - r >= 8 && !(ws[(unsigned char)r[0]] | ws[(unsigned char)r[1]] | ws[(unsigned char)r[2]] | ws[(unsigned char)r[3]] |
                                  ws[(unsigned char)r[4]] | ws[(unsigned char)r[5]] | ws[(unsigned char)r[6]] | ws[(unsigned char)r[7]])) r += 8;
            else r++;
        }
        p = r;
        n = (int)(p - b);
        return true;
    }
    bool str(std::string& out) { const char* b; int n; if (!next(b, n)) return false; out.assign(b, n); return true; }
    bool i64(long long& out) { std::string t; if (!str(t)) return false; char* q; out = strtoll(t.c_str(), &q, 10); return q != t.c_str(); }
    bool i32(int& out) { long long v; if (!i64(v)) return false; out = (int)v; return true; }
    bool dbl(double& out) { std::string t; if (!str(t)) return false; char* q; out = strtod(t.c_str(), &q); return q != t.c_str(); }
};


// Parsed image of a GWAS summary file (rsid chr bp a1 a2 z), kept per process and shared by every window of a
// chromosome run that names the same file (path + size + mtime): the reference re-reads the text once per call
// (gauss.cpp:146-152).  Rows are in file order and carry the reference's parsing-state semantics: a field that
// fails to parse keeps the value of the previous line (the variables live outside the loop there too).
struct GwasRow { std::string rsid, a1, a2; int chr; long long bp; double z; };
struct GwasCache {
    std::vector<GwasRow> rows;
    std::vector<uint32_t> by_pos;      // row numbers ordered by (chr, bp), file order among equals: a window takes its range by binary search
    // positions the study lists more than once, or under equal alleles: the only ones where the reference's duplicate check
    // (gauss.cpp:386-392) can fire -- the gene drivers keep them in their SNP map whatever the annotation names (sorted)
    std::vector<std::pair<int, long long>> odd_positions;
};

// ------------------------------------------------------------------------------------------
// prepared window / gene set
// ------------------------------------------------------------------------------------------
struct gauss_prepared {
    int kind = 0;
    Args args;
    SnpMap snp_map;
    std::vector<Snp*> snp_vec;                 // after the AF filter, map order
    std::vector<Snp*> measured, unmeasured;    // matrix row order
    std::vector<int32_t> measured_rows, unmeasured_rows;
    std::vector<uint8_t> gm, gu;
    int64_t ld = 0;
    int N = 0;
    std::vector<int32_t> pop_off;
    std::vector<double> pop_wgt, z1;
    std::vector<int32_t> gene_off;
    std::vector<double> out_z, out_info, out_r, out_b11, out_b21;
    int n_head = 0, n_predm = 0;               // QCAT: measured SNPs left of / inside the prediction window
    // packed panel: rows stay in the mmap'd file and are named by index (zero-copy); gm / gu are only
    // materialised (as ASCII) for the kinds that need bytes on the host
    bool packed_rows = false;
    std::vector<int32_t> store_rows_m, store_rows_u, pop_src_off;
    int32_t num_eig = 0;
    int32_t status = 0;
    gauss_table snps;
    bool snps_built = false;
};

struct GeneResult {
    std::string geneid = ".", top_categ = ".", top_snp = ".";
    double chisq = -1.0, jepeg_pval = -1.0, top_categ_pval = -1.0, top_snp_pval = -1.0;
    int df = 0, num_snp = 0;
};

// minimal fork-join helper: fn(i) for i in [0, n) on up to nt threads
template <typename F>
inline void parallel_for(int n, int nt, F fn)
{
    if (n <= 0) return;
    nt = std::max(1, std::min(nt, n));
    if (nt == 1) { for (int i = 0; i < n; i++) fn(i); return; }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    auto body = [&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
    for (int t = 1; t < nt; t++) th.emplace_back(body);
    body();
    for (std::thread& x : th) x.join();
}

inline bool env_flag(const char* name, bool dflt)
{
    const char* e = getenv(name);
    return e ? atoi(e) != 0 : dflt;
}

inline double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- the chromosome driver's window (host_chrom.cpp: a merge of the sorted study and panel tables); also behind the one-window
// entry points of host_calls.cpp ----
struct ChromSetup {                     // what prepare() derives from a call's arguments alone: once per call, not once per window
    int kind = 0;
    bool mix = false, qcat = false;
    bool measured_only = false;         // computeLD: unmeasured SNPs are never read (computeLD.cpp:80-86) -- not entered at all
    Args a;                             // population table, flags, weights, cutoffs (start_bp / end_bp are the windows', not set here)
    std::vector<int> sel;               // the selected populations, panel order
    std::vector<int32_t> pop_off, pop_src_off;
    std::vector<double> pop_wgt;
    double two_subj = 0;                // 2 x the selected samples (gauss.cpp:589)
    std::shared_ptr<const GwasCache> gw;
};

struct LeanSnp {
    int64_t row;                        // panel row = fpos of the packed feeder
    long long bp;
    double z, info, af;
    int32_t type, qcat_m;
    double qcat_t, qcat_chisq;
};

struct LeanWindow {
    const ChromSetup* cs = nullptr;
    long long start_bp = 0, end_bp = 0;
    std::vector<LeanSnp> v;             // prepare()'s snp_vec: after the AF filter, map order
    std::vector<int32_t> measured, unmeasured;      // into v, matrix row order
    std::vector<int32_t> store_rows_m, store_rows_u;
    std::vector<double> z1, out_z, out_info, out_r;
    int n_head = 0, n_predm = 0;
    int32_t num_eig = 0, status = 0;
    std::unique_ptr<gauss_table> pre;   // the output table, built while the GPU works (lean_table_prebuild); the results are filled in after
    std::vector<int32_t> out_row;       // v -> row of the table, -1 outside the prediction window
    // the chromosome driver builds no table per window: the window's rows are a slice [tab_off, tab_off + n_out) of the call's ONE table
    size_t tab_off = 0;
    int n_out = -1;                     // rows of the prediction window (lean_table_count)
    // A window handed out by the window cache owns none of the lists above: `base` is the cached (immutable) window they are read
    // from -- lean_src() -- and this object holds the call's own state only (outputs, status, its slice of the table).  The chromosome
    // driver's functions (lean_window_desc, lean_table_count, lean_table_prebuild_into, lean_window_finish_into) read through lean_src.
    std::shared_ptr<const LeanWindow> base;
};
static inline const LeanWindow& lean_src(const LeanWindow& w) { return w.base ? *w.base : w; }
// The slice form of lean_table_prebuild / lean_window_finish: the window's rows written straight into the columns of `all` (which must
// hold the reference's column set, sized to cover the slice) -- everything the results do not change before the GPU is waited for,
// the unmeasured SNPs' z / info / pval (QCAT: the four test columns) after.  Slices of different windows may be written concurrently.
int lean_table_count(LeanWindow& w);
void lean_table_prebuild_into(LeanWindow& w, gauss_table& all, size_t off);
void lean_window_finish_into(LeanWindow& w, gauss_table& all);
// A pristine copy of a built window (the inputs; nothing a call writes) -- what the window cache keeps and hands out
std::unique_ptr<LeanWindow> lean_window_clone(const LeanWindow& w);

int chrom_setup(ChromSetup& cs, int kind, int chr, int64_t wing_size, const char* study_pop, const char* const* pop_names,
                const double* pop_wgts, int n_pop_wgt, const char* input_file, const std::string& packed_path, const char* desc_file,
                double af1_cutoff, const std::shared_ptr<PackedPanel>& pk, const std::shared_ptr<const GwasCache>& gw);
int lean_window_build(LeanWindow& w, const ChromSetup& cs, long long start_bp, long long end_bp);
int lean_window_desc(LeanWindow& w, gauss_window_desc* d);
void lean_table_prebuild(LeanWindow& w);
gauss_table* lean_window_finish(LeanWindow& w);

// ---- routines that cross translation units ----
int read_ref_desc(Args& a);
int init_pop_flag_vec(Args& a);
void init_pop_flag_wgt_vec(Args& a);
void set_pop_wgt_map(Args& a, const char* const* names, const double* w, int n);
std::shared_ptr<const GwasCache> load_gwas_cached(const std::string& path, std::string& err);
int ReadInputZ(SnpMap& m, const Args& a, bool All);
int merge_index_entry(SnpMap& m, const Args& a, bool All, const std::string& rsid, int chr, long long bp,
                             const std::string& a1, const std::string& a2, long long fpos);
int ReadReferenceIndex(SnpMap& m, const Args& a, bool All);
void load_line(BgzfReader& fp, Snp& s, const Args& a, std::vector<double>* af_out);
std::shared_ptr<PackedPanel> open_packed_shared(const std::string& path, std::string& err);
int resolve_packed_panel(const std::string& index_file, const std::string& data_file, const std::string& desc_file,
                                bool create, std::string& out, std::string& err, int64_t* packed_now = nullptr);
int auto_pack_mode();
void fill_matrix(std::vector<uint8_t>& G, const std::vector<Snp*>& rows, int64_t ld);
void unpack_rows(const gauss_prepared& p, const std::vector<Snp*>& rows, std::vector<uint8_t>& G);
void materialise_from_packed(gauss_prepared& p);
void build_snp_table(gauss_prepared& p);
int prepare(gauss_prepared& p);
double pnorm_upper(double x);
double pchisq_upper(double x, int df);
GeneResult jepeg_tail(const std::vector<Snp*>& gs, const double* CorG, const Args& a);
gauss_table* dist_output(gauss_prepared& p);
gauss_table* qcat_output(gauss_prepared& p);
gauss_table* prep_output(gauss_prepared& p);
int panel_make_resident(gauss_ctx* ctx, const std::string& path, void** dev, int64_t* uploaded, bool async = false);
bool panel_is_resident(gauss_ctx* ctx, const std::string& path, void** dev, bool wait = true);
