// libgauss_host.so -- host data layer + the five reference entry points (include/gauss_host.h).
//
// Restates, in plain C++ without Rcpp, the reference's feeder semantics:
//   Arguments defaults          src/gauss.cpp:18-35
//   ReadInputZ                  src/gauss.cpp:121-190
//   ReadReferenceIndex / ...All src/gauss.cpp:293-399 / 431-518
//   MakeSnpVec / MakeSnpVecMix  src/gauss.cpp:543-604 / 631-693
//   ReadGenotype                src/gauss.cpp:720-785
//   read_ref_desc               src/gauss.cpp:951-993
//   init_pop_flag_vec / _wgt_   src/gauss.cpp:1019-1066 / 1093-1117
//   ReadAnnotation              src/gauss.cpp:1275-1361
//   MakeGeneStartEndVec         src/gauss.cpp:1383-1439
//   SNP ordering (MapKey)       src/gauss.h:72-99
// and the drivers computeLD.cpp:26-166, dist.cpp:30-126, distmix.cpp:30-135, jepeg.cpp:28-153,
// jepegmix.cpp:26-161 with the numeric hot path delegated to libgauss_hip.so.
#include "../../../include/gauss_host.h"

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
static thread_local std::string g_err;

// Diagnostics on stderr: GAUSS_TRACE=chrom,prep (any subset, or "all"; libgauss_hip reads job, upload, stream from the same variable)
static bool host_trace(const char* what)
{
    const char* e = getenv("GAUSS_TRACE");
    return e && *e && (strstr(e, "all") != nullptr || strstr(e, what) != nullptr);
}

static int herr(const char* fmt, ...)
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
// spread over the worker threads' fresh malloc arenas (measured).  Blocks are 2 MB, 2 MB-aligned,
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
static BlockPools& block_pools() { static BlockPools* bp = new BlockPools(); return *bp; }

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

// read_ref_desc (gauss.cpp:951-993)
static int read_ref_desc(Args& a)
{
    std::ifstream in(a.reference_pop_desc_file.c_str());
    if (!in) return herr("ERROR: can't open reference population description file '%s'", a.reference_pop_desc_file.c_str());
    std::string line, pop, sup;
    int n = 0;
    std::getline(in, line);   // header
    while (std::getline(in, line)) {
        Tok t(line);
        if (!t.str(pop)) continue;
        t.i32(n); t.str(sup);
        a.ref_pop_vec.push_back(pop);
        a.ref_pop_size_vec.push_back(n);
        a.ref_sup_pop_vec.push_back(sup);
    }
    a.num_pops = (int)a.ref_pop_vec.size();
    return 0;
}

// init_pop_flag_vec (gauss.cpp:1019-1066): study_pop names a population or a super population
static int init_pop_flag_vec(Args& a)
{
    const int in_pop = (int)std::count(a.ref_pop_vec.begin(), a.ref_pop_vec.end(), a.study_pop);
    const int in_sup = (int)std::count(a.ref_sup_pop_vec.begin(), a.ref_sup_pop_vec.end(), a.study_pop);
    const std::vector<std::string>* pv = nullptr;
    if (in_pop != 0 && in_sup == 0) pv = &a.ref_pop_vec;
    if (in_pop == 0 && in_sup != 0) pv = &a.ref_sup_pop_vec;
    if (in_pop == 0 && in_sup == 0) return herr("ERROR: invalid population name '%s'", a.study_pop.c_str());
    if (!pv) return herr("ERROR: population name '%s' is both a population and a super population", a.study_pop.c_str());
    int cnt = 0;
    for (int i = 0; i < a.num_pops; i++) {
        if ((*pv)[i] == a.study_pop) { a.pop_flag_vec.push_back(1); cnt += a.ref_pop_size_vec[i]; }
        else a.pop_flag_vec.push_back(0);
    }
    a.num_samples = cnt;
    return 0;
}

// init_pop_flag_wgt_vec (gauss.cpp:1093-1117): weights re-ordered into panel order, unknown names ignored
static void init_pop_flag_wgt_vec(Args& a)
{
    for (int i = 0; i < a.num_pops; i++) {
        auto it = a.pop_wgt_map.find(a.ref_pop_vec[i]);
        if (it != a.pop_wgt_map.end()) { a.pop_flag_vec.push_back(1); a.pop_wgt_vec.push_back(it->second); }
        else a.pop_flag_vec.push_back(0);
    }
}

static void set_pop_wgt_map(Args& a, const char* const* names, const double* w, int n)
{
    for (int i = 0; i < n; i++) {       // distmix.cpp:48-54: names upper-cased
        std::string pop = names[i];
        std::transform(pop.begin(), pop.end(), pop.begin(), ::toupper);
        a.pop_wgt_map[pop] = w[i];
    }
}

// Parsed image of a GWAS summary file (rsid chr bp a1 a2 z), kept per process and shared by every window of a
// chromosome run that names the same file (path + size + mtime): the reference re-reads the text once per call
// (gauss.cpp:146-152).  Rows are in file order and carry the reference's parsing-state semantics: a field that
// fails to parse keeps the value of the previous line (the variables live outside the loop there too).
struct GwasRow { std::string rsid, a1, a2; int chr; long long bp; double z; };
struct GwasCache {
    std::vector<GwasRow> rows;
    std::vector<uint32_t> by_pos;      // row numbers ordered by (chr, bp), file order among equals: a window takes its range by binary search
};

static std::shared_ptr<const GwasCache> load_gwas_cached(const std::string& path, std::string& err)
{
    static std::mutex mu;
    static std::map<std::string, std::shared_ptr<const GwasCache>> cache;
    struct stat st;
    if (stat(path.c_str(), &st) != 0) { err = "ERROR: can't open input file '" + path + "'"; return nullptr; }
    char key[64];
    snprintf(key, sizeof(key), "|%lld|%lld.%ld", (long long)st.st_size, (long long)st.st_mtim.tv_sec, (long)st.st_mtim.tv_nsec);
    const std::string k = path + key;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(k);
    if (it != cache.end()) return it->second;
    std::ifstream in(path.c_str());
    if (!in) { err = "ERROR: can't open input file '" + path + "'"; return nullptr; }
    std::shared_ptr<GwasCache> c = std::make_shared<GwasCache>();
    std::string line, rsid, a1, a2;
    int chr = 0; long long bp = 0; double z = 0;
    std::getline(in, line);   // header
    while (std::getline(in, line)) {
        Tok t(line);
        if (t.str(rsid) && t.i32(chr) && t.i64(bp) && t.str(a1) && t.str(a2)) t.dbl(z);
        c->rows.push_back(GwasRow{rsid, a1, a2, chr, bp, z});
    }
    c->by_pos.resize(c->rows.size());
    for (size_t i = 0; i < c->rows.size(); i++) c->by_pos[i] = (uint32_t)i;
    std::stable_sort(c->by_pos.begin(), c->by_pos.end(), [&](uint32_t x, uint32_t y) {
        const GwasRow &a = c->rows[x], &b = c->rows[y];
        return a.chr < b.chr || (a.chr == b.chr && a.bp < b.bp);
    });
    if (cache.size() >= 8) cache.clear();          // a handful of studies per process at most
    cache[k] = c;
    return c;
}

// ReadInputZ (gauss.cpp:121-190)
static int ReadInputZ(SnpMap& m, const Args& a, bool All)
{
    std::string err;
    std::shared_ptr<const GwasCache> gw = load_gwas_cached(a.input_file, err);
    if (!gw) return herr("%s", err.c_str());
    // A window of one chromosome: its rows are a range of the (chr, bp)-ordered index (the reference scans the whole file for every
    // window, gauss.cpp:133-140).  Rows of one position keep their file order, so a key listed twice ends with its later row either way.
    size_t q0 = 0, q1 = gw->rows.size();
    const bool ranged = !All && a.chr > 0;
    if (ranged) {
        const long long lo = a.start_bp - a.wing_size, hi = a.end_bp + a.wing_size;
        auto before = [&](uint32_t x, long long bp) { const GwasRow& r = gw->rows[x]; return r.chr < a.chr || (r.chr == a.chr && r.bp < bp); };
        q0 = (size_t)(std::lower_bound(gw->by_pos.begin(), gw->by_pos.end(), lo, before) - gw->by_pos.begin());
        q1 = (size_t)(std::lower_bound(gw->by_pos.begin(), gw->by_pos.end(), hi + 1, before) - gw->by_pos.begin());
    }
    for (size_t q = q0; q < q1; q++) {
        const GwasRow& r = gw->rows[ranged ? gw->by_pos[q] : q];
        if (!All) {
            if ((a.chr > 0) && (a.chr != r.chr)) continue;
            if ((a.start_bp - a.wing_size) > r.bp || (a.end_bp + a.wing_size) < r.bp) continue;
        }
        SnpPtr s = m.make();
        s->rsid = r.rsid; s->chr = r.chr; s->bp = r.bp; s->a1 = r.a1; s->a2 = r.a2; s->z = r.z;
        s->info = 1.0;     // gauss.cpp:142
        s->type = 2;       // gauss.cpp:176
        // (a study file sorted by position appends: no descent through the tree; a later row of the same key still replaces the earlier one)
        MapKey key{r.chr, r.bp, r.a1, r.a2};
        auto it = (m.empty() || m.rbegin()->first < key) ? m.emplace_hint(m.end(), std::move(key), SnpPtr()) : m.try_emplace(std::move(key)).first;
        it->second = std::move(s);
    }
    return 0;
}

// Parsed image of a BGZF text index file (rsid chr bp a1 a2 af1ref fpos per line), kept per process and
// shared by every call that names the same file (path + size + mtime).  Entries are in file order and carry the
// reference's parsing state semantics: a field that fails to parse keeps the value of the previous line, as the
// reference's variables do (they are declared outside its loop, gauss.cpp:317-321).
struct IndexCache {
    struct Entry { int32_t chr; uint32_t rsid, a1, a2; long long bp, fpos; };
    std::vector<Entry> e;
    std::vector<char> pool;
    bool sorted = true;
    size_t lower_bound(int chr, long long bp) const
    {
        size_t lo = 0, hi = e.size();
        while (lo < hi) {
            const size_t mid = lo + (hi - lo) / 2;
            if (e[mid].chr < chr || (e[mid].chr == chr && e[mid].bp < bp)) lo = mid + 1; else hi = mid;
        }
        return lo;
    }
};

static std::shared_ptr<const IndexCache> load_index_cached(const std::string& path, std::string& err)
{
    static std::mutex mu;
    static std::map<std::string, std::shared_ptr<const IndexCache>> cache;      // a handful of panels per process
    struct stat st;
    if (stat(path.c_str(), &st) != 0) { err = "ERROR: can't open reference index file '" + path + "'"; return nullptr; }
    char key[64];
    snprintf(key, sizeof(key), "|%lld|%lld.%ld", (long long)st.st_size, (long long)st.st_mtim.tv_sec, (long)st.st_mtim.tv_nsec);
    const std::string k = path + key;
    std::lock_guard<std::mutex> lock(mu);             // concurrent windows of a farm: the first one parses
    auto it = cache.find(k);
    if (it != cache.end()) return it->second;
    BgzfReader fp;
    if (!fp.open(path)) { err = "ERROR: can't open reference index file '" + path + "'"; return nullptr; }
    std::shared_ptr<IndexCache> ic = std::make_shared<IndexCache>();
    std::string line, rsid, a1, a2;
    int chr = 0; double af1ref = 0; long long bp = 0, fpos = 0;
    auto add = [&](const std::string& v) { const uint32_t o = (uint32_t)ic->pool.size(); ic->pool.insert(ic->pool.end(), v.begin(), v.end()); ic->pool.push_back(0); return o; };
    for (;;) {
        const int last = fp.getline(line);
        if (last == -2) { err = "Error: can't read reference index file '" + path + "'"; return nullptr; }
        if (last == -1) break;
        Tok t(line);
        if (t.str(rsid) && t.i32(chr) && t.i64(bp) && t.str(a1) && t.str(a2) && t.dbl(af1ref)) t.i64(fpos);
        IndexCache::Entry en{chr, add(rsid), add(a1), add(a2), bp, fpos};
        if (!ic->e.empty() && (chr < ic->e.back().chr || (chr == ic->e.back().chr && bp < ic->e.back().bp))) ic->sorted = false;
        ic->e.push_back(en);
    }
    if (cache.size() >= 4) cache.clear();              // bound the memory of a long-lived process
    cache[k] = ic;
    return ic;
}

// One index entry merged into the SNP map: the body of the loops of ReadReferenceIndex (gauss.cpp:340-390)
// and ReadReferenceIndexAll (gauss.cpp:478-512).
static int merge_index_entry(SnpMap& m, const Args& a, bool All, const std::string& rsid, int chr, long long bp,
                             const std::string& a1, const std::string& a2, long long fpos)
{
    if (!All) {
        if ((a.chr > 0) && (a.chr != chr)) return 0;
        if ((a.start_bp - a.wing_size) > bp || (a.end_bp + a.wing_size) < bp) return 0;
    }
    // Most panel SNPs share their position with no GWAS SNP: one ordered lookup at (chr, bp) settles that neither
    // allele order is present and doubles as the insertion hint (same outcome as the two finds below, which only run
    // when something already sits at this position).
    auto pos = m.lower_bound(MapKey{chr, bp, std::string(), std::string()});
    if (pos == m.end() || pos->first.chr != chr || pos->first.bp != bp) {
        if (!All) {       // gauss.cpp:373-385; ReadReferenceIndexAll never adds unmeasured SNPs
            SnpPtr s = m.make();
            s->rsid = rsid; s->chr = chr; s->bp = bp; s->a1 = a1; s->a2 = a2; s->type = 0; s->fpos = fpos;
            m.emplace_hint(pos, MapKey{chr, bp, a1, a2}, std::move(s));
        }
        return 0;
    }
    auto it1 = m.find(MapKey{chr, bp, a1, a2});
    auto it2 = m.find(MapKey{chr, bp, a2, a1});
    if (it1 != m.end() && it2 == m.end()) {
        it1->second->rsid = rsid; it1->second->type = 1; it1->second->fpos = fpos;
    } else if (it1 == m.end() && it2 != m.end()) {
        // GWAS alleles are swapped relative to the panel: adopt the panel's order, flip z
        SnpPtr s = std::move(it2->second);
        m.erase(it2);
        s->rsid = rsid; s->a1 = a1; s->a2 = a2; s->z = s->z * (-1); s->type = 1; s->fpos = fpos;
        m[MapKey{chr, bp, a1, a2}] = std::move(s);
    } else if (it1 == m.end() && it2 == m.end()) {
        if (!All) {       // gauss.cpp:373-385; ReadReferenceIndexAll never adds unmeasured SNPs
            SnpPtr s = m.make();
            s->rsid = rsid; s->chr = chr; s->bp = bp; s->a1 = a1; s->a2 = a2; s->type = 0; s->fpos = fpos;
            m[MapKey{chr, bp, a1, a2}] = std::move(s);
        }
    } else {
        return herr("ERROR: input file contains duplicates");
    }
    return 0;
}

// ReadReferenceIndex (gauss.cpp:293-399) and ReadReferenceIndexAll (gauss.cpp:431-518)
static int ReadReferenceIndex(SnpMap& m, const Args& a, bool All)
{
    if (a.pk) {
        // packed panel: the SNP table is in memory; a sorted panel is entered by binary search instead of the
        // reference's genome-wide scan.  fpos is the row number.
        const PackedPanel& pk = *a.pk;
        int64_t i0 = 0, i1 = pk.n_snp();
        if (!All && a.chr > 0 && pk.header().sorted) {
            i0 = pk.lower_bound(a.chr, a.start_bp - a.wing_size);
            i1 = pk.lower_bound(a.chr, a.end_bp + a.wing_size + 1);
        }
        if (All && pk.header().sorted) {
            // ReadReferenceIndexAll never adds a SNP (gauss.cpp:478-512): only panel entries at a position the map already holds
            // can change anything, and an entry only touches map entries of its own position.  So instead of looking every
            // panel SNP up in the map (100 000 ordered lookups for a chromosome's panel against 13 000 study SNPs) walk the map's
            // positions and find each one's panel entries by binary search; entries of one position keep the panel's order.
            // (positions with several study SNPs or several panel entries -- multi-allelic sites -- go through
            // merge_index_entry; one study SNP against one panel entry, the rule, is settled on the spot: same alleles, swapped
            // alleles, or different alleles, exactly the three outcomes merge_index_entry has for it)
            std::vector<std::pair<int, long long>> slow;
            for (auto it = m.begin(); it != m.end();) {
                const int chr = it->first.chr;
                const long long bp = it->first.bp;
                auto nx = std::next(it);
                const bool single = (nx == m.end() || nx->first.chr != chr || nx->first.bp != bp);
                if (!single) {
                    slow.emplace_back(chr, bp);
                    while (nx != m.end() && nx->first.chr == chr && nx->first.bp == bp) ++nx;
                    it = nx;
                    continue;
                }
                const int64_t i = pk.lower_bound(chr, bp);
                const bool have = i < pk.n_snp() && pk.snp(i).chr == chr && pk.snp(i).bp == bp;
                if (have && ((i + 1 < pk.n_snp() && pk.snp(i + 1).chr == chr && pk.snp(i + 1).bp == bp) ||
                             strcmp(pk.str(pk.snp(i).a1), pk.str(pk.snp(i).a2)) == 0)) {       // (equal alleles: both lookups of merge_index_entry hit the same entry)
                    slow.emplace_back(chr, bp); it = nx; continue;
                }
                if (have) {
                    const PkSnp& s = pk.snp(i);
                    const char *pa1 = pk.str(s.a1), *pa2 = pk.str(s.a2);
                    if (it->first.a1 == pa1 && it->first.a2 == pa2) {
                        it->second->rsid = pk.str(s.rsid); it->second->type = 1; it->second->fpos = i;
                    } else if (it->first.a1 == pa2 && it->first.a2 == pa1) {
                        // GWAS alleles are swapped relative to the panel: adopt the panel's order, flip z (the new key sorts inside this position: `nx` stays the next position)
                        SnpPtr sp = std::move(it->second);
                        m.erase(it);
                        sp->rsid = pk.str(s.rsid); sp->a1 = pa1; sp->a2 = pa2; sp->z = sp->z * (-1); sp->type = 1; sp->fpos = i;
                        m[MapKey{chr, bp, pa1, pa2}] = std::move(sp);
                    }
                }
                it = nx;
            }
            for (const auto& cb : slow)
                for (int64_t i = pk.lower_bound(cb.first, cb.second); i < pk.n_snp(); i++) {
                    const PkSnp& s = pk.snp(i);
                    if (s.chr != cb.first || s.bp != cb.second) break;
                    if (merge_index_entry(m, a, All, pk.str(s.rsid), s.chr, s.bp, pk.str(s.a1), pk.str(s.a2), i)) return -1;
                }
            return 0;
        }
        if (!All && a.chr > 0 && pk.header().sorted) {
            // One window of a sorted panel: the panel's SNPs and the map ascend together, so the map position of each panel SNP is
            // found by walking an iterator forward instead of descending the tree for every one of the ~3 000 (`at` = first map entry
            // at or after the SNP's position).  A position the map holds nothing at -- the rule: an unmeasured SNP -- is entered (or,
            // in a wing of a dist / distmix window, left out) on the spot; a position that holds something, or one the previous panel
            // SNP shared (multi-allelic sites: its entry was put in FRONT of `at`), goes through merge_index_entry and `at` is found anew.
            auto key_before = [](const MapKey& k, int chr, long long bp) { return k.chr < chr || (k.chr == chr && k.bp < bp); };
            auto at = m.end();
            bool have_at = false;
            int pchr = -1;
            long long pbp = -1;
            for (int64_t i = i0; i < i1; i++) {
                const PkSnp& s = pk.snp(i);
                if ((a.start_bp - a.wing_size) > s.bp || (a.end_bp + a.wing_size) < s.bp || s.chr != a.chr) continue;      // (merge_index_entry's own filter)
                const bool same_site = (s.chr == pchr && s.bp == pbp);
                pchr = s.chr; pbp = s.bp;
                if (!have_at || same_site) { at = m.lower_bound(MapKey{s.chr, s.bp, std::string(), std::string()}); have_at = true; }
                else while (at != m.end() && key_before(at->first, s.chr, s.bp)) ++at;
                if (at == m.end() || at->first.chr != s.chr || at->first.bp != s.bp) {
                    // (a wing SNP is only left out when it is the panel's one entry at its position: a panel that lists a site twice
                    // makes the second entry find the first -- gauss.cpp:356-361 turns that into a measured SNP -- so both are entered)
                    const bool alone = !(i + 1 < i1 && pk.snp(i + 1).chr == s.chr && pk.snp(i + 1).bp == s.bp);
                    if (a.drop_wing_unmeasured && alone && (s.bp < a.start_bp || s.bp > a.end_bp)) continue;
                    SnpPtr sp = m.make();                      // gauss.cpp:373-385
                    sp->rsid = pk.str(s.rsid); sp->chr = s.chr; sp->bp = s.bp; sp->a1 = pk.str(s.a1); sp->a2 = pk.str(s.a2); sp->type = 0; sp->fpos = i;
                    m.emplace_hint(at, MapKey{s.chr, s.bp, sp->a1, sp->a2}, std::move(sp));
                    continue;
                }
                if (merge_index_entry(m, a, All, pk.str(s.rsid), s.chr, s.bp, pk.str(s.a1), pk.str(s.a2), i)) return -1;
                have_at = false;                               // (an entry of this position may have been erased and entered again)
            }
            return 0;
        }
        // (an unsorted panel, or a call over every chromosome: every entry goes through the map; wing SNPs are not left out here --
        // entries of one position need not be neighbours, so "the panel's one entry at its position" cannot be told on the spot)
        for (int64_t i = i0; i < i1; i++) {
            const PkSnp& s = pk.snp(i);
            if (merge_index_entry(m, a, All, pk.str(s.rsid), s.chr, s.bp, pk.str(s.a1), pk.str(s.a2), i)) return -1;
        }
        return 0;
    }
    // text index: parsed once per file and process (read-once index) -- the
    // reference inflates and parses the whole genome-wide index on every call (gauss.cpp:322-392)
    std::string err;
    std::shared_ptr<const IndexCache> ic = load_index_cached(a.reference_index_file, err);
    if (!ic) return herr("%s", err.c_str());
    size_t i0 = 0, i1 = ic->e.size();
    if (!All && a.chr > 0 && ic->sorted) {
        i0 = ic->lower_bound(a.chr, a.start_bp - a.wing_size);
        i1 = ic->lower_bound(a.chr, a.end_bp + a.wing_size + 1);
    }
    std::string rsid, a1, a2;
    for (size_t i = i0; i < i1; i++) {
        const IndexCache::Entry& e = ic->e[i];
        rsid = ic->pool.data() + e.rsid; a1 = ic->pool.data() + e.a1; a2 = ic->pool.data() + e.a2;
        if (merge_index_entry(m, a, All, rsid, e.chr, e.bp, a1, a2, e.fpos)) return -1;
    }
    return 0;
}

// Read the panel data line of a SNP once and split it into the P genotype strings and P
// allele frequencies (gauss.cpp:755-763 and 660-674 parse the same line twice).
static void load_line(BgzfReader& fp, Snp& s, const Args& a, std::vector<double>* af_out)
{
    if (!s.have_line) {
        fp.seek(s.fpos);
        fp.getline(s.line);     // a seek past EOF (fpos = -1) yields an empty line, like the reference
        s.have_line = true;
    }
    s.geno.clear();
    Tok t(s.line);
    for (int k = 0; k < a.num_pops; k++) {
        const char* b = nullptr; int n = 0;
        if (!t.next(b, n)) { b = s.line.data() + s.line.size(); n = 0; }
        if (a.pop_flag_vec[k]) s.geno.push_back(std::make_pair(b, n));
    }
    if (af_out) {
        af_out->clear();
        for (int k = 0; k < a.num_pops; k++) {
            double af = 0.0;            // a failed extraction leaves 0 (C++11 num_get)
            t.dbl(af);
            if (a.pop_flag_vec[k]) af_out->push_back(af);
        }
    }
}

// Panel lines are independent: inflate + split them on several host threads, each with its own reader
// (the reference reads them one by one through a single BGZF handle, gauss.cpp:546-566).
static std::atomic<int> g_host_threads{4};

static int preload_lines(SnpMap& m, const Args& a, bool want_af, std::vector<std::vector<double>>* afs)
{
    std::vector<Snp*> v;
    v.reserve(m.size());
    for (auto& kv : m) v.push_back(kv.second.get());
    if (afs) afs->assign(v.size(), std::vector<double>());
    const int nt = std::max(1, std::min<int>(g_host_threads.load(), (int)(v.size() / 64) + 1));
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&]() {
        BgzfReader fp;
        if (!fp.open(a.reference_data_file)) { failed = 1; return; }
        for (;;) {
            const size_t i0 = next.fetch_add(32);
            if (i0 >= v.size()) break;
            for (size_t i = i0; i < std::min(v.size(), i0 + 32); i++)
                load_line(fp, *v[i], a, (want_af && afs) ? &(*afs)[i] : nullptr);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (failed) return herr("ERROR: can't open reference data file '%s'", a.reference_data_file.c_str());
    return 0;
}

// MakeSnpVec / MakeSnpVecMix on a packed panel: the per-population allele counts and allele frequencies
// were tabulated when the panel was packed, so no genotype line is touched for the AF filter.
static int MakeSnpVecPacked(std::vector<Snp*>& v, SnpMap& m, const Args& a, bool mix)
{
    const PackedPanel& pk = *a.pk;
    for (auto& kv : m) {
        Snp& s = *kv.second;
        if (s.fpos < 0 || s.fpos >= pk.n_snp()) continue;      // GWAS-only SNP: the text path reads an empty line, AF = NaN / 0
        if (!mix) {
            double allele_counter = 0, num_subj = 0;           // gauss.cpp:574-591 (integer-valued sums)
            const int32_t* c = pk.cnt(s.fpos);
            for (int k = 0; k < a.num_pops; k++)
                if (a.pop_flag_vec[k]) { allele_counter += (double)c[k]; num_subj += a.ref_pop_size_vec[k]; }
            double af1ref = allele_counter / (2 * num_subj);
            af1ref = std::ceil(af1ref * 100000.0) / 100000.0;
            s.af1ref = af1ref;
            if ((af1ref > a.af1_cutoff) && (af1ref < (1 - a.af1_cutoff))) v.push_back(&s);
        } else {
            double af1_mix = 0;                                // gauss.cpp:676-682
            const double* f = pk.af(s.fpos);
            int j = 0;
            for (int k = 0; k < a.num_pops; k++)
                if (a.pop_flag_vec[k]) af1_mix += f[k] * a.pop_wgt_vec[j++];
            if ((af1_mix > a.af1_cutoff) && (af1_mix < (1 - a.af1_cutoff))) { s.af1mix = af1_mix; v.push_back(&s); }
        }
    }
    return 0;
}

// MakeSnpVec (gauss.cpp:543-604)
static int MakeSnpVec(std::vector<Snp*>& v, SnpMap& m, const Args& a)
{
    if (a.pk) return MakeSnpVecPacked(v, m, a, false);
    if (preload_lines(m, a, false, nullptr)) return -1;
    for (auto& kv : m) {
        Snp& s = *kv.second;
        double allele_counter = 0, num_subj = 0;
        for (auto& g : s.geno) {
            num_subj += g.second;
            for (int i = 0; i < g.second; i++) allele_counter += (double)(g.first[i] - '0');
        }
        double af1ref = allele_counter / (2 * num_subj);
        af1ref = std::ceil(af1ref * 100000.0) / 100000.0;      // gauss.cpp:591
        s.af1ref = af1ref;
        if ((af1ref > a.af1_cutoff) && (af1ref < (1 - a.af1_cutoff))) v.push_back(&s);
    }
    return 0;
}

// MakeSnpVecMix (gauss.cpp:631-693)
static int MakeSnpVecMix(std::vector<Snp*>& v, SnpMap& m, const Args& a)
{
    if (a.pk) return MakeSnpVecPacked(v, m, a, true);
    std::vector<std::vector<double>> afs;
    if (preload_lines(m, a, true, &afs)) return -1;
    size_t idx = 0;
    for (auto& kv : m) {
        Snp& s = *kv.second;
        const std::vector<double>& af1_vec = afs[idx++];
        double af1_mix = 0;
        for (size_t k = 0; k < af1_vec.size(); k++) af1_mix += af1_vec[k] * a.pop_wgt_vec[k];
        if ((af1_mix > a.af1_cutoff) && (af1_mix < (1 - a.af1_cutoff))) {
            s.af1mix = af1_mix;
            v.push_back(&s);
        }
    }
    return 0;
}

// ReadAnnotation (gauss.cpp:1275-1361)
// Parsed image of an annotation file, kept per process like the study file's (path + size + mtime): a gene-level call over
// the same annotation parses it once.  Rows carry what the reference's loop variables hold after each line (gauss.cpp:1308-1330:
// a field that fails to parse keeps the previous line's value, an unknown category name the previous number).
struct AnnotRow { int chr, categ_num; long long bp; double wgt; std::string a1, a2, geneid; };
struct AnnotCache { std::vector<AnnotRow> rows; };

static std::shared_ptr<const AnnotCache> load_annotation_cached(const std::string& path, std::string& err)
{
    static std::mutex mu;
    static std::map<std::string, std::shared_ptr<const AnnotCache>> cache;
    struct stat st;
    if (stat(path.c_str(), &st) != 0) { err = "ERROR: can't open snp annotation data file '" + path + "'"; return nullptr; }
    char key[64];
    snprintf(key, sizeof(key), "|%lld|%lld.%ld", (long long)st.st_size, (long long)st.st_mtim.tv_sec, (long)st.st_mtim.tv_nsec);
    const std::string k = path + key;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(k);
    if (it != cache.end()) return it->second;
    std::ifstream in(path.c_str());
    if (!in) { err = "ERROR: can't open snp annotation data file '" + path + "'"; return nullptr; }
    std::shared_ptr<AnnotCache> c = std::make_shared<AnnotCache>();
    std::string line, rsid, a1, a2, geneid, categ;
    int chr = 0, categ_num = 0; long long bp = 0; double wgt = 0;
    std::getline(in, line);
    while (std::getline(in, line)) {
        Tok t(line);
        if (t.str(rsid) && t.i32(chr) && t.i64(bp) && t.str(a1) && t.str(a2) && t.str(geneid) && t.str(categ)) t.dbl(wgt);
        if (categ == "PROTEIN") categ_num = 0;
        else if (categ == "TFBS") categ_num = 1;
        else if (categ == "WTH_HAIR") categ_num = 2;
        else if (categ == "WTH_TARGET") categ_num = 3;
        else if (categ == "CIS_EQTL") categ_num = 4;
        else if (categ == "TRANS_EQTL") categ_num = 5;      // unknown names keep the previous number (gauss.cpp:1319-1330)
        c->rows.push_back(AnnotRow{chr, categ_num, bp, wgt, a1, a2, geneid});
    }
    if (cache.size() >= 4) cache.clear();
    cache[k] = c;
    return c;
}

static int ReadAnnotation(SnpMap& m, const Args& a)
{
    std::string err;
    std::shared_ptr<const AnnotCache> an = load_annotation_cached(a.annotation_file, err);
    if (!an) return herr("%s", err.c_str());
    for (const AnnotRow& r : an->rows) {
        const int chr = r.chr, categ_num = r.categ_num;
        const long long bp = r.bp;
        const double wgt = r.wgt;
        const std::string &a1 = r.a1, &a2 = r.a2, &geneid = r.geneid;
        // nothing of the study at this position (most of a genome-wide annotation): neither allele order can be there
        auto pos = m.lower_bound(MapKey{chr, bp, std::string(), std::string()});
        if (pos == m.end() || pos->first.chr != chr || pos->first.bp != bp) continue;
        auto it1 = m.find(MapKey{chr, bp, a1, a2});
        auto it2 = m.find(MapKey{chr, bp, a2, a1});
        if (it1 != m.end() && it2 == m.end()) {
            it1->second->geneid = geneid;
            it1->second->categ[categ_num] = wgt;
        } else if (it1 == m.end() && it2 != m.end()) {
            SnpPtr s = std::move(it2->second);
            m.erase(it2);
            s->a1 = a1; s->a2 = a2;
            s->af1ref = 1 - s->af1ref;
            s->z = s->z * (-1);
            s->geneid = geneid;
            s->categ[categ_num] = wgt;
            m[MapKey{chr, bp, a1, a2}] = std::move(s);
        }
    }
    return 0;
}

// One mapping per packed panel file and process: every window of a chromosome shares it (and so the farm
// sees one store to make resident).  Keyed by path + size + mtime; dropped when the last window closes.
static std::shared_ptr<PackedPanel> open_packed_shared(const std::string& path, std::string& err)
{
    static std::mutex mu;
    static std::map<std::string, std::weak_ptr<PackedPanel>> cache;
    struct stat st;
    if (stat(path.c_str(), &st) != 0) { err = "ERROR: can't open reference data file '" + path + "'"; return nullptr; }
    char key[64];
    snprintf(key, sizeof(key), "|%lld|%lld.%ld", (long long)st.st_size, (long long)st.st_mtim.tv_sec, (long)st.st_mtim.tv_nsec);
    const std::string k = path + key;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(k);
    if (it != cache.end())
        if (std::shared_ptr<PackedPanel> sp = it->second.lock()) return sp;
    std::shared_ptr<PackedPanel> sp = std::make_shared<PackedPanel>();
    if (!sp->open(path, err)) return nullptr;
    cache[k] = sp;
    return sp;
}

// ------------------------------------------------------------------------------------------
// Packed-panel cache ("auto-pack on first use").  The reference's panel is three files (BGZF index, BGZF data,
// population description: gauss.cpp:293-399, 720-785); the packed panel made from them lives in a cache directory
// under a name that carries the identity (path, size, mtime) of all three, so a changed panel is packed again and a
// stale file is never picked up.  Directory: $GAUSS_PANEL_CACHE, else ".gauss_panel_cache" beside the data file, else
// (read-only panel directory) /tmp/gauss_panel_cache_<uid>.  Several processes may ask at once (one rank per GPU):
// the first one packs under an flock, the others wait for it; the file appears by rename, never half written.
// ------------------------------------------------------------------------------------------
static std::string file_identity(const std::string& path)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return path + "|missing";
    char buf[96];
    snprintf(buf, sizeof(buf), "|%lld|%lld.%09ld", (long long)st.st_size, (long long)st.st_mtim.tv_sec, (long)st.st_mtim.tv_nsec);
    char real[4096];
    const char* rp = realpath(path.c_str(), real);
    return std::string(rp ? rp : path.c_str()) + buf;
}

// A cache directory this process may create, use and TRUST: made with mode 0700 when missing; an existing one must be a
// real directory (no symlink), owned by this user and writable by nobody else -- a packed panel found in a directory
// that someone else can write to (the predictable /tmp fallback, pre-created by another local user) is hostile data.
// `trusted_shared`: a directory the user named (GAUSS_PANEL_CACHE) or one beside the panel files may be group-shared on
// purpose (a lab's panel directory); only the ownership-free checks apply there.
static bool dir_usable(const std::string& d, bool create, bool private_only)
{
    if (create && mkdir(d.c_str(), private_only ? 0700 : 0777) != 0 && errno != EEXIST) return false;
    struct stat st;
    if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
    if (private_only && (st.st_uid != getuid() || (st.st_mode & (S_IWGRP | S_IWOTH)))) return false;
    return access(d.c_str(), (create ? W_OK : R_OK) | X_OK) == 0;
}

// 0: `out` names a packed panel (the data file itself if it already is one).  1: no cached panel and create == false.
// -1: error (message in err).
static int resolve_packed_panel(const std::string& index_file, const std::string& data_file, const std::string& desc_file,
                                bool create, std::string& out, std::string& err, int64_t* packed_now = nullptr)
{
    if (packed_now) *packed_now = 0;
    if (PackedPanel::is_packed(data_file)) { out = data_file; return 0; }
    const std::string ident = file_identity(index_file) + "\n" + file_identity(data_file) + "\n" + file_identity(desc_file);
    uint64_t h1 = 1469598103934665603ull, h2 = 0x9E3779B97F4A7C15ull;           // two FNV-1a style lanes: a 128-bit name
    for (unsigned char c : ident) { h1 = (h1 ^ c) * 1099511628211ull; h2 = (h2 ^ (c + 0x5Bu)) * 0x100000001B3ull; h2 ^= h2 >> 29; }
    char hex[40];
    snprintf(hex, sizeof(hex), "%016llx%016llx", (unsigned long long)h1, (unsigned long long)h2);
    std::string base = data_file;
    const size_t slash = base.find_last_of('/');
    const std::string dir_of_data = slash == std::string::npos ? "." : base.substr(0, slash);
    if (slash != std::string::npos) base = base.substr(slash + 1);
    std::vector<std::pair<std::string, bool>> dirs;                               // (directory, must be private to this user)
    if (const char* e = getenv("GAUSS_PANEL_CACHE")) dirs.emplace_back(e, false);
    else { dirs.emplace_back(dir_of_data + "/.gauss_panel_cache", false); dirs.emplace_back("/tmp/gauss_panel_cache_" + std::to_string((long)getuid()), true); }
    // an existing entry anywhere on the list wins (in a directory that passes the trust check)
    for (const auto& dp : dirs) {
        if (!dir_usable(dp.first, false, dp.second)) continue;
        const std::string p = dp.first + "/" + base + "." + hex + ".gpk";
        if (PackedPanel::is_packed(p)) { out = p; return 0; }
    }
    if (!create) return 1;
    for (const auto& dp : dirs) {
        const std::string& d = dp.first;
        if (!dir_usable(d, true, dp.second)) continue;
        const std::string p = d + "/" + base + "." + hex + ".gpk";
        const std::string lockp = p + ".lock";
        // The lock file is removed by its holder WHILE it holds the lock; whoever gets the lock next checks that the file it
        // locked is still the one the name points at (same inode) and starts over otherwise -- so a waiter on the old inode
        // and a newcomer that created a new file can never both "hold the lock".
        int fd = -1;
        for (int attempt = 0; attempt < 100 && fd < 0; attempt++) {
            const int f = open(lockp.c_str(), O_CREAT | O_RDWR | O_NOFOLLOW, 0600);
            if (f < 0) break;
            if (flock(f, LOCK_EX) != 0) { close(f); break; }
            struct stat a, b;
            if (fstat(f, &a) == 0 && stat(lockp.c_str(), &b) == 0 && a.st_ino == b.st_ino && a.st_dev == b.st_dev) fd = f;
            else close(f);                                             // unlinked under us: the name is a new file now
        }
        if (fd < 0) continue;
        int rc = 0;
        if (!PackedPanel::is_packed(p)) {                                 // nobody packed it while we waited for the lock
            const std::string tmp = p + ".tmp." + std::to_string((long)getpid());
            const int64_t n = gauss_host::pack_panel(index_file, data_file, desc_file, tmp, err);
            if (n < 0) { unlink(tmp.c_str()); rc = -1; }
            else if (rename(tmp.c_str(), p.c_str()) != 0) { err = "ERROR: can't move the packed panel into the cache: " + p; unlink(tmp.c_str()); rc = -1; }
            else if (packed_now) *packed_now = n;
        }
        unlink(lockp.c_str());                                        // still locked: see above
        flock(fd, LOCK_UN);
        close(fd);
        if (rc) return rc;
        out = p;
        return 0;
    }
    err = "ERROR: no writable directory for the packed-panel cache (set GAUSS_PANEL_CACHE)";
    return -1;
}

// The policy of the entry points.  GAUSS_AUTO_PACK=0: never look at the cache.  GAUSS_AUTO_PACK=1: pack on first use,
// everywhere.  Unset: a one-window entry point uses a cached panel when one exists but does not make one (packing a
// genome-wide panel takes minutes; one window from the text files takes a second), the chromosome driver -- which needs
// the packed form -- packs on first use.
static int auto_pack_mode()
{
    const char* e = getenv("GAUSS_AUTO_PACK");
    return e ? (atoi(e) != 0 ? 1 : 0) : -1;
}

static int panel_make_resident(gauss_ctx* ctx, const std::string& path, void** dev, int64_t* uploaded, bool async = false);
static bool panel_is_resident(gauss_ctx* ctx, const std::string& path, void** dev, bool wait = true);

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

static void fill_matrix(std::vector<uint8_t>& G, const std::vector<Snp*>& rows, int64_t ld)
{
    G.assign((size_t)std::max<size_t>(rows.size(), 1) * ld, (uint8_t)'0');
    for (size_t r = 0; r < rows.size(); r++) {
        uint8_t* dst = G.data() + r * ld;
        uint8_t* const row0 = dst;
        for (auto& g : rows[r]->geno) { memcpy(dst, g.first, (size_t)g.second); dst += g.second; }
        if (rows[r]->flip_geno)                                  // gauss.cpp:1165-1176: only '0'..'2' are flipped
            for (uint8_t* c = row0; c < dst; c++)
                if (*c >= '0' && *c <= '2') *c = (uint8_t)('0' + (2 - (*c - '0')));
    }
}

// ASCII genotype matrices out of the packed rows (selected populations, panel order), with the
// minor-allele flip of UpdateSnpToMinorAllele applied where flagged.
static void unpack_rows(const gauss_prepared& p, const std::vector<Snp*>& rows, std::vector<uint8_t>& G)
{
    const PackedPanel& pk = *p.args.pk;
    G.assign((size_t)std::max<size_t>(rows.size(), 1) * p.ld, (uint8_t)'0');
    for (size_t r = 0; r < rows.size(); r++) {
        uint8_t* dst = G.data() + r * p.ld;
        const uint8_t* src = pk.row(rows[r]->fpos);
        const bool flip = rows[r]->flip_geno;
        for (int k = 0; k < p.args.num_pops; k++) {
            if (!p.args.pop_flag_vec[k]) continue;
            const uint8_t* b = src + pk.pop(k).byte_off;
            const int m = (int)pk.pop(k).size;
            for (int i = 0; i < m; i++) {
                int c = (b[i >> 2] >> (2 * (i & 3))) & 3;
                if (flip && c <= 2) c = 2 - c;
                *dst++ = (uint8_t)('0' + c);
            }
        }
    }
}

static void materialise_from_packed(gauss_prepared& p)
{
    unpack_rows(p, p.measured, p.gm);
    unpack_rows(p, p.unmeasured, p.gu);
}

static void build_snp_table(gauss_prepared& p)
{
    gauss_table& t = p.snps;
    t.cols.clear();
    const bool mix = (p.kind == GAUSS_KIND_COMPUTELD || p.kind == GAUSS_KIND_DISTMIX || p.kind == GAUSS_KIND_JEPEGMIX ||
                      p.kind == GAUSS_KIND_QCATMIX || p.kind == GAUSS_KIND_PREP_RECESSIVE);
    Column& rsid = t.add("rsid", GAUSS_COL_STR);
    for (Snp* s : p.snp_vec) rsid.s.push_back(s->rsid);
    Column& chr = t.add("chr", GAUSS_COL_INT);
    for (Snp* s : p.snp_vec) chr.i.push_back(s->chr);
    Column& bp = t.add("bp", GAUSS_COL_INT);
    for (Snp* s : p.snp_vec) bp.i.push_back((int)s->bp);
    Column& a1 = t.add("a1", GAUSS_COL_STR);
    for (Snp* s : p.snp_vec) a1.s.push_back(s->a1);
    Column& a2 = t.add("a2", GAUSS_COL_STR);
    for (Snp* s : p.snp_vec) a2.s.push_back(s->a2);
    Column& af = t.add(mix ? "af1mix" : "af1ref", GAUSS_COL_DBL);
    for (Snp* s : p.snp_vec) af.d.push_back(mix ? s->af1mix : s->af1ref);
    Column& z = t.add("z", GAUSS_COL_DBL);
    for (Snp* s : p.snp_vec) z.d.push_back(s->z);
    Column& info = t.add("info", GAUSS_COL_DBL);
    for (Snp* s : p.snp_vec) info.d.push_back(s->info);
    Column& type = t.add("type", GAUSS_COL_INT);
    for (Snp* s : p.snp_vec) type.i.push_back(s->type);
    Column& fpos = t.add("fpos", GAUSS_COL_DBL);
    for (Snp* s : p.snp_vec) fpos.d.push_back((double)s->fpos);
    Column& gid = t.add("geneid", GAUSS_COL_STR);
    for (Snp* s : p.snp_vec) gid.s.push_back(s->geneid);
    p.snps_built = true;
}

static int prepare(gauss_prepared& p)
{
    Args& a = p.args;
    const int kind = p.kind;
    const bool mix = (kind == GAUSS_KIND_COMPUTELD || kind == GAUSS_KIND_DISTMIX || kind == GAUSS_KIND_JEPEGMIX ||
                      kind == GAUSS_KIND_QCATMIX || kind == GAUSS_KIND_PREP_RECESSIVE);
    const bool gene = (kind == GAUSS_KIND_JEPEG || kind == GAUSS_KIND_JEPEGMIX);
    const bool qcat = (kind == GAUSS_KIND_QCAT || kind == GAUSS_KIND_QCATMIX);
    const bool prep = (kind == GAUSS_KIND_PREP_QCAT || kind == GAUSS_KIND_PREP_RECESSIVE);
    static const bool trace = host_trace("prep");
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tt[8] = {0};
    tt[0] = tnow();
    if (read_ref_desc(a)) return -1;
    if (a.pk) {
        if (a.pk->n_pop() != a.num_pops) return herr("packed panel has %d populations, the description file %d", a.pk->n_pop(), a.num_pops);
        for (int k = 0; k < a.num_pops; k++)
            if (a.ref_pop_vec[k] != a.pk->pop(k).name || a.ref_pop_size_vec[k] != (int)a.pk->pop(k).size)
                return herr("packed panel population %d is %s (%u samples), the description file says %s (%d)", k,
                            a.pk->pop(k).name, a.pk->pop(k).size, a.ref_pop_vec[k].c_str(), a.ref_pop_size_vec[k]);
    }
    if (mix) init_pop_flag_wgt_vec(a);
    else if (init_pop_flag_vec(a)) return -1;
    tt[1] = tnow();
    if (ReadInputZ(p.snp_map, a, gene)) return -1;
    tt[2] = tnow();
    if (ReadReferenceIndex(p.snp_map, a, gene)) return -1;
    tt[3] = tnow();
    if (gene && ReadAnnotation(p.snp_map, a)) return -1;
    const double t_annot = tnow();
    if (mix) { if (MakeSnpVecMix(p.snp_vec, p.snp_map, a)) return -1; }
    else if (MakeSnpVec(p.snp_vec, p.snp_map, a)) return -1;
    tt[4] = tnow();

    // populations selected, in panel order; N = sum of their sizes as found in the panel lines
    p.pop_off.assign(1, 0);
    for (int k = 0; k < a.num_pops; k++)
        if (a.pop_flag_vec[k]) p.pop_off.push_back(p.pop_off.back() + a.ref_pop_size_vec[k]);
    p.N = p.pop_off.back();
    p.ld = ((int64_t)p.N + 15) / 16 * 16;
    if (mix) p.pop_wgt = a.pop_wgt_vec;
    else p.pop_wgt.assign(p.pop_off.size() - 1, 1.0);

    if (kind == GAUSS_KIND_PREP_RECESSIVE) {
        for (Snp* s : p.snp_vec)                                // UpdateSnpToMinorAllele, gauss.cpp:1137-1184
            if (s->af1mix > 0.5) {
                s->af1mix = 1 - s->af1mix;
                s->z = -s->z;
                std::swap(s->a1, s->a2);
                s->flip_geno = true;
            }
    }
    if (prep) {
        for (size_t r = 0; r < p.snp_vec.size(); r++) {        // prep_qcat.cpp:69-78, prep_qcatmix.cpp:104-117
            Snp* s = p.snp_vec[r];
            // "unmeasured" here = every panel SNP of the prediction window, measured ones included
            if (s->type != 2 && (s->bp >= a.start_bp && s->bp <= a.end_bp)) { p.unmeasured.push_back(s); p.unmeasured_rows.push_back((int32_t)r); }
            if (s->type == 1) { p.measured.push_back(s); p.measured_rows.push_back((int32_t)r); }
        }
    } else if (kind == GAUSS_KIND_DIST || kind == GAUSS_KIND_DISTMIX || qcat) {
        for (size_t r = 0; r < p.snp_vec.size(); r++) {        // dist.cpp:132-140, qcat.cpp:140-152
            Snp* s = p.snp_vec[r];
            if (s->type == 0 && (s->bp >= a.start_bp && s->bp <= a.end_bp)) { p.unmeasured.push_back(s); p.unmeasured_rows.push_back((int32_t)r); }
            else if (s->type == 1) {
                p.measured.push_back(s); p.measured_rows.push_back((int32_t)r);
                if (s->bp < a.start_bp) p.n_head++;
                else if (s->bp <= a.end_bp) p.n_predm++;
            }
        }
    } else if (kind == GAUSS_KIND_COMPUTELD) {
        for (size_t r = 0; r < p.snp_vec.size(); r++)          // computeLD.cpp:80-86
            if (p.snp_vec[r]->type == 1) { p.measured.push_back(p.snp_vec[r]); p.measured_rows.push_back((int32_t)r); }
    } else {
        // jepeg.cpp:73-87: measured SNPs with a gene id, sorted by gene id with std::sort
        std::vector<std::pair<Snp*, int32_t>> gv;
        for (size_t r = 0; r < p.snp_vec.size(); r++)
            if (p.snp_vec[r]->geneid != "." && p.snp_vec[r]->type == 1) gv.push_back(std::make_pair(p.snp_vec[r], (int32_t)r));
        std::sort(gv.begin(), gv.end(), [](const std::pair<Snp*, int32_t>& x, const std::pair<Snp*, int32_t>& y) {
            return x.first->geneid < y.first->geneid;          // LessThanGeneid, snp.h:131-135
        });
        for (auto& g : gv) { p.measured.push_back(g.first); p.measured_rows.push_back(g.second); }
        // MakeGeneStartEndVec (gauss.cpp:1383-1439): runs of equal gene id
        p.gene_off.clear();
        for (size_t i = 0; i < p.measured.size(); i++)
            if (i == 0 || p.measured[i]->geneid != p.measured[i - 1]->geneid) p.gene_off.push_back((int32_t)i);
        p.gene_off.push_back((int32_t)p.measured.size());
    }
    if (a.pk) {
        const PackedPanel& pk = *a.pk;
        for (int k = 0; k < a.num_pops; k++)
            if (a.pop_flag_vec[k]) p.pop_src_off.push_back((int32_t)pk.pop(k).byte_off);
        for (Snp* s : p.measured) p.store_rows_m.push_back((int32_t)s->fpos);
        for (Snp* s : p.unmeasured) p.store_rows_u.push_back((int32_t)s->fpos);
        // every numeric entry point accepts row lists (windows: gauss_window_desc.rows_m/rows_u; LD-only calls:
        // gauss_ld_rows, gauss_gene_ld_batch_rows), so the genotypes stay in the mmap'd panel / in HBM; only the
        // minor-allele flip of prep_recessive_impute needs bytes on the host
        p.packed_rows = (kind != GAUSS_KIND_PREP_RECESSIVE);
        if (!p.packed_rows) materialise_from_packed(p);
    } else {
        // every selected population string must have its panel length, otherwise the matrix is ragged
        for (Snp* s : p.measured) {
            int n = 0;
            for (auto& g : s->geno) n += g.second;
            if (n != p.N) return herr("ERROR: genotype line of %s has %d samples, population table says %d", s->rsid.c_str(), n, p.N);
        }
        for (Snp* s : p.unmeasured) {
            int n = 0;
            for (auto& g : s->geno) n += g.second;
            if (n != p.N) return herr("ERROR: genotype line of %s has %d samples, population table says %d", s->rsid.c_str(), n, p.N);
        }
        fill_matrix(p.gm, p.measured, p.ld);                       // ReadGenotype, gauss.cpp:720-785
        fill_matrix(p.gu, p.unmeasured, p.ld);
    }
    p.z1.clear();
    for (Snp* s : p.measured) p.z1.push_back(s->z);
    tt[5] = tnow();
    tt[6] = tnow();       // the SNP-list table (gauss_prepared_snps) is built on first request
    if (trace)
        fprintf(stderr, "[prepare] desc %.2f  gwas %.2f  index %.2f  annotation %.2f  af-filter %.2f  partition %.2f  snp-table %.2f ms (map %zu, kept %zu)\n",
                tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], t_annot - tt[3], tt[4] - t_annot, tt[5] - tt[4], tt[6] - tt[5], p.snp_map.size(), p.snp_vec.size());
    return 0;
}

// ------------------------------------------------------------------------------------------
// small dense helpers for the JEPEG k x k tail (k <= 6): gene.cpp:317-550
// ------------------------------------------------------------------------------------------
static double pnorm_upper(double x) { return 0.5 * erfc(x / 1.4142135623730951); }   // R::pnorm5(x,0,1,0,0)

static double pchisq_upper(double x, int df)                                            // R::pchisq(x,df,0,0)
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

struct GeneResult {
    std::string geneid = ".", top_categ = ".", top_snp = ".";
    double chisq = -1.0, jepeg_pval = -1.0, top_categ_pval = -1.0, top_snp_pval = -1.0;
    int df = 0, num_snp = 0;
};

static const char* categ_name(int c)       // Categ::GetName, gene.cpp:17-33
{
    static const char* nm[6] = {"PFS", "TFB", "STR", "TAR", "CIS", "TRN"};
    return (c >= 0 && c < 6) ? nm[c] : "";
}

// Gene::RunJepeg + CalJepegPval tail (gene.cpp:88-185, 317-550) given CorG (n x n row-major)
static GeneResult jepeg_tail(const std::vector<Snp*>& gs, const double* CorG, const Args& a)
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

// ------------------------------------------------------------------------------------------
// outputs
// ------------------------------------------------------------------------------------------
static gauss_table* dist_output(gauss_prepared& p)     // dist.cpp:91-124 / distmix.cpp:100-133
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

static gauss_table* qcat_output(gauss_prepared& p)     // qcat.cpp:94-131 / qcatmix.cpp:102-139
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

static gauss_table* prep_output(gauss_prepared& p)     // prep_qcat.cpp:135-204 / prep_qcatmix.cpp:262-315
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

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
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

// Re-block a BGZF text file line by line (exercises reader + writer; used by tests and by tools that
// rewrite panels).  Returns the number of lines copied, or -1.
int64_t gauss_host_bgzf_copy(const char* in_path, const char* out_path)
{
    BgzfReader r;
    if (!in_path || !out_path || !r.open(in_path)) return herr("can't open '%s'", in_path ? in_path : "(null)");
    gauss_host::BgzfWriter w;
    if (!w.open(out_path)) return herr("can't create '%s'", out_path);
    std::string line;
    int64_t n = 0;
    for (;;) {
        const int last = r.getline(line);
        if (last == -2) return herr("codec error in '%s'", in_path);
        if (last == -1 && line.empty()) break;
        line.push_back('\n');
        if (!w.write(line.data(), line.size())) return herr("write error");
        n++;
        if (last == -1) break;
    }
    if (!w.close()) return herr("write error");
    return n;
}

int64_t gauss_host_pack_panel(const char* index_file, const char* data_file, const char* desc_file, const char* out_file)
{
    if (!index_file || !data_file || !desc_file || !out_file) { herr("file name is NULL"); return -1; }
    std::string err;
    const int64_t n = gauss_host::pack_panel(index_file, data_file, desc_file, out_file, err);
    if (n < 0) herr("%s", err.c_str());
    return n;
}

int gauss_host_panel_cache(const char* index_file, const char* data_file, const char* desc_file, int create,
                           char* out_path, int out_len, int64_t* snps_packed_now)
{
    if (!index_file || !data_file || !desc_file || !out_path || out_len < 2) return herr("bad arguments");
    std::string out, err;
    const int rc = resolve_packed_panel(index_file, data_file, desc_file, create != 0, out, err, snps_packed_now);
    if (rc < 0) return herr("%s", err.c_str());
    if (rc == 1) { out_path[0] = 0; return 1; }
    if ((int)out.size() + 1 > out_len) return herr("path buffer too small for '%s'", out.c_str());
    memcpy(out_path, out.c_str(), out.size() + 1);
    return 0;
}

void gauss_host_set_threads(int n) { g_host_threads = n < 1 ? 1 : (n > 64 ? 64 : n); }

int gauss_host_prepare(int kind, int chr, int64_t start_bp, int64_t end_bp, int64_t wing_size, const char* study_pop,
                       const char* const* pop_names, const double* pop_wgts, int n_pop_wgt, const char* input_file,
                       const char* annotation_file, const char* reference_index_file, const char* reference_data_file,
                       const char* reference_pop_desc_file, double af1_cutoff, gauss_prepared** out)
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

static int run_impute(gauss_ctx* ctx, int kind, int chr, int64_t start_bp, int64_t end_bp, int64_t wing, const char* study_pop,
                      const char* const* names, const double* wgts, int nw, const char* input, const char* index,
                      const char* data, const char* desc, double af1_cutoff, gauss_table** out)
{
    if (!ctx || !out) return herr("bad arguments");
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

static int run_jepeg(gauss_ctx* ctx, int kind, const char* study_pop, const char* const* names, const double* wgts, int nw,
                     const char* input, const char* annotation, const char* index, const char* data, const char* desc,
                     double af1_cutoff, gauss_table** out)
{
    if (!ctx || !out) return herr("bad arguments");
    gauss_prepared* p = nullptr;
    if (gauss_host_prepare(kind, 0, 0, 0, 0, study_pop, names, wgts, nw, input, annotation, index, data, desc, af1_cutoff, &p)) return -1;
    std::unique_ptr<gauss_prepared> hold(p);
    const Args& a = p->args;
    const int S = (int)p->measured.size();
    const int ng = p->gene_off.empty() ? 0 : (int)p->gene_off.size() - 1;
    std::vector<double> blocks;
    std::vector<size_t> boff;
    size_t tot = 0;
    for (int g = 0; g < ng; g++) { boff.push_back(tot); const size_t n = p->gene_off[g + 1] - p->gene_off[g]; tot += n * n; }
    blocks.assign(std::max<size_t>(tot, 1), 0.0);
    if (S > 0 && ng > 0) {
        // CorG of every gene in one launch, diagonal 1 + lambda (gene.cpp:306-315 / 576-586)
        const int mode = (kind == GAUSS_KIND_JEPEG) ? GAUSS_MODE_POOLED : GAUSS_MODE_WEIGHTED;
        if (p->packed_rows) {
            const uint8_t* store = nullptr;
            int on_device = 0;
            if (packed_row_source(ctx, *p, &store, &on_device)) return -1;
            if (gauss_gene_ld_batch_rows(ctx, mode, store, a.pk->row_bytes(), GAUSS_GENO_2BIT, p->store_rows_m.data(), S,
                                         p->pop_off.data(), p->pop_src_off.data(), p->pop_wgt.data(), (int)p->pop_off.size() - 1,
                                         p->gene_off.data(), ng, 1.0 + a.lambda, on_device, blocks.data()) != 0)
                return herr("%s", gauss_last_error());
        } else if (gauss_gene_ld_batch(ctx, mode, p->gm.data(), S, p->ld, p->pop_off.data(), p->pop_wgt.data(),
                                       (int)p->pop_off.size() - 1, p->gene_off.data(), ng, 1.0 + a.lambda, blocks.data()) != 0)
            return herr("%s", gauss_last_error());
    }
    std::unique_ptr<gauss_table> t(new gauss_table());
    Column geneid{"geneid", GAUSS_COL_STR, {}, {}, {}}, chisq{"chisq", GAUSS_COL_DBL, {}, {}, {}}, df{"df", GAUSS_COL_INT, {}, {}, {}};
    Column jp{"jepeg_pval", GAUSS_COL_DBL, {}, {}, {}}, ns{"num_snp", GAUSS_COL_INT, {}, {}, {}}, tc{"top_categ", GAUSS_COL_STR, {}, {}, {}};
    Column tcp{"top_categ_pval", GAUSS_COL_DBL, {}, {}, {}}, ts{"top_snp", GAUSS_COL_STR, {}, {}, {}}, tsp{"top_snp_pval", GAUSS_COL_DBL, {}, {}, {}};
    for (int g = 0; g < ng; g++) {
        std::vector<Snp*> gs(p->measured.begin() + p->gene_off[g], p->measured.begin() + p->gene_off[g + 1]);
        const GeneResult r = jepeg_tail(gs, blocks.data() + boff[g], a);
        geneid.s.push_back(r.geneid); chisq.d.push_back(r.chisq); df.i.push_back(r.df); jp.d.push_back(r.jepeg_pval);
        ns.i.push_back(r.num_snp); tc.s.push_back(r.top_categ); tcp.d.push_back(r.top_categ_pval);
        ts.s.push_back(r.top_snp); tsp.d.push_back(r.top_snp_pval);
    }
    t->cols = {geneid, chisq, df, jp, ns, tc, tcp, ts, tsp};          // jepeg.cpp:143-151
    *out = t.release();
    return 0;
}

int gauss_host_jepeg(gauss_ctx* ctx, const char* study_pop, const char* input_file, const char* annotation_file,
                     const char* reference_index_file, const char* reference_data_file, const char* reference_pop_desc_file,
                     double af1_cutoff, gauss_table** out)
{
    return run_jepeg(ctx, GAUSS_KIND_JEPEG, study_pop, nullptr, nullptr, 0, input_file, annotation_file, reference_index_file,
                     reference_data_file, reference_pop_desc_file, af1_cutoff, out);
}

int gauss_host_jepegmix(gauss_ctx* ctx, const char* const* pop_names, const double* pop_wgts, int n_pop_wgt,
                        const char* input_file, const char* annotation_file, const char* reference_index_file,
                        const char* reference_data_file, const char* reference_pop_desc_file, double af1_cutoff,
                        gauss_table** out)
{
    return run_jepeg(ctx, GAUSS_KIND_JEPEGMIX, nullptr, pop_names, pop_wgts, n_pop_wgt, input_file, annotation_file,
                     reference_index_file, reference_data_file, reference_pop_desc_file, af1_cutoff, out);
}

}  // extern "C"

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
static int panel_make_resident(gauss_ctx* ctx, const std::string& path, void** dev, int64_t* uploaded, bool async)
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
static bool panel_is_resident(gauss_ctx* ctx, const std::string& path, void** dev, bool wait)
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

// minimal fork-join helper: fn(i) for i in [0, n) on up to nt threads
template <typename F>
static void parallel_for(int n, int nt, F fn)
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

static bool env_flag(const char* name, bool dflt)
{
    const char* e = getenv(name);
    return e ? atoi(e) != 0 : dflt;
}

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
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

// ------------------------------------------------------------------------------------------
// The chromosome driver's window: prepare()'s data layer as ONE merge of two sorted arrays.
//
// prepare() restates the reference literally: every SNP of the extended window becomes an object with five strings in a std::map
// keyed by (chr, bp, a1, a2) (ReadInputZ, ReadReferenceIndex, MakeSnpVec: gauss.cpp:121-190, 293-399, 543-693) -- 0.65-1.0 ms
// for a window of 3 000 SNPs, paid again by every window of a chromosome, and with one rank of eight holding four windows it is
// what the GPU waits for.  A window of a SORTED packed panel needs none of it.  The study's rows
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
// What a window costs now: DESIGN.md section 9e item 11.  tests/test_feeder.py holds it against prepare() on random
// studies with multi-allelic, duplicated, swapped and study-only sites, for all four kinds.
// ------------------------------------------------------------------------------------------
struct ChromSetup {                     // what prepare() derives from a call's arguments alone: once per call, not once per window
    int kind = 0;
    bool mix = false, qcat = false;
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
};

// 0, or -1 with the message prepare() would have given every window (the caller then lets prepare() give it)
static int chrom_setup(ChromSetup& cs, int kind, int chr, int64_t wing_size, const char* study_pop, const char* const* pop_names,
                       const double* pop_wgts, int n_pop_wgt, const char* input_file, const std::string& packed_path,
                       const char* desc_file, double af1_cutoff, const std::shared_ptr<PackedPanel>& pk,
                       const std::shared_ptr<const GwasCache>& gw)
{
    cs.kind = kind;
    cs.mix = (kind == GAUSS_KIND_DISTMIX || kind == GAUSS_KIND_QCATMIX);
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

static int lean_window_build(LeanWindow& w, const ChromSetup& cs, long long start_bp, long long end_bp)
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
        if (type == 0 && (bp < start_bp || bp > end_bp)) return;       // a wing's unmeasured SNP: nothing reads it
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
static int lean_window_desc(LeanWindow& w, gauss_window_desc* d)
{
    const ChromSetup& cs = *w.cs;
    const Args& a = cs.a;
    const int M = (int)w.measured.size(), U = (int)w.unmeasured.size();
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
    d->rows_m = w.store_rows_m.data(); d->rows_u = w.store_rows_u.data();
    d->pop_src_off = cs.pop_src_off.data();
    d->z1 = w.z1.data(); d->lambda = a.lambda; d->min_abs_eig = a.min_abs_eig;
    d->out_status = &w.status;
    if (cs.qcat) {
        w.out_r.assign((size_t)w.n_predm + U, 0.0);
        w.num_eig = M;
        d->kind = GAUSS_WIN_QCAT;
        d->n_head_measured = w.n_head; d->n_pred_measured = w.n_predm; d->eig_cutoff = a.eig_cutoff;
        d->out_r = w.out_r.data(); d->out_num_eig = &w.num_eig;
        if (w.n_predm + U < 1) return herr("QCAT window has no SNP to test");
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
static void lean_table_prebuild(LeanWindow& w)
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

static gauss_table* lean_window_finish(LeanWindow& w)
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
    int64_t counters0[4] = {0, 0, 0, 0};
    (void)gauss_hip_counters(ctx, counters0);

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
    std::vector<long long> gbp;
    for (const GwasRow& r : gw->rows)
        if (chr <= 0 || r.chr == chr) gbp.push_back(r.bp);
    std::sort(gbp.begin(), gbp.end());
    struct Win { int64_t s, e; double cost; int owner, status, M, U; std::string why; };
    std::vector<Win> wins;
    for (int64_t s0 = start_bp; s0 <= end_bp; s0 += window_size) {
        Win w;
        w.s = s0; w.e = std::min(end_bp, s0 + window_size - 1);
        const double m = (double)(std::upper_bound(gbp.begin(), gbp.end(), (long long)(w.e + wing_size)) -
                                  std::lower_bound(gbp.begin(), gbp.end(), (long long)(w.s - wing_size)));
        double u = 0;
        if (pk->header().sorted && chr > 0) {
            const double in_panel = (double)(pk->lower_bound(chr, w.e + 1) - pk->lower_bound(chr, w.s));
            const double m_pred = (double)(std::upper_bound(gbp.begin(), gbp.end(), (long long)w.e) -
                                           std::lower_bound(gbp.begin(), gbp.end(), (long long)w.s));
            u = std::max(0.0, in_panel - m_pred);
        }
        w.cost = issued_cost_per_sample((int)m, (int)u) + 1.0;   // what the Gram kernel issues for the window, per sample (above)
        w.owner = 0; w.status = -1; w.M = (int)m; w.U = (int)u;
        wins.push_back(w);
    }
    {   // Whole windows by longest processing time first (ties by index), then a local search -- the idea of
        // farm.level_windows without the cuts, on a simpler cost model: a window costs its LD flops per sample (the
        // Python planner also prices B11's factorisation and the per-SNP tail, so the two plans may pick different
        // owners; each is used consistently by every rank of its own driver).  A rank's load is the cost of its windows
        // plus the factorisation chain of its tallest one (the chain is latency bound on the few windows a rank
        // holds: DESIGN.md section 6; farm.CHAIN_STEP_COST, here per sample).
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
    // window has nothing to hide its factorisation chain under; 8-rank shares 7.4-8.4 ms against 7.8-8.2.  DESIGN.md 9e item 10.)
    const size_t lead = 0;
    const size_t n_rest = mine.size() - lead;
    if (n_batches < 1) n_batches = (int)lead + (n_rest >= 16 ? (first_use && async_upload ? 6 : 4) : (n_rest >= 9 ? 3 : (n_rest >= 4 ? 2 : 1)));
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
    const bool lean = chr > 0 && pk->header().sorted &&
                      chrom_setup(cs, kind, chr, wing_size, study_pop, pop_names, pop_wgts, n_pop_wgt, input_file, packed_path,
                                  reference_pop_desc_file, af1_cutoff, pk, gw) == 0;
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
    std::thread feeder([&]() {
        parallel_for((int)order.size(), nthreads, [&](int q) {
            const int b = order[q].first, k = order[q].second;
            Win& w = wins[batches[b][k]];
            Slot& sl = slots[b][k];
            gauss_prepared* p = nullptr;
            if (lean) {
                std::unique_ptr<LeanWindow> lw(new LeanWindow());
                if (lean_window_build(*lw, cs, w.s, w.e)) { w.status = 2; w.why = gauss_host_last_error(); }
                else {
                    w.M = (int)lw->measured.size(); w.U = (int)lw->unmeasured.size();
                    if (lean_window_desc(*lw, &sl.d)) { w.status = 1; w.why = gauss_host_last_error(); }      // the ">10" guards (dist.cpp:145-151)
                    else { sl.lw = std::move(lw); sl.ok = true; }
                }
            } else if (gauss_host_prepare(kind, chr, w.s, w.e, wing_size, study_pop, pop_names, pop_wgts, n_pop_wgt, input_file, nullptr,
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
        });
    });

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
    bool first = true;
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
    auto retire = [&](int b) {
        // results of batch b -> SNP objects -> per-window tables (host threads; the GPU is on batch b+1 meanwhile)
        // (what the tables hold that the results do not change is built BEFORE the wait: the last batch has no batch b+1 to hide under)
        double tp = now_s();
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
                    Win& w = wins[batches[b][k]];
                    w.status = 2; w.why = std::string(gauss_last_error()) + " (batch error: " + why + ")";
                    sl.ok = false;
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
            if (sl.lw) { sl.tab = lean_window_finish(*sl.lw); wins[batches[b][k]].status = 0; }
            else if (gauss_prepared_finish(sl.p.get(), &t) == 0) { sl.tab = t; wins[batches[b][k]].status = 0; }
            else { wins[batches[b][k]].status = 2; wins[batches[b][k]].why = gauss_host_last_error(); }
            sl.p.reset();
            sl.lw.reset();
        });
        const double t_fin = now_s();
        append_batch(b);
        st.t_tables += now_s() - tt;
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
                    for (int32_t r : sq.lw ? sq.lw->store_rows_m : sq.p->store_rows_m) top = std::max<int64_t>(top, r);
                    for (int32_t r : sq.lw ? sq.lw->store_rows_u : sq.p->store_rows_u) top = std::max<int64_t>(top, r);
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
                        Win& w = wins[batches[b][k]];
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
    feeder.join();
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
    if (first) {       // no window produced rows: still hand back the reference's column set
        const bool mix = (kind == GAUSS_KIND_DISTMIX || kind == GAUSS_KIND_QCATMIX);
        const bool qc = (kind == GAUSS_KIND_QCAT || kind == GAUSS_KIND_QCATMIX);
        all->add("rsid", GAUSS_COL_STR); all->add("chr", GAUSS_COL_INT); all->add("bp", GAUSS_COL_INT);
        all->add("a1", GAUSS_COL_STR); all->add("a2", GAUSS_COL_STR); all->add(mix ? "af1mix" : "af1ref", GAUSS_COL_DBL);
        all->add("z", GAUSS_COL_DBL);
        if (qc) { all->add("qcat_m", GAUSS_COL_INT); all->add("qcat_t", GAUSS_COL_DBL); all->add("qcat_chisq", GAUSS_COL_DBL); all->add("qcat_pval", GAUSS_COL_DBL); }
        else { all->add("pval", GAUSS_COL_DBL); all->add("info", GAUSS_COL_DBL); }
        all->add("type", GAUSS_COL_INT);
    }
    all->cols.push_back(win_col);
    {
        NamedMat nm;
        nm.name = "windows"; nm.nrow = (int)wins.size(); nm.ncol = 6;
        nm.d.assign((size_t)nm.nrow * 6, 0.0);
        for (int i = 0; i < nm.nrow; i++) {
            const Win& w = wins[i];
            const double v[6] = {(double)w.s, (double)w.e, (double)w.owner, (double)w.status, (double)w.M, (double)w.U};
            for (int c = 0; c < 6; c++) nm.d[(size_t)c * nm.nrow + i] = v[c];
            if (w.status == 1) st.n_skipped++;
            if (w.status == 2) { st.n_failed++; if (all->messages.size() < 64) all->messages.push_back("window " + std::to_string(i) + ": " + w.why); }
        }
        all->named.push_back(nm);
    }
    st.t_tables += now_s() - tt;
    st.t_total = now_s() - t_begin;
    {
        int64_t c1[4] = {0, 0, 0, 0};
        if (gauss_hip_counters(ctx, c1) == 0) st.n_merged_giveups = (int32_t)(c1[2] - counters0[2]);
    }
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

// A whole string column as one fixed-width, NUL-padded byte matrix [nrow x *width] (numpy dtype "S<width>"):
// 90 000 rows come across the boundary as one buffer instead of 90 000 Python strings.
const char* gauss_table_strcol_fixed(const gauss_table* t, int c, int* width)
{
    if (!t || c < 0 || c >= (int)t->cols.size() || t->cols[c].type != GAUSS_COL_STR) return nullptr;
    const Column& col = t->cols[c];
    size_t w = 1;
    for (const std::string& v : col.s) w = std::max(w, v.size());
    if (col.fixed.size() != w * col.s.size() || col.fixed_w != (int)w) {
        col.fixed.assign(w * col.s.size(), '\0');
        for (size_t r = 0; r < col.s.size(); r++) memcpy(&col.fixed[r * w], col.s[r].data(), col.s[r].size());
        col.fixed_w = (int)w;
    }
    if (width) *width = (int)w;
    return col.fixed.data();
}

}  // extern "C"
