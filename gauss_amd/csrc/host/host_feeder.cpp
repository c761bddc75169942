// libgauss_host.so -- host data layer + the five reference entry points (include/gauss_host.h).
//
// This file: the readers and filters, the packed-panel cache, prepare().
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
#include "host_internal.h"

BlockPools& block_pools() { static BlockPools* bp = new BlockPools(); return *bp; }

// read_ref_desc (gauss.cpp:951-993)
int read_ref_desc(Args& a)
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
int init_pop_flag_vec(Args& a)
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
void init_pop_flag_wgt_vec(Args& a)
{
    for (int i = 0; i < a.num_pops; i++) {
        auto it = a.pop_wgt_map.find(a.ref_pop_vec[i]);
        if (it != a.pop_wgt_map.end()) { a.pop_flag_vec.push_back(1); a.pop_wgt_vec.push_back(it->second); }
        else a.pop_flag_vec.push_back(0);
    }
}

void set_pop_wgt_map(Args& a, const char* const* names, const double* w, int n)
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

std::shared_ptr<const GwasCache> load_gwas_cached(const std::string& path, std::string& err)
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
    for (size_t q = 0; q < c->by_pos.size(); q++) {
        const GwasRow& r = c->rows[c->by_pos[q]];
        const bool twice = (q + 1 < c->by_pos.size() && c->rows[c->by_pos[q + 1]].chr == r.chr && c->rows[c->by_pos[q + 1]].bp == r.bp);
        if ((twice || r.a1 == r.a2) && (c->odd_positions.empty() || c->odd_positions.back() != std::make_pair(r.chr, r.bp)))
            c->odd_positions.emplace_back(r.chr, r.bp);
    }
    if (cache.size() >= 8) cache.clear();          // a handful of studies per process at most
    cache[k] = c;
    return c;
}

struct AnnotCache;
static std::shared_ptr<const AnnotCache> load_annotation_cached(const std::string& path, std::string& err);
static const std::vector<std::pair<int, long long>>& annotation_positions(const AnnotCache& an);

// ReadInputZ (gauss.cpp:121-190)
int ReadInputZ(SnpMap& m, const Args& a, bool All)
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
    auto enter = [&](const GwasRow& r) {
        SnpPtr s = m.make();
        s->rsid = r.rsid; s->chr = r.chr; s->bp = r.bp; s->a1 = r.a1; s->a2 = r.a2; s->z = r.z;
        s->info = 1.0;     // gauss.cpp:142
        s->type = 2;       // gauss.cpp:176
        // (a study file sorted by position appends: no descent through the tree; a later row of the same key still replaces the earlier one)
        MapKey key{r.chr, r.bp, r.a1, r.a2};
        auto it = (m.empty() || m.rbegin()->first < key) ? m.emplace_hint(m.end(), std::move(key), SnpPtr()) : m.try_emplace(std::move(key)).first;
        it->second = std::move(s);
    };
    // The gene drivers' map (Args::annotated_only): the study's rows at the positions the annotation names and at the study's odd
    // ones, each position's rows found in the (chr, bp)-ordered index -- positions ascending, a position's rows in file order (a
    // key listed twice ends with its later row, as in the file-order loop below; rows of different keys never meet).
    if (All && a.annotated_only && !a.annotation_file.empty()) {
        std::shared_ptr<const AnnotCache> an = load_annotation_cached(a.annotation_file, err);
        if (!an) return herr("%s", err.c_str());
        const std::vector<std::pair<int, long long>>& named = annotation_positions(*an);
        const std::vector<std::pair<int, long long>>& odd = gw->odd_positions;
        auto before = [&](uint32_t x, const std::pair<int, long long>& at) { const GwasRow& r = gw->rows[x]; return r.chr < at.first || (r.chr == at.first && r.bp < at.second); };
        size_t i = 0, j = 0;
        while (i < named.size() || j < odd.size()) {
            std::pair<int, long long> at;
            if (j >= odd.size() || (i < named.size() && named[i] <= odd[j])) { at = named[i]; if (j < odd.size() && odd[j] == at) j++; i++; }
            else at = odd[j++];
            for (auto it = std::lower_bound(gw->by_pos.begin(), gw->by_pos.end(), at, before); it != gw->by_pos.end(); ++it) {
                const GwasRow& r = gw->rows[*it];
                if (r.chr != at.first || r.bp != at.second) break;
                enter(r);
            }
        }
        return 0;
    }
    for (size_t q = q0; q < q1; q++) {
        const GwasRow& r = gw->rows[ranged ? gw->by_pos[q] : q];
        if (!All) {
            if ((a.chr > 0) && (a.chr != r.chr)) continue;
            if ((a.start_bp - a.wing_size) > r.bp || (a.end_bp + a.wing_size) < r.bp) continue;
        }
        enter(r);
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
int merge_index_entry(SnpMap& m, const Args& a, bool All, const std::string& rsid, int chr, long long bp,
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
int ReadReferenceIndex(SnpMap& m, const Args& a, bool All)
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
void load_line(BgzfReader& fp, Snp& s, const Args& a, std::vector<double>* af_out)
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
struct AnnotCache {
    std::vector<AnnotRow> rows;
    std::vector<std::pair<int, long long>> positions;      // the (chr, bp) the file names, sorted, each once
};

static const std::vector<std::pair<int, long long>>& annotation_positions(const AnnotCache& an) { return an.positions; }

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
    for (const AnnotRow& r : c->rows) c->positions.emplace_back(r.chr, r.bp);
    std::sort(c->positions.begin(), c->positions.end());
    c->positions.erase(std::unique(c->positions.begin(), c->positions.end()), c->positions.end());
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
        // the entries of this position follow `pos`: the two allele orders are looked for among them (what two more descents
        // through the tree with freshly built keys would find)
        auto it1 = m.end(), it2 = m.end();
        for (auto it = pos; it != m.end() && it->first.chr == chr && it->first.bp == bp; ++it) {
            if (it->first.a1 == a1 && it->first.a2 == a2) it1 = it;
            if (it->first.a1 == a2 && it->first.a2 == a1) it2 = it;
        }
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
std::shared_ptr<PackedPanel> open_packed_shared(const std::string& path, std::string& err)
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
int resolve_packed_panel(const std::string& index_file, const std::string& data_file, const std::string& desc_file,
                                bool create, std::string& out, std::string& err, int64_t* packed_now)
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
int auto_pack_mode()
{
    const char* e = getenv("GAUSS_AUTO_PACK");
    return e ? (atoi(e) != 0 ? 1 : 0) : -1;
}


void fill_matrix(std::vector<uint8_t>& G, const std::vector<Snp*>& rows, int64_t ld)
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
void unpack_rows(const gauss_prepared& p, const std::vector<Snp*>& rows, std::vector<uint8_t>& G)
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

void materialise_from_packed(gauss_prepared& p)
{
    unpack_rows(p, p.measured, p.gm);
    unpack_rows(p, p.unmeasured, p.gu);
}

void build_snp_table(gauss_prepared& p)
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

int prepare(gauss_prepared& p)
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


extern "C" {

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

}  // extern "C"
