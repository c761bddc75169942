#include "packed_panel.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <fstream>
#include <mutex>
#include <thread>

#include "bgzf_io.h"

namespace gauss_host {

static const char kMagic[8] = {'G', 'A', 'U', 'S', 'S', 'P', 'K', '1'};

PackedPanel::~PackedPanel() { close(); }

void PackedPanel::close()
{
    if (base_) munmap(const_cast<uint8_t*>(base_), bytes_);
    if (fd_ >= 0) ::close(fd_);
    base_ = nullptr; fd_ = -1; bytes_ = 0; hdr_ = nullptr;
}

bool PackedPanel::is_packed(const std::string& path)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char m[8] = {0};
    const size_t n = fread(m, 1, 8, f);
    fclose(f);
    return n == 8 && memcmp(m, kMagic, 8) == 0;
}

bool PackedPanel::open(const std::string& path, std::string& err)
{
    close();
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) { err = "ERROR: can't open reference data file '" + path + "'"; return false; }
    struct stat st;
    if (fstat(fd_, &st) != 0 || (size_t)st.st_size < sizeof(PkHeader)) { err = "packed panel '" + path + "' is truncated"; close(); return false; }
    bytes_ = (size_t)st.st_size;
    void* p = mmap(nullptr, bytes_, PROT_READ, MAP_SHARED, fd_, 0);
    if (p == MAP_FAILED) { err = "mmap of packed panel '" + path + "' failed"; base_ = nullptr; close(); return false; }
    base_ = (const uint8_t*)p;
    hdr_ = (const PkHeader*)base_;
    const PkHeader& h = *hdr_;
    // the header is untrusted input: every section range is checked with overflow-safe arithmetic, then the
    // records that carry offsets (population blocks, SNP strings) are checked against what they point into
    auto section = [&](uint64_t off, uint64_t count, uint64_t elem, uint64_t align) {
        uint64_t sz = 0;
        if (__builtin_mul_overflow(count, elem, &sz)) return false;
        return off % align == 0 && off <= (uint64_t)bytes_ && sz <= (uint64_t)bytes_ - off;
    };
    uint64_t n_cells = 0;
    bool ok = memcmp(h.magic, kMagic, 8) == 0 && h.version == 1 && h.file_bytes == bytes_ && h.row_bytes % 16 == 0 &&
              h.row_bytes > 0 && h.n_pop >= 1 && h.n_pop <= 4096 && h.n_snp <= (uint64_t)INT32_MAX &&
              !__builtin_mul_overflow(h.n_snp, (uint64_t)h.n_pop, &n_cells) &&
              section(h.off_pops, h.n_pop, sizeof(PkPop), 8) && section(h.off_snps, h.n_snp, sizeof(PkSnp), 8) &&
              section(h.off_strings, 0, 1, 1) && h.off_strings <= h.off_af &&
              section(h.off_af, n_cells, sizeof(double), 8) && section(h.off_cnt, n_cells, sizeof(int32_t), 4) &&
              section(h.off_geno, h.n_snp, h.row_bytes, 16);
    if (!ok) { err = "packed panel '" + path + "' has a bad header"; close(); return false; }
    {
        const PkPop* pp = (const PkPop*)(base_ + h.off_pops);
        for (uint32_t k = 0; k < h.n_pop && ok; k++) {
            const uint64_t blk = ((uint64_t)pp[k].size + 63) / 64 * 16;
            ok = pp[k].byte_off % 16 == 0 && (uint64_t)pp[k].byte_off + blk <= h.row_bytes &&
                 memchr(pp[k].name, 0, sizeof(pp[k].name)) != nullptr && memchr(pp[k].super, 0, sizeof(pp[k].super)) != nullptr;
        }
        if (!ok) { err = "packed panel '" + path + "': a population block lies outside the genotype row"; close(); return false; }
        // strings live in [off_strings, off_af); the last byte of that range must be NUL so that every offset
        // below its length names a terminated string
        const uint64_t str_bytes = h.off_af - h.off_strings;
        const PkSnp* sp = (const PkSnp*)(base_ + h.off_snps);
        ok = h.n_snp == 0 || (str_bytes > 0 && base_[h.off_af - 1] == 0);
        for (uint64_t i = 0; i < h.n_snp && ok; i++)
            ok = sp[i].rsid < str_bytes && sp[i].a1 < str_bytes && sp[i].a2 < str_bytes;
        if (!ok) { err = "packed panel '" + path + "': a SNP record points outside the string table"; close(); return false; }
    }
    // The tables in front of the genotype section (strings, per-population AF and allele counts: ~41 MB for 100 000 SNPs
    // x 29 populations) are read by every window's data layer, from many threads at once; left to demand paging the
    // first windows of a process spent 9-22 ms each in page faults on this mapping (0.5 ms once mapped).  Map them now,
    // in one batched call where the kernel has it.
    {
        const size_t page = (size_t)sysconf(_SC_PAGESIZE);
        const size_t lo = (size_t)h.off_pops / page * page;
        const size_t hi = std::min((size_t)bytes_, ((size_t)h.off_geno + page - 1) / page * page);
        bool mapped = false;
#ifdef MADV_POPULATE_READ
        mapped = hi > lo && madvise(const_cast<uint8_t*>(base_) + lo, hi - lo, MADV_POPULATE_READ) == 0;
#endif
        if (!mapped && hi > lo) {
            (void)madvise(const_cast<uint8_t*>(base_) + lo, hi - lo, MADV_WILLNEED);
            volatile uint8_t sink = 0;
            for (size_t o = lo; o < hi; o += page) sink = sink + base_[o];
        }
    }
    pops_ = (const PkPop*)(base_ + h.off_pops);
    snps_ = (const PkSnp*)(base_ + h.off_snps);
    strings_ = (const char*)(base_ + h.off_strings);
    af_ = (const double*)(base_ + h.off_af);
    cnt_ = (const int32_t*)(base_ + h.off_cnt);
    geno_ = base_ + h.off_geno;
    return true;
}

int64_t PackedPanel::lower_bound(int chr, int64_t bp) const
{
    int64_t lo = 0, hi = n_snp();
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        const PkSnp& s = snps_[mid];
        if (s.chr < chr || (s.chr == chr && s.bp < bp)) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

namespace {
// white space as isspace() in the "C" locale, from a table: a panel line is ~33 kB and every byte of it passes through here
struct WsTable { bool t[256]; WsTable() { for (int c = 0; c < 256; c++) t[c] = (c == ' ' || (c >= 9 && c <= 13)); } };
static const WsTable kWs;
struct Fields {
    const char* p; const char* e;
    bool next(const char*& b, int& n)
    {
        while (p < e && kWs.t[(unsigned char)*p]) p++;
        if (p >= e) return false;
        b = p;
        // long tokens (a population's genotype digits) end at a blank in practice: memchr finds it, then make sure that no
        // other white-space character came first
        const char* q = (const char*)memchr(p, ' ', (size_t)(e - p));
        const char* lim = q ? q : e;
        const char* r = p;
        if (lim - r > 64) {
            for (const char* t = r; t < lim; t++) if (kWs.t[(unsigned char)*t]) { lim = t; break; }
            r = lim;
        } else {
            while (r < lim && !kWs.t[(unsigned char)*r]) r++;
        }
        p = r;
        n = (int)(p - b);
        return true;
    }
};
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}  // namespace

int64_t pack_panel(const std::string& index_path, const std::string& data_path, const std::string& desc_path,
                   const std::string& out_path, std::string& err)
{
    // population description: name, size, super population (gauss.cpp:951-993)
    std::vector<PkPop> pops;
    {
        std::ifstream in(desc_path.c_str());
        if (!in) { err = "ERROR: can't open reference population description file '" + desc_path + "'"; return -1; }
        std::string line;
        std::getline(in, line);
        uint32_t off = 0;
        while (std::getline(in, line)) {
            Fields t{line.data(), line.data() + line.size()};
            const char* b; int n;
            if (!t.next(b, n)) continue;
            PkPop p;
            memset(&p, 0, sizeof(p));
            memcpy(p.name, b, (size_t)std::min(n, 23));
            if (t.next(b, n)) p.size = (uint32_t)strtoul(std::string(b, n).c_str(), nullptr, 10);
            if (t.next(b, n)) memcpy(p.super, b, (size_t)std::min(n, 23));
            p.byte_off = off;
            off += (p.size + 63) / 64 * 16;
            pops.push_back(p);
        }
        if (pops.empty()) { err = "population description '" + desc_path + "' lists no populations"; return -1; }
    }
    const int P = (int)pops.size();
    const uint64_t row_bytes = std::max<uint64_t>(16, pops.back().byte_off + (pops.back().size + 63) / 64 * 16);

    BgzfReader idx;
    if (!idx.open(index_path)) { err = "ERROR: can't open reference index file '" + index_path + "'"; return -1; }
    { BgzfReader probe; if (!probe.open(data_path)) { err = "ERROR: can't open reference data file '" + data_path + "'"; return -1; } }

    // ---- pass 1: the index (one short line per SNP) -- with it every section's place in the output file is known ----
    std::vector<PkSnp> snps;
    std::vector<char> strings;
    std::vector<long long> fpos;
    auto add_str = [&](const char* b, int n) { const uint32_t o = (uint32_t)strings.size(); strings.insert(strings.end(), b, b + n); strings.push_back(0); return o; };
    bool sorted = true;
    {
        std::string line;
        for (;;) {
            const int last = idx.getline(line);
            if (last == -2) { err = "Error: can't read reference index file '" + index_path + "'"; return -1; }
            if (last == -1 && line.empty()) break;
            Fields t{line.data(), line.data() + line.size()};
            const char *b_rs, *b_chr, *b_bp, *b_a1, *b_a2, *b_af, *b_fp;
            int n_rs, n_chr, n_bp, n_a1, n_a2, n_af, n_fp;
            if (t.next(b_rs, n_rs) && t.next(b_chr, n_chr) && t.next(b_bp, n_bp) && t.next(b_a1, n_a1) && t.next(b_a2, n_a2) &&
                t.next(b_af, n_af) && t.next(b_fp, n_fp)) {
                PkSnp sn;
                sn.chr = (int32_t)strtol(std::string(b_chr, n_chr).c_str(), nullptr, 10);
                sn.bp = strtoll(std::string(b_bp, n_bp).c_str(), nullptr, 10);
                sn.rsid = add_str(b_rs, n_rs); sn.a1 = add_str(b_a1, n_a1); sn.a2 = add_str(b_a2, n_a2);
                if (!snps.empty() && (sn.chr < snps.back().chr || (sn.chr == snps.back().chr && sn.bp < snps.back().bp))) sorted = false;
                snps.push_back(sn);
                fpos.push_back(strtoll(std::string(b_fp, n_fp).c_str(), nullptr, 10));
            }
            line.clear();
            if (last == -1) break;
        }
    }
    const size_t S = snps.size();

    PkHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, kMagic, 8);
    h.version = 1; h.n_pop = (uint32_t)P; h.n_snp = S; h.row_bytes = row_bytes; h.sorted = sorted ? 1u : 0u;
    size_t off = sizeof(PkHeader);
    h.off_pops = off; off = align_up(off + (size_t)P * sizeof(PkPop), 64);
    h.off_snps = off; off = align_up(off + S * sizeof(PkSnp), 64);
    h.off_strings = off; off = align_up(off + strings.size(), 64);
    h.off_af = off; off = align_up(off + S * (size_t)P * sizeof(double), 64);
    h.off_cnt = off; off = align_up(off + S * (size_t)P * sizeof(int32_t), 4096);
    h.off_geno = off; off += S * (size_t)row_bytes;
    h.file_bytes = off;

    const int fd = ::open(out_path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) { err = "can't write '" + out_path + "'"; return -1; }
    auto fail_out = [&](const std::string& why) { err = why; ::close(fd); remove(out_path.c_str()); return (int64_t)-1; };
    if (ftruncate(fd, (off_t)h.file_bytes) != 0) return fail_out("can't size '" + out_path + "'");
    auto put_at = [&](size_t at, const void* src, size_t n) {
        const char* q = (const char*)src;
        while (n > 0) {
            const ssize_t w = pwrite(fd, q, n, (off_t)at);
            if (w <= 0) return false;
            q += w; at += (size_t)w; n -= (size_t)w;
        }
        return true;
    };

    // ---- pass 2: the data lines (one ~33 kB text line per SNP, found by its BGZF virtual offset) are inflated, parsed and
    // 2-bit packed by a pool of threads, each with its own reader, in runs of consecutive SNPs (neighbouring lines share BGZF
    // blocks -- a 33 kB line is half a block -- so interleaving single lines would make every thread inflate every block); a
    // run's rows go straight to their place in the output file (pwrite), the allele tables into the SNP's slots.  The
    // conversion is a one-off per panel, but a genome-wide 33KG panel has millions of lines. ----
    std::vector<double> af(S * (size_t)P, 0.0);
    std::vector<int32_t> cnt(S * (size_t)P, 0);
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (int)std::max<size_t>(1, std::min<size_t>(std::min(16u, hw ? hw : 4u), (S + 127) / 128));
    std::vector<BgzfReader> readers((size_t)nt);
    for (BgzfReader& r : readers)
        if (!r.open(data_path)) return fail_out("ERROR: can't open reference data file '" + data_path + "'");
    const size_t RUN = 128;
    std::atomic<size_t> next{0};
    std::mutex emu;
    std::string first_err;
    std::atomic<bool> stop{false};
    auto work = [&](int tid) {
        BgzfReader& dat = readers[(size_t)tid];
        std::string dline;
        std::vector<uint8_t> rows(RUN * row_bytes);
        auto give_up = [&](const std::string& why) {
            std::lock_guard<std::mutex> lock(emu);
            if (first_err.empty()) first_err = why;
            stop = true;
        };
        for (size_t c0 = next.fetch_add(RUN); c0 < S && !stop; c0 = next.fetch_add(RUN)) {
            const size_t c1 = std::min(S, c0 + RUN);
            memset(rows.data(), 0, (c1 - c0) * row_bytes);
            for (size_t i = c0; i < c1; i++) {
                dat.seek(fpos[i]);
                dline.clear();
                dat.getline(dline);
                Fields d{dline.data(), dline.data() + dline.size()};
                uint8_t* row = rows.data() + (i - c0) * row_bytes;
                for (int k = 0; k < P; k++) {
                    const char* g; int n;
                    // A population's genotype string is pops[k].size digits and a blank: taken by LENGTH (a byte-by-byte search for
                    // the token's end walked all 33 kB of every line); the digit test of the packing loop below rejects a blank or any
                    // other character inside it, and a line that is not shaped like that goes through the tokeniser for its message.
                    while (d.p < d.e && kWs.t[(unsigned char)*d.p]) d.p++;
                    const int want = (int)pops[k].size;
                    if (d.e - d.p >= want && (d.p + want == d.e || kWs.t[(unsigned char)d.p[want]])) { g = d.p; n = want; d.p += want; }
                    else if (!d.next(g, n) || n != want)
                        return give_up(std::string("panel line of ") + (strings.data() + snps[i].rsid) + ": population " + pops[k].name + " does not have " + std::to_string(pops[k].size) + " genotypes");
                    int32_t c = 0;
                    uint8_t* dst = row + pops[k].byte_off;
                    int q = 0;
                    bool bad = false;
                    // eight genotype digits at a time: one 64-bit word, '0' taken off every byte, the eight 2-bit codes folded
                    // into 16 bits in three shift-or steps (byte j of the word is digit q + j: little endian)
                    for (; q + 8 <= n; q += 8) {
                        uint64_t x;
                        memcpy(&x, g + q, 8);
                        x -= 0x3030303030303030ull;
                        if (x & 0xFCFCFCFCFCFCFCFCull) { bad = true; break; }       // a byte outside '0'..'3' (a borrow shows up here too)
                        c += (int32_t)((x * 0x0101010101010101ull) >> 56);
                        x = (x | (x >> 6)) & 0x000F000F000F000Full;
                        x = (x | (x >> 12)) & 0x000000FF000000FFull;
                        x = (x | (x >> 24)) & 0xFFFFull;
                        dst[q >> 2] = (uint8_t)x;
                        dst[(q >> 2) + 1] = (uint8_t)(x >> 8);
                    }
                    for (; !bad && q < n; q++) {
                        const unsigned code = (unsigned)(g[q] - '0');
                        if (code > 3) { bad = true; break; }
                        c += (int32_t)code;
                        dst[q >> 2] |= (uint8_t)(code << (2 * (q & 3)));
                    }
                    if (bad) {
                        // (a white-space character inside the string: the token is shorter than the population)
                        for (int t = 0; t < n; t++)
                            if (kWs.t[(unsigned char)g[t]])
                                return give_up(std::string("panel line of ") + (strings.data() + snps[i].rsid) + ": population " + pops[k].name + " does not have " + std::to_string(pops[k].size) + " genotypes");
                        return give_up(std::string("panel line of ") + (strings.data() + snps[i].rsid) + " has a genotype outside 0..3");
                    }
                    cnt[i * (size_t)P + k] = c;
                }
                for (int k = 0; k < P; k++) {
                    const char* g; int n;
                    double v = 0.0;                           // a missing column reads as 0, like the text feeder
                    if (d.next(g, n)) { char tmp[64]; const int m = std::min(n, 63); memcpy(tmp, g, (size_t)m); tmp[m] = 0; v = strtod(tmp, nullptr); }
                    af[i * (size_t)P + k] = v;
                }
            }
            if (!put_at(h.off_geno + c0 * (size_t)row_bytes, rows.data(), (c1 - c0) * (size_t)row_bytes)) return give_up("short write to '" + out_path + "'");
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(work, t);
        work(0);
        for (std::thread& x : th) x.join();
    }
    if (!first_err.empty()) return fail_out(first_err);

    bool ok = put_at(0, &h, sizeof(h)) && put_at(h.off_pops, pops.data(), (size_t)P * sizeof(PkPop)) &&
              put_at(h.off_snps, snps.data(), S * sizeof(PkSnp)) && put_at(h.off_strings, strings.data(), strings.size()) &&
              put_at(h.off_af, af.data(), af.size() * sizeof(double)) && put_at(h.off_cnt, cnt.data(), cnt.size() * sizeof(int32_t));
    ok = (::close(fd) == 0) && ok;
    if (!ok) { err = "short write to '" + out_path + "'"; remove(out_path.c_str()); return -1; }
    return (int64_t)S;
}

}  // namespace gauss_host
