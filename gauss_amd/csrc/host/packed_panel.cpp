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
struct Fields {
    const char* p; const char* e;
    bool next(const char*& b, int& n)
    {
        while (p < e && isspace((unsigned char)*p)) p++;
        if (p >= e) return false;
        b = p;
        while (p < e && !isspace((unsigned char)*p)) p++;
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

    std::vector<PkSnp> snps;
    std::vector<char> strings;
    std::vector<double> af;
    std::vector<int32_t> cnt;
    const std::string geno_tmp = out_path + ".geno.tmp";
    FILE* gf = fopen(geno_tmp.c_str(), "wb");
    if (!gf) { err = "can't write '" + geno_tmp + "'"; return -1; }
    auto add_str = [&](const char* b, int n) { const uint32_t o = (uint32_t)strings.size(); strings.insert(strings.end(), b, b + n); strings.push_back(0); return o; };
    bool sorted = true;

    // The index is read sequentially in blocks of BLK lines; the data lines of a block (one ~33 kB text line per SNP,
    // found by its BGZF virtual offset) are inflated, parsed and 2-bit packed by a pool of threads, each with its own
    // reader, straight into the block's slots -- the conversion is a one-off per panel, but a genome-wide 33KG panel has
    // millions of lines (0.29 ms per line on one thread).
    const int BLK = 16384;
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (int)std::max(1u, std::min(16u, hw ? hw : 4u));
    std::vector<BgzfReader> readers((size_t)nt);
    for (BgzfReader& r : readers)
        if (!r.open(data_path)) { err = "ERROR: can't open reference data file '" + data_path + "'"; fclose(gf); remove(geno_tmp.c_str()); return -1; }
    struct Pending { long long fpos; std::string rsid; };
    std::vector<Pending> pend;
    std::vector<uint8_t> rows;
    std::vector<double> baf;
    std::vector<int32_t> bcnt;
    std::string line;
    bool eof = false;
    while (!eof) {
        pend.clear();
        while ((int)pend.size() < BLK) {
            const int last = idx.getline(line);
            if (last == -2) { err = "Error: can't read reference index file '" + index_path + "'"; fclose(gf); remove(geno_tmp.c_str()); return -1; }
            if (last == -1 && line.empty()) { eof = true; break; }
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
                pend.push_back(Pending{strtoll(std::string(b_fp, n_fp).c_str(), nullptr, 10), std::string(b_rs, n_rs)});
            }
            line.clear();
            if (last == -1) { eof = true; break; }
        }
        const int nb = (int)pend.size();
        if (nb == 0) break;
        rows.assign((size_t)nb * row_bytes, 0);
        baf.assign((size_t)nb * P, 0.0);
        bcnt.assign((size_t)nb * P, 0);
        std::atomic<int> next{0};
        std::mutex emu;
        std::string first_err;
        auto work = [&](int tid) {
            BgzfReader& dat = readers[tid];
            std::string dline;
            // runs of consecutive SNPs per grab: neighbouring lines share BGZF blocks (a 33 kB line is half a block),
            // interleaving single lines over the threads would make every thread inflate every block
            const int RUN = 128;
            for (int c0 = next.fetch_add(RUN); c0 < nb; c0 = next.fetch_add(RUN))
            for (int i = c0; i < std::min(nb, c0 + RUN); i++) {
                dat.seek(pend[i].fpos);
                dline.clear();
                dat.getline(dline);
                Fields d{dline.data(), dline.data() + dline.size()};
                uint8_t* row = rows.data() + (size_t)i * row_bytes;
                for (int k = 0; k < P; k++) {
                    const char* g; int n;
                    if (!d.next(g, n) || n != (int)pops[k].size) {
                        std::lock_guard<std::mutex> lock(emu);
                        if (first_err.empty()) first_err = "panel line of " + pend[i].rsid + ": population " + pops[k].name + " does not have " + std::to_string(pops[k].size) + " genotypes";
                        return;
                    }
                    int32_t c = 0;
                    uint8_t* dst = row + pops[k].byte_off;
                    for (int q = 0; q < n; q++) {
                        const unsigned code = (unsigned)(g[q] - '0');
                        if (code > 3) {
                            std::lock_guard<std::mutex> lock(emu);
                            if (first_err.empty()) first_err = "panel line of " + pend[i].rsid + " has a genotype outside 0..3";
                            return;
                        }
                        c += (int32_t)code;
                        dst[q >> 2] |= (uint8_t)(code << (2 * (q & 3)));
                    }
                    bcnt[(size_t)i * P + k] = c;
                }
                for (int k = 0; k < P; k++) {
                    const char* g; int n;
                    double v = 0.0;                           // a missing column reads as 0, like the text feeder
                    if (d.next(g, n)) v = strtod(std::string(g, n).c_str(), nullptr);
                    baf[(size_t)i * P + k] = v;
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < std::min(nt, nb); t++) th.emplace_back(work, t);
        work(0);
        for (std::thread& x : th) x.join();
        if (!first_err.empty()) { err = first_err; fclose(gf); remove(geno_tmp.c_str()); return -1; }
        af.insert(af.end(), baf.begin(), baf.end());
        cnt.insert(cnt.end(), bcnt.begin(), bcnt.end());
        if (fwrite(rows.data(), 1, rows.size(), gf) != rows.size()) { err = "short write to '" + geno_tmp + "'"; fclose(gf); remove(geno_tmp.c_str()); return -1; }
    }
    fclose(gf);

    PkHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, kMagic, 8);
    h.version = 1; h.n_pop = (uint32_t)P; h.n_snp = snps.size(); h.row_bytes = row_bytes; h.sorted = sorted ? 1u : 0u;
    size_t off = sizeof(PkHeader);
    h.off_pops = off; off = align_up(off + (size_t)P * sizeof(PkPop), 64);
    h.off_snps = off; off = align_up(off + snps.size() * sizeof(PkSnp), 64);
    h.off_strings = off; off = align_up(off + strings.size(), 64);
    h.off_af = off; off = align_up(off + af.size() * sizeof(double), 64);
    h.off_cnt = off; off = align_up(off + cnt.size() * sizeof(int32_t), 4096);
    h.off_geno = off; off += snps.size() * (size_t)row_bytes;
    h.file_bytes = off;

    FILE* out = fopen(out_path.c_str(), "wb");
    if (!out) { err = "can't write '" + out_path + "'"; remove(geno_tmp.c_str()); return -1; }
    auto put_at = [&](size_t at, const void* p, size_t n) { fseek(out, (long)at, SEEK_SET); return n == 0 || fwrite(p, 1, n, out) == n; };
    bool ok = put_at(0, &h, sizeof(h)) && put_at(h.off_pops, pops.data(), (size_t)P * sizeof(PkPop)) &&
              put_at(h.off_snps, snps.data(), snps.size() * sizeof(PkSnp)) && put_at(h.off_strings, strings.data(), strings.size()) &&
              put_at(h.off_af, af.data(), af.size() * sizeof(double)) && put_at(h.off_cnt, cnt.data(), cnt.size() * sizeof(int32_t));
    if (ok) {
        fseek(out, (long)h.off_geno, SEEK_SET);
        FILE* in = fopen(geno_tmp.c_str(), "rb");
        std::vector<uint8_t> buf(1 << 20);
        size_t n;
        while (in && (n = fread(buf.data(), 1, buf.size(), in)) > 0) ok = ok && fwrite(buf.data(), 1, n, out) == n;
        if (in) fclose(in); else ok = false;
        if (snps.empty()) { const char z = 0; ok = ok && put_at(h.file_bytes ? h.file_bytes - 1 : 0, &z, h.file_bytes ? 1 : 0); }
    }
    ok = (fclose(out) == 0) && ok;
    remove(geno_tmp.c_str());
    if (!ok) { err = "short write to '" + out_path + "'"; return -1; }
    return (int64_t)snps.size();
}

}  // namespace gauss_host
