// Packed reference panel ("GAUSSPK1"): the read-once, mmap-able replacement for the BGZF text panel
// (SURVEY.md section 8f row N3).  The reference keeps two BGZF text files -- an index with one line per
// SNP (rsid chr bp a1 a2 af1ref fpos; scanned genome-wide on every call, gauss.cpp:322-392) and a data
// file with one ~33 kB text line per SNP (P genotype strings + P allele frequencies; sought and inflated
// twice per SNP, gauss.cpp:546-566, 755-763).  One packed file carries the same information:
//
//   header   128 bytes, little endian (struct PkHeader)
//   pops     n_pop x PkPop      name, super population, size, byte offset of the block in a row
//   snps     n_snp x PkSnp      chr, bp, string offsets (rsid, a1, a2); file order = index order
//   strings  NUL-terminated
//   af       n_snp x n_pop f64  the panel's per-population allele frequencies, as parsed by strtod
//   cnt      n_snp x n_pop i32  per-population allele counts (sum of the genotype codes)
//   geno     n_snp x row_bytes  2-bit genotypes in the GAUSS_GENO_2BIT layout of include/gauss_hip.h:
//                               population blocks 16-byte aligned, zero padded to 64 samples
//
// A window's rows go to the GPU as they are (a quarter of the bytes of the text strings) or the whole
// geno section is uploaded once (gauss_store_upload) and windows name rows by index.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace gauss_host {

struct PkHeader {
    char magic[8];             // "GAUSSPK1"
    uint32_t version;          // 1
    uint32_t n_pop;
    uint64_t n_snp;
    uint64_t row_bytes;        // multiple of 16
    uint64_t off_pops, off_snps, off_strings, off_af, off_cnt, off_geno, file_bytes;
    uint32_t sorted;           // 1: (chr, bp) non-decreasing in file order -> windows found by binary search
    uint32_t pad_[9];
};
static_assert(sizeof(PkHeader) == 128, "PkHeader must be 128 bytes");

struct PkPop { char name[24]; char super[24]; uint32_t size; uint32_t byte_off; };
struct PkSnp { int32_t chr; uint32_t rsid, a1, a2; int64_t bp; };
static_assert(sizeof(PkPop) == 56 && sizeof(PkSnp) == 24, "packed record sizes");

class PackedPanel {
public:
    PackedPanel() = default;
    ~PackedPanel();
    PackedPanel(const PackedPanel&) = delete;
    PackedPanel& operator=(const PackedPanel&) = delete;

    static bool is_packed(const std::string& path);     // magic check, false on any error
    bool open(const std::string& path, std::string& err);
    void close();

    const PkHeader& header() const { return *hdr_; }
    int n_pop() const { return (int)hdr_->n_pop; }
    int64_t n_snp() const { return (int64_t)hdr_->n_snp; }
    int64_t row_bytes() const { return (int64_t)hdr_->row_bytes; }
    const PkPop& pop(int k) const { return pops_[k]; }
    const PkSnp& snp(int64_t i) const { return snps_[i]; }
    const char* str(uint32_t off) const { return strings_ + off; }
    const double* af(int64_t i) const { return af_ + (size_t)i * hdr_->n_pop; }
    const int32_t* cnt(int64_t i) const { return cnt_ + (size_t)i * hdr_->n_pop; }
    const uint8_t* row(int64_t i) const { return geno_ + (size_t)i * hdr_->row_bytes; }
    const uint8_t* geno() const { return geno_; }
    int fd() const { return fd_; }                                  // the open file (pread source of the resident upload)
    int64_t geno_file_offset() const { return (int64_t)(geno_ - base_); }
    // first row with (chr, bp) >= the key, for sorted panels
    int64_t lower_bound(int chr, int64_t bp) const;

private:
    int fd_ = -1;
    const uint8_t* base_ = nullptr;
    size_t bytes_ = 0;
    const PkHeader* hdr_ = nullptr;
    const PkPop* pops_ = nullptr;
    const PkSnp* snps_ = nullptr;
    const char* strings_ = nullptr;
    const double* af_ = nullptr;
    const int32_t* cnt_ = nullptr;
    const uint8_t* geno_ = nullptr;
};

// BGZF text panel (index + data + population description) -> packed panel.  Returns the number of
// SNPs written or -1 (err filled).  Genotype characters other than '0'..'3' cannot be packed.
int64_t pack_panel(const std::string& index_path, const std::string& data_path, const std::string& desc_path,
                   const std::string& out_path, std::string& err);

}  // namespace gauss_host
