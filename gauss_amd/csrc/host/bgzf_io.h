// BGZF (blocked gzip) reader/writer over zlib -- host side of the panel feeder.
//
// The reference reads its panel through samtools-era src/bgzf.c (bgzf_open/bgzf_seek/bgzf_getc,
// util.cpp:488-507).  In the Rcpp integration that file stays in place; this is the standalone
// equivalent for libgauss_host.so, written from the BGZF format description (SAM spec 4.1):
// a BGZF file is a series of gzip members, each <= 64 KiB, carrying an extra field "BC" with the
// member's total size minus one; a virtual offset is (member file offset << 16) | offset inside
// the inflated member (bgzf.c:702-727 uses the same convention).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace gauss_host {

class BgzfReader {
public:
    BgzfReader() = default;
    ~BgzfReader();
    BgzfReader(const BgzfReader&) = delete;
    BgzfReader& operator=(const BgzfReader&) = delete;

    bool open(const std::string& path);
    void close();
    bool is_open() const { return fp_ != nullptr || map_ != nullptr; }
    // Position at a virtual offset.  Offsets that land beyond the file read as EOF (the reference
    // relies on this for SNPs whose fpos is -1, gauss.cpp:561-597).
    void seek(int64_t voffset);
    // Append characters up to (not including) '\n' or EOF.  Returns '\n' (10) or -1 at EOF,
    // -2 on a codec error -- the return convention of BgzfGetLine (util.cpp:488-507).
    int getline(std::string& line);
    int64_t tell() const { return (block_address_ << 16) | (int64_t)block_offset_; }

private:
    int read_block();   // 0 ok (block_length_ may be 0 at EOF), -1 error
    FILE* fp_ = nullptr;
    // the file is mapped when possible: a block is then parsed straight out of the page cache (no lseek / read
    // system calls per block -- they dominate, and serialise threads, in sandboxed containers); fp_ is the fallback
    const unsigned char* map_ = nullptr;
    size_t map_size_ = 0;
    int64_t block_address_ = 0;
    int block_offset_ = 0;
    int block_length_ = 0;
    int64_t next_address_ = 0;
    bool loaded_ = false;
    std::vector<unsigned char> comp_, data_;
    void* fast_ = nullptr;      // libdeflate decompressor, when that library is present at run time
};

class BgzfWriter {
public:
    ~BgzfWriter();
    bool open(const std::string& path, int level = 6);
    int64_t tell() const { return (block_address_ << 16) | (int64_t)buf_.size(); }
    bool write(const void* data, size_t n);
    bool close();

private:
    bool flush_block();
    FILE* fp_ = nullptr;
    int level_ = 6;
    int64_t block_address_ = 0;
    std::vector<unsigned char> buf_;
};

}  // namespace gauss_host
