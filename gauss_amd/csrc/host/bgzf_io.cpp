#include "bgzf_io.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <zlib.h>

#include <cstdlib>
#include <cstring>

#include <dlfcn.h>

namespace gauss_host {

// Optional libdeflate binding (resolved once, by name; the three entry points have been ABI-stable since 1.0).
struct FastInflate {
    void* (*alloc)() = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*release)(void*) = nullptr;
    uint32_t (*crc32)(uint32_t, const void*, size_t) = nullptr;      // libdeflate_crc32: carry-less multiply, several GB/s (zlib 1.2's
                                                                      // table walk checks a 64 KB block in ~50 us -- as long as inflating it)
};
static const FastInflate& fast_inflate()
{
    static const FastInflate fi = [] {
        FastInflate f;
        if (getenv("GAUSS_NO_LIBDEFLATE")) return f;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return f;
        f.alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        f.decompress = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
        f.release = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        f.crc32 = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
        if (!f.alloc || !f.decompress || !f.release) f = FastInflate();
        return f;
    }();
    return fi;
}


static const int kMaxBlock = 64 * 1024;
static const int kWriteBlock = 0xff00;    // input bytes per block, leaves room for the gzip wrapper

BgzfReader::~BgzfReader() { close(); }

bool BgzfReader::open(const std::string& path)
{
    close();
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) == 0 && st.st_size > 0 && !getenv("GAUSS_BGZF_NO_MMAP")) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { map_ = (const unsigned char*)m; map_size_ = (size_t)st.st_size; }
        }
        ::close(fd);
    }
    if (!map_) {
        fp_ = fopen(path.c_str(), "rb");
        if (!fp_) return false;
    }
    comp_.resize(kMaxBlock);
    data_.resize(kMaxBlock);
    block_address_ = 0; block_offset_ = 0; block_length_ = 0; next_address_ = 0; loaded_ = false;
    return true;
}

void BgzfReader::close()
{
    if (fp_) fclose(fp_);
    fp_ = nullptr;
    if (map_) munmap(const_cast<unsigned char*>(map_), map_size_);
    map_ = nullptr; map_size_ = 0;
    if (fast_) { fast_inflate().release(fast_); fast_ = nullptr; }
}

void BgzfReader::seek(int64_t voffset)
{
    const int64_t addr = (voffset >> 16) & 0xFFFFFFFFFFFFLL;
    const int off = (int)(voffset & 0xFFFF);
    if (!(loaded_ && addr == block_address_)) {      // same block: no re-inflate
        block_address_ = addr;
        loaded_ = false;
    }
    block_offset_ = off;
}

int BgzfReader::read_block()
{
    block_length_ = 0;
    next_address_ = block_address_;
    const unsigned char* cdata = nullptr;      // the compressed payload followed by the 8-byte trailer
    int xlen = 0, total = 0, clen = 0;
    auto parse_extra = [&](const unsigned char* extra) {
        int bsize = -1;
        for (int p = 0; p + 4 <= xlen;) {
            const int slen = extra[p + 2] | (extra[p + 3] << 8);
            if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= xlen) bsize = extra[p + 4] | (extra[p + 5] << 8);
            p += 4 + slen;
        }
        return bsize;
    };
    if (map_) {
        if (block_address_ < 0 || (uint64_t)block_address_ >= map_size_) return 0;    // unreachable offset reads as EOF
        const unsigned char* hdr = map_ + block_address_;
        const size_t left = map_size_ - (size_t)block_address_;
        if (left < 12 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) return -1;
        xlen = hdr[10] | (hdr[11] << 8);
        if (xlen < 6 || xlen > 4096 || left < (size_t)12 + xlen) return -1;
        const int bsize = parse_extra(hdr + 12);
        if (bsize < 0) return -1;
        total = bsize + 1;
        clen = total - 12 - xlen - 8;
        if (clen < 0 || clen > kMaxBlock || left < (size_t)total) return -1;
        cdata = hdr + 12 + xlen;
    } else {
        if (fseeko(fp_, (off_t)block_address_, SEEK_SET) != 0) return 0;   // unreachable offset reads as EOF
        unsigned char hdr[12];
        size_t n = fread(hdr, 1, 12, fp_);
        if (n == 0) return 0;               // EOF
        if (n != 12 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) return -1;
        xlen = hdr[10] | (hdr[11] << 8);
        if (xlen < 6 || xlen > 4096) return -1;
        unsigned char extra[4096];
        if (fread(extra, 1, xlen, fp_) != (size_t)xlen) return -1;
        const int bsize = parse_extra(extra);
        if (bsize < 0) return -1;
        total = bsize + 1;
        clen = total - 12 - xlen - 8;
        if (clen < 0 || clen > kMaxBlock) return -1;
        if (fread(comp_.data(), 1, (size_t)clen + 8, fp_) != (size_t)clen + 8) return -1;
        cdata = comp_.data();
    }
    const unsigned char* ft = cdata + clen;
    const uint32_t crc = ft[0] | (ft[1] << 8) | (ft[2] << 16) | ((uint32_t)ft[3] << 24);
    const uint32_t isize = ft[4] | (ft[5] << 8) | (ft[6] << 16) | ((uint32_t)ft[7] << 24);
    // Raw deflate payload.  libdeflate (whole-buffer decoder, ~2-3x zlib) when its runtime library is present --
    // the image ships libdeflate.so.0 without headers, so it is bound by name; zlib otherwise and on any doubt.
    bool done = false;
    const FastInflate& fi = fast_inflate();
    if (fi.decompress && isize <= (uint32_t)kMaxBlock) {
        if (!fast_) fast_ = fi.alloc();
        size_t got = 0;
        if (fast_ && fi.decompress(fast_, cdata, (size_t)clen, data_.data(), (size_t)kMaxBlock, &got) == 0 && got == isize) {
            block_length_ = (int)got;
            done = true;
        }
    }
    if (!done) {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        zs.next_in = const_cast<unsigned char*>(cdata);
        zs.avail_in = (uInt)clen;
        zs.next_out = data_.data();
        zs.avail_out = kMaxBlock;
        if (inflateInit2(&zs, -15) != Z_OK) return -1;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END) return -1;
        block_length_ = (int)zs.total_out;
    }
    const uint32_t have = fi.crc32 ? fi.crc32(0, data_.data(), (size_t)block_length_)
                                   : (uint32_t)::crc32(::crc32(0L, Z_NULL, 0), data_.data(), (uInt)block_length_);
    if (have != crc) return -1;
    next_address_ = block_address_ + total;
    return 0;
}

int BgzfReader::getline(std::string& line)
{
    line.clear();
    if (!fp_ && !map_) return -2;
    for (;;) {
        if (!loaded_) {
            if (read_block() != 0) return -2;
            loaded_ = true;
            if (block_length_ == 0) return -1;       // end of file (or the empty EOF-marker block)
        }
        if (block_offset_ >= block_length_) {        // block exhausted: continue in the next one
            if (block_length_ == 0) return -1;
            block_address_ = next_address_;
            block_offset_ = 0;
            loaded_ = false;
            continue;
        }
        const unsigned char* p = data_.data() + block_offset_;
        const int avail = block_length_ - block_offset_;
        const void* nl = memchr(p, '\n', (size_t)avail);
        if (nl) {
            const int len = (int)((const unsigned char*)nl - p);
            line.append((const char*)p, (size_t)len);
            block_offset_ += len + 1;
            return 10;
        }
        line.append((const char*)p, (size_t)avail);
        block_offset_ = block_length_;
    }
}

// ------------------------------------------------------------------------------------------
BgzfWriter::~BgzfWriter() { close(); }

bool BgzfWriter::open(const std::string& path, int level)
{
    fp_ = fopen(path.c_str(), "wb");
    level_ = level;
    block_address_ = 0;
    buf_.clear();
    buf_.reserve(kWriteBlock);
    return fp_ != nullptr;
}

bool BgzfWriter::flush_block()
{
    std::vector<unsigned char> out(kMaxBlock + 64);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level_, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = buf_.data();
    zs.avail_in = (uInt)buf_.size();
    zs.next_out = out.data() + 18;
    zs.avail_out = (uInt)(out.size() - 18 - 8);
    const int rc = deflate(&zs, Z_FINISH);
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return false;
    const int clen = (int)zs.total_out;
    const int total = clen + 18 + 8;
    if (total > kMaxBlock) return false;
    unsigned char* h = out.data();
    const unsigned char hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0,
                                   (unsigned char)((total - 1) & 0xff), (unsigned char)((total - 1) >> 8)};
    memcpy(h, hdr, 18);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf_.data(), (uInt)buf_.size());
    const uint32_t isz = (uint32_t)buf_.size();
    unsigned char* f = h + 18 + clen;
    for (int i = 0; i < 4; i++) { f[i] = (crc >> (8 * i)) & 0xff; f[4 + i] = (isz >> (8 * i)) & 0xff; }
    if (fwrite(h, 1, (size_t)total, fp_) != (size_t)total) return false;
    block_address_ += total;
    buf_.clear();
    return true;
}

bool BgzfWriter::write(const void* data, size_t n)
{
    const unsigned char* p = (const unsigned char*)data;
    while (n > 0) {
        const size_t room = (size_t)kWriteBlock - buf_.size();
        const size_t take = n < room ? n : room;
        buf_.insert(buf_.end(), p, p + take);
        p += take; n -= take;
        if (buf_.size() == (size_t)kWriteBlock && !flush_block()) return false;
    }
    return true;
}

bool BgzfWriter::close()
{
    if (!fp_) return true;
    bool ok = true;
    if (!buf_.empty()) ok = flush_block();
    ok = ok && flush_block();            // empty block = BGZF end-of-file marker
    fclose(fp_);
    fp_ = nullptr;
    return ok;
}

}  // namespace gauss_host
