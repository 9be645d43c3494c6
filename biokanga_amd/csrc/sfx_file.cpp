// sfx_file.cpp - .sfx reader/writer (see sfx_file.h for the format and reference citations)
#include "sfx_file.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <cerrno>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <thread>

namespace bk {

namespace {
constexpr size_t kHdrSize = 1224;        // sizeof(tsSfxHeaderV3) with pack(4)
constexpr size_t kBlkHdrSize = 20;       // sizeof(tsSfxBlock) - 1
constexpr size_t kEntrySize = 111;       // sizeof(tsSfxEntry) with pack(1)
// header field offsets, pack(4)
constexpr size_t oVersion = 4, oAttr = 8, oFileLen = 12, oEntOfs = 20, oEntSize = 28, oNumBlocks = 32,
                 oBlkSize = 36, oBlkOfs = 44, oDataset = 52, oDescr = 133, oTitle = 1157;

template <typename T> T rd(const uint8_t *p) { T v; memcpy(&v, p, sizeof(T)); return v; }
template <typename T> void wr(uint8_t *p, T v) { memcpy(p, &v, sizeof(T)); }

int fail(std::string *err, int rc, const std::string &msg)
{
    if (err) *err = msg;
    return rc;
}

// one stretch of the file from `nthreads` threads, each pwrite()-ing slices of 64 MB at their own offsets (a 15 GB image through one
// write() loop is six seconds of one core's copying into fresh page-cache pages)
bool pwrite_all(int fd, const uint8_t *p, uint64_t len, uint64_t at, int nthreads)
{
    const uint64_t slice = 64ull << 20, ns = (len + slice - 1) / slice;
    std::atomic<uint64_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&]() {
        for (uint64_t i; ok && (i = next.fetch_add(1)) < ns;) {
            uint64_t o = i * slice, n = std::min(slice, len - o);
            while (n) {
                const ssize_t w = ::pwrite(fd, p + o, (size_t)n, (off_t)(at + o));
                if (w < 0 && errno == EINTR) continue;
                if (w <= 0) {                   // (nothing written without an error counts as one: errno is then whatever an earlier call left)
                    ok = false;
                    break;
                }
                o += (uint64_t)w;
                n -= (uint64_t)w;
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads && (uint64_t)t < ns; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    return ok;
}

bool write_all(int fd, const uint8_t *p, uint64_t len)
{
    while (len) {
        size_t chunk = len > (1ull << 30) ? (1ull << 30) : (size_t)len;
        ssize_t n = ::write(fd, p, chunk);
        if (n <= 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += n;
        len -= (uint64_t)n;
    }
    return true;
}
}  // namespace

SfxFile::~SfxFile()
{
    if (map_base && map_base != MAP_FAILED) munmap(map_base, map_len);
}

// CUtility::GenHash16 (libbiokanga/Utility.cpp:17-37): seed 19937, per char h=(h^tolower(c))*3119,
// h^=h>>13, h&=0xffff; 0 is remapped to 19937
uint16_t gen_hash16(const char *name)
{
    if (!name || !name[0]) return 0;
    int h = 19937;
    for (const char *p = name; *p; ++p) {
        h = (h ^ (int)tolower((unsigned char)*p)) * 3119;
        h ^= (h >> 13);
        h &= 0xffff;
    }
    if (h == 0) h = 19937;
    return (uint16_t)h;
}

int sfx_open(const char *path, SfxFile &out, std::string *err)
{
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) return fail(err, -90, std::string("unable to open ") + path + ": " + strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < kHdrSize) {
        ::close(fd);
        return fail(err, -94, std::string(path) + ": not a biokanga suffix array file (too short)");
    }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) return fail(err, -95, std::string("mmap failed on ") + path);
    out.map_base = m;
    out.map_len = (size_t)st.st_size;
    out.path = path;
    const uint8_t *img = (const uint8_t *)m;

    // CSfxArrayV3::Disk2Hdr, SfxArrayV2.cpp:551-631
    if (tolower(img[0]) != 's' || tolower(img[1]) != 'f' || tolower(img[2]) != 'x' || img[3] < '3' || img[3] > '5')
        return fail(err, -94, std::string(path) + ": invalid magic signature - not a Biokanga generated suffix array file");
    out.version = rd<int32_t>(img + oVersion);
    if (out.version < 3 || out.version > 5)
        return fail(err, -86, std::string(path) + ": structure version incompatible");
    if (out.version == 3)   // 36-char name header layout of pre-2.75 releases
        return fail(err, -86, std::string(path) + ": V3 (36 char names) suffix files are not supported");
    out.attributes = rd<uint32_t>(img + oAttr);
    if (out.attributes & 0x3)
        return fail(err, -100, std::string(path) + ": bisulfite/colorspace indexes are outside the supported hot path");
    uint64_t ent_ofs = rd<uint64_t>(img + oEntOfs);
    uint32_t ent_size = rd<uint32_t>(img + oEntSize);
    uint32_t nblocks = rd<uint32_t>(img + oNumBlocks);
    uint64_t blk_size = rd<uint64_t>(img + oBlkSize);
    uint64_t blk_ofs = rd<uint64_t>(img + oBlkOfs);
    auto zstr = [&](size_t ofs, size_t max) {
        const char *s = (const char *)img + ofs;
        return std::string(s, strnlen(s, max));
    };
    out.dataset = zstr(oDataset, 81);
    out.description = zstr(oDescr, 1024);
    out.title = zstr(oTitle, 64);
    if (nblocks != 1 || ent_ofs == 0 || ent_size < 8)
        return fail(err, -1, std::string(path) + ": no suffix block / entries");
    // every header field comes from outside: compare against what is left of the mapping, never add first (sums can wrap)
    if (blk_ofs > out.map_len || out.map_len - blk_ofs < kBlkHdrSize || ent_ofs > out.map_len || out.map_len - ent_ofs < ent_size)
        return fail(err, -85, std::string(path) + ": truncated file");

    const uint8_t *blk = img + blk_ofs;
    out.block_id = rd<uint32_t>(blk);
    out.concat_len = rd<uint64_t>(blk + 8);
    out.el_size = rd<uint32_t>(blk + 16);
    if (out.el_size != 4 && out.el_size != 5) return fail(err, -1, "unsupported suffix element size");
    const uint64_t room = out.map_len - blk_ofs - kBlkHdrSize;
    if (out.concat_len == 0 || out.concat_len > room / (1 + (uint64_t)out.el_size))
        return fail(err, -85, std::string(path) + ": suffix block size mismatch");
    uint64_t need = kBlkHdrSize + out.concat_len + out.concat_len * out.el_size;
    if (blk_size && blk_size != need)
        return fail(err, -85, std::string(path) + ": suffix block size mismatch");
    out.seq = blk + kBlkHdrSize;
    out.sa = out.seq + out.concat_len;

    // Disk2Entries, SfxArrayV2.cpp:636-747
    const uint8_t *eb = img + ent_ofs;
    uint32_t n = rd<uint32_t>(eb);
    if ((uint64_t)8 + (uint64_t)n * kEntrySize > ent_size) return fail(err, -85, "entries block truncated");
    out.entries.resize(n);
    const uint8_t *e = eb + 8;
    out.tot_seq_len = 0;
    for (uint32_t i = 0; i < n; i++, e += kEntrySize) {
        SfxEntry &d = out.entries[i];
        d.entry_id = rd<uint32_t>(e);
        d.fblock_id = rd<uint32_t>(e + 4);
        memcpy(d.name, e + 8, 81);
        d.name[80] = 0;
        d.name_hash = rd<uint16_t>(e + 89);
        d.seq_len = rd<uint32_t>(e + 91);
        d.start_ofs = rd<uint64_t>(e + 95);
        d.end_ofs = rd<uint64_t>(e + 103);
        out.tot_seq_len += d.seq_len;
        // an entry is a span of the concatenated bases (the EOS after it included in concat_len)
        if (d.seq_len == 0 || d.start_ofs >= out.concat_len || d.end_ofs >= out.concat_len || d.end_ofs < d.start_ofs ||
            d.end_ofs - d.start_ofs + 1 != d.seq_len)
            return fail(err, -85, std::string(path) + ": sequence entry outside the suffix block");
    }
    return 0;
}

int sfx_write(const char *path, const std::string &dataset, const std::string &description,
              const std::string &title, const std::vector<SfxEntry> &entries, const uint8_t *seq,
              uint64_t concat_len, const uint8_t *sa, uint32_t el_size, std::string *err, int nthreads)
{
    if (entries.empty() || concat_len == 0 || (el_size != 4 && el_size != 5))
        return fail(err, -100, "sfx_write: nothing to write");
    int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0600);   // reference creates with S_IREAD|S_IWRITE
    if (fd < 0) return fail(err, -89, std::string("unable to create ") + path + ": " + strerror(errno));

    // header: InitHdr (SfxArrayV2.cpp:282-296) then the fields set by SfxBlock2Disk/Entries2Disk
    std::vector<uint8_t> hdr(kHdrSize, 0);
    hdr[0] = 's'; hdr[1] = 'f'; hdr[2] = 'x'; hdr[3] = '5';
    wr<int32_t>(&hdr[oVersion], 5);
    wr<uint32_t>(&hdr[oAttr], 0);
    uint64_t file_len = kHdrSize;
    uint64_t blk_ofs = file_len;
    uint64_t blk_size = kBlkHdrSize + concat_len + concat_len * el_size;
    file_len += blk_size;
    uint64_t ent_ofs = file_len;
    uint32_t ent_size = (uint32_t)(8 + kEntrySize * entries.size());
    file_len += ent_size;
    wr<uint64_t>(&hdr[oFileLen], file_len);
    wr<uint64_t>(&hdr[oEntOfs], ent_ofs);
    wr<uint32_t>(&hdr[oEntSize], ent_size);
    wr<uint32_t>(&hdr[oNumBlocks], 1);
    wr<uint64_t>(&hdr[oBlkSize], blk_size);
    wr<uint64_t>(&hdr[oBlkOfs], blk_ofs);
    strncpy((char *)&hdr[oDataset], dataset.c_str(), 80);
    strncpy((char *)&hdr[oDescr], description.c_str(), 1023);
    strncpy((char *)&hdr[oTitle], title.c_str(), 63);

    uint8_t bh[kBlkHdrSize];
    wr<uint32_t>(bh, 1);
    wr<uint32_t>(bh + 4, (uint32_t)entries.size());
    wr<uint64_t>(bh + 8, concat_len);
    wr<uint32_t>(bh + 16, el_size);

    std::vector<uint8_t> eb(ent_size, 0);
    wr<uint32_t>(&eb[0], (uint32_t)entries.size());
    wr<uint32_t>(&eb[4], (uint32_t)entries.size());
    uint8_t *e = &eb[8];
    for (const SfxEntry &s : entries) {
        wr<uint32_t>(e, s.entry_id);
        wr<uint32_t>(e + 4, s.fblock_id);
        memcpy(e + 8, s.name, 81);
        wr<uint16_t>(e + 89, s.name_hash);
        wr<uint32_t>(e + 91, s.seq_len);
        wr<uint64_t>(e + 95, s.start_ofs);
        wr<uint64_t>(e + 103, s.end_ofs);
        e += kEntrySize;
    }
    bool ok;
    if (nthreads > 1 && concat_len >= (16ull << 20)) {
        // the small parts at their places, the bases and the suffix array by all threads
        const uint64_t seq_at = blk_ofs + kBlkHdrSize, sa_at = seq_at + concat_len;
        ok = ::pwrite(fd, hdr.data(), hdr.size(), 0) == (ssize_t)hdr.size() && ::pwrite(fd, bh, sizeof(bh), (off_t)blk_ofs) == (ssize_t)sizeof(bh) &&
             ::pwrite(fd, eb.data(), eb.size(), (off_t)ent_ofs) == (ssize_t)eb.size() &&
             pwrite_all(fd, seq, concat_len, seq_at, nthreads) && pwrite_all(fd, sa, concat_len * el_size, sa_at, nthreads);
    } else
        ok = write_all(fd, hdr.data(), hdr.size()) && write_all(fd, bh, sizeof(bh)) &&
             write_all(fd, seq, concat_len) && write_all(fd, sa, concat_len * el_size) &&
             write_all(fd, eb.data(), eb.size());
    if (fsync(fd) != 0) ok = false;
    ::close(fd);
    if (!ok) {
        (void)::unlink(path);                      // (no partly written index under the final name)
        return fail(err, -85, std::string("write failed on ") + path);
    }
    return 0;
}

}  // namespace bk
