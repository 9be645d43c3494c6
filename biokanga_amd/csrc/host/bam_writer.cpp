// bam_writer.cpp - see bam_writer.h
#include "bam_writer.h"

#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>

namespace bk {

namespace {
constexpr uint64_t kBlock = 0xff00;          // BGZF_BLOCK_SIZE, bgzf.h:46
constexpr int kNumBins = 37450;              // cNumSAIBins, SAMfile.h:41

// bgzf_compress (bgzf.cpp): header, raw deflate, crc32, isize
bool bgzf_block(const uint8_t *src, size_t n, int level, std::vector<uint8_t> &out)
{
    out.resize(0x10000 + 64);
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(out.data(), hdr, 16);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    zs.next_in = (Bytef *)src;
    zs.avail_in = (uInt)n;
    zs.next_out = out.data() + 18;
    zs.avail_out = (uInt)(out.size() - 18 - 8);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); return false; }
    if (deflateEnd(&zs) != Z_OK) return false;
    size_t total = 18 + zs.total_out + 8;
    uint16_t bsize = (uint16_t)(total - 1);
    memcpy(out.data() + 16, &bsize, 2);
    uint32_t crc = (uint32_t)crc32(crc32(0L, nullptr, 0), src, (uInt)n), isize = (uint32_t)n;
    memcpy(out.data() + 18 + zs.total_out, &crc, 4);
    memcpy(out.data() + 18 + zs.total_out + 4, &isize, 4);
    out.resize(total);
    return true;
}

bool write_all(int fd, const void *p, size_t n)
{
    const uint8_t *q = (const uint8_t *)p;
    while (n) {
        ssize_t w = ::write(fd, q, n);
        if (w <= 0) return false;
        q += w;
        n -= (size_t)w;
    }
    return true;
}

struct Chunk { uint32_t start, end; uint64_t start_va, end_va; };
}  // namespace

int bam_reg2bin(int beg, int end)
{
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (beg >> 26);
    return 0;
}

// CSIreg2bin of the CSI specification as the reference has it (SAMfile.cpp:2059-2070); end exclusive
static int csi_reg2bin(int64_t beg, int64_t end, int min_shift, int depth)
{
    int l, s = min_shift, t = ((1 << depth * 3) - 1) / 7;
    for (--end, l = depth; l > 0; --l, s += 3, t -= 1 << l * 3)
        if (beg >> s == end >> s) return t + (int)(beg >> s);
    return 0;
}

int write_bam_and_bai(const std::string &path, const std::vector<uint8_t> &stream, const std::vector<BamAligned> &aligned,
                      uint64_t flush_at, uint32_t n_refs, uint64_t max_ref_len, int nthreads, std::string *err)
{
    // the reference switches from BAI to CSI when a header sequence reaches the 512 Mbp a BAI can address (CSAMfile::StartAlignments,
    // SAMfile.cpp:1602-1607); R-tree depth from the longest sequence, never below the BAI's 5 (CSIDepth :2089, :1666-1668)
    const bool csi = max_ref_len >= 0x20000000ULL;
    const int min_shift = 14;
    int depth = 5;
    if (csi) {
        int lv = 0;
        for (int64_t sz = (int64_t)1 << min_shift; (int64_t)max_ref_len > sz; ++lv, sz <<= 3) {}
        depth = lv < 5 ? 5 : lv;
    }
    const uint64_t max_idx_len = csi ? 0x7fffffffULL : 0x20000000ULL;          // cMaxCSIRefSeqLen / cMaxSAIRefSeqLen
    for (const BamAligned &a : aligned)
        if ((uint64_t)(uint32_t)a.end >= max_idx_len) {
            if (err) *err = "alignment ending at " + std::to_string((uint32_t)a.end) + " is beyond what a " + (csi ? "CSI" : "BAI") + " index of this build covers";
            return -100;
        }
    const uint64_t total = stream.size();
    // block table: [0, flush_at) and [flush_at, total) are each cut every kBlock bytes
    std::vector<uint64_t> beg;
    uint64_t n_first = 0;
    for (uint64_t u = 0; u < flush_at; u += kBlock) beg.push_back(u);
    n_first = beg.size();
    for (uint64_t u = flush_at; u < total; u += kBlock) beg.push_back(u);
    const size_t nb = beg.size();
    auto block_end = [&](size_t b) { return b + 1 < nb ? (b + 1 == n_first ? flush_at : beg[b + 1]) : total; };
    std::vector<std::vector<uint8_t>> comp(nb);
    std::vector<uint8_t> ok(nb, 1);
    if (nthreads < 1) nthreads = 1;
    {
        std::vector<std::thread> th;
        auto work = [&](int w) {
            for (size_t b = (size_t)w; b < nb; b += (size_t)nthreads)
                ok[b] = bgzf_block(stream.data() + beg[b], (size_t)(block_end(b) - beg[b]), 6, comp[b]) ? 1 : 0;
        };
        for (int w = 1; w < nthreads; w++) th.emplace_back(work, w);
        work(0);
        for (auto &t : th) t.join();
    }
    for (size_t b = 0; b < nb; b++)
        if (!ok[b]) { if (err) *err = "BGZF compression failed"; return -100; }
    std::vector<uint64_t> caddr(nb + 1, 0);
    for (size_t b = 0; b < nb; b++) caddr[b + 1] = caddr[b] + comp[b].size();

    int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) { if (err) *err = "unable to create '" + path + "'"; return -91; }
    for (size_t b = 0; b < nb; b++)
        if (!write_all(fd, comp[b].data(), comp[b].size())) { ::close(fd); if (err) *err = "write failed"; return -96; }
    {   // bgzf_close: an empty block at the default compression level
        std::vector<uint8_t> eof;
        if (!bgzf_block(nullptr, 0, Z_DEFAULT_COMPRESSION, eof) || !write_all(fd, eof.data(), eof.size())) { ::close(fd); return -96; }
    }
    fsync(fd);
    ::close(fd);

    // bgzf_tell at stream offset u of an aligned record boundary (u <= flush_at)
    auto va = [&](uint64_t u) -> uint64_t {
        if (u == flush_at) return caddr[n_first] << 16;                  // right after the flush: next block, offset 0
        return (caddr[u / kBlock] << 16) | (u % kBlock);
    };

    // ---- BAI / CSI (CSAMfile::AddAlignment / AddChunk / UpdateSAIIndex) ----
    std::vector<uint8_t> bai;
    auto put32 = [&](uint32_t v) { bai.insert(bai.end(), (uint8_t *)&v, (uint8_t *)&v + 4); };
    auto put64 = [&](uint64_t v) { bai.insert(bai.end(), (uint8_t *)&v, (uint8_t *)&v + 8); };
    if (csi) {
        bai.insert(bai.end(), {'C', 'S', 'I', 1});
        put32((uint32_t)min_shift);
        put32((uint32_t)depth);
        put32(0);                            // l_aux
    } else
        bai.insert(bai.end(), {'B', 'A', 'I', 1});
    put32(n_refs);
    std::vector<std::vector<Chunk>> bins(csi ? (size_t)(((1ULL << (depth + 1) * 3) - 1) / 7) : (size_t)kNumBins);
    std::vector<int> used;                   // bins owning chunks for the current reference
    std::vector<uint64_t> lin;
    uint32_t n_lin = 0;
    auto flush_ref = [&]() {                 // UpdateSAIIndex
        put32((uint32_t)used.size());
        if (!used.empty()) {
            std::vector<int> order(used);
            std::sort(order.begin(), order.end());
            for (int b : order) {
                put32((uint32_t)b);
                if (csi) {                   // loffset: the lowest virtual address any chunk of the bin starts at (tsBAIbin.StartVA)
                    uint64_t lo = bins[b][0].start_va;
                    for (const Chunk &c : bins[b]) lo = std::min(lo, c.start_va);
                    put64(lo);
                }
                put32((uint32_t)bins[b].size());
                for (const Chunk &c : bins[b]) { put64(c.start_va); put64(c.end_va); }
                bins[b].clear();
            }
            if (!csi) {                      // the 16 kb linear index exists in the BAI only
                put32(n_lin);
                for (uint32_t k = 0; k < n_lin; k++) put64(lin[k]);
            }
        } else
            put32(0);                        // the reference writes this second zero ("n_intv") in both formats
        used.clear();
        std::fill(lin.begin(), lin.end(), 0);
        n_lin = 0;
    };
    int cur_ref = 0;
    bool any = false;
    for (const BamAligned &a : aligned) {
        any = true;
        while (cur_ref < a.ref) { flush_ref(); cur_ref++; }
        const uint64_t sva = va(a.u_beg), eva = va(a.u_end);
        const uint32_t start = (uint32_t)a.pos, end = (uint32_t)a.end;
        const uint32_t k = start / 0x4000;
        if (lin.size() <= k) lin.resize((size_t)k + 1024, 0);
        if (lin[k] == 0) { n_lin = k + 1; lin[k] = sva; }
        const int bin = csi ? csi_reg2bin((int64_t)start, (int64_t)end, min_shift, depth)
                            : bam_reg2bin((int)start, (int)end);         // inclusive end, as AddChunk passes it
        if ((size_t)bin >= bins.size()) { if (err) *err = "index bin out of range"; return -1; }
        std::vector<Chunk> &cl = bins[bin];
        if (cl.empty()) {
            used.push_back(bin);
            cl.push_back({start, end, sva, eva});
        } else {
            Chunk &c = cl.back();
            if (start > c.end + 1) cl.push_back({start, end, sva, eva});
            else {
                if (c.start > start) { c.start = start; c.start_va = sva; }
                if (c.end < end) c.end = end;
                c.end_va = eva;
            }
        }
    }
    (void)any;
    flush_ref();                             // Close(): the reference being indexed when the records ended
    const std::string bpath = path + (csi ? ".csi" : ".bai");
    fd = ::open(bpath.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) { if (err) *err = "unable to create '" + bpath + "'"; return -91; }
    bool good = true;
    if (csi) {
        // the CSI goes through bgzf_write (SAMfile.cpp:1688, WriteIdxToDisk :1826): 0xff00-byte blocks at the BAM's compression
        // level, then bgzf_close's empty block
        std::vector<uint8_t> blk;
        for (uint64_t u = 0; u < bai.size() && good; u += kBlock)
            good = bgzf_block(bai.data() + u, (size_t)std::min<uint64_t>(kBlock, bai.size() - u), 6, blk) && write_all(fd, blk.data(), blk.size());
        if (good) good = bgzf_block(nullptr, 0, Z_DEFAULT_COMPRESSION, blk) && write_all(fd, blk.data(), blk.size());
    } else
        good = write_all(fd, bai.data(), bai.size());
    fsync(fd);
    ::close(fd);
    if (!good) { if (err) *err = "write failed"; return -96; }
    return 0;
}

}  // namespace bk
