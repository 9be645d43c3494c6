// cli_common.h - what the pieces of the command-line front end share: log lines in the reference's format, the option map, the
// read store, buffered (optionally gzip'd) file output, the NAR tags.
#pragma once
#include <fcntl.h>
#include <atomic>
#include <thread>
#include <algorithm>
#include <sys/time.h>
#include <unistd.h>
#include <zlib.h>

#include <cerrno>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <sys/mman.h>
#include <map>
#include <string>
#include "../bk_cpus.h"
#include <vector>

#include "noinit.h"

namespace bkcli {


// [0, n) cut into one contiguous range per thread: fn(lo, hi, t) with t < nthreads.  Small n runs on the caller's thread.
template <class Fn>
inline void par_ranges(size_t n, int nthreads, Fn fn)
{
    const size_t nt = std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, nthreads), n / 65536));
    if (nt <= 1) { fn((size_t)0, n, 0); return; }
    std::vector<std::thread> th;
    for (size_t t = 1; t < nt; t++) th.emplace_back([&fn, n, nt, t]() { fn(n * t / nt, n * (t + 1) / nt, (int)t); });
    fn((size_t)0, n / nt, 0);
    for (auto &x : th) x.join();
}

inline const char *kProgVer = "4.4.2";          // cpszProgVer of the release whose formats are kept (biokanga.cpp:33)
inline std::string g_proc = "biokanga";          // gszProcName = basename(argv[0]) (used for @PG ID:)
inline FILE *g_logfile = nullptr;

// CDiagnostics::DiagOut style line: [Www Mmm dd hh:mm:ss.mmm yyyy](proc) text
inline void diag(const char *fmt, ...)
{
    char msg[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof(msg), fmt, ap);
    va_end(ap);
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    struct tm tmv;
    localtime_r(&tv.tv_sec, &tmv);
    char ts[64], line[4300];
    strftime(ts, sizeof(ts), "%b %e %H:%M:%S", &tmv);
    snprintf(line, sizeof(line), "[%s.%03d %d](%s) %s\n", ts, (int)(tv.tv_usec / 1000), tmv.tm_year + 1900, g_proc.c_str(), msg);
    fputs(line, stdout);
    fflush(stdout);
    if (g_logfile) { fputs(line, g_logfile); fflush(g_logfile); }
}

// ---------------------------------------------------------------------------------------------
// tiny option parser: -x val, -xval, --long=val, --long val; repeated options accumulate
struct Args {
    std::map<std::string, std::vector<std::string>> v;
    bool has(const std::string &k) const { return v.count(k) != 0; }
    std::string str(const std::string &k, const std::string &d = "") const { return has(k) ? v.at(k).back() : d; }
    int num(const std::string &k, int d) const { return has(k) ? atoi(v.at(k).back().c_str()) : d; }
};

inline const char *kNarTag[20] = {"NA", "AA", "EN", "NL", "MH", "ML", "ET", "OJ", "OM", "DP", "DS", "FC", "PR", "UI", "OI", "UP", "IS", "IT", "NP", "LC"};
inline const char *kNarDescr[20] = {"Not processed for alignment", "Alignment accepted", "Excessive indeterminate (Ns) bases",
                             "No potential alignment loci", "Mismatch delta (minimum Hamming) criteria not met",
                             "Aligned to multiloci", "Excessively end trimmed", "Aligned as orphaned splice junction",
                             "Aligned as orphaned microInDel", "Duplicate PCR", "Duplicate read sequence",
                             "Aligned to filtered target sequence", "Aligned to a priority region", "PE under minimum insert size",
                             "PE over maximum insert size", "PE partner not aligned", "PE partner aligned to inconsistent strand",
                             "PE partner aligned to different target sequence", "PE alignment not accepted",
                             "Alignment violated loci base constraints"};

// (CPUs this process can actually keep busy - affinity mask and cgroup quota: the library's own reader, bk_cpus.h)
inline int effective_cpus() { return bk::effective_cpus(); }

struct ReadStore {
    bk::RawVec<uint8_t> bases;             // (RawVec: sized once, filled by all threads - see noinit.h)
    bk::RawVec<uint64_t> offs;
    bk::RawVec<uint32_t> lens;
    bk::RawVec<char> names;                // '\0' separated
    bk::RawVec<uint64_t> name_ofs;
    // bytes of `bases` the reads really occupy: the buffer itself may be the parser's, as large as the input file, with the reads lying
    // where its pieces wrote them (fasta.h, ParsedFile); 0 = the buffer holds nothing else
    uint64_t used_bases = 0;
    uint64_t base_bytes() const { return used_bases ? used_bases : bases.size(); }
    uint64_t name_bytes() const { return names.size(); }
    size_t size() const { return lens.size(); }
    const char *name(size_t i) const { return names.data() + name_ofs[i]; }
};

// The output file of a SAM run, created early: one background thread allocates its pages (Linux fallocate - which fails where the
// file system cannot do it natively instead of emulating it with racy reads and writes; then nothing is preallocated) 256 MB at a
// time from an estimate of the text's size, while the run does everything else.  The writers later copy into pages that exist.
struct SamPrealloc {
    int fd = -1;
    std::atomic<off_t> done{0};         // bytes from the file's start whose pages exist (and, with `map`, are in the page table)
    std::atomic<bool> quit{false}, ended{false};
    off_t est = 0;
    char *map = nullptr;                // the whole estimate mapped shared: writers memcpy into it below `done` without a single fault
    bool kept = false;                  // (set once the text is complete: a file abandoned before that is emptied)
    std::vector<std::thread> th;
    std::mutex mu;
    std::vector<uint8_t> chunk_done;
    static constexpr off_t kStep = 128LL << 20;
    // Pages of a new tmpfs / ext4 file are allocated and zeroed one by one whoever asks: fallocate() does it under the file's lock - one
    // thread's worth, 3.5 GB/s, which a 7 GB SAM file of a 50 M-read run waited for - while MADV_POPULATE_WRITE (Linux 5.14) on a shared
    // mapping of the hole allocates, zeroes and maps from as many threads as call it, and reports failure instead of raising SIGBUS.
    // Without it: the old single fallocate thread, and writers map the ranges they need.
    void start(const char *path, uint64_t estimate, int nthreads = 3)
    {
        fd = ::open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) return;
        est = (off_t)estimate;
        if (est > 0 && ftruncate(fd, est) == 0) {
            void *m = mmap(nullptr, (size_t)est, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m != MAP_FAILED) {
                if (madvise(m, 4096, 23 /* MADV_POPULATE_WRITE */) == 0) map = (char *)m;
                else munmap(m, (size_t)est);
            }
        }
        if (map) {
            n_chunks = (size_t)((est + kStep - 1) / kStep);
            chunk_done.assign(n_chunks, 0);
            add_threads(nthreads);
            return;
        }
        th.emplace_back([this]() {
            const off_t step = 256LL << 20;
            for (off_t at = 0; at < est && !quit.load(); at += step) {
                const off_t len = std::min<off_t>(step, est - at);
                if (fallocate(fd, 0, at, len) != 0) break;
                done.store(at + len);
            }
            ended.store(true);
        });
    }
    // (more threads for the populating form; nothing to add to a finished or a fallocate-only one)
    size_t n_chunks = 0;
    std::atomic<size_t> next_chunk{0};
    std::atomic<int> live{0};
    void add_threads(int n)
    {
        if (!map || ended.load()) return;
        for (int t = 0; t < n; t++) {
            live.fetch_add(1);
            th.emplace_back([this]() {
                for (;;) {
                    const size_t k = next_chunk.fetch_add(1);
                    if (k >= n_chunks || quit.load()) break;
                    const off_t at = (off_t)k * kStep, len = std::min<off_t>(kStep, est - at);
                    // (a few MB per call: the call holds the address space's lock shared, and a thread that wants it exclusively - any
                    // large allocation of the parser's - makes everybody else's page faults queue up behind it until the call returns)
                    bool ok = true;
                    for (off_t o = 0; o < len && ok && !quit.load(); o += (4 << 20)) ok = madvise(map + at + o, (size_t)std::min<off_t>(4 << 20, len - o), 23) == 0;
                    if (!ok) { quit.store(true); break; }
                    if (quit.load()) break;
                    std::lock_guard<std::mutex> g(mu);
                    chunk_done[k] = 1;
                    off_t d = done.load();
                    while (d < est && chunk_done[(size_t)(d / kStep)]) d = std::min<off_t>(est, (d / kStep + 1) * kStep);
                    done.store(d);
                }
                if (live.fetch_sub(1) == 1) ended.store(true);
            });
        }
    }
    void finish() { quit.store(true); for (auto &t : th) if (t.joinable()) t.join(); }
    // Written ranges leave the page table behind the writers (the file keeps the pages): 60 ms per GB of page-table and reverse-map work
    // that would otherwise be the process's last act.  A thread of its own does it, nobody waits for it; a large run ends the process
    // under it (what it has not reached goes with the address space), a small one joins it here.
    std::mutex um;
    std::condition_variable ucv;
    std::vector<std::pair<char *, size_t>> uq;
    bool ustop = false;
    std::vector<std::thread> uth;                  // (two: one falls behind a 50 GB/s stream of text)
    void unmap_behind(char *p, size_t n)
    {
        {
            std::lock_guard<std::mutex> lk(um);
            // (pieces of 64 MB: both threads get work out of one slice)
            for (size_t o = 0; o < n; o += (64u << 20)) uq.emplace_back(p + o, std::min<size_t>(64u << 20, n - o));
            while (uth.size() < 2)
                uth.emplace_back([this]() {
                    for (;;) {
                        std::pair<char *, size_t> r;
                        { std::unique_lock<std::mutex> lk2(um); ucv.wait(lk2, [&] { return ustop || !uq.empty(); }); if (uq.empty()) return; r = uq.back(); uq.pop_back(); }
                        (void)madvise(r.first, r.second, MADV_DONTNEED);
                    }
                });
        }
        ucv.notify_all();
    }
    ~SamPrealloc()
    {
        finish();
        { std::lock_guard<std::mutex> lk(um); ustop = true; }
        ucv.notify_all();
        for (auto &t : uth) if (t.joinable()) t.join();
        if (map) munmap(map, (size_t)est);
        if (fd >= 0) { if (!kept && ftruncate(fd, 0) != 0) {} ::close(fd); }
    }
};

// `n` bytes as gzip members appended to `out`: a gzip file is any number of members one after the other, so threads can compress
// their own stretches of an output and the members go out in order.  The members are bgzip's (BGZF: at most 0xff00 bytes of text
// each, its whole size in a 'BC' extra field; level 6, gzopen's default): every gzip reader takes the file as it would one stream,
// and readers that know the framing - samtools, tabix, this package's own loader - can inflate it by all their threads.
inline bool gzip_members(const char *s, size_t n, std::vector<uint8_t> &out)
{
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    const size_t kText = 0xff00, kRoom = 0x10000;
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out.reserve(out.size() + n / 3 + kRoom);
    bool ok = true;
    for (size_t o = 0; ok && o < n; o += kText) {
        const size_t k = std::min(kText, n - o), at = out.size();
        out.resize(at + kRoom);
        uint8_t *b = out.data() + at;
        memcpy(b, hdr, 16);
        z.next_in = (Bytef *)const_cast<char *>(s + o);
        z.avail_in = (uInt)k;
        z.next_out = b + 18;
        z.avail_out = (uInt)(kRoom - 18 - 8);
        ok = deflate(&z, Z_FINISH) == Z_STREAM_END;
        const size_t made = kRoom - 18 - 8 - z.avail_out, total = 18 + made + 8;
        const uint16_t bsize = (uint16_t)(total - 1);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)(s + o), (uInt)k), isize = (uint32_t)k;
        memcpy(b + 16, &bsize, 2);
        memcpy(b + 18 + made, &crc, 4);
        memcpy(b + 18 + made + 4, &isize, 4);
        out.resize(at + (ok ? total : 0));
        deflateReset(&z);
    }
    deflateEnd(&z);
    return ok;
}

// the empty member that ends a bgzip'd file (and is all of an empty one)
inline const uint8_t *bgzf_eof(size_t *n)
{
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    *n = sizeof(eof);
    return eof;
}

struct OutBuf {
    int fd = -1;
    // set when the name ends in ".gz" (CAligner::FileReqWriteCompr, Aligner.cpp:4337): what is put goes out as gzip members - one per
    // flush from here, or made by the caller's own threads (put_members) - where the reference has gzwrite's single stream; readers
    // see the same text
    bool gz = false, gz_wrote = false;
    std::atomic<bool> failed{false};        // a write came up short (several formatting threads may say so)
    bool pipe = false;                      // a FIFO or a pipe (-o >(samtools ..)): no offsets - everything in order through write()
    off_t pos = 0;                          // file offset of the next byte (everything goes through pwrite)
    std::vector<char> b;
    void open(const char *path)
    {
        size_t n = strlen(path);
        // read-write: the SAM writer maps ranges of the file (a shared mapping needs a readable descriptor); write-only for what cannot
        // be opened that way (a FIFO, /dev/stdout) - those are written with pwrite() / write()
        fd = ::open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        gz = fd >= 0 && n > 3 && !strcasecmp(path + n - 3, ".gz");
        pipe = fd >= 0 && lseek(fd, 0, SEEK_CUR) == (off_t)-1 && errno == ESPIPE;
        b.reserve(8 << 20);
        pos = 0;
    }
    void put(const char *s, size_t n)
    {
        if (!gz && n >= (1u << 20)) { flush(); write_raw(s, n); return; }          // (a thread's whole stretch: no second copy)
        b.insert(b.end(), s, s + n);
        if (b.size() > (4u << 20)) flush();
    }
    void put(const std::string &s) { put(s.data(), s.size()); }
    void write_raw(const void *p, size_t n)
    {
        size_t o = 0;
        while (o < n) {
            ssize_t w = ::pwrite(fd, (const char *)p + o, n - o, pos + (off_t)o);
            if (w < 0 && errno == ESPIPE) w = ::write(fd, (const char *)p + o, n - o);      // (a pipe has no offsets)
            if (w <= 0) { failed = true; break; }
            o += (size_t)w;
        }
        pos += (off_t)o;
    }
    void flush()
    {
        if (gz) {
            if (!b.empty()) {
                std::vector<uint8_t> m;
                if (gzip_members(b.data(), b.size(), m)) { write_raw(m.data(), m.size()); gz_wrote = true; } else failed = true;
            }
            b.clear();
            return;
        }
        write_raw(b.data(), b.size());
        b.clear();
    }
    // gzip members the caller made of the text that follows what was put so far
    void put_members(const uint8_t *m, size_t n) { flush(); if (n) { write_raw(m, n); gz_wrote = true; } }
    bool borrowed = false;                  // the descriptor belongs to a SamPrealloc
    void close()
    {
        flush();
        if (gz && fd >= 0) { size_t n; const uint8_t *e = bgzf_eof(&n); write_raw(e, n); }       // (bgzip's end mark; an empty text is just that)
        if (fd >= 0) { fsync(fd); if (!borrowed) ::close(fd); }
        fd = -1;
    }
};

}  // namespace bkcli
