// fasta.cpp - see fasta.h
#include "fasta.h"

#include "fast_inflate.h"

#include <cmath>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <new>
#include <cctype>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cstdlib>
#include <cstring>
#include <thread>

namespace bk {

namespace {
struct A2S {
    uint8_t t[256];
    A2S()
    {
        for (int i = 0; i < 256; i++) t[i] = 4;   // default: eBaseN
        const char *lo = "acgt", *up = "ACGT";
        for (int i = 0; i < 4; i++) { t[(uint8_t)lo[i]] = (uint8_t)(i | 0x08); t[(uint8_t)up[i]] = (uint8_t)i; }
        t[(uint8_t)'u'] = 3 | 0x08;
        t[(uint8_t)'U'] = 3;
        t[(uint8_t)'-'] = 6;
    }
};
const A2S g_a2s;
}  // namespace

static inline uint8_t a2s(uint8_t c) { return g_a2s.t[c]; }

SeqReader::~SeqReader() { close(); }

int SeqReader::open(const std::string &path, std::string *err)
{
    close();
    path_ = path;
    gz_ = gzopen(path.c_str(), "rb");          // transparently reads plain files too
    if (!gz_) {
        if (err) *err = "unable to open '" + path + "'";
        return -90;
    }
    gzbuffer(gz_, 1 << 20);
    buf_.resize(4 << 20);
    pos_ = len_ = 0;
    eof_ = false;
    started_ = false;
    // file type from the first non-whitespace character
    int c;
    while ((c = getc_()) >= 0 && isspace(c)) {}
    if (c < 0) { fastq_ = false; return 0; }
    fastq_ = c == '@';
    if (c != '>' && c != '@') {
        if (err) *err = "'" + path + "' is not a multifasta short reads or fastq file";
        return -93;      // eBSFerrNotFasta
    }
    ungetc_();
    return 0;
}

void SeqReader::stop_ahead()
{
    if (!ahead_) return;
    { std::lock_guard<std::mutex> lk(ahead_->mu); ahead_->stop = true; }
    ahead_->cv.notify_all();
    if (ahead_->th.joinable()) ahead_->th.join();
    ahead_.reset();
}

void SeqReader::close()
{
    stop_ahead();
    if (gz_) gzclose(gz_);
    gz_ = nullptr;
}

int SeqReader::fill()
{
    if (eof_) return 0;
    if (!ahead_) {
        // the first buffer is read here; from the second on a thread reads one buffer ahead (only it touches gz_ from now on)
        int n = gzread(gz_, buf_.data(), (unsigned)buf_.size());
        if (n <= 0) { eof_ = true; len_ = pos_ = 0; return n < 0 ? -85 : 0; }
        len_ = (size_t)n;
        pos_ = 0;
        if ((size_t)n == buf_.size()) {              // (a file that fits one buffer needs no thread)
            ahead_.reset(new Ahead());
            Ahead *a = ahead_.get();
            a->buf.resize(buf_.size());
            a->want = true;
            gzFile gz = gz_;
            a->th = std::thread([a, gz]() {
                for (;;) {
                    { std::unique_lock<std::mutex> lk(a->mu); a->cv.wait(lk, [&] { return a->want || a->stop; }); if (a->stop) return; a->want = false; }
                    const int got = gzread(gz, a->buf.data(), (unsigned)a->buf.size());
                    { std::lock_guard<std::mutex> lk(a->mu); a->n = got; a->ready = true; }
                    a->cv.notify_all();
                    if (got <= 0) return;
                }
            });
        }
        return n;
    }
    Ahead *a = ahead_.get();
    int n;
    {
        std::unique_lock<std::mutex> lk(a->mu);
        a->cv.wait(lk, [&] { return a->ready; });
        a->ready = false;
        n = a->n;
        if (n > 0) { buf_.swap(a->buf); a->want = true; }
    }
    a->cv.notify_all();
    if (n <= 0) { eof_ = true; len_ = pos_ = 0; return n < 0 ? -85 : 0; }
    len_ = (size_t)n;
    pos_ = 0;
    return n;
}

int SeqReader::getc_()
{
    if (pos_ >= len_) {
        if (fill() <= 0) return -1;
    }
    return buf_[pos_++];
}

int SeqReader::next(std::string &descr, std::vector<uint8_t> &bases)
{
    descr.clear();
    bases.clear();
    int c;
    if (fastq_) {
        // CFasta::ParseFastQblockQ (libbiokanga/Fasta.cpp:1206-1440): @id / sequence / +[id] / qualities, one line
        // each, blank lines between elements sloughed; the sequence may only hold acgtn (either case) - IUPAC codes
        // and SOLiD colour calls end the run, as they do in the reference (colour space is out of scope here);
        // sequence and quality lengths must agree; characters outside 0x20..0x7f are errors
        auto bad_chr = [](int ch) { return !isspace(ch) && (ch < 0x20 || ch > 0x7f); };
        while ((c = getc_()) >= 0 && isspace(c)) {}
        if (c < 0) return 0;
        if (c != '@') return -77;                    // eBSFerrFastqSeqID
        while ((c = getc_()) >= 0 && c != '\n' && c != '\r') {
            if (bad_chr(c)) return -76;              // eBSFerrFastqChr
            if (descr.size() < 8192) descr.push_back((char)c);              // cMaxFastaDescrLen
        }
        for (;;) {                                   // sequence line (leading blank lines sloughed)
            c = getc_();
            if (c < 0) break;
            if (c == '\n' || c == '\r') { if (bases.empty()) continue; break; }
            switch (c) {
            case 'a': case 'A': case 'c': case 'C': case 'g': case 'G': case 't': case 'T': case 'n': case 'N':
                if (bases.size() < 0x30000) bases.push_back(a2s((uint8_t)c));   // cMaxFastQSeqLen (commdefs.h:161): silently truncated
                break;
            default:
                return -78;                          // eBSFerrFastqSeq
            }
        }
        while ((c = getc_()) >= 0 && (c == '\n' || c == '\r')) {}
        if (c != '+') return -79;                    // eBSFerrFastqDescr
        while ((c = getc_()) >= 0 && c != '\n' && c != '\r') if (bad_chr(c)) return -76;
        size_t nq = 0;
        qual_.clear();
        for (;;) {                                   // quality line
            c = getc_();
            if (c < 0) break;
            if (c == '\n' || c == '\r') { if (nq == 0) continue; break; }
            if (bad_chr(c)) return -76;
            if (nq < 0x30000) { nq++; if (qmode_ != 3) qual_.push_back((uint8_t)c); }
        }
        if (descr.empty() || bases.empty() || nq != bases.size()) return -85;   // eBSFerrFileAccess: empty or unequal elements
        if (qmode_ != 3) {
            // Aligner.cpp:11133-11194: clamp to the encoding's range, Phred capped at 40, 4 bits per base
            for (size_t i = 0; i < nq; i++) {
                int q = qual_[i], ph;
                switch (qmode_) {
                case 0: if (q < 33) q = 33; else if (q >= 126) q = 125; ph = q - 33; break;
                case 1: if (q < 64) q = 64; else if (q >= 126) q = 125; ph = q - 64; break;
                default:
                    if (q < 59 || q >= 126) q = q < 64 ? 64 : 125;
                    ph = q - 59;
                    ph = (int)(uint8_t)(10 * log(1 + pow(10.0, ((double)ph / 10.0) / log(10.0))));
                    break;
                }
                if (ph > 40) ph = 40;
                bases[i] |= (uint8_t)((((uint32_t)ph + 2) * 15) / 40) << 4;
            }
        }
        return 1;
    }
    // FASTA
    while ((c = getc_()) >= 0 && c != '>') {}        // skip to the next descriptor
    if (c < 0) return 0;
    while ((c = getc_()) >= 0 && c != '\n' && c != '\r') {
        if ((unsigned)c > 0x7f) c = '?';
        descr.push_back((char)c);
    }
    for (;;) {
        c = getc_();
        if (c < 0) break;
        if (c == '>') { ungetc_(); break; }
        if (isalpha(c) || c == '-') bases.push_back(a2s((uint8_t)c));
    }
    return 1;
}

// ------------------------------------------------------------------------------------------------
namespace {

// same state machine as SeqReader::next (FASTA branch) over [p, e); e is a record start or the file end.  One table look-up per
// sequence character (0xff = not a sequence character: everything but letters and '-'), output through raw pointers into
// buffers sized for the worst case up front.
struct SeqLut {
    uint8_t t[256];
    SeqLut() { for (int i = 0; i < 256; i++) t[i] = (isalpha(i) || i == '-') ? a2s((uint8_t)i) : 0xff; t[(uint8_t)'>'] = 0xfe; }
};
const SeqLut g_seq_lut;

// 32 sequence characters at a time: while they are all of a,c,g,t / A,C,G,T (what a read's line is made of but for its end) their codes
// come from bit arithmetic - ((c >> 1) ^ (c >> 2)) & 3 is 0,1,2,3 for a,c,g,t in either case, bit 5 of the letter is the soft-mask flag
// (0x08 of the code).  Returns how many leading characters of the 32 it translated (all stored; the caller advances by that many).
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline unsigned acgt_run32(const uint8_t *p, uint8_t *bw)
{
    const __m256i v = _mm256_loadu_si256((const __m256i *)p);
    const __m256i up = _mm256_and_si256(v, _mm256_set1_epi8((char)0xdf));
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(up, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(up, _mm256_set1_epi8('C'))),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(up, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(up, _mm256_set1_epi8('T'))));
    const unsigned m = (unsigned)_mm256_movemask_epi8(ok);
    const __m256i code = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), _mm256_set1_epi8(3));
    const __m256i soft = _mm256_srli_epi16(_mm256_and_si256(v, _mm256_set1_epi8(0x20)), 2);
    _mm256_storeu_si256((__m256i *)bw, _mm256_or_si256(code, soft));
    return m == 0xffffffffu ? 32u : (unsigned)__builtin_ctz(~m);
}
const bool g_have_avx2 = __builtin_cpu_supports("avx2");
#else
inline unsigned acgt_run32(const uint8_t *, uint8_t *) { return 0; }
const bool g_have_avx2 = false;
#endif

// bw: where this piece's bases go (the piece's offset in the file-sized buffer of its ParsedFile)
void parse_range(const uint8_t *p, const uint8_t *e, uint8_t *bw, ParsedChunk &out, bool mid = false)
{
    const size_t approx = (size_t)(e - p);
    out.bases = bw;
    out.lens.reserve(approx / 64 + 16);
    out.descr_lens.reserve(approx / 64 + 16);
    RawVec<char> &dv = out.descr_own;
    dv.resize(approx / 4 + 256);
    size_t dn = 0;
    const uint8_t *lut = g_seq_lut.t;
    out.continues = mid;
    while (p < e) {
        if (mid) {
            // a piece that starts at a line inside a record (split_records): its first bases go on with the record in the piece in front
            mid = false;
            out.descr_lens.push_back(0);
        } else {
            p = (const uint8_t *)memchr(p, '>', (size_t)(e - p));      // skip to the next descriptor
            if (!p) break;
            p++;
            // the descriptor: up to the line's end ('\n' or '\r')
            const uint8_t *q = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
            if (!q) q = e;
            if (const uint8_t *cr = (const uint8_t *)memchr(p, '\r', (size_t)(q - p))) q = cr;
            const size_t dl = (size_t)(q - p);
            if (dn + dl > dv.size()) dv.resize(std::max(dv.size() * 2, dn + dl + 256));
            char *d0 = dv.data() + dn;
            uint8_t high = 0;
            for (size_t i = 0; i < dl; i++) { d0[i] = (char)p[i]; high |= p[i]; }
            if (high & 0x80) for (size_t i = 0; i < dl; i++) if ((uint8_t)d0[i] > 0x7f) d0[i] = '?';
            dn += dl;
            p = q;
            out.descr_lens.push_back((uint32_t)dl);
        }
        uint8_t *b0 = bw;
        while (p < e) {
            // (a piece never writes more bases than it has read characters: 32 bytes stored at bw stay inside the piece while p + 32 <= e)
            if (g_have_avx2 && p + 32 <= e) {
                const unsigned k = acgt_run32(p, bw);
                p += k; bw += k;
                if (k == 32) continue;
            }
            const uint8_t v = lut[*p];
            if (v == 0xfe) break;                                      // '>': the next record
            p++;
            *bw = v;
            bw += v != 0xff;
        }
        out.lens.push_back((uint32_t)(bw - b0));
    }
    out.descr = dv.data();
}

// first record start at or after q: a '>' with no other '>' between it and the preceding line break
// (a '>' inside a descriptor line is part of the descriptor)
const uint8_t *next_record_start(const uint8_t *base, const uint8_t *q, const uint8_t *end)
{
    while (q < end) {
        while (q < end && *q != '>') q++;
        if (q >= end) return end;
        const uint8_t *r = q;
        bool inside = false;
        while (r > base) {
            --r;
            if (*r == '\n' || *r == '\r') break;
            if (*r == '>') { inside = true; break; }
        }
        if (!inside) return q;
        q++;
    }
    return end;
}


// ---- FASTQ, the same way ----------------------------------------------------------------------------------------------------------
// SeqReader::next's FASTQ state machine over [p, e), p a record start: '@' id line, ONE sequence line of a,c,g,t,n (either case), '+' line,
// ONE quality line of the sequence's length; blank lines between the elements are sloughed.  Anything the serial reader would refuse -
// and anything this form does not want to decide (a record cut by e) - makes the piece give up: the file then goes through the serial
// reader, which says what is wrong and where.
inline bool fq_bad_chr(uint8_t c) { return !isspace(c) && (c < 0x20 || c > 0x7f); }

// no character of [p, p + n) is one the reader refuses (eight at a time: a byte below 0x20 or above 0x7f sends the stretch to the exact test)
inline bool fq_clean(const uint8_t *p, size_t n)
{
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x;
        memcpy(&x, p + i, 8);
        // high bit set, or a byte below 0x20 (the subtraction borrows into bit 7 of such a byte; bytes with the high bit set are caught by the first test)
        if ((x & 0x8080808080808080ULL) || ((x - 0x2020202020202020ULL) & ~x & 0x8080808080808080ULL)) {
            for (size_t k = i; k < i + 8; k++) if (fq_bad_chr(p[k])) return false;
        }
    }
    for (; i < n; i++) if (fq_bad_chr(p[i])) return false;
    return true;
}

inline uint8_t fq_score_nibble(int q, int qmode)
{
    int ph;
    switch (qmode) {
    case 0: if (q < 33) q = 33; else if (q >= 126) q = 125; ph = q - 33; break;
    case 1: if (q < 64) q = 64; else if (q >= 126) q = 125; ph = q - 64; break;
    default:
        if (q < 59 || q >= 126) q = q < 64 ? 64 : 125;
        ph = q - 59;
        ph = (int)(uint8_t)(10 * log(1 + pow(10.0, ((double)ph / 10.0) / log(10.0))));
        break;
    }
    if (ph > 40) ph = 40;
    return (uint8_t)((((uint32_t)ph + 2) * 15) / 40);
}

bool parse_fastq_range(const uint8_t *p, const uint8_t *e, uint8_t *bw, int qmode, ParsedChunk &out)
{
    const size_t approx = (size_t)(e - p);
    out.bases = bw;
    out.lens.reserve(approx / 200 + 16);
    out.descr_lens.reserve(approx / 200 + 16);
    RawVec<char> &dv = out.descr_own;
    dv.resize(approx / 6 + 256);
    size_t dn = 0;
    uint8_t score[256];
    if (qmode != 3) for (int q = 0; q < 256; q++) score[q] = (uint8_t)(fq_score_nibble(q, qmode) << 4);
    for (;;) {
        while (p < e && isspace(*p)) p++;
        if (p >= e) break;
        if (*p != '@') return false;
        p++;
        // id line
        const uint8_t *q = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
        if (!q) return false;
        if (const uint8_t *cr = (const uint8_t *)memchr(p, '\r', (size_t)(q - p))) q = cr;
        if (!fq_clean(p, (size_t)(q - p))) return false;
        size_t dl = std::min<size_t>((size_t)(q - p), 8192);               // cMaxFastaDescrLen
        if (dn + dl > dv.size()) dv.resize(std::max(dv.size() * 2, dn + dl + 256));
        memcpy(dv.data() + dn, p, dl);
        p = q;
        // sequence line (leading blank lines sloughed)
        while (p < e && (*p == '\n' || *p == '\r')) p++;
        uint8_t *b0 = bw;
        size_t nb = 0;
        while (p < e && *p != '\n' && *p != '\r') {
            // (32 letters of a,c,g,t at a time, as in the FASTA pieces; an N, the line's end or anything else goes the exact way)
            if (g_have_avx2 && p + 32 <= e && nb + 32 <= 0x30000) {
                const unsigned k = acgt_run32(p, bw);
                p += k; bw += k; nb += k;
                if (k == 32) continue;
                if (p >= e || *p == '\n' || *p == '\r') break;
            }
            const uint8_t c = *p++;
            uint8_t v;
            switch (c) {
            case 'a': v = 0 | 8; break; case 'A': v = 0; break;
            case 'c': v = 1 | 8; break; case 'C': v = 1; break;
            case 'g': v = 2 | 8; break; case 'G': v = 2; break;
            case 't': v = 3 | 8; break; case 'T': v = 3; break;
            case 'n': case 'N': v = 4; break;
            default: return false;
            }
            if (nb < 0x30000) { *bw++ = v; nb++; }                          // cMaxFastQSeqLen: silently truncated
        }
        if (p >= e) return false;
        // '+' line
        while (p < e && (*p == '\n' || *p == '\r')) p++;
        if (p >= e || *p != '+') return false;
        p++;
        {
            const uint8_t *z = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
            if (!z) return false;
            if (const uint8_t *cr = (const uint8_t *)memchr(p, '\r', (size_t)(z - p))) z = cr;
            if (!fq_clean(p, (size_t)(z - p))) return false;
            p = z;
        }
        // quality line (leading blank lines sloughed); the file's last line may end without a line break
        while (p < e && (*p == '\n' || *p == '\r')) p++;
        size_t nq = 0;
        {
            const uint8_t *z = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
            if (!z) z = e;
            if (const uint8_t *cr = (const uint8_t *)memchr(p, '\r', (size_t)(z - p))) z = cr;
            const size_t n = (size_t)(z - p);
            if (!fq_clean(p, n)) return false;
            nq = std::min<size_t>(n, 0x30000);
            if (qmode != 3) for (size_t i = 0; i < nq && i < nb; i++) b0[i] |= score[p[i]];
            p = z;
        }
        if (dl == 0 || nb == 0 || nq != nb) return false;
        dn += dl;
        out.descr_lens.push_back((uint32_t)dl);
        out.lens.push_back((uint32_t)nb);
    }
    out.descr = dv.data();
    return true;
}

// first record start at or after q: a line that begins with '@', followed by a line of nothing but a,c,g,t,n, a line that begins with
// '+', and a line of the sequence line's length - in a file of such records no other line passes (the one quality line that may begin with
// '@' is followed by an id line, which is no sequence)
const uint8_t *next_fastq_record(const uint8_t *base, const uint8_t *q, const uint8_t *end)
{
    auto line_end = [&](const uint8_t *x) { while (x < end && *x != '\n' && *x != '\r') x++; return x; };
    auto next_line = [&](const uint8_t *x) { while (x < end && (*x == '\n' || *x == '\r')) x++; return x; };
    // to the start of a line
    while (q > base && q[-1] != '\n' && q[-1] != '\r') q--;
    for (int tries = 0; q < end && tries < 64; tries++) {
        q = next_line(q);
        if (q >= end) return end;
        if (*q == '@') {
            const uint8_t *l1e = line_end(q), *l2 = next_line(l1e), *l2e = line_end(l2), *l3 = next_line(l2e), *l3e = line_end(l3), *l4 = next_line(l3e), *l4e = line_end(l4);
            bool ok = l2 < l2e && l3 < end && *l3 == '+' && (l4e - l4) == (l2e - l2);
            for (const uint8_t *x = l2; ok && x < l2e; x++) {
                const uint8_t c = (uint8_t)(*x | 0x20);
                ok = c == 'a' || c == 'c' || c == 'g' || c == 't' || c == 'n';
            }
            if (ok) return q;
        }
        q = line_end(q);
    }
    return nullptr;                                     // nothing like a record near here: the serial reader takes the file
}

}  // namespace

// A bgzip'd file (the BGZF framing of SAMtools' bgzip: gzip members of <= 64 KB that carry their own size in a 'BC' extra field) is
// the one gzip layout whose members can be found without inflating: walk the headers, then every thread inflates members of its
// own straight to their place in the text.  Anything else - an ordinary .gz, a member without the field, a size, length or CRC-32
// that does not hold - is not ours: false, and the serial reader (gzread) takes the file and says what it finds.
static bool inflate_bgzf(const uint8_t *base, size_t size, int nthreads, RawVec<uint8_t> &text)
{
    struct Member { size_t src, clen, dst; uint32_t isize, crc; };
    std::vector<Member> mem;
    size_t o = 0, total = 0;
    while (o < size) {
        if (size - o < 28) return false;
        const uint8_t *h = base + o;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) return false;
        const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
        if (12 + xlen + 8 > size - o) return false;
        size_t bsize = 0;
        for (size_t x = 0; x + 4 <= xlen;) {
            const uint8_t *f = h + 12 + x;
            const size_t slen = (size_t)f[2] | ((size_t)f[3] << 8);
            if (x + 4 + slen > xlen) return false;
            if (f[0] == 'B' && f[1] == 'C' && slen == 2) bsize = ((size_t)f[4] | ((size_t)f[5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || bsize > size - o) return false;
        auto le32 = [](const uint8_t *q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
        Member m{o + 12 + xlen, bsize - 12 - xlen - 8, total, le32(h + bsize - 4), le32(h + bsize - 8)};
        if (m.isize > 65536) return false;                 // (a BGZF member holds at most 64 KB of text: a file whose trailers claim more is not one - the serial reader takes it)
        mem.push_back(m);
        total += m.isize;
        o += bsize;
    }
    if (total < (1u << 20)) return false;
    try { text.resize(total + 64); } catch (const std::bad_alloc &) { return false; }       // (no room: the serial reader takes the file)
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    if (nthreads < 1) nthreads = 1;
    std::vector<std::thread> th;
    for (int w = 0; w < nthreads; w++)
        th.emplace_back([&]() {
            z_stream z;
            memset(&z, 0, sizeof(z));
            bool z_ready = false;
            for (;;) {
                const size_t i0 = next.fetch_add(16);
                if (i0 >= mem.size() || bad) break;
                for (size_t i = i0; i < std::min(mem.size(), i0 + 16); i++) {
                    const Member &m = mem[i];
                    uint8_t *dst = text.data() + m.dst;
                    size_t used = 0;
                    bool ok = inflate_raw(base + m.src, m.clen, dst, m.isize, dst, &used) == (long)m.isize && used == m.clen;
                    if (!ok) {                              // (a member our decoder does not take: zlib's word on it)
                        if (!z_ready && inflateInit2(&z, -15) != Z_OK) { bad = 1; break; }
                        z_ready = true;
                        z.next_in = const_cast<Bytef *>(base + m.src);
                        z.avail_in = (uInt)m.clen;
                        z.next_out = dst;
                        z.avail_out = m.isize;
                        ok = inflate(&z, Z_FINISH) == Z_STREAM_END && z.avail_in == 0 && z.avail_out == 0;
                        inflateReset(&z);
                    }
                    if (!ok || (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, m.isize) != m.crc) { bad = 1; break; }
                }
            }
            if (z_ready) inflateEnd(&z);
        });
    for (auto &t : th) t.join();
    if (bad) { RawVec<uint8_t>().swap(text); return false; }
    memset(text.data() + total, '\n', 64);
    text.resize(total);
    return true;
}

// Any other gzip file: its members one after the other (a deflate stream has no entry points) by the decoder of fast_inflate.h - a
// large member of text by several threads that start at guessed block boundaries, see there - into one buffer, and the CRC-32 of
// every member by all threads afterwards.  The text's size is not known up front (the trailer's length is modulo 4 GB and says
// nothing of other members): the buffer is sized for eight times the file, untouched pages costing nothing, and a file that
// inflates beyond that - or holds anything the decoder or this reader of RFC 1952 headers does not take - goes to the serial reader.
// the memory this process may use: the machine's, or the control group's limit where one is set
static uint64_t memory_budget()
{
    uint64_t m = (uint64_t)sysconf(_SC_PHYS_PAGES) * (uint64_t)sysconf(_SC_PAGESIZE);
    for (const char *f : {"/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"})
        if (FILE *h = fopen(f, "r")) {
            unsigned long long v = 0;
            if (fscanf(h, "%llu", &v) == 1 && v > 0 && v < m) m = v;
            fclose(h);
        }
    return m;
}

static bool inflate_gzip(const uint8_t *base, size_t size, int nthreads, RawVec<uint8_t> &text)
{
    if (size < 18) return false;
    const size_t cap = std::max<size_t>(size * 8, (size_t)64 << 20);
    try { text.resize(cap); } catch (const std::bad_alloc &) { return false; }
    struct Member { size_t dst, n; uint32_t crc; };
    std::vector<Member> mem;
    auto le32 = [](const uint8_t *q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
    size_t o = 0, w = 0;
    // (several threads hold the text once as bytes and about twice as 16-bit symbols until the pieces are joined: some three times a text
    // of about four times the file - only where that is well within the memory at hand)
    bool ok = true, several = nthreads > 1 && (uint64_t)size * 13 < memory_budget() / 2;
    while (ok && o < size) {
        const uint8_t *h = base + o;
        if (size - o < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 0xe0)) { ok = false; break; }
        size_t p = o + 10;
        if (h[3] & 4) {
            if (size - p < 2) { ok = false; break; }
            const size_t xlen = (size_t)base[p] | ((size_t)base[p + 1] << 8);
            if (size - p - 2 < xlen) { ok = false; break; }
            p += 2 + xlen;
        }
        for (int f = 8; f <= 16 && ok; f <<= 1)
            if (h[3] & f) {
                const void *z = memchr(base + p, 0, size - p);
                if (!z) ok = false; else p = (size_t)((const uint8_t *)z - base) + 1;
            }
        if (ok && (h[3] & 2)) p += 2;
        if (!ok || p + 8 > size) { ok = false; break; }
        size_t used = 0;
        // (several threads for a large member - until one attempt comes back as one thread's work: then the file is not one stream of
        // text to its end, and every further attempt would decode the rest of the file only to find that out again)
        int pieces = 1;
        const long n = several ? inflate_raw_parallel(base + p, size - p - 8, text.data() + w, cap - w, &used, nthreads, &pieces)
                               : inflate_raw(base + p, size - p - 8, text.data() + w, cap - w, text.data() + w, &used);
        if (pieces < 2) several = false;
        if (n < 0 || size - p - used < 8) { ok = false; break; }
        p += used;
        if ((uint32_t)n != le32(base + p + 4)) { ok = false; break; }
        mem.push_back(Member{w, (size_t)n, le32(base + p)});
        w += (size_t)n;
        o = p + 8;
    }
    if (ok && w >= (1u << 20)) {
        // every member's CRC-32 out of pieces the threads take, joined by crc32_combine
        struct Piece { size_t at, n; uint32_t crc; };
        std::vector<Piece> pc;
        std::vector<size_t> first(mem.size() + 1, 0);
        const size_t step = std::max<size_t>((size_t)4 << 20, w / (size_t)(std::max(1, nthreads) * 4));
        for (size_t i = 0; i < mem.size(); i++) {
            first[i] = pc.size();
            for (size_t a = 0; a < mem[i].n; a += step) pc.push_back(Piece{mem[i].dst + a, std::min(step, mem[i].n - a), 0});
        }
        first[mem.size()] = pc.size();
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        auto work = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < pc.size();) {
                uLong c = crc32(0L, Z_NULL, 0);
                for (size_t a = 0; a < pc[i].n; a += (size_t)1 << 30) c = crc32(c, text.data() + pc[i].at + a, (uInt)std::min<size_t>((size_t)1 << 30, pc[i].n - a));
                pc[i].crc = (uint32_t)c;
            }
        };
        for (int t = 1; t < nthreads; t++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        for (size_t i = 0; i < mem.size() && ok; i++) {
            uLong c = crc32(0L, Z_NULL, 0);
            for (size_t k = first[i]; k < first[i + 1]; k++) c = crc32_combine(c, pc[k].crc, (z_off_t)pc[k].n);
            ok = (uint32_t)c == mem[i].crc;
        }
    } else
        ok = false;
    if (!ok) { RawVec<uint8_t>().swap(text); return false; }
    text.resize(w);
    return true;
}

int parse_fasta_parallel(const std::string &path, int nthreads, ParsedFile &out, std::string *err, int qmode, bool split_records)
{
    out.chunks.clear();
    struct stat st;
    if (stat(path.c_str(), &st) == 0 && !S_ISREG(st.st_mode)) return 0;       // a FIFO, a pipe of a process substitution: the serial reader's
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) {
        if (err) *err = "unable to open '" + path + "'";
        return -90;
    }
    if (fstat(fd, &st) != 0 || st.st_size < (64 << 10)) { ::close(fd); return 0; }
    size_t size = (size_t)st.st_size;
    void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) return 0;
    if (nthreads < 1) nthreads = 1;
    const uint8_t *base = (const uint8_t *)m;
    RawVec<uint8_t> text;                                 // a gzip'd file's text: bgzip members by all threads, other layouts by one
    if (base[0] == 0x1f && base[1] == 0x8b) {
        const bool ours = inflate_bgzf(base, size, nthreads, text) || inflate_gzip(base, size, nthreads, text);
        munmap(m, size);
        m = nullptr;
        if (!ours) return 0;
        base = text.data();
        size = text.size();
    }
    auto unmap = [&]() { if (m) munmap(m, size); };
    if (size < (1u << 20)) { unmap(); return 0; }
    const uint8_t *end = base + size;
    const uint8_t *p = base;
    while (p < end && isspace(*p)) p++;
    if (p >= end || (*p != '>' && *p != '@')) { unmap(); return 0; }
    const bool fastq = *p == '@';
    size_t pieces = (size_t)nthreads * 4;                 // a few pieces per thread evens out the tail
    if (pieces > size / (256 << 10) + 1) pieces = size / (256 << 10) + 1;
    std::vector<const uint8_t *> cut(pieces + 1);
    cut[0] = p;
    cut[pieces] = end;
    for (size_t t = 1; t < pieces; t++) {
        const uint8_t *mid = base + size / pieces * t;
        if (!fastq && split_records) {
            // (a genome: a few records of any length; the pieces are cut at line starts, wherever in a record those are)
            const uint8_t *nl = (const uint8_t *)memchr(mid, '\n', (size_t)(end - mid));
            cut[t] = nl ? nl + 1 : end;
            continue;
        }
        cut[t] = fastq ? next_fastq_record(base, mid, end) : next_record_start(base, mid, end);
        if (cut[t] == nullptr) { unmap(); return 0; }
    }
    for (size_t t = 1; t < pieces; t++) if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    out.chunks.resize(pieces);
    try { out.bases.resize(size + 64); } catch (const std::bad_alloc &) { unmap(); return 0; }      // (sized, not touched: the pieces' own pages are the only ones that become real; no room: the serial reader takes the file)
    std::atomic<int> gave_up{0};
    std::vector<std::thread> th;
    for (int w = 0; w < nthreads; w++)
        th.emplace_back([&, w]() {
            for (size_t t = (size_t)w; t < pieces; t += (size_t)nthreads) {
                if (!fastq) parse_range(cut[t], cut[t + 1], out.bases.data() + (cut[t] - base), out.chunks[t], split_records && t > 0 && cut[t] < cut[t + 1] && *cut[t] != '>');
                else if (!parse_fastq_range(cut[t], cut[t + 1], out.bases.data() + (cut[t] - base), qmode, out.chunks[t])) gave_up = 1;
            }
        });
    for (auto &t : th) t.join();
    unmap();
    if (gave_up) { out.chunks.clear(); RawVec<uint8_t>().swap(out.bases); return 0; }      // (the serial reader says what is wrong, and where)
    return 1;
}

uint64_t text_bytes_estimate(const std::string &path)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) return 0;     // (a FIFO is not opened just to be looked at:
    int fd = ::open(path.c_str(), O_RDONLY);                                                      // its writer would see its reader go away)
    if (fd < 0) return 0;
    const size_t size = (size_t)st.st_size;
    uint8_t head[18] = {0}, tail[8] = {0};
    const bool gz = size >= 26 && pread(fd, head, 18, 0) == 18 && head[0] == 0x1f && head[1] == 0x8b && pread(fd, tail, 8, (off_t)(size - 8)) == 8;
    if (!gz) { ::close(fd); return size; }
    auto le32 = [](const uint8_t *q) { return (uint64_t)q[0] | ((uint64_t)q[1] << 8) | ((uint64_t)q[2] << 16) | ((uint64_t)q[3] << 24); };
    if (head[3] == 4 && head[12] == 'B' && head[13] == 'C') {
        // bgzip: from header to header (the common layout: 'BC' is the only extra field)
        void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            const uint8_t *base = (const uint8_t *)m;
            uint64_t total = 0;
            size_t o = 0;
            bool ok = true;
            while (o < size) {
                const uint8_t *h = base + o;
                if (size - o < 28 || h[0] != 0x1f || h[1] != 0x8b || h[3] != 4 || h[12] != 'B' || h[13] != 'C') { ok = false; break; }
                const size_t bsize = ((size_t)h[16] | ((size_t)h[17] << 8)) + 1;
                if (bsize < 26 || bsize > size - o) { ok = false; break; }
                total += le32(h + bsize - 4);
                o += bsize;
            }
            munmap(m, size);
            if (ok) { ::close(fd); return total; }
        }
    }
    ::close(fd);
    // the length word counts modulo 4 GB: of the texts it can stand for, the one whose ratio to the file is nearest to 3.5 (read files
    // deflate to between a fifth and a third)
    const uint64_t word = le32(tail + 4);
    uint64_t best = word;
    double best_off = 1e30;
    for (uint64_t est = word; est <= (uint64_t)size * 12 + (1ull << 32); est += 1ull << 32) {
        const double off = fabs((double)est / (double)size - 3.5);
        if (off < best_off) { best_off = off; best = est; }
    }
    return best;
}

int RecordStream::open(const std::string &path, int nthreads, std::string *err)
{
    int rc = nthreads > 1 ? parse_fasta_parallel(path, nthreads, file_, err, rd_.quality_mode()) : 0;
    if (rc < 0) return rc;
    parsed_ = rc == 1;
    ci_ = ri_ = bo_ = dofs_ = 0;
    return parsed_ ? 0 : rd_.open(path, err);
}

int RecordStream::next(const char *&d, size_t &dl, const uint8_t *&b, size_t &bl)
{
    if (!parsed_) {
        int rc = rd_.next(d_, b_);
        if (rc <= 0) return rc;
        d = d_.data(); dl = d_.size(); b = b_.data(); bl = b_.size();
        return 1;
    }
    std::vector<ParsedChunk> &chunks_ = file_.chunks;
    while (ci_ < chunks_.size() && ri_ >= chunks_[ci_].lens.size()) { ci_++; ri_ = 0; bo_ = 0; dofs_ = 0; }
    if (ci_ >= chunks_.size()) return 0;
    const ParsedChunk &c = chunks_[ci_];
    d = c.descr + dofs_; dl = c.descr_lens[ri_];
    b = c.bases + bo_; bl = c.lens[ri_];
    dofs_ += dl; bo_ += bl; ri_++;
    return 1;
}

}  // namespace bk
