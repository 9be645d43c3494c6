// fasta.cpp - see fasta.h
#include "fasta.h"

#include <cctype>
#include <cstring>

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

void SeqReader::close()
{
    if (gz_) gzclose(gz_);
    gz_ = nullptr;
}

int SeqReader::fill()
{
    if (eof_) return 0;
    int n = gzread(gz_, buf_.data(), (unsigned)buf_.size());
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
        // @descr \n seq \n + \n qual
        while ((c = getc_()) >= 0 && (c == '\n' || c == '\r')) {}
        if (c < 0) return 0;
        if (c != '@') return -77;                    // eBSFerrFastqSeqID-ish: malformed record
        while ((c = getc_()) >= 0 && c != '\n') if (c != '\r') descr.push_back((char)c);
        while ((c = getc_()) >= 0 && c != '\n') {
            if (c == '\r') continue;
            if (isalpha(c) || c == '-') bases.push_back(a2s((uint8_t)c));
        }
        while ((c = getc_()) >= 0 && c != '\n') {}   // '+' line
        size_t nq = 0;
        while ((c = getc_()) >= 0 && c != '\n') if (c != '\r') nq++;
        (void)nq;
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

}  // namespace bk
