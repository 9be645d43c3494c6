// fasta.h - FASTA / FASTQ (optionally gzip'd) record reader for the host front end.
// Follows the ingest semantics of the reference's CFasta (libbiokanga/Fasta.cpp:907-1137,1518-1560):
//   * a FASTA descriptor runs from '>' to end of line; '>' anywhere outside a descriptor starts a new one
//   * sequence characters: every non-alphabetic character except '-' is sloughed
//   * CFasta::Ascii2Sense: a/c/g/t/u -> 0..3 | cRptMskFlg(0x08), A/C/G/T/U -> 0..3, '-' -> eBaseInDel(6),
//     everything else -> eBaseN(4)
//   * FASTQ: 4-line records (@id, sequence, +, qualities); qualities are dropped (-g3 default) or packed 4 bits per base
#pragma once
#include <zlib.h>

#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "noinit.h"

namespace bk {

class SeqReader {
public:
    SeqReader() = default;
    ~SeqReader();
    // returns 0 or a negative teBSFrsltCodes value
    int open(const std::string &path, std::string *err);
    void close();
    // next record: descriptor (without '>' / '@') and bases already mapped by Ascii2Sense.
    // returns 1 = record, 0 = end of file, <0 = error
    int next(std::string &descr, std::vector<uint8_t> &bases);
    bool is_fastq() const { return fastq_; }
    // FASTQ quality scores (`-g`, CAligner::LoadReads biokanga/Aligner.cpp:11121-11200): 0 Sanger / Illumina 1.8+, 1 Illumina 1.3+,
    // 2 Solexa, 3 ignore (default).  Unless ignored, the score of every base is reduced to 4 bits, ((Qphred + 2) * 15) / 40, and
    // travels in bits 4..7 of the base's byte - where the reference keeps it.
    void set_quality_mode(int m) { qmode_ = m; }
    int quality_mode() const { return qmode_; }

private:
    int fill();
    int getc_();
    void ungetc_() { --pos_; }
    gzFile gz_ = nullptr;
    // the next buffer is inflated by a thread of the reader's own while the parser works through the current one: inflate is nine tenths
    // of a gzip'd file's ingest, and the mate files of a paired run inflate side by side
    struct Ahead {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::vector<uint8_t> buf;
        int n = 0;
        bool ready = false, want = false, stop = false;
    };
    std::unique_ptr<Ahead> ahead_;
    void stop_ahead();
    std::vector<uint8_t> buf_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false, fastq_ = false, started_ = false;
    int qmode_ = 3;
    std::vector<uint8_t> qual_;
    std::string path_;
};

// A run of parsed records (file order inside a chunk, chunks in file order): views into the two buffers of a ParsedFile.
struct ParsedChunk {
    uint8_t *bases = nullptr;           // the chunk's bases back to back, Ascii2Sense-mapped: a view into its ParsedFile's buffer
    char *descr = nullptr;              // its descriptors back to back (as SeqReader::next returns them): descr_own's bytes
    std::vector<uint32_t> lens;
    std::vector<uint32_t> descr_lens;
    RawVec<char> descr_own;
    bool continues = false;             // split_records: the chunk's first record (descriptor length 0) goes on with the chunk's in front
};

// A whole file parsed in pieces.  The bases buffer is as large as the file and every piece writes from its own offset in the file on
// (a record's bases are never longer than the record's text), so the pieces need no placement pass: a consumer that keeps the reads
// can take the buffer as it is (the read store of `biokanga align` does).  Descriptors - a small part of a read file, copied on by
// every consumer - are the pieces' own.
struct ParsedFile {
    RawVec<uint8_t> bases;
    std::vector<ParsedChunk> chunks;
};

// (FASTQ files go the same way: pieces cut at records whose four lines check out, scores - unless qmode is 3 - into bits 4..7.)
// Parses a whole plain-text (not gzip'd) FASTA file with `nthreads` threads: the file is mapped, cut at
// record starts (the first '>' of a line) and every piece goes through the same state machine as
// SeqReader::next.  Returns 1 and fills `out` when it handled the file, 0 when the file is not eligible
// (gzip, FASTQ, tiny: use SeqReader), < 0 on error.
// split_records (FASTA only): the pieces are cut at line starts wherever those fall, for files of a few long records - a genome -
// and a piece that starts inside a record says so (ParsedChunk::continues); the read loaders do not ask for this.
int parse_fasta_parallel(const std::string &path, int nthreads, ParsedFile &out, std::string *err, int qmode = 3, bool split_records = false);

// The text a read file holds, in bytes, without reading it: a plain file's size; a bgzip'd file's members' lengths, summed from
// their headers (exact); for any other gzip file the last member's length word - which counts modulo 4 GB - raised by the multiple
// of 4 GB that brings the text nearest to 3.5 times the file's size (an estimate: sizes buffers and the output file ahead of time,
// nothing depends on its being right).  0 when the file cannot be read.
uint64_t text_bytes_estimate(const std::string &path);

// One stream of records, from either source.
class RecordStream {
public:
    int open(const std::string &path, int nthreads, std::string *err);
    void set_quality_mode(int m) { rd_.set_quality_mode(m); }       // before open()
    // 1 = record, 0 = end, < 0 = error; pointers stay valid until the next call
    int next(const char *&d, size_t &dl, const uint8_t *&b, size_t &bl);
    // whole-file parse available: the chunks in file order (then next() need not be used)
    bool parsed() const { return parsed_; }
    std::vector<ParsedChunk> &chunks() { return file_.chunks; }
    ParsedFile &file() { return file_; }

private:
    SeqReader rd_;
    bool parsed_ = false;
    ParsedFile file_;
    size_t ci_ = 0, ri_ = 0, bo_ = 0, dofs_ = 0;
    std::string d_;
    std::vector<uint8_t> b_;
};

}  // namespace bk
