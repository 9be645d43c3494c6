// fasta.h - FASTA / FASTQ (optionally gzip'd) record reader for the host front end.
// Follows the ingest semantics of the reference's CFasta (libbiokanga/Fasta.cpp:907-1137,1518-1560):
//   * a FASTA descriptor runs from '>' to end of line; '>' anywhere outside a descriptor starts a new one
//   * sequence characters: every non-alphabetic character except '-' is sloughed
//   * CFasta::Ascii2Sense: a/c/g/t/u -> 0..3 | cRptMskFlg(0x08), A/C/G/T/U -> 0..3, '-' -> eBaseInDel(6),
//     everything else -> eBaseN(4)
//   * FASTQ: 4-line records (@id, sequence, +, qualities); qualities are not retained (-g3 default)
#pragma once
#include <zlib.h>

#include <cstdint>
#include <string>
#include <vector>

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

private:
    int fill();
    int getc_();
    void ungetc_() { --pos_; }
    gzFile gz_ = nullptr;
    std::vector<uint8_t> buf_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false, fastq_ = false, started_ = false;
    std::string path_;
};

}  // namespace bk
