// read_loader.h - the read store's loaders of `biokanga align`: CAligner::LoadRawReads' acceptance rules (biokanga/Aligner.cpp:10724-11427)
// over single-end files and over the mate files of a paired run.  Plain and bgzip'd files are parsed whole (fasta.h) and accepted by
// all threads; the record-by-record loops take everything else, and decide what is said about a file that does not parse.
#pragma once
#include <string>
#include <vector>

#include "cli_common.h"

namespace bkcli {

extern int g_qual_mode;    // -g: FASTQ scores 0 Sanger, 1 Illumina 1.3+, 2 Solexa, 3 ignored (fasta.h)
extern int g_sample_nth;   // -#: every Nth raw read (or pair) of each file is processed, starting with the first (Aligner.cpp:10943,11027-11033)

extern int g_whole_file_loads;   // files that went through the whole-file parse and the all-thread acceptance (the tests' question)

// 0, or a negative teBSFrsltCodes value after the reference's message
int load_reads(const std::vector<std::string> &files, int trim5, int trim3, int min_len, int max_len, int nthreads, ReadStore &rs);
// mates in lockstep, both must pass the length rules (Aligner.cpp:11080-11130); stored PE1, PE2, PE1, PE2 ..
int load_reads_pe(const std::vector<std::string> &f1, const std::vector<std::string> &f2, int trim5, int trim3, int min_len, int max_len,
                  int nthreads, ReadStore &rs);

}  // namespace bkcli
