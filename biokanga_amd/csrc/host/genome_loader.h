// genome_loader.h - the sequences `biokanga index` indexes, as kangax takes them in (CreateBioseqSuffixFile / ProcessFastaFile,
// kangax.cpp:545-690,774-926): every FASTA record of at least `min_seq_len` bases becomes an entry, its bases - soft-mask flag off, long
// N runs thinned out (nrun_mutate.h) - followed by an end-of-sequence mark in one concatenated store.  Large plain, bgzip'd and gzip'd
// files are parsed by all threads in pieces that may start inside a record (fasta.h, split_records); other files record by record.
#pragma once
#include <string>
#include <vector>

#include "../sfx_file.h"
#include "cli_common.h"

namespace bkcli {

struct Genome {
    bk::RawVec<uint8_t> seq;
    std::vector<bk::SfxEntry> entries;
    uint32_t n_under = 0;               // records not taken: shorter than min_seq_len
    int whole_files = 0;                // files that went through the all-thread parse (the tests' question)
};

// `files` in the order they are to be taken (the caller sorts them as the reference's glob does).  0, or 1 after the message.
int load_genome(const std::vector<std::string> &files, int min_seq_len, int nthreads, Genome &g);

}  // namespace bkcli
