// genome_loader.cpp - see genome_loader.h
#include "genome_loader.h"

#include <strings.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

#include "fasta.h"
#include "nrun_mutate.h"

namespace bkcli {

namespace {

// One file's records, one after the other.  The reference's read buffer bookkeeping decides where its 16 M-base chunks fall within a
// record, and the N-run counter starts again at every chunk (kangax.cpp:572,589,626-660); the random bases come from one sequence over
// all files, as from the process-wide rand().
struct Taker {
    Genome &g;
    const std::string &fn;
    int min_seq_len;
    bk::GlibcRand &rnd;
    static constexpr size_t kChunk = 0x00ffffff;        // cMaxAllocBuffChunk, kangax.cpp:37
    size_t allocd = kChunk * 16, avail = kChunk * 16;
    int seq_id = 0;

    // the record whose bases were appended to g.seq from `start` on
    int take(const std::string &descr, size_t start)
    {
        seq_id++;
        char name[256];
        if (sscanf(descr.c_str(), " %255s", name) != 1) snprintf(name, sizeof(name), "%s.%d", fn.c_str(), ++seq_id);
        const size_t len = g.seq.size() - start;
        size_t buff_ofs = 0;
        while (buff_ofs < len) {
            const size_t chunk = std::min(std::min(avail, kChunk), len - buff_ofs);
            bk::mutate_n_runs(g.seq.data() + start + buff_ofs, chunk, rnd);
            buff_ofs += chunk;
            avail -= chunk;
            if (avail < kChunk / 8) {
                allocd += kChunk;
                avail = allocd - buff_ofs;
            }
        }
        if (len < (size_t)min_seq_len) { g.n_under++; g.seq.resize(start); return 0; }
        if (len > 0xfff00000ULL) { diag("AddEntry: SeqLen %zu not in range 1..%u", len, 0xfff00000u); return 1; }
        bk::SfxEntry e;
        e.entry_id = (uint32_t)g.entries.size() + 1;
        e.fblock_id = 1;
        strncpy(e.name, name, 80);
        e.name_hash = bk::gen_hash16(name);
        e.seq_len = (uint32_t)len;
        e.start_ofs = start;
        e.end_ofs = start + len - 1;
        for (const bk::SfxEntry &o : g.entries)
            if (!strcasecmp(o.name, e.name)) { diag("CreateBioseqSuffixFile, duplicate sequence entry name '%s' in file '%s'", e.name, fn.c_str()); return 1; }
        g.entries.push_back(e);
        g.seq.push_back(bk::kBaseEOS);
        return 0;
    }
};

}  // namespace

int load_genome(const std::vector<std::string> &files, int min_seq_len, int nthreads, Genome &g)
{
    bk::GlibcRand nrun_rand;
    uint64_t text = 0;
    for (const std::string &fn : files) text += bk::text_bytes_estimate(fn);
    g.seq.reserve(text + 1024);                                            // (bases and marks are fewer than the files' characters)
    for (const std::string &fn : files) {
        std::string err;
        Taker t{g, fn, min_seq_len, nrun_rand};
        bk::ParsedFile pf;
        const int whole = nthreads > 1 ? bk::parse_fasta_parallel(fn, nthreads, pf, &err, 3, true) : 0;
        if (whole < 0) { diag("ProcessFastaFile: Unable to open '%s' %s", fn.c_str(), err.c_str()); return 1; }
        if (whole == 1) {
            diag("ProcessFastaFile:- Adding %s..", fn.c_str());
            g.whole_files++;
            // the pieces' records in file order; a piece's first record may go on with the last one of the piece in front
            std::string descr;
            size_t start = g.seq.size();
            bool open = false;
            for (const bk::ParsedChunk &c : pf.chunks) {
                size_t bo = 0, dofs = 0;
                for (size_t i = 0; i < c.lens.size(); i++) {
                    const bool goes_on = i == 0 && c.continues;
                    if (!goes_on) {
                        if (open && t.take(descr, start)) return 1;
                        descr.assign(c.descr + dofs, c.descr_lens[i]);
                        start = g.seq.size();
                        open = true;
                    }
                    if (open) g.seq.insert(g.seq.end(), c.bases + bo, c.bases + bo + c.lens[i]);
                    bo += c.lens[i];
                    dofs += c.descr_lens[i];
                }
            }
            if (open && t.take(descr, start)) return 1;
            continue;
        }
        bk::SeqReader rd;
        int rc = rd.open(fn, &err);
        if (rc) { diag("ProcessFastaFile: Unable to open '%s' %s", fn.c_str(), err.c_str()); return 1; }
        diag("ProcessFastaFile:- Adding %s..", fn.c_str());
        std::string d;
        std::vector<uint8_t> bases;
        while ((rc = rd.next(d, bases)) > 0) {
            const size_t start = g.seq.size();
            g.seq.insert(g.seq.end(), bases.begin(), bases.end());
            if (t.take(d, start)) return 1;
        }
        if (rc < 0) { diag("ProcessFastaFile: errors whilst reading '%s'", fn.c_str()); return 1; }
    }
    return 0;
}

}  // namespace bkcli
