// report_harness - host/report.cpp's report_text() on a made-up result set: the CSV forms (-M0..3) and BED (-M4, with its .ind / .jct
// files), plain or .gz, with a given number of threads.  The records are a function of the seed alone, so the files of any two
// thread counts - and of any two builds - can be compared.   report_harness <fmt> <threads> <records> <seed> <out> [sfx]
//   (an <out> that ends in .bam with fmt 5 / 6: report_bam(), the BAM file and its index)
//   (fmt 1 / 3 print target bases: <sfx> is a .sfx the harness writes first and report_text reads back)
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../biokanga_amd/csrc/host/report.h"
#include "../../biokanga_amd/csrc/sfx_file.h"

using namespace bkcli;

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    const int fmt = atoi(argv[1]), nthreads = atoi(argv[2]);
    const size_t nr = (size_t)atol(argv[3]);
    unsigned seed = (unsigned)atoi(argv[4]);
    auto rnd = [&]() { seed = seed * 1103515245u + 12345u; return (seed >> 8) & 0xffffff; };
    // three sequences
    const uint32_t n_ent = 3, lens[3] = {400000, 90000, 250000};
    std::vector<bk_entry_info> ents(n_ent);
    std::vector<bk::SfxEntry> se(n_ent);
    std::vector<uint8_t> seq;
    for (uint32_t c = 0; c < n_ent; c++) {
        ents[c].entry_id = c + 1; ents[c].seq_len = lens[c]; ents[c].start_ofs = seq.size(); ents[c].end_ofs = seq.size() + lens[c] - 1;
        snprintf(ents[c].name, sizeof ents[c].name, "chr%u", c + 1);
        se[c].entry_id = c + 1; strcpy(se[c].name, ents[c].name); se[c].name_hash = bk::gen_hash16(se[c].name); se[c].seq_len = lens[c];
        se[c].start_ofs = ents[c].start_ofs; se[c].end_ofs = ents[c].end_ofs;
        for (uint32_t i = 0; i < lens[c]; i++) seq.push_back((uint8_t)(rnd() % 97 == 0 ? 4 : rnd() % 4));
        seq.push_back(7);
    }
    Args a;
    a.v["o"] = {argv[5]};
    a.v["t"] = {"title of the run"};
    if (argc > 6) {
        std::vector<uint8_t> sa(seq.size() * 4, 0);
        std::string err;
        if (bk::sfx_write(argv[6], "ds", "ds", "ds", se, seq.data(), seq.size(), sa.data(), 4, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 3; }
        a.v["I"] = {argv[6]};
    }
    ReadStore rs;
    std::vector<bk_hit> hits(nr);
    std::vector<bk_seg2> seg2(nr);
    bk::FlankTrims trims;
    const bool with_trims = (seed & 1) != 0;
    for (size_t i = 0; i < nr; i++) {
        const uint32_t len = 30 + rnd() % 170;
        rs.offs.push_back(rs.bases.size());
        rs.lens.push_back(len);
        for (uint32_t k = 0; k < len; k++) rs.bases.push_back((uint8_t)((rnd() % 53 == 0 ? 4 : rnd() % 4) | (rnd() % 5 == 0 ? 8 : 0)));
        char nm[64];
        const int nl = snprintf(nm, sizeof nm, "read_%zu/%u", i, rnd() % 1000);
        rs.name_ofs.push_back(rs.names.size());
        rs.names.insert(rs.names.end(), nm, nm + nl + 1);
        bk_hit &h = hits[i];
        memset(&h, 0, sizeof h);
        const unsigned kind = rnd() % 10;
        h.nar = kind < 7 ? BK_NAR_ACCEPTED : (uint8_t)(2 + rnd() % 4);
        h.chrom_id = 1 + rnd() % n_ent;
        h.match_len = (uint16_t)len;
        h.match_loci = rnd() % (lens[h.chrom_id - 1] - 2 * len - 2000);
        h.strand = rnd() % 2 ? '+' : '-';
        h.mismatches = (uint8_t)(rnd() % 4);
        bk_seg2 &g = seg2[i];
        memset(&g, 0, sizeof g);
        if (kind == 0 || kind == 1) {                   // a second segment: microInDel (deletion or insertion) or splice junction
            const uint16_t first = (uint16_t)(10 + rnd() % (len - 20));
            g.flags = kind == 0 ? (uint8_t)(rnd() % 2 ? 1 : 3) : 4;
            h.match_len = first;
            g.read_ofs = (uint16_t)(first + ((g.flags & 2) ? 1 + rnd() % 3 : 0));
            g.match_len = (uint16_t)(len - g.read_ofs);
            g.match_loci = h.match_loci + first + ((g.flags & 2) ? 0 : (g.flags & 4) ? 50 + rnd() % 1500 : 1 + rnd() % 5);
            g.mismatches = (uint8_t)(rnd() % 3);
        }
        if (with_trims) {
            const bool plain = !(g.flags & 5);
            trims.left.push_back((uint16_t)(plain ? rnd() % 4 : 0));
            trims.right.push_back((uint16_t)(plain ? rnd() % 4 : 0));
            trims.mismatches.push_back((uint8_t)(rnd() % 3));
        }
    }
    std::vector<uint32_t> order(nr), src;
    for (size_t i = 0; i < nr; i++) order[i] = (uint32_t)i;
    for (size_t i = nr; i > 1; i--) std::swap(order[i - 1], order[rnd() % i]);
    std::vector<int> multi_dist;
    const std::string species = "synthetic";
    Report R{a, rs, hits, ents, species, n_ent, src, seg2, trims, multi_dist, order, 0, 0, 1, fmt, nthreads, 5, 2000, 10000};
    const std::string op = argv[5];
    const int rc = (op.size() > 4 && op.substr(op.size() - 4) == ".bam") ? report_bam(R, op) : report_text(R);     // (as cmd_align picks)
    printf("rc %d\n", rc);
    return rc ? 1 : 0;
}
