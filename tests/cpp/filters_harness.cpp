// drives host/post_filters.h the way the command line does:
//   filters_harness <min_flank> <paired 0|1> <splice 0|1> <indel 0|1> hits.bin seg2.bin bases.bin offs.bin seq.bin ent_start.bin out_hits.bin out_trims.bin
// seq.bin: the index's concatenated target (1 byte/base), ent_start.bin: uint64 start offset per entry (EntryID - 1)
// out_trims.bin: per record uint16 left, uint16 right, uint16 trimmed mismatches
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../biokanga_amd/csrc/host/mtqsort.h"
#include "../../biokanga_amd/csrc/host/post_filters.h"

template <typename T>
static std::vector<T> slurp(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v((size_t)n / sizeof(T));
    if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n) exit(2);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc == 5 && std::string(argv[1]) == "pcr") {
        // filters_harness pcr <win_len> hits.bin out_hits.bin : CAligner::ReducePCRduplicates over the reference's sorted order
        const int win = atoi(argv[2]);
        std::vector<bk_hit> hits = slurp<bk_hit>(argv[3]);
        std::vector<uint32_t> ord(hits.size());
        for (size_t i = 0; i < ord.size(); i++) ord[i] = (uint32_t)i;
        auto cmp = [&](uint32_t x, uint32_t y) -> int {          // CAligner::SortHitMatch
            const bk_hit &p = hits[x], &q = hits[y];
            if (p.nar != q.nar) return p.nar < q.nar ? -1 : 1;
            if (p.num_hits == 1 && q.num_hits != 1) return -1;
            if (p.num_hits != 1 && q.num_hits == 1) return 1;
            if (p.num_hits != 1 && q.num_hits != 1) return p.num_hits < q.num_hits ? -1 : (p.num_hits > q.num_hits ? 1 : 0);
            if (p.chrom_id != q.chrom_id) return p.chrom_id < q.chrom_id ? -1 : 1;
            if (p.match_loci != q.match_loci) return p.match_loci < q.match_loci ? -1 : 1;
            if (p.match_len != q.match_len) return p.match_len < q.match_len ? -1 : 1;
            if (p.strand != q.strand) return p.strand < q.strand ? -1 : 1;
            if (p.low_mm != q.low_mm) return p.low_mm < q.low_mm ? -1 : 1;
            return 0;
        };
        bk::ref_order_sort(ord.data(), (int64_t)ord.size(), cmp, 4);
        bk::reduce_pcr_duplicates(hits, ord, [&](size_t i) { return hits[i].match_loci; }, [&](size_t i) { return (uint32_t)hits[i].match_len; }, win);
        FILE *f = fopen(argv[4], "wb");
        fwrite(hits.data(), sizeof(bk_hit), hits.size(), f);
        fclose(f);
        return 0;
    }
    if (argc != 13) return 2;
    const int min_flank = atoi(argv[1]);
    const bool paired = atoi(argv[2]) != 0, splice = atoi(argv[3]) != 0, indel = atoi(argv[4]) != 0;
    std::vector<bk_hit> hits = slurp<bk_hit>(argv[5]);
    std::vector<bk_seg2> seg2 = slurp<bk_seg2>(argv[6]);
    std::vector<uint8_t> bases = slurp<uint8_t>(argv[7]);
    std::vector<uint64_t> offs = slurp<uint64_t>(argv[8]);
    std::vector<uint8_t> seq = slurp<uint8_t>(argv[9]);
    std::vector<uint64_t> ent_start = slurp<uint64_t>(argv[10]);
    if (seg2.empty()) seg2.assign(hits.size(), bk_seg2{});
    bk::FlankTrims tr;
    if (min_flank > 0)
        bk::auto_trim_flanks(hits, [&](size_t i) { return (seg2[i].flags & 5) != 0; }, [&](size_t i) { return bases.data() + offs[i]; },
                             [&](size_t i) -> const uint8_t * {
                                 const bk_hit &h = hits[i];
                                 return (h.chrom_id >= 1 && h.chrom_id <= ent_start.size()) ? seq.data() + ent_start[h.chrom_id - 1] + h.match_loci : nullptr;
                             },
                             min_flank, paired, 3, tr);
    if (splice) bk::remove_orphan_segs(hits, seg2, 4, 7);
    if (indel) bk::remove_orphan_segs(hits, seg2, 1, 8);
    FILE *f = fopen(argv[11], "wb");
    fwrite(hits.data(), sizeof(bk_hit), hits.size(), f);
    fclose(f);
    f = fopen(argv[12], "wb");
    for (size_t i = 0; i < hits.size(); i++) {
        uint16_t t[3] = {tr.empty() ? (uint16_t)0 : tr.left[i], tr.empty() ? (uint16_t)0 : tr.right[i],
                         tr.empty() ? (uint16_t)hits[i].mismatches : (uint16_t)tr.mismatches[i]};
        fwrite(t, 2, 3, f);
    }
    fclose(f);
    return 0;
}
