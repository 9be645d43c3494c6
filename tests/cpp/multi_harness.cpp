// drives host/multi_assign.h and host/glibc_rand.h the way the command line does for -r2 / -r3 / -r4:
//   multi_harness <mode 2|3|4> <threads> <max_reads_len> <clamp 0|1> hits.bin offs.bin loci.bin out_hits.bin [trims.bin out_trims.bin]
// trims.bin: one bk_loci_trims per locus (-c with the multi-loci modes); out_trims.bin: per read the trims of the locus it took (zeros
// for reads the policy left alone)
// hits.bin: bk_hit records (as bk_align_batch returns them with max_ml > 1), offs.bin: uint64[n+1], loci.bin: bk_loci
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>
#include "../../biokanga_amd/csrc/host/glibc_rand.h"
#include "../../biokanga_amd/csrc/host/multi_assign.h"

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
    if (argc == 3 && std::string(argv[1]) == "sorttest") {          // par_sort against std::sort on pseudo-random keys with many ties
        const size_t n = (size_t)atol(argv[2]);
        std::vector<std::pair<uint32_t, uint32_t>> a(n);
        uint64_t x = 88172645463325252ULL;
        for (size_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; a[i] = {(uint32_t)(x % 5000), (uint32_t)i}; }
        auto b = a;
        auto less = [](const std::pair<uint32_t, uint32_t> &p, const std::pair<uint32_t, uint32_t> &q) { return p < q; };
        std::sort(b.begin(), b.end(), less);
        for (int t : {1, 3, 8, 64}) { auto c = a; bk::par_sort(c, less, t); if (c != b) { printf("mismatch at %d threads\n", t); return 1; } }
        printf("ok\n");
        return 0;
    }
    if (argc != 9 && argc != 11) return 2;
    const int mode = atoi(argv[1]), threads = atoi(argv[2]);
    const uint32_t max_reads_len = (uint32_t)atoi(argv[3]);
    const bool clamp = atoi(argv[4]) != 0;
    std::vector<bk_hit> hits = slurp<bk_hit>(argv[5]);
    std::vector<uint64_t> offs = slurp<uint64_t>(argv[6]);
    std::vector<bk_loci> loci = slurp<bk_loci>(argv[7]);
    const size_t nr = hits.size();
    std::vector<bk_loci_trims> trims, rec_trims(nr, bk_loci_trims{});
    if (argc == 11) trims = slurp<bk_loci_trims>(argv[9]);
    auto trims_of = [&](uint64_t l) { return l < trims.size() ? trims[l] : bk_loci_trims{}; };
    auto count = [&](size_t i) -> uint32_t {
        const bk_hit &h = hits[i];
        if (h.rslt == BK_HR_HITS || (clamp && h.rslt == BK_HR_HITINSTS)) return (uint32_t)(offs[i + 1] - offs[i]);
        return 0;
    };
    auto take = [&](bk_hit &h, const bk_loci &L) {
        h.chrom_id = L.chrom_id; h.match_loci = L.match_loci; h.match_len = L.match_len; h.strand = L.strand;
        h.mismatches = L.mismatches; h.nar = BK_NAR_ACCEPTED; h.num_hits = 1; h.low_hit_instances = 1;
    };
    if (mode == 2) {
        bk::GlibcRand pick;
        for (size_t i = 0; i < nr; i++) {
            const uint32_t c = count(i);
            if (c) {
                const uint64_t l = offs[i] + (uint32_t)pick.next() % c;
                take(hits[i], loci[l]);
                rec_trims[i] = trims_of(l);
            }
        }
    } else {
        bk::MultiAssign ma;
        for (size_t i = 0; i < nr; i++) {
            const uint32_t c = count(i);
            for (uint32_t k = 0; k < c; k++) {
                const bk_loci_trims t = trims_of(offs[i] + k);
                ma.add((uint32_t)i + 1, loci[offs[i] + k], c > 1, t.left, t.right, (uint32_t)(offs[i] + k));
            }
        }
        ma.assign(mode == 3, threads, max_reads_len);
        for (const bk::MultiHitRec &m : ma.recs)
            if (m.multi && m.assigned) { take(hits[m.read_id - 1], m.loci); rec_trims[m.read_id - 1] = trims_of(m.src); }
    }
    FILE *f = fopen(argv[8], "wb");
    fwrite(hits.data(), sizeof(bk_hit), nr, f);
    fclose(f);
    if (argc == 11) {
        f = fopen(argv[10], "wb");
        fwrite(rec_trims.data(), sizeof(bk_loci_trims), nr, f);
        fclose(f);
    }
    return 0;
}
