// genome_harness - the sequence store `biokanga index` builds (host/genome_loader.cpp) from the given files with <threads> threads:
// entries, sequences not taken, and a checksum over every entry's fields and the whole store.  The test wants the same of one thread
// (record by record) and of many (pieces cut at line starts, wherever in a record).
//   genome_harness <threads> <minseqlen> file [file ..]
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../biokanga_amd/csrc/host/genome_loader.h"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    std::vector<std::string> files;
    for (int i = 3; i < argc; i++) files.push_back(argv[i]);
    bkcli::Genome g;
    if (bkcli::load_genome(files, atoi(argv[2]), atoi(argv[1]), g)) { printf("failed\n"); return 0; }
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; for (size_t i = 0; i < n; i++) { h ^= q[i]; h *= 1099511628211ull; } };
    for (const bk::SfxEntry &e : g.entries) {
        mix(&e.entry_id, 4); mix(e.name, strlen(e.name) + 1); mix(&e.name_hash, 2); mix(&e.seq_len, 4); mix(&e.start_ofs, 8); mix(&e.end_ofs, 8);
    }
    mix(g.seq.data(), g.seq.size());
    size_t ns = 0;
    for (size_t i = 0; i < g.seq.size(); i++) ns += g.seq[i] == 4;
    fprintf(stderr, "whole %d\n", g.whole_files);
    printf("entries %zu under %u store %zu Ns %zu sum %016llx\n", g.entries.size(), g.n_under, g.seq.size(), ns, (unsigned long long)h);
    return 0;
}
