// loader_harness - the read store as `biokanga align` loads it (host/read_loader.cpp), printed as counts and a checksum over every
// read's length, name and bases in store order; the log lines go to stderr as the command line prints them.  The test runs it with
// one thread (the record-by-record loops) and with many (whole-file parse, all-thread acceptance) and wants the same of both.
//   loader_harness se|pe <threads> <trim5> <trim3> <minlen> <maxlen> <qmode> <nth> file [file ..]      (pe: mates alternate a1 b1 a2 b2)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../biokanga_amd/csrc/host/read_loader.h"

int main(int argc, char **argv)
{
    if (argc < 10) { fprintf(stderr, "usage\n"); return 2; }
    const bool pe = !strcmp(argv[1], "pe");
    const int nt = atoi(argv[2]), t5 = atoi(argv[3]), t3 = atoi(argv[4]), mn = atoi(argv[5]), mx = atoi(argv[6]);
    bkcli::g_qual_mode = atoi(argv[7]);
    bkcli::g_sample_nth = atoi(argv[8]);
    std::vector<std::string> f1, f2;
    for (int i = 9; i < argc; i++) (pe && ((i - 9) & 1) ? f2 : f1).push_back(argv[i]);
    bkcli::ReadStore rs;
    const int rc = pe ? bkcli::load_reads_pe(f1, f2, t5, t3, mn, mx, nt, rs) : bkcli::load_reads(f1, t5, t3, mn, mx, nt, rs);
    if (rc) { fprintf(stderr, "whole %d\n", bkcli::g_whole_file_loads); printf("rc %d\n", rc); return 0; }
    uint64_t h = 1469598103934665603ull, nb = 0;
    auto mix = [&](const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; for (size_t i = 0; i < n; i++) { h ^= q[i]; h *= 1099511628211ull; } };
    for (size_t i = 0; i < rs.size(); i++) {
        mix(&rs.lens[i], 4);
        mix(rs.name(i), strlen(rs.name(i)) + 1);
        mix(rs.bases.data() + rs.offs[i], rs.lens[i]);
        nb += rs.lens[i];
    }
    fprintf(stderr, "whole %d\n", bkcli::g_whole_file_loads);
    printf("reads %zu bases %llu names %llu sum %016llx\n", rs.size(), (unsigned long long)nb, (unsigned long long)rs.name_bytes(), (unsigned long long)h);
    return 0;
}
