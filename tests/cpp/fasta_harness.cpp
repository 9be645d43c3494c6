// test harness: parses a FASTA file with the serial SeqReader and with parse_fasta_parallel
// (biokanga_amd/csrc/host/fasta.cpp) and reports whether both give the same records.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include "../../biokanga_amd/csrc/host/fasta.h"

int main(int argc, char **argv)
{
    if (argc != 3 && argc != 4) return 2;
    const int nthreads = atoi(argv[2]);
    std::string err;
    const int qmode = (argc == 4 && argv[3][0] == 'q') ? atoi(argv[3] + 1) : 3;      // "q0" .. "q3": how FASTQ scores are kept (-g)
    if (argc == 4 && !strcmp(argv[3], "est")) {     // the text the file holds, by its own account
        printf("text %llu\n", (unsigned long long)bk::text_bytes_estimate(argv[1]));
        return 0;
    }
    if (argc == 4 && argv[3][0] != 'q') {          // "time": the parallel parser alone, three times
        for (int k = 0; k < 3; k++) {
            timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            bk::ParsedFile pf;
            int h = bk::parse_fasta_parallel(argv[1], nthreads, pf, &err);
            clock_gettime(CLOCK_MONOTONIC, &t1);
            size_t n = 0;
            for (auto &c : pf.chunks) n += c.lens.size();
            printf("parse %d: %.3f s, %zu records, handled %d\n", k, (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec), n, h);
        }
        return 0;
    }
    bk::SeqReader rd;
    rd.set_quality_mode(qmode);
    if (rd.open(argv[1], &err)) { fprintf(stderr, "%s\n", err.c_str()); return 3; }
    bk::RecordStream rs;
    rs.set_quality_mode(qmode);
    if (rs.open(argv[1], nthreads, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 3; }
    bk::ParsedFile probe;
    int handled = bk::parse_fasta_parallel(argv[1], nthreads, probe, &err, qmode);
    std::string d;
    std::vector<uint8_t> b;
    unsigned long n = 0, nbases = 0;
    unsigned long long sum = 1469598103934665603ULL;                      // FNV-1a over every descriptor and base: the same for a file and its gzip'd copy
    for (;;) {
        int r1 = rd.next(d, b);
        const char *pd; const uint8_t *pb; size_t dl, bl;
        int r2 = rs.next(pd, dl, pb, bl);
        if (r1 != r2) { printf("MISMATCH rc %d %d at record %lu\n", r1, r2, n); return 1; }
        if (r1 <= 0) break;
        if (d.size() != dl || memcmp(d.data(), pd, dl) || b.size() != bl || memcmp(b.data(), pb, bl)) {
            printf("MISMATCH at record %lu (descr %zu/%zu bases %zu/%zu)\n", n, d.size(), dl, b.size(), bl);
            return 1;
        }
        n++;
        nbases += bl;
        for (size_t i = 0; i < dl; i++) sum = (sum ^ (unsigned char)pd[i]) * 1099511628211ULL;
        for (size_t i = 0; i < bl; i++) sum = (sum ^ pb[i]) * 1099511628211ULL;
    }
    printf("OK records %lu bases %lu sum %016llx parallel %d pieces %zu\n", n, nbases, sum, handled, probe.chunks.size());
    return 0;
}
