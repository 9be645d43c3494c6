// nrun_harness - host/nrun_mutate.h against the reference's loop as written (kangax.cpp:626-660), on chunks of bases with runs of N and
// n of every length, placed at the chunks' starts and ends too.   nrun_harness <seed> <rounds>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../biokanga_amd/csrc/host/nrun_mutate.h"

static void as_written(uint8_t *p, size_t chunk, bk::GlibcRand &rnd)
{
    int seq_ns = 0;
    for (size_t k = 0; k < chunk; k++) {
        p[k] &= ~0x08;
        if (p[k] == 4 && (k + 5) < chunk) {
            if (++seq_ns > 25 && p[k + 1] == 4 && p[k + 2] == 4 && p[k + 3] == 4 && p[k + 4] == 4) {
                if (!(seq_ns % 13)) p[k] = (uint8_t)(rnd.next() % 4);
            }
        } else
            seq_ns = 0;
    }
}

int main(int argc, char **argv)
{
    unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 200;
    auto rnd = [&]() { seed = seed * 1103515245u + 12345u; return (seed >> 8) & 0xffffff; };
    bk::GlibcRand ra, rb;                                    // one sequence over all chunks, as the process-wide rand()
    size_t mutated = 0;
    for (int r = 0; r < rounds; r++) {
        const size_t n = r % 7 == 0 ? rnd() % 12 : 1 + rnd() % 5000;
        std::vector<uint8_t> a(n);
        for (size_t i = 0; i < n;) {
            const unsigned kind = rnd() % 10;
            size_t run = kind < 3 ? 1 + rnd() % 200 : 1 + rnd() % 30;
            if (r % 5 == 0 && i == 0) run = 60;              // a run at the chunk's start
            for (size_t j = 0; j < run && i < n; j++, i++)
                a[i] = kind == 0 ? 4 : kind == 1 ? 12 : kind == 2 ? (uint8_t)(rnd() % 3 ? 4 : 12) : (uint8_t)((rnd() % 4) | (rnd() % 2 ? 8 : 0));
        }
        if (r % 3 == 0) for (size_t i = n > 70 ? n - 70 : 0; i < n; i++) a[i] = 4;          // and one to its end
        std::vector<uint8_t> b = a, raw = a;
        as_written(a.data(), n, ra);
        bk::mutate_n_runs(b.data(), n, rb);
        if (a != b || ra.next() != rb.next()) { printf("MISMATCH round %d (chunk of %zu)\n", r, n); return 1; }
        for (size_t i = 0; i < n; i++) mutated += (raw[i] & 0xF7) == 4 && a[i] != 4;
    }
    printf("OK rounds %d mutated %zu\n", rounds, mutated);
    return 0;
}
