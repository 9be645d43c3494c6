// inflate_harness - host/fast_inflate.cpp against zlib on raw deflate streams.
//   inflate_harness check <file.deflate> <file.expected>     the decoder's output must be the expected bytes, all input consumed: "OK n"
//                                                            (or "NO" when it declines; never anything else, never out of bounds)
//   inflate_harness time <file.deflate> <expected size>      the best of seven timed runs of either decoder
//   inflate_harness par <file.deflate> <threads>             inflate_raw_parallel must give inflate_raw's bytes: "OK bytes n pieces p"
//   inflate_harness fuzz <file.deflate> <seed> <rounds> [threads]   damaged copies: the decoder may decline or agree with zlib, nothing else
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../biokanga_amd/csrc/host/fast_inflate.h"

static std::vector<uint8_t> slurp(const char *p)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(p, "rb");
    if (!f) { perror(p); exit(2); }
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static long zlib_raw(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *used)
{
    z_stream z;
    memset(&z, 0, sizeof z);
    inflateInit2(&z, -15);
    z.next_in = const_cast<Bytef *>(in); z.avail_in = (uInt)n; z.next_out = out; z.avail_out = (uInt)cap;
    const int rc = inflate(&z, Z_FINISH);
    const long got = rc == Z_STREAM_END ? (long)z.total_out : -1;
    *used = z.total_in;
    inflateEnd(&z);
    return got;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const std::string mode = argv[1];
    std::vector<uint8_t> in = slurp(argv[2]);
    if (mode == "check") {
        std::vector<uint8_t> want = slurp(argv[3]);
        // exact-size heap blocks (so that a sanitizer sees any step outside), at the exact capacity and at a roomy one
        for (size_t room : {(size_t)0, (size_t)1000}) {
            uint8_t *src = (uint8_t *)malloc(in.size() ? in.size() : 1), *out = (uint8_t *)malloc(want.size() + room + 1);
            memcpy(src, in.data(), in.size());
            size_t used = 0;
            const long n = bk::inflate_raw(src, in.size(), out, want.size() + room, out, &used);
            if (n < 0) { printf("NO\n"); return 0; }
            if ((size_t)n != want.size() || (n > 0 && memcmp(out, want.data(), want.size())) || used != in.size()) { printf("WRONG n %ld want %zu used %zu of %zu\n", n, want.size(), used, in.size()); return 1; }
            if (want.size() > 0) {                       // one byte short: must decline
                const long m = bk::inflate_raw(src, in.size(), out, want.size() - 1, out, &used);
                if (m >= 0) { printf("WRONG: fitted %zu bytes into %zu\n", want.size(), want.size() - 1); return 1; }
            }
            free(src); free(out);
        }
        printf("OK %zu\n", want.size());
        return 0;
    }
    if (mode == "time") {
        const size_t cap = (size_t)atoll(argv[3]);
        std::vector<uint8_t> a(cap + 64), b(cap + 64);
        double best1 = 1e9, best2 = 1e9;
        long n1 = 0, n2 = 0;
        bool same = true;
        for (int r = 0; r < 7; r++) {
            size_t u1 = 0, u2 = 0;
            double t0 = now();
            n1 = bk::inflate_raw(in.data(), in.size(), a.data(), cap, a.data(), &u1);
            double t1 = now();
            n2 = zlib_raw(in.data(), in.size(), b.data(), cap, &u2);
            double t2 = now();
            best1 = std::min(best1, t1 - t0);
            best2 = std::min(best2, t2 - t1);
            same = same && n1 == n2 && u1 == u2 && !memcmp(a.data(), b.data(), (size_t)(n1 > 0 ? n1 : 0));
        }
        printf("best of 7: ours %ld bytes %.3f s %.0f MB/s | zlib %ld bytes %.3f s %.0f MB/s | %s\n", n1, best1, n1 / best1 / 1e6, n2, best2, n2 / best2 / 1e6, same ? "same" : "DIFFERENT");
        return 0;
    }
    if (mode == "par") {                                 // <threads>: the several-thread decoder against the one-thread one
        const int nt = atoi(argv[3]);
        const size_t cap = in.size() * 12 + (1u << 20);
        std::vector<uint8_t> a(cap), b(cap);
        size_t u1 = 0, u2 = 0;
        int pieces = 0;
        double t0 = now();
        const long n1 = bk::inflate_raw(in.data(), in.size(), a.data(), cap, a.data(), &u1);
        double t1 = now();
        const long n2 = bk::inflate_raw_parallel(in.data(), in.size(), b.data(), cap, &u2, nt, &pieces);
        double t2 = now();
        const bool same = n1 == n2 && u1 == u2 && (n1 <= 0 || !memcmp(a.data(), b.data(), (size_t)n1));
        printf("%s bytes %ld pieces %d | one thread %.3f s, %d threads %.3f s\n", same ? "OK" : "WRONG", n1, pieces, t1 - t0, nt, t2 - t1);
        return same ? 0 : 1;
    }
    if (mode == "fuzz") {
        unsigned seed = (unsigned)atoi(argv[3]);
        const int rounds = argc > 4 ? atoi(argv[4]) : 1000;
        const int par = argc > 5 ? atoi(argv[5]) : 0;      // > 0: through inflate_raw_parallel with that many threads
        const size_t cap = 1 << 22;
        std::vector<uint8_t> a(cap), b(cap);
        int declined = 0, agreed = 0, zlib_no = 0;
        for (int r = 0; r < rounds; r++) {
            std::vector<uint8_t> d = in;
            auto rnd = [&]() { seed = seed * 1103515245u + 12345u; return (seed >> 8) & 0xffffff; };
            const int kind = (int)(rnd() % 4);
            if (kind == 0) d[rnd() % d.size()] ^= (uint8_t)(1u << (rnd() % 8));
            else if (kind == 1) d.resize(rnd() % d.size());
            else if (kind == 2) { size_t at = rnd() % d.size(); for (size_t i = at; i < d.size() && i < at + 8; i++) d[i] = (uint8_t)rnd(); }
            else { for (int k = 0; k < 3; k++) d[rnd() % d.size()] = (uint8_t)rnd(); }
            uint8_t *src = (uint8_t *)malloc(d.size() ? d.size() : 1);
            memcpy(src, d.data(), d.size());
            size_t u1 = 0, u2 = 0;
            const long n1 = par ? bk::inflate_raw_parallel(src, d.size(), a.data(), cap, &u1, par) : bk::inflate_raw(src, d.size(), a.data(), cap, a.data(), &u1);
            const long n2 = zlib_raw(src, d.size(), b.data(), cap, &u2);
            free(src);
            if (n2 < 0) zlib_no++;
            if (n1 < 0) { declined++; continue; }
            if (n1 != n2 || u1 != u2 || memcmp(a.data(), b.data(), (size_t)n1)) { printf("WRONG round %d: ours %ld (%zu used) zlib %ld (%zu used)\n", r, n1, u1, n2, u2); return 1; }
            agreed++;
        }
        printf("OK rounds %d declined %d agreed %d zlib refused %d\n", rounds, declined, agreed, zlib_no);
        return 0;
    }
    return 2;
}
