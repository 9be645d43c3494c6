// outbuf_harness - host/cli_common.h OutBuf on a name that ends in ".gz": text put piece by piece, stretches handed over as ready-made gzip
// members (as the SAM writer's threads make them), or nothing at all.   outbuf_harness <out.gz> <lines> ; prints the text's FNV-1a and size
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../biokanga_amd/csrc/host/cli_common.h"

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    const long lines = atol(argv[2]);
    bkcli::OutBuf out;
    out.open(argv[1]);
    if (out.fd < 0) return 3;
    uint64_t h = 1469598103934665603ull, total = 0;
    auto seen = [&](const std::string &s) { for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; } total += s.size(); };
    std::string stretch;
    for (long i = 0; i < lines; i++) {
        char line[128];
        const int n = snprintf(line, sizeof line, "read%ld\t%ld\tchr%ld\t%ld\tACGTACGTTTGACCAGT%ld\n", i, i % 16, i % 7, i * 13, i % 1000);
        const std::string s(line, (size_t)n);
        seen(s);
        if ((i / 50000) % 2 == 0) {                     // this stretch goes through put(), the next one as members made here
            if (!stretch.empty()) { std::vector<uint8_t> m; if (!bkcli::gzip_members(stretch.data(), stretch.size(), m)) return 4; out.put_members(m.data(), m.size()); stretch.clear(); }
            out.put(s);
        } else
            stretch += s;
    }
    if (!stretch.empty()) { std::vector<uint8_t> m; if (!bkcli::gzip_members(stretch.data(), stretch.size(), m)) return 4; out.put_members(m.data(), m.size()); }
    out.close();
    printf("%016llx %llu %d\n", (unsigned long long)h, (unsigned long long)total, out.failed ? 1 : 0);
    return 0;
}
