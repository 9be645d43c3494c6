// test harness (CPU): re-writes a BAM + BAI (CSI when a header sequence reaches 512 Mbp) from the uncompressed BAM stream of a reference-made file through
// biokanga_amd/csrc/host/bam_writer.cpp; the test compares the two files with the reference's.
//   bam_harness <uncompressed.bam.bin> <out.bam> <threads>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../biokanga_amd/csrc/host/bam_writer.h"

int main(int argc, char **argv)
{
    if (argc != 4) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    std::vector<uint8_t> s;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) s.insert(s.end(), buf, buf + n);
    fclose(f);
    auto r32 = [&](size_t o) { uint32_t v; memcpy(&v, s.data() + o, 4); return v; };
    size_t p = 4;
    uint32_t l_text = r32(p); p += 4 + l_text;
    uint32_t n_ref = r32(p); p += 4;
    uint64_t max_ref_len = 0;
    for (uint32_t i = 0; i < n_ref; i++) {
        uint32_t ln = r32(p);
        const uint32_t l_ref = r32(p + 4 + ln);
        if (l_ref > max_ref_len) max_ref_len = l_ref;
        p += 4 + ln + 4;
    }
    std::vector<bk::BamAligned> al;
    uint64_t flush_at = 0;
    while (p < s.size()) {
        uint32_t block = r32(p);
        int32_t ref = (int32_t)r32(p + 4), pos = (int32_t)r32(p + 8);
        uint32_t l_qn = r32(p + 12) & 0xff, n_cig = r32(p + 16) & 0xffff;
        if (ref >= 0) {
            int32_t span = 0;
            for (uint32_t c = 0; c < n_cig; c++) { uint32_t op = r32(p + 36 + l_qn + 4 * c); if ((op & 15) == 0) span += (int32_t)(op >> 4); }
            al.push_back({p, p + 4 + block, ref, pos, pos + span - 1});
            flush_at = p + 4 + block;
        }
        p += 4 + block;
    }
    std::string err;
    int rc = bk::write_bam_and_bai(argv[2], s, al, flush_at, n_ref, max_ref_len, atoi(argv[3]), &err);
    if (rc) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    return 0;
}
