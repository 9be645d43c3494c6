// sfx_harness - bk::sfx_write (the .sfx image `biokanga index` writes) with a given number of writer threads, from bases and suffix-array
// bytes held in files.   sfx_harness <bases.bin> <sa.bin> <el> <threads> <out.sfx> name:len [name:len ..]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../biokanga_amd/csrc/sfx_file.h"

static std::vector<uint8_t> slurp(const char *p)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(p, "rb");
    if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END);
    v.resize((size_t)ftell(f));
    fseek(f, 0, SEEK_SET);
    if (fread(v.data(), 1, v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    std::vector<uint8_t> seq = slurp(argv[1]), sa = slurp(argv[2]);
    const uint32_t el = (uint32_t)atoi(argv[3]);
    std::vector<bk::SfxEntry> entries;
    uint64_t ofs = 0;
    for (int i = 6; i < argc; i++) {
        char *colon = strchr(argv[i], ':');
        *colon = 0;
        bk::SfxEntry e;
        e.entry_id = (uint32_t)entries.size() + 1;
        strncpy(e.name, argv[i], 80);
        e.name_hash = bk::gen_hash16(e.name);
        e.seq_len = (uint32_t)atoll(colon + 1);
        e.start_ofs = ofs;
        e.end_ofs = ofs + e.seq_len - 1;
        ofs += (uint64_t)e.seq_len + 1;
        entries.push_back(e);
    }
    std::string err;
    const int rc = bk::sfx_write(argv[5], "ds", "ds", "ds", entries, seq.data(), seq.size(), sa.data(), el, &err, atoi(argv[4]));
    printf("rc %d %s\n", rc, err.c_str());
    return rc ? 1 : 0;
}
