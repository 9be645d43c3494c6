// sfx_file.h - reader/writer of the reference's on-disk `.sfx` suffix-array index (host side).
//
// Format (verified against files written by the reference):
//   tsSfxHeaderV3   pack(4), 1224 B   libbiokanga/SfxArrayV2.h:174-187
//   tsSfxBlock      pack(1), 20 B header {BlockID u32, NumEntries u32, ConcatSeqLen u64, SfxElSize u32}
//                   + ConcatSeqLen base bytes (1 B/base, eBaseEOS=7 after each entry)
//                   + ConcatSeqLen * SfxElSize suffix array bytes (4- or 5-byte LE elements)
//                                       libbiokanga/SfxArrayV2.h:97-104, SfxArrayV2.cpp:421-502
//   tsSfxEntriesBlock pack(1), {NumEntries u32, MaxEntries u32} + 111 B per tsSfxEntry
//                                       libbiokanga/SfxArrayV2.h:79-95, SfxArrayV2.cpp:505-548
// File order written by the reference: header, block, entries.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace bk {

constexpr int kMaxNameLen = 81;          // cMaxDatasetSpeciesChrom, commdefs.h:172
constexpr uint8_t kBaseN = 4;            // eBaseN
constexpr uint8_t kBaseEOS = 7;          // eBaseEOS
constexpr uint64_t kThres5ByteEls = 4000000000ULL;   // cThres8ByteSfxEls, SfxArrayV2.h:163

struct SfxEntry {
    uint32_t entry_id = 0;      // 1..n
    uint32_t fblock_id = 1;     // (BlockID & 0xff) | Flags << 8
    char     name[kMaxNameLen] = {0};
    uint16_t name_hash = 0;     // CUtility::GenHash16
    uint32_t seq_len = 0;
    uint64_t start_ofs = 0;
    uint64_t end_ofs = 0;
};

// read-only view of an opened .sfx (file is mmap'ed; seq/sa point into the mapping)
struct SfxFile {
    std::string path;
    std::string dataset, description, title;
    int32_t  version = 0;
    uint32_t attributes = 0;
    uint32_t block_id = 0;
    uint64_t concat_len = 0;
    uint32_t el_size = 0;
    const uint8_t *seq = nullptr;
    const uint8_t *sa = nullptr;
    std::vector<SfxEntry> entries;
    uint64_t tot_seq_len = 0;   // GetTotSeqsLen

    void *map_base = nullptr;
    size_t map_len = 0;
    ~SfxFile();
    SfxFile() = default;
    SfxFile(const SfxFile &) = delete;
    SfxFile &operator=(const SfxFile &) = delete;
};

// returns 0 or a negative teBSFrsltCodes value; err receives a message
int sfx_open(const char *path, SfxFile &out, std::string *err = nullptr);

uint16_t gen_hash16(const char *name);   // CUtility::GenHash16, libbiokanga/Utility.cpp:17-37

// writes header + block + entries exactly as CSfxArrayV3::Finalise does.  sa holds concat_len
// elements of el_size bytes.  entries must have start/end offsets filled in.  Images of 16 M bases and more are written by
// `nthreads` threads, each at its own offsets.
int sfx_write(const char *path, const std::string &dataset, const std::string &description,
              const std::string &title, const std::vector<SfxEntry> &entries, const uint8_t *seq,
              uint64_t concat_len, const uint8_t *sa, uint32_t el_size, std::string *err = nullptr, int nthreads = 1);

}  // namespace bk
