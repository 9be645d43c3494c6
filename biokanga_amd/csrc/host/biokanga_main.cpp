// biokanga_main.cpp - host front end: `biokanga index` and `biokanga align` (C++ above the C ABI).
//
// Keeps the reference's option letters and defaults for the subset that reaches the hot path
// (biokanga/kanga.cpp:194-294, biokanga/kangax.cpp:95-116), the .sfx index format and the SAM text
// the reference writes (libbiokanga/SAMfile.cpp:1521,1573-1575,1766,2110-2283).  The alignment
// itself is ONLY available through libbiokanga_amd (HIP kernels); there is no host fallback.
//
//   biokanga index -i genome.fa [-i more.fa] -o genome.sfx -r name [-l minseqlen] [-d descr] [-t title]
//   biokanga align -i reads.fa[.gz] -I genome.sfx -o out.sam [-s subs] [-e 1|2] [-Q 0|1|2] [-m 0..3]
//                  [-n maxNs] [-l minlen] [-L maxlen] [-y trim5] [-Y trim3] [-M 0|5|6] [-O stats.csv]
//                  [-U 1..4 -u mates.fa -d minins -D maxins [-E]] [-T threads(ignored)] [-F logfile] [--device n]
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>
#include <regex.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/biokanga_amd.h"
#include "../sfx_file.h"
#include "bam_writer.h"
#include "fasta.h"
#include "glibc_rand.h"
#include "mtqsort.h"
#include "multi_assign.h"
#include "post_filters.h"

namespace {

const char *kProgVer = "4.4.2";          // cpszProgVer of the release whose formats are kept (biokanga.cpp:33)
std::string g_proc = "biokanga";          // gszProcName = basename(argv[0]) (used for @PG ID:)
FILE *g_logfile = nullptr;

// CDiagnostics::DiagOut style line: [Www Mmm dd hh:mm:ss.mmm yyyy](proc) text
void diag(const char *fmt, ...)
{
    char msg[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof(msg), fmt, ap);
    va_end(ap);
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    struct tm tmv;
    localtime_r(&tv.tv_sec, &tmv);
    char ts[64], line[4300];
    strftime(ts, sizeof(ts), "%b %e %H:%M:%S", &tmv);
    snprintf(line, sizeof(line), "[%s.%03d %d](%s) %s\n", ts, (int)(tv.tv_usec / 1000), tmv.tm_year + 1900, g_proc.c_str(), msg);
    fputs(line, stdout);
    fflush(stdout);
    if (g_logfile) { fputs(line, g_logfile); fflush(g_logfile); }
}

// ---------------------------------------------------------------------------------------------
// tiny option parser: -x val, -xval, --long=val, --long val; repeated options accumulate
struct Args {
    std::map<std::string, std::vector<std::string>> v;
    bool has(const std::string &k) const { return v.count(k) != 0; }
    std::string str(const std::string &k, const std::string &d = "") const { return has(k) ? v.at(k).back() : d; }
    int num(const std::string &k, int d) const { return has(k) ? atoi(v.at(k).back().c_str()) : d; }
};

bool parse_args(int argc, char **argv, int first, const std::map<std::string, std::string> &longnames,
                const std::string &flags_with_val, const std::string &flags_no_val, Args &out, std::string &err)
{
    for (int i = first; i < argc; i++) {
        std::string a = argv[i];
        if (a.size() >= 2 && a[0] == '-' && a[1] == '-') {
            std::string name = a.substr(2), val;
            bool hasval = false;
            size_t eq = name.find('=');
            if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); hasval = true; }
            auto it = longnames.find(name);
            if (it == longnames.end()) { err = "unknown option --" + name; return false; }
            std::string key = it->second;
            bool wants = flags_with_val.find(key) != std::string::npos || key.size() > 1;
            if (flags_no_val.find(key) != std::string::npos && key.size() == 1) wants = false;
            if (wants && !hasval) {
                if (i + 1 >= argc) { err = "option --" + name + " needs a value"; return false; }
                val = argv[++i];
            }
            out.v[key].push_back(wants ? val : "1");
        } else if (a.size() >= 2 && a[0] == '-') {
            std::string key(1, a[1]);
            if (flags_no_val.find(key) != std::string::npos) { out.v[key].push_back("1"); continue; }
            if (flags_with_val.find(key) == std::string::npos) { err = "unknown option " + a; return false; }
            std::string val = a.substr(2);
            if (val.empty()) {
                if (i + 1 >= argc) { err = "option " + a + " needs a value"; return false; }
                val = argv[++i];
            }
            out.v[key].push_back(val);
        } else {
            err = "unexpected argument '" + a + "'";
            return false;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// index

int cmd_index(int argc, char **argv, int first)
{
    Args a;
    std::string err;
    std::map<std::string, std::string> ln = {{"mode", "m"}, {"minseqlen", "l"}, {"descr", "d"}, {"title", "t"}, {"ref", "r"},
                                             {"threads", "T"}, {"log", "F"}, {"FileLogLevel", "f"}, {"device", "device"}};
    if (!parse_args(argc, argv, first, ln, "mlidtroTFf", "", a, err)) {
        fprintf(stderr, "%s index: %s\n", g_proc.c_str(), err.c_str());
        return 1;
    }
    if (!a.has("i") || !a.has("o") || !a.has("r")) {
        fprintf(stderr, "usage: %s index -i <fasta> [-i <fasta>..] -o <out.sfx> -r <refspecies> [-l minseqlen] [-d descr] [-t title]\n", g_proc.c_str());
        return 1;
    }
    if (a.has("F")) g_logfile = fopen(a.str("F").c_str(), "a");
    if (a.num("m", 0) != 0) { diag("Error: only standard indexing mode '-m0' is supported"); return 1; }
    int min_seq_len = a.has("l") ? a.num("l", 50) : 50;                  // kangax.cpp:275-280
    min_seq_len = std::max(1, std::min(1000000, min_seq_len));
    std::string ref = a.str("r").substr(0, 80);
    std::string title = a.has("t") ? a.str("t").substr(0, 63) : ref.substr(0, 63);
    std::string descr = a.has("d") ? a.str("d").substr(0, 1023) : ref;
    diag("Subprocess index Version %s starting", kProgVer);

    // CreateBioseqSuffixFile / ProcessFastaFile (kangax.cpp:545-690,774-926)
    std::vector<uint8_t> seq;
    std::vector<bk::SfxEntry> entries;
    const size_t kChunk = 0x00ffffff;                                    // cMaxAllocBuffChunk, kangax.cpp:37
    uint32_t n_under = 0;
    bk::GlibcRand nrun_rand;                                             // one sequence over all files, as the process-wide rand()
    std::vector<std::string> files = a.v["i"];
    std::sort(files.begin(), files.end());                               // SG_GLOB_FULLSORT
    for (const std::string &fn : files) {
        bk::SeqReader rd;
        int rc = rd.open(fn, &err);
        if (rc) { diag("ProcessFastaFile: Unable to open '%s' %s", fn.c_str(), err.c_str()); return 1; }
        diag("ProcessFastaFile:- Adding %s..", fn.c_str());
        // the reference's read buffer bookkeeping decides where its 16 M-base chunks fall, and the
        // N-run mutation state is reset at every chunk boundary (kangax.cpp:572,589,626-660)
        size_t allocd = kChunk * 16, avail = allocd;
        std::string d;
        std::vector<uint8_t> bases;
        int seq_id = 0;
        while ((rc = rd.next(d, bases)) > 0) {
            seq_id++;
            char name[256];
            if (sscanf(d.c_str(), " %255s", name) != 1) snprintf(name, sizeof(name), "%s.%d", fn.c_str(), ++seq_id);
            size_t buff_ofs = 0, len = bases.size();
            while (buff_ofs < len) {
                size_t chunk = std::min(std::min(avail, kChunk), len - buff_ofs);
                uint8_t *p = bases.data() + buff_ofs;
                int seq_ns = 0;
                for (size_t k = 0; k < chunk; k++) {
                    p[k] &= ~0x08;
                    if (p[k] == bk::kBaseN && (k + 5) < chunk) {
                        if (++seq_ns > 25 && p[k + 1] == bk::kBaseN && p[k + 2] == bk::kBaseN && p[k + 3] == bk::kBaseN &&
                            p[k + 4] == bk::kBaseN) {
                            if (!(seq_ns % 13)) p[k] = (uint8_t)(nrun_rand.next() % 4);      // the reference's unseeded rand()
                        }
                    } else
                        seq_ns = 0;
                }
                buff_ofs += chunk;
                avail -= chunk;
                if (avail < kChunk / 8) {
                    allocd += kChunk;
                    avail = allocd - buff_ofs;
                }
            }
            if (len < (size_t)min_seq_len) { n_under++; continue; }
            if (len > 0xfff00000ULL) { diag("AddEntry: SeqLen %zu not in range 1..%u", len, 0xfff00000u); return 1; }
            bk::SfxEntry e;
            e.entry_id = (uint32_t)entries.size() + 1;
            e.fblock_id = 1;
            strncpy(e.name, name, 80);
            e.name_hash = bk::gen_hash16(name);
            e.seq_len = (uint32_t)len;
            e.start_ofs = seq.size();
            e.end_ofs = seq.size() + len - 1;
            for (const bk::SfxEntry &o : entries)
                if (!strcasecmp(o.name, e.name)) { diag("CreateBioseqSuffixFile, duplicate sequence entry name '%s' in file '%s'", e.name, fn.c_str()); return 1; }
            entries.push_back(e);
            seq.insert(seq.end(), bases.begin(), bases.end());
            seq.push_back(bk::kBaseEOS);
        }
        if (rc < 0) { diag("ProcessFastaFile: errors whilst reading '%s'", fn.c_str()); return 1; }
    }
    if (n_under) diag("ProcessFastaFile - %u sequences not accepted for indexing as length under %dbp ", n_under, min_seq_len);
    if (entries.empty()) { diag("Nothing to index"); return 1; }

    diag("CreateBioseqSuffixFile: sorting suffix array...");
    uint64_t n = seq.size();
    uint32_t el = n < bk::kThres5ByteEls ? 4 : 5;
    int dev = a.num("device", 0);
    if (bk_device_count() < 1) { diag("Fatal: no HIP device - the suffix sort runs on the GPU only"); return 1; }
    if (hipSetDevice(dev) != hipSuccess) { diag("Fatal: unable to select HIP device %d", dev); return 1; }
    uint8_t *d_seq = nullptr, *d_sa = nullptr;
    if (hipMalloc(&d_seq, n) != hipSuccess || hipMalloc(&d_sa, n * el) != hipSuccess) { diag("Fatal: unable to allocate device memory"); return 1; }
    if (hipMemcpy(d_seq, seq.data(), n, hipMemcpyHostToDevice) != hipSuccess) { diag("Fatal: upload failed"); return 1; }
    int rc = bk_build_sa_device(d_seq, n, d_sa, (int)el, dev);
    if (rc) { diag("Fatal: suffix sort failed: %s", bk_strerror(rc)); return 1; }
    std::vector<uint8_t> sa(n * el);
    if (hipMemcpy(sa.data(), d_sa, n * el, hipMemcpyDeviceToHost) != hipSuccess) { diag("Fatal: download failed"); return 1; }
    (void)hipFree(d_seq);
    (void)hipFree(d_sa);
    rc = bk::sfx_write(a.str("o").c_str(), ref, descr, title, entries, seq.data(), n, sa.data(), el, &err);
    if (rc) { diag("Fatal: %s", err.c_str()); return 1; }
    diag("CreateBioseqSuffixFile: completed...");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// align

const char *kNarTag[20] = {"NA", "AA", "EN", "NL", "MH", "ML", "ET", "OJ", "OM", "DP", "DS", "FC", "PR", "UI", "OI", "UP", "IS", "IT", "NP", "LC"};
const char *kNarDescr[20] = {"Not processed for alignment", "Alignment accepted", "Excessive indeterminate (Ns) bases",
                             "No potential alignment loci", "Mismatch delta (minimum Hamming) criteria not met",
                             "Aligned to multiloci", "Excessively end trimmed", "Aligned as orphaned splice junction",
                             "Aligned as orphaned microInDel", "Duplicate PCR", "Duplicate read sequence",
                             "Aligned to filtered target sequence", "Aligned to a priority region", "PE under minimum insert size",
                             "PE over maximum insert size", "PE partner not aligned", "PE partner aligned to inconsistent strand",
                             "PE partner aligned to different target sequence", "PE alignment not accepted",
                             "Alignment violated loci base constraints"};

struct ReadStore {
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offs;
    std::vector<uint32_t> lens;
    std::vector<char> names;               // '\0' separated
    std::vector<uint64_t> name_ofs;
    size_t size() const { return lens.size(); }
    const char *name(size_t i) const { return names.data() + name_ofs[i]; }
};

// The acceptance rules of load_reads() applied to a file that was parsed whole (fasta.h, ParsedChunk): every
// chunk is filtered and measured by its own thread, a prefix sum gives each chunk its place in the read
// store, and the threads copy their accepted records there.  Same records, same order, same log lines.
int accept_chunks(std::vector<bk::ParsedChunk> &chunks, const std::string &fn, int trim5, int trim3, int min_len, int max_len,
                  int nthreads, ReadStore &rs)
{
    const size_t nc = chunks.size();
    bool sim = false;
    for (const auto &c : chunks)
        if (!c.lens.empty()) {
            size_t dl = std::min<size_t>(c.descr_lens[0], 127);
            sim = dl >= 14 && (!strncmp(c.descr.data(), "lcl|usimreads|", 14) || !strncmp(c.descr.data(), "lcr|usimreads|", 14));
            break;
        }
    struct Tot { uint64_t n_acc = 0, n_bases = 0, n_names = 0, n_under = 0, n_over = 0, n_rec = 0; long bad_at = -1; };
    std::vector<Tot> tot(nc);
    std::vector<std::vector<uint32_t>> keep_name_len(nc);       // per record: accepted name length + 1, or 0 when sloughed
    auto name_len = [&](const char *d, size_t dl) {
        if (dl > 127) dl = 127;
        if (sim) return dl;
        size_t k = 0;
        while (k < 79 && k < dl && !isspace((unsigned char)d[k])) k++;
        return k;
    };
    auto run = [&](auto fn_) {
        std::vector<std::thread> th;
        for (int w = 1; w < nthreads; w++) th.emplace_back([&, w]() { for (size_t c = (size_t)w; c < nc; c += (size_t)nthreads) fn_(c); });
        for (size_t c = 0; c < nc; c += (size_t)nthreads) fn_(c);
        for (auto &t : th) t.join();
    };
    run([&](size_t ci) {
        const bk::ParsedChunk &c = chunks[ci];
        Tot &t = tot[ci];
        auto &kn = keep_name_len[ci];
        kn.assign(c.lens.size(), 0);
        size_t dofs = 0;
        for (size_t i = 0; i < c.lens.size(); i++) {
            const int len = (int)c.lens[i];
            const size_t dl = c.descr_lens[i];
            t.n_rec++;
            if (len < 1 || len > 0x30000) { if (t.bad_at < 0) t.bad_at = (long)i; }
            else if (trim5 + trim3 + min_len > len) t.n_under++;
            else if (trim5 + trim3 + max_len < len) t.n_over++;
            else {
                size_t nl = name_len(c.descr.data() + dofs, dl);
                kn[i] = (uint32_t)nl + 1;
                t.n_acc++;
                t.n_bases += (uint64_t)(len - trim5 - trim3);
                t.n_names += nl + 1;
            }
            dofs += dl;
        }
    });
    // log lines in file order, as the serial loader prints them
    uint64_t n_descr = 0, n_under = 0, n_over = 0, n_acc = 0;
    for (size_t ci = 0; ci < nc; ci++) {
        const bk::ParsedChunk &c = chunks[ci];
        if (tot[ci].bad_at >= 0) { diag("Problem parsing sequence after %llu reads parsed", (unsigned long long)(n_descr + tot[ci].bad_at + 1)); return -63; }
        if ((n_under < 10 && tot[ci].n_under) || (n_over < 10 && tot[ci].n_over))
            for (size_t i = 0; i < c.lens.size(); i++) {
                const int len = (int)c.lens[i];
                if (trim5 + trim3 + min_len > len) { if (++n_under <= 10) diag("Load: under length (%d) sequence in '%s' after end trims has been sloughed..", len, fn.c_str()); }
                else if (trim5 + trim3 + max_len < len) { if (++n_over <= 10) diag("Load: over length (%d) sequence in '%s' after end trims has been sloughed..", len, fn.c_str()); }
            }
        else { n_under += tot[ci].n_under; n_over += tot[ci].n_over; }
        n_descr += tot[ci].n_rec;
        n_acc += tot[ci].n_acc;
    }
    // placement
    std::vector<uint64_t> r0(nc + 1), b0(nc + 1), m0(nc + 1);
    r0[0] = rs.lens.size(); b0[0] = rs.bases.size(); m0[0] = rs.names.size();
    for (size_t ci = 0; ci < nc; ci++) {
        r0[ci + 1] = r0[ci] + tot[ci].n_acc;
        b0[ci + 1] = b0[ci] + tot[ci].n_bases;
        m0[ci + 1] = m0[ci] + tot[ci].n_names;
    }
    rs.lens.resize(r0[nc]); rs.offs.resize(r0[nc]); rs.name_ofs.resize(r0[nc]);
    rs.bases.resize(b0[nc]);
    rs.names.resize(m0[nc]);
    run([&](size_t ci) {
        bk::ParsedChunk &c = chunks[ci];
        const auto &kn = keep_name_len[ci];
        uint64_t r = r0[ci], bo = b0[ci], mo = m0[ci];
        size_t dofs = 0, sofs = 0;
        for (size_t i = 0; i < c.lens.size(); i++) {
            const uint32_t len = c.lens[i];
            if (kn[i]) {
                const uint32_t keep = len - (uint32_t)trim5 - (uint32_t)trim3, nl = kn[i] - 1;
                rs.offs[r] = bo;
                rs.lens[r] = keep;
                memcpy(rs.bases.data() + bo, c.bases.data() + sofs + trim5, keep);
                rs.name_ofs[r] = mo;
                memcpy(rs.names.data() + mo, c.descr.data() + dofs, nl);
                rs.names[mo + nl] = '\0';
                r++; bo += keep; mo += nl + 1;
            }
            dofs += c.descr_lens[i];
            sofs += len;
        }
        bk::ParsedChunk().bases.swap(c.bases);
        bk::ParsedChunk().descr.swap(c.descr);
    });
    diag("Load: %llu reads parsed, %llu accepted, %llu under length, %llu over length from '%s'", (unsigned long long)n_descr,
         (unsigned long long)n_acc, (unsigned long long)n_under, (unsigned long long)n_over, fn.c_str());
    return 0;
}

// CAligner::LoadRawReads (Aligner.cpp:10724-11427): descriptor rule, -y/-Y trims, -l/-L acceptance
int g_qual_mode = 3;
int g_sample_nth = 1;      // -#: every Nth raw read (or pair) of each file is processed, starting with the first (Aligner.cpp:10943,11027-11033)

int load_reads(const std::vector<std::string> &files, int trim5, int trim3, int min_len, int max_len, int nthreads, ReadStore &rs)
{
    for (const std::string &fn : files) {
        bk::RecordStream rd;
        std::string err;
        rd.set_quality_mode(g_qual_mode);
        int rc = rd.open(fn, nthreads, &err);
        if (rc) { diag("Load: %s", err.c_str()); return rc; }
        diag("Loading reads from '%s'", fn.c_str());
        if (rd.parsed() && g_sample_nth <= 1) {
            rc = accept_chunks(rd.chunks(), fn, trim5, trim3, min_len, max_len, nthreads, rs);
            if (rc) return rc;
            continue;
        }
        const char *d;
        const uint8_t *b;
        size_t dl, bl;
        bool sim = false;
        uint32_t n_descr = 0, n_under = 0, n_over = 0, n_acc = 0;
        int nxt_sample = g_sample_nth;
        while ((rc = rd.next(d, dl, b, bl)) > 0) {
            n_descr++;
            if (dl > 127) dl = 127;                                       // cMaxDescrLen-1
            if (n_descr == 1) sim = dl >= 14 && (!strncmp(d, "lcl|usimreads|", 14) || !strncmp(d, "lcr|usimreads|", 14));
            if (g_sample_nth > 1) {
                nxt_sample++;
                if (g_sample_nth > nxt_sample) continue;
                nxt_sample = 0;
            }
            int len = (int)bl;
            if (bl < 1 || bl > 0x30000) { diag("Problem parsing sequence after %u reads parsed", n_descr); return -63; }
            if (trim5 + trim3 + min_len > len) {
                if (++n_under <= 10) diag("Load: under length (%d) sequence in '%s' after end trims has been sloughed..", len, fn.c_str());
                continue;
            }
            if (trim5 + trim3 + max_len < len) {
                if (++n_over <= 10) diag("Load: over length (%d) sequence in '%s' after end trims has been sloughed..", len, fn.c_str());
                continue;
            }
            if (!sim) {                                                   // cut at first whitespace, < cMaxDescrIDLen
                size_t k = 0;
                while (k < 79 && k < dl && !isspace((unsigned char)d[k])) k++;
                dl = k;
            }
            int keep = len - trim5 - trim3;
            rs.offs.push_back(rs.bases.size());
            rs.lens.push_back((uint32_t)keep);
            rs.bases.insert(rs.bases.end(), b + trim5, b + trim5 + keep);
            rs.name_ofs.push_back(rs.names.size());
            rs.names.insert(rs.names.end(), d, d + dl);
            rs.names.push_back('\0');
            n_acc++;
        }
        if (rc < 0) { diag("Load: errors whilst parsing '%s'", fn.c_str()); return rc; }
        diag("Load: %u reads parsed, %u accepted, %u under length, %u over length from '%s'", n_descr, n_acc, n_under, n_over, fn.c_str());
    }
    return 0;
}

// paired end loading: PE1/PE2 records in lockstep, both ends must pass the length acceptance
// (Aligner.cpp:11080-11130); stored interleaved PE1, PE2
int load_reads_pe(const std::vector<std::string> &f1, const std::vector<std::string> &f2, int trim5, int trim3, int min_len, int max_len,
                  int nthreads, ReadStore &rs)
{
    for (size_t k = 0; k < f1.size(); k++) {
        bk::RecordStream rd[2];
        std::string err;
        rd[0].set_quality_mode(g_qual_mode);
        rd[1].set_quality_mode(g_qual_mode);
        int rc = rd[0].open(f1[k], nthreads, &err);
        if (rc) { diag("Load: %s", err.c_str()); return rc; }
        rc = rd[1].open(f2[k], nthreads, &err);
        if (rc) { diag("Load: %s", err.c_str()); return rc; }
        diag("Loading paired end reads from '%s' and '%s'", f1[k].c_str(), f2[k].c_str());
        const char *d[2];
        const uint8_t *b[2];
        size_t dl[2], bl[2];
        bool sim[2] = {false, false};
        uint32_t n_descr = 0, n_under = 0, n_over = 0, n_acc = 0;
        int nxt_sample = g_sample_nth;
        for (;;) {
            int rc1 = rd[0].next(d[0], dl[0], b[0], bl[0]);
            if (rc1 < 0) { diag("Load: errors whilst parsing '%s'", f1[k].c_str()); return rc1; }
            if (rc1 == 0) break;
            int rc2 = rd[1].next(d[1], dl[1], b[1], bl[1]);
            if (rc2 <= 0) { diag("Load: '%s' has fewer reads than '%s'", f2[k].c_str(), f1[k].c_str()); return -63; }
            n_descr++;
            bool skip = false;
            for (int e = 0; e < 2; e++) {
                if (dl[e] > 127) dl[e] = 127;
                if (n_descr == 1) sim[e] = dl[e] >= 14 && (!strncmp(d[e], "lcl|usimreads|", 14) || !strncmp(d[e], "lcr|usimreads|", 14));
                if (bl[e] < 1 || bl[e] > 0x30000) { diag("Problem parsing sequence after %u reads parsed", n_descr); return -63; }
            }
            if (g_sample_nth > 1) {
                nxt_sample++;
                if (g_sample_nth > nxt_sample) continue;
                nxt_sample = 0;
            }
            for (int e = 0; e < 2 && !skip; e++) {
                int len = (int)bl[e];
                if (trim5 + trim3 + min_len > len) { n_under++; skip = true; }
                else if (trim5 + trim3 + max_len < len) { n_over++; skip = true; }
            }
            if (skip) continue;
            for (int e = 0; e < 2; e++) {
                if (!sim[e]) {
                    size_t q = 0;
                    while (q < 79 && q < dl[e] && !isspace((unsigned char)d[e][q])) q++;
                    dl[e] = q;
                }
                int keep = (int)bl[e] - trim5 - trim3;
                rs.offs.push_back(rs.bases.size());
                rs.lens.push_back((uint32_t)keep);
                rs.bases.insert(rs.bases.end(), b[e] + trim5, b[e] + trim5 + keep);
                rs.name_ofs.push_back(rs.names.size());
                rs.names.insert(rs.names.end(), d[e], d[e] + dl[e]);
                rs.names.push_back('\0');
            }
            n_acc++;
        }
        diag("Load: %u pairs parsed, %u accepted, %u under length, %u over length", n_descr, n_acc, n_under, n_over);
    }
    return 0;
}

struct OutBuf {
    int fd = -1;
    gzFile gz = nullptr;                    // set when the name ends in ".gz" (CAligner::FileReqWriteCompr, Aligner.cpp:4337)
    off_t pos = 0;                          // file offset of the next byte (everything goes through pwrite)
    std::vector<char> b;
    void open(const char *path)
    {
        size_t n = strlen(path);
        fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd >= 0 && n > 3 && !strcasecmp(path + n - 3, ".gz")) gz = gzdopen(fd, "wb");
        b.reserve(8 << 20);
        pos = 0;
    }
    void put(const char *s, size_t n) { b.insert(b.end(), s, s + n); if (b.size() > (4u << 20)) flush(); }
    void put(const std::string &s) { put(s.data(), s.size()); }
    void flush()
    {
        size_t o = 0;
        if (gz) { if (!b.empty()) gzwrite(gz, b.data(), (unsigned)b.size()); b.clear(); return; }
        while (o < b.size()) { ssize_t w = ::pwrite(fd, b.data() + o, b.size() - o, pos + (off_t)o); if (w <= 0) break; o += (size_t)w; }
        pos += (off_t)o;
        b.clear();
    }
    void close() { flush(); if (gz) { gzclose(gz); gz = nullptr; fd = -1; } if (fd >= 0) { fsync(fd); ::close(fd); } fd = -1; }
};

// ---------------------------------------------------------------------------------------------
// Everything the reporting functions need about one `align` run (references into cmd_align's state)
struct Report {
    const Args &a;
    ReadStore &rs;
    std::vector<bk_hit> &hits;                    // one per record (a read, or a reported locus in -r5)
    const std::vector<bk_entry_info> &ents;
    const std::string &species;
    uint32_t n_ent;
    const std::vector<uint32_t> &src;             // -r5: record -> read
    const std::vector<bk_seg2> &seg2;             // per read: second segment / chimeric trims
    const bk::FlankTrims &trims;                  // per record
    const std::vector<int> &multi_dist;
    const std::vector<uint32_t> &order;           // records in the reference's output order
    int pe_mode, ml_mode, max_ml, fmt, nthreads, micro_indel, splice_len, max_rpt_sam_seqs;

    size_t RD(size_t i) const { return src.empty() ? i : (size_t)src[i]; }
    bool has_seg2(size_t i) const { return !seg2.empty() && (seg2[RD(i)].flags & 5); }       // FlgInDel or FlgSplice
    uint32_t TL(size_t i) const { return trims.empty() ? 0u : trims.left[i]; }
    uint32_t TR(size_t i) const { return trims.empty() ? 0u : trims.right[i]; }
    uint32_t a_start(const bk_hit &h, size_t i) const { return h.match_loci + (h.strand == '+' ? TL(i) : TR(i)); }       // AdjStartLoci
    uint32_t a_len(const bk_hit &h, size_t i) const { return (uint32_t)h.match_len - TL(i) - TR(i); }                      // AdjHitLen
    uint32_t a_mm(const bk_hit &h, size_t i) const { return trims.empty() ? h.mismatches : trims.mismatches[i]; }         // TrimMismatches
};

    // -j / -J: reads that found no alignment at all (NAR EN, NL) / multi-loci reads (NAR ML) as FASTA, in the sorted
    // order, 70 columns (CAligner::ReportNoneAligned / ReportMultiAlign, Aligner.cpp:3826-4010)
void report_read_subset(Report &R, const char *opt, const char *tag, bool (*want)(uint8_t))
{
    [[maybe_unused]] const Args &a = R.a;
    [[maybe_unused]] auto &hits = R.hits;
    [[maybe_unused]] auto &rs = R.rs;
    [[maybe_unused]] auto &ents = R.ents;
    [[maybe_unused]] const std::string &species = R.species;
    [[maybe_unused]] const uint32_t n_ent = R.n_ent;
    [[maybe_unused]] const size_t nr = R.hits.size();
    [[maybe_unused]] const auto &order = R.order;
    [[maybe_unused]] const auto &seg2 = R.seg2;
    [[maybe_unused]] const auto &multi_dist = R.multi_dist;
    [[maybe_unused]] const int pe_mode = R.pe_mode, ml_mode = R.ml_mode, max_ml = R.max_ml, fmt = R.fmt, nthreads = R.nthreads, micro_indel = R.micro_indel,
                               splice_len = R.splice_len, max_rpt_sam_seqs = R.max_rpt_sam_seqs;
    [[maybe_unused]] auto RD = [&](size_t i) -> size_t { return R.RD(i); };
    [[maybe_unused]] auto has_seg2 = [&](size_t i) -> bool { return R.has_seg2(i); };
    [[maybe_unused]] auto TL = [&](size_t i) -> uint32_t { return R.TL(i); };
    [[maybe_unused]] auto TR = [&](size_t i) -> uint32_t { return R.TR(i); };
    [[maybe_unused]] auto a_start = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_start(h, i); };
    [[maybe_unused]] auto a_len = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_len(h, i); };
    [[maybe_unused]] auto a_mm = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_mm(h, i); };
    if (!a.has(opt) || ml_mode == 5) return;                         // kanga.cpp:1045-1066
    OutBuf o;
    o.open(a.str(opt).c_str());
    if (o.fd < 0) { diag("Unable to create '%s'", a.str(opt).c_str()); return; }
    static const char up[8] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N'};
    std::string rec;
    for (size_t k = 0; k < nr; k++) {
        const uint32_t i = order[k];
        if (!want(hits[i].nar)) continue;
        const uint32_t len = rs.lens[RD(i)];
        const uint8_t *sq = rs.bases.data() + rs.offs[RD(i)];
        char hd[400];
        int n = snprintf(hd, sizeof(hd), ">lcl|%s|%u %s %u|1|%u\n", tag, i + 1, rs.name(RD(i)), i + 1, len);
        rec.assign(hd, (size_t)n);
        for (uint32_t q = 0; q < len; q++) {
            rec.push_back(up[sq[q] & 7]);
            if ((q + 1) % 70 == 0 || q + 1 == len) rec.push_back('\n');
        }
        o.put(rec);
    }
    o.close();
}

    // -O: CAligner::ProcessPairedEnds' insert length table (PE only, Aligner.cpp:3024-3040), WriteBasicCountStats
    // (:4186-4330, fed by WriteSubDist :6275-6336 for every accepted read) and ReportTargHitCnts (:5475-5537)
void report_stats(Report &R)
{
    [[maybe_unused]] const Args &a = R.a;
    [[maybe_unused]] auto &hits = R.hits;
    [[maybe_unused]] auto &rs = R.rs;
    [[maybe_unused]] auto &ents = R.ents;
    [[maybe_unused]] const std::string &species = R.species;
    [[maybe_unused]] const uint32_t n_ent = R.n_ent;
    [[maybe_unused]] const size_t nr = R.hits.size();
    [[maybe_unused]] const auto &order = R.order;
    [[maybe_unused]] const auto &seg2 = R.seg2;
    [[maybe_unused]] const auto &multi_dist = R.multi_dist;
    [[maybe_unused]] const int pe_mode = R.pe_mode, ml_mode = R.ml_mode, max_ml = R.max_ml, fmt = R.fmt, nthreads = R.nthreads, micro_indel = R.micro_indel,
                               splice_len = R.splice_len, max_rpt_sam_seqs = R.max_rpt_sam_seqs;
    [[maybe_unused]] auto RD = [&](size_t i) -> size_t { return R.RD(i); };
    [[maybe_unused]] auto has_seg2 = [&](size_t i) -> bool { return R.has_seg2(i); };
    [[maybe_unused]] auto TL = [&](size_t i) -> uint32_t { return R.TL(i); };
    [[maybe_unused]] auto TR = [&](size_t i) -> uint32_t { return R.TR(i); };
    [[maybe_unused]] auto a_start = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_start(h, i); };
    [[maybe_unused]] auto a_len = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_len(h, i); };
    [[maybe_unused]] auto a_mm = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_mm(h, i); };
    if (!a.has("O")) return;
    FILE *f = fopen(a.str("O").c_str(), "w");
    if (!f) { diag("Unable to create '%s'", a.str("O").c_str()); return; }
    if (pe_mode) {
        std::vector<int> len_dist(100001, 0);                    // cPairMaxLen + 1
        for (size_t i = 0; i + 1 < nr; i += 2) {
            const bk_hit &p1 = hits[i], &p2 = hits[i + 1];
            if (!((p1.flags & 0x80) && (p2.flags & 0x80))) continue;
            long s1 = p1.match_loci, e1 = s1 + p1.match_len - 1, s2 = p2.match_loci, e2 = s2 + p2.match_len - 1;
            long frag = p1.strand == '+' ? 1 + e2 - s1 : 1 + e1 - s2;
            if (frag >= 0 && frag <= 100000) len_dist[(size_t)frag]++;
        }
        for (int i = 0; i <= 100000; i++) fprintf(f, "%d,%d\n", i, len_dist[(size_t)i]);
    }
    size_t n_acc = 0;
    uint32_t max_len = 0;
    for (size_t i = 0; i < nr; i++) if (hits[i].nar == BK_NAR_ACCEPTED) { n_acc++; max_len = std::max(max_len, rs.lens[RD(i)]); }
    if (n_acc && max_len) {
        bk::SfxFile sf;
        std::string serr;
        if (bk::sfx_open(a.str("I").c_str(), sf, &serr) == 0) {
            // per read position: accepted reads covering it, and those whose base differs from the target there
            // (read orientation; the target is reverse complemented for '-' alignments); no qualities are
            // loaded, so everything falls into the lowest Phred band
            // qi / sb hold the four Phred bands back to back (band * max_len + position)
            std::vector<std::vector<uint64_t>> qi((size_t)nthreads, std::vector<uint64_t>((size_t)max_len * 4, 0)), sb(qi),
                ms((size_t)nthreads, std::vector<uint64_t>(max_len, 0));
            auto work = [&](int w) {
                auto &Q = qi[(size_t)w], &S = sb[(size_t)w], &M = ms[(size_t)w];
                for (size_t i = (size_t)w; i < nr; i += (size_t)nthreads) {
                    const bk_hit &h = hits[i];
                    if (h.nar != BK_NAR_ACCEPTED || h.chrom_id < 1 || h.chrom_id > n_ent || has_seg2(i)) continue;     // FlagSegs reads are sloughed (:6286)
                    const uint8_t *rd = rs.bases.data() + rs.offs[RD(i)];
                    const uint32_t len = rs.lens[RD(i)];
                    const uint8_t *tg = sf.seq + ents[h.chrom_id - 1].start_ofs + a_start(h, i);
                    const uint32_t alen = a_len(h, i), tl0 = TL(i);
                    uint32_t nsub = 0;
                    for (uint32_t k = 0; k < alen && tl0 + k < len; k++) {      // read positions TrimLeft .. ReadLen - TrimRight (:6303-6306)
                        uint8_t t = h.strand == '-' ? tg[alen - 1 - k] & 7 : tg[k] & 7;
                        if (h.strand == '-' && t < 4) t = (uint8_t)(3 - t);
                        const uint32_t q4 = (rd[tl0 + k] >> 4) & 15;           // 4-bit score -> band (WriteSubDist :6309-6320)
                        const size_t at = (size_t)(q4 <= 3 ? 0 : q4 <= 7 ? 1 : q4 <= 11 ? 2 : 3) * max_len + tl0 + k;
                        Q[at]++;
                        if ((rd[tl0 + k] & 7) != t) { S[at]++; nsub++; }
                    }
                    M[nsub < max_len ? nsub : max_len - 1]++;
                }
            };
            std::vector<std::thread> th;
            for (int w = 1; w < nthreads; w++) th.emplace_back(work, w);
            work(0);
            for (auto &t : th) t.join();
            for (int w = 1; w < nthreads; w++)
                for (size_t k = 0; k < (size_t)max_len * 4; k++) { qi[0][k] += qi[(size_t)w][k]; sb[0][k] += sb[(size_t)w][k]; if (k < max_len) ms[0][k] += ms[(size_t)w][k]; }
            static const char *band_a[4] = {"Phred 0..9", "Phred 10..19", "Phred 20..29", "Phred 30+"};
            static const char *band_b[4] = {"Phred 0..8", "Phred 9..19", "Phred 20..29", "Phred 30+"};
            if (ml_mode) {                                           // WriteBasicCountStats, Aligner.cpp:4203-4227
                fprintf(f, "\"Multihit distribution\",");
                for (int k = 0; k < max_ml; k++) fprintf(f, ",%d", k + 1);
                fprintf(f, "\n,\"Instances\"");
                for (int k = 0; k < max_ml; k++) fprintf(f, ",%d", multi_dist[(size_t)k]);
                fprintf(f, "\n");
            }
            fprintf(f, "\"Phred Score Instances\",");
            for (uint32_t k = 0; k < max_len; k++) fprintf(f, ",%u", k + 1);
            for (int bnd = 0; bnd < 4; bnd++) {
                fprintf(f, "\n,\"%s\"", band_a[bnd]);
                for (uint32_t k = 0; k < max_len; k++) fprintf(f, ",%llu", (unsigned long long)qi[0][(size_t)bnd * max_len + k]);
            }
            fprintf(f, "\n\n\"Aligner Induced Subs\",");
            for (uint32_t k = 0; k < max_len; k++) fprintf(f, ",%u", k + 1);
            for (int bnd = 0; bnd < 4; bnd++) {
                fprintf(f, "\n,\"%s\"", band_b[bnd]);
                for (uint32_t k = 0; k < max_len; k++) fprintf(f, ",%llu", (unsigned long long)sb[0][(size_t)bnd * max_len + k]);
            }
            fprintf(f, "\n\n\"Multiple substitutions\",");
            for (uint32_t k = 0; k < max_len; k++) fprintf(f, ",%u", k);
            fprintf(f, "\n,\"Instances\"");
            for (uint32_t k = 0; k < max_len; k++) fprintf(f, ",%llu", (unsigned long long)ms[0][k]);
            fprintf(f, "\n");
        } else
            diag("Unable to reopen '%s' for the substitution statistics: %s", a.str("I").c_str(), serr.c_str());
        std::vector<uint64_t> cnt(n_ent, 0);                  // final NAR (after any PE processing), in entry order
        for (size_t i = 0; i < nr; i++)
            if (hits[i].nar == BK_NAR_ACCEPTED && hits[i].chrom_id >= 1 && hits[i].chrom_id <= n_ent) cnt[hits[i].chrom_id - 1]++;
        fprintf(f, "\"TargSeq\",\"TargLen\",\"NumHits\"\n");
        for (uint32_t c = 0; c < n_ent; c++)
            if (cnt[c]) fprintf(f, "\"%s\",%u,%llu\n", ents[c].name, ents[c].seq_len, (unsigned long long)cnt[c]);
    }
    fclose(f);
}

    // -A with SAM / BAM output: the junctions are still reported as BED lines in "<out>.jct" (Aligner.cpp:713-721,4440-4462); the track
    // title is empty in these modes
void report_jct_for_sam(Report &R)
{
    [[maybe_unused]] const Args &a = R.a;
    [[maybe_unused]] auto &hits = R.hits;
    [[maybe_unused]] auto &rs = R.rs;
    [[maybe_unused]] auto &ents = R.ents;
    [[maybe_unused]] const std::string &species = R.species;
    [[maybe_unused]] const uint32_t n_ent = R.n_ent;
    [[maybe_unused]] const size_t nr = R.hits.size();
    [[maybe_unused]] const auto &order = R.order;
    [[maybe_unused]] const auto &seg2 = R.seg2;
    [[maybe_unused]] const auto &multi_dist = R.multi_dist;
    [[maybe_unused]] const int pe_mode = R.pe_mode, ml_mode = R.ml_mode, max_ml = R.max_ml, fmt = R.fmt, nthreads = R.nthreads, micro_indel = R.micro_indel,
                               splice_len = R.splice_len, max_rpt_sam_seqs = R.max_rpt_sam_seqs;
    [[maybe_unused]] auto RD = [&](size_t i) -> size_t { return R.RD(i); };
    [[maybe_unused]] auto has_seg2 = [&](size_t i) -> bool { return R.has_seg2(i); };
    [[maybe_unused]] auto TL = [&](size_t i) -> uint32_t { return R.TL(i); };
    [[maybe_unused]] auto TR = [&](size_t i) -> uint32_t { return R.TR(i); };
    [[maybe_unused]] auto a_start = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_start(h, i); };
    [[maybe_unused]] auto a_len = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_len(h, i); };
    [[maybe_unused]] auto a_mm = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_mm(h, i); };
    if (!splice_len || fmt < 5) return;
    std::string jp = a.str("o");
    if (jp.size() > 3 && !strcasecmp(jp.c_str() + jp.size() - 3, ".gz")) jp.resize(jp.size() - 3);
    OutBuf j;
    j.open((jp + ".jct").c_str());
    if (j.fd < 0) { diag("Unable to create '%s.jct'", jp.c_str()); return; }
    char ln[1024];
    int m = snprintf(ln, sizeof(ln), "track type=bed name=\"JCT_\" description=\"\"\n");
    j.put(ln, (size_t)m);
    for (size_t k = 0; k < nr; k++) {
        const uint32_t i = order[k];
        const bk_hit &h = hits[i];
        if (h.nar != BK_NAR_ACCEPTED || !has_seg2(i) || !(seg2[RD(i)].flags & 4)) continue;
        const bk_seg2 &g = seg2[RD(i)];
        const uint32_t end1 = g.match_loci + g.match_len;
        m = snprintf(ln, sizeof(ln), "%s\t%u\t%u\tarj\t0\t%c\t%u\t%u\t0\t2\t%u,%u\t0,%u\n", ents[h.chrom_id - 1].name, h.match_loci, end1, (char)h.strand,
                     h.match_loci, end1, (unsigned)h.match_len, (unsigned)g.match_len, g.match_loci - h.match_loci);
        j.put(ln, (size_t)m);
    }
    j.close();
}

    // ".bam" (more than 5 characters of name, kanga.cpp:848-857): BGZF-compressed BAM with its BAI index
int report_bam(Report &R, const std::string &opath)
{
    [[maybe_unused]] const Args &a = R.a;
    [[maybe_unused]] auto &hits = R.hits;
    [[maybe_unused]] auto &rs = R.rs;
    [[maybe_unused]] auto &ents = R.ents;
    [[maybe_unused]] const std::string &species = R.species;
    [[maybe_unused]] const uint32_t n_ent = R.n_ent;
    [[maybe_unused]] const size_t nr = R.hits.size();
    [[maybe_unused]] const auto &order = R.order;
    [[maybe_unused]] const auto &seg2 = R.seg2;
    [[maybe_unused]] const auto &multi_dist = R.multi_dist;
    [[maybe_unused]] const int pe_mode = R.pe_mode, ml_mode = R.ml_mode, max_ml = R.max_ml, fmt = R.fmt, nthreads = R.nthreads, micro_indel = R.micro_indel,
                               splice_len = R.splice_len, max_rpt_sam_seqs = R.max_rpt_sam_seqs;
    [[maybe_unused]] auto RD = [&](size_t i) -> size_t { return R.RD(i); };
    [[maybe_unused]] auto has_seg2 = [&](size_t i) -> bool { return R.has_seg2(i); };
    [[maybe_unused]] auto TL = [&](size_t i) -> uint32_t { return R.TL(i); };
    [[maybe_unused]] auto TR = [&](size_t i) -> uint32_t { return R.TR(i); };
    [[maybe_unused]] auto a_start = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_start(h, i); };
    [[maybe_unused]] auto a_len = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_len(h, i); };
    [[maybe_unused]] auto a_mm = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_mm(h, i); };
    int rc = 0;
    std::vector<uint8_t> has_hit(n_ent + 1, 0);
    for (const bk_hit &h : hits) if (h.nar == BK_NAR_ACCEPTED && h.chrom_id <= n_ent) has_hit[h.chrom_id] = 1;
    const bool all = (uint32_t)max_rpt_sam_seqs >= n_ent;
    std::string text = "@HD\tVN:1.4\tSO:coordinate";
    std::vector<int32_t> ref_of(n_ent + 1, -1);
    std::vector<uint32_t> refs;
    int n_with = 0;
    char tmp[512];
    for (uint32_t c = 1; c <= n_ent; c++) {
        if (!has_hit[c] && !all) continue;
        int n = snprintf(tmp, sizeof(tmp), "\n@SQ\tAS:%s\tSN:%s\tLN:%u", species.empty() ? "NA" : species.c_str(), ents[c - 1].name, ents[c - 1].seq_len);
        text.append(tmp, (size_t)n);
        ref_of[c] = (int32_t)refs.size();
        refs.push_back(c);
        n_with += has_hit[c];
    }
    int n = snprintf(tmp, sizeof(tmp), "\n@PG\tID:%s\tVN:%s\n", g_proc.c_str(), kProgVer);
    text.append(tmp, (size_t)n);
    diag("Header written with references to %d sequences of which %d have at least 1 alignments", (int)refs.size(), n_with);
    std::vector<uint8_t> stream;
    auto p32 = [](std::vector<uint8_t> &v, uint32_t x) { v.insert(v.end(), (uint8_t *)&x, (uint8_t *)&x + 4); };
    stream.insert(stream.end(), {'B', 'A', 'M', 1});
    p32(stream, (uint32_t)text.size());
    stream.insert(stream.end(), text.begin(), text.end());
    p32(stream, (uint32_t)refs.size());
    for (uint32_t c : refs) {
        uint32_t ln = (uint32_t)strlen(ents[c - 1].name) + 1;
        p32(stream, ln);
        stream.insert(stream.end(), ents[c - 1].name, ents[c - 1].name + ln);
        p32(stream, ents[c - 1].seq_len);
    }
    // records (CAligner::ReportBAMread, Aligner.cpp:5768-6126; CSAMfile::AddAlignment, SAMfile.cpp:2283-2540),
    // formatted in stripes of the sorted order by all host threads
    static const uint8_t code4[8] = {1, 2, 4, 8, 15, 15, 15, 15}, comp4[8] = {8, 4, 2, 1, 15, 15, 15, 15};
    struct Stripe { std::vector<uint8_t> bytes; std::vector<bk::BamAligned> al; uint64_t n = 0; };
    const size_t per_thread = 32768;
    const size_t n_stripes = (nr + per_thread - 1) / per_thread;
    std::vector<Stripe> stripes(n_stripes);
    auto format_stripe = [&](size_t si) {
        Stripe &S = stripes[si];
        const size_t lo = si * per_thread, hi = std::min(nr, lo + per_thread);
        std::vector<uint8_t> &v = S.bytes;
        for (size_t k = lo; k < hi; k++) {
            const uint32_t i = order[k];
            const bk_hit &h = hits[i];
            const bool acc = h.nar == BK_NAR_ACCEPTED;
            if (!acc && fmt != 6) continue;
            const uint8_t *sq = rs.bases.data() + rs.offs[RD(i)];
            const uint32_t len = rs.lens[RD(i)];
            int flag = 0, tlen = 0;
            long pnext = -1;
            if (!pe_mode) flag = acc ? (h.strand == '+' ? 0 : 16) : 4;
            else {
                const bool first_of_pair = (i & 1) == 0;
                const bk_hit &m = hits[first_of_pair ? i + 1 : i - 1];
                flag = 0x1 | 0x2 | (first_of_pair ? 0x40 : 0x80);
                flag |= acc ? (h.strand == '+' ? 0 : 0x10) : 0x4;
                if ((h.flags & 0x80) && (m.flags & 0x80) && m.nar == BK_NAR_ACCEPTED) {
                    flag |= m.strand == '+' ? 0 : 0x20;
                    if (acc) {
                        const size_t mi = first_of_pair ? i + 1 : i - 1;
                        pnext = (long)a_start(m, mi);
                        long s0 = (long)a_start(h, i), s1 = (long)a_start(m, mi);
                        tlen = (int)(s0 <= s1 ? (s1 - s0) + (long)a_len(m, mi) : (s0 - s1) + (long)a_len(h, i));
                    }
                } else
                    flag |= 0x8;
            }
            const char *qn = rs.name(RD(i));
            const uint32_t l_qn = (uint32_t)strlen(qn) + 1;
            const char *tag = acc ? nullptr : kNarTag[h.nar < 20 ? h.nar : 0];
            const uint32_t aux = tag ? 3 + (uint32_t)strlen(tag) + 1 : 0;
            const bool two = acc && has_seg2(i);
            // soft clips in target order: read orientation for '+', swapped for '-' (Aligner.cpp:5961-5984)
            const uint32_t clip5 = acc ? (h.strand == '+' ? TL(i) : TR(i)) : 0u, clip3 = acc ? (h.strand == '+' ? TR(i) : TL(i)) : 0u;
            const uint32_t n_cig = (two ? 3u : 1u) + (clip5 ? 1u : 0u) + (clip3 ? 1u : 0u);
            const uint32_t pos0 = acc ? a_start(h, i) : 0u;
            const uint32_t hit_len = acc ? a_len(h, i) + (two ? seg2[RD(i)].match_len : 0u) : 0u;      // AdjAlignHitLen
            const uint32_t block = 32 + l_qn + 4 * n_cig + (len + 1) / 2 + len + aux;
            const size_t at = v.size();
            v.resize(at + 4 + block);
            uint8_t *q = v.data() + at;
            auto w32 = [&](uint32_t x) { memcpy(q, &x, 4); q += 4; };
            w32(block);
            const int32_t ref = acc ? ref_of[h.chrom_id] : -1;
            w32((uint32_t)ref);
            w32(acc ? pos0 : 0xFFFFFFFFu);
            const uint32_t bin = acc ? (uint32_t)bk::bam_reg2bin((int)pos0, (int)(pos0 + hit_len)) : 0u;
            w32(bin << 16 | 255u << 8 | l_qn);
            w32((uint32_t)flag << 16 | n_cig);
            w32(len);
            w32(acc && pnext >= 0 ? (uint32_t)ref : 0xFFFFFFFFu);
            w32(acc ? (uint32_t)pnext : 0xFFFFFFFFu);
            w32((uint32_t)tlen);
            memcpy(q, qn, l_qn); q += l_qn;
            if (clip5) w32(clip5 << 4 | 4u);
            w32((acc ? a_len(h, i) : len) << 4);
            if (clip3) w32(clip3 << 4 | 4u);
            if (two) {
                const bk_seg2 &g = seg2[RD(i)];
                if (g.flags & 4) w32((uint32_t)((long)g.match_loci - ((long)h.match_loci + h.match_len)) << 4 | 3u);
                else if (g.flags & 2) w32((uint32_t)((long)len - ((long)h.match_len + g.match_len)) << 4 | 1u);
                else { long gap = (long)g.match_loci - ((long)h.match_loci + h.match_len); w32((uint32_t)(gap < 0 ? -gap : gap) << 4 | 2u); }
                w32((uint32_t)g.match_len << 4);
            }
            uint8_t byte = 0;
            for (uint32_t o = 0; o < len; o++) {
                uint8_t c4 = (acc && h.strand != '+') ? comp4[sq[len - 1 - o] & 7] : code4[sq[o] & 7];
                if (!(o & 1)) byte = (uint8_t)(c4 << 4);
                else byte |= c4;
                if ((o & 1) || o == len - 1) *q++ = byte;
            }
            {
                uint32_t sum = 0;
                for (uint32_t o = 0; o < len; o++) sum += (sq[o] >> 4) & 15;
                if (!sum) memset(q, 0xff, len);
                else {                                       // the reference stores the ASCII form here as well (SAMfile.cpp:2374)
                    const bool rev = acc && h.strand != '+';
                    for (uint32_t o = 0; o < len; o++) q[o] = (uint8_t)(33 + ((((rev ? sq[len - 1 - o] : sq[o]) >> 4) & 15) * 40) / 15);
                }
                q += len;
            }
            if (tag) { *q++ = 'Y'; *q++ = 'U'; *q++ = 'Z'; size_t tl = strlen(tag) + 1; memcpy(q, tag, tl); q += tl; }
            if (acc) S.al.push_back({(uint64_t)at, (uint64_t)(at + 4 + block), ref, (int32_t)pos0, (int32_t)(pos0 + hit_len - 1)});
            S.n++;
        }
    };
    {
        std::vector<std::thread> th;
        auto work = [&](int w) { for (size_t si = (size_t)w; si < n_stripes; si += (size_t)nthreads) format_stripe(si); };
        for (int w = 1; w < nthreads; w++) th.emplace_back(work, w);
        work(0);
        for (auto &t : th) t.join();
    }
    std::vector<bk::BamAligned> aligned;
    uint64_t flush_at = 0, n_rep = 0;
    for (Stripe &S : stripes) {
        const uint64_t base = stream.size();
        for (bk::BamAligned al : S.al) { al.u_beg += base; al.u_end += base; aligned.push_back(al); flush_at = al.u_end; }
        stream.insert(stream.end(), S.bytes.begin(), S.bytes.end());
        n_rep += S.n;
        std::vector<uint8_t>().swap(S.bytes);
    }
    std::string berr;
    rc = bk::write_bam_and_bai(opath, stream, aligned, flush_at, (uint32_t)refs.size(), nthreads, &berr);
    if (rc) { diag("Fatal: %s", berr.c_str()); return 1; }
    report_jct_for_sam(R);
    diag("Completed reporting BAM %llu read alignments", (unsigned long long)n_rep);
    diag("Reporting of aligned result set completed");
    report_read_subset(R, "j", "na", [](uint8_t nar) { return nar == BK_NAR_NS || nar == BK_NAR_NOHIT; });
    report_read_subset(R, "J", "ml", [](uint8_t nar) { return nar == BK_NAR_MULTIALIGN; });
    report_stats(R);
    return 0;
}

// SAM text (-M5 / -M6, optionally gzip'd), CSV (-M0..3) and BED (-M4) output
int report_text(Report &R)
{
    [[maybe_unused]] const Args &a = R.a;
    [[maybe_unused]] auto &hits = R.hits;
    [[maybe_unused]] auto &rs = R.rs;
    [[maybe_unused]] auto &ents = R.ents;
    [[maybe_unused]] const std::string &species = R.species;
    [[maybe_unused]] const uint32_t n_ent = R.n_ent;
    [[maybe_unused]] const size_t nr = R.hits.size();
    [[maybe_unused]] const auto &order = R.order;
    [[maybe_unused]] const auto &seg2 = R.seg2;
    [[maybe_unused]] const auto &multi_dist = R.multi_dist;
    [[maybe_unused]] const int pe_mode = R.pe_mode, ml_mode = R.ml_mode, max_ml = R.max_ml, fmt = R.fmt, nthreads = R.nthreads, micro_indel = R.micro_indel,
                               splice_len = R.splice_len, max_rpt_sam_seqs = R.max_rpt_sam_seqs;
    [[maybe_unused]] auto RD = [&](size_t i) -> size_t { return R.RD(i); };
    [[maybe_unused]] auto has_seg2 = [&](size_t i) -> bool { return R.has_seg2(i); };
    [[maybe_unused]] auto TL = [&](size_t i) -> uint32_t { return R.TL(i); };
    [[maybe_unused]] auto TR = [&](size_t i) -> uint32_t { return R.TR(i); };
    [[maybe_unused]] auto a_start = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_start(h, i); };
    [[maybe_unused]] auto a_len = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_len(h, i); };
    [[maybe_unused]] auto a_mm = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_mm(h, i); };
    OutBuf out;
    out.open(a.str("o").c_str());
    if (out.fd < 0) { diag("Fatal: unable to create '%s'", a.str("o").c_str()); return 1; }
    char line[8192];
    uint64_t n_reported = 0;
    if (fmt >= 5) {
        // header: CSAMfile::Create/AddRefSeq/StartAlignments
        std::vector<uint8_t> has_hit(n_ent + 1, 0);
        for (const bk_hit &h : hits) if (h.nar == BK_NAR_ACCEPTED && h.chrom_id <= n_ent) has_hit[h.chrom_id] = 1;
        bool all = (uint32_t)max_rpt_sam_seqs >= n_ent;
        out.put("@HD\tVN:1.4\tSO:coordinate");
        int n_hdr = 0, n_with = 0;
        for (uint32_t c = 1; c <= n_ent; c++) {
            if (!has_hit[c] && !all) continue;
            int n = snprintf(line, sizeof(line), "\n@SQ\tAS:%s\tSN:%s\tLN:%u", species.empty() ? "NA" : species.c_str(), ents[c - 1].name, ents[c - 1].seq_len);
            out.put(line, (size_t)n);
            n_hdr++;
            n_with += has_hit[c];
        }
        int n = snprintf(line, sizeof(line), "\n@PG\tID:%s\tVN:%s\n", g_proc.c_str(), kProgVer);
        out.put(line, (size_t)n);
        diag("Header written with references to %d sequences of which %d have at least 1 alignments", n_hdr, n_with);
        static const char comp[8] = {'T', 'G', 'C', 'A', 'N', 'N', 'N', 'N'};
        static const char fwd[8] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N'};
        auto put_num = [](std::string &r, long v) {
            char t[24];
            int n = 0;
            bool neg = v < 0;
            unsigned long u = neg ? (unsigned long)(-v) : (unsigned long)v;
            do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
            if (neg) r.push_back('-');
            while (n) r.push_back(t[--n]);
        };
        // QUAL (ReportBAMread :5928-5955): '*' when no base carries a score, else 33 + q4 * 40 / 15 per base, reversed with the read
        auto put_qual = [](std::string &r, const uint8_t *sq, uint32_t n, bool reversed) {
            uint32_t sum = 0;
            for (uint32_t q = 0; q < n; q++) sum += (sq[q] >> 4) & 15;
            if (!sum) { r.push_back('*'); return; }
            const size_t o = r.size();
            r.resize(o + n);
            for (uint32_t q = 0; q < n; q++) r[o + q] = (char)(33 + ((((reversed ? sq[n - 1 - q] : sq[q]) >> 4) & 15) * 40) / 15);
        };
        // one record (CAligner::ReportBAMread, Aligner.cpp:5850-5924,6036-6054); false when the read is not reported
        auto format_rec = [&](size_t k, std::string &rec) -> bool {
            uint32_t i = order[k];
            const bk_hit &h = hits[i];
            bool acc = h.nar == BK_NAR_ACCEPTED;
            if (!acc && fmt != 6) return false;
            const uint8_t *s = rs.bases.data() + rs.offs[RD(i)];
            uint32_t len = rs.lens[RD(i)];
            rec += rs.name(RD(i));
            int flag = 0, tlen = 0;
            long pnext = -1;
            if (!pe_mode) flag = acc ? (h.strand == '+' ? 0 : 16) : 4;
            else {
                const bool first_of_pair = (i & 1) == 0;
                const bk_hit &m = hits[first_of_pair ? i + 1 : i - 1];
                flag = 0x1 | 0x2 | (first_of_pair ? 0x40 : 0x80);
                flag |= acc ? (h.strand == '+' ? 0 : 0x10) : 0x4;
                if ((h.flags & 0x80) && (m.flags & 0x80) && m.nar == BK_NAR_ACCEPTED) {
                    flag |= m.strand == '+' ? 0 : 0x20;
                    if (acc) {
                        const size_t mi = first_of_pair ? i + 1 : i - 1;
                        pnext = (long)a_start(m, mi);
                        long s0 = (long)a_start(h, i), s1 = (long)a_start(m, mi);
                        tlen = (int)(s0 <= s1 ? (s1 - s0) + (long)a_len(m, mi) : (s0 - s1) + (long)a_len(h, i));
                    }
                } else
                    flag |= 0x8;
            }
            rec.push_back('\t');
            put_num(rec, flag);
            if (acc) {
                rec.push_back('\t');
                rec += ents[h.chrom_id - 1].name;
                rec.push_back('\t');
                put_num(rec, (long)a_start(h, i) + 1);
                rec += "\t255\t";
                const uint32_t clip5 = h.strand == '+' ? TL(i) : TR(i), clip3 = h.strand == '+' ? TR(i) : TL(i);
                if (clip5) { put_num(rec, clip5); rec.push_back('S'); }
                put_num(rec, a_len(h, i));
                rec += "M";
                if (clip3) { put_num(rec, clip3); rec.push_back('S'); }
                if (has_seg2(i)) {                                       // CAligner::ReportBAMread, Aligner.cpp:5986-6033
                    const bk_seg2 &g = seg2[RD(i)];
                    if (g.flags & 4) { put_num(rec, (long)g.match_loci - ((long)h.match_loci + h.match_len)); rec.push_back('N'); }
                    else if (g.flags & 2) { put_num(rec, (long)len - ((long)h.match_len + g.match_len)); rec.push_back('I'); }
                    else { long gap = (long)g.match_loci - ((long)h.match_loci + h.match_len); put_num(rec, gap < 0 ? -gap : gap); rec.push_back('D'); }
                    put_num(rec, g.match_len);
                    rec.push_back('M');
                }
                rec.push_back('\t');
                rec.push_back(pnext < 0 ? '*' : '=');
                rec.push_back('\t');
                put_num(rec, pnext < 0 ? 0L : pnext + 1);
                rec.push_back('\t');
                put_num(rec, tlen);
                rec.push_back('\t');
                size_t o = rec.size();
                rec.resize(o + len);
                if (h.strand == '+') for (uint32_t q = 0; q < len; q++) rec[o + q] = fwd[s[q] & 7];
                else for (uint32_t q = 0; q < len; q++) rec[o + q] = comp[s[len - 1 - q] & 7];
                rec.push_back('\t');
                put_qual(rec, s, len, h.strand != '+');
                rec.push_back('\n');
            } else {
                rec += "\t*\t0\t255\t";
                put_num(rec, len);
                rec += "M\t*\t0\t0\t";
                size_t o = rec.size();
                rec.resize(o + len);
                for (uint32_t q = 0; q < len; q++) rec[o + q] = fwd[s[q] & 7];
                rec.push_back('\t');
                put_qual(rec, s, len, false);
                rec += "\t\tYU:Z:";                                    // the doubled TAB is what the reference writes
                rec += kNarTag[h.nar < 20 ? h.nar : 0];
                rec.push_back('\n');
            }
            return true;
        };
        // records are formatted by all host threads into per-thread buffers, one stripe of the sorted order
        // each, and written out in order (the reference formats serially, ~4.5 us per read)
        const size_t per_thread = 32768;
        const int nt = (int)std::min<size_t>((size_t)nthreads, (nr + per_thread - 1) / per_thread ? (nr + per_thread - 1) / per_thread : 1);
        std::vector<std::string> bufs((size_t)nt);
        std::vector<uint64_t> cnts((size_t)nt);
        for (size_t k0 = 0; k0 < nr; k0 += per_thread * (size_t)nt) {
            auto work = [&](int t) {
                size_t lo = k0 + (size_t)t * per_thread, hi = std::min(nr, lo + per_thread);
                std::string &buf = bufs[(size_t)t];
                buf.clear();
                uint64_t c = 0;
                for (size_t k = lo; k < hi; k++) c += format_rec(k, buf) ? 1 : 0;
                cnts[(size_t)t] = c;
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; t++) th.emplace_back(work, t);
            work(0);
            for (auto &t : th) t.join();
            if (out.gz) {                    // compressed SAM: one deflate stream, in order
                for (int t = 0; t < nt; t++) { out.put(bufs[(size_t)t]); n_reported += cnts[(size_t)t]; }
                continue;
            }
            // the stripes go to their places in the file in parallel as well
            out.flush();
            std::vector<off_t> at((size_t)nt + 1);
            at[0] = out.pos;
            for (int t = 0; t < nt; t++) { at[(size_t)t + 1] = at[(size_t)t] + (off_t)bufs[(size_t)t].size(); n_reported += cnts[(size_t)t]; }
            auto put = [&](int t) {
                const std::string &bf = bufs[(size_t)t];
                size_t o = 0;
                while (o < bf.size()) {
                    ssize_t w = ::pwrite(out.fd, bf.data() + o, bf.size() - o, at[(size_t)t] + (off_t)o);
                    if (w <= 0) break;
                    o += (size_t)w;
                }
            };
            th.clear();
            for (int t = 1; t < nt; t++) th.emplace_back(put, t);
            put(0);
            for (auto &t : th) t.join();
            out.pos = at[(size_t)nt];
        }
        report_jct_for_sam(R);
        diag("Completed reporting SAM %llu read alignments", (unsigned long long)n_reported);
    } else {
        // -M0..3 CSV (loci; 1: + match sequence, 2: + read sequence, 3: + both) and -M4 UCSC BED
        // (CAligner::WriteReadHits, Aligner.cpp:6336-6660); the site-preference score column is 0 as in the
        // reference when no -8 preferences are computed
        static const char up[8] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N'};
        bk::SfxFile sf;
        if (fmt == 1 || fmt == 3) {
            std::string serr;
            if (bk::sfx_open(a.str("I").c_str(), sf, &serr) != 0) { diag("Fatal: %s", serr.c_str()); return 1; }
        }
        if (fmt == 4) {
            std::string title = a.str("t", "kanga");
            int m = snprintf(line, sizeof(line), "track type=bed name=\"%s\" description=\"%s\"\n", title.c_str(), title.c_str());
            out.put(line, (size_t)m);
            if (ml_mode == 5) out.put(line, (size_t)m);      // written at file creation AND by WriteReadHits (Aligner.cpp:4405-4413,6356-6362)
        }
        // -a with -M4: reads aligned with a microInDel go to "<out>.ind" as 12-column BED lines (Aligner.cpp:4417-4438,6372-6376,6519-6526)
        OutBuf ind;
        if (fmt == 4 && micro_indel) {
            ind.open((a.str("o") + ".ind").c_str());
            if (ind.fd < 0) { diag("Fatal: unable to create '%s.ind'", a.str("o").c_str()); return 1; }
            std::string title = a.str("t", "kanga");
            int m = snprintf(line, sizeof(line), "track type=bed name=\"IND_%s\" description=\"%s\"\n", title.c_str(), title.c_str());
            ind.put(line, (size_t)m);
        }
        OutBuf jct;
        if (fmt == 4 && splice_len) {
            jct.open((a.str("o") + ".jct").c_str());
            if (jct.fd < 0) { diag("Fatal: unable to create '%s.jct'", a.str("o").c_str()); return 1; }
            std::string title = a.str("t", "kanga");
            int m = snprintf(line, sizeof(line), "track type=bed name=\"JCT_%s\" description=\"%s\"\n", title.c_str(), title.c_str());
            jct.put(line, (size_t)m);
        }
        std::string rec;
        for (size_t k = 0; k < nr; k++) {
            uint32_t i = order[k];
            const bk_hit &h = hits[i];
            if (h.nar != BK_NAR_ACCEPTED) continue;
            const bool two = has_seg2(i);
            if (fmt == 4) {
                if (two) {
                    const bk_seg2 &g = seg2[RD(i)];
                    const bool sj = (g.flags & 4) != 0;
                    const uint32_t end1 = g.match_loci + g.match_len;          // AdjAlignEndLoci + 1
                    int m = snprintf(line, sizeof(line), "%s\t%u\t%u\t%s\t0\t%c\t%u\t%u\t0\t2\t%u,%u\t0,%u\n", ents[h.chrom_id - 1].name, h.match_loci, end1,
                                     sj ? "arj" : "ari", (char)h.strand, h.match_loci, end1, (unsigned)h.match_len, (unsigned)g.match_len, g.match_loci - h.match_loci);
                    (sj ? jct : ind).put(line, (size_t)m);
                } else {
                    int m = snprintf(line, sizeof(line), "%s\t%u\t%u\tar\t0\t%c\n", ents[h.chrom_id - 1].name, a_start(h, i), a_start(h, i) + a_len(h, i),
                                     (char)h.strand);
                    out.put(line, (size_t)m);
                }
                n_reported++;
                continue;
            }
            // one line per segment (WriteReadHits, Aligner.cpp:6566-6627)
            const uint32_t len = rs.lens[RD(i)];
            for (int sg = 0; sg < (two ? 2 : 1); sg++) {
                const uint32_t s_loci = sg ? seg2[RD(i)].match_loci : a_start(h, i), s_len = sg ? seg2[RD(i)].match_len : a_len(h, i);
                const uint32_t s_mm = sg ? seg2[RD(i)].mismatches : a_mm(h, i), s_rofs = sg ? seg2[RD(i)].read_ofs : TL(i);       // ReadOfs + TrimLeft
                int m = snprintf(line, sizeof(line), "%u,\"%s\",\"%s\",\"%s\",%u,%u,%u,\"%c\",0,0,1,%u,\"N/A\",\"%s\"", i + 1,
                                 two ? ((seg2[RD(i)].flags & 4) ? "arj" : "ari") : "ar", species.c_str(),
                                 ents[h.chrom_id - 1].name, s_loci, s_loci + s_len - 1, (unsigned)s_len, (char)h.strand, (unsigned)s_mm, rs.name(RD(i)));
                rec.assign(line, (size_t)m);
                if (fmt >= 2) {                                          // the read as loaded, from the segment's read offset
                    const uint8_t *sq = rs.bases.data() + rs.offs[RD(i)];
                    rec += ",\"";
                    for (uint32_t q = 0; q < s_len && s_rofs + q < len; q++) rec.push_back(up[sq[s_rofs + q] & 7]);
                    rec.push_back('"');
                }
                if (fmt == 1 || fmt == 3) {                              // the target it matched, in read orientation
                    const uint8_t *tg = sf.seq + ents[h.chrom_id - 1].start_ofs + s_loci;
                    rec += ",\"";
                    for (uint32_t q = 0; q < s_len; q++) {
                        uint8_t t = h.strand == '-' ? tg[s_len - 1 - q] & 7 : tg[q] & 7;
                        if (h.strand == '-' && t < 4) t = (uint8_t)(3 - t);
                        rec.push_back(up[t]);
                    }
                    rec.push_back('"');
                }
                rec.push_back('\n');
                out.put(rec);
            }
            n_reported++;
        }
        if (ind.fd >= 0) ind.close();
        if (jct.fd >= 0) jct.close();
    }
    out.close();
    diag("Reporting of aligned result set completed");

    report_read_subset(R, "j", "na", [](uint8_t nar) { return nar == BK_NAR_NS || nar == BK_NAR_NOHIT; });
    report_read_subset(R, "J", "ml", [](uint8_t nar) { return nar == BK_NAR_MULTIALIGN; });
    report_stats(R);
    return 0;
}

int cmd_align(int argc, char **argv, int first)
{
    Args a;
    std::string err;
    std::map<std::string, std::string> ln = {
        {"mode", "m"}, {"alignstrand", "Q"}, {"editdelta", "e"}, {"substitutions", "s"}, {"maxns", "n"}, {"trim5", "y"},
        {"trim3", "Y"}, {"minacceptreadlen", "l"}, {"maxacceptreadlen", "L"}, {"format", "M"}, {"in", "i"}, {"sfx", "I"},
        {"out", "o"}, {"stats", "O"}, {"threads", "T"}, {"log", "F"}, {"FileLogLevel", "f"}, {"pemode", "U"}, {"mlmode", "r"},
        {"quality", "g"}, {"device", "device"}, {"rptsamseqsthres", "4"}, {"pair", "u"}, {"pairminlen", "d"}, {"pairmaxlen", "D"},
        {"pairstrand", "E"}, {"nonealign", "j"}, {"multialign", "J"}, {"title", "t"}, {"maxmulti", "R"}, {"clampmaxmulti", "X"},
        {"bestmatches", "N"}, {"microindellen", "a"}, {"minflankexacts", "x"}, {"splicejunctlen", "A"}, {"minchimeric", "c"}, {"pcrwin", "k"}, {"samplenthrawread", "#"}, {"chromexclude", "Z"}, {"chromeinclude", "z"}};
    if (!parse_args(argc, argv, first, ln, "mQesnyYlLMiIoOTFfUrg4udDjJtRaxAck#Zz", "EXN", a, err)) {
        fprintf(stderr, "%s align: %s\n", g_proc.c_str(), err.c_str());
        return 1;
    }
    if (!a.has("i") || !a.has("I") || !a.has("o")) {
        fprintf(stderr, "usage: %s align -i <reads> -I <genome.sfx> -o <out.sam> [-s subs] [-e delta] [-Q strand] [-m mode] [-n maxNs] "
                        "[-l minlen] [-L maxlen] [-M 0|5|6] [-O stats]\n", g_proc.c_str());
        return 1;
    }
    if (a.has("F")) g_logfile = fopen(a.str("F").c_str(), "a");
    diag("Subprocess align Version %s starting", kProgVer);
    const int pe_mode = a.num("U", 0);
    if (pe_mode < 0 || pe_mode > 4) { diag("Error: paired end processing mode '-U%d' must be in range 0..4", pe_mode); return 1; }
    if (pe_mode && (!a.has("u") || a.v["u"].size() != a.v["i"].size())) {
        diag("Error: paired end processing '-U%d' needs as many '-u' PE2 files as '-i' PE1 files", pe_mode);
        return 1;
    }
    bk_pe_params PE = {};
    PE.pe_mode = pe_mode;
    PE.pair_min_len = a.num("d", 100);            // cDfltPairMinLen
    PE.pair_max_len = a.num("D", 1000);           // cDfltPairMaxLen
    PE.pair_strand = a.has("E") ? 1 : 0;
    if (pe_mode && (PE.pair_min_len < 25 || PE.pair_max_len < PE.pair_min_len || PE.pair_max_len > 100000)) {
        diag("Error: paired end insert size range '-d%d -D%d' not accepted", PE.pair_min_len, PE.pair_max_len);
        return 1;
    }
    // -r multi-loci modes (kanga.cpp:482-486,535-539,666-694): 0 slough, 1 stats only, 2 random pick, 3 cluster with
    // uniques, 4 cluster with uniques + other multi-loci reads, 5 report all loci; -R loci limit, -X clamp
    const int ml_mode = a.num("r", 0);
    if (ml_mode < 0 || ml_mode > 5) { diag("Error: multiple aligned reads processing mode '-r%d' specified outside of range 0..5", ml_mode); return 1; }
    if (pe_mode && ml_mode) { diag("Error: Sorry, currently multiloci processing '-r%d' not supported in paired end '-U%d' processing", ml_mode, pe_mode); return 1; }
    int max_ml = 1;
    bool clamp_ml = false, best_matches = false;
    if (ml_mode) {
        max_ml = a.num("R", 5);                                         // cDfltMaxMultiHits
        const int lim = ml_mode == 5 ? 100000 : 500;                    // cMaxAllHits / cMaxMultiHits
        if (max_ml < 2 || max_ml > lim) { diag("Error: multiple aligned reads '-R%d' specified outside of range 2..%d", max_ml, lim); return 1; }
        if (max_ml > BK_MAX_ML) { diag("Error: '-R%d' is above the %d loci per read this build keeps", max_ml, BK_MAX_ML); return 1; }
        best_matches = a.has("N");                                      // bLocateBestMatches (implies the clamp, kanga.cpp:686-694)
        clamp_ml = a.has("X") || best_matches;
    }
    // -g FASTQ quality scores: 0 Sanger / Illumina 1.8+, 1 Illumina 1.3+, 2 Solexa, 3 ignore (default; QUAL is then '*')
    g_sample_nth = a.num("#", 1);
    if (g_sample_nth < 1 || g_sample_nth > 10000) { diag("Error: sample every Nth raw read '-#%d' specified outside of range 1..10000", g_sample_nth); return 1; }
    g_qual_mode = a.num("g", 3);
    if (g_qual_mode < 0 || g_qual_mode > 3) { diag("Error: fastq quality '-g%d' specified outside of range 0..3", g_qual_mode); return 1; }
    // -a microInDels (kanga.cpp:696-710): looked for in reads the substitution-only phases leave unaligned
    const int micro_indel = a.num("a", 0);
    if (micro_indel < 0 || micro_indel > 20) { diag("Error: microInDel length maximum '-a%d' specified outside of range 0..20", micro_indel); return 1; }
    if (micro_indel && ml_mode == 5) { diag("Error: microInDels not supported when reporting multiloci alignments"); return 1; }
    if (micro_indel && (ml_mode || pe_mode)) { diag("Error: microInDels '-a%d' together with '-r%d' / '-U%d' are not available in this build", micro_indel, ml_mode, pe_mode); return 1; }
    // -x: trim aligned reads back from both ends until that many consecutive bases match (CAligner::AutoTrimFlanks)
    // -A RNA-seq splice junctions (kanga.cpp:726-742,810-811): looked for after the microInDel pass; switches flank trimming on
    const int splice_len = a.num("A", 0);
    if (splice_len != 0 && (splice_len < 25 || splice_len > 100000)) { diag("Error: RNAseq maximum splice junction separation '-A%d' must be either 0 or in the range 25..100000", splice_len); return 1; }
    if (splice_len && ml_mode == 5) { diag("Error: in report all multiloci mode '-r5', there is no splice junction processing.."); return 1; }
    if (splice_len && pe_mode) { diag("Error: Sorry, currently RNA-seq splice junction processing '-A%d' not supported in paired end '-U%d' processing", splice_len, pe_mode); return 1; }
    if (splice_len && ml_mode) { diag("Error: splice junctions '-A%d' together with '-r%d' are not available in this build", splice_len, ml_mode); return 1; }
    // -c chimeric trimming (kanga.cpp:648-664): reads nothing else aligned may be placed with their ends trimmed off
    const int min_chim = a.num("c", 0);
    if (min_chim != 0 && (min_chim < 50 || min_chim > 99)) { diag("Error: minimum chimeric length percentage '-c%d' specified outside of range 50..99", min_chim < 0 ? -min_chim : min_chim); return 1; }
    if (min_chim && (ml_mode || pe_mode)) { diag("Error: chimeric trimming '-c%d' together with '-r%d' / '-U%d' is not available in this build", min_chim, ml_mode, pe_mode); return 1; }
    // -k PCR differential amplification artefact reduction (kanga.cpp:718-724): window 0..250, off by default
    const int pcr_win = a.has("k") ? a.num("k", -1) : -1;
    if (a.has("k") && (pcr_win < 0 || pcr_win > 250)) { diag("Error: PCR differential amplification artefacts window length '-k%d' specified outside of range 0..250", pcr_win); return 1; }
    if (pcr_win >= 0 && ml_mode == 5) { diag("Error: '-k%d' together with '-r5' is not available in this build", pcr_win); return 1; }
    // -Z / -z chromosome exclude / include filters: POSIX extended regular expressions, case-insensitive (Aligner.cpp:4770-4795)
    std::vector<regex_t> re_excl, re_incl;
    for (const char *opt : {"Z", "z"})
        if (a.has(opt))
            for (const std::string &pat : a.v[opt]) {
                regex_t re;
                if (regcomp(&re, pat.c_str(), REG_EXTENDED | REG_ICASE)) { diag("Error: ProcessAlign: %s chrom RE '%s' error", opt[0] == 'Z' ? "exclude" : "include", pat.c_str()); return 1; }
                (opt[0] == 'Z' ? re_excl : re_incl).push_back(re);
            }
    if ((!re_excl.empty() || !re_incl.empty()) && (pe_mode || ml_mode == 5)) { diag("Error: chromosome filters '-Z/-z' together with '-U%d' / '-r5' are not available in this build", pe_mode); return 1; }
    int min_flank = a.num("x", 0);
    if (min_flank < 0 || min_flank > 7) { diag("Error: Max flank trimming '-x%d' specified outside of range 0..7", min_flank); return 1; }      // cMaxAllowedSubs / 2
    if (min_flank && ml_mode == 5) { diag("Error: flank trimming '-x%d' together with '-r5' is not available in this build", min_flank); return 1; }
    bk_align_params P = {};
    P.micro_indel_len = micro_indel;
    P.splice_junct_len = splice_len;
    P.pmode = a.num("m", 0);
    P.align_strand = a.num("Q", 0);
    P.min_edit_dist = a.num("e", 1);
    P.max_subs = a.num("s", 10);                  // cDfltAllowedSubs per 100bp
    if (splice_len > 0 && min_chim == 0 && min_flank == 0) min_flank = P.max_subs;      // MinFlankExacts = MaxSubs (kanga.cpp:810-811)
    P.min_chimeric_len = min_chim;
    P.max_ns = a.num("n", 1);
    P.max_ml = max_ml;
    P.clamp_ml = clamp_ml ? 1 : 0;
    P.best_matches = best_matches ? 1 : 0;
    int fmt = a.num("M", 5);
    if (ml_mode == 5 && !(fmt == 0 || fmt == 4 || fmt == 5 || fmt == 6)) {      // kanga.cpp:830-834
        diag("Error: reporting all multiloci alignments '-r5' is only available with output formats '-M0', '-M4', '-M5' and '-M6'");
        return 1;
    }
    if (a.has("O") && fmt == 6) {                 // kanga.cpp:1015-1021
        diag("Error: Output induced substitution mode '-O<file>' not available in '-M6' output mode\n");
        return 1;
    }
    int min_len = a.num("l", 50), max_len = a.num("L", 500);
    int trim5 = a.num("y", 0), trim3 = a.num("Y", 0);
    int max_rpt_sam_seqs = a.num("4", 10000);
    // -T: host threads for parsing, sorting and formatting (0 = all cores, capped like the reference's cMaxWorkerThreads)
    int nthreads = a.num("T", 0);
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 128) nthreads = 128;
    if (P.pmode < 0 || P.pmode > 3 || P.align_strand < 0 || P.align_strand > 2 || P.min_edit_dist < 1 || P.min_edit_dist > 2 ||
        P.max_subs < 0 || P.max_subs > 25 || P.max_ns < 0 || P.max_ns > 5 || min_len < 15 || min_len > 2000 || max_len < min_len ||
        max_len > 2000 || fmt < 0 || fmt > 6) {
        diag("Error: an option value is outside its accepted range");
        return 1;
    }

    diag("Loading suffix array file '%s'", a.str("I").c_str());
    bk_ctx *ctx = nullptr;
    int rc = bk_ctx_create(&ctx, a.str("I").c_str(), a.num("device", 0), &P);
    if (rc) { diag("Fatal: unable to load genome assembly suffix array: %s", bk_strerror(rc)); return 1; }
    std::string species = bk_dataset_name(ctx);
    uint32_t n_ent = bk_num_entries(ctx);
    std::vector<bk_entry_info> ents(n_ent);
    for (uint32_t i = 0; i < n_ent; i++) bk_get_entry(ctx, i, &ents[i]);
    diag("Genome Assembly Name: '%s'", species.c_str());
    diag("Genome assembly suffix array loaded");

    ReadStore rs;
    if (pe_mode) rc = load_reads_pe(a.v["i"], a.v["u"], trim5, trim3, min_len, max_len, nthreads, rs);
    else rc = load_reads(a.v["i"], trim5, trim3, min_len, max_len, nthreads, rs);
    if (rc) { bk_ctx_destroy(ctx); return 1; }
    size_t nr = rs.size();
    diag("Now aligning with minimum core size of %dbp...\n", bk_min_core_len(ctx));
    std::vector<bk_hit> hits(nr);
    std::vector<uint64_t> l_offs;                  // multi-loci modes: read i owns loci [l_offs[i], l_offs[i+1])
    std::vector<bk_loci> loci;
    std::vector<bk_seg2> seg2;                     // -a: second segment of each read (flags 0 = none)
    if (ml_mode) l_offs.assign(1, 0);
    const size_t kBatch = 16u << 20;
    for (size_t lo = 0; lo < nr; lo += kBatch) {
        size_t n = std::min(kBatch, nr - lo);
        rc = bk_align_batch(ctx, rs.bases.data(), rs.offs.data() + lo, rs.lens.data() + lo, (uint32_t)n, hits.data() + lo);
        if (rc) { diag("Fatal: alignment failed: %s", bk_strerror(rc)); bk_ctx_destroy(ctx); return 1; }
        if (micro_indel || splice_len || min_chim) {
            const bk_seg2 *bs = nullptr;
            uint64_t ns = 0;
            rc = bk_batch_seg2(ctx, &bs, &ns);
            if (rc || !bs || ns != n) { diag("Fatal: microInDel segments unavailable: %s", bk_strerror(rc)); bk_ctx_destroy(ctx); return 1; }
            seg2.insert(seg2.end(), bs, bs + ns);
        }
        if (ml_mode) {
            const uint64_t *bo = nullptr;
            const bk_loci *bl = nullptr;
            uint64_t nl = 0;
            rc = bk_batch_loci(ctx, &bo, &bl, &nl);
            if (rc || !bo) { diag("Fatal: loci lists unavailable: %s", bk_strerror(rc)); bk_ctx_destroy(ctx); return 1; }
            const uint64_t base = loci.size();
            for (size_t i = 1; i <= n; i++) l_offs.push_back(base + bo[i]);
            loci.insert(loci.end(), bl, bl + nl);
        }
    }
    diag("Alignment of %zu from %zu loaded completed", nr, nr);

    std::vector<uint32_t> src;                     // -r5: record -> read it came from (records replace the reads)
    std::vector<int> multi_dist((size_t)max_ml, 0);
    if (ml_mode) {
        // CAligner::ProcCoredApprox for MLMode != eMLdefault (Aligner.cpp:9241-9424); reads in load order, as -T1 runs them
        auto eff_count = [&](size_t i) -> uint32_t {            // LowHitInstances of a read that counts as eHRhits (after -X)
            const bk_hit &h = hits[i];
            if (h.rslt == BK_HR_HITS || (clamp_ml && h.rslt == BK_HR_HITINSTS)) return (uint32_t)(l_offs[i + 1] - l_offs[i]);
            return 0;
        };
        auto take = [&](bk_hit &h, const bk_loci &L) {
            h.chrom_id = L.chrom_id; h.match_loci = L.match_loci; h.match_len = L.match_len; h.strand = L.strand;
            h.mismatches = L.mismatches; h.nar = BK_NAR_ACCEPTED; h.num_hits = 1; h.low_hit_instances = 1;
        };
        uint64_t n_uniq = 0, n_multi = 0, n_loci = 0;
        for (size_t i = 0; i < nr; i++) {
            const uint32_t c = eff_count(i);
            if (!c) continue;
            n_loci += c;
            (c == 1 ? n_uniq : n_multi)++;
            if (ml_mode != 5) multi_dist[c - 1]++;
            if (clamp_ml && hits[i].rslt == BK_HR_HITINSTS) hits[i].low_hit_instances = (int16_t)c;
        }
        diag("Provisionally accepted %llu aligned reads (%llu uniquely, %llu aligning to multiloci) aligning to a total of %llu loci",
             (unsigned long long)(n_uniq + n_multi), (unsigned long long)n_uniq, (unsigned long long)n_multi, (unsigned long long)n_loci);
        if (ml_mode == 2) {
            // eMLrand: rand() % LowHitInstances for every eHRhits read, unique ones included (:9365-9366); the sequence is
            // glibc's unseeded one (glibc_rand.h), which is what a single-threaded reference run consumes in the same order
            bk::GlibcRand pick;
            for (size_t i = 0; i < nr; i++) {
                const uint32_t c = eff_count(i);
                if (!c) continue;
                const uint32_t k = (uint32_t)pick.next() % c;
                take(hits[i], loci[l_offs[i] + k]);
            }
        } else if (ml_mode == 3 || ml_mode == 4) {
            uint32_t max_reads_len = 0;
            for (size_t i = 0; i < nr; i++) max_reads_len = std::max(max_reads_len, rs.lens[i]);
            bk::MultiAssign ma;
            for (size_t i = 0; i < nr; i++) {
                const uint32_t c = eff_count(i);
                for (uint32_t k = 0; k < c; k++) ma.add((uint32_t)i + 1, loci[l_offs[i] + k], c > 1);
            }
            diag("Assigning %llu reads which aligned to multiple loci to a single loci", (unsigned long long)n_multi);
            bk::MultiAssignStats st = ma.assign(ml_mode == 3, nthreads, max_reads_len);
            for (const bk::MultiHitRec &m : ma.recs)
                if (m.multi && m.assigned) take(hits[m.read_id - 1], m.loci);
            diag("Clustering completed, removed %d unclustered orphans from %d putative resulting in %d (%d clustered near unique, %d clustered near other multiloci reads) multihit reads accepted as assigned",
                 st.putative - st.assigned, st.putative, st.assigned, st.near_unique, st.near_multi);
        } else if (ml_mode == 5) {
            // eMLall: every locus becomes a record of its own, ReadID = order of creation (CAligner::WriteHitLoci / AddMultiHit,
            // Aligner.cpp:6666-6800); with -M6 reads without alignment (eHRnone, eHRHitInsts) are kept as one unaligned
            // record, everything else (EN, MMDelta) drops out (:9311-9352,9441-9449)
            std::vector<bk_hit> recs;
            for (size_t i = 0; i < nr; i++) {
                const bk_hit &h = hits[i];
                const uint32_t c = eff_count(i);
                if (c) {
                    for (uint32_t k = 0; k < c; k++) {
                        bk_hit r = h;
                        take(r, loci[l_offs[i] + k]);
                        recs.push_back(r);
                        src.push_back((uint32_t)i);
                    }
                } else if (fmt == 6 && h.nar != BK_NAR_NS && (h.rslt == BK_HR_NONE || h.rslt == BK_HR_HITINSTS)) {
                    bk_hit r = h;
                    r.num_hits = 0;
                    r.low_mm = 0;
                    recs.push_back(r);
                    src.push_back((uint32_t)i);
                }
            }
            diag("Treating accepted %llu multialigned reads as uniquely aligned %llu source reads in subsequent processing",
                 (unsigned long long)n_multi, (unsigned long long)(n_loci - n_uniq));
            hits.swap(recs);
            nr = hits.size();
        }
    }
    auto RD = [&](size_t i) -> size_t { return src.empty() ? i : (size_t)src[i]; };
    auto has_seg2 = [&](size_t i) -> bool { return !seg2.empty() && (seg2[RD(i)].flags & 5); };       // FlgInDel or FlgSplice
    if (pe_mode) {
        // CAligner::ProcessPairedEnds: reads are held interleaved PE1,PE2 (Aligner.cpp:11349-11355); before the flank trimmer, as in
        // CAligner::Align (:573-622)
        diag("Paired end association and partner alignment processing started..");
        const size_t kPairs = 8u << 20;
        for (size_t lo = 0; lo < nr / 2; lo += kPairs) {
            size_t n = std::min(kPairs, nr / 2 - lo);
            rc = bk_pair_batch(ctx, rs.bases.data(), rs.offs.data() + 2 * lo, rs.lens.data() + 2 * lo, (uint32_t)n, hits.data() + 2 * lo, &PE);
            if (rc) { diag("Fatal: paired end processing failed: %s", bk_strerror(rc)); bk_ctx_destroy(ctx); return 1; }
        }
        size_t n_pe = 0;
        for (size_t i = 0; i < nr; i += 2) n_pe += (hits[i].flags & 0x80) && (hits[i + 1].flags & 0x80);
        diag("From %zu paired reads there were %zu accepted as paired", nr / 2, n_pe);
    }

    // per-record flank trims in READ orientation (tsSegLoci.TrimLeft / TrimRight / TrimMismatches of Seg[0]), host/post_filters.h
    bk::FlankTrims trims;
    auto TL = [&](size_t i) -> uint32_t { return trims.empty() ? 0u : trims.left[i]; };
    auto TR = [&](size_t i) -> uint32_t { return trims.empty() ? 0u : trims.right[i]; };
    auto a_start = [&](const bk_hit &h, size_t i) -> uint32_t { return h.match_loci + (h.strand == '+' ? TL(i) : TR(i)); };      // AdjStartLoci
    auto a_len = [&](const bk_hit &h, size_t i) -> uint32_t { return (uint32_t)h.match_len - TL(i) - TR(i); };                     // AdjHitLen
    auto is_chimeric = [&](size_t i) -> bool { return !seg2.empty() && (seg2[RD(i)].flags & 8); };     // FlgChimeric: trims come with the hit
    if (min_chim) {
        // chimeric placements keep the trims AdaptiveTrim found (ProcCoredApprox :9292-9299); the flank trimmer leaves them alone (:1641)
        if (trims.empty()) {
            trims.left.assign(nr, 0); trims.right.assign(nr, 0); trims.mismatches.resize(nr);
            for (size_t i = 0; i < nr; i++) trims.mismatches[i] = hits[i].mismatches;
        }
        size_t n_ch = 0;
        for (size_t i = 0; i < nr; i++)
            if (hits[i].nar == BK_NAR_ACCEPTED && is_chimeric(i)) { trims.left[i] = seg2[RD(i)].match_len; trims.right[i] = seg2[RD(i)].read_ofs; n_ch++; }
        diag("Of the accepted aligned reads, %zu were chimeric", n_ch);
    }
    // CAligner::SortHitMatch (Aligner.cpp:10069-10114) over record indexes, ties left to the replica of the reference's sort
    auto cmp = [&](uint32_t x, uint32_t y) -> int {
        const bk_hit &p = hits[x], &q = hits[y];
        if (p.nar != q.nar) return p.nar < q.nar ? -1 : 1;
        if (p.num_hits == 1 && q.num_hits != 1) return -1;
        if (p.num_hits != 1 && q.num_hits == 1) return 1;
        if (p.num_hits != 1 && q.num_hits != 1) return p.num_hits < q.num_hits ? -1 : (p.num_hits > q.num_hits ? 1 : 0);
        if (p.chrom_id != q.chrom_id) return p.chrom_id < q.chrom_id ? -1 : 1;
        const uint32_t ps = a_start(p, x), qs = a_start(q, y), pl = a_len(p, x), ql = a_len(q, y);
        if (ps != qs) return ps < qs ? -1 : 1;
        if (pl != ql) return pl < ql ? -1 : 1;
        if (p.strand != q.strand) return p.strand < q.strand ? -1 : 1;
        if (p.low_mm != q.low_mm) return p.low_mm < q.low_mm ? -1 : 1;
        return 0;
    };
    if (pcr_win >= 0 && !pe_mode) {
        // CAligner::ReducePCRduplicates runs on the sorted set, before the flank trimmer (Aligner.cpp:598-610)
        diag("Processing to reduce PCR differential amplification artefacts processing started..");
        std::vector<uint32_t> ord(nr);
        for (size_t i = 0; i < nr; i++) ord[i] = (uint32_t)i;
        bk::ref_order_sort(ord.data(), (int64_t)nr, cmp, nthreads);
        const size_t n_dup = bk::reduce_pcr_duplicates(hits, ord, [&](size_t i) { return a_start(hits[i], i); }, [&](size_t i) { return a_len(hits[i], i); }, pcr_win);
        diag("Removed %zu potential PCR artefact reads", n_dup);
    }
    if (min_flank > 0) {
        diag("Starting 5' and 3' flank sequence autotrim processing...");
        bk::SfxFile sft;
        std::string serr;
        if (bk::sfx_open(a.str("I").c_str(), sft, &serr) != 0) { diag("Fatal: %s", serr.c_str()); bk_ctx_destroy(ctx); return 1; }
        bk::auto_trim_flanks(hits, [&](size_t i) { return has_seg2(i) || is_chimeric(i); }, [&](size_t i) { return rs.bases.data() + rs.offs[RD(i)]; },
                             [&](size_t i) -> const uint8_t * {
                                 const bk_hit &h = hits[i];
                                 return (h.chrom_id >= 1 && h.chrom_id <= n_ent) ? sft.seq + ents[h.chrom_id - 1].start_ofs + h.match_loci : nullptr;
                             },
                             min_flank, pe_mode != 0, nthreads, trims);
        diag("Finished 5' and 3' flank sequence autotriming, %zu plus strand and %zu minus strand aligned reads removed", trims.removed_plus,
             trims.removed_minus);
    }
    // orphan junction filters, splice junctions first (Aligner.cpp:630-650)
    if (splice_len) {
        diag("Removal of orphan splice junction processing started..");
        auto r = bk::remove_orphan_segs(hits, seg2, 4, 7);
        diag("From %zu reads with putative splice junctions %zu orphans were removed", r.first, r.second);
    }
    if (micro_indel) {
        diag("Removal of orphan microInDels processing started..");
        auto r = bk::remove_orphan_segs(hits, seg2, 1, 8);
        diag("From %zu reads with putative microIndels %zu orphans were removed", r.first, r.second);
    }

    if (!re_excl.empty() || !re_incl.empty()) {
        // CAligner::FiltByChroms (Aligner.cpp:4019-4120): a sequence stays if an include expression matches its name, or - with no
        // include expressions at all - if no exclude expression does; accepted reads on the others become eNARChromFilt
        diag("Now filtering matches by chromosome");
        std::vector<uint8_t> keep(n_ent + 1, 1);
        for (uint32_t c = 1; c <= n_ent; c++) {
            regmatch_t mc;
            bool ok = false;
            for (regex_t &re : re_incl) if (!regexec(&re, ents[c - 1].name, 1, &mc, 0)) { ok = true; break; }
            if (!ok && re_incl.empty()) {
                ok = true;
                for (regex_t &re : re_excl) if (!regexec(&re, ents[c - 1].name, 1, &mc, 0)) { ok = false; break; }
            }
            keep[c] = ok ? 1 : 0;
        }
        size_t n_filt = 0;
        for (size_t i = 0; i < nr; i++) {
            bk_hit &h = hits[i];
            if (h.nar == BK_NAR_ACCEPTED && h.chrom_id <= n_ent && !keep[h.chrom_id]) { h.nar = 11; h.num_hits = 0; h.low_hit_instances = 0; n_filt++; }
        }
        diag("Filtering by chromosome completed - removed %zu  matches", n_filt);
    }

    // CAligner::ReportAlignStats (Aligner.cpp:3493-3822): NAR histogram
    uint64_t nar[20] = {0};
    for (const bk_hit &h : hits) nar[h.nar < 20 ? h.nar : 0]++;
    diag("Unable to align %llu source reads of which %llu were not aligned as they contained excessive number of indeterminate 'N' bases",
         (unsigned long long)(nr - nar[1]), (unsigned long long)nar[2]);
    diag("Read nonalignment reason summary:");
    for (int k = 0; k < 20; k++) diag("   %llu (%s) %s", (unsigned long long)nar[k], kNarTag[k], kNarDescr[k]);

    // SortReadHits(eRSMHitMatch): index in load (ReadID) order -> reference order
    diag("Reporting of aligned result set started...");
    diag("Sorting alignments by ascending chrom.loci");
    std::vector<uint32_t> order(nr);
    for (size_t i = 0; i < nr; i++) order[i] = (uint32_t)i;
    bk::ref_order_sort(order.data(), (int64_t)nr, cmp, nthreads);

    Report R{a, rs, hits, ents, species, n_ent, src, seg2, trims, multi_dist, order, pe_mode, ml_mode, max_ml, fmt, nthreads, micro_indel, splice_len, max_rpt_sam_seqs};
    // ".bam" (more than 5 characters of name, kanga.cpp:848-857): BGZF-compressed BAM with its BAI index; else SAM / CSV / BED text
    const std::string opath = a.str("o");
    const int rr = (fmt >= 5 && opath.size() > 5 && !strcasecmp(opath.c_str() + opath.size() - 4, ".bam")) ? report_bam(R, opath) : report_text(R);
    bk_ctx_destroy(ctx);
    return rr;
}

}  // namespace

int main(int argc, char **argv)
{
    // gszProcName: basename of argv[0] without extension (biokanga.cpp:236-246)
    std::string p = argv[0];
    size_t sl = p.find_last_of('/');
    if (sl != std::string::npos) p = p.substr(sl + 1);
    size_t dot = p.find_last_of('.');
    if (dot != std::string::npos && dot > 0) p = p.substr(0, dot);
    if (!p.empty()) g_proc = p;
    if (argc < 2) {
        fprintf(stderr, "%s: MI355X build of the BioKanga `index` and `align` sub-processes (%s)\nusage: %s index|align <options>\n",
                g_proc.c_str(), bk_version(), g_proc.c_str());
        return 1;
    }
    time_t t0 = time(nullptr);
    int rc;
    std::string sub = argv[1];
    if (sub == "index" || sub == "kangax") rc = cmd_index(argc, argv, 2);
    else if (sub == "align" || sub == "kanga") rc = cmd_align(argc, argv, 2);
    else {
        fprintf(stderr, "%s: sub-process '%s' is outside the supported hot path (index, align)\n", g_proc.c_str(), sub.c_str());
        return 1;
    }
    diag("Exit code: %d Total processing time: %ld seconds", rc, (long)(time(nullptr) - t0));
    return rc;
}
