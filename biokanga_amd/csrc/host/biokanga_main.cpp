// biokanga_main.cpp - host front end: `biokanga index` and `biokanga align` (C++ above the C ABI).
//
// Keeps the reference's option letters and defaults for the subset that reaches the hot path
// (biokanga/kanga.cpp:194-294, biokanga/kangax.cpp:95-116), the .sfx index format and the SAM text
// the reference writes (libbiokanga/SAMfile.cpp:1521,1573-1575,1766,2110-2283).  The alignment
// itself is ONLY available through libbiokanga_amd (HIP kernels); there is no host fallback.
//
//   biokanga index -i genome.fa[.gz] [-i more.fa] -o genome.sfx -r name [-l minseqlen] [-d descr] [-t title] [-T threads] [--device n]
//   biokanga align -i reads.fa[.gz] -I genome.sfx -o out.sam [-s subs] [-e 1|2] [-Q 0|1|2] [-m 0..3]
//                  [-n maxNs] [-l minlen] [-L maxlen] [-y trim5] [-Y trim3] [-M 0|5|6] [-O stats.csv]
//                  [-U 1..4 -u mates.fa -d minins -D maxins [-E]] [-T host threads] [-F logfile] [--device n | --devices 0-7]
//   (reads: FASTA / FASTQ, plain, gzip'd or bgzip'd, also through a FIFO; -o: a file, a name ending in .gz or .bam, or a FIFO)
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>
#include <regex.h>
#include <zlib.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cctype>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <memory>
#include <ctime>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/biokanga_amd.h"
#include "../sfx_file.h"
#include "../bk_env.h"
#include "bam_writer.h"
#include "fasta.h"
#include "genome_loader.h"
#include "glibc_rand.h"
#include "mtqsort.h"
#include "multi_assign.h"
#include "post_filters.h"
#include "snp.h"
#include "read_loader.h"
#include "report.h"

namespace {
using namespace bkcli;

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
        fprintf(stderr, "usage: %s index -i <fasta[.gz]> [-i <fasta>..] -o <out.sfx> -r <refspecies> [-l minseqlen] [-d descr] [-t title] [-T threads] [--device n]\n", g_proc.c_str());
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

    // CreateBioseqSuffixFile / ProcessFastaFile (kangax.cpp:545-690,774-926): genome_loader.cpp
    std::vector<std::string> files = a.v["i"];
    std::sort(files.begin(), files.end());                               // SG_GLOB_FULLSORT
    int nthreads = a.num("T", 0);
    if (nthreads <= 0) nthreads = effective_cpus();
    // (BK_TIMING=1: the stages' wall-clock on stderr)
    const bool timing = bk::env::timing();
    timespec ts0;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    auto lap = [&](const char *what) {
        if (!timing) return;
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        fprintf(stderr, "bk timing: index: %-44s %8.1f ms\n", what, 1e3 * (double)(ts.tv_sec - ts0.tv_sec) + 1e-6 * (double)(ts.tv_nsec - ts0.tv_nsec));
        ts0 = ts;
    };
    Genome G;
    if (load_genome(files, min_seq_len, std::max(1, std::min(nthreads, 128)), G)) return 1;
    lap("genome files taken in (parse, N runs)");
    bk::RawVec<uint8_t> &seq = G.seq;
    std::vector<bk::SfxEntry> &entries = G.entries;
    const uint32_t n_under = G.n_under;
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
    lap("HIP start-up, device buffers, bases uploaded");
    int rc = bk_build_sa_device(d_seq, n, d_sa, (int)el, dev);
    if (rc) { diag("Fatal: suffix sort failed: %s", bk_strerror(rc)); return 1; }
    lap("suffix array sorted on the device");
    bk::RawVec<uint8_t> sa;                                              // (12 GB for a human genome: sized, not zeroed first)
    sa.resize(n * el);
    if (hipMemcpy(sa.data(), d_sa, n * el, hipMemcpyDeviceToHost) != hipSuccess) { diag("Fatal: download failed"); return 1; }
    lap("suffix array downloaded (pageable)");
    (void)hipFree(d_seq);
    (void)hipFree(d_sa);
    lap("device buffers given back");
    rc = bk::sfx_write(a.str("o").c_str(), ref, descr, title, entries, seq.data(), n, sa.data(), el, &err, std::max(1, std::min(nthreads, 16)));
    if (rc) { diag("Fatal: %s", err.c_str()); return 1; }
    lap(".sfx written");
    diag("CreateBioseqSuffixFile: completed...");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// align

// BK_TIMING=1: wall-clock of the front end's own stages on stderr (the library prints its own)
struct HostClock {
    bool on = bk::env::timing();
    double t0 = now();
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    static double wall() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return (double)(ts.tv_sec % 60) + 1e-9 * (double)ts.tv_nsec; }      // (the log's seconds)
    void lap(const char *what) { if (!on) return; const double t = now(); fprintf(stderr, "bk timing: host: %-40s %7.1f ms   (done at :%06.3f)\n", what, 1e3 * (t - t0), wall()); t0 = t; }
};

// Everything `biokanga align` was asked for, validated the way kanga.cpp:298-1066 validates it
struct AlignOpts {
    int pe_mode = 0, ml_mode = 0, max_ml = 1, micro_indel = 0, splice_len = 0, min_chim = 0, pcr_win = -1, min_flank = 0, fmt = 5;
    int min_len = 50, max_len = 500, trim5 = 0, trim3 = 0, max_rpt_sam_seqs = 10000, nthreads = 1;
    bool clamp_ml = false, best_matches = false;
    bk_pe_params PE{};
    bk_align_params P{};
    SnpOpts snp;
    std::vector<regex_t> re_excl, re_incl;
    std::vector<int> devices;              // --device n | --devices a,b,c | a-b: one context (+ upload / align / download pipeline) per entry
};

// 0, or 1 after the error message has been logged
int read_align_opts(Args &a, AlignOpts &o)
{
    o.pe_mode = a.num("U", 0);
    if (o.pe_mode < 0 || o.pe_mode > 4) { diag("Error: paired end processing mode '-U%d' must be in range 0..4", o.pe_mode); return 1; }
    if (o.pe_mode && (!a.has("u") || a.v["u"].size() != a.v["i"].size())) {
        diag("Error: paired end processing '-U%d' needs as many '-u' PE2 files as '-i' PE1 files", o.pe_mode);
        return 1;
    }
    o.PE = bk_pe_params{};
    o.PE.pe_mode = o.pe_mode;
    o.PE.pair_min_len = a.num("d", 100);            // cDfltPairMinLen
    o.PE.pair_max_len = a.num("D", 1000);           // cDfltPairMaxLen
    o.PE.pair_strand = a.has("E") ? 1 : 0;
    if (o.pe_mode && (o.PE.pair_min_len < 25 || o.PE.pair_max_len < o.PE.pair_min_len || o.PE.pair_max_len > 100000)) {
        diag("Error: paired end insert size range '-d%d -D%d' not accepted", o.PE.pair_min_len, o.PE.pair_max_len);
        return 1;
    }
    // -r multi-loci modes (kanga.cpp:482-486,535-539,666-694): 0 slough, 1 stats only, 2 random pick, 3 cluster with
    // uniques, 4 cluster with uniques + other multi-loci reads, 5 report all loci; -R loci limit, -X clamp
    o.ml_mode = a.num("r", 0);
    if (o.ml_mode < 0 || o.ml_mode > 5) { diag("Error: multiple aligned reads processing mode '-r%d' specified outside of range 0..5", o.ml_mode); return 1; }
    if (o.pe_mode && o.ml_mode) { diag("Error: Sorry, currently multiloci processing '-r%d' not supported in paired end '-U%d' processing", o.ml_mode, o.pe_mode); return 1; }
    o.max_ml = 1;

    if (o.ml_mode) {
        o.max_ml = a.num("R", 5);                                         // cDfltMaxMultiHits
        const int lim = o.ml_mode == 5 ? 100000 : 500;                    // cMaxAllHits / cMaxMultiHits
        if (o.max_ml < 2 || o.max_ml > lim) { diag("Error: multiple aligned reads '-R%d' specified outside of range 2..%d", o.max_ml, lim); return 1; }
        if (o.max_ml > BK_MAX_ML) { diag("Error: '-R%d' is above the %d loci per read this build keeps", o.max_ml, BK_MAX_ML); return 1; }
        o.best_matches = a.has("N");                                      // bLocateBestMatches (implies the clamp, kanga.cpp:686-694)
        o.clamp_ml = a.has("X") || o.best_matches;
    }
    // -g FASTQ quality scores: 0 Sanger / Illumina 1.8+, 1 Illumina 1.3+, 2 Solexa, 3 ignore (default; QUAL is then '*')
    g_sample_nth = a.num("#", 1);
    if (g_sample_nth < 1 || g_sample_nth > 10000) { diag("Error: sample every Nth raw read '-#%d' specified outside of range 1..10000", g_sample_nth); return 1; }
    g_qual_mode = a.num("g", 3);
    if (g_qual_mode < 0 || g_qual_mode > 3) { diag("Error: fastq quality '-g%d' specified outside of range 0..3", g_qual_mode); return 1; }
    // -a microInDels (kanga.cpp:696-710): looked for in reads the substitution-only phases leave unaligned
    o.micro_indel = a.num("a", 0);
    if (o.micro_indel < 0 || o.micro_indel > 20) { diag("Error: microInDel length maximum '-a%d' specified outside of range 0..20", o.micro_indel); return 1; }
    if (o.micro_indel && o.ml_mode == 5) { diag("Error: microInDels not supported when reporting multiloci alignments"); return 1; }
    if (o.micro_indel && o.pe_mode) { diag("Error: Sorry, currently microInDel processing '-a%d' not supported in paired end '-U%d' processing", o.micro_indel, o.pe_mode); return 1; }   // kanga.cpp:541-545
    // -x: trim aligned reads back from both ends until that many consecutive bases match (CAligner::AutoTrimFlanks)
    // -A RNA-seq splice junctions (kanga.cpp:726-742,810-811): looked for after the microInDel pass; switches flank trimming on
    o.splice_len = a.num("A", 0);
    if (o.splice_len != 0 && (o.splice_len < 25 || o.splice_len > 100000)) { diag("Error: RNAseq maximum splice junction separation '-A%d' must be either 0 or in the range 25..100000", o.splice_len); return 1; }
    if (o.splice_len && o.ml_mode == 5) { diag("Error: in report all multiloci mode '-r5', there is no splice junction processing.."); return 1; }
    if (o.splice_len && o.pe_mode) { diag("Error: Sorry, currently RNA-seq splice junction processing '-A%d' not supported in paired end '-U%d' processing", o.splice_len, o.pe_mode); return 1; }
    // -c chimeric trimming (kanga.cpp:648-664): reads nothing else aligned may be placed with their ends trimmed off
    o.min_chim = a.num("c", 0);
    if (o.min_chim != 0 && (o.min_chim < 50 || o.min_chim > 99)) { diag("Error: minimum chimeric length percentage '-c%d' specified outside of range 50..99", o.min_chim < 0 ? -o.min_chim : o.min_chim); return 1; }
    // (-c with -N: the reference refuses it itself, kanga.cpp:712-716)
    if (o.min_chim && o.ml_mode && a.has("N")) {
        diag("Error: Sorry, chimeric read processing not supported in this release if either SOLiD or locating multiple best matches also requested");
        return 1;
    }
    // (-c with -r1..4 and -a / -A: AlignReads runs the microInDel / splice junction searches between the substitution-only phases and the
    // chimeric call on one set of counts and hits, SfxArrayV2.cpp:7722-7757 - five reference runs byte-identical, tests/golden/chimmlindel)
    // -p / -P / -1 / -S SNP calling (kanga.cpp:866-925)

    o.snp.min_reads = a.num("p", 0);
    if (o.snp.min_reads != 0 && (o.snp.min_reads < 1 || o.snp.min_reads > 100)) { diag("Error: Minimum read coverage at any loci '-p%d' must be in range 1..100", o.snp.min_reads); return 1; }
    if (o.snp.min_reads > 0) {
        o.snp.qvalue = a.has("P") ? atof(a.str("P").c_str()) : 0.0;
        if (o.snp.qvalue < 0.0 || o.snp.qvalue > 0.40) { diag("Error: QValue '-P%1.5f' for controlling SNP FDR (Benjamini-Hochberg) must be in range 0.0 to 0.4", o.snp.qvalue); return 1; }
        if (o.snp.qvalue == 0.0) o.snp.qvalue = 0.05;
        const double pcnt = a.has("1") ? atof(a.str("1").c_str()) : 25.0;
        if (pcnt < 0.1 || pcnt > 35.0) { diag("Error: SNP minimum non-ref '-1%f' for controlling SNP FDR must be in range 0.1 to 35.0", pcnt); return 1; }
        o.snp.nonref_prop = pcnt / 100.0;
        if (o.ml_mode == 5) { diag("Error: SNP processing not currently supported if reporting multiloci alignments"); return 1; }
        o.snp.marker_len = a.num("K", 0);                    // kanga.cpp:928-952
        o.snp.centroid_path = a.str("7", "");                // kanga.cpp:954-960
        if (o.snp.marker_len != 0 && (o.snp.marker_len < 25 || o.snp.marker_len > 500)) { diag("Error: Marker length specified with '-K%d' must be in range 25 to 500", o.snp.marker_len); return 1; }
        if (o.snp.marker_len) {
            o.snp.marker_poly_thres = a.has("G") ? atof(a.str("G").c_str()) : (1.0 / 3.0);
            if (o.snp.marker_poly_thres < 0.0 || o.snp.marker_poly_thres > 0.50) { diag("Error: Max marker sequence base poymorphism specified with '-G%1.3f' must be in range 0.0 to 0.5", o.snp.marker_poly_thres); return 1; }
        }
    }
    // -k PCR differential amplification artefact reduction (kanga.cpp:718-724): window 0..250, off by default
    o.pcr_win = a.has("k") ? a.num("k", -1) : -1;
    if (a.has("k") && (o.pcr_win < 0 || o.pcr_win > 250)) { diag("Error: PCR differential amplification artefacts window length '-k%d' specified outside of range 0..250", o.pcr_win); return 1; }
    // -Z / -z chromosome exclude / include filters: POSIX extended regular expressions, case-insensitive (Aligner.cpp:4770-4795)

    for (const char *opt : {"Z", "z"})
        if (a.has(opt))
            for (const std::string &pat : a.v[opt]) {
                regex_t re;
                if (regcomp(&re, pat.c_str(), REG_EXTENDED | REG_ICASE)) { diag("Error: ProcessAlign: %s chrom RE '%s' error", opt[0] == 'Z' ? "exclude" : "include", pat.c_str()); return 1; }
                (opt[0] == 'Z' ? o.re_excl : o.re_incl).push_back(re);
            }
    // (with -U the reference consults the filters inside its pair rules, AcceptThisChromID at Aligner.cpp:2771-2786,3224,3323,3445: not built)
    o.min_flank = a.num("x", 0);
    if (o.min_flank < 0 || o.min_flank > 7) { diag("Error: Max flank trimming '-x%d' specified outside of range 0..7", o.min_flank); return 1; }      // cMaxAllowedSubs / 2
    o.P = bk_align_params{};
    // with -N the reads go through LocateBestMatches, which has no microInDel / splice junction branches (Aligner.cpp:9197-9218): the
    // options are accepted as the reference accepts them and only their host-side consequences remain (-A still switches flank trimming on)
    o.P.micro_indel_len = o.best_matches ? 0 : o.micro_indel;
    o.P.splice_junct_len = o.best_matches ? 0 : o.splice_len;
    o.P.pmode = a.num("m", 0);
    o.P.align_strand = a.num("Q", 0);
    o.P.min_edit_dist = a.num("e", 1);
    o.P.max_subs = a.num("s", 10);                  // cDfltAllowedSubs per 100bp
    if (o.splice_len > 0 && o.min_chim == 0 && o.min_flank == 0) o.min_flank = o.P.max_subs;      // MinFlankExacts = MaxSubs (kanga.cpp:810-811)
    o.P.min_chimeric_len = o.min_chim;
    o.P.max_ns = a.num("n", 1);
    o.P.max_ml = o.max_ml;
    o.P.clamp_ml = o.clamp_ml ? 1 : 0;
    o.P.best_matches = o.best_matches ? 1 : 0;
    o.fmt = a.num("M", 5);
    if (o.ml_mode == 5 && !(o.fmt == 0 || o.fmt == 4 || o.fmt == 5 || o.fmt == 6)) {      // kanga.cpp:830-834
        diag("Error: reporting all multiloci alignments '-r5' is only available with output formats '-M0', '-M4', '-M5' and '-M6'");
        return 1;
    }
    if (a.has("O") && o.fmt == 6) {                 // kanga.cpp:1015-1021
        diag("Error: Output induced substitution mode '-O<file>' not available in '-M6' output mode\n");
        return 1;
    }
    o.min_len = a.num("l", 50), o.max_len = a.num("L", 500);
    o.trim5 = a.num("y", 0), o.trim3 = a.num("Y", 0);
    o.max_rpt_sam_seqs = a.num("4", 10000);
    // -T: host threads for parsing, sorting and formatting (0 = all cores, capped like the reference's cMaxWorkerThreads)
    o.nthreads = a.num("T", 0);
    if (o.nthreads <= 0) o.nthreads = effective_cpus();                  // the cores the process may really use (cgroup quota included)
    if (o.nthreads < 1) o.nthreads = 1;
    if (o.nthreads > 128) o.nthreads = 128;
    if (o.P.pmode < 0 || o.P.pmode > 3 || o.P.align_strand < 0 || o.P.align_strand > 2 || o.P.min_edit_dist < 1 || o.P.min_edit_dist > 2 ||
        o.P.max_subs < 0 || o.P.max_subs > 25 || o.P.max_ns < 0 || o.P.max_ns > 5 || o.min_len < 15 || o.min_len > 2000 || o.max_len < o.min_len ||
        o.max_len > 2000 || o.fmt < 0 || o.fmt > 6) {
        diag("Error: an option value is outside its accepted range");
        return 1;
    }
    // --device n (one GPU) or --devices 0-7 / 0,1,2: the read batches are dealt round-robin over one context per entry
    if (a.has("devices")) {
        const std::string spec = a.str("devices");
        size_t at = 0;
        while (at < spec.size()) {
            size_t end = spec.find(',', at);
            if (end == std::string::npos) end = spec.size();
            const std::string part = spec.substr(at, end - at);
            const size_t dash = part.find('-');
            const int lo = atoi(part.c_str()), hi = dash == std::string::npos ? lo : atoi(part.c_str() + dash + 1);
            if (part.empty() || lo < 0 || hi < lo || hi > 1023) { diag("Error: '--devices %s' is not a list of device numbers / ranges", spec.c_str()); return 1; }
            for (int d = lo; d <= hi; d++) o.devices.push_back(d);
            at = end + 1;
        }
    } else
        o.devices.push_back(a.num("device", 0));
    if (o.devices.empty() || o.devices.size() > 64) { diag("Error: between 1 and 64 devices can be used"); return 1; }
    return 0;
}

// Long runs - from this many reads per device on - are cut into larger batches.
constexpr unsigned long long kLongRunMinReads = 600000000ULL;
// The index image a job gets - the suffix-ordered window array (a sixth of a 3.1 Gbp index, 25 GB), the k-mer table's second words, the
// third- and fourth-level search keys: all made behind the suffix array's upload, slice by slice while the next slice crosses PCIe -
// follows the library's one rule, bk_image_policy(reads per device): everything from BK_POLICY_MIN_READS reads on (0.1 - 0.2 s more load
// on a 3.1 Gbp index for 0.7 ns less per read of a hundred bases), the lean image that grows should the run turn out long below.
// bench.py measures its headline on the image this rule picks.

// what the alignment pass leaves for the policies above the boundary
struct AlignedSet {
    std::vector<bk_hit> hits;                      // one per read, load order
    std::vector<uint64_t> l_offs;                  // multi-loci modes: read i owns loci [l_offs[i], l_offs[i+1])
    std::vector<bk_loci> loci;
    std::vector<bk_loci_trims> loci_trims;         // -c with the multi-loci modes: end trims of every locus (else empty)
    std::vector<bk_seg2> seg2;                     // -a / -A / -c: second segment of each read (flags 0 = none)
    std::vector<uint64_t> seq_counts;              // per sequence: reads the SE pass accepted, summed over the devices (RCCL when > 1)
};


// The reads as they cross the boundary (the reference's loader hands its workers 1 byte/base, Aligner.cpp:9038-9055, of which three bits
// reach the hot path): 2 bit/base words + 16-bit lengths + the bases that are not a,c,g,t (bk_pack_reads), cut into the batches the
// pipelines will be fed, in page-locked memory - made by the host threads while the index image is still on its way to the device.
struct Submission {
    struct Batch { size_t lo, hi; uint64_t w0, w1, e0, e1, ticket; };
    std::vector<Batch> batches;
    uint32_t *words = nullptr;
    uint16_t *lens16 = nullptr;
    bk_nbase *exc = nullptr;
    uint64_t n_words = 0, cap_exc = 0, max_words = 1;
    uint64_t cap_words = 0, cap_lens = 0;          // what `early` made room for
    size_t max_reads = 1;
    uint32_t max_len = 0;
    void *registered = nullptr;                    // the result array, page-locked in place while results arrive in it
    // Page-locking memory is slow (0.2 s per GB with the parser's threads running): for plain-text inputs the buffers are made by a
    // thread of their own from the moment the input files' sizes are known - a read's words cannot outnumber its bytes in the file / 16
    std::thread early;
    void start_early(uint64_t input_bytes, uint32_t min_len)
    {
        early = std::thread([this, input_bytes, min_len]() {
            const uint64_t w = input_bytes / 16 + 4096, r = input_bytes / ((uint64_t)min_len + 3) + 4096, e = input_bytes / 448 + (1u << 20);
            words = (uint32_t *)bk_host_alloc((w + 64) * 4);
            lens16 = (uint16_t *)bk_host_alloc((r + 64) * 2);
            exc = (bk_nbase *)bk_host_alloc(e * sizeof(bk_nbase));
            if (words && lens16 && exc) { cap_words = w; cap_lens = r; cap_exc = e; }
            else { bk_host_free(words); bk_host_free(lens16); bk_host_free(exc); words = nullptr; lens16 = nullptr; exc = nullptr; }
        });
    }
    void release_results() { if (registered) { bk_host_unregister(registered); registered = nullptr; } }
    // the packed reads have served once the last batch is back: their page-locked memory is given back (0.18 s per GB) by a thread of its
    // own while the records are sorted, instead of at the end of the run
    std::thread releaser;
    std::vector<bk_nbase> exc_job;                 // the exceptions of all batches with `read` counting from the first read of the run
    // (with a SAM head start that reads the packed form: once that has taken it to the device; kept if the head start failed - the
    // formatter then uploads for itself)
    void release_packed_in_background(bk_sam_prep *after = nullptr, std::function<void()> then = nullptr)
    {
        if (early.joinable()) early.join();
        uint32_t *w = words; uint16_t *l = lens16; bk_nbase *e = exc;
        words = nullptr; lens16 = nullptr; exc = nullptr; cap_words = cap_lens = cap_exc = 0;
        looked = std::async(std::launch::deferred, []() {});       // (reset below when there is a head start to look at)
        if (after != nullptr) {
            auto seen = std::make_shared<std::promise<void>>();
            looked = seen->get_future();
            releaser = std::thread([w, l, e, after, seen, then]() {
                const int rc = bk_sam_prep_wait(after);
                seen->set_value();                      // (`after` may be consumed from here on)
                if (rc != BK_OK) return;
                if (then) then();
                bk_host_free(w); bk_host_free(l); bk_host_free(e);
            });
        } else
            releaser = std::thread([w, l, e]() { bk_host_free(w); bk_host_free(l); bk_host_free(e); });
    }
    std::future<void> looked;                      // ready once the releaser no longer needs the head start's handle
    void done_with_head_start() { if (looked.valid()) looked.wait(); }
    ~Submission()
    {
        if (early.joinable()) early.join();
        if (releaser.joinable()) releaser.join();
        release_results(); bk_host_free(words); bk_host_free(lens16); bk_host_free(exc);
    }
};

// Reads per batch.  A batch costs less per read the larger it is (every phase's wave-per-read launch lasts at least as long as its
// heaviest read: 3 M reads take 3.3 ns each, 12 M 2.2 ns, 50 M 2.0 ns) - but its scratch is allocated at 16 ms per GB (0.5 KB per
// read), and nothing overlaps the upload of a device's first batch.  Runs below the window array's threshold are over in a fraction
// of a second: 4 M reads first, then batches of 12 M; long runs grow to 32 M.  Small inputs: two batches per device.
int prepare_submission(const AlignOpts &o, const ReadStore &rs, size_t ndev, bool long_run, AlignedSet &A, Submission &S)
{
    const size_t nr = rs.size();
    HostClock clk;
    std::thread results([&]() {                    // (the record array is touched page by page: a thread of its own)
        HostClock c2;
        A.hits.reserve(nr);                        // (room first, marked for huge pages, then the records: 500 page faults per GB instead of 260 000)
        if (nr * sizeof(bk_hit) >= bk::kHugeFrom) {
            const uintptr_t lo = ((uintptr_t)A.hits.data() + bk::kHugePage - 1) & ~(uintptr_t)(bk::kHugePage - 1);
            const uintptr_t hi = ((uintptr_t)A.hits.data() + nr * sizeof(bk_hit)) & ~(uintptr_t)(bk::kHugePage - 1);
            if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
        }
        A.hits.resize(nr);
        c2.lap("result array sized");
        if (nr && bk_host_register(A.hits.data(), nr * sizeof(bk_hit)) == BK_OK) S.registered = A.hits.data();
        c2.lap("result array page-locked");
    });
    size_t per = long_run ? (32u << 20) : (12u << 20);
    const bool ramp = nr / ndev >= (12u << 20);
    if (!ramp) {
        per = 8u << 20;
        if (nr / ndev / 2 + 1 < per) per = std::max<size_t>(65536, nr / ndev / 2 + 1);
    }
    per += per & 1;
    for (size_t lo = 0, k = 0; lo < nr; k++) {
        size_t n = per;
        if (ramp) {
            const size_t round = k / ndev;                                           // this device's round-th batch
            n = round == 0 ? (4u << 20) : (round == 1 ? (12u << 20) : per);
        }
        if (nr - lo < n + n / 4) n = nr - lo;                                        // (no small last batch: it joins the one before)
        n += n & 1;
        Submission::Batch b{lo, std::min(nr, lo + n), 0, 0, 0, 0, 0};
        S.max_reads = std::max(S.max_reads, b.hi - b.lo);
        S.batches.push_back(b);
        lo = b.hi;
    }
    // words per batch (a read takes ceil(len / 16) of them), longest read
    {
        const int nt = std::max(1, std::min(o.nthreads, (int)S.batches.size()));
        std::vector<uint32_t> ml(S.batches.size(), 0);
        std::atomic<size_t> next{0};
        auto count = [&]() {
            for (size_t k; (k = next.fetch_add(1)) < S.batches.size();) {
                Submission::Batch &b = S.batches[k];
                uint64_t w = 0;
                uint32_t m = 0;
                for (size_t i = b.lo; i < b.hi; i++) { w += ((uint64_t)rs.lens[i] + 15) >> 4; m = std::max(m, rs.lens[i]); }
                b.w1 = w;
                ml[k] = m;
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(count);
        count();
        for (auto &t : th) t.join();
        uint64_t at = 0;
        for (size_t k = 0; k < S.batches.size(); k++) {
            Submission::Batch &b = S.batches[k];
            S.max_words = std::max(S.max_words, b.w1);
            b.w0 = at; at += b.w1; b.w1 = at;
            S.max_len = std::max(S.max_len, ml[k]);
        }
        S.n_words = at;
    }
    clk.lap("batches cut, words counted");
    if (S.early.joinable()) S.early.join();
    if (!(S.words && S.cap_words >= S.n_words && S.cap_lens >= nr)) {          // no head start, or one that guessed too low
        bk_host_free(S.words); bk_host_free(S.lens16); bk_host_free(S.exc);
        S.words = (uint32_t *)bk_host_alloc((S.n_words + 64) * 4);
        S.lens16 = (uint16_t *)bk_host_alloc((nr + 64) * 2);
        S.cap_exc = nr / 4 + (1u << 20);
        S.exc = (bk_nbase *)bk_host_alloc(S.cap_exc * sizeof(bk_nbase));
    }
    int rc = (S.words && S.lens16 && S.exc) ? BK_OK : BK_ERR_MEM;
    clk.lap("page-locked buffers for the packed reads");
    uint64_t e_at = 0;
    for (size_t k = 0; k < S.batches.size() && rc == BK_OK; k++) {
        Submission::Batch &b = S.batches[k];
        uint64_t n_exc = 0;
        rc = bk_pack_reads(rs.bases.data(), rs.offs.data() + b.lo, rs.lens.data() + b.lo, (uint32_t)(b.hi - b.lo), S.words + b.w0, S.lens16 + b.lo,
                           S.exc + e_at, S.cap_exc - e_at, &n_exc);
        if (rc == BK_ERR_MEM) {
            // more bases that are not a,c,g,t than room was made for: a larger list (what is packed so far is kept), this batch again
            const uint64_t cap2 = std::max<uint64_t>(2 * S.cap_exc, e_at + n_exc + (1u << 20));
            bk_nbase *e2 = (bk_nbase *)bk_host_alloc(cap2 * sizeof(bk_nbase));
            if (!e2) break;
            memcpy(e2, S.exc, e_at * sizeof(bk_nbase));
            bk_host_free(S.exc);
            S.exc = e2;
            S.cap_exc = cap2;
            rc = BK_OK;
            k--;
            continue;
        }
        b.e0 = e_at; e_at += n_exc; b.e1 = e_at;
    }
    if (rc == BK_OK) {
        S.exc_job.assign(S.exc, S.exc + e_at);
        for (const Submission::Batch &b : S.batches)
            for (uint64_t e = b.e0; e < b.e1; e++) S.exc_job[e].read += (uint32_t)b.lo;
    }
    clk.lap("reads packed to 2 bit/base");
    results.join();
    clk.lap("waited for the result array");
    if (rc) diag("Fatal: unable to pack the reads for the device: %s", bk_strerror(rc));
    return rc;
}

// CAligner::LocateCoredApprox over every loaded read: the packed batches go through one upload / align / download pipeline per device
// (bk_stream_*), batch b on device b mod N; with -U the paired-end association runs on the device-resident batch right after its SE
// pass (batches hold whole pairs).  Results land in load order whatever the number of devices, so everything downstream - the
// reference's sort order included - is the same as on one GPU.  The pipelines (their device buffers, the contexts' batch scratch)
// are set up by open_pipelines() BEFORE the clock of T_align starts: align_reads() only submits and collects.
int open_pipelines(const std::vector<bk_ctx *> &ctxs, const AlignOpts &o, const Submission &S, std::vector<bk_stream *> &st)
{
    st.assign(ctxs.size(), nullptr);
    std::vector<int> rcs(ctxs.size(), 0);
    std::vector<std::thread> th;
    for (size_t d = 0; d < ctxs.size(); d++)
        th.emplace_back([&, d]() {
            rcs[d] = bk_ctx_reserve(ctxs[d], (uint32_t)S.max_reads, std::max<uint32_t>(S.max_len, 16));
            if (!rcs[d]) rcs[d] = bk_stream_create_packed(&st[d], ctxs[d], (uint32_t)S.max_reads, S.max_words, 3, o.pe_mode ? &o.PE : nullptr);
        });
    for (auto &t : th) t.join();
    for (size_t d = 0; d < ctxs.size(); d++)
        if (rcs[d]) {
            diag("Fatal: unable to set up the device pipeline: %s", bk_strerror(rcs[d]));
            for (bk_stream *s : st) bk_stream_destroy(s);
            st.clear();
            return rcs[d];
        }
    return BK_OK;
}

int align_reads(const std::vector<bk_ctx *> &ctxs, std::vector<bk_stream *> &st, const AlignOpts &o, const ReadStore &rs, Submission &S, AlignedSet &A)
{
    const size_t ndev = ctxs.size();
    const bool lists = o.ml_mode != 0, segs = o.P.micro_indel_len || o.P.splice_junct_len || o.P.min_chimeric_len;
    if (lists) A.l_offs.assign(1, 0);
    std::vector<Submission::Batch> &batches = S.batches;
    auto close = [&]() { for (bk_stream *s : st) bk_stream_destroy(s); st.clear(); };
    auto collect = [&](size_t k) -> int {
        Submission::Batch &b = batches[k];
        bk_stream *s = st[k % ndev];
        int rc = bk_stream_wait(s, b.ticket);
        if (rc) { diag("Fatal: alignment failed: %s", bk_strerror(rc)); return rc; }
        const size_t n = b.hi - b.lo;
        if (segs) {
            const bk_seg2 *bs = nullptr;
            uint64_t ns = 0;
            rc = bk_stream_batch_seg2(s, b.ticket, &bs, &ns);
            if (rc || !bs || ns != n) { diag("Fatal: microInDel segments unavailable: %s", bk_strerror(rc)); return rc ? rc : BK_ERR_INTERNAL; }
            A.seg2.insert(A.seg2.end(), bs, bs + ns);
        }
        if (lists) {
            const uint64_t *bo = nullptr;
            const bk_loci *bl = nullptr;
            uint64_t nl = 0;
            rc = bk_stream_batch_loci(s, b.ticket, &bo, &bl, &nl);
            if (rc || !bo) { diag("Fatal: loci lists unavailable: %s", bk_strerror(rc)); return rc ? rc : BK_ERR_INTERNAL; }
            const uint64_t base = A.loci.size();
            for (size_t i = 1; i <= n; i++) A.l_offs.push_back(base + bo[i]);
            A.loci.insert(A.loci.end(), bl, bl + nl);
            if (o.P.min_chimeric_len) {
                const bk_loci_trims *bt = nullptr;
                uint64_t nt = 0;
                rc = bk_stream_batch_loci_trims(s, b.ticket, &bt, &nt);
                if (rc || (nl && (!bt || nt != nl))) { diag("Fatal: loci trims unavailable: %s", bk_strerror(rc)); return rc ? rc : BK_ERR_INTERNAL; }
                A.loci_trims.insert(A.loci_trims.end(), bt, bt + nt);
            }
        }
        if (segs || lists) bk_stream_release(s, b.ticket);
        return BK_OK;
    };
    // submit() blocks while a device's three buffer sets are busy, and a batch completes without being waited for, so the
    // loop below keeps every pipeline full; results are collected in load order, 3 * ndev batches behind the submissions
    size_t next_collect = 0;
    const size_t lag = 3 * ndev;
    for (size_t k = 0; k < batches.size(); k++) {
        Submission::Batch &b = batches[k];
        int rc = bk_stream_submit_packed(st[k % ndev], S.words + b.w0, b.w1 - b.w0, S.lens16 + b.lo, (uint32_t)(b.hi - b.lo), S.exc + b.e0, b.e1 - b.e0,
                                         A.hits.data() + b.lo, &b.ticket);
        if (rc) { diag("Fatal: alignment failed: %s", bk_strerror(rc)); close(); return rc; }
        while (next_collect + lag <= k) { rc = collect(next_collect++); if (rc) { close(); return rc; } }
    }
    while (next_collect < batches.size()) { int rc = collect(next_collect++); if (rc) { close(); return rc; } }
    // what crossed PCIe, and the pipelines' own clock (first batch submitted -> last result back, the slowest device)
    {
        bk_stream_stats tot{};
        for (bk_stream *s : st) {
            bk_stream_stats x{};
            if (bk_stream_get_stats(s, &x, 0) == BK_OK) {
                tot.batches += x.batches; tot.reads += x.reads; tot.bytes_h2d += x.bytes_h2d; tot.bytes_d2h += x.bytes_d2h;
                tot.seconds_first_submit_to_last_result = std::max(tot.seconds_first_submit_to_last_result, x.seconds_first_submit_to_last_result);
            }
        }
        if (tot.reads)
            diag("Device pipeline: %llu reads in %llu batches over %zu device(s), %.1f bytes per read host to device (2 bit/base), %.1f back, first submit to last result %.3f seconds",
                 (unsigned long long)tot.reads, (unsigned long long)tot.batches, ndev, (double)tot.bytes_h2d / (double)tot.reads, (double)tot.bytes_d2h / (double)tot.reads,
                 tot.seconds_first_submit_to_last_result);
    }
    close();
    // (the result array stays page-locked while nothing replaces it: the SAM formatter's upload of the records then is one DMA)
    if (o.ml_mode) S.release_results();
    // the path's one exchange step: per-sequence accepted-read counts summed over the devices
    A.seq_counts.assign(bk_num_entries(ctxs[0]), 0);
    int rc = bk_seq_counts_allreduce((bk_ctx *const *)ctxs.data(), (int)ndev, A.seq_counts.data(), (uint32_t)A.seq_counts.size(), 1);
    if (rc) { diag("Fatal: unable to reduce the per sequence hit counts over the devices: %s", bk_strerror(rc)); return rc; }
    return BK_OK;
}

// CAligner::ProcCoredApprox for MLMode != eMLdefault (Aligner.cpp:9241-9424): what becomes of the reads that aligned to more than one
// locus (-r1 statistics, -r2 random pick, -r3 / -r4 clustering, -r5 every locus its own record).  `src` maps the records of -r5 back
// to their reads; `nr` is the record count afterwards.
// `rec_trims` (with -c only): per record the end trims its placement carries - those of the locus chosen for it, or of the read's own
// unique chimeric placement.
void resolve_multi_loci(const AlignOpts &o, const ReadStore &rs, AlignedSet &A, std::vector<uint32_t> &src, std::vector<int> &multi_dist, size_t &nr,
                        std::vector<bk_loci_trims> &rec_trims, const std::vector<uint8_t> *chrom_ok)
{
    std::vector<bk_hit> &hits = A.hits;
    const std::vector<uint64_t> &l_offs = A.l_offs;
    const std::vector<bk_loci> &loci = A.loci;
    const bool with_trims = o.P.min_chimeric_len > 0;
    auto trims_of = [&](uint64_t locus) -> bk_loci_trims { return locus < A.loci_trims.size() ? A.loci_trims[locus] : bk_loci_trims{}; };
    if (with_trims) {
        rec_trims.assign(nr, bk_loci_trims{});
        for (size_t i = 0; i < nr && i < A.seg2.size(); i++)
            if (A.seg2[i].flags & 8) { rec_trims[i].left = A.seg2[i].match_len; rec_trims[i].right = A.seg2[i].read_ofs; rec_trims[i].chimeric = 1; }
    }
    // CAligner::ProcCoredApprox for MLMode != eMLdefault (Aligner.cpp:9241-9424); reads in load order, as -T1 runs them
    auto eff_count = [&](size_t i) -> uint32_t {            // LowHitInstances of a read that counts as eHRhits (after -X)
        const bk_hit &h = hits[i];
        if (h.rslt == BK_HR_HITS || (o.clamp_ml && h.rslt == BK_HR_HITINSTS)) return (uint32_t)(l_offs[i + 1] - l_offs[i]);
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
        if (o.ml_mode != 5) multi_dist[c - 1]++;
        if (o.clamp_ml && hits[i].rslt == BK_HR_HITINSTS) hits[i].low_hit_instances = (int16_t)c;
    }
    diag("Provisionally accepted %llu aligned reads (%llu uniquely, %llu aligning to multiloci) aligning to a total of %llu loci",
         (unsigned long long)(n_uniq + n_multi), (unsigned long long)n_uniq, (unsigned long long)n_multi, (unsigned long long)n_loci);
    if (o.ml_mode == 2) {
        // eMLrand: rand() % LowHitInstances for every eHRhits read, unique ones included (:9365-9366); the sequence is
        // glibc's unseeded one (glibc_rand.h), which is what a single-threaded reference run consumes in the same order
        bk::GlibcRand pick;
        for (size_t i = 0; i < nr; i++) {
            const uint32_t c = eff_count(i);
            if (!c) continue;
            const uint32_t k = (uint32_t)pick.next() % c;
            take(hits[i], loci[l_offs[i] + k]);
            if (with_trims) rec_trims[i] = trims_of(l_offs[i] + k);
        }
    } else if (o.ml_mode == 3 || o.ml_mode == 4) {
        uint32_t max_reads_len = 0;
        for (size_t i = 0; i < nr; i++) max_reads_len = std::max(max_reads_len, rs.lens[i]);
        bk::MultiAssign ma;
        for (size_t i = 0; i < nr; i++) {
            const uint32_t c = eff_count(i);
            for (uint32_t k = 0; k < c; k++) {
                const bk_loci_trims t = with_trims ? trims_of(l_offs[i] + k) : bk_loci_trims{};
                ma.add((uint32_t)i + 1, loci[l_offs[i] + k], c > 1, t.left, t.right, (uint32_t)(l_offs[i] + k));
            }
        }
        diag("Assigning %llu reads which aligned to multiple loci to a single loci", (unsigned long long)n_multi);
        bk::MultiAssignStats st = ma.assign(o.ml_mode == 3, o.nthreads, max_reads_len);
        for (const bk::MultiHitRec &m : ma.recs)
            if (m.multi && m.assigned) {
                take(hits[m.read_id - 1], m.loci);
                if (with_trims) rec_trims[m.read_id - 1] = trims_of(m.src);
            }
        diag("Clustering completed, removed %d unclustered orphans from %d putative resulting in %d (%d clustered near unique, %d clustered near other multiloci reads) multihit reads accepted as assigned",
             st.putative - st.assigned, st.putative, st.assigned, st.near_unique, st.near_multi);
    } else if (o.ml_mode == 5) {
        // eMLall: every locus becomes a record of its own, ReadID = order of creation (CAligner::WriteHitLoci / AddMultiHit,
        // Aligner.cpp:6666-6800); with -M6 reads without alignment (eHRnone, eHRHitInsts) are kept as one unaligned
        // record, everything else (EN, MMDelta) drops out (:9311-9352,9441-9449)
        std::vector<bk_hit> recs;
        std::vector<bk_loci_trims> rt;
        for (size_t i = 0; i < nr; i++) {
            const bk_hit &h = hits[i];
            const uint32_t c = eff_count(i);
            if (c) {
                // loci on sequences the -Z / -z filters reject never become records (CAligner::WriteHitLoci, Aligner.cpp:6739-6760); a read
                // that loses all of them is kept as one record without a hit in '-M6' only
                // The reference compacts the accepted loci in place with `if(pAcceptHit != pHit) *pAcceptHit++ = *pHit;` (:6748-6752): the
                // write position only moves when it differs from the read position, so an accepted locus at the front is overwritten by the
                // next accepted one and the LAST one kept comes out twice.  Reproduced as it stands.
                uint32_t kept = c;
                std::vector<uint32_t> pick(c);
                for (uint32_t k = 0; k < c; k++) pick[k] = k;
                if (chrom_ok) {
                    uint32_t acc = 0;
                    kept = 0;
                    for (uint32_t k = 0; k < c; k++) {
                        if (!(*chrom_ok)[loci[l_offs[i] + pick[k]].chrom_id]) continue;
                        kept++;
                        if (acc != k) pick[acc++] = pick[k];
                    }
                }
                for (uint32_t q = 0; q < kept; q++) {
                    const uint32_t k = pick[q];
                    bk_hit r = h;
                    take(r, loci[l_offs[i] + k]);
                    recs.push_back(r);
                    src.push_back((uint32_t)i);
                    if (with_trims) rt.push_back(trims_of(l_offs[i] + k));
                }
                if (!kept && o.fmt == 6) {
                    // (the record the reference keeps of such a read comes out of its FiltByChroms pass as "aligned to filtered target
                    // sequence": observed on the reference's own output, tests/golden/multi/r5R5ZmB.m6.sam.gz)
                    bk_hit r = h;
                    r.nar = 11;
                    r.num_hits = 0;
                    r.low_hit_instances = 0;
                    recs.push_back(r);
                    src.push_back((uint32_t)i);
                    if (with_trims) rt.push_back(bk_loci_trims{});
                }
            } else if (o.fmt == 6 && h.nar != BK_NAR_NS && (h.rslt == BK_HR_NONE || h.rslt == BK_HR_HITINSTS)) {
                bk_hit r = h;
                r.num_hits = 0;
                r.low_mm = 0;
                recs.push_back(r);
                src.push_back((uint32_t)i);
                if (with_trims) rt.push_back(bk_loci_trims{});
            }
        }
        if (with_trims) rec_trims.swap(rt);
        diag("Treating accepted %llu multialigned reads as uniquely aligned %llu source reads in subsequent processing",
             (unsigned long long)n_multi, (unsigned long long)(n_loci - n_uniq));
        hits.swap(recs);
        nr = hits.size();
    }
}

time_t g_t0 = 0;
// The last line of the log, then the process ends at once: every output file is closed and synced by now, and what the destructors of a
// large run would do - hand 20 GB of read store, records and page-locked buffers back page by page, take the HIP runtime down - the
// exit does wholesale (0.9 s of a 5.5 s run went there).
[[noreturn]] void end_process(int rc)
{
    diag("Exit code: %d Total processing time: %ld seconds", rc, (long)(time(nullptr) - g_t0));
    if (bk::env::timing()) {                     // what the exit has to give back
        if (FILE *f = fopen("/proc/self/smaps_rollup", "r")) {
            char line[256];
            while (fgets(line, sizeof line, f))
                if (!strncmp(line, "Rss:", 4) || !strncmp(line, "Anonymous:", 10) || !strncmp(line, "AnonHugePages:", 14) || !strncmp(line, "Shared_", 7) || !strncmp(line, "Private_", 8) || !strncmp(line, "ShmemPmdMapped:", 15))
                    fprintf(stderr, "bk timing: host: at exit %s", line);
            fclose(f);
        }
    }
    fflush(nullptr);
    _exit(rc);
}

int cmd_align(int argc, char **argv, int first)
{
    Args a;
    std::string err;
    std::map<std::string, std::string> ln = {
        {"mode", "m"}, {"alignstrand", "Q"}, {"editdelta", "e"}, {"substitutions", "s"}, {"maxns", "n"}, {"trim5", "y"},
        {"trim3", "Y"}, {"minacceptreadlen", "l"}, {"maxacceptreadlen", "L"}, {"format", "M"}, {"in", "i"}, {"sfx", "I"},
        {"out", "o"}, {"stats", "O"}, {"threads", "T"}, {"log", "F"}, {"FileLogLevel", "f"}, {"pemode", "U"}, {"mlmode", "r"},
        {"quality", "g"}, {"device", "device"}, {"devices", "devices"}, {"window-array", "window-array"}, {"index-image", "index-image"}, {"rptsamseqsthres", "4"}, {"pair", "u"}, {"pairminlen", "d"}, {"pairmaxlen", "D"},
        {"pairstrand", "E"}, {"nonealign", "j"}, {"multialign", "J"}, {"title", "t"}, {"maxmulti", "R"}, {"clampmaxmulti", "X"},
        {"bestmatches", "N"}, {"microindellen", "a"}, {"minflankexacts", "x"}, {"splicejunctlen", "A"}, {"minchimeric", "c"}, {"pcrwin", "k"}, {"samplenthrawread", "#"}, {"chromexclude", "Z"}, {"chromeinclude", "z"},
        {"minsnpreads", "p"}, {"qvalue", "P"}, {"snpnonrefpcnt", "1"}, {"snpfile", "S"}, {"markerlen", "K"}, {"markerpolythres", "G"}, {"snpcentroid", "7"}};
    if (!parse_args(argc, argv, first, ln, "mQesnyYlLMiIoOTFfUrg4udDjJtRaxAck#ZzpP1SKG7", "EXN", a, err)) {
        fprintf(stderr, "%s align: %s\n", g_proc.c_str(), err.c_str());
        return 1;
    }
    if (!a.has("i") || !a.has("I") || !a.has("o")) {
        fprintf(stderr, "usage: %s align -i <reads> -I <genome.sfx> -o <out.sam> [-s subs] [-e delta] [-Q strand] [-m mode] [-n maxNs] "
                        "[-l minlen] [-L maxlen] [-M 0|5|6] [-O stats] [--device n | --devices a-b] [--window-array on|off] [--index-image lean|full]\n", g_proc.c_str());
        return 1;
    }
    if (a.has("F")) g_logfile = fopen(a.str("F").c_str(), "a");
    diag("Subprocess align Version %s starting", kProgVer);
    AlignOpts o;
    if (read_align_opts(a, o)) return 1;

    // index images travel to the devices (one loader thread each) while this thread parses the reads: the two longest serial steps
    // of a run overlap (the reference loads its reads in the background of the alignment instead, Aligner.cpp:4820-4860)
    diag("Loading suffix array file '%s'", a.str("I").c_str());
    const size_t ndev = o.devices.size();
    std::vector<bk_ctx *> ctxs(ndev, nullptr);
    std::vector<int> ctx_rc(ndev, 0);
    std::vector<std::thread> loaders;
    // the first device reads the .sfx and builds the tables; the others receive the finished image device to device (xGMI)
    // The image pays from BK_POLICY_MIN_READS reads per device on (bk_image_policy).  The decision is made HERE, from the size of
    // the input files, before a single read is parsed: the array then comes with the index image (made behind the slices of the suffix
    // array's upload) instead of in front of the first batch.  A FASTA record of a 100-base read is about 120 bytes, a FASTQ one about
    // 250; gzip'd files hold about four times their size.
    uint64_t est_reads = 0, plain_bytes = 0;
    bool all_plain = true;
    for (const char *opt : {"i", "u"})
        if (a.has(opt))
            for (const std::string &fn : a.v[opt]) {
                // (gzip'd files too: they are inflated whole like plain ones are mapped, and say - bgzip'd ones exactly - how much text they hold)
                const uint64_t text = bk::text_bytes_estimate(fn);
                if (!text) continue;
                const bool fq = fn.find(".fq") != std::string::npos || fn.find(".fastq") != std::string::npos;
                est_reads += text / (fq ? 250 : 120);
                plain_bytes += text;
                all_plain = all_plain && o.nthreads > 1;
            }
    Submission S;
    if (all_plain && plain_bytes >= (256u << 20)) S.start_early(plain_bytes, (uint32_t)std::max(15, o.min_len));
    // SAM text goes into a file of known approximate size.  A large plain-text input tells that size now (a record is the read's name and
    // bases - and qualities - as the input holds them plus about 31 bytes, 55 for a paired end; a FASTA record of 100 bases is 117 bytes):
    // the file's pages are allocated, zeroed and mapped by background threads from here on - 7 GB for 50 M reads of 100 bases, which the
    // writers would otherwise wait for.  Other inputs: once the reads are loaded (below).
    SamPrealloc pre;
    const std::string opath0 = a.str("o");
    struct stat ost;
    const bool out_special = stat(opath0.c_str(), &ost) == 0 && !S_ISREG(ost.st_mode);      // a FIFO, /dev/fd/N of a process substitution, /dev/null
    const bool sam_plain = o.fmt >= 5 && !out_special && !(opath0.size() > 5 && !strcasecmp(opath0.c_str() + opath0.size() - 4, ".bam")) &&
                           !(opath0.size() > 3 && !strcasecmp(opath0.c_str() + opath0.size() - 3, ".gz"));
    const bool pre_early = sam_plain && all_plain && plain_bytes >= bk::env::sam_early_min(256ULL << 20);      // (tests lower the input size from which the file is started early)
    const uint64_t pre_early_est = (1u << 20) + plain_bytes + plain_bytes / (o.pe_mode ? 2 : 3);
    if (pre_early) pre.start(opath0.c_str(), pre_early_est, 2);
    const bool long_run = est_reads / ndev >= kLongRunMinReads;
    // bk_image_policy, from the reads a device is going to align; --window-array on | off and --index-image lean | full override its two halves
    uint32_t ctx_flags = bk_image_policy(est_reads / ndev);
    const std::string wa = a.has("window-array") ? a.str("window-array") : std::string();
    const bool wa_off = wa == "off" || wa == "0" || wa == "no";
    if (wa_off) ctx_flags &= ~BK_CTX_WINDOW_ARRAY_EAGER;
    else if (!wa.empty()) ctx_flags |= BK_CTX_WINDOW_ARRAY_EAGER;
    const bool want_array = (ctx_flags & BK_CTX_WINDOW_ARRAY_EAGER) != 0;
    const std::string img = a.has("index-image") ? a.str("index-image") : std::string();
    if (img == "full") ctx_flags &= ~(BK_CTX_GROW_IMAGE | BK_CTX_LEAN_IMAGE);
    else if (img == "lean") ctx_flags = (ctx_flags & ~BK_CTX_GROW_IMAGE) | BK_CTX_LEAN_IMAGE;
    loaders.emplace_back([&]() {
        ctx_rc[0] = bk_ctx_create_ex(&ctxs[0], a.str("I").c_str(), o.devices[0], &o.P, ctx_flags);
        if (ctx_rc[0] || ndev == 1) return;
        std::vector<std::thread> cloners;
        for (size_t d = 1; d < ndev; d++) cloners.emplace_back([&, d]() { ctx_rc[d] = bk_ctx_clone(&ctxs[d], ctxs[0], o.devices[d]); });
        for (auto &t : cloners) t.join();
    });
    auto destroy_ctxs = [&]() { for (bk_ctx *c : ctxs) bk_ctx_destroy(c); };
    ReadStore rs;
    int rc;
    // (the index loader's threads - the HIP runtime coming up, four feeding the upload - the page-locking of the packed reads' buffers and
    // the output file's pages run meanwhile: the parser leaves them their share of the cores, or a CPU quota stalls all of them in turn)
    const int parse_threads = std::max(std::min(o.nthreads, 4), o.nthreads - 6);
    if (o.pe_mode) rc = load_reads_pe(a.v["i"], a.v["u"], o.trim5, o.trim3, o.min_len, o.max_len, parse_threads, rs);
    else rc = load_reads(a.v["i"], o.trim5, o.trim3, o.min_len, o.max_len, parse_threads, rs);
    // .. and, still behind the index load: the reads packed for the boundary, the result array page-locked
    AlignedSet A;
    // (declared after the read store and A: runs before either goes - the releaser thread may still be giving the read store's bases back)
    struct Unlock { Submission &S; ~Unlock() { if (S.releaser.joinable()) S.releaser.join(); S.release_results(); } } unlock_results{S};
    if (!rc && rs.size()) rc = prepare_submission(o, rs, ndev, long_run, A, S);
    if (!rc && pre_early) pre.add_threads(2);       // (two while the parser's threads allocate their own memory, four from here on)
    { HostClock jc; for (auto &t : loaders) t.join(); jc.lap("waited for the index image"); }
    for (size_t d = 0; d < ndev; d++)
        if (ctx_rc[d]) { diag("Fatal: unable to load genome assembly suffix array: %s", bk_strerror(ctx_rc[d])); destroy_ctxs(); return 1; }
    if (rc) { destroy_ctxs(); return 1; }
    bk_ctx *ctx = ctxs[0];
    std::string species = bk_dataset_name(ctx);
    uint32_t n_ent = bk_num_entries(ctx);
    std::vector<bk_entry_info> ents(n_ent);
    for (uint32_t i = 0; i < n_ent; i++) bk_get_entry(ctx, i, &ents[i]);
    diag("Genome Assembly Name: '%s'", species.c_str());
    diag("Genome assembly suffix array loaded");

    size_t nr = rs.size();
    // (the other devices' contexts, copies of the first's image, make their window arrays when their first batch arrives)
    // (--window-array on asks for it whatever the index: one of 5-byte elements gets it only so - "use_swin" 2 -, the policy's own
    // choice leaves it without: making it goes over a 17 Gbp index twice)
    for (bk_ctx *c : ctxs) (void)bk_ctx_tune(c, "use_swin", want_array ? ((!wa.empty() && !wa_off) ? 2 : 1) : 0);
    diag("Index image: %s, suffix-ordered window array %s (%zu reads per device, %llu estimated from the input files' sizes; every table comes with the index from %llu reads per device on; --index-image lean | full and --window-array on | off override)",
         (ctx_flags & BK_CTX_LEAN_IMAGE) ? "lean" : ((ctx_flags & BK_CTX_GROW_IMAGE) ? "lean, grows on a long run" : "every table"),
         !want_array ? "off" : ((bk_sfx_el_size(ctx) == 5 && (wa.empty() || wa_off)) ? "off (an index of 5-byte elements gets it with --window-array on only)" : "on"), nr / ctxs.size(),
         (unsigned long long)(est_reads / ndev), (unsigned long long)BK_POLICY_MIN_READS);
    // (the SAM file of any other large run is started now: name + bases (+ qualities) + about 31 bytes per record, 55 for a paired end)
    if (sam_plain && pre.fd < 0) {
        if (nr >= (size_t)bk::env::sam_device_min(200000ULL)) {               // (tests lower the size from which the large-run machinery is used)
            uint64_t est = (1u << 20) + rs.name_bytes() + 40ULL * nr + (o.pe_mode ? 24ULL * nr : 0) + (uint64_t)n_ent * 128;
            est += (uint64_t)(a.num("g", 3) != 3 ? 2 : 1) * rs.base_bytes();
            pre.start(opath0.c_str(), est);
        }
    }
    // CAligner::AcceptThisChromID (Aligner.cpp:2651-2715): a sequence passes unless an exclude expression matches its name, and - with include
    // expressions present - only if one of those does too (exclusion first: not the rule of FiltByChroms further down)
    std::vector<uint8_t> chrom_ok;
    if (!o.re_excl.empty() || !o.re_incl.empty()) {
        chrom_ok.assign(n_ent + 1, 1);
        for (uint32_t c = 1; c <= n_ent; c++) {
            regmatch_t mc;
            bool ok = true;
            for (regex_t &re : o.re_excl) if (!regexec(&re, ents[c - 1].name, 1, &mc, 0)) { ok = false; break; }
            if (ok && !o.re_incl.empty()) {
                ok = false;
                for (regex_t &re : o.re_incl) if (!regexec(&re, ents[c - 1].name, 1, &mc, 0)) { ok = true; break; }
            }
            chrom_ok[c] = ok ? 1 : 0;
        }
    }
    // the paired-end rules consult the filters while they pair (AcceptProvPE, the orphan recovery's anchors, the single-end acceptance)
    if (o.pe_mode && !chrom_ok.empty())
        for (bk_ctx *c : ctxs)
            if ((rc = bk_ctx_set_chrom_filter(c, chrom_ok.data(), (uint32_t)chrom_ok.size())) != BK_OK) { diag("Error: chromosome filter table: %s", bk_strerror(rc)); destroy_ctxs(); return 1; }
    // the pipelines' device buffers and the contexts' batch scratch are in place before the clock of T_align starts (the reference sizes its
    // per-thread scratch before its workers start, Aligner.cpp:8771-8790)
    std::vector<bk_stream *> streams;
    if (nr && open_pipelines(ctxs, o, S, streams)) { destroy_ctxs(); return 1; }
    diag("Now aligning with minimum core size of %dbp...\n", bk_min_core_len(ctx));
    if (o.pe_mode) diag("Paired end association and partner alignment processing runs with the alignment of each batch");
    rc = nr ? align_reads(ctxs, streams, o, rs, S, A) : BK_OK;
    if (rc) { destroy_ctxs(); return 1; }
    if (!nr) A.seq_counts.assign(n_ent, 0);
    std::vector<bk_hit> &hits = A.hits;
    std::vector<bk_seg2> &seg2 = A.seg2;
    diag("Alignment of %zu from %zu loaded completed", nr, nr);
    // plain SAM records of a large run are formatted by the device (report.cpp): what it needs of the reads travels there while the
    // host resolves, filters and sorts
    bk_sam_prep *sam_prep = nullptr;
    bk_sam_job pk_job{};
    struct PrepGuard { bk_sam_prep *&p; Submission &S; ~PrepGuard() { if (p) { S.done_with_head_start(); bk_sam_prep_free(p); p = nullptr; } } } prep_guard{sam_prep, S};      // (not consumed: given back)
    if (pre.fd >= 0 && o.ml_mode != 5 && !o.micro_indel && !o.splice_len && !o.min_chim && !o.min_flank && nr == rs.lens.size()) {
        bk_sam_job hj{};
        hj.bases = rs.bases.data(); hj.n_bases = rs.bases.size(); hj.offs = rs.offs.data(); hj.lens = rs.lens.data();
        hj.names = rs.names.data(); hj.n_name_bytes = rs.names.size(); hj.name_ofs = rs.name_ofs.data(); hj.n_reads = nr;
        // without scores in the read store (FASTA, or -g3) the formatter reads the packed form the alignment was fed from: page-locked,
        // a quarter of the bytes, no host thread copies anything
        if (g_qual_mode == 3 && S.words != nullptr && S.lens16 != nullptr) {
            hj.pk_words = S.words; hj.n_pk_words = S.n_words; hj.pk_lens16 = S.lens16; hj.pk_exc = S.exc_job.data(); hj.n_pk_exc = S.exc_job.size();
            hj.bases = nullptr; hj.n_bases = 0; hj.offs = nullptr; hj.lens = nullptr;
        }
        pk_job = hj;
        const uint64_t per_rec = (rs.name_bytes() + (uint64_t)(a.num("g", 3) != 3 ? 2 : 1) * rs.base_bytes()) / std::max<size_t>(nr, 1) + 64 + (o.pe_mode ? 24 : 0);
        if (bk_sam_prepare(ctx, &hj, (uint32_t)std::min<uint64_t>(per_rec + per_rec / 8, 1u << 20), &sam_prep) != BK_OK) { sam_prep = nullptr; pk_job = bk_sam_job{}; }
    }
    // The read store's bases - 5 GB of a 50 M-read run, a third of a second to hand back at the exit - have served too once the device
    // holds the packed reads, unless an output of this run still reads them (-j / -J, -O, SNP calling); should the device decline
    // after all, the host formatter loads them again (restore_reads)
    bool bases_dropped = false;
    // (.. and only when every read file can be read a second time - a FIFO or a process substitution cannot: should the device decline
    // the SAM records, the host formatter reloads the reads)
    bool inputs_regular = true;
    for (const char *opt : {"i", "u"})
        if (a.has(opt))
            for (const std::string &fn : a.v[opt]) { struct stat ist; inputs_regular = inputs_regular && stat(fn.c_str(), &ist) == 0 && S_ISREG(ist.st_mode); }
    const bool may_drop_bases = inputs_regular && pk_job.pk_words != nullptr && sam_prep != nullptr && !a.has("j") && !a.has("J") && !a.has("O") && o.snp.min_reads <= 0;
    if (nr)
        S.release_packed_in_background(pk_job.pk_words != nullptr ? sam_prep : nullptr,
                                       may_drop_bases ? std::function<void()>([&rs, &bases_dropped]() { bk::RawVec<uint8_t> none; rs.bases.swap(none); bases_dropped = true; })
                                                      : std::function<void()>());
    auto restore_reads = [&]() -> int {
        if (S.releaser.joinable()) S.releaser.join();
        if (!bases_dropped) return 0;
        diag("The device declined the SAM records: loading the reads again for the host's formatter");
        ReadStore again;
        const int rl = o.pe_mode ? load_reads_pe(a.v["i"], a.v["u"], o.trim5, o.trim3, o.min_len, o.max_len, o.nthreads, again)
                                 : load_reads(a.v["i"], o.trim5, o.trim3, o.min_len, o.max_len, o.nthreads, again);
        if (rl || again.size() != rs.size()) return 1;
        rs.bases.swap(again.bases); rs.offs.swap(again.offs); rs.used_bases = again.used_bases;
        bases_dropped = false;
        return 0;
    };

    std::vector<uint32_t> src;                     // -r5: record -> read it came from (records replace the reads)
    std::vector<int> multi_dist((size_t)o.max_ml, 0);
    std::vector<bk_loci_trims> rec_trims;          // -c with -r: per record (after -r5's expansion) the trims of the placement it took
    if (o.ml_mode) resolve_multi_loci(o, rs, A, src, multi_dist, nr, rec_trims, chrom_ok.empty() ? nullptr : &chrom_ok);
    auto RD = [&](size_t i) -> size_t { return src.empty() ? i : (size_t)src[i]; };
    auto has_seg2 = [&](size_t i) -> bool { return !seg2.empty() && (seg2[RD(i)].flags & 5); };       // FlgInDel or FlgSplice
    if (o.pe_mode) {
        // CAligner::ProcessPairedEnds ran on the device with each batch (reads are held interleaved PE1,PE2, Aligner.cpp:11349-11355)
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
    auto is_chimeric = [&](size_t i) -> bool {                                                          // FlgChimeric: trims come with the hit
        if (!rec_trims.empty()) return rec_trims[i].chimeric != 0;
        return !seg2.empty() && (seg2[RD(i)].flags & 8);
    };
    if (o.min_chim) {
        // chimeric placements keep the trims AdaptiveTrim found (ProcCoredApprox :9292-9299); the flank trimmer leaves them alone (:1641)
        if (trims.empty()) {
            trims.left.assign(nr, 0); trims.right.assign(nr, 0); trims.mismatches.resize(nr);
            for (size_t i = 0; i < nr; i++) trims.mismatches[i] = hits[i].mismatches;
        }
        size_t n_ch = 0;
        for (size_t i = 0; i < nr; i++)
            if (hits[i].nar == BK_NAR_ACCEPTED && is_chimeric(i)) {
                trims.left[i] = rec_trims.empty() ? seg2[RD(i)].match_len : rec_trims[i].left;
                trims.right[i] = rec_trims.empty() ? seg2[RD(i)].read_ofs : rec_trims[i].right;
                n_ch++;
            }
        diag("Of the accepted aligned reads, %zu were chimeric", n_ch);
    }
    // CAligner::SortHitMatch (Aligner.cpp:10069-10114), ties left to the replica of the reference's sort.  The comparator's fields are
    // packed into three words per record first (NAR | NumHits class: 1 before the others | NumHits | ChromID, then AdjStartLoci |
    // AdjHitLen, then Strand | LowMMCnt; records whose NumHits is not 1 compare equal beyond NumHits, so their other fields stay 0): the
    // quicksort then walks keys lying next to each other instead of chasing record indexes through the result array - same
    // comparisons, same outcome, same order
    struct SortRec { uint64_t hi, lo; uint32_t tail, idx; };
    auto sort_cmp = [](const SortRec &x, const SortRec &y) -> int {
        if (x.hi != y.hi) return x.hi < y.hi ? -1 : 1;
        if (x.lo != y.lo) return x.lo < y.lo ? -1 : 1;
        if (x.tail != y.tail) return x.tail < y.tail ? -1 : 1;
        return 0;
    };
    auto sorted_order = [&](std::vector<uint32_t> &ord) {
        bk::RawVec<SortRec> recs(nr);
        auto fill = [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                const bk_hit &p = hits[i];
                SortRec &r = recs[i];
                r.idx = (uint32_t)i;
                if (p.num_hits == 1) {
                    r.hi = ((uint64_t)p.nar << 41) | (uint64_t)p.chrom_id;
                    r.lo = ((uint64_t)a_start(p, i) << 32) | (uint64_t)a_len(p, i);
                    r.tail = ((uint32_t)p.strand << 8) | (uint32_t)(uint8_t)(p.low_mm + 128);
                } else {
                    r.hi = ((uint64_t)p.nar << 41) | (1ULL << 40) | ((uint64_t)p.num_hits << 32);
                    r.lo = 0;
                    r.tail = 0;
                }
            }
        };
        HostClock clk;
        par_ranges(nr, o.nthreads, [&](size_t lo, size_t hi, int) { fill(lo, hi); });
        clk.lap("sort records filled");
        bk::ref_order_sort(recs.data(), (int64_t)nr, sort_cmp, o.nthreads);
        clk.lap("sorted (the reference's order of equal records)");
        ord.resize(nr);
        par_ranges(nr, o.nthreads, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; i++) ord[i] = recs[i].idx; });
        clk.lap("order taken");
    };
    if (o.pcr_win >= 0 && !o.pe_mode) {
        // CAligner::ReducePCRduplicates runs on the sorted set, before the flank trimmer (Aligner.cpp:598-610)
        diag("Processing to reduce PCR differential amplification artefacts processing started..");
        std::vector<uint32_t> ord;
        sorted_order(ord);
        const size_t n_dup = bk::reduce_pcr_duplicates(hits, ord, [&](size_t i) { return a_start(hits[i], i); }, [&](size_t i) { return a_len(hits[i], i); }, o.pcr_win);
        diag("Removed %zu potential PCR artefact reads", n_dup);
    }
    if (o.min_flank > 0) {
        diag("Starting 5' and 3' flank sequence autotrim processing...");
        bk::SfxFile sft;
        std::string serr;
        if (bk::sfx_open(a.str("I").c_str(), sft, &serr) != 0) { diag("Fatal: %s", serr.c_str()); destroy_ctxs(); return 1; }
        bk::auto_trim_flanks(hits, [&](size_t i) { return has_seg2(i) || is_chimeric(i); }, [&](size_t i) { return rs.bases.data() + rs.offs[RD(i)]; },
                             [&](size_t i) -> const uint8_t * {
                                 const bk_hit &h = hits[i];
                                 return (h.chrom_id >= 1 && h.chrom_id <= n_ent) ? sft.seq + ents[h.chrom_id - 1].start_ofs + h.match_loci : nullptr;
                             },
                             o.min_flank, o.pe_mode != 0, o.nthreads, trims);
        diag("Finished 5' and 3' flank sequence autotriming, %zu plus strand and %zu minus strand aligned reads removed", trims.removed_plus,
             trims.removed_minus);
    }
    // orphan junction filters, splice junctions first (Aligner.cpp:630-650)
    if (o.splice_len) {
        diag("Removal of orphan splice junction processing started..");
        auto r = bk::remove_orphan_segs(hits, seg2, 4, 7);
        diag("From %zu reads with putative splice junctions %zu orphans were removed", r.first, r.second);
    }
    if (o.micro_indel) {
        diag("Removal of orphan microInDels processing started..");
        auto r = bk::remove_orphan_segs(hits, seg2, 1, 8);
        diag("From %zu reads with putative microIndels %zu orphans were removed", r.first, r.second);
    }

    if (!o.re_excl.empty() || !o.re_incl.empty()) {
        // CAligner::FiltByChroms (Aligner.cpp:4019-4120): a sequence stays if an include expression matches its name, or - with no
        // include expressions at all - if no exclude expression does; accepted reads on the others become eNARChromFilt
        diag("Now filtering matches by chromosome");
        std::vector<uint8_t> keep(n_ent + 1, 1);
        for (uint32_t c = 1; c <= n_ent; c++) {
            regmatch_t mc;
            bool ok = false;
            for (regex_t &re : o.re_incl) if (!regexec(&re, ents[c - 1].name, 1, &mc, 0)) { ok = true; break; }
            if (!ok && o.re_incl.empty()) {
                ok = true;
                for (regex_t &re : o.re_excl) if (!regexec(&re, ents[c - 1].name, 1, &mc, 0)) { ok = false; break; }
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

    // CAligner::ReportAlignStats (Aligner.cpp:3493-3822): strand counts, the simulated-reads truth check, NAR histogram
    uint64_t nar[20] = {0};
    uint64_t tot_acc = 0, tot_plus = 0;
    size_t first_acc = nr;                         // the first accepted read, in load order
    {
        const int nt = std::max(1, o.nthreads);
        std::vector<std::array<uint64_t, 24>> part((size_t)nt);
        std::vector<size_t> first((size_t)nt, nr);
        for (auto &pt : part) pt.fill(0);
        par_ranges(nr, nt, [&](size_t lo, size_t hi, int t) {
            std::array<uint64_t, 24> c{};
            size_t f = nr;
            for (size_t i = lo; i < hi; i++) {
                const bk_hit &h = hits[i];
                c[h.nar < 20 ? h.nar : 0]++;
                if (h.nar == BK_NAR_ACCEPTED) { c[20]++; c[21] += h.strand == '+'; if (f == nr) f = i; }
            }
            part[(size_t)t] = c; first[(size_t)t] = f;
        });
        for (int t = 0; t < nt; t++) {
            for (int k = 0; k < 20; k++) nar[k] += part[(size_t)t][(size_t)k];
            tot_acc += part[(size_t)t][20]; tot_plus += part[(size_t)t][21];
            first_acc = std::min(first_acc, first[(size_t)t]);
        }
    }
    {
        // Reads named by `biokanga simreads` carry where they came from (lcl|usimreads|id|chrom|start|end|len|strand|..): an accepted
        // alignment on the named sequence counts as high confidence when one or both of its ends are the named ones, anything
        // else as misaligned.  Reads are visited in load order; the first accepted read decides whether the set is simulated, and
        // a later accepted read that does not parse ends the check - and drops the line - exactly as the reference's loop does
        // (:3560-3650).  The descriptors are parsed with the reference's own sscanf formats.
        uint64_t n_plus = 0, n_acc = 0, n2 = 0, n1 = 0, n_mis = 0;
        bool sim = false;
        n_acc = tot_acc; n_plus = tot_plus;        // (counted above by all threads; the truth check below visits reads only while the set looks simulated)
        for (size_t i = first_acc; i < nr; i++) {
            const bk_hit &h = hits[i];
            if (h.nar != BK_NAR_ACCEPTED) continue;
            if (!(sim || i == first_acc)) break;
            char typ[100], xchrom[300], x1c[100], x2c[100], xstrand;
            int id, xs, xe, xl, xerrs;
            const char *nm = rs.name(RD(i));
            if (strlen(nm) > 127) continue;
            int its = sscanf(nm, "%99[^|]|usimreads|%d|%99[^|]|%d|%d|%d|%c|%d", typ, &id, xchrom, &xs, &xe, &xl, &xstrand, &xerrs);
            if (its >= 6) sim = true;
            else {
                its = sscanf(nm, "%99[^|]|usimreads|%d|%99[^|]|%99[^|]|%99[^|]|%d|%d|%d|%c|%d", typ, &id, xchrom, x1c, x2c, &xs, &xe, &xl, &xstrand, &xerrs);
                if (its < 8) sim = false;
                else { strcat(xchrom, "|"); strcat(xchrom, x1c); strcat(xchrom, "|"); strcat(xchrom, x2c); sim = true; }
            }
            if (!sim) continue;
            if (h.chrom_id < 1 || h.chrom_id > n_ent || strcasecmp(ents[h.chrom_id - 1].name, xchrom)) { n_mis++; continue; }
            const long left = (long)h.match_loci;
            const long right = has_seg2(i) ? (long)seg2[RD(i)].match_loci + seg2[RD(i)].match_len - 1 : left + h.match_len - 1;
            if (left == xs || right == xe) (left != xs || right != xe ? n1 : n2)++;
            else n_mis++;
        }
        diag("From %zu source reads there are %llu accepted alignments, %llu on '+' strand, %llu on '-' strand", rs.size(), (unsigned long long)n_acc,
             (unsigned long long)n_plus, (unsigned long long)(n_acc - n_plus));
        if (sim)
            diag("There are %llu (%llu 2 edge, %llu 1 edge) high confidence aligned simulated reads with %llu misaligned", (unsigned long long)(n2 + n1),
                 (unsigned long long)n2, (unsigned long long)n1, (unsigned long long)n_mis);
    }
    diag("Unable to align %llu source reads of which %llu were not aligned as they contained excessive number of indeterminate 'N' bases",
         (unsigned long long)(nr - nar[1]), (unsigned long long)nar[2]);
    diag("Read nonalignment reason summary:");
    for (int k = 0; k < 20; k++) diag("   %llu (%s) %s", (unsigned long long)nar[k], kNarTag[k], kNarDescr[k]);
    if (!o.pe_mode && !o.ml_mode && !o.micro_indel && !o.splice_len && !o.min_chim && o.pcr_win < 0 && !o.min_flank && o.re_excl.empty() && o.re_incl.empty()) {
        // nothing above changed an acceptance: the per-sequence counts the devices kept (summed over them by the exchange step)
        // must add up to the accepted reads counted here
        uint64_t tot = 0;
        for (uint64_t c : A.seq_counts) tot += c;
        if (tot != nar[1]) { diag("Fatal: the devices counted %llu accepted reads, the result records hold %llu", (unsigned long long)tot, (unsigned long long)nar[1]); destroy_ctxs(); return 1; }
        if (ndev > 1) diag("Accepted read counts per sequence reduced over %zu devices: %llu reads", ndev, (unsigned long long)tot);
    }

    // SortReadHits(eRSMHitMatch): index in load (ReadID) order -> reference order
    diag("Reporting of aligned result set started...");
    diag("Sorting alignments by ascending chrom.loci");
    std::vector<uint32_t> order;
    sorted_order(order);

    Report R{a, rs, hits, ents, species, n_ent, src, seg2, trims, multi_dist, order, o.pe_mode, o.ml_mode, o.max_ml, o.fmt, o.nthreads, o.micro_indel, o.splice_len, o.max_rpt_sam_seqs};
    R.ctx = ctx;
    R.pre = pre.fd >= 0 ? &pre : nullptr;
    S.done_with_head_start();
    R.restore_reads = restore_reads;
    R.sam_prep = sam_prep;
    R.pk_words = pk_job.pk_words; R.n_pk_words = pk_job.n_pk_words; R.pk_lens16 = pk_job.pk_lens16; R.pk_exc = pk_job.pk_exc; R.n_pk_exc = pk_job.n_pk_exc;
    sam_prep = nullptr;                            // (the report owns it from here)
    // ".bam" (more than 5 characters of name, kanga.cpp:848-857): BGZF-compressed BAM with its BAI index; else SAM / CSV / BED text
    const std::string opath = a.str("o");
    int rr = (o.fmt >= 5 && opath.size() > 5 && !strcasecmp(opath.c_str() + opath.size() - 4, ".bam")) ? report_bam(R, opath) : report_text(R);
    if (R.sam_prep) { bk_sam_prep_free(R.sam_prep); R.sam_prep = nullptr; }        // (the report took another path)
    // SNPs: the file is only opened for '-M0' .. '-M5' (Aligner.cpp:4488), and only processed when reads were accepted (:746)
    if (rr == 0 && o.snp.min_reads > 0 && o.fmt <= 5) {
        o.snp.path = a.has("S") ? a.str("S") : opath + ".snp";
        o.snp.bed = o.fmt == 4;
        o.snp.vcf = !o.snp.bed && o.snp.path.size() >= 4 && !strcasecmp(o.snp.path.c_str() + o.snp.path.size() - 4, ".vcf");     // Aligner.cpp:157-165
        o.snp.title = a.str("t", "kanga");
        o.snp.sfx_path = a.str("I");
        bool any = false;
        for (const bk_hit &h : hits) if (h.nar == BK_NAR_ACCEPTED) { any = true; break; }
        if (any) rr = process_snps(ctx, R, o.snp);
        else for (const char *ext : {"", ".disnp.csv", ".trisnp.csv", ".markers"}) { if (ext[1] == 'm' && !o.snp.marker_len) continue; OutBuf e; e.open((o.snp.path + ext).c_str()); e.close(); }
    }
    if (rr == 0 && nr >= 1000000) {                // (small runs unwind normally: leak checkers and tests see every destructor)
        pre.finish();
        end_process(rr);                           // (contexts, page-locked buffers and the address space go with the process)
    }
    { HostClock clk; destroy_ctxs(); clk.lap("contexts destroyed"); }
    return rr;
}


// CUtility::arg_parsefromfile (libbiokanga/Utility.cpp:793-910; called first thing by every sub-process, kanga.cpp:298): an argument
// "@file" is replaced by the options found in that file - one or more per line, separated by blanks or tabs except inside quotes
// (the quote characters stay part of the option, as they do there); empty lines and lines starting with '#', ';' or "//" are skipped
bool expand_param_files(int argc, char **argv, std::vector<std::string> &out)
{
    for (int i = 0; i < argc; i++) {
        if (argv[i][0] != '@') { out.push_back(argv[i]); continue; }
        std::string fn = argv[i] + 1;
        while (!fn.empty() && isspace((unsigned char)fn.back())) fn.pop_back();
        size_t b0 = 0;
        while (b0 < fn.size() && isspace((unsigned char)fn[b0])) b0++;
        fn = fn.substr(b0);
        FILE *f = fn.empty() ? nullptr : fopen(fn.c_str(), "r");
        if (!f) { printf("Unable to open options file '%s'\nError: %s", fn.c_str(), strerror(errno)); return false; }
        char line[8192];
        while (fgets(line, sizeof(line), f)) {
            std::string l = line;
            while (!l.empty() && isspace((unsigned char)l.back())) l.pop_back();
            size_t a = 0;
            while (a < l.size() && isspace((unsigned char)l[a])) a++;
            l = l.substr(a);
            if (l.empty() || l[0] == '#' || l[0] == ';' || (l[0] == '/' && l.size() > 1 && l[1] == '/')) continue;
            std::string opt;
            bool in_quotes = false, in_param = false;
            for (size_t k = 0; k <= l.size(); k++) {
                char ch = k < l.size() ? l[k] : '\0';
                if (ch == 0x16) ch = '-';
                if (ch == '"' || ch == '\'') { in_quotes = !in_quotes; in_param = true; opt.push_back(ch); continue; }
                if ((ch == ' ' || ch == '\t') && in_quotes) { opt.push_back(ch); continue; }
                if (ch == ' ' || ch == '\t' || ch == '\0') {
                    if (!in_param && ch != '\0') continue;
                    out.push_back(opt);
                    opt.clear();
                    in_quotes = in_param = false;
                    continue;
                }
                in_param = true;
                opt.push_back(ch);
            }
        }
        fclose(f);
    }
    return true;
}

}  // namespace

int main(int argc_in, char **argv_in)
{
    std::vector<std::string> expanded;
    if (!expand_param_files(argc_in, argv_in, expanded)) return 1;
    std::vector<char *> argv_vec;
    for (std::string &a : expanded) argv_vec.push_back(&a[0]);
    argv_vec.push_back(nullptr);
    const int argc = (int)expanded.size();
    char **argv = argv_vec.data();
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
    g_t0 = time(nullptr);
    int rc;
    std::string sub = argv[1];
    if (sub == "index" || sub == "kangax") rc = cmd_index(argc, argv, 2);
    else if (sub == "align" || sub == "kanga") rc = cmd_align(argc, argv, 2);
    else {
        fprintf(stderr, "%s: sub-process '%s' is outside the supported hot path (index, align)\n", g_proc.c_str(), sub.c_str());
        return 1;
    }
    diag("Exit code: %d Total processing time: %ld seconds", rc, (long)(time(nullptr) - g_t0));
    return rc;
}
