/*
 * biokanga_amd.h - C ABI of the MI355X-native `biokanga align` hot path (libbiokanga_amd.so).
 *
 * Drop-in boundary.  The reference (csiro-crop-informatics/biokanga v4.4.2) has no FFI of its own;
 * the seam is the pair of call sites SURVEY.md §8(b) names:
 *   - per read : CSfxArrayV3::AlignReads            libbiokanga/SfxArrayV2.h:585-606, called from
 *                CAligner::ProcCoredApprox          biokanga/Aligner.cpp:9220-9237
 *   - per batch: CAligner::LocateCoredApprox        biokanga/Aligner.h:930-931 (Aligner.cpp:8651)
 * bk_align_batch() is the batch form of the first placed where the second is: it consumes the
 * reads exactly as CAligner holds them (1 byte/base: bits 0-2 base, bit 3 soft-mask, bits 4-7
 * quality, biokanga/Aligner.cpp:9038-9055) and fills, per read, the fields ProcCoredApprox writes
 * into tsReadHit (NAR, NumHits, LowHitInstances, LowMMCnt, NxtLowMMCnt, HitLoci.Hit.Seg[0]).
 *
 * Plain pointers and sizes only; no C++/torch types.  Every function returns 0 (eBSFSuccess) or a
 * negative teBSFrsltCodes value (libbiokanga/ErrorCodes.h:15-95); HIP/RCCL failures map to
 * BK_ERR_INTERNAL.  The library never calls exit().  There is NO CPU fallback: without a usable
 * HIP device every compute entry point fails with BK_ERR_NODEVICE.
 */
#ifndef BIOKANGA_AMD_H
#define BIOKANGA_AMD_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* teBSFrsltCodes subset (libbiokanga/ErrorCodes.h) */
#define BK_OK               0
#define BK_ERR_INTERNAL    (-1)    /* eBSFerrInternal  */
#define BK_ERR_PARAMS      (-100)  /* eBSFerrParams    */
#define BK_ERR_MEM         (-95)   /* eBSFerrMem       */
#define BK_ERR_NOTBIOSEQ   (-94)   /* eBSFerrNotBioseq */
#define BK_ERR_OPNFILE     (-90)   /* eBSFerrOpnFile   */
#define BK_ERR_CREATEFILE  (-89)   /* eBSFerrCreateFile*/
#define BK_ERR_FILEVER     (-86)   /* eBSFerrFileVer   */
#define BK_ERR_FILEACCESS  (-85)   /* eBSFerrFileAccess*/
#define BK_ERR_NODEVICE    (-2)    /* ours: no HIP device / kernels unavailable */

/* tHRslt, libbiokanga/SfxArrayV2.h:68-74 */
enum { BK_HR_NONE = 0, BK_HR_HITS = 1, BK_HR_MMDELTA = 2, BK_HR_HITINSTS = 3, BK_HR_RMMDELTA = 4 };
/* teNAR, biokanga/Aligner.h:106-128 (values produced by the hot path) */
enum { BK_NAR_UNALIGNED = 0, BK_NAR_ACCEPTED = 1, BK_NAR_NS = 2, BK_NAR_NOHIT = 3, BK_NAR_MMDELTA = 4,
       BK_NAR_MULTIALIGN = 5 };

/* alignment parameters = the `biokanga align` options that reach the hot path
 * (biokanga/kanga.cpp:194-294) */
typedef struct bk_align_params {
    int32_t max_subs;       /* -s  max substitutions per 100 bp (reference default 10)         */
    int32_t min_edit_dist;  /* -e  minimum Hamming delta to the next best hit, 1 or 2           */
    int32_t align_strand;   /* -Q  0 both strands, 1 sense only, 2 antisense only               */
    int32_t pmode;          /* -m  0 default, 1 more sensitive, 2 ultra sensitive, 3 less sens. */
    int32_t max_ns;         /* -n  max indeterminate bases per 100 bp (default 1)               */
    int32_t max_ml;         /* -R  MaxMLmatches (MaxHits of AlignReads): 1 in the default -r0 mode, 2..BK_MAX_ML in the
                             *     multi-loci modes -r1..-r5, whose loci lists bk_batch_loci() then returns         */
    int32_t clamp_ml;       /* -X  with max_ml > 1: reads with more than max_ml loci (rslt eHRHitInsts) also get a loci
                             *     list, of their first max_ml loci (bClampMaxMLmatches, Aligner.cpp:9243-9248)        */
    int32_t best_matches;   /* -N  with max_ml > 1: CSfxArrayV3::LocateBestMatches instead of AlignReads - per read the (up to)
                             *     max_ml loci with the fewest mismatches, none above the -s limit, ordered by mismatches
                             *     then discovery; bk_hit.rslt is eHRhits / eHRnone, LowMMCnt and NxtLowMMCnt stay 0          */
    int32_t micro_indel_len;/* -a  0, or the longest microInDel (1..20) looked for in reads the AlignReads phases leave unaligned
                             *     (CSfxArrayV3::LocateInDels); second segments come back through bk_batch_seg2()             */
    int32_t splice_junct_len;/* -A  0, or the longest splice junction (25..100000) looked for in reads still unaligned after that
                             *     (CSfxArrayV3::LocateSpliceJuncts); second segments through bk_batch_seg2(), flags bit 2      */
    int32_t min_chimeric_len;/* -c  0, or 50..99: reads nothing else aligned may be placed end-trimmed, keeping at least this percentage of
                             *     their length (chimeric form of LocateCoreMultiples); trims through bk_batch_seg2(), flags bit 3 */
    int32_t reserved2;
} bk_align_params;

/* per-read result: the tsReadHit fields written by ProcCoredApprox (Aligner.cpp:9311-9479) and
 * tsHitLoci.Seg[0] (SfxArrayV2.h:219-240).  20 bytes, little endian. */
typedef struct bk_hit {
    uint32_t chrom_id;           /* Seg[0].ChromID = tsSfxEntry.EntryID (1..n), 0 if none        */
    uint32_t match_loci;         /* Seg[0].MatchLoci, 0-based within the entry                   */
    uint16_t match_len;          /* Seg[0].MatchLen                                              */
    int16_t  low_hit_instances;  /* tsReadHit.LowHitInstances                                    */
    uint8_t  rslt;               /* tHRslt returned by AlignReads                                */
    uint8_t  nar;                /* teNAR                                                        */
    uint8_t  strand;             /* '+', '-' or '?'                                              */
    int8_t   low_mm;             /* tsReadHit.LowMMCnt                                           */
    int8_t   nxt_low_mm;         /* tsReadHit.NxtLowMMCnt                                        */
    uint8_t  num_hits;           /* tsReadHit.NumHits                                            */
    uint8_t  mismatches;         /* Seg[0].Mismatches (Hamming score of the accepted hit)        */
    uint8_t  flags;              /* reserved                                                     */
} bk_hit;

/* counters behind the "algorithmic bytes" figure of SURVEY.md §8(d); all are counts of what the
 * REFERENCE algorithm does for the same reads (independent of our data layout) */
typedef struct bk_counters {
    uint64_t n_reads;
    uint64_t n_search;      /* LocateFirstExact calls                                  */
    uint64_t n_cand;        /* candidates Hamming-extended (new, in-bounds targets)    */
    uint64_t n_lcm_calls;   /* LocateCoreMultiples invocations                         */
    uint64_t n_heavy;       /* (ours) LocateCoreMultiples calls routed to the wave-per-read kernels */
    uint64_t n_cand_heavy;  /* (ours) the part of n_cand processed by the wave-per-read kernels    */
    uint64_t reserved[2];
} bk_counters;

/* timing of the device work of the last bk_align_batch*() call, measured with HIP events on the
 * stream the kernels were launched on */
typedef struct bk_timing {
    float    ms_total;          /* first kernel start -> last kernel end                       */
    float    ms_search;         /* sum over launches of the SA-interval search kernel          */
    float    ms_extend;         /* sum over launches of the candidate walk / Hamming kernel    */
    float    ms_heavy;          /* sum over launches of the wave-per-read kernel               */
    float    ms_other;          /* pack / finalize                                             */
    uint32_t n_search_launches;
    uint32_t n_extend_launches;
    uint32_t n_heavy_launches;
    uint32_t n_search_b_launches;
    /* the search stage by kernel (ms_search = their sum + the clears of the interval records): pass A (k-mer table + small buckets),
     * the grouping of pass B's work list (keys + radix sort), pass B (bisection of the big buckets) */
    float    ms_search_a;
    float    ms_search_sort;
    float    ms_search_b;
    float    ms_prep;           /* read preparation (2-bit rows of both strands, N policy, first active list); part of ms_other */
} bk_timing;

typedef struct bk_entry_info {
    uint32_t entry_id;
    uint32_t seq_len;
    uint64_t start_ofs;
    uint64_t end_ofs;
    char     name[81];
} bk_entry_info;

typedef struct bk_ctx bk_ctx;

const char *bk_version(void);
const char *bk_strerror(int rc);
/* number of usable HIP devices (0 if none); never fails */
int  bk_device_count(void);

/* Replaces CSfxArrayV3::Open + SetTargBlock (libbiokanga/SfxArrayV2.cpp:891-1103,1836-1890) and the
 * MinCoreLen/MaxIter set-up of CAligner::Align / LocateCoredApprox (Aligner.cpp:341-356,8725-8761):
 * reads the .sfx file, uploads target + suffix array to HBM, builds the k-mer interval table. */
int  bk_ctx_create(bk_ctx **out, const char *sfx_path, int device_id, const bk_align_params *p);
/* The same with flags.  BK_CTX_WINDOW_ARRAY_EAGER: the suffix-ordered window array - 48 bytes for every suffix of the part of the suffix
 * array the wave kernel's long walks visit, a sixth of a 3.1 Gbp index (25 GB) - is part of the image from the start: made slice by
 * slice behind the suffix array's upload, as the other tables are, for reads of a hundred bases (a first batch that is searched with
 * other core lengths makes it again: 0.1 s); without the flag the array is made when the first batch it can serve arrives (tuning knob
 * "use_swin").  Replaces the candidates' random target reads of LocateCoreMultiples' walk (libbiokanga/SfxArrayV2.cpp:5862-5915) by a
 * streaming read; results never depend on it. */
#define BK_CTX_WINDOW_ARRAY_EAGER 1u
/* BK_CTX_LEAN_IMAGE: the tables that only pay from tens of millions of reads on are left out - the k-mer table's second words (a
 * bucket's only key, or the map of its keys' first five bits; "use_ktab2": 17 GB more at 3.1 Gbp, 0.18 ns per read of a hundred bases
 * together with the next) and the third- and fourth-level search keys ("use_k3": 4 bytes per suffix each).  Without the flag they are
 * part of the image wherever a fifth of the HBM stays free behind them, made behind the suffix array's upload like the other tables
 * (0.05 - 0.2 s of a 3.1 Gbp index's load on an idle device: profiles/r06_*_e2e_image*.txt).  Results never depend on it. */
#define BK_CTX_LEAN_IMAGE 2u
/* BK_CTX_NO_DEEP_KEYS: the third- and fourth-level search keys alone are left out (8 bytes per suffix: 26 GB at 3.1 Gbp).  Results
 * never depend on it. */
#define BK_CTX_NO_DEEP_KEYS 4u
/* BK_CTX_GROW_IMAGE: the context starts with the lean image - what a job of a few million reads wants - and, should it turn out to
 * align 5 x BK_POLICY_MIN_READS reads after all (tuning knob "grow_after_reads"), makes the tables BK_CTX_LEAN_IMAGE leaves out in the
 * background: a thread of its own allocates and fills them on a stream of its own and the next batch after they are complete takes
 * them in.  "image_wait" makes them at once and waits; the environment's BK_GROW_AFTER_READS=<n> sets the threshold of every context
 * created with the flag and makes the batch after the one that started the worker wait for it (tests: a small run that grows).
 * Results never depend on it. */
#define BK_CTX_GROW_IMAGE 8u
/* The image a job gets, by ONE rule for the command line (`biokanga align`, which knows its read count from the input files' sizes),
 * the benchmark and any other caller: the flags for bk_ctx_create_ex from the reads one device is going to align.  From
 * BK_POLICY_MIN_READS on: every table and the window array from the start (BK_CTX_WINDOW_ARRAY_EAGER - 0.1 - 0.2 s more load for 0.7 ns
 * less per read of a hundred bases at 3.1 Gbp); below, or when the count is not known (0): BK_CTX_GROW_IMAGE. */
#define BK_POLICY_MIN_READS 20000000ULL
uint32_t bk_image_policy(uint64_t reads_per_device);
int  bk_ctx_create_ex(bk_ctx **out, const char *sfx_path, int device_id, const bk_align_params *p, uint32_t flags);

/* Same, from an index image already resident in HBM (synthetic benchmarks, GPU-built indexes):
 * d_seq  = concat_len bytes, 1 byte/base, eBaseEOS(7) after every entry (tsSfxBlock.SeqSuffix)
 * d_sa   = concat_len suffix array elements of sfx_el_size (4 or 5) bytes, little endian
 * entries (host) describe the sequences as tsSfxEntry does.  The buffers are copied/re-packed;
 * the caller keeps ownership and may free them after the call returns. */
int  bk_ctx_create_from_device(bk_ctx **out, const void *d_seq, uint64_t concat_len, const void *d_sa,
                               int sfx_el_size, const bk_entry_info *entries, uint32_t n_entries,
                               int device_id, const bk_align_params *p);

/* A second context on another (or the same) device from the finished index image of `src`: every table is copied device to device
 * (hipMemcpyPeer: over xGMI between two GPUs) instead of reading the .sfx again and rebuilding the tables there - how one process
 * replicates the index over the GPUs of a node.  Parameters are those of `src`; tuning knobs set on `src` after its
 * creation that change the image (kmer_bits, use_k2, use_isa, use_tgt2) are inherited with it. */
int  bk_ctx_clone(bk_ctx **out, const bk_ctx *src, int device_id);

void bk_ctx_destroy(bk_ctx *ctx);

/* Sizes the context's batch scratch (read rows, interval records, work lists, sort buffers) for batches of up to max_batch_reads reads
 * of up to max_read_len bases NOW, so that no allocation is left for the first batch to pay inside T_align - the reference sizes its
 * per-thread scratch before its workers start, too (CAligner::LocateCoredApprox, Aligner.cpp:8771-8790).  Optional for the blocking
 * calls and the pipeline (batches grow the scratch on demand), required by bk_align_batch_device_async. */
int  bk_ctx_reserve(bk_ctx *ctx, uint32_t max_batch_reads, uint32_t max_read_len);

/* The chromosome filters (-Z / -z) as the paired-end rules see them: accept[id] != 0 <=> CAligner::AcceptThisChromID(id) (Aligner.cpp:2651-2715),
 * consulted by AcceptProvPE (:2771-2786), by the anchors of the orphan recovery (:3224,3323) and by the single-end acceptance at the end of
 * ProcessPairedEnds (:3445-3473).  ids >= n pass; n == 0 removes the table.  Only bk_pair_batch* (and pipelines created with `pe`) look at it:
 * the filters never change a single read's alignment. */
int  bk_ctx_set_chrom_filter(bk_ctx *ctx, const uint8_t *accept, uint32_t n);

/* change alignment parameters (re-derives MinCoreLen, MaxIter, slides) */
int  bk_ctx_set_params(bk_ctx *ctx, const bk_align_params *p);
/* cross-check knobs: results never depend on them; the test-suite runs independent implementations of the same step against
 * each other through these.  name =
 *   "kmer_bits" (k of the k-mer table, 2..16)   "use_ktab" (0: plain bisection)   "use_k2" (second-level key array)
 *   "use_iv32" (phase 0 hands the interval of a read's first k + 16 bases to the later phases)   "lazy_search" (small buckets handed on unverified)
 *   "use_ktab2" (k-mer table entries of two words, 17 GB more at k = 16: 1, the default - a bucket of one suffix carries its second-level key and is
 *   settled by the line that names it; 2 - it carries the suffix array element and the search hands the suffix on for the extension to check:
 *   no trip to the suffix array for it, but every bucket whose key the search would have turned down is a candidate - slower, round 6)
 *   "grow_after_reads" / "grow_state" / "image_wait" (BK_CTX_GROW_IMAGE: the reads after which the long-run tables are made; 0 not started, 1 being
 *   made, 2 made, 3 nothing made, 4 taken in, 5 not a growing context; make them now and wait: returns key arrays + 4 if the k-mer table carries keys)
 *   "use_k3" (0..2: key arrays of the 15 bases behind the second-level keys' and of the 15 behind those - 4 bytes per suffix each, where the
 *   HBM has the room; how many there are: "k3_resident")
 *   "sort_lists" (bit 0: search work list grouped by bucket - by default only where the index has no third-level keys -, bit 1: wave list
 *   sorted, bit 2: .. longest read first)
 *   "use_wave" (wave kernel, 0: general hash-set kernel)   "use_isa" (inverse suffix
 *   array dedupe, 0: hash set)   "use_tgt2" (2 bit/base window compare)   "heavy_thresh" (longest interval the lane/flat kernels take, 0..100)
 *   "use_swin" (suffix-ordered window array: 0 none - every window from the 2-bit target; 1 for the part of the suffix array the wave kernel's long
 *   walks visit, reads of <= 100 bases and the middle cores of reads of <= 160; 2 the same whatever the batch's longest read; 3 for every suffix.
 *   An index of 5-byte elements gets the partial array only at 2: making it goes over such an index twice - once to find the shortest run
 *   length whose coverage fits the memory that is free -, 2 s at 17 Gbp for a seventh of its wave kernel's time)
 *   "swin_budget_kb" (most the partial array may take; 0: a third of what every suffix would, within half of the free HBM - nearly all of it for an
 *   index beyond 2^32 suffixes)   "swin_skip_short" (the coverage rule without the reads' this many shortest core lengths; 0)
 *   "ktab_wide" (tests: 1 = the k-mer table with 64-bit bucket starts whatever the index's size, packed as an index beyond 2^32 suffixes has
 *   it - a 64-bit start per 2^16 codes + 32-bit offsets, half the bytes -, 2 = unpacked; rebuilt)   "ktab_packed" (read only)
 *   "swin_resident", "swin_mbytes", "swin_setup_us", "swin_covered_ppm", "swin_core_lens" (read only, value ignored: whether the window array
 *   is in HBM right now, what it occupies, what making it took, the share of the suffix array it holds, the core lengths its coverage is for)
 *   "async_phases" (1: the main path's phase loop never reads a count back - launches sized by bounds, sorts by the previous chunk's
 *   needs; 0: counts read back between launches, as every other configuration does)
 *   "force_rccl" (bk_seq_counts_allreduce goes through RCCL - a communicator of one rank - even when every context sits on one device; how many
 *   of a context's reductions did, and the ranks of the last one's communicator: "rccl_allreduces", "rccl_ranks", read only)
 *   "debug_stop_phase" (test hook, see bk_debug_intervals)
 *   "chunk_reads" (reads per pass over the phases)   "max_read_len"
 * returns the old value or <0 */
int64_t bk_ctx_tune(bk_ctx *ctx, const char *name, int64_t value);

uint32_t bk_num_entries(const bk_ctx *ctx);
int  bk_get_entry(const bk_ctx *ctx, uint32_t idx, bk_entry_info *out);
const char *bk_dataset_name(const bk_ctx *ctx);
uint64_t bk_concat_len(const bk_ctx *ctx);
int  bk_sfx_el_size(const bk_ctx *ctx);
int  bk_min_core_len(const bk_ctx *ctx);

/* Batch form of CSfxArrayV3::AlignReads over host buffers (blocking).
 * bases: all reads concatenated, 1 byte/base as CAligner holds them; offs[i] = start of read i in
 * bases; lens[i] = read length (<= 2000); out[nreads]. */
int  bk_align_batch(bk_ctx *ctx, const uint8_t *bases, const uint64_t *offs, const uint32_t *lens,
                    uint32_t nreads, bk_hit *out);

/* Same over buffers already resident in HBM on the context's device.  The kernels are launched on `stream` (a
 * hipStream_t, NULL = the context's own stream), after whatever that stream already holds.  The call is
 * HOST-SYNCHRONOUS: it measures the batch's longest read, sizes (and may allocate) the batch scratch, and returns when the
 * results are in d_out; `sync` is accepted and has no effect.  The overlapped form (uploads, kernels and downloads of
 * consecutive batches running concurrently) is bk_stream_*; the form that only enqueues is below. */
int  bk_align_batch_device(bk_ctx *ctx, const void *d_bases, const void *d_offs, const void *d_lens,
                           uint32_t nreads, void *d_out, void *stream, int sync);

/* The same batch ENQUEUED on `stream` - every phase of AlignReads, none of it waited for: the call returns as soon as the launches
 * are made, d_out is complete when `stream` reaches that point.  A consumer kernel enqueued on the same stream (or on another
 * one behind an event recorded there) sees the results without the host ever waiting: the counts the phases produce - reads still
 * unaligned, work items, reads for the wave kernel - stay in device memory and the kernels size themselves by them
 * (the reference's workers never leave their loop either, Aligner.cpp:8943-9527).
 * Needs: the scratch in place for batches of this size and read length (bk_ctx_reserve), max_read_len >= the longest read of the
 * batch (the kernel family is picked by it; a longer read is reported by the NEXT call as BK_ERR_PARAMS), reads of at most 512
 * bases, the default result form (no multi-loci lists, -a / -A / -c, -N), one batch of a context in flight per stream - the
 * scratch belongs to the context.  Otherwise BK_ERR_PARAMS before anything is launched: use the blocking call. */
int  bk_align_batch_device_async(bk_ctx *ctx, const void *d_bases, const void *d_offs, const void *d_lens,
                                 uint32_t nreads, uint32_t max_read_len, void *d_out, void *stream);

/* ---- packed reads: 2 bit/base across PCIe ------------------------------------------------------------------------
 * The 1 byte/base form above is how CAligner holds reads; of its 8 bits the hot path uses 3 (Aligner.cpp:9038-9055) and almost
 * always only 2.  Packed form of a batch:
 *   words  read i owns ceil(lens[i] / 16) consecutive 32-bit words, reads back to back in order; base j of a read sits in bits
 *          31-2(j%16) .. 30-2(j%16) of its word j/16 (first base in the top bits): 0 a, 1 c, 2 g, 3 t.  Bits past a read's end are ignored.
 *   lens   16 bits per read (reads are <= 2000 bases)
 *   exc    every base whose code (bits 0-2 of its byte in the 1 byte/base form) is not 0..3 - the indeterminate base N (4), or anything
 *          else (5..7, which make the reference refuse the read) - as runs: `run` + 1 consecutive bases of a read from `pos` on
 *          carry `code`; strictly ascending by (read, pos), runs do not overlap; the 2-bit fields of such bases are ignored.  Empty
 *          for most batches; a read of nothing but N is one entry.
 * 100-base reads cross PCIe as 30 bytes instead of 104.  Results are those of the 1 byte/base calls, bit for bit. */
typedef struct bk_nbase {
    uint32_t read;               /* index of the read within the batch */
    uint16_t pos;                /* first base of the run              */
    uint8_t  code;               /* 4 (N) .. 7                         */
    uint8_t  run;                /* further bases with the same code (0..255) */
} bk_nbase;                      /* 8 bytes */
/* number of words the packed form of these reads takes */
uint64_t bk_packed_words(const uint32_t *lens, uint32_t nreads);
/* host-side packer (the loader thread's job; uses a few threads): 1 byte/base reads -> the packed form.  offs == NULL: reads lie back
 * to back.  words must hold bk_packed_words(lens, nreads) entries, lens16 nreads, exc exc_cap entries; *n_exc receives the number of exceptions FOUND - when it exceeds exc_cap only the first exc_cap were stored and the call fails
 * with BK_ERR_MEM, to be repeated with a larger array.  Reads longer than 2000 bases: BK_ERR_PARAMS. */
int  bk_pack_reads(const uint8_t *bases, const uint64_t *offs, const uint32_t *lens, uint32_t nreads, uint32_t *words, uint16_t *lens16,
                   bk_nbase *exc, uint64_t exc_cap, uint64_t *n_exc);
/* bk_align_batch over the packed form (host buffers, blocking); n_words = bk_packed_words() of the batch */
int  bk_align_batch_packed(bk_ctx *ctx, const uint32_t *words, uint64_t n_words, const uint16_t *lens, uint32_t nreads,
                           const bk_nbase *exc, uint64_t n_exc, bk_hit *out);

/* Paired-end association after the SE pass = CAligner::ProcessPairedEnds (biokanga/Aligner.cpp:
 * 2876-3489) incl. orphan recovery by CSfxArrayV3::AlignPairedRead (libbiokanga/SfxArrayV2.cpp:8247).
 * Reads and hits are interleaved PE1, PE2, PE1, PE2 ... (2 * n_pairs of each); hits must be the
 * results bk_align_batch() returned for exactly these reads and are updated in place
 * (NAR, NumHits, LowHitInstances, LowMMCnt, Seg[0]); bk_hit.flags bit 7 = FlgPEAligned. */
typedef struct bk_pe_params {
    int32_t pe_mode;        /* -U  1 orphan recovery, 2 unique only, 3 = 1 + leftover ends as SE, 4 = 2 + SE */
    int32_t pair_min_len;   /* -d  minimum insert (default 100)  */
    int32_t pair_max_len;   /* -D  maximum insert (default 1000) */
    int32_t pair_strand;    /* -E  both ends on the same strand  */
} bk_pe_params;
int  bk_pair_batch(bk_ctx *ctx, const uint8_t *bases, const uint64_t *offs, const uint32_t *lens, uint32_t n_pairs,
                   bk_hit *hits, const bk_pe_params *pe);
/* the same on buffers already resident in HBM on the context's GPU (d_hits = what bk_align_batch_device wrote
 * for exactly these reads); nothing crosses PCIe */
int  bk_pair_batch_device(bk_ctx *ctx, const void *d_bases, const void *d_offs, const void *d_lens, uint32_t n_pairs,
                          void *d_hits, const bk_pe_params *pe);
/* The same with the reads' bk_seg2 records (2 * n_pairs, what bk_batch_seg2() returned for exactly these reads; see below), updated in
 * place.  Required on a context with min_chimeric_len > 0 (`-c` together with `-U`; BK_ERR_PARAMS without them): ProcessPairedEnds then
 * measures inserts between the trimmed ends (AdjStartLoci / AdjEndLoci, Aligner.cpp:1528-1544) and AlignPairedRead may place the orphan's
 * partner end-trimmed down to min_chimeric_len percent of its length (MinPutLen, SfxArrayV2.cpp:8327-8330), taking the longest trimmed
 * stretch, then the fewest mismatches, first in scan order; a recovered partner's record is replaced as a whole - flags = FlgChimeric
 * with its trims, or 0 (whatever second segment it had is gone).  A pipeline created with pe != NULL does this by itself. */
struct bk_seg2;
int  bk_pair_batch_seg2(bk_ctx *ctx, const uint8_t *bases, const uint64_t *offs, const uint32_t *lens, uint32_t n_pairs,
                        bk_hit *hits, struct bk_seg2 *seg2, const bk_pe_params *pe);
int  bk_pair_batch_seg2_device(bk_ctx *ctx, const void *d_bases, const void *d_offs, const void *d_lens, uint32_t n_pairs,
                               void *d_hits, void *d_seg2, const bk_pe_params *pe);

/* ---- multi-loci modes (-r1..-r5: AlignReads called with MaxHits = -R > 1) ---------------------- */
/* One locus of a read that aligned to 1..MaxHits places with the same, lowest, number of mismatches: the
 * Seg[0] fields of the tsHitLoci entries LocateCoreMultiples leaves in pHits[] (SfxArrayV2.cpp:6157-6205),
 * in the reference's discovery order ('+' strand first, cores in order, suffix-array order within a core) -
 * the order `-r5` reports them in and `-r2` indexes with rand() (Aligner.cpp:9337-9366).  12 bytes. */
#define BK_MAX_ML 32766
typedef struct bk_loci {
    uint32_t chrom_id;           /* Seg[0].ChromID                  */
    uint32_t match_loci;         /* Seg[0].MatchLoci                */
    uint16_t match_len;          /* Seg[0].MatchLen                 */
    uint8_t  strand;             /* '+' or '-'                      */
    uint8_t  mismatches;         /* Seg[0].Mismatches               */
} bk_loci;
/* Loci lists of the reads of the LAST bk_align_batch()/bk_align_batch_device() call on a context created
 * with max_ml > 1: read r owns loci[offs[r] .. offs[r+1]), that is LowHitInstances entries when its
 * bk_hit.rslt is eHRhits (1 for a unique read, whose record it repeats), max_ml entries when it is eHRHitInsts
 * and clamp_ml is set, and none otherwise.  The pointers
 * are host memory owned by the context, valid until its next align call.  With max_ml == 1 there are no
 * lists (*n_loci = 0, NULL pointers).  What to do with them (-r1 statistics, -r2 random pick, -r3/-r4
 * clustering, -r5 report all; Aligner.cpp:9328-9424,5105-5272) is host policy above this boundary. */
int  bk_batch_loci(bk_ctx *ctx, const uint64_t **offs, const bk_loci **loci, uint64_t *n_loci);
/* With min_chimeric_len > 0 as well (`-c` together with `-r1..5`): the chimeric LocateCoreMultiples call is made with MaxHits = max_ml
 * too and every locus it returns keeps its own end trims (tsSegLoci.TrimLeft / TrimRight of a FlgChimeric hit, read orientation,
 * SfxArrayV2.cpp:6027-6070).  One entry per locus of bk_batch_loci(), in the same order; all zero for loci of the ordinary calls.
 * NULL / 0 on other contexts.  6 bytes. */
typedef struct bk_loci_trims {
    uint16_t left, right;        /* Seg[0].TrimLeft / TrimRight                                  */
    uint8_t  chimeric;           /* FlgChimeric                                                  */
    uint8_t  reserved;
} bk_loci_trims;
int  bk_batch_loci_trims(bk_ctx *ctx, const bk_loci_trims **trims, uint64_t *n_loci);

/* ---- microInDels (-a) ---------------------------------------------------------------------- */
/* Second segment of a read aligned with a microInDel (tsHitLoci.Seg[1] and flags, SfxArrayV2.h:219-240): the read's bk_hit holds
 * the first segment (match_loci / match_len / mismatches of Seg[0]; low_mm = mismatches of both, nxt_low_mm = low_mm + 2 as
 * LocateInDels returns them, SfxArrayV2.cpp:7655-7657).  12 bytes. */
typedef struct bk_seg2 {
    uint32_t match_loci;         /* Seg[1].MatchLoci, 0-based within the entry of Seg[0]          */
    uint16_t match_len;          /* Seg[1].MatchLen                                               */
    uint16_t read_ofs;           /* Seg[1].ReadOfs: first read base of the second segment        */
    uint8_t  mismatches;         /* Seg[1].Mismatches                                             */
    uint8_t  flags;              /* bit 0 FlgInDel, bit 1 FlgInsert (gap is in the read), bit 2 FlgSplice; bit 3 FlgChimeric: no second segment,
                                  * match_len = Seg[0].TrimLeft, read_ofs = Seg[0].TrimRight (read orientation); 0 = plain hit */
    uint16_t score;              /* tsHitLoci.Score                                               */
} bk_seg2;
/* One entry per read of the LAST align call on a context created with micro_indel_len > 0 (host memory owned by the context,
 * valid until its next align call; NULL / 0 otherwise).  Orphan removal (CAligner::RemoveOrphanMicroInDels, Aligner.cpp:2382-2470)
 * is host policy above this boundary. */
int  bk_batch_seg2(bk_ctx *ctx, const bk_seg2 **seg2, uint64_t *n);

/* ---- overlapped pipeline: host buffers in -> host results out ---------------------------------------------------
 * Batch form of the reference's loader-thread / aligner-threads overlap (CAligner::LoadReads + ThreadedIterReads,
 * biokanga/Aligner.cpp:4820-4860,9636-9704): batch k+1 crosses PCIe while batch k runs through the AlignReads phases
 * and the results of batch k-1 travel back, on three HIP streams driven by three host threads owned by the stream
 * object.  T_align of the metric (first batch submitted -> last result back) is what bk_stream_get_stats() reports.
 * While a stream exists the context's own align / pair / loci / seg2 calls must not be used from other threads. */
typedef struct bk_stream bk_stream;
typedef struct bk_stream_stats {
    uint64_t batches, reads;
    uint64_t bytes_h2d, bytes_d2h;                      /* what crossed PCIe */
    double   seconds_first_submit_to_last_result;       /* T_align of the batches since the last reset */
} bk_stream_stats;
/* page-locked host memory (hipHostMalloc): buffers the stream DMAs from / to directly.  Pageable buffers are accepted
 * too and are staged by the HIP runtime (slower). */
void *bk_host_alloc(size_t bytes);
void  bk_host_free(void *p);
/* .. or memory the caller already owns, page-locked in place (hipHostRegister) for as long as results or reads travel through it:
 * what the command line does with its result array.  0 or a negative code; unregister before the memory is freed. */
int   bk_host_register(void *p, size_t bytes);
void  bk_host_unregister(void *p);
/* depth = sets of device buffers (2..4 is useful: one uploading, one aligning, one downloading); every batch may hold up to
 * max_batch_reads reads in max_batch_bases bytes.  pe != NULL: batches hold whole pairs interleaved PE1, PE2 and the
 * paired-end association (bk_pair_batch_device) runs on the resident buffers right after the SE pass. */
int  bk_stream_create(bk_stream **out, bk_ctx *ctx, uint32_t max_batch_reads, uint64_t max_batch_bases, int depth, const bk_pe_params *pe);
/* The same for a caller that only submits packed batches (bk_stream_submit_packed) of at most max_batch_words 32-bit words: the device
 * buffers hold 4 bytes per 16 bases instead of 16, a quarter of the memory (and of the allocation time) of the general form.
 * bk_stream_submit() on such a pipeline is refused. */
int  bk_stream_create_packed(bk_stream **out, bk_ctx *ctx, uint32_t max_batch_reads, uint64_t max_batch_words, int depth, const bk_pe_params *pe);
/* bases[0 .. nbases) holds the reads of this batch, 1 byte/base as CAligner holds them; offs[i] = start of read i within
 * bases, or offs == NULL when the reads lie back to back in lens order (the offsets are then computed on the device and
 * 8 bytes per read stay off PCIe); out[nreads].  Returns once the batch is queued (it blocks only while all `depth`
 * buffer sets are busy); every buffer must stay valid and untouched until bk_stream_wait(ticket) has returned. */
int  bk_stream_submit(bk_stream *s, const uint8_t *bases, uint64_t nbases, const uint64_t *offs, const uint32_t *lens, uint32_t nreads,
                      bk_hit *out, uint64_t *ticket);
/* the same with the batch in the packed form (see bk_align_batch_packed) */
int  bk_stream_submit_packed(bk_stream *s, const uint32_t *words, uint64_t n_words, const uint16_t *lens, uint32_t nreads,
                             const bk_nbase *exc, uint64_t n_exc, bk_hit *out, uint64_t *ticket);
/* blocks until the results of that batch are in its `out`; returns the batch's result code (each ticket once, unless the
 * context runs a list mode - then bk_stream_batch_loci / _seg2 stay available until bk_stream_release) */
/* Device-resident, asynchronous form of bk_align_batch_device (+ bk_pair_batch_device on a pipeline with pe): reads and result
 * buffer already in HBM on the context's GPU.  The call records a point on `producer_stream` (a hipStream_t, NULL = the default stream)
 * and returns; the batch is aligned, in submission order with the pipeline's other batches, once everything enqueued on that stream
 * before the call has run - the caller's thread never waits for the phase loop (CAligner's loader || workers overlap, Aligner.cpp:4820-4860,
 * for a producer that is itself device code).  d_hits is complete, visible to every stream, when bk_stream_wait(ticket) has returned; the
 * buffers must stay untouched until then.  No size limit other than the context's own. */
int  bk_stream_submit_device(bk_stream *s, const void *d_bases, const void *d_offs, const void *d_lens, uint32_t nreads, void *d_hits,
                             void *producer_stream, uint64_t *ticket);
int  bk_stream_wait(bk_stream *s, uint64_t ticket);
int  bk_stream_batch_loci(bk_stream *s, uint64_t ticket, const uint64_t **offs, const bk_loci **loci, uint64_t *n_loci);
int  bk_stream_batch_seg2(bk_stream *s, uint64_t ticket, const bk_seg2 **seg2, uint64_t *n);
int  bk_stream_batch_loci_trims(bk_stream *s, uint64_t ticket, const bk_loci_trims **trims, uint64_t *n_loci);
int  bk_stream_release(bk_stream *s, uint64_t ticket);
/* waits for everything submitted; first failing batch's code or BK_OK */
int  bk_stream_drain(bk_stream *s);
int  bk_stream_get_stats(bk_stream *s, bk_stream_stats *out, int reset);
void bk_stream_destroy(bk_stream *s);

/* counters/timing accumulated since the last reset */
int  bk_get_counters(bk_ctx *ctx, bk_counters *out, int reset);
int  bk_get_timing(bk_ctx *ctx, bk_timing *out, int reset);

/* per-sequence count of accepted reads since the last reset (CAligner::ReportTargHitCnts,
 * Aligner.cpp:5475-5537) - the vector the multi-GPU run sum-reduces over RCCL.  n = bk_num_entries */
int  bk_seq_counts(bk_ctx *ctx, uint64_t *per_entry_hits, uint32_t n, int reset);
/* The path's one exchange step (what CAligner::ReportTargHitCnts and the "@SQ has hits" decision of
 * Aligner.cpp:5475-5537,5606-5637 see on a single host): the count vectors of the `n` contexts of one process - one context per
 * GPU, each fed its own shard of the reads - are sum-reduced where they lie in HBM, with RCCL (ncclAllReduce over xGMI) between
 * distinct devices; contexts sharing a device are added up there first.  `out` (n_entries values, may be NULL) receives the
 * total; reset != 0 then clears the per-context counts.  Multi-process runs reduce bk_seq_counts() through their own
 * communicator instead (bench.py: torch.distributed / RCCL). */
int  bk_seq_counts_allreduce(bk_ctx *const *ctxs, int n, uint64_t *out, uint32_t n_entries, int reset);

/* ---- SNP calling (-p / -P / -1 / -S) --------------------------------------------------------- */
/* The pile-up and the per-locus screening of CAligner::ProcessSNPs (biokanga/Aligner.cpp:7737-7960) and of the first
 * loop of CAligner::OutputSNPs (:6880-7110) on the GPU: per-locus counts of reference / non-reference read bases live
 * in HBM (6 x uint32 per target base), reads are piled up with atomic adds, and only the loci that qualify as putative
 * SNPs travel back.  P-values (CStats::Binomial), the Benjamini-Hochberg cut and the CSV/BED/VCF writers are host
 * policy above this boundary (biokanga_amd/csrc/host/snp.cpp). */
typedef struct bk_snp_aln {      /* one accepted alignment without InDel/splice, after every host filter */
    uint32_t read_idx;           /* read of this call the alignment belongs to                              */
    uint32_t chrom_id;           /* Seg[0].ChromID                                                          */
    uint32_t loci;               /* AdjStartLoci(Seg[0])  (Aligner.cpp:1528)                                */
    uint16_t len;                /* AdjHitLen(Seg[0])     (Aligner.cpp:1546)                                */
    uint16_t read_ofs;           /* Seg[0].ReadOfs + TrimLeft: first read base of the aligned part          */
    uint8_t  strand;             /* '+' | '-' : '-' piles up the reverse complement of those read bases     */
    uint8_t  reserved[3];
} bk_snp_aln;                    /* 20 bytes */
typedef struct bk_snp_site {     /* a locus passing the coverage / non-reference proportion screen (:6932-6960) */
    uint32_t loci;
    uint32_t num_ref;            /* tsSNPcnts.NumRefBases                                                   */
    uint32_t non_ref[5];         /* tsSNPcnts.NonRefBaseCnts a,c,g,t,n (their sum = NumNonRefBases)         */
    uint32_t win_mismatches;     /* LocalTotMismatches of the 51 base background window at this locus       */
    uint32_t win_matches;        /* LocalTotMatches                                                         */
    uint32_t ref_base;           /* tsSNPcnts.RefBase 0..3                                                  */
} bk_snp_site;                   /* 40 bytes */
typedef struct bk_snp_chrom {    /* whole-sequence totals OutputSNPs needs (:6881, :6927-6931) */
    uint64_t tot_match, tot_mismatch;        /* tsChromSNPs.TotMatch / TotMismatch                          */
    uint64_t loci_covered, bases_coverage;   /* contribution to m_LociBasesCovered / m_LociBasesCoverage    */
} bk_snp_chrom;
/* allocate (first call) and zero the per-locus counts */
int  bk_snp_reset(bk_ctx *ctx);
/* pile up `n_alns` alignments of the `nreads` host-resident reads; may be called any number of times */
int  bk_snp_pileup(bk_ctx *ctx, const uint8_t *bases, const uint64_t *offs, const uint32_t *lens, uint32_t nreads,
                   const bk_snp_aln *alns, uint64_t n_alns);
/* the same on reads already resident in HBM on the context's GPU (e.g. the buffers bk_align_batch_device was given) and a device
 * array of alignments; asynchronous on the context's stream unless `sync`.  The caller guarantees what bk_snp_pileup checks:
 * read_idx < nreads, read_ofs + len within the read, chrom_id a sequence of the index */
int  bk_snp_pileup_device(bk_ctx *ctx, const void *d_bases, const void *d_offs, uint32_t nreads, const void *d_alns, uint64_t n_alns, int sync);
/* screen sequence `chrom_id`: loci covered by >= min_reads bases of which >= 1 and a proportion >= min_nonref_prop differ
 * from the target, in ascending loci order (host memory owned by the context, valid until its next call) */
int  bk_snp_sites(bk_ctx *ctx, uint32_t chrom_id, int32_t min_reads, double min_nonref_prop,
                  const bk_snp_site **sites, uint64_t *n_sites, bk_snp_chrom *totals);
/* the counts of `n` consecutive loci of sequence `chrom_id` from `loci` on, 7 uint32 per locus: NumRefBases, NonRefBaseCnts a,c,g,t,n and
 * the target base - what the marker sequences of `-K` are assembled from (OutputSNPs, Aligner.cpp:7006-7086) */
int  bk_snp_counts(bk_ctx *ctx, uint32_t chrom_id, uint32_t loci, uint32_t n, uint32_t *out);
/* SNP centroids (`-7 <file>`): adds, for every locus of sequence `chrom_id` covered by >= min_reads bases and at least 3 bases away from
 * both ends, one to num_insts[k], k = the 7 target bases centred on the locus as a base-4 number (first base most significant; loci with a
 * non-ACGT base in the window are skipped) - tsSNPCentroid.NumInsts of OutputSNPs (Aligner.cpp:6934-6953).  num_insts: 16384 uint32. */
#define BK_SNP_CENTROIDS 16384
int  bk_snp_centroid_insts(bk_ctx *ctx, uint32_t chrom_id, int32_t min_reads, uint32_t *num_insts);

/* ---- SAM records formatted on the device -------------------------------------------------------------------------
 * CAligner::ReportBAMread (biokanga/Aligner.cpp:5768-6126; CSAMfile::AddAlignment, SAMfile.cpp:2100-2283) prints one text line per
 * read; for tens of millions of reads that is byte-parallel work.  bk_sam_format() takes the reads, their names, their records and
 * the order the lines are wanted in (the reference's sort order is the host's business) and hands the lines back as text, in
 * slices: a lane per record measures its line, a prefix sum places it, the lane writes it.  Plain records only - one segment, no
 * end trims (no -a / -A / -c / -x); with pe_mode the reads are interleaved PE1, PE2 and carry the flags bk_pair_batch left.
 * The sink is called once per slice with the slice's text and its offset within the whole text; up to two calls may run at the
 * same time (on threads of the library), each must return 0.  Lines are those of the reference's SAM body, byte for byte. */
struct bk_sam_prep;
typedef struct bk_sam_job {
    const uint8_t  *bases;        /* all reads, 1 byte/base as CAligner holds them (quality in bits 4-7 when loaded)      */
    uint64_t        n_bases;
    const uint64_t *offs;         /* n_reads: start of read i in bases                                                     */
    const uint32_t *lens;         /* n_reads                                                                               */
    const char     *names;        /* the reads' names, NUL-terminated, back to back in read order                          */
    uint64_t        n_name_bytes;
    const uint64_t *name_ofs;     /* n_reads: start of name i in names                                                     */
    const bk_hit   *hits;         /* n_reads records                                                                       */
    uint64_t        n_reads;
    const uint32_t *order;        /* n_order read numbers: the lines in output order                                       */
    uint64_t        n_order;
    int32_t         report_unaligned;   /* -M6: reads without an accepted alignment get a line too (FLAG 4, YU:Z:<reason>) */
    int32_t         pe_mode;            /* 0, or the -U mode the records were paired under                                 */
    struct bk_sam_prep *prep;           /* NULL, or what bk_sam_prepare() started for exactly these reads (consumed by the call)   */
    /* The reads in the packed form of bk_pack_reads instead of bases / offs / lens (those may then be NULL; QUAL is '*'): all reads'
     * words back to back in read order, their 16-bit lengths, and the exceptions with `read` counting from the job's first read.
     * A 100-base read then travels as 30 bytes instead of 112 - from the buffers the alignment was fed from, when those are still there. */
    const uint32_t *pk_words;
    uint64_t        n_pk_words;
    const uint16_t *pk_lens16;
    const bk_nbase *pk_exc;
    uint64_t        n_pk_exc;
} bk_sam_job;
/* Optional head start: everything of a job that is known once the reads are aligned - the read store, the names, the buffers the
 * text leaves the device through - can travel while the host still sorts.  bk_sam_prepare() returns at once (a thread of the library
 * does the work; `job`'s bases / offs / lens / names / name_ofs must stay valid until bk_sam_format() or bk_sam_prep_free()), hits and
 * order are not looked at.  text_bytes_per_record: expected length of a line (sizes the pinned text buffers; 0 = let the first slice
 * size them).  Pass the handle in bk_sam_job.prep; a handle that is not used is given back with bk_sam_prep_free(). */
typedef struct bk_sam_prep bk_sam_prep;
int  bk_sam_prepare(bk_ctx *ctx, const bk_sam_job *job, uint32_t text_bytes_per_record, bk_sam_prep **out);
void bk_sam_prep_free(bk_sam_prep *prep);
/* blocks until the head start's transfers are over (the job's read-side arrays may be released then); returns their result code */
int  bk_sam_prep_wait(bk_sam_prep *prep);
typedef int (*bk_sam_sink)(void *user, const char *text, uint64_t n_bytes, uint64_t text_offset);
int  bk_sam_format(bk_ctx *ctx, const bk_sam_job *job, bk_sam_sink sink, void *user, uint64_t *n_reported, uint64_t *n_bytes);

/* ---- test hook: the search stage on its own ------------------------------------------------------------------------
 * LocateFirstExact / LocateLastExact (SfxArrayV2.cpp:7765-7876, :7914-8027) are two kernels here (k-mer table + key arrays, bk_search.hip); to
 * hold them to the reference's functions probe by probe, a context can be told to stop a batch behind the search of one phase of
 * AlignReads' schedule - bk_ctx_tune(ctx, "debug_stop_phase", p + 1), 0 = off; the batch's result records are then meaningless - and this call
 * returns what that search wrote for the reads still unaligned in that phase (one chunk: batches of at most "chunk_reads" reads):
 *   n_act, iv_cores   reads on the phase's list, cores per strand the records are numbered for
 *   act[n_act]        their numbers in the batch
 *   first / count     [(strand * iv_cores + core) * n_act + position in act]: the suffix array interval [first, first + (count & 0x3fffffff))
 *                     of the core's exact matches; bit 31 of count: a k-mer bucket of at most four suffixes handed on unverified
 *                     ("lazy_search" 1 - the extension drops the members whose core bases differ); bits 31 and 30: a bucket of ONE suffix
 *                     handed on as the suffix itself - `first` is its suffix array ELEMENT, the target position ("use_ktab2" 2: the k-mer
 *                     table's entry carries it, and the extension fetches the window without a trip to the suffix array); a core the
 *                     read does not have: count 0
 * act == NULL: only n_act and iv_cores.  cap_reads: what the arrays hold per plane. */
int  bk_debug_intervals(bk_ctx *ctx, uint32_t cap_reads, uint32_t *n_act, uint32_t *iv_cores, uint32_t *act, uint64_t *first, uint32_t *count);

/* ---- .sfx index construction (CSfxArrayV3::AddEntry/Finalise, kangax.cpp:774-926) ------------ */
/* Suffix-sorts `concat_len` bases resident in HBM (1 byte/base, EOS terminated entries) into
 * d_sa_out (sfx_el_size bytes/element) on the device: order = nibble-lexicographic exactly as
 * QSortSeqCmp32/40 (SfxArrayV2.cpp:9491-9542). */
int  bk_build_sa_device(const void *d_seq, uint64_t concat_len, void *d_sa_out, int sfx_el_size,
                        int device_id);

#ifdef __cplusplus
}
#endif
#endif
