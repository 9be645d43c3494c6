// bk_device.h - device-side data layout and helpers shared by the HIP kernels (gfx950 only).
//
// HBM layout of the index image
//   tgt4   : target bases packed 4 bit/base, 16 bases per uint64, FIRST base in the MOST significant
//            nibble, so that an unsigned compare of two words is the reference's low-nibble
//            lexicographic compare (A0 C1 G2 T3 N4 < EOS7; SfxArrayV2.cpp:7791-7812).  Padded with
//            EOS nibbles past concat_len so windows never read out of bounds.
//   tgt2   : the same bases at 2 bit/base (32 per uint64, first base in the top bits) for the window compare of
//            the extend kernels: a 100-base window is 25 bytes = 1.8 32-byte sectors instead of 2.6.  N/EOS
//            cannot be held; nflag (1 bit per 2^flag_shift bases, sized to stay <= 16 KB so that it lives in the L1 -
//            a bigger bitmap costs one more cache miss per candidate, which is what bounds these kernels) says where the
//            4-bit copy has to be used instead
//   sa_lo  : uint32[N] low 32 bits of each suffix array element; sa_hi: uint8[N] bits 32..39 when
//            the .sfx uses 5-byte elements (SfxOfsToLoci, SfxArrayV2.cpp:33-44)
//   ktab   : uint32/uint64[4^k + 1]; ktab[c] = number of suffixes sorting before k-mer code c, i.e.
//            the LocateFirstExact lower bound of c; a core whose first k bases have code c can only
//            match inside [ktab[c], ktab[c+1])
//   k2     : uint32[N] second-level search keys in suffix-array order: the 15 bases that follow the first k of suffix sa[i]
//            at 2 bit/base + a kind (see bk_dev_k2.h); sorted inside every k-mer bucket, so cores are located by
//            bisecting contiguous keys - sixteen to a cache line - instead of chasing sa -> target
//   isa    : uint32[N] inverse suffix array (isa[sa[i]] = i); lets the wave kernel decide whether a
//            target start already reached through an earlier, TRUNCATED core interval was inside
//            the part of that interval the reference processed (replaces its per-thread hash set)
//   ent_*  : tsSfxEntry StartOfs/EndOfs/EntryID (MapChunkHit2Entry, SfxArrayV2.cpp:2530-2575)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/biokanga_amd.h"

namespace bk {

// sampled levels of the second-level keys, stored behind them in the same allocation (bk_dev_k2.h)
constexpr int kMoreKeys = 2;                        // key arrays behind the second-level keys (DevIndex::kx)
constexpr int kK2Levels = 7;                        // 16^8 > the largest interval a work item carries (2^kKindShift)

__host__ __device__ __forceinline__ uint64_t k2s_pad(uint64_t words) { return (words + 15) & ~15ULL; }
// entries of level j
__host__ __device__ __forceinline__ uint64_t k2s_count(uint64_t n, int j) { return (n + (1ULL << (4 * j)) - 1) >> (4 * j); }
// where level j starts, in 4-byte words from k2[0] (level kK2Levels + 1 "starts" where the allocation ends); every level starts on
// a 64-byte line and is followed by a line of padding, so whole lines may be loaded around any entry
__host__ __device__ __forceinline__ uint64_t k2s_start(uint64_t n, int j)
{
    uint64_t o = k2s_pad(n) + 16;
    for (int i = 1; i < j; i++) o += k2s_pad(k2s_count(n, i)) + 16;
    return o;
}


// A work list that every block of a launch appends to.  One shared counter means one L2 line retiring every block's atomic in
// turn, 12 ns each (`tools/rand_access_bench atomic`: 200 000 blocks, one atomic each, 2.4 ms on one line, 0.07 ms on 64 lines) -
// that, not memory latency, was what k_flat, pass A of the search and the read packing waited for.  So a block appends to stripe
// (block number mod kListStripes): counters on the stripe's own line, its own region of `cap` entries per list; a small kernel then
// copies the stripes into the dense list the next kernel reads and adds their sizes to the list's count (launch_compact).  The
// "most cores any read of the next phase has" maximum travels the same way.
constexpr int kListStripes = 64;
struct StripeSet {
    uint32_t *cnt;              // 2 * kListStripes lines of 16 words: line s [0..2] = sizes of stripe s of lists 0..2, line kListStripes + s [0] = a running maximum
    uint32_t *stage[3];         // per list: kListStripes regions of cap entries
    uint32_t cap;
};
constexpr int kCtrStripes = 64;       // copies of the DevBatch::ctr block (8 counters = one 64-byte line each)
constexpr int kNwLong = 20;            // 16-base words of the wider register-window kernel families: reads of up to 320 bases (2 x 250, 2 x 300)
constexpr int kNwLongest = 32;         // .. and of up to 512 bases
constexpr int kPackedPadWords = 32;    // 32-bit words every buffer of packed reads (DevBatch::pk_words) is padded by: the read preparation loads a
                                       //   whole register window (up to kNwLongest words) from a read's first word, however short the read
constexpr int kMaxCoresFast = 16;      // cores per strand the lane-per-read path handles
constexpr int kWave = 64;
constexpr int kMaxReadLenAbs = 2000;   // cMaxFastQSeqLen upper bound of -L (Aligner.h:94)
constexpr uint32_t kNodeCap = 1024000; // cMaxNumIdentNodes, SfxArrayV2.h:15
constexpr uint64_t kEosWord = 0x7777777777777777ULL;
constexpr int kSwBases = 192, kSwPre = 92, kSwLen = 100;      // DevIndex::swin, entries of three 16-byte words (reads of up to 128 bases)
// .. and by the entry's size in 16-byte words: 3 = 192 bases from 92 before the suffix on (every core of a read of <= 100 bases), 5 = 320
// bases from 160 before it on (every core of a read of <= 160 bases: 2 x 150), which the 16-word kernel family takes its windows from
template <int E> struct SwGeo { static constexpr int bases = 64 * E, pre = E == 3 ? 92 : 32 * E; };
// Where word q (0 .. E - 1) of entry idx lies in swin, in 16-byte units: the entries of a block of 32 are stored word by word - 32 first
// words, 32 second words, .. - so that the 64 lanes of a round read each word from contiguous memory, and a core whose window does not
// reach into an entry's first or last 64 bases (two of a 100-base read's four cores at 25 bases) leaves those words' lines untouched.
template <int E> __host__ __device__ inline uint64_t sw_word_at(uint64_t idx, int q) { return (idx >> 5) * (32ULL * E) + (uint64_t)q * 32 + (idx & 31); }
constexpr int kSwBlkShift = 5;                                // DevIndex::swmap: coverage goes by blocks of 32 suffix array indexes
constexpr uint32_t kSwNone = 0xFFFFFFFFu;                     //   .. a block the window array does not hold
constexpr uint32_t kSwMinRun = 65, kSwHead = 192;             //   .. the coverage rule (bk_index.hip, k_swin_cover)
constexpr int kSwLevels = 8;                                  //   .. applied for this many core lengths at most (the reads' last phases)
constexpr uint32_t kReadHasN = 1u << 15;   // DevBatch::rmeta
constexpr uint32_t kReadLenMask = kReadHasN - 1;

struct DevIndex {
    const uint64_t *tgt4;
    const uint64_t *tgt2;       // 2 bit/base copy of the target (32 bases per word), may be null
    const uint64_t *tgt2s;      // tgt2 again, physically shifted by 32 bytes (windows never straddle a 64-byte line); may be null
    const uint8_t *nflag;       // bit per 2^flag_shift bases: region holds N/EOS (tgt2 unusable there); <= 16 KB, L1 resident
    int flag_shift;
    uint32_t nflag_bytes;       // .. its size (a kernel may keep a copy of it in LDS)
    const uint32_t *sa_lo;
    const uint8_t *sa_hi;       // null unless 5-byte elements
    const uint64_t *ent_start;
    const uint64_t *ent_end;
    const uint32_t *ent_id;
    const uint32_t *id2idx;     // EntryID -> entry index (0xffffffff = no such id), max_id + 1 elements: ids need not be 1..n in file order
    uint32_t max_id;
    const uint32_t *ktab32;     // one of ktab32/ktab64 when k > 0
    const uint64_t *ktab64;
    const uint64_t *ktab_hi;    // with ktab32, an index beyond 2^32 suffixes: bucket start = ktab_hi[c >> 16] + ktab32[c] - half of ktab64's bytes (k_pack_ktab64)
    const uint2 *ktab2;         // instead of ktab32 when the second-level keys exist: {ktab32[c], y} - y = the key of a bucket of one suffix (no second line), the map of the first five bits of a larger bucket's keys (k_make_ktab2)
    int ktab2_elem;             // ktab2's second word of a bucket of one suffix is that suffix's array ELEMENT (its target position), not its second-level key: the search hands it on (kElemFlag)
    const uint32_t *k2;         // second-level keys: the 15 bases following the first k of suffix sa[i], 2 bits each + kind; may be null
    const uint32_t *kx[kMoreKeys];   // third-, fourth-level keys: the 15 bases after those and the 15 after these, same form; all ones where the level before is not of kind 0; null from the first level the index does without
    const uint32_t *isa;        // inverse suffix array (rank of every position), 4-byte indexes only; may be null
    const uint4 *swin;          // suffix-ordered windows: for every suffix array index i the kSwBases bases of the 2-bit target from
                                //   sa[i] - kSwPre on, 48 bytes each.  The candidates of a core interval - consecutive suffix array
                                //   elements - then fetch their target windows from CONSECUTIVE entries (a streaming read, 130 G
                                //   entries/s) instead of one random cache line each (50 G/s).  Reads of up to kSwLen bases whose
                                //   core offsets stay within kSwPre; 4-byte indexes with 48 bytes per base of HBM to spare; may be null
    int sw_words;               // 16-byte words per entry of swin: 3 or 5 (SwGeo)
    const uint32_t *swmap;      // null: swin holds every suffix.  Else swin holds only the blocks of 2^kSwBlkShift suffix array indexes that the
                                //   wave kernel's long walks visit: swmap[i >> kSwBlkShift] = the block's number in swin, or kSwNone.
                                //   Neighbouring covered blocks have neighbouring numbers, so a core interval whose first and last
                                //   block are as far apart in swin as in the suffix array reads its entries from ONE run of swin
    uint64_t n;                 // concat_len
    uint32_t n_ent;
    int k;                      // k-mer table order (0 = none)
};

struct DevAlignCfg {            // derived once per context (CAligner::LocateCoredApprox, Align)
    int max_subs, mm_delta, align_strand, max_ns, max_hits;
    int min_core_len, slides_per100, max_iter;
    int heavy_thresh;           // intervals longer than this go to the wave-per-read kernel
};

struct DevBatch {
    const uint8_t *bases;       // 1 B/base as CAligner holds them; null when the batch came packed (pk_words)
    const uint64_t *offs;       // start of every read in bases (or, packed batches, its first word in pk_words)
    const uint32_t *lens;
    const uint32_t *pk_words;   // packed batches (bk_align_batch_packed): 16 bases per word at 2 bit/base, first base in the top bits
    const bk_nbase *pk_exc;     //   the bases that are not a,c,g,t, sorted by (read, position); read numbers count from the first read
    uint64_t pk_nexc;           //   of the submitted batch, of which this chunk holds reads pk_read0 .. pk_read0 + n_reads - 1
    uint32_t pk_read0;
    uint64_t *rd4;              // [read][strand][wpr] packed nibble words (fwd, revcomp).  With rd2 set ("lean" batches) only the rows of
                                //   reads that hold an N are written by the read preparation; k_expand_rd4 fills in the rows of the few
                                //   reads a kernel of the general family is about to see
    uint64_t *rd2;              // [read][strand][nw/2]: the read and its reverse complement at 2 bit/base, 32 bases per word, first base
                                //   in the top bits, N held as A, zero beyond the read's end; may be null (then rd4 holds every read)
    uint32_t *rmeta;            // per read, written by the read preparation: length | has-N flag << 15 (kReadHasN)
    uint64_t *iv_first;         // core intervals [strand][core][read]: start (suffix array index) ...
    uint32_t *iv_n;             // ... and count | flags - separate arrays only for 5-byte indexes
    uint2 *iv2;                 // 4-byte indexes: {start, count | flags} in one word; then iv_first/iv_n are null
    uint32_t iv_stride;         // records per (strand, core) plane of the interval arrays in this phase = length of its active list
    const uint32_t *act;        // the phase's active list (read numbers): interval records and the wave list are indexed by position in it
    uint32_t *wave_work;        // per read: candidates the wave kernel will walk (sum of its core intervals), left by k_flat when it hands the read on; may be null
    uint2 *iv32;                // [strand][read]: suffix array interval of the read's first k + 16 bases {start, count; count 0xffffffff = not known},
                                //   left by phase 0 for the offset-0 cores of the later phases (4-byte indexes); may be null
    bk_hit *out;
    unsigned long long *seq_counts;   // per entry accepted reads
    unsigned long long *ctr;          // kCtrStripes x 8: [0] n_search [1] n_cand [2] n_lcm [3] n_heavy
    uint32_t wpr;
    uint32_t n_reads;
    uint32_t nw;                // 4-bit words covered by an rd2 row (8 or 16), 0 without rd2
    uint32_t iv_cores;          // cores per strand the interval slots are numbered for (<= kMaxCoresFast: the most a read of this batch can have)
};

// The counts of one AlignReads phase over a chunk of reads, in device memory: the phase loop's kernels size themselves by them, so
// the host never has to read a count back between two launches (a launch's grid comes from a bound the host does know - a phase's
// active list is never longer than the one before it - and blocks beyond the count have nothing to do).  One 64-byte line per phase,
// zeroed when a chunk starts; phase p's kernels read ctl[p] and leave n_act of the next phase in ctl[p + 1].
struct PhaseCtl {
    uint32_t n_act;             // reads on the phase's active list
    uint32_t cmax;              // most cores any of them has (informative: launches use the bound of the batch's longest read)
    uint32_t n_slist;           // work items pass A left for pass B
    uint32_t n_wave;            // reads k_flat handed to the wave kernel
    uint32_t n_heavy;           // .. and to the general kernel
    uint32_t wave_cursor, heavy_cursor;
    uint32_t pad[9];
};
constexpr int kMaxPhases = 72;          // AlignReads runs at most MaxTotMM + 2 <= 65 LocateCoreMultiples calls (cMaxTotAllowedSubs 63)

struct HeavyScratch {
    unsigned long long *htab;   // slots * tab_size entries of (epoch<<32 | key)
    uint32_t *slot_epoch;
    uint32_t tab_size;          // power of two
    uint32_t n_slots;
};

// ------------------------------------------------------------------------------------------------
struct ReadPlan {
    int len, max_tot_mm, core_len, core_delta, max_slides, n_loop, n_phases;
};

// CAligner::ProcCoredApprox parameter derivation (Aligner.cpp:9085-9095) + the phase schedule of
// CSfxArrayV3::AlignReads (SfxArrayV2.cpp:7695-7719)
__host__ __device__ inline ReadPlan make_plan(int len, const DevAlignCfg &c)
{
    ReadPlan p;
    p.len = len;
    int m = c.max_subs == 0 ? 0 : (int)(0.5 + (double)(len * c.max_subs) / 100.0);
    if (c.max_subs != 0 && m < 1) m = 1;
    if (m > 63) m = 63;
    p.max_tot_mm = m;
    int cl = len / (c.mm_delta == 1 ? m + 1 : m + 2);
    p.core_len = cl > c.min_core_len ? cl : c.min_core_len;
    int ms = (c.slides_per100 * len + 99) / 100;
    p.max_slides = ms > 1 ? ms : 1;
    int cd = len / p.max_slides - 1;
    p.core_delta = cd > p.core_len ? cd : p.core_len;
    p.n_loop = 0;
    int has_final = 1;
    if (m > 0) {
        int a;
        for (a = 0; a <= m; a++) {
            if (len / (a + c.mm_delta) <= p.core_len) break;
            p.n_loop++;
        }
        has_final = a <= m;
    }
    p.n_phases = p.n_loop + has_final;
    return p;
}

__host__ __device__ inline void phase_params(const ReadPlan &p, const DevAlignCfg &c, int phase, int &mm, int &cl, int &cd)
{
    if (phase < p.n_loop) {
        mm = phase;
        cl = p.len / (phase + c.mm_delta);
        cd = cl;
    } else {
        mm = p.max_tot_mm;
        cl = p.core_len;
        cd = p.core_delta;
    }
}

// core sliding rule of LocateCoreMultiples (SfxArrayV2.cpp:5836-5847); returns the number of cores,
// writes the first `maxn` offsets.  The rule advances by `cd` until a core would overrun the read,
// then places one last core flush with the read's end, so offset i is min(i*cd, plen-cl).  The
// offsets are written with compile-time indexes after the counting loop: a data-dependent
// `if (n < maxn) ofs[n] = o` inside the loop was miscompiled for gfx950 (hipcc 7.2) when the count
// exceeded maxn - the 17th offset landed in a neighbouring register.
__host__ __device__ inline int core_offsets(int plen, int cl, int cd, int max_slides, int *ofs, int maxn)
{
    int cur = cd, o = 0, n = 0;
    while (n < max_slides && o <= plen - cl && cur > cl / 3) {
        if (o + cl + cur > plen) cur = plen - (o + cl);
        n++;
        o += cur;
    }
    const int last = plen - cl;
    for (int i = 0; i < maxn; i++) {
        int v = i * cd;
        ofs[i] = v < last ? v : last;
    }
    return n;
}

}  // namespace bk
