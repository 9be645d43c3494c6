// bk_dev_util.h - device helpers every kernel family shares (gfx950 only): nibble / 2-bit row access, the interval records of a
// phase, striped work-list appends, suffix array access, the core compare, entry look-up, Hamming loops, the result record.
// Host-side helpers that more than one kernel file needs (stripe capacity, list compaction, row expansion) are declared at the end.
#pragma once
#include "bk_device.h"

namespace bk {

// ------------------------------------------------------------------------------------------------
// small device helpers

__device__ __forceinline__ uint64_t nib16(const uint64_t *__restrict__ w, uint64_t pos)
{
    // 16 nibbles starting at base position pos.  Both words are always loaded (every array is
    // padded by at least one word) - straight-line code, no exec-masked second load.
    uint64_t i = pos >> 4;
    unsigned s = (unsigned)(pos & 15) << 2;
    uint64_t a = w[i];
    uint64_t b = w[i + 1];
    return (a << s) | ((b >> 1) >> (63 - s));
}

__device__ __forceinline__ uint64_t top_mask(int nibs)   // mask keeping the first `nibs` (1..16) nibbles
{
    return nibs >= 16 ? ~0ULL : (~0ULL << (64 - 4 * nibs));
}

// 16 bases at 2 bit/base (first base in the top bits) -> 16 nibbles
__device__ __forceinline__ uint64_t spread2to4(uint32_t v)
{
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x << 2)) & 0x3333333333333333ULL;
    return x;
}

// 32 bases from base position pos of a 2 bit/base row (both words are always loaded: rows are followed by at least one more word)
__device__ __forceinline__ uint64_t bits64_2(const uint64_t *__restrict__ w, int pos)
{
    const int i = pos >> 5;
    const unsigned s = (unsigned)(pos & 31) << 1;
    const uint64_t a = w[i], b = w[i + 1];
    return (a << s) | ((b >> 1) >> (63 - s));
}

// One strand's row of a read, whichever form the batch holds it in: 4 bit/base words (rd4), or - lean batches, reads without an N -
// 2 bit/base words (rd2) widened on the fly.  Bases beyond the read's end are whatever follows the row; every user masks by length.
struct RdRow {
    const uint64_t *p;
    bool four;
    __device__ __forceinline__ uint64_t nib16(int pos) const          // 16 nibbles from base position pos
    {
        return four ? bk::nib16(p, (uint64_t)pos) : spread2to4((uint32_t)(bits64_2(p, pos) >> 32));
    }
    __device__ __forceinline__ uint64_t word16(int k) const           // nibbles of bases 16k .. 16k + 15
    {
        if (four) return p[k];
        const uint64_t v = p[k >> 1];
        return spread2to4((k & 1) ? (uint32_t)v : (uint32_t)(v >> 32));
    }
};

__device__ __forceinline__ RdRow read_row(const DevBatch &b, uint32_t r, int strand, bool has_n)
{
    RdRow q;
    q.four = b.rd2 == nullptr || has_n;
    q.p = q.four ? b.rd4 + ((uint64_t)r * 2 + strand) * b.wpr : b.rd2 + ((uint64_t)r * 2 + strand) * (b.nw / 2);
    return q;
}

// core interval slot of (read, strand, core): [strand][core][a], a = the read's position in the phase's ACTIVE list (DevBatch::act),
// so that lanes working on neighbouring active reads touch neighbouring words in every phase - indexed by read number, a later
// phase found one record in two to four of a line still in use.  The records live for one phase: written by its search passes,
// read by its extend kernels (which reach a read of the wave list through its position, act[a]).
__device__ __forceinline__ uint64_t iv_slot(const DevBatch &b, uint32_t a, int st, int c)
{
    return (uint64_t)(st * (int)b.iv_cores + c) * b.iv_stride + a;
}

// core interval records.  4-byte indexes keep {start, count} of a slot in ONE 8-byte word (b.iv2), so every
// access is one line; 5-byte indexes (starts beyond 2^32) keep the two arrays
__device__ __forceinline__ uint32_t iv_count(const DevBatch &b, uint64_t slot)
{
    return b.iv2 ? b.iv2[slot].y : b.iv_n[slot];
}
__device__ __forceinline__ uint64_t iv_start(const DevBatch &b, uint64_t slot)
{
    return b.iv2 ? (uint64_t)b.iv2[slot].x : b.iv_first[slot];
}
__device__ __forceinline__ void iv_get(const DevBatch &b, uint64_t slot, uint64_t &first, uint32_t &n)
{
    if (b.iv2) { const uint2 v = b.iv2[slot]; first = v.x; n = v.y; }
    else { first = b.iv_first[slot]; n = b.iv_n[slot]; }
}
__device__ __forceinline__ void iv_put(const DevBatch &b, uint64_t slot, uint64_t first, uint32_t n)
{
    if (b.iv2) b.iv2[slot] = make_uint2((uint32_t)first, n);
    else { b.iv_first[slot] = first; b.iv_n[slot] = n; }
}

// The counters are kept kCtrStripes times, one 64-byte line each, and summed when they are read back: every block of every launch
// adds to them, and atomics on ONE line are retired one after the other by the L2 (12 ns each, `tools/rand_access_bench atomic`).
__device__ __forceinline__ uint32_t stripe_reserve(const StripeSet &l, int li, uint32_t n)
{
    return atomicAdd(&l.cnt[(blockIdx.x & (kListStripes - 1)) * 16 + li], n);
}
__device__ __forceinline__ void stripe_put(const StripeSet &l, int li, uint32_t at, uint32_t v)
{
    l.stage[li][(uint64_t)(blockIdx.x & (kListStripes - 1)) * l.cap + at] = v;
}
// the same for a kernel whose blocks take several tiles of work: the stripe goes by the tile's number (the capacity is counted in tiles)
__device__ __forceinline__ uint32_t stripe_reserve_tile(const StripeSet &l, int li, uint32_t n, uint64_t tile)
{
    return atomicAdd(&l.cnt[(uint32_t)(tile & (kListStripes - 1)) * 16 + li], n);
}
__device__ __forceinline__ void stripe_put_tile(const StripeSet &l, int li, uint32_t at, uint32_t v, uint64_t tile)
{
    l.stage[li][(tile & (kListStripes - 1)) * l.cap + at] = v;
}
__device__ __forceinline__ void stripe_max(const StripeSet &l, uint32_t v)
{
    atomicMax(&l.cnt[(kListStripes + (blockIdx.x & (kListStripes - 1))) * 16], v);        // a line of its own (see StripeSet)
}

__device__ __forceinline__ uint32_t ctr_stripe() { return (blockIdx.x & (kCtrStripes - 1)) * 8; }

template <bool WIDE>
__device__ __forceinline__ uint64_t sa_get(const DevIndex &ix, uint64_t i)
{
    uint64_t v = ix.sa_lo[i];
    if (WIDE) v |= (uint64_t)ix.sa_hi[i] << 32;
    return v;
}

// probe core (cl bases at read offset ofs; p0 = its first 16 nibbles, masked) vs the suffix at pos.
// <0 / 0 / >0 exactly as the reference's compare loops (EOS in target makes the probe lower).
__device__ __forceinline__ uint64_t row_nib16(const uint64_t *__restrict__ rdw, int pos) { return nib16(rdw, (uint64_t)pos); }
__device__ __forceinline__ uint64_t row_nib16(const RdRow &rdw, int pos) { return rdw.nib16(pos); }

template <typename Row>
__device__ __forceinline__ int cmp_core(const Row &rdw, int ofs, int cl, uint64_t p0,
                                        const uint64_t *__restrict__ tgt, uint64_t pos)
{
    uint64_t t0 = nib16(tgt, pos) & top_mask(cl);
    if (p0 != t0) return p0 < t0 ? -1 : 1;
    for (int i = 16; i < cl; i += 16) {
        uint64_t m = top_mask(cl - i);
        uint64_t p = row_nib16(rdw, ofs + i) & m;
        uint64_t t = nib16(tgt, pos + i) & m;
        if (p != t) return p < t ? -1 : 1;
    }
    return 0;
}

// entry table in LDS for kernels that look an entry up per candidate (the table is a few dozen sequences for
// assembled genomes; bigger tables stay in global memory)
struct LdsEntries {
    uint64_t start[128], end[128];
    bool on;
};
__device__ __forceinline__ void lds_entries_load(LdsEntries &le, const DevIndex &ix)
{
    le.on = ix.n_ent <= 128;
    if (le.on)
        for (uint32_t i = threadIdx.x; i < ix.n_ent; i += blockDim.x) { le.start[i] = ix.ent_start[i]; le.end[i] = ix.ent_end[i]; }
    __syncthreads();
}

__device__ __forceinline__ int find_entry(const DevIndex &ix, uint64_t t)
{
    int lo = 0, hi = (int)ix.n_ent - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if (t < ix.ent_start[mid]) hi = mid - 1;
        else if (t > ix.ent_end[mid]) lo = mid + 1;
        else return mid;
    }
    return -1;
}

__device__ __forceinline__ int find_entry_lds(const LdsEntries &le, const DevIndex &ix, uint64_t t)
{
    if (!le.on) return find_entry(ix, t);
    int lo = 0, hi = (int)ix.n_ent - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if (t < le.start[mid]) hi = mid - 1;
        else if (t > le.end[mid]) lo = mid + 1;
        else return mid;
    }
    return -1;
}

// Hamming distance of the whole read against the target window at t; stops once > limit
__device__ __forceinline__ int hamming(const uint64_t *__restrict__ rdw, int len, const uint64_t *__restrict__ tgt,
                                       uint64_t t, int limit)
{
    int mm = 0;
    for (int i = 0; i < len; i += 16) {
        uint64_t x = (nib16(rdw, i) ^ nib16(tgt, t + i)) & top_mask(len - i);
        x = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ULL;
        mm += __popcll(x);
        if (mm > limit) break;
    }
    return mm;
}

// the same for a window that has not been checked against the entry table: 127 when it holds an EOS (code 7, only
// ever found in the target) - LocateBestMatches' Hamming loop is what keeps its hits inside one entry (:6826-6833)
__device__ __forceinline__ int hamming_eos(const uint64_t *__restrict__ rdw, int len, const uint64_t *__restrict__ tgt,
                                           uint64_t t, int limit)
{
    int mm = 0;
    uint64_t eos = 0;
    for (int i = 0; i < len; i += 16) {
        const uint64_t w = nib16(tgt, t + i), m = top_mask(len - i);
        eos |= w & (w >> 1) & (w >> 2) & m & 0x1111111111111111ULL;
        uint64_t x = (nib16(rdw, i) ^ w) & m;
        x = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ULL;
        mm += __popcll(x);
    }
    return (eos || mm > limit) ? 127 : mm;
}

// 2-bit code of the first 16 nibbles (first base in the top 2 bits)
__device__ __forceinline__ uint32_t squeeze2(uint64_t x)
{
    x &= 0x3333333333333333ULL;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFULL;
    return (uint32_t)x;
}

__device__ __forceinline__ uint32_t flags_to_bits16(uint64_t f)   // f: flags at bit 0 of each nibble, base 0 in the top nibble
{
    uint64_t g = __brevll(f) >> 3;                                 // flag of base k now at bit 4k
    g = (g | (g >> 3)) & 0x0303030303030303ULL;
    g = (g | (g >> 6)) & 0x000F000F000F000FULL;
    g = (g | (g >> 12)) & 0x000000FF000000FFULL;
    g = (g | (g >> 24)) & 0xFFFFULL;
    return (uint32_t)g;                                            // bit k = base k
}

__device__ __forceinline__ uint64_t ktab_get(const DevIndex &ix, uint64_t c)
{
    if (ix.ktab2) return (uint64_t)ix.ktab2[c].x;
    if (ix.ktab_hi) return ix.ktab_hi[c >> 16] + (uint64_t)ix.ktab32[c];       // (a 512 KB array: it stays in the caches)
    return ix.ktab32 ? (uint64_t)ix.ktab32[c] : ix.ktab64[c];
}

// a bucket's start and end (the next bucket's start) in one go: the packed table's 64-bit start is looked up once for both
__device__ __forceinline__ void ktab_get_pair(const DevIndex &ix, uint64_t c, uint64_t &lo, uint64_t &hi)
{
    if (ix.ktab_hi && !ix.ktab2) {
        const uint64_t base = ix.ktab_hi[c >> 16];
        const uint32_t o0 = ix.ktab32[c], o1 = ix.ktab32[c + 1];
        lo = base + o0;
        hi = ((c + 1) & 0xFFFFULL) ? base + o1 : ix.ktab_hi[(c + 1) >> 16];       // (a group's first offset is 0)
        return;
    }
    lo = ktab_get(ix, c);
    hi = ktab_get(ix, c + 1);
}

// SA index range [lo, hi) that can contain suffixes starting with the core
__device__ __forceinline__ void core_range(const DevIndex &ix, uint64_t p0, int cl, uint64_t &lo, uint64_t &hi)
{
    lo = 0;
    hi = ix.n;
    int k = ix.k;
    if (k <= 0) return;
    int kk = cl < k ? cl : k;
    if (p0 & 0x4444444444444444ULL & top_mask(kk)) return;   // an N inside the indexed prefix
    uint32_t code = squeeze2(p0);
    uint64_t c_lo = (uint64_t)(code >> (32 - 2 * kk)) << (2 * (k - kk));
    uint64_t c_hi = c_lo | ((1ULL << (2 * (k - kk))) - 1);
    lo = ktab_get(ix, c_lo);
    hi = ktab_get(ix, c_hi + 1);
}

// lower bound (LocateFirstExact) + length of the matching run, capped at `cap`
template <bool WIDE, typename Row>
__device__ __forceinline__ void search_core(const DevIndex &ix, const Row &rdw, int ofs, int cl,
                                            uint64_t cap, uint64_t &first, uint64_t &count)
{
    uint64_t p0 = row_nib16(rdw, ofs) & top_mask(cl);
    uint64_t lo, hi;
    core_range(ix, p0, cl, lo, hi);
    uint64_t end = hi;
    while (lo < hi) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        int c = cmp_core(rdw, ofs, cl, p0, ix.tgt4, sa_get<WIDE>(ix, mid));
        if (c > 0) lo = mid + 1;
        else hi = mid;
    }
    first = lo;
    count = 0;
    if (lo >= end) return;
    if (cmp_core(rdw, ofs, cl, p0, ix.tgt4, sa_get<WIDE>(ix, lo)) != 0) return;
    if (end - lo == 1) { count = 1; return; }            // bucket of one suffix: nothing else can match
    // gallop over the run of matches, then bisect its end
    uint64_t limit = end - lo < cap ? end : lo + cap;    // exclusive
    uint64_t cur = lo, step = 1;
    while (cur + step < limit && cmp_core(rdw, ofs, cl, p0, ix.tgt4, sa_get<WIDE>(ix, cur + step)) == 0) {
        cur += step;
        step <<= 1;
    }
    uint64_t l2 = cur + 1, h2 = cur + step < limit ? cur + step : limit;
    while (l2 < h2) {
        uint64_t mid = l2 + ((h2 - l2) >> 1);
        if (cmp_core(rdw, ofs, cl, p0, ix.tgt4, sa_get<WIDE>(ix, mid)) == 0) l2 = mid + 1;
        else h2 = mid;
    }
    count = l2 - lo;
}

// classification at the end of LocateCoreMultiples for a call that started from the fresh state
// (SfxArrayV2.cpp:6238-6261); init = MaxTotMM + MMDelta + 1
__device__ __forceinline__ int classify(int low_inst, int low_mm, int nxt, int init, int mm_delta, int max_hits)
{
    if (low_inst == 0 && low_mm == init) return BK_HR_NONE;
    if (low_inst >= 1 && (nxt - low_mm) < mm_delta) return BK_HR_MMDELTA;
    if (low_inst > max_hits) return BK_HR_HITINSTS;
    return BK_HR_HITS;
}

// HitRslt -> tsReadHit fields, default MLMode (Aligner.cpp:9241,9311-9479)
__device__ __forceinline__ void write_result(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, uint32_t r,
                                             int len, int rslt, int low_inst, int low_mm, int nxt, uint64_t hit_left,
                                             int hit_ent, int hit_strand, int diag)
{
    bk_hit h;
    h.chrom_id = 0; h.match_loci = 0; h.match_len = 0; h.low_hit_instances = 0;
    h.rslt = (uint8_t)rslt; h.nar = BK_NAR_NOHIT; h.strand = '?'; h.low_mm = 0; h.nxt_low_mm = 0;
    h.num_hits = 0; h.mismatches = 0;
    h.flags = (uint8_t)diag;      // diagnostics only: (AlignReads phase << 1) | resolved by k_heavy
    if (low_inst > cfg.max_hits) low_inst = cfg.max_hits + 1;
    switch (rslt) {
    case BK_HR_HITS:
        if (low_inst == 1) {
            h.nar = BK_NAR_ACCEPTED;
            h.num_hits = 1;
            h.strand = (uint8_t)hit_strand;
            h.chrom_id = ix.ent_id[hit_ent];
            h.match_loci = (uint32_t)(hit_left - ix.ent_start[hit_ent]);
            h.match_len = (uint16_t)len;
            h.mismatches = (uint8_t)low_mm;
        } else
            h.nar = BK_NAR_MULTIALIGN;
        break;
    case BK_HR_MMDELTA:
        h.nar = BK_NAR_MMDELTA;
        h.match_len = (uint16_t)len;
        break;
    case BK_HR_HITINSTS:
        h.nar = BK_NAR_MULTIALIGN;
        h.match_len = (uint16_t)len;
        break;
    default:
        break;
    }
    h.low_hit_instances = (int16_t)low_inst;
    h.low_mm = (int8_t)low_mm;
    h.nxt_low_mm = (int8_t)nxt;
    b.out[r] = h;
}

__device__ __forceinline__ uint64_t uniform64(uint64_t v)      // value known to be wave-uniform -> scalar registers
{
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// a core interval of at most kLazyBucket suffixes may be handed on by the search unverified (kLazyFlag in its count): the extend
// kernels keep the members whose core bases are clean in the window they evaluate anyway (bk_search.hip)
constexpr uint32_t kLazyBucket = 4;
constexpr uint32_t kLazyFlag = 0x80000000u;
// .. and of a bucket of ONE suffix the search may hand on the suffix itself: with kElemFlag (and kLazyFlag, count 1) the record's `first` is
// the suffix array ELEMENT - the target position - that the k-mer table's entry carried (DevIndex::ktab2_elem), not the suffix's index:
// the candidate's window is fetched without the trip to the suffix array in front of it (k_flat's chain: record -> window instead of
// record -> element -> window; the kernel that evaluates the window checks the core's bases there, as for every unverified bucket)
constexpr uint32_t kElemFlag = 0x40000000u;
constexpr uint32_t kIvFlags = kLazyFlag | kElemFlag;
constexpr int kWaveGrab = 8;            // work items a wave of the wave-per-read kernels claims per atomic on the shared cursor

// ---- host side, shared by the launchers of the kernel files ----------------------------------
void launch_fill_u64(unsigned long long *p, uint64_t n, unsigned long long v, hipStream_t s);       // bk_index.hip
// 4-bit rows of both strands for every read of the batch (general kernel family, paired-end kernels): k_pack_reads (+ the exceptions
// of a packed batch), bk_prep.hip
void launch_pack_rows(const DevBatch &b, hipStream_t s, const uint32_t *pairs = nullptr, uint32_t n_pairs = 0);
struct CompactJobs { StripeSet set; uint32_t *dense[3]; uint32_t *total[3]; uint32_t *max_out; int n; };
// capacity of one stripe of a list that `blocks` blocks append at most `per_block` entries each to (bk::StripeSet)
inline uint32_t stripe_cap(unsigned blocks, unsigned per_block) { return ((blocks + kListStripes - 1) / kListStripes) * per_block; }
// the stripes of up to three lists -> their dense forms (bk_prep.hip)
void launch_compact(const StripeSet &set, uint32_t *const *dense, uint32_t *const *total, int n, uint32_t *max_out, hipStream_t s);
// 4-bit rows for the reads of `list` in a lean batch (bk_prep.hip)
void expand_rd4(const DevBatch &b, const uint32_t *list, uint32_t n_list, hipStream_t s);

}  // namespace bk
