// bk_kernels.hip - hand-written gfx950 (CDNA4, wave64) kernels of the `biokanga align` hot path.
//
// index set-up
//   k_pack_target / k_pack_target2   1 B/base target -> 4 bit/base words; 2 bit/base copy + N/EOS region bitmap
//   k_split_sa5                      5-byte suffix elements -> lo32 + hi8 arrays
//   k_build_ktab / k_build_k2 / k_check_k2 / k_build_isa   k-mer table, second-level keys, inverse suffix array
// per batch (register-kernel path: reads of <= 256 bases, <= 16 cores per strand)
//   k_prep_fused    pack read + reverse complement (SeqTrans.cpp:458-512) as 4-bit and 2-bit rows, N policy
//                   (Aligner.cpp:9041-9063), result record, first active list
//   k_search_a/_b   LocateFirstExact (SfxArrayV2.cpp:7765) + extent of the matching run for every core of the phase:
//                   k-mer table + contiguous second-level keys, work list grouped by bucket for the bisection pass
//   k_flat          LocateCoreMultiples (SfxArrayV2.cpp:5830-6261) for reads whose core intervals are all short:
//                   one candidate per lane, the best / next-best / instances outcome reduced over the candidates' lanes
//   k_wave          the same call for repeat reads: one wave per call, 64 candidates per round, ballot prefix sums
//                   reproduce the reference's sequential order (100-candidate copy-count cut-off, MaxIter, node
//                   cap, early exit); dedupe by inverse suffix array, or by the reference's hash set (5-byte indexes)
//   k_count_seqs    per-sequence accepted-read histogram
//   k_pe_classify / k_pe_orphan      paired-end association and orphan recovery (Aligner.cpp:2726-3489)
// general path (longer reads, more cores; also selectable for cross-checks)
//   k_pack_reads / k_init_reads, k_search, k_light / k_extend, k_heavy
//
// Integer / bit-compare work bound by random cache-line misses: no MFMA anywhere.
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
__device__ __forceinline__ void stripe_max(const StripeSet &l, uint32_t v)
{
    atomicMax(&l.cnt[(kListStripes + (blockIdx.x & (kListStripes - 1))) * 16], v);        // a line of its own (see StripeSet)
}

// stripes -> dense lists, appended behind what each list already holds; k_finish_lists then adds the sizes to the lists' counts,
// folds the maximum into *max_out and clears the stripes' lines for the next launch
struct CompactJobs { StripeSet set; uint32_t *dense[3]; uint32_t *total[3]; uint32_t *max_out; int n; };

__global__ void __launch_bounds__(256) k_compact_lists(CompactJobs J)
{
    const int li = blockIdx.y;
    __shared__ uint32_t s_pre[kListStripes + 1];
    if (threadIdx.x < 64) {
        uint32_t run = 0;
        for (int s0 = 0; s0 < kListStripes; s0 += 64) {
            uint32_t v = J.set.cnt[(s0 + threadIdx.x) * 16 + li];
            for (int off = 1; off < 64; off <<= 1) { uint32_t u = __shfl_up(v, off); if ((int)threadIdx.x >= off) v += u; }
            s_pre[s0 + threadIdx.x + 1] = run + v;
            run += __shfl(v, 63);
        }
        if (threadIdx.x == 0) s_pre[0] = 0;
    }
    __syncthreads();
    const uint32_t tot = s_pre[kListStripes], old = *J.total[li];
    const uint32_t *__restrict__ stage = J.set.stage[li];
    uint32_t *__restrict__ dense = J.dense[li];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += gridDim.x * 256) {
        int lo = 0, hi = kListStripes - 1;                  // stripe s: s_pre[s] <= i < s_pre[s + 1]
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_pre[mid] <= i) lo = mid; else hi = mid - 1;
        }
        dense[old + i] = stage[(uint64_t)lo * J.set.cap + (i - s_pre[lo])];
    }
}

__global__ void k_finish_lists(CompactJobs J)
{
    uint32_t sum[3] = {0, 0, 0}, mx = 0;
    for (int s0 = threadIdx.x; s0 < kListStripes; s0 += 64) {
        uint32_t *line = J.set.cnt + s0 * 16;
#pragma unroll
        for (int li = 0; li < 3; li++) { sum[li] += line[li]; line[li] = 0; }
        uint32_t *mline = J.set.cnt + (kListStripes + s0) * 16;
        mx = mline[0] > mx ? mline[0] : mx;
        mline[0] = 0;
    }
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int li = 0; li < 3; li++) sum[li] += __shfl_down(sum[li], off);
        const uint32_t q = __shfl_down(mx, off);
        mx = q > mx ? q : mx;
    }
    if (threadIdx.x == 0) {
        for (int li = 0; li < J.n; li++) if (sum[li]) *J.total[li] += sum[li];
        if (J.max_out && mx > *J.max_out) *J.max_out = mx;
    }
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
    return ix.ktab32 ? (uint64_t)ix.ktab32[c] : ix.ktab64[c];
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

// ------------------------------------------------------------------------------------------------
// index upload kernels

__global__ void k_pack_target(const uint8_t *__restrict__ seq, uint64_t n, uint64_t *__restrict__ tgt4, uint64_t nwords)
{
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; w < nwords; w += stride) {
        uint64_t base = w << 4;
        uint64_t v = 0;
        if (base + 16 <= n) {
            const uint4 q = *reinterpret_cast<const uint4 *>(seq + base);
            uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = 0; k < 4; k++) v = (v << 4) | ((d[j] >> (8 * k)) & 0x0f);
        } else {
            for (int k = 0; k < 16; k++) {
                uint64_t p = base + k;
                uint64_t nb = p < n ? (uint64_t)(seq[p] & 0x0f) : 7ULL;
                v = (v << 4) | nb;
            }
        }
        tgt4[w] = v;
    }
}

// 2 bit/base copy + "block holds N/EOS" bitmap, derived from the packed 4-bit target (padding included)
__global__ void k_pack_target2(const uint64_t *__restrict__ tgt4, uint64_t nwords4, uint64_t *__restrict__ tgt2,
                               unsigned int *__restrict__ nflag32, int flag_shift)
{
    // one thread per 64-base block = 4 nibble words -> 2 words of tgt2; flags per 2^flag_shift bases
    uint64_t nblocks = nwords4 / 4;
    for (uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; blk < nblocks; blk += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t a = tgt4[4 * blk], b = tgt4[4 * blk + 1], c = tgt4[4 * blk + 2], d = tgt4[4 * blk + 3];
        tgt2[2 * blk] = ((uint64_t)squeeze2(a) << 32) | squeeze2(b);
        tgt2[2 * blk + 1] = ((uint64_t)squeeze2(c) << 32) | squeeze2(d);
        if ((a | b | c | d) & 0x4444444444444444ULL) {
            uint64_t g = blk >> (flag_shift - 6);
            atomicOr(&nflag32[g >> 5], 1u << (g & 31));
        }
    }
}

__global__ void k_split_sa5(const uint8_t *__restrict__ sa5, uint64_t n, uint32_t *__restrict__ lo, uint8_t *__restrict__ hi)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint8_t *p = sa5 + i * 5;
        lo[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        hi[i] = p[4];
    }
}

// bucket of a suffix = 2-bit code of its first k bases; a suffix that meets N/EOS after j < k bases
// sorts after every real k-mer sharing those j bases, i.e. in the bucket "prefix padded with T"
__device__ __forceinline__ uint64_t suffix_bucket(const uint64_t *__restrict__ tgt, uint64_t pos, int k)
{
    uint64_t w = nib16(tgt, pos);
    uint64_t bad = w & 0x4444444444444444ULL;           // N(4) and EOS(7) have bit 2 set
    if (bad) {
        int j = __clzll(bad) >> 2;                       // first offending nibble
        if (j < 16) w |= (~0ULL >> (4 * j)) & 0x3333333333333333ULL;   // pad with T from there on
    }
    return (uint64_t)(squeeze2(w) >> (32 - 2 * k));
}

template <bool WIDE, typename TabT>
__global__ void k_build_ktab(DevIndex ix, TabT *__restrict__ tab, int k)
{
    uint64_t n = ix.n;
    uint64_t ncodes = 1ULL << (2 * k);
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i <= n; i += stride) {
        // entries (prev, cur] receive i; prev = bucket(i-1) (or -1), cur = bucket(i) (or ncodes at i == n)
        uint64_t cur = i < n ? suffix_bucket(ix.tgt4, sa_get<WIDE>(ix, i), k) : ncodes;
        uint64_t from = i > 0 ? suffix_bucket(ix.tgt4, sa_get<WIDE>(ix, i - 1), k) + 1 : 0;
        for (uint64_t c = from; c <= cur; c++) tab[c] = (TabT)i;
    }
}

// ------------------------------------------------------------------------------------------------
// read preparation: k_pack_reads (one lane per packed 16-base word of the read or of its reverse
// complement: 16 byte loads, one 8-byte store) then k_init_reads (one lane per read: N policy of
// Aligner.cpp:9041-9063 from the packed words, default result record, first active list).
// Nibble written for a base byte v: v & 7 (quality / mask bits dropped); values 5..7 mark a byte
// the reference would refuse ((*pSeq = (*pSeqVal & 0x07)) > eBaseN).

struct __attribute__((packed)) Bytes16 { uint64_t lo, hi; };      // unaligned 16-byte load (one global_load_dwordx4)

// 8 base bytes, first base in the MOST significant byte -> 8 nibbles (bits 0..2 of each byte kept)
__device__ __forceinline__ uint64_t pack8_msb(uint64_t y)
{
    y &= 0x0707070707070707ULL;
    y = (y | (y >> 4)) & 0x00FF00FF00FF00FFULL;
    y = (y | (y >> 8)) & 0x0000FFFF0000FFFFULL;
    y = (y | (y >> 16)) & 0x00000000FFFFFFFFULL;
    return y;
}

__device__ __forceinline__ uint64_t complement8(uint64_t x)       // A<->T, C<->G on 3-bit codes, others unchanged
{
    x &= 0x0707070707070707ULL;
    return x ^ (((~x >> 2) & 0x0101010101010101ULL) * 3);
}

// ---- packed batches (bk_align_batch_packed): 16 bases per 32-bit word at 2 bit/base, first base in the top bits ----------------

__device__ __forceinline__ uint32_t rev2_32(uint32_t x)            // the 16 2-bit fields in reverse order
{
    x = __brev(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
__device__ __forceinline__ uint64_t rev2_64(uint64_t x)            // the 32 2-bit fields in reverse order
{
    x = __brevll(x);
    return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
}

// bases 16w .. 16w + 15 of a packed read (rc: of its reverse complement) as one 2-bit word, zero beyond the read's end.  W = the
// read's words; the word behind its last one may be loaded (the buffers are followed by one more word), its bits are never used.
__device__ __forceinline__ uint32_t packed_word16(const uint32_t *__restrict__ W, int len, int w, bool rc)
{
    const int rem = len - 16 * w;                     // bases of the read in this word
    if (rem <= 0) return 0;
    const uint32_t keep = rem >= 16 ? 0xFFFFFFFFu : ~0u << (32 - 2 * rem);
    if (!rc) return W[w] & keep;
    // reverse complement: its bases 16w .. are the complement of the forward bases p + 15 .. p, p = len - 16w - 16
    const int p = len - 16 * w - 16;
    uint32_t x;
    if (p >= 0) {
        const int i = p >> 4;
        const unsigned s = (unsigned)(p & 15) << 1;
        const uint64_t c = ((uint64_t)W[i] << 32) | W[i + 1];
        x = (uint32_t)(c >> (32 - s));
    } else
        x = W[0] >> (unsigned)(2 * (-p));             // the read's first 16 + p bases, at the low end
    return ~rev2_32(x) & keep;
}

// reverse complement of a read held as W words of 32 bases (first base in the top bits, zero beyond its end).  Straight-line code:
// the word shift is a cascade of selects, one per bit of the shift count (a version that copied t[] into place under
// `if (shift == k)` inside an unrolled loop came out of hipcc 7.2 reading registers it had never written - rows wrong only in
// the blocks that did not start on a freshly zeroed register file; tools/prep_check.hip)
template <int W>
__device__ __forceinline__ void revcomp2(const uint64_t (&f)[W], int len, uint64_t (&r)[W])
{
    uint64_t v[W + 1];
#pragma unroll
    for (int i = 0; i < W; i++) v[i] = ~rev2_64(f[W - 1 - i]);      // the whole row reversed: the read now ends flush with the row's end
    v[W] = 0;
    const int sh = 2 * (32 * W - len), q = sh >> 6;                 // shift left by q words and bsh bits: 0 <= q <= W
    const unsigned bsh = (unsigned)(sh & 63);
#pragma unroll
    for (int step = 1; step <= W; step <<= 1) {
        const bool on = (q & step) != 0;
        uint64_t nv[W + 1];
#pragma unroll
        for (int i = 0; i <= W; i++) nv[i] = on ? (i + step <= W ? v[i + step <= W ? i + step : W] : 0ULL) : v[i];
#pragma unroll
        for (int i = 0; i <= W; i++) v[i] = nv[i];
    }
#pragma unroll
    for (int i = 0; i < W; i++) r[i] = bsh ? ((v[i] << bsh) | (v[i + 1] >> (64 - bsh))) : v[i];
}

// exceptions of a packed batch, one lane each.  k_mark_exc counts them into the reads' meta words BEFORE the read preparation runs
// (bits 16..30: N bases, bit 31: a code the reference refuses - what the N policy needs to know); k_exc_rows gives the reads that
// have one their 4-bit rows (lean batches; widened from the 2-bit rows), k_apply_exc then writes the codes into the rows of both strands
__global__ void __launch_bounds__(256) k_mark_exc(DevBatch b)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.pk_nexc) return;
    const bk_nbase e = b.pk_exc[i];
    const uint32_t rr = e.read - b.pk_read0;
    if (rr >= b.n_reads) return;
    if (e.code == 4) atomicAdd(&b.rmeta[rr], ((uint32_t)e.run + 1u) << 16);
    else atomicOr(&b.rmeta[rr], 1u << 31);
}

__global__ void __launch_bounds__(256) k_exc_rows(DevBatch b)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.pk_nexc) return;
    const uint32_t rd = b.pk_exc[i].read;
    const uint32_t rr = rd - b.pk_read0;
    if (rr >= b.n_reads || (i > 0 && b.pk_exc[i - 1].read == rd)) return;       // the first exception of a read does the read
    for (uint32_t st = 0; st < 2; st++)
        for (uint32_t w = 0; w < b.wpr; w++) {
            uint64_t v = 0;
            if (w < b.nw) {
                const uint64_t x = b.rd2[((uint64_t)rr * 2 + st) * (b.nw / 2) + (w >> 1)];
                v = spread2to4((w & 1) ? (uint32_t)x : (uint32_t)(x >> 32));
            }
            b.rd4[((uint64_t)rr * 2 + st) * b.wpr + w] = v;
        }
}

__global__ void __launch_bounds__(256) k_apply_exc(DevBatch b)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.pk_nexc) return;
    const bk_nbase e = b.pk_exc[i];
    const uint32_t rr = e.read - b.pk_read0;
    if (rr >= b.n_reads) return;
    const int len = (int)b.lens[rr];
    // the run's bases [lo, hi] of each strand (codes 4..7 are their own complement, SeqTrans.cpp:458-512), one 16-base word at a time
    const int lo2[2] = {(int)e.pos, len - 1 - (int)e.pos - (int)e.run}, hi2[2] = {(int)e.pos + (int)e.run, len - 1 - (int)e.pos};
    const unsigned long long code16 = 0x1111111111111111ULL * (unsigned long long)(e.code & 7);
    for (int st = 0; st < 2; st++) {
        for (int w = lo2[st] >> 4; w <= (hi2[st] >> 4); w++) {
            const int a = lo2[st] > 16 * w ? lo2[st] - 16 * w : 0, z = hi2[st] < 16 * w + 15 ? hi2[st] - 16 * w : 15;      // nibbles a..z of the word
            const unsigned long long m = (~0ULL >> (4 * a)) & (~0ULL << (60 - 4 * z));
            unsigned long long *p = reinterpret_cast<unsigned long long *>(b.rd4 + ((uint64_t)rr * 2 + st) * b.wpr + w);
            atomicAnd(p, ~m);
            atomicOr(p, code16 & m);
        }
    }
}

// one-time check of a packed batch's exception list (whole batch): codes 4..7, reads and positions in range, strictly ascending
__global__ void __launch_bounds__(256) k_check_exc(const bk_nbase *__restrict__ exc, uint64_t n_exc, const uint32_t *__restrict__ lens, uint32_t n_reads,
                                                   uint32_t *__restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_exc) return;
    const bk_nbase e = exc[i];
    bool ok = e.read < n_reads && e.code >= 4 && e.code <= 7;
    if (ok) ok = (uint32_t)e.pos + e.run < lens[e.read];
    if (ok && i > 0) {
        const bk_nbase q = exc[i - 1];
        ok = q.read < e.read || (q.read == e.read && (uint32_t)q.pos + q.run < e.pos);
    }
    if (!ok) atomicAdd(bad, 1u);
}

// lens16 -> lens32 and the words each read takes (the input of the offset scan)
__global__ void __launch_bounds__(256) k_widen_lens(const uint16_t *__restrict__ lens16, uint32_t n, uint32_t *__restrict__ lens32,
                                                    unsigned long long *__restrict__ nwords)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = lens16[i];
    lens32[i] = l;
    nwords[i] = (l + 15) >> 4;
}

// packed batch: [0] max over reads of (first word + words taken), [1] longest read
__global__ void __launch_bounds__(256) k_packed_extent(const uint64_t *__restrict__ offs, const uint32_t *__restrict__ lens, uint32_t n,
                                                       unsigned long long *__restrict__ out)
{
    unsigned long long e = 0, l = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long len = lens[i], end = offs[i] + ((len + 15) >> 4);
        e = end > e ? end : e;
        l = len > l ? len : l;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long e2 = __shfl_down(e, off), l2 = __shfl_down(l, off);
        e = e2 > e ? e2 : e;
        l = l2 > l ? l2 : l;
    }
    if ((threadIdx.x & 63) == 0) {
        if (e) atomicMax(out + 0, e);
        if (l) atomicMax(out + 1, l);
    }
}

void launch_packed_extent(const uint64_t *offs, const uint32_t *lens, uint32_t n, unsigned long long *out, hipStream_t s)
{
    unsigned blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (n) hipLaunchKernelGGL(k_packed_extent, dim3(blocks), dim3(256), 0, s, offs, lens, n, out);
}

void launch_widen_lens(const uint16_t *lens16, uint32_t n, uint32_t *lens32, unsigned long long *nwords, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_widen_lens, dim3((n + 255) / 256), dim3(256), 0, s, lens16, n, lens32, nwords);
}

void launch_check_exc(const bk_nbase *exc, uint64_t n_exc, const uint32_t *lens, uint32_t n_reads, uint32_t *bad, hipStream_t s)
{
    if (n_exc) hipLaunchKernelGGL(k_check_exc, dim3((unsigned)((n_exc + 255) / 256)), dim3(256), 0, s, exc, n_exc, lens, n_reads, bad);
}

__global__ void __launch_bounds__(256) k_pack_reads(DevBatch b)
{
    // a block packs 256 / (2 * wpr) whole reads: 32-bit index arithmetic only
    const uint32_t wpr = b.wpr;
    const uint32_t per_read = 2 * wpr;                      // <= 256 (kMaxReadLenAbs)
    const uint32_t rpb = 256 / per_read;
    const uint32_t lr = threadIdx.x / per_read;
    if (lr >= rpb) return;
    const uint64_t r = (uint64_t)blockIdx.x * rpb + lr;
    if (r >= b.n_reads) return;
    const uint32_t rem = threadIdx.x - lr * per_read;
    const uint32_t st = rem >= wpr ? 1 : 0, w = rem - st * wpr;
    int len = (int)b.lens[r];
    if (b.pk_words != nullptr) {
        // packed batch: the word, or the 16 bases of the forward read whose reverse complement it is, straight from the 2-bit words
        // (bases that are not a,c,g,t read as whatever their field holds; k_apply_exc writes their codes afterwards)
        b.rd4[r * per_read + rem] = spread2to4(packed_word16(b.pk_words + b.offs[r], len, (int)w, st != 0));
        return;
    }
    const uint8_t *s = b.bases + b.offs[r];
    uint64_t v = 0;
    int base0 = 16 * (int)w;
    if (base0 + 16 <= len) {
        // a full word: the 16 source bytes lie inside the read, fetch them with one load
        if (st == 0) {
            Bytes16 q = *reinterpret_cast<const Bytes16 *>(s + base0);
            v = (pack8_msb(__builtin_bswap64(q.lo)) << 32) | pack8_msb(__builtin_bswap64(q.hi));
        } else {
            // reverse complement (SeqTrans.cpp:458-512): output base k = complement of s[len-1-base0-k]
            Bytes16 q = *reinterpret_cast<const Bytes16 *>(s + (len - 16 - base0));
            v = (pack8_msb(complement8(q.hi)) << 32) | pack8_msb(complement8(q.lo));
        }
    } else if (base0 < len) {
        int cnt = len - base0;
        if (st == 0) {
            for (int k = 0; k < cnt; k++) v |= (uint64_t)(s[base0 + k] & 7) << (60 - 4 * k);
        } else {
            for (int k = 0; k < cnt; k++) {
                uint8_t x = s[len - 1 - base0 - k] & 7;
                x = x < 4 ? (uint8_t)(3 - x) : x;
                v |= (uint64_t)x << (60 - 4 * k);
            }
        }
    }
    b.rd4[r * per_read + rem] = v;
}

// k_pack_reads + k_init_reads in one pass for reads of <= 16*NW bases (the register-kernel path): one lane per read builds the
// rows of both strands in registers - from 16-byte loads of the read's bytes, or (PACKED) from its 2-bit words, of which the forward
// row is a copy - applies the N policy, initialises the result record and appends the read to the first active list.  Lean batches
// (b.rd2 set) get 2 bit/base rows, and 4 bit/base rows only for the reads that hold an N: 60 bytes written per 100-base read
// instead of the 248 of full rows in both forms.
template <int NW, bool PACKED>
__global__ void __launch_bounds__(256) k_prep_fused(DevAlignCfg cfg, DevBatch b, StripeSet out)
{
    __shared__ uint32_t s_cnt, s_base, s_cmax;
    if (threadIdx.x == 0) { s_cnt = 0; s_cmax = 0; }
    __syncthreads();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool lean = b.rd2 != nullptr;
    bool go = false, has_n = false;
    uint32_t my_cmax = 0;
    int len = 0;
    uint64_t fw[PACKED ? 1 : NW], rv[PACKED ? 1 : NW];              // 4 bit/base rows (1 byte/base input)
    uint64_t f2[NW / 2], r2[NW / 2];                                // 2 bit/base rows
    bk_hit h;
    if (r < b.n_reads) {
        len = (int)b.lens[r];
        int num_ns = 0;
        bool bad = false;
        if (PACKED) {
            // (all NW words are loaded whatever the read's length - straight-line loads, no exec-masked ones in an unrolled loop: see
            // DESIGN.md on hipcc 7.2; the words buffer is followed by NW more words, and what lies behind the read is masked off)
            const uint32_t *__restrict__ W = b.pk_words + b.offs[r];
            uint32_t wv[NW];
#pragma unroll
            for (int k = 0; k < NW; k++) wv[k] = W[k];
#pragma unroll
            for (int k = 0; k < NW / 2; k++) {
                uint64_t v = ((uint64_t)wv[2 * k] << 32) | wv[2 * k + 1];
                const int rem = len - 32 * k;                       // bases of the read in this word
                if (rem < 32) v = rem <= 0 ? 0ULL : (v & (~0ULL << (64 - 2 * rem)));
                f2[k] = v;
            }
            revcomp2<NW / 2>(f2, len, r2);
            const uint32_t pre = b.rmeta[r];                        // k_mark_exc: what the exception list holds for this read
            num_ns = (int)((pre >> 16) & 0x7FFFu);
            bad = (pre >> 31) != 0;
        } else {
            // forward rows from the bytes; the reverse complement rows come from them by bit work (revcomp2) unless the read has an N
            // or the batch keeps 4-bit rows for every read - then the bytes are walked a second time from the other end
            const uint8_t *s = b.bases + b.offs[r];
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const int base0 = 16 * w;
                uint64_t f = 0;
                if (base0 + 16 <= len) {
                    Bytes16 q = *reinterpret_cast<const Bytes16 *>(s + base0);
                    f = (pack8_msb(__builtin_bswap64(q.lo)) << 32) | pack8_msb(__builtin_bswap64(q.hi));
                } else if (base0 < len) {
                    const int cnt = len - base0;
                    for (int k = 0; k < cnt; k++) f |= (uint64_t)(s[base0 + k] & 7) << (60 - 4 * k);
                }
                fw[w] = f;
                rv[w] = 0;
            }
#pragma unroll
            for (int w = 0; w < NW; w++) {
                if (16 * w < len) {
                    uint64_t x = fw[w] & top_mask(len - 16 * w);
                    uint64_t hi = x & 0x4444444444444444ULL;
                    uint64_t lo = (x | (x >> 1)) & 0x1111111111111111ULL;
                    bad |= ((hi >> 2) & lo) != 0;
                    num_ns += __popcll(hi);
                }
            }
#pragma unroll
            for (int k = 0; k < NW / 2; k++) f2[k] = ((uint64_t)squeeze2(fw[2 * k]) << 32) | squeeze2(fw[2 * k + 1]);
            if (lean && !bad && num_ns == 0) revcomp2<NW / 2>(f2, len, r2);
            else {
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    const int base0 = 16 * w;
                    uint64_t v = 0;
                    if (base0 + 16 <= len) {
                        Bytes16 p = *reinterpret_cast<const Bytes16 *>(s + (len - 16 - base0));
                        v = (pack8_msb(complement8(p.hi)) << 32) | pack8_msb(complement8(p.lo));
                    } else if (base0 < len) {
                        const int cnt = len - base0;
                        for (int k = 0; k < cnt; k++) {
                            uint8_t x = s[len - 1 - base0 - k] & 7;
                            x = x < 4 ? (uint8_t)(3 - x) : x;
                            v |= (uint64_t)x << (60 - 4 * k);
                        }
                    }
                    rv[w] = v;
                }
#pragma unroll
                for (int k = 0; k < NW / 2; k++) r2[k] = ((uint64_t)squeeze2(rv[2 * k]) << 32) | squeeze2(rv[2 * k + 1]);
            }
        }
        // N policy and result record, as k_init_reads
        h.chrom_id = 0; h.match_loci = 0; h.match_len = 0; h.low_hit_instances = 0; h.rslt = 0;
        h.nar = BK_NAR_NOHIT; h.strand = '?'; h.low_mm = 0; h.nxt_low_mm = 0; h.num_hits = 0; h.mismatches = 0; h.flags = 0;
        int max_ns_seq = 0;
        if (cfg.max_ns) {
            max_ns_seq = (len * cfg.max_ns) / 100;
            if (max_ns_seq < cfg.max_ns) max_ns_seq = cfg.max_ns;
        }
        if (bad || num_ns > max_ns_seq) h.nar = BK_NAR_NS;
        has_n = bad || num_ns > 0;
        if (h.nar != BK_NAR_NS) {
            ReadPlan p = make_plan(len, cfg);
            if (p.n_phases > 0) {
                int mm, cl, cd, ofs[1];
                phase_params(p, cfg, 0, mm, cl, cd);
                int nc = core_offsets(len, cl, cd, p.max_slides, ofs, 0);
                if (nc <= kMaxCoresFast) my_cmax = (uint32_t)nc;
                go = true;
            }
        }
    }
    // the list append comes before the rows are written: its barriers wait for every store the wave has issued
    const int lane = threadIdx.x & 63;
    uint64_t m = __ballot(go);
    uint32_t my_off = 0;
    for (int off = 32; off > 0; off >>= 1) { uint32_t q = __shfl_down(my_cmax, off); my_cmax = q > my_cmax ? q : my_cmax; }
    if (m) {
        uint32_t w = 0;
        if (lane == 0) { w = atomicAdd(&s_cnt, (uint32_t)__popcll(m)); if (my_cmax) atomicMax(&s_cmax, my_cmax); }
        w = __builtin_amdgcn_readfirstlane(w);
        my_off = w + (uint32_t)__popcll(m & ((1ULL << lane) - 1));
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) s_base = stripe_reserve(out, 0, s_cnt);
    if (threadIdx.x == 64 && s_cmax) stripe_max(out, s_cmax);
    __syncthreads();
    if (go) stripe_put(out, 0, s_base + my_off, r);
    if (r < b.n_reads) {
        b.rmeta[r] = (uint32_t)len | (has_n ? kReadHasN : 0u);
        // 4 bit/base rows [read][strand][wpr] (zero padded, 16-byte aligned): every read of a batch without 2-bit rows; in lean
        // batches only the reads with an N (1 byte/base input: here; packed input: k_exc_rows + k_apply_exc after this kernel)
        if (PACKED ? !lean : (!lean || has_n)) {
            const uint32_t wpr = b.wpr;
            uint4 *row0 = reinterpret_cast<uint4 *>(b.rd4 + (uint64_t)r * 2 * wpr);
            uint4 *row1 = reinterpret_cast<uint4 *>(b.rd4 + ((uint64_t)r * 2 + 1) * wpr);
#pragma unroll
            for (int q = 0; q < NW / 2; q++) {
                if (2 * q < (int)wpr) {
                    uint64_t a0, a1, c0, c1;
                    if (PACKED) {
                        a0 = spread2to4((uint32_t)(f2[q] >> 32)); a1 = spread2to4((uint32_t)f2[q]);
                        c0 = spread2to4((uint32_t)(r2[q] >> 32)); c1 = spread2to4((uint32_t)r2[q]);
                    } else { a0 = fw[PACKED ? 0 : 2 * q]; a1 = fw[PACKED ? 0 : 2 * q + 1]; c0 = rv[PACKED ? 0 : 2 * q]; c1 = rv[PACKED ? 0 : 2 * q + 1]; }
                    row0[q] = make_uint4((uint32_t)a0, (uint32_t)(a0 >> 32), (uint32_t)a1, (uint32_t)(a1 >> 32));
                    row1[q] = make_uint4((uint32_t)c0, (uint32_t)(c0 >> 32), (uint32_t)c1, (uint32_t)(c1 >> 32));
                }
            }
            for (uint32_t q = NW / 2; 2 * q < wpr; q++) { row0[q] = make_uint4(0, 0, 0, 0); row1[q] = make_uint4(0, 0, 0, 0); }
        }
        if (lean) {
            uint4 *t0 = reinterpret_cast<uint4 *>(b.rd2 + (uint64_t)r * 2 * (NW / 2));
            uint4 *t1 = reinterpret_cast<uint4 *>(b.rd2 + ((uint64_t)r * 2 + 1) * (NW / 2));
#pragma unroll
            for (int q = 0; q < NW / 4; q++) {
                t0[q] = make_uint4((uint32_t)f2[2 * q], (uint32_t)(f2[2 * q] >> 32), (uint32_t)f2[2 * q + 1], (uint32_t)(f2[2 * q + 1] >> 32));
                t1[q] = make_uint4((uint32_t)r2[2 * q], (uint32_t)(r2[2 * q] >> 32), (uint32_t)r2[2 * q + 1], (uint32_t)(r2[2 * q + 1] >> 32));
            }
        }
        b.out[r] = h;
    }
}

__global__ void __launch_bounds__(1024) k_init_reads(DevAlignCfg cfg, DevBatch b, uint32_t *__restrict__ act,
                                                      uint32_t *__restrict__ act_cnt, uint32_t *__restrict__ cmax)
{
    __shared__ uint32_t s_cnt, s_base, s_cmax;
    if (threadIdx.x == 0) { s_cnt = 0; s_cmax = 0; }
    __syncthreads();
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    bool go = false;
    uint32_t my_cmax = 0;
    if (r < b.n_reads) {
        int len = (int)b.lens[r];
        bk_hit h;
        h.chrom_id = 0; h.match_loci = 0; h.match_len = 0; h.low_hit_instances = 0; h.rslt = 0;
        h.nar = BK_NAR_NOHIT; h.strand = '?'; h.low_mm = 0; h.nxt_low_mm = 0; h.num_hits = 0; h.mismatches = 0; h.flags = 0;
        int max_ns_seq = 0;
        if (cfg.max_ns) {
            max_ns_seq = (len * cfg.max_ns) / 100;
            if (max_ns_seq < cfg.max_ns) max_ns_seq = cfg.max_ns;
        }
        const uint64_t *fw = b.rd4 + (uint64_t)r * 2 * b.wpr;
        int num_ns = 0;
        bool bad = false;
        for (int w = 0; 16 * w < len; w++) {
            uint64_t x = fw[w] & top_mask(len - 16 * w);
            uint64_t hi = x & 0x4444444444444444ULL;                    // values 4..7
            uint64_t lo = (x | (x >> 1)) & 0x1111111111111111ULL;       // low two bits non-zero
            bad |= ((hi >> 2) & lo) != 0;                               // 5,6,7: not a base the reference accepts
            num_ns += __popcll(hi);
        }
        if (bad || num_ns > max_ns_seq) h.nar = BK_NAR_NS;
        b.out[r] = h;
        b.rmeta[r] = (uint32_t)len | ((bad || num_ns > 0) ? kReadHasN : 0u);
        if (h.nar != BK_NAR_NS) {
            ReadPlan p = make_plan(len, cfg);
            if (p.n_phases > 0) {
                int mm, cl, cd, ofs[1];
                phase_params(p, cfg, 0, mm, cl, cd);
                int nc = core_offsets(len, cl, cd, p.max_slides, ofs, 0);
                if (nc <= kMaxCoresFast) my_cmax = (uint32_t)nc;
                go = true;
            }
        }
    }
    // one global append per block (see k_light)
    const int lane = threadIdx.x & 63;
    uint64_t m = __ballot(go);
    uint32_t my_off = 0;
    for (int off = 32; off > 0; off >>= 1) { uint32_t q = __shfl_down(my_cmax, off); my_cmax = q > my_cmax ? q : my_cmax; }
    if (m) {
        uint32_t w = 0;
        if (lane == 0) { w = atomicAdd(&s_cnt, (uint32_t)__popcll(m)); if (my_cmax) atomicMax(&s_cmax, my_cmax); }
        w = __builtin_amdgcn_readfirstlane(w);
        my_off = w + (uint32_t)__popcll(m & ((1ULL << lane) - 1));
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) s_base = atomicAdd(act_cnt, s_cnt);
    if (threadIdx.x == 64 && s_cmax) atomicMax(cmax, s_cmax);
    __syncthreads();
    if (go) act[s_base + my_off] = r;
}

// ------------------------------------------------------------------------------------------------
// K1: SA interval search, one lane per (active read, strand, core)

// With `lazy` set (register-window path), a core whose k-mer table bucket holds <= kLazyBucket suffixes
// is NOT bisected/verified here: the bucket is handed on as is (bit 31 of iv_n set) and the extend
// kernels keep only the members whose core bases are clean in the window they evaluate anyway -
// same candidates in the same SA order, two dependent HBM round trips fewer per probe.
constexpr uint32_t kLazyBucket = 4;
constexpr uint32_t kLazyFlag = 0x80000000u;

template <bool WIDE>
__global__ void __launch_bounds__(256) k_search(DevIndex ix, DevAlignCfg cfg, DevBatch b, const uint32_t *__restrict__ act,
                                                uint32_t n_act, int phase, int cmax, int nstr, int lazy)
{
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t per_read = (uint32_t)(nstr * cmax);
    uint64_t a = tid / per_read;
    if (a >= n_act) return;
    uint32_t rem = (uint32_t)(tid - a * per_read);
    int si = (int)(rem / (uint32_t)cmax), c = (int)(rem % (uint32_t)cmax);
    uint32_t r = act[a];
    const uint32_t meta = b.rmeta[r];
    int len = (int)(meta & kReadLenMask);
    ReadPlan p = make_plan(len, cfg);
    int mm, cl, cd;
    phase_params(p, cfg, phase, mm, cl, cd);
    // offset of core c by the sliding rule
    int cur = cd, o = 0, n = 0, my_ofs = -1;
    while (n < p.max_slides && o <= len - cl && cur > cl / 3) {
        if (o + cl + cur > len) cur = len - (o + cl);
        if (n == c) my_ofs = o;
        n++;
        o += cur;
    }
    if (my_ofs < 0 || n > kMaxCoresFast) return;
    int strand = cfg.align_strand == 2 ? 1 : si;
    const RdRow rdw = read_row(b, r, strand, (meta & kReadHasN) != 0);
    uint64_t first, count;
    uint64_t slot = iv_slot(b, (uint32_t)a, strand, c);
    if (lazy && ix.k > 0 && cl >= ix.k) {
        uint64_t p0 = rdw.nib16(my_ofs) & top_mask(cl);
        uint64_t lo, hi;
        core_range(ix, p0, cl, lo, hi);
        if (hi - lo <= kLazyBucket && !(lo == 0 && hi == ix.n)) {
            iv_put(b, slot, lo, (uint32_t)(hi - lo) | (hi > lo ? kLazyFlag : 0u));
            return;
        }
    }
    search_core<WIDE>(ix, rdw, my_ofs, cl, ~0ULL >> 1, first, count);     // exact run length
    iv_put(b, slot, first, count > 0x7FFFFFFFULL ? 0x7FFFFFFFu : (uint32_t)count);
}

// ------------------------------------------------------------------------------------------------
// K2/K3: candidate walk + Hamming extension + classification, one lane per active read

// ------------------------------------------------------------------------------------------------
// Two-pass search over the second-level key array (DevIndex::k2).
//   k2[i] = the kK2Bases = 15 bases that FOLLOW the first k bases of suffix sa[i], 2 bits each in the top 30 bits of a 32-bit
//           word, and a kind in the low two: 0 = all of them a,c,g,t; 1 = an N or a sequence end among them - the bases in front
//           of it are kept, everything behind is filled with ones, so that the key sorts where the suffix does (above every key
//           that continues the same bases with a,c,g,t); the word 0xFFFFFFFF = an N / sequence end already inside the first k
//           bases (such suffixes sit at the end of the k-mer bucket they sort into and compare above every N-free probe).
//           Inside one k-mer bucket k2 is non-decreasing, so bases k .. k+14 of a core are resolved by a bisection over
//           CONTIGUOUS 4-byte keys - one load per step instead of the dependent suffix-array-then-target pair, sixteen keys to a
//           cache line - and a bucket of <= 16 suffixes is settled from one or two lines.
//           A key of kind 1 can compare EQUAL to a probe that ends in t's where the suffix has its N (the fill): such keys lie at
//           the END of the run of equal keys (the suffix sorts above every true match), so only the upper bound can be off; it
//           is walked back over them with a look at the target itself.  Thousands of suffixes in a genome are of that kind.
//   pass A (lane per read/strand/core): k-mer table lookup; empty buckets and buckets of <= 16 keys are
//           finished here, everything else is appended to a work list.
//   pass B (lane per work item): bisection over k2, then - for cores longer than k+15 bases whose
//           sub-bucket is not handed on unverified - over suffix array + target from base k+15 on.
// The split keeps the lanes of pass B uniformly busy: in one combined kernel ~70 % of the lanes
// finished after the table lookup and idled while their wave's longest bisection ran.
// Work items: the slot index; its iv_first/iv_n entry carries (range start, size | kind << 30).

constexpr uint32_t kKindShift = 30;
constexpr uint32_t kKindK2 = 1;        // bisect k2 over [first, first+size)
constexpr uint32_t kKindDeep = 2;      // [first, first+size) shares k+15 bases with the core: resolve the rest
constexpr uint32_t kKindFull = 3;      // no usable k-mer bucket: full search
constexpr uint32_t kInlineBucket = 16;
constexpr int kK2Bases = 15;
constexpr uint32_t kK2Above = 0xFFFFFFFFu;

// -1 / 0 / +1: key (masked to the core's bases) vs probe; the all-ones key sorts above everything
__device__ __forceinline__ int k2_cmp(uint32_t key, uint32_t m, uint32_t q2)
{
    if (key == kK2Above) return 1;
    key &= m;
    return key < q2 ? -1 : (key > q2 ? 1 : 0);
}
__device__ __forceinline__ bool k2_nkind(uint32_t key) { return key != kK2Above && (key & 3u) == 1u; }

// mask of the first L = min(rem2, kK2Bases) bases of a key; rem2 = bases of the core beyond the k-mer table's k
__device__ __forceinline__ uint32_t k2_mask(int rem2)
{
    const int L = rem2 < kK2Bases ? rem2 : kK2Bases;
    return L <= 0 ? 0u : ~0u << (32 - 2 * L);
}

// the key of the suffix at pos
__device__ __forceinline__ uint32_t k2_make(const uint64_t *__restrict__ tgt4, uint64_t pos, int k)
{
    const uint64_t w0 = nib16(tgt4, pos);
    if (w0 & top_mask(k) & 0x4444444444444444ULL) return kK2Above;
    const uint64_t w1 = nib16(tgt4, pos + (uint64_t)k);
    const uint64_t bad = w1 & 0x4444444444444440ULL;                 // N / sequence end among the 15 bases
    const uint32_t code = squeeze2(w1) & ~3u;
    if (!bad) return code;
    const int j = __clzll((long long)bad) >> 2;                       // the first of them
    return ((code | (0xFFFFFFFFu >> (2 * j))) & ~3u) | 1u;
}

// as cmp_core, but only bases [start, cl) of the core are compared
template <typename Row>
__device__ __forceinline__ int cmp_core_from(const Row &rdw, int ofs, int cl, int start,
                                             const uint64_t *__restrict__ tgt, uint64_t pos)
{
    for (int i = start; i < cl; i += 16) {
        uint64_t m = top_mask(cl - i);
        uint64_t p = row_nib16(rdw, ofs + i) & m;
        uint64_t t = nib16(tgt, pos + i) & m;
        if (p != t) return p < t ? -1 : 1;
    }
    return 0;
}

template <bool WIDE>
__global__ void k_build_k2(DevIndex ix, uint32_t *__restrict__ k2)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ix.n; i += (uint64_t)gridDim.x * blockDim.x)
        k2[i] = k2_make(ix.tgt4, sa_get<WIDE>(ix, i), ix.k);
}

// the bisection needs k2 non-decreasing inside every k-mer bucket; count the places where it is not
template <bool WIDE>
__global__ void k_check_k2(DevIndex ix, const uint32_t *__restrict__ k2, unsigned long long *__restrict__ bad)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < ix.n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (k2[i] <= k2[i + 1]) continue;
        if (suffix_bucket(ix.tgt4, sa_get<WIDE>(ix, i), ix.k) == suffix_bucket(ix.tgt4, sa_get<WIDE>(ix, i + 1), ix.k))
            atomicAdd(bad, 1ULL);
    }
}

// Pass A with ILP searches per lane, written stage by stage so that the loads of a stage (read row, k-mer table, second-level
// keys) of all ILP searches are in flight together: item u of a lane is search number tid + u * (lanes of the grid), i.e. every u
// maps neighbouring lanes to neighbouring searches.  Same records and work list for every ILP (order aside).
// -DBK_PROF=1 (k_flat) / 2 (k_search_a_ilp): where a block's time goes - thread 0 adds the cycles between its section marks to
// g_prof (summed over the blocks, read with bk_debug_prof(); `BK_DIAG=1 python bench.py` prints them)
__device__ unsigned long long g_prof[64 * 16];
#ifdef BK_PROF
#define PROF_AT(k) do { if (threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const long long now_ = clock64(); pt[k] = now_ - last_; last_ = now_; } } while (0)
#define PROF_BEGIN long long pt[10] = {0,0,0,0,0,0,0,0,0,0}, last_ = clock64()
#define PROF_END do { if (threadIdx.x == 0) { for (int k_ = 0; k_ < 10; k_++) atomicAdd(&g_prof[(blockIdx.x & 63) * 16 + k_], (unsigned long long)pt[k_]); atomicAdd(&g_prof[(blockIdx.x & 63) * 16 + 10], 1ULL); } } while (0)
#else
#define PROF_AT(k) do { } while (0)
#define PROF_BEGIN do { } while (0)
#define PROF_END do { } while (0)
#endif
#if defined(BK_PROF) && BK_PROF == 1
#define PROF(k) PROF_AT(k)
#else
#define PROF(k) do { } while (0)
#endif
#if defined(BK_PROF) && BK_PROF == 2
#define PROFS(k) PROF_AT(k)
#else
#define PROFS(k) do { } while (0)
#endif
template <int ILP>
__global__ void __launch_bounds__(256) k_search_a_ilp(DevIndex ix, DevAlignCfg cfg, DevBatch b, const uint32_t *__restrict__ act,
                                                      uint32_t n_act, int phase, int cmax, int nstr, int lazy,
                                                      StripeSet out)
{
    __shared__ uint32_t s_cnt, s_base;
#if defined(BK_PROF) && BK_PROF == 2
    PROF_BEGIN;
#endif
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const uint32_t per_read = (uint32_t)(nstr * cmax);
    const uint64_t total = (uint64_t)n_act * per_read;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const int k = ix.k;
    bool on[ILP], push[ILP], have_code[ILP];
    uint64_t slot[ILP], p0[ILP], first[ILP], lo[ILP], hi[ILP];
    uint32_t nval[ILP], q2raw[ILP];            // q2raw: the 16 bases behind the k-mer's, 2 bits each
    int cl[ILP];
    // The core at offset 0 of phases 0, 1, 2 .. begins with the same k + 15 bases whenever it is that long, so the interval those bases
    // select is looked up once: phase 0 leaves it in iv32 (here, or pass B after its key bisection), the later phases' offset-0 lanes
    // take it from there instead of fetching a k-mer table line and a key line each (a quarter of the searches at C2).
    constexpr uint32_t kNoIv32 = 0xFFFFFFFFu;
    uint32_t cix[ILP];                  // entry of iv32 this lane reads (phase > 0) or writes (phase 0); kNoIv32 = neither
    uint2 cv[ILP];
    bool cached[ILP];
    uint32_t key0[ILP];
    // stage 1: the item, its read row
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + (uint64_t)u * stride;
        on[u] = false; push[u] = false; have_code[u] = false; slot[u] = 0; p0[u] = 0; q2raw[u] = 0; first[u] = 0; nval[u] = 0; cl[u] = 1; lo[u] = hi[u] = 0;
        cix[u] = kNoIv32; cv[u] = make_uint2(0, kNoIv32); cached[u] = false;
        if (tid < total) {
            const uint64_t a = tid / per_read;
            const uint32_t rem = (uint32_t)(tid - a * per_read);
            const int si = (int)(rem / (uint32_t)cmax), c = (int)(rem % (uint32_t)cmax);
            const uint32_t r = act[a];
            const uint32_t meta = b.rmeta[r];
            const int len = (int)(meta & kReadLenMask);
            const int strand_c = cfg.align_strand == 2 ? 1 : si;
            if (b.iv32 != nullptr && c == 0 && phase > 0) cv[u] = b.iv32[(uint32_t)strand_c * b.n_reads + r];     // (requested with the length)
            ReadPlan p = make_plan(len, cfg);
            int mm, cd, dummy[1];
            phase_params(p, cfg, phase, mm, cl[u], cd);
            const int nc = core_offsets(len, cl[u], cd, p.max_slides, dummy, 0);
            if (c < nc && nc <= kMaxCoresFast) {
                on[u] = true;
                if (b.iv32 != nullptr && c == 0 && cl[u] >= k + kK2Bases) {
                    cix[u] = (uint32_t)strand_c * b.n_reads + r;
                    cached[u] = phase > 0 && cv[u].y < (1u << kKindShift);
                }
                const int my_ofs = c * cd < len - cl[u] ? c * cd : len - cl[u];
                const int strand = cfg.align_strand == 2 ? 1 : si;
                slot[u] = iv_slot(b, (uint32_t)a, strand, c);
                if (b.rd2 != nullptr && !(meta & kReadHasN)) {
                    // 32 bases from the core's start out of the 2-bit row: the k-mer code's bases and the 16 that follow them
                    const uint64_t *row = b.rd2 + ((uint64_t)r * 2 + strand) * (b.nw / 2);
                    const uint64_t x = bits64_2(row, my_ofs);
                    p0[u] = spread2to4((uint32_t)(x >> 32)) & top_mask(cl[u]);
                    q2raw[u] = k == 16 ? (uint32_t)x : (uint32_t)(bits64_2(row, my_ofs + k) >> 32);
                } else {
                    const uint64_t *rdw = b.rd4 + ((uint64_t)r * 2 + strand) * b.wpr;
                    p0[u] = nib16(rdw, my_ofs) & top_mask(cl[u]);
                    const uint64_t q4 = nib16(rdw, my_ofs + k);
                    q2raw[u] = squeeze2(q4);
                    // an N among the core's bases behind the k-mer cannot be put to the 2-bit keys: the full search takes the core
                    const int rem2 = cl[u] - k;
                    if (rem2 > 0 && (q4 & 0x4444444444444444ULL & top_mask(rem2 < kK2Bases ? rem2 : kK2Bases))) p0[u] |= 0x4000000000000000ULL;
                }
            }
        }
    }
    PROFS(0);
    // stage 2: k-mer table
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        nval[u] = kKindFull << kKindShift;
        push[u] = on[u];
        have_code[u] = on[u] && !cached[u] && cl[u] >= k && !(p0[u] & 0x4444444444444444ULL & top_mask(k));
        key0[u] = kK2Above;
        if (have_code[u]) {
            const uint64_t code = (uint64_t)(squeeze2(p0[u]) >> (32 - 2 * k));
            if (ix.ktab2 != nullptr) {
                // {bucket start, second-level key of its first suffix}: a bucket of one - every second one a read of a unique region
                // meets, and a third of those its other strand runs into by chance - is settled by the line that names it
                const uint2 e0 = ix.ktab2[code], e1 = ix.ktab2[code + 1];
                lo[u] = e0.x; hi[u] = e1.x; key0[u] = e0.y;
            } else {
                lo[u] = ktab_get(ix, code);
                hi[u] = ktab_get(ix, code + 1);
            }
        }
    }
    PROFS(1);
    // stage 3: small buckets from the key array
    // (a lane loads the keys its bucket has - one to three for most - and no more: this kernel lives on the rate at which the
    // texture path takes lane requests, and sixteen keys for every lane cost a third more time than the search saved)
    uint32_t key[ILP][kInlineBucket];
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        const uint64_t size = hi[u] - lo[u];
#pragma unroll
        for (uint32_t j = 0; j < kInlineBucket; j++)
            key[u][j] = (have_code[u] && size <= kInlineBucket && j < size) ? ((j == 0 && ix.ktab2 != nullptr) ? key0[u] : ix.k2[lo[u] + j]) : kK2Above;
    }
    PROFS(2);
    // stage 4: results
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        uint2 leave = make_uint2(0, kNoIv32);               // what phase 0 leaves in iv32 for this read and strand
        if (cached[u]) {
            // the interval of the first k + 15 bases, as the bucket compare below would have produced it
            first[u] = cv[u].x;
            const uint32_t cnt = cv[u].y;
            if (cnt == 0 || cl[u] <= k + kK2Bases) { nval[u] = cnt; push[u] = false; }
            else if (lazy && cnt <= kLazyBucket) { nval[u] = cnt | kLazyFlag; push[u] = false; }
            else nval[u] = cnt | (kKindDeep << kKindShift);
        }
        if (have_code[u]) {
            const uint64_t size = hi[u] - lo[u];
            if (size == 0) { first[u] = lo[u]; nval[u] = 0; push[u] = false; leave = make_uint2((uint32_t)lo[u], 0u); }
            else if (size <= kInlineBucket) {
                const uint32_t m = k2_mask(cl[u] - k), q2 = q2raw[u] & m;
                uint32_t lb = 0, ub = 0;
                bool suspect = false;                          // a key of the N kind counted as equal: pass B has a look at the target
#pragma unroll
                for (uint32_t j = 0; j < kInlineBucket; j++) {
                    const int cm = k2_cmp(key[u][j], m, q2);
                    lb += cm < 0;
                    ub += cm <= 0;
                    suspect |= cm == 0 && k2_nkind(key[u][j]);
                }
                if (suspect) {
                    first[u] = lo[u];
                    nval[u] = (uint32_t)size | (kKindK2 << kKindShift);
                } else {
                    first[u] = lo[u] + lb;
                    const uint32_t cnt = ub - lb;
                    leave = make_uint2((uint32_t)first[u], cnt);
                    if (cnt == 0 || cl[u] <= k + kK2Bases) { nval[u] = cnt; push[u] = false; }
                    else if (lazy && cnt <= kLazyBucket) { nval[u] = cnt | kLazyFlag; push[u] = false; }
                    else nval[u] = cnt | (kKindDeep << kKindShift);
                }
            } else if (size < (1ULL << kKindShift)) {
                first[u] = lo[u];
                nval[u] = (uint32_t)size | (kKindK2 << kKindShift);        // (pass B leaves the interval in iv32 after its key bisection)
            }
        }
        if (phase == 0 && cix[u] != kNoIv32) b.iv32[cix[u]] = leave;
    }
    PROFS(3);
    // work-list appends, one global atomic per block.  The interval records are stored after them: the barriers of the append
    // wait for every store the wave has issued.
    const int lane = threadIdx.x & 63;
    uint32_t my_off[ILP];
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        my_off[u] = 0;
        const uint64_t m = __ballot(push[u]);
        if (m) {
            uint32_t w = 0;
            if (lane == 0) w = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
            w = __builtin_amdgcn_readfirstlane(w);
            my_off[u] = w + (uint32_t)__popcll(m & ((1ULL << lane) - 1));
        }
    }
    PROFS(4);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) s_base = stripe_reserve(out, 0, s_cnt);
    __syncthreads();
    PROFS(5);
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        if (push[u]) stripe_put(out, 0, s_base + my_off[u], (uint32_t)slot[u]);
        if (on[u] && nval[u] != 0) iv_put(b, slot[u], first[u], nval[u]);
    }
#if defined(BK_PROF) && BK_PROF == 2
    PROFS(6);
    PROF_END;
#endif
}

template <bool WIDE>
__global__ void __launch_bounds__(256) k_search_b(DevIndex ix, DevAlignCfg cfg, DevBatch b, int phase, int lazy,
                                                  const uint32_t *__restrict__ list, uint32_t n_list)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_list) return;
#ifdef BK_DIAG_B
    unsigned long long d_k2 = 0, d_deep = 0;
    struct Fin { unsigned long long &a, &b; DevBatch &bb; __device__ ~Fin() { if (a) atomicAdd(&bb.ctr[ctr_stripe() + 5], a); if (b) atomicAdd(&bb.ctr[ctr_stripe() + 6], b); } } fin{d_k2, d_deep, b};
#endif
    const uint64_t slot = list[i];
    const uint32_t r = b.act[(uint32_t)(slot % b.iv_stride)], sc = (uint32_t)(slot / b.iv_stride);
    const int strand = (int)(sc / b.iv_cores), c = (int)(sc % b.iv_cores);
    const uint32_t meta = b.rmeta[r];
    const int len = (int)(meta & kReadLenMask);
    ReadPlan p = make_plan(len, cfg);
    int mm, cl, cd;
    phase_params(p, cfg, phase, mm, cl, cd);
    const int my_ofs = c * cd < len - cl ? c * cd : len - cl;
    const RdRow rdw = read_row(b, r, strand, (meta & kReadHasN) != 0);
    uint64_t first;
    uint32_t raw;
    iv_get(b, slot, first, raw);
    const uint32_t kind = raw >> kKindShift;
    uint64_t cnt = raw & ((1u << kKindShift) - 1);
    const int k = ix.k;
    if (kind == kKindFull) {
        search_core<WIDE>(ix, rdw, my_ofs, cl, ~0ULL >> 1, first, cnt);
        iv_put(b, slot, first, cnt > 0x7FFFFFFFULL ? 0x7FFFFFFFu : (uint32_t)cnt);
        return;
    }
    if (kind == kKindK2) {
        const uint32_t m = k2_mask(cl - k);
        const uint32_t q2 = squeeze2(rdw.nib16(my_ofs + k)) & m;
        // lower and upper bound in lock step: two independent loads per round
        uint64_t l1 = first, h1 = first + cnt, l2 = first, h2 = first + cnt;
        while (l1 < h1 || l2 < h2) {
            const bool a1 = l1 < h1, a2 = l2 < h2;
            const uint64_t m1 = l1 + ((h1 - l1) >> 1), m2 = l2 + ((h2 - l2) >> 1);
#ifdef BK_DIAG_B
            d_k2 += 1 + (1ULL << 32) * ((a1 && (h1 - l1) > 8) + (a2 && (h2 - l2) > 8 && (m1 >> 3) != (m2 >> 3)));
#endif
            const uint32_t v1 = a1 ? ix.k2[m1] : 0, v2 = a2 ? ix.k2[m2] : 0;
            if (a1) { if (k2_cmp(v1, m, q2) < 0) l1 = m1 + 1; else h1 = m1; }
            if (a2) { if (k2_cmp(v2, m, q2) <= 0) l2 = m2 + 1; else h2 = m2; }
        }
        // keys of the N kind at the end of the run of equal keys may be there for their fill only: the target decides
        {
            const int upto = cl < k + kK2Bases ? cl : k + kK2Bases;
            while (l2 > l1) {
                const uint32_t kv = ix.k2[l2 - 1];
                if (!k2_nkind(kv) || cmp_core_from(rdw, my_ofs, upto, k, ix.tgt4, sa_get<WIDE>(ix, l2 - 1)) == 0) break;
                l2--;
            }
        }
        first = l1;
        cnt = l2 - l1;
        if (!WIDE && b.iv32 != nullptr && phase == 0 && c == 0 && cl >= k + kK2Bases)       // see k_search_a_ilp
            b.iv32[(uint32_t)strand * b.n_reads + r] = make_uint2((uint32_t)first, (uint32_t)cnt);
        if (cnt == 0 || cl <= k + kK2Bases) {
            iv_put(b, slot, first, cnt > 0x7FFFFFFFULL ? 0x7FFFFFFFu : (uint32_t)cnt);
            return;
        }
    }
    // [first, first+cnt) agrees with the core on its first k+15 bases
    if (lazy && cnt <= kLazyBucket) {
        iv_put(b, slot, first, (uint32_t)cnt | kLazyFlag);
        return;
    }
    {
        const int start = k + kK2Bases;
        uint64_t l1 = first, h1 = first + cnt, l2 = first, h2 = first + cnt;
        while (l1 < h1 || l2 < h2) {
            const bool a1 = l1 < h1, a2 = l2 < h2;
            const uint64_t m1 = l1 + ((h1 - l1) >> 1), m2 = l2 + ((h2 - l2) >> 1);
#ifdef BK_DIAG_B
            d_deep += 1 + (1ULL << 32) * (a1 + (a2 && m1 != m2));
#endif
            const uint64_t s1 = a1 ? sa_get<WIDE>(ix, m1) : 0, s2 = a2 ? sa_get<WIDE>(ix, m2) : 0;
            const int c1 = a1 ? cmp_core_from(rdw, my_ofs, cl, start, ix.tgt4, s1) : 0;
            const int c2 = a2 ? cmp_core_from(rdw, my_ofs, cl, start, ix.tgt4, s2) : 0;
            if (a1) { if (c1 > 0) l1 = m1 + 1; else h1 = m1; }
            if (a2) { if (c2 >= 0) l2 = m2 + 1; else h2 = m2; }
        }
        first = l1;
        cnt = l2 - l1;
    }
    iv_put(b, slot, first, cnt > 0x7FFFFFFFULL ? 0x7FFFFFFFu : (uint32_t)cnt);
}

template <bool WIDE>
__global__ void __launch_bounds__(256) k_extend(DevIndex ix, DevAlignCfg cfg, DevBatch b, const uint32_t *__restrict__ act,
                                                uint32_t n_act, int phase, uint32_t *__restrict__ next_act,
                                                uint32_t *__restrict__ next_cnt, uint32_t *__restrict__ heavy,
                                                uint32_t *__restrict__ heavy_cnt, uint32_t *__restrict__ cmax_next)
{
    __shared__ LdsEntries s_le;
    lds_entries_load(s_le, ix);
    uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_search = 0, n_cand = 0, n_lcm = 0;
    if (a < n_act) {
        uint32_t r = act[a];
        int len = (int)b.lens[r];
        ReadPlan p = make_plan(len, cfg);
        int mm, cl, cd, ofs[kMaxCoresFast];
        phase_params(p, cfg, phase, mm, cl, cd);
        int nc = core_offsets(len, cl, cd, p.max_slides, ofs, kMaxCoresFast);
        int s0 = cfg.align_strand == 2 ? 1 : 0, s1 = cfg.align_strand == 1 ? 0 : 1;
        bool is_heavy = nc > kMaxCoresFast;
        if (!is_heavy)
            for (int st = s0; st <= s1; st++)
                for (int c = 0; c < nc; c++)
                    if (iv_count(b, iv_slot(b, a, st, c)) > (uint32_t)cfg.heavy_thresh) is_heavy = true;
        if (is_heavy) {
            heavy[atomicAdd(heavy_cnt, 1u)] = r;
        } else {
            n_lcm = 1;
            const int init = mm + cfg.mm_delta + 1;
            int low_inst = 0, low_mm = init, nxt = init;
            uint64_t hit_left = 0;
            int hit_ent = -1, hit_strand = '?';
            bool done = false;
            for (int st = s0; st <= s1 && !done; st++) {
                const uint64_t *rdw = b.rd4 + ((uint64_t)r * 2 + st) * b.wpr;
                for (int c = 0; c < nc && !done; c++) {
                    n_search++;
                    uint64_t slot = iv_slot(b, a, st, c);
                    uint32_t n;
                    uint64_t first;
                    iv_get(b, slot, first, n);
                    for (uint32_t j = 0; j < n; j++) {
                        uint64_t loci = sa_get<WIDE>(ix, first + j);
                        if (loci < (uint64_t)ofs[c]) continue;
                        uint64_t t = loci - (uint64_t)ofs[c];
                        int e = find_entry_lds(s_le, ix, t);
                        if (e < 0 || t + (uint64_t)len - 1 > ix.ent_end[e]) continue;
                        // already processed through an earlier core of this strand pass?  (no core
                        // interval is truncated here, so "processed" == "that core matches at t")
                        bool dup = false;
                        for (int c2 = 0; c2 < c && !dup; c2++) {
                            uint64_t q0 = nib16(rdw, ofs[c2]) & top_mask(cl);
                            dup = cmp_core(rdw, ofs[c2], cl, q0, ix.tgt4, t + (uint64_t)ofs[c2]) == 0;
                        }
                        if (dup) continue;
                        n_cand++;
                        int lim = mm < nxt - 1 ? mm : nxt - 1;
                        int cm = hamming(rdw, len, ix.tgt4, t, lim);
                        if (cm > lim) continue;
                        if (cm < low_mm) {
                            low_inst = 1; nxt = low_mm; low_mm = cm;
                            hit_left = t; hit_ent = e; hit_strand = st ? '-' : '+';
                        } else if (cm == low_mm)
                            low_inst++;
                        else
                            nxt = cm;
                        if (low_inst > cfg.max_hits && low_mm == 0) { done = true; break; }
                    }
                }
            }
            int rslt = classify(low_inst, low_mm, nxt, init, cfg.mm_delta, cfg.max_hits);
            if (rslt != BK_HR_NONE)
                write_result(ix, cfg, b, r, len, rslt, low_inst, low_mm, nxt, hit_left, hit_ent, hit_strand, phase << 1);
            else if (phase + 1 < p.n_phases) {
                int mm2, cl2, cd2, dummy[1];
                phase_params(p, cfg, phase + 1, mm2, cl2, cd2);
                int nc2 = core_offsets(len, cl2, cd2, p.max_slides, dummy, 0);
                if (nc2 <= kMaxCoresFast) atomicMax(cmax_next, (uint32_t)nc2);
                next_act[atomicAdd(next_cnt, 1u)] = r;
            }
        }
    }
    // counters: wave reduce, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        n_search += __shfl_down(n_search, off);
        n_cand += __shfl_down(n_cand, off);
        n_lcm += __shfl_down(n_lcm, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (n_search) atomicAdd(&b.ctr[ctr_stripe() + 0], n_search);
        if (n_cand) atomicAdd(&b.ctr[ctr_stripe() + 1], n_cand);
        if (n_lcm) atomicAdd(&b.ctr[ctr_stripe() + 2], n_lcm);
    }
}

// ------------------------------------------------------------------------------------------------
// Register-resident candidate evaluation for reads of <= 16*NW bases (NW = 8 or 16 sixteen-base words).
// One pass over the target window gives everything the reference's loop derives per candidate:
//   * EOS inside the window  <=> the read would cross an entry boundary (MapChunkHit2Entry bounds test)
//   * a bit-per-base mismatch map: Hamming distance = popcount, "core c' matches exactly here" = its
//     bit range is clear (which is what the reference's dedupe set encodes, see k_wave)
// The nw+1 window words are independent loads, all in flight together.

template <int NW>
struct Window {
    uint64_t bm[NW / 4];        // mismatch bit map, bit b of bm[b/64] = base b differs
    int mm;
    bool eos;
};


template <int NW>
__device__ __forceinline__ void eval_window(const uint64_t (&rw)[NW], int len, const uint64_t *__restrict__ tgt, uint64_t t,
                                            Window<NW> &w)
{
    // window fetched with 16-byte loads: ceil((len/16 + 2) / 2) instructions instead of len/16 + 1
    const uint64_t i0 = t >> 4;
    const unsigned s = (unsigned)(t & 15) << 2;
    const bool odd = (i0 & 1) != 0;
    const uint4 *__restrict__ blk = reinterpret_cast<const uint4 *>(tgt) + (i0 >> 1);
    constexpr int NB = NW / 2 + 1;
    uint64_t r2[2 * NB];
    // words touched from i0 on: the window's bases plus the shift spill; 16-byte blocks from the even word
    // at or below i0.  Only what is needed is fetched (a 100-base window needs the 5th block in 3 of 32
    // alignments) - these kernels are bound by 32-byte sectors moved, see DESIGN.md
    const int nwords = ((int)(t & 15) + len + 15) >> 4;
    const int nblk = ((odd ? 1 : 0) + nwords + 1) >> 1;
#pragma unroll
    for (int q = 0; q < NB; q++) {
        // block q holds words 2q, 2q+1 counted from the even word at or below i0
        if (q < nblk) {
            uint4 v = blk[q];
            r2[2 * q] = ((uint64_t)v.y << 32) | v.x;
            r2[2 * q + 1] = ((uint64_t)v.w << 32) | v.z;
        } else {
            r2[2 * q] = 0;
            r2[2 * q + 1] = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < NW / 4; k++) w.bm[k] = 0;
    uint64_t eosacc = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        if (16 * k < len) {
            uint64_t a = odd ? r2[k + 1] : r2[k];
            uint64_t b = odd ? r2[k + 2] : r2[k + 1];
            uint64_t win = (a << s) | ((b >> 1) >> (63 - s));
            uint64_t m = top_mask(len - 16 * k);
            uint64_t x = (rw[k] ^ win) & m;
            uint64_t f = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ULL;
            eosacc |= win & (win >> 1) & (win >> 2) & m & 0x1111111111111111ULL;      // nibble 7 = EOS
            w.bm[k >> 2] |= (uint64_t)flags_to_bits16(f) << (16 * (k & 3));
        }
    }
    int mm = 0;
#pragma unroll
    for (int k = 0; k < NW / 4; k++) mm += __popcll(w.bm[k]);
    w.mm = mm;
    w.eos = eosacc != 0;
}

// eval_window for the rare window that the 2-bit compare cannot decide (N or a sequence end nearby): the same result, one 16-base
// word of read and target at a time from memory, so that the path costs the kernels that carry it a few registers instead of the
// NW + NW/2 + 2 words the all-at-once form holds
template <int NW>
__device__ __forceinline__ void eval_window_rare(const RdRow &rdrow, int len, const uint64_t *__restrict__ tgt, uint64_t t,
                                              Window<NW> &w)
{
#pragma unroll
    for (int k = 0; k < NW / 4; k++) w.bm[k] = 0;
    uint64_t eosacc = 0;
    const int nk = (len + 15) >> 4;
#pragma unroll 1
    for (int k = 0; k < nk; k++) {
        const uint64_t win = nib16(tgt, t + 16 * (uint64_t)k);
        const uint64_t m = top_mask(len - 16 * k);
        const uint64_t x = (rdrow.word16(k) ^ win) & m;
        const uint64_t f = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ULL;
        eosacc |= win & (win >> 1) & (win >> 2) & m & 0x1111111111111111ULL;
        const uint64_t bits = (uint64_t)flags_to_bits16(f) << (16 * (k & 3));
#pragma unroll
        for (int q = 0; q < NW / 4; q++) w.bm[q] |= (k >> 2) == q ? bits : 0ULL;
    }
    int mm = 0;
#pragma unroll
    for (int k = 0; k < NW / 4; k++) mm += __popcll(w.bm[k]);
    w.mm = mm;
    w.eos = eosacc != 0;
}

// true when bases [o, o+cl) of the read all match the window (cl >= 1)
template <int NW>
__device__ __forceinline__ bool core_clean(const Window<NW> &w, int o, int cl)
{
    const int hi = o + cl;
    bool dirty = false;
#pragma unroll
    for (int k = 0; k < NW / 4; k++) {
        int a = o > 64 * k ? o - 64 * k : 0;
        int b = hi < 64 * k + 64 ? hi - 64 * k : 64;
        if (a < b) {
            uint64_t m = (b >= 64 ? ~0ULL : ((1ULL << b) - 1)) & ~((1ULL << a) - 1);
            dirty |= (w.bm[k] & m) != 0;
        }
    }
    return !dirty;
}

__device__ __forceinline__ uint64_t uniform64(uint64_t v)      // value known to be wave-uniform -> scalar registers
{
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

template <int NW>
__device__ __forceinline__ void load_read_words(const uint64_t *__restrict__ rdw, int len, uint64_t (&rw)[NW])
{
    // rd4 rows are 16-byte aligned (wpr is even): 16-byte loads
    const uint4 *__restrict__ p = reinterpret_cast<const uint4 *>(rdw);
#pragma unroll
    for (int q = 0; q < NW / 2; q++) {
        if (32 * q < len) {
            uint4 v = p[q];
            rw[2 * q] = ((uint64_t)v.y << 32) | v.x;
            rw[2 * q + 1] = ((uint64_t)v.w << 32) | v.z;
        } else {
            rw[2 * q] = 0;
            rw[2 * q + 1] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2-bit window compare.  Read side: per read and strand NW/2 words at 2 bit/base (N held as A) plus NW/4
// words of "this read base is N" in the mismatch-map format; target side: DevIndex::tgt2.  Only for
// windows whose 64-base blocks hold no N/EOS (window_flagged), where a read N always is a mismatch and
// nothing else can differ from the 4-bit compare.

__device__ __forceinline__ bool window_flagged(const DevIndex &ix, uint64_t t, int len)
{
    const uint64_t g0 = t >> ix.flag_shift, g1 = (t + (uint64_t)len - 1) >> ix.flag_shift;     // a window spans <= 2 regions
    return (((ix.nflag[g0 >> 3] >> (g0 & 7)) | (ix.nflag[g1 >> 3] >> (g1 & 7))) & 1) != 0;
}

// The window's 16-byte blocks are requested by window2_load and consumed by window2_compare: a kernel that has several
// candidates per lane (k_flat) issues the loads of all of them before it touches the first result.
template <int NW>
__device__ __forceinline__ void window2_load(const uint64_t *__restrict__ tgt2, const uint64_t *__restrict__ tgt2s, uint64_t t, int len,
                                             uint4 (&v)[NW / 4 + 1])
{
    const uint64_t i0 = t >> 5;
    const bool odd = (i0 & 1) != 0;
    // tgt2s (optional) is the same data stored again 32 bytes later: a window that would straddle a 64-byte
    // line in one copy lies inside a line of the other (16-byte block index 2 or 3 within the line -> 0 or 1)
    const uint64_t blk0 = i0 >> 1;
    const uint4 *__restrict__ blk = (tgt2s != nullptr && (blk0 & 2)) ? reinterpret_cast<const uint4 *>(tgt2s) + (blk0 - 2)
                                                                      : reinterpret_cast<const uint4 *>(tgt2) + blk0;
    const int nwords = ((int)(t & 31) + len + 31) >> 5;
    const int nblk = ((odd ? 1 : 0) + nwords + 1) >> 1;
#pragma unroll
    for (int q = 0; q < NW / 4 + 1; q++) v[q] = q < nblk ? blk[q] : make_uint4(0, 0, 0, 0);
}

template <int NW>
__device__ __forceinline__ void window2_compare(const uint64_t (&r2w)[NW / 2], const uint64_t (&rnm)[NW / 4], int len, uint64_t t,
                                                const uint4 (&v)[NW / 4 + 1], Window<NW> &w)
{
    const unsigned s = (unsigned)(t & 31) << 1;
    const bool odd = ((t >> 5) & 1) != 0;
    constexpr int NB = NW / 4 + 1;
    uint64_t r[2 * NB];
#pragma unroll
    for (int q = 0; q < NB; q++) {
        r[2 * q] = ((uint64_t)v[q].y << 32) | v[q].x;
        r[2 * q + 1] = ((uint64_t)v[q].w << 32) | v[q].z;
    }
#pragma unroll
    for (int k = 0; k < NW / 4; k++) w.bm[k] = rnm[k];
#pragma unroll
    for (int k = 0; k < NW / 2; k++) {
        if (32 * k < len) {
            uint64_t a = odd ? r[k + 1] : r[k];
            uint64_t b = odd ? r[k + 2] : r[k + 1];
            uint64_t win = (a << s) | ((b >> 1) >> (63 - s));
            uint64_t x = r2w[k] ^ win;
            uint64_t y = (x | (x >> 1)) & 0x5555555555555555ULL;        // base j of the word: bit 62 - 2j
            const int rem = len - 32 * k;
            if (rem < 32) y &= ~0ULL << (64 - 2 * rem);
            uint64_t g = __brevll(y) >> 1;                               // base j: bit 2j
            g = (g | (g >> 1)) & 0x3333333333333333ULL;
            g = (g | (g >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
            g = (g | (g >> 4)) & 0x00FF00FF00FF00FFULL;
            g = (g | (g >> 8)) & 0x0000FFFF0000FFFFULL;
            g = (g | (g >> 16)) & 0x00000000FFFFFFFFULL;
            w.bm[k >> 1] |= g << (32 * (k & 1));
        }
    }
    int mm = 0;
#pragma unroll
    for (int k = 0; k < NW / 4; k++) mm += __popcll(w.bm[k]);
    w.mm = mm;
    w.eos = false;
}

template <int NW>
__device__ __forceinline__ void eval_window2(const uint64_t (&r2w)[NW / 2], const uint64_t (&rnm)[NW / 4], int len,
                                             const uint64_t *__restrict__ tgt2, const uint64_t *__restrict__ tgt2s, uint64_t t,
                                             Window<NW> &w)
{
    uint4 v[NW / 4 + 1];
    window2_load<NW>(tgt2, tgt2s, t, len, v);
    window2_compare<NW>(r2w, rnm, len, t, v, w);
}

template <int NW>
__device__ __forceinline__ void load_read_words2(const uint64_t *__restrict__ row, uint64_t (&r2w)[NW / 2])
{
    // rows are NW/2 words = a multiple of 16 bytes
    const uint4 *__restrict__ p = reinterpret_cast<const uint4 *>(row);
#pragma unroll
    for (int q = 0; q < NW / 4; q++) {
        uint4 u = p[q];
        r2w[2 * q] = ((uint64_t)u.y << 32) | u.x;
        r2w[2 * q + 1] = ((uint64_t)u.w << 32) | u.z;
    }
}

// ------------------------------------------------------------------------------------------------
// The wave kernels' form of the 2-bit compare.  k_wave spends two thirds of its issue slots on vector ALU work, and half of
// eval_window2 is the squeeze of the pair-per-base difference into one bit per base, which only exists so that core_clean can
// build its masks in base units.  Here the map stays where the compare leaves it: word i covers bases 64i .. 64i + 63, base
// 64i + j (j < 32) at bit 62 - 2j and base 64i + 32 + j at bit 63 - 2j (two 32-base compare words laid into each other, one
// shift-or).  The Hamming distance still is the popcount, and "core c matches exactly here" is an AND with a mask of the same
// layout that the wave computes once per read (the core geometry is the same for every candidate) and keeps in LDS.

template <int NW>
struct IWindow {
    uint64_t im[NW / 4];
    int mm;
    bool eos;
};

__device__ __forceinline__ uint64_t spread32(uint32_t v)       // bit p -> bit 2p
{
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x << 2)) & 0x3333333333333333ULL;
    x = (x | (x << 1)) & 0x5555555555555555ULL;
    return x;
}

// one bit per base (bit b = base b of the 64, Window::bm / the read rows' N words) -> the laid-together form
__device__ __forceinline__ uint64_t bits_to_imap(uint64_t bm)
{
    return spread32(__brev((uint32_t)bm)) | (spread32(__brev((uint32_t)(bm >> 32))) << 1);
}

__device__ __forceinline__ uint64_t pair_mask32(int a, int b)   // bases [a, b) of one 32-base compare word, clamped to it
{
    a = a < 0 ? 0 : a;
    b = b > 32 ? 32 : b;
    return a < b ? ((~0ULL >> (2 * a)) & (~0ULL << (64 - 2 * b)) & 0x5555555555555555ULL) : 0ULL;
}

// word i of the mask selecting bases [o, h) in an IWindow map
__device__ __forceinline__ uint64_t imask_word(int o, int h, int i)
{
    return pair_mask32(o - 64 * i, h - 64 * i) | (pair_mask32(o - 64 * i - 32, h - 64 * i - 32) << 1);
}

template <int NW>
__device__ __forceinline__ bool im_clean(const uint64_t (&im)[NW / 4], const uint64_t *__restrict__ mask)
{
    uint64_t a = 0;
#pragma unroll
    for (int i = 0; i < NW / 4; i++) a |= im[i] & mask[i];
    return a == 0;
}

template <int NW>
__device__ __forceinline__ void window_to_iwindow(const Window<NW> &w4, IWindow<NW> &w)
{
#pragma unroll
    for (int i = 0; i < NW / 4; i++) w.im[i] = bits_to_imap(w4.bm[i]);
    w.mm = w4.mm;
    w.eos = w4.eos;
}

template <bool WIDE>
__device__ __forceinline__ bool window_flagged_t(const DevIndex &ix, uint64_t t, int len)
{
    if (WIDE) return window_flagged(ix, t, len);
    // 4-byte indexes: the same test in 32-bit arithmetic
    const uint32_t t0 = (uint32_t)t;
    uint32_t t1 = t0 + (uint32_t)(len - 1);
    t1 = t1 < t0 ? 0xFFFFFFFFu : t1;
    const uint32_t g0 = t0 >> ix.flag_shift, g1 = t1 >> ix.flag_shift;
    return (((ix.nflag[g0 >> 3] >> (g0 & 7)) | (ix.nflag[g1 >> 3] >> (g1 & 7))) & 1) != 0;
}

template <int NW, bool WIDE>
__device__ __forceinline__ void eval_window2i(const uint64_t (&r2w)[NW / 2], const uint64_t (&rni)[NW / 4], int len,
                                              const uint64_t *__restrict__ tgt2, const uint64_t *__restrict__ tgt2s, uint64_t t,
                                              IWindow<NW> &w)
{
    // the loads of eval_window2 (4-byte indexes: block numbers fit 32 bits)
    const unsigned s = (unsigned)(t & 31) << 1;
    const bool odd = ((t >> 5) & 1) != 0;
    const uint4 *__restrict__ blk;
    if (WIDE) {
        const uint64_t blk0 = t >> 6;
        blk = (tgt2s != nullptr && (blk0 & 2)) ? reinterpret_cast<const uint4 *>(tgt2s) + (blk0 - 2) : reinterpret_cast<const uint4 *>(tgt2) + blk0;
    } else {
        const uint32_t blk0 = (uint32_t)t >> 6;
        blk = (tgt2s != nullptr && (blk0 & 2)) ? reinterpret_cast<const uint4 *>(tgt2s) + (blk0 - 2) : reinterpret_cast<const uint4 *>(tgt2) + blk0;
    }
    constexpr int NB = NW / 4 + 1;
    uint64_t r[2 * NB];
    const int nwords = ((int)(t & 31) + len + 31) >> 5;
    const int nblk = ((odd ? 1 : 0) + nwords + 1) >> 1;
#pragma unroll
    for (int q = 0; q < NB; q++) {
        if (q < nblk) {
            uint4 v = blk[q];
            r[2 * q] = ((uint64_t)v.y << 32) | v.x;
            r[2 * q + 1] = ((uint64_t)v.w << 32) | v.z;
        } else {
            r[2 * q] = 0;
            r[2 * q + 1] = 0;
        }
    }
    int mm = 0;
    uint64_t even = 0;
#pragma unroll
    for (int k = 0; k < NW / 2; k++) {
        uint64_t y = 0;
        if (32 * k < len) {
            uint64_t a = odd ? r[k + 1] : r[k];
            uint64_t b = odd ? r[k + 2] : r[k + 1];
            uint64_t win = (a << s) | ((b >> 1) >> (63 - s));
            uint64_t x = r2w[k] ^ win;
            y = (x | (x >> 1)) & 0x5555555555555555ULL;                  // base j of the word: bit 62 - 2j
            const int rem = len - 32 * k;
            if (rem < 32) y &= ~0ULL << (64 - 2 * rem);
        }
        if (k & 1) {
            // only even bits are set in y: the shift does not carry between the halves
            const uint32_t lo = ((uint32_t)y << 1) | (uint32_t)even, hi = ((uint32_t)(y >> 32) << 1) | (uint32_t)(even >> 32);
            const uint64_t m = (((uint64_t)hi << 32) | lo) | rni[k >> 1];
            w.im[k >> 1] = m;
            mm += __popcll(m);
        } else
            even = y;
    }
    w.mm = mm;
    w.eos = false;
}

// eval_window2i with the window taken from the candidate's entry of the suffix-ordered window array (DevIndex::swin): the three
// 16-byte words of entry `e`; the window starts bofs = kSwPre - (core offset) bases into it - the same for every lane of the wave.
// the compare of eval_swin2i once the five words the window starts in are known
template <int NW>
__device__ __forceinline__ void swin2i_compare(const uint64_t (&r2w)[NW / 2], const uint64_t (&rni)[NW / 4], int len, const uint64_t (&q)[5], unsigned s,
                                               IWindow<NW> &w)
{
    int mm = 0;
    uint64_t even = 0;
#pragma unroll
    for (int k = 0; k < NW / 2; k++) {
        uint64_t y = 0;
        if (k < 4 && 32 * k < len) {
            const uint64_t win = (q[k < 4 ? k : 0] << s) | ((q[k < 4 ? k + 1 : 0] >> 1) >> (63 - s));
            const uint64_t x = r2w[k] ^ win;
            y = (x | (x >> 1)) & 0x5555555555555555ULL;
            const int rem = len - 32 * k;
            if (rem < 32) y &= ~0ULL << (64 - 2 * rem);
        }
        if (k & 1) {
            const uint32_t lo = ((uint32_t)y << 1) | (uint32_t)even, hi = ((uint32_t)(y >> 32) << 1) | (uint32_t)(even >> 32);
            const uint64_t m = (((uint64_t)hi << 32) | lo) | rni[k >> 1];
            w.im[k >> 1] = m;
            mm += __popcll(m);
        } else
            even = y;
    }
    w.mm = mm;
    w.eos = false;
}

// The same compare in 16-base dwords for a window start that is the same in every lane: window dword w is one funnel shift
// (v_alignbit_b32) of entry dwords D2 + w and D2 + w + 1, whose registers are known at compile time once the scalar unit has branched
// on D2; five vector instructions per 16 bases and six per 64-base map word instead of the 64-bit shifts, selects and splits above.
template <int NW, int D2>
__device__ __forceinline__ void swin2i_compare32(const uint64_t (&r2w)[NW / 2], const uint64_t (&rni)[NW / 4], int len, const uint32_t (&S)[13], unsigned sh,
                                                 IWindow<NW> &w)
{
    static_assert(D2 >= -1 && D2 <= kSwPre / 16, "the window starts inside the entry's lead");
    int mm = 0;
#pragma unroll
    for (int i = 0; i < NW / 4; i++) {
        uint32_t y[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int wd = 4 * i + d;
            y[d] = 0;
            // (dwords the entry does not hold are never asked for: the caller only comes here with the whole window inside the entry)
            if (D2 + wd + 1 <= 12 && 16 * wd < len) {
                const uint32_t hi = D2 + wd >= 0 ? S[D2 + wd >= 0 ? D2 + wd : 0] : 0u;
                const uint32_t lo = S[D2 + wd + 1 <= 12 ? D2 + wd + 1 : 12];
                const uint32_t win = __builtin_amdgcn_alignbit(hi, lo, sh);
                const uint32_t rw = (wd & 1) ? (uint32_t)r2w[wd >> 1] : (uint32_t)(r2w[wd >> 1] >> 32);
                const uint32_t x = rw ^ win;
                uint32_t yy = (x | (x >> 1)) & 0x55555555u;                 // base j of the dword: bit 30 - 2j
                const int rem = len - 16 * wd;
                if (rem < 16) yy &= ~0u << (32 - 2 * rem);
                y[d] = yy;
            }
        }
        // the map word of these 64 bases: base j < 32 at bit 62 - 2j, base 32 + j at bit 63 - 2j
        const uint32_t mhi = y[0] | (y[2] << 1), mlo = y[1] | (y[3] << 1);
        const uint64_t m = (((uint64_t)mhi << 32) | mlo) | rni[i];
        w.im[i] = m;
        mm += __popcll(m);
    }
    w.mm = mm;
    w.eos = false;
}

// UNIFORM: bofs is the same for every lane of the wave (one core per round) - the choice of the starting word is then a branch the scalar
// unit takes instead of ten selects per lane
template <int NW, bool UNIFORM>
__device__ __forceinline__ void eval_swin2i(const uint64_t (&r2w)[NW / 2], const uint64_t (&rni)[NW / 4], int len, const uint4 (&e)[3],
                                            int bofs, IWindow<NW> &w)
{
    uint64_t r[6], q[5];                                       // (reads of up to kSwLen <= 128 bases: four compare words at most)
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const uint4 v = e[i];
        r[2 * i] = ((uint64_t)v.y << 32) | v.x;
        r[2 * i + 1] = ((uint64_t)v.w << 32) | v.z;
    }
    if (UNIFORM) {
        // the entry as thirteen 16-base dwords in base order (the 64-bit words hold their first base in the top bits)
        const uint32_t S[13] = {e[0].y, e[0].x, e[0].w, e[0].z, e[1].y, e[1].x, e[1].w, e[1].z, e[2].y, e[2].x, e[2].w, e[2].z, 0u};
        const int ub = __builtin_amdgcn_readfirstlane(bofs);
        const int rb = (ub & 15) << 1;
        const unsigned sh = (unsigned)(32 - rb) & 31u;
        const int d2 = (ub >> 4) - (rb ? 0 : 1);                  // -1 .. 5 (bofs <= kSwPre)
        switch (d2) {
        case -1: swin2i_compare32<NW, -1>(r2w, rni, len, S, sh, w); break;
        case 0: swin2i_compare32<NW, 0>(r2w, rni, len, S, sh, w); break;
        case 1: swin2i_compare32<NW, 1>(r2w, rni, len, S, sh, w); break;
        case 2: swin2i_compare32<NW, 2>(r2w, rni, len, S, sh, w); break;
        case 3: swin2i_compare32<NW, 3>(r2w, rni, len, S, sh, w); break;
        case 4: swin2i_compare32<NW, 4>(r2w, rni, len, S, sh, w); break;
        default: swin2i_compare32<NW, 5>(r2w, rni, len, S, sh, w); break;
        }
        return;
    }
    const int w0 = bofs >> 5;                                  // 0..2
    const unsigned s = (unsigned)(bofs & 31) << 1;
#pragma unroll
    for (int i = 0; i < 5; i++) q[i] = w0 == 0 ? r[i] : (w0 == 1 ? r[i + 1] : (i + 2 < 6 ? r[i + 2 < 6 ? i + 2 : 5] : 0ULL));
    swin2i_compare<NW>(r2w, rni, len, q, s, w);
}

// entry i of the suffix-ordered window array: kSwBases bases of the 2-bit target from sa[i] - kSwPre on (bases before the target's
// start read as 0: no window that uses them passes the "candidate starts before the read does" test)
__global__ void __launch_bounds__(256) k_build_swin(DevIndex ix, uint4 *__restrict__ swin)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ix.n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t base0 = (int64_t)ix.sa_lo[i] - kSwPre;
        uint64_t wd[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int64_t pos = base0 + 32 * k;
            uint64_t v;
            if (pos >= 0) {
                const uint64_t wi = (uint64_t)pos >> 5;
                const unsigned sh = (unsigned)(pos & 31) << 1;
                const uint64_t a = ix.tgt2[wi], bq = ix.tgt2[wi + 1];
                v = (a << sh) | ((bq >> 1) >> (63 - sh));
            } else if (pos > -32)
                v = ix.tgt2[0] >> (unsigned)(2 * (-pos));
            else
                v = 0;
            wd[k] = v;
        }
#pragma unroll
        for (int q = 0; q < 3; q++)
            swin[i * 3 + q] = make_uint4((uint32_t)wd[2 * q], (uint32_t)(wd[2 * q] >> 32), (uint32_t)wd[2 * q + 1], (uint32_t)(wd[2 * q + 1] >> 32));
    }
}

void launch_build_swin(const DevIndex &ix, void *swin, hipStream_t s)
{
    hipLaunchKernelGGL(k_build_swin, dim3(65536), dim3(256), 0, s, ix, reinterpret_cast<uint4 *>(swin));
}

// ------------------------------------------------------------------------------------------------
// k_light: one lane per read, reads of <= 16*NW bases whose core intervals are all <= heavy_thresh
// long.  Same contract as k_extend (which stays for longer reads); differences: the window is
// evaluated once in registers, bounds come from the EOS test instead of the entry table, and calls
// it cannot take go to the wave kernel (`wave`) or to the general kernel (`heavy`).

template <bool WIDE, int NW>
__global__ void __launch_bounds__(256, NW <= 8 ? 4 : 2) k_light(DevIndex ix, DevAlignCfg cfg, DevBatch b, const uint32_t *__restrict__ act,
                                               uint32_t n_act, int phase, uint32_t *__restrict__ next_act,
                                               uint32_t *__restrict__ next_cnt, uint32_t *__restrict__ heavy,
                                               uint32_t *__restrict__ heavy_cnt, uint32_t *__restrict__ wave,
                                               uint32_t *__restrict__ wave_cnt, uint32_t *__restrict__ cmax_next)
{
    // list appends, the next phase's core maximum and the counters are combined per block in LDS:
    // one global atomic per block and list instead of one per wave (same-address returning atomics
    // retire at only ~170 M/s on this part)
    __shared__ uint32_t s_cnt[4], s_base[4], s_cmax;
    __shared__ unsigned long long s_ctr[3];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 4) s_cmax = 0;
    if (threadIdx.x >= 8 && threadIdx.x < 11) s_ctr[threadIdx.x - 8] = 0;
    __syncthreads();
    uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_search = 0, n_cand = 0, n_lcm = 0;
    int dest = 0;                   // 1 = next phase, 2 = wave kernel, 3 = general kernel
    uint32_t r = 0, my_cmax = 0;
    if (a < n_act) {
        r = act[a];
        int len = (int)b.lens[r];
        ReadPlan p = make_plan(len, cfg);
        int mm, cl, cd, ofs[kMaxCoresFast];
        phase_params(p, cfg, phase, mm, cl, cd);
        int nc = core_offsets(len, cl, cd, p.max_slides, ofs, kMaxCoresFast);
        int s0 = cfg.align_strand == 2 ? 1 : 0, s1 = cfg.align_strand == 1 ? 0 : 1;
        bool fits = nc <= kMaxCoresFast && len <= 16 * NW;
        bool is_heavy = !fits;
        if (fits)
            for (int st = s0; st <= s1; st++)
                for (int c = 0; c < nc; c++)
                    if ((iv_count(b, iv_slot(b, a, st, c)) & ~kLazyFlag) > (uint32_t)cfg.heavy_thresh) is_heavy = true;
        if (is_heavy) {
            dest = (fits && wave != nullptr) ? 2 : 3;
        } else {
            n_lcm = 1;
            const int init = mm + cfg.mm_delta + 1;
            int low_inst = 0, low_mm = init, nxt = init;
            uint64_t hit_left = 0;
            int hit_strand = '?';
            bool done = false;
            for (int st = s0; st <= s1 && !done; st++) {
                uint64_t rw[NW];
                load_read_words<NW>(b.rd4 + ((uint64_t)r * 2 + st) * b.wpr, len, rw);
                for (int c = 0; c < nc && !done; c++) {
                    n_search++;
                    uint64_t slot = iv_slot(b, a, st, c);
                    uint32_t nraw;
                    uint64_t first;
                    iv_get(b, slot, first, nraw);
                    const uint32_t n = nraw & ~kLazyFlag;
                    const bool lazy = (nraw & kLazyFlag) != 0;
                    for (uint32_t j = 0; j < n; j++) {
                        uint64_t loci = sa_get<WIDE>(ix, first + j);
                        if (loci < (uint64_t)ofs[c]) continue;
                        uint64_t t = loci - (uint64_t)ofs[c];
                        Window<NW> w;
                        eval_window<NW>(rw, len, ix.tgt4, t, w);
                        if (lazy && !core_clean<NW>(w, ofs[c], cl)) continue;    // bucket member that is not a match of this core
                        if (w.eos) continue;                                    // crosses an entry boundary
                        bool dup = false;                                       // reached through an earlier core already?
                        for (int c2 = 0; c2 < c; c2++) dup |= core_clean<NW>(w, ofs[c2], cl);
                        if (dup) continue;
                        n_cand++;
                        int cm = w.mm;
                        if (cm > mm || cm >= nxt) continue;
                        if (cm < low_mm) {
                            low_inst = 1; nxt = low_mm; low_mm = cm;
                            hit_left = t; hit_strand = st ? '-' : '+';
                        } else if (cm == low_mm)
                            low_inst++;
                        else
                            nxt = cm;
                        if (low_inst > cfg.max_hits && low_mm == 0) { done = true; break; }
                    }
                }
            }
            int rslt = classify(low_inst, low_mm, nxt, init, cfg.mm_delta, cfg.max_hits);
            if (rslt != BK_HR_NONE) {
                int e = low_inst >= 1 ? find_entry(ix, hit_left) : -1;
                write_result(ix, cfg, b, r, len, rslt, low_inst, low_mm, nxt, hit_left, e, hit_strand, phase << 1);
            } else if (phase + 1 < p.n_phases) {
                int mm2, cl2, cd2, dummy[1];
                phase_params(p, cfg, phase + 1, mm2, cl2, cd2);
                int nc2 = core_offsets(len, cl2, cd2, p.max_slides, dummy, 0);
                if (nc2 <= kMaxCoresFast) my_cmax = (uint32_t)nc2;
                dest = 1;
            }
        }
    }
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = (1ULL << lane) - 1;
    for (int off = 32; off > 0; off >>= 1) {
        n_search += __shfl_down(n_search, off);
        n_cand += __shfl_down(n_cand, off);
        n_lcm += __shfl_down(n_lcm, off);
        uint32_t m = __shfl_down(my_cmax, off);
        my_cmax = m > my_cmax ? m : my_cmax;
    }
    uint32_t my_off = 0;
#pragma unroll
    for (int d = 1; d <= 3; d++) {
        uint64_t m = __ballot(dest == d);
        if (m) {
            uint32_t w = 0;
            if (lane == 0) w = atomicAdd(&s_cnt[d], (uint32_t)__popcll(m));
            w = __builtin_amdgcn_readfirstlane(w);
            if (dest == d) my_off = w + (uint32_t)__popcll(m & lt_mask);
        }
    }
    if (lane == 0) {
        if (my_cmax) atomicMax(&s_cmax, my_cmax);
        if (n_search) atomicAdd(&s_ctr[0], n_search);
        if (n_cand) atomicAdd(&s_ctr[1], n_cand);
        if (n_lcm) atomicAdd(&s_ctr[2], n_lcm);
    }
    __syncthreads();
    {
        const uint32_t t = threadIdx.x;
        if (t == 1 && s_cnt[1]) s_base[1] = atomicAdd(next_cnt, s_cnt[1]);
        if (t == 2 && s_cnt[2]) s_base[2] = atomicAdd(wave_cnt, s_cnt[2]);
        if (t == 3 && s_cnt[3]) s_base[3] = atomicAdd(heavy_cnt, s_cnt[3]);
        if (t == 4 && s_cmax) atomicMax(cmax_next, s_cmax);
        if (t >= 8 && t < 11 && s_ctr[t - 8]) atomicAdd(&b.ctr[ctr_stripe() + t - 8], s_ctr[t - 8]);
    }
    __syncthreads();
    if (dest == 1) next_act[s_base[1] + my_off] = r;
    else if (dest == 2) wave[s_base[2] + my_off] = a;          // (the wave kernel finds the read's interval records by its position)
    else if (dest == 3) heavy[s_base[3] + my_off] = r;
}

// ------------------------------------------------------------------------------------------------
// k_flat: the same contract as k_light, organised so that every lane does the same amount of work.
// A block owns 256 consecutive active reads.  Their candidates (every suffix of every core interval,
// in the reference's walk order strand -> core -> suffix) are numbered consecutively and EVALUATED
// one per lane - suffix array load, window compare, one result byte in LDS (mismatch count, or
// "skip": off the read's start / unverified bucket member that does not match / crosses an entry
// boundary / already reached through an earlier core).  The Low/NxtLow/instances outcome of a read is
// then reduced over its candidates' lanes (the state machine is order-independent up to its early exit,
// whose reads are replayed in order by their own lane; on 5-byte indexes the reference's truncated-key
// rule - a candidate is taken for seen when an earlier one of the strand pass has the same low word - is
// applied first, as a pass over the candidates' lanes).
// In k_light a lane walked all candidates of its read itself, so a wave ran as long as its read with
// the most candidates (up to 4 x 64) while the typical read has one or two.
// Valid while no interval is longer than 100: then the reference's IterCnt==100 copy-count check and
// MaxIter cannot trigger, every interval is walked to its end, and "already reached through an
// earlier core" is exactly "that earlier core matches here" (see k_wave for the general case).

constexpr uint32_t kLdsEntries = 128;     // entry tables up to this size are searched in LDS
constexpr uint32_t kFlatCap = 8192;        // result bytes held in LDS per pass over a block's reads
constexpr uint8_t kRecSkip = 255;

template <bool WIDE> struct FlatEntT { typedef uint64_t type; };
template <> struct FlatEntT<false> { typedef uint32_t type; };
__host__ __device__ constexpr bool flat_caches_first(bool wide, int bs, int slots_max)
{
    return !wide && slots_max <= 16 && bs * slots_max * 6 <= 24576;
}

// per-read rows of the 2-bit read copy are staged in LDS (fetched once per block, with the lengths and the interval records)
__host__ __device__ constexpr bool flat_rows_in_lds(bool wide, int nw, int bs) { return !wide && nw <= 8 && bs <= 256; }

template <bool WIDE, int NW, int BS>
__global__ void __launch_bounds__(BS) k_flat(DevIndex ix, DevAlignCfg cfg, DevBatch b, const uint32_t *__restrict__ act,
                                              uint32_t n_act, int phase, int slots_max, StripeSet out, int have_wave)
{
    // A block's time is a chain of dependent memory round trips (its four waves per SIMD do not hide them), so the kernel is laid
    // out to keep that chain short: everything that depends on the read number only - length, interval records, the read's 2-bit
    // rows - is requested together; the suffix array elements of up to KB candidates per lane are requested together, then their
    // windows, and only then the first compare runs.
    constexpr bool ROWS = flat_rows_in_lds(WIDE, NW, BS);
#ifdef BK_FLAT_KB
    constexpr int KB = BK_FLAT_KB;
#else
    constexpr int KB = 1;                                   // candidates a lane has in flight (2 .. 4 measured: the registers cost more occupancy than the overlap buys)
#endif
    constexpr int NBLK = NW / 4 + 1;
    constexpr uint32_t CAP = (ROWS ? kFlatCap / 4 : kFlatCap) * BS / 256;        // (LDS: four blocks per CU must fit 160 KB)
    constexpr int SPEC = 8;                                 // interval records requested before the length is known
    extern __shared__ uint32_t s_dyn[];
    // 4-byte indexes: the interval starts (and the "unverified bucket" bits) the counting pass has loaded anyway stay in LDS, so that
    // the evaluation's chain of dependent loads is suffix array element -> window instead of record -> element -> window
    const bool cf = flat_caches_first(WIDE, BS, slots_max);
    uint32_t *s_first = s_dyn;                          // [BS][slots_max] (when cf)
    uint16_t *s_sp = reinterpret_cast<uint16_t *>(s_dyn + (cf ? BS * slots_max : 0));   // [BS][slots_max] running candidate count after each slot
    __shared__ uint32_t s_lazy[BS];                     // bit q: slot q is an unverified bucket (when cf)
    __shared__ uint32_t s_off[BS + 1];                     // first candidate number of each read of the block
    __shared__ uint32_t s_r[BS];
    __shared__ uint32_t s_geo[BS];                         // read length | core length << 10 | core step << 20
    __shared__ uint8_t s_hasn[BS];
    // the per-read outcome is reduced over the candidates' lanes (4-byte indexes): smallest (mismatches << 16 | candidate number) of
    // the acceptable candidates, how many share that mismatch count (low half) and how many were looked at (high half), and the
    // smallest count above it.  One lane per read walking its own bytes made a wave wait for its read with the most candidates.
    __shared__ uint32_t s_k1[BS], s_c2[BS], s_nx[BS];
    __shared__ uint8_t s_mm[BS];                         // bit st: the read's strand-st row holds an N (the 4-bit compare decides its windows)
    __shared__ uint8_t s_rec[CAP];
    __shared__ uint4 s_row[ROWS ? BS * 2 * (NW / 4) : 1];  // [read][strand]: NW/2 words at 2 bit/base
    // 5-byte indexes: the reference's set of seen targets is keyed by the target start truncated to 32 bits (SfxArrayV2.cpp:5932), so a
    // candidate whose start lies a multiple of 2^32 bases from an earlier candidate of the same strand pass is taken for seen and
    // skipped.  The low words travel with the result bytes and the replay applies exactly that rule.
    __shared__ uint32_t s_key[WIDE ? CAP : 1];
    __shared__ uint32_t s_wsum[BS / 64];
    __shared__ uint32_t s_cnt[4], s_base[4], s_cmax;
    __shared__ unsigned long long s_ctr[3];
    using EntT = typename FlatEntT<WIDE>::type;
    __shared__ EntT s_es[kLdsEntries], s_ee[kLdsEntries];              // entry table, when it is small enough
    const uint32_t t = threadIdx.x;
#if defined(BK_PROF) && BK_PROF == 1
    PROF_BEGIN;
#endif
    const int lane = t & 63, wid = t >> 6;
    if (t < 4) s_cnt[t] = 0;
    if (t == 4) s_cmax = 0;
    if (t >= 8 && t < 11) s_ctr[t - 8] = 0;
    const bool ent_lds = ix.n_ent <= kLdsEntries;
    uint64_t es_v = 0, ee_v = 0;                            // stored after the counting pass: nothing here waits for them
    if (ent_lds && t < ix.n_ent) { es_v = ix.ent_start[t]; ee_v = ix.ent_end[t]; }

    const uint32_t a = blockIdx.x * blockDim.x + t;
    const int s0 = cfg.align_strand == 2 ? 1 : 0, s1 = cfg.align_strand == 1 ? 0 : 1;
    // slot q of a read = (strand pass q / cmaxs, core q % cmaxs): the numbering does not depend on the read's own core count,
    // cores it does not have are empty slots
    const int cmaxs = slots_max / (s1 - s0 + 1);
    const bool two_bit = b.rd2 != nullptr;
    uint32_t n_search = 0, n_cand = 0, n_lcm = 0;
    int dest = 0;                   // 1 = next phase, 2 = wave kernel, 3 = general kernel
    uint32_t r = 0, my_cmax = 0, my_total = 0;
    int len = 0, mm = 0, cl = 1, cd = 1, nc = 0, n_phases = 0;
    bool mine = false;              // this lane's read is resolved here
    if (a < n_act) {
        r = act[a];
        const uint32_t len_v = b.rmeta[r];
        const bool spec = !WIDE && slots_max <= SPEC;
        uint2 sv[SPEC];
        if (!WIDE) {
#pragma unroll
            for (int u = 0; u < SPEC; u++) {
                sv[u] = make_uint2(0, 0);
                if (spec && u < slots_max) {
                    const int sti = u >= cmaxs ? 1 : 0;
                    sv[u] = b.iv2[iv_slot(b, a, s0 + sti, u - sti * cmaxs)];
                }
            }
        }
        // both strands' rows (ROWS implies NW == 8): two 16-byte blocks of bases each, one 64-byte line per read.  Named values, not
        // an array: the compiler kept an array of them in scratch memory
        uint4 rb00 = make_uint4(0, 0, 0, 0), rb01 = rb00, rb10 = rb00, rb11 = rb00;
        if (ROWS && two_bit) {
            const uint4 *__restrict__ rp = reinterpret_cast<const uint4 *>(b.rd2 + (uint64_t)r * 2 * (NW / 2));
            rb00 = rp[0]; rb01 = rp[1]; rb10 = rp[2]; rb11 = rp[3];
        }
        len = (int)(len_v & kReadLenMask);
        ReadPlan p = make_plan(len, cfg);
        n_phases = p.n_phases;
        int dummy[1];
        phase_params(p, cfg, phase, mm, cl, cd);
        nc = core_offsets(len, cl, cd, p.max_slides, dummy, 0);
        const bool fits = nc <= kMaxCoresFast && len <= 16 * NW && nc <= cmaxs;
        bool is_heavy = !fits;
        if (fits) {
            uint32_t run = 0, lazy_bits = 0, work = 0;          // work: every candidate of the read (the wave kernel's job size, should it go there)
            if (!WIDE && spec) {
#pragma unroll
                for (int u = 0; u < SPEC; u++)
                    if (u < slots_max) {
                        const int c = u >= cmaxs ? u - cmaxs : u;
                        const uint32_t raw = c < nc ? sv[u].y : 0u;          // a core the read does not have: whatever the slot held
                        if (raw & kLazyFlag) lazy_bits |= 1u << u;
                        const uint32_t cnt = raw & ~kLazyFlag;
                        if (cnt > (uint32_t)cfg.heavy_thresh) is_heavy = true;
                        run += is_heavy ? 0 : cnt;
                        work = work + cnt < work ? 0xFFFFFFFFu : work + cnt;
                        if (cf) s_first[t * slots_max + u] = sv[u].x;
                        s_sp[t * slots_max + u] = (uint16_t)run;
                    }
            } else
                for (int q = 0; q < slots_max; q++) {
                    const int sti = q >= cmaxs ? 1 : 0, c = q - sti * cmaxs;
                    uint64_t f64 = 0;
                    uint32_t cnt = 0;
                    if (c < nc) iv_get(b, iv_slot(b, a, s0 + sti, c), f64, cnt);
                    if (cnt & kLazyFlag) lazy_bits |= 1u << (q & 31);
                    cnt &= ~kLazyFlag;
                    if (cnt > (uint32_t)cfg.heavy_thresh) is_heavy = true;
                    run += is_heavy ? 0 : cnt;
                    work = work + cnt < work ? 0xFFFFFFFFu : work + cnt;
                    if (cf) s_first[t * slots_max + q] = (uint32_t)f64;
                    s_sp[t * slots_max + q] = (uint16_t)run;
                }
            s_lazy[t] = lazy_bits;
            if (run > CAP) is_heavy = true;                 // more candidates than one pass's result bytes hold (many cores, all near heavy_thresh): the wave kernel's
            my_total = is_heavy ? 0 : run;
            if (is_heavy && have_wave && b.wave_work != nullptr) b.wave_work[a] = work;
        }
        if (is_heavy) dest = (fits && have_wave) ? 2 : 3;
        else { mine = true; n_lcm = 1; }
        if (ROWS && two_bit) { s_row[t * 4 + 0] = rb00; s_row[t * 4 + 1] = rb01; s_row[t * 4 + 2] = rb10; s_row[t * 4 + 3] = rb11; }
        s_hasn[t] = (len_v & kReadHasN) ? 1 : 0;           // a read with an N: the 4-bit compare decides its windows
    }
    PROF(0);
    if (ent_lds && t < ix.n_ent) { s_es[t] = (EntT)es_v; s_ee[t] = (EntT)ee_v; }
    s_k1[t] = 0xFFFFFFFFu; s_c2[t] = 0; s_nx[t] = 0xFFFFFFFFu;
    s_mm[t] = (uint8_t)(mm < 255 ? mm : 255);
    s_r[t] = r; s_geo[t] = (uint32_t)len | ((uint32_t)cl << 10) | ((uint32_t)cd << 20);
    // block-wide exclusive prefix sum of the candidate counts
    {
        uint32_t v = my_total;
        for (int off = 1; off < 64; off <<= 1) { uint32_t u = __shfl_up(v, off); if (lane >= off) v += u; }
        if (lane == 63) s_wsum[wid] = v;
        __syncthreads();
        uint32_t add = 0;
        for (int w = 0; w < wid; w++) add += s_wsum[w];
        s_off[t] = add + v - my_total;
        if (t == BS - 1) s_off[BS] = add + v;
    }
    __syncthreads();
    PROF(1);

    // replay state of this lane's read
    const int init = mm + cfg.mm_delta + 1;
    int low_inst = 0, low_mm = init, nxt = init;
    int best_q = -1;
    uint32_t best_j = 0;
    constexpr uint32_t kNone = 0xFFFFFFFFu, kOffStart = 1u << 18, kLazyBit = 1u << 16;

    for (uint32_t start = 0; start < BS;) {
        // reads [start, end): as many as fit the result buffer (a single read never exceeds it)
        const uint32_t base = s_off[start];
        uint32_t lo = start + 1, hi = BS;
        while (lo < hi) {                                   // largest end with s_off[end] - base <= CAP
            uint32_t mid = (lo + hi + 1) >> 1;
            if (s_off[mid] - base <= CAP) lo = mid; else hi = mid - 1;
        }
        const uint32_t end = lo;
        const uint32_t total = s_off[end] - base;
        for (uint32_t f0 = 0; f0 < total; f0 += KB * BS) {
            uint64_t tv[KB];                // suffix array element, then the window's start
            uint32_t meta[KB];              // read of the block | slot << 10 | flags; kNone = no candidate
            // ---- A: which candidate, and its suffix array element
#pragma unroll
            for (int i = 0; i < KB; i++) {
                const uint32_t f = f0 + (uint32_t)i * BS + t;
                meta[i] = kNone;
                tv[i] = 0;
                if (f < total) {
                    const uint32_t g = base + f;
                    uint32_t l2 = start, h2 = end - 1;              // read ri: last one with s_off[ri] <= g
                    while (l2 < h2) {
                        uint32_t mid = (l2 + h2 + 1) >> 1;
                        if (s_off[mid] <= g) l2 = mid; else h2 = mid - 1;
                    }
                    const uint32_t ri = l2;
                    const uint32_t local = g - s_off[ri];
                    const uint16_t *sp = s_sp + ri * slots_max;
                    int q = 0;
                    while (sp[q] <= local) q++;                      // slot holding candidate `local`
                    const uint32_t j = local - (q ? sp[q - 1] : 0);
                    uint64_t iv_f;
                    bool lazy;
                    if (cf) { iv_f = s_first[ri * slots_max + q]; lazy = ((s_lazy[ri] >> q) & 1) != 0; }
                    else {
                        const int sti = q >= cmaxs ? 1 : 0;
                        uint32_t iv_c;
                        iv_get(b, iv_slot(b, blockIdx.x * blockDim.x + ri, s0 + sti, q - sti * cmaxs), iv_f, iv_c);
                        lazy = (iv_c & kLazyFlag) != 0;
                    }
                    tv[i] = sa_get<WIDE>(ix, iv_f + j);
                    meta[i] = ri | ((uint32_t)q << 10) | (lazy ? kLazyBit : 0u);
                }
            }
            // ---- B: window start; the region flags and the window's blocks are requested, nothing waits for them here
            uint4 wv[KB][NBLK];
            uint8_t fb0[KB], fb1[KB];
#pragma unroll
            for (int i = 0; i < KB; i++) {
                fb0[i] = 0; fb1[i] = 0;
#pragma unroll
                for (int u = 0; u < NBLK; u++) wv[i][u] = make_uint4(0, 0, 0, 0);
                if (meta[i] != kNone) {
                    const uint32_t ri = meta[i] & 1023u;
                    const int q = (int)((meta[i] >> 10) & 63u);
                    const int c = q >= cmaxs ? q - cmaxs : q;
                    const uint32_t geo = s_geo[ri];
                    const int c_len = (int)(geo & 1023u), c_cl = (int)((geo >> 10) & 1023u), c_cd = (int)(geo >> 20);
                    const int last = c_len - c_cl;
                    const int ofs = c * c_cd < last ? c * c_cd : last;
                    if (tv[i] >= (uint64_t)ofs) {
                        const uint64_t t0 = tv[i] - (uint64_t)ofs;
                        tv[i] = t0;
                        if (two_bit) {
                            const uint64_t g0 = t0 >> ix.flag_shift, g1 = (t0 + (uint64_t)c_len - 1) >> ix.flag_shift;
                            fb0[i] = ix.nflag[g0 >> 3];
                            fb1[i] = ix.nflag[g1 >> 3];
                            window2_load<NW>(ix.tgt2, ix.tgt2s, t0, c_len, wv[i]);
                        }
                    } else
                        meta[i] |= kOffStart;
                }
            }
            // ---- C: compare, one result byte per candidate
#pragma unroll
            for (int i = 0; i < KB; i++) {
                if (meta[i] == kNone) continue;
                const uint32_t f = f0 + (uint32_t)i * BS + t;
                uint8_t rec = kRecSkip;
                if (!(meta[i] & kOffStart)) {
                    const uint32_t ri = meta[i] & 1023u;
                    const int q = (int)((meta[i] >> 10) & 63u);
                    const int sti = q >= cmaxs ? 1 : 0, c = q - sti * cmaxs, st = s0 + sti;
                    const bool lazy = (meta[i] & kLazyBit) != 0;
                    const uint32_t geo = s_geo[ri];
                    const int c_len = (int)(geo & 1023u), c_cl = (int)((geo >> 10) & 1023u), c_cd = (int)(geo >> 20);
                    const int last = c_len - c_cl;
                    const int ofs = c * c_cd < last ? c * c_cd : last;
                    const uint64_t t0 = tv[i];
                    const uint32_t cr = s_r[ri];
                    Window<NW> w;
                    bool flg = true;
                    if (two_bit) {
                        const uint64_t g0 = t0 >> ix.flag_shift, g1 = (t0 + (uint64_t)c_len - 1) >> ix.flag_shift;
                        flg = ((((uint32_t)fb0[i] >> (g0 & 7)) | ((uint32_t)fb1[i] >> (g1 & 7))) & 1) != 0;
                        uint64_t r2w[NW / 2], rnm[NW / 4];
                        flg |= s_hasn[ri] != 0;
                        if (ROWS) {
#pragma unroll
                            for (int u = 0; u < NW / 4; u++) {
                                const uint4 v = s_row[(ri * 2 + st) * (NW / 4) + u];
                                r2w[2 * u] = ((uint64_t)v.y << 32) | v.x;
                                r2w[2 * u + 1] = ((uint64_t)v.w << 32) | v.z;
                            }
                        } else
                            load_read_words2<NW>(b.rd2 + ((uint64_t)cr * 2 + st) * (NW / 2), r2w);
#pragma unroll
                        for (int u = 0; u < NW / 4; u++) rnm[u] = 0;
                        window2_compare<NW>(r2w, rnm, c_len, t0, wv[i], w);
                    }
                    if (flg) eval_window_rare<NW>(read_row(b, cr, st, s_hasn[ri] != 0), c_len, ix.tgt4, t0, w);       // N/EOS nearby, or a read with an N (rare): the 4-bit compare decides
                    bool skip = w.eos || (lazy && !core_clean<NW>(w, ofs, c_cl));
#pragma unroll 1
                    for (int c2 = 0; c2 < c; c2++) skip |= core_clean<NW>(w, c2 * c_cd, c_cl);   // earlier cores never sit at the clipped offset
                    if (!skip) rec = (uint8_t)(w.mm < 127 ? w.mm : 127);
                    if (WIDE) s_key[f] = (uint32_t)t0;
                    if (!WIDE && rec != kRecSkip) {
                        atomicAdd(&s_c2[ri], 1u << 16);
                        if (rec <= s_mm[ri]) atomicMin(&s_k1[ri], ((uint32_t)rec << 16) | (base + f - s_off[ri]));
                    }
                }
                s_rec[f] = rec;
            }
        }
        PROF(2);
        __syncthreads();
        PROF(3);
        auto replay_sequential = [&]() __attribute__((always_inline)) {
            const uint16_t *sp = s_sp + t * slots_max;
            const uint32_t rb = s_off[t] - base;
            bool done = false;
            uint32_t prev = 0;
            for (int q = 0; q < slots_max && !done; q++) {
                if ((q >= cmaxs ? q - cmaxs : q) >= nc) continue;           // not a core of this read (an empty slot)
                n_search++;
                const uint32_t upto = sp[q];
                for (uint32_t x = prev; x < upto; x++) {
                    const int cm = s_rec[rb + x];
                    if (cm == kRecSkip) continue;
                    n_cand++;
                    if (cm > mm || cm >= nxt) continue;
                    if (cm < low_mm) {
                        low_inst = 1; nxt = low_mm; low_mm = cm;
                        best_q = q; best_j = x - prev;
                    } else if (cm == low_mm)
                        low_inst++;
                    else
                        nxt = cm;
                    if (low_inst > cfg.max_hits && low_mm == 0) { done = true; break; }
                }
                prev = upto;
            }
        };
        if (WIDE) {
            // 5-byte indexes: the reference keys its set of seen targets by the target start truncated to 32 bits (SfxArrayV2.cpp:5932): a
            // candidate is taken for seen when an earlier candidate of the same strand pass (inside its entry, a match of its core) has
            // the same low word.  That depends on the candidates' positions only, so it is a pass of its own over the candidates'
            // lanes, in front of the reduction.  (A candidate marked here while another lane still scans past it changes nothing: the
            // first candidate with a key is never marked, and every later one finds it.)
            for (uint32_t f = t; f < total; f += BS) {
                if (s_rec[f] == kRecSkip) continue;
                const uint32_t g = base + f;
                uint32_t l2 = start, h2 = end - 1;
                while (l2 < h2) {
                    uint32_t mid = (l2 + h2 + 1) >> 1;
                    if (s_off[mid] <= g) l2 = mid; else h2 = mid - 1;
                }
                const uint32_t rb = s_off[l2] - base, local = f - rb;
                const uint32_t second = (s1 > s0) ? (uint32_t)s_sp[l2 * slots_max + cmaxs - 1] : 0xFFFFFFFFu;      // first candidate of the second strand pass
                const uint32_t from = local >= second ? second : 0u;
                const uint32_t kx = s_key[f];
                bool seen = false;
                for (uint32_t y = from; y < local && !seen; y++) seen = s_rec[rb + y] != kRecSkip && s_key[rb + y] == kx;
                if (seen) s_rec[f] = kRecSkip;
            }
            __syncthreads();
            for (uint32_t f = t; f < total; f += BS) {
                const uint8_t rec = s_rec[f];
                if (rec == kRecSkip) continue;
                const uint32_t g = base + f;
                uint32_t l2 = start, h2 = end - 1;
                while (l2 < h2) {
                    uint32_t mid = (l2 + h2 + 1) >> 1;
                    if (s_off[mid] <= g) l2 = mid; else h2 = mid - 1;
                }
                atomicAdd(&s_c2[l2], 1u << 16);
                if (rec <= s_mm[l2]) atomicMin(&s_k1[l2], ((uint32_t)rec << 16) | (g - s_off[l2]));
            }
            __syncthreads();
        }
        {
            // second pass over the candidates: how many reach the read's smallest count, and the smallest count above it
            for (uint32_t f = t; f < total; f += BS) {
                const int cm = s_rec[f];
                if (cm == kRecSkip) continue;
                const uint32_t g = base + f;
                uint32_t l2 = start, h2 = end - 1;
                while (l2 < h2) {
                    uint32_t mid = (l2 + h2 + 1) >> 1;
                    if (s_off[mid] <= g) l2 = mid; else h2 = mid - 1;
                }
                if (cm > (int)s_mm[l2]) continue;
                if ((uint32_t)cm == (s_k1[l2] >> 16)) atomicAdd(&s_c2[l2], 1u);
                else atomicMin(&s_nx[l2], (uint32_t)cm);
            }
            __syncthreads();
            if (mine && t >= start && t < end) {
                const uint32_t k1 = s_k1[t], c2 = s_c2[t];
                if (k1 != 0xFFFFFFFFu && (k1 >> 16) == 0 && (int)(c2 & 0xFFFFu) > cfg.max_hits)
                    replay_sequential();            // the reference stops at hit max_hits + 1 of an exact match: what it had seen until then counts
                else {
                    n_search += (uint32_t)((s1 - s0 + 1) * nc);
                    n_cand += c2 >> 16;
                    if (k1 != 0xFFFFFFFFu) {
                        low_mm = (int)(k1 >> 16);
                        low_inst = (int)(c2 & 0xFFFFu);
                        const uint32_t nx = s_nx[t];
                        nxt = nx < (uint32_t)init ? (int)nx : init;
                        const uint32_t local = k1 & 0xFFFFu;
                        const uint16_t *sp = s_sp + t * slots_max;
                        int q = 0;
                        while (sp[q] <= local) q++;
                        best_q = q;
                        best_j = local - (q ? sp[q - 1] : 0);
                    }
                }
            }
        }
        PROF(4);
        __syncthreads();
        PROF(5);
        start = end;
    }

    // what becomes of the read is decided first and the list appends are done BEFORE the result record is written: the barriers of
    // the append wait for every store the wave has issued, and the scattered 20-byte records take long to drain
    int rslt = BK_HR_NONE;
    if (mine) {
        rslt = classify(low_inst, low_mm, nxt, init, cfg.mm_delta, cfg.max_hits);
        if (rslt == BK_HR_NONE && phase + 1 < n_phases) {
            ReadPlan p = make_plan(len, cfg);
            int mm2, cl2, cd2, dummy[1];
            phase_params(p, cfg, phase + 1, mm2, cl2, cd2);
            int nc2 = core_offsets(len, cl2, cd2, p.max_slides, dummy, 0);
            if (nc2 <= kMaxCoresFast) my_cmax = (uint32_t)nc2;
            dest = 1;
        }
    }
    const uint64_t lt_mask = (1ULL << lane) - 1;
    for (int off = 32; off > 0; off >>= 1) {
        n_search += __shfl_down(n_search, off);
        n_cand += __shfl_down(n_cand, off);
        n_lcm += __shfl_down(n_lcm, off);
        uint32_t m = __shfl_down(my_cmax, off);
        my_cmax = m > my_cmax ? m : my_cmax;
    }
    uint32_t my_off = 0;
#pragma unroll
    for (int d = 1; d <= 3; d++) {
        uint64_t m = __ballot(dest == d);
        if (m) {
            uint32_t w = 0;
            if (lane == 0) w = atomicAdd(&s_cnt[d], (uint32_t)__popcll(m));
            w = __builtin_amdgcn_readfirstlane(w);
            if (dest == d) my_off = w + (uint32_t)__popcll(m & lt_mask);
        }
    }
    if (lane == 0) {
        if (my_cmax) atomicMax(&s_cmax, my_cmax);
        if (n_search) atomicAdd(&s_ctr[0], (unsigned long long)n_search);
        if (n_cand) atomicAdd(&s_ctr[1], (unsigned long long)n_cand);
        if (n_lcm) atomicAdd(&s_ctr[2], (unsigned long long)n_lcm);
    }
    PROF(6);
    __syncthreads();
    PROF(7);
    // lists of the stripe set: 0 = next phase, 1 = wave kernel, 2 = general kernel
    if (t >= 1 && t <= 3 && s_cnt[t]) s_base[t] = stripe_reserve(out, (int)t - 1, s_cnt[t]);
    if (t == 4 && s_cmax) stripe_max(out, s_cmax);
    if (t >= 8 && t < 11 && s_ctr[t - 8]) atomicAdd(&b.ctr[ctr_stripe() + t - 8], s_ctr[t - 8]);
    __syncthreads();
    PROF(8);
    if (dest) stripe_put(out, dest - 1, s_base[dest] + my_off, dest == 2 ? a : r);      // (wave list: the read's position, see iv_slot)
    if (mine && rslt != BK_HR_NONE) {
        uint64_t hit_left = 0;
        int hit_strand = '?', e = -1;
        if (low_inst >= 1) {
            const int sti = best_q >= cmaxs ? 1 : 0, c = best_q - sti * cmaxs, st = s0 + sti;
            const int last = len - cl;
            const int ofs = c * cd < last ? c * cd : last;
            const uint64_t bf = cf ? (uint64_t)s_first[t * slots_max + best_q] : iv_start(b, iv_slot(b, a, st, c));
            hit_left = sa_get<WIDE>(ix, bf + best_j) - (uint64_t)ofs;
            hit_strand = st ? '-' : '+';
            if (ent_lds) {
                int lo = 0, hi = (int)ix.n_ent - 1;
                while (lo <= hi) {
                    int mid = (lo + hi) >> 1;
                    if (hit_left < (uint64_t)s_es[mid]) hi = mid - 1;
                    else if (hit_left > (uint64_t)s_ee[mid]) lo = mid + 1;
                    else { e = mid; break; }
                }
            } else
                e = find_entry(ix, hit_left);
        }
        write_result(ix, cfg, b, r, len, rslt, low_inst, low_mm, nxt, hit_left, e, hit_strand, phase << 1);
    }
#if defined(BK_PROF) && BK_PROF == 1
    PROF(9);
    PROF_END;
#endif
}

// ------------------------------------------------------------------------------------------------
// k_wave: one wave per LocateCoreMultiples call for reads of <= 16*NW bases and <= 16 cores per
// strand (4-byte suffix arrays).  64 candidates of a core interval per step; the reference's
// SEQUENTIAL semantics are reproduced exactly with ballot prefix sums, as in k_heavy:
//   IterCnt counts only new, in-bounds targets; at the first loop top with IterCnt == 100 the
//   remaining copy count (n - j + 2) abandons the core when > MaxIter; MaxIter and the 1 024 000
//   node cap stop it; after MaxHits+1 exact instances everything stops (SfxArrayV2.cpp:5857-5875,6206).
// The reference's hash set of already-seen target starts is replaced by an equivalent test: target
// start T reached through core c was already processed in this strand pass  <=>  for some earlier
// core c2 the read's core c2 matches the target at T (so T+ofs[c2] lies in c2's suffix interval) AND
// that suffix lay inside the prefix of c2's interval that was actually walked (rank from the inverse
// suffix array; only looked up when c2's walk was cut short).

__device__ __forceinline__ uint32_t hash_key(uint32_t key, uint32_t mask)
{
    return (key * 2654435761u) & mask;      // table size is a power of two
}

__device__ __forceinline__ bool htab_contains(unsigned long long *tab, uint32_t mask, uint32_t epoch, uint32_t key)
{
    unsigned long long mine = ((unsigned long long)epoch << 32) | key;
    uint32_t h = hash_key(key, mask);
    for (;;) {
        unsigned long long v = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(v >> 32) != epoch) return false;
        if (v == mine) return true;
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ void htab_insert(unsigned long long *tab, uint32_t mask, uint32_t epoch, uint32_t key)
{
    unsigned long long mine = ((unsigned long long)epoch << 32) | key;
    uint32_t h = hash_key(key, mask);
    for (;;) {
        unsigned long long v = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(v >> 32) != epoch) {
            unsigned long long old = atomicCAS(&tab[h], v, mine);
            if (old == v) return;
            continue;                       // somebody else took the slot: look at it again
        }
        if (v == mine) return;
        h = (h + 1) & mask;
    }
}

// 5-byte indexes only: two candidates of one 64-candidate round whose target starts lie a multiple of 2^32 bases apart carry the
// same truncated key (SfxArrayV2.cpp:5932).  The reference, walking them one after the other, takes the later one for seen; the
// hash set is only consulted for what EARLIER rounds left in it, so the round is checked against itself here.
__device__ __forceinline__ bool same_key_earlier_in_round(bool cand, uint32_t key, int lane)
{
    bool dup = false;
    uint64_t vm = __ballot(cand);
    if (__popcll(vm) > 1) {
        // first a cheap look at six bits of a hash of the keys: lanes that share all six with no other candidate cannot have a twin
        // (almost every round ends here); the exact pass over the candidates runs only for the others
        const uint32_t h6 = (key * 2654435761u) >> 26;
        uint64_t peers = vm;
#pragma unroll
        for (int bit = 0; bit < 6; bit++) {
            const uint64_t bm = __ballot(cand && ((h6 >> bit) & 1));
            peers &= ((h6 >> bit) & 1) ? bm : ~bm;
        }
        vm = __ballot(cand && (peers & (peers - 1)) != 0);          // candidates that share their six bits with another candidate
    }
    if (__popcll(vm) > 1)
        while (vm) {
            const int l = __ffsll((unsigned long long)vm) - 1;
            vm &= vm - 1;
            const uint32_t k2 = __shfl(key, l);
            dup |= cand && lane > l && key == k2;
        }
    return dup;
}

// the wave kernel's LDS set of seen keys (HASH form)
constexpr uint32_t kLdsSet = 2048, kLdsSetFill = 1536, kLdsEmpty = 0xFFFFFFFFu;

__device__ __forceinline__ bool lset_contains(const uint32_t *set, uint32_t key)
{
    uint32_t h = hash_key(key, kLdsSet - 1);
    for (;;) {
        const uint32_t v = __hip_atomic_load(&set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v == kLdsEmpty) return false;
        if (v == key) return true;
        h = (h + 1) & (kLdsSet - 1);
    }
}

__device__ __forceinline__ void lset_insert(uint32_t *set, uint32_t key)
{
    uint32_t h = hash_key(key, kLdsSet - 1);
    for (;;) {
        const uint32_t old = atomicCAS(&set[h], kLdsEmpty, key);
        if (old == kLdsEmpty || old == key) return;
        h = (h + 1) & (kLdsSet - 1);
    }
}

constexpr int kWaveGrab = 8;

struct WaveCoreInfo {
    unsigned long long first;
    uint32_t n;
    uint32_t walked;        // number of leading SA entries of the interval whose loop body was reached
    int ofs;
};

// HASH: the reference's own dedupe instead - a per-wave set of the 32-bit truncated target-start keys
// (SfxArrayV2.cpp:5932), kept in HBM with epoch tags as in k_heavy.  This is the form for 5-byte indexes (no
// inverse suffix array; and only the truncated keys reproduce the reference there, where two starts 2^32 apart
// count as one) and for 4-byte indexes whose inverse suffix array was not built.
// SW: the index holds the suffix-ordered window array (DevIndex::swin) - reads it covers take their candidates' windows from it.
template <int NW, bool WIDE, bool HASH, bool SW, bool GROUP>
__global__ void __launch_bounds__(256, NW <= 8 ? 4 : 2) k_wave(DevIndex ix, DevAlignCfg cfg, DevBatch b, HeavyScratch hs,
                                              const uint32_t *__restrict__ list,
                                              uint32_t n_list, int phase, uint32_t *__restrict__ cursor,
                                              uint32_t *__restrict__ next_act, uint32_t *__restrict__ next_cnt,
                                              uint32_t *__restrict__ cmax_next)
{
    __shared__ WaveCoreInfo s_core[4][kMaxCoresFast];
    __shared__ uint64_t s_cmask[4][kMaxCoresFast][NW / 4];       // per core: its bases in the IWindow layout
    // HASH: the set of seen target keys of a strand pass lives in LDS (kLdsSet keys per wave, open addressing) and spills into the
    // wave's HBM table only when a pass inserts more than kLdsSetFill keys - a look-up and an insert in HBM are two or three more
    // random cache lines (and a compare-and-swap) on the dependent chain of every candidate
    __shared__ uint32_t s_set[HASH ? 4 : 1][HASH ? kLdsSet : 1];
    // the sequences' first and last base (up to 64 of them): a finished read finds its sequence without a trip to HBM at its very end
    __shared__ uint64_t s_es[64], s_ee[64];
    if (ix.n_ent <= 64) {
        if (threadIdx.x < ix.n_ent) { s_es[threadIdx.x] = ix.ent_start[threadIdx.x]; s_ee[threadIdx.x] = ix.ent_end[threadIdx.x]; }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    WaveCoreInfo *core = s_core[wib];
    uint64_t (*cmask)[NW / 4] = s_cmask[wib];
    uint32_t *lset = s_set[HASH ? wib : 0];
    uint32_t lset_n = kLdsSet;                   // keys in the LDS set (kLdsSet: not cleared yet)
    bool spilled = false;                        // this strand pass has keys in the HBM table as well
    const uint32_t wave_slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned long long *tab = nullptr;
    uint32_t tmask = 0, epoch = 0;
    if (HASH) {
        if (wave_slot >= hs.n_slots) return;
        tab = hs.htab + (uint64_t)wave_slot * hs.tab_size;
        tmask = hs.tab_size - 1;
        epoch = hs.slot_epoch[wave_slot];
    }
    const uint64_t lt_mask = (1ULL << lane) - 1;
    unsigned long long n_search = 0, n_cand = 0, n_lcm = 0, n_fetch = 0, n_dup = 0;
    constexpr bool kDiag = false;

    // work items are claimed kWaveGrab at a time: one device-scope atomic on the shared cursor per
    // item serialises 8192 resident waves on a single address
    uint32_t grab_next = 0, grab_left = 0;
    const int grab = kWaveGrab;
    // reads that go on to the next phase are parked one per lane and appended 64 at a time
    uint32_t pend_r = 0, pend_n = 0, cmax_loc = 0;
    auto flush_pending = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(next_cnt, pend_n);
        base = __builtin_amdgcn_readfirstlane(base);
        if ((uint32_t)lane < pend_n) next_act[base + lane] = pend_r;
        pend_n = 0;
    };
    int geo_len = -1, geo_mm = 0, geo_cl = 1, geo_cd = 1, geo_nc = 0;      // geometry of the reads of length geo_len in this phase
    ReadPlan geo_p{};
    for (;;) {
        if (grab_left == 0) {
            uint32_t g = 0;
            if (lane == 0) g = atomicAdd(cursor, (uint32_t)grab);
            grab_next = __builtin_amdgcn_readfirstlane(g);
            grab_left = (uint32_t)grab;
        }
        const uint32_t item = grab_next++;
        grab_left--;
        if (item >= n_list) break;
        // (wave-uniform values that arrive through vector loads are handed to the scalar unit explicitly: the read's plan, its loop
        // bounds and the window geometry then cost scalar instructions once instead of vector instructions in every lane)
        const uint32_t pos = __builtin_amdgcn_readfirstlane(list[item]);          // position in the phase's active list: where its interval records lie
        const uint32_t r = __builtin_amdgcn_readfirstlane(b.act[pos]);
        const uint32_t meta = __builtin_amdgcn_readfirstlane(b.rmeta[r]);
        const int len = (int)(meta & kReadLenMask);
        const bool has_n = (meta & kReadHasN) != 0;
        // the plan of the read's length, the core offsets and the cores' masks only change with the length: a batch of equal-length reads
        // computes them once per wave (they live in registers and in the wave's LDS words), not once per read
        if (len != geo_len) {
            geo_len = len;
            geo_p = make_plan(len, cfg);
            phase_params(geo_p, cfg, phase, geo_mm, geo_cl, geo_cd);
            int ofs_tmp[kMaxCoresFast];
            geo_nc = core_offsets(len, geo_cl, geo_cd, geo_p.max_slides, ofs_tmp, kMaxCoresFast);
            __builtin_amdgcn_wave_barrier();
            if (lane < geo_nc && lane < kMaxCoresFast) {
                int o = 0;
#pragma unroll
                for (int q = 0; q < kMaxCoresFast; q++) if (q == lane) o = ofs_tmp[q];
                core[lane].ofs = o;
            }
            __builtin_amdgcn_wave_barrier();
            const int ncm = geo_nc < kMaxCoresFast ? geo_nc : kMaxCoresFast;
            for (int idx = lane; idx < ncm * (NW / 4); idx += 64) {
                const int cc = idx / (NW / 4), i = idx % (NW / 4);
                const int o = core[cc].ofs;
                cmask[cc][i] = imask_word(o, o + geo_cl, i);
            }
        }
        const ReadPlan p = geo_p;
        const int mm = geo_mm, cl = geo_cl, cd = geo_cd, nc = geo_nc;
        (void)cd;
        n_lcm++;
        const int init = mm + cfg.mm_delta + 1;
        int low_inst = 0, low_mm = init, nxt = init;
        uint64_t hit_left = 0;
        int hit_strand = '?';
        bool done = false;
        int s0 = cfg.align_strand == 2 ? 1 : 0, s1 = cfg.align_strand == 1 ? 0 : 1;
        const bool sw_read = SW && len <= kSwLen && len - cl <= kSwPre;       // every core offset of the read lies within an entry's lead
        for (int st = s0; st <= s1 && !done; st++) {
            if (HASH) {                      // a new dedupe set per strand pass (SfxArrayV2.cpp:5834)
                if (lset_n) {
                    for (uint32_t i = lane; i < kLdsSet; i += 64) lset[i] = kLdsEmpty;
                    lset_n = 0;
                    __builtin_amdgcn_wave_barrier();
                }
                spilled = false;             // (the HBM table gets its new epoch when a pass first spills into it)
            }
            // the read's 2 bit/base row (the same for the whole wave: scalar registers).  The 4 bit/base words that a window near an
            // N or a sequence end needs are not kept in registers: that path (eval_window_rare) fetches them as it goes - from the
            // 2-bit row again, or, a read with an N, from its rd4 row, which also gives its N positions
            uint64_t r2w[NW / 2], rni[NW / 4];                           // rni: "read base is N", in the IWindow layout
            const bool two_bit = b.rd2 != nullptr;
            const RdRow row4 = read_row(b, r, st, has_n);
            if (two_bit) {
                load_read_words2<NW>(b.rd2 + ((uint64_t)r * 2 + st) * (NW / 2), r2w);
#pragma unroll
                for (int k = 0; k < NW / 2; k++) r2w[k] = uniform64(r2w[k]);
            } else {
#pragma unroll
                for (int k = 0; k < NW / 2; k++) r2w[k] = 0;
            }
#pragma unroll
            for (int k = 0; k < NW / 4; k++) rni[k] = 0;
            if (two_bit && has_n) {
                const uint64_t *__restrict__ rp = b.rd4 + ((uint64_t)r * 2 + st) * b.wpr;
#pragma unroll
                for (int q = 0; q < NW / 4; q++) {
                    uint64_t nm = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (16 * (4 * q + k) < len) nm |= (uint64_t)flags_to_bits16((uniform64(rp[4 * q + k]) >> 2) & 0x1111111111111111ULL) << (16 * k);
                    rni[q] = nm ? uniform64(bits_to_imap(nm)) : 0ULL;
                }
            }
            uint32_t my_cn = 0;                       // lane l < nc: suffixes in core l's interval
            if (lane < nc) {
                uint64_t slot = iv_slot(b, pos, st, lane);
                uint64_t f;
                uint32_t cn;
                iv_get(b, slot, f, cn);
                core[lane].first = f;
                core[lane].n = cn;                    // bit 31: unverified bucket (<= kLazyBucket members)
                core[lane].walked = 0;
                my_cn = cn & ~kLazyFlag;
            }
            __builtin_amdgcn_wave_barrier();
            // GROUP: consecutive cores with small intervals share a round - one candidate per lane in walk order (core, then suffix) -
            // instead of a round each: with a dozen cores per strand a wave otherwise spends most of its rounds on two or three
            // candidates.  None of the reference's iteration rules can fire inside such a round (fewer than 100 candidates per core,
            // the node cap checked before it), and a small interval is always walked to its end.
            uint32_t pre_ex = 0;                      // GROUP: candidates of the cores before this lane's (cores counted as min(n, 65))
            if (GROUP) {
                uint32_t v = my_cn > 64 ? 65u : my_cn;
                const uint32_t own = v;
                for (int off = 1; off < 16; off <<= 1) { const uint32_t u = __shfl_up(v, off); if (lane >= off) v += u; }
                pre_ex = v - own;
            }
            uint32_t nodes = 0;
            for (int c = 0; c < nc && !done && nodes < kNodeCap;) {
                // (every lane reads the same LDS words: told so, the compiler keeps them and what follows from them in scalar registers)
                const uint64_t first = uniform64(core[c].first);
                const uint32_t cn_c = __builtin_amdgcn_readfirstlane(core[c].n);
                const bool lazy = (cn_c & kLazyFlag) != 0;
                const uint64_t n = cn_c & ~kLazyFlag;
                const int ofs = __builtin_amdgcn_readfirstlane(core[c].ofs);
                // the cores of this step: c alone (a long interval, 64 suffixes a round), or c .. ce - 1 in one round
                int ce = c + 1;
                uint32_t gtot = (uint32_t)(n > 64 ? 65 : n);
                const bool grouped = GROUP && n <= 64 && nodes + 64 < kNodeCap;
                uint32_t pre_c = 0;
                if (GROUP && grouped) {
                    pre_c = __shfl(pre_ex, c);
                    const uint64_t stop_at = __ballot(lane > c && (lane >= nc || my_cn > 64 || pre_ex + my_cn - pre_c > 64));
                    ce = stop_at ? __ffsll((unsigned long long)stop_at) - 1 : nc;
                    gtot = __shfl(pre_ex, ce < 64 ? ce : 63) - pre_c;
                    if (ce >= nc) gtot = __shfl(pre_ex + (my_cn > 64 ? 65u : my_cn), nc - 1) - pre_c;
                    if (lane >= c && lane < ce) core[lane].walked = my_cn;       // (small intervals are walked whole)
                    __builtin_amdgcn_wave_barrier();
                }
                int lc = c;                              // this lane's core, its suffix within the interval
                uint64_t lfirst = first;
                int lofs = ofs;
                bool llazy = lazy;
                uint32_t lj_g = 0;
                if (GROUP && grouped) {
                    for (int l = c + 1; l < ce; l++) lc += (__shfl(pre_ex, l) - pre_c) <= (uint32_t)lane ? 1 : 0;
                    lj_g = (uint32_t)lane - (__shfl(pre_ex, lc) - pre_c);
                    lfirst = core[lc].first;
                    lofs = core[lc].ofs;
                    llazy = (core[lc].n & kLazyFlag) != 0;
                }
                n_search += (unsigned long long)(ce - c);
                uint32_t iter = 0;
                bool copies_checked = false;
                uint64_t walked = n;
                // the window array serves a core when the read's whole window lies inside the candidate's entry: bases kSwPre - ofs ..
                // + len of its kSwBases (every core of a read of up to kSwLen bases; of a longer read - 2 x 150 - the cores in the
                // middle, when they are walked a round per 64 suffixes; rounds shared by several cores of such a read go to the target)
                const bool sw_now = SW && ((GROUP && grouped) ? sw_read : (ofs <= kSwPre && len - ofs <= kSwBases - kSwPre));
                for (uint64_t j0 = 0; j0 < ((GROUP && grouped) ? 1u : n) && !done; j0 += 64) {
                    const uint64_t j = (GROUP && grouped) ? (uint64_t)lj_g : j0 + lane;
                    const bool active = (GROUP && grouped) ? (uint32_t)lane < gtot : j < n;
                    // (the candidate's entry of the window array is requested together with its suffix array element: one round trip)
                    uint4 ev[3] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
                    if (SW && sw_now && active) {
                        const uint4 *__restrict__ ep = ix.swin + (lfirst + j) * 3;
                        ev[0] = ep[0]; ev[1] = ep[1]; ev[2] = ep[2];
                    }
                    const uint64_t loci = active ? sa_get<WIDE>(ix, lfirst + j) : 0;
                    const uint64_t t = loci - (uint64_t)lofs;
                    bool valid = active && loci >= (uint64_t)lofs;
                    IWindow<NW> w;
                    w.mm = 127; w.eos = true;
#pragma unroll
                    for (int k = 0; k < NW / 4; k++) w.im[k] = ~0ULL;
                    if (valid) {
                        // the block-flag load and the 2-bit window loads are issued together; only the rare
                        // flagged window is then fetched again from the 4-bit copy
                        if (two_bit) {
                            const bool flg = window_flagged_t<WIDE>(ix, t, len);
                            if (SW && sw_now) {
                                if constexpr (SW) {
                                    if (GROUP && grouped) eval_swin2i<NW, false>(r2w, rni, len, ev, kSwPre - lofs, w);
                                    else eval_swin2i<NW, true>(r2w, rni, len, ev, kSwPre - lofs, w);
                                }
                            } else
                                eval_window2i<NW, WIDE>(r2w, rni, len, ix.tgt2, ix.tgt2s, t, w);
                            if (flg) {
                                Window<NW> w4;
                                eval_window_rare<NW>(row4, len, ix.tgt4, t, w4);
                                window_to_iwindow<NW>(w4, w);
                            }
                        } else {
                            Window<NW> w4;
                            eval_window_rare<NW>(row4, len, ix.tgt4, t, w4);
                            window_to_iwindow<NW>(w4, w);
                        }
                        valid = !w.eos && (!llazy || im_clean<NW>(w.im, cmask[lc]));
                    }
                    bool dup = false;
                    const uint32_t key = (uint32_t)(1 + loci - (uint32_t)lofs);       // 32-bit truncation as :5932
                    if (HASH) {
                        dup = valid && lset_n != 0 && lset_contains(lset, key);
                        if (spilled) dup = dup || (valid && htab_contains(tab, tmask, epoch, key));
                        // (a round against itself: two starts 2^32 apart in one interval; the same start reached through two cores of a group)
                        if (WIDE || (GROUP && grouped)) dup |= same_key_earlier_in_round(valid && !dup, key, lane);
                    } else for (int c2 = 0; c2 < ce - 1; c2++) {
                        bool m = valid && !dup && c2 < lc && im_clean<NW>(w.im, cmask[c2]);
                        if (__ballot(m)) {
                            if (m) {
                                if (core[c2].walked >= (core[c2].n & ~kLazyFlag)) dup = true;
                                else {
                                    uint64_t rank = (uint64_t)ix.isa[t + (uint64_t)core[c2].ofs] - core[c2].first;
                                    dup = rank < (uint64_t)core[c2].walked;
                                }
                            }
                        }
                    }
                    const bool isnew = valid && !dup;
                    if (kDiag) { n_fetch += __popcll(__ballot(active && loci >= (uint64_t)ofs)); n_dup += __popcll(__ballot(dup)); }
                    const uint64_t newmask = __ballot(isnew);
                    const uint32_t pre = (uint32_t)__popcll(newmask & lt_mask);
                    const uint32_t iter_before = iter + pre;
                    const uint32_t nodes_before = nodes + pre;
                    bool stop = active && !(GROUP && grouped) && ((cfg.max_iter && iter_before >= (uint32_t)cfg.max_iter) || nodes_before >= kNodeCap);
                    uint64_t cutoff = (GROUP && grouped) ? 64 : n;
                    uint64_t stopmask = __ballot(stop);
                    if (stopmask) cutoff = j0 + (uint64_t)(__ffsll((unsigned long long)stopmask) - 1);
                    if (!copies_checked && !(GROUP && grouped)) {
                        bool chk = active && j > 0 && iter_before == 100;
                        uint64_t chkmask = __ballot(chk);
                        if (chkmask) {
                            uint64_t jc = j0 + (uint64_t)(__ffsll((unsigned long long)chkmask) - 1);
                            if (jc < cutoff) {
                                copies_checked = true;
                                uint64_t num_copies = n - jc + 2;
                                if (cfg.max_iter && (uint32_t)num_copies > (uint32_t)cfg.max_iter) cutoff = jc;
                            }
                        }
                    }
                    const bool proc = active && ((GROUP && grouped) || j < cutoff) && isnew;
                    if (HASH) {
                        const uint32_t nins = (uint32_t)__popcll(__ballot(proc));
                        if (nins) {
                            // (the key that looks like an empty slot, one in 2^32, always goes to the HBM table)
                            const bool to_lds = lset_n + nins <= kLdsSetFill;
                            if ((!to_lds || __ballot(proc && key == kLdsEmpty)) && !spilled) {
                                spilled = true;
                                epoch++;
                                if (epoch == 0) {            // wrapped: really clear the table
                                    for (uint32_t i = lane; i < hs.tab_size; i += 64) tab[i] = 0;
                                    epoch = 1;
                                    __builtin_amdgcn_wave_barrier();
                                }
                            }
                            if (proc) {
                                if (to_lds && key != kLdsEmpty) lset_insert(lset, key);
                                else htab_insert(tab, tmask, epoch, key);
                            }
                            if (to_lds) lset_n += nins;
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                    int cm = (proc && w.mm <= mm && w.mm < nxt) ? w.mm : 127;
                    bool acc = cm != 127;
                    uint64_t keep = ~0ULL;
                    uint64_t zmask = __ballot(acc && cm == 0);
                    int zc0 = low_mm == 0 ? low_inst : 0;
                    bool exit_now = false;
                    if (zmask && zc0 + __popcll(zmask) > cfg.max_hits) {
                        int need = cfg.max_hits + 1 - zc0;
                        uint64_t z = zmask;
                        for (int q = 1; q < need; q++) z &= z - 1;
                        int cut_lane = __ffsll((unsigned long long)z) - 1;
                        keep = cut_lane >= 63 ? ~0ULL : ((2ULL << cut_lane) - 1);
                        exit_now = true;
                        if (GROUP && grouped) n_search -= (unsigned long long)(ce - 1 - __shfl(lc, cut_lane));      // the cores behind the exit are never searched
                    }
                    uint64_t procmask = __ballot(proc) & keep;
                    uint32_t nproc = (uint32_t)__popcll(procmask);
                    iter += nproc;
                    nodes += nproc;
                    n_cand += (lane == 0) ? nproc : 0;
                    acc = acc && ((keep >> lane) & 1);
                    uint64_t accmask = __ballot(acc);
                    if (accmask) {
                        // smallest and second smallest count among the accepted lanes: the counts are at most mm, so one ballot per
                        // value (scalar work) instead of two butterfly reductions through the LDS crossbar
                        int bmin = 127, bsec = 127;
                        uint64_t minmask = 0;
                        for (int m = 0; m <= mm; m++) {
                            const uint64_t bm = __ballot(acc && cm == m);
                            if (!bm) continue;
                            if (bmin == 127) { bmin = m; minmask = bm; }
                            else { bsec = m; break; }
                        }
                        int cnt = __popcll(minmask);
                        int fl = __ffsll((unsigned long long)minmask) - 1;
                        if (bmin < low_mm) {
                            nxt = low_mm < bsec ? low_mm : bsec;
                            low_mm = bmin;
                            low_inst = cnt;
                            hit_left = __shfl(t, fl);
                            hit_strand = st ? '-' : '+';
                        } else if (bmin == low_mm) {
                            low_inst += cnt;
                            if (bsec < nxt) nxt = bsec;
                        } else if (bmin < nxt)
                            nxt = bmin;
                    }
                    if (exit_now) done = true;
                    if (!(GROUP && grouped) && cutoff < j0 + 64) { walked = cutoff; break; }
                }
                if (!(GROUP && grouped) && lane == 0) core[c].walked = walked > 0x7FFFFFFFULL ? 0x7FFFFFFFu : (uint32_t)walked;
                __builtin_amdgcn_wave_barrier();
                c = ce;
            }
        }
        int rslt = classify(low_inst, low_mm, nxt, init, cfg.mm_delta, cfg.max_hits);      // wave-uniform
        if (rslt != BK_HR_NONE) {
            int e = -1;
            if (low_inst >= 1) {
                if (ix.n_ent <= 64) {       // one entry per lane
                    bool in = (uint32_t)lane < ix.n_ent && hit_left >= s_es[lane] && hit_left <= s_ee[lane];
                    uint64_t m = __ballot(in);
                    e = m ? __ffsll((unsigned long long)m) - 1 : -1;
                } else if (lane == 0)
                    e = find_entry(ix, hit_left);
            }
            if (lane == 0) write_result(ix, cfg, b, r, len, rslt, low_inst, low_mm, nxt, hit_left, e, hit_strand, (phase << 1) | 1);
        } else if (phase + 1 < p.n_phases) {
            int mm2, cl2, cd2, dummy[1];
            phase_params(p, cfg, phase + 1, mm2, cl2, cd2);
            int nc2 = core_offsets(len, cl2, cd2, p.max_slides, dummy, 0);
            if (nc2 <= kMaxCoresFast && (uint32_t)nc2 > cmax_loc) cmax_loc = (uint32_t)nc2;
            if ((uint32_t)lane == pend_n) pend_r = r;
            if (++pend_n == 64) flush_pending();
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (HASH && lane == 0) hs.slot_epoch[wave_slot] = epoch;
    if (pend_n) flush_pending();
    if (lane == 0 && cmax_loc) atomicMax(cmax_next, cmax_loc);
    if (lane == 0) {
        if (n_search) atomicAdd(&b.ctr[ctr_stripe() + 0], n_search);
        if (n_cand) atomicAdd(&b.ctr[ctr_stripe() + 1], n_cand);
        if (n_lcm) { atomicAdd(&b.ctr[ctr_stripe() + 2], n_lcm); atomicAdd(&b.ctr[ctr_stripe() + 3], n_lcm); }
        if (n_cand) atomicAdd(&b.ctr[ctr_stripe() + 4], n_cand);
        if (kDiag) { atomicAdd(&b.ctr[ctr_stripe() + 5], n_fetch); atomicAdd(&b.ctr[ctr_stripe() + 6], n_dup); }
    }
}

__global__ void k_build_isa(const uint32_t *__restrict__ sa, uint64_t n, uint32_t *__restrict__ isa)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) isa[sa[i]] = (uint32_t)i;
}

// CSfxArrayV3::AdaptiveTrim (SfxArrayV2.cpp:5482-5682) for one candidate: the longest stretch of the read that starts and ends in a
// run of >= min_flank matching bases, is at least min_trim long and stays under (max_mm + 1) % mismatches - counting, as the
// reference does, every base of a mismatching run against the length from the stretch's start (first test) and against the
// stretch itself (second test), in double precision.  The reference keeps a table of match / mismatch runs; here the runs are
// read off a bitmap of the mismatching positions (bit i of word i / 64): ATW = 8 words for reads of up to 512 bases (registers), 32 for
// the longest reads the boundary takes (2000 bases; the map then lives in scratch memory - its own instantiation of the kernel).
template <int ATW>
__device__ __forceinline__ int at_run_end(const uint64_t (&bm)[ATW], int p, int n)    // first q > p with bit(q) != bit(p), or n
{
    constexpr int kATWords = ATW;
    const int bit = (int)((bm[p >> 6] >> (p & 63)) & 1);
    int w = p >> 6;
    uint64_t x = (bit ? ~bm[w] : bm[w]) >> (p & 63);
    if (x) { const int q = p + (__ffsll((unsigned long long)x) - 1); return q < n ? q : n; }
    for (w++; w < kATWords && (w << 6) < n; w++) {
        x = bit ? ~bm[w] : bm[w];
        if (x) { const int q = (w << 6) + (__ffsll((unsigned long long)x) - 1); return q < n ? q : n; }
    }
    return n;
}

// returns the trimmed length (0 = nothing acceptable); trim5 / trim3 = bases cut from the start / end of the read as given
template <int ATW>
__device__ int adaptive_trim_dev(const uint64_t *__restrict__ rdw, const uint64_t *__restrict__ tgt, uint64_t t, int len, int min_trim, int max_mm,
                                 int min_flank, int &trim_mm, int &trim5, int &trim3)
{
    constexpr int kATWords = ATW;
    trim_mm = 0; trim5 = 0; trim3 = 0;
    if (len < 25 || len > 64 * kATWords || min_trim < 15 || min_trim > len || max_mm > 15 || min_flank > 10) return 0;
    if (min_flank == 0) min_flank = 1;
    uint64_t bm[kATWords];
#pragma unroll
    for (int w = 0; w < kATWords; w++) bm[w] = 0;
    for (int i = 0; i < len; i += 16) {
        const int nv = len - i < 16 ? len - i : 16;
        const uint64_t x = (nib16(rdw, i) ^ nib16(tgt, t + i)) & top_mask(nv);
        const uint64_t f = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ULL;
        bm[i >> 6] |= (uint64_t)flags_to_bits16(f) << (i & 63);            // bit k of the 16 = base i + k mismatches
    }
    // pass 1: is there an exact run of >= 8; first / last run that may start a stretch, last run that may end one
    bool have8 = false;
    int first_start = -1, last_start = -1, last_end = -1, first_end = -1;
    for (int p = 0; p < len;) {
        const int q = at_run_end<ATW>(bm, p, len), rl = q - p;
        const bool mm = ((bm[p >> 6] >> (p & 63)) & 1) != 0;
        if (!mm) {
            if (rl >= 8) have8 = true;
            if (rl >= min_flank) {
                if (p <= len - min_trim) { last_start = p; if (first_start < 0) first_start = p; }
                if (p + rl >= min_trim) { last_end = p; if (first_end < 0) first_end = p; }
            }
        }
        p = q;
    }
    if (!have8 || first_start < 0 || first_end < 0) return 0;
    const double lim = (max_mm + 1.0) / 100.0;
    int best_len = 0, best_mm = 0, best_start = 0, best_end = 0;
    for (int sp = first_start; sp <= last_start;) {
        const int sq = at_run_end<ATW>(bm, sp, len);
        const bool smm = ((bm[sp >> 6] >> (sp & 63)) & 1) != 0;
        const bool can_start = !smm && (sq - sp) >= min_flank && sp <= len - min_trim;
        if (can_start) {
            int cur_len = 0, cur_mm = 0;
            for (int p = sp; p < len && p <= last_end;) {
                const int q = at_run_end<ATW>(bm, p, len), rl = q - p;
                const bool mm = ((bm[p >> 6] >> (p & 63)) & 1) != 0;
                const bool can_end = !mm && rl >= min_flank && p + rl >= min_trim;
                cur_len += rl;
                p = q;
                if (mm) {
                    if (max_mm == 0) break;
                    cur_mm += rl;
                    if (lim <= (double)cur_mm / (double)(len - sp)) break;
                } else if (best_len == 0) {
                    best_start = sp; best_end = len - (sp + cur_len); best_len = cur_len; best_mm = 0;
                    continue;
                }
                if (cur_len < min_trim || !can_end) continue;
                if (lim <= (double)cur_mm / (double)cur_len) continue;
                if (best_len < cur_len || (best_len == cur_len && (best_mm == 0 || cur_mm < best_mm))) {
                    best_start = sp; best_end = len - (sp + cur_len); best_len = cur_len; best_mm = cur_mm;
                }
            }
        }
        sp = sq;
    }
    if (best_len < min_trim) return 0;
    trim_mm = best_mm; trim5 = best_start; trim3 = best_end;
    return best_len;
}

// ------------------------------------------------------------------------------------------------
// general wave-per-read form of one LocateCoreMultiples call

//
// ENUM form (multi-loci modes, MaxHits > 1): the read's result is already known; the call that produced it (its
// AlignReads phase is kept in bk_hit.flags) is replayed with the same cut-off rules and every candidate whose
// Hamming distance equals the final LowMMCnt is written out in discovery order - the contents of the
// reference's pHits[] when LocateCoreMultiples returns (SfxArrayV2.cpp:6157-6205: '+' strand first, cores in
// order, suffix array order within a core).  `enum_err` counts reads whose replay did not reproduce
// LowHitInstances (must stay 0).
//
// BEST form (`-N`, CSfxArrayV3::LocateBestMatches, SfxArrayV2.cpp:6654-7019): one call with the caller's MaxTotMM /
// CoreLen / CoreDelta (no phase schedule, no Hamming-delta rule); the answer is the first MaxHits candidates in
// (mismatches, discovery order) - what the reference's insertion list holds at the end (a new hit goes in front
// of the first one with more mismatches; once full, the worst entry falls off and the mismatch limit tightens to
// the new worst, :6917-6961).  Two replays per read: the first histograms the candidates by mismatches, the second
// writes each kept candidate straight to its final place (class offset + rank within the class).  Candidates are
// hashed before the entry table is consulted (only the concatenation end is checked, :6816), entry boundaries are
// caught by the EOS test of the Hamming loop - both as in the reference, they change which candidates count
// towards the iteration limits.  Output: dense rows of MaxHits loci per read + the count; the result record is
// written here (eHRhits / eHRnone, LowMMCnt and NxtLowMMCnt stay 0 as ProcCoredApprox leaves them, Aligner.cpp:9197-9218).
template <bool WIDE, int MODE>
__global__ void __launch_bounds__(256) k_heavy(DevIndex ix, DevAlignCfg cfg, DevBatch b, HeavyScratch hs,
                                               const uint32_t *__restrict__ list, uint32_t n_list, int phase_arg,
                                               uint32_t *__restrict__ cursor, uint32_t *__restrict__ next_act,
                                               uint32_t *__restrict__ next_cnt, uint32_t *__restrict__ cmax_next,
                                               const unsigned long long *__restrict__ loci_offs, bk_loci *__restrict__ loci_out,
                                               uint32_t *__restrict__ enum_err, bk_seg2 *__restrict__ seg2_aux = nullptr,
                                               bk_loci_trims *__restrict__ trims_out = nullptr)
{
    constexpr bool ENUM = MODE == 1 || MODE == 5, BEST = MODE == 2, CHIM = MODE == 3 || MODE == 4;
    constexpr int ATW = (MODE == 4 || MODE == 5) ? 32 : 8;  // MODE 4 / 5: the chimeric form / its replay for reads of more than 512 bases
    __shared__ LdsEntries s_le;
    __shared__ uint32_t s_hist[BEST ? 4 : 1][64], s_pre[BEST ? 4 : 1][64], s_run[BEST ? 4 : 1][64];
    __shared__ bk_loci s_first[BEST ? 4 : 1];
    lds_entries_load(s_le, ix);
    const int lane = threadIdx.x & 63;
    const int wib = BEST ? (int)(threadIdx.x >> 6) : 0;
    const uint32_t wave_slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave_slot >= hs.n_slots) return;
    unsigned long long *tab = hs.htab + (uint64_t)wave_slot * hs.tab_size;
    const uint32_t tmask = hs.tab_size - 1;
    uint32_t epoch = hs.slot_epoch[wave_slot];
    const uint64_t lt_mask = (1ULL << lane) - 1;
    unsigned long long n_search = 0, n_cand = 0, n_lcm = 0;

    // items are claimed kWaveGrab at a time and next-phase reads parked one per lane (see k_wave)
    uint32_t grab_next = 0, grab_left = 0, pend_r = 0, pend_n = 0, cmax_loc = 0;
    auto flush_pending = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(next_cnt, pend_n);
        base = __builtin_amdgcn_readfirstlane(base);
        if ((uint32_t)lane < pend_n) next_act[base + lane] = pend_r;
        pend_n = 0;
    };
    for (;;) {
        if (grab_left == 0) {
            uint32_t g = 0;
            if (lane == 0) g = atomicAdd(cursor, (uint32_t)kWaveGrab);
            grab_next = __builtin_amdgcn_readfirstlane(g);
            grab_left = kWaveGrab;
        }
        const uint32_t item = grab_next++;
        grab_left--;
        if (item >= n_list) break;
        uint32_t r = list[item];
        int len = (int)b.lens[r];
        ReadPlan p = make_plan(len, cfg);
        int mm, cl, cd;
        int phase = phase_arg, want_mm = 0;
        uint32_t want_n = 0, emitted = 0;
        unsigned long long emit_base = 0;
        bool clamped = false;
        // ENUM of a read whose result came from the chimeric call (`-c` with the multi-loci modes): that call is replayed and every
        // candidate whose trimmed length and mismatches equal the best pair is a locus, with its own end trims.  k_heavy<.., CHIM>
        // left the best trimmed length in the read's bk_seg2 record (flags 0x40, match_len; the host never sees it).
        bool chim_replay = false;
        int want_len = 0;
        if (ENUM) {
            const bk_hit h = b.out[r];
            phase = h.flags >> 1;
            want_mm = h.low_mm;
            emit_base = loci_offs[r];
            want_n = (uint32_t)(loci_offs[r + 1] - emit_base);
            clamped = h.rslt == BK_HR_HITINSTS;      // -X: only the first MaxHits loci of a read with more
            if (seg2_aux != nullptr && phase == p.n_phases) {
                const bk_seg2 g = seg2_aux[r];
                chim_replay = g.flags == 0x40;
                want_len = g.match_len;
            }
        }
        if (BEST) phase = p.n_phases - 1;                 // the caller's own MaxTotMM / CoreLen / CoreDelta
        phase_params(p, cfg, phase, mm, cl, cd);
        // CHIM (`-c`, AlignReads :7750-7757): the last call, with shorter cores; a placement is a candidate trimmed at its ends by
        // AdaptiveTrim to at least min_chimeric_len % of the read; longest first, then fewest mismatches (:5959-6080)
        int chim_min = 0, chim_best_len = 0, chim_best_mm = 0, chim_t5 = 0, chim_t3 = 0;
        if (CHIM || chim_replay) {
            phase = p.n_phases;
            mm = p.max_tot_mm;
            cl = len / (mm + 4) > cfg.min_core_len ? len / (mm + 4) : cfg.min_core_len;
            cd = p.max_slides > 1 ? len / (p.max_slides - 1) : len;
            if (cd < cl) cd = cl;
            chim_min = (phase_arg * len) / 100;            // phase_arg carries the percentage
        }
        // state an ambiguous microInDel / splice search left behind (k_indel); LocateCoreMultiples starts from it (:5775-5800)
        int in_inst = 0, in_mm = 0, in_nxt = 0;
        bool inherited = false;
        if (CHIM) {
            bk_seg2 *sg = reinterpret_cast<bk_seg2 *>(loci_out);
            const bk_seg2 st0 = sg[r];
            if (st0.flags == 0x80) {
                inherited = true;
                in_inst = st0.match_len; in_mm = st0.mismatches; in_nxt = in_mm + 2;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) { bk_seg2 z; z.match_loci = 0; z.match_len = 0; z.read_ofs = 0; z.mismatches = 0; z.flags = 0; z.score = 0; sg[r] = z; }
                if (in_inst > cfg.max_hits && in_mm == 0) {                     // :5775 - nothing is searched
                    if (lane == 0) write_result(ix, cfg, b, r, len, BK_HR_HITINSTS, in_inst, in_mm, in_nxt, 0, -1, '?', (phase << 1) | 1);
                    continue;
                }
            }
        }
        n_lcm++;
        const int init = mm + cfg.mm_delta + 1;
        int low_inst = 0, low_mm = init, nxt = init;
        if (CHIM && inherited) { low_inst = in_inst; low_mm = in_mm; nxt = in_nxt; }
        uint64_t hit_left = 0;
        int hit_ent = -1, hit_strand = '?';
        bool done = false;
        int s0 = cfg.align_strand == 2 ? 1 : 0, s1 = cfg.align_strand == 1 ? 0 : 1;
        int best_t = 0;                                    // BEST: last mismatch class kept, how many of it, total kept
        uint32_t best_need = 0, best_count = 0;
        if (BEST) { s_hist[wib][lane] = 0; s_run[wib][lane] = 0; __builtin_amdgcn_wave_barrier(); }
      for (int pass = 0; pass < (BEST ? 2 : 1); pass++) {
        if (BEST && pass == 1) {
            // classes 0..mm: inclusive scan of the histogram over the lanes
            const uint32_t hcnt = s_hist[wib][lane];
            uint32_t cum = hcnt;
            for (int d = 1; d < 64; d <<= 1) { uint32_t v = __shfl_up(cum, d); if (lane >= d) cum += v; }
            const uint32_t total = __shfl(cum, 63);
            best_count = total < (uint32_t)cfg.max_hits ? total : (uint32_t)cfg.max_hits;
            if (best_count == 0) break;
            const uint64_t reach = __ballot(cum >= best_count);
            best_t = __ffsll((unsigned long long)reach) - 1;
            s_pre[wib][lane] = cum - hcnt;
            __builtin_amdgcn_wave_barrier();
            best_need = best_count - s_pre[wib][best_t];
        }
        for (int st = s0; st <= s1 && !done; st++) {
            const uint64_t *rdw = b.rd4 + ((uint64_t)r * 2 + st) * b.wpr;
            // new dedupe set for this strand pass (memset of the hash heads, SfxArrayV2.cpp:5834)
            epoch++;
            if (epoch == 0) {               // wrapped: really clear the table
                for (uint32_t i = lane; i < hs.tab_size; i += 64) tab[i] = 0;
                epoch = 1;
            }
            uint32_t nodes = 0;
            // walk the cores in order; every 64 cores the lanes search one core each
            int cur = cd, o = 0, ci = 0;
            uint64_t my_first = 0, my_n = 0;
            int my_ofs = 0;
            while (ci < p.max_slides && o <= len - cl && cur > cl / 3 && nodes < kNodeCap && !done) {
                if ((ci & 63) == 0) {
                    // replay the sliding rule from here for the next 64 cores; lane l takes core ci + l
                    int cur2 = cur, o2 = o, c2 = ci;
                    bool have = false;
                    while (c2 < ci + 64 && c2 < p.max_slides && o2 <= len - cl && cur2 > cl / 3) {
                        if (o2 + cl + cur2 > len) cur2 = len - (o2 + cl);
                        if (c2 - ci == lane) { my_ofs = o2; have = true; }
                        c2++;
                        o2 += cur2;
                    }
                    my_first = 0; my_n = 0;
                    if (have) search_core<WIDE>(ix, rdw, my_ofs, cl, ~0ULL >> 1, my_first, my_n);
                }
                if (o + cl + cur > len) cur = len - (o + cl);
                const int ofs = o;
                const uint64_t first = __shfl(my_first, ci & 63);
                const uint64_t n = __shfl(my_n, ci & 63);
                if (!BEST || pass == 0) n_search++;
                // candidate walk of this core, 64 SA elements per step
                uint32_t iter = 0;
                bool copies_checked = false;
                for (uint64_t j0 = 0; j0 < n && !done; j0 += 64) {
                    uint64_t j = j0 + lane;
                    bool active = j < n;
                    uint64_t loci = active ? sa_get<WIDE>(ix, first + j) : 0;
                    uint64_t t = loci - (uint64_t)ofs;
                    int e = -1;
                    bool valid = active && loci >= (uint64_t)ofs;
                    if (BEST) valid = valid && t + (uint64_t)len <= ix.n;
                    else if (valid) {
                        e = find_entry_lds(s_le, ix, t);
                        valid = e >= 0 && t + (uint64_t)len - 1 <= ix.ent_end[e];
                    }
                    uint32_t key = (uint32_t)(1 + loci - (uint32_t)ofs);       // 32-bit truncation as :5932
                    bool isnew = valid && !htab_contains(tab, tmask, epoch, key);
                    if (WIDE) isnew = isnew && !same_key_earlier_in_round(isnew, key, lane);
                    uint64_t newmask = __ballot(isnew);
                    uint32_t pre = (uint32_t)__popcll(newmask & lt_mask);
                    uint32_t iter_before = iter + pre;
                    uint32_t nodes_before = nodes + pre;
                    // loop-top conditions of the reference's while() for candidate j (:5857-5875)
                    bool stop = active && ((cfg.max_iter && iter_before >= (uint32_t)cfg.max_iter) || nodes_before >= kNodeCap);
                    uint64_t cutoff = n;                               // first candidate index NOT processed
                    uint64_t stopmask = __ballot(stop);
                    if (stopmask) cutoff = j0 + (uint64_t)(__ffsll((unsigned long long)stopmask) - 1);
                    if (!copies_checked) {
                        bool chk = active && j > 0 && iter_before == 100;
                        uint64_t chkmask = __ballot(chk);
                        if (chkmask) {
                            uint64_t jc = j0 + (uint64_t)(__ffsll((unsigned long long)chkmask) - 1);
                            if (jc < cutoff) {
                                copies_checked = true;
                                uint64_t num_copies = n - jc + 2;      // 1 + LastTargIdx - TargIdx, :5871-5872
                                if (cfg.max_iter && (uint32_t)num_copies > (uint32_t)cfg.max_iter) cutoff = jc;
                            }
                        }
                    }
                    bool proc = active && j < cutoff && isnew;
                    if (proc) htab_insert(tab, tmask, epoch, key);
                    int cm = 127;
                    if (CHIM || chim_replay) {
                        int c_len = 0, c_mm = 0, c_t5 = 0, c_t3 = 0, e2 = -1;
                        if (proc) {
                            c_len = adaptive_trim_dev<ATW>(rdw, ix.tgt4, t, len, chim_min, mm, 3, c_mm, c_t5, c_t3);
                            if (c_len < chim_min) c_len = 0;
                            if (c_len) { e2 = find_entry_lds(s_le, ix, t + (uint64_t)c_t5); if (e2 < 0) e2 = e; }
                        }
                        const uint32_t np = (uint32_t)__popcll(__ballot(proc));
                        iter += np;
                        nodes += np;
                        n_cand += (lane == 0) ? np : 0;
                        if (ENUM) {
                            const bool hit = c_len > 0 && c_len == want_len && c_mm == want_mm;
                            const uint64_t hmask = __ballot(hit);
                            if (hit) {
                                const uint32_t k = emitted + (uint32_t)__popcll(hmask & lt_mask);
                                if (k < want_n) {
                                    bk_loci L;
                                    L.chrom_id = ix.ent_id[e2];
                                    L.match_loci = (uint32_t)(t - ix.ent_start[e2]);
                                    L.match_len = (uint16_t)len;
                                    L.strand = (uint8_t)(st ? '-' : '+');
                                    L.mismatches = (uint8_t)c_mm;
                                    loci_out[emit_base + k] = L;
                                    if (trims_out != nullptr) {
                                        bk_loci_trims T;
                                        T.left = (uint16_t)(st ? c_t3 : c_t5); T.right = (uint16_t)(st ? c_t5 : c_t3); T.chimeric = 1; T.reserved = 0;
                                        trims_out[emit_base + k] = T;
                                    }
                                }
                            }
                            emitted += (uint32_t)__popcll(hmask);
                            if (clamped && emitted >= want_n) done = true;
                            if (cutoff < j0 + 64) break;
                            continue;
                        }
                        uint64_t hm = __ballot(c_len > 0);
                        while (hm && !done) {                       // in suffix-array order, as the reference meets them
                            const int src = __ffsll((unsigned long long)hm) - 1;
                            hm &= hm - 1;
                            const int l2 = __shfl(c_len, src), m2 = __shfl(c_mm, src);
                            if (l2 > chim_best_len || (l2 == chim_best_len && m2 < chim_best_mm)) {
                                if (chim_best_len > 0 && l2 > chim_best_len) low_mm = m2 + cfg.mm_delta + 1;
                                chim_best_len = l2; chim_best_mm = m2;
                                low_inst = 1;
                                nxt = low_mm;
                                low_mm = m2;
                                hit_left = __shfl(t, src); hit_ent = __shfl(e2, src); hit_strand = st ? '-' : '+';
                                chim_t5 = __shfl(c_t5, src); chim_t3 = __shfl(c_t3, src);
                            } else if (l2 == chim_best_len && m2 == chim_best_mm)
                                low_inst++;
                            else if (l2 == chim_best_len && m2 < nxt)
                                nxt = m2;
                            if (l2 == len && low_inst > cfg.max_hits && low_mm == 0) done = true;
                        }
                        if (cutoff < j0 + 64) break;
                        continue;
                    }
                    if (BEST) {
                        if (proc) cm = hamming_eos(rdw, len, ix.tgt4, t, pass ? best_t : mm);
                        const bool hit = cm != 127;
                        if (pass == 0) {
                            if (hit) atomicAdd(&s_hist[wib][cm], 1u);
                        } else if (__ballot(hit)) {
                            uint32_t my_pos = 0xffffffffu;
                            for (int m = 0; m <= best_t; m++) {
                                const uint64_t bm = __ballot(hit && cm == m);
                                if (!bm) continue;
                                const uint32_t run = s_run[wib][m];
                                if (hit && cm == m) {
                                    const uint32_t k = run + (uint32_t)__popcll(bm & lt_mask);
                                    if (m < best_t || k < best_need) my_pos = s_pre[wib][m] + k;
                                }
                                __builtin_amdgcn_wave_barrier();
                                if (lane == 0) s_run[wib][m] = run + (uint32_t)__popcll(bm);
                                __builtin_amdgcn_wave_barrier();
                            }
                            if (my_pos != 0xffffffffu) {
                                e = find_entry_lds(s_le, ix, t);
                                bk_loci L;
                                L.chrom_id = ix.ent_id[e];
                                L.match_loci = (uint32_t)(t - ix.ent_start[e]);
                                L.match_len = (uint16_t)len;
                                L.strand = (uint8_t)(st ? '-' : '+');
                                L.mismatches = (uint8_t)cm;
                                loci_out[(unsigned long long)r * (unsigned)cfg.max_hits + my_pos] = L;
                                if (my_pos == 0) s_first[wib] = L;
                            }
                        }
                        const uint32_t np = (uint32_t)__popcll(__ballot(proc));
                        iter += np;
                        nodes += np;
                        if (pass == 0) n_cand += (lane == 0) ? np : 0;
                        if (cutoff < j0 + 64) break;
                        continue;
                    }
                    if (proc) {
                        int lim = ENUM ? want_mm : (mm < nxt - 1 ? mm : nxt - 1);
                        cm = hamming(rdw, len, ix.tgt4, t, lim);
                        if (cm > lim) cm = 127;
                    }
                    if (ENUM) {
                        const bool hit = proc && cm == want_mm;
                        const uint64_t hmask = __ballot(hit);
                        if (hit) {
                            const uint32_t k = emitted + (uint32_t)__popcll(hmask & lt_mask);
                            if (k < want_n) {
                                bk_loci L;
                                L.chrom_id = ix.ent_id[e];
                                L.match_loci = (uint32_t)(t - ix.ent_start[e]);
                                L.match_len = (uint16_t)len;
                                L.strand = (uint8_t)(st ? '-' : '+');
                                L.mismatches = (uint8_t)cm;
                                loci_out[emit_base + k] = L;
                            }
                        }
                        emitted += (uint32_t)__popcll(hmask);
                        if (clamped && emitted >= want_n) done = true;
                        const uint32_t np = (uint32_t)__popcll(__ballot(proc));
                        iter += np;
                        nodes += np;
                        if (cutoff < j0 + 64) break;
                        continue;
                    }
                    bool acc = cm != 127;
                    // early exit once MaxHits+1 exact instances have been seen, in order (:6206)
                    uint64_t keep = ~0ULL;
                    uint64_t zmask = __ballot(acc && cm == 0);
                    int zc0 = low_mm == 0 ? low_inst : 0;
                    if (zmask && zc0 + __popcll(zmask) > cfg.max_hits) {
                        int need = cfg.max_hits + 1 - zc0;
                        uint64_t z = zmask;
                        for (int q = 1; q < need; q++) z &= z - 1;
                        int cut_lane = __ffsll((unsigned long long)z) - 1;
                        keep = cut_lane >= 63 ? ~0ULL : ((2ULL << cut_lane) - 1);
                        done = true;
                    }
                    uint64_t procmask = __ballot(proc) & keep;
                    uint32_t nproc = (uint32_t)__popcll(procmask);
                    iter += nproc;
                    nodes += nproc;
                    n_cand += (lane == 0) ? nproc : 0;
                    acc = acc && ((keep >> lane) & 1);
                    uint64_t accmask = __ballot(acc);
                    if (accmask) {
                        int v = acc ? cm : 127;
                        int bmin = v;
                        for (int off = 32; off > 0; off >>= 1) { int w = __shfl_xor(bmin, off); bmin = w < bmin ? w : bmin; }
                        int v2 = (acc && cm > bmin) ? cm : 127;
                        int bsec = v2;
                        for (int off = 32; off > 0; off >>= 1) { int w = __shfl_xor(bsec, off); bsec = w < bsec ? w : bsec; }
                        uint64_t minmask = __ballot(acc && cm == bmin);
                        int cnt = __popcll(minmask);
                        int fl = __ffsll((unsigned long long)minmask) - 1;
                        if (bmin < low_mm) {
                            nxt = low_mm < bsec ? low_mm : bsec;
                            low_mm = bmin;
                            low_inst = cnt;
                            hit_left = __shfl(t, fl);
                            hit_ent = __shfl(e, fl);
                            hit_strand = st ? '-' : '+';
                        } else if (bmin == low_mm) {
                            low_inst += cnt;
                            if (bsec < nxt) nxt = bsec;
                        } else if (bmin < nxt)
                            nxt = bmin;
                    }
                    if (cutoff < j0 + 64) break;                    // core abandoned / iteration limit
                }
                if (CHIM && low_inst > cfg.max_hits && low_mm == 0) done = true;        // :6206-6213
                ci++;
                o += cur;
            }
        }
      }   // pass
        if (ENUM) {
            if (lane == 0 && (clamped ? emitted < want_n : emitted != want_n)) atomicAdd(enum_err, 1u);
            continue;
        }
        if (BEST) {
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                bk_hit h;
                h.chrom_id = 0; h.match_loci = 0; h.match_len = 0; h.low_hit_instances = (int16_t)best_count;
                h.rslt = best_count ? BK_HR_HITS : BK_HR_NONE; h.nar = BK_NAR_NOHIT; h.strand = '?'; h.low_mm = 0; h.nxt_low_mm = 0;
                h.num_hits = 0; h.mismatches = 0; h.flags = (uint8_t)((phase << 1) | 1);
                if (best_count == 1) {
                    const bk_loci L = s_first[wib];
                    h.nar = BK_NAR_ACCEPTED; h.num_hits = 1; h.strand = L.strand; h.chrom_id = L.chrom_id; h.match_loci = L.match_loci;
                    h.match_len = L.match_len; h.mismatches = L.mismatches;
                } else if (best_count > 1)
                    h.nar = BK_NAR_MULTIALIGN;
                b.out[r] = h;
                ((unsigned long long *)loci_offs)[r] = best_count;          // here: the per-read count array
            }
            continue;
        }
        int rslt = classify(low_inst, low_mm, nxt, init, cfg.mm_delta, cfg.max_hits);      // wave-uniform
        if (CHIM && inherited) {                            // the general tail of LocateCoreMultiples (:6238-6261)
            if (low_mm == in_mm && low_inst == in_inst) {
                if (in_nxt > nxt) rslt = (nxt - in_mm) < cfg.mm_delta ? BK_HR_MMDELTA : BK_HR_RMMDELTA;
                else rslt = BK_HR_NONE;
            } else if (low_inst >= 1 && (nxt - low_mm) < cfg.mm_delta) rslt = BK_HR_MMDELTA;
            else if (low_inst > cfg.max_hits) rslt = BK_HR_HITINSTS;
            else rslt = BK_HR_HITS;
            if (rslt == BK_HR_RMMDELTA) {                   // ProcCoredApprox only takes the new NxtLowMMCnt (Aligner.cpp:9470-9473)
                if (lane == 0) { bk_hit h = b.out[r]; h.rslt = BK_HR_RMMDELTA; h.nxt_low_mm = (int8_t)nxt; h.flags = (uint8_t)((phase << 1) | 1); b.out[r] = h; }
                continue;
            }
        }
        if (rslt != BK_HR_NONE) {
            if (lane == 0) {
                write_result(ix, cfg, b, r, len, rslt, low_inst, low_mm, nxt, hit_left, hit_ent, hit_strand, (phase << 1) | 1);
                if (CHIM && rslt == BK_HR_HITS && low_inst == 1) {          // Seg[0].TrimLeft / TrimRight in read orientation (:6027-6036)
                    bk_seg2 g;
                    g.match_loci = 0; g.mismatches = 0; g.score = 0; g.flags = 8;
                    g.match_len = (uint16_t)(hit_strand == '+' ? chim_t5 : chim_t3);
                    g.read_ofs = (uint16_t)(hit_strand == '+' ? chim_t3 : chim_t5);
                    reinterpret_cast<bk_seg2 *>(loci_out)[r] = g;
                } else if (CHIM && cfg.max_hits > 1 && low_inst > 1 && chim_best_len > 0 && (rslt == BK_HR_HITS || rslt == BK_HR_HITINSTS)) {
                    bk_seg2 g;                                              // for the replay that lists the loci (see ENUM above)
                    g.match_loci = 0; g.mismatches = 0; g.score = 0; g.flags = 0x40; g.read_ofs = 0;
                    g.match_len = (uint16_t)chim_best_len;
                    reinterpret_cast<bk_seg2 *>(loci_out)[r] = g;
                }
            }
        } else if (phase + 1 < p.n_phases) {
            int mm2, cl2, cd2, dummy[1];
            phase_params(p, cfg, phase + 1, mm2, cl2, cd2);
            int nc2 = core_offsets(len, cl2, cd2, p.max_slides, dummy, 0);
            if (nc2 <= kMaxCoresFast && (uint32_t)nc2 > cmax_loc) cmax_loc = (uint32_t)nc2;
            if ((uint32_t)lane == pend_n) pend_r = r;
            if (++pend_n == 64) flush_pending();
        }
    }
    if (pend_n) flush_pending();
    if (lane == 0 && cmax_loc) atomicMax(cmax_next, cmax_loc);
    if (lane == 0) {
        hs.slot_epoch[wave_slot] = epoch;
        if (!ENUM) {                                   // the ENUM replay is ours, not work the reference does
            if (n_search) atomicAdd(&b.ctr[ctr_stripe() + 0], n_search);
            if (n_cand) atomicAdd(&b.ctr[ctr_stripe() + 1], n_cand);
            if (n_lcm) { atomicAdd(&b.ctr[ctr_stripe() + 2], n_lcm); atomicAdd(&b.ctr[ctr_stripe() + 3], n_lcm); }
            if (n_cand) atomicAdd(&b.ctr[ctr_stripe() + 4], n_cand);
        }
    }
}

// multi-loci bookkeeping: per read the number of loci to report (LowHitInstances of a read whose AlignReads
// returned eHRhits, else 0); after the scan, reads with one locus copy it from their result record and reads
// with several are queued for the replay above
__global__ void __launch_bounds__(256) k_loci_count(const bk_hit *__restrict__ out, uint32_t n, int clamp_to,
                                                    unsigned long long *__restrict__ cnt)
{
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const bk_hit h = out[r];
    unsigned long long c = 0;
    if (h.rslt == BK_HR_HITS && h.low_hit_instances > 0) c = (unsigned long long)h.low_hit_instances;
    else if (h.rslt == BK_HR_HITINSTS && clamp_to > 0) c = (unsigned long long)clamp_to;
    cnt[r] = c;
}

__global__ void __launch_bounds__(256) k_loci_single(const bk_hit *__restrict__ out, uint32_t n, const unsigned long long *__restrict__ offs,
                                                     bk_loci *__restrict__ loci, uint32_t *__restrict__ list, uint32_t *__restrict__ list_cnt,
                                                     const bk_seg2 *__restrict__ seg2, bk_loci_trims *__restrict__ trims)
{
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const bk_hit h = out[r];
    if (offs[r + 1] == offs[r]) return;
    if (h.rslt == BK_HR_HITS && h.low_hit_instances == 1) {
        bk_loci L;
        L.chrom_id = h.chrom_id; L.match_loci = h.match_loci; L.match_len = h.match_len; L.strand = h.strand; L.mismatches = h.mismatches;
        loci[offs[r]] = L;
        if (trims != nullptr && seg2 != nullptr) {             // a unique chimeric placement: its trims travel in the bk_seg2 record
            const bk_seg2 g = seg2[r];
            if (g.flags & 8) { bk_loci_trims T; T.left = g.match_len; T.right = g.read_ofs; T.chimeric = 1; T.reserved = 0; trims[offs[r]] = T; }
        }
    } else
        list[atomicAdd(list_cnt, 1u)] = r;
}

// ------------------------------------------------------------------------------------------------
// microInDels (`-a`): CSfxArrayV3::LocateInDels (SfxArrayV2.cpp:7348-7660) for the reads the AlignReads phases left
// unaligned, called as AlignReads does (:7722-7734): core = min(2 CoreLen, (len-1)/2), at most cMaxMicroInDelMM (2)
// mismatches, MaxHits 1.  One wave per read; per strand two anchor cores (5' end explored rightwards, 3' end
// leftwards); a lane takes one suffix-array member of the anchor's interval and runs ExploreInDelMatchRight / -Left
// (:8943-9405) on it: mismatch positions of the ungapped compare (only placements with >= 7 of them are explored),
// then for each of the first MaxTotMM+1 positions every gap length 1..microInDelLen as insertion and as deletion,
// keeping the best score.  The lanes' results are then folded in suffix-array order with the reference's rule:
// a higher score replaces, an equal score at another Seg[0] start makes the read ambiguous.

struct IndelPlacement {
    int r, score, is_insert;
    uint64_t s0_loci, s1_loci;
    int s0_len, s0_mm, s1_len, s1_mm, s1_ofs;
};

__device__ __forceinline__ int rd_base4(const uint64_t *__restrict__ rdw, int i)
{
    return (int)((rdw[i >> 4] >> (60 - 4 * (i & 15))) & 7);
}
__device__ __forceinline__ int tg_base4(const uint64_t *__restrict__ tgt, uint64_t pos)
{
    return (int)((tgt[pos >> 4] >> (60 - 4 * (int)(pos & 15))) & 7);
}

constexpr int kIndelMMExplore = 7, kIndelMinSeq = 7, kIndelMaxMM = 2, kIndelBase = 500, kIndelMaxScore = 1000, kIndelMatch = 3,
              kIndelMismatch = 5, kIndelOpen = 20, kIndelExt = 1;

// right == true: ExploreInDelMatchRight, else ExploreInDelMatchLeft.  t = target offset of the read's first base,
// ent_left = bases of the entry from t to its end (right) - only used for the two "enough target left" tests.
template <bool RIGHT>
__device__ void explore_indel(const uint64_t *__restrict__ rdw, const uint64_t *__restrict__ tgt, int plen, uint64_t t, uint32_t targ_seq_len,
                              int max_len, int max_mm, IndelPlacement &out)
{
    out.r = 0; out.score = 0; out.is_insert = 0; out.s0_loci = 0; out.s1_loci = 0; out.s0_len = 0; out.s0_mm = 0; out.s1_len = 0; out.s1_mm = 0; out.s1_ofs = 0;
    int mm_ofs[kIndelMMExplore + 2];
    int n_mm = 0;
    const int lim = max_mm > kIndelMMExplore ? max_mm : kIndelMMExplore;      // max_mm <= 2 here
    for (int q = 0; q < plen && n_mm <= lim; q++) {
        const int i = RIGHT ? q : plen - 1 - q;
        const int pb = rd_base4(rdw, i), tb = tg_base4(tgt, t + (uint64_t)i);
        if (tb > 4 || pb > 4) return;
        if (pb == tb && pb <= 3) continue;
        if (n_mm < kIndelMMExplore + 2) mm_ofs[n_mm] = i;
        n_mm++;
    }
    if (n_mm < kIndelMMExplore || (RIGHT ? kIndelMinSeq > plen - mm_ofs[0] : kIndelMinSeq > mm_ofs[0])) {
        if (n_mm > max_mm) return;
        out.r = 1; out.s0_len = plen; out.s0_loci = t; out.s0_mm = n_mm;
        out.score = kIndelBase + plen * kIndelMatch - n_mm * kIndelMismatch;
        return;
    }
    const int tot = max_mm < n_mm ? max_mm : n_mm;
    IndelPlacement ins = out, del = out;
    for (int pass = 0; pass < 2; pass++) {                 // 0: insertion into the read, 1: deletion from it
        IndelPlacement &best = pass ? del : ins;
        for (int k = 0; k <= tot; k++) {
            const int mo = mm_ofs[k];
            if (RIGHT ? !(kIndelMinSeq < plen - mo) : !(kIndelMinSeq < mo)) break;
            for (int l = 1; l <= max_len; l++) {
                int score = kIndelBase + plen * kIndelMatch - ((l - 1) * kIndelExt + kIndelOpen);
                int rest, p0;                              // bases compared, first probe index
                long long t0;                              // first target offset relative to t
                if (RIGHT) {
                    if (pass == 0) { rest = plen - (mo + l); p0 = mo + l; t0 = mo; }
                    else { rest = plen - mo; p0 = mo; t0 = mo + l; }
                    if (rest < kIndelMinSeq) break;
                    const uint32_t trest = targ_seq_len - (uint32_t)(pass == 0 ? mo : mo + l);
                    if (trest < (uint32_t)rest) break;
                } else {
                    score -= k * kIndelMismatch;
                    if (score < best.score) break;
                    if (pass == 0) { rest = mo - (l - 1); p0 = mo - l; t0 = mo; }
                    else { rest = mo + 1; p0 = mo; t0 = (long long)mo - l; }
                    if (rest < kIndelMinSeq) break;
                    if (pass == 1 && (uint64_t)l > t) break;
                }
                int imm = 0, i;
                for (i = 0; i < rest && (k + imm) <= max_mm; i++) {
                    const int pi = RIGHT ? p0 + i : p0 - i;
                    const long long ti = RIGHT ? t0 + i : t0 - i;
                    const int pb = rd_base4(rdw, pi), tb = tg_base4(tgt, (uint64_t)((long long)t + ti));
                    if (pb > 4 || tb > 4) break;
                    if (pb == tb && pb <= 3) continue;
                    imm++;
                    score -= kIndelMismatch;
                    if ((uint32_t)rest < (uint32_t)(kIndelMinSeq * imm)) break;
                }
                if (i != rest) continue;
                if (score > best.score) {
                    best.score = score; best.is_insert = pass == 0 ? 1 : 0;
                    if (RIGHT) {
                        best.s0_len = mo; best.s0_loci = t; best.s0_mm = k;
                        best.s1_len = rest; best.s1_mm = imm;
                        best.s1_loci = pass == 0 ? t + (uint64_t)mo : t + (uint64_t)mo + (uint64_t)l;
                        best.s1_ofs = pass == 0 ? mo + l : mo;
                    } else {
                        best.s0_len = rest; best.s0_mm = imm; best.s1_mm = k;
                        if (pass == 0) {
                            best.s0_loci = (uint64_t)(uint32_t)(t + (uint64_t)l);
                            best.s1_len = plen - (rest + l); best.s1_loci = best.s0_loci + (uint64_t)rest; best.s1_ofs = rest + l;
                        } else {
                            best.s0_loci = (uint64_t)(uint32_t)(t - (uint64_t)l);
                            best.s1_len = plen - rest; best.s1_loci = best.s0_loci + (uint64_t)rest + (uint64_t)l; best.s1_ofs = rest;
                        }
                    }
                }
            }
        }
    }
    if (del.score == 0 && ins.score == 0) return;
    if (del.score > ins.score) { out = del; out.r = 3; }
    else { out = ins; out.r = 2; }
}

// ExploreSpliceRight / ExploreSpliceLeft (SfxArrayV2.cpp:8437-8940) for one placement of the read at target offset t: beyond the
// anchor core collect the mismatch positions; with >= 8 of them try, for each of the first MaxTotMM+1, to move the rest of the read
// 25 .. max_junct bases further along the target - candidates are pre-filtered by a rolling sum of base codes, compared with the
// reference's strict mismatch budget, scored (GT..AG / CT..AC ends earn a bonus, every 1000 bases of intron cost 10).
constexpr int kJunctSep = 25, kJunctMM = 2, kJunctSeg = 10, kSpliceBonus = 50, kSpliceLenCost = 10;

__device__ __forceinline__ int splice_bonus(bool plus, int d0, int d1, int a0, int a1)
{
    const bool gt_ag = d0 == 2 && d1 == 3 && a0 == 2 && a1 == 0, ct_ac = d0 == 1 && d1 == 3 && a0 == 1 && a1 == 0;
    if (plus) return gt_ag ? kSpliceBonus : (ct_ac ? kSpliceBonus / 2 : 0);
    return ct_ac ? kSpliceBonus : (gt_ag ? kSpliceBonus / 2 : 0);
}

// one base per call from consecutive target positions (ascending when FWD, else descending): the 16-base word is reloaded only
// when the position crosses into the next one - the junction scan below walks up to 100 000 positions per candidate
template <bool FWD>
struct NibStream {
    const uint64_t *__restrict__ tgt;
    uint64_t pos, word;
    __device__ __forceinline__ void start(const uint64_t *__restrict__ t4, uint64_t p) { tgt = t4; pos = p; word = t4[p >> 4]; }
    __device__ __forceinline__ int next()
    {
        const int v = (int)((word >> (60 - 4 * (int)(pos & 15))) & 7);
        if (FWD) { pos++; if ((pos & 15) == 0) word = tgt[pos >> 4]; }
        else { if ((pos & 15) == 0) word = tgt[(pos - 1) >> 4]; pos--; }
        return v;
    }
};

template <bool RIGHT>
__device__ void explore_splice(const uint64_t *__restrict__ rdw, const uint64_t *__restrict__ tgt, int plen, uint64_t t, uint64_t targ_len,
                               int max_junct, int max_mm, int core_len, bool plus, IndelPlacement &out)
{
    out.r = 0; out.score = 0; out.is_insert = 0; out.s0_loci = 0; out.s1_loci = 0; out.s0_len = 0; out.s0_mm = 0; out.s1_len = 0; out.s1_mm = 0; out.s1_ofs = 0;
    if (RIGHT) { if (t + (uint64_t)plen + kJunctSep > targ_len) return; }
    else if (t < (uint64_t)(kJunctSep + kJunctSeg)) return;
    if (max_mm > kJunctMM) max_mm = kJunctMM;
    const int pe = plen - 1;
    // probe / target base i positions away from the scan origin (5' end going right, or 3' end going left)
    auto P = [&](int i) -> int { return rd_base4(rdw, RIGHT ? i : pe - i); };
    auto T = [&](long long i) -> int { return tg_base4(tgt, RIGHT ? t + (uint64_t)i : (uint64_t)((long long)t + pe - i)); };
    int mm_ofs[kJunctMM * 5 + 2];
    int n_mm = 0, pb = 0, tb = 0;
    const int lim = kJunctMM * 5;
    for (int i = core_len; i < plen && n_mm <= lim; i++) {
        pb = P(i); tb = T(i);
        if (tb > 4 || pb > 4) return;
        if (pb == tb && pb <= 3) continue;
        mm_ofs[n_mm++] = i;
    }
    if (n_mm < kJunctMM * 4 || kJunctSeg > plen - mm_ofs[0]) {
        if (n_mm > max_mm) return;
        out.r = 1; out.s0_len = plen; out.s0_loci = t; out.s0_mm = n_mm;
        out.score = kIndelBase + plen * kIndelMatch - n_mm * kIndelMismatch;
        return;
    }
    const int tot = n_mm < max_mm ? n_mm : max_mm;
    if (RIGHT && tot < 1) return;
    {
        const int from = RIGHT ? mm_ofs[tot - 1] : mm_ofs[tot];
        for (int i = 0; i < kJunctSep + kJunctSeg; i++)
            if (T((long long)from + i) > 4) return;
    }
    IndelPlacement cur = out;
    for (int k = 0; k <= tot && kJunctSeg < plen - mm_ofs[k]; k++) {
        if (cur.score >= kIndelMaxScore) break;
        const int seg_len = plen - mm_ofs[k], mo = mm_ofs[k];
        const int hash_diff = 4 * (max_mm - k);
        int probe_hash = 100000;
        for (int i = 0; i < seg_len; i++) probe_hash += P(mo + i);
        const int min_hash = probe_hash - hash_diff, max_hash = probe_hash + hash_diff;
        int targ_hash = 100000, i;
        for (i = 0; i < seg_len - 1; i++) {
            const int b = T((long long)mo + kJunctSep + i);
            if (b > 4) break;
            targ_hash += b;
        }
        if (i < seg_len - 1) break;
        // the two ends of the sliding window as streams (scan coordinate i is target t + i for RIGHT, t + pe - i for LEFT)
        NibStream<RIGHT> s_te, s_ts;
        {
            const long long te0 = (long long)mo + kJunctSep + seg_len - 1, ts0 = (long long)mo + kJunctSep;
            s_te.start(tgt, RIGHT ? t + (uint64_t)te0 : (uint64_t)((long long)t + pe - te0));
            s_ts.start(tgt, RIGHT ? t + (uint64_t)ts0 : (uint64_t)((long long)t + pe - ts0));
        }
        for (int gap = kJunctSep; gap < max_junct - seg_len; gap++) {
            const long long ts = (long long)mo + gap;                                // start of the moved segment, scan coordinates
            if ((tb = s_te.next()) > 4) break;
            targ_hash += tb;
            const bool in_range = !(targ_hash < min_hash || targ_hash > max_hash);
            targ_hash -= s_ts.next();
            if (!in_range) continue;
            if (RIGHT) { if ((uint32_t)(targ_len - (t + (uint64_t)mo + (uint64_t)gap + 1)) < (uint32_t)seg_len) break; }
            else if ((uint32_t)(t - (uint64_t)gap) < 1u) break;
            int cmm = 0;
            for (i = 0; i < seg_len && (k + cmm) < max_mm; i++) {
                pb = P(mo + i); tb = T(ts + i);
                if (pb > 4 || tb > 4) break;
                if (pb == tb && pb <= 3) continue;
                cmm++;
            }
            if (i != seg_len) {
                if (pb > 4 || tb > 4) break;
                continue;
            }
            int score = kIndelBase + plen * kIndelMatch - ((k + cmm) * kIndelMismatch + (gap / 1000) * kSpliceLenCost);
            // donor = first two intron bases after the kept part, acceptor = last two before the moved part (target order)
            if (RIGHT) score += splice_bonus(plus, T(mo), T(mo + 1), T(ts - 1), T(ts - 2));
            else score += splice_bonus(plus, T(ts - 1), T(ts - 2), T(mo), T(mo + 1));
            if (score > cur.score) {
                cur.score = score; cur.r = 3;
                if (RIGHT) {
                    cur.s0_len = mo; cur.s0_loci = t; cur.s0_mm = k;
                    cur.s1_len = seg_len; cur.s1_loci = t + (uint64_t)mo + (uint64_t)gap; cur.s1_mm = cmm; cur.s1_ofs = mo;
                } else {
                    cur.s0_len = seg_len; cur.s0_loci = t - (uint64_t)gap; cur.s0_mm = cmm;
                    cur.s1_len = mo; cur.s1_loci = cur.s0_loci + (uint64_t)seg_len + (uint64_t)gap; cur.s1_mm = k; cur.s1_ofs = seg_len;
                }
            }
        }
    }
    if (cur.score == 0) return;
    out = cur;
}

template <bool WIDE>
__global__ void __launch_bounds__(256) k_indel(DevIndex ix, DevAlignCfg cfg, DevBatch b, const uint32_t *__restrict__ list, uint32_t n_list,
                                               int max_indel, int max_junct, int keep_state, uint32_t *__restrict__ cursor, bk_seg2 *__restrict__ seg2)
{
    __shared__ LdsEntries s_le;
    lds_entries_load(s_le, ix);
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = (1ULL << lane) - 1;
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(cursor, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_list) break;
        const uint32_t r = list[item];
        const int len = (int)b.lens[r];
        const ReadPlan p = make_plan(len, cfg);
        const int core = p.core_len * 2 < (len - 1) / 2 ? p.core_len * 2 : (len - 1) / 2;
        const int max_mm = p.max_tot_mm > kIndelMaxMM ? kIndelMaxMM : p.max_tot_mm;       // cMaxMicroInDelMM == cMaxJunctAlignMM == 2
        if (core < 1) continue;
        // AlignReads: LocateInDels first, LocateSpliceJuncts only if that returned nothing (SfxArrayV2.cpp:7722-7748).  Both write
        // their caller's LowHitInstances / LowMMCnt / NxtLowMMCnt even when they then return "nothing" because the best placement
        // is ambiguous (:7655-7659, :7340-7344); the chimeric call that may follow starts from those values (left_*).
        int left_inst = 0, left_mm = 0;
        bool aligned = false;
        for (int mode = max_indel > 0 ? 0 : 1; mode < 2; mode++) {
            if (mode == 1 && max_junct <= 0) break;
            const bool splice = mode == 1;
            left_inst = 0; left_mm = 0;
            // best placement so far (wave-uniform)
            int best_score = 0, best_inst = 0, b_insert = 0, b_kind = 0, b_s0_len = 0, b_s0_mm = 0, b_s1_len = 0, b_s1_mm = 0, b_s1_ofs = 0, b_strand = '+';
            uint64_t b_s0 = 0, b_s1 = 0;
            bool done = false;
            const int s0 = cfg.align_strand == 2 ? 1 : 0, s1 = cfg.align_strand == 1 ? 0 : 1;
            for (int st = s0; st <= s1 && !done; st++) {
                const uint64_t *rdw = b.rd4 + ((uint64_t)r * 2 + st) * b.wpr;
                for (int phase = 0; phase < 2; phase++) {
                    const int ofs = phase == 0 ? 0 : len - core;
                    uint64_t first = 0, n = 0;
                    search_core<WIDE>(ix, rdw, ofs, core, ~0ULL >> 1, first, n);       // every lane the same search
                    uint32_t iter = 0;
                    bool copies_checked = false;
                    for (uint64_t j0 = 0; j0 < n; j0 += 64) {
                        const uint64_t j = j0 + lane;
                        const bool active = j < n;
                        const uint64_t loci = active ? sa_get<WIDE>(ix, first + j) : 0;
                        const uint64_t t = loci - (uint64_t)ofs;
                        bool valid = active && loci >= (uint64_t)ofs;
                        int e = -1;
                        if (valid && splice) valid = t + (uint64_t)len < ix.n;                                   // :7132
                        if (valid) {
                            e = find_entry_lds(s_le, ix, loci);                   // MapChunkHit2Entry of the ANCHOR position (:7497 / :7135)
                            if (splice) valid = e >= 0 && t >= ix.ent_start[e] && t + (uint64_t)len <= ix.ent_end[e];
                            else valid = e >= 0 && t >= ix.ent_start[e] && t + (uint64_t)len - 1 <= ix.ent_end[e] && t + (uint64_t)len <= ix.n;
                        }
                        const uint64_t newmask = __ballot(valid);
                        const uint32_t pre = (uint32_t)__popcll(newmask & lt_mask);
                        const uint32_t iter_before = iter + pre;
                        bool stop = active && cfg.max_iter && iter_before >= (uint32_t)cfg.max_iter;
                        // the splice walk also ends at a suffix too close to the end of the concatenation (:7087)
                        if (splice && active && j > 0 && loci + (uint64_t)(phase == 0 ? len : core) >= ix.n) stop = true;
                        uint64_t cutoff = n;
                        const uint64_t stopmask = __ballot(stop);
                        if (stopmask) cutoff = j0 + (uint64_t)(__ffsll((unsigned long long)stopmask) - 1);
                        if (!copies_checked) {
                            const bool chk = active && j > 0 && iter_before == 100;
                            const uint64_t chkmask = __ballot(chk);
                            if (chkmask) {
                                const uint64_t jc = j0 + (uint64_t)(__ffsll((unsigned long long)chkmask) - 1);
                                if (jc < cutoff) {
                                    copies_checked = true;
                                    const uint64_t num_copies = n - jc + 2;
                                    if (cfg.max_iter && (uint32_t)num_copies > (uint32_t)cfg.max_iter) cutoff = jc;
                                }
                            }
                        }
                        const bool proc = valid && j < cutoff;
                        iter += (uint32_t)__popcll(__ballot(proc));
                        IndelPlacement pl;
                        pl.r = 0; pl.score = 0;
                        if (proc && !splice) {
                            const uint32_t seq_left = (uint32_t)(ix.ent_end[e] + 1 - t);      // SeqLen - (TargOfs - StartOfs)
                            if (phase == 0) explore_indel<true>(rdw, ix.tgt4, len, t, seq_left, max_indel, max_mm, pl);
                            else explore_indel<false>(rdw, ix.tgt4, len, t, seq_left, max_indel, max_mm, pl);
                        } else if (proc) {
                            if (phase == 0) {
                                int limit = (int)(ix.n - t);
                                if (limit > kJunctSep + kJunctSeg) {
                                    limit -= kJunctSep + kJunctSeg;
                                    if (limit > max_junct) limit = max_junct;
                                    explore_splice<true>(rdw, ix.tgt4, len, t, ix.n, limit, max_mm, core, st == 0, pl);
                                }
                            } else if (t >= (uint64_t)(uint32_t)(ofs + kJunctSeg)) {
                                int limit = (int)(t < (uint64_t)(uint32_t)max_junct ? t : (uint64_t)(uint32_t)max_junct);
                                if (limit >= kJunctSep + kJunctSeg) {
                                    limit -= kJunctSeg;
                                    explore_splice<false>(rdw, ix.tgt4, len, t, ix.n, limit, max_mm, core, st == 0, pl);
                                }
                            }
                        }
                        // fold this round's placements in suffix-array order (:7517-7561 / :7160-7200)
                        uint64_t hm = __ballot(proc && pl.r > 0);
                        while (hm) {
                            const int src = __ffsll((unsigned long long)hm) - 1;
                            hm &= hm - 1;
                            const int sc = __shfl(pl.score, src);
                            if (sc < best_score) continue;
                            const uint64_t c_s0 = __shfl(pl.s0_loci, src);
                            if (sc == best_score) {
                                if (b_s0 == c_s0) continue;
                                if (++best_inst > 1) continue;
                            } else
                                best_inst = 0;
                            best_score = sc; b_s0 = c_s0; b_s1 = __shfl(pl.s1_loci, src);
                            b_s0_len = __shfl(pl.s0_len, src); b_s0_mm = __shfl(pl.s0_mm, src); b_s1_len = __shfl(pl.s1_len, src);
                            b_s1_mm = __shfl(pl.s1_mm, src); b_s1_ofs = __shfl(pl.s1_ofs, src); b_insert = __shfl(pl.is_insert, src);
                            b_kind = __shfl(pl.r, src) > 1 ? (splice ? 4 : 1) : 0;
                            b_strand = st ? '-' : '+';
                            best_inst++;
                        }
                        if (cutoff < j0 + 64) break;
                    }
                    if (best_inst >= 1 && best_score >= kIndelMaxScore) { done = true; break; }
                }
            }
            if (best_inst == 0) continue;                   // nothing: on to the next mode
            if (best_score > kIndelMaxScore) best_score = kIndelMaxScore;
            // offsets -> entry + position; LocateInDels insists on one entry for both segments (a placement without a second
            // segment looks up offset 0 there), LocateSpliceJuncts only looks the second one up when there is one
            int e0 = -1, e1 = -1;
            if (lane == 0) {
                e0 = find_entry(ix, b_s0);
                e1 = (splice && b_s1 == 0) ? e0 : find_entry(ix, b_s1);
            }
            e0 = __shfl(e0, 0); e1 = __shfl(e1, 0);
            bool ok = e0 >= 0 && e1 >= 0;
            if (ok && !splice) ok = ix.ent_id[e0] == ix.ent_id[e1];
            if (!ok) continue;
            if (best_inst > 1) {                            // ambiguous: reported as nothing, but the counts stay behind
                left_inst = splice ? best_inst : 1;
                left_mm = b_s0_mm + b_s1_mm;
                continue;
            }
            aligned = true;
            if (lane == 0) {
                bk_hit h;
                h.chrom_id = ix.ent_id[e0]; h.match_loci = (uint32_t)(b_s0 - ix.ent_start[e0]); h.match_len = (uint16_t)b_s0_len;
                h.low_hit_instances = 1; h.rslt = BK_HR_HITS; h.nar = BK_NAR_ACCEPTED; h.strand = (uint8_t)b_strand;
                h.low_mm = (int8_t)(b_s0_mm + b_s1_mm); h.nxt_low_mm = (int8_t)(b_s0_mm + b_s1_mm + 2); h.num_hits = 1;
                h.mismatches = (uint8_t)b_s0_mm; h.flags = (uint8_t)(((p.n_phases) << 1) | 1);
                b.out[r] = h;
                bk_seg2 g;
                g.match_loci = b_s1 > 0 ? (uint32_t)(b_s1 - ix.ent_start[e1]) : 0u; g.match_len = (uint16_t)b_s1_len; g.read_ofs = (uint16_t)b_s1_ofs;
                g.mismatches = (uint8_t)b_s1_mm; g.flags = (uint8_t)(b_kind | (b_insert ? 2 : 0)); g.score = (uint16_t)best_score;
                seg2[r] = g;
            }
            break;                                          // aligned: no further mode
        }
        if (!aligned && keep_state && left_inst > 0 && lane == 0) {
            bk_seg2 g;
            g.match_loci = 0; g.read_ofs = 0; g.score = 0; g.flags = 0x80;          // not a placement: state for the chimeric call
            g.match_len = (uint16_t)(left_inst > 65535 ? 65535 : left_inst); g.mismatches = (uint8_t)left_mm;
            seg2[r] = g;
        }
    }
}

// reads AlignReads' phases left without any result (candidates for the -a pass)
// Lean batches keep 4 bit/base rows only for the reads with an N.  The kernels of the general family (k_heavy in all its forms,
// k_indel) read rd4 rows; the few reads they are handed get theirs here, widened from the 2-bit rows: one lane per 16-base word.
__global__ void __launch_bounds__(256) k_expand_rd4(DevBatch b, const uint32_t *__restrict__ list, uint32_t n_list)
{
    const uint32_t per_read = 2 * b.wpr;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t item = tid / per_read;
    if (item >= n_list) return;
    const uint32_t rem = (uint32_t)(tid - item * per_read);
    const uint32_t r = list ? list[item] : (uint32_t)item;
    if (b.rmeta[r] & kReadHasN) return;                          // its rows were written by the read preparation
    const uint32_t st = rem >= b.wpr ? 1 : 0, w = rem - st * b.wpr;
    uint64_t v = 0;
    if (w < b.nw) {
        const uint64_t x = b.rd2[((uint64_t)r * 2 + st) * (b.nw / 2) + (w >> 1)];
        v = spread2to4((w & 1) ? (uint32_t)x : (uint32_t)(x >> 32));
    }
    b.rd4[((uint64_t)r * 2 + st) * b.wpr + w] = v;
}

static void expand_rd4(const DevBatch &b, const uint32_t *list, uint32_t n_list, hipStream_t s)
{
    if (b.rd2 == nullptr || !n_list) return;
    const uint64_t threads = (uint64_t)n_list * 2 * b.wpr;
    hipLaunchKernelGGL(k_expand_rd4, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, b, list, n_list);
}

__global__ void __launch_bounds__(256) k_unaligned_list(const bk_hit *__restrict__ out, uint32_t n, uint32_t *__restrict__ list, uint32_t *__restrict__ cnt)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const bk_hit h = out[r];
    if (h.nar == BK_NAR_NOHIT && h.rslt == BK_HR_NONE) list[atomicAdd(cnt, 1u)] = r;
}

void launch_unaligned_list(const bk_hit *out, uint32_t n, uint32_t *list, uint32_t *cnt, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_unaligned_list, dim3((n + 255) / 256), dim3(256), 0, s, out, n, list, cnt);
}

void launch_indel(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, uint32_t n, int max_indel, int max_junct, int keep_state, uint32_t *list,
                  uint32_t *list_cnt_dev, uint32_t *list_cnt_host, uint32_t *cursor, bk_seg2 *seg2, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_unaligned_list, dim3((n + 255) / 256), dim3(256), 0, s, b.out, n, list, list_cnt_dev);
    (void)hipMemcpyAsync(list_cnt_host, list_cnt_dev, 4, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    const uint32_t n_list = *list_cnt_host;
    if (!n_list) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>(((uint64_t)n_list + 3) / 4, 8192);
    expand_rd4(b, list, n_list, s);
    if (ix.sa_hi) hipLaunchKernelGGL((k_indel<true>), dim3(blocks), dim3(256), 0, s, ix, cfg, b, list, n_list, max_indel, max_junct, keep_state, cursor, seg2);
    else hipLaunchKernelGGL((k_indel<false>), dim3(blocks), dim3(256), 0, s, ix, cfg, b, list, n_list, max_indel, max_junct, keep_state, cursor, seg2);
}

__global__ void k_fill_u64(unsigned long long *__restrict__ p, uint64_t n, unsigned long long v)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

__global__ void __launch_bounds__(256) k_max_len(const uint32_t *__restrict__ lens, uint32_t n, uint32_t *__restrict__ out)
{
    __shared__ uint32_t s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    uint32_t v = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w = lens[i];
        v = w > v ? w : v;
    }
    for (int off = 32; off > 0; off >>= 1) { uint32_t w = __shfl_down(v, off); v = w > v ? w : v; }
    if ((threadIdx.x & 63) == 0 && v) atomicMax(&s_max, v);
    __syncthreads();
    if (threadIdx.x == 0 && s_max) atomicMax(out, s_max);
}

// per-sequence counts of accepted reads (feeds the -O CSV and the cross-rank reduction): one pass over
// a finished chunk's hit records with a block-private LDS histogram, so that the alignment kernels
// carry no same-address global atomics (47 M of them per 50 M reads cost ~30 ms inside k_light)
constexpr uint32_t kHistLds = 4096;

__global__ void __launch_bounds__(256) k_count_seqs(const bk_hit *__restrict__ out, uint32_t n, const uint32_t *__restrict__ id2idx,
                                                    uint32_t n_ent, unsigned long long *__restrict__ counts)
{
    __shared__ uint32_t s_hist[kHistLds];
    const uint32_t nl = n_ent < kHistLds ? n_ent : kHistLds;
    for (uint32_t i = threadIdx.x; i < nl; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (out[i].nar != BK_NAR_ACCEPTED) continue;
        uint32_t e = id2idx[out[i].chrom_id];
        if (e < nl) atomicAdd(&s_hist[e], 1u);
        else atomicAdd(&counts[e], 1ULL);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nl; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&counts[i], (unsigned long long)s_hist[i]);
}

// ------------------------------------------------------------------------------------------------
// K4: paired-end association after the SE pass - CAligner::ProcessPairedEnds (biokanga/Aligner.cpp:
// 3055-3489), AcceptProvPE / PEInsertSize (:2726-2850) and the orphan recovery of
// CSfxArrayV3::AlignPairedRead (libbiokanga/SfxArrayV2.cpp:8247-8433) whose AdaptiveTrim (:5482-5682)
// is called with MinTrimLen == read length, i.e. it accepts a window iff: the first and the last 3
// bases match, some run of >= 8 bases matches, and (MaxMM+1.0)/100.0 > mismatches/length (double
// compare, MaxMM = the -s value, <= 15); the FIRST window with the fewest mismatches wins.
// hits[2i] / hits[2i+1] = PE1 / PE2; bk_hit.flags bit 7 = FlgPEAligned.

struct DevPE { int pe_mode, min_len, max_len, pair_strand; };

enum { NAR_CHROMFILT = 11, NAR_PEINSERTMIN = 13, NAR_PEINSERTMAX = 14, NAR_PENOHIT = 15, NAR_PESTRAND = 16, NAR_PECHROM = 17,
       NAR_PEUNALIGN = 18 };

__device__ __forceinline__ int pe_insert_size(const DevPE &pe, uint8_t s1, uint32_t st1, uint32_t en1, uint8_t s2, uint32_t st2, uint32_t en2)
{
    int frag;
    if ((pe.pair_strand && s1 != s2) || (!pe.pair_strand && s1 == s2)) return -1;
    if (s1 == '+') frag = 1 + (int)en2 - (int)st1;
    else frag = 1 + (int)en1 - (int)st2;
    if (frag < 0) return -1;
    if (frag < pe.min_len) return -6;
    if (frag > pe.max_len) return -7;
    return frag;
}

__device__ __forceinline__ bool pe_unaligned(const bk_hit &h) { return h.nar == BK_NAR_NS || h.nar == BK_NAR_NOHIT || h.nar == BK_NAR_UNALIGNED; }

// the tail of ProcessPairedEnds once no PE could be formed (:3440-3480)
__device__ __forceinline__ void pe_finish(const DevPE &pe, bk_hit &f, bk_hit &r)
{
    if (!(pe.pe_mode == 3 || pe.pe_mode == 4)) {
        f.num_hits = 0; f.low_hit_instances = 0; r.num_hits = 0; r.low_hit_instances = 0;
        if (f.nar == BK_NAR_ACCEPTED) f.nar = NAR_PENOHIT;
        if (r.nar == BK_NAR_ACCEPTED) r.nar = NAR_PENOHIT;
        return;
    }
    bk_hit *hh[2] = {&f, &r};
    for (int k = 0; k < 2; k++) {
        bk_hit &h = *hh[k];
        if (h.num_hits != 1) {
            h.num_hits = 0; h.low_hit_instances = 0;
            if (h.nar == BK_NAR_ACCEPTED) h.nar = NAR_PEUNALIGN;
        } else
            h.nar = BK_NAR_ACCEPTED;
    }
}

// AdjStartLoci / AdjEndLoci of a read's Seg[0] (Aligner.cpp:1528-1544): a chimeric placement (bk_seg2.flags bit 3) carries its end trims
__device__ __forceinline__ void pe_adj_loci(const bk_hit &h, const bk_seg2 *__restrict__ seg2, uint32_t idx, uint32_t &start, uint32_t &end)
{
    uint32_t tl = 0, tr = 0;
    if (seg2 != nullptr) {
        const bk_seg2 g = seg2[idx];
        if (g.flags & 8) { tl = g.match_len; tr = g.read_ofs; }
    }
    if (h.strand == '+') { start = h.match_loci + tl; end = h.match_loci + (h.match_len - tr - 1); }
    else { start = h.match_loci + tr; end = h.match_loci + (h.match_len - tl - 1); }
}

__global__ void __launch_bounds__(256) k_pe_classify(DevPE pe, bk_hit *__restrict__ hits, uint32_t n_pairs,
                                                      uint32_t *__restrict__ orphans, uint32_t *__restrict__ orphan_cnt,
                                                      const bk_seg2 *__restrict__ seg2)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    bk_hit f = hits[2 * i], r = hits[2 * i + 1];
    f.flags &= 0x7f; r.flags &= 0x7f;
    bool f_un = pe_unaligned(f), r_un = pe_unaligned(r);
    bool store = true;
    if (!(f.nar == BK_NAR_ACCEPTED || r.nar == BK_NAR_ACCEPTED)) {
        // neither end accepted: nothing to pair
    } else if (pe.pe_mode == 2 && (f_un || r_un)) {
        f.num_hits = 0; f.low_hit_instances = 0; r.num_hits = 0; r.low_hit_instances = 0;
        if (f.nar == BK_NAR_ACCEPTED) f.nar = NAR_PENOHIT;
        if (r.nar == BK_NAR_ACCEPTED) r.nar = NAR_PENOHIT;
    } else {
        bool paired = false, stop = false;
        if (f.nar == BK_NAR_ACCEPTED && r.nar == BK_NAR_ACCEPTED) {
            int frag = 0;
            if (f.num_hits == 1 && r.num_hits == 1) {
                if (f.chrom_id != r.chrom_id) frag = -2;
                else {
                    uint32_t fs, fe, rs, re;
                    pe_adj_loci(f, seg2, 2 * i, fs, fe);
                    pe_adj_loci(r, seg2, 2 * i + 1, rs, re);
                    frag = pe_insert_size(pe, f.strand, fs, fe, r.strand, rs, re);
                }
            }
            if (frag > 0) { f.flags |= 0x80; r.flags |= 0x80; paired = true; }
            else {
                switch (frag) {
                case -1: f.nar = r.nar = NAR_PESTRAND; break;
                case -2: f.nar = r.nar = NAR_PECHROM; break;
                case -6: f.nar = r.nar = NAR_PEINSERTMIN; break;
                case -7: f.nar = r.nar = NAR_PEINSERTMAX; break;
                }
                if (pe.pe_mode == 2) {
                    f.num_hits = 0; f.low_hit_instances = 0; r.num_hits = 0; r.low_hit_instances = 0;
                    if (f.nar == BK_NAR_ACCEPTED) f.nar = NAR_PENOHIT;
                    if (r.nar == BK_NAR_ACCEPTED) r.nar = NAR_PENOHIT;
                    stop = true;
                }
            }
        }
        if (!paired && !stop) {
            bool try_orphan = (pe.pe_mode == 1 || pe.pe_mode == 3) && ((f.num_hits == 1 && !r_un) || (r.num_hits == 1 && !f_un));
            if (try_orphan) orphans[atomicAdd(orphan_cnt, 1u)] = i;      // finished by k_pe_orphan
            else pe_finish(pe, f, r);
        }
    }
    if (store) { hits[2 * i] = f; hits[2 * i + 1] = r; }
}

// AdaptiveTrim(full length) acceptance of the read (packed words rdw) against the target at t
__device__ __forceinline__ bool pe_window_ok(const uint64_t *__restrict__ rdw, int len, const uint64_t *__restrict__ tgt, uint64_t t,
                                             int max_mm, int &mm_out)
{
    int mm = 0, run = 0;
    bool have8 = false, first3 = false;
    for (int i = 0; i < len; i += 16) {
        int nv = len - i < 16 ? len - i : 16;
        uint64_t x = (nib16(rdw, i) ^ nib16(tgt, t + i)) & top_mask(nv);
        uint64_t f = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ULL;
        uint32_t bits = flags_to_bits16(f);                         // bit k = base i+k mismatches
        uint32_t valid = nv >= 16 ? 0xFFFFu : ((1u << nv) - 1);
        uint32_t g = ~bits & valid;                                 // bit k = base matches
        mm += __popc(bits);
        if (i == 0) first3 = (g & 7u) == 7u;
        // runs of >= 8 matches: inside the word, or continuing the run carried from previous words
        uint32_t y = g & (g >> 1);
        y &= y >> 2;
        y &= y >> 4;
        int lead = __ffs((int)(~g & 0x1FFFFu)) - 1;                 // matches at the start of this word (0..16)
        if (lead > nv) lead = nv;
        if (y != 0 || run + lead >= 8) have8 = true;
        if (g == valid) run += nv;
        else run = nv - (32 - __clz((int)(~g & valid)));            // matches after the last mismatch of this word
    }
    mm_out = mm;
    if (len < 25 || len > 2048 || max_mm > 15) return false;         // AdaptiveTrim parameter validation -> eBSFerrParams
    if (!have8 || !first3 || run < 3) return false;
    if (mm > 0 && max_mm == 0) return false;
    if ((max_mm + 1.0) / 100.0 <= (double)mm / (double)len) return false;
    return true;
}

// One candidate window of AlignPairedRead.  ATW == 0: the partner must fit whole (MinChimericLen == 0, MinPutLen = ReadLen); otherwise
// AdaptiveTrim may cut its ends down to min_put bases (SfxArrayV2.cpp:8327-8330,8400-8470).  The reference raises MinPutLen to the length of
// every placement it takes and hands that to the next AdaptiveTrim call; a call with the initial MinPutLen returns the same stretch whenever
// that stretch is at least as long as the raised limit and nothing acceptable otherwise (the limit only removes shorter candidates from
// AdaptiveTrim's scan), so the outcome of the scan is the first window with the longest stretch and, among those, the fewest mismatches.
// key: smaller = better; ~0 = not a candidate.  Bits 52.. = 4095 - trimmed length, bits 40..51 = mismatches, low 40 bits = scan order.
template <int ATW>
__device__ __forceinline__ unsigned long long pe_window_key(const uint64_t *__restrict__ rdw, int len, const uint64_t *__restrict__ tgt, uint64_t t,
                                                            int max_mm, int min_put, unsigned long long order, int &t5, int &t3)
{
    t5 = 0; t3 = 0;
    if constexpr (ATW == 0) {
        int mm;
        if (!(pe_window_ok(rdw, len, tgt, t, max_mm, mm) && mm <= max_mm)) return ~0ULL;
        return ((unsigned long long)(4095 - len) << 52) | ((unsigned long long)mm << 40) | order;
    } else {
        int mm;
        const int r = adaptive_trim_dev<ATW>(rdw, tgt, t, len, min_put, max_mm, 3, mm, t5, t3);
        if (r < min_put || r == 0 || (r == min_put && mm > max_mm)) return ~0ULL;
        return ((unsigned long long)(4095 - r) << 52) | ((unsigned long long)mm << 40) | order;
    }
}

template <int ATW>
__global__ void __launch_bounds__(256) k_pe_orphan(DevIndex ix, DevAlignCfg cfg, DevPE pe, DevBatch b, bk_hit *__restrict__ hits,
                                                    const uint32_t *__restrict__ list, uint32_t n_list, uint32_t *__restrict__ cursor,
                                                    bk_seg2 *__restrict__ seg2, int min_chim)
{
    const int lane = threadIdx.x & 63;
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(cursor, 1u);
        item = __shfl(item, 0);
        if (item >= n_list) break;
        const uint32_t i = list[item];
        bk_hit f = hits[2 * i], r = hits[2 * i + 1];
        const bool f_un = pe_unaligned(f), r_un = pe_unaligned(r);
        bool done = false;
        for (int anchor = 0; anchor < 2 && !done; anchor++) {
            bk_hit &a = anchor == 0 ? f : r;
            bk_hit &o = anchor == 0 ? r : f;
            const bool o_un = anchor == 0 ? r_un : f_un;
            if (!(a.num_hits == 1 && !o_un)) continue;
            const uint32_t oi = 2 * i + (anchor == 0 ? 1 : 0);
            bool b3, anti;
            if (anchor == 0) {
                b3 = a.strand == '+';
                anti = pe.pair_strand ? (a.strand != '+') : (a.strand == '+');
            } else {
                b3 = a.strand == '+'; anti = a.strand == '+';
                if (pe.pair_strand) { b3 = !b3; anti = !anti; }
            }
            uint32_t a_start, a_end;
            pe_adj_loci(a, seg2, 2 * i + (anchor == 0 ? 0 : 1), a_start, a_end);
            const int read_len = (int)b.lens[oi];
            const int max_allowed = cfg.max_subs;
            const int min_put = ATW != 0 && min_chim > 0 ? (read_len * min_chim + 50) / 100 : read_len;
            // AlignPairedRead set-up (:8270-8330)
            if (pe.min_len < read_len || pe.min_len > pe.max_len) continue;
            if (a.chrom_id < 1 || a.chrom_id > ix.max_id) continue;
            const uint32_t a_ent = ix.id2idx[a.chrom_id];                  // EntryIDs need not be 1..n in file order
            if (a_ent >= ix.n_ent) continue;
            const uint64_t c_start = ix.ent_start[a_ent];
            const uint32_t targ_len = (uint32_t)(ix.ent_end[a_ent] - c_start + 1);
            int targ_loci;
            if (b3) { targ_loci = (int)a_start; if ((uint32_t)(targ_loci + pe.min_len) > targ_len) continue; }
            else { targ_loci = (int)a_end; if (targ_loci < pe.min_len || (uint32_t)targ_loci >= targ_len) continue; }
            uint32_t start_put, end_put;
            if (b3) {
                start_put = (uint32_t)(targ_loci + pe.min_len);
                if (start_put + (uint32_t)min_put >= targ_len) continue;
                end_put = (uint32_t)(targ_loci + pe.max_len);
            } else {
                start_put = a_end < (uint32_t)pe.max_len ? 0 : a_end - (uint32_t)pe.max_len;
                end_put = a_end - (uint32_t)pe.min_len;
            }
            const uint64_t *rdw = b.rd4 + ((uint64_t)oi * 2 + (anti ? 1 : 0)) * b.wpr;
            // best = smallest pe_window_key
            unsigned long long best = ~0ULL;
            uint32_t best_loci = 0;
            int best_t5 = 0, best_t3 = 0;
            if (end_put - start_put >= 1000) {
                // cores of the read located through the suffix array (IterateExactsRange, :3382-3474)
                int match_len = read_len - 1;
                int m = cfg.max_subs == 0 ? 0 : (int)(0.5 + (double)(match_len * cfg.max_subs) / 100.0);
                if (cfg.max_subs != 0 && m < 1) m = 1;
                if (m > 63) m = 63;
                int core_len = read_len / (cfg.mm_delta == 1 ? m + 1 : m + 2);
                if (core_len < cfg.min_core_len) core_len = cfg.min_core_len;
                int core_delta = read_len / cfg.slides_per100 - 1;
                if (core_delta < core_len) core_delta = core_len;
                unsigned long long order = 0;
                for (int core_ofs = 0; core_ofs + core_len <= read_len; core_ofs += core_delta) {
                    uint64_t first, n;
                    if (ix.sa_hi) search_core<true>(ix, rdw, core_ofs, core_len, ~0ULL >> 1, first, n);
                    else search_core<false>(ix, rdw, core_ofs, core_len, ~0ULL >> 1, first, n);
                    first = uniform64(first); n = uniform64(n);
                    for (uint64_t j0 = 0; j0 < n; j0 += 64) {
                        uint64_t j = j0 + lane;
                        unsigned long long key = ~0ULL;
                        uint32_t loci = 0;
                        int t5 = 0, t3 = 0;
                        if (j < n) {
                            uint64_t pos = ix.sa_hi ? sa_get<true>(ix, first + j) : sa_get<false>(ix, first + j);
                            if (pos >= c_start && pos <= ix.ent_end[a_ent]) {
                                uint32_t hit = (uint32_t)(pos - c_start);
                                if (hit >= start_put && hit <= end_put && (uint32_t)core_ofs <= hit &&
                                    (hit + (uint32_t)read_len - (uint32_t)core_ofs) < targ_len) {
                                    key = pe_window_key<ATW>(rdw, read_len, ix.tgt4, c_start + hit - (uint32_t)core_ofs, max_allowed, min_put, order + j, t5, t3);
                                    loci = hit - (uint32_t)core_ofs;
                                }
                            }
                        }
                        unsigned long long k2 = key;
                        for (int off = 32; off > 0; off >>= 1) { unsigned long long q = __shfl_xor(k2, off); k2 = q < k2 ? q : k2; }
                        if (k2 != ~0ULL && (k2 >> 40) < (best >> 40)) {      // strictly better than the best so far
                            int src = __ffsll((unsigned long long)__ballot(key == k2)) - 1;
                            best = k2;
                            best_loci = __shfl(loci, src);
                            best_t5 = __shfl(t5, src);
                            best_t3 = __shfl(t3, src);
                        }
                    }
                    order += n;
                }
            } else {
                for (uint32_t h0 = start_put; h0 <= end_put; h0 += 64) {
                    uint32_t hit = h0 + (uint32_t)lane;
                    unsigned long long key = ~0ULL;
                    int t5 = 0, t3 = 0;
                    if (hit <= end_put && hit >= h0) key = pe_window_key<ATW>(rdw, read_len, ix.tgt4, c_start + hit, max_allowed, min_put, hit, t5, t3);
                    unsigned long long k2 = key;
                    for (int off = 32; off > 0; off >>= 1) { unsigned long long q = __shfl_xor(k2, off); k2 = q < k2 ? q : k2; }
                    if (k2 != ~0ULL && (k2 >> 40) < (best >> 40)) {
                        best = k2;
                        best_loci = (uint32_t)(k2 & 0xFFFFFFFFFFULL);
                        const int src = (int)(best_loci - h0);
                        best_t5 = __shfl(t5, src);
                        best_t3 = __shfl(t3, src);
                    }
                    if (h0 + 64 < h0) break;
                }
            }
            if (best == ~0ULL) continue;
            const int mm = (int)((best >> 40) & 0xFFF);
            if (mm > max_allowed) continue;                                  // (a longer stretch may have displaced an acceptable one, :8472)
            const int trimmed_len = 4095 - (int)(best >> 52);
            const uint8_t h_strand = anti ? '-' : '+';
            // AdjStartLoci / AdjEndLoci of the placement: the trims are those of the sequence as matched, i.e. in target direction
            const uint32_t h_start = best_loci + (uint32_t)best_t5, h_end = best_loci + (uint32_t)(read_len - best_t3) - 1;
            int frag;
            if (anchor == 0) frag = pe_insert_size(pe, a.strand, a_start, a_end, h_strand, h_start, h_end);
            else frag = pe_insert_size(pe, h_strand, h_start, h_end, a.strand, a_start, a_end);
            if (frag <= 0) continue;
            if (seg2 != nullptr && lane == 0) {                              // the whole tsHitLoci is replaced (Aligner.cpp:3420)
                bk_seg2 g{};
                if (trimmed_len != read_len) {                               // FlgChimeric: TrimLeft / TrimRight in read orientation
                    g.flags = 8;
                    g.match_len = (uint16_t)(anti ? best_t3 : best_t5);
                    g.read_ofs = (uint16_t)(anti ? best_t5 : best_t3);
                }
                seg2[oi] = g;
            }
            o.chrom_id = a.chrom_id; o.match_loci = best_loci; o.match_len = (uint16_t)read_len; o.strand = h_strand;
            o.mismatches = (uint8_t)mm; o.num_hits = 1; o.low_mm = (int8_t)mm; o.low_hit_instances = 1;
            f.flags |= 0x80; r.flags |= 0x80;
            f.nar = BK_NAR_ACCEPTED; r.nar = BK_NAR_ACCEPTED;
            done = true;
        }
        if (!done) pe_finish(pe, f, r);
        if (lane == 0) { hits[2 * i] = f; hits[2 * i + 1] = r; }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// launchers (called from bk_engine.cpp through plain function pointers-free C++ interface)

void launch_pack_target(const uint8_t *seq, uint64_t n, uint64_t *tgt4, uint64_t nwords, hipStream_t s)
{
    uint64_t blocks = (nwords + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_pack_target, dim3((unsigned)blocks), dim3(256), 0, s, seq, n, tgt4, nwords);
}

void launch_split_sa5(const uint8_t *sa5, uint64_t n, uint32_t *lo, uint8_t *hi, hipStream_t s)
{
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_split_sa5, dim3((unsigned)blocks), dim3(256), 0, s, sa5, n, lo, hi);
}

void launch_build_ktab(const DevIndex &ix, void *tab, int k, bool tab64, hipStream_t s)
{
    uint64_t blocks = (ix.n + 1 + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    bool wide = ix.sa_hi != nullptr;
    if (wide) {
        if (tab64) hipLaunchKernelGGL((k_build_ktab<true, uint64_t>), dim3((unsigned)blocks), dim3(256), 0, s, ix, (uint64_t *)tab, k);
        else hipLaunchKernelGGL((k_build_ktab<true, uint32_t>), dim3((unsigned)blocks), dim3(256), 0, s, ix, (uint32_t *)tab, k);
    } else {
        if (tab64) hipLaunchKernelGGL((k_build_ktab<false, uint64_t>), dim3((unsigned)blocks), dim3(256), 0, s, ix, (uint64_t *)tab, k);
        else hipLaunchKernelGGL((k_build_ktab<false, uint32_t>), dim3((unsigned)blocks), dim3(256), 0, s, ix, (uint32_t *)tab, k);
    }
}

// hipMemsetAsync is not trusted with >= 4 GiB spans: clear with our own grid-stride kernel
void launch_fill_u64(unsigned long long *p, uint64_t n, unsigned long long v, hipStream_t s)
{
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (!blocks) return;
    hipLaunchKernelGGL(k_fill_u64, dim3((unsigned)blocks), dim3(256), 0, s, p, n, v);
}

void launch_count_seqs(const bk_hit *out, uint32_t n, const uint32_t *id2idx, uint32_t n_ent, unsigned long long *counts, hipStream_t s)
{
    if (!n) return;
    uint32_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_count_seqs, dim3(blocks), dim3(256), 0, s, out, n, id2idx, n_ent, counts);
}

void launch_max_len(const uint32_t *lens, uint32_t n, uint32_t *out, hipStream_t s)
{
    uint32_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (n) hipLaunchKernelGGL(k_max_len, dim3(blocks), dim3(256), 0, s, lens, n, out);
}

void launch_pack_target2(const uint64_t *tgt4, uint64_t nwords4, uint64_t *tgt2, unsigned int *nflag32, int flag_shift, hipStream_t s)
{
    uint64_t blocks = (nwords4 / 4 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_pack_target2, dim3((unsigned)blocks), dim3(256), 0, s, tgt4, nwords4, tgt2, nflag32, flag_shift);
}

static uint32_t stripe_cap(unsigned blocks, unsigned per_block)
{
    return ((blocks + kListStripes - 1) / kListStripes) * per_block;
}

// the stripes of up to three lists -> their dense forms (appended behind total[i] entries; the counts and the maximum follow)
static void launch_compact(const StripeSet &set, uint32_t *const *dense, uint32_t *const *total, int n, uint32_t *max_out, hipStream_t s)
{
    CompactJobs J;
    J.set = set;
    for (int i = 0; i < 3; i++) { J.dense[i] = dense[i < n ? i : 0]; J.total[i] = total[i < n ? i : 0]; }
    J.max_out = max_out;
    J.n = n;
    hipLaunchKernelGGL(k_compact_lists, dim3(1024, (unsigned)n), dim3(256), 0, s, J);
    hipLaunchKernelGGL(k_finish_lists, dim3(1), dim3(64), 0, s, J);
}

// stage: at least n_reads + (kListStripes + 2) * 1024 entries; stripe_cnt: kListStripes * 16 words, zero between launches
void launch_prep(const DevAlignCfg &cfg, const DevBatch &b, uint32_t *act, uint32_t *act_cnt, uint32_t *cmax, uint32_t *stage,
                 uint32_t *stripe_cnt, hipStream_t s)
{
    const bool packed = b.pk_words != nullptr;
    const unsigned eblocks = (unsigned)((b.pk_nexc + 255) / 256);
    if (b.nw == 8 || b.nw == 16 || b.nw == kNwLong || b.nw == kNwLongest) {          // register-kernel path: fused pack + init
        const unsigned blocks = (b.n_reads + 255) / 256;
        StripeSet out;
        out.cnt = stripe_cnt;
        out.stage[0] = out.stage[1] = out.stage[2] = stage;
        out.cap = stripe_cap(blocks, 256);
        if (packed) {
            // the exception list first tells every read how many N it holds (the N policy is decided in the fused kernel), and
            // afterwards writes the codes into 4-bit rows: existing ones, or - lean batches - rows made for just these reads
            launch_fill_u64(reinterpret_cast<unsigned long long *>(b.rmeta), ((uint64_t)b.n_reads + 1) / 2, 0ULL, s);      // (rmeta is allocated in whole 8-byte words)
            if (eblocks) hipLaunchKernelGGL(k_mark_exc, dim3(eblocks), dim3(256), 0, s, b);
            if (b.nw == 8) hipLaunchKernelGGL((k_prep_fused<8, true>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
            else if (b.nw == 16) hipLaunchKernelGGL((k_prep_fused<16, true>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
            else if (b.nw == kNwLong) hipLaunchKernelGGL((k_prep_fused<kNwLong, true>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
            else hipLaunchKernelGGL((k_prep_fused<kNwLongest, true>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
            if (eblocks && b.rd2 != nullptr) hipLaunchKernelGGL(k_exc_rows, dim3(eblocks), dim3(256), 0, s, b);
            if (eblocks) hipLaunchKernelGGL(k_apply_exc, dim3(eblocks), dim3(256), 0, s, b);
        } else if (b.nw == 8) hipLaunchKernelGGL((k_prep_fused<8, false>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
        else if (b.nw == 16) hipLaunchKernelGGL((k_prep_fused<16, false>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
        else if (b.nw == kNwLong) hipLaunchKernelGGL((k_prep_fused<kNwLong, false>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
        else hipLaunchKernelGGL((k_prep_fused<kNwLongest, false>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
        launch_compact(out, &act, &act_cnt, 1, cmax, s);
        return;
    }
    const uint32_t rpb = 256 / (2 * b.wpr);
    hipLaunchKernelGGL(k_pack_reads, dim3((b.n_reads + rpb - 1) / rpb), dim3(256), 0, s, b);
    if (packed && eblocks) hipLaunchKernelGGL(k_apply_exc, dim3(eblocks), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_init_reads, dim3((b.n_reads + 1023) / 1024), dim3(1024), 0, s, cfg, b, act, act_cnt, cmax);
}

void launch_search(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, uint32_t n_act,
                   int phase, int cmax, int nstr, int lazy, hipStream_t s)
{
    uint64_t threads = (uint64_t)n_act * (uint64_t)(cmax * nstr);
    unsigned blocks = (unsigned)((threads + 255) / 256);
    if (ix.sa_hi) hipLaunchKernelGGL(k_search<true>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, n_act, phase, cmax, nstr, lazy);
    else hipLaunchKernelGGL(k_search<false>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, n_act, phase, cmax, nstr, lazy);
}

void launch_extend(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, uint32_t n_act,
                   int phase, uint32_t *next_act, uint32_t *next_cnt, uint32_t *heavy, uint32_t *heavy_cnt,
                   uint32_t *cmax_next, hipStream_t s)
{
    unsigned blocks = (n_act + 255) / 256;
    if (ix.sa_hi) hipLaunchKernelGGL(k_extend<true>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, n_act, phase, next_act, next_cnt, heavy, heavy_cnt, cmax_next);
    else hipLaunchKernelGGL(k_extend<false>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, n_act, phase, next_act, next_cnt, heavy, heavy_cnt, cmax_next);
}

void launch_pe(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, int pe_mode, int min_len, int max_len, int pair_strand,
               bk_hit *hits, uint32_t n_pairs, uint32_t *orphans, uint32_t *counters /*[0] count [1] cursor, zeroed*/,
               uint32_t *h_count, bk_seg2 *seg2, int min_chim, int long_reads, hipStream_t s)
{
    DevPE pe{pe_mode, min_len, max_len, pair_strand};
    const uint32_t rpb = 256 / (2 * b.wpr);
    hipLaunchKernelGGL(k_pack_reads, dim3((b.n_reads + rpb - 1) / rpb), dim3(256), 0, s, b);
    if (b.pk_words != nullptr && b.pk_nexc) hipLaunchKernelGGL(k_apply_exc, dim3((unsigned)((b.pk_nexc + 255) / 256)), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_pe_classify, dim3((n_pairs + 255) / 256), dim3(256), 0, s, pe, hits, n_pairs, orphans, counters, seg2);
    (void)hipMemcpyAsync(h_count, counters, 4, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    uint32_t n = *h_count;
    if (n) {
        uint32_t waves = n < 8192 ? n : 8192;
#define BK_ORPH(W) hipLaunchKernelGGL(k_pe_orphan<W>, dim3((waves + 3) / 4), dim3(256), 0, s, ix, cfg, pe, b, hits, orphans, n, counters + 1, seg2, min_chim)
        if (min_chim <= 0 || seg2 == nullptr) BK_ORPH(0);
        else if (long_reads) BK_ORPH(32);                                     // reads of more than 512 bases: 2048-base mismatch map per lane
        else BK_ORPH(8);
#undef BK_ORPH
    }
}

// k-mer table entries {bucket start, second-level key of the bucket's first suffix} (DevIndex::ktab2)
__global__ void __launch_bounds__(256) k_make_ktab2(const uint32_t *__restrict__ tab, const uint32_t *__restrict__ k2, uint64_t n_entries, uint64_t n,
                                                    uint2 *__restrict__ out)
{
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_entries; c += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t lo = tab[c];
        out[c] = make_uint2(lo, lo < n ? k2[lo] : 0u);
    }
}

void launch_make_ktab2(const uint32_t *tab, const uint32_t *k2, uint64_t n_entries, uint64_t n, void *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_make_ktab2, dim3(65536), dim3(256), 0, s, tab, k2, n_entries, n, reinterpret_cast<uint2 *>(out));
}

void launch_build_k2(const DevIndex &ix, uint32_t *k2, unsigned long long *bad, hipStream_t s)
{
    uint64_t blocks = (ix.n + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    DevIndex t = ix;
    t.k2 = k2;
    if (ix.sa_hi) {
        hipLaunchKernelGGL(k_build_k2<true>, dim3((unsigned)blocks), dim3(256), 0, s, t, k2);
        hipLaunchKernelGGL(k_check_k2<true>, dim3((unsigned)blocks), dim3(256), 0, s, t, k2, bad);
    } else {
        hipLaunchKernelGGL(k_build_k2<false>, dim3((unsigned)blocks), dim3(256), 0, s, t, k2);
        hipLaunchKernelGGL(k_check_k2<false>, dim3((unsigned)blocks), dim3(256), 0, s, t, k2, bad);
    }
}

// keys for grouping work items that touch the same part of the index (see bk_engine.cpp, sort_work):
// search items by the start of their k-mer bucket, wave items by the start of their longest core interval
__global__ void __launch_bounds__(256) k_keys_search(DevBatch b, const uint32_t *__restrict__ list, uint32_t n, int shift,
                                                     uint32_t *__restrict__ keys)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = (uint32_t)(iv_start(b, list[i]) >> shift);
}

__global__ void __launch_bounds__(256) k_keys_wave(DevAlignCfg cfg, DevBatch b, int phase, const uint32_t *__restrict__ list,
                                                   uint32_t n, int shift, uint32_t *__restrict__ keys, const uint32_t *__restrict__ work_of)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pos = list[i], r = b.act[pos];
    if (shift < 0 && work_of != nullptr) { keys[i] = 0xFFFFFFFFu - work_of[pos]; return; }      // k_flat has added the intervals up already
    const int len = (int)b.lens[r];
    ReadPlan p = make_plan(len, cfg);
    int mm, cl, cd, dummy[1];
    phase_params(p, cfg, phase, mm, cl, cd);
    int nc = core_offsets(len, cl, cd, p.max_slides, dummy, 0);
    if (nc > kMaxCoresFast) nc = kMaxCoresFast;
    const int s0 = cfg.align_strand == 2 ? 1 : 0, s1 = cfg.align_strand == 1 ? 0 : 1;
    uint32_t best_n = 0;
    uint64_t best_first = 0, work = 0;
    for (int st = s0; st <= s1; st++)
        for (int c = 0; c < nc; c++) {
            uint64_t slot = iv_slot(b, pos, st, c);
            uint64_t f;
            uint32_t raw;
            iv_get(b, slot, f, raw);
            uint32_t cnt = raw & ~kLazyFlag;
            work += cnt;
            if (cnt > best_n) { best_n = cnt; best_first = f; }
        }
    // shift < 0: longest job first (the reads are dealt to the waves in list order; a read with 10^5 candidates that comes up last
    // keeps one wave busy long after the others have run dry)
    keys[i] = shift < 0 ? 0xFFFFFFFFu - (uint32_t)(work < 0xFFFFFFFFULL ? work : 0xFFFFFFFFULL) : (uint32_t)(best_first >> shift);
}

void launch_keys_search(const DevBatch &b, const uint32_t *list, uint32_t n, int shift, uint32_t *keys, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_keys_search, dim3((n + 255) / 256), dim3(256), 0, s, b, list, n, shift, keys);
}

void launch_keys_wave(const DevAlignCfg &cfg, const DevBatch &b, int phase, const uint32_t *list, uint32_t n, int shift, uint32_t *keys,
                      const uint32_t *work_of, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_keys_wave, dim3((n + 255) / 256), dim3(256), 0, s, cfg, b, phase, list, n, shift, keys, work_of);
}

// stage: at least n_act * cmax * nstr + (kListStripes + 2) * 1024 entries; stripe_cnt: kListStripes * 16 words, zero between launches
void launch_search_a(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, uint32_t n_act,
                     int phase, int cmax, int nstr, int lazy, uint32_t *list, uint32_t *list_cnt, uint32_t *stage, uint32_t *stripe_cnt,
                     hipStream_t s)
{
    uint64_t threads = (uint64_t)n_act * (uint64_t)(cmax * nstr);
    int ilp = lazy >> 8;                               // bits 8..: searches per lane (0 / 1 = the plain kernel)
    lazy &= 0xff;
    if (ilp < 2) ilp = 1;
    else if (ilp != 2) ilp = 4;
    const uint64_t per = (uint64_t)256 * (uint64_t)ilp;
    const unsigned blocks = (unsigned)((threads + per - 1) / per);
    StripeSet out;
    out.cnt = stripe_cnt;
    out.stage[0] = out.stage[1] = out.stage[2] = stage;
    out.cap = stripe_cap(blocks, (unsigned)per);
    if (ilp == 2) hipLaunchKernelGGL(k_search_a_ilp<2>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, n_act, phase, cmax, nstr, lazy, out);
    else if (ilp == 4) hipLaunchKernelGGL(k_search_a_ilp<4>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, n_act, phase, cmax, nstr, lazy, out);
    else hipLaunchKernelGGL(k_search_a_ilp<1>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, n_act, phase, cmax, nstr, lazy, out);
    launch_compact(out, &list, &list_cnt, 1, nullptr, s);
}

void launch_search_b(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, int phase, int lazy, const uint32_t *list,
                     uint32_t n_list, hipStream_t s)
{
    if (!n_list) return;
    unsigned blocks = (n_list + 255) / 256;
    if (ix.sa_hi) hipLaunchKernelGGL(k_search_b<true>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, phase, lazy, list, n_list);
    else hipLaunchKernelGGL(k_search_b<false>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, phase, lazy, list, n_list);
}

void launch_build_isa(const uint32_t *sa, uint64_t n, uint32_t *isa, hipStream_t s)
{
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    hipLaunchKernelGGL(k_build_isa, dim3((unsigned)blocks), dim3(256), 0, s, sa, n, isa);
}

void launch_light(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, uint32_t n_act, int phase,
                  uint32_t *next_act, uint32_t *next_cnt, uint32_t *heavy, uint32_t *heavy_cnt, uint32_t *wave, uint32_t *wave_cnt,
                  uint32_t *cmax_next, int nw, hipStream_t s)
{
    unsigned blocks = (n_act + 255) / 256;
    bool wide = ix.sa_hi != nullptr;
#define BK_LIGHT(W, N) hipLaunchKernelGGL((k_light<W, N>), dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, n_act, phase, next_act, next_cnt, heavy, heavy_cnt, wave, wave_cnt, cmax_next)
    if (nw <= 8) { if (wide) BK_LIGHT(true, 8); else BK_LIGHT(false, 8); }
    else if (nw <= 16) { if (wide) BK_LIGHT(true, 16); else BK_LIGHT(false, 16); }
    else if (nw <= kNwLong) { if (wide) BK_LIGHT(true, kNwLong); else BK_LIGHT(false, kNwLong); }
    else { if (wide) BK_LIGHT(true, kNwLongest); else BK_LIGHT(false, kNwLongest); }
#undef BK_LIGHT
}

// stage: three buffers of at least n_act + (kListStripes + 2) * 1024 entries, stripe_cnt: kListStripes * 16 words, zero between launches
void launch_flat(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, uint32_t n_act, int phase,
                 int slots_max, uint32_t *next_act, uint32_t *next_cnt, uint32_t *heavy, uint32_t *heavy_cnt, uint32_t *wave,
                 uint32_t *wave_cnt, uint32_t *cmax_next, uint32_t *const *stage, uint32_t *stripe_cnt, int nw, hipStream_t s)
{
    int bs = (nw >> 8) ? (nw >> 8) : 256;              // bits 8..: reads (= threads) per block, 64 .. 1024
    nw &= 0xff;
    bool wide = ix.sa_hi != nullptr;
    if (wide && bs > 256) bs = 256;                    // (the low words of the 5-byte form need 4 more bytes of LDS per candidate)
    unsigned blocks = (n_act + (unsigned)bs - 1) / (unsigned)bs;
    if (slots_max < 1) slots_max = 1;
    size_t lds = (size_t)bs * slots_max * (flat_caches_first(wide, bs, slots_max) ? 6 : 2);
    StripeSet out;
    out.cnt = stripe_cnt;
    for (int i = 0; i < 3; i++) out.stage[i] = stage[i];
    out.cap = stripe_cap(blocks, (unsigned)bs);
    const int have_wave = wave != nullptr;
#define BK_FLAT(W, N, B) hipLaunchKernelGGL((k_flat<W, N, B>), dim3(blocks), dim3(B), lds, s, ix, cfg, b, act, n_act, phase, slots_max, out, have_wave)
#define BK_FLAT_W(N) do { if (bs == 64) BK_FLAT(true, N, 64); else if (bs == 128) BK_FLAT(true, N, 128); else BK_FLAT(true, N, 256); } while (0)
#define BK_FLAT_B(N) do { if (bs == 64) BK_FLAT(false, N, 64); else if (bs == 128) BK_FLAT(false, N, 128); else if (bs == 512) BK_FLAT(false, N, 512); else if (bs == 1024) BK_FLAT(false, N, 1024); else BK_FLAT(false, N, 256); } while (0)
    if (nw <= 8) { if (wide) BK_FLAT_W(8); else BK_FLAT_B(8); }
    else if (nw <= 16) { if (wide) BK_FLAT_W(16); else BK_FLAT_B(16); }
    else if (nw <= kNwLong) { if (wide) BK_FLAT_W(kNwLong); else BK_FLAT_B(kNwLong); }
    else { if (wide) BK_FLAT_W(kNwLongest); else BK_FLAT_B(kNwLongest); }
#undef BK_FLAT_W
#undef BK_FLAT_B
#undef BK_FLAT
    // list order of the set: next phase, wave kernel, general kernel (the wave list may be absent: then it stays empty)
    uint32_t *dense[3] = {next_act, have_wave ? wave : heavy, heavy}, *total[3] = {next_cnt, have_wave ? wave_cnt : heavy_cnt, heavy_cnt};
    launch_compact(out, dense, total, 3, cmax_next, s);
}

void launch_wave(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const HeavyScratch &hs, const uint32_t *list, uint32_t n_list,
                 int phase, uint32_t *cursor, uint32_t *next_act, uint32_t *next_cnt, uint32_t *cmax_next, int nw, uint32_t max_waves,
                 hipStream_t s)
{
    const bool wide = ix.sa_hi != nullptr, hash = ix.isa == nullptr;
    uint32_t waves = n_list < max_waves ? n_list : max_waves;
    if (hash && waves > hs.n_slots) waves = hs.n_slots;
    unsigned blocks = (waves + 3) / 4;
#define BK_WAVE(N, W, H, S, G) hipLaunchKernelGGL((k_wave<N, W, H, S, G>), dim3(blocks), dim3(256), 0, s, ix, cfg, b, hs, list, n_list, phase, cursor, next_act, next_cnt, cmax_next)
    const bool sw = ix.swin != nullptr && b.rd2 != nullptr;
    const bool group8 = (nw & 0x100) != 0;                 // 8-word form: small intervals share rounds (the 16-word form always does)
    nw &= 0xff;
    if (nw <= 8) {
        if (wide) BK_WAVE(8, true, true, false, true);
        else if (hash) BK_WAVE(8, false, true, false, true);
        else if (sw) { if (group8) BK_WAVE(8, false, false, true, true); else BK_WAVE(8, false, false, true, false); }
        else { if (group8) BK_WAVE(8, false, false, false, true); else BK_WAVE(8, false, false, false, false); }
    } else if (nw <= 16) {
        if (wide) BK_WAVE(16, true, true, false, true);
        else if (hash) BK_WAVE(16, false, true, false, true);
        else if (sw) BK_WAVE(16, false, false, true, true);
        else BK_WAVE(16, false, false, false, true);
    } else if (nw <= kNwLong) {                            // reads of 257 .. 16 * kNwLong bases
        if (wide) BK_WAVE(kNwLong, true, true, false, true);
        else if (hash) BK_WAVE(kNwLong, false, true, false, true);
        else BK_WAVE(kNwLong, false, false, false, true);
    } else {                                               // .. 16 * kNwLongest bases
        if (wide) BK_WAVE(kNwLongest, true, true, false, true);
        else if (hash) BK_WAVE(kNwLongest, false, true, false, true);
        else BK_WAVE(kNwLongest, false, false, false, true);
    }
#undef BK_WAVE
}

void launch_heavy(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const HeavyScratch &hs, const uint32_t *list,
                  uint32_t n_list, int phase, uint32_t *cursor, uint32_t *next_act, uint32_t *next_cnt, uint32_t *cmax_next,
                  hipStream_t s)
{
    uint32_t waves = n_list < hs.n_slots ? n_list : hs.n_slots;
    unsigned blocks = (waves + 3) / 4;
    expand_rd4(b, list, n_list, s);
    if (ix.sa_hi) hipLaunchKernelGGL((k_heavy<true, 0>), dim3(blocks), dim3(256), 0, s, ix, cfg, b, hs, list, n_list, phase, cursor, next_act, next_cnt, cmax_next, nullptr, nullptr, nullptr);
    else hipLaunchKernelGGL((k_heavy<false, 0>), dim3(blocks), dim3(256), 0, s, ix, cfg, b, hs, list, n_list, phase, cursor, next_act, next_cnt, cmax_next, nullptr, nullptr, nullptr);
}

// -c: the chimeric LocateCoreMultiples call for every read of `list` (reads nothing else aligned); trims into seg2[]
void launch_chimeric(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const HeavyScratch &hs, const uint32_t *list, uint32_t n_list,
                     int min_pct, int long_reads, uint32_t *cursor, bk_seg2 *seg2, hipStream_t s)
{
    if (!n_list) return;
    uint32_t waves = n_list < hs.n_slots ? n_list : hs.n_slots;
    unsigned blocks = (waves + 3) / 4;
    expand_rd4(b, list, n_list, s);
#define BK_CHIM(W, M) hipLaunchKernelGGL((k_heavy<W, M>), dim3(blocks), dim3(256), 0, s, ix, cfg, b, hs, list, n_list, min_pct, cursor, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<bk_loci *>(seg2), nullptr)
    if (long_reads) { if (ix.sa_hi) BK_CHIM(true, 4); else BK_CHIM(false, 4); }      // reads of more than 512 bases: 2048-base mismatch map per lane
    else { if (ix.sa_hi) BK_CHIM(true, 3); else BK_CHIM(false, 3); }
#undef BK_CHIM
}

// -N: LocateBestMatches for every read of `list`; cnt[r] = loci kept, dense[r * MaxHits ..] = the loci
void launch_best(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const HeavyScratch &hs, const uint32_t *list,
                 uint32_t n_list, uint32_t *cursor, unsigned long long *cnt, bk_loci *dense, hipStream_t s)
{
    if (!n_list) return;
    uint32_t waves = n_list < hs.n_slots ? n_list : hs.n_slots;
    unsigned blocks = (waves + 3) / 4;
    expand_rd4(b, list, n_list, s);
    if (ix.sa_hi) hipLaunchKernelGGL((k_heavy<true, 2>), dim3(blocks), dim3(256), 0, s, ix, cfg, b, hs, list, n_list, 0, cursor, nullptr, nullptr, nullptr, cnt, dense, nullptr);
    else hipLaunchKernelGGL((k_heavy<false, 2>), dim3(blocks), dim3(256), 0, s, ix, cfg, b, hs, list, n_list, 0, cursor, nullptr, nullptr, nullptr, cnt, dense, nullptr);
}

// dense rows of `width` loci -> packed lists at offs[]
__global__ void __launch_bounds__(256) k_loci_compact(const bk_loci *__restrict__ dense, uint32_t width, const unsigned long long *__restrict__ offs,
                                                      uint32_t n, bk_loci *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t r = i / width;
    const uint32_t k = (uint32_t)(i % width);
    if (r >= n) return;
    const unsigned long long o = offs[r];
    if (k < offs[r + 1] - o) out[o + k] = dense[i];
}

void launch_loci_compact(const bk_loci *dense, uint32_t width, const unsigned long long *offs, uint32_t n, bk_loci *out, hipStream_t s)
{
    const uint64_t tot = (uint64_t)n * width;
    if (tot) hipLaunchKernelGGL(k_loci_compact, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, dense, width, offs, n, out);
}

void launch_loci_count(const bk_hit *out, uint32_t n, int clamp_to, unsigned long long *cnt, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_loci_count, dim3((n + 255) / 256), dim3(256), 0, s, out, n, clamp_to, cnt);
}

void launch_loci_single(const bk_hit *out, uint32_t n, const unsigned long long *offs, bk_loci *loci, uint32_t *list, uint32_t *list_cnt,
                        const bk_seg2 *seg2, bk_loci_trims *trims, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_loci_single, dim3((n + 255) / 256), dim3(256), 0, s, out, n, offs, loci, list, list_cnt, seg2, trims);
}

// min_pct / seg2 / trims: contexts that trim chimeric reads (`-c` with the multi-loci modes) - the loci of a read the chimeric call placed
// come from a replay of that call, each with its end trims
void launch_loci_enum(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const HeavyScratch &hs, const uint32_t *list,
                      uint32_t n_list, uint32_t *cursor, const unsigned long long *offs, bk_loci *loci, uint32_t *err, int min_pct, int long_reads,
                      bk_seg2 *seg2, bk_loci_trims *trims, hipStream_t s)
{
    if (!n_list) return;
    uint32_t waves = n_list < hs.n_slots ? n_list : hs.n_slots;
    unsigned blocks = (waves + 3) / 4;
    expand_rd4(b, list, n_list, s);
#define BK_ENUM(W, M) hipLaunchKernelGGL((k_heavy<W, M>), dim3(blocks), dim3(256), 0, s, ix, cfg, b, hs, list, n_list, min_pct, cursor, nullptr, nullptr, nullptr, offs, loci, err, seg2, trims)
    if (seg2 != nullptr && long_reads) { if (ix.sa_hi) BK_ENUM(true, 5); else BK_ENUM(false, 5); }
    else { if (ix.sa_hi) BK_ENUM(true, 1); else BK_ENUM(false, 1); }
#undef BK_ENUM
}


// ------------------------------------------------------------------------------------------------
// SNP pile-up and screening (CAligner::ProcessSNPs :7737-7960, OutputSNPs :6880-7110).
// Counts are six planes of uint32 over the concatenated target: plane 0 = NumRefBases, planes 1..5 = NonRefBaseCnts
// a,c,g,t,n; a wave walks one alignment, its lanes consecutive loci, so the adds of a plane coalesce.

__global__ void __launch_bounds__(256) k_snp_pileup(DevIndex ix, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offs,
                                                    const uint32_t *__restrict__ id2idx, const bk_snp_aln *__restrict__ alns, uint64_t n_alns,
                                                    uint32_t *__restrict__ planes)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t a = wave; a < n_alns; a += n_waves) {
        const bk_snp_aln al = alns[a];
        const uint32_t e = id2idx[al.chrom_id];
        const uint64_t g0 = ix.ent_start[e] + al.loci;
        uint32_t len = al.len;
        const uint64_t chrom_len = ix.ent_end[e] - ix.ent_start[e] + 1;
        if ((uint64_t)al.loci + len > chrom_len) {                       // :7826-7830
            if (al.loci + 10 > chrom_len) continue;
            len = (uint32_t)(chrom_len - al.loci);
        }
        const uint8_t *rd = bases + offs[al.read_idx] + al.read_ofs;
        const bool rev = al.strand == '-';
        for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
            uint32_t rb = rev ? rd[al.len - 1 - i] & 7u : rd[i] & 7u;
            if (rev && rb < 4) rb = 3 - rb;
            const uint64_t g = g0 + i;
            const uint32_t tb = (uint32_t)(ix.tgt4[g >> 4] >> (60 - 4 * (unsigned)(g & 15))) & 7u;
            if (tb >= 4 || rb > 4) continue;
            atomicAdd(&planes[(tb == rb ? 0 : (uint64_t)(1 + rb) * ix.n) + g], 1u);
        }
    }
}

__global__ void __launch_bounds__(256) k_snp_sites(DevIndex ix, const uint32_t *__restrict__ planes, uint64_t g0, uint32_t chrom_len, uint32_t min_reads,
                                                   double min_prop, bk_snp_site *__restrict__ sites, uint32_t cap, uint32_t *__restrict__ n_sites,
                                                   unsigned long long *__restrict__ totals /*[4]*/)
{
    constexpr uint32_t kFlank = 25, kWin = 2 * kFlank + 1;                 // cSNPBkgndRateWindow = 51
    __shared__ unsigned long long s_tot[4];
    if (threadIdx.x < 4) s_tot[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long t_m = 0, t_mm = 0, t_cov = 0, t_bases = 0;
    for (uint64_t l64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; l64 < chrom_len; l64 += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = (uint32_t)l64;
        const uint64_t g = g0 + l;
        const uint32_t ref = planes[g];
        uint32_t nr[5], nonref = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) { nr[k] = planes[(uint64_t)(1 + k) * ix.n + g]; nonref += nr[k]; }
        const uint32_t tot = ref + nonref;
        t_m += ref; t_mm += nonref;
        if (tot) { t_cov++; t_bases += tot; }
        if (tot < min_reads || nonref < 1u) continue;
        if ((double)nonref / (double)(int)tot < min_prop) continue;
        // the window OutputSNPs has slid to by the time it looks at this locus (:6885-6925)
        uint32_t w_lo = 0, w_n = chrom_len < kWin ? chrom_len : kWin;
        if (l > kFlank && chrom_len > kWin) {
            const uint32_t last = chrom_len - 1 - kFlank;
            w_lo = (l < last ? l : last) - kFlank;
        }
        uint32_t wm = 0, wmm = 0;
        for (uint32_t k = 0; k < w_n; k++) {
            const uint64_t q = g0 + w_lo + k;
            wm += planes[q];
#pragma unroll
            for (int b = 0; b < 5; b++) wmm += planes[(uint64_t)(1 + b) * ix.n + q];
        }
        const uint32_t slot = atomicAdd(n_sites, 1u);
        if (slot < cap) {
            bk_snp_site st;
            st.loci = l; st.num_ref = ref;
#pragma unroll
            for (int k = 0; k < 5; k++) st.non_ref[k] = nr[k];
            st.win_mismatches = wmm; st.win_matches = wm;
            st.ref_base = (uint32_t)(ix.tgt4[g >> 4] >> (60 - 4 * (unsigned)(g & 15))) & 7u;
            sites[slot] = st;
        }
    }
    atomicAdd(&s_tot[0], t_m); atomicAdd(&s_tot[1], t_mm); atomicAdd(&s_tot[2], t_cov); atomicAdd(&s_tot[3], t_bases);
    __syncthreads();
    if (threadIdx.x < 4 && s_tot[threadIdx.x]) atomicAdd(&totals[threadIdx.x], s_tot[threadIdx.x]);
}

__global__ void __launch_bounds__(256) k_snp_gather(DevIndex ix, const uint32_t *__restrict__ planes, uint64_t g0, uint32_t n, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t g = g0 + i;
#pragma unroll
    for (int k = 0; k < 6; k++) out[(uint64_t)i * 7 + k] = planes[(uint64_t)k * ix.n + g];
    out[(uint64_t)i * 7 + 6] = (uint32_t)(ix.tgt4[g >> 4] >> (60 - 4 * (unsigned)(g & 15))) & 7u;
}

void launch_snp_gather(const DevIndex &ix, const uint32_t *planes, uint64_t g0, uint32_t n, uint32_t *out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_snp_gather, dim3((n + 255) / 256), dim3(256), 0, s, ix, planes, g0, n, out);
}

__global__ void __launch_bounds__(256) k_snp_centroids(DevIndex ix, const uint32_t *__restrict__ planes, uint64_t g0, uint32_t chrom_len, uint32_t min_reads,
                                                       uint32_t *__restrict__ hist)
{
    __shared__ uint32_t s_hist[BK_SNP_CENTROIDS];                          // 64 KB of the CU's 160 KB LDS
    for (uint32_t i = threadIdx.x; i < BK_SNP_CENTROIDS; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    for (uint64_t l64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 3; l64 + 3 < chrom_len; l64 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t g = g0 + l64;
        uint32_t tot = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) tot += planes[(uint64_t)k * ix.n + g];
        if (tot < min_reads) continue;
        const uint64_t w = nib16(ix.tgt4, g - 3);                         // 7 target bases from the top nibble down
        uint32_t idx = 0;
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const uint32_t b = (uint32_t)(w >> (60 - 4 * j)) & 7u;
            ok = ok && b < 4;
            idx = (idx << 2) | (b & 3u);
        }
        if (ok) atomicAdd(&s_hist[idx], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < BK_SNP_CENTROIDS; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}

void launch_snp_centroids(const DevIndex &ix, const uint32_t *planes, uint64_t g0, uint32_t chrom_len, uint32_t min_reads, uint32_t *hist, hipStream_t s)
{
    if (chrom_len < 7) return;
    uint64_t blocks = ((uint64_t)chrom_len + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_snp_centroids, dim3((unsigned)blocks), dim3(256), 0, s, ix, planes, g0, chrom_len, min_reads, hist);
}

void launch_snp_pileup(const DevIndex &ix, const uint8_t *bases, const uint64_t *offs, const uint32_t *id2idx, const bk_snp_aln *alns, uint64_t n_alns,
                       uint32_t *planes, hipStream_t s)
{
    if (!n_alns) return;
    uint64_t blocks = (n_alns + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_snp_pileup, dim3((unsigned)blocks), dim3(256), 0, s, ix, bases, offs, id2idx, alns, n_alns, planes);
}

void launch_snp_sites(const DevIndex &ix, const uint32_t *planes, uint64_t g0, uint32_t chrom_len, uint32_t min_reads, double min_prop,
                      bk_snp_site *sites, uint32_t cap, uint32_t *n_sites, unsigned long long *totals, hipStream_t s)
{
    if (!chrom_len) return;
    uint64_t blocks = ((uint64_t)chrom_len + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_snp_sites, dim3((unsigned)blocks), dim3(256), 0, s, ix, planes, g0, chrom_len, min_reads, min_prop, sites, cap, n_sites, totals);
}

}  // namespace bk

extern "C" int bk_debug_prof(unsigned long long *out16)
{
    unsigned long long h[64 * 16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(bk::g_prof), sizeof(h)) != hipSuccess) return 1;
    for (int k = 0; k < 16; k++) { out16[k] = 0; for (int s0 = 0; s0 < 64; s0++) out16[k] += h[s0 * 16 + k]; }
    return 0;
}

