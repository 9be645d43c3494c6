// bk_dev_k2.h - the second-level search keys (DevIndex::k2): their layout, compare and construction helpers, shared by the index
// set-up kernels (bk_index.hip) and the search passes (bk_search.hip).
#pragma once
#include "bk_dev_util.h"

namespace bk {

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
//           sub-bucket is not handed on unverified - over the key arrays of the next 15 and the 15 bases after those (DevIndex::kx,
//           when the index has them) and over suffix array + target from the first base no key holds on.
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

// DevIndex::ktab2's word of a bucket of two or more suffixes (k_make_ktab2): does no key of the bucket agree with the probe on the bits
// the mask keeps of the first five behind the k-mer?  (A mask that keeps fewer asks for any of the values they leave open.)
constexpr uint64_t kTab2BitmapMax = 64;
__device__ __forceinline__ bool ktab2_absent(uint32_t bitmap, uint32_t m, uint32_t q2)
{
    const uint32_t m5 = m >> 27, lo5 = (q2 >> 27) & m5, hi5 = lo5 | (~m5 & 31u);
    const uint32_t upto = hi5 == 31u ? 0xFFFFFFFFu : (1u << (hi5 + 1)) - 1u;
    return (bitmap & upto & ~((1u << lo5) - 1u)) == 0u;
}

// -1 / 0 / +1: key (masked to the core's bases) vs probe; the all-ones key sorts above everything
__device__ __forceinline__ int k2_cmp(uint32_t key, uint32_t m, uint32_t q2)
{
    if (key == kK2Above) return 1;
    key &= m;
    return key < q2 ? -1 : (key > q2 ? 1 : 0);
}
__device__ __forceinline__ bool k2_nkind(uint32_t key) { return key != kK2Above && (key & 3u) == 1u; }

// mask of the first L = min(rem2, kK2Bases) bases of a key; rem2 = bases of the core beyond the k-mer table's k
__host__ __device__ __forceinline__ uint32_t k2_mask(int rem2)
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

// ------------------------------------------------------------------------------------------------
// Sampled levels behind the keys (same allocation): level j (1 .. kK2Levels) holds the key of every 16^j-th suffix - entry g is
// k2[(g + 1) * 16^j - 1] - so a bucket of S keys is bisected sixteen ways a step: a line of the highest level with a sample inside
// the bucket, one line of every level below it, one line of the keys themselves - ceil(log16 S) + 1 lines per bound where halving the
// interval over the keys costs log2 S - 3, and as many dependent trips.  A sample may be used whenever the key it copies lies inside
// the interval at hand (the keys of other buckets say nothing), which is all the narrowing below relies on.
// (layout: kK2Levels, k2s_start .. in bk_device.h)

// entries [a, b) of L (sorted where it matters: inside the interval): how many lie below the probe, how many not above it.  Whole
// aligned lines are loaded - four 16-byte requests, no dependent trips inside a line - and the second line of a range that straddles
// two is only fetched when the first did not end both counts.
#ifdef BK_DIAG_B
#define BK_K2_LINE(ctr) (ctr)++
#else
#define BK_K2_LINE(ctr) (void)0
#endif
__device__ __forceinline__ void k2_count_range(const uint32_t *__restrict__ L, uint64_t a, uint64_t b, uint32_t m, uint32_t q2, uint32_t &n_lt, uint32_t &n_le,
                                               unsigned long long &lines)
{
    n_lt = 0; n_le = 0;
    (void)lines;
    for (uint64_t base = a & ~15ULL; base < b; base += 16) {
        const uint4 *__restrict__ p = reinterpret_cast<const uint4 *>(L + base);
        const uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        BK_K2_LINE(lines);
        const uint32_t key[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
        const uint32_t lo = a > base ? (uint32_t)(a - base) : 0u, hi = b - base < 16 ? (uint32_t)(b - base) : 16u;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            const bool in = j >= lo && j < hi;
            const uint32_t km = key[j] == kK2Above ? 0xFFFFFFFFu : (key[j] & m);         // (the probe's low two bits are clear)
            n_lt += (in && km < q2) ? 1u : 0u;
            n_le += (in && km <= q2) ? 1u : 0u;
        }
        if ((uint64_t)n_le < base + hi - a) break;
    }
}

// lower and upper bound of the probe among the keys [first, first + cnt) of one k-mer bucket; lv[j] = k2s_start(n, j)
__device__ __forceinline__ void k2_bounds(const uint32_t *__restrict__ k2, const uint64_t *lv, uint64_t first, uint64_t cnt, uint32_t m, uint32_t q2,
                                          uint64_t &lb, uint64_t &ub, unsigned long long &lines)
{
    uint64_t l1 = first, h1 = first + cnt, l2 = l1, h2 = h1;
    int ktop = cnt >= 16 ? (63 - __clzll((long long)cnt)) >> 2 : 0;                    // the highest level with a sample inside
    if (ktop > kK2Levels) ktop = kK2Levels;
    for (int j = ktop; j >= 0; j--) {
        const int sh = 4 * j;
        const uint32_t *__restrict__ L = j ? k2 + lv[j] : k2;
        // sample g of this level is key (g + 1) * 16^j - 1: inside [l, h) for g in [l >> sh, h >> sh)
        const uint64_t a1 = l1 >> sh, b1 = h1 >> sh, a2 = l2 >> sh, b2 = h2 >> sh;
        uint32_t lt = 0, le = 0;
        if (a1 < b1) {
            k2_count_range(L, a1, b1, m, q2, lt, le, lines);
            const uint64_t g = a1 + lt;                  // the first sample that is not below the probe
            if (lt) l1 = g << sh;
            if (g < b1) h1 = ((g + 1) << sh) - 1;
        }
        if (a2 < b2) {
            if (a2 != a1 || b2 != b1) { uint32_t lt2; k2_count_range(L, a2, b2, m, q2, lt2, le, lines); }
            const uint64_t g = a2 + le;                  // the first sample above the probe
            if (le) l2 = g << sh;
            if (g < b2) h2 = ((g + 1) << sh) - 1;
        }
    }
    lb = l1;
    ub = l2;
}

// the next key of the suffix at pos (DevIndex::kx): the 15 bases from `from` on (k + 15 for the third-level key, k + 30 for the fourth), in
// the form of the second-level key; `before` = its key of the level before.  Inside a run of suffixes whose keys of the level before are
// equal and of kind 0 - the only runs a level is ever consulted for: an interval that agrees with an N-free core on every base before
// `from` - these keys are non-decreasing, for the reasons the second-level keys are inside a bucket.
__device__ __forceinline__ uint32_t kx_make(const uint64_t *__restrict__ tgt4, uint64_t pos, int from, uint32_t before)
{
    if (before == kK2Above || (before & 3u) != 0u) return kK2Above;
    const uint64_t w1 = nib16(tgt4, pos + (uint64_t)from);
    const uint64_t bad = w1 & 0x4444444444444440ULL;
    const uint32_t code = squeeze2(w1) & ~3u;
    if (!bad) return code;
    const int j = __clzll((long long)bad) >> 2;
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

}  // namespace bk
