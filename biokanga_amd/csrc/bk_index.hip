// bk_index.hip - index set-up kernels (gfx950): the .sfx image as the path's kernels want it.
//   k_pack_target / k_pack_target2   1 B/base target -> 4 bit/base words; 2 bit/base copy + N/EOS region bitmap
//   k_split_sa5                      5-byte suffix elements -> lo32 + hi8 arrays
//   k_build_ktab / k_make_ktab2      k-mer table (+ entries that carry their bucket's first second-level key)
//   k_build_k2 / k_check_k2          second-level keys; k_build_k2_levels: every 16th, 256th .. of them behind the keys
//   k_build_isa                      inverse suffix array
//   k_build_swin                     suffix-ordered window array
//   k_swin_*                         .. for the part of the suffix array the wave kernel's long walks visit
#include <algorithm>

#include "bk_dev_k2.h"
#include "bk_dev_window.h"

namespace bk {

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
__global__ void k_build_ktab(DevIndex ix, TabT *__restrict__ tab, int k, uint64_t i0, uint64_t i1, unsigned long long *__restrict__ starts, int tab_stride)
{
    // stride 2 (TabT = uint32_t): the entries of DevIndex::ktab2, {bucket start, y} - the starts are written where they stay and
    // k_fill_ktab2_y adds the second words once the second-level keys are there
    // (suffix array indexes [i0, i1) of 0 .. n: a table can be made range by range, as the array arrives)
    // starts (optional; i0 a multiple of 64, the bitmap zeroed): bit i = a bucket starts at suffix array index i - what this pass finds
    // out anyway, kept for the partial window array's coverage (k_swin_breaks)
    uint64_t n = ix.n;
    uint64_t ncodes = 1ULL << (2 * k);
    uint64_t i = i0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < i1; i += stride) {
        // entries (prev, cur] receive i; prev = bucket(i-1) (or -1), cur = bucket(i) (or ncodes at i == n)
        uint64_t cur = i < n ? suffix_bucket(ix.tgt4, sa_get<WIDE>(ix, i), k) : ncodes;
        uint64_t from = i > 0 ? suffix_bucket(ix.tgt4, sa_get<WIDE>(ix, i - 1), k) + 1 : 0;
        for (uint64_t c = from; c <= cur; c++) tab[c * (uint64_t)tab_stride] = (TabT)i;
        if (starts != nullptr) {
            const unsigned long long m = __ballot(from != cur + 1);           // (the lanes of a wave hold 64 consecutive indexes from a multiple of 64 on)
            if ((i & 63) == 0 && m) atomicOr(&starts[i >> 6], m);
        }
    }
}

template <bool WIDE>
__global__ void k_build_k2(DevIndex ix, uint32_t *__restrict__ k2, uint32_t *__restrict__ k3, uint32_t *__restrict__ k4, uint64_t i0, uint64_t i1)
{
    for (uint64_t i = i0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < i1; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t pos = sa_get<WIDE>(ix, i);
        const uint32_t key = k2_make(ix.tgt4, pos, ix.k);
        if (k2 != nullptr) k2[i] = key;                    // (null: the keys are there already and in use - BK_CTX_GROW_IMAGE adds the arrays behind them)
        if (k3 != nullptr) {                                 // (the same lines of the target, or the next)
            const uint32_t key3 = kx_make(ix.tgt4, pos, ix.k + kK2Bases, key);
            k3[i] = key3;
            if (k4 != nullptr) k4[i] = kx_make(ix.tgt4, pos, ix.k + 2 * kK2Bases, key3);
        }
    }
}

// the sampled levels behind the keys (bk_dev_k2.h): one thread per word between the keys' end and the allocation's - level entries
// and the padding between levels alike
__global__ void k_build_k2_levels(uint32_t *__restrict__ k2, uint64_t n)
{
    const uint64_t w0 = n, w1 = k2s_start(n, kK2Levels + 1);
    for (uint64_t w = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < w1; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = kK2Above;
        uint64_t start = k2s_start(n, 1);
        for (int j = 1; j <= kK2Levels && w >= start; j++) {
            const uint64_t cnt = k2s_count(n, j);
            if (w - start < cnt) {
                const uint64_t src = ((w - start + 1) << (4 * j)) - 1;
                v = k2[src < n ? src : n - 1];
                break;
            }
            start += k2s_pad(cnt) + 16;
        }
        k2[w] = v;
    }
}

// the bisection needs k2 non-decreasing inside every k-mer bucket; count the places where it is not
template <bool WIDE>
__global__ void k_check_k2(DevIndex ix, const uint32_t *__restrict__ k2, unsigned long long *__restrict__ bad, uint64_t i0, uint64_t i1)
{
    // (the pairs (i, i + 1) with i in [i0, i1))
    for (uint64_t i = i0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < i1 && i + 1 < ix.n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = k2[i], c = k2[i + 1];
        if (a < c) continue;
        // equal second-level keys of kind 0: the third-level keys must be in order, and the fourth-level keys where those are equal and
        // of kind 0 - bad[1] counts where they are not
        if (a == c) {
            if ((a & 3u) != 0u || ix.kx[0] == nullptr) continue;
            const uint32_t a3 = ix.kx[0][i], c3 = ix.kx[0][i + 1];
            if (a3 < c3) continue;
            if (a3 == c3 && ((a3 & 3u) != 0u || ix.kx[1] == nullptr || ix.kx[1][i] <= ix.kx[1][i + 1])) continue;
        }
        if (suffix_bucket(ix.tgt4, sa_get<WIDE>(ix, i), ix.k) == suffix_bucket(ix.tgt4, sa_get<WIDE>(ix, i + 1), ix.k))
            atomicAdd(bad + (a == c ? 1 : 0), 1ULL);
    }
}

// entry i of the suffix-ordered window array: SwGeo<E>::bases bases of the 2-bit target from sa[i] - SwGeo<E>::pre on (bases before the
// target's start read as 0: no window that uses them passes the "candidate starts before the read does" test)
template <int E>
__device__ __forceinline__ void swin_entry(const DevIndex &ix, uint64_t i, uint4 *__restrict__ swin, uint64_t idx)
{
    const int64_t base0 = (int64_t)(ix.sa_hi != nullptr ? sa_get<true>(ix, i) : (uint64_t)ix.sa_lo[i]) - SwGeo<E>::pre;      // (5-byte elements: their fifth byte too)
    uint64_t wd[2 * E];
#pragma unroll
    for (int k = 0; k < 2 * E; k++) {
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
    for (int q = 0; q < E; q++)
        swin[sw_word_at<E>(idx, q)] = make_uint4((uint32_t)wd[2 * q], (uint32_t)(wd[2 * q] >> 32), (uint32_t)wd[2 * q + 1], (uint32_t)(wd[2 * q + 1] >> 32));
}

// (the array for every suffix: entry i is suffix i's; room for whole blocks of 32 entries)
template <int E>
__global__ void __launch_bounds__(256) k_build_swin(DevIndex ix, uint4 *__restrict__ swin)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ix.n; i += (uint64_t)gridDim.x * blockDim.x)
        swin_entry<E>(ix, i, swin, i);
}

void launch_build_swin(const DevIndex &ix, void *swin, int words, hipStream_t s)
{
    if (words == 5) hipLaunchKernelGGL(k_build_swin<5>, dim3(65536), dim3(256), 0, s, ix, reinterpret_cast<uint4 *>(swin));
    else hipLaunchKernelGGL(k_build_swin<3>, dim3(65536), dim3(256), 0, s, ix, reinterpret_cast<uint4 *>(swin));
}

// ------------------------------------------------------------------------------------------------
// The window array for PART of the suffix array (DevIndex::swmap).  k_wave's candidates are not spread evenly: it walks core intervals
// of more than 64 suffixes, whole when they hold up to MaxIter + ~100 of them and for their first ~100 - 130 entries when they hold more
// (the reference's copy-count check at IterCnt == 100, SfxArrayV2.cpp:5868-5875) - profiles/r05_a_cand_hist.csv: a tenth of the suffix
// array holds 98 % of the windows a C2 step fetches.  Which tenth follows from the index alone: the runs of suffixes that share their
// first W bases (W = the core length of the reads' last phase, the shortest cores they are searched with; longer cores select parts of
// such runs).  A run of kSwMinRun .. max_run suffixes is covered whole, a longer one for its first kSwHead suffixes, a shorter one -
// the lane-per-candidate kernel's - not at all; coverage goes by blocks of 2^kSwBlkShift suffixes that hold any covered suffix.
// The same rule is applied for the core lengths of the reads' earlier phases (a 50-base core's interval is a run of ITS length, and the
// start of one that long lies anywhere inside the 25-base run around it): a block is covered when any level's rule covers it.
//   k_swin_breaks         per level: bit i = a run starts at i, as far as second-level keys and target tell
//   k_swin_bucket_starts  .. and as far as the k-mer table tells (first k bases differ)
//   k_swin_cover          per block: covered or not, level by level
//   k_swin_map / _fill    block numbers in the array (scan of the flags), the entries themselves
// Whatever these decide changes no result: an uncovered candidate takes its window from the 2-bit target as before.

// up to kSwLevels core lengths (ascending): level l's bitmap has bit i set when suffixes i - 1 and i share fewer than w[l] bases.  Levels
// up to k + 15 bases are read off the second-level keys; deeper ones compare the 2-bit target from base k + 15 on (two random lines per
// pair of neighbours - but only for the pairs that agree that far, a tenth of a genome's suffixes)
struct SwinLevels {
    int n;
    int w[kSwLevels];
    unsigned long long *brk[kSwLevels];
};

__device__ __forceinline__ uint64_t tgt2_bases32(const uint64_t *__restrict__ tgt2, uint64_t pos)
{
    const uint64_t wi = pos >> 5;
    const unsigned sh = (unsigned)(pos & 31) << 1;
    const uint64_t a = tgt2[wi], bq = tgt2[wi + 1];
    return (a << sh) | ((bq >> 1) >> (63 - sh));
}

// the suffix array indexes [a, e) (a multiple of 64): bit i - a of a level's bitmap.  The range is taken for itself - a run starts at a
// and one ends at e - so that an array can be made range by range, behind the suffix array's upload.  starts (optional): the bucket-start
// bitmap k_build_ktab left (else k_swin_bucket_starts adds those bits)
__global__ void __launch_bounds__(256) k_swin_breaks(DevIndex ix, SwinLevels lv, uint64_t a, uint64_t e, uint64_t n_words, const unsigned long long *__restrict__ starts)
{
    const int lane = threadIdx.x & 63;
    const uint64_t n = ix.n;
    const int k = ix.k, deep_from = k + kK2Bases;
    const int w_max = lv.w[lv.n - 1];
    for (uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_words; w += ((uint64_t)gridDim.x * blockDim.x) >> 6) {
        const uint64_t i = a + (w << 6) + lane;
        int shared = (i == a || i == e) ? 0 : 1 << 20;           // bases suffixes i - 1 and i share, as far as this kernel looks (past the end: no break)
        if (starts != nullptr && i < e && ((starts[i >> 6] >> (i & 63)) & 1)) shared = 0;
        if (shared && i > a && i < e) {
            const uint32_t a = ix.k2[i - 1], c = ix.k2[i];
            if (a == kK2Above || c == kK2Above) shared = 0;
            else {
                const uint32_t x = (a ^ c) & ~3u;
                if (x) shared = k + (__clz((int)x) >> 1);
                else if ((a | c) & 3u) shared = deep_from - 1;   // an N or a sequence end among the 15 bases: no deeper than the keys
                else if (w_max > deep_from) {
                    const bool wide_el = ix.sa_hi != nullptr;
                    const uint64_t pa = (wide_el ? sa_get<true>(ix, i - 1) : (uint64_t)ix.sa_lo[i - 1]) + (uint64_t)deep_from, pb = (wide_el ? sa_get<true>(ix, i) : (uint64_t)ix.sa_lo[i]) + (uint64_t)deep_from;
                    const int span = w_max - deep_from;
                    shared = deep_from;
                    if (pa + (uint64_t)span < n && pb + (uint64_t)span < n && !window_flagged(ix, pa, span) && !window_flagged(ix, pb, span)) {
                        for (int q = 0; q < span; q += 32) {
                            const uint64_t d = tgt2_bases32(ix.tgt2, pa + (uint64_t)q) ^ tgt2_bases32(ix.tgt2, pb + (uint64_t)q);
                            if (d) { shared += __clzll((long long)d) >> 1; break; }
                            shared += 32;
                        }
                    }
                }
            }
        }
        for (int l = 0; l < lv.n; l++) {
            const unsigned long long m = __ballot(i <= e && shared < lv.w[l]);
            if (lane == 0) lv.brk[l][w] = m;
        }
    }
}

__global__ void __launch_bounds__(256) k_swin_bucket_starts(DevIndex ix, uint64_t n_codes, SwinLevels lv, uint64_t a, uint64_t e)
{
    const int lane = threadIdx.x & 63;
    const uint64_t span = (n_codes + 63) & ~63ULL;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < span; c += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = ~0ULL, hi = ~0ULL;
        if (c < n_codes) {
            if (ix.ktab2) { lo = ix.ktab2[c].x; hi = ix.ktab2[c + 1].x; }
            else { lo = ktab_get(ix, c); hi = ktab_get(ix, c + 1); }           // (32- or 64-bit entries)
        }
        // neighbouring codes' buckets start in the same word of the bitmap: their bits are OR-ed along the lanes first (the starts
        // are non-decreasing), so that a word takes one atomic from the wave instead of twenty
        const bool in = c < n_codes && hi != lo && lo >= a && lo < e;
        const uint32_t w = in ? (uint32_t)((lo - a) >> 5) : 0xFFFFFFFFu;            // (a range holds fewer than 2^37 indexes)
        uint32_t bits = in ? 1u << (uint32_t)(lo & 31) : 0u;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t ob = __shfl_up(bits, off), ow = __shfl_up(w, off);
            if (lane >= off && ow == w) bits |= ob;
        }
        const uint32_t nw = __shfl_down(w, 1);
        if (bits && (lane == 63 || nw != w))
            for (int l = 0; l < lv.n; l++) atomicOr(reinterpret_cast<uint32_t *>(lv.brk[l]) + w, bits);
    }
}

// first set bit above position p, looking at no more than `limit` positions (p + 1 .. p + limit); 0 = none there
__device__ __forceinline__ uint64_t swin_next_break(const unsigned long long *__restrict__ brk, uint64_t p, uint64_t limit)
{
    const uint64_t last = p + limit;
    uint64_t q = p + 1;
    while (q <= last) {
        unsigned long long w = brk[q >> 6] >> (q & 63);
        if (w) { const uint64_t r = q + (uint64_t)(__ffsll(w) - 1); return r <= last ? r : 0; }
        q = (q | 63) + 1;
    }
    return 0;
}

// last set bit at or below p, looking down to p - limit; ~0 = none there
__device__ __forceinline__ uint64_t swin_prev_break(const unsigned long long *__restrict__ brk, uint64_t p, uint64_t limit)
{
    const uint64_t first = p > limit ? p - limit : 0;
    uint64_t q = p;
    for (;;) {
        unsigned long long w = brk[q >> 6] << (63 - (q & 63));
        if (w) { const uint64_t r = q - (uint64_t)__clzll((long long)w); return r >= first ? r : ~0ULL; }
        if ((q >> 6) == 0 || (q & ~63ULL) <= first) return ~0ULL;
        q = (q & ~63ULL) - 1;
    }
}

__global__ void __launch_bounds__(256) k_swin_cover(const unsigned long long *__restrict__ brk, uint64_t n, uint32_t max_run, uint32_t min_run, uint32_t *__restrict__ flags, uint64_t n_blocks, int first_level)
{
    constexpr uint64_t B = 1ULL << kSwBlkShift;
    for (uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; blk < n_blocks; blk += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b0 = blk << kSwBlkShift;
        bool cov = !first_level && flags[blk] != 0;           // (covered by an earlier level's rule)
        if (cov) continue;
        // the run the block's first suffix lies in
        const uint64_t s0 = swin_prev_break(brk, b0, max_run);
        if (s0 != ~0ULL) {
            const uint64_t e0 = swin_next_break(brk, b0, s0 + max_run - b0);        // (0: the run is longer than max_run)
            if (e0) cov = e0 - s0 >= min_run;
            else cov = b0 - s0 < kSwHead;
        }
        // the runs that start inside the block: only the last of them can be long enough (the others end inside the block)
        if (!cov) {
            const unsigned long long w = (brk[b0 >> 6] >> (b0 & 63)) & ((1ULL << B) - 1) & ~1ULL;
            if (w) {
                const uint64_t s = b0 + (uint64_t)(63 - __clzll((long long)w));
                if (s < n) cov = swin_next_break(brk, s, min_run - 1) == 0;
            }
        }
        flags[blk] = cov ? 1u : 0u;
    }
}

// incl: inclusive scan of the flags.  A covered block's number is its rank among the covered ones; blocks beyond the budget stay out
// used: covered blocks of the ranges before this one (device memory: no range waits for the host)
__global__ void __launch_bounds__(256) k_swin_map(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ incl, uint64_t n_blocks, uint32_t cap_blocks,
                                                  const uint32_t *__restrict__ used, uint32_t *__restrict__ map)
{
    const uint32_t base = *used;
    for (uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; blk < n_blocks; blk += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t slot = (uint64_t)base + incl[blk] - 1;
        map[blk] = (flags[blk] && slot < cap_blocks) ? (uint32_t)slot : kSwNone;
    }
}

__global__ void k_swin_advance(const uint32_t *__restrict__ incl_last, uint32_t cap_blocks, uint32_t *__restrict__ used)
{
    const uint64_t u = (uint64_t)*used + *incl_last;
    *used = u < cap_blocks ? (uint32_t)u : cap_blocks;
}

template <int E>
__global__ void __launch_bounds__(256) k_swin_fill(DevIndex ix, const uint32_t *__restrict__ map, uint4 *__restrict__ swin, uint64_t a, uint64_t e)
{
    for (uint64_t i = a + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = map[i >> kSwBlkShift];
        if (s == kSwNone) continue;
        swin_entry<E>(ix, i, swin, ((uint64_t)s << kSwBlkShift) + (i & ((1u << kSwBlkShift) - 1)));
    }
}

// w: n_levels core lengths, ascending; brk: a bitmap of n_words 64-bit words per level
// the suffix array indexes [a, e), a a multiple of 64; brk: a bitmap of n_words = (e - a) / 64 + 2 or more 64-bit words per level
void launch_swin_breaks(const DevIndex &ix, const int *w, int n_levels, unsigned long long *const *brk, uint64_t a, uint64_t e, uint64_t n_words,
                        const unsigned long long *starts, hipStream_t s)
{
    SwinLevels lv{};
    lv.n = n_levels;
    for (int l = 0; l < n_levels; l++) { lv.w[l] = w[l]; lv.brk[l] = brk[l]; }
    const unsigned blocks = (unsigned)std::min<uint64_t>((n_words + 3) / 4, 32768);
    hipLaunchKernelGGL(k_swin_breaks, dim3(blocks), dim3(256), 0, s, ix, lv, a, e, n_words, starts);
    if (starts == nullptr) {
        const uint64_t n_codes = 1ULL << (2 * ix.k);
        hipLaunchKernelGGL(k_swin_bucket_starts, dim3((unsigned)std::min<uint64_t>((n_codes + 255) / 256, 65536)), dim3(256), 0, s, ix, n_codes, lv, a, e);
    }
}

void launch_swin_cover(const unsigned long long *brk, uint64_t n, uint32_t max_run, uint32_t min_run, uint32_t *flags, uint64_t n_blocks, int first_level, hipStream_t s)
{
    const unsigned blocks = (unsigned)std::min<uint64_t>((n_blocks + 255) / 256, 262144);
    hipLaunchKernelGGL(k_swin_cover, dim3(blocks), dim3(256), 0, s, brk, n, max_run, min_run, flags, n_blocks, first_level);
}

// map: the n_blocks entries of the range's blocks; `used` moves on by the range's covered blocks
void launch_swin_map(const uint32_t *flags, const uint32_t *incl, uint64_t n_blocks, uint32_t cap_blocks, uint32_t *used, uint32_t *map, hipStream_t s)
{
    if (!n_blocks) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>((n_blocks + 255) / 256, 262144);
    hipLaunchKernelGGL(k_swin_map, dim3(blocks), dim3(256), 0, s, flags, incl, n_blocks, cap_blocks, used, map);
    hipLaunchKernelGGL(k_swin_advance, dim3(1), dim3(1), 0, s, incl + (n_blocks - 1), cap_blocks, used);
}

// adds the number of non-zero flags to *count
__global__ void __launch_bounds__(256) k_count_nonzero(const uint32_t *__restrict__ flags, uint64_t n, unsigned long long *__restrict__ count)
{
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ((n + 63) & ~63ULL); i += (uint64_t)gridDim.x * blockDim.x)
        mine += (unsigned long long)__popcll(__ballot(i < n && flags[i] != 0));
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
}

void launch_count_nonzero(const uint32_t *flags, uint64_t n, unsigned long long *count, hipStream_t s)
{
    if (!n) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, 16384);
    hipLaunchKernelGGL(k_count_nonzero, dim3(blocks), dim3(256), 0, s, flags, n, count);
}

// map: the whole map (indexed by suffix array index >> kSwBlkShift); entries of the indexes [a, e)
void launch_swin_fill(const DevIndex &ix, const uint32_t *map, void *swin, int words, uint64_t a, uint64_t e, hipStream_t s)
{
    if (e <= a) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>((e - a + 255) / 256, 65536);
    if (words == 5) hipLaunchKernelGGL(k_swin_fill<5>, dim3(blocks), dim3(256), 0, s, ix, map, reinterpret_cast<uint4 *>(swin), a, e);
    else hipLaunchKernelGGL(k_swin_fill<3>, dim3(blocks), dim3(256), 0, s, ix, map, reinterpret_cast<uint4 *>(swin), a, e);
}

__global__ void k_build_isa(const uint32_t *__restrict__ sa, uint64_t i0, uint64_t i1, uint32_t *__restrict__ isa)
{
    uint64_t i = i0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < i1; i += stride) isa[sa[i]] = (uint32_t)i;
}

__global__ void k_fill_u64(unsigned long long *__restrict__ p, uint64_t n, unsigned long long v)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
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

// (every table builder takes a range [i0, i1) of suffix array indexes; i1 = 0 stands for the whole array)
void launch_build_ktab(const DevIndex &ix, void *tab, int k, bool tab64, hipStream_t s, uint64_t i0, uint64_t i1, unsigned long long *starts, bool pairs)
{
    const int stride = (pairs && !tab64) ? 2 : 1;
    if (i0 & 63) starts = nullptr;
    if (i1 == 0) i1 = ix.n + 1;                      // (the entry past the last bucket comes with index n)
    if (i1 <= i0) return;
    uint64_t blocks = (i1 - i0 + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    bool wide = ix.sa_hi != nullptr;
    if (wide) {
        if (tab64) hipLaunchKernelGGL((k_build_ktab<true, uint64_t>), dim3((unsigned)blocks), dim3(256), 0, s, ix, (uint64_t *)tab, k, i0, i1, starts, stride);
        else hipLaunchKernelGGL((k_build_ktab<true, uint32_t>), dim3((unsigned)blocks), dim3(256), 0, s, ix, (uint32_t *)tab, k, i0, i1, starts, stride);
    } else {
        if (tab64) hipLaunchKernelGGL((k_build_ktab<false, uint64_t>), dim3((unsigned)blocks), dim3(256), 0, s, ix, (uint64_t *)tab, k, i0, i1, starts, stride);
        else hipLaunchKernelGGL((k_build_ktab<false, uint32_t>), dim3((unsigned)blocks), dim3(256), 0, s, ix, (uint32_t *)tab, k, i0, i1, starts, stride);
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

// A k-mer table of 64-bit bucket starts (an index beyond 2^32 suffixes: 34 GB at k = 16) as 32-bit offsets from the start of every
// 2^16-th code's bucket: half the bytes.  *overflow is set when a group of 2^16 codes spans 2^32 suffixes or more (the table then stays
// as it is)
__global__ void __launch_bounds__(256) k_pack_ktab64(const uint64_t *__restrict__ tab, uint64_t n_entries, uint32_t *__restrict__ off, uint64_t *__restrict__ hi,
                                                     uint32_t *__restrict__ overflow)
{
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_entries; c += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t base = tab[c & ~0xFFFFULL], v = tab[c] - base;
        if (v >> 32) atomicOr(overflow, 1u);
        off[c] = (uint32_t)v;
        if ((c & 0xFFFFULL) == 0) hi[c >> 16] = base;
    }
}

void launch_pack_ktab64(const uint64_t *tab, uint64_t n_entries, uint32_t *off, uint64_t *hi, uint32_t *overflow, hipStream_t s)
{
    hipLaunchKernelGGL(k_pack_ktab64, dim3(65536), dim3(256), 0, s, tab, n_entries, off, hi, overflow);
}

void launch_pack_target2(const uint64_t *tgt4, uint64_t nwords4, uint64_t *tgt2, unsigned int *nflag32, int flag_shift, hipStream_t s)
{
    uint64_t blocks = (nwords4 / 4 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_pack_target2, dim3((unsigned)blocks), dim3(256), 0, s, tgt4, nwords4, tgt2, nflag32, flag_shift);
}

// k-mer table entries {bucket start, y} (DevIndex::ktab2).  y of a bucket of one suffix: its second-level key - the line that names the
// bucket settles the search.  y of a larger bucket: which values the first five bits behind the k-mer take among its keys (bit v set:
// some suffix of the bucket continues with v; keys of suffixes with an N / sequence end inside the k-mer match nothing and set nothing) -
// a probe whose own five bits find no bit set has an empty interval, and the key line is not fetched to learn it (ktab2_absent)
__global__ void __launch_bounds__(256) k_make_ktab2(const uint32_t *__restrict__ tab, const uint32_t *__restrict__ k2, uint64_t n_entries, uint64_t n,
                                                    uint2 *__restrict__ out, const uint32_t *__restrict__ sa_elem)
{
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_entries; c += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t lo = tab[c];
        const uint64_t hi = c + 1 < n_entries ? (uint64_t)tab[c + 1] : (uint64_t)lo;
        uint32_t y = 0u;
        if (hi == (uint64_t)lo + 1) y = sa_elem != nullptr ? sa_elem[lo] : k2[lo];
        else if (hi > (uint64_t)lo + kTab2BitmapMax) y = 0xFFFFFFFFu;           // (a bucket this large has every bit set, or as good as)
        else
            for (uint64_t i = lo; i < hi; i++) {
                const uint32_t key = k2[i];
                if (key != kK2Above) y |= 1u << (key >> 27);
            }
        out[c] = make_uint2(lo, y);
    }
}

// the same for a table whose entries already are pairs with their first words - the bucket starts - in place (k_build_ktab, stride 2):
// the second words alone.  k2 == nullptr (the keys turned out unusable: the one-pass search reads the starts only): maps that hide nothing
__global__ void __launch_bounds__(256) k_fill_ktab2_y(uint2 *__restrict__ tab, const uint32_t *__restrict__ k2, uint64_t n_entries, const uint32_t *__restrict__ sa_elem)
{
    // sa_elem (optional): the suffix array - a bucket of one suffix then carries the suffix's element instead of its key (DevIndex::ktab2_elem)
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_entries; c += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t lo = tab[c].x;
        const uint64_t hi = c + 1 < n_entries ? (uint64_t)tab[c + 1].x : (uint64_t)lo;
        uint32_t y = 0u;
        if (k2 == nullptr) y = 0xFFFFFFFFu;
        else if (hi == (uint64_t)lo + 1) y = sa_elem != nullptr ? sa_elem[lo] : k2[lo];
        else if (hi > (uint64_t)lo + kTab2BitmapMax) y = 0xFFFFFFFFu;
        else
            for (uint64_t i = lo; i < hi; i++) {
                const uint32_t key = k2[i];
                if (key != kK2Above) y |= 1u << (key >> 27);
            }
        tab[c].y = y;
    }
}

void launch_fill_ktab2_y(void *tab2, const uint32_t *k2, uint64_t n_entries, hipStream_t s, const uint32_t *sa_elem)
{
    hipLaunchKernelGGL(k_fill_ktab2_y, dim3(65536), dim3(256), 0, s, reinterpret_cast<uint2 *>(tab2), k2, n_entries, sa_elem);
}

void launch_make_ktab2(const uint32_t *tab, const uint32_t *k2, uint64_t n_entries, uint64_t n, void *out, hipStream_t s, const uint32_t *sa_elem)
{
    hipLaunchKernelGGL(k_make_ktab2, dim3(65536), dim3(256), 0, s, tab, k2, n_entries, n, reinterpret_cast<uint2 *>(out), sa_elem);
}

void launch_build_k2(const DevIndex &ix, uint32_t *k2, uint32_t *k3, uint32_t *k4, unsigned long long *bad, hipStream_t s, uint64_t i0, uint64_t i1, bool write_k2)
{
    if (i1 == 0) i1 = ix.n;
    if (i1 <= i0) return;
    uint64_t blocks = (i1 - i0 + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    DevIndex t = ix;
    t.k2 = k2;
    t.kx[0] = k3;
    t.kx[1] = k3 ? k4 : nullptr;
    // the order check looks at pairs (i, i + 1): a range checks the pair that straddles its start, and leaves the one at its end to the next
    const uint64_t c0 = i0 ? i0 - 1 : 0, c1 = i1 == ix.n ? i1 : i1 - 1;
    if (ix.sa_hi) {
        hipLaunchKernelGGL(k_build_k2<true>, dim3((unsigned)blocks), dim3(256), 0, s, t, write_k2 ? k2 : nullptr, k3, t.kx[1] ? k4 : nullptr, i0, i1);
        hipLaunchKernelGGL(k_check_k2<true>, dim3((unsigned)blocks), dim3(256), 0, s, t, k2, bad, c0, c1);
    } else {
        hipLaunchKernelGGL(k_build_k2<false>, dim3((unsigned)blocks), dim3(256), 0, s, t, write_k2 ? k2 : nullptr, k3, t.kx[1] ? k4 : nullptr, i0, i1);
        hipLaunchKernelGGL(k_check_k2<false>, dim3((unsigned)blocks), dim3(256), 0, s, t, k2, bad, c0, c1);
    }
}

void launch_build_k2_levels(uint32_t *k2, uint64_t n, hipStream_t s)
{
    const uint64_t words = k2s_start(n, kK2Levels + 1) - n;
    uint64_t blocks = (words + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    hipLaunchKernelGGL(k_build_k2_levels, dim3((unsigned)blocks), dim3(256), 0, s, k2, n);
}

void launch_build_isa(const uint32_t *sa, uint64_t n, uint32_t *isa, hipStream_t s, uint64_t i0, uint64_t i1)
{
    if (i1 == 0) i1 = n;
    if (i1 <= i0) return;
    uint64_t blocks = (i1 - i0 + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    hipLaunchKernelGGL(k_build_isa, dim3((unsigned)blocks), dim3(256), 0, s, sa, i0, i1, isa);
}

}  // namespace bk
