// bk_search.hip - LocateFirstExact (SfxArrayV2.cpp:7765-7876) + the extent of the matching run for every core of a phase (gfx950):
//   k_search          one pass over suffix array + target (indexes without second-level keys)
//   k_search_a_ilp    pass A: k-mer table + small buckets out of the second-level keys
//   k_search_b        pass B: bisection of the big buckets, work list grouped by bucket
#include <algorithm>

#include "bk_dev_k2.h"
#include "bk_dev_prof.h"

namespace bk {

// ------------------------------------------------------------------------------------------------
// K1: SA interval search, one lane per (active read, strand, core)

// With `lazy` set (register-window path), a core whose k-mer table bucket holds <= kLazyBucket suffixes
// is NOT bisected/verified here: the bucket is handed on as is (bit 31 of iv_n set) and the extend
// kernels keep only the members whose core bases are clean in the window they evaluate anyway -
// same candidates in the same SA order, two dependent HBM round trips fewer per probe.

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

// Pass A with ILP searches per lane, written stage by stage so that the loads of a stage (read row, k-mer table, second-level
// keys) of all ILP searches are in flight together: item u of a lane is search number tid + u * (lanes of the grid), i.e. every u
// maps neighbouring lanes to neighbouring searches.  Same records and work list for every ILP (order aside).

template <int ILP>
__global__ void __launch_bounds__(256) k_search_a_ilp(DevIndex ix, DevAlignCfg cfg, DevBatch b, const uint32_t *__restrict__ act,
                                                      const uint32_t *__restrict__ p_n_act, int phase, int cmax, int nstr, int lazy,
                                                      StripeSet out)
{
    __shared__ uint32_t s_cnt, s_base;
#if defined(BK_PROF) && BK_PROF == 2
    PROF_BEGIN;
#endif
    // the length of the active list lives in device memory (PhaseCtl); a block takes tiles of 256 * ILP searches until the list is done
    const uint32_t n_act = *p_n_act;
    const uint32_t per_read = (uint32_t)(nstr * cmax);
    const uint64_t total = (uint64_t)n_act * per_read;
    constexpr uint64_t stride = 256;                        // item u of a lane: tile start + lane + 256 u - neighbouring lanes, neighbouring searches
    const int k = ix.k;
  for (uint64_t tile0 = (uint64_t)blockIdx.x * (256 * ILP); tile0 < total; tile0 += (uint64_t)gridDim.x * (256 * ILP)) {
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
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
        const uint64_t tid = tile0 + threadIdx.x + (uint64_t)u * stride;
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
                    // (a plain count, or the record of a bucket of one suffix handed on as it is - kElemFlag: the same bucket, the same suffix,
                    // whatever the core's length - where unverified buckets may be handed on at all)
                    cached[u] = phase > 0 && cv[u].y != kNoIv32 && (cv[u].y < (1u << kKindShift) || (lazy && (cv[u].y & kIvFlags) == kIvFlags));
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
                lo[u] = e0.x; hi[u] = e1.x; key0[u] = e0.y;             // (a bucket of one: its key; a larger one: the map of its keys' first five bits)
            } else {
                ktab_get_pair(ix, code, lo[u], hi[u]);
            }
        }
    }
    PROFS(1);
    // stage 3: small buckets from the key array
    // (a lane loads the keys its bucket has - one to three for most - and no more: this kernel lives on the rate at which the
    // texture path takes lane requests, and sixteen keys for every lane cost a third more time than the search saved)
    uint32_t key[ILP][kInlineBucket];
    bool absent[ILP];                   // the table's line says that no key of the bucket continues the way the core does
    bool carry[ILP];                    // a bucket of one suffix whose table entry carries the suffix itself: handed on as it is (kElemFlag)
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        const uint64_t size = hi[u] - lo[u];
        const uint32_t m = k2_mask(cl[u] - k);
        absent[u] = have_code[u] && ix.ktab2 != nullptr && size >= 2 && ktab2_absent(key0[u], m, q2raw[u] & m);
        carry[u] = have_code[u] && lazy && ix.ktab2 != nullptr && ix.ktab2_elem && size == 1;
        const bool key_in_entry = size == 1 && ix.ktab2 != nullptr && !ix.ktab2_elem;
#pragma unroll
        for (uint32_t j = 0; j < kInlineBucket; j++)
            key[u][j] = (have_code[u] && !absent[u] && !carry[u] && size <= kInlineBucket && j < size) ? (key_in_entry ? key0[u] : ix.k2[lo[u] + j]) : kK2Above;
    }
    PROFS(2);
    // stage 4: results
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        uint2 leave = make_uint2(0, kNoIv32);               // what phase 0 leaves in iv32 for this read and strand
        if (cached[u] && (cv[u].y & kElemFlag)) {
            // phase 0 met a bucket of one suffix here and handed the suffix on: so does this phase
            first[u] = cv[u].x;
            nval[u] = cv[u].y;
            push[u] = false;
        } else if (cached[u]) {
            // the interval of the first k + 15 bases, as the bucket compare below would have produced it
            first[u] = cv[u].x;
            const uint32_t cnt = cv[u].y;
            if (cnt == 0 || cl[u] <= k + kK2Bases) { nval[u] = cnt; push[u] = false; }
            else if (lazy && cnt <= kLazyBucket) { nval[u] = cnt | kLazyFlag; push[u] = false; }
            else nval[u] = cnt | (kKindDeep << kKindShift);
        }
        if (carry[u]) {
            // (the table's line named the bucket and the suffix in it: no key line, no suffix array line - the window the extension fetches
            // says whether the core is there)
            first[u] = key0[u];
            nval[u] = 1u | kIvFlags;
            push[u] = false;
            leave = make_uint2(key0[u], 1u | kIvFlags);
        } else if (have_code[u]) {
            const uint64_t size = hi[u] - lo[u];
            if (size == 0 || absent[u]) { first[u] = lo[u]; nval[u] = 0; push[u] = false; leave = make_uint2((uint32_t)lo[u], 0u); }
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
    const uint64_t tile = tile0 / (256 * ILP);
    if (threadIdx.x == 0 && s_cnt) s_base = stripe_reserve_tile(out, 0, s_cnt, tile);
    __syncthreads();
    PROFS(5);
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        if (push[u]) stripe_put_tile(out, 0, s_base + my_off[u], (uint32_t)slot[u], tile);
        if (on[u] && nval[u] != 0) iv_put(b, slot[u], first[u], nval[u]);
    }
    __syncthreads();                                        // (s_cnt / s_base are the next tile's, too)
  }
#if defined(BK_PROF) && BK_PROF == 2
    PROFS(6);
    PROF_END;
#endif
}

template <bool WIDE>
__global__ void __launch_bounds__(256) k_search_b(DevIndex ix, DevAlignCfg cfg, DevBatch b, int phase, int lazy,
                                                  const uint32_t *__restrict__ list, const uint32_t *__restrict__ sorted, uint32_t n_sorted,
                                                  const uint32_t *__restrict__ p_n_list)
{
    // the work list's length lives in device memory; its first min(length, n_sorted) items are taken from `sorted` (the grouped copy:
    // the launch had to size the sort before the length was known), the others from the list as pass A left it - the order of the
    // items never changes a result
    __shared__ uint64_t s_lv[kK2Levels + 1];                // where the sampled levels of the keys start (k2s_start)
    if (threadIdx.x <= (unsigned)kK2Levels) s_lv[threadIdx.x] = threadIdx.x ? k2s_start(ix.n, (int)threadIdx.x) : 0;
    __syncthreads();
    const uint32_t n_list = *p_n_list;
    const uint32_t n_grouped = sorted != nullptr ? (n_list < n_sorted ? n_list : n_sorted) : 0u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_list; i += gridDim.x * blockDim.x) {
#ifdef BK_DIAG_B
    // counted per item kind, low word items, high word 64-byte lines: counter 5 = the bisections over the second- and the third-level
    // keys (items of both added up), counter 6 = what is left for suffix array + target after them
    unsigned long long d_k2 = 0, d_deep = 0, d_sa = 0;
    struct Fin { unsigned long long &a, &b, &c; DevBatch &bb; __device__ ~Fin() { if (a + b) atomicAdd(&bb.ctr[ctr_stripe() + 5], a + b); if (c) atomicAdd(&bb.ctr[ctr_stripe() + 6], c); } } fin{d_k2, d_deep, d_sa, b};
#endif
    const uint64_t slot = i < n_grouped ? sorted[i] : list[i];
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
        continue;
    }
    if (kind == kKindK2) {
        const uint32_t m = k2_mask(cl - k);
        const uint32_t q2 = squeeze2(rdw.nib16(my_ofs + k)) & m;
        // lower and upper bound through the sampled levels: a line per level and bound (bk_dev_k2.h)
        uint64_t l1, l2;
        unsigned long long k2_lines = 0;                    // (counted by -DBK_DIAG_B builds only)
        k2_bounds(ix.k2, s_lv, first, cnt, m, q2, l1, l2, k2_lines);
#ifdef BK_DIAG_B
        d_k2 += 1 + (k2_lines << 32);                       // (low word: items of this kind; high word: their lines)
#endif
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
            continue;
        }
    }
    // [first, first+cnt) agrees with the core on its first k+15 bases
    if (lazy && cnt <= kLazyBucket) {
        iv_put(b, slot, first, (uint32_t)cnt | kLazyFlag);
        continue;
    }
    int start = k + kK2Bases;
    // the next 15 bases from the third-level keys and the 15 after those from the fourth-level keys, the way the second-level keys gave
    // theirs (an N among them in the read: not expressible in 2 bits, the suffix array and the target take over there)
    bool settled = false;
    for (int lv = 0; lv < kMoreKeys && ix.kx[lv] != nullptr; lv++) {
        const uint32_t *__restrict__ kx = ix.kx[lv];
        const uint64_t p3 = rdw.nib16(my_ofs + start);
        if (p3 & 0x4444444444444444ULL & top_mask(cl - start < kK2Bases ? cl - start : kK2Bases)) break;
        const uint32_t m3 = k2_mask(cl - start);
        const uint32_t q3 = squeeze2(p3) & m3;
        uint64_t l1, l2;
        unsigned long long k3_lines = 0;
        k2_bounds(kx, s_lv, first, cnt, m3, q3, l1, l2, k3_lines);
#ifdef BK_DIAG_B
        d_deep += 1 + (k3_lines << 32);                     // (low word: items that consult the third-level keys; high word: their lines)
#endif
        {
            const int upto = cl < start + kK2Bases ? cl : start + kK2Bases;
            while (l2 > l1) {
                const uint32_t kv = kx[l2 - 1];
                if (!k2_nkind(kv) || cmp_core_from(rdw, my_ofs, upto, start, ix.tgt4, sa_get<WIDE>(ix, l2 - 1)) == 0) break;
                l2--;
            }
        }
        first = l1;
        cnt = l2 - l1;
        start += kK2Bases;
        if (cnt == 0 || cl <= start) {
            iv_put(b, slot, first, cnt > 0x7FFFFFFFULL ? 0x7FFFFFFFu : (uint32_t)cnt);
            settled = true;
            break;
        }
        if (lazy && cnt <= kLazyBucket) {
            iv_put(b, slot, first, (uint32_t)cnt | kLazyFlag);
            settled = true;
            break;
        }
    }
    if (settled) continue;
    {
        uint64_t l1 = first, h1 = first + cnt, l2 = first, h2 = first + cnt;
        while (l1 < h1 || l2 < h2) {
            const bool a1 = l1 < h1, a2 = l2 < h2;
            const uint64_t m1 = l1 + ((h1 - l1) >> 1), m2 = l2 + ((h2 - l2) >> 1);
#ifdef BK_DIAG_B
            d_sa += (2ULL << 32) * (a1 + (a2 && m1 != m2));            // (an element of the suffix array and a window of the target per probe)
#endif
            const uint64_t s1 = a1 ? sa_get<WIDE>(ix, m1) : 0, s2 = a2 ? sa_get<WIDE>(ix, m2) : 0;
            const int c1 = a1 ? cmp_core_from(rdw, my_ofs, cl, start, ix.tgt4, s1) : 0;
            const int c2 = a2 ? cmp_core_from(rdw, my_ofs, cl, start, ix.tgt4, s2) : 0;
            if (a1) { if (c1 > 0) l1 = m1 + 1; else h1 = m1; }
            if (a2) { if (c2 >= 0) l2 = m2 + 1; else h2 = m2; }
        }
        first = l1;
#ifdef BK_DIAG_B
        d_sa += 1;
#endif
        cnt = l2 - l1;
    }
    iv_put(b, slot, first, cnt > 0x7FFFFFFFULL ? 0x7FFFFFFFu : (uint32_t)cnt);
  }
}

void launch_search(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, uint32_t n_act,
                   int phase, int cmax, int nstr, int lazy, hipStream_t s)
{
    uint64_t threads = (uint64_t)n_act * (uint64_t)(cmax * nstr);
    unsigned blocks = (unsigned)((threads + 255) / 256);
    if (ix.sa_hi) hipLaunchKernelGGL(k_search<true>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, n_act, phase, cmax, nstr, lazy);
    else hipLaunchKernelGGL(k_search<false>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, n_act, phase, cmax, nstr, lazy);
}

// keys for grouping work items that touch the same part of the index (see bk_engine.cpp, sort_work):
// search items by the start of their k-mer bucket, wave items by the start of their longest core interval
__global__ void __launch_bounds__(256) k_keys_search(DevBatch b, const uint32_t *__restrict__ list, const uint32_t *__restrict__ p_n, uint32_t n_sort, int shift,
                                                     uint32_t *__restrict__ keys)
{
    // keys of the first min(*p_n, n_sort) items; what the sort was sized for beyond the list's real length sorts to the end
    const uint32_t n = *p_n;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_sort; i += gridDim.x * blockDim.x)
        keys[i] = i < n ? (uint32_t)(iv_start(b, list[i]) >> shift) : 0xFFFFFFFFu;
}

void launch_keys_search(const DevBatch &b, const uint32_t *list, const uint32_t *p_n, uint32_t n_sort, int shift, uint32_t *keys, hipStream_t s)
{
    if (!n_sort) return;
    unsigned blocks = (n_sort + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_keys_search, dim3(blocks), dim3(256), 0, s, b, list, p_n, n_sort, shift, keys);
}

// stage: at least n_act * cmax * nstr + (kListStripes + 2) * 1024 entries; stripe_cnt: kListStripes * 16 words, zero between launches
// n_act_bound: no more reads than this are on the active list (its length is *p_n_act, in device memory)
void launch_search_a(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, const uint32_t *p_n_act, uint32_t n_act_bound,
                     int phase, int cmax, int nstr, int lazy, uint32_t *list, uint32_t *list_cnt, uint32_t *stage, uint32_t *stripe_cnt,
                     hipStream_t s)
{
    uint64_t threads = (uint64_t)n_act_bound * (uint64_t)(cmax * nstr);
    if (!threads) return;
    lazy &= 0xff;
    constexpr int ilp = 2;                             // searches per lane: 1, 2, 4 measured (45.5 / 42.7 / 43.6 ms of search per C2 step, round 2)
    const uint64_t per = (uint64_t)256 * (uint64_t)ilp;
    const uint64_t tiles = (threads + per - 1) / per;
    const unsigned blocks = (unsigned)std::min<uint64_t>(tiles, 16384);       // (a block takes tiles until the list is done)
    StripeSet out;
    out.cnt = stripe_cnt;
    out.stage[0] = out.stage[1] = out.stage[2] = stage;
    out.cap = stripe_cap((unsigned)tiles, (unsigned)per);             // (stripes go by tile number)
    hipLaunchKernelGGL(k_search_a_ilp<2>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, p_n_act, phase, cmax, nstr, lazy, out);
    launch_compact(out, &list, &list_cnt, 1, nullptr, s);
}

// list: the work items as pass A left them (*p_n_list of them, at most n_bound); sorted / n_sorted: the grouped copy of the first n_sorted (or null)
void launch_search_b(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, int phase, int lazy, const uint32_t *list, const uint32_t *sorted,
                     uint32_t n_sorted, const uint32_t *p_n_list, uint64_t n_bound, hipStream_t s)
{
    if (!n_bound) return;
    unsigned blocks = (unsigned)std::min<uint64_t>((n_bound + 255) / 256, 32768);
    if (ix.sa_hi) hipLaunchKernelGGL(k_search_b<true>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, phase, lazy, list, sorted, n_sorted, p_n_list);
    else hipLaunchKernelGGL(k_search_b<false>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, phase, lazy, list, sorted, n_sorted, p_n_list);
}

// the interval records of the slots a phase can use, zeroed: [strand][core][position in the active list] for every core below cmax
__global__ void __launch_bounds__(256) k_clear_iv(DevBatch b, const uint32_t *__restrict__ p_n_act, int cmax, int st0, int st1)
{
    const uint32_t n_act = *p_n_act;
    const uint64_t per_strand = (uint64_t)cmax * n_act, total = per_strand * (uint64_t)(st1 - st0 + 1);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const int st = st0 + (int)(i / per_strand);
        const uint64_t q = i % per_strand;
        const uint64_t slot = iv_slot(b, (uint32_t)(q % n_act), st, (int)(q / n_act));
        if (b.iv2) b.iv2[slot] = make_uint2(0, 0);
        else b.iv_n[slot] = 0;
    }
}

void launch_clear_iv(const DevBatch &b, const uint32_t *p_n_act, uint32_t n_act_bound, int cmax, int st0, int st1, hipStream_t s)
{
    const uint64_t total = (uint64_t)n_act_bound * (uint64_t)cmax * (uint64_t)(st1 - st0 + 1);
    if (!total) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(k_clear_iv, dim3(blocks), dim3(256), 0, s, b, p_n_act, cmax, st0, st1);
}

}  // namespace bk

// in-kernel section timers (bk_dev_prof.h): BK_PROF == 1 reads k_flat's, 2 pass A's
namespace bk { int prof_read_flat(unsigned long long *out16); }
extern "C" int bk_debug_prof(unsigned long long *out16)
{
#if defined(BK_PROF) && BK_PROF == 2
    return bk::prof_read(out16);
#elif defined(BK_PROF)
    return bk::prof_read_flat(out16);
#else
    (void)out16;
    return 1;
#endif
}
