// bk_extend.hip - LocateCoreMultiples (SfxArrayV2.cpp:5693-6262) for the calls without a long core interval (gfx950):
//   k_flat      block-cooperative: one candidate per lane, the outcome of a read reduced over its candidates' lanes
//   k_light     lane per read (register windows), kept selectable for cross-checks
//   k_extend    lane per read for reads beyond the register-window families
#include "bk_dev_window.h"
#include "bk_dev_prof.h"

namespace bk {

// ------------------------------------------------------------------------------------------------
// K2/K3: candidate walk + Hamming extension + classification, one lane per active read

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
// k_flat: reads of <= 16*NW bases whose core intervals are all <= heavy_thresh long - the same contract as k_extend (which stays for
// longer reads), organised so that every lane does the same amount of work; calls it cannot take go to the wave kernel or to the general one.
// A block owns 256 consecutive active reads.  Their candidates (every suffix of every core interval,
// in the reference's walk order strand -> core -> suffix) are numbered consecutively and EVALUATED
// one per lane - suffix array load, window compare, one result byte in LDS (mismatch count, or
// "skip": off the read's start / unverified bucket member that does not match / crosses an entry
// boundary / already reached through an earlier core).  The Low/NxtLow/instances outcome of a read is
// then reduced over its candidates' lanes (the state machine is order-independent up to its early exit,
// whose reads are replayed in order by their own lane; on 5-byte indexes the reference's truncated-key
// rule - a candidate is taken for seen when an earlier one of the strand pass has the same low word - is
// applied first, as a pass over the candidates' lanes).
// (A lane per read that walks all candidates of its read itself makes a wave run as long as its read with the most candidates - up to
// 4 x 64 - while the typical read has one or two: that form, k_light, left the tree in round 5.)
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

// Rows of the 2-bit read copy: a candidate's lane fetches its read's row (one strand: 32 bytes of a 100-base read) together with the
// candidate's window.  (They used to be staged in LDS, both strands of every read of the block fetched up front: 64 bytes for each of
// the many reads of a phase that turn out to have no candidate at all - three reads in four in phase 0 - and 16 KB of LDS per block.)
__host__ __device__ constexpr bool flat_rows_in_lds(bool wide, int nw, int bs) { return false; }
// .. but the result bytes of a pass stay at what that form left room for (the target starts of a pass's candidates take their place)
__host__ __device__ constexpr bool flat_small_pass(bool wide, int nw, int bs) { return !wide && nw <= 8 && bs <= 256; }

template <bool WIDE, int NW, int BS>
__global__ void __launch_bounds__(BS) k_flat(DevIndex ix, DevAlignCfg cfg, DevBatch b, const uint32_t *__restrict__ act,
                                              const uint32_t *__restrict__ p_n_act, int phase, int slots_max, StripeSet out, int have_wave)
{
    // (the active list's length lives in device memory - PhaseCtl; the launch covers a bound of it, a block beyond the list is done)
    const uint32_t n_act = *p_n_act;
    if ((uint64_t)blockIdx.x * BS >= n_act) return;
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
    constexpr bool SMALL = flat_small_pass(WIDE, NW, BS);
    constexpr uint32_t CAP = (SMALL ? kFlatCap / 4 : kFlatCap) * BS / 256;       // (LDS: four blocks per CU must fit 160 KB)
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
    __shared__ alignas(16) uint8_t s_rec[CAP];
    __shared__ uint4 s_row[ROWS ? BS * 2 * (NW / 4) : 1];  // [read][strand]: NW/2 words at 2 bit/base
    // 5-byte indexes: the reference's set of seen targets is keyed by the target start truncated to 32 bits (SfxArrayV2.cpp:5932), so a
    // candidate whose start lies a multiple of 2^32 bases from an earlier candidate of the same strand pass is taken for seen and
    // skipped.  The low words travel with the result bytes and the replay applies exactly that rule.
    __shared__ alignas(16) uint32_t s_key[WIDE ? CAP : 1];
    // 4-byte indexes: the target start of every candidate of the pass - the accepted read's locus is read from here instead of from
    // the suffix array again (a random line, and a dependent trip at the block's very end)
    __shared__ uint32_t s_t[SMALL ? CAP : 1];
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
                        if (raw & kElemFlag) lazy_bits |= 0x10000u << u;     // (a bucket of one suffix handed on as its suffix array element: the upper half, slot by slot)
                        const uint32_t cnt = raw & ~kIvFlags;
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
                    if ((cnt & kElemFlag) && cf) lazy_bits |= 0x10000u << q;       // (cf: at most sixteen slots; else the evaluation reads the records again)
                    cnt &= ~kIvFlags;
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
    uint32_t best_j = 0, best_x = 0, my_hit_t = 0;       // best_x: the best candidate's number among the read's; my_hit_t: its target start (SMALL)
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
                    bool lazy, elem;
                    if (cf) { iv_f = s_first[ri * slots_max + q]; lazy = ((s_lazy[ri] >> q) & 1) != 0; elem = ((s_lazy[ri] >> (16 + q)) & 1) != 0; }
                    else {
                        const int sti = q >= cmaxs ? 1 : 0;
                        uint32_t iv_c;
                        iv_get(b, iv_slot(b, blockIdx.x * blockDim.x + ri, s0 + sti, q - sti * cmaxs), iv_f, iv_c);
                        lazy = (iv_c & kLazyFlag) != 0;
                        elem = (iv_c & kElemFlag) != 0;
                    }
                    // (kElemFlag: the record holds the suffix itself - no trip to the suffix array in front of the window's)
                    tv[i] = elem ? iv_f : sa_get<WIDE>(ix, iv_f + j);
                    meta[i] = ri | ((uint32_t)q << 10) | (lazy ? kLazyBit : 0u);
                }
            }
            // ---- B: window start; the region flags and the window's blocks are requested, nothing waits for them here
            uint4 wv[KB][NBLK];
            uint4 rowv[KB][NW / 4];         // the read's 2-bit row of the candidate's strand (when it is not taken from LDS)
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
                            if (!ROWS && NW <= 8) {            // (the wider forms fetch the row when they compare: sixteen more live registers cost them more than the trip)
                                const int sti = q >= cmaxs ? 1 : 0;
                                const uint4 *__restrict__ rp = reinterpret_cast<const uint4 *>(b.rd2 + ((uint64_t)s_r[ri] * 2 + (uint64_t)(s0 + sti)) * (NW / 2));
#pragma unroll
                                for (int u = 0; u < NW / 4; u++) rowv[i][u] = rp[u];
                            }
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
                        } else if (NW <= 8) {
#pragma unroll
                            for (int u = 0; u < NW / 4; u++) {
                                r2w[2 * u] = ((uint64_t)rowv[i][u].y << 32) | rowv[i][u].x;
                                r2w[2 * u + 1] = ((uint64_t)rowv[i][u].w << 32) | rowv[i][u].z;
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
                    if (SMALL) s_t[f] = (uint32_t)t0;
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
                        best_q = q; best_j = x - prev; best_x = x;
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
                // (four earlier candidates per step: their keys as one 16-byte word, their result bytes as one dword)
                uint32_t i = rb + from;
                const uint32_t i1 = rb + local;
                for (; i < i1 && (i & 3) != 0; i++) seen |= s_rec[i] != kRecSkip && s_key[i] == kx;
                for (; i + 4 <= i1 && !seen; i += 4) {
                    const uint4 k4 = *reinterpret_cast<const uint4 *>(&s_key[i]);
                    const uint32_t r4 = *reinterpret_cast<const uint32_t *>(&s_rec[i]);
                    seen = (k4.x == kx && (r4 & 0xffu) != kRecSkip) || (k4.y == kx && ((r4 >> 8) & 0xffu) != kRecSkip) ||
                           (k4.z == kx && ((r4 >> 16) & 0xffu) != kRecSkip) || (k4.w == kx && (r4 >> 24) != kRecSkip);
                }
                for (; i < i1 && !seen; i++) seen = s_rec[i] != kRecSkip && s_key[i] == kx;
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
                        best_x = local;
                    }
                }
                if (SMALL && best_q >= 0) my_hit_t = s_t[s_off[t] - base + best_x];      // (this pass's candidates are still in LDS)
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
            const bool belem = cf ? ((s_lazy[t] >> (16 + best_q)) & 1) != 0 : (iv_count(b, iv_slot(b, a, st, c)) & kElemFlag) != 0;
            hit_left = SMALL ? (uint64_t)my_hit_t : (belem ? bf : sa_get<WIDE>(ix, bf + best_j)) - (uint64_t)ofs;
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

void launch_extend(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, uint32_t n_act,
                   int phase, uint32_t *next_act, uint32_t *next_cnt, uint32_t *heavy, uint32_t *heavy_cnt,
                   uint32_t *cmax_next, hipStream_t s)
{
    unsigned blocks = (n_act + 255) / 256;
    if (ix.sa_hi) hipLaunchKernelGGL(k_extend<true>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, n_act, phase, next_act, next_cnt, heavy, heavy_cnt, cmax_next);
    else hipLaunchKernelGGL(k_extend<false>, dim3(blocks), dim3(256), 0, s, ix, cfg, b, act, n_act, phase, next_act, next_cnt, heavy, heavy_cnt, cmax_next);
}

// stage: three buffers of at least n_act + (kListStripes + 2) * 1024 entries, stripe_cnt: kListStripes * 16 words, zero between launches
// n_act_bound: no more reads than this are on the active list (its length is *p_n_act, in device memory)
void launch_flat(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, const uint32_t *p_n_act, uint32_t n_act_bound, int phase,
                 int slots_max, uint32_t *next_act, uint32_t *next_cnt, uint32_t *heavy, uint32_t *heavy_cnt, uint32_t *wave,
                 uint32_t *wave_cnt, uint32_t *cmax_next, uint32_t *const *stage, uint32_t *stripe_cnt, int nw, hipStream_t s)
{
    constexpr int bs = 256;                            // reads (= threads) per block: 64 .. 1024 measured, 256 best (profiles/NOTES.md, round 2)
    nw &= 0xff;
    bool wide = ix.sa_hi != nullptr;
    if (!n_act_bound) return;
    unsigned blocks = (n_act_bound + (unsigned)bs - 1) / (unsigned)bs;
    if (slots_max < 1) slots_max = 1;
    size_t lds = (size_t)bs * slots_max * (flat_caches_first(wide, bs, slots_max) ? 6 : 2);
    StripeSet out;
    out.cnt = stripe_cnt;
    for (int i = 0; i < 3; i++) out.stage[i] = stage[i];
    out.cap = stripe_cap(blocks, (unsigned)bs);
    const int have_wave = wave != nullptr;
#define BK_FLAT(W, N) hipLaunchKernelGGL((k_flat<W, N, 256>), dim3(blocks), dim3(256), lds, s, ix, cfg, b, act, p_n_act, phase, slots_max, out, have_wave)
    if (nw <= 8) { if (wide) BK_FLAT(true, 8); else BK_FLAT(false, 8); }
    else if (nw <= 16) { if (wide) BK_FLAT(true, 16); else BK_FLAT(false, 16); }
    else if (nw <= kNwLong) { if (wide) BK_FLAT(true, kNwLong); else BK_FLAT(false, kNwLong); }
    else { if (wide) BK_FLAT(true, kNwLongest); else BK_FLAT(false, kNwLongest); }
#undef BK_FLAT
    // list order of the set: next phase, wave kernel, general kernel (the wave list may be absent: then it stays empty)
    uint32_t *dense[3] = {next_act, have_wave ? wave : heavy, heavy}, *total[3] = {next_cnt, have_wave ? wave_cnt : heavy_cnt, heavy_cnt};
    launch_compact(out, dense, total, 3, cmax_next, s);
}

#ifdef BK_PROF
int prof_read_flat(unsigned long long *out16) { return prof_read(out16); }
#endif

}  // namespace bk
