// bk_wave.hip - LocateCoreMultiples for repeat reads (gfx950): one wave per call, 64 candidates of a core interval per round.
#include "bk_dev_window.h"
#include "bk_dev_sets.h"

namespace bk {

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


// -DBK_CAND_HIST (a measurement build, tools/cand_hist.py): where in the suffix array the windows this kernel fetches lie - per block of
// 2^kHistShift suffix array indexes - and how they spread over the lengths of the core intervals they came from
#ifdef BK_CAND_HIST
constexpr int kHistShift = 6;
__device__ uint32_t *g_hist_blk;                      // windows fetched per block
__device__ unsigned long long g_hist_len[3][40];      // by floor(log2(interval length)): intervals, windows fetched, candidates processed
__device__ unsigned long long g_hist_ph[8][40][2];    // by phase and floor(log2(interval length)): windows fetched, .. of them from the window array
#endif

#ifndef BK_WAVE_NFLAG_LDS
#define BK_WAVE_NFLAG_LDS 1
#endif
constexpr int kNflagLds = 12288;

template <bool WIDE> struct PosT { typedef uint64_t type; };
template <> struct PosT<false> { typedef uint32_t type; };

struct WaveCoreInfo {
    unsigned long long first;
    uint32_t n;
    uint32_t walked;        // number of leading SA entries of the interval whose loop body was reached
    int ofs;
    uint32_t sw_base;       // SW: the window array's entry of the interval's first suffix ..
    uint32_t sw_n;          // .. and how many of the interval's leading suffixes have their entries in one run from there (0: none)
};

// HASH: the reference's own dedupe instead - a per-wave set of the 32-bit truncated target-start keys
// (SfxArrayV2.cpp:5932), kept in HBM with epoch tags as in k_heavy.  This is the form for 5-byte indexes (no
// inverse suffix array; and only the truncated keys reproduce the reference there, where two starts 2^32 apart
// count as one) and for 4-byte indexes whose inverse suffix array was not built.
// SW: the index holds the suffix-ordered window array (DevIndex::swin) - reads it covers take their candidates' windows from it.
// (16-word inverse-suffix-array forms: five waves per SIMD at 96 registers and a dozen of them spilled beat four at 116 - 2 x 150's
// k_wave 77.1 -> 70.5 ms per step, round 5; the hash-set forms lose by the same move, round 4)
#ifndef BK_WAVE16_BLOCKS
#define BK_WAVE16_BLOCKS 5
#endif
// (the 8-word window-array form: eight waves per SIMD at 64 registers - two spilled - against seven at 70: C2's k_wave 31.0 -> 29.6 ms; the form
// without the array has its eight waves at 53 registers anyway and lost 4 % under the same bound)
#ifndef BK_WAVE8_BLOCKS
#define BK_WAVE8_BLOCKS 8
#endif
#if defined(BK_PROF) && BK_PROF == 3
// where a wave's cycles go (lane 0 of every wave; sums over the launch): 0 claiming an item and its read's plan, 1 a strand pass's set-up
// (read row, interval records, window array map), 2 a core's set-up, 3 the rounds, 4 a read's end; 5 = strand passes, 6 = rounds
static __device__ unsigned long long g_wprof[64 * 8];
#define WPROF_DECL long long wp_last = clock64(); unsigned long long wp[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define WPROF(k) do { const long long now_ = clock64(); wp[k] += (unsigned long long)(now_ - wp_last); wp_last = now_; } while (0)
#define WPROF_N(k) wp[k]++
#define WPROF_END do { if (lane == 0) for (int k_ = 0; k_ < 8; k_++) atomicAdd(&g_wprof[(blockIdx.x & 63) * 8 + k_], wp[k_]); } while (0)
#define RPROF(k) do { } while (0)
#define RPROF_N(k) do { } while (0)
#elif defined(BK_PROF) && BK_PROF == 4
// where a ROUND's cycles go (lane 0 of every wave, every mark behind a wait for what is in flight): 0 the suffix array elements in hand,
// 1 windows fetched and compared, 2 the look-up in the set of seen keys, 3 the keys' insertion, 4 the round's bookkeeping; 5 = rounds,
// 6 = rounds of a strand pass that lives in the HBM table, 7 = cycles outside the rounds
static __device__ unsigned long long g_wprof[64 * 8];
#define WPROF_DECL long long wp_last = clock64(); unsigned long long wp[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define WPROF(k) do { const long long now_ = clock64(); wp[7] += (unsigned long long)(now_ - wp_last); wp_last = now_; } while (0)
#define WPROF_N(k) do { if ((k) == 6) wp[5]++; } while (0)
#define RPROF(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long now_ = clock64(); wp[k] += (unsigned long long)(now_ - wp_last); wp_last = now_; } while (0)
#define RPROF_N(k) wp[k]++
#define WPROF_END do { if (lane == 0) for (int k_ = 0; k_ < 8; k_++) atomicAdd(&g_wprof[(blockIdx.x & 63) * 8 + k_], wp[k_]); } while (0)
#else
#define RPROF(k) do { } while (0)
#define RPROF_N(k) do { } while (0)
#define WPROF_DECL do { } while (0)
#define WPROF(k) do { } while (0)
#define WPROF_N(k) do { } while (0)
#define WPROF_END do { } while (0)
#endif

// the kernel's arguments as one structure
struct WaveArgs {
    DevIndex ix; DevAlignCfg cfg; DevBatch b; HeavyScratch hs;
    const uint32_t *list, *sorted;
    uint32_t n_sorted;
    const uint32_t *p_n_list;
    int phase;
    uint32_t *cursor, *next_act, *next_cnt, *cmax_next;
};
template <int NW, bool WIDE, bool HASH, bool SW, bool GROUP>
__global__ void __launch_bounds__(256, NW <= 8 ? ((HASH || !SW) ? 4 : BK_WAVE8_BLOCKS) : ((NW == 16 && !HASH) ? BK_WAVE16_BLOCKS : ((NW == 16 && SW) ? 4 : 2))) k_wave(WaveArgs wa)
{
    const DevIndex &ix = wa.ix;
    const DevAlignCfg &cfg = wa.cfg;
    const DevBatch &b = wa.b;
    const HeavyScratch &hs = wa.hs;
    const uint32_t *__restrict__ list = wa.list, *__restrict__ sorted = wa.sorted, *__restrict__ p_n_list = wa.p_n_list;
    const uint32_t n_sorted = wa.n_sorted;
    const int phase = wa.phase;
    uint32_t *__restrict__ cursor = wa.cursor, *__restrict__ next_act = wa.next_act, *__restrict__ next_cnt = wa.next_cnt, *__restrict__ cmax_next = wa.cmax_next;
    // Both strand passes of a read are set up at once - interval records, window array map, 2-bit rows: one chain of dependent trips per
    // read where a chain per strand pass left the wave without work twice - into the wave's own words of LDS
    __shared__ WaveCoreInfo s_core[4][2][kMaxCoresFast];
    __shared__ uint64_t s_row2[4][2][NW / 2];                    // the read's 2 bit/base rows, first and second strand pass
    // .. and the work items are resolved a claim at a time: lane l follows item l of the claim through list -> active list -> read
    // metadata, kGrabMax chains side by side where every item used to walk its own
    __shared__ uint32_t s_grab[4][3][kWaveGrab];
    __shared__ uint64_t s_cmask[4][kMaxCoresFast][NW / 4];       // per core: its bases in the IWindow layout
    // NFL: the block's own copy of the "region holds N / sequence end" bitmap (<= kNflagLds bytes: a 3.1 Gbp index has 12 KB).  A round's
    // chain of dependent trips is suffix array element -> the two flag bytes of the window it names -> compare; the flag bytes came out
    // of the L1, which is still a trip of the vector memory path, in order behind every load the wave has in flight.  From LDS they cost
    // a hundred cycles and wait for nothing else.
    constexpr bool NFL = SW && !HASH && BK_WAVE_NFLAG_LDS;
    __shared__ uint32_t s_nflag[NFL ? kNflagLds / 4 : 1];
    const bool nfl = NFL && ix.nflag != nullptr && ix.nflag_bytes <= (uint32_t)kNflagLds;
    if (NFL && nfl) {
        for (uint32_t i = threadIdx.x; i < ix.nflag_bytes / 4; i += blockDim.x) s_nflag[i] = reinterpret_cast<const uint32_t *>(ix.nflag)[i];
        __syncthreads();
    }
    // HASH: the set of seen target keys of a strand pass lives in LDS (kLdsSet keys per wave, open addressing) and spills into the
    // wave's HBM table only when a pass inserts more than kLdsSetFill keys - a look-up and an insert in HBM are two or three more
    // random cache lines (and a compare-and-swap) on the dependent chain of every candidate
    __shared__ __attribute__((aligned(16))) uint32_t s_set[HASH ? 4 : 1][HASH ? kLdsSet : 4];
    // the sequences' first and last base (up to 64 of them): a finished read finds its sequence without a trip to HBM at its very end
    __shared__ uint64_t s_es[64], s_ee[64];
    if (ix.n_ent <= 64) {
        if (threadIdx.x < ix.n_ent) { s_es[threadIdx.x] = ix.ent_start[threadIdx.x]; s_ee[threadIdx.x] = ix.ent_end[threadIdx.x]; }
        __syncthreads();
    }
    // the list's length lives in device memory (PhaseCtl); its first min(length, n_sorted) items come from the sorted copy, whose size the
    // launch had to fix before the length was known - the order of the items never changes a result
    const uint32_t n_list = *p_n_list;
    const uint32_t n_grouped = sorted != nullptr ? (n_list < n_sorted ? n_list : n_sorted) : 0u;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // (the compiler does not know it for wave-uniform: addresses in the wave's LDS words would each keep a vector register)
    WaveCoreInfo *core = s_core[wib][0];
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
    // .. but no more of them than leaves every wave several claims to make: the list comes longest job first, and eight items at a time
    // hand the eight heaviest reads of a short list to ONE wave (a batch of 3 M reads then lasts as long as those eight, one after the other)
    uint32_t grab_next = 0, grab_left = 0, grab_at = 0;
    const uint32_t per_wave = n_list / (gridDim.x * 4u * 4u);
    const int grab = per_wave >= (uint32_t)kWaveGrab ? kWaveGrab : (per_wave < 1 ? 1 : (int)per_wave);
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
    WPROF_DECL;
    for (;;) {
        if (grab_left == 0) {
            uint32_t g = 0;
            if (lane == 0) g = atomicAdd(cursor, (uint32_t)grab);
            grab_next = __builtin_amdgcn_readfirstlane(g);
            grab_left = (uint32_t)grab;
            grab_at = 0;
            __builtin_amdgcn_wave_barrier();
            // (the lane's number through an opaque move: what is derived from it here - addresses in the wave's LDS words - is then made
            // where it is used, not kept in vector registers across the whole item loop as loop invariants)
            int lg = lane;
            asm volatile("" : "+v"(lg));
            if (lg < grab && grab_next + (uint32_t)lg < n_list) {
                const uint32_t it = grab_next + (uint32_t)lg;
                const uint32_t gp = it < n_grouped ? sorted[it] : list[it];       // position in the phase's active list: where its interval records lie
                const uint32_t gr = b.act[gp];
                s_grab[wib][0][lg] = gp;
                s_grab[wib][1][lg] = gr;
                s_grab[wib][2][lg] = b.rmeta[gr];
            }
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t item = grab_next++;
        grab_left--;
        if (item >= n_list) break;
        // (wave-uniform values that arrive through vector loads are handed to the scalar unit explicitly: the read's plan, its loop
        // bounds and the window geometry then cost scalar instructions once instead of vector instructions in every lane)
        const uint32_t pos = __builtin_amdgcn_readfirstlane(s_grab[wib][0][grab_at]);
        const uint32_t r = __builtin_amdgcn_readfirstlane(s_grab[wib][1][grab_at]);
        const uint32_t meta = __builtin_amdgcn_readfirstlane(s_grab[wib][2][grab_at]);
        grab_at++;
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
                s_core[wib][0][lane].ofs = o;
                s_core[wib][1][lane].ofs = o;
            }
            __builtin_amdgcn_wave_barrier();
            const int ncm = geo_nc < kMaxCoresFast ? geo_nc : kMaxCoresFast;
            for (int idx = lane; idx < ncm * (NW / 4); idx += 64) {
                const int cc = idx / (NW / 4), i = idx % (NW / 4);
                const int o = s_core[wib][0][cc].ofs;
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
        constexpr int E = NW <= 8 ? 3 : 5;                     // 16-byte words of a window array entry of this kernel family (SwGeo)
        // every core offset of the read lies within an entry's lead, and the read's end within the entry; shared rounds take their
        // windows from three-word entries only
        const bool sw_read = SW && E == 3 && len <= SwGeo<E>::bases - SwGeo<E>::pre && len - cl <= SwGeo<E>::pre;
        const bool two_bit = b.rd2 != nullptr;
        WPROF(0);
        {
            // lanes 0 .. 15: the cores of the first strand pass, 16 .. 31: of the second; lanes 32 ..: the words of the two rows
            int ls = lane;
            asm volatile("" : "+v"(ls));                // (as in the claim: nothing derived from the lane's number here outlives the block)
            const int sti_l = ls >> 4, c_l = ls & 15, st_l = s0 + sti_l;
            if (ls < 32 && c_l < nc && st_l <= s1) {
                uint64_t f;
                uint32_t cn;
                iv_get(b, iv_slot(b, pos, st_l, c_l), f, cn);
                WaveCoreInfo &ci = s_core[wib][sti_l][c_l];
                ci.first = f;
                ci.n = cn;                    // bit 31: unverified bucket (<= kLazyBucket members)
                ci.walked = 0;
                const uint32_t cnt_l = cn & ~kIvFlags;
                if (SW) {
                    // which of the interval's candidates take their windows from the window array: all of them when it holds every suffix;
                    // else (DevIndex::swmap) the interval when its first and last block lie as far apart in the array as in the suffix
                    // array - every block between them is there too, in order -, or, failing that, its first kSwHead suffixes by the
                    // same test (an interval beyond MaxIter is left after a hundred-odd candidates)
                    uint32_t sb = (uint32_t)f, sn = cnt_l;
                    if (cn & kElemFlag) sn = 0;         // (the record holds the suffix itself, not its place in the suffix array: its window comes from the target)
                    else if (ix.swmap != nullptr) {
                        sn = 0;
                        if (cnt_l) {
                            const uint32_t head = cnt_l < kSwHead ? cnt_l : kSwHead;
                            const uint32_t b0 = (uint32_t)(f >> kSwBlkShift), bl = (uint32_t)((f + cnt_l - 1) >> kSwBlkShift), bh = (uint32_t)((f + head - 1) >> kSwBlkShift);
                            const uint32_t m0 = ix.swmap[b0], ml = ix.swmap[bl], mh = ix.swmap[bh];
                            if (m0 != kSwNone) {
                                sb = (m0 << kSwBlkShift) + ((uint32_t)f & ((1u << kSwBlkShift) - 1));
                                if (ml != kSwNone && ml - m0 == bl - b0) sn = cnt_l;
                                else if (mh != kSwNone && mh - m0 == bh - b0) {
                                    const uint64_t upto = ((uint64_t)(bh + 1) << kSwBlkShift) - f;
                                    sn = upto < cnt_l ? (uint32_t)upto : cnt_l;
                                }
                            }
                        }
                    }
                    ci.sw_base = sb;
                    ci.sw_n = sn;
                }
            }
            if (two_bit && ls >= 32 && ls < 32 + NW) {
                const int q = ls - 32, sti_r = q / (NW / 2), k = q % (NW / 2);
                if (s0 + sti_r <= s1) s_row2[wib][sti_r][k] = b.rd2[((uint64_t)r * 2 + (uint64_t)(s0 + sti_r)) * (NW / 2) + k];
            }
            __builtin_amdgcn_wave_barrier();
        }
        WPROF(1);
        for (int st = s0; st <= s1 && !done; st++) {
            core = s_core[wib][st - s0];
            WPROF(2);
            WPROF_N(5);
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
            const RdRow row4 = read_row(b, r, st, has_n);
#pragma unroll
            for (int k = 0; k < NW / 2; k++) r2w[k] = two_bit ? uniform64(s_row2[wib][st - s0][k]) : 0ULL;
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
            const uint32_t my_cn = lane < nc ? core[lane].n & ~kIvFlags : 0u;       // lane l < nc: suffixes in core l's interval
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
                const bool elem = (cn_c & kElemFlag) != 0;               // the record's start is the one suffix's array ELEMENT (its target position)
                const uint32_t n = cn_c & ~kIvFlags;                     // (an interval's count stays below 2^30)
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
                bool llazy = lazy, lelem = elem;
                uint32_t lj_g = 0;
                if (GROUP && grouped) {
                    for (int l = c + 1; l < ce; l++) lc += (__shfl(pre_ex, l) - pre_c) <= (uint32_t)lane ? 1 : 0;
                    lj_g = (uint32_t)lane - (__shfl(pre_ex, lc) - pre_c);
                    lfirst = core[lc].first;
                    lofs = core[lc].ofs;
                    llazy = (core[lc].n & kLazyFlag) != 0;
                    lelem = (core[lc].n & kElemFlag) != 0;
                }
                // SW: the window array's entry of this lane's first candidate, and how far the array serves the interval
                uint32_t lsw_base = 0, sw_n_c = 0;
                bool grp_sw = false;
                if (SW) {
                    lsw_base = (GROUP && grouped) ? core[lc].sw_base : (uint32_t)__builtin_amdgcn_readfirstlane(core[c].sw_base);
                    sw_n_c = __builtin_amdgcn_readfirstlane(core[c].sw_n);
                    // a shared round goes to the array when it serves every one of its cores whole
                    if (GROUP && grouped) grp_sw = __ballot((uint32_t)lane < gtot && core[lc].sw_n < (core[lc].n & ~kIvFlags)) == 0;
                }
                n_search += (unsigned long long)(ce - c);
                uint32_t iter = 0;
                bool copies_checked = false;
                uint32_t walked = n;
                // the window array serves a core when the read's whole window lies inside the candidate's entry: bases kSwPre - ofs ..
                // + len of its kSwBases (every core of a read of up to kSwLen bases; of a longer read - 2 x 150 - the cores in the
                // middle, when they are walked a round per 64 suffixes; rounds shared by several cores of such a read go to the target)
                const bool sw_core = SW && ((GROUP && grouped) ? (sw_read && grp_sw) : (ofs <= SwGeo<E>::pre && len - ofs <= SwGeo<E>::bases - SwGeo<E>::pre && sw_n_c != 0));
                // the entry's 16-byte words (64 bases each) the window lies in; a shared round's cores differ in it: every word
                const int sw_q0 = (GROUP && grouped) ? 0 : (SwGeo<E>::pre - ofs) >> 6, sw_q1 = (GROUP && grouped) ? E - 1 : (SwGeo<E>::pre - ofs + len - 1) >> 6;
                WPROF(2);
                using P = typename PosT<WIDE>::type;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
#pragma clang diagnostic ignored "-Wconditional-uninitialized"
                auto request = [&](uint32_t jj0, bool &sw_r, uint4 (&e)[E], P &lo) __attribute__((always_inline)) {
                    const uint32_t jj = (GROUP && grouped) ? lj_g : jj0 + (uint32_t)lane;
                    const bool act = (GROUP && grouped) ? (uint32_t)lane < gtot : jj < n;
                    // (a round takes its windows from the array when every candidate of it has its entry there)
                    sw_r = SW && sw_core && ((GROUP && grouped) || (jj0 + 64 < n ? jj0 + 64 : n) <= sw_n_c);
                    // (the candidate's entry of the window array is requested together with its suffix array element: one round trip - the
                    // element FIRST: the compiler waits for the entry's last word right where it is issued, and an element asked for
                    // behind that left a trip later, 3 % of the kernel's time (profiles/r06_g_kwave_ab.txt).  Target positions of an index
                    // of 4-byte elements: 32-bit arithmetic; a lane without a candidate leaves the words undefined - nothing of them is
                    // looked at below)
                    lo = act ? (lelem ? (P)lfirst : (P)sa_get<WIDE>(ix, lfirst + jj)) : (P)0;
                    if (SW && sw_r && act) {
                        // (only the words the core's window reaches into: which, is the same for every candidate of the core)
                        const uint32_t eidx = lsw_base + jj;
                        const uint4 *__restrict__ ep = ix.swin + ((uint64_t)(eidx >> 5) * (32u * E) + (eidx & 31u));
#pragma unroll
                        for (int q = 0; q < E; q++)
                            if (q >= sw_q0 && q <= sw_q1) e[q] = ep[q * 32];
                    }
                };
                for (uint32_t j0 = 0; j0 < ((GROUP && grouped) ? 1u : n) && !done; j0 += 64) {
                    WPROF_N(6);
                    const uint32_t j = (GROUP && grouped) ? lj_g : j0 + (uint32_t)lane;
                    const bool active = (GROUP && grouped) ? (uint32_t)lane < gtot : j < n;
                    uint4 ev[E];
                    P loci;
                    bool sw_now;
                    request(j0, sw_now, ev, loci);
                    RPROF(0);
                    const P t = loci - (P)lofs;
                    bool valid = active && loci >= (P)lofs;
#ifdef BK_CAND_HIST
                    {
                        const uint32_t ln = core[lc].n & ~kIvFlags;
                        const int lb = ln ? 31 - __clz((int)ln) : 0;
                        if (valid) {
                            if (g_hist_blk) atomicAdd(&g_hist_blk[(lfirst + j) >> kHistShift], 1u);
                            atomicAdd(&g_hist_len[1][lb], 1ULL);
                        }
                        if (active && j == 0) atomicAdd(&g_hist_len[0][lb], 1ULL);
                    }
#endif
#ifdef BK_CAND_HIST
                    if (valid) {
                        const uint32_t ln = core[lc].n & ~kIvFlags;
                        const int lb = ln ? 31 - __clz((int)ln) : 0;
                        atomicAdd(&g_hist_ph[phase < 8 ? phase : 7][lb][0], 1ULL);
                        if (SW && sw_now) atomicAdd(&g_hist_ph[phase < 8 ? phase : 7][lb][1], 1ULL);
                    }
#endif
                    IWindow<NW> w;
                    w.mm = 127; w.eos = true;                             // (the map of a lane without a valid candidate stays undefined: every use is behind `valid`)
                    if (valid) {
                        // the block-flag load and the 2-bit window loads are issued together; only the rare
                        // flagged window is then fetched again from the 4-bit copy
                        if (two_bit) {
                            bool flg;
                            if (NFL && nfl) {
                                // (4-byte indexes: 32-bit arithmetic, as window_flagged_t)
                                const uint32_t t0 = (uint32_t)t;
                                uint32_t t1 = t0 + (uint32_t)(len - 1);
                                t1 = t1 < t0 ? 0xFFFFFFFFu : t1;
                                const uint32_t g0 = t0 >> ix.flag_shift, g1 = t1 >> ix.flag_shift;
                                const uint8_t *fl = reinterpret_cast<const uint8_t *>(s_nflag);
                                flg = ((((uint32_t)fl[g0 >> 3] >> (g0 & 7)) | ((uint32_t)fl[g1 >> 3] >> (g1 & 7))) & 1) != 0;
                            } else
                                flg = window_flagged_t<WIDE>(ix, t, len);
                            if (SW && sw_now) {
                                if constexpr (SW) {
                                    if (GROUP && grouped) eval_swin2i<NW, false, E>(r2w, rni, len, ev, SwGeo<E>::pre - lofs, w);
                                    else eval_swin2i<NW, true, E>(r2w, rni, len, ev, SwGeo<E>::pre - lofs, w);
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
                    RPROF(1);
                    bool dup = false;
                    const uint32_t key = (uint32_t)(1 + loci - (uint32_t)lofs);       // 32-bit truncation as :5932
                    const bool fused = HASH && spilled;          // the pass lives in the HBM table since an earlier round: look-up and insert are one walk
                    uint32_t hslot = 0xFFFFFFFFu;               // .. and this is the slot this lane's key went into
                    if (HASH) {
                        dup = valid && lset_n != 0 && lset_contains(lset, key);
                        // (a round against itself: two starts 2^32 apart in one interval; the same start reached through two cores of a group)
                        if (WIDE || (GROUP && grouped)) dup |= same_key_earlier_in_round(valid && !dup, key, lane);
                        if (fused && valid && !dup) dup = htab_find_or_insert(tab, tmask, epoch, key, hslot);
                    } else for (int c2 = 0; c2 < ce - 1; c2++) {
                        bool m = valid && !dup && c2 < lc && im_clean<NW>(w.im, cmask[c2]);
                        if (__ballot(m)) {
                            if (m) {
                                if (core[c2].walked >= (core[c2].n & ~kIvFlags)) dup = true;
                                else {
                                    uint64_t rank = (uint64_t)ix.isa[(P)(t + (P)core[c2].ofs)] - core[c2].first;
                                    dup = rank < (uint64_t)core[c2].walked;
                                }
                            }
                        }
                    }
                    const bool isnew = valid && !dup;
                    RPROF(2);
                    if (HASH && spilled) RPROF_N(6);
                    if (kDiag) { n_fetch += __popcll(__ballot(active && loci >= (uint64_t)ofs)); n_dup += __popcll(__ballot(dup)); }
                    const uint64_t newmask = __ballot(isnew);
                    const uint32_t pre = (uint32_t)__popcll(newmask & lt_mask);
                    const uint32_t iter_before = iter + pre;
                    const uint32_t nodes_before = nodes + pre;
                    bool stop = active && !(GROUP && grouped) && ((cfg.max_iter && iter_before >= (uint32_t)cfg.max_iter) || nodes_before >= kNodeCap);
                    uint32_t cutoff = (GROUP && grouped) ? 64u : n;
                    uint64_t stopmask = __ballot(stop);
                    if (stopmask) cutoff = j0 + (uint32_t)(__ffsll((unsigned long long)stopmask) - 1);
                    if (!copies_checked && !(GROUP && grouped)) {
                        bool chk = active && j > 0 && iter_before == 100;
                        uint64_t chkmask = __ballot(chk);
                        if (chkmask) {
                            uint32_t jc = j0 + (uint32_t)(__ffsll((unsigned long long)chkmask) - 1);
                            if (jc < cutoff) {
                                copies_checked = true;
                                uint32_t num_copies = n - jc + 2;
                                if (cfg.max_iter && num_copies > (uint32_t)cfg.max_iter) cutoff = jc;
                            }
                        }
                    }
                    const bool proc = active && ((GROUP && grouped) || j < cutoff) && isnew;
                    if (fused) {
                        // (the keys are in the table; that of a candidate the cuts above left unprocessed comes out again)
                        if (hslot != 0xFFFFFFFFu && !proc) htab_retract(tab, hslot, epoch);
                        __builtin_amdgcn_wave_barrier();
                    } else if (HASH) {
                        const uint32_t nins = (uint32_t)__popcll(__ballot(proc));
                        if (nins) {
                            // (the key that looks like an empty slot, one in 2^32, always goes to the HBM table)
                            const bool to_lds = lset_n + nins <= kLdsSetFill;
                            if ((!to_lds || __ballot(proc && key == kLdsEmpty)) && !spilled) {
                                spilled = true;
                                epoch++;
                                if (epoch >= kTombBit) {     // wrapped: really clear the table
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
                    RPROF(3);
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
#ifdef BK_CAND_HIST
                    if (proc && ((keep >> lane) & 1)) { const uint32_t ln = core[lc].n & ~kIvFlags; atomicAdd(&g_hist_len[2][ln ? 31 - __clz((int)ln) : 0], 1ULL); }
#endif
                    uint64_t procmask = __ballot(proc) & keep;
                    uint32_t nproc = (uint32_t)__popcll(procmask);
                    iter += nproc;
                    nodes += nproc;
                    n_cand += nproc;                                        // (the same in every lane: a scalar)
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
                            hit_left = (uint64_t)__shfl(t, fl);
                            hit_strand = st ? '-' : '+';
                        } else if (bmin == low_mm) {
                            low_inst += cnt;
                            if (bsec < nxt) nxt = bsec;
                        } else if (bmin < nxt)
                            nxt = bmin;
                    }
                    RPROF(4);
                    if (exit_now) done = true;
                    if (!(GROUP && grouped) && cutoff < j0 + 64) { walked = cutoff; break; }
#pragma clang diagnostic pop
                }
                WPROF(3);
                if (!(GROUP && grouped) && lane == 0) core[c].walked = walked;
                __builtin_amdgcn_wave_barrier();
                c = ce;
            }
        }
        WPROF(2);
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
        WPROF(4);
    }
    WPROF_END;
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

__global__ void __launch_bounds__(256) k_keys_wave(DevAlignCfg cfg, DevBatch b, int phase, const uint32_t *__restrict__ list,
                                                   const uint32_t *__restrict__ p_n, uint32_t n_sort, int shift, uint32_t *__restrict__ keys, const uint32_t *__restrict__ work_of)
{
  // keys of the first min(*p_n, n_sort) items; what the sort was sized for beyond the list's real length sorts to the end
  const uint32_t n = *p_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_sort; i += gridDim.x * blockDim.x) {
    if (i >= n) { keys[i] = 0xFFFFFFFFu; continue; }
    const uint32_t pos = list[i], r = b.act[pos];
    if (shift < 0 && work_of != nullptr) { keys[i] = 0xFFFFFFFEu - (work_of[pos] < 0xFFFFFFFEu ? work_of[pos] : 0xFFFFFFFEu); continue; }      // k_flat has added the intervals up already
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
            uint32_t cnt = raw & ~kIvFlags;
            work += cnt;
            if (cnt > best_n) { best_n = cnt; best_first = f; }
        }
    // shift < 0: longest job first (the reads are dealt to the waves in list order; a read with 10^5 candidates that comes up last
    // keeps one wave busy long after the others have run dry)
    keys[i] = shift < 0 ? 0xFFFFFFFEu - (uint32_t)(work < 0xFFFFFFFEULL ? work : 0xFFFFFFFEULL) : (uint32_t)(best_first >> shift);
  }
}

void launch_keys_wave(const DevAlignCfg &cfg, const DevBatch &b, int phase, const uint32_t *list, const uint32_t *p_n, uint32_t n_sort, int shift, uint32_t *keys,
                      const uint32_t *work_of, hipStream_t s)
{
    if (!n_sort) return;
    unsigned blocks = (n_sort + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_keys_wave, dim3(blocks), dim3(256), 0, s, cfg, b, phase, list, p_n, n_sort, shift, keys, work_of);
}

// list: the reads k_flat handed on (*p_n_list of them, at most n_bound); sorted / n_sorted: the sorted copy of the first n_sorted (or null)
void launch_wave(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const HeavyScratch &hs, const uint32_t *list, const uint32_t *sorted,
                 uint32_t n_sorted, const uint32_t *p_n_list, uint32_t n_bound,
                 int phase, uint32_t *cursor, uint32_t *next_act, uint32_t *next_cnt, uint32_t *cmax_next, int nw, uint32_t max_waves,
                 hipStream_t s)
{
    if (!n_bound) return;
    const bool wide = ix.sa_hi != nullptr, hash = ix.isa == nullptr;
    uint32_t waves = n_bound < max_waves ? n_bound : max_waves;
    if (hash && waves > hs.n_slots) waves = hs.n_slots;
    unsigned blocks = (waves + 3) / 4;
    const WaveArgs wa{ix, cfg, b, hs, list, sorted, n_sorted, p_n_list, phase, cursor, next_act, next_cnt, cmax_next};
#define BK_WAVE(N, W, H, S, G) hipLaunchKernelGGL((k_wave<N, W, H, S, G>), dim3(blocks), dim3(256), 0, s, wa)
    const bool sw = ix.swin != nullptr && b.rd2 != nullptr && ix.sw_words == ((nw & 0xff) <= 8 ? 3 : 5);      // (entries of this kernel family's size)
    // (reads of up to 128 bases have four cores or so per strand: their small intervals take a round each; sharing rounds, as the wider
    // forms and the hash-set forms do, cost the 8-word inverse-suffix-array forms more registers than it saved rounds - round 3)
    nw &= 0xff;
    if (nw <= 8) {
        if (wide && sw) BK_WAVE(8, true, true, true, true);                 // (5-byte elements: the set of seen keys, windows from the array where it holds them)
        else if (wide) BK_WAVE(8, true, true, false, true);
        else if (hash) BK_WAVE(8, false, true, false, true);
#ifdef BK_WAVE8_GROUP
        else if (sw) BK_WAVE(8, false, false, true, true);
#else
        else if (sw) BK_WAVE(8, false, false, true, false);
#endif
        else BK_WAVE(8, false, false, false, false);
    } else if (nw <= 16) {
        if (wide && sw) BK_WAVE(16, true, true, true, true);
        else if (wide) BK_WAVE(16, true, true, false, true);
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

}  // namespace bk

#ifdef BK_CAND_HIST
// op 0: (re)allocate and zero the per-block histogram for n suffix array indexes, zero the per-length one; op 1: copy the per-block counts
// (n of them) to out; op 2: the per-length histogram (3 x 40 x 8 bytes); op 3: free; op 4: the per-phase one (8 x 40 x 2 x 8 bytes)
extern "C" int bk_debug_cand_hist(int op, void *out, unsigned long long n)
{
    static uint32_t *d_blk = nullptr;
    static unsigned long long n_blk = 0;
    if (op == 0 || op == 3) {
        if (d_blk) (void)hipFree(d_blk);
        d_blk = nullptr;
        n_blk = 0;
        if (op == 0) {
            n_blk = (n >> bk::kHistShift) + 2;
            if (hipMalloc(&d_blk, n_blk * 4) != hipSuccess || hipMemset(d_blk, 0, n_blk * 4) != hipSuccess) return 1;
        }
        unsigned long long z[3][40] = {}, zp[8][40][2] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(bk::g_hist_blk), &d_blk, sizeof(d_blk)) != hipSuccess) return 1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(bk::g_hist_len), z, sizeof(z)) != hipSuccess) return 1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(bk::g_hist_ph), zp, sizeof(zp)) != hipSuccess) return 1;
        return hipDeviceSynchronize() != hipSuccess;
    }
    if (op == 1) return hipMemcpy(out, d_blk, (n < n_blk ? n : n_blk) * 4, hipMemcpyDeviceToHost) != hipSuccess;
    if (op == 2) return hipMemcpyFromSymbol(out, HIP_SYMBOL(bk::g_hist_len), 3 * 40 * 8) != hipSuccess;
    if (op == 4) return hipMemcpyFromSymbol(out, HIP_SYMBOL(bk::g_hist_ph), 8 * 40 * 2 * 8) != hipSuccess;
    return 1;
}
#endif

#if defined(BK_PROF) && (BK_PROF == 3 || BK_PROF == 4)
extern "C" int bk_debug_prof_wave(unsigned long long *out8)
{
    unsigned long long h[64 * 8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(bk::g_wprof), sizeof(h)) != hipSuccess) return 1;
    for (int k = 0; k < 8; k++) { out8[k] = 0; for (int q = 0; q < 64; q++) out8[k] += h[q * 8 + k]; }
    return 0;
}
#endif
