// bk_rescue.hip - what AlignReads does for reads its phases left unaligned, and the paired-end association (gfx950):
//   k_indel                       LocateInDels / LocateSpliceJuncts (SfxArrayV2.cpp:7022-7660, 8437-9405)
//   k_pe_classify / k_pe_orphan   ProcessPairedEnds, PEInsertSize, AlignPairedRead (Aligner.cpp:2726-3489, SfxArrayV2.cpp:8247-8433)
#include "bk_dev_util.h"
#include "bk_dev_trim.h"

namespace bk {

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

// ------------------------------------------------------------------------------------------------
// K4: paired-end association after the SE pass - CAligner::ProcessPairedEnds (biokanga/Aligner.cpp:
// 3055-3489), AcceptProvPE / PEInsertSize (:2726-2850) and the orphan recovery of
// CSfxArrayV3::AlignPairedRead (libbiokanga/SfxArrayV2.cpp:8247-8433) whose AdaptiveTrim (:5482-5682)
// is called with MinTrimLen == read length, i.e. it accepts a window iff: the first and the last 3
// bases match, some run of >= 8 bases matches, and (MaxMM+1.0)/100.0 > mismatches/length (double
// compare, MaxMM = the -s value, <= 15); the FIRST window with the fewest mismatches wins.
// hits[2i] / hits[2i+1] = PE1 / PE2; bk_hit.flags bit 7 = FlgPEAligned.

struct DevPE {
    int pe_mode, min_len, max_len, pair_strand;
    const uint8_t *accept;      // the -Z / -z filters as CAligner::AcceptThisChromID answers them, by sequence id; null = every sequence passes
    uint32_t n_accept;
};

__device__ __forceinline__ bool pe_chrom_ok(const DevPE &pe, uint32_t id) { return pe.accept == nullptr || id >= pe.n_accept || pe.accept[id] != 0; }

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
        // (a unique alignment on a filtered sequence is not accepted as single-ended either; Aligner.cpp:3445-3473 names it "PE partner not aligned")
        if (h.num_hits != 1 || !pe_chrom_ok(pe, h.chrom_id)) {
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
                // AcceptProvPE with the chromosome filters (Aligner.cpp:2771-2786): -3 both ends filtered, -4 / -5 the first / second
                const bool f_ok = pe_chrom_ok(pe, f.chrom_id);
                if (f.chrom_id != r.chrom_id) {
                    const bool r_ok = pe_chrom_ok(pe, r.chrom_id);
                    frag = (f_ok && r_ok) ? -2 : ((!f_ok && !r_ok) ? -3 : (!f_ok ? -4 : -5));
                } else if (!f_ok) frag = -3;
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
                case -3: f.num_hits = 0; f.low_hit_instances = 0; r.num_hits = 0; r.low_hit_instances = 0; f.nar = r.nar = NAR_CHROMFILT; stop = true; break;
                case -4: f.nar = NAR_CHROMFILT; f.num_hits = 0; f.low_hit_instances = 0; break;
                case -5: r.nar = NAR_CHROMFILT; r.num_hits = 0; r.low_hit_instances = 0; break;
                case -6: f.nar = r.nar = NAR_PEINSERTMIN; break;
                case -7: f.nar = r.nar = NAR_PEINSERTMAX; break;
                }
                if (pe.pe_mode == 2 && !stop) {
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
            if (!pe_chrom_ok(pe, a.chrom_id)) {
                // an anchor on a filtered sequence is not used; the reference marks the FIRST read's record for either anchor (:3316-3322,3416-3421)
                if (a.nar == BK_NAR_ACCEPTED) { f.num_hits = 0; f.low_hit_instances = 0; f.nar = NAR_CHROMFILT; }
                continue;
            }
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

void launch_pe(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, int pe_mode, int min_len, int max_len, int pair_strand,
               bk_hit *hits, uint32_t n_pairs, uint32_t *orphans, uint32_t *counters /*[0] count [1] cursor, zeroed*/,
               uint32_t *h_count, bk_seg2 *seg2, int min_chim, int long_reads, const uint8_t *accept, uint32_t n_accept, hipStream_t s)
{
    DevPE pe{pe_mode, min_len, max_len, pair_strand, accept, n_accept};
    hipLaunchKernelGGL(k_pe_classify, dim3((n_pairs + 255) / 256), dim3(256), 0, s, pe, hits, n_pairs, orphans, counters, seg2);
    (void)hipMemcpyAsync(h_count, counters, 4, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    uint32_t n = *h_count;
    if (n) {
        // the 4-bit rows of the pairs the recovery will look at (a lean batch holds rows for its N reads only): 6 % of a C3 step went into
        // packing the rows of every read here
        launch_pack_rows(b, s, orphans, n);
        uint32_t waves = n < 8192 ? n : 8192;
#define BK_ORPH(W) hipLaunchKernelGGL(k_pe_orphan<W>, dim3((waves + 3) / 4), dim3(256), 0, s, ix, cfg, pe, b, hits, orphans, n, counters + 1, seg2, min_chim)
        if (min_chim <= 0 || seg2 == nullptr) BK_ORPH(0);
        else if (long_reads) BK_ORPH(32);                                     // reads of more than 512 bases: 2048-base mismatch map per lane
        else BK_ORPH(8);
#undef BK_ORPH
    }
}

}  // namespace bk
