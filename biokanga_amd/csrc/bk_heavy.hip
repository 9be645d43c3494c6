// bk_heavy.hip - the general wave-per-read form of LocateCoreMultiples (gfx950): reads beyond the register-window families, the
// multi-loci replays (ENUM / BEST), the chimeric call (CHIM), and the loci list bookkeeping.
#include "bk_dev_window.h"
#include "bk_dev_sets.h"
#include "bk_dev_trim.h"

namespace bk {

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

}  // namespace bk
