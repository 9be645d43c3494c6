// Host-side filters the reference applies to the aligned read set after the hot path, restated over the C ABI's records:
//   * auto_trim_flanks    - CAligner::AutoTrimFlanks (biokanga/Aligner.cpp:1608-1800), `-x` (and forced by `-A`)
//   * remove_orphan_segs  - CAligner::RemoveOrphanSpliceJuncts / RemoveOrphanMicroInDels (Aligner.cpp:2287-2470)
// Records are indexed like bk_hit[] (one per read, or per reported locus in -r5); reads are looked up through `read_of`.
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <thread>
#include <vector>

#include "../../../include/biokanga_amd.h"

namespace bk {

// per-record flank trims in READ orientation (tsSegLoci.TrimLeft / TrimRight / TrimMismatches of Seg[0])
struct FlankTrims {
    std::vector<uint16_t> left, right;
    std::vector<uint8_t> mismatches;
    size_t removed_plus = 0, removed_minus = 0;
    bool empty() const { return left.empty(); }
};

// From each end of the read walk inwards until min_flank consecutive bases match the target; what lies outside is trimmed.
// SE: a read that cannot keep half its length (>= 15) between two such runs becomes eNARTrim (6); PE: the walks stop after a
// third of the read and nothing is eliminated.  Reads with a second segment are left alone.
//   read_bases(i) -> the read of record i as loaded (1 byte/base), target(i) -> first target base of its alignment (1 byte/base)
inline void auto_trim_flanks(std::vector<bk_hit> &hits, const std::function<bool(size_t)> &has_seg2, const std::function<const uint8_t *(size_t)> &read_bases,
                             const std::function<const uint8_t *(size_t)> &target, int min_flank, bool paired, int nthreads, FlankTrims &out)
{
    const size_t nr = hits.size();
    if (out.left.size() != nr) {                        // may already hold the trims of chimeric placements, which are skipped below
        out.left.assign(nr, 0);
        out.right.assign(nr, 0);
        out.mismatches.resize(nr);
        for (size_t i = 0; i < nr; i++) out.mismatches[i] = hits[i].mismatches;
    }
    if (nthreads < 1) nthreads = 1;
    std::vector<size_t> plus((size_t)nthreads, 0), minus((size_t)nthreads, 0);
    auto work = [&](int w) {
        std::vector<uint8_t> tg;
        for (size_t i = (size_t)w; i < nr; i += (size_t)nthreads) {
            bk_hit &h = hits[i];
            if (h.nar != BK_NAR_ACCEPTED || has_seg2(i)) continue;
            const uint8_t *t0 = target(i);
            if (!t0) continue;
            const uint32_t mlen = h.match_len;
            int min_trimmed = (int)(mlen + 1) / 2;
            if (min_trimmed < 15) min_trimmed = 15;
            const uint8_t *rd = read_bases(i);
            tg.resize(mlen);
            for (uint32_t k = 0; k < mlen; k++) {                          // target in read orientation
                uint8_t t = h.strand == '-' ? t0[mlen - 1 - k] & 7 : t0[k] & 7;
                if (h.strand == '-' && t < 4) t = (uint8_t)(3 - t);
                tg[k] = t;
            }
            int exact = 0, tmm = 0;
            const int core_l = paired ? (int)mlen / 3 : (int)mlen;
            uint32_t idx;
            for (idx = 0; idx <= mlen - (uint32_t)min_trimmed && idx < (uint32_t)core_l; idx++) {
                if ((rd[idx] & 7) != tg[idx]) { exact = 0; tmm++; continue; }
                if (++exact == min_flank) break;
            }
            auto eliminate = [&]() { h.num_hits = 0; h.nar = 6; (h.strand == '+' ? plus : minus)[(size_t)w]++; };      // eNARTrim
            if (!paired && ((idx + (uint32_t)min_trimmed) > mlen || exact < min_flank)) { eliminate(); continue; }
            const int left = (int)idx - (min_flank - 1);
            exact = 0;
            const int core_r = paired ? (int)(mlen * 2) / 3 : 0;
            for (idx = mlen - 1; idx >= (uint32_t)(left + min_trimmed) && idx > (uint32_t)core_r; idx--) {
                if ((rd[idx] & 7) != tg[idx]) { exact = 0; tmm++; continue; }
                if (++exact == min_flank) break;
            }
            if (!paired && (exact != min_flank || idx < (uint32_t)(left + min_trimmed))) { eliminate(); continue; }
            const int right = (int)idx + min_flank;
            out.left[i] = (uint16_t)left;
            out.right[i] = (uint16_t)(mlen - (uint32_t)right);
            if (left || (mlen - (uint32_t)right)) out.mismatches[i] = (uint8_t)(h.mismatches - tmm);
        }
    };
    std::vector<std::thread> th;
    for (int w = 1; w < nthreads; w++) th.emplace_back(work, w);
    work(0);
    for (auto &t : th) t.join();
    for (int w = 0; w < nthreads; w++) { out.removed_plus += plus[(size_t)w]; out.removed_minus += minus[(size_t)w]; }
}

// A two-segment placement (flag bit `want` of bk_seg2.flags: 4 splice junction, 1 microInDel) stands only if another read's
// junction lies within 3 bases of it on both sides; the others get `orphan_nar` (7 eNARSpliceJctn / 8 eNARmicroInDel).
// seg2 is indexed by record.  Returns {placements, orphans removed}.
inline std::pair<size_t, size_t> remove_orphan_segs(std::vector<bk_hit> &hits, const std::vector<bk_seg2> &seg2, uint8_t want, uint8_t orphan_nar)
{
    struct Junct { uint32_t chrom, starts, ends; size_t read; };
    std::vector<Junct> jn;
    if (seg2.empty()) return {0, 0};                 // no segmented alignments were looked for (-N)
    for (size_t i = 0; i < hits.size(); i++)
        if (hits[i].nar == BK_NAR_ACCEPTED && (seg2[i].flags & want))
            jn.push_back({hits[i].chrom_id, hits[i].match_loci + hits[i].match_len - 1u, seg2[i].match_loci, i});
    std::sort(jn.begin(), jn.end(), [](const Junct &x, const Junct &y) {
        if (x.chrom != y.chrom) return x.chrom < y.chrom;
        if (x.starts != y.starts) return x.starts < y.starts;
        return x.ends < y.ends;
    });
    std::vector<uint8_t> supported(jn.size(), 0);
    for (size_t k = 0; k + 1 < jn.size(); k++) {
        const Junct &x = jn[k], &y = jn[k + 1];
        if (x.chrom == y.chrom && x.starts <= y.starts + 3u && x.starts >= y.starts - 3u && x.ends <= y.ends + 3u &&
            x.ends >= y.ends - 3u)                      // UINT32 arithmetic as in tsSegJuncts
            supported[k] = supported[k + 1] = 1;
    }
    size_t n_orphan = 0;
    for (size_t k = 0; k < jn.size(); k++)
        if (!supported[k]) {
            bk_hit &h = hits[jn[k].read];
            h.nar = orphan_nar;
            h.num_hits = 0;
            h.low_hit_instances = 0;
            n_orphan++;
        }
    return {jn.size(), n_orphan};
}

// CAligner::ReducePCRduplicates (Aligner.cpp:2184-2285), `-k`: among accepted reads with the same sequence, start, strand and
// length only the first few of the sorted order are kept - none beyond the first when win_len is 0, else a number that grows with the
// density of distinct read starts within win_len bases (NumUpUniques / NumDnUniques, :9817-9920, quirks included: the upstream
// count looks at the read itself first and gives up at the first non-aligned read of another "sequence").  `order` = the
// reads in the reference's sorted order (SortHitMatch).  Returns the number of reads turned into eNARPCRdup (9).
inline size_t reduce_pcr_duplicates(std::vector<bk_hit> &hits, const std::vector<uint32_t> &order, const std::function<uint32_t(size_t)> &adj_start,
                                    const std::function<uint32_t(size_t)> &adj_len, int win_len)
{
    const size_t n = order.size();
    auto acc = [&](size_t p) { return hits[order[p]].nar == BK_NAR_ACCEPTED; };
    auto uniques = [&](size_t p, bool up) -> int {
        const bk_hit &c = hits[order[p]];
        const int cur_start = (int)adj_start(order[p]);
        int prv = cur_start, cnt = 0;
        if (up) {
            if (p == 0) return 0;                                   // ReadHitIdx == 1
            for (size_t q = p + 1; q-- > 0;) {                      // starts with the read itself (m_ppReadHitsIdx[ReadHitIdx - 1])
                const bk_hit &h = hits[order[q]];
                if (c.chrom_id != h.chrom_id) return cnt;
                if (h.nar != BK_NAR_ACCEPTED) continue;
                const int st = (int)adj_start(order[q]);
                if (cur_start > win_len && (cur_start - win_len) > st) return cnt;
                if (c.strand != h.strand) continue;
                if (st != prv) { cnt++; prv = st; }
            }
            return cnt;
        }
        if (p + 1 == n) return 0;
        for (size_t q = p + 1; q < n; q++) {
            const bk_hit &h = hits[order[q]];
            if (c.chrom_id != h.chrom_id) return cnt;
            if (h.nar != BK_NAR_ACCEPTED) continue;
            const int st = (int)adj_start(order[q]);
            if ((cur_start + win_len) < st) return cnt;
            if (c.strand != h.strand) continue;
            if (st != prv) { cnt++; prv = st; }
        }
        return cnt;
    };
    size_t removed = 0;
    for (size_t p = 0; p < n; p++) {
        if (!acc(p)) continue;
        int limit = 0;
        if (win_len > 0) {
            const int up = uniques(p, true), dn = uniques(p, false);
            limit = up > dn ? up : dn;
            const int prop = (int)(((double)limit / win_len) * 100.0);
            limit = prop < 5 ? 1 : prop <= 10 ? 2 : prop <= 20 ? 3 : prop <= 40 ? 4 : prop <= 60 ? 5 : prop <= 80 ? 10 : 50;
        }
        const bk_hit c = hits[order[p]];
        const uint32_t c_start = adj_start(order[p]), c_len = adj_len(order[p]);
        size_t mark = p;
        for (size_t q = p + 1; q < n; q++) {
            bk_hit &h = hits[order[q]];
            if (h.nar != BK_NAR_ACCEPTED) continue;
            if (c.chrom_id == h.chrom_id && c_start == adj_start(order[q]) && c.strand == h.strand) {
                if (c_len != adj_len(order[q])) continue;
                if (limit > 0) { limit--; continue; }
                h.num_hits = 0; h.low_hit_instances = 0; h.nar = 9;          // eNARPCRdup
                mark = q;
                removed++;
            } else
                break;
        }
        p = mark;
    }
    return removed;
}

}  // namespace bk
