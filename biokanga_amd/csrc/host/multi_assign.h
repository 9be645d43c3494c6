// Assignment of multi-loci reads to one of their loci by clustering (`biokanga align -r3 / -r4`): host-side
// restatement of CAligner::AssignMultiMatches / ProcAssignMultiMatches / GetClusterStartEnd
// (biokanga/Aligner.cpp:4925-5272) and its two sort orders (SortMultiHits :10119-10155, SortMultiHitReadIDs
// :10160-10199).  Input: every locus of every read AlignReads accepted (eHRhits), as bk_batch_loci() returns them;
// reads with one locus take part as the "unique" neighbours the others cluster with.
//
// Scores are 16 bit as in tsHitLoci.Score: bit 15 = clustered with uniquely aligned reads, low 15 bits the
// accumulated overlap score (upstream pass clamps at 0x1fff, downstream pass at 0x3fff).  The per-thread block
// hand-out is reproduced because a record whose start, length, strand and sequence equal those of the record
// processed just before it IN THE SAME BLOCK copies that record's score instead of computing its own (:4984-4992).
#pragma once
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

#include "../../../include/biokanga_amd.h"

namespace bk {

struct MultiHitRec {
    uint32_t read_id;        // 1-based load order
    bk_loci loci;
    bool multi;              // FlagMH: the read has more than one locus
    bool assigned;           // FlagMHA
    uint8_t how;             // FlagHL: 3 = clustered near unique reads, 4 = near other multi-loci reads
    uint16_t score;
    uint16_t trim_left = 0, trim_right = 0;      // Seg[0].TrimLeft / TrimRight of a chimeric placement (read orientation), else 0
    uint32_t src = 0;        // the caller's number for this locus
};

struct MultiAssignStats { int putative = 0, assigned = 0, near_unique = 0, near_multi = 0; };

// sort with a strict total order on all host threads: chunks sorted independently, then merged pairwise, round by round
template <typename T, typename Less>
inline void par_sort(std::vector<T> &v, Less less, int nthreads)
{
    const size_t n = v.size();
    size_t parts = 1;
    while (parts * 2 <= (size_t)(nthreads < 1 ? 1 : nthreads) && n / (parts * 2) >= 65536) parts *= 2;
    if (parts == 1) { std::sort(v.begin(), v.end(), less); return; }
    std::vector<size_t> cut(parts + 1);
    for (size_t i = 0; i <= parts; i++) cut[i] = n * i / parts;
    {
        std::vector<std::thread> th;
        for (size_t i = 1; i < parts; i++) th.emplace_back([&, i]() { std::sort(v.begin() + (ptrdiff_t)cut[i], v.begin() + (ptrdiff_t)cut[i + 1], less); });
        std::sort(v.begin(), v.begin() + (ptrdiff_t)cut[1], less);
        for (auto &t : th) t.join();
    }
    std::vector<T> tmp(n);
    std::vector<T> *src = &v, *dst = &tmp;
    for (size_t width = 1; width < parts; width *= 2) {
        std::vector<std::thread> th;
        for (size_t i = 0; i < parts; i += 2 * width) {
            const size_t a = cut[i], m = cut[std::min(i + width, parts)], b = cut[std::min(i + 2 * width, parts)];
            th.emplace_back([=, &less]() { std::merge(src->begin() + (ptrdiff_t)a, src->begin() + (ptrdiff_t)m, src->begin() + (ptrdiff_t)m, src->begin() + (ptrdiff_t)b,
                                                       dst->begin() + (ptrdiff_t)a, less); });
        }
        for (auto &t : th) t.join();
        std::swap(src, dst);
    }
    if (src != &v) v.swap(tmp);
}

class MultiAssign {
public:
    std::vector<MultiHitRec> recs;

    void add(uint32_t read_id, const bk_loci &l, bool multi, uint16_t trim_left = 0, uint16_t trim_right = 0, uint32_t src = 0)
    {
        recs.push_back({read_id, l, multi, false, 0, 0, trim_left, trim_right, src});
    }

    // uniq_only: -r3 (cluster with uniquely aligned reads only); nthreads: the -T of the run; max_reads_len: longest loaded read
    MultiAssignStats assign(bool uniq_only, int nthreads, uint32_t max_reads_len)
    {
        MultiAssignStats st;
        if (recs.empty()) return st;
        if (nthreads < 1) nthreads = 1;
        threads_ = nthreads;
        sort_by_loci();
        // blocks as the clustering threads would claim them (GetClusterStartEnd :4929-4957)
        std::vector<std::pair<size_t, size_t>> blocks;
        const size_t n = recs.size();
        for (size_t from = 0; from < n;) {
            const uint32_t left = (uint32_t)(n - from);
            uint32_t num = left;
            if (left >= 100) {
                num = std::min<uint32_t>(2000u, (uint32_t)nthreads + left / (uint32_t)nthreads);
                num = std::min(num, left);
            }
            blocks.push_back({from, from + num});
            from += num;
        }
        {
            const int nt = (int)std::min<size_t>((size_t)nthreads, blocks.size());
            auto work = [&](int w) { for (size_t b = (size_t)w; b < blocks.size(); b += (size_t)nt) score_block(blocks[b].first, blocks[b].second, uniq_only, max_reads_len); };
            std::vector<std::thread> th;
            for (int w = 1; w < nt; w++) th.emplace_back(work, w);
            work(0);
            for (auto &t : th) t.join();
        }
        // best scoring locus per read first (:5134-5181)
        par_sort(recs, [](const MultiHitRec &a, const MultiHitRec &b) {
            if (a.read_id != b.read_id) return a.read_id < b.read_id;
            if (a.score != b.score) return a.score > b.score;
            if (a.loci.chrom_id != b.loci.chrom_id) return a.loci.chrom_id < b.loci.chrom_id;
            if (len_of(a) != len_of(b)) return len_of(a) < len_of(b);
            if (a.loci.mismatches != b.loci.mismatches) return a.loci.mismatches < b.loci.mismatches;
            if (start_of(a) != start_of(b)) return start_of(a) < start_of(b);
            return a.loci.strand < b.loci.strand;
        }, threads_);
        uint32_t cur_read = 0;
        for (size_t i = 0; i < n; i++) {
            MultiHitRec &c = recs[i];
            if (!c.multi || c.read_id == cur_read) continue;
            cur_read = c.read_id;
            st.putative++;
            const uint32_t best = c.score & 0x7fffu;
            if (best < kMinScore) continue;
            if (i + 1 < n && (c.score & kUniqueFlag) == (recs[i + 1].score & kUniqueFlag)) {
                const uint32_t nxt = recs[i + 1].score & 0x7fffu;
                if (best < nxt * 2) continue;
            }
            c.assigned = true;
            c.how = (c.score & kUniqueFlag) ? 3 : 4;
        }
        // a locus chosen for its multi-loci neighbours only stands if one of them is still in play (:5183-5264)
        sort_by_loci();
        for (size_t i = 0; i < n; i++) {
            MultiHitRec &c = recs[i];
            if (!c.assigned) continue;
            bool accept = c.how != 4;
            if (!accept) {
                for (size_t k = i; k-- > 0;) {
                    const MultiHitRec &q = recs[k];
                    const uint32_t dist = start_of(c) - start_of(q);
                    if (dist > (uint32_t)(kOverlap + (int)len_of(q))) break;
                    if (q.loci.chrom_id != c.loci.chrom_id) break;
                    if (!q.multi || q.assigned) { accept = true; break; }
                }
                for (size_t k = i + 1; !accept && k < n; k++) {
                    const MultiHitRec &q = recs[k];
                    const uint32_t dist = start_of(q) - start_of(c);
                    if (dist > (uint32_t)(kOverlap + (int)len_of(c))) break;
                    if (q.loci.chrom_id != c.loci.chrom_id) break;
                    if (!q.multi || q.assigned) { accept = true; break; }
                }
                if (!accept) c.assigned = false;
            }
            if (accept) {
                st.assigned++;
                (c.how == 3 ? st.near_unique : st.near_multi)++;
            }
        }
        return st;
    }

private:
    static constexpr uint16_t kUniqueFlag = 0x8000;   // cUniqueClustFlg
    static constexpr int kOverlap = 10;               // cClustMultiOverLap
    static constexpr int kUniqueScore = 5;            // cClustUniqueScore
    static constexpr int kMultiScore = 1;             // cClustMultiScore
    static constexpr int kScale = 10;                 // cClustScaleFact
    static constexpr uint32_t kMinScore = 50;         // cMHminScore

    // AdjStartLoci / AdjEndLoci / AdjHitLen (Aligner.cpp:1528-1552): the end trims of a chimeric placement move them inwards
    static uint32_t start_of(const MultiHitRec &r) { return r.loci.match_loci + (r.loci.strand == '+' ? r.trim_left : r.trim_right); }
    static uint32_t end_of(const MultiHitRec &r) { return r.loci.match_loci + ((uint32_t)r.loci.match_len - (r.loci.strand == '+' ? r.trim_right : r.trim_left) - 1u); }
    static uint32_t len_of(const MultiHitRec &r) { return (uint32_t)r.loci.match_len - r.trim_left - r.trim_right; }

    int threads_ = 1;

    void sort_by_loci()
    {
        par_sort(recs, [](const MultiHitRec &a, const MultiHitRec &b) {
            if (a.loci.chrom_id != b.loci.chrom_id) return a.loci.chrom_id < b.loci.chrom_id;
            if (start_of(a) != start_of(b)) return start_of(a) < start_of(b);
            if (len_of(a) != len_of(b)) return len_of(a) < len_of(b);
            if (a.loci.mismatches != b.loci.mismatches) return a.loci.mismatches < b.loci.mismatches;
            if (a.loci.strand != b.loci.strand) return a.loci.strand < b.loci.strand;
            return a.read_id < b.read_id;
        }, threads_);
    }

    // one pass of the neighbour scoring; `cap` = 0x1fff upstream, 0x3fff downstream.  Returns true when the scan can stop.
    static bool score_one(MultiHitRec &c, const MultiHitRec &q, int overlap, bool uniq_only, uint32_t cap)
    {
        if ((uniq_only && q.multi) || ((c.score & kUniqueFlag) && (uint32_t)(c.score & 0x7fff) >= cap)) return false;
        if (q.loci.strand != c.loci.strand || q.read_id == c.read_id) return false;
        if (!q.multi) {                              // a uniquely aligned neighbour outranks any number of multi-loci ones
            uint32_t sc = (uint32_t)(1 + (overlap * kUniqueScore) / kScale);
            if (c.score & kUniqueFlag) sc += c.score & 0x7fffu;
            if (sc > cap) sc = cap;
            c.score = (uint16_t)(sc | kUniqueFlag);
            return sc == cap;
        }
        if (!(c.score & kUniqueFlag)) {
            uint32_t sc = (uint32_t)(1 + (overlap * kMultiScore) / kScale);
            sc += c.score & 0x7fffu;
            if (sc > cap) sc = cap;
            c.score = (uint16_t)sc;
        }
        return false;
    }

    void score_block(size_t from, size_t until, bool uniq_only, uint32_t max_reads_len)
    {
        const size_t n = recs.size();
        const MultiHitRec *prev = nullptr;
        for (size_t i = from; i < until; i++) {
            MultiHitRec &c = recs[i];
            if (!c.multi) continue;
            if (prev && start_of(*prev) == start_of(c) && len_of(*prev) == len_of(c) && prev->loci.strand == c.loci.strand &&
                prev->loci.chrom_id == c.loci.chrom_id) {
                c.score = prev->score;
                continue;
            }
            prev = nullptr;
            c.score = 0;
            for (size_t k = i; k-- > 0;) {                                   // upstream
                const MultiHitRec &q = recs[k];
                if (q.loci.chrom_id != c.loci.chrom_id) break;
                if ((uint32_t)(start_of(c) - start_of(q)) >= max_reads_len) break;
                const uint32_t q_end = end_of(q);
                if (q_end < start_of(c) + (uint32_t)kOverlap) continue;
                const int overlap = (int)std::min<uint32_t>(len_of(c), q_end - start_of(c));
                if (score_one(c, q, overlap, uniq_only, 0x1fff)) break;
            }
            for (size_t k = i + 1; k < n; k++) {                             // downstream
                const MultiHitRec &q = recs[k];
                if (q.loci.chrom_id != c.loci.chrom_id) break;
                const uint32_t c_end = end_of(c);
                if (start_of(q) > c_end - (uint32_t)kOverlap) break;
                const int overlap = (int)std::min<uint32_t>(len_of(q), c_end - start_of(q));
                if (score_one(c, q, overlap, uniq_only, 0x3fff)) break;
            }
            prev = &c;
        }
    }
};

}  // namespace bk
