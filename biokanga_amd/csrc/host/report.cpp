// report.cpp - see report.h
#include "report.h"

#include <strings.h>
#include <time.h>
#include <sys/mman.h>

#include <algorithm>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <thread>

#include "../sfx_file.h"
#include "../bk_env.h"
#include "bam_writer.h"

namespace bkcli {


    // -j / -J: reads that found no alignment at all (NAR EN, NL) / multi-loci reads (NAR ML) as FASTA, in the sorted
    // order, 70 columns (CAligner::ReportNoneAligned / ReportMultiAlign, Aligner.cpp:3826-4010)
void report_read_subset(Report &R, const char *opt, const char *tag, bool (*want)(uint8_t))
{
    const Args &a = R.a;
    auto &hits = R.hits;
    auto &rs = R.rs;
    const size_t nr = R.hits.size();
    const auto &order = R.order;
    auto RD = [&](size_t i) -> size_t { return R.RD(i); };
    const int ml_mode = R.ml_mode;
    if (!a.has(opt) || ml_mode == 5) return;                         // kanga.cpp:1045-1066
    OutBuf o;
    o.open(a.str(opt).c_str());
    if (o.fd < 0) { diag("Unable to create '%s'", a.str(opt).c_str()); return; }
    static const char up[8] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N'};
    std::string rec;
    for (size_t k = 0; k < nr; k++) {
        const uint32_t i = order[k];
        if (!want(hits[i].nar)) continue;
        const uint32_t len = rs.lens[RD(i)];
        const uint8_t *sq = rs.bases.data() + rs.offs[RD(i)];
        char hd[400];
        int n = snprintf(hd, sizeof(hd), ">lcl|%s|%u %s %u|1|%u\n", tag, i + 1, rs.name(RD(i)), i + 1, len);
        rec.assign(hd, (size_t)n);
        for (uint32_t q = 0; q < len; q++) {
            rec.push_back(up[sq[q] & 7]);
            if ((q + 1) % 70 == 0 || q + 1 == len) rec.push_back('\n');
        }
        o.put(rec);
    }
    o.close();
}

    // -O: CAligner::ProcessPairedEnds' insert length table (PE only, Aligner.cpp:3024-3040), WriteBasicCountStats
    // (:4186-4330, fed by WriteSubDist :6275-6336 for every accepted read) and ReportTargHitCnts (:5475-5537)
void report_stats(Report &R)
{
    const Args &a = R.a;
    auto &hits = R.hits;
    auto &rs = R.rs;
    auto &ents = R.ents;
    const uint32_t n_ent = R.n_ent;
    const size_t nr = R.hits.size();
    const auto &multi_dist = R.multi_dist;
    auto RD = [&](size_t i) -> size_t { return R.RD(i); };
    auto has_seg2 = [&](size_t i) -> bool { return R.has_seg2(i); };
    auto TL = [&](size_t i) -> uint32_t { return R.TL(i); };
    auto a_start = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_start(h, i); };
    auto a_len = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_len(h, i); };
    const int pe_mode = R.pe_mode, ml_mode = R.ml_mode, max_ml = R.max_ml, nthreads = R.nthreads;
    if (!a.has("O")) return;
    FILE *f = fopen(a.str("O").c_str(), "w");
    if (!f) { diag("Unable to create '%s'", a.str("O").c_str()); return; }
    if (pe_mode) {
        std::vector<int> len_dist(100001, 0);                    // cPairMaxLen + 1
        for (size_t i = 0; i + 1 < nr; i += 2) {
            const bk_hit &p1 = hits[i], &p2 = hits[i + 1];
            if (!((p1.flags & 0x80) && (p2.flags & 0x80))) continue;
            long s1 = p1.match_loci, e1 = s1 + p1.match_len - 1, s2 = p2.match_loci, e2 = s2 + p2.match_len - 1;
            long frag = p1.strand == '+' ? 1 + e2 - s1 : 1 + e1 - s2;
            if (frag >= 0 && frag <= 100000) len_dist[(size_t)frag]++;
        }
        for (int i = 0; i <= 100000; i++) fprintf(f, "%d,%d\n", i, len_dist[(size_t)i]);
    }
    size_t n_acc = 0;
    uint32_t max_len = 0;
    for (size_t i = 0; i < nr; i++) if (hits[i].nar == BK_NAR_ACCEPTED) { n_acc++; max_len = std::max(max_len, rs.lens[RD(i)]); }
    if (n_acc && max_len) {
        bk::SfxFile sf;
        std::string serr;
        if (bk::sfx_open(a.str("I").c_str(), sf, &serr) == 0) {
            // per read position: accepted reads covering it, and those whose base differs from the target there
            // (read orientation; the target is reverse complemented for '-' alignments); no qualities are
            // loaded, so everything falls into the lowest Phred band
            // qi / sb hold the four Phred bands back to back (band * max_len + position)
            std::vector<std::vector<uint64_t>> qi((size_t)nthreads, std::vector<uint64_t>((size_t)max_len * 4, 0)), sb(qi),
                ms((size_t)nthreads, std::vector<uint64_t>(max_len, 0));
            auto work = [&](int w) {
                auto &Q = qi[(size_t)w], &S = sb[(size_t)w], &M = ms[(size_t)w];
                for (size_t i = (size_t)w; i < nr; i += (size_t)nthreads) {
                    const bk_hit &h = hits[i];
                    if (h.nar != BK_NAR_ACCEPTED || h.chrom_id < 1 || h.chrom_id > n_ent || has_seg2(i)) continue;     // FlagSegs reads are sloughed (:6286)
                    const uint8_t *rd = rs.bases.data() + rs.offs[RD(i)];
                    const uint32_t len = rs.lens[RD(i)];
                    const uint8_t *tg = sf.seq + ents[h.chrom_id - 1].start_ofs + a_start(h, i);
                    const uint32_t alen = a_len(h, i), tl0 = TL(i);
                    uint32_t nsub = 0;
                    for (uint32_t k = 0; k < alen && tl0 + k < len; k++) {      // read positions TrimLeft .. ReadLen - TrimRight (:6303-6306)
                        uint8_t t = h.strand == '-' ? tg[alen - 1 - k] & 7 : tg[k] & 7;
                        if (h.strand == '-' && t < 4) t = (uint8_t)(3 - t);
                        const uint32_t q4 = (rd[tl0 + k] >> 4) & 15;           // 4-bit score -> band (WriteSubDist :6309-6320)
                        const size_t at = (size_t)(q4 <= 3 ? 0 : q4 <= 7 ? 1 : q4 <= 11 ? 2 : 3) * max_len + tl0 + k;
                        Q[at]++;
                        if ((rd[tl0 + k] & 7) != t) { S[at]++; nsub++; }
                    }
                    M[nsub < max_len ? nsub : max_len - 1]++;
                }
            };
            std::vector<std::thread> th;
            for (int w = 1; w < nthreads; w++) th.emplace_back(work, w);
            work(0);
            for (auto &t : th) t.join();
            for (int w = 1; w < nthreads; w++)
                for (size_t k = 0; k < (size_t)max_len * 4; k++) { qi[0][k] += qi[(size_t)w][k]; sb[0][k] += sb[(size_t)w][k]; if (k < max_len) ms[0][k] += ms[(size_t)w][k]; }
            static const char *band_a[4] = {"Phred 0..9", "Phred 10..19", "Phred 20..29", "Phred 30+"};
            static const char *band_b[4] = {"Phred 0..8", "Phred 9..19", "Phred 20..29", "Phred 30+"};
            if (ml_mode) {                                           // WriteBasicCountStats, Aligner.cpp:4203-4227
                fprintf(f, "\"Multihit distribution\",");
                for (int k = 0; k < max_ml; k++) fprintf(f, ",%d", k + 1);
                fprintf(f, "\n,\"Instances\"");
                for (int k = 0; k < max_ml; k++) fprintf(f, ",%d", multi_dist[(size_t)k]);
                fprintf(f, "\n");
            }
            fprintf(f, "\"Phred Score Instances\",");
            for (uint32_t k = 0; k < max_len; k++) fprintf(f, ",%u", k + 1);
            for (int bnd = 0; bnd < 4; bnd++) {
                fprintf(f, "\n,\"%s\"", band_a[bnd]);
                for (uint32_t k = 0; k < max_len; k++) fprintf(f, ",%llu", (unsigned long long)qi[0][(size_t)bnd * max_len + k]);
            }
            fprintf(f, "\n\n\"Aligner Induced Subs\",");
            for (uint32_t k = 0; k < max_len; k++) fprintf(f, ",%u", k + 1);
            for (int bnd = 0; bnd < 4; bnd++) {
                fprintf(f, "\n,\"%s\"", band_b[bnd]);
                for (uint32_t k = 0; k < max_len; k++) fprintf(f, ",%llu", (unsigned long long)sb[0][(size_t)bnd * max_len + k]);
            }
            fprintf(f, "\n\n\"Multiple substitutions\",");
            for (uint32_t k = 0; k < max_len; k++) fprintf(f, ",%u", k);
            fprintf(f, "\n,\"Instances\"");
            for (uint32_t k = 0; k < max_len; k++) fprintf(f, ",%llu", (unsigned long long)ms[0][k]);
            fprintf(f, "\n");
        } else
            diag("Unable to reopen '%s' for the substitution statistics: %s", a.str("I").c_str(), serr.c_str());
        std::vector<uint64_t> cnt(n_ent, 0);                  // final NAR (after any PE processing), in entry order
        for (size_t i = 0; i < nr; i++)
            if (hits[i].nar == BK_NAR_ACCEPTED && hits[i].chrom_id >= 1 && hits[i].chrom_id <= n_ent) cnt[hits[i].chrom_id - 1]++;
        fprintf(f, "\"TargSeq\",\"TargLen\",\"NumHits\"\n");
        for (uint32_t c = 0; c < n_ent; c++)
            if (cnt[c]) fprintf(f, "\"%s\",%u,%llu\n", ents[c].name, ents[c].seq_len, (unsigned long long)cnt[c]);
    }
    fclose(f);
}

    // -A with SAM / BAM output: the junctions are still reported as BED lines in "<out>.jct" (Aligner.cpp:713-721,4440-4462); the track
    // title is empty in these modes
void report_jct_for_sam(Report &R)
{
    const Args &a = R.a;
    auto &hits = R.hits;
    auto &ents = R.ents;
    const size_t nr = R.hits.size();
    const auto &order = R.order;
    const auto &seg2 = R.seg2;
    auto RD = [&](size_t i) -> size_t { return R.RD(i); };
    auto has_seg2 = [&](size_t i) -> bool { return R.has_seg2(i); };
    const int fmt = R.fmt, splice_len = R.splice_len;
    if (!splice_len || fmt < 5) return;
    std::string jp = a.str("o");
    if (jp.size() > 3 && !strcasecmp(jp.c_str() + jp.size() - 3, ".gz")) jp.resize(jp.size() - 3);
    OutBuf j;
    j.open((jp + ".jct").c_str());
    if (j.fd < 0) { diag("Unable to create '%s.jct'", jp.c_str()); return; }
    char ln[1024];
    int m = snprintf(ln, sizeof(ln), "track type=bed name=\"JCT_\" description=\"\"\n");
    j.put(ln, (size_t)m);
    for (size_t k = 0; k < nr; k++) {
        const uint32_t i = order[k];
        const bk_hit &h = hits[i];
        if (h.nar != BK_NAR_ACCEPTED || !has_seg2(i) || !(seg2[RD(i)].flags & 4)) continue;
        const bk_seg2 &g = seg2[RD(i)];
        const uint32_t end1 = g.match_loci + g.match_len;
        m = snprintf(ln, sizeof(ln), "%s\t%u\t%u\tarj\t0\t%c\t%u\t%u\t0\t2\t%u,%u\t0,%u\n", ents[h.chrom_id - 1].name, h.match_loci, end1, (char)h.strand,
                     h.match_loci, end1, (unsigned)h.match_len, (unsigned)g.match_len, g.match_loci - h.match_loci);
        j.put(ln, (size_t)m);
    }
    j.close();
}

// which sequences hold at least one accepted alignment (the header lists those, CSAMfile::AddRefSeq)
static std::vector<uint8_t> seqs_with_hits(const std::vector<bk_hit> &hits, uint32_t n_ent, int nthreads)
{
    const int nt = std::max(1, nthreads);
    std::vector<std::vector<uint8_t>> part((size_t)nt);
    par_ranges(hits.size(), nt, [&](size_t lo, size_t hi, int t) {
        std::vector<uint8_t> m(n_ent + 1, 0);
        for (size_t i = lo; i < hi; i++) { const bk_hit &h = hits[i]; if (h.nar == BK_NAR_ACCEPTED && h.chrom_id <= n_ent) m[h.chrom_id] = 1; }
        part[(size_t)t].swap(m);
    });
    std::vector<uint8_t> has(n_ent + 1, 0);
    for (auto &m : part) for (size_t c = 0; c < m.size(); c++) has[c] |= m[c];
    return has;
}

    // ".bam" (more than 5 characters of name, kanga.cpp:848-857): BGZF-compressed BAM with its BAI index
int report_bam(Report &R, const std::string &opath)
{
    auto &hits = R.hits;
    auto &rs = R.rs;
    auto &ents = R.ents;
    const std::string &species = R.species;
    const uint32_t n_ent = R.n_ent;
    const size_t nr = R.hits.size();
    const auto &order = R.order;
    const auto &seg2 = R.seg2;
    auto RD = [&](size_t i) -> size_t { return R.RD(i); };
    auto has_seg2 = [&](size_t i) -> bool { return R.has_seg2(i); };
    auto TL = [&](size_t i) -> uint32_t { return R.TL(i); };
    auto TR = [&](size_t i) -> uint32_t { return R.TR(i); };
    auto a_start = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_start(h, i); };
    auto a_len = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_len(h, i); };
    const int pe_mode = R.pe_mode, fmt = R.fmt, nthreads = R.nthreads, max_rpt_sam_seqs = R.max_rpt_sam_seqs;
    int rc = 0;
    std::vector<uint8_t> has_hit = seqs_with_hits(hits, n_ent, nthreads);
    const bool all = (uint32_t)max_rpt_sam_seqs >= n_ent;
    std::string text = "@HD\tVN:1.4\tSO:coordinate";
    std::vector<int32_t> ref_of(n_ent + 1, -1);
    std::vector<uint32_t> refs;
    int n_with = 0;
    char tmp[512];
    for (uint32_t c = 1; c <= n_ent; c++) {
        if (!has_hit[c] && !all) continue;
        int n = snprintf(tmp, sizeof(tmp), "\n@SQ\tAS:%s\tSN:%s\tLN:%u", species.empty() ? "NA" : species.c_str(), ents[c - 1].name, ents[c - 1].seq_len);
        text.append(tmp, (size_t)n);
        ref_of[c] = (int32_t)refs.size();
        refs.push_back(c);
        n_with += has_hit[c];
    }
    int n = snprintf(tmp, sizeof(tmp), "\n@PG\tID:%s\tVN:%s\n", g_proc.c_str(), kProgVer);
    text.append(tmp, (size_t)n);
    diag("Header written with references to %d sequences of which %d have at least 1 alignments", (int)refs.size(), n_with);
    std::vector<uint8_t> stream;
    auto p32 = [](std::vector<uint8_t> &v, uint32_t x) { v.insert(v.end(), (uint8_t *)&x, (uint8_t *)&x + 4); };
    stream.insert(stream.end(), {'B', 'A', 'M', 1});
    p32(stream, (uint32_t)text.size());
    stream.insert(stream.end(), text.begin(), text.end());
    p32(stream, (uint32_t)refs.size());
    for (uint32_t c : refs) {
        uint32_t ln = (uint32_t)strlen(ents[c - 1].name) + 1;
        p32(stream, ln);
        stream.insert(stream.end(), ents[c - 1].name, ents[c - 1].name + ln);
        p32(stream, ents[c - 1].seq_len);
    }
    // records (CAligner::ReportBAMread, Aligner.cpp:5768-6126; CSAMfile::AddAlignment, SAMfile.cpp:2283-2540),
    // formatted in stripes of the sorted order by all host threads
    static const uint8_t code4[8] = {1, 2, 4, 8, 15, 15, 15, 15}, comp4[8] = {8, 4, 2, 1, 15, 15, 15, 15};
    struct Stripe { std::vector<uint8_t> bytes; std::vector<bk::BamAligned> al; uint64_t n = 0; };
    const size_t per_thread = 32768;
    const size_t n_stripes = (nr + per_thread - 1) / per_thread;
    std::vector<Stripe> stripes(n_stripes);
    auto format_stripe = [&](size_t si) {
        Stripe &S = stripes[si];
        const size_t lo = si * per_thread, hi = std::min(nr, lo + per_thread);
        std::vector<uint8_t> &v = S.bytes;
        for (size_t k = lo; k < hi; k++) {
            const uint32_t i = order[k];
            const bk_hit &h = hits[i];
            const bool acc = h.nar == BK_NAR_ACCEPTED;
            if (!acc && fmt != 6) continue;
            const uint8_t *sq = rs.bases.data() + rs.offs[RD(i)];
            const uint32_t len = rs.lens[RD(i)];
            int flag = 0, tlen = 0;
            long pnext = -1;
            if (!pe_mode) flag = acc ? (h.strand == '+' ? 0 : 16) : 4;
            else {
                const bool first_of_pair = (i & 1) == 0;
                const bk_hit &m = hits[first_of_pair ? i + 1 : i - 1];
                flag = 0x1 | 0x2 | (first_of_pair ? 0x40 : 0x80);
                flag |= acc ? (h.strand == '+' ? 0 : 0x10) : 0x4;
                if ((h.flags & 0x80) && (m.flags & 0x80) && m.nar == BK_NAR_ACCEPTED) {
                    flag |= m.strand == '+' ? 0 : 0x20;
                    if (acc) {
                        const size_t mi = first_of_pair ? i + 1 : i - 1;
                        pnext = (long)a_start(m, mi);
                        long s0 = (long)a_start(h, i), s1 = (long)a_start(m, mi);
                        tlen = (int)(s0 <= s1 ? (s1 - s0) + (long)a_len(m, mi) : (s0 - s1) + (long)a_len(h, i));
                    }
                } else
                    flag |= 0x8;
            }
            const char *qn = rs.name(RD(i));
            const uint32_t l_qn = (uint32_t)strlen(qn) + 1;
            const char *tag = acc ? nullptr : kNarTag[h.nar < 20 ? h.nar : 0];
            const uint32_t aux = tag ? 3 + (uint32_t)strlen(tag) + 1 : 0;
            const bool two = acc && has_seg2(i);
            // soft clips in target order: read orientation for '+', swapped for '-' (Aligner.cpp:5961-5984)
            const uint32_t clip5 = acc ? (h.strand == '+' ? TL(i) : TR(i)) : 0u, clip3 = acc ? (h.strand == '+' ? TR(i) : TL(i)) : 0u;
            const uint32_t n_cig = (two ? 3u : 1u) + (clip5 ? 1u : 0u) + (clip3 ? 1u : 0u);
            const uint32_t pos0 = acc ? a_start(h, i) : 0u;
            const uint32_t hit_len = acc ? a_len(h, i) + (two ? seg2[RD(i)].match_len : 0u) : 0u;      // AdjAlignHitLen
            const uint32_t block = 32 + l_qn + 4 * n_cig + (len + 1) / 2 + len + aux;
            const size_t at = v.size();
            v.resize(at + 4 + block);
            uint8_t *q = v.data() + at;
            auto w32 = [&](uint32_t x) { memcpy(q, &x, 4); q += 4; };
            w32(block);
            const int32_t ref = acc ? ref_of[h.chrom_id] : -1;
            w32((uint32_t)ref);
            w32(acc ? pos0 : 0xFFFFFFFFu);
            const uint32_t bin = acc ? (uint32_t)bk::bam_reg2bin((int)pos0, (int)(pos0 + hit_len)) : 0u;
            w32(bin << 16 | 255u << 8 | l_qn);
            w32((uint32_t)flag << 16 | n_cig);
            w32(len);
            w32(acc && pnext >= 0 ? (uint32_t)ref : 0xFFFFFFFFu);
            w32(acc ? (uint32_t)pnext : 0xFFFFFFFFu);
            w32((uint32_t)tlen);
            memcpy(q, qn, l_qn); q += l_qn;
            if (clip5) w32(clip5 << 4 | 4u);
            w32((acc ? a_len(h, i) : len) << 4);
            if (clip3) w32(clip3 << 4 | 4u);
            if (two) {
                const bk_seg2 &g = seg2[RD(i)];
                if (g.flags & 4) w32((uint32_t)((long)g.match_loci - ((long)h.match_loci + h.match_len)) << 4 | 3u);
                else if (g.flags & 2) w32((uint32_t)((long)len - ((long)h.match_len + g.match_len)) << 4 | 1u);
                else { long gap = (long)g.match_loci - ((long)h.match_loci + h.match_len); w32((uint32_t)(gap < 0 ? -gap : gap) << 4 | 2u); }
                w32((uint32_t)g.match_len << 4);
            }
            uint8_t byte = 0;
            for (uint32_t o = 0; o < len; o++) {
                uint8_t c4 = (acc && h.strand != '+') ? comp4[sq[len - 1 - o] & 7] : code4[sq[o] & 7];
                if (!(o & 1)) byte = (uint8_t)(c4 << 4);
                else byte |= c4;
                if ((o & 1) || o == len - 1) *q++ = byte;
            }
            {
                uint32_t sum = 0;
                for (uint32_t o = 0; o < len; o++) sum += (sq[o] >> 4) & 15;
                if (!sum) memset(q, 0xff, len);
                else {                                       // the reference stores the ASCII form here as well (SAMfile.cpp:2374)
                    const bool rev = acc && h.strand != '+';
                    for (uint32_t o = 0; o < len; o++) q[o] = (uint8_t)(33 + ((((rev ? sq[len - 1 - o] : sq[o]) >> 4) & 15) * 40) / 15);
                }
                q += len;
            }
            if (tag) { *q++ = 'Y'; *q++ = 'U'; *q++ = 'Z'; size_t tl = strlen(tag) + 1; memcpy(q, tag, tl); q += tl; }
            if (acc) S.al.push_back({(uint64_t)at, (uint64_t)(at + 4 + block), ref, (int32_t)pos0, (int32_t)(pos0 + hit_len - 1)});
            S.n++;
        }
    };
    {
        std::vector<std::thread> th;
        auto work = [&](int w) { for (size_t si = (size_t)w; si < n_stripes; si += (size_t)nthreads) format_stripe(si); };
        for (int w = 1; w < nthreads; w++) th.emplace_back(work, w);
        work(0);
        for (auto &t : th) t.join();
    }
    std::vector<bk::BamAligned> aligned;
    uint64_t flush_at = 0, n_rep = 0;
    for (Stripe &S : stripes) {
        const uint64_t base = stream.size();
        for (bk::BamAligned al : S.al) { al.u_beg += base; al.u_end += base; aligned.push_back(al); flush_at = al.u_end; }
        stream.insert(stream.end(), S.bytes.begin(), S.bytes.end());
        n_rep += S.n;
        std::vector<uint8_t>().swap(S.bytes);
    }
    std::string berr;
    uint64_t max_ref_len = 0;
    for (uint32_t c : refs) max_ref_len = std::max<uint64_t>(max_ref_len, ents[c - 1].seq_len);
    rc = bk::write_bam_and_bai(opath, stream, aligned, flush_at, (uint32_t)refs.size(), max_ref_len, nthreads, &berr);
    if (rc) { diag("Fatal: %s", berr.c_str()); return 1; }
    report_jct_for_sam(R);
    diag("Completed reporting BAM %llu read alignments", (unsigned long long)n_rep);
    diag("Reporting of aligned result set completed");
    report_read_subset(R, "j", "na", [](uint8_t nar) { return nar == BK_NAR_NS || nar == BK_NAR_NOHIT; });
    report_read_subset(R, "J", "ml", [](uint8_t nar) { return nar == BK_NAR_MULTIALIGN; });
    report_stats(R);
    return 0;
}

// SAM text (-M5 / -M6, optionally gzip'd), CSV (-M0..3) and BED (-M4) output
int report_text(Report &R)
{
    const Args &a = R.a;
    auto &hits = R.hits;
    auto &rs = R.rs;
    auto &ents = R.ents;
    const std::string &species = R.species;
    const uint32_t n_ent = R.n_ent;
    const size_t nr = R.hits.size();
    const auto &order = R.order;
    const auto &seg2 = R.seg2;
    auto RD = [&](size_t i) -> size_t { return R.RD(i); };
    auto has_seg2 = [&](size_t i) -> bool { return R.has_seg2(i); };
    auto TL = [&](size_t i) -> uint32_t { return R.TL(i); };
    auto TR = [&](size_t i) -> uint32_t { return R.TR(i); };
    auto a_start = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_start(h, i); };
    auto a_len = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_len(h, i); };
    auto a_mm = [&](const bk_hit &h, size_t i) -> uint32_t { return R.a_mm(h, i); };
    const int pe_mode = R.pe_mode, ml_mode = R.ml_mode, fmt = R.fmt, nthreads = R.nthreads, micro_indel = R.micro_indel, splice_len = R.splice_len, max_rpt_sam_seqs = R.max_rpt_sam_seqs;
    OutBuf out;
    if (R.pre != nullptr && fmt >= 5) { out.fd = R.pre->fd; out.borrowed = true; out.b.reserve(8 << 20); }
    else out.open(a.str("o").c_str());
    if (out.fd < 0) { diag("Fatal: unable to create '%s'", a.str("o").c_str()); return 1; }
    char line[8192];
    uint64_t n_reported = 0;
    if (fmt >= 5) {
        // header: CSAMfile::Create/AddRefSeq/StartAlignments
        std::vector<uint8_t> has_hit = seqs_with_hits(hits, n_ent, nthreads);
        bool all = (uint32_t)max_rpt_sam_seqs >= n_ent;
        out.put("@HD\tVN:1.4\tSO:coordinate");
        int n_hdr = 0, n_with = 0;
        for (uint32_t c = 1; c <= n_ent; c++) {
            if (!has_hit[c] && !all) continue;
            int n = snprintf(line, sizeof(line), "\n@SQ\tAS:%s\tSN:%s\tLN:%u", species.empty() ? "NA" : species.c_str(), ents[c - 1].name, ents[c - 1].seq_len);
            out.put(line, (size_t)n);
            n_hdr++;
            n_with += has_hit[c];
        }
        int n = snprintf(line, sizeof(line), "\n@PG\tID:%s\tVN:%s\n", g_proc.c_str(), kProgVer);
        out.put(line, (size_t)n);
        diag("Header written with references to %d sequences of which %d have at least 1 alignments", n_hdr, n_with);
        static const char comp[8] = {'T', 'G', 'C', 'A', 'N', 'N', 'N', 'N'};
        static const char fwd[8] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N'};
        // Records are written through a raw pointer into a stripe buffer that is grown to the record's worst case first (the
        // reference formats with sprintf, ~4.5 us per read; the host's core-seconds are what bounds T_e2e here).
        struct Stripe {
            char *d = nullptr;
            size_t n = 0, cap = 0;
            ~Stripe() { free(d); }
            char *room(size_t k)
            {
                if (n + k > cap) { cap = std::max(cap * 2, n + k + (1u << 20)); d = (char *)realloc(d, cap); }
                return d + n;
            }
        };
        auto put_num = [](char *w, long v) -> char * {
            char t[24];
            int n = 0;
            bool neg = v < 0;
            unsigned long u = neg ? (unsigned long)(-v) : (unsigned long)v;
            do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
            if (neg) *w++ = '-';
            while (n) *w++ = t[--n];
            return w;
        };
        auto put_str = [](char *w, const char *z) -> char * { const size_t k = strlen(z); memcpy(w, z, k); return w + k; };
        // QUAL (ReportBAMread :5928-5955): '*' when no base carries a score, else 33 + q4 * 40 / 15 per base, reversed with the read
        auto put_qual = [](char *w, const uint8_t *sq, uint32_t n, bool reversed) -> char * {
            uint32_t sum = 0;
            for (uint32_t q = 0; q < n; q++) sum |= sq[q] & 0xf0u;
            if (!sum) { *w++ = '*'; return w; }
            for (uint32_t q = 0; q < n; q++) w[q] = (char)(33 + ((((reversed ? sq[n - 1 - q] : sq[q]) >> 4) & 15) * 40) / 15);
            return w + n;
        };
        // one record (CAligner::ReportBAMread, Aligner.cpp:5850-5924,6036-6054); false when the read is not reported
        auto format_rec = [&](size_t k, Stripe &st) -> bool {
            uint32_t i = order[k];
            const bk_hit &h = hits[i];
            bool acc = h.nar == BK_NAR_ACCEPTED;
            if (!acc && fmt != 6) return false;
            const uint8_t *s = rs.bases.data() + rs.offs[RD(i)];
            uint32_t len = rs.lens[RD(i)];
            const char *nm = rs.name(RD(i));
            const size_t nml = strlen(nm);
            char *w = st.room(nml + 2 * (size_t)len + 400);
            memcpy(w, nm, nml);
            w += nml;
            int flag = 0, tlen = 0;
            long pnext = -1;
            if (!pe_mode) flag = acc ? (h.strand == '+' ? 0 : 16) : 4;
            else {
                const bool first_of_pair = (i & 1) == 0;
                const bk_hit &m = hits[first_of_pair ? i + 1 : i - 1];
                flag = 0x1 | 0x2 | (first_of_pair ? 0x40 : 0x80);
                flag |= acc ? (h.strand == '+' ? 0 : 0x10) : 0x4;
                if ((h.flags & 0x80) && (m.flags & 0x80) && m.nar == BK_NAR_ACCEPTED) {
                    flag |= m.strand == '+' ? 0 : 0x20;
                    if (acc) {
                        const size_t mi = first_of_pair ? i + 1 : i - 1;
                        pnext = (long)a_start(m, mi);
                        long s0 = (long)a_start(h, i), s1 = (long)a_start(m, mi);
                        tlen = (int)(s0 <= s1 ? (s1 - s0) + (long)a_len(m, mi) : (s0 - s1) + (long)a_len(h, i));
                    }
                } else
                    flag |= 0x8;
            }
            *w++ = '\t';
            w = put_num(w, flag);
            if (acc) {
                *w++ = '\t';
                w = put_str(w, ents[h.chrom_id - 1].name);
                *w++ = '\t';
                w = put_num(w, (long)a_start(h, i) + 1);
                w = put_str(w, "\t255\t");
                const uint32_t clip5 = h.strand == '+' ? TL(i) : TR(i), clip3 = h.strand == '+' ? TR(i) : TL(i);
                if (clip5) { w = put_num(w, clip5); *w++ = 'S'; }
                w = put_num(w, a_len(h, i));
                *w++ = 'M';
                if (clip3) { w = put_num(w, clip3); *w++ = 'S'; }
                if (has_seg2(i)) {                                       // CAligner::ReportBAMread, Aligner.cpp:5986-6033
                    const bk_seg2 &g = seg2[RD(i)];
                    if (g.flags & 4) { w = put_num(w, (long)g.match_loci - ((long)h.match_loci + h.match_len)); *w++ = 'N'; }
                    else if (g.flags & 2) { w = put_num(w, (long)len - ((long)h.match_len + g.match_len)); *w++ = 'I'; }
                    else { long gap = (long)g.match_loci - ((long)h.match_loci + h.match_len); w = put_num(w, gap < 0 ? -gap : gap); *w++ = 'D'; }
                    w = put_num(w, g.match_len);
                    *w++ = 'M';
                }
                *w++ = '\t';
                *w++ = pnext < 0 ? '*' : '=';
                *w++ = '\t';
                w = put_num(w, pnext < 0 ? 0L : pnext + 1);
                *w++ = '\t';
                w = put_num(w, tlen);
                *w++ = '\t';
                if (h.strand == '+') for (uint32_t q = 0; q < len; q++) w[q] = fwd[s[q] & 7];
                else for (uint32_t q = 0; q < len; q++) w[q] = comp[s[len - 1 - q] & 7];
                w += len;
                *w++ = '\t';
                w = put_qual(w, s, len, h.strand != '+');
                *w++ = '\n';
            } else {
                w = put_str(w, "\t*\t0\t255\t");
                w = put_num(w, len);
                w = put_str(w, "M\t*\t0\t0\t");
                for (uint32_t q = 0; q < len; q++) w[q] = fwd[s[q] & 7];
                w += len;
                *w++ = '\t';
                w = put_qual(w, s, len, false);
                w = put_str(w, "\t\tYU:Z:");                           // the doubled TAB is what the reference writes
                w = put_str(w, kNarTag[h.nar < 20 ? h.nar : 0]);
                *w++ = '\n';
            }
            st.n = (size_t)(w - st.d);
            return true;
        };
        // Plain records - one segment, no end trims, one record per read - are formatted on the device (bk_sam_format): the host
        // hands over reads, names, records and the output order, and copies the text it gets back, slice by slice, into the file.
        // Whatever the device path cannot take, or fails on, is formatted below by the host threads.
        bool device_done = false;
        const size_t dev_min = (size_t)bk::env::sam_device_min(100000ULL);       // (records from which the device formats: tests set it to 1, a huge value keeps the host path)
        if (R.ctx != nullptr && !out.gz && !out.pipe && R.src.empty() && R.seg2.empty() && R.trims.empty() && nr >= dev_min) {
            const bool timing0 = bk::env::timing();
            timespec t0s; clock_gettime(CLOCK_MONOTONIC, &t0s);
            out.flush();
            struct SinkState { int fd; off_t base; int nthreads; SamPrealloc *pre; std::atomic<long> us_wait{0}, us_copy{0}, us_zap{0}; };
            SinkState st{out.fd, out.pos, std::max(1, nthreads / 2), R.pre};
            auto sink = [](void *user, const char *text, uint64_t n, uint64_t ofs) -> int {
                SinkState *S = static_cast<SinkState *>(user);
                const off_t at = S->base + (off_t)ofs;
                auto us_now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (long)ts.tv_sec * 1000000L + ts.tv_nsec / 1000; };
                const long u0 = us_now();
                // the slice's range of the file is allocated and mapped, and the threads copy into the mapping (concurrent pwrite()s to one
                // file queue up behind the inode lock); pwrite() remains for files that cannot be mapped
                char *map = nullptr;
                const off_t map_lo = at & ~(off_t)4095;
                const size_t map_len = (size_t)(at + (off_t)n - map_lo);
                // (pages the background thread has allocated, or is about to, are not allocated twice)
                bool have = false;
                if (S->pre != nullptr && S->pre->est >= at + (off_t)n) {
                    while (S->pre->done.load() < at + (off_t)n && !S->pre->ended.load()) {
                        timespec ts{0, 200000};
                        nanosleep(&ts, nullptr);
                    }
                    have = S->pre->done.load() >= at + (off_t)n;
                }
                // (a range the background threads have populated is written through their mapping: no fault, no mmap)
                const bool through_pre = have && S->pre->map != nullptr;
                if (through_pre) map = S->pre->map + map_lo;
                else if (have || fallocate(S->fd, 0, at, (off_t)n) == 0) {
                    void *m = mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, S->fd, map_lo);
                    if (m != MAP_FAILED) map = (char *)m;
                }
                const long u1 = us_now();
                const int nt = S->nthreads;
                std::vector<std::thread> th;
                std::atomic<int> bad{0};
                auto put = [&](int t) {
                    const uint64_t lo = n * (uint64_t)t / (uint64_t)nt, hi = n * (uint64_t)(t + 1) / (uint64_t)nt;
                    if (map) { memcpy(map + (at - map_lo) + lo, text + lo, hi - lo); return; }
                    for (uint64_t o = lo; o < hi;) {
                        ssize_t w = ::pwrite(S->fd, text + o, hi - o, at + (off_t)o);
                        if (w <= 0) { bad = 1; return; }
                        o += (uint64_t)w;
                    }
                };
                for (int t = 1; t < nt; t++) th.emplace_back(put, t);
                put(0);
                for (auto &t : th) t.join();
                const long u2 = us_now();
                S->us_wait += u1 - u0; S->us_copy += u2 - u1;
                struct Zap { SinkState *S; long u; decltype(us_now) &now; ~Zap() { S->us_zap += now() - u; } } zap{S, u2, us_now};
                if (map && !through_pre) munmap(map, map_len);
                else if (through_pre) {
                    // (the written range leaves the page table behind us: the unmapper thread's work)
                    const off_t lo_al = (at + 4095) & ~(off_t)4095, hi_al = (at + (off_t)n) & ~(off_t)4095;
                    if (hi_al > lo_al) S->pre->unmap_behind(S->pre->map + lo_al, (size_t)(hi_al - lo_al));
                }
                return bad.load();
            };
            bk_sam_job job{};
            if (R.pk_words == nullptr) { job.bases = rs.bases.data(); job.n_bases = rs.bases.size(); job.offs = rs.offs.data(); job.lens = rs.lens.data(); }
            job.names = rs.names.data(); job.n_name_bytes = rs.names.size(); job.name_ofs = rs.name_ofs.data();
            job.hits = hits.data(); job.n_reads = nr; job.order = order.data(); job.n_order = nr;
            job.report_unaligned = fmt == 6 ? 1 : 0; job.pe_mode = pe_mode;
            job.pk_words = R.pk_words; job.n_pk_words = R.n_pk_words; job.pk_lens16 = R.pk_lens16; job.pk_exc = R.pk_exc; job.n_pk_exc = R.n_pk_exc;
            job.prep = R.sam_prep;
            R.sam_prep = nullptr;                                                    // (consumed by the call whatever its outcome)
            uint64_t n_rep = 0, n_bytes = 0;
            int drc = BK_ERR_PARAMS;
            if (bk::env::sam_device_fail()) { if (job.prep) bk_sam_prep_free(job.prep); }       // (tests: the device declines after its head start)
            else if (rs.lens.size() == nr) drc = bk_sam_format(R.ctx, &job, sink, &st, &n_rep, &n_bytes);
            if (drc == BK_OK) {
                out.pos += (off_t)n_bytes;
                timespec ta; clock_gettime(CLOCK_MONOTONIC, &ta);
                if (R.pre != nullptr) R.pre->finish();
                if (ftruncate(out.fd, out.pos) != 0) { diag("Fatal: unable to size '%s'", a.str("o").c_str()); return 1; }
                if (R.pre != nullptr) R.pre->kept = true;
                if (timing0) { timespec tb; clock_gettime(CLOCK_MONOTONIC, &tb); fprintf(stderr, "bk timing: SAM file cut to its size: %.0f ms\n", 1e3 * ((double)(tb.tv_sec - ta.tv_sec) + 1e-9 * (double)(tb.tv_nsec - ta.tv_nsec))); }
                n_reported = n_rep;
                device_done = true;
                if (timing0) fprintf(stderr, "bk timing: SAM sink (summed over its calls, two may overlap): waited for the file's pages %ld ms, copied %ld ms, unmapped %ld ms\n", st.us_wait.load() / 1000, st.us_copy.load() / 1000, st.us_zap.load() / 1000);
                if (timing0) { timespec t1s; clock_gettime(CLOCK_MONOTONIC, &t1s); fprintf(stderr, "bk timing: SAM formatted on the device and copied out: %.0f ms (%llu bytes)\n", 1e3 * ((double)(t1s.tv_sec - t0s.tv_sec) + 1e-9 * (double)(t1s.tv_nsec - t0s.tv_nsec)), (unsigned long long)n_bytes); }
            } else if (timing0)
                fprintf(stderr, "bk timing: device SAM formatter declined (%s): host threads format\n", bk_strerror(drc));
        }
        if (!device_done && R.restore_reads && R.restore_reads() != 0) { diag("Fatal: unable to reload the reads for the host formatter"); return 1; }
        // records are formatted by all host threads into per-thread buffers, one stripe of the sorted order
        // each, and written out in order (the reference formats serially, ~4.5 us per read)
        const size_t per_thread = 131072;
        const int nt = (int)std::min<size_t>((size_t)nthreads, (nr + per_thread - 1) / per_thread ? (nr + per_thread - 1) / per_thread : 1);
        std::vector<Stripe> bufs((size_t)nt);
        std::vector<uint64_t> cnts((size_t)nt);
        std::vector<std::vector<uint8_t>> zbufs((size_t)nt);
        const bool timing = bk::env::timing();
        auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
        double t_fmt = 0, t_grow = 0, t_put = 0;
        unsigned rounds_mapped = 0, rounds_pwrite = 0;
        // The file's pages are allocated ahead of the writers by one background thread (Linux fallocate - which fails with EOPNOTSUPP
        // where the file system cannot do it natively, instead of emulating it with reads and writes that would race the writers as
        // glibc's posix_fallocate does; any failure just means "no preallocation" - 256 MB at a time, from an
        // upper estimate of the text's size): a store into a page that already exists costs a minor fault, one into a hole of a tmpfs
        // or ext4 file an allocation under the file's locks - with every writer thread doing that at once the copy-out ran at 3.7 GB/s.
        // The file is cut to its real size at the end.
        std::atomic<off_t> prealloc_size{0};
        std::thread prealloc;
        if (R.pre != nullptr && !device_done) { R.pre->finish(); prealloc_size.store(R.pre->done.load()); }    // (what the early thread allocated counts; the rest as before)
        if (!out.gz && !out.pipe && nr >= 200000 && !device_done && R.pre == nullptr) {
            out.flush();
            const bool with_qual = a.num("g", 3) != 3;                                 // QUAL is '*' unless FASTQ scores were loaded (-g0..2)
            uint64_t est = (uint64_t)out.pos + rs.name_bytes() + 64ULL * nr + (pe_mode ? 24ULL * nr : 0);
            for (size_t k = 0; k < nr; k++) est += (with_qual ? 2ULL : 1ULL) * rs.lens[RD(k)];
            const int pfd = out.fd;
            prealloc = std::thread([pfd, est, &prealloc_size]() {
                const off_t step = 256LL << 20;
                for (off_t at = 0; at < (off_t)est; at += step) {
                    const off_t len = std::min<off_t>(step, (off_t)est - at);
                    if (fallocate(pfd, 0, at, len) != 0) return;                         // e.g. a pipe, NFS: the writers grow the file themselves
                    prealloc_size.store(at + len);
                }
            });
        }
        for (size_t k0 = 0; k0 < nr && !device_done; k0 += per_thread * (size_t)nt) {
            const double tA = now();
            auto work = [&](int t) {
                size_t lo = k0 + (size_t)t * per_thread, hi = std::min(nr, lo + per_thread);
                Stripe &buf = bufs[(size_t)t];
                buf.n = 0;
                uint64_t c = 0;
                // the sorted order walks the read store at random: the records' lines are requested a few records ahead
                for (size_t k = lo; k < hi; k++) {
                    if (k + 16 < hi) {
                        const uint32_t j = order[k + 16];
                        __builtin_prefetch(&hits[j]);
                        const size_t r = RD(j);
                        __builtin_prefetch(&rs.offs[r]);
                        __builtin_prefetch(&rs.lens[r]);
                        __builtin_prefetch(&rs.name_ofs[r]);
                    }
                    if (k + 8 < hi) {
                        const size_t r = RD(order[k + 8]);
                        const uint8_t *b = rs.bases.data() + rs.offs[r];
                        __builtin_prefetch(b);
                        __builtin_prefetch(b + 64);
                        __builtin_prefetch(rs.names.data() + rs.name_ofs[r]);
                    }
                    c += format_rec(k, buf) ? 1 : 0;
                }
                cnts[(size_t)t] = c;
                // compressed SAM: every thread makes gzip members of its own stretch (one deflate stream through one thread takes minutes
                // for what the threads format in a second; a gzip file may be any number of members)
                if (out.gz) { zbufs[(size_t)t].clear(); if (buf.n && !gzip_members(buf.d, buf.n, zbufs[(size_t)t])) out.failed = true; }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; t++) th.emplace_back(work, t);
            work(0);
            for (auto &t : th) t.join();
            const double tB = now();
            t_fmt += tB - tA;
            if (out.gz) {                    // compressed SAM: the threads' members, in order
                for (int t = 0; t < nt; t++) { out.put_members(zbufs[(size_t)t].data(), zbufs[(size_t)t].size()); n_reported += cnts[(size_t)t]; }
                continue;
            }
            if (out.pipe) {                  // a FIFO / pipe takes the text in order, through write()
                for (int t = 0; t < nt; t++) { out.put(bufs[(size_t)t].d, bufs[(size_t)t].n); n_reported += cnts[(size_t)t]; }
                continue;
            }
            // the stripes go to their places in the file in parallel as well: the file is grown by the round's bytes and the threads
            // copy into a shared mapping of that range (concurrent pwrite()s to one file queue up behind the inode lock - 1.3 GB/s
            // on tmpfs - while page faults on a mapping do not); pwrite() remains for outputs that cannot be mapped
            out.flush();
            std::vector<off_t> at((size_t)nt + 1);
            at[0] = out.pos;
            for (int t = 0; t < nt; t++) { at[(size_t)t + 1] = at[(size_t)t] + (off_t)bufs[(size_t)t].n; n_reported += cnts[(size_t)t]; }
            const off_t map_lo = at[0] & ~(off_t)4095;
            const size_t map_len = (size_t)(at[(size_t)nt] - map_lo);
            char *map = nullptr;
            // the range must exist before it is mapped: already preallocated, or allocated here (fallocate only ever grows a file, so
            // it cannot collide with the background thread the way an ftruncate could)
            const bool have_range = at[(size_t)nt] > at[0] &&
                                    ((prealloc.joinable() || R.pre != nullptr) ? (prealloc_size.load() >= at[(size_t)nt] || fallocate(out.fd, 0, at[0], at[(size_t)nt] - at[0]) == 0)
                                                         : ftruncate(out.fd, at[(size_t)nt]) == 0);
            if (have_range) {
                void *m = mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, out.fd, map_lo);
                if (m != MAP_FAILED) map = (char *)m;
            }
            if (map) rounds_mapped++; else rounds_pwrite++;
            const double tC = now();
            t_grow += tC - tB;
            auto put = [&](int t) {
                const Stripe &bf = bufs[(size_t)t];
                if (map) { memcpy(map + (at[(size_t)t] - map_lo), bf.d, bf.n); return; }
                size_t o = 0;
                while (o < bf.n) {
                    ssize_t w = ::pwrite(out.fd, bf.d + o, bf.n - o, at[(size_t)t] + (off_t)o);
                    if (w < 0 && errno == EINTR) continue;
                    if (w <= 0) { out.failed = true; break; }          // (a full disk ends the report with an error, not with a short file)
                    o += (size_t)w;
                }
            };
            th.clear();
            for (int t = 1; t < nt; t++) th.emplace_back(put, t);
            put(0);
            for (auto &t : th) t.join();
            if (map) munmap(map, map_len);
            t_put += now() - tC;
            out.pos = at[(size_t)nt];
        }
        if (R.pre != nullptr && !device_done) {
            out.flush();
            if (ftruncate(out.fd, out.pos) != 0) { diag("Fatal: unable to size '%s'", a.str("o").c_str()); return 1; }
            R.pre->kept = true;
        }
        if (prealloc.joinable()) {
            prealloc.join();
            out.flush();
            if (ftruncate(out.fd, out.pos) != 0) { diag("Fatal: unable to size '%s'", a.str("o").c_str()); return 1; }
        }
        if (timing) fprintf(stderr, "bk timing: SAM format %.0f ms, grow + map %.0f ms, copy out %.0f ms (%d threads; %u rounds through a shared mapping, %u through pwrite)\n", 1e3 * t_fmt, 1e3 * t_grow, 1e3 * t_put, nt, rounds_mapped, rounds_pwrite);
        report_jct_for_sam(R);
        diag("Completed reporting SAM %llu read alignments", (unsigned long long)n_reported);
    } else {
        // -M0..3 CSV (loci; 1: + match sequence, 2: + read sequence, 3: + both) and -M4 UCSC BED
        // (CAligner::WriteReadHits, Aligner.cpp:6336-6660); the site-preference score column is 0 as in the
        // reference when no -8 preferences are computed
        static const char up[8] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N'};
        bk::SfxFile sf;
        if (fmt == 1 || fmt == 3) {
            std::string serr;
            if (bk::sfx_open(a.str("I").c_str(), sf, &serr) != 0) { diag("Fatal: %s", serr.c_str()); return 1; }
        }
        if (fmt == 4) {
            std::string title = a.str("t", "kanga");
            int m = snprintf(line, sizeof(line), "track type=bed name=\"%s\" description=\"%s\"\n", title.c_str(), title.c_str());
            out.put(line, (size_t)m);
            if (ml_mode == 5) out.put(line, (size_t)m);      // written at file creation AND by WriteReadHits (Aligner.cpp:4405-4413,6356-6362)
        }
        // -a with -M4: reads aligned with a microInDel go to "<out>.ind" as 12-column BED lines (Aligner.cpp:4417-4438,6372-6376,6519-6526)
        OutBuf ind;
        if (fmt == 4 && micro_indel) {
            ind.open((a.str("o") + ".ind").c_str());
            if (ind.fd < 0) { diag("Fatal: unable to create '%s.ind'", a.str("o").c_str()); return 1; }
            std::string title = a.str("t", "kanga");
            int m = snprintf(line, sizeof(line), "track type=bed name=\"IND_%s\" description=\"%s\"\n", title.c_str(), title.c_str());
            ind.put(line, (size_t)m);
        }
        OutBuf jct;
        if (fmt == 4 && splice_len) {
            jct.open((a.str("o") + ".jct").c_str());
            if (jct.fd < 0) { diag("Fatal: unable to create '%s.jct'", a.str("o").c_str()); return 1; }
            std::string title = a.str("t", "kanga");
            int m = snprintf(line, sizeof(line), "track type=bed name=\"JCT_%s\" description=\"%s\"\n", title.c_str(), title.c_str());
            jct.put(line, (size_t)m);
        }
        // The lines are made by all threads - a stretch of the output order each, into buffers of its own, compressed there too when the
        // output is a .gz - and go to the files in order (one thread's snprintf over 50 M records was 12 s of a run whose alignment
        // takes 0.2).
        struct Made { std::string out, ind, jct; std::vector<uint8_t> z; uint64_t n = 0; };
        const size_t per_thread = 65536;
        const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, nthreads), (nr + per_thread - 1) / per_thread));
        std::vector<Made> made((size_t)nt);
        for (size_t k0 = 0; k0 < nr; k0 += per_thread * (size_t)nt) {
            auto work = [&](int t) {
                Made &M = made[(size_t)t];
                M.out.clear(); M.ind.clear(); M.jct.clear(); M.z.clear();
                M.n = 0;
                char line[8192];
                std::string rec;
                const size_t lo = k0 + (size_t)t * per_thread, hi = std::min(nr, lo + per_thread);
                for (size_t k = lo; k < hi; k++) {
                    uint32_t i = order[k];
                    const bk_hit &h = hits[i];
                    if (h.nar != BK_NAR_ACCEPTED) continue;
                    const bool two = has_seg2(i);
                    if (fmt == 4) {
                        if (two) {
                            const bk_seg2 &g = seg2[RD(i)];
                            const bool sj = (g.flags & 4) != 0;
                            const uint32_t end1 = g.match_loci + g.match_len;          // AdjAlignEndLoci + 1
                            int m = snprintf(line, sizeof(line), "%s\t%u\t%u\t%s\t0\t%c\t%u\t%u\t0\t2\t%u,%u\t0,%u\n", ents[h.chrom_id - 1].name, h.match_loci, end1,
                                             sj ? "arj" : "ari", (char)h.strand, h.match_loci, end1, (unsigned)h.match_len, (unsigned)g.match_len, g.match_loci - h.match_loci);
                            (sj ? M.jct : M.ind).append(line, (size_t)m);
                        } else {
                            int m = snprintf(line, sizeof(line), "%s\t%u\t%u\tar\t0\t%c\n", ents[h.chrom_id - 1].name, a_start(h, i), a_start(h, i) + a_len(h, i),
                                             (char)h.strand);
                            M.out.append(line, (size_t)m);
                        }
                        M.n++;
                        continue;
                    }
                    // one line per segment (WriteReadHits, Aligner.cpp:6566-6627)
                    const uint32_t len = rs.lens[RD(i)];
                    for (int sg = 0; sg < (two ? 2 : 1); sg++) {
                        const uint32_t s_loci = sg ? seg2[RD(i)].match_loci : a_start(h, i), s_len = sg ? seg2[RD(i)].match_len : a_len(h, i);
                        const uint32_t s_mm = sg ? seg2[RD(i)].mismatches : a_mm(h, i), s_rofs = sg ? seg2[RD(i)].read_ofs : TL(i);       // ReadOfs + TrimLeft
                        int m = snprintf(line, sizeof(line), "%u,\"%s\",\"%s\",\"%s\",%u,%u,%u,\"%c\",0,0,1,%u,\"N/A\",\"%s\"", i + 1,
                                         two ? ((seg2[RD(i)].flags & 4) ? "arj" : "ari") : "ar", species.c_str(),
                                         ents[h.chrom_id - 1].name, s_loci, s_loci + s_len - 1, (unsigned)s_len, (char)h.strand, (unsigned)s_mm, rs.name(RD(i)));
                        rec.assign(line, (size_t)m);
                        if (fmt >= 2) {                                          // the read as loaded, from the segment's read offset
                            const uint8_t *sq = rs.bases.data() + rs.offs[RD(i)];
                            rec += ",\"";
                            for (uint32_t q = 0; q < s_len && s_rofs + q < len; q++) rec.push_back(up[sq[s_rofs + q] & 7]);
                            rec.push_back('"');
                        }
                        if (fmt == 1 || fmt == 3) {                              // the target it matched, in read orientation
                            const uint8_t *tg = sf.seq + ents[h.chrom_id - 1].start_ofs + s_loci;
                            rec += ",\"";
                            for (uint32_t q = 0; q < s_len; q++) {
                                uint8_t t = h.strand == '-' ? tg[s_len - 1 - q] & 7 : tg[q] & 7;
                                if (h.strand == '-' && t < 4) t = (uint8_t)(3 - t);
                                rec.push_back(up[t]);
                            }
                            rec.push_back('"');
                        }
                        rec.push_back('\n');
                        M.out += rec;
                    }
                    M.n++;
                }
                if (out.gz && !M.out.empty() && !gzip_members(M.out.data(), M.out.size(), M.z)) out.failed = true;
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; t++) th.emplace_back(work, t);
            work(0);
            for (auto &t : th) t.join();
            for (int t = 0; t < nt; t++) {
                Made &M = made[(size_t)t];
                if (out.gz) out.put_members(M.z.data(), M.z.size());
                else out.put(M.out.data(), M.out.size());
                if (!M.ind.empty()) ind.put(M.ind);
                if (!M.jct.empty()) jct.put(M.jct);
                n_reported += M.n;
            }
        }
        if (ind.fd >= 0) ind.close();
        if (jct.fd >= 0) jct.close();
        if (ind.failed || jct.failed) out.failed = true;                    // (the microInDel / junction files count like the results file)
    }
    out.close();
    if (out.failed) {                                // (eBSFerrFileAccess: the reference's WriteReadHits gives up on a short write too)
        diag("Fatal error: unable to write all of the results to '%s' - disk full?", a.str("o").c_str());
        return -85;
    }
    diag("Reporting of aligned result set completed");

    report_read_subset(R, "j", "na", [](uint8_t nar) { return nar == BK_NAR_NS || nar == BK_NAR_NOHIT; });
    report_read_subset(R, "J", "ml", [](uint8_t nar) { return nar == BK_NAR_MULTIALIGN; });
    report_stats(R);
    return 0;
}

}  // namespace bkcli
