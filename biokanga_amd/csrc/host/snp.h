// SNP calling of `biokanga align -p<minreads> [-P<qvalue>] [-1<nonref%>] [-S<file>]`: host side of CAligner::ProcessSNPs /
// OutputSNPs (biokanga/Aligner.cpp:7609-8071, 6803-7607).  The pile-up of read bases over the target and the per-locus
// screen run on the GPU (bk_snp_pileup / bk_snp_sites); here: the binomial P-values (CStats::Binomial,
// libbiokanga/Stats.cpp:475-551), the Benjamini-Hochberg cut, the CSV / VCF / BED writers and the DiSNP / TriSNP
// haplotype tables the reference writes beside the SNP file (`<snpfile>.disnp.csv`, `.trisnp.csv`).
// Marker sequences (-K / -G, `<snpfile>.markers`) are assembled from the device counts around each putative SNP (bk_snp_counts).
// SNP centroids (-7 <file>): NumInsts per 7-mer from the device (bk_snp_centroid_insts), the per-SNP sums on the host.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cli_common.h"
#include "report.h"

namespace bkcli {

struct SnpOpts {
    int min_reads = 0;            // -p  m_MinSNPreads
    double qvalue = 0.05;         // -P  m_QValue
    double nonref_prop = 0.25;    // -1 / 100  m_SNPNonRefPcnt
    std::string path;             // -S or <out>.snp
    bool vcf = false, bed = false;
    std::string title;            // BED track title
    std::string sfx_path;         // ##reference= of the VCF header
    int marker_len = 0;           // -K  0 or 25..500
    std::string centroid_path;    // -7
    double marker_poly_thres = 0; // -G  m_MarkerPolyThres
};

namespace snp_detail {

// CStats::Calc_nCk (Stats.cpp:475-509): the product form in extended precision
inline double calc_nck(uint32_t n, uint32_t k)
{
    if (k > n) return 0.0;
    if (k > n / 2) k = n - k;
    long double accum = 1;
    for (uint32_t i = 1; i <= k; i++) accum = accum * (n - k + i) / i;
    return (double)accum;
}
// CStats::ProbKeqlk (:514-523)
inline double prob_k_eql(uint32_t n, uint32_t k, double p)
{
    if (p < 0 || p > 1) return -1;
    double nck = calc_nck(n, k);
    double p2 = pow(p, (double)(int32_t)k);
    double q2 = pow(1 - p, (double)(int32_t)(n - k));
    return nck * p2 * q2;
}
// CStats::Binomial (:527-547), its clamp of n to 5000 (k scaled by 1000/n) included
inline double binomial(int n, int k, double p)
{
    if (k > n) return 0.0;
    if (n > 5000) { k = (int)((1000.0 / n) * k); n = 5000; }
    double sum = 0;
    for (int i = 0; i <= k; i++) {
        sum += prob_k_eql((uint32_t)n, (uint32_t)i, p);
        if (sum >= 1.0) break;
    }
    return sum < 1.0 ? sum : 1.0;
}

struct LociP {                    // tsLociPValues
    uint32_t loci, rank;
    double pvalue, bkgnd_rate;
    uint32_t local_reads, local_subs, num_reads, num_subs, ref_base, non_ref[5];
    uint32_t marker_id, n_polymorphic;
};

inline char base_uc(uint32_t b) { return b < 4 ? "ACGT"[b] : 'N'; }          // CSeqTrans::MapBase2Ascii
inline char base_lc(uint32_t b) { return b < 4 ? "acgt"[b] : 'n'; }
inline const char *kCentroidCase = "ACGT";     // CSeqTrans::MapSeq2Ascii

}  // namespace snp_detail

// returns 0, or 1 after a fatal message
inline int process_snps(bk_ctx *ctx, Report &R, const SnpOpts &o)
{
    using namespace snp_detail;
    const size_t nrec = R.hits.size();
    OutBuf snp_out, di_out, tri_out;
    snp_out.open(o.path.c_str());
    if (snp_out.fd < 0) { diag("Fatal: Unable to create/truncate SNP file '%s'", o.path.c_str()); return 1; }
    const std::string di_path = o.path + ".disnp.csv", tri_path = o.path + ".trisnp.csv";
    di_out.open(di_path.c_str());
    tri_out.open(tri_path.c_str());
    if (di_out.fd < 0 || tri_out.fd < 0) { diag("Fatal: Unable to create/truncate DiSNP/TriSNP files beside '%s'", o.path.c_str()); return 1; }
    diag("Processing for SNPs and writing out SNPs to file '%s", o.path.c_str());
    OutBuf marker_out;
    const std::string marker_path = o.path + ".markers";
    const int marker5 = o.marker_len / 2, marker3 = o.marker_len - 1 - marker5;       // Aligner.cpp:224-225
    uint32_t marker_id = 0;
    if (o.marker_len) {
        marker_out.open(marker_path.c_str());
        if (marker_out.fd < 0) { diag("Fatal: Unable to create/truncate markers file '%s'", marker_path.c_str()); return 1; }
        diag("Processing for Markers and writing out marker sequences to file '%s", marker_path.c_str());
    }
    std::vector<uint32_t> mcnt;
    struct Centroid { uint32_t num_snps = 0, ref_cnt = 0, non_ref[5] = {0, 0, 0, 0, 0}; };
    std::vector<Centroid> cents;
    std::vector<uint32_t> cent_insts;
    OutBuf cent_out;
    if (!o.centroid_path.empty()) {
        cent_out.open(o.centroid_path.c_str());
        if (cent_out.fd < 0) { diag("Fatal: Unable to create/truncate SNP centroids file '%s'", o.centroid_path.c_str()); return 1; }
        cents.resize(BK_SNP_CENTROIDS);
        cent_insts.assign(BK_SNP_CENTROIDS, 0);
    }
    char line[4096];
    // headers (ProcessSNPs :7633-7718)
    if (o.bed) snp_out.put(line, (size_t)snprintf(line, sizeof(line), "track type=bed name=\"%s_SNPs\" description=\"%s SNPs\"\n", o.title.c_str(), o.title.c_str()));
    else if (o.vcf)
        snp_out.put(line, (size_t)snprintf(line, sizeof(line), "##fileformat=VCFv4.2\n##source=biokangaV%s\n##reference=%s\n##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele Frequency\">\n"
                                           "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Read Depth\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n",
                                           kProgVer, o.sfx_path.c_str()));
    else
        snp_out.put(std::string("\"SNP_ID\",\"ElType\",\"Species\",\"Chrom\",\"StartLoci\",\"EndLoci\",\"Len\",\"Strand\",\"Rank\",\"PValue\",\"Bases\",\"Mismatches\",\"RefBase\",\"MMBaseA\",\"MMBaseC\","
                                "\"MMBaseG\",\"MMBaseT\",\"MMBaseN\",\"BackgroundSubRate\",\"TotWinBases\",\"TotWinMismatches\",\"MarkerID\",\"NumPolymorphicSites\"\n"));
    {
        std::string h = "\"DiSNPs_ID\",\"ElType\",\"Species\",\"Chrom\"";
        for (int k = 1; k <= 2; k++) {
            const std::string n = "SNP" + std::to_string(k);
            h += ",\"" + n + "Loci\",\"" + n + "RefBase\",\"" + n + "BaseAcnt\",\"" + n + "BaseCcnt\",\"" + n + "BaseGcnt\",\"" + n + "BaseTcnt\",\"" + n + "BaseNcnt\"";
        }
        h += ",\"Depth\",\"Antisense\",\"Haplotypes\"";
        for (int i = 0; i < 16; i++) { h += ",\""; h += "acgt"[(i >> 2) & 3]; h += "acgt"[i & 3]; h += "\""; }
        h += "\n";
        di_out.put(h);
        h = "\"TriSNPs_ID\",\"ElType\",\"Species\",\"Chrom\"";
        for (int k = 1; k <= 3; k++) {
            const std::string n = "SNP" + std::to_string(k);
            h += ",\"" + n + "Loci\",\"" + n + "RefBase\",\"" + n + "BaseAcnt\",\"" + n + "BaseCcnt\",\"" + n + "BaseGcnt\",\"" + n + "BaseTcnt\",\"" + n + "BaseNcnt\"";
        }
        h += ",\"Depth\",\"Antisense\",\"Haplotypes\"";
        for (int i = 0; i < 64; i++) { h += ",\""; h += "acgt"[(i >> 4) & 3]; h += "acgt"[(i >> 2) & 3]; h += "acgt"[i & 3]; h += "\""; }
        h += "\n";
        tri_out.put(h);
    }

    // the reads ProcessSNPs piles up: accepted, neither microInDel nor spliced (:7739-7744)
    auto piled = [&](size_t i) { return R.hits[i].nar == BK_NAR_ACCEPTED && !R.has_seg2(i); };
    int rc = bk_snp_reset(ctx);
    if (rc) { diag("Fatal: SNP pile-up could not be set up: %s", bk_strerror(rc)); return 1; }
    struct ChromAcc { uint64_t tot_len = 0, n_reads = 0; size_t first = (size_t)-1, last = 0; };
    std::vector<ChromAcc> acc(R.n_ent + 1);
    std::vector<uint32_t> chrom_order;
    for (size_t k = 0; k < R.order.size(); k++) {          // sequences in the order the sorted reads reach them
        const size_t i = R.order[k];
        if (!piled(i)) continue;
        const bk_hit &h = R.hits[i];
        ChromAcc &c = acc[h.chrom_id];
        if (c.first == (size_t)-1) { c.first = k; chrom_order.push_back(h.chrom_id); }
        c.last = k;
        uint32_t len = R.a_len(h, i);
        const uint32_t clen = (uint32_t)R.ents[h.chrom_id - 1].seq_len, st = R.a_start(h, i);
        if (st + len > clen) { if ((int)clen - (int)st < 10) continue; len = clen - st; }
        c.tot_len += len; c.n_reads++;
    }
    {
        const size_t chunk = 8u << 20;
        std::vector<bk_snp_aln> alns;
        for (size_t r0 = 0; r0 < nrec; r0 += chunk) {
            const size_t r1 = std::min(nrec, r0 + chunk);
            alns.clear();
            for (size_t i = r0; i < r1; i++) {
                if (!piled(i)) continue;
                const bk_hit &h = R.hits[i];
                bk_snp_aln a{};
                a.read_idx = (uint32_t)(R.RD(i) - R.RD(r0)); a.chrom_id = h.chrom_id; a.loci = R.a_start(h, i); a.len = (uint16_t)R.a_len(h, i);
                a.read_ofs = (uint16_t)R.TL(i); a.strand = h.strand;
                alns.push_back(a);
            }
            if (alns.empty()) continue;
            const size_t rd0 = R.RD(r0), rd1 = R.RD(r1 - 1) + 1;
            rc = bk_snp_pileup(ctx, R.rs.bases.data(), R.rs.offs.data() + rd0, R.rs.lens.data() + rd0, (uint32_t)(rd1 - rd0), alns.data(), alns.size());
            if (rc) { diag("Fatal: SNP pile-up failed: %s", bk_strerror(rc)); return 1; }
        }
    }

    uint64_t tot_snps = 0, loci_covered = 0, bases_coverage = 0;
    char sz_alts[100] = "", sz_freq[100] = "";             // as in the reference these keep their last contents when no allele qualifies
    for (uint32_t chrom : chrom_order) {
        const bk_snp_site *sites = nullptr;
        uint64_t n_sites = 0;
        bk_snp_chrom tot{};
        rc = bk_snp_sites(ctx, chrom, o.min_reads, o.nonref_prop, &sites, &n_sites, &tot);
        if (rc) { diag("Fatal: SNP screening failed: %s", bk_strerror(rc)); return 1; }
        loci_covered += tot.loci_covered; bases_coverage += tot.bases_coverage;
        if (!cents.empty()) {
            rc = bk_snp_centroid_insts(ctx, chrom, o.min_reads, cent_insts.data());
            if (rc) { diag("Fatal: SNP centroid counting failed: %s", bk_strerror(rc)); return 1; }
        }
        const char *chrom_name = R.ents[chrom - 1].name;
        const ChromAcc &ca = acc[chrom];
        const int max_disnp_sep = (int)(uint32_t)((ca.tot_len + ca.n_reads - 1) / ca.n_reads);      // MeanReadLen (:7750)
        const double global_rate = std::max(0.01, (double)tot.tot_mismatch / (double)(1 + tot.tot_match + tot.tot_mismatch));   // cMinSeqErrRate
        std::vector<LociP> lp;
        lp.reserve(n_sites);
        for (uint64_t k = 0; k < n_sites; k++) {
            const bk_snp_site &s = sites[k];
            const uint32_t nonref = s.non_ref[0] + s.non_ref[1] + s.non_ref[2] + s.non_ref[3] + s.non_ref[4];
            const int tot_bases = (int)(nonref + s.num_ref);
            const uint32_t loc_tmm = nonref <= s.win_mismatches ? s.win_mismatches - nonref : 0;
            const uint32_t loc_tm = s.num_ref < s.win_matches ? s.win_matches - s.num_ref : 0;
            double rate;
            if (loc_tmm + loc_tm == 0) rate = global_rate;
            else {
                rate = (double)loc_tmm / (double)(loc_tmm + loc_tm);
                if (rate < global_rate) rate = global_rate;
            }
            if (rate > 0.20) continue;                     // cMaxBkgdNoiseThres
            LociP e{};
            if (o.marker_len) {                            // :7006-7086: only SNPs a marker sequence can be built around are reported
                if (s.loci < (uint32_t)marker5 || s.loci + (uint32_t)marker3 >= (uint32_t)R.ents[chrom - 1].seq_len) continue;
                if ((double)nonref / tot_bases < 0.5) continue;
                const int mlen = 1 + marker5 + marker3, mstart = (int)s.loci - marker5;
                mcnt.resize((size_t)mlen * 7);
                rc = bk_snp_counts(ctx, chrom, (uint32_t)mstart, (uint32_t)mlen, mcnt.data());
                if (rc) { diag("Fatal: SNP marker counts could not be fetched: %s", bk_strerror(rc)); return 1; }
                std::string mseq((size_t)mlen, 'N');
                int n_poly = 0, q = 0;
                for (; q < mlen; q++) {
                    const uint32_t *c7 = &mcnt[(size_t)q * 7];
                    const uint32_t nr = c7[1] + c7[2] + c7[3] + c7[4] + c7[5];
                    const int tb = (int)(nr + c7[0]);
                    if (tb < o.min_reads) break;
                    double prop = (double)nr / tb;
                    if (prop <= o.marker_poly_thres) {
                        if (prop > 0.1) n_poly++;
                        mseq[(size_t)q] = base_uc(c7[6]);
                        continue;
                    }
                    int al = 0;
                    for (; al < 5; al++)
                        if (c7[1 + al] > 0 && (prop = (double)c7[1 + al] / tb) >= (1.0 - o.marker_poly_thres)) {
                            if (prop < 0.9) n_poly++;
                            mseq[(size_t)q] = base_uc((uint32_t)al);
                            break;
                        }
                    if (al == 5) break;
                }
                if (q != mlen) continue;
                const char ref_c = base_uc(s.ref_base), snp_c = mseq[(size_t)marker5];
                if (ref_c == snp_c) continue;
                marker_id++;
                e.marker_id = marker_id; e.n_polymorphic = (uint32_t)n_poly;
                marker_out.put(line, (size_t)snprintf(line, sizeof(line), ">Marker%d %s %d|%d|%d|%d|%c|%c|%d\n", (int)marker_id, chrom_name, mstart, mlen, (int)s.loci, marker5, snp_c, ref_c, n_poly));
                marker_out.put(mseq);
                marker_out.put("\n", 1);
            }
            e.loci = s.loci; e.rank = 0;
            e.pvalue = 1.0 - binomial(tot_bases, (int)nonref, rate);
            e.bkgnd_rate = rate; e.local_reads = loc_tmm + loc_tm; e.local_subs = loc_tmm;
            e.num_reads = (uint32_t)tot_bases; e.num_subs = nonref; e.ref_base = s.ref_base;
            for (int b = 0; b < 5; b++) e.non_ref[b] = s.non_ref[b];
            lp.push_back(e);
        }
        if (lp.empty()) continue;
        // glibc qsort() is a stable merge sort for the < 25 000 elements CMTqsort hands it (MTqsort.cpp:456-457); beyond that the
        // reference's own tie order depends on its thread scheduling
        std::stable_sort(lp.begin(), lp.end(), [](const LociP &a, const LociP &b) { return a.pvalue < b.pvalue; });
        size_t n_snps = 0;
        for (size_t k = 0; k < lp.size(); k++) {           // Benjamini-Hochberg (:7120-7128)
            const double adj = ((double)(k + 1) / (double)lp.size()) * o.qvalue;
            if (lp[k].pvalue >= adj) break;
            n_snps++;
            lp[k].rank = (uint32_t)(k + 1);
        }
        lp.resize(n_snps);
        std::sort(lp.begin(), lp.end(), [](const LociP &a, const LociP &b) { return a.loci < b.loci; });
        auto ref_at = [&](int loci) -> uint32_t {          // tsSNPcnts.RefBase of an accepted SNP locus
            auto it = std::lower_bound(lp.begin(), lp.end(), (uint32_t)loci, [](const LociP &a, uint32_t l) { return a.loci < l; });
            return it != lp.end() && it->loci == (uint32_t)loci ? it->ref_base : 0u;
        };

        // ---- reads overlapping two / three SNP loci: CAligner::IterateReadsOverlapping (:9741-9815) over the sorted records
        struct Adjacent { int start = 0, end = 0; size_t first_iter = (size_t)-1, prev_iter = (size_t)-1; } adj[2];
        const size_t kNone = (size_t)-1, n_order = R.order.size();
        auto next_of = [&](size_t k) { return k + 1 < n_order ? k + 1 : kNone; };
        auto iterate = [&](bool tri, int start_loci, int end_loci) -> size_t {
            Adjacent &a = adj[tri ? 1 : 0];
            bool fresh = false;
            size_t nxt = kNone;
            if (a.first_iter == kNone || start_loci < a.start || end_loci < a.end) {
                nxt = ca.first; a.first_iter = ca.first; a.prev_iter = kNone; a.start = start_loci; a.end = end_loci; fresh = true;
            } else if (start_loci == a.start && end_loci == a.end) {
                if (a.prev_iter == ca.last) return kNone;
                nxt = next_of(a.prev_iter);
            } else {
                nxt = a.first_iter; a.prev_iter = nxt; a.start = start_loci; a.end = end_loci; fresh = true;
            }
            for (;;) {
                if (nxt == kNone) return kNone;
                const size_t i = R.order[nxt];
                if (piled(i)) {
                    const bk_hit &h = R.hits[i];
                    if (h.chrom_id != chrom) return kNone;
                    const int cs = (int)R.a_start(h, i), ce = (int)(R.a_start(h, i) + R.a_len(h, i) - 1);
                    if (cs <= start_loci && ce >= end_loci) {
                        a.prev_iter = nxt;
                        if (fresh) a.first_iter = nxt;
                        return nxt;
                    }
                    if (cs > start_loci) return kNone;
                }
                if (nxt == ca.last) return kNone;
                nxt = next_of(nxt);
            }
        };
        auto snp_base = [&](size_t k, uint32_t loci) -> uint32_t {     // CAligner::AdjAlignSNPBase (:1475-1518)
            const size_t i = R.order[k];
            const bk_hit &h = R.hits[i];
            const uint32_t as = R.a_start(h, i), ae = as + R.a_len(h, i) - 1;
            if (as > loci || ae < loci) return 7;
            const uint8_t *b = R.rs.bases.data() + R.rs.offs[R.RD(i)];
            if (h.strand == '+') return b[loci - h.match_loci] & 7u;
            uint32_t v = b[h.match_loci + h.match_len - loci - 1] & 7u;
            return v < 4 ? 3 - v : v;
        };

        int prev_di = -1, cur_tri = 0, prev_tri = -1, first_tri = -1, n_di = 0, n_tri = 0;
        for (size_t k = 0; k < lp.size(); k++) {
            LociP &e = lp[k];
            tot_snps++;
            const int rel_rank = std::max(1, (int)(999u - ((999u * e.rank) / (uint32_t)lp.size())));
            if (o.bed)
                snp_out.put(line, (size_t)snprintf(line, sizeof(line), "%s\t%d\t%d\tSNP_%d\t%d\t+\n", chrom_name, (int)e.loci, (int)e.loci + 1, (int)tot_snps, rel_rank));
            else if (o.vcf) {
                uint32_t thres = 0;
                for (uint32_t b = 0; b < 4; b++) if (b != e.ref_base && e.non_ref[b] > thres) thres = e.non_ref[b];
                thres = std::max((thres + 5) / 10, 1u);
                int ao = 0, fo = 0;
                for (uint32_t b = 0; b < 4; b++) {
                    if (b == e.ref_base || e.non_ref[b] < thres) continue;
                    if (ao > 0) { sz_alts[ao++] = ','; sz_freq[fo++] = ','; }
                    sz_alts[ao++] = base_uc(b); sz_alts[ao] = '\0';
                    fo += sprintf(&sz_freq[fo], "%1.4f", (double)e.non_ref[b] / e.num_reads);
                }
                const int phred = e.pvalue < 0.0000000001 ? 100 : (int)(0.5 + (10.0 * log10(1.0 / e.pvalue)));
                snp_out.put(line, (size_t)snprintf(line, sizeof(line), "%s\t%u\tSNP%d\t%c\t%s\t%d\tPASS\tAF=%s;DP=%d\n", chrom_name, e.loci + 1, (int)tot_snps, base_uc(e.ref_base),
                                                   sz_alts, phred, sz_freq, (int)e.num_reads));
            } else {
                uint32_t c5[5];
                for (int b = 0; b < 5; b++) c5[b] = e.non_ref[b];
                c5[e.ref_base] = e.num_reads - e.num_subs;
                snp_out.put(line, (size_t)snprintf(line, sizeof(line), "%d,\"SNP\",\"%s\",\"%s\",%d,%d,1,\"+\",%d,%f,%d,%d,\"%c\",%d,%d,%d,%d,%d,%f,%d,%d,%d,%d\n", (int)tot_snps, R.species.c_str(),
                                                   chrom_name, (int)e.loci, (int)e.loci, rel_rank, e.pvalue, (int)e.num_reads, (int)e.num_subs, base_uc(e.ref_base), (int)c5[0], (int)c5[1],
                                                   (int)c5[2], (int)c5[3], (int)c5[4], e.bkgnd_rate, (int)e.local_reads, (int)e.local_subs, (int)e.marker_id, (int)e.n_polymorphic));
            }
            // ---- DiSNPs / TriSNPs (:7246-7560)
            const int cur_di = (int)e.loci;
            if (prev_di != -1 && cur_di > 0 && (cur_di - prev_di) <= max_disnp_sep) {
                int n_over = 0, n_anti = 0, cnts[16] = {0};
                uint32_t refb[2] = {e.ref_base, ref_at(prev_di)}, bc[2][4] = {{0}};
                size_t rd;
                while ((rd = iterate(false, prev_di, cur_di)) != kNone) {
                    const uint32_t pb = snp_base(rd, (uint32_t)prev_di);
                    if (pb > 3) continue;
                    const uint32_t cb = snp_base(rd, (uint32_t)cur_di);
                    if (cb > 3) continue;
                    bc[1][pb]++; bc[0][cb]++;
                    n_over++;
                    if (R.hits[R.order[rd]].strand == '-') n_anti++;
                    cnts[((pb & 3) << 2) | (cb & 3)]++;
                }
                if (n_over >= o.min_reads) {
                    int n_hap = 0;
                    const int th = std::max(3, n_over / 20);
                    for (int q = 0; q < 16; q++) if (cnts[q] >= th) n_hap++;
                    n_di++;
                    std::string row;
                    row += std::to_string(n_di) + ",\"DiSNPs\",\"" + R.species + "\",\"" + chrom_name + "\",";
                    const int lab[2] = {prev_di, cur_di};
                    for (int q = 0; q < 2; q++) {
                        snprintf(line, sizeof(line), "%d,\"%c\",%d,%d,%d,%d,0,", lab[q], base_lc(refb[q]), (int)bc[q][0], (int)bc[q][1], (int)bc[q][2], (int)bc[q][3]);
                        row += line;
                    }
                    row += std::to_string(n_over) + "," + std::to_string(n_anti) + "," + std::to_string(n_hap);
                    for (int q = 0; q < 16; q++) row += "," + std::to_string(cnts[q]);
                    row += "\n";
                    di_out.put(row);
                }
            }
            cur_tri = (int)e.loci;
            if (first_tri != -1 && prev_tri > 0 && cur_tri > 0 && (cur_tri - first_tri) <= max_disnp_sep) {
                int n_over = 0, n_anti = 0, cnts[64] = {0};
                uint32_t refb[3] = {e.ref_base, ref_at(prev_tri), ref_at(first_tri)}, bc[3][4] = {{0}};
                size_t rd;
                while ((rd = iterate(true, first_tri, cur_tri)) != kNone) {
                    const uint32_t fb = snp_base(rd, (uint32_t)first_tri);
                    if (fb > 3) continue;
                    const uint32_t pb = snp_base(rd, (uint32_t)prev_tri);
                    if (pb > 3) continue;
                    const uint32_t cb = snp_base(rd, (uint32_t)cur_tri);
                    if (cb > 3) continue;
                    bc[2][fb]++; bc[1][pb]++; bc[0][cb]++;
                    n_over++;
                    if (R.hits[R.order[rd]].strand == '-') n_anti++;
                    cnts[((fb & 3) << 4) | ((pb & 3) << 2) | (cb & 3)]++;
                }
                if (n_over >= o.min_reads) {
                    int n_hap = 0;
                    const int th = std::max(3, n_over / 20);
                    for (int q = 0; q < 64; q++) if (cnts[q] >= th) n_hap++;
                    n_tri++;
                    std::string row;
                    row += std::to_string(n_tri) + ",\"TriSNPs\",\"" + R.species + "\",\"" + chrom_name + "\",";
                    const int lab[3] = {first_tri, prev_tri, cur_tri};
                    for (int q = 0; q < 3; q++) {
                        snprintf(line, sizeof(line), "%d,\"%c\",%d,%d,%d,%d,0,", lab[q], base_lc(refb[q]), (int)bc[q][0], (int)bc[q][1], (int)bc[q][2], (int)bc[q][3]);
                        row += line;
                    }
                    row += std::to_string(n_over) + "," + std::to_string(n_anti) + "," + std::to_string(n_hap);
                    for (int q = 0; q < 64; q++) row += "," + std::to_string(cnts[q]);
                    row += "\n";
                    tri_out.put(row);
                }
            }
            prev_di = cur_di;
            first_tri = prev_tri;
            prev_tri = cur_tri;
            if (!cents.empty() && e.loci >= 3 && e.loci < (uint32_t)R.ents[chrom - 1].seq_len - 3) {          // :7563-7590
                uint32_t c7[7 * 7];
                rc = bk_snp_counts(ctx, chrom, e.loci - 3, 7, c7);
                if (rc) { diag("Fatal: SNP centroid counts could not be fetched: %s", bk_strerror(rc)); return 1; }
                uint32_t idx = 0;
                bool ok = true;
                for (int q = 0; q < 7; q++) { ok = ok && c7[q * 7 + 6] < 4; idx = (idx << 2) | (c7[q * 7 + 6] & 3u); }
                if (ok) {
                    Centroid &c = cents[idx];
                    c.ref_cnt += c7[3 * 7 + 0];
                    for (int b = 0; b < 5; b++) c.non_ref[b] += c7[3 * 7 + 1 + b];
                    c.num_snps++;
                }
            }
        }
    }
    snp_out.close(); di_out.close(); tri_out.close();
    if (!cents.empty()) {                                  // ProcessSNPs :8001-8032
        cent_out.put(std::string("\"CentroidID\",\"Seq\",\"NumInsts\",\"NumSNPs\",\"RefBase\",\"RefBaseCnt\",\"BaseA\",\"BaseC\",\"BaseG\",\"BaseT\",\"BaseN\"\n"));
        for (uint32_t k = 0; k < BK_SNP_CENTROIDS; k++) {
            char seq7[8];
            for (int q = 0; q < 7; q++) seq7[q] = kCentroidCase[(k >> (2 * (6 - q))) & 3u];
            seq7[7] = '\0';
            const Centroid &c = cents[k];
            cent_out.put(line, (size_t)snprintf(line, sizeof(line), "%d,\"%s\",%d,%d,\"%c\",%d,%d,%d,%d,%d,%d\n", (int)k + 1, seq7, (int)cent_insts[k], (int)c.num_snps, base_uc((k >> 6) & 3u),
                                                (int)c.ref_cnt, (int)c.non_ref[0], (int)c.non_ref[1], (int)c.non_ref[2], (int)c.non_ref[3], (int)c.non_ref[4]));
        }
        cent_out.close();
    }
    if (o.marker_len) {
        marker_out.close();
        diag("Marker processing completed with %d marker sequences writtten to file '%s", (int)marker_id, marker_path.c_str());
    }
    diag("SNP processing completed with %d putative SNPs discovered", (int)tot_snps);
    diag("There are %lld aligned loci bases which are covered by %lld read bases with mean coverage of %1.2f", (long long)loci_covered, (long long)bases_coverage,
         (double)bases_coverage / (double)loci_covered);
    return 0;
}

}  // namespace bkcli
