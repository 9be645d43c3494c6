// report.h - output side of `biokanga align`: SAM / BAM / CSV / BED writers, -O statistics, -j / -J read subsets, .jct / .ind files
// (CAligner::WriteBAMReadHits, ReportBAMread, WriteReadHits, WriteBasicCountStats, ReportNoneAligned / ReportMultiAlign).
#pragma once
#include <functional>
#include "../../../include/biokanga_amd.h"
#include "cli_common.h"
#include "post_filters.h"

namespace bkcli {

// ---------------------------------------------------------------------------------------------
// Everything the reporting functions need about one `align` run (references into cmd_align's state)
struct Report {
    const Args &a;
    ReadStore &rs;
    std::vector<bk_hit> &hits;                    // one per record (a read, or a reported locus in -r5)
    const std::vector<bk_entry_info> &ents;
    const std::string &species;
    uint32_t n_ent;
    const std::vector<uint32_t> &src;             // -r5: record -> read
    const std::vector<bk_seg2> &seg2;             // per read: second segment / chimeric trims
    const bk::FlankTrims &trims;                  // per record
    const std::vector<int> &multi_dist;
    const std::vector<uint32_t> &order;           // records in the reference's output order
    int pe_mode, ml_mode, max_ml, fmt, nthreads, micro_indel, splice_len, max_rpt_sam_seqs;
    bk_ctx *ctx = nullptr;                        // a context whose device formats plain SAM records (bk_sam_format); may be null
    SamPrealloc *pre = nullptr;                   // SAM text: the output file, created early and being preallocated; may be null
    bk_sam_prep *sam_prep = nullptr;              // the device formatter's head start (bk_sam_prepare), consumed or freed by report_text / report_bam
    // the reads in the packed form the alignment was fed from, as the head start was given them (bk_sam_job.pk_*); the arrays themselves
    // may be gone once the head start has taken them to the device
    // .. and the read store's bases too, when nothing but the host's own formatter could still want them: restore_reads() brings them
    // back (null: they never left)
    std::function<int()> restore_reads;
    const uint32_t *pk_words = nullptr;
    uint64_t n_pk_words = 0;
    const uint16_t *pk_lens16 = nullptr;
    const bk_nbase *pk_exc = nullptr;
    uint64_t n_pk_exc = 0;

    size_t RD(size_t i) const { return src.empty() ? i : (size_t)src[i]; }
    bool has_seg2(size_t i) const { return !seg2.empty() && (seg2[RD(i)].flags & 5); }       // FlgInDel or FlgSplice
    uint32_t TL(size_t i) const { return trims.empty() ? 0u : trims.left[i]; }
    uint32_t TR(size_t i) const { return trims.empty() ? 0u : trims.right[i]; }
    uint32_t a_start(const bk_hit &h, size_t i) const { return h.match_loci + (h.strand == '+' ? TL(i) : TR(i)); }       // AdjStartLoci
    uint32_t a_len(const bk_hit &h, size_t i) const { return (uint32_t)h.match_len - TL(i) - TR(i); }                      // AdjHitLen
    uint32_t a_mm(const bk_hit &h, size_t i) const { return trims.empty() ? h.mismatches : trims.mismatches[i]; }         // TrimMismatches
};

void report_read_subset(Report &R, const char *opt, const char *tag, bool (*want)(uint8_t));
void report_stats(Report &R);
void report_jct_for_sam(Report &R);
int report_bam(Report &R, const std::string &opath);       // 0, or 1 after a fatal message
int report_text(Report &R);

}  // namespace bkcli
