// bk_snp_host.cpp - SNP pile-up and screening entry points (include/biokanga_amd.h: bk_snp_*; CAligner::ProcessSNPs, Aligner.cpp:7609-8071);
// the kernels are in bk_snp.hip.
#include "bk_engine_int.h"

using namespace bk;

extern "C" {

// ---- SNP pile-up and screening (see include/biokanga_amd.h) -------------------------------------
int bk_snp_reset(bk_ctx *c)
{
    if (!c) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)c->ix.n * 6 * sizeof(uint32_t);
    if (!c->d_snp_planes) HIP_TRY(dev_malloc(&c->d_snp_planes, bytes));
    if (!c->d_snp_tot) HIP_TRY(dev_malloc(&c->d_snp_tot, 4 * 8));
    HIP_TRY(clear_dev(c->d_snp_planes, bytes, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BK_OK;
}

int bk_snp_pileup(bk_ctx *c, const uint8_t *bases, const uint64_t *offs, const uint32_t *lens, uint32_t nreads, const bk_snp_aln *alns,
                  uint64_t n_alns)
{
    if (!c || (n_alns && (!bases || !offs || !lens || !alns || !nreads))) return BK_ERR_PARAMS;
    if (!c->d_snp_planes) return BK_ERR_PARAMS;                      // bk_snp_reset() first
    if (!n_alns) return BK_OK;
    uint64_t lo = ~0ULL, hi = 0;
    for (uint32_t i = 0; i < nreads; i++) { lo = std::min(lo, offs[i]); hi = std::max(hi, offs[i] + lens[i]); }
    uint32_t max_id = 0;
    for (const auto &e : c->entries) max_id = std::max(max_id, e.entry_id);
    for (uint64_t i = 0; i < n_alns; i++) {
        const bk_snp_aln &a = alns[i];
        if (a.read_idx >= nreads || (uint32_t)a.read_ofs + a.len > lens[a.read_idx] || (a.strand != '+' && a.strand != '-')) return BK_ERR_PARAMS;
        if (a.chrom_id == 0 || a.chrom_id > max_id) return BK_ERR_PARAMS;
    }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    uint8_t *d_bases = nullptr;
    uint64_t *d_offs = nullptr;
    bk_snp_aln *d_alns = nullptr;
    std::vector<uint64_t> rel(nreads);
    for (uint32_t i = 0; i < nreads; i++) rel[i] = offs[i] - lo;
    int rc = BK_OK;
    auto try_ = [&](hipError_t e) { if (e != hipSuccess && rc == BK_OK) rc = e == hipErrorOutOfMemory ? BK_ERR_MEM : BK_ERR_INTERNAL; return e == hipSuccess; };
    if (try_(dev_malloc(&d_bases, hi - lo + 16)) && try_(dev_malloc(&d_offs, (size_t)nreads * 8)) && try_(dev_malloc(&d_alns, (size_t)n_alns * sizeof(bk_snp_aln)))) {
        try_(hipMemcpyAsync(d_bases, bases + lo, hi - lo, hipMemcpyHostToDevice, s));
        try_(hipMemcpyAsync(d_offs, rel.data(), (size_t)nreads * 8, hipMemcpyHostToDevice, s));
        try_(hipMemcpyAsync(d_alns, alns, (size_t)n_alns * sizeof(bk_snp_aln), hipMemcpyHostToDevice, s));
        if (rc == BK_OK) {
            launch_snp_pileup(c->ix, d_bases, d_offs, c->d_id2idx, d_alns, n_alns, c->d_snp_planes, s);
            try_(hipGetLastError());
        }
        try_(hipStreamSynchronize(s));
    }
    free_dev(d_bases); free_dev(d_offs); free_dev(d_alns);
    return rc;
}

int bk_snp_pileup_device(bk_ctx *c, const void *d_bases, const void *d_offs, uint32_t nreads, const void *d_alns, uint64_t n_alns, int sync)
{
    if (!c || (n_alns && (!d_bases || !d_offs || !d_alns || !nreads))) return BK_ERR_PARAMS;
    if (!c->d_snp_planes) return BK_ERR_PARAMS;
    if (!n_alns) return BK_OK;
    HIP_TRY(hipSetDevice(c->device));
    launch_snp_pileup(c->ix, (const uint8_t *)d_bases, (const uint64_t *)d_offs, c->d_id2idx, (const bk_snp_aln *)d_alns, n_alns, c->d_snp_planes, c->stream);
    HIP_TRY(hipGetLastError());
    if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
    return BK_OK;
}

int bk_snp_counts(bk_ctx *c, uint32_t chrom_id, uint32_t loci, uint32_t n, uint32_t *out)
{
    if (!c || !out || !c->d_snp_planes) return BK_ERR_PARAMS;
    const bk_entry_info *ent = nullptr;
    for (const auto &e : c->entries) if (e.entry_id == chrom_id) { ent = &e; break; }
    if (!ent || (uint64_t)loci + n > ent->seq_len) return BK_ERR_PARAMS;
    if (!n) return BK_OK;
    HIP_TRY(hipSetDevice(c->device));
    uint32_t *d_out = nullptr;
    HIP_TRY(dev_malloc(&d_out, (size_t)n * 7 * 4));
    launch_snp_gather(c->ix, c->d_snp_planes, ent->start_ofs + loci, n, d_out, c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n * 7 * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    free_dev(d_out);
    return e == hipSuccess ? BK_OK : BK_ERR_INTERNAL;
}

int bk_snp_centroid_insts(bk_ctx *c, uint32_t chrom_id, int32_t min_reads, uint32_t *num_insts)
{
    if (!c || !num_insts || min_reads < 1 || !c->d_snp_planes) return BK_ERR_PARAMS;
    const bk_entry_info *ent = nullptr;
    for (const auto &e : c->entries) if (e.entry_id == chrom_id) { ent = &e; break; }
    if (!ent) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(c->device));
    uint32_t *d_hist = nullptr;
    HIP_TRY(dev_malloc(&d_hist, BK_SNP_CENTROIDS * 4));
    std::vector<uint32_t> h(BK_SNP_CENTROIDS);
    hipError_t e = hipMemsetAsync(d_hist, 0, BK_SNP_CENTROIDS * 4, c->stream);
    if (e == hipSuccess) { launch_snp_centroids(c->ix, c->d_snp_planes, ent->start_ofs, (uint32_t)ent->seq_len, (uint32_t)min_reads, d_hist, c->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_hist, BK_SNP_CENTROIDS * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    free_dev(d_hist);
    if (e != hipSuccess) return BK_ERR_INTERNAL;
    for (int i = 0; i < BK_SNP_CENTROIDS; i++) num_insts[i] += h[i];
    return BK_OK;
}

int bk_snp_sites(bk_ctx *c, uint32_t chrom_id, int32_t min_reads, double min_nonref_prop, const bk_snp_site **sites, uint64_t *n_sites,
                 bk_snp_chrom *totals)
{
    if (!c || !sites || !n_sites || !totals || min_reads < 1 || !(min_nonref_prop >= 0.0)) return BK_ERR_PARAMS;
    if (!c->d_snp_planes) return BK_ERR_PARAMS;
    const bk_entry_info *ent = nullptr;
    for (const auto &e : c->entries) if (e.entry_id == chrom_id) { ent = &e; break; }
    if (!ent) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (!c->cap_snp_sites) {
        HIP_TRY(dev_malloc(&c->d_snp_sites, (size_t)(1u << 20) * sizeof(bk_snp_site)));
        c->cap_snp_sites = 1u << 20;
    }
    unsigned long long h_tot[4] = {0, 0, 0, 0};
    uint32_t n = 0;
    for (;;) {                                                        // second pass only when the list outgrew its buffer
        HIP_TRY(hipMemsetAsync(c->d_small, 0, 16 * 4, s));
        HIP_TRY(hipMemsetAsync(c->d_snp_tot, 0, 4 * 8, s));
        launch_snp_sites(c->ix, c->d_snp_planes, ent->start_ofs, (uint32_t)ent->seq_len, (uint32_t)min_reads, min_nonref_prop, c->d_snp_sites,
                         c->cap_snp_sites, c->d_small, c->d_snp_tot, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&n, c->d_small, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(h_tot, c->d_snp_tot, sizeof(h_tot), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (n <= c->cap_snp_sites) break;
        free_dev(c->d_snp_sites);
        c->d_snp_sites = nullptr;
        c->cap_snp_sites = 0;
        HIP_TRY(dev_malloc(&c->d_snp_sites, (size_t)n * sizeof(bk_snp_site)));
        c->cap_snp_sites = n;
    }
    c->snp_sites.resize(n);
    if (n) HIP_TRY(hipMemcpy(c->snp_sites.data(), c->d_snp_sites, (size_t)n * sizeof(bk_snp_site), hipMemcpyDeviceToHost));
    std::sort(c->snp_sites.begin(), c->snp_sites.end(), [](const bk_snp_site &a, const bk_snp_site &b) { return a.loci < b.loci; });
    *sites = n ? c->snp_sites.data() : nullptr;
    *n_sites = n;
    totals->tot_match = h_tot[0]; totals->tot_mismatch = h_tot[1]; totals->loci_covered = h_tot[2]; totals->bases_coverage = h_tot[3];
    return BK_OK;
}


}  // extern "C"
