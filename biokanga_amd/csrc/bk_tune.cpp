// bk_tune.cpp - a context's parameters and knobs (include/biokanga_amd.h: bk_ctx_reserve, bk_ctx_set_chrom_filter, bk_ctx_set_params,
// bk_ctx_tune, the index's descriptive getters).  Results never depend on a knob: the test-suite runs independent implementations of the
// same step against each other through them.
#include "bk_engine_int.h"

using namespace bk;

extern "C" {

int bk_ctx_reserve(bk_ctx *c, uint32_t max_batch_reads, uint32_t max_read_len)
{
    if (!c || !max_batch_reads || !max_read_len || max_read_len > (uint32_t)kMaxReadLenAbs) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t n = std::min(max_batch_reads, c->chunk_reads);
    const uint32_t wpr = words_per_read(max_read_len);
    const bool reg_path = c->use_wave && max_read_len <= 16u * (uint32_t)kNwLongest;
    const bool two_bit = reg_path && c->ix.tgt2 != nullptr;
    const uint32_t ivc = iv_cores_for(c, max_read_len);
    int rc = ensure_batch_scratch(c, n, wpr, two_bit ? rd2w_for(max_read_len) : 0u, ivc);
    if (rc) return rc;
    if (c->ix.k2) {
        // pass B's work list: at most one item per (read, strand, core)
        const uint64_t lanes = (uint64_t)n * ivc * (c->cfg.align_strand == 0 ? 2u : 1u);
        if (lanes > c->cap_slist) {
            free_dev(c->d_slist);
            free_dev(c->d_slist_stage);
            c->d_slist = c->d_slist_stage = nullptr;
            c->cap_slist = 0;
            HIP_TRY(dev_malloc(&c->d_slist, lanes * 4));
            HIP_TRY(dev_malloc(&c->d_slist_stage, (lanes + (kListStripes + 2) * 1024) * 4));
            c->cap_slist = lanes;
        }
    }
    if (c->sort_lists) { rc = ensure_sort_scratch(c, n, c->stream); if (rc) return rc; }      // (grown when a phase's list is longer)
    if (c->use_wave && c->ix.isa == nullptr) { rc = size_heavy_scratch(c); if (rc) return rc; }     // hash-set dedupe of the wave kernel
    return BK_OK;
}

int bk_ctx_set_chrom_filter(bk_ctx *c, const uint8_t *accept, uint32_t n)
{
    if (!c || (n && !accept)) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_dev(c->d_chrom_accept);
    c->d_chrom_accept = nullptr;
    c->n_chrom_accept = 0;
    if (!n) return BK_OK;
    HIP_TRY(dev_malloc(&c->d_chrom_accept, n));
    HIP_TRY(hipMemcpy(c->d_chrom_accept, accept, n, hipMemcpyHostToDevice));
    c->n_chrom_accept = n;
    return BK_OK;
}

int bk_ctx_set_params(bk_ctx *c, const bk_align_params *p)
{
    if (!c || !p) return BK_ERR_PARAMS;
    bk_align_params old = c->params;
    c->params = *p;
    if (c->params.max_ml == 0) c->params.max_ml = 1;
    int rc = derive_cfg(c);
    if (rc) { c->params = old; derive_cfg(c); return rc; }
    (void)hipSetDevice(c->device);
    return BK_OK;
}

int64_t bk_ctx_tune(bk_ctx *c, const char *name, int64_t value)
{
    if (!c || !name) return BK_ERR_PARAMS;
    (void)hipSetDevice(c->device);
    std::string n(name);
    if (n == "heavy_thresh") {
        int64_t old = c->cfg.heavy_thresh;
        if (value < 0 || value > 100) return BK_ERR_PARAMS;
        c->cfg.heavy_thresh = (int)value;     // 0 routes every call with a non-empty interval to k_heavy
        return old;
    }
    if (n == "chunk_reads") {
        int64_t old = c->chunk_reads;
        if (value < 1 || value > (1LL << 30)) return BK_ERR_PARAMS;
        c->chunk_reads = (uint32_t)value;
        return old;
    }
    if (n == "kmer_bits" || n == "use_ktab") {
        int64_t old = n == "use_ktab" ? c->use_ktab : c->ix.k;
        if (n == "use_ktab") c->use_ktab = value ? 1 : 0;
        else { if (value < 2 || value > 16) return BK_ERR_PARAMS; c->k_req = (int)value; }
        int rc = build_tables(c);
        return rc ? rc : old;
    }
    if (n == "ktab_wide") {                // the k-mer table of an index beyond 2^32 suffixes on any index (tests): 1 packed (ktab_hi + 32-bit offsets), 2 plain 64-bit starts
        int64_t old = c->ktab_wide;
        c->ktab_wide = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
        int rc = build_tables(c);
        return rc ? rc : old;
    }
    if (n == "ktab_packed") return c->ix.ktab_hi != nullptr;       // (read only)
    if (n == "use_k3") {                   // how many key arrays behind the second-level keys (rebuilt with the tables)
        int64_t old = c->use_k3;
        c->use_k3 = value < 0 ? 0 : (value > kMoreKeys ? kMoreKeys : (int)value);
        int rc = build_tables(c);
        return rc ? rc : old;
    }
    if (n == "k3_resident") return (c->ix.kx[0] != nullptr) + (c->ix.kx[1] != nullptr);
    if (n == "ktab2_resident") return c->ix.ktab2 != nullptr;
    if (n == "ktab2_elem") return c->ix.ktab2 != nullptr && c->ix.ktab2_elem;      // (read only) .. and its buckets of one suffix carry suffix array elements
    if (n == "grow_after_reads") { int64_t old = (int64_t)c->grow_after; if (value > 0) c->grow_after = (uint64_t)value; return old; }
    if (n == "grow_state") return c->grow_enabled ? c->grow_state.load() : 5;          // (0 .. 4: bk_ctx_int.h; 5: not a growing context)
    if (n == "image_wait") {               // BK_CTX_GROW_IMAGE: make the long-run tables now if they are not under way, wait for them, take them in
        if (c->grow_enabled) {
            grow_tick(c, 0, true);
            if (c->grow_state.load() != 4 && c->grow_state.load() != 0) grow_take_in(c);
        }
        return (c->ix.kx[0] != nullptr) + (c->ix.kx[1] != nullptr) + (c->ix.ktab2 != nullptr ? 4 : 0);
    }
    if (n == "use_ktab2") {                // k-mer table entries of two words: 0 no, 1 a bucket of one suffix carries its key, 2 its suffix array element (rebuilt with the tables)
        int64_t old = c->use_ktab2;
        c->use_ktab2 = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
        c->grow_elem = c->use_ktab2 == 2;
        int rc = build_tables(c);
        return rc ? rc : old;
    }
    if (n == "sort_lists") {
        int64_t old = c->sort_lists;
        c->sort_lists = (int)value & 7;
        c->sort_lists_set = true;
        return old;
    }
    if (n == "use_tgt2") {
        int64_t old = c->use_tgt2;
        c->use_tgt2 = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
        int rc = build_tgt2(c);
        return rc ? rc : old;
    }
    if (n == "wave_waves") {
        int64_t old = c->wave_waves;
        if (value < 64 || value > 65536) return BK_ERR_PARAMS;
        c->wave_waves = (uint32_t)value;
        return old;
    }
    if (n == "async_phases") {             // 0: the phase loop reads its counts back between launches (exact launch sizes), as every other configuration does
        int64_t old = c->async_phases;
        c->async_phases = value ? 1 : 0;
        return old;
    }
    if (n == "swin_resident") return c->d_swin != nullptr ? 1 : 0;      // (read only: whether the window array is in HBM right now)
    if (n == "use_swin") {
        int64_t old = c->use_swin;
        c->use_swin = value < 0 ? 0 : (value > 3 ? 3 : (int)value);
        c->swin_denied = false;
        if (c->d_swin && (!c->use_swin || (c->use_swin == 3) != (c->d_swmap == nullptr))) {
            if (c->stream) (void)hipStreamSynchronize(c->stream);
            free_dev(c->d_swin);
            free_dev(c->d_swmap);
            c->d_swin = nullptr;
            c->d_swmap = nullptr;
            c->ix.swin = nullptr;
            c->ix.swmap = nullptr;
            c->swin_bytes = 0;
        }
        return old;
    }
    if (n == "swin_skip_short") {           // the partial window array's rule without the reads' this many shortest core lengths; applies to the next build
        int64_t old = c->swin_skip_short;
        c->swin_skip_short = value < 0 ? 0 : (int)std::min<int64_t>(value, kSwLevels - 1);
        return old;
    }
    if (n == "swin_budget_kb") {            // most the partial window array may take (0: what the free memory allows); applies to the next build
        int64_t old = (int64_t)(c->swin_budget >> 10);
        c->swin_budget = value > 0 ? (uint64_t)value << 10 : 0;
        return old;
    }
    // (read only) what the window array occupies, what making it took, how much of the suffix array it holds
    if (n == "swin_mbytes") return (int64_t)(c->swin_bytes >> 20);
    if (n == "swin_setup_us") return (int64_t)(c->swin_setup_s * 1e6);
    if (n == "swin_covered_ppm") return c->d_swin ? (int64_t)(c->swin_covered * 1e6) : 0;
    if (n == "swin_core_lens") return c->d_swmap ? c->swin_w : 0;      // shortest | longest << 8 | levels << 16 of the core lengths its coverage was made for
    if (n == "use_isa") {
        int64_t old = c->use_isa;
        c->use_isa = value ? 1 : 0;
        int rc = build_tables(c);
        return rc ? rc : old;
    }
    if (n == "use_k2") {
        int64_t old = c->use_k2;
        c->use_k2 = value ? 1 : 0;
        int rc = build_tables(c);
        return rc ? rc : old;
    }
    if (n == "use_iv32") {
        int64_t old = c->use_iv32;
        c->use_iv32 = value ? 1 : 0;
        return old;
    }
    if (n == "lazy_search") {
        int64_t old = c->lazy_search;
        c->lazy_search = value ? 1 : 0;
        return old;
    }
    if (n == "use_wave") {
        int64_t old = c->use_wave;
        c->use_wave = value ? 1 : 0;
        int rc = build_tables(c);
        return rc ? rc : old;
    }
    if (n == "force_rccl") {              // bk_seq_counts_allreduce goes through RCCL even when every context sits on one device
        int64_t old = c->force_rccl ? 1 : 0;
        c->force_rccl = value != 0;
        return old;
    }
    if (n == "rccl_allreduces") return (int64_t)c->rccl_allreduces;      // (read only) reductions of this context that went through RCCL
    if (n == "rccl_ranks") return (int64_t)c->rccl_ranks;                // (read only) .. and the ranks of the last one's communicator
    if (n == "debug_stop_phase") {        // test hook: the next batches stop behind the search of phase value - 1 (bk_debug_intervals); 0: off
        int64_t old = c->dbg_stop_phase + 1;
        c->dbg_stop_phase = value <= 0 ? -1 : (int)value - 1;
        c->dbg_valid = false;
        return old;
    }
    if (n == "max_read_len") {
        int64_t old = c->max_read_len;
        if (value < 16 || value > kMaxReadLenAbs) return BK_ERR_PARAMS;
        c->max_read_len = (int)value;
        return old;
    }
    return BK_ERR_PARAMS;
}


uint32_t bk_num_entries(const bk_ctx *c) { return c ? (uint32_t)c->entries.size() : 0; }
int bk_get_entry(const bk_ctx *c, uint32_t idx, bk_entry_info *out)
{
    if (!c || !out || idx >= c->entries.size()) return BK_ERR_PARAMS;
    *out = c->entries[idx];
    return BK_OK;
}
const char *bk_dataset_name(const bk_ctx *c) { return c ? c->dataset.c_str() : ""; }
uint64_t bk_concat_len(const bk_ctx *c) { return c ? c->ix.n : 0; }
int bk_sfx_el_size(const bk_ctx *c) { return c ? (int)c->el_size : 0; }
int bk_min_core_len(const bk_ctx *c) { return c ? c->cfg.min_core_len : 0; }


}  // extern "C"
