// bk_engine.cpp - the batch driver behind the C ABI in include/biokanga_amd.h: batch scratch, the phase loop of CSfxArrayV3::AlignReads over
// whole batches (libbiokanga/SfxArrayV2.cpp:7666-7760), the paired-end pass, the packed and device-resident batch forms, counters, timing.
// Compiled with hipcc; device code lives in the kernel files (bk_prep / bk_search / bk_extend / bk_wave / bk_heavy / bk_rescue .hip); the index
// image and the context's life are bk_image.cpp, knobs bk_tune.cpp, the exchange step bk_exchange.cpp.  No CPU fallback exists: every
// compute entry point needs a HIP device and fails with BK_ERR_NODEVICE otherwise.
#include "bk_engine_int.h"

namespace bk {
int size_heavy_scratch(bk_ctx *c)
{
    // worst-case inserts per strand pass = min(node cap, cores(max_read_len) * MaxIter)
    int slides = std::max(1, (c->cfg.slides_per100 * c->max_read_len + 99) / 100);
    uint64_t worst = c->cfg.max_iter ? (uint64_t)slides * (uint64_t)c->cfg.max_iter : kNodeCap;
    if (worst > kNodeCap) worst = kNodeCap;
    uint32_t ts = 1024;
    while (ts < 2 * worst) ts <<= 1;
    uint32_t slots = 4096;                  // one per resident wave of the hash-set forms (four blocks of four waves a CU: 40 KB of LDS each) when they fit in 16 GB
    while ((uint64_t)slots * ts * 8 > (16ULL << 30) && slots > 64) slots >>= 1;
    if (c->hs.htab && c->hs.tab_size == ts && c->hs.n_slots == slots) return BK_OK;
    free_dev(c->hs.htab);
    free_dev(c->hs.slot_epoch);
    c->hs = HeavyScratch{};
    HIP_TRY(dev_malloc(&c->hs.htab, (size_t)slots * ts * 8));
    HIP_TRY(dev_malloc(&c->hs.slot_epoch, (size_t)slots * 4));
    launch_fill_u64(c->hs.htab, (uint64_t)slots * ts, 0ULL, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(c->hs.slot_epoch, 0, (size_t)slots * 4, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->hs.tab_size = ts;
    c->hs.n_slots = slots;
    return BK_OK;
}


int ensure_batch_scratch(bk_ctx *c, uint32_t n_reads, uint32_t wpr, uint32_t rd2w, uint32_t iv_cores)
{
    if (n_reads <= c->cap_reads && wpr <= c->cap_wpr && rd2w <= c->cap_rd2w && iv_cores <= c->cap_iv_cores) return BK_OK;
    uint32_t nr = std::max(n_reads, c->cap_reads), w = std::max(wpr, c->cap_wpr), w2 = std::max(rd2w, c->cap_rd2w);
    const uint32_t ivc = std::max(iv_cores, c->cap_iv_cores);
    free_dev(c->d_rd4); free_dev(c->d_iv_first); free_dev(c->d_iv_n); free_dev(c->d_rd2); free_dev(c->d_iv2); free_dev(c->d_rmeta);
    c->d_rmeta = nullptr;
    free_dev(c->d_act[0]); free_dev(c->d_act[1]); free_dev(c->d_heavy); free_dev(c->d_wave); free_dev(c->d_iv32); free_dev(c->d_wave_work);
    c->d_iv32 = nullptr;
    c->d_wave_work = nullptr;
    c->d_rd4 = nullptr; c->d_iv_first = nullptr; c->d_iv_n = nullptr; c->d_rd2 = nullptr; c->d_iv2 = nullptr;
    c->d_act[0] = c->d_act[1] = c->d_heavy = c->d_wave = nullptr;
    for (int i = 0; i < 3; i++) { free_dev(c->d_stage[i]); c->d_stage[i] = nullptr; }
    c->cap_reads = 0;
    HIP_TRY(dev_malloc(&c->d_rd4, (size_t)nr * 2 * w * 8));
    if (w2) HIP_TRY(dev_malloc(&c->d_rd2, (size_t)nr * 2 * w2 * 8 + 64));        // (+ the words a 32-base fetch at a row's end runs into)
    HIP_TRY(dev_malloc(&c->d_rmeta, ((size_t)nr + 2) / 2 * 8));
    if (c->d_sa_hi == nullptr && c->ix.n < (1ULL << 32))
        HIP_TRY(dev_malloc(&c->d_iv2, (size_t)nr * 2 * ivc * 8));
    else {
        HIP_TRY(dev_malloc(&c->d_iv_first, (size_t)nr * 2 * ivc * 8));
        HIP_TRY(dev_malloc(&c->d_iv_n, (size_t)nr * 2 * ivc * 4));
    }
    HIP_TRY(dev_malloc(&c->d_act[0], (size_t)nr * 4));
    HIP_TRY(dev_malloc(&c->d_act[1], (size_t)nr * 4));
    HIP_TRY(dev_malloc(&c->d_heavy, (size_t)nr * 4));
    HIP_TRY(dev_malloc(&c->d_wave, (size_t)nr * 4));
    for (int i = 0; i < 3; i++) HIP_TRY(dev_malloc(&c->d_stage[i], ((size_t)nr + (kListStripes + 2) * 1024) * 4));      // striped forms of the lists (StripedList)
    if (!c->d_stripe_cnt) {
        HIP_TRY(dev_malloc(&c->d_stripe_cnt, (size_t)2 * kListStripes * 16 * 4));
        HIP_TRY(dev_zero_now(c->d_stripe_cnt, (size_t)2 * kListStripes * 16 * 4));
    }
    if (c->d_iv2) HIP_TRY(dev_malloc(&c->d_iv32, (size_t)nr * 2 * 8));
    HIP_TRY(dev_malloc(&c->d_wave_work, (size_t)nr * 4));
    c->cap_reads = nr;
    c->cap_wpr = w;
    c->cap_rd2w = w2;
    c->cap_iv_cores = ivc;
    return BK_OK;
}

struct EvTimer {
    bk_ctx *c;
    bool on = true;                     // off: no events (a call that returns before its kernels have run cannot read them)
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> spans;   // kind, (start, stop)
    size_t next_ev = 0;
    hipEvent_t get()
    {
        if (!on) return nullptr;
        if (next_ev == c->ev_pool.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            c->ev_pool.push_back(e);
        }
        return c->ev_pool[next_ev++];
    }
    hipEvent_t begin(hipStream_t s)
    {
        hipEvent_t e = get();
        if (e) (void)hipEventRecord(e, s);
        return e;
    }
    void end(int kind, hipEvent_t b, hipStream_t s)
    {
        hipEvent_t e = get();
        if (!e) return;
        (void)hipEventRecord(e, s);
        spans.push_back({kind, {b, e}});
    }
};

// one chunk of reads, all phases.  Blocking (host reads back the active counts between phases).
// Work-list grouping.  A list is reordered so that items touching the same part of the index (same k-mer
// bucket / same core interval: reads from one repeat family) sit next to each other: neighbouring lanes then
// walk the same bisection path (their loads coalesce) and neighbouring waves fetch the same target windows
// (L2 hits instead of HBM misses).  Results do not depend on the order.
int ensure_sort_scratch(bk_ctx *c, uint32_t n, hipStream_t s)
{
    if (n <= c->cap_sort) return BK_OK;
    HIP_TRY(hipStreamSynchronize(s));
    for (auto &p : c->d_sort) { free_dev(p); p = nullptr; }
    free_dev(c->d_sort_tmp);
    c->d_sort_tmp = nullptr;
    c->cap_sort = 0;
    const uint64_t cap = (uint64_t)n + n / 4;
    for (auto &p : c->d_sort) HIP_TRY(dev_malloc(&p, cap * 4));
    size_t tb = 0;
    if (sort_list_by_key(nullptr, nullptr, nullptr, nullptr, (uint32_t)cap, nullptr, &tb, s)) return BK_ERR_INTERNAL;
    HIP_TRY(dev_malloc(&c->d_sort_tmp, tb));
    c->sort_tmp_bytes = tb;
    c->cap_sort = cap;
    return BK_OK;
}

// keys are expected in d_sort[0]; returns the reordered list (d_sort[2])
int sort_work(bk_ctx *c, const uint32_t *list, uint32_t n, hipStream_t s, const uint32_t **out)
{
    size_t tb = c->sort_tmp_bytes;
    if (sort_list_by_key(c->d_sort[0], c->d_sort[1], list, c->d_sort[2], n, c->d_sort_tmp, &tb, s)) return BK_ERR_INTERNAL;
    *out = c->d_sort[2];
    return BK_OK;
}

// Multi-loci modes: the loci lists of one chunk (reads whose AlignReads returned eHRhits own LowHitInstances
// entries each).  Counts -> offsets (scan) -> single loci copied from the result records, the others replayed
// by the ENUM form of the wave-per-read kernel; appended to the context's host vectors.
int collect_loci(bk_ctx *c, const DevBatch &b, uint32_t n, uint32_t maxlen, hipStream_t s)
{
    unsigned long long *d_cnt = nullptr, *d_offs = nullptr;
    void *d_tmp = nullptr;
    bk_loci *d_loci = nullptr;
    bk_loci_trims *d_trims = nullptr;
    const bool chim = c->params.min_chimeric_len > 0 && c->d_seg2 != nullptr;       // every locus carries its end trims
    int rc = BK_OK;
    auto cleanup = [&]() { free_dev(d_cnt); free_dev(d_offs); free_dev(d_tmp); free_dev(d_loci); free_dev(d_trims); };
#define LOCI_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return e_ == hipErrorOutOfMemory ? BK_ERR_MEM : BK_ERR_INTERNAL; } } while (0)
    LOCI_TRY(dev_malloc(&d_cnt, ((size_t)n + 1) * 8));
    LOCI_TRY(dev_malloc(&d_offs, ((size_t)n + 1) * 8));
    LOCI_TRY(hipMemsetAsync(d_cnt, 0, ((size_t)n + 1) * 8, s));
    launch_loci_count(b.out, n, c->params.clamp_ml ? c->cfg.max_hits : 0, d_cnt, s);
    size_t tb = 0;
    if (scan_counts_u64(nullptr, nullptr, n + 1, nullptr, &tb, s)) { cleanup(); return BK_ERR_INTERNAL; }
    LOCI_TRY(dev_malloc(&d_tmp, tb ? tb : 16));
    if (scan_counts_u64(d_cnt, d_offs, n + 1, d_tmp, &tb, s)) { cleanup(); return BK_ERR_INTERNAL; }
    const size_t base = c->loci_offs.empty() ? 0 : c->loci_offs.size() - 1;      // reads of earlier chunks
    const uint64_t loci_base = c->loci.size();
    if (c->loci_offs.empty()) c->loci_offs.push_back(0);
    c->loci_offs.resize(base + n + 1);
    LOCI_TRY(hipMemcpyAsync(c->loci_offs.data() + base, d_offs, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, s));
    LOCI_TRY(hipStreamSynchronize(s));
    const uint64_t total = c->loci_offs[base + n];
    if (total) {
        uint32_t *sm = c->d_small;
        LOCI_TRY(dev_malloc(&d_loci, (size_t)total * sizeof(bk_loci)));
        if (chim) {
            LOCI_TRY(dev_malloc(&d_trims, (size_t)total * sizeof(bk_loci_trims)));
            LOCI_TRY(clear_dev(d_trims, (size_t)total * sizeof(bk_loci_trims), s));
        }
        LOCI_TRY(hipMemsetAsync(sm, 0, 16 * 4, s));
        uint32_t *list = c->d_act[0];                  // the phase work lists are free by now
        launch_loci_single(b.out, n, d_offs, d_loci, list, sm + 0, chim ? c->d_seg2 : nullptr, d_trims, s);
        LOCI_TRY(hipMemcpyAsync(c->h_small, sm, 16 * 4, hipMemcpyDeviceToHost, s));
        LOCI_TRY(hipStreamSynchronize(s));
        const uint32_t n_multi = c->h_small[0];
        if (n_multi) {
            rc = size_heavy_scratch(c);
            if (rc) { cleanup(); return rc; }
            launch_loci_enum(c->ix, c->cfg, b, c->hs, list, n_multi, sm + 1, d_offs, d_loci, sm + 2, chim ? c->params.min_chimeric_len : 0, maxlen > 512 ? 1 : 0,
                             chim ? c->d_seg2 : nullptr, d_trims, s);
            LOCI_TRY(hipGetLastError());
            LOCI_TRY(hipMemcpyAsync(c->h_small, sm, 16 * 4, hipMemcpyDeviceToHost, s));
        }
        c->loci.resize(loci_base + total);
        LOCI_TRY(hipMemcpyAsync(c->loci.data() + loci_base, d_loci, (size_t)total * sizeof(bk_loci), hipMemcpyDeviceToHost, s));
        if (chim) {
            c->loci_trims.resize(loci_base + total);
            LOCI_TRY(hipMemcpyAsync(c->loci_trims.data() + loci_base, d_trims, (size_t)total * sizeof(bk_loci_trims), hipMemcpyDeviceToHost, s));
        }
        LOCI_TRY(hipStreamSynchronize(s));
        if (n_multi && c->h_small[2] != 0) {           // a replay that did not reproduce LowHitInstances: never ignore
            fprintf(stderr, "bk: loci replay disagreed with LowHitInstances for %u reads\n", c->h_small[2]);
            cleanup();
            return BK_ERR_INTERNAL;
        }
    }
    if (loci_base)
        for (size_t i = 0; i <= n; i++) c->loci_offs[base + i] += loci_base;
#undef LOCI_TRY
    cleanup();
    return BK_OK;
}

// -N (LocateBestMatches): one wave-per-read pass over the reads the N policy let through, dense rows of MaxHits
// loci per read compacted into the same host-side lists the other multi-loci modes return
int best_matches_chunk(bk_ctx *c, const DevBatch &b, uint32_t n, const uint32_t *d_list, uint32_t n_list, hipStream_t s)
{
    unsigned long long *d_cnt = nullptr, *d_offs = nullptr;
    void *d_tmp = nullptr;
    bk_loci *d_dense = nullptr, *d_loci = nullptr;
    const uint32_t width = (uint32_t)c->cfg.max_hits;
    auto cleanup = [&]() { free_dev(d_cnt); free_dev(d_offs); free_dev(d_tmp); free_dev(d_dense); free_dev(d_loci); };
#define BEST_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return e_ == hipErrorOutOfMemory ? BK_ERR_MEM : BK_ERR_INTERNAL; } } while (0)
    BEST_TRY(dev_malloc(&d_cnt, ((size_t)n + 1) * 8));
    BEST_TRY(dev_malloc(&d_offs, ((size_t)n + 1) * 8));
    BEST_TRY(dev_malloc(&d_dense, (size_t)n * width * sizeof(bk_loci)));
    BEST_TRY(hipMemsetAsync(d_cnt, 0, ((size_t)n + 1) * 8, s));
    int rc = size_heavy_scratch(c);
    if (rc) { cleanup(); return rc; }
    uint32_t *sm = c->d_small;
    BEST_TRY(hipMemsetAsync(sm + 4, 0, 4, s));
    launch_best(c->ix, c->cfg, b, c->hs, d_list, n_list, sm + 4, d_cnt, d_dense, s);
    BEST_TRY(hipGetLastError());
    size_t tb = 0;
    if (scan_counts_u64(nullptr, nullptr, n + 1, nullptr, &tb, s)) { cleanup(); return BK_ERR_INTERNAL; }
    BEST_TRY(dev_malloc(&d_tmp, tb ? tb : 16));
    if (scan_counts_u64(d_cnt, d_offs, n + 1, d_tmp, &tb, s)) { cleanup(); return BK_ERR_INTERNAL; }
    const size_t base = c->loci_offs.empty() ? 0 : c->loci_offs.size() - 1;
    const uint64_t loci_base = c->loci.size();
    if (c->loci_offs.empty()) c->loci_offs.push_back(0);
    c->loci_offs.resize(base + n + 1);
    BEST_TRY(hipMemcpyAsync(c->loci_offs.data() + base, d_offs, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, s));
    BEST_TRY(hipStreamSynchronize(s));
    const uint64_t total = c->loci_offs[base + n];
    if (total) {
        BEST_TRY(dev_malloc(&d_loci, (size_t)total * sizeof(bk_loci)));
        launch_loci_compact(d_dense, width, d_offs, n, d_loci, s);
        BEST_TRY(hipGetLastError());
        c->loci.resize(loci_base + total);
        BEST_TRY(hipMemcpyAsync(c->loci.data() + loci_base, d_loci, (size_t)total * sizeof(bk_loci), hipMemcpyDeviceToHost, s));
        BEST_TRY(hipStreamSynchronize(s));
    }
    if (loci_base)
        for (size_t i = 0; i <= n; i++) c->loci_offs[base + i] += loci_base;
#undef BEST_TRY
    cleanup();
    return BK_OK;
}

int align_chunk(bk_ctx *c, const DevReads &in, uint32_t first, uint32_t n, uint32_t maxlen, bk_hit *d_out, hipStream_t s, EvTimer &tm)
{
    const uint8_t *d_bases = in.bases;
    const uint64_t *d_offs = in.offs + first;
    const uint32_t *d_lens = in.lens + first;
    uint32_t *sm = c->d_small, *hm = c->h_small;
    HIP_TRY(hipMemsetAsync(sm, 0, 16 * 4, s));
    const uint32_t wpr = words_per_read(maxlen);
    // register-resident window kernels handle reads of <= 128 / <= 256 / <= 16 * kNwLong / <= 16 * kNwLongest bases
    const bool reg_path = c->use_wave && maxlen <= 16u * (uint32_t)kNwLongest;
    const int nw16 = maxlen <= 128 ? 8 : (maxlen <= 256 ? 16 : (maxlen <= 16u * (uint32_t)kNwLong ? kNwLong : kNwLongest));
    const bool two_bit = reg_path && c->ix.tgt2 != nullptr;
    const uint32_t ivc = iv_cores_for(c, maxlen);
    int rc = ensure_batch_scratch(c, n, wpr, two_bit ? (uint32_t)(nw16 / 2) : 0u, ivc);
    if (rc && c->d_swin) {                        // the window array is a luxury: it goes before a batch is refused for want of memory
        (void)hipGetLastError();
        bk::release_swin(c);
        rc = ensure_batch_scratch(c, n, wpr, two_bit ? (uint32_t)(nw16 / 2) : 0u, ivc);
    }
    if (rc) return rc;

    DevBatch b{};
    b.bases = d_bases; b.offs = d_offs; b.lens = d_lens;
    b.pk_words = in.words; b.pk_exc = in.exc; b.pk_nexc = in.words ? in.n_exc : 0; b.pk_read0 = first;
    b.rd4 = c->d_rd4; b.iv_first = c->d_iv_first; b.iv_n = c->d_iv_n; b.iv2 = c->d_iv2;
    b.rd2 = two_bit ? c->d_rd2 : nullptr;
    b.rmeta = c->d_rmeta;
    // (the wave list's job sizes come from k_flat only when every read on that list went through it)
    b.wave_work = (reg_path && c->cfg.heavy_thresh <= 100) ? c->d_wave_work : nullptr;
    b.iv32 = (c->use_iv32 && c->ix.k2) ? c->d_iv32 : nullptr;      // (written by k_search_a_ilp and pass B in phase 0)
    b.nw = reg_path ? (uint32_t)nw16 : 0u;       // the fused prep kernel packs reads of the register-kernel path
    b.out = d_out; b.seq_counts = c->d_seq_counts; b.ctr = c->d_ctr;
    b.wpr = wpr; b.n_reads = n; b.iv_cores = ivc;
    const int nstr = c->cfg.align_strand == 0 ? 2 : 1;

    // ---- the phase loop ---------------------------------------------------------------------------------------------------------
    // Every count the phases produce (active reads, pass B's work items, reads for the wave kernel) stays in device memory (PhaseCtl,
    // one line per phase) and the kernels size themselves by it.  On the main path - register-window kernels, k_flat, second-level
    // keys - the host therefore launches the whole schedule without reading anything back: grids come from bounds it knows (an active
    // list is never longer than the chunk; a read of up to maxlen bases has at most so many cores and phases), the two work-list sorts
    // are sized from what the previous chunk needed (items beyond that run unsorted: order never changes a result).  The other
    // configurations (general kernel family, lane-per-read kernels, no key array, -N, BK_DEBUG) keep reading the counts back, which
    // sizes their launches exactly.
    PhaseCtl *ctl = c->d_ctl;
    HIP_TRY(hipMemsetAsync(ctl, 0, sizeof(PhaseCtl) * (kMaxPhases + 2), s));
    auto P = [&](int ph) { return reinterpret_cast<uint32_t *>(ctl + ph); };      // words of ctl[ph]: [0] n_act [1] cmax [2] n_slist [3] n_wave [4] n_heavy [5] wave cursor [6] heavy cursor
    // bounds of a read of up to maxlen bases: phases, cores per strand in each
    int max_phases = 0, cmax_bound[kMaxPhases + 1] = {0};
    bool cores_fit = true;
    for (uint32_t len = 1; len <= maxlen; len++) {
        const ReadPlan p = make_plan((int)len, c->cfg);
        max_phases = std::max(max_phases, p.n_phases);
        for (int ph = 0; ph < p.n_phases && ph <= kMaxPhases; ph++) {
            int mm, cl, cd, dummy[1];
            phase_params(p, c->cfg, ph, mm, cl, cd);
            const int nc = core_offsets((int)len, cl, cd, p.max_slides, dummy, 0);
            if (nc > kMaxCoresFast) cores_fit = false;
            cmax_bound[ph] = std::max(cmax_bound[ph], std::min(nc, (int)kMaxCoresFast));
        }
    }
    if (max_phases > kMaxPhases) return BK_ERR_INTERNAL;
    const bool no_readback = reg_path && c->cfg.heavy_thresh <= 100 && c->ix.k2 != nullptr && cores_fit && !c->params.best_matches &&
                             !c->debug && c->async_phases;
    const bool check_maxlen = !tm.on;                 // (a call that only enqueues: its caller named the longest read, nobody has looked)
    if (check_maxlen) launch_max_len(d_lens, n, P(kMaxPhases + 1) + 0, s);
    hipEvent_t e0 = tm.begin(s);
    launch_prep(c->cfg, b, c->d_act[0], P(0) + 0, P(0) + 1, c->d_stage[0], c->d_stripe_cnt, s);
    HIP_TRY(hipGetLastError());
    tm.end(7, e0, s);
    uint32_t n_act = n;
    int cmax = cmax_bound[0];
    auto read_ctl = [&](int ph) -> int {           // ctl[ph], ctl[ph + 1] -> hm[0..31]
        HIP_TRY(hipMemcpyAsync(hm, ctl + ph, 2 * sizeof(PhaseCtl), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return BK_OK;
    };
    if (!no_readback) {
        int rr = read_ctl(0);
        if (rr) return rr;
        n_act = hm[0];
        cmax = (int)hm[1];
    }
    int cur = 0;
    if (c->params.best_matches) {
        hipEvent_t eb = tm.begin(s);
        int rb = best_matches_chunk(c, b, n, c->d_act[0], n_act, s);
        if (rb) return rb;
        tm.end(2, eb, s);
        n_act = 0;
    }
    b.iv_stride = n;                               // interval records and the wave list go by position in the phase's active list
    if (no_readback && c->ix.isa == nullptr) { int rh = size_heavy_scratch(c); if (rh) return rh; }      // hash-set dedupe of the wave kernel
    for (int phase = 0; no_readback ? phase < max_phases : n_act > 0; phase++) {
        const uint32_t *ext_list = c->d_act[cur];
        b.act = ext_list;
        if (no_readback) cmax = cmax_bound[phase];
        const uint32_t n_bound = n_act;            // (no read-back: the chunk's size; else the list's length)
        uint32_t *ctl_p = P(phase), *ctl_n = P(phase + 1);
        const int lazy = (reg_path && c->lazy_search) ? 1 : 0;
        if (cmax > 0) {
            hipEvent_t e1 = tm.begin(s);
            if (c->ix.k2) {
                const uint64_t lanes = (uint64_t)n_bound * (uint64_t)(cmax * nstr);
                if (lanes > c->cap_slist) {
                    HIP_TRY(hipStreamSynchronize(s));
                    free_dev(c->d_slist);
                    free_dev(c->d_slist_stage);
                    c->d_slist = c->d_slist_stage = nullptr;
                    c->cap_slist = 0;
                    HIP_TRY(dev_malloc(&c->d_slist, lanes * 4));
                    HIP_TRY(dev_malloc(&c->d_slist_stage, (lanes + (kListStripes + 2) * 1024) * 4));     // its striped form (StripeSet)
                    c->cap_slist = lanes;
                }
                // interval counts of the slots this phase can use, zeroed (empty search results store nothing)
                launch_clear_iv(b, ctl_p + 0, n_bound, cmax, c->cfg.align_strand == 2 ? 1 : 0, c->cfg.align_strand == 1 ? 0 : 1, s);
                hipEvent_t ea = tm.begin(s);
                launch_search_a(c->ix, c->cfg, b, c->d_act[cur], ctl_p + 0, n_bound, phase, cmax, nstr, lazy, c->d_slist, ctl_p + 2,
                                c->d_slist_stage, c->d_stripe_cnt, s);
                HIP_TRY(hipGetLastError());
                tm.end(4, ea, s);
                // pass B's work list, grouped by k-mer bucket: the sort's size is the list's length when that was read back, else what
                // the previous chunk's phase needed (plus a margin; capped at the sort buffers)
                uint64_t n_slist_bound = lanes;
                uint32_t n_sort = 0;
                if (!no_readback) {
                    HIP_TRY(hipMemcpyAsync(hm + 2, ctl_p + 2, 4, hipMemcpyDeviceToHost, s));
                    HIP_TRY(hipStreamSynchronize(s));
                    n_slist_bound = hm[2];
                    n_sort = hm[2];
                } else {
                    const double f = c->hist_valid ? c->hist_slist[phase] * 1.05 : 0.30;
                    n_sort = (uint32_t)std::min<uint64_t>({(uint64_t)(f * n) + 4096, lanes, (uint64_t)0x7FFFFFF0});
                }
                if (no_readback && c->cap_sort >= 4096) n_sort = (uint32_t)std::min<uint64_t>(n_sort, c->cap_sort);      // (no growing - it waits for the stream - for a guess)
                const uint32_t *sorted = nullptr;
                if ((c->sort_lists & 1) && n_sort >= 4096) {
                    int rs = ensure_sort_scratch(c, n_sort, s);
                    if (rs) return rs;
                    hipEvent_t es = tm.begin(s);
                    launch_keys_search(b, c->d_slist, ctl_p + 2, n_sort, c->sort_shift, c->d_sort[0], s);
                    rs = sort_work(c, c->d_slist, n_sort, s, &sorted);
                    if (rs) return rs;
                    tm.end(5, es, s);
                } else
                    n_sort = 0;
                if (n_slist_bound) {
                    hipEvent_t eb = tm.begin(s);
                    launch_search_b(c->ix, c->cfg, b, phase, lazy, c->d_slist, sorted, n_sort, ctl_p + 2, n_slist_bound, s);
                    tm.end(6, eb, s);
                }
            } else
                launch_search(c->ix, c->cfg, b, c->d_act[cur], n_act, phase, cmax, nstr, lazy, s);
            HIP_TRY(hipGetLastError());
            tm.end(0, e1, s);
        }
        if (phase == c->dbg_stop_phase) {
            // (test hook, bk_debug_intervals: the interval records the search of this phase wrote stay where they are; nothing of the
            // phase's extension or of the phases behind it runs, so the batch's result records are not to be used)
            c->dbg_n = n; c->dbg_ivc = ivc; c->dbg_cur = cur; c->dbg_phase = phase; c->dbg_valid = true;
            break;
        }
        hipEvent_t e2 = tm.begin(s);
        // 5-byte indexes: the reference's seen-target set is keyed by the target start truncated to 32 bits (SfxArrayV2.cpp:5932), a
        // rule that depends on every earlier candidate of the strand pass.  k_flat and the hash-set kernels reproduce it; the
        // lane-per-read kernels dedupe on exact starts, so on an index of more than 2^32 bases they only finish the calls without candidates and hand
        // every other one to the hash-set kernels.
        DevAlignCfg cfg_lane = c->cfg;
        if (c->ix.n > (1ULL << 32)) cfg_lane.heavy_thresh = 0;            // (below that the truncated keys are the exact ones)
        if (n_bound == 0) {}
        else if (reg_path && c->cfg.heavy_thresh <= 100)
            launch_flat(c->ix, c->cfg, b, ext_list, ctl_p + 0, n_bound, phase, nstr * std::max(cmax, 1), c->d_act[cur ^ 1], ctl_n + 0, c->d_heavy, ctl_p + 4,
                        c->d_wave, ctl_p + 3, ctl_n + 1, c->d_stage, c->d_stripe_cnt, nw16, s);
        else
            launch_extend(c->ix, cfg_lane, b, ext_list, n_act, phase, c->d_act[cur ^ 1], ctl_n + 0, c->d_heavy, ctl_p + 4, ctl_n + 1, s);
        HIP_TRY(hipGetLastError());
        tm.end(1, e2, s);
        uint32_t n_heavy = 0, n_wave = n_bound;    // (no read-back: any read of the list may have gone to the wave kernel)
        if (!no_readback) {
            int rr = read_ctl(phase);
            if (rr) return rr;
            n_heavy = hm[4];
            n_wave = hm[3];
        }
        if (n_wave) {
            hipEvent_t e3 = tm.begin(s);
            const uint32_t *wsorted = nullptr;
            uint32_t n_sort = n_wave;
            if (no_readback) {
                const double f = c->hist_valid ? c->hist_wave[phase] * 1.05 : 0.25;
                n_sort = (uint32_t)std::min<uint64_t>((uint64_t)(f * n) + 4096, n_wave);
                if (c->cap_sort >= 4096) n_sort = (uint32_t)std::min<uint64_t>(n_sort, c->cap_sort);
            }
            if ((c->sort_lists & 2) && n_sort >= 4096) {
                int rs = ensure_sort_scratch(c, n_sort, s);
                if (rs) return rs;
                launch_keys_wave(c->cfg, b, phase, c->d_wave, ctl_p + 3, n_sort, (c->sort_lists & 4) ? -1 : c->sort_shift, c->d_sort[0], b.wave_work, s);
                rs = sort_work(c, c->d_wave, n_sort, s, &wsorted);
                if (rs) return rs;
            } else
                n_sort = 0;
            if (!no_readback && c->ix.isa == nullptr) { int rh = size_heavy_scratch(c); if (rh) return rh; }      // hash-set dedupe
            launch_wave(c->ix, c->cfg, b, c->hs, c->d_wave, wsorted, n_sort, ctl_p + 3, n_wave, phase, ctl_p + 5, c->d_act[cur ^ 1], ctl_n + 0, ctl_n + 1,
                        nw16, c->wave_waves, s);
            HIP_TRY(hipGetLastError());
            tm.end(2, e3, s);
        }
        if (n_heavy) {
            hipEvent_t e3 = tm.begin(s);
            { int rh = size_heavy_scratch(c); if (rh) return rh; }
            launch_heavy(c->ix, c->cfg, b, c->hs, c->d_heavy, n_heavy, phase, ctl_p + 6, c->d_act[cur ^ 1], ctl_n + 0, ctl_n + 1, s);
            HIP_TRY(hipGetLastError());
            tm.end(2, e3, s);
        }
        if (!no_readback) {
            if (n_wave || n_heavy) { int rr = read_ctl(phase); if (rr) return rr; }
            if (c->debug)
                fprintf(stderr, "bk: phase %d n_act %u cmax %d n_heavy %u n_wave %u -> next n_act %u cmax %u\n", phase, n_act, cmax, n_heavy, n_wave, hm[16], hm[17]);
            n_act = hm[16];                        // ctl[phase + 1].n_act, .cmax
            cmax = (int)hm[17];
        }
        cur ^= 1;
        if (phase > 70) return BK_ERR_INTERNAL;
    }
    if (c->dbg_stop_phase >= 0 && !c->dbg_valid) {       // (no read reached that phase: an empty list - ctl[kMaxPhases] is never written)
        c->dbg_n = n; c->dbg_ivc = ivc; c->dbg_cur = cur; c->dbg_phase = kMaxPhases; c->dbg_valid = true;
    }
    if (no_readback) {
        // what the phases needed goes to the host on its own time: it sizes the next chunk's sorts (and says whether a read was handed to the
        // general kernel, which this schedule never launches: the core bound above rules it out)
        HIP_TRY(hipMemcpyAsync(c->h_ctl, ctl, sizeof(PhaseCtl) * (kMaxPhases + 2), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(c->ev_ctl, s));
        c->ctl_pending = true;
        c->ctl_pending_reads = n;
        c->ctl_pending_phases = max_phases;
        c->ctl_pending_maxlen = check_maxlen ? maxlen : 0;
    }
    if (c->params.micro_indel_len > 0 || c->params.splice_junct_len > 0 || c->params.min_chimeric_len > 0) {
        // AlignReads' branches for what is still unaligned (SfxArrayV2.cpp:7722-7757): microInDels, then splice junctions, then the
        // chimeric (end-trimmed) placement
        hipEvent_t ei = tm.begin(s);
        if (n > c->cap_seg2) {                                     // kept with the batch scratch, grown on demand
            free_dev(c->d_seg2);
            c->d_seg2 = nullptr;
            c->cap_seg2 = 0;
            HIP_TRY(dev_malloc(&c->d_seg2, (size_t)n * sizeof(bk_seg2)));
            c->cap_seg2 = n;
        }
        bk_seg2 *d_seg2 = c->d_seg2;
        hipError_t eh = clear_dev(d_seg2, (size_t)n * sizeof(bk_seg2), s);
        int rc2 = BK_OK;
        if (eh == hipSuccess && (c->params.micro_indel_len > 0 || c->params.splice_junct_len > 0)) {
            eh = hipMemsetAsync(sm, 0, 16 * 4, s);
            if (eh == hipSuccess) {
                launch_indel(c->ix, c->cfg, b, n, c->params.micro_indel_len, c->params.splice_junct_len, c->params.min_chimeric_len > 0 ? 1 : 0, c->d_act[0], sm + 0,
                             hm + 0, sm + 1, d_seg2, s);
                eh = hipGetLastError();
            }
        }
        if (eh == hipSuccess && c->params.min_chimeric_len > 0) {
            rc2 = size_heavy_scratch(c);
            if (rc2 == BK_OK) {
                eh = hipMemsetAsync(sm, 0, 16 * 4, s);
                if (eh == hipSuccess) {
                    launch_unaligned_list(b.out, n, c->d_act[0], sm + 0, s);
                    eh = hipMemcpyAsync(hm, sm, 4, hipMemcpyDeviceToHost, s);
                }
                if (eh == hipSuccess) eh = hipStreamSynchronize(s);
                if (eh == hipSuccess && hm[0]) {
                    launch_chimeric(c->ix, c->cfg, b, c->hs, c->d_act[0], hm[0], c->params.min_chimeric_len, maxlen > 512 ? 1 : 0, sm + 1, d_seg2, s);
                    eh = hipGetLastError();
                }
            }
        }
        const size_t at = c->seg2.size();
        c->seg2.resize(at + n);
        if (eh == hipSuccess && rc2 == BK_OK) eh = hipMemcpyAsync(c->seg2.data() + at, d_seg2, (size_t)n * sizeof(bk_seg2), hipMemcpyDeviceToHost, s);
        if (eh == hipSuccess) eh = hipStreamSynchronize(s);
        if (rc2 != BK_OK) return rc2;
        if (eh != hipSuccess) return BK_ERR_INTERNAL;
        if (c->params.min_chimeric_len > 0 && c->cfg.max_hits > 1)          // (the chimeric call's note to the loci replay, see k_heavy)
            for (size_t i = at; i < at + n; i++)
                if (c->seg2[i].flags == 0x40) c->seg2[i] = bk_seg2{};
        tm.end(2, ei, s);
    }
    hipEvent_t e4 = tm.begin(s);
    launch_count_seqs(d_out, n, c->d_id2idx, c->ix.n_ent, c->d_seq_counts, s);
    HIP_TRY(hipGetLastError());
    tm.end(3, e4, s);
    if (c->cfg.max_hits > 1 && !c->params.best_matches) {
        int rl = collect_loci(c, b, n, maxlen, s);
        if (rl) return rl;
    }
    return BK_OK;
}


// What the last chunk that ran without read-backs needed, once its counts have arrived in h_ctl (the copy was enqueued behind its
// kernels): work items of pass B and reads for the wave kernel per phase, as fractions of the chunk - the sizes of the next chunk's
// sorts.  wait: block until they are there.  A read on a general-kernel list would mean the core bound of align_chunk was wrong.
int take_phase_history(bk_ctx *c, bool wait)
{
    if (!c->ctl_pending) return BK_OK;
    if (wait) HIP_TRY(bk::wait_event(c->ev_ctl));
    else if (hipEventQuery(c->ev_ctl) != hipSuccess) { (void)hipGetLastError(); return BK_OK; }
    c->ctl_pending = false;
    const PhaseCtl *h = c->h_ctl;
    const double n = (double)std::max<uint32_t>(c->ctl_pending_reads, 1);
    for (int ph = 0; ph < kMaxPhases; ph++) {
        c->hist_slist[ph] = ph < c->ctl_pending_phases ? (double)h[ph].n_slist / n : 0.0;
        c->hist_wave[ph] = ph < c->ctl_pending_phases ? (double)h[ph].n_wave / n : 0.0;
        if (h[ph].n_heavy) {
            fprintf(stderr, "biokanga_amd: %u reads of phase %d were left for the general kernel by a schedule that does not run it\n", h[ph].n_heavy, ph);
            c->async_error = BK_ERR_INTERNAL;
        }
    }
    // (bk_align_batch_device_async: the longest read its caller promised against the longest the batch really held)
    if (c->ctl_pending_maxlen && h[kMaxPhases + 1].n_act > c->ctl_pending_maxlen) {
        fprintf(stderr, "biokanga_amd: a batch held a read of %u bases, its caller had promised at most %u\n", h[kMaxPhases + 1].n_act, c->ctl_pending_maxlen);
        c->async_error = BK_ERR_PARAMS;
    }
    c->hist_valid = true;
    if (c->async_error && wait) { const int e = c->async_error; c->async_error = 0; return e; }
    return BK_OK;
}

// enqueue_only: everything is launched on `s` and the call returns without waiting for any of it (bk_align_batch_device_async): the
// caller has named the longest read, the scratch is in place (bk_ctx_reserve) and the configuration is one whose phase loop reads
// nothing back - else BK_ERR_PARAMS, before anything is launched.
int align_device(bk_ctx *c, const DevReads &in, uint32_t nreads, bk_hit *d_out, hipStream_t s, uint32_t maxlen_known, bool enqueue_only)
{
    grow_tick(c, nreads);                      // (BK_CTX_GROW_IMAGE: the long-run tables are started, or taken in, between batches)
    const uint32_t *d_lens = in.lens;
    if (enqueue_only) {
        const uint32_t ml = maxlen_known;
        const bool plain = c->cfg.max_hits == 1 && !c->params.best_matches && !c->params.micro_indel_len && !c->params.splice_junct_len && !c->params.min_chimeric_len;
        const bool fits = ml >= 1 && ml <= 16u * (uint32_t)kNwLongest && nreads <= c->cap_reads && nreads <= c->chunk_reads && words_per_read(ml) <= c->cap_wpr &&
                          iv_cores_for(c, ml) <= c->cap_iv_cores && rd2w_for(ml) <= c->cap_rd2w &&
                          (uint64_t)nreads * iv_cores_for(c, ml) * (c->cfg.align_strand == 0 ? 2u : 1u) <= c->cap_slist && (c->ix.isa != nullptr || c->hs.htab != nullptr);
        const bool main_path = c->use_wave && c->cfg.heavy_thresh <= 100 && c->ix.k2 != nullptr && c->ix.tgt2 != nullptr && !c->debug && c->async_phases;
        if (!plain || !fits || !main_path) return BK_ERR_PARAMS;
    }
    EvTimer tm{c};
    tm.on = !enqueue_only;
    hipEvent_t t0 = tm.begin(s);
    c->loci_offs.clear();
    c->loci.clear();
    c->loci_trims.clear();
    c->seg2.clear();
    // longest read of the call -> row width of the packed reads and the kernel family used (the pipeline knows it already)
    uint32_t maxlen = maxlen_known;
    if (!maxlen) {
        HIP_TRY(hipMemsetAsync(c->d_small, 0, 16 * 4, s));
        launch_max_len(d_lens, nreads, c->d_small + 5, s);
        HIP_TRY(hipMemcpyAsync(c->h_small, c->d_small, 16 * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        maxlen = c->h_small[5];
    }
    if (maxlen > (uint32_t)kMaxReadLenAbs) return BK_ERR_PARAMS;
    c->last_maxlen = maxlen;
    (void)take_phase_history(c, false);
    if ((int)maxlen > c->max_read_len) {
        c->max_read_len = (int)maxlen;
    }
    if (!enqueue_only) { int rw = maybe_build_swin(c, maxlen, nreads, s); if (rw) return rw; }      // (the window array is used when it is there)
    // chunk size: as many reads as the knob allows and as fit in about half of the HBM still free
    // (the phase kernels run better the more reads they see: fewer launches, shorter tails)
    uint32_t chunk = c->chunk_reads;
    // (a batch the scratch already holds needs no look at the free memory)
    if (!(std::min(chunk, nreads) <= c->cap_reads && words_per_read(maxlen) <= c->cap_wpr && iv_cores_for(c, maxlen) <= c->cap_iv_cores &&
          rd2w_for(maxlen) <= std::max(c->cap_rd2w, 1u) && !c->params.best_matches)) {
        const uint64_t per_read = scratch_bytes_per_read(words_per_read(maxlen), rd2w_for(maxlen), iv_cores_for(c, maxlen));
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const uint64_t have = (uint64_t)c->cap_reads * scratch_bytes_per_read(c->cap_wpr, c->cap_rd2w, c->cap_iv_cores);
        uint64_t fit = ((uint64_t)free_b + have) / 4 * 3 / per_read;
        if (fit < 65536) fit = 65536;
        if (fit < chunk) chunk = (uint32_t)fit;
        if (chunk > (1u << 27)) chunk = 1u << 27;       // 32 interval slots per read are indexed with 32 bits
        if (c->params.best_matches) {                   // dense rows of MaxHits loci per read: keep them within 4 GB
            const uint64_t lim = std::max<uint64_t>(1024, (4ULL << 30) / (sizeof(bk_loci) * (uint64_t)c->cfg.max_hits));
            if (lim < chunk) chunk = (uint32_t)lim;
        }
    }
    for (uint32_t done = 0; done < nreads;) {
        uint32_t n = std::min(chunk, nreads - done);
        int rc = align_chunk(c, in, done, n, maxlen, d_out + done, s, tm);
        if (rc) return rc;
        done += n;
    }
    if (enqueue_only) return BK_OK;          // (the chunk's counts reach the host behind its kernels: take_phase_history of the next call)
    hipEvent_t t1 = tm.begin(s);
    HIP_TRY(bk::wait_stream(s, c->ev_wait));             // (asleep: the caller's thread leaves its CPU to others meanwhile)
    { int rh = take_phase_history(c, true); if (rh) return rh; }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, t0, t1);
    c->timing.ms_total += ms;
    for (auto &sp : tm.spans) {
        float m = 0;
        (void)hipEventElapsedTime(&m, sp.second.first, sp.second.second);
        switch (sp.first) {
        case 0: c->timing.ms_search += m; c->timing.n_search_launches++; break;
        case 1: c->timing.ms_extend += m; c->timing.n_extend_launches++; break;
        case 2: c->timing.ms_heavy += m; c->timing.n_heavy_launches++; break;
        case 4: c->timing.ms_search_a += m; break;           // (inside a span of kind 0)
        case 5: c->timing.ms_search_sort += m; break;
        case 6: c->timing.ms_search_b += m; c->timing.n_search_b_launches++; break;
        case 7: c->timing.ms_prep += m; c->timing.ms_other += m; break;
        default: c->timing.ms_other += m; break;
        }
    }
    return BK_OK;
}


}  // namespace bk

using namespace bk;

int bk::engine_align_device(bk_ctx *c, const DevReads &in, uint32_t nreads, bk_hit *d_out, hipStream_t s, uint32_t maxlen_known)
{
    return align_device(c, in, nreads, d_out, s, maxlen_known);
}

namespace {
struct CastU64 {
    __host__ __device__ unsigned long long operator()(const uint32_t &v) const { return (unsigned long long)v; }
};
}

int bk::engine_prepare_packed(bk_ctx *c, const uint16_t *d_lens16, uint32_t nreads, uint64_t n_words, const bk_nbase *d_exc, uint64_t n_exc,
                              uint32_t *d_lens32, uint64_t *d_offs, uint32_t *maxlen, hipStream_t s)
{
    // words per read -> exclusive scan in place = first word of every read
    launch_widen_lens(d_lens16, nreads, d_lens32, (unsigned long long *)d_offs, s);
    HIP_TRY(hipGetLastError());
    size_t need = 0;
    HIP_TRY(bk::prim::exclusive_sum(nullptr, need, (unsigned long long *)d_offs, (unsigned long long *)d_offs, (size_t)nreads, s));
    if (need > c->scan_tmp_bytes) {
        HIP_TRY(hipStreamSynchronize(s));
        free_dev(c->d_scan_tmp);
        c->d_scan_tmp = nullptr;
        c->scan_tmp_bytes = 0;
        HIP_TRY(dev_malloc(&c->d_scan_tmp, need + 256));
        c->scan_tmp_bytes = need + 256;
    }
    size_t tb = c->scan_tmp_bytes;
    HIP_TRY(bk::prim::exclusive_sum(c->d_scan_tmp, tb, (unsigned long long *)d_offs, (unsigned long long *)d_offs, (size_t)nreads, s));
    // [0] max over reads of (first word + words) = the batch's word count, [1] longest read; exceptions in range and ascending
    HIP_TRY(hipMemsetAsync(c->d_ctr_aux, 0, 32, s));
    launch_packed_extent(d_offs, d_lens32, nreads, c->d_ctr_aux, s);
    launch_check_exc(d_exc, n_exc, d_lens32, nreads, reinterpret_cast<uint32_t *>(c->d_ctr_aux + 2), s);
    HIP_TRY(hipGetLastError());
    unsigned long long h[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(h, c->d_ctr_aux, 24, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (h[0] != n_words || h[1] > (unsigned long long)kMaxReadLenAbs || (uint32_t)h[2] != 0) return BK_ERR_PARAMS;
    *maxlen = (uint32_t)h[1];
    return BK_OK;
}

// ------------------------------------------------------------------------------------------------
// K4 on buffers resident in HBM (shared by bk_pair_batch, bk_pair_batch_device and the stream pipeline).  seg2_host / seg2_dev: the
// bk_seg2 records of these reads (one of the two, or neither) - required when the context trims chimeric reads (-c), whose pair rules
// look at the trimmed loci and whose orphan recovery may place the partner end-trimmed; updated in place
int bk::engine_pair_device(bk_ctx *c, const DevReads &in, uint32_t n_pairs, bk_hit *d_hits, uint32_t maxlen, const bk_pe_params *pe, hipStream_t s,
                           bk_seg2 *seg2_host, bk_seg2 *seg2_dev)
{
    const uint8_t *d_bases = in.bases;
    const uint64_t *d_offs = in.offs;
    const uint32_t *d_lens = in.lens;
    const uint32_t nreads = 2 * n_pairs;
    const uint32_t wpr = words_per_read(maxlen);
    // (launch_pe never touches the interval records: the slots keep whatever core count the SE pass sized them for)
    const uint32_t ivc = c->cap_iv_cores ? c->cap_iv_cores : iv_cores_for(c, maxlen);
    int rc = ensure_batch_scratch(c, nreads, wpr, 0, ivc);
    if (rc && c->d_swin) {                        // the window array goes before a batch is refused for want of memory (as in align_chunk)
        (void)hipGetLastError();
        bk::release_swin(c);
        rc = ensure_batch_scratch(c, nreads, wpr, 0, ivc);
    }
    if (rc) return rc;
    DevBatch b{};
    b.bases = d_bases; b.offs = d_offs; b.lens = d_lens;
    b.pk_words = in.words; b.pk_exc = in.exc; b.pk_nexc = in.words ? in.n_exc : 0; b.pk_read0 = 0;
    b.rd4 = c->d_rd4; b.iv_first = c->d_iv_first; b.iv_n = c->d_iv_n; b.iv2 = c->d_iv2;
    b.rmeta = c->d_rmeta;
    b.out = d_hits; b.seq_counts = c->d_seq_counts; b.ctr = c->d_ctr;
    b.wpr = wpr; b.n_reads = nreads; b.iv_cores = c->cap_iv_cores ? c->cap_iv_cores : kMaxCoresFast;
    if (c->params.min_chimeric_len > 0 && !seg2_host && !seg2_dev) return BK_ERR_PARAMS;
    bk_seg2 *d_seg2 = seg2_dev;
    if (!d_seg2 && seg2_host) {
        if (nreads > c->cap_seg2) {
            free_dev(c->d_seg2);
            c->d_seg2 = nullptr;
            c->cap_seg2 = 0;
            HIP_TRY(dev_malloc(&c->d_seg2, (size_t)nreads * sizeof(bk_seg2)));
            c->cap_seg2 = nreads;
        }
        d_seg2 = c->d_seg2;
        HIP_TRY(hipMemcpyAsync(d_seg2, seg2_host, (size_t)nreads * sizeof(bk_seg2), hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemsetAsync(c->d_small, 0, 16 * 4, s));
    launch_pe(c->ix, c->cfg, b, pe->pe_mode, pe->pair_min_len, pe->pair_max_len, pe->pair_strand ? 1 : 0, d_hits, n_pairs,
              c->d_heavy, c->d_small, c->h_small, d_seg2, c->params.min_chimeric_len, maxlen > 512 ? 1 : 0, c->d_chrom_accept, c->n_chrom_accept, s);
    HIP_TRY(hipGetLastError());
    if (seg2_host && !seg2_dev) HIP_TRY(hipMemcpyAsync(seg2_host, d_seg2, (size_t)nreads * sizeof(bk_seg2), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return BK_OK;
}


extern "C" {


int bk_batch_seg2(bk_ctx *c, const bk_seg2 **seg2, uint64_t *n)
{
    if (!c || !seg2 || !n) return BK_ERR_PARAMS;
    *seg2 = c->seg2.empty() ? nullptr : c->seg2.data();
    *n = c->seg2.size();
    return BK_OK;
}

int bk_batch_loci(bk_ctx *c, const uint64_t **offs, const bk_loci **loci, uint64_t *n_loci)
{
    if (!c || !offs || !loci || !n_loci) return BK_ERR_PARAMS;
    if (c->loci_offs.empty()) { *offs = nullptr; *loci = nullptr; *n_loci = 0; return BK_OK; }
    *offs = c->loci_offs.data();
    *loci = c->loci.data();
    *n_loci = c->loci.size();
    return BK_OK;
}

int bk_batch_loci_trims(bk_ctx *c, const bk_loci_trims **trims, uint64_t *n_loci)
{
    if (!c || !trims || !n_loci) return BK_ERR_PARAMS;
    *trims = c->loci_trims.empty() ? nullptr : c->loci_trims.data();
    *n_loci = c->loci_trims.size();
    return BK_OK;
}


const char *bk_version(void) { return "biokanga_amd 0.1 (gfx950; reference biokanga 4.4.2)"; }

const char *bk_strerror(int rc)
{
    switch (rc) {
    case BK_OK: return "success";
    case BK_ERR_INTERNAL: return "internal processing error (HIP failure or inconsistency)";
    case BK_ERR_NODEVICE: return "no usable HIP device - this library has no CPU fallback";
    case BK_ERR_PARAMS: return "parameter error";
    case BK_ERR_MEM: return "unable to allocate memory";
    case BK_ERR_NOTBIOSEQ: return "file exists but is not a biokanga suffix array file";
    case BK_ERR_OPNFILE: return "unable to open file";
    case BK_ERR_CREATEFILE: return "unable to create file";
    case BK_ERR_FILEVER: return "file version error";
    case BK_ERR_FILEACCESS: return "file access (seek/read/write) failed";
    default: return "error";
    }
}

int bk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}


int bk_align_batch_device(bk_ctx *c, const void *d_bases, const void *d_offs, const void *d_lens, uint32_t nreads,
                          void *d_out, void *stream, int sync)
{
    (void)sync;   // the phase loop reads active counts back, so the call always completes before returning
    if (!c || (nreads && (!d_bases || !d_offs || !d_lens || !d_out))) return BK_ERR_PARAMS;
    if (!nreads) return BK_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    DevReads in;
    in.bases = (const uint8_t *)d_bases; in.offs = (const uint64_t *)d_offs; in.lens = (const uint32_t *)d_lens;
    return align_device(c, in, nreads, (bk_hit *)d_out, s);
}

int bk_align_batch_device_async(bk_ctx *c, const void *d_bases, const void *d_offs, const void *d_lens, uint32_t nreads, uint32_t max_read_len,
                                void *d_out, void *stream)
{
    if (!c || !max_read_len || (nreads && (!d_bases || !d_offs || !d_lens || !d_out))) return BK_ERR_PARAMS;
    if (!nreads) return BK_OK;
    HIP_TRY(hipSetDevice(c->device));
    { int rh = take_phase_history(c, false); if (rh) return rh; }       // (also: what an earlier call of this kind found wrong with its batch)
    if (c->async_error) { const int e = c->async_error; c->async_error = 0; return e; }
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    DevReads in;
    in.bases = (const uint8_t *)d_bases; in.offs = (const uint64_t *)d_offs; in.lens = (const uint32_t *)d_lens;
    return align_device(c, in, nreads, (bk_hit *)d_out, s, max_read_len, true);
}

int bk_align_batch(bk_ctx *c, const uint8_t *bases, const uint64_t *offs, const uint32_t *lens, uint32_t nreads, bk_hit *out)
{
    if (!c || (nreads && (!bases || !offs || !lens || !out))) return BK_ERR_PARAMS;
    if (!nreads) return BK_OK;
    HIP_TRY(hipSetDevice(c->device));
    // the reads need not be contiguous in `bases`: find the extent referenced
    uint64_t lo = ~0ULL, hi = 0;
    for (uint32_t i = 0; i < nreads; i++) {
        if (lens[i] > (uint32_t)kMaxReadLenAbs) return BK_ERR_PARAMS;
        lo = std::min(lo, offs[i]);
        hi = std::max(hi, offs[i] + lens[i]);
    }
    uint64_t nbytes = hi - lo;
    if (nbytes + 16 > c->cap_in_bases) {
        free_dev(c->d_in_bases);
        c->d_in_bases = nullptr;
        c->cap_in_bases = 0;
        HIP_TRY(dev_malloc(&c->d_in_bases, nbytes + 16));
        c->cap_in_bases = nbytes + 16;
    }
    if (nreads > c->cap_in_reads) {
        free_dev(c->d_in_offs); free_dev(c->d_in_lens); free_dev(c->d_in_out);
        c->d_in_offs = nullptr; c->d_in_lens = nullptr; c->d_in_out = nullptr;
        c->cap_in_reads = 0;
        HIP_TRY(dev_malloc(&c->d_in_offs, (size_t)nreads * 8));
        HIP_TRY(dev_malloc(&c->d_in_lens, (size_t)nreads * 4));
        HIP_TRY(dev_malloc(&c->d_in_out, (size_t)nreads * sizeof(bk_hit)));
        c->cap_in_reads = nreads;
    }
    std::vector<uint64_t> rel(nreads);
    for (uint32_t i = 0; i < nreads; i++) rel[i] = offs[i] - lo;
    HIP_TRY(hipMemcpyAsync(c->d_in_bases, bases + lo, nbytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_in_offs, rel.data(), (size_t)nreads * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_in_lens, lens, (size_t)nreads * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    DevReads in;
    in.bases = c->d_in_bases; in.offs = c->d_in_offs; in.lens = c->d_in_lens;
    int rc = align_device(c, in, nreads, c->d_in_out, c->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, c->d_in_out, (size_t)nreads * sizeof(bk_hit), hipMemcpyDeviceToHost));
    return BK_OK;
}

uint64_t bk_packed_words(const uint32_t *lens, uint32_t nreads)
{
    uint64_t n = 0;
    if (lens) for (uint32_t i = 0; i < nreads; i++) n += ((uint64_t)lens[i] + 15) >> 4;
    return n;
}

namespace {
// 16 bases (bytes, bits 0-2 = the code) -> one packed word, when none of them is above 3: the two low bits of every byte gathered with the
// bit-extract instruction (first base into the top bits: the bytes are swapped first).  *clean = every code was 0..3.
__attribute__((target("bmi2"))) inline uint32_t pack16_bmi2(const uint8_t *s, bool *clean)
{
    uint64_t a, b;
    memcpy(&a, s, 8);
    memcpy(&b, s + 8, 8);
    *clean = ((a | b) & 0x0404040404040404ULL) == 0;
    const uint64_t m = 0x0303030303030303ULL;
    return (uint32_t)(__builtin_ia32_pext_di(__builtin_bswap64(a), m) << 16) | (uint32_t)__builtin_ia32_pext_di(__builtin_bswap64(b), m);
}
inline uint32_t pack16_plain(const uint8_t *s, bool *clean)
{
    uint32_t v = 0, bad = 0;
    for (uint32_t k = 0; k < 16; k++) { const uint32_t code = s[k] & 7u; bad |= code & 4u; v |= (code & 3u) << (30 - 2 * k); }
    *clean = bad == 0;
    return v;
}
}  // namespace

int bk_pack_reads(const uint8_t *bases, const uint64_t *offs, const uint32_t *lens, uint32_t nreads, uint32_t *words, uint16_t *lens16,
                  bk_nbase *exc, uint64_t exc_cap, uint64_t *n_exc)
{
    static const bool have_bmi2 = __builtin_cpu_supports("bmi2");
    if (!n_exc || (nreads && (!bases || !lens || !words || !lens16)) || (exc_cap && !exc)) return BK_ERR_PARAMS;
    *n_exc = 0;
    if (!nreads) return BK_OK;
    // slices of reads, one thread each: first word and first base of every slice, then pack; exceptions are collected per slice
    // and laid behind each other afterwards (ascending by read and position as the slices are)
    unsigned nt = (unsigned)std::max(1, std::min(16, bk::effective_cpus()));         // (affinity mask and cgroup quota, not the host's hardware threads)
    if (nreads < 65536) nt = 1;
    std::vector<uint64_t> w0(nt + 1, 0), b0(nt + 1, 0);
    std::vector<uint32_t> r0(nt + 1);
    for (unsigned t = 0; t <= nt; t++) r0[t] = (uint32_t)((uint64_t)nreads * t / nt);
    bool too_long = false;
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                uint64_t w = 0, bsum = 0;
                for (uint32_t i = r0[t]; i < r0[t + 1]; i++) { w += ((uint64_t)lens[i] + 15) >> 4; bsum += lens[i]; if (lens[i] > (uint32_t)kMaxReadLenAbs) too_long = true; }
                w0[t + 1] = w;
                b0[t + 1] = bsum;
            });
        for (auto &x : th) x.join();
    }
    if (too_long) return BK_ERR_PARAMS;
    for (unsigned t = 0; t < nt; t++) { w0[t + 1] += w0[t]; b0[t + 1] += b0[t]; }
    std::vector<std::vector<bk_nbase>> found(nt);
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                uint32_t *wp = words + w0[t];
                uint64_t at = b0[t];
                for (uint32_t i = r0[t]; i < r0[t + 1]; i++) {
                    const uint32_t len = lens[i];
                    const uint8_t *s = bases + (offs ? offs[i] : at);
                    at += len;
                    lens16[i] = (uint16_t)len;
                    for (uint32_t j0 = 0; j0 < len; j0 += 16) {
                        const uint32_t cnt = std::min<uint32_t>(16, len - j0);
                        if (cnt == 16) {                       // a whole word of a, c, g, t: sixteen bytes at once
                            bool clean;
                            const uint32_t w16 = have_bmi2 ? pack16_bmi2(s + j0, &clean) : pack16_plain(s + j0, &clean);
                            if (clean) { *wp++ = w16; continue; }
                        }
                        uint32_t v = 0;
                        for (uint32_t k = 0; k < cnt; k++) {
                            const uint32_t code = s[j0 + k] & 7u;
                            if (code > 3) {
                                std::vector<bk_nbase> &f = found[t];
                                if (!f.empty() && f.back().read == i && f.back().code == code && f.back().run < 255 &&
                                    (uint32_t)f.back().pos + f.back().run + 1 == j0 + k)
                                    f.back().run++;
                                else
                                    f.push_back(bk_nbase{i, (uint16_t)(j0 + k), (uint8_t)code, 0});
                            } else
                                v |= code << (30 - 2 * k);
                        }
                        *wp++ = v;
                    }
                }
            });
        for (auto &x : th) x.join();
    }
    uint64_t total = 0;
    for (unsigned t = 0; t < nt; t++) {
        for (const bk_nbase &e : found[t]) { if (total < exc_cap) exc[total] = e; total++; }
    }
    *n_exc = total;
    return total > exc_cap ? BK_ERR_MEM : BK_OK;
}

int bk_align_batch_packed(bk_ctx *c, const uint32_t *words, uint64_t n_words, const uint16_t *lens, uint32_t nreads, const bk_nbase *exc,
                          uint64_t n_exc, bk_hit *out)
{
    if (!c || (nreads && (!lens || !out)) || (n_words && !words) || (n_exc && !exc)) return BK_ERR_PARAMS;
    if (!nreads) return BK_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // (k_prep_fused<NW, PACKED> loads NW words from a read's first one whatever its length - up to kNwLongest for a batch whose
    // longest read has 257 .. 512 bases - so the buffer is padded by that many words behind the last read)
    static_assert(kPackedPadWords >= kNwLongest, "packed read words must be padded by the widest register-window family");
    if (n_words + kPackedPadWords > c->cap_in_words) {
        free_dev(c->d_in_words);
        c->d_in_words = nullptr;
        c->cap_in_words = 0;
        HIP_TRY(dev_malloc(&c->d_in_words, (n_words + kPackedPadWords) * 4));
        c->cap_in_words = n_words + kPackedPadWords;
    }
    if (n_exc > c->cap_in_exc) {
        free_dev(c->d_in_exc);
        c->d_in_exc = nullptr;
        c->cap_in_exc = 0;
        HIP_TRY(dev_malloc(&c->d_in_exc, n_exc * sizeof(bk_nbase)));
        c->cap_in_exc = n_exc;
    }
    if (nreads > c->cap_in_reads) {
        free_dev(c->d_in_offs); free_dev(c->d_in_lens); free_dev(c->d_in_out); free_dev(c->d_in_lens16);
        c->d_in_offs = nullptr; c->d_in_lens = nullptr; c->d_in_out = nullptr; c->d_in_lens16 = nullptr;
        c->cap_in_reads = 0;
        HIP_TRY(dev_malloc(&c->d_in_offs, (size_t)nreads * 8));
        HIP_TRY(dev_malloc(&c->d_in_lens, (size_t)nreads * 4));
        HIP_TRY(dev_malloc(&c->d_in_out, (size_t)nreads * sizeof(bk_hit)));
        c->cap_in_reads = nreads;
    }
    if (!c->d_in_lens16 || nreads > c->cap_in_lens16) {
        free_dev(c->d_in_lens16);
        c->d_in_lens16 = nullptr;
        HIP_TRY(dev_malloc(&c->d_in_lens16, (size_t)std::max(nreads, c->cap_in_reads) * 2));
        c->cap_in_lens16 = std::max(nreads, c->cap_in_reads);
    }
    if (n_words) HIP_TRY(hipMemcpyAsync(c->d_in_words, words, n_words * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_in_lens16, lens, (size_t)nreads * 2, hipMemcpyHostToDevice, s));
    if (n_exc) HIP_TRY(hipMemcpyAsync(c->d_in_exc, exc, n_exc * sizeof(bk_nbase), hipMemcpyHostToDevice, s));
    uint32_t maxlen = 0;
    int rc = engine_prepare_packed(c, c->d_in_lens16, nreads, n_words, c->d_in_exc, n_exc, c->d_in_lens, c->d_in_offs, &maxlen, s);
    if (rc) return rc;
    DevReads in;
    in.offs = c->d_in_offs; in.lens = c->d_in_lens; in.words = c->d_in_words; in.exc = c->d_in_exc; in.n_exc = n_exc;
    rc = align_device(c, in, nreads, c->d_in_out, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, c->d_in_out, (size_t)nreads * sizeof(bk_hit), hipMemcpyDeviceToHost));
    return BK_OK;
}

int bk_pair_batch(bk_ctx *c, const uint8_t *bases, const uint64_t *offs, const uint32_t *lens, uint32_t n_pairs, bk_hit *hits,
                  const bk_pe_params *pe)
{
    return bk_pair_batch_seg2(c, bases, offs, lens, n_pairs, hits, nullptr, pe);
}

int bk_pair_batch_seg2(bk_ctx *c, const uint8_t *bases, const uint64_t *offs, const uint32_t *lens, uint32_t n_pairs, bk_hit *hits,
                       bk_seg2 *seg2, const bk_pe_params *pe)
{
    if (!c || !pe || (n_pairs && (!bases || !offs || !lens || !hits))) return BK_ERR_PARAMS;
    if (c->params.min_chimeric_len > 0 && n_pairs && !seg2) return BK_ERR_PARAMS;
    if (pe->pe_mode < 1 || pe->pe_mode > 4 || pe->pair_min_len < 1 || pe->pair_max_len < pe->pair_min_len) return BK_ERR_PARAMS;
    if (!n_pairs) return BK_OK;
    if (n_pairs > 0x7fffffffu) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t nreads = 2 * n_pairs;
    uint64_t lo = ~0ULL, hi = 0;
    uint32_t maxlen = 0;
    for (uint32_t i = 0; i < nreads; i++) {
        if (lens[i] > (uint32_t)kMaxReadLenAbs) return BK_ERR_PARAMS;
        lo = std::min(lo, offs[i]);
        hi = std::max(hi, offs[i] + lens[i]);
        maxlen = std::max(maxlen, lens[i]);
    }
    uint64_t nbytes = hi - lo;
    if (nbytes + 16 > c->cap_in_bases) {
        free_dev(c->d_in_bases);
        c->d_in_bases = nullptr;
        c->cap_in_bases = 0;
        HIP_TRY(dev_malloc(&c->d_in_bases, nbytes + 16));
        c->cap_in_bases = nbytes + 16;
    }
    if (nreads > c->cap_in_reads) {
        free_dev(c->d_in_offs); free_dev(c->d_in_lens); free_dev(c->d_in_out);
        c->d_in_offs = nullptr; c->d_in_lens = nullptr; c->d_in_out = nullptr;
        c->cap_in_reads = 0;
        HIP_TRY(dev_malloc(&c->d_in_offs, (size_t)nreads * 8));
        HIP_TRY(dev_malloc(&c->d_in_lens, (size_t)nreads * 4));
        HIP_TRY(dev_malloc(&c->d_in_out, (size_t)nreads * sizeof(bk_hit)));
        c->cap_in_reads = nreads;
    }
    std::vector<uint64_t> rel(nreads);
    for (uint32_t i = 0; i < nreads; i++) rel[i] = offs[i] - lo;
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(c->d_in_bases, bases + lo, nbytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_in_offs, rel.data(), (size_t)nreads * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_in_lens, lens, (size_t)nreads * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_in_out, hits, (size_t)nreads * sizeof(bk_hit), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    DevReads in;
    in.bases = c->d_in_bases; in.offs = c->d_in_offs; in.lens = c->d_in_lens;
    int rc = engine_pair_device(c, in, n_pairs, c->d_in_out, maxlen, pe, s, seg2, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(hits, c->d_in_out, (size_t)nreads * sizeof(bk_hit), hipMemcpyDeviceToHost));
    return BK_OK;
}

// device-resident form: reads and their bk_hit records (the output of bk_align_batch_device for exactly these
// reads, interleaved PE1, PE2) already in HBM; hits are updated in place
int bk_pair_batch_device(bk_ctx *c, const void *d_bases, const void *d_offs, const void *d_lens, uint32_t n_pairs, void *d_hits,
                         const bk_pe_params *pe)
{
    return bk_pair_batch_seg2_device(c, d_bases, d_offs, d_lens, n_pairs, d_hits, nullptr, pe);
}

int bk_pair_batch_seg2_device(bk_ctx *c, const void *d_bases, const void *d_offs, const void *d_lens, uint32_t n_pairs, void *d_hits,
                              void *d_seg2, const bk_pe_params *pe)
{
    if (!c || !pe || (n_pairs && (!d_bases || !d_offs || !d_lens || !d_hits))) return BK_ERR_PARAMS;
    if (c->params.min_chimeric_len > 0 && n_pairs && !d_seg2) return BK_ERR_PARAMS;
    if (pe->pe_mode < 1 || pe->pe_mode > 4 || pe->pair_min_len < 1 || pe->pair_max_len < pe->pair_min_len) return BK_ERR_PARAMS;
    if (!n_pairs) return BK_OK;
    if (n_pairs > 0x7fffffffu) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t nreads = 2 * n_pairs;
    hipStream_t s = c->stream;
    HIP_TRY(hipMemsetAsync(c->d_small, 0, 16 * 4, s));
    launch_max_len((const uint32_t *)d_lens, nreads, c->d_small + 5, s);
    HIP_TRY(hipMemcpyAsync(c->h_small, c->d_small, 16 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint32_t maxlen = c->h_small[5];
    if (maxlen > (uint32_t)kMaxReadLenAbs) return BK_ERR_PARAMS;
    DevReads in;
    in.bases = (const uint8_t *)d_bases; in.offs = (const uint64_t *)d_offs; in.lens = (const uint32_t *)d_lens;
    return engine_pair_device(c, in, n_pairs, (bk_hit *)d_hits, maxlen, pe, s, nullptr, (bk_seg2 *)d_seg2);
}


// Test hook for the search stage (tests/test_gpu_search_stage.py): what LocateFirstExact / LocateLastExact's device form left for every
// (read, strand, core) of the phase named with "debug_stop_phase", as the kernels behind it would have read it
int bk_debug_intervals(bk_ctx *c, uint32_t cap_reads, uint32_t *n_act, uint32_t *iv_cores, uint32_t *act, uint64_t *first, uint32_t *count)
{
    if (!c || !n_act || !iv_cores) return BK_ERR_PARAMS;
    if (!c->dbg_valid) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    PhaseCtl ctl{};
    HIP_TRY(hipMemcpy(&ctl, c->d_ctl + c->dbg_phase, sizeof(PhaseCtl), hipMemcpyDeviceToHost));
    *n_act = ctl.n_act;
    *iv_cores = c->dbg_ivc;
    if (!act && !first && !count) return BK_OK;
    if (!act || !first || !count || ctl.n_act > cap_reads || ctl.n_act > c->dbg_n) return BK_ERR_PARAMS;
    const uint32_t na = ctl.n_act, stride = c->dbg_n, planes = 2 * c->dbg_ivc;
    HIP_TRY(hipMemcpy(act, c->d_act[c->dbg_cur], (size_t)na * 4, hipMemcpyDeviceToHost));
    std::vector<uint2> h2;
    std::vector<uint32_t> hn;
    if (c->d_iv2) h2.resize(na);
    else hn.resize(na);
    for (uint32_t pl = 0; pl < planes; pl++) {
        uint64_t *f = first + (size_t)pl * na;
        uint32_t *q = count + (size_t)pl * na;
        if (c->d_iv2) {
            HIP_TRY(hipMemcpy(h2.data(), c->d_iv2 + (size_t)pl * stride, (size_t)na * 8, hipMemcpyDeviceToHost));
            for (uint32_t a = 0; a < na; a++) { f[a] = h2[a].x; q[a] = h2[a].y; }
        } else {
            HIP_TRY(hipMemcpy(f, c->d_iv_first + (size_t)pl * stride, (size_t)na * 8, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(q, c->d_iv_n + (size_t)pl * stride, (size_t)na * 4, hipMemcpyDeviceToHost));
        }
    }
    return BK_OK;
}


int bk_get_counters(bk_ctx *c, bk_counters *out, int reset)
{
    if (!c || !out) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long hs[kCtrStripes * 8], h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpy(hs, c->d_ctr, sizeof(hs), hipMemcpyDeviceToHost));
    for (int i = 0; i < kCtrStripes * 8; i++) h[i & 7] += hs[i];
    memset(out, 0, sizeof(*out));
    out->n_search = h[0]; out->n_cand = h[1]; out->n_lcm_calls = h[2]; out->n_heavy = h[3]; out->n_cand_heavy = h[4]; out->reserved[0] = h[5]; out->reserved[1] = h[6];
    if (reset) HIP_TRY(dev_zero_now(c->d_ctr, sizeof(hs)));
    return BK_OK;
}

int bk_get_timing(bk_ctx *c, bk_timing *out, int reset)
{
    if (!c || !out) return BK_ERR_PARAMS;
    *out = c->timing;
    if (reset) c->timing = bk_timing{};
    return BK_OK;
}


int bk_build_sa_device(const void *d_seq, uint64_t concat_len, void *d_sa_out, int sfx_el_size, int device_id)
{
    if (!d_seq || !d_sa_out || !concat_len || (sfx_el_size != 4 && sfx_el_size != 5)) return BK_ERR_PARAMS;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BK_ERR_NODEVICE;
    if (device_id < 0 || device_id >= ndev) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(device_id));
    return build_sa_device((const uint8_t *)d_seq, concat_len, d_sa_out, sfx_el_size, nullptr);
}


}  // extern "C"
