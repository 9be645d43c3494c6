// bk_image.cpp - the index image in HBM and the life of a context (include/biokanga_amd.h: bk_ctx_create[_ex], _from_device, _clone, _destroy,
// bk_image_policy): .sfx -> packed target, suffix array, k-mer table, second- to fourth-level keys, inverse suffix array, 2-bit target, the
// suffix-ordered window array; the worker that grows a lean image while batches run.  Replaces CSfxArrayV3::Open + SetTargBlock
// (libbiokanga/SfxArrayV2.cpp:891-1103,1836-1890) and the MinCoreLen / MaxIter set-up of CAligner::Align / LocateCoredApprox
// (Aligner.cpp:341-356,8725-8761).  Compiled with hipcc; the kernels are in bk_index.hip.
#include "bk_engine_int.h"

namespace bk {
// hipMemset that has happened when it returns: a memset only joins the null stream's queue, and the pipelines' streams (non-blocking) do
// not wait for that queue - a kernel launched on one of them right after could meet the old bytes, or have its own writes zeroed later
hipError_t dev_zero_now(void *p, size_t bytes)
{
    hipError_t e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}

hipError_t dev_malloc_bytes(void **p, size_t bytes)
{
    const int poison = env::poison();
    const bool timing = env::timing();
    timespec ta, tb;
    if (timing) clock_gettime(CLOCK_MONOTONIC, &ta);
    hipError_t e = hipMalloc(p, bytes);
    if (timing) {          // (BK_TIMING: an allocation that took the driver more than 2 ms says so)
        clock_gettime(CLOCK_MONOTONIC, &tb);
        const double ms = 1e3 * (double)(tb.tv_sec - ta.tv_sec) + 1e-6 * (double)(tb.tv_nsec - ta.tv_nsec);
        if (ms > 2.0) fprintf(stderr, "bk timing: hipMalloc of %.2f GB took %.1f ms\n", (double)bytes / 1e9, ms);
    }
    if (e == hipSuccess && poison >= 0 && bytes) {
        e = hipMemset(*p, poison & 0xff, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();       // (a memset returns before it is done, and the contexts' streams do not wait for the null stream)
    }
    return e;
}


int derive_cfg(bk_ctx *c)
{
    const bk_align_params &p = c->params;
    if (p.max_subs < 0 || p.max_subs > 25 || p.min_edit_dist < 1 || p.min_edit_dist > 2 || p.align_strand < 0 ||
        p.align_strand > 2 || p.pmode < 0 || p.pmode > 3 || p.max_ns < 0 || p.max_ns > 5)
        return BK_ERR_PARAMS;
    if (p.max_ml < 0 || p.max_ml > BK_MAX_ML) return BK_ERR_PARAMS;
    if (p.best_matches && p.max_ml < 2) return BK_ERR_PARAMS;
    if (p.micro_indel_len < 0 || p.micro_indel_len > 20) return BK_ERR_PARAMS;           // cMaxMicroInDelLen
    if ((p.micro_indel_len || p.splice_junct_len) && p.max_ml > 1 && p.best_matches) return BK_ERR_PARAMS;   // LocateBestMatches has no such branches
    if (p.min_chimeric_len != 0 && (p.min_chimeric_len < 50 || p.min_chimeric_len > 99)) return BK_ERR_PARAMS;          // kanga.cpp:648-653
    // -c with the multi-loci modes: the chimeric call lists its loci - also together with -a / -A (the microInDel / splice searches and the
    // chimeric call run on one set of counts and hits; pinned by tests/golden/chimmlindel).  Only -N is refused, as the reference refuses
    // it itself (no chimeric branch in LocateBestMatches; kanga.cpp:712-716)
    if (p.min_chimeric_len != 0 && p.max_ml > 1 && p.best_matches) return BK_ERR_PARAMS;
    if (p.splice_junct_len != 0 && (p.splice_junct_len < 25 || p.splice_junct_len > 100000)) return BK_ERR_PARAMS;   // cMin/cMaxJunctAlignSep
    DevAlignCfg &g = c->cfg;
    g.max_subs = p.max_subs;
    g.mm_delta = p.min_edit_dist;
    g.align_strand = p.align_strand;
    g.max_ns = p.max_ns;
    g.max_hits = p.max_ml > 1 ? p.max_ml : 1;
    // CAligner::LocateCoredApprox, Aligner.cpp:8725-8761
    uint64_t t = c->tot_seq_len;
    int m;
    if (t <= 500000ULL) m = 4;
    else if (t <= 20000000ULL) m = 7;
    else if (t <= 250000000ULL) m = 11;
    else if (t <= 3500000000ULL) m = 12;
    else m = 15;
    switch (p.pmode) {
    case 2: g.slides_per100 = 9; break;
    case 1: m += 1; g.slides_per100 = 8; break;
    case 0: m += 2; g.slides_per100 = 8; break;
    default: m += 4; g.slides_per100 = 6; break;
    }
    g.min_core_len = m;
    // CAligner::Align, Aligner.cpp:341-356
    switch (p.pmode) {
    case 0: g.max_iter = 5000; break;
    case 1: g.max_iter = 10000; break;
    case 2: g.max_iter = 20000; break;
    default: g.max_iter = 2500; break;
    }
    if (g.heavy_thresh < 0 || g.heavy_thresh > 100) g.heavy_thresh = 64;
    return BK_OK;
}


void free_dev(void *p)
{
    if (!p) return;
    const bool timing = env::timing();
    timespec ta, tb;
    if (timing) clock_gettime(CLOCK_MONOTONIC, &ta);
    (void)hipFree(p);
    if (timing) {
        clock_gettime(CLOCK_MONOTONIC, &tb);
        const double ms = 1e3 * (double)(tb.tv_sec - ta.tv_sec) + 1e-6 * (double)(tb.tv_nsec - ta.tv_nsec);
        if (ms > 2.0) fprintf(stderr, "bk timing: hipFree took %.1f ms\n", ms);
    }
}

// zero-fill that stays correct for spans of 4 GiB and more: hipMemsetAsync is not trusted with those (bk_index.hip,
// k_fill_u64), so large clears go through the fill kernel (8-byte words, plus a byte tail through hipMemsetAsync)
hipError_t clear_dev(void *p, size_t bytes, hipStream_t s)
{
    if (bytes < (1ULL << 30) || ((uintptr_t)p & 7)) {
        for (size_t at = 0; at < bytes; at += (1ULL << 30)) {
            hipError_t e = hipMemsetAsync((uint8_t *)p + at, 0, std::min<size_t>(1ULL << 30, bytes - at), s);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    const size_t words = bytes / 8;
    launch_fill_u64((unsigned long long *)p, words, 0ULL, s);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && (bytes & 7)) e = hipMemsetAsync((uint8_t *)p + words * 8, 0, bytes & 7, s);
    return e;
}


int pick_k(uint64_t n)
{
    int k = 1;
    while (k < 16 && (1ULL << (2 * k)) < n) k++;
    return k < 8 ? 8 : k;
}

// The k-mer table, the second-level keys and the inverse suffix array are all made by one pass over suffix array indexes, so they can be
// made range by range: behind the suffix array's upload (bk_ctx_create_ex sends it in slices and these kernels work on a slice while the
// next crosses PCIe), or in one go.  tables_begin decides and allocates, tables_range enqueues, tables_end checks and publishes.

// ------------------------------------------------------------------------------------------------
// BK_CTX_GROW_IMAGE: a context starts with the image a short job wants and grows the tables that pay over thousands of millions of reads
// - the key arrays behind the second-level keys, the k-mer table entries that carry their bucket's first key - while it works: a thread
// of its own allocates and fills them on a stream of its own (they are made of the suffix array, the target and the second-level keys,
// which the batches under way only read), and the next batch after they are complete takes them in.  Results never depend on which
// image a batch ran on.
void grow_worker(bk_ctx *c)
{
    int st = 3;
    hipStream_t s = nullptr;
    unsigned long long *d_bad = nullptr;
    do {
        if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
        const DevIndex &ix = c->grow_ix;             // (set by the thread that started this one, before it did)
        const uint64_t n = ix.n;
        const uint64_t need = k2s_start(n, kK2Levels + 1) * 4;
        int nk = 0;
        if (ix.k2 != nullptr && ix.kx[0] == nullptr) {
            for (int i = 0; i < kMoreKeys; i++) {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need > free_b || free_b - need < total_b / 5) break;
                if (dev_malloc(&c->grow_kx[i], need) != hipSuccess) { (void)hipGetLastError(); c->grow_kx[i] = nullptr; break; }
                nk = i + 1;
            }
        }
        if (nk) {
            unsigned long long bad2[2] = {0, 0};
            bool ok = dev_malloc(&d_bad, 16) == hipSuccess && hipMemsetAsync(d_bad, 0, 16, s) == hipSuccess;
            if (ok) {
                launch_build_k2(ix, const_cast<uint32_t *>(ix.k2), c->grow_kx[0], nk > 1 ? c->grow_kx[1] : nullptr, d_bad, s, 0, n, false);
                ok = hipMemcpyAsync(bad2, d_bad, 16, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess && !bad2[0] && !bad2[1];
            }
            if (ok) {
                for (int i = 0; i < nk; i++) launch_build_k2_levels(c->grow_kx[i], n, s);
                ok = hipStreamSynchronize(s) == hipSuccess;
            }
            if (!ok) {
                (void)hipGetLastError();
                for (int i = 0; i < kMoreKeys; i++) { free_dev(c->grow_kx[i]); c->grow_kx[i] = nullptr; }
                nk = 0;
            }
        }
        if (c->grow_want_ktab2 && ix.k2 != nullptr && ix.ktab32 != nullptr) {
            const uint64_t n_entries = (1ULL << (2 * ix.k)) + 1;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && n_entries * 8 + (total_b / 5) < free_b && dev_malloc(&c->grow_ktab2, n_entries * 8) == hipSuccess) {
                launch_make_ktab2(ix.ktab32, ix.k2, n_entries, n, c->grow_ktab2, s, c->grow_want_elem ? ix.sa_lo : nullptr);
                if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); free_dev(c->grow_ktab2); c->grow_ktab2 = nullptr; }
            } else
                (void)hipGetLastError();
        }
        st = (nk || c->grow_ktab2) ? 2 : 3;
    } while (false);
    free_dev(d_bad);
    if (s) (void)hipStreamDestroy(s);
    c->grow_state.store(st, std::memory_order_release);
}

// the worker's tables become the context's (between two batches; the device is made idle before the k-mer table it replaces is given back)
void grow_take_in(bk_ctx *c)
{
    if (c->grow_thread.joinable()) c->grow_thread.join();
    if (c->grow_state.load(std::memory_order_acquire) == 2) {
        for (int i = 0; i < kMoreKeys; i++)
            if (c->grow_kx[i]) { c->d_kx[i] = c->grow_kx[i]; c->ix.kx[i] = c->grow_kx[i]; c->grow_kx[i] = nullptr; }
        if (c->ix.kx[0]) c->use_k3 = kMoreKeys;
        if (c->grow_ktab2) {
            // (earlier batches may still run on a caller's stream - bk_stream's, bk_align_batch_device_async's - with the old table's
            // address in their kernel arguments: the whole device is waited for, not the context's own stream)
            (void)hipDeviceSynchronize();
            free_dev(c->d_ktab);
            c->d_ktab = c->grow_ktab2;
            c->grow_ktab2 = nullptr;
            c->ktab_bytes = (size_t)((1ULL << (2 * c->ix.k)) + 1) * 8;
            c->ktab_is2 = true;
            c->use_ktab2 = c->grow_want_elem ? 2 : 1;
            c->ix.ktab2_elem = c->grow_want_elem ? 1 : 0;
            c->ix.ktab32 = nullptr;
            c->ix.ktab2 = reinterpret_cast<const uint2 *>(c->d_ktab);
        }
        if (!c->sort_lists_set) c->sort_lists = (c->sort_lists & ~1) | (c->ix.kx[0] == nullptr ? 1 : 0);
        if (env::timing())
            fprintf(stderr, "biokanga_amd: long-run tables taken in after %llu reads: %d key array(s) behind the second-level keys%s\n",
                    (unsigned long long)c->grow_seen, (c->ix.kx[0] != nullptr) + (c->ix.kx[1] != nullptr), c->ix.ktab2 ? ", first keys in the k-mer table" : "");
    }
    c->grow_state.store(4, std::memory_order_release);
}

// whatever the worker has made is dropped (the tables are about to be rebuilt, or the context ends)
void grow_drop(bk_ctx *c)
{
    if (c->grow_thread.joinable()) c->grow_thread.join();
    for (int i = 0; i < kMoreKeys; i++) { free_dev(c->grow_kx[i]); c->grow_kx[i] = nullptr; }
    free_dev(c->grow_ktab2);
    c->grow_ktab2 = nullptr;
    if (c->grow_state.load() != 0) c->grow_state.store(4);
}

// called with every batch: starts the worker once the context has seen enough reads, takes its tables in when they are complete
void grow_tick(bk_ctx *c, uint64_t nreads, bool now)
{
    if (!c->grow_enabled) return;
    const int st = c->grow_state.load(std::memory_order_acquire);
    if (st != 0 && st != 4) c->grow_seen += nreads;
    if (st == 0) {
        c->grow_seen += nreads;
        if ((now || c->grow_seen >= c->grow_after) && c->tables_built && c->ix.k2 != nullptr) {
            c->grow_ix = c->ix;
            c->grow_want_ktab2 = !c->ktab64 && !c->ktab_is2;
            c->grow_want_elem = c->grow_elem && c->d_sa_hi == nullptr;
            c->grow_state.store(1);
            c->grow_thread = std::thread(grow_worker, c);
        }
    } else if (st == 2 || st == 3 || (st == 1 && c->grow_wait))
        grow_take_in(c);
}

int tables_begin(bk_ctx *c, TablePlan &tp)
{
    grow_drop(c);
    free_dev(c->d_ktab); free_dev(c->d_ktab_hi); free_dev(c->d_k2); free_dev(c->d_isa);
    c->d_ktab = nullptr; c->d_ktab_hi = nullptr; c->d_k2 = nullptr; c->d_isa = nullptr;
    for (int i = 0; i < kMoreKeys; i++) { free_dev(c->d_kx[i]); c->d_kx[i] = nullptr; c->ix.kx[i] = nullptr; }
    c->ix.ktab32 = nullptr; c->ix.ktab64 = nullptr; c->ix.ktab_hi = nullptr; c->ix.ktab2 = nullptr; c->ix.k2 = nullptr; c->ix.isa = nullptr;
    c->ktab_is2 = false;
    c->ix.ktab2_elem = 0;
    c->ix.k = 0;
    // What the HBM has room for is decided before anything is allocated, in the order of what a byte buys: k-mer table, second-level
    // keys, inverse suffix array, the key arrays behind the second-level keys, then the k-mer table's second words - each only where a
    // fifth of the HBM stays free behind it (batch scratch, window array).  Nothing is given back or allocated again afterwards: an
    // allocation made after a large hipFree waits for the driver to wipe what was freed (profiles/NOTES.md, round 6).
    if (c->use_ktab) {
        int k = c->k_req > 0 ? c->k_req : pick_k(c->ix.n);
        if (k > 16) k = 16;
        if (k < 2) k = 2;
        const uint64_t ncodes = 1ULL << (2 * k);
        c->ktab64 = c->ix.n >= (1ULL << 32) || c->ktab_wide != 0;
        const uint64_t ktab_bytes = (ncodes + 1) * (c->ktab64 ? 8 : 4);
        const uint64_t need = k2s_start(c->ix.n, kK2Levels + 1) * 4;          // (the keys and their sampled levels, bk_dev_k2.h)
        const bool want_isa = c->use_wave && c->use_isa && c->d_sa_hi == nullptr && c->ix.n < (1ULL << 32);
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        uint64_t planned = ktab_bytes;
        auto fits = [&](uint64_t more) { return planned + more <= free_b && free_b - planned - more >= total_b / 5; };
        // second-level key array; needs the k-mer table.  Skipped (search falls back to the one-pass kernel) when it would not leave a
        // fifth of the HBM free, or - found in tables_end - if the suffix array is not ordered the way the bisection needs (never
        // seen; checked because .sfx files come from outside).
        if (c->use_k2 && fits(need)) { tp.k2 = true; planned += need; }
        if (want_isa) planned += c->ix.n * 4;
        // third- and fourth-level keys (the 15 bases after those, and the 15 after these): as much again each, for the cores of
        // more than k + 15 bases
        for (int i = 0; tp.k2 && i < kMoreKeys && i < c->use_k3; i++) {
            if (!fits(need)) break;
            tp.kx = i + 1;
            planned += need;
        }
        // the k-mer table's entries as pairs (4-byte indexes; 17 GB more at k = 16): see DevIndex::ktab2
        if (tp.k2 && c->use_ktab2 && !c->ktab64 && fits(ktab_bytes)) tp.ktab2 = true;
        const size_t bytes = (size_t)ktab_bytes * (tp.ktab2 ? 2 : 1);
        HIP_TRY(dev_malloc(&c->d_ktab, bytes));
        c->ktab_bytes = bytes;
        tp.ktab = true;
        tp.k = k;
        if (tp.k2) {
            HIP_TRY(dev_malloc(&c->d_k2, need));
            HIP_TRY(dev_malloc(&tp.d_bad, 16));
            HIP_TRY(hipMemsetAsync(tp.d_bad, 0, 16, c->stream));
            for (int i = 0; i < tp.kx; i++) HIP_TRY(dev_malloc(&c->d_kx[i], need));
        }
    }
    if (c->use_wave && c->use_isa && c->d_sa_hi == nullptr && c->ix.n < (1ULL << 32)) {
        HIP_TRY(dev_malloc(&c->d_isa, c->ix.n * 4));
        tp.isa = true;
    }
    return BK_OK;
}

// suffix array indexes [i0, i1) have arrived
int tables_range(bk_ctx *c, const TablePlan &tp, uint64_t i0, uint64_t i1, unsigned long long *bucket_starts)
{
    DevIndex ix = c->ix;
    ix.k = tp.k;
    const bool last = i1 >= c->ix.n;
    if (tp.ktab) launch_build_ktab(ix, c->d_ktab, tp.k, c->ktab64, c->stream, i0, last ? c->ix.n + 1 : i1, bucket_starts, tp.ktab2);
    if (tp.k2) launch_build_k2(ix, c->d_k2, tp.kx > 0 ? c->d_kx[0] : nullptr, tp.kx > 1 ? c->d_kx[1] : nullptr, tp.d_bad, c->stream, i0, i1);
    if (tp.isa) launch_build_isa(c->d_sa_lo, c->ix.n, c->d_isa, c->stream, i0, i1);
    HIP_TRY(hipGetLastError());
    return BK_OK;
}

int tables_end(bk_ctx *c, TablePlan &tp)
{
    unsigned long long bad2[2] = {0, 0};
    if (tp.k2) HIP_TRY(hipMemcpyAsync(bad2, tp.d_bad, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (tp.ktab && c->ktab64 && c->ktab_wide != 2) {
        // the table of an index beyond 2^32 suffixes at half its size (17 of 34 GB at k = 16: room for the window array such an index may
        // ask for): 32-bit offsets from a 64-bit start per 2^16 codes - unless a group of codes spans 2^32 suffixes
        const uint64_t n_entries = (1ULL << (2 * tp.k)) + 1, n_hi = (n_entries >> 16) + 2;
        uint32_t *d_off = nullptr, *d_flag = nullptr, flag = 1;
        uint64_t *d_hi = nullptr;
        if (dev_malloc(&d_off, n_entries * 4) == hipSuccess && dev_malloc(&d_hi, n_hi * 8) == hipSuccess && dev_malloc(&d_flag, 4) == hipSuccess &&
            hipMemsetAsync(d_flag, 0, 4, c->stream) == hipSuccess) {
            launch_pack_ktab64((const uint64_t *)c->d_ktab, n_entries, d_off, d_hi, d_flag, c->stream);
            if (hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) flag = 1;
        }
        (void)hipGetLastError();
        free_dev(d_flag);
        if (flag == 0) {
            free_dev(c->d_ktab);
            c->d_ktab = d_off;
            c->ktab_bytes = (size_t)n_entries * 4;
            c->d_ktab_hi = d_hi;
        } else { free_dev(d_off); free_dev(d_hi); }
    }
    if (tp.ktab) {
        if (c->ktab64 && c->d_ktab_hi) { c->ix.ktab32 = (const uint32_t *)c->d_ktab; c->ix.ktab_hi = c->d_ktab_hi; }
        else if (c->ktab64) c->ix.ktab64 = (const uint64_t *)c->d_ktab;
        else if (tp.ktab2) { c->ix.ktab2 = reinterpret_cast<const uint2 *>(c->d_ktab); c->ktab_is2 = true; c->ix.ktab2_elem = (c->use_ktab2 >= 2 && c->d_sa_hi == nullptr) ? 1 : 0; }
        else c->ix.ktab32 = (const uint32_t *)c->d_ktab;
        c->ix.k = tp.k;
    }
    const unsigned long long bad = bad2[0];
    if (tp.kx && (bad || bad2[1])) {
        if (!bad) fprintf(stderr, "biokanga_amd: suffix array not in nibble order at %llu place(s) beyond the second-level keys; third-level keys disabled\n", bad2[1]);
        for (int i = 0; i < kMoreKeys; i++) { free_dev(c->d_kx[i]); c->d_kx[i] = nullptr; }
        tp.kx = 0;
    }
    if (tp.k2 && bad) {
        fprintf(stderr, "biokanga_amd: suffix array not in nibble order inside %llu k-mer bucket(s); second-level keys disabled\n", bad);
        free_dev(c->d_k2);
        c->d_k2 = nullptr;
    } else if (tp.k2) {
        launch_build_k2_levels(c->d_k2, c->ix.n, c->stream);
        for (int i = 0; i < tp.kx; i++) launch_build_k2_levels(c->d_kx[i], c->ix.n, c->stream);
        HIP_TRY(hipGetLastError());
        c->ix.k2 = c->d_k2;
        for (int i = 0; i < kMoreKeys; i++) c->ix.kx[i] = i < tp.kx ? c->d_kx[i] : nullptr;
    }
    // the second words of a k-mer table of pairs (its first words, the bucket starts, are in place): a bucket's only key or the map of
    // its keys' first five bits - or maps that hide nothing where the keys turned out unusable
    if (tp.ktab2) {
        if (c->d_k2 == nullptr) c->ix.ktab2_elem = 0;       // (no keys, no two-pass search: nothing hands elements on)
        launch_fill_ktab2_y(c->d_ktab, c->d_k2, (1ULL << (2 * tp.k)) + 1, c->stream, c->ix.ktab2_elem ? c->d_sa_lo : nullptr);
        HIP_TRY(hipGetLastError());
    }
    if (tp.isa) c->ix.isa = c->d_isa;
    HIP_TRY(hipStreamSynchronize(c->stream));       // (batches run on their callers' streams, which do not wait for this one)
    // pass B's items grouped by bucket: 1.6 ms of a C2 step's pass B for 2.2 ms of sorting once the deep bisections run over key arrays
    // (profiles/NOTES.md, round 5) - grouped only where they still run over suffix array + target
    if (!c->sort_lists_set) c->sort_lists = (c->sort_lists & ~1) | (c->ix.kx[0] == nullptr ? 1 : 0);
    c->tables_built = true;
    return BK_OK;
}

// (re)builds all three over the whole array: contexts made from a device image, and the knobs that change a table's shape
int build_tables(bk_ctx *c)
{
    TablePlan tp;
    int rc = tables_begin(c, tp);
    if (!rc) rc = tables_range(c, tp, 0, c->ix.n);
    if (!rc) rc = tables_end(c, tp);
    return rc;
}

// 2 bit/base target copy + N/EOS block bitmap for the window compare of the extend kernels
int build_tgt2(bk_ctx *c)
{
    free_dev(c->d_tgt2); free_dev(c->d_nflag); free_dev(c->d_tgt2s); c->d_tgt2s = nullptr; c->ix.tgt2s = nullptr;
    c->d_tgt2 = nullptr; c->d_nflag = nullptr;
    c->ix.tgt2 = nullptr; c->ix.nflag = nullptr;
    if (!c->use_tgt2) return BK_OK;
    const uint64_t nblocks = c->n_tgt4_words / 4;
    // flag granule: the smallest power of two that keeps the bitmap within 16 KB; at least 512 bases so that a
    // window of the register kernels (<= 16 * kNwLongest bases) spans at most two regions
    static_assert(16 * kNwLongest <= 512, "a register-kernel window must not span more than two flag regions");
    int shift = 9;
    while ((((nblocks * 64) >> shift) + 7) / 8 > 16384) shift++;
    const uint64_t flag_bytes = (((((nblocks * 64) >> shift) + 1) + 31) / 32) * 4 + 16;
    HIP_TRY(dev_malloc(&c->d_tgt2, nblocks * 16 + 64));
    HIP_TRY(dev_malloc(&c->d_nflag, flag_bytes));
    c->nflag_bytes = flag_bytes;
    HIP_TRY(hipMemsetAsync(c->d_nflag, 0, flag_bytes, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_tgt2 + nblocks * 2, 0, 64, c->stream));
    launch_pack_target2(c->d_tgt4, c->n_tgt4_words, c->d_tgt2, (unsigned int *)c->d_nflag, shift, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->ix.tgt2 = c->d_tgt2;
    c->ix.nflag = c->d_nflag;
    c->ix.nflag_bytes = (uint32_t)flag_bytes;
    free_dev(c->d_tgt2s);
    c->d_tgt2s = nullptr;
    c->ix.tgt2s = nullptr;
    if (c->use_tgt2 >= 2) {
        // second copy: element j holds tgt2[j + 4], i.e. logical byte p sits at physical byte p - 32
        HIP_TRY(dev_malloc(&c->d_tgt2s, nblocks * 16 + 64));
        HIP_TRY(clear_dev(c->d_tgt2s, nblocks * 16 + 64, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_tgt2s, c->d_tgt2 + 4, (nblocks * 2 - 4) * 8, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->ix.tgt2s = c->d_tgt2s;
    }
    c->ix.flag_shift = shift;
    return BK_OK;
}


// entry table, counters and the small per-context buffers; parameters -> DevAlignCfg
int setup_entries(bk_ctx *c, const bk_entry_info *entries, uint32_t n_entries)
{
    StageClock clk0;
    // entries
    c->entries.assign(entries, entries + n_entries);
    c->tot_seq_len = 0;
    std::vector<uint64_t> es(n_entries), ee(n_entries);
    std::vector<uint32_t> ei(n_entries);
    for (uint32_t i = 0; i < n_entries; i++) {
        es[i] = entries[i].start_ofs;
        ee[i] = entries[i].end_ofs;
        ei[i] = entries[i].entry_id;
        c->tot_seq_len += entries[i].seq_len;
        if (i && es[i] <= ee[i - 1]) return BK_ERR_PARAMS;
    }
    HIP_TRY(dev_malloc(&c->d_ent_start, n_entries * 8));
    HIP_TRY(dev_malloc(&c->d_ent_end, n_entries * 8));
    HIP_TRY(dev_malloc(&c->d_ent_id, n_entries * 4));
    HIP_TRY(hipMemcpy(c->d_ent_start, es.data(), n_entries * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_ent_end, ee.data(), n_entries * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_ent_id, ei.data(), n_entries * 4, hipMemcpyHostToDevice));
    c->ix.ent_start = c->d_ent_start;
    c->ix.ent_end = c->d_ent_end;
    c->ix.ent_id = c->d_ent_id;
    {   // EntryID -> entry index, for the per-sequence hit counts
        uint32_t max_id = 0;
        for (uint32_t i = 0; i < n_entries; i++) max_id = std::max(max_id, ei[i]);
        if ((uint64_t)max_id > 16ULL * n_entries + (1u << 20)) return BK_ERR_PARAMS;
        std::vector<uint32_t> map((size_t)max_id + 1, 0xFFFFFFFFu);
        for (uint32_t i = 0; i < n_entries; i++) map[ei[i]] = i;
        HIP_TRY(dev_malloc(&c->d_id2idx, map.size() * 4));
        HIP_TRY(hipMemcpy(c->d_id2idx, map.data(), map.size() * 4, hipMemcpyHostToDevice));
        c->ix.id2idx = c->d_id2idx;
        c->ix.max_id = max_id;
    }
    c->ix.n_ent = n_entries;
    HIP_TRY(dev_malloc(&c->d_seq_counts, n_entries * 8));
    HIP_TRY(dev_zero_now(c->d_seq_counts, n_entries * 8));
    HIP_TRY(dev_malloc(&c->d_ctr, (size_t)kCtrStripes * 8 * 8));
    HIP_TRY(dev_zero_now(c->d_ctr, (size_t)kCtrStripes * 8 * 8));
    HIP_TRY(dev_malloc(&c->d_small, 16 * 4));
    HIP_TRY(dev_malloc(&c->d_ctl, sizeof(PhaseCtl) * (kMaxPhases + 2)));
    HIP_TRY(hipHostMalloc(&c->h_ctl, sizeof(PhaseCtl) * (kMaxPhases + 2)));
    HIP_TRY(bk::make_wait_event(&c->ev_ctl));
    HIP_TRY(bk::make_wait_event(&c->ev_wait));
    HIP_TRY(dev_malloc(&c->d_ctr_aux, 32));
    HIP_TRY(hipHostMalloc(&c->h_small, 2 * sizeof(PhaseCtl)));
    int rc = derive_cfg(c);
    clk0.lap("entry table, small buffers");
    return rc;
}

int finish_ctx(bk_ctx *c, const bk_entry_info *entries, uint32_t n_entries)
{
    int rc = c->entries_set ? BK_OK : setup_entries(c, entries, n_entries);
    if (rc) return rc;
    c->entries_set = true;
    StageClock clk;
    if (!c->tables_built) {                // (bk_ctx_create_ex makes them behind the suffix array's upload)
        rc = build_tables(c);
        clk.lap("k-mer table, second-level keys, inverse suffix array");
        if (rc) return rc;
    }
    if (!c->tgt2_built) {
        rc = build_tgt2(c);      // the hash scratch of the general kernels is sized when they first run
        clk.lap("2-bit target");
    }
    return rc;
}

int new_ctx(bk_ctx **out, int device_id, const bk_align_params *p, bk_ctx **pc)
{
    if (!out || !p) return BK_ERR_PARAMS;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BK_ERR_NODEVICE;
    if (device_id < 0 || device_id >= ndev) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(device_id));
    bk_ctx *c = new bk_ctx();
    bk::live_contexts()++;                         // (how host threads wait depends on how many contexts share the process's CPUs: bk_wait.h)
    c->device = device_id;
    c->params = *p;
    c->cfg.heavy_thresh = 64;
    c->debug = env::debug_phases();
    if (c->params.max_ml == 0) c->params.max_ml = 1;
    if (hipStreamCreate(&c->stream) != hipSuccess) { bk::live_contexts()--; delete c; return BK_ERR_INTERNAL; }
    *pc = c;
    return BK_OK;
}

// uploads the 1 B/base sequence + suffix array that already sit in device memory
int adopt_device_image(bk_ctx *c, const uint8_t *d_seq, uint64_t n, const uint8_t *d_sa, int el)
{
    c->el_size = (uint32_t)el;
    c->ix.n = n;
    uint64_t nwords = ((n + 15) / 16 + (kMaxReadLenAbs / 16) + 4 + 3) & ~3ULL;      // whole 64-base blocks
    HIP_TRY(dev_malloc(&c->d_tgt4, nwords * 8));
    launch_pack_target(d_seq, n, c->d_tgt4, nwords, c->stream);
    HIP_TRY(hipGetLastError());
    c->n_tgt4_words = nwords;
    c->sort_shift = 0;
    while ((n >> c->sort_shift) >= (1ULL << 32)) c->sort_shift++;
    if (d_sa == nullptr) {
        // (4-byte elements that the caller has put where they stay: c->d_sa_lo is allocated and filled)
        if (el != 4 || !c->d_sa_lo) return BK_ERR_INTERNAL;
    } else {
        HIP_TRY(dev_malloc(&c->d_sa_lo, n * 4));
        if (el == 5) {
            HIP_TRY(dev_malloc(&c->d_sa_hi, n));
            launch_split_sa5(d_sa, n, c->d_sa_lo, c->d_sa_hi, c->stream);
            HIP_TRY(hipGetLastError());
        } else
            HIP_TRY(hipMemcpyAsync(c->d_sa_lo, d_sa, n * 4, hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->ix.tgt4 = c->d_tgt4;
    c->ix.sa_lo = c->d_sa_lo;
    c->ix.sa_hi = c->d_sa_hi;
    return BK_OK;
}


// The suffix-ordered window array (DevIndex::swin, 48 bytes per suffix it holds) is built when the first batch it can serve arrives - reads
// of up to kSwLen bases whose core offsets stay within kSwPre - for the part of the suffix array the wave kernel's long walks visit
// (bk_index.hip, k_swin_cover: a tenth of a 3.1 Gbp index), within a budget of the HBM that is free next to this batch's scratch.
}  // namespace bk

// frees the suffix-ordered window array (and does not build it again): called when something else needs the HBM
void bk::release_swin(bk_ctx *c)
{
    if (!c || !c->d_swin) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipDeviceSynchronize();
    free_dev(c->d_swin);
    free_dev(c->d_swmap);
    c->d_swin = nullptr;
    c->d_swmap = nullptr;
    c->ix.swin = nullptr;
    c->ix.swmap = nullptr;
    c->swin_bytes = 0;
    c->swin_denied = true;
    fprintf(stderr, "biokanga_amd: window array released to make room\n");
}


namespace bk {
// The partial array is made range by range of the suffix array (one range when the index is already there; behind the slices of the
// suffix array's upload when it is still arriving - bk_ctx_create_ex): per level (core length) the break bitmap of the runs of suffixes
// sharing that many bases and the per-block coverage it implies, block numbers by a scan that continues the ranges before, then the
// entries - no more of them than `budget` bytes hold (blocks beyond it stay uncovered: coverage never changes a result).  Nothing waits
// for the host between ranges: the number of covered blocks lives in device memory until swin_end.
struct SwinBuild {
    int w[kSwLevels] = {}, n_levels = 0;
    int words = 3;                            // 16-byte words per entry (SwGeo)
    uint32_t max_run = 0, cap_blocks = 0;
    uint32_t min_run = kSwMinRun;             // shortest run covered whole (raised where the rule's coverage would not fit the budget: swin_fit)
    int mode = 0;                             // 0: one range, entries sized by its coverage; 1: ranges behind the upload's slices; 2: ranges of an index in place
    uint64_t done = 0;                        // suffix array indexes below this are dealt with (a multiple of 64, or n)
    uint64_t range_cap = 0;                   // most indexes one range may hold (what the scratch is sized for)
    unsigned long long *d_brk[kSwLevels] = {};
    uint32_t *d_flags = nullptr, *d_incl = nullptr, *d_map = nullptr, *d_used = nullptr;
    void *d_tmp = nullptr, *d_ent = nullptr;
    size_t tmp_bytes = 0;
    unsigned long long *d_starts = nullptr;   // bucket-start bitmap from k_build_ktab (null: read off the finished k-mer table)
    // sliced builds: the entries' memory is allocated by a thread of its own from the moment the index's size is known (a large
    // allocation takes the driver 16 ms per GB and more when another process has just given memory back): ranges whose turn comes
    // before it is there have their entries made later
    std::thread ent_alloc;
    std::atomic<int> ent_state{0};            // 0 not asked for, 1 being allocated, 2 there, 3 failed
    void *ent_mem = nullptr;
    uint64_t filled = 0;                      // entries of the suffix array indexes below this are made
    double t0 = 0;
    void drop_scratch() { for (auto &q : d_brk) { free_dev(q); q = nullptr; } free_dev(d_flags); free_dev(d_incl); free_dev(d_tmp); free_dev(d_starts); d_flags = d_incl = nullptr; d_tmp = nullptr; d_starts = nullptr; }
    ~SwinBuild() { if (ent_alloc.joinable()) ent_alloc.join(); if (ent_mem && ent_mem != d_ent) free_dev(ent_mem); drop_scratch(); free_dev(d_map); free_dev(d_ent); free_dev(d_used); }
    // the entries' memory, asked for ahead of swin_begin: `bytes` on `device`
    void alloc_ahead(int device, uint64_t bytes)
    {
        ent_state = 1;
        ent_alloc = std::thread([this, device, bytes]() {
            void *p = nullptr;
            if (hipSetDevice(device) != hipSuccess || dev_malloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); ent_state = 3; return; }
            ent_mem = p;
            ent_state = 2;
        });
    }
};


// the core lengths reads of maxlen bases are searched with, shortest first (LocateCoreMultiples' CoreLen per phase of AlignReads' schedule):
// the last phase's, then the ones before it; at most kSwLevels of them.  Returns their number
// wave_share (optional): per level, the reads the wave kernel took in the phases that search with that core length, per read of the last
// chunk whose counts have come back (bk_ctx::hist_wave) - what the level's intervals are walked by
int swin_core_lens_k(const bk_ctx *c, uint32_t maxlen, int k, int *w, double *wave_share = nullptr)
{
    const ReadPlan p = make_plan((int)std::max<uint32_t>(maxlen, 1), c->cfg);
    int n = 0;
    for (int ph = p.n_phases - 1; ph >= 0; ph--) {
        int mm, cl, cd;
        phase_params(p, c->cfg, ph, mm, cl, cd);
        cl = std::min(std::max(cl, k), 120);
        if (n == 0 || cl > w[n - 1]) {
            if (n == kSwLevels) break;
            w[n] = cl;
            if (wave_share) wave_share[n] = 0.0;
            n++;
        }
        if (wave_share && ph < kMaxPhases) wave_share[n - 1] += c->hist_wave[ph];
    }
    return n;
}
// skip_auto: (an index of 5-byte elements, whose array is cut to the memory that is free) leave out the shortest core lengths while the
// phases that use them send the wave kernel less than a twentieth of its reads: the last phases' cores select the longest runs of all,
// which the coverage would be spent on first, for the few reads that get that far
int swin_core_lens(const bk_ctx *c, uint32_t maxlen, int *w, bool skip_auto = false)
{
    double share[kSwLevels];
    int n = swin_core_lens_k(c, maxlen, c->ix.k, w, share);
    int skip = c->swin_skip_short > 0 ? std::min(c->swin_skip_short, n - 1) : 0;         // ("swin_skip_short": this many, whatever the phases say)
    if (!skip && skip_auto && c->hist_valid) {
        double total = 0.0, acc = 0.0;
        for (int i = 0; i < n; i++) total += share[i];
        while (skip < n - 1 && acc + share[skip] < 0.05 * total) acc += share[skip++];
    }
    if (skip) { for (int i = skip; i < n; i++) w[i - skip] = w[i]; n -= skip; }
    return n;
}


// sliced: the suffix array arrives in ranges (swin_range per range, entries allocated by the budget up front, bucket starts noted by the
// k-mer table's builder); else ONE swin_range call over the whole array, which allocates what its coverage turned out to need
int swin_begin(bk_ctx *c, SwinBuild &sb, const int *w, int n_levels, int words, uint64_t budget, uint64_t range_cap, int mode, hipStream_t s)
{
    const uint64_t n = c->ix.n;
    const uint64_t n_blocks = (n + (1u << kSwBlkShift) - 1) >> kSwBlkShift;
    sb.words = words;
    const uint64_t block_bytes = (uint64_t)(16 * words) << kSwBlkShift;
    sb.t0 = StageClock::now();
    sb.n_levels = n_levels;
    for (int l = 0; l < n_levels; l++) sb.w[l] = w[l];
    // a run is walked whole when the copy-count check at IterCnt == 100 lets it pass: up to MaxIter + 100-odd suffixes
    sb.max_run = c->cfg.max_iter > 0 ? (uint32_t)c->cfg.max_iter + 256u : 1u << 20;
    sb.cap_blocks = (uint32_t)std::min<uint64_t>(n_blocks, budget / block_bytes);
    if (sb.cap_blocks == 0) return 1;
    sb.range_cap = std::min<uint64_t>(range_cap, n) + 64;
    const uint64_t brk_words = (sb.range_cap >> 6) + 4, blocks = (sb.range_cap >> kSwBlkShift) + 2;
#define SW_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { (void)hipGetLastError(); return e_ == hipErrorOutOfMemory ? BK_ERR_MEM : BK_ERR_INTERNAL; } } while (0)
    for (int l = 0; l < n_levels; l++) SW_TRY(dev_malloc(&sb.d_brk[l], brk_words * 8));
    SW_TRY(dev_malloc(&sb.d_flags, blocks * 4));
    SW_TRY(dev_malloc(&sb.d_incl, blocks * 4));
    SW_TRY(dev_malloc(&sb.d_map, n_blocks * 4));
    SW_TRY(dev_malloc(&sb.d_used, 4));
    SW_TRY(hipMemsetAsync(sb.d_used, 0, 4, s));
    SW_TRY(bk::prim::inclusive_sum(nullptr, sb.tmp_bytes, sb.d_flags, sb.d_incl, (size_t)blocks, s));
    SW_TRY(dev_malloc(&sb.d_tmp, sb.tmp_bytes + 256));
    sb.mode = mode;
    if (mode == 1) {
        if (sb.ent_state == 0) SW_TRY(dev_malloc(&sb.d_ent, (uint64_t)sb.cap_blocks * block_bytes));      // (else: alloc_ahead's thread brings it)
        SW_TRY(dev_malloc(&sb.d_starts, ((n >> 6) + 4) * 8));
        SW_TRY(clear_dev(sb.d_starts, ((n >> 6) + 4) * 8, s));
    }
    return BK_OK;
}

// suffix array indexes below `upto` are in place (with their second-level keys): whole 64-index words of them are dealt with
int swin_range(bk_ctx *c, SwinBuild &sb, const DevIndex &ix, uint64_t upto, hipStream_t s)
{
    const uint64_t n = ix.n;
    while (sb.done < n) {
        uint64_t e = upto >= n ? n : (upto & ~63ULL);
        if (e > sb.done && e - sb.done > sb.range_cap - 64) e = (sb.done + sb.range_cap - 64) & ~63ULL;      // (no more than the scratch holds at a time)
        if (e <= sb.done) break;
        const uint64_t a = sb.done, len = e - a;
        const uint64_t n_words = (len >> 6) + 2, n_blocks = (len + (1u << kSwBlkShift) - 1) >> kSwBlkShift;
        launch_swin_breaks(ix, sb.w, sb.n_levels, sb.d_brk, a, e, n_words, sb.d_starts, s);
        for (int l = 0; l < sb.n_levels; l++) launch_swin_cover(sb.d_brk[l], len, sb.max_run, sb.min_run, sb.d_flags, n_blocks, l == 0, s);
        SW_TRY(hipGetLastError());
        size_t tb = sb.tmp_bytes;
        SW_TRY(bk::prim::inclusive_sum(sb.d_tmp, tb, sb.d_flags, sb.d_incl, (size_t)n_blocks, s));
        launch_swin_map(sb.d_flags, sb.d_incl, n_blocks, sb.cap_blocks, sb.d_used, sb.d_map + (a >> kSwBlkShift), s);
        if (sb.d_ent == nullptr && sb.ent_state == 2) sb.d_ent = sb.ent_mem;
        if (sb.d_ent == nullptr && sb.ent_state == 1) { sb.done = e; continue; }       // (its memory is not there yet: the entries follow)
        if (sb.d_ent == nullptr && sb.ent_state == 3) return BK_ERR_MEM;
        if (sb.d_ent == nullptr) {
            // (the whole array in one range: the entries take what the coverage needs, known now)
            if (a != 0 || e != n) return BK_ERR_INTERNAL;
            uint32_t used = 0;
            SW_TRY(hipMemcpyAsync(&used, sb.d_used, 4, hipMemcpyDeviceToHost, s));
            SW_TRY(hipStreamSynchronize(s));
            if (used == 0) { sb.done = e; return BK_OK; }
            sb.cap_blocks = used;
            SW_TRY(dev_malloc(&sb.d_ent, (uint64_t)used * ((uint64_t)(16 * sb.words) << kSwBlkShift)));
        }
        launch_swin_fill(ix, sb.d_map, sb.d_ent, sb.words, sb.filled, e, s);
        SW_TRY(hipGetLastError());
        sb.filled = e;
        sb.done = e;
    }
    return BK_OK;
}

// An index whose rule covers more than the budget holds (a 17 Gbp genome that is 85 % repeats: a tenth of its suffixes lie in runs of 65
// and more) gets the part of it that is walked most: a run's suffixes are each fetched once per read that meets the run, and a run of R
// copies is met by R times as many reads as a unique place - so the longest runs first.  One pass over the ranges counts the blocks the
// rule would cover for a ladder of shortest-run lengths (the heads of the runs beyond MaxIter are in every one of them); the shortest
// that fits is taken, and the entries are sized by its count.  Mode 2 only (the index is in place and is gone over twice).
int swin_fit(bk_ctx *c, SwinBuild &sb, const DevIndex &ix, hipStream_t s)
{
    static const uint32_t ladder[] = {kSwMinRun, 96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 416, 448, 480, 512, 576, 640, 704, 768, 832, 896, 960, 1024, 1088, 1152, 1216, 1280,
                                      1344, 1408, 1472, 1536, 1664, 1792, 1920, 2048, 2304, 2560, 3072, 4096, 6144, 8192};
    constexpr int NL = (int)(sizeof(ladder) / sizeof(ladder[0]));
    const uint64_t n = ix.n;
    unsigned long long *d_cnt = nullptr;
    SW_TRY(dev_malloc(&d_cnt, NL * 8));
    SW_TRY(hipMemsetAsync(d_cnt, 0, NL * 8, s));
    for (uint64_t a = 0; a < n;) {
        uint64_t e = std::min<uint64_t>(n, (a + sb.range_cap - 64) & ~63ULL);
        const uint64_t len = e - a, n_words = (len >> 6) + 2, n_blocks = (len + (1u << kSwBlkShift) - 1) >> kSwBlkShift;
        launch_swin_breaks(ix, sb.w, sb.n_levels, sb.d_brk, a, e, n_words, nullptr, s);
        for (int q = 0; q < NL; q++) {
            if (ladder[q] > sb.max_run) break;
            for (int l = 0; l < sb.n_levels; l++) launch_swin_cover(sb.d_brk[l], len, sb.max_run, ladder[q], sb.d_flags, n_blocks, l == 0, s);
            launch_count_nonzero(sb.d_flags, n_blocks, d_cnt + q, s);
        }
        SW_TRY(hipGetLastError());
        a = e;
    }
    unsigned long long cnt[NL] = {};
    SW_TRY(hipMemcpyAsync(cnt, d_cnt, NL * 8, hipMemcpyDeviceToHost, s));
    SW_TRY(hipStreamSynchronize(s));
    free_dev(d_cnt);
    int pick = 0;
    while (pick + 1 < NL && ladder[pick + 1] <= sb.max_run && cnt[pick] > sb.cap_blocks) pick++;
    sb.min_run = ladder[pick];
    const uint64_t blocks = std::min<uint64_t>(cnt[pick], sb.cap_blocks);
    if (blocks == 0) return 1;
    sb.cap_blocks = (uint32_t)blocks;
    SW_TRY(dev_malloc(&sb.d_ent, blocks * ((uint64_t)(16 * sb.words) << kSwBlkShift)));
    StageClock clk;
    if (clk.on) fprintf(stderr, "biokanga_amd: window array: runs of %u and more suffixes covered whole (the rule at 65: %.1f %% of the suffix array, this: %.1f %%, room for %.1f %%)\n", sb.min_run,
                        100.0 * (double)cnt[0] * 32.0 / (double)n, 100.0 * (double)cnt[pick] * 32.0 / (double)n, 100.0 * (double)blocks * 32.0 / (double)n);
    return BK_OK;
}

// publishes the array (the context takes the buffers over); 1 = nothing was worth covering
int swin_end(bk_ctx *c, SwinBuild &sb, hipStream_t s)
{
    if (sb.ent_alloc.joinable()) sb.ent_alloc.join();
    if (sb.ent_state == 3) return BK_ERR_MEM;
    if (sb.ent_state == 2 && sb.d_ent == nullptr) sb.d_ent = sb.ent_mem;
    if (sb.d_ent != nullptr && sb.filled < sb.done) {       // (the ranges that came before the entries' memory did)
        launch_swin_fill(c->ix, sb.d_map, sb.d_ent, sb.words, sb.filled, sb.done, s);
        sb.filled = sb.done;
    }
    uint32_t used = 0;
    SW_TRY(hipMemcpyAsync(&used, sb.d_used, 4, hipMemcpyDeviceToHost, s));
    SW_TRY(hipStreamSynchronize(s));
    sb.drop_scratch();
    if (sb.done < c->ix.n) return BK_ERR_INTERNAL;
    if (used == 0) return 1;
    const uint64_t block_bytes = (uint64_t)(16 * sb.words) << kSwBlkShift;
#undef SW_TRY
    const uint64_t n_blocks = (c->ix.n + (1u << kSwBlkShift) - 1) >> kSwBlkShift;
    c->d_swin = sb.d_ent;
    c->d_swmap = sb.d_map;
    sb.d_ent = nullptr;
    sb.ent_mem = nullptr;
    sb.d_map = nullptr;
    c->swin_w = sb.w[0] | (sb.w[sb.n_levels - 1] << 8) | (sb.n_levels << 16) | (sb.words << 24);
    c->ix.sw_words = sb.words;
    c->swin_bytes = (uint64_t)std::max(used, sb.cap_blocks) * block_bytes + n_blocks * 4;      // (what is allocated: a sliced build's entries were sized before its coverage was known)
    c->swin_covered = (double)used / (double)n_blocks;
    c->ix.swin = reinterpret_cast<const uint4 *>(c->d_swin);
    c->ix.swmap = c->d_swmap;
    c->swin_setup_s = StageClock::now() - sb.t0;
    return BK_OK;
}

// most bytes the partial array may take: a third of what every suffix would, half of what is free beyond `reserve`, the caller's cap
uint64_t swin_budget_for(const bk_ctx *c, uint64_t free_b, uint64_t reserve, int words)
{
    // (flags, scan, map, break bitmaps while it is made: of the whole index, or - beyond 2^32 suffixes - of a range of 2^31 and the map)
    const bool ranged = c->ix.n >= (1ULL << 32);
    const uint64_t span = ranged ? (1ULL << 31) : c->ix.n;
    const uint64_t work = ((span >> kSwBlkShift) + 1) * 8 + ((c->ix.n >> kSwBlkShift) + 1) * 4 + (span >> 3) * (kSwLevels + 1) + (64ULL << 20);
    if (free_b < reserve + work + (1ULL << 30)) return 0;
    // (such an index's array is only made when asked for, and its walks are where its time goes: what is free beyond the reserve, not half of it - a batch that then
    // finds no room for its scratch has the array released first, align_chunk)
    uint64_t budget = std::min<uint64_t>(c->ix.n * 16 * (uint64_t)words / 3, ranged ? (free_b - reserve - work) / 16 * 15 : (free_b - reserve - work) / 2);
    if (c->swin_budget) budget = std::min<uint64_t>(budget, c->swin_budget);
    return budget;
}

int maybe_build_swin(bk_ctx *c, uint32_t maxlen, uint32_t nreads, hipStream_t s)
{
    // (an index of 5-byte elements has no inverse suffix array: its wave kernel forms keep the reference's set of seen keys, and take
    // windows from the array all the same)
    const bool wide = c->d_sa_hi != nullptr || c->ix.n >= (1ULL << 32);
    if (!c->use_swin || c->swin_denied || !c->ix.tgt2 || (!wide && !c->ix.isa) || !c->ix.k2 || !c->use_wave) return BK_OK;
    // (.. when asked to - "use_swin" 2, `--window-array on`: making it goes over such an index twice, seconds at 17 Gbp, which a job of
    // BASELINE config 5's size per device does not earn back; the policy's 1 leaves such an index without)
    if (wide && c->use_swin != 2) return BK_OK;
    const bool full = c->use_swin == 3;
    // (.. and from the second batch on: which core lengths are worth covering is read off the batch before, swin_core_lens)
    if (wide && !c->hist_valid && c->swin_skip_short == 0) return BK_OK;
    int w[kSwLevels];
    const int n_levels = swin_core_lens(c, maxlen, w, wide);
    // entries of three 16-byte words for the kernel family of reads of up to 128 bases, of five for the one of up to 256 (SwGeo)
    const int words = maxlen <= 128 ? 3 : 5;
    const int w_key = w[0] | (w[n_levels - 1] << 8) | (n_levels << 16) | (words << 24);
    if (wide && c->d_swin && c->ix.sw_words == words) return BK_OK;      // (such an index's array is kept as it was made: its levels follow a batch's phases, which differ a little every time)
    if (c->d_swin) {
        // (a partial array made for other core lengths is made again ONCE - the eager build's guess of a hundred bases against what the
        // first batch really holds; after that an array of the right entry size is kept whatever the next batch's longest read: coverage
        // never changes a result, and batches of variable-length reads would otherwise drop and rebuild 25 GB every time their longest
        // read crosses a core length)
        if (full == (c->d_swmap == nullptr) && (full ? c->ix.sw_words == words : (c->swin_w == w_key || (c->swin_rebuilt && c->ix.sw_words == words)))) return BK_OK;
        c->swin_rebuilt = true;
        HIP_TRY(hipStreamSynchronize(s));
        free_dev(c->d_swin); free_dev(c->d_swmap);
        c->d_swin = nullptr; c->d_swmap = nullptr; c->ix.swin = nullptr; c->ix.swmap = nullptr; c->swin_bytes = 0;
    }
    // (it serves the register-window kernel families of reads of up to 128 and up to 256 bases: every core of reads of up to 100 / 160
    // bases, the middle cores of longer ones; 2: made whatever the batch)
    if (c->use_swin == 1 && maxlen > 256) return BK_OK;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const uint64_t want = (uint64_t)std::min(nreads, c->chunk_reads) * scratch_bytes_per_read(words_per_read(maxlen), rd2w_for(maxlen), iv_cores_for(c, maxlen));
    const uint64_t have = (uint64_t)c->cap_reads * scratch_bytes_per_read(c->cap_wpr, c->cap_rd2w, c->cap_iv_cores);
    const uint64_t missing = want > have ? want - have : 0;
    const uint64_t reserve = missing * 4 / 3 + (6ULL << 30);                // (the chunk size is set from 3/4 of the free memory)
    StageClock clk;
    const double t0 = StageClock::now();
    if (full) {
        const uint64_t need = ((c->ix.n + 31) & ~31ULL) * 16 * (uint64_t)words;             // (whole blocks of 32 entries: sw_word_at)
        if ((uint64_t)free_b < need + reserve) { c->swin_denied = true; return BK_OK; }      // (asked once)
        if (dev_malloc(&c->d_swin, need) != hipSuccess) { (void)hipGetLastError(); c->d_swin = nullptr; c->swin_denied = true; return BK_OK; }
        launch_build_swin(c->ix, c->d_swin, words, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
        c->swin_bytes = need;
        c->swin_covered = 1.0;
        c->ix.sw_words = words;
    } else {
        const uint64_t budget = swin_budget_for(c, (uint64_t)free_b, reserve, words);
        if (!budget) { c->swin_denied = true; return BK_OK; }
        SwinBuild sb;
        // (an index beyond 2^32 suffixes is gone over in ranges of 2^31 - its break bitmaps would be 2 GB a level otherwise - and twice:
        // what the rule covers there does not fit, swin_fit)
        const bool ranged = c->ix.n >= (1ULL << 32);
        int rb = swin_begin(c, sb, w, n_levels, words, budget, ranged ? (1ULL << 31) : c->ix.n, ranged ? 2 : 0, s);
        if (!rb && ranged) rb = swin_fit(c, sb, c->ix, s);
        if (!rb) rb = swin_range(c, sb, c->ix, c->ix.n, s);
        if (!rb) rb = swin_end(c, sb, s);
        if (rb == BK_ERR_INTERNAL) return rb;
        if (rb) { c->swin_denied = true; return BK_OK; }                   // (no room, or nothing worth covering: asked once)
    }
    c->ix.swin = reinterpret_cast<const uint4 *>(c->d_swin);
    c->ix.swmap = c->d_swmap;
    if (full) c->swin_setup_s = StageClock::now() - t0;
    if (clk.on) fprintf(stderr, "biokanga_amd: window array for %.1f %% of the suffix array (runs sharing %d .. %d bases, %d levels), entries of %d bytes, %.2f GB\n", 100.0 * c->swin_covered, w[0], w[n_levels - 1], n_levels, 16 * words, c->swin_bytes / 1e9);
    clk.lap("suffix-ordered windows");
    return BK_OK;
}


}  // namespace bk

using namespace bk;

extern "C" {

// One rule for the command line, the benchmark and any other caller (include/biokanga_amd.h)
uint32_t bk_image_policy(uint64_t reads_per_device)
{
    return reads_per_device >= BK_POLICY_MIN_READS ? BK_CTX_WINDOW_ARRAY_EAGER : BK_CTX_GROW_IMAGE;
}

int bk_ctx_create(bk_ctx **out, const char *sfx_path, int device_id, const bk_align_params *p)
{
    return bk_ctx_create_ex(out, sfx_path, device_id, p, 0);
}

int bk_ctx_create_ex(bk_ctx **out, const char *sfx_path, int device_id, const bk_align_params *p, uint32_t flags)
{
    if (!sfx_path) return BK_ERR_PARAMS;
    bk_ctx *c = nullptr;
    StageClock clk;
    int rc = new_ctx(out, device_id, p, &c);
    if (rc) return rc;
    clk.lap("HIP runtime + device + stream");
    if (flags & BK_CTX_LEAN_IMAGE) { c->use_ktab2 = 0; c->use_k3 = 0; }          // (BK_CTX_GROW_IMAGE below: the worker makes the table's second words the default way, grow_elem)
    if (flags & BK_CTX_NO_DEEP_KEYS) c->use_k3 = 0;
    if (flags & BK_CTX_GROW_IMAGE) {
        c->use_ktab2 = 0; c->use_k3 = 0; c->grow_enabled = true;
        // (tests: a small run that grows - and, so that it does before it is over, waits for the tables at the batch after the one that started them)
        if (const unsigned long long v = env::grow_after_reads()) { c->grow_after = v; c->grow_wait = true; }
    }
    SfxFile f;
    std::string err;
    rc = sfx_open(sfx_path, f, &err);
    clk.lap("sfx_open");
    if (rc) {
        fprintf(stderr, "biokanga_amd: %s\n", err.c_str());
        bk_ctx_destroy(c);
        return rc;
    }
    c->dataset = f.dataset;
    const bool eager_swin = (flags & BK_CTX_WINDOW_ARRAY_EAGER) && f.el_size == 4 && f.concat_len < (1ULL << 32) && c->use_swin && c->use_swin != 3;
    // The window array's entries, when the caller wants the array from the start: 10 bytes per suffix (a 3.1 Gbp genome with 45 % of
    // repeat-derived bases needs 8), no more than a quarter of the HBM - allocated by a thread of its own while the suffix array crosses
    // PCIe, once every other allocation of the set-up is made (a large allocation holds the driver's lock for as long as it takes)
    SwinBuild sb;
    uint64_t swin_ahead = 0;
    if (eager_swin) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > f.concat_len * 24 + (16ULL << 30)) {
            swin_ahead = std::min<uint64_t>(f.concat_len * 10, total_b / 4);
            if (c->swin_budget) swin_ahead = std::min<uint64_t>(swin_ahead, c->swin_budget);
        } else
            (void)hipGetLastError();
    }
    std::vector<bk_entry_info> ents(f.entries.size());
    for (size_t i = 0; i < ents.size(); i++) {
        ents[i].entry_id = f.entries[i].entry_id;
        ents[i].seq_len = f.entries[i].seq_len;
        ents[i].start_ofs = f.entries[i].start_ofs;
        ents[i].end_ofs = f.entries[i].end_ofs;
        memcpy(ents[i].name, f.entries[i].name, 81);
    }
    // stage the file image through HBM: bases and suffix array as they are on disk
    // (4-byte suffix array elements are stored as the file holds them: they travel straight to where they stay)
    uint8_t *d_seq = nullptr, *d_sa = nullptr;
    const bool sa_in_place = f.el_size == 4;
    auto cleanup = [&]() { free_dev(d_seq); free_dev(d_sa); };
    if (dev_malloc(&d_seq, f.concat_len + 16) != hipSuccess ||
        (sa_in_place ? dev_malloc(&c->d_sa_lo, f.concat_len * 4) : dev_malloc(&d_sa, f.concat_len * f.el_size)) != hipSuccess) {
        cleanup(); bk_ctx_destroy(c); return BK_ERR_MEM;
    }
    clk.lap("device allocations");
    // (read() into the staging buffers, not through the mapping: its pages would be faulted in one by one, and handed back one by one at exit)
    const int fd = ::open(sfx_path, O_RDONLY);
    if (fd < 0) { cleanup(); bk_ctx_destroy(c); return BK_ERR_OPNFILE; }
    const uint64_t seq_ofs = (uint64_t)(f.seq - (const uint8_t *)f.map_base), sa_ofs = (uint64_t)(f.sa - (const uint8_t *)f.map_base);
    bool sent = upload_file(d_seq, fd, seq_ofs, f.concat_len, device_id) == BK_OK;
    clk.lap("upload bases");
    if (sent && sa_in_place) {
        // 4-byte elements: the bases are packed at once, and the suffix array follows in slices - the tables that are one pass over its
        // indexes (k-mer table, second-level keys, inverse suffix array) are made of slice i while slice i + 1 crosses PCIe
        rc = adopt_device_image(c, d_seq, f.concat_len, nullptr, 4);
        free_dev(d_seq);
        d_seq = nullptr;
        TablePlan tp;
        if (!rc) rc = tables_begin(c, tp);
        const uint64_t n = f.concat_len;
        uint64_t n_slices = std::max<uint64_t>(1, std::min<uint64_t>(16, n >> 26));            // (slices of at least 256 MB; what the tables still owe when the last one has arrived is a slice's worth)
        if (const unsigned long long ts = env::table_slices()) n_slices = std::max<uint64_t>(1, std::min<uint64_t>(ts, n));      // (tests: small indexes in several slices)
        // The window array, when the caller wants it from the start, is made behind the slices as well (for reads of a hundred bases: a
        // first batch of another shortest core length makes it again, which costs little): what it reads besides suffix array and keys -
        // entry table, alignment parameters, 2-bit target - is made now instead of after the upload.
        bool swin_sliced = false;
        if (!rc && eager_swin && swin_ahead && tp.ktab && tp.k2 && tp.isa && !c->ktab64 && c->use_wave && c->use_tgt2) {
            rc = setup_entries(c, ents.data(), (uint32_t)ents.size());
            if (!rc) { c->entries_set = true; rc = build_tgt2(c); }
            if (!rc) {
                c->tgt2_built = true;
                clk.lap("entry table, 2-bit target");
                int w[kSwLevels];
                const int n_levels = swin_core_lens_k(c, 100, tp.k, w);
                sb.ent_state = 1;                                   // (its entries come from alloc_ahead's thread, started below)
                swin_sliced = swin_begin(c, sb, w, n_levels, 3, swin_ahead, n / n_slices + 128, 1, c->stream) == BK_OK;
                if (!swin_sliced) { (void)hipGetLastError(); sb.ent_state = 0; }
                else sb.alloc_ahead(device_id, swin_ahead);
            }
        }
        for (uint64_t k = 0; k < n_slices && !rc && sent; k++) {
            // (slices start at multiples of 64 indexes: the k-mer table's builder notes the bucket starts a word of a bitmap at a time)
            const uint64_t i0 = (n * k / n_slices) & ~63ULL, i1 = k + 1 == n_slices ? n : (n * (k + 1) / n_slices) & ~63ULL;
            if (i1 <= i0) continue;
            sent = upload_file(c->d_sa_lo + i0, fd, sa_ofs + i0 * 4, (i1 - i0) * 4, device_id) == BK_OK;
            if (sent) rc = tables_range(c, tp, i0, i1, swin_sliced ? sb.d_starts : nullptr);
            if (sent && !rc && swin_sliced) {
                DevIndex ix = c->ix;
                ix.k = tp.k;
                ix.k2 = c->d_k2;
                if (swin_range(c, sb, ix, i1, c->stream) != BK_OK) { (void)hipGetLastError(); swin_sliced = false; }
            }
        }
        clk.lap("upload suffix array, tables enqueued behind its slices");
        if (!rc && sent) rc = tables_end(c, tp);
        clk.lap("tables finished");
        if (!rc && sent && swin_sliced && c->ix.k2 != nullptr) {
            const int re = swin_end(c, sb, c->stream);
            if (re == BK_ERR_INTERNAL) rc = re;
            clk.lap("window array finished");
            if (clk.on && c->d_swin) fprintf(stderr, "biokanga_amd: window array for %.1f %% of the suffix array, %.2f GB, made behind the upload\n", 100.0 * c->swin_covered, c->swin_bytes / 1e9);
        }
    } else if (sent) {
        sent = upload_file(d_sa, fd, sa_ofs, f.concat_len * f.el_size, device_id) == BK_OK;
        clk.lap("upload suffix array");
        if (sent) rc = adopt_device_image(c, d_seq, f.concat_len, d_sa, (int)f.el_size);
        clk.lap("pack target, adopt");
    }
    ::close(fd);
    cleanup();
    if (!sent) { bk_ctx_destroy(c); return BK_ERR_INTERNAL; }
    if (rc) { bk_ctx_destroy(c); return rc; }
    rc = finish_ctx(c, ents.data(), (uint32_t)ents.size());
    if (rc) { bk_ctx_destroy(c); return rc; }
    clk.lap("(rest of bk_ctx_create)");
    if (eager_swin) {
        // (made for reads of a hundred bases; a first batch of another shortest core length makes it again, which costs little)
        rc = maybe_build_swin(c, 100, c->chunk_reads, c->stream);
        if (rc) { bk_ctx_destroy(c); return rc; }
    }
    *out = c;
    return BK_OK;
}

int bk_ctx_create_from_device(bk_ctx **out, const void *d_seq, uint64_t concat_len, const void *d_sa, int sfx_el_size,
                              const bk_entry_info *entries, uint32_t n_entries, int device_id, const bk_align_params *p)
{
    if (!d_seq || !d_sa || !entries || !n_entries || !concat_len || (sfx_el_size != 4 && sfx_el_size != 5)) return BK_ERR_PARAMS;
    bk_ctx *c = nullptr;
    int rc = new_ctx(out, device_id, p, &c);
    if (rc) return rc;
    c->dataset = "device";
    rc = adopt_device_image(c, (const uint8_t *)d_seq, concat_len, (const uint8_t *)d_sa, sfx_el_size);
    if (!rc) rc = finish_ctx(c, entries, n_entries);
    if (rc) { bk_ctx_destroy(c); return rc; }
    *out = c;
    return BK_OK;
}

// The finished index image of `src` (packed target, suffix array, k-mer table, second-level keys, inverse suffix array, 2-bit target
// copies) copied device to device - over xGMI between two GPUs - instead of loading the .sfx again over PCIe and rebuilding every
// table: what `biokanga align --devices` does for the second and later GPUs.
int bk_ctx_clone(bk_ctx **out, const bk_ctx *src, int device_id)
{
    if (!src) return BK_ERR_PARAMS;
    bk_ctx *c = nullptr;
    int rc = new_ctx(out, device_id, &src->params, &c);
    if (rc) return rc;
    StageClock clk;
    c->dataset = src->dataset;
    c->el_size = src->el_size;
    c->n_tgt4_words = src->n_tgt4_words;
    c->sort_shift = src->sort_shift;
    c->ktab64 = src->ktab64;
    c->ktab_wide = src->ktab_wide;
    c->ktab_is2 = src->ktab_is2;
    c->use_ktab2 = src->use_ktab2;
    c->grow_elem = src->grow_elem;
    c->ix.ktab2_elem = src->ix.ktab2_elem;
    c->use_k3 = src->use_k3;
    c->sort_lists = src->sort_lists; c->sort_lists_set = src->sort_lists_set;
    c->grow_enabled = src->grow_enabled && src->grow_state.load() != 4; c->grow_after = src->grow_after; c->grow_wait = src->grow_wait;       // (a clone of a grown context has what it grew)
    c->ktab_bytes = src->ktab_bytes;
    c->nflag_bytes = src->nflag_bytes;
    c->use_ktab = src->use_ktab; c->k_req = src->k_req; c->use_k2 = src->use_k2; c->use_isa = src->use_isa; c->use_wave = src->use_wave; c->use_tgt2 = src->use_tgt2;
    c->ix.n = src->ix.n;
    c->ix.k = src->ix.k;
    c->ix.flag_shift = src->ix.flag_shift;
    c->ix.nflag_bytes = src->ix.nflag_bytes;
    const uint64_t n = src->ix.n;
    const uint64_t nblocks = src->n_tgt4_words / 4;
    bool ok = true;
    auto dup = [&](auto *&dst, const auto *from, size_t bytes) {
        if (!ok || !from) return;
        void *p = nullptr;
        if (dev_malloc(&p, bytes) != hipSuccess) { ok = false; rc = BK_ERR_MEM; return; }
        dst = static_cast<std::remove_reference_t<decltype(dst)>>(p);
        hipError_t e = c->device == src->device ? hipMemcpyAsync(p, from, bytes, hipMemcpyDeviceToDevice, c->stream)
                                                : hipMemcpyPeerAsync(p, c->device, from, src->device, bytes, c->stream);
        if (e != hipSuccess) { ok = false; rc = BK_ERR_INTERNAL; }
    };
    dup(c->d_tgt4, src->d_tgt4, (size_t)src->n_tgt4_words * 8);
    dup(c->d_sa_lo, src->d_sa_lo, (size_t)n * 4);
    dup(c->d_sa_hi, src->d_sa_hi, (size_t)n);
    dup(c->d_ktab, (const uint8_t *)src->d_ktab, src->ktab_bytes);
    dup(c->d_ktab_hi, src->d_ktab_hi, (size_t)(((((1ULL << (2 * src->ix.k)) + 1) >> 16) + 2) * 8));
    dup(c->d_k2, src->d_k2, (size_t)k2s_start(n, kK2Levels + 1) * 4);
    for (int i = 0; i < kMoreKeys; i++) dup(c->d_kx[i], src->d_kx[i], (size_t)k2s_start(n, kK2Levels + 1) * 4);
    dup(c->d_isa, src->d_isa, (size_t)n * 4);
    dup(c->d_tgt2, src->d_tgt2, (size_t)nblocks * 16 + 64);
    dup(c->d_tgt2s, src->d_tgt2s, (size_t)nblocks * 16 + 64);
    dup(c->d_nflag, src->d_nflag, src->nflag_bytes);
    c->use_swin = src->use_swin;                                      // (the window array is built here when the first batch asks for it)
    if (ok && hipStreamSynchronize(c->stream) != hipSuccess) { ok = false; rc = BK_ERR_INTERNAL; }
    if (!ok) { bk_ctx_destroy(c); return rc; }
    c->ix.tgt4 = c->d_tgt4; c->ix.sa_lo = c->d_sa_lo; c->ix.sa_hi = c->d_sa_hi;
    if (c->d_ktab) {
        if (c->ktab64 && c->d_ktab_hi) { c->ix.ktab32 = (const uint32_t *)c->d_ktab; c->ix.ktab_hi = c->d_ktab_hi; }
        else if (c->ktab64) c->ix.ktab64 = (const uint64_t *)c->d_ktab;
        else if (c->ktab_is2) c->ix.ktab2 = (const uint2 *)c->d_ktab;
        else c->ix.ktab32 = (const uint32_t *)c->d_ktab;
    }
    for (int i = 0; i < kMoreKeys; i++) c->ix.kx[i] = c->d_kx[i];
    c->ix.k2 = c->d_k2; c->ix.isa = c->d_isa; c->ix.tgt2 = c->d_tgt2; c->ix.tgt2s = c->d_tgt2s; c->ix.nflag = c->d_nflag;
    clk.lap("index image copied from the first device");
    rc = setup_entries(c, src->entries.data(), (uint32_t)src->entries.size());
    if (rc) { bk_ctx_destroy(c); return rc; }
    *out = c;
    return BK_OK;
}

void bk_ctx_destroy(bk_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    grow_drop(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_dev(c->d_tgt4); free_dev(c->d_sa_lo); free_dev(c->d_sa_hi);
    free_dev(c->d_snp_planes); free_dev(c->d_snp_tot); free_dev(c->d_snp_sites); free_dev(c->d_ent_start); free_dev(c->d_ent_end); free_dev(c->d_ent_id); free_dev(c->d_id2idx); free_dev(c->d_ktab); free_dev(c->d_ktab_hi); free_dev(c->d_k2); free_dev(c->d_kx[0]); free_dev(c->d_kx[1]); free_dev(c->d_slist); free_dev(c->d_slist_stage); free_dev(c->d_sort[0]); free_dev(c->d_sort[1]); free_dev(c->d_sort[2]); free_dev(c->d_sort_tmp); free_dev(c->d_tgt2); free_dev(c->d_tgt2s); free_dev(c->d_nflag); free_dev(c->d_rd2); free_dev(c->d_rmeta);
    free_dev(c->d_rd4); free_dev(c->d_iv_first); free_dev(c->d_iv_n); free_dev(c->d_iv2);
    free_dev(c->d_act[0]); free_dev(c->d_act[1]); free_dev(c->d_heavy); free_dev(c->d_wave); free_dev(c->d_iv32); free_dev(c->d_wave_work); free_dev(c->d_small);
    for (int i = 0; i < 3; i++) free_dev(c->d_stage[i]);
    free_dev(c->d_stripe_cnt);
    free_dev(c->d_isa); free_dev(c->d_swin); free_dev(c->d_swmap); free_dev(c->d_seg2); free_dev(c->d_seq_global);
    free_dev(c->d_seq_counts); free_dev(c->d_ctr); free_dev(c->hs.htab); free_dev(c->hs.slot_epoch);
    free_dev(c->d_in_bases); free_dev(c->d_in_offs); free_dev(c->d_in_lens); free_dev(c->d_in_out);
    free_dev(c->d_in_words); free_dev(c->d_in_lens16); free_dev(c->d_in_exc); free_dev(c->d_scan_tmp); free_dev(c->d_ctr_aux);
    if (c->h_small) (void)hipHostFree(c->h_small);
    free_dev(c->d_ctl);
    free_dev(c->d_chrom_accept);
    for (void *&t : c->sam_text) if (t) { (void)hipHostFree(t); t = nullptr; }
    if (c->h_ctl) (void)hipHostFree(c->h_ctl);
    if (c->ev_ctl) (void)hipEventDestroy(c->ev_ctl);
    if (c->ev_wait) (void)hipEventDestroy(c->ev_wait);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    bk::live_contexts()--;
    delete c;
}


}  // extern "C"
