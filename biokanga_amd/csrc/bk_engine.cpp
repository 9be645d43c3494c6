// bk_engine.cpp - implementation of the C ABI in include/biokanga_amd.h: context (index image in
// HBM), batch driver (phase loop of CSfxArrayV3::AlignReads over whole batches), counters, timing.
// Compiled with hipcc; device code lives in the kernel files (bk_index / bk_prep / bk_search / bk_extend / bk_wave / bk_heavy / bk_rescue / bk_snp .hip).  No CPU fallback exists: every compute
// entry point needs a HIP device and fails with BK_ERR_NODEVICE otherwise.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>
#include "bk_prim.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <atomic>
#include <thread>
#include <type_traits>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "bk_ctx_int.h"
#include "bk_cpus.h"
#include "bk_wait.h"
#include "sfx_file.h"

namespace bk {
// launchers defined in the kernel files (*.hip)
void launch_pack_target(const uint8_t *seq, uint64_t n, uint64_t *tgt4, uint64_t nwords, hipStream_t s);
void launch_pack_target2(const uint64_t *tgt4, uint64_t nwords4, uint64_t *tgt2, unsigned int *nflag32, int flag_shift, hipStream_t s);
void launch_split_sa5(const uint8_t *sa5, uint64_t n, uint32_t *lo, uint8_t *hi, hipStream_t s);
void launch_build_ktab(const DevIndex &ix, void *tab, int k, bool tab64, hipStream_t s, uint64_t i0 = 0, uint64_t i1 = 0, unsigned long long *starts = nullptr, bool pairs = false);
void launch_fill_ktab2_y(void *tab2, const uint32_t *k2, uint64_t n_entries, hipStream_t s);
void launch_max_len(const uint32_t *lens, uint32_t n, uint32_t *out, hipStream_t s);
void launch_widen_lens(const uint16_t *lens16, uint32_t n, uint32_t *lens32, unsigned long long *nwords, hipStream_t s);
void launch_check_exc(const bk_nbase *exc, uint64_t n_exc, const uint32_t *lens, uint32_t n_reads, uint32_t *bad, hipStream_t s);
void launch_packed_extent(const uint64_t *offs, const uint32_t *lens, uint32_t n, unsigned long long *out, hipStream_t s);
void launch_snp_pileup(const DevIndex &ix, const uint8_t *bases, const uint64_t *offs, const uint32_t *id2idx, const bk_snp_aln *alns, uint64_t n_alns,
                       uint32_t *planes, hipStream_t s);
void launch_snp_gather(const DevIndex &ix, const uint32_t *planes, uint64_t g0, uint32_t n, uint32_t *out, hipStream_t s);
void launch_snp_centroids(const DevIndex &ix, const uint32_t *planes, uint64_t g0, uint32_t chrom_len, uint32_t min_reads, uint32_t *hist, hipStream_t s);
void launch_snp_sites(const DevIndex &ix, const uint32_t *planes, uint64_t g0, uint32_t chrom_len, uint32_t min_reads, double min_prop,
                      bk_snp_site *sites, uint32_t cap, uint32_t *n_sites, unsigned long long *totals, hipStream_t s);
void launch_count_seqs(const bk_hit *out, uint32_t n, const uint32_t *id2idx, uint32_t n_ent, unsigned long long *counts, hipStream_t s);
void launch_fill_u64(unsigned long long *p, uint64_t n, unsigned long long v, hipStream_t s);
void launch_prep(const DevAlignCfg &cfg, const DevBatch &b, uint32_t *act, uint32_t *act_cnt, uint32_t *cmax, uint32_t *stage,
                 uint32_t *stripe_cnt, hipStream_t s);
void launch_search(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, uint32_t n_act,
                   int phase, int cmax, int nstr, int lazy, hipStream_t s);
void launch_extend(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, uint32_t n_act,
                   int phase, uint32_t *next_act, uint32_t *next_cnt, uint32_t *heavy, uint32_t *heavy_cnt,
                   uint32_t *cmax_next, hipStream_t s);
void launch_build_isa(const uint32_t *sa, uint64_t n, uint32_t *isa, hipStream_t s, uint64_t i0 = 0, uint64_t i1 = 0);
void launch_build_swin(const DevIndex &ix, void *swin, int words, hipStream_t s);
void launch_swin_breaks(const DevIndex &ix, const int *w, int n_levels, unsigned long long *const *brk, uint64_t a, uint64_t e, uint64_t n_words,
                        const unsigned long long *starts, hipStream_t s);
void launch_swin_cover(const unsigned long long *brk, uint64_t n, uint32_t max_run, uint32_t *flags, uint64_t n_blocks, int first_level, hipStream_t s);
void launch_swin_map(const uint32_t *flags, const uint32_t *incl, uint64_t n_blocks, uint32_t cap_blocks, uint32_t *used, uint32_t *map, hipStream_t s);
void launch_swin_fill(const DevIndex &ix, const uint32_t *map, void *swin, int words, uint64_t a, uint64_t e, hipStream_t s);
void launch_build_k2(const DevIndex &ix, uint32_t *k2, uint32_t *k3, uint32_t *k4, unsigned long long *bad, hipStream_t s, uint64_t i0 = 0, uint64_t i1 = 0, bool write_k2 = true);
void launch_build_k2_levels(uint32_t *k2, uint64_t n, hipStream_t s);
void launch_make_ktab2(const uint32_t *tab, const uint32_t *k2, uint64_t n_entries, uint64_t n, void *out, hipStream_t s);
void launch_search_a(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, const uint32_t *p_n_act, uint32_t n_act_bound,
                     int phase, int cmax, int nstr, int lazy, uint32_t *list, uint32_t *list_cnt, uint32_t *stage, uint32_t *stripe_cnt,
                     hipStream_t s);
void launch_search_b(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, int phase, int lazy, const uint32_t *list, const uint32_t *sorted,
                     uint32_t n_sorted, const uint32_t *p_n_list, uint64_t n_bound, hipStream_t s);
void launch_clear_iv(const DevBatch &b, const uint32_t *p_n_act, uint32_t n_act_bound, int cmax, int st0, int st1, hipStream_t s);
void launch_pe(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, int pe_mode, int min_len, int max_len, int pair_strand,
               bk_hit *hits, uint32_t n_pairs, uint32_t *orphans, uint32_t *counters, uint32_t *h_count, bk_seg2 *seg2, int min_chim,
               int long_reads, const uint8_t *accept, uint32_t n_accept, hipStream_t s);
void launch_flat(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const uint32_t *act, const uint32_t *p_n_act, uint32_t n_act_bound, int phase,
                 int slots_max, uint32_t *next_act, uint32_t *next_cnt, uint32_t *heavy, uint32_t *heavy_cnt, uint32_t *wave,
                 uint32_t *wave_cnt, uint32_t *cmax_next, uint32_t *const *stage, uint32_t *stripe_cnt, int nw, hipStream_t s);
void launch_wave(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const HeavyScratch &hs, const uint32_t *list, const uint32_t *sorted,
                 uint32_t n_sorted, const uint32_t *p_n_list, uint32_t n_bound, int phase, uint32_t *cursor, uint32_t *next_act, uint32_t *next_cnt,
                 uint32_t *cmax_next, int nw, uint32_t max_waves, hipStream_t s);
void launch_heavy(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const HeavyScratch &hs, const uint32_t *list,
                  uint32_t n_list, int phase, uint32_t *cursor, uint32_t *next_act, uint32_t *next_cnt, uint32_t *cmax_next,
                  hipStream_t s);
int build_sa_device(const uint8_t *d_seq, uint64_t n, void *d_sa_out, int el_size, hipStream_t s);
int scan_counts_u64(const unsigned long long *in, unsigned long long *out, uint32_t n, void *tmp, size_t *tmp_bytes, hipStream_t s);
void launch_loci_count(const bk_hit *out, uint32_t n, int clamp_to, unsigned long long *cnt, hipStream_t s);
void launch_best(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const HeavyScratch &hs, const uint32_t *list,
                 uint32_t n_list, uint32_t *cursor, unsigned long long *cnt, bk_loci *dense, hipStream_t s);
void launch_indel(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, uint32_t n, int max_indel, int max_junct, int keep_state, uint32_t *list,
                  uint32_t *list_cnt_dev, uint32_t *list_cnt_host, uint32_t *cursor, bk_seg2 *seg2, hipStream_t s);
void launch_unaligned_list(const bk_hit *out, uint32_t n, uint32_t *list, uint32_t *cnt, hipStream_t s);
void launch_chimeric(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const HeavyScratch &hs, const uint32_t *list, uint32_t n_list,
                     int min_pct, int long_reads, uint32_t *cursor, bk_seg2 *seg2, hipStream_t s);
void launch_loci_compact(const bk_loci *dense, uint32_t width, const unsigned long long *offs, uint32_t n, bk_loci *out, hipStream_t s);
void launch_loci_single(const bk_hit *out, uint32_t n, const unsigned long long *offs, bk_loci *loci, uint32_t *list, uint32_t *list_cnt,
                        const bk_seg2 *seg2, bk_loci_trims *trims, hipStream_t s);
void launch_loci_enum(const DevIndex &ix, const DevAlignCfg &cfg, const DevBatch &b, const HeavyScratch &hs, const uint32_t *list,
                      uint32_t n_list, uint32_t *cursor, const unsigned long long *offs, bk_loci *loci, uint32_t *err, int min_pct, int long_reads,
                      bk_seg2 *seg2, bk_loci_trims *trims, hipStream_t s);
int sort_list_by_key(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n,
                     void *tmp, size_t *tmp_bytes, hipStream_t s);
void launch_keys_search(const DevBatch &b, const uint32_t *list, const uint32_t *p_n, uint32_t n_sort, int shift, uint32_t *keys, hipStream_t s);
void launch_keys_wave(const DevAlignCfg &cfg, const DevBatch &b, int phase, const uint32_t *list, const uint32_t *p_n, uint32_t n_sort, int shift, uint32_t *keys,
                      const uint32_t *work_of, hipStream_t s);
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
    static const int poison = getenv("BK_POISON") ? atoi(getenv("BK_POISON")) : -1;
    static const bool timing = getenv("BK_TIMING") != nullptr;
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

}  // namespace bk

using namespace bk;

namespace {

// BK_TIMING=1: wall-clock of the set-up stages on stderr
struct StageClock {
    bool on = getenv("BK_TIMING") != nullptr;
    double t0 = now();
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    static double wall() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return (double)(ts.tv_sec % 60) + 1e-9 * (double)ts.tv_nsec; }      // (a log's seconds)
    void lap(const char *what) { if (!on) return; const double t = now(); fprintf(stderr, "bk timing: %-28s %7.1f ms   (done at :%06.3f)\n", what, 1e3 * (t - t0), wall()); t0 = t; }
};

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
    static const bool timing = getenv("BK_TIMING") != nullptr;
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
struct TablePlan {
    bool ktab = false, k2 = false, isa = false;
    bool ktab2 = false;                            // the k-mer table's entries are pairs {bucket start, y} from the start (DevIndex::ktab2): starts written in place, y filled in tables_end
    int kx = 0;                                    // key arrays behind the second-level keys (DevIndex::kx)
    int k = 0;
    unsigned long long *d_bad = nullptr;           // places where the second-level keys are not in order inside a bucket; the third-level keys inside a run of equal second-level keys
    ~TablePlan() { free_dev(d_bad); }
};

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
                launch_make_ktab2(ix.ktab32, ix.k2, n_entries, n, c->grow_ktab2, s);
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
            c->use_ktab2 = 1;
            c->ix.ktab32 = nullptr;
            c->ix.ktab2 = reinterpret_cast<const uint2 *>(c->d_ktab);
        }
        if (!c->sort_lists_set) c->sort_lists = (c->sort_lists & ~1) | (c->ix.kx[0] == nullptr ? 1 : 0);
        if (getenv("BK_TIMING"))
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
void grow_tick(bk_ctx *c, uint64_t nreads, bool now = false)
{
    if (!c->grow_enabled) return;
    const int st = c->grow_state.load(std::memory_order_acquire);
    if (st != 0 && st != 4) c->grow_seen += nreads;
    if (st == 0) {
        c->grow_seen += nreads;
        if ((now || c->grow_seen >= c->grow_after) && c->tables_built && c->ix.k2 != nullptr) {
            c->grow_ix = c->ix;
            c->grow_want_ktab2 = !c->ktab64 && !c->ktab_is2;
            c->grow_state.store(1);
            c->grow_thread = std::thread(grow_worker, c);
        }
    } else if (st == 2 || st == 3 || (st == 1 && c->grow_wait))
        grow_take_in(c);
}

int tables_begin(bk_ctx *c, TablePlan &tp)
{
    grow_drop(c);
    free_dev(c->d_ktab); free_dev(c->d_k2); free_dev(c->d_isa);
    c->d_ktab = nullptr; c->d_k2 = nullptr; c->d_isa = nullptr;
    for (int i = 0; i < kMoreKeys; i++) { free_dev(c->d_kx[i]); c->d_kx[i] = nullptr; c->ix.kx[i] = nullptr; }
    c->ix.ktab32 = nullptr; c->ix.ktab64 = nullptr; c->ix.ktab2 = nullptr; c->ix.k2 = nullptr; c->ix.isa = nullptr;
    c->ktab_is2 = false;
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
        c->ktab64 = c->ix.n >= (1ULL << 32);
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
int tables_range(bk_ctx *c, const TablePlan &tp, uint64_t i0, uint64_t i1, unsigned long long *bucket_starts = nullptr)
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
    if (tp.ktab) {
        if (c->ktab64) c->ix.ktab64 = (const uint64_t *)c->d_ktab;
        else if (tp.ktab2) { c->ix.ktab2 = reinterpret_cast<const uint2 *>(c->d_ktab); c->ktab_is2 = true; }
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
        launch_fill_ktab2_y(c->d_ktab, c->d_k2, (1ULL << (2 * tp.k)) + 1, c->stream);
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

int size_heavy_scratch(bk_ctx *c)
{
    // worst-case inserts per strand pass = min(node cap, cores(max_read_len) * MaxIter)
    int slides = std::max(1, (c->cfg.slides_per100 * c->max_read_len + 99) / 100);
    uint64_t worst = c->cfg.max_iter ? (uint64_t)slides * (uint64_t)c->cfg.max_iter : kNodeCap;
    if (worst > kNodeCap) worst = kNodeCap;
    uint32_t ts = 1024;
    while (ts < 2 * worst) ts <<= 1;
    uint32_t slots = 8192;                  // one per resident wave of the wave kernels when they fit in 16 GB
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
    c->debug = getenv("BK_DEBUG") != nullptr;
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

int ensure_batch_scratch(bk_ctx *c, uint32_t n_reads, uint32_t wpr, uint32_t rd2w = 0, uint32_t iv_cores = kMaxCoresFast)
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
static int ensure_sort_scratch(bk_ctx *c, uint32_t n, hipStream_t s)
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
static int sort_work(bk_ctx *c, const uint32_t *list, uint32_t n, hipStream_t s, const uint32_t **out)
{
    size_t tb = c->sort_tmp_bytes;
    if (sort_list_by_key(c->d_sort[0], c->d_sort[1], list, c->d_sort[2], n, c->d_sort_tmp, &tb, s)) return BK_ERR_INTERNAL;
    *out = c->d_sort[2];
    return BK_OK;
}

static inline uint32_t words_per_read(uint32_t maxlen)
{
    return ((maxlen + 15) / 16 + 2) & ~1u;      // even: every packed row starts 16-byte aligned
}

// per-read bytes of batch scratch (packed fwd+revcomp rows, core intervals, work lists)
// the most cores per strand a read of up to maxlen bases can have (LocateCoreMultiples' MaxNumSlides), capped at what the
// interval-slot kernels take
static inline uint32_t iv_cores_for(const bk_ctx *c, uint32_t maxlen)
{
    const uint32_t ms = std::max(1u, ((uint32_t)c->cfg.slides_per100 * maxlen + 99) / 100);
    return std::min<uint32_t>(ms, kMaxCoresFast);
}

// 64-bit words of a read's 2 bit/base row in the register-window kernel family that takes reads of up to maxlen bases
static inline uint32_t rd2w_for(uint32_t maxlen)
{
    return maxlen <= 128 ? 4u : (maxlen <= 256 ? 8u : (maxlen <= 16u * (uint32_t)kNwLong ? (uint32_t)kNwLong / 2 : (uint32_t)kNwLongest / 2));
}

// per-read bytes of batch scratch: packed rows in both forms, interval records, work lists (reads, search items and their striped
// forms), sort buffers
static inline uint64_t scratch_bytes_per_read(uint32_t wpr, uint32_t rd2w = 8, uint32_t iv_cores = kMaxCoresFast)
{
    return 2ULL * wpr * 8 + 2ULL * rd2w * 8 + 2ULL * iv_cores * (12 + 8) + 52 + 24;
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

// The suffix-ordered window array (DevIndex::swin, 48 bytes per suffix it holds) is built when the first batch it can serve arrives - reads
// of up to kSwLen bases whose core offsets stay within kSwPre - for the part of the suffix array the wave kernel's long walks visit
// (bk_index.hip, k_swin_cover: a tenth of a 3.1 Gbp index), within a budget of the HBM that is free next to this batch's scratch.
}  // (anonymous)

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

namespace {

// the core lengths reads of maxlen bases are searched with, shortest first (LocateCoreMultiples' CoreLen per phase of AlignReads' schedule):
// the last phase's, then the ones before it; at most kSwLevels of them.  Returns their number
int swin_core_lens_k(const bk_ctx *c, uint32_t maxlen, int k, int *w)
{
    const ReadPlan p = make_plan((int)std::max<uint32_t>(maxlen, 1), c->cfg);
    int n = 0;
    for (int ph = p.n_phases - 1; ph >= 0 && n < kSwLevels; ph--) {
        int mm, cl, cd;
        phase_params(p, c->cfg, ph, mm, cl, cd);
        cl = std::min(std::max(cl, k), 120);
        if (n == 0 || cl > w[n - 1]) w[n++] = cl;
    }
    return n;
}
int swin_core_lens(const bk_ctx *c, uint32_t maxlen, int *w) { return swin_core_lens_k(c, maxlen, c->ix.k, w); }

// The partial array is made range by range of the suffix array (one range when the index is already there; behind the slices of the
// suffix array's upload when it is still arriving - bk_ctx_create_ex): per level (core length) the break bitmap of the runs of suffixes
// sharing that many bases and the per-block coverage it implies, block numbers by a scan that continues the ranges before, then the
// entries - no more of them than `budget` bytes hold (blocks beyond it stay uncovered: coverage never changes a result).  Nothing waits
// for the host between ranges: the number of covered blocks lives in device memory until swin_end.
struct SwinBuild {
    int w[kSwLevels] = {}, n_levels = 0;
    int words = 3;                            // 16-byte words per entry (SwGeo)
    uint32_t max_run = 0, cap_blocks = 0;
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

// sliced: the suffix array arrives in ranges (swin_range per range, entries allocated by the budget up front, bucket starts noted by the
// k-mer table's builder); else ONE swin_range call over the whole array, which allocates what its coverage turned out to need
int swin_begin(bk_ctx *c, SwinBuild &sb, const int *w, int n_levels, int words, uint64_t budget, uint64_t range_cap, bool sliced, hipStream_t s)
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
    if (sliced) {
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
        for (int l = 0; l < sb.n_levels; l++) launch_swin_cover(sb.d_brk[l], len, sb.max_run, sb.d_flags, n_blocks, l == 0, s);
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
uint64_t swin_budget_for(const bk_ctx *c, uint64_t free_b, uint64_t reserve, int words = 3)
{
    const uint64_t work = ((c->ix.n >> kSwBlkShift) + 1) * 12 + (c->ix.n >> 3) * (kSwLevels + 1) + (64ULL << 20);      // (flags, scan, map, break bitmaps while it is made)
    if (free_b < reserve + work + (1ULL << 30)) return 0;
    uint64_t budget = std::min<uint64_t>(c->ix.n * 16 * (uint64_t)words / 3, (free_b - reserve - work) / 2);
    if (c->swin_budget) budget = std::min<uint64_t>(budget, c->swin_budget);
    return budget;
}

int maybe_build_swin(bk_ctx *c, uint32_t maxlen, uint32_t nreads, hipStream_t s)
{
    if (!c->use_swin || c->swin_denied || c->d_sa_hi || c->ix.n >= (1ULL << 32) || !c->ix.tgt2 || !c->ix.isa || !c->ix.k2 || !c->use_wave) return BK_OK;
    const bool full = c->use_swin == 3;
    int w[kSwLevels];
    const int n_levels = swin_core_lens(c, maxlen, w);
    // entries of three 16-byte words for the kernel family of reads of up to 128 bases, of five for the one of up to 256 (SwGeo)
    const int words = maxlen <= 128 ? 3 : 5;
    const int w_key = w[0] | (w[n_levels - 1] << 8) | (n_levels << 16) | (words << 24);
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
        int rb = swin_begin(c, sb, w, n_levels, words, budget, c->ix.n, false, s);
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
int align_device(bk_ctx *c, const DevReads &in, uint32_t nreads, bk_hit *d_out, hipStream_t s, uint32_t maxlen_known = 0, bool enqueue_only = false)
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

}  // namespace

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
    if (flags & BK_CTX_LEAN_IMAGE) { c->use_ktab2 = 0; c->use_k3 = 0; }
    if (flags & BK_CTX_NO_DEEP_KEYS) c->use_k3 = 0;
    if (flags & BK_CTX_GROW_IMAGE) {
        c->use_ktab2 = 0; c->use_k3 = 0; c->grow_enabled = true;
        // (tests: a small run that grows - and, so that it does before it is over, waits for the tables at the batch after the one that started them)
        if (const char *e = getenv("BK_GROW_AFTER_READS")) { const unsigned long long v = strtoull(e, nullptr, 10); if (v) { c->grow_after = v; c->grow_wait = true; } }
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
        if (const char *e = getenv("BK_TABLE_SLICES")) n_slices = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)atoi(e), n));      // (tests: small indexes in several slices)
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
                swin_sliced = swin_begin(c, sb, w, n_levels, 3, swin_ahead, n / n_slices + 128, true, c->stream) == BK_OK;
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
    c->ktab_is2 = src->ktab_is2;
    c->use_ktab2 = src->use_ktab2;
    c->use_k3 = src->use_k3;
    c->sort_lists = src->sort_lists; c->sort_lists_set = src->sort_lists_set;
    c->grow_enabled = src->grow_enabled && src->grow_state.load() != 4; c->grow_after = src->grow_after; c->grow_wait = src->grow_wait;       // (a clone of a grown context has what it grew)
    c->ktab_bytes = src->ktab_bytes;
    c->nflag_bytes = src->nflag_bytes;
    c->use_ktab = src->use_ktab; c->k_req = src->k_req; c->use_k2 = src->use_k2; c->use_isa = src->use_isa; c->use_wave = src->use_wave; c->use_tgt2 = src->use_tgt2;
    c->ix.n = src->ix.n;
    c->ix.k = src->ix.k;
    c->ix.flag_shift = src->ix.flag_shift;
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
        if (c->ktab64) c->ix.ktab64 = (const uint64_t *)c->d_ktab;
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
    free_dev(c->d_snp_planes); free_dev(c->d_snp_tot); free_dev(c->d_snp_sites); free_dev(c->d_ent_start); free_dev(c->d_ent_end); free_dev(c->d_ent_id); free_dev(c->d_id2idx); free_dev(c->d_ktab); free_dev(c->d_k2); free_dev(c->d_kx[0]); free_dev(c->d_kx[1]); free_dev(c->d_slist); free_dev(c->d_slist_stage); free_dev(c->d_sort[0]); free_dev(c->d_sort[1]); free_dev(c->d_sort[2]); free_dev(c->d_sort_tmp); free_dev(c->d_tgt2); free_dev(c->d_tgt2s); free_dev(c->d_nflag); free_dev(c->d_rd2); free_dev(c->d_rmeta);
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
    if (n == "use_k3") {                   // how many key arrays behind the second-level keys (rebuilt with the tables)
        int64_t old = c->use_k3;
        c->use_k3 = value < 0 ? 0 : (value > kMoreKeys ? kMoreKeys : (int)value);
        int rc = build_tables(c);
        return rc ? rc : old;
    }
    if (n == "k3_resident") return (c->ix.kx[0] != nullptr) + (c->ix.kx[1] != nullptr);
    if (n == "ktab2_resident") return c->ix.ktab2 != nullptr;
    if (n == "grow_after_reads") { int64_t old = (int64_t)c->grow_after; if (value > 0) c->grow_after = (uint64_t)value; return old; }
    if (n == "grow_state") return c->grow_enabled ? c->grow_state.load() : 5;          // (0 .. 4: bk_ctx_int.h; 5: not a growing context)
    if (n == "image_wait") {               // BK_CTX_GROW_IMAGE: make the long-run tables now if they are not under way, wait for them, take them in
        if (c->grow_enabled) {
            grow_tick(c, 0, true);
            if (c->grow_state.load() != 4 && c->grow_state.load() != 0) grow_take_in(c);
        }
        return (c->ix.kx[0] != nullptr) + (c->ix.kx[1] != nullptr) + (c->ix.ktab2 != nullptr ? 4 : 0);
    }
    if (n == "use_ktab2") {                // k-mer table entries with the first key of their bucket (rebuilt with the tables)
        int64_t old = c->use_ktab2;
        c->use_ktab2 = value ? 1 : 0;
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

int bk_seq_counts(bk_ctx *c, uint64_t *per_entry_hits, uint32_t n, int reset)
{
    if (!c || !per_entry_hits || n != c->entries.size()) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpy(per_entry_hits, c->d_seq_counts, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(dev_zero_now(c->d_seq_counts, (size_t)n * 8));
    return BK_OK;
}

// ---- the multi-GPU exchange step: per-sequence accepted-read counts summed over the contexts of one process ----------
namespace {
__global__ void k_add_u64(unsigned long long *__restrict__ acc, const unsigned long long *__restrict__ src, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += src[i];
}

// RCCL is bound on first use (dlopen): a single-GPU run never loads it, and inside a PyTorch process the copy PyTorch already
// brought in (same SONAME) is the one that answers
struct Rccl {
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
    Rccl()
    {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { fprintf(stderr, "biokanga_amd: unable to load librccl: %s\n", dlerror()); return; }
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce");
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && AllReduce;
    }
};
}  // namespace

int bk_seq_counts_allreduce(bk_ctx *const *ctxs, int n, uint64_t *out, uint32_t n_entries, int reset)
{
    if (!ctxs || n < 1 || n > 64) return BK_ERR_PARAMS;
    for (int i = 0; i < n; i++)
        if (!ctxs[i] || ctxs[i]->entries.size() != n_entries) return BK_ERR_PARAMS;
    // one leader per distinct device: contexts sharing a GPU are summed there first
    std::vector<int> leaders;
    std::vector<int> leader_of(n);
    for (int i = 0; i < n; i++) {
        int l = -1;
        for (int j : leaders) if (ctxs[j]->device == ctxs[i]->device) { l = j; break; }
        if (l < 0) { leaders.push_back(i); l = i; }
        leader_of[i] = l;
    }
    const size_t bytes = (size_t)n_entries * 8;
    for (int i = 0; i < n; i++) {
        bk_ctx *c = ctxs[i];
        HIP_TRY(hipSetDevice(c->device));
        if (!c->d_seq_global) HIP_TRY(dev_malloc(&c->d_seq_global, bytes ? bytes : 8));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    for (int i = 0; i < n; i++) {
        bk_ctx *c = ctxs[i], *L = ctxs[leader_of[i]];
        HIP_TRY(hipSetDevice(c->device));
        if (c == L) HIP_TRY(hipMemcpyAsync(L->d_seq_global, c->d_seq_counts, bytes, hipMemcpyDeviceToDevice, L->stream));
        else {
            hipLaunchKernelGGL(k_add_u64, dim3((n_entries + 255) / 256), dim3(256), 0, L->stream, L->d_seq_global, c->d_seq_counts, n_entries);
            HIP_TRY(hipGetLastError());
        }
    }
    // ("force_rccl": a run on ONE device takes the RCCL branch too - a communicator of one rank - so that the binding, the communicator
    // set-up and the grouped all-reduce run on hardware wherever the library does, not only on a multi-GPU node)
    if (leaders.size() > 1 || ctxs[0]->force_rccl) {
        static Rccl rccl;
        if (!rccl.ok) return BK_ERR_INTERNAL;
        std::vector<int> devs;
        for (int j : leaders) devs.push_back(ctxs[j]->device);
        // one set of communicators per set of devices, made on first use and kept for the life of the process: creating them costs
        // far more than the reduction of a few hundred bytes they carry
        static std::mutex comm_mu;
        static std::map<std::vector<int>, std::vector<ncclComm_t>> comm_cache;
        std::lock_guard<std::mutex> comm_lock(comm_mu);
        auto hit = comm_cache.find(devs);
        if (hit == comm_cache.end()) {
            std::vector<ncclComm_t> fresh(devs.size());
            if (rccl.CommInitAll(fresh.data(), (int)devs.size(), devs.data()) != ncclSuccess) {
                fprintf(stderr, "biokanga_amd: ncclCommInitAll failed\n");
                return BK_ERR_INTERNAL;
            }
            hit = comm_cache.emplace(devs, std::move(fresh)).first;
        }
        const std::vector<ncclComm_t> &comms = hit->second;
        ncclResult_t r = rccl.GroupStart();
        for (size_t k = 0; k < leaders.size() && r == ncclSuccess; k++) {
            bk_ctx *L = ctxs[leaders[k]];
            (void)hipSetDevice(L->device);
            r = rccl.AllReduce(L->d_seq_global, L->d_seq_global, n_entries, ncclUint64, ncclSum, comms[k], L->stream);
        }
        ncclResult_t r2 = rccl.GroupEnd();
        for (int j : leaders) { (void)hipSetDevice(ctxs[j]->device); (void)hipStreamSynchronize(ctxs[j]->stream); }
        if (r != ncclSuccess || r2 != ncclSuccess) { fprintf(stderr, "biokanga_amd: ncclAllReduce failed\n"); return BK_ERR_INTERNAL; }
        for (int i = 0; i < n; i++) { ctxs[i]->rccl_allreduces++; ctxs[i]->rccl_ranks = (int)leaders.size(); }
    }
    for (int i = 0; i < n; i++) {
        bk_ctx *c = ctxs[i], *L = ctxs[leader_of[i]];
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(L->stream));
        if (c != L) HIP_TRY(hipMemcpy(c->d_seq_global, L->d_seq_global, bytes, hipMemcpyDeviceToDevice));
        if (reset) HIP_TRY(dev_zero_now(c->d_seq_counts, bytes));
    }
    if (out) {
        HIP_TRY(hipSetDevice(ctxs[0]->device));
        HIP_TRY(hipMemcpy(out, ctxs[0]->d_seq_global, bytes, hipMemcpyDeviceToHost));
    }
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
