// bk_engine_int.h - internal: what the translation units behind the C ABI share (bk_image.cpp: the index image in HBM and the context's
// life; bk_engine.cpp: batch scratch and the phase loop; bk_tune.cpp: parameters and knobs; bk_exchange.cpp: the multi-GPU exchange step;
// bk_snp_host.cpp: the SNP pile-up's entry points).  Not part of the boundary.
#pragma once
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
#include "bk_env.h"
#include "sfx_file.h"


namespace bk {
// launchers defined in the kernel files (*.hip)
// launchers defined in the kernel files (*.hip)
void launch_pack_target(const uint8_t *seq, uint64_t n, uint64_t *tgt4, uint64_t nwords, hipStream_t s);
void launch_pack_target2(const uint64_t *tgt4, uint64_t nwords4, uint64_t *tgt2, unsigned int *nflag32, int flag_shift, hipStream_t s);
void launch_split_sa5(const uint8_t *sa5, uint64_t n, uint32_t *lo, uint8_t *hi, hipStream_t s);
void launch_build_ktab(const DevIndex &ix, void *tab, int k, bool tab64, hipStream_t s, uint64_t i0 = 0, uint64_t i1 = 0, unsigned long long *starts = nullptr, bool pairs = false);
void launch_fill_ktab2_y(void *tab2, const uint32_t *k2, uint64_t n_entries, hipStream_t s, const uint32_t *sa_elem = nullptr);
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
void launch_swin_cover(const unsigned long long *brk, uint64_t n, uint32_t max_run, uint32_t min_run, uint32_t *flags, uint64_t n_blocks, int first_level, hipStream_t s);
void launch_pack_ktab64(const uint64_t *tab, uint64_t n_entries, uint32_t *off, uint64_t *hi, uint32_t *overflow, hipStream_t s);
void launch_count_nonzero(const uint32_t *flags, uint64_t n, unsigned long long *count, hipStream_t s);
void launch_swin_map(const uint32_t *flags, const uint32_t *incl, uint64_t n_blocks, uint32_t cap_blocks, uint32_t *used, uint32_t *map, hipStream_t s);
void launch_swin_fill(const DevIndex &ix, const uint32_t *map, void *swin, int words, uint64_t a, uint64_t e, hipStream_t s);
void launch_build_k2(const DevIndex &ix, uint32_t *k2, uint32_t *k3, uint32_t *k4, unsigned long long *bad, hipStream_t s, uint64_t i0 = 0, uint64_t i1 = 0, bool write_k2 = true);
void launch_build_k2_levels(uint32_t *k2, uint64_t n, hipStream_t s);
void launch_make_ktab2(const uint32_t *tab, const uint32_t *k2, uint64_t n_entries, uint64_t n, void *out, hipStream_t s, const uint32_t *sa_elem = nullptr);
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

// BK_TIMING=1: wall-clock of the set-up stages on stderr
struct StageClock {
    bool on = env::timing();
    double t0 = now();
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    static double wall() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return (double)(ts.tv_sec % 60) + 1e-9 * (double)ts.tv_nsec; }      // (a log's seconds)
    void lap(const char *what) { if (!on) return; const double t = now(); fprintf(stderr, "bk timing: %-28s %7.1f ms   (done at :%06.3f)\n", what, 1e3 * (t - t0), wall()); t0 = t; }
};


// ---- sizes of the batch scratch (bk_engine.cpp allocates by them, bk_image.cpp leaves room for them beside the window array)
inline uint32_t words_per_read(uint32_t maxlen)
{
    return ((maxlen + 15) / 16 + 2) & ~1u;      // even: every packed row starts 16-byte aligned
}

// per-read bytes of batch scratch (packed fwd+revcomp rows, core intervals, work lists)
// the most cores per strand a read of up to maxlen bases can have (LocateCoreMultiples' MaxNumSlides), capped at what the
// interval-slot kernels take
inline uint32_t iv_cores_for(const bk_ctx *c, uint32_t maxlen)
{
    const uint32_t ms = std::max(1u, ((uint32_t)c->cfg.slides_per100 * maxlen + 99) / 100);
    return std::min<uint32_t>(ms, kMaxCoresFast);
}

// 64-bit words of a read's 2 bit/base row in the register-window kernel family that takes reads of up to maxlen bases
inline uint32_t rd2w_for(uint32_t maxlen)
{
    return maxlen <= 128 ? 4u : (maxlen <= 256 ? 8u : (maxlen <= 16u * (uint32_t)kNwLong ? (uint32_t)kNwLong / 2 : (uint32_t)kNwLongest / 2));
}

// per-read bytes of batch scratch: packed rows in both forms, interval records, work lists (reads, search items and their striped
// forms), sort buffers
inline uint64_t scratch_bytes_per_read(uint32_t wpr, uint32_t rd2w = 8, uint32_t iv_cores = kMaxCoresFast)
{
    return 2ULL * wpr * 8 + 2ULL * rd2w * 8 + 2ULL * iv_cores * (12 + 8) + 52 + 24;
}


// ---- bk_image.cpp: parameters -> DevAlignCfg, the tables of the index image, the window array, contexts
int derive_cfg(bk_ctx *c);
void free_dev(void *p);
hipError_t clear_dev(void *p, size_t bytes, hipStream_t s);
struct TablePlan {
    bool ktab = false, k2 = false, isa = false;
    bool ktab2 = false;                            // the k-mer table's entries are pairs {bucket start, y} from the start (DevIndex::ktab2): starts written in place, y filled in tables_end
    int kx = 0;                                    // key arrays behind the second-level keys (DevIndex::kx)
    int k = 0;
    unsigned long long *d_bad = nullptr;           // places where the second-level keys are not in order inside a bucket; the third-level keys inside a run of equal second-level keys
    ~TablePlan() { free_dev(d_bad); }
};

int tables_begin(bk_ctx *c, TablePlan &tp);
int tables_range(bk_ctx *c, const TablePlan &tp, uint64_t i0, uint64_t i1, unsigned long long *bucket_starts = nullptr);
int tables_end(bk_ctx *c, TablePlan &tp);
int build_tables(bk_ctx *c);
int build_tgt2(bk_ctx *c);
void grow_take_in(bk_ctx *c);
void grow_drop(bk_ctx *c);
void grow_tick(bk_ctx *c, uint64_t nreads, bool now = false);
int setup_entries(bk_ctx *c, const bk_entry_info *entries, uint32_t n_entries);
int finish_ctx(bk_ctx *c, const bk_entry_info *entries, uint32_t n_entries);
int new_ctx(bk_ctx **out, int device_id, const bk_align_params *p, bk_ctx **pc);
int adopt_device_image(bk_ctx *c, const uint8_t *d_seq, uint64_t n, const uint8_t *d_sa, int el);
int maybe_build_swin(bk_ctx *c, uint32_t maxlen, uint32_t nreads, hipStream_t s);
// ---- bk_engine.cpp: batch scratch, the phase loop
int size_heavy_scratch(bk_ctx *c);
int ensure_batch_scratch(bk_ctx *c, uint32_t n_reads, uint32_t wpr, uint32_t rd2w = 0, uint32_t iv_cores = kMaxCoresFast);
int ensure_sort_scratch(bk_ctx *c, uint32_t n, hipStream_t s);
int align_device(bk_ctx *c, const DevReads &in, uint32_t nreads, bk_hit *d_out, hipStream_t s, uint32_t maxlen_known = 0, bool enqueue_only = false);
}  // namespace bk
