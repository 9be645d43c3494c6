// bk_ctx_int.h - internal: the context behind the C ABI (include/biokanga_amd.h), shared by bk_engine.cpp (batch driver)
// and bk_stream.cpp (overlapped host <-> device pipeline).  Not part of the boundary.
#pragma once
#include <atomic>
#include <thread>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "bk_device.h"

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            fprintf(stderr, "biokanga_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return e__ == hipErrorOutOfMemory ? BK_ERR_MEM : BK_ERR_INTERNAL;                      \
        }                                                                                          \
    } while (0)

struct bk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bk_align_params params{};
    bk::DevAlignCfg cfg{};
    bk::DevIndex ix{};
    // owned device allocations of the index image
    uint64_t *d_tgt4 = nullptr;
    uint32_t *d_sa_lo = nullptr;
    uint8_t *d_sa_hi = nullptr;
    uint64_t *d_ent_start = nullptr, *d_ent_end = nullptr;
    uint32_t *d_ent_id = nullptr, *d_id2idx = nullptr;
    void *d_ktab = nullptr;
    size_t ktab_bytes = 0, nflag_bytes = 0;
    uint64_t *d_tgt2 = nullptr;           // 2 bit/base target copy (DevIndex::tgt2)
    uint64_t *d_tgt2s = nullptr;          // the same, stored 32 bytes later (DevIndex::tgt2s)
    uint8_t *d_nflag = nullptr;
    uint64_t *d_rd2 = nullptr;            // 2-bit read rows
    uint32_t *d_rmeta = nullptr;          // DevBatch::rmeta
    uint64_t n_tgt4_words = 0;
    uint32_t cap_rd2w = 0;
    int use_tgt2 = 2;        // 0: 4-bit windows only, 1: 2-bit copy, 2: 2-bit copy stored twice (32 bytes apart)
    uint32_t *d_k2 = nullptr;             // second-level search keys (DevIndex::k2)
    uint32_t *d_kx[bk::kMoreKeys] = {nullptr, nullptr};   // third-, fourth-level search keys (DevIndex::kx)
    uint32_t *d_slist = nullptr;          // work list of the two-pass search
    uint32_t *d_sort[3] = {nullptr, nullptr, nullptr};   // keys in, keys out, list out of sort_work
    void *d_sort_tmp = nullptr;
    uint64_t cap_sort = 0;
    size_t sort_tmp_bytes = 0;
    int sort_lists = 7;      // bit 0: search work list grouped by index position; bit 1: wave list sorted, by index position or (bit 2) longest read first
    bool sort_lists_set = false;   // .. as the caller's knob left it; else bit 0 follows the index: off where it has third-level keys (tables_end)
    int sort_shift = 0;      // keys = suffix array index >> sort_shift (fits 32 bits)
    uint64_t cap_slist = 0;
    int use_k2 = 1;
    int use_k3 = bk::kMoreKeys;  // key arrays behind the second-level keys (DevIndex::kx): at most this many, where the HBM has the room
    // BK_CTX_GROW_IMAGE: the tables that only pay over long runs (key arrays behind the second-level keys, k-mer table entries with their
    // bucket's first key) are made by a thread of their own once the context has aligned grow_after reads, and taken in between two batches
    bool grow_enabled = false, grow_wait = false;
    uint64_t grow_after = 5 * BK_POLICY_MIN_READS, grow_seen = 0;      // (a context that was started lean for a short job and turned out to run a long one)
    std::atomic<int> grow_state{0};          // 0 not started, 1 being made, 2 made, 3 nothing made (no room / not in order), 4 taken in
    std::thread grow_thread;
    uint32_t *grow_kx[bk::kMoreKeys] = {nullptr, nullptr};
    void *grow_ktab2 = nullptr;
    bk::DevIndex grow_ix{};                  // the index as it was when the worker started (its own copy: the batches' thread goes on changing ix - the window array)
    bool grow_want_ktab2 = false;
    bool grow_elem = false, grow_want_elem = false;        // the worker's k-mer table of pairs carries suffix array elements for buckets of one suffix ("use_ktab2" 2)
    int use_ktab2 = 1;       // k-mer table entries of two words (DevIndex::ktab2): 1 - a bucket of one suffix carries its second-level key; 2 - its suffix array element (DevIndex::ktab2_elem: the search hands the suffix on, kElemFlag - measured slower, round 6: the buckets whose key the search would have turned down reach the extension as candidates)
    bool ktab_is2 = false;
    int use_iv32 = 1;        // phase 0 leaves the interval of a read's first k + 16 bases for the offset-0 cores of the later phases
    uint32_t wave_waves = 256u * 8u * 4u;   // resident waves the wave kernel is launched with
    int use_isa = 1;         // 0: no inverse suffix array - the wave kernel dedupes with its hash set (as it does for 5-byte indexes)
    uint32_t *d_isa = nullptr;
    void *d_swin = nullptr;               // suffix-ordered window array (DevIndex::swin), built when the first batch it serves arrives
    uint32_t *d_swmap = nullptr;          // .. and which blocks of the suffix array it holds (DevIndex::swmap; null: all of them)
    int use_swin = 1;         // 0: none; 1: the part of the suffix array the wave kernel's long walks visit; 2: the same, whatever the batch's read lengths; 3: every suffix
    uint64_t swin_budget = 0; // most bytes the partial array may take (0: by the free memory)
    int swin_skip_short = 0;  // the coverage rule leaves out this many of the reads' shortest core lengths
    int swin_w = 0;           // the core length the partial array's coverage was made for
    uint64_t swin_bytes = 0;  // what array and map occupy
    double swin_setup_s = 0;  // .. and what making them took (allocation included)
    double swin_covered = 0;  // .. share of the suffix array it holds
    bool swin_denied = false; // it did not fit beside a batch's scratch when first asked for
    bool swin_rebuilt = false; // a batch has already made the partial array again for its own core lengths (maybe_build_swin does that once)
    int use_wave = 1;        // 1: k_light / k_wave for reads <= 256 bp, 0: k_extend / k_heavy only
    int lazy_search = 1;     // 1: small k-mer buckets are handed to the extend kernels unverified
    bool ktab64 = false;
    int ktab_wide = 0;                    // "ktab_wide": 0 bucket starts of 64 bits only where the index needs them; 1 always, packed as ktab_hi + offsets; 2 always, unpacked (tests)
    uint64_t *d_ktab_hi = nullptr;        // .. stored as 32-bit offsets from a 64-bit start per 2^16 codes (DevIndex::ktab_hi) when every such group spans less than 2^32 suffixes
    int k_req = -1;          // requested k (-1 auto)
    int use_ktab = 1;
    uint32_t el_size = 4;
    uint64_t tot_seq_len = 0;
    std::string dataset;
    std::vector<bk_entry_info> entries;
    // SNP pile-up: 6 count planes over the concatenated target, site list of the last bk_snp_sites call
    uint32_t *d_snp_planes = nullptr;
    unsigned long long *d_snp_tot = nullptr;
    bk_snp_site *d_snp_sites = nullptr;
    uint32_t cap_snp_sites = 0;
    std::vector<bk_snp_site> snp_sites;

    // batch scratch (grown on demand)
    uint32_t cap_reads = 0, cap_wpr = 0, cap_iv_cores = 0;
    uint64_t *d_rd4 = nullptr, *d_iv_first = nullptr;
    uint2 *d_iv2 = nullptr;
    uint32_t *d_iv_n = nullptr, *d_act[2] = {nullptr, nullptr}, *d_heavy = nullptr, *d_wave = nullptr;
    uint2 *d_iv32 = nullptr;              // DevBatch::iv32
    uint32_t *d_wave_work = nullptr;      // DevBatch::wave_work
    uint32_t *d_stage[3] = {nullptr, nullptr, nullptr}, *d_stripe_cnt = nullptr, *d_slist_stage = nullptr;       // striped work lists (bk::StripeSet)
    uint32_t *d_small = nullptr;          // [0] act_cnt [1] next_cnt [2] heavy_cnt [3] cmax [4] cursor [5] maxlen [6] wave_cnt [7] wave cursor
    uint32_t *h_small = nullptr;          // pinned mirror (two PhaseCtl lines when the phase loop reads its counts back)
    bk::PhaseCtl *d_ctl = nullptr;        // kMaxPhases + 2 lines: the counts of a chunk's phases (bk_device.h)
    bk::PhaseCtl *h_ctl = nullptr;        // pinned: the last chunk's counts, copied behind its kernels
    hipEvent_t ev_ctl = nullptr;
    bool ctl_pending = false, hist_valid = false;
    uint32_t ctl_pending_reads = 0, ctl_pending_maxlen = 0;
    int ctl_pending_phases = 0;
    double hist_slist[bk::kMaxPhases] = {0}, hist_wave[bk::kMaxPhases] = {0};      // per phase: pass B items / wave-kernel reads per read of the chunk
    int async_error = 0;     // what take_phase_history found wrong with a batch of bk_align_batch_device_async (reported by the next call)
    int async_phases = 1;    // 1: the main path's phase loop launches without reading counts back (see align_chunk)
    unsigned long long *d_seq_counts = nullptr, *d_ctr = nullptr, *d_ctr_aux = nullptr;
    unsigned long long *d_seq_global = nullptr;   // bk_seq_counts_allreduce: the counts summed over every context of the run
    bool force_rccl = false;                      // .. through RCCL even on one device ("force_rccl")
    uint64_t rccl_allreduces = 0;                 // .. how many of this context's reductions went through RCCL, the ranks of the last one's communicator
    int rccl_ranks = 0;
    // heavy path scratch
    bk::HeavyScratch hs{};
    int max_read_len = 500;
    uint32_t last_maxlen = 0;    // longest read of the last align call
    // test hook (bk_debug_intervals): batches stop behind the search of this phase and leave its interval records in the scratch
    int dbg_stop_phase = -1, dbg_phase = 0, dbg_cur = 0;
    uint32_t dbg_n = 0, dbg_ivc = 0;
    bool dbg_valid = false;
    bool debug = false;      // BK_DEBUG in the environment when the context was created: per-phase counts on stderr
    uint32_t chunk_reads = 64u << 20;
    // staging for host-buffer batches
    uint8_t *d_in_bases = nullptr;
    uint64_t *d_in_offs = nullptr;
    uint32_t *d_in_lens = nullptr;
    bk_hit *d_in_out = nullptr;
    bk_seg2 *d_seg2 = nullptr;            // -a / -A / -c: second segments of a chunk (kept with the batch scratch)
    // packed host batches (bk_align_batch_packed) and the offset scan of packed batches
    uint32_t *d_in_words = nullptr;
    uint16_t *d_in_lens16 = nullptr;
    bk_nbase *d_in_exc = nullptr;
    uint64_t cap_in_words = 0, cap_in_exc = 0;
    uint32_t cap_in_lens16 = 0;
    void *d_scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    uint32_t cap_seg2 = 0;
    uint64_t cap_in_bases = 0;
    uint32_t cap_in_reads = 0;

    hipEvent_t ev_wait = nullptr;         // an event the caller's thread sleeps on (bk_wait.h)
    bool entries_set = false, tgt2_built = false;     // .. and so do the entry table / the 2-bit target (made early for a window array that is made behind the upload too)
    bool tables_built = false;            // k-mer table, second-level keys, inverse suffix array exist (made behind the suffix array's upload)
    void *sam_text[2] = {nullptr, nullptr};   // bk_sam_format's page-locked text buffers, kept for the next call (giving page-locked memory back costs 0.1 s per GB)
    uint64_t sam_text_cap = 0;
    uint8_t *d_chrom_accept = nullptr;    // bk_ctx_set_chrom_filter: by sequence id, what the PE rules ask of the -Z / -z filters
    uint32_t n_chrom_accept = 0;
    bk_timing timing{};
    std::vector<hipEvent_t> ev_pool;
    // multi-loci modes: loci lists of the last align call (host side, see bk_batch_loci)
    std::vector<uint64_t> loci_offs;
    std::vector<bk_loci> loci;
    std::vector<bk_loci_trims> loci_trims;   // -c with the multi-loci modes: one per locus
    std::vector<bk_seg2> seg2;           // -a: second segments of the last align call, one per read
};


namespace bk {
// a batch of reads resident in HBM, in either form the boundary takes
struct DevReads {
    const uint8_t *bases = nullptr;       // 1 byte/base form: the bases, offs[i] = start of read i in them
    const uint64_t *offs = nullptr;       //   (packed form: first word of read i in `words`)
    const uint32_t *lens = nullptr;
    const uint32_t *words = nullptr;      // packed form (bk_align_batch_packed): 2 bit/base words, then bases == nullptr
    const bk_nbase *exc = nullptr;        //   + the bases that are not a,c,g,t
    uint64_t n_exc = 0;
};
// batch driver entry points of bk_engine.cpp used by the stream pipeline (all blocking on `s`: the phase loop reads the
// active counts back between phases)
int engine_align_device(bk_ctx *c, const DevReads &in, uint32_t nreads, bk_hit *d_out, hipStream_t s, uint32_t maxlen_known = 0);
void release_swin(bk_ctx *c);
int engine_pair_device(bk_ctx *c, const DevReads &in, uint32_t n_pairs, bk_hit *d_hits, uint32_t maxlen, const bk_pe_params *pe, hipStream_t s,
                       bk_seg2 *seg2_host = nullptr, bk_seg2 *seg2_dev = nullptr);
// packed batches: lens16 -> d_lens32, word offsets of the reads -> d_offs (scan), the batch checked (word count, read lengths,
// exception list); *maxlen = longest read.  Blocking on `s`.
int engine_prepare_packed(bk_ctx *c, const uint16_t *d_lens16, uint32_t nreads, uint64_t n_words, const bk_nbase *d_exc, uint64_t n_exc,
                          uint32_t *d_lens32, uint64_t *d_offs, uint32_t *maxlen, hipStream_t s);
}  // namespace bk

namespace bk {
// Host -> device copy that runs near PCIe rate whatever the kind of host memory: pinned sources are DMA'd directly; pageable
// ones (file mappings, std::vector) are copied by a few host threads into a pool of pinned staging buffers, each thread feeding
// its own HIP stream, so that the DRAM / page-cache reads and the DMAs of different slices overlap (a plain hipMemcpy stages
// through one thread: 6-8 GB/s).  Returns when the bytes are on the device.  bk_upload.cpp
// hipMalloc for everything the context and its pipelines own.  BK_POISON=<byte> (debugging aid) fills every allocation with that byte, so a
// kernel that reads device memory nobody wrote meets the same bytes on every box, not whatever the previous process left in HBM.
hipError_t dev_malloc_bytes(void **p, size_t bytes);
hipError_t dev_zero_now(void *p, size_t bytes);
template <class T> inline hipError_t dev_malloc(T **p, size_t bytes) { return dev_malloc_bytes(reinterpret_cast<void **>(p), bytes); }
int upload_host(void *d_dst, const void *h_src, size_t bytes, int device);
int upload_file(void *d_dst, int fd, uint64_t file_ofs, size_t bytes, int device);       // the same from a file range, read() into the staging buffers
bool host_is_pinned(const void *p);
}  // namespace bk
