// bk_stream.cpp - overlapped host <-> device pipeline over the batch driver (bk_stream_* of include/biokanga_amd.h).
//
// The reference overlaps its loader thread with the aligner threads: reads are handed out in blocks while the
// file is still being parsed (CAligner::LoadReads / ThreadedIterReads, biokanga/Aligner.cpp:4820-4860,9636-9704),
// and T_align of SURVEY.md §8(d) runs from the first block handed out to the last result stored.  Here the same
// overlap is between PCIe and the kernels: three host threads, three HIP streams, `depth` sets of device buffers.
//
//   submit() -> [uploader: H2D of batch k+1 on s_up] -> [aligner: every AlignReads phase of batch k on s_al,
//                (+ paired-end association on the resident buffers)] -> [downloader: D2H of batch k-1 on s_dn] -> wait()
//
// The batch scratch (packed reads, core intervals, work lists) belongs to the context and is used by one batch at a
// time - the phase kernels fill the whole chip - so only the 1 B/base reads, their offsets/lengths and the 20-byte
// result records are multi-buffered.  Host buffers obtained from bk_host_alloc() (pinned) are DMA'd directly;
// pageable ones go through the HIP runtime's staging copies (slower, still overlapped with the kernels).
#include <hip/hip_runtime.h>
#include "bk_prim.h"

#include <algorithm>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

#include "bk_ctx_int.h"
#include "bk_wait.h"
#include "bk_env.h"

namespace {

struct CastU64 {
    __host__ __device__ unsigned long long operator()(const uint32_t &v) const { return (unsigned long long)v; }
};

// max over reads of offs[i] + lens[i] (reads must lie inside the uploaded bases) and of lens[i].  Checked per element without a
// sum that can wrap: a read that starts beyond `nbases` or runs past it reports an end of ~0.
__global__ void __launch_bounds__(256) k_extent(const uint64_t *__restrict__ offs, const uint32_t *__restrict__ lens, uint32_t n,
                                                unsigned long long nbases, unsigned long long *__restrict__ out)
{
    unsigned long long e = 0, l = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long len = lens[i], o = offs[i];
        const unsigned long long end = (o > nbases || len > nbases - o) ? ~0ULL : o + len;
        e = end > e ? end : e;
        l = len > l ? len : l;
    }
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long e2 = __shfl_down(e, off), l2 = __shfl_down(l, off);
        e = e2 > e ? e2 : e;
        l = l2 > l ? l2 : l;
    }
    if ((threadIdx.x & 63) == 0) {
        if (e) atomicMax(out + 0, e);
        if (l) atomicMax(out + 1, l);
    }
}

struct Job {
    uint64_t ticket = 0;
    const uint8_t *bases = nullptr;
    uint64_t nbases = 0;
    const uint64_t *offs = nullptr;
    const uint32_t *lens = nullptr;
    // packed form (bk_stream_submit_packed): words != nullptr or n_words == 0 with lens16 set
    const uint32_t *words = nullptr;
    uint64_t n_words = 0;
    const uint16_t *lens16 = nullptr;
    const bk_nbase *exc = nullptr;
    uint64_t n_exc = 0;
    uint32_t n = 0;
    bk_hit *out = nullptr;
    // device-resident form (bk_stream_submit_device): the caller's buffers in HBM, results written where they say; ev_in = the point
    // of the caller's stream the buffers are ready at
    bool dev = false;
    const uint8_t *d_bases = nullptr;
    const uint64_t *d_offs = nullptr;
    const uint32_t *d_lens = nullptr;
    bk_hit *d_out = nullptr;
    hipEvent_t ev_in = nullptr;
    ~Job() { if (ev_in) (void)hipEventDestroy(ev_in); }
    int slot = 0;
    int rc = BK_OK;
    bool done = false;
    double t_submit = 0, t_done = 0;
    // list modes: what the context held after this batch's align call
    std::vector<uint64_t> loci_offs;
    std::vector<bk_loci> loci;
    std::vector<bk_loci_trims> loci_trims;
    std::vector<bk_seg2> seg2;
};

struct Slot {
    uint8_t *d_bases = nullptr;           // 1 byte/base reads, or the words of a packed batch
    uint16_t *d_lens16 = nullptr;
    bk_nbase *d_exc = nullptr;            // grown on demand
    uint64_t cap_exc = 0;
    uint64_t *d_offs = nullptr;
    uint32_t *d_lens = nullptr;
    bk_hit *d_out = nullptr;
    hipEvent_t ev_up = nullptr, ev_al = nullptr;
};

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

}  // namespace

struct bk_stream {
    bk_ctx *ctx = nullptr;
    int depth = 0;
    uint32_t max_reads = 0;
    uint64_t max_bases = 0;
    uint64_t max_words = 0;            // != 0: packed batches only, of at most this many words
    std::vector<Slot> slots;
    hipStream_t s_up = nullptr, s_al = nullptr, s_dn = nullptr;
    void *d_scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    unsigned long long *d_ext = nullptr, *h_ext = nullptr;     // [0] max read end [1] max read length
    std::thread t_up, t_al, t_dn;
    hipEvent_t ev_w_al = nullptr, ev_w_dn = nullptr;       // what the aligner and the download thread sleep on (bk_wait.h)
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job *> q_up, q_al, q_dn;
    std::map<uint64_t, Job *> jobs;
    uint64_t next_ticket = 1, n_done = 0;
    bool stop = false, list_modes = false, has_pe = false, span_open = false;
    bk_pe_params pe{};
    bk_stream_stats stats{};
    double t_first_submit = 0, t_last_done = 0;

    void fail(Job *j, int rc) { if (j->rc == BK_OK) j->rc = rc; }
    static int rc_of(hipError_t e) { return e == hipSuccess ? BK_OK : (e == hipErrorOutOfMemory ? BK_ERR_MEM : BK_ERR_INTERNAL); }

    Job *pop(std::deque<Job *> &q)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return nullptr;
        Job *j = q.front();
        q.pop_front();
        return j;
    }
    void push(std::deque<Job *> &q, Job *j)
    {
        { std::lock_guard<std::mutex> lk(mu); q.push_back(j); }
        cv.notify_all();
    }

    void run_upload()
    {
        (void)hipSetDevice(ctx->device);
        while (Job *j = pop(q_up)) {
            Slot &sl = slots[j->slot];
            hipError_t e = hipSuccess;
            if (j->dev) { push(q_al, j); continue; }          // nothing to upload: the batch keeps its place in the queue
            if (j->n && j->lens16) {
                // packed batch: 2 bit/base words, 16-bit lengths, the few bases that are not a,c,g,t
                const size_t wb = (size_t)j->n_words * 4;
                if (wb) {
                    if (bk::host_is_pinned(j->words)) e = hipMemcpyAsync(sl.d_bases, j->words, wb, hipMemcpyHostToDevice, s_up);
                    else if (bk::upload_host(sl.d_bases, j->words, wb, ctx->device) != BK_OK) e = hipErrorUnknown;
                }
                if (e == hipSuccess) e = hipMemcpyAsync(sl.d_lens16, j->lens16, (size_t)j->n * 2, hipMemcpyHostToDevice, s_up);
                if (e == hipSuccess && j->n_exc > sl.cap_exc) {
                    // (the slot is idle: its previous batch has been waited for)
                    if (sl.d_exc) (void)hipFree(sl.d_exc);
                    sl.d_exc = nullptr;
                    sl.cap_exc = 0;
                    e = bk::dev_malloc(&sl.d_exc, (size_t)(j->n_exc + j->n_exc / 4 + 1024) * sizeof(bk_nbase));
                    if (e == hipSuccess) sl.cap_exc = j->n_exc + j->n_exc / 4 + 1024;
                }
                if (e == hipSuccess && j->n_exc) e = hipMemcpyAsync(sl.d_exc, j->exc, (size_t)j->n_exc * sizeof(bk_nbase), hipMemcpyHostToDevice, s_up);
            } else if (j->n) {
                // the bases are the bulk: pageable buffers are staged by several threads (bk::upload_host), pinned ones DMA'd as they are
                if (bk::host_is_pinned(j->bases)) e = hipMemcpyAsync(sl.d_bases, j->bases, j->nbases, hipMemcpyHostToDevice, s_up);
                else if (bk::upload_host(sl.d_bases, j->bases, j->nbases, ctx->device) != BK_OK) e = hipErrorUnknown;
                if (e == hipSuccess && j->offs) e = hipMemcpyAsync(sl.d_offs, j->offs, (size_t)j->n * 8, hipMemcpyHostToDevice, s_up);
                if (e == hipSuccess) e = hipMemcpyAsync(sl.d_lens, j->lens, (size_t)j->n * 4, hipMemcpyHostToDevice, s_up);
            }
            if (e == hipSuccess) e = hipEventRecord(sl.ev_up, s_up);
            if (e != hipSuccess) fail(j, rc_of(e));
            push(q_al, j);
        }
    }

    void run_align()
    {
        (void)hipSetDevice(ctx->device);
        const bool timing = bk::env::timing();       // per batch on stderr: how long the aligner thread waited for it, prepared it, aligned it
        double t_free = now_s();
        for (;;) {
            Job *j = pop(q_al);
            if (!j) break;
            const double t_got = now_s();
            double t_prep = t_got;
            Slot &sl = slots[j->slot];
            if (j->dev) {
                if (j->rc == BK_OK && j->n) {
                    hipError_t e = hipStreamWaitEvent(s_al, j->ev_in, 0);          // the caller's kernels that fill the buffers come first
                    bk::DevReads in;
                    in.bases = j->d_bases; in.offs = j->d_offs; in.lens = j->d_lens;
                    int rc = e == hipSuccess ? bk::engine_align_device(ctx, in, j->n, j->d_out, s_al, 0) : rc_of(e);
                    if (rc == BK_OK && has_pe) {
                        // (the longest read of the batch: the SE pass has just measured it)
                        rc = bk::engine_pair_device(ctx, in, j->n / 2, j->d_out, (uint32_t)ctx->last_maxlen, &pe, s_al, ctx->seg2.empty() ? nullptr : ctx->seg2.data());
                    }
                    if (rc) fail(j, rc);
                    else if (list_modes) {
                        j->loci_offs.swap(ctx->loci_offs);
                        j->loci.swap(ctx->loci);
                        j->loci_trims.swap(ctx->loci_trims);
                        j->seg2.swap(ctx->seg2);
                    }
                }
                push(q_dn, j);
                t_free = now_s();
                continue;
            }
            if (j->rc == BK_OK && j->n && j->lens16) {
                hipError_t e = hipStreamWaitEvent(s_al, sl.ev_up, 0);
                uint32_t maxlen = 0;
                int rc = e == hipSuccess ? bk::engine_prepare_packed(ctx, sl.d_lens16, j->n, j->n_words, sl.d_exc, j->n_exc, sl.d_lens, sl.d_offs, &maxlen, s_al)
                                         : rc_of(e);
                t_prep = now_s();
                bk::DevReads in;
                in.offs = sl.d_offs; in.lens = sl.d_lens; in.words = reinterpret_cast<const uint32_t *>(sl.d_bases); in.exc = sl.d_exc; in.n_exc = j->n_exc;
                if (rc == BK_OK) rc = bk::engine_align_device(ctx, in, j->n, sl.d_out, s_al, maxlen);
                if (rc == BK_OK && has_pe) rc = bk::engine_pair_device(ctx, in, j->n / 2, sl.d_out, maxlen, &pe, s_al, ctx->seg2.empty() ? nullptr : ctx->seg2.data());
                if (rc) fail(j, rc);
                else if (list_modes) {
                    j->loci_offs.swap(ctx->loci_offs);
                    j->loci.swap(ctx->loci);
                    j->loci_trims.swap(ctx->loci_trims);
                    j->seg2.swap(ctx->seg2);
                }
                if (j->rc == BK_OK) {
                    e = hipEventRecord(sl.ev_al, s_al);
                    if (e != hipSuccess) fail(j, rc_of(e));
                }
            } else if (j->rc == BK_OK && j->n) {
                hipError_t e = hipStreamWaitEvent(s_al, sl.ev_up, 0);
                if (e == hipSuccess && !j->offs) {          // contiguous reads: offsets = exclusive prefix sum of the lengths
                    size_t tb = scan_tmp_bytes;
                    rocprim::transform_iterator<const uint32_t *, CastU64, unsigned long long> in(sl.d_lens, CastU64());
                    e = bk::prim::exclusive_sum(d_scan_tmp, tb, in, (unsigned long long *)sl.d_offs, (size_t)j->n, s_al);
                }
                if (e == hipSuccess) e = hipMemsetAsync(d_ext, 0, 16, s_al);
                if (e == hipSuccess) {
                    unsigned blocks = (j->n + 255) / 256;
                    if (blocks > 2048) blocks = 2048;
                    hipLaunchKernelGGL(k_extent, dim3(blocks), dim3(256), 0, s_al, sl.d_offs, sl.d_lens, j->n, (unsigned long long)j->nbases, d_ext);
                    e = hipGetLastError();
                }
                if (e == hipSuccess) e = hipMemcpyAsync(h_ext, d_ext, 16, hipMemcpyDeviceToHost, s_al);
                if (e == hipSuccess) e = bk::wait_stream(s_al, ev_w_al);
                if (e != hipSuccess) fail(j, rc_of(e));
                else if (h_ext[0] > j->nbases || h_ext[1] > (unsigned long long)bk::kMaxReadLenAbs) fail(j, BK_ERR_PARAMS);
                const uint32_t maxlen = (uint32_t)h_ext[1];
                if (j->rc == BK_OK) {
                    bk::DevReads in;
                    in.bases = sl.d_bases; in.offs = sl.d_offs; in.lens = sl.d_lens;
                    int rc = bk::engine_align_device(ctx, in, j->n, sl.d_out, s_al, maxlen);
                    if (rc == BK_OK && has_pe) rc = bk::engine_pair_device(ctx, in, j->n / 2, sl.d_out, maxlen, &pe, s_al, ctx->seg2.empty() ? nullptr : ctx->seg2.data());
                    if (rc) fail(j, rc);
                    else if (list_modes) {
                        j->loci_offs.swap(ctx->loci_offs);
                        j->loci.swap(ctx->loci);
                    j->loci_trims.swap(ctx->loci_trims);
                        j->seg2.swap(ctx->seg2);
                    }
                }
                if (j->rc == BK_OK) {
                    e = hipEventRecord(sl.ev_al, s_al);
                    if (e != hipSuccess) fail(j, rc_of(e));
                }
            }
            if (ctx->debug || timing)
                fprintf(stderr, "bk: stream batch of %u reads: waited %.2f ms for it, prepared in %.2f ms, aligned in %.2f ms\n", j->n, 1e3 * (t_got - t_free),
                        1e3 * (t_prep - t_got), 1e3 * (now_s() - t_prep));
            push(q_dn, j);
            t_free = now_s();
        }
    }

    void run_download()
    {
        (void)hipSetDevice(ctx->device);
        while (Job *j = pop(q_dn)) {
            Slot &sl = slots[j->slot];
            if (j->rc != BK_OK) {
                // nothing of a failed batch - device-resident ones included: their kernels read the caller's buffers and write the
                // context's batch scratch - may still be in flight when its slot (and that scratch) is reused or the caller frees
                (void)hipStreamSynchronize(s_up);
                (void)hipStreamSynchronize(s_al);
                (void)hipStreamSynchronize(s_dn);
            } else if (j->dev) {
                // results were written where the caller said and the aligner thread has waited for its last kernel
            } else if (j->n) {
                hipError_t e = hipStreamWaitEvent(s_dn, sl.ev_al, 0);
                if (e == hipSuccess) e = hipMemcpyAsync(j->out, sl.d_out, (size_t)j->n * sizeof(bk_hit), hipMemcpyDeviceToHost, s_dn);
                if (e == hipSuccess) e = bk::wait_stream(s_dn, ev_w_dn);         // (asleep until the records are back)
                if (e != hipSuccess) fail(j, rc_of(e));
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                j->done = true;
                j->t_done = now_s();
                t_last_done = j->t_done;
                n_done++;
                stats.batches++;
                stats.reads += j->n;
                if (!j->dev) stats.bytes_h2d += j->lens16 ? j->n_words * 4 + (uint64_t)j->n * 2 + j->n_exc * sizeof(bk_nbase) : j->nbases + (uint64_t)j->n * (j->offs ? 12 : 4);
                if (!j->dev) stats.bytes_d2h += (uint64_t)j->n * sizeof(bk_hit);
                stats.seconds_first_submit_to_last_result = t_last_done - t_first_submit;
            }
            cv.notify_all();
        }
    }
};

extern "C" {

void *bk_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }     // (every device's pipeline may DMA from it)
    return p;
}

void bk_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int bk_host_register(void *p, size_t bytes)
{
    if (!p || !bytes) return BK_ERR_PARAMS;
    if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return BK_ERR_MEM; }
    return BK_OK;
}

void bk_host_unregister(void *p)
{
    if (p && hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
}

void bk_stream_destroy(bk_stream *s)
{
    if (!s) return;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->n_done + 1 == s->next_ticket; });      // let what was submitted finish
        s->stop = true;
    }
    s->cv.notify_all();
    if (s->t_up.joinable()) s->t_up.join();
    if (s->t_al.joinable()) s->t_al.join();
    if (s->t_dn.joinable()) s->t_dn.join();
    (void)hipSetDevice(s->ctx->device);
    for (Slot &sl : s->slots) {
        if (sl.d_bases) (void)hipFree(sl.d_bases);
        if (sl.d_lens16) (void)hipFree(sl.d_lens16);
        if (sl.d_exc) (void)hipFree(sl.d_exc);
        if (sl.d_offs) (void)hipFree(sl.d_offs);
        if (sl.d_lens) (void)hipFree(sl.d_lens);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.ev_up) (void)hipEventDestroy(sl.ev_up);
        if (sl.ev_al) (void)hipEventDestroy(sl.ev_al);
    }
    if (s->d_scan_tmp) (void)hipFree(s->d_scan_tmp);
    if (s->d_ext) (void)hipFree(s->d_ext);
    if (s->h_ext) (void)hipHostFree(s->h_ext);
    if (s->s_up) (void)hipStreamDestroy(s->s_up);
    if (s->s_al) (void)hipStreamDestroy(s->s_al);
    if (s->s_dn) (void)hipStreamDestroy(s->s_dn);
    if (s->ev_w_al) (void)hipEventDestroy(s->ev_w_al);
    if (s->ev_w_dn) (void)hipEventDestroy(s->ev_w_dn);
    for (auto &kv : s->jobs) delete kv.second;
    delete s;
}

static int stream_create(bk_stream **out, bk_ctx *ctx, uint32_t max_batch_reads, uint64_t max_batch_bases, uint64_t max_batch_words, int depth, const bk_pe_params *pe);

int bk_stream_create(bk_stream **out, bk_ctx *ctx, uint32_t max_batch_reads, uint64_t max_batch_bases, int depth, const bk_pe_params *pe)
{
    return stream_create(out, ctx, max_batch_reads, max_batch_bases, 0, depth, pe);
}

int bk_stream_create_packed(bk_stream **out, bk_ctx *ctx, uint32_t max_batch_reads, uint64_t max_batch_words, int depth, const bk_pe_params *pe)
{
    if (!max_batch_words) return BK_ERR_PARAMS;
    return stream_create(out, ctx, max_batch_reads, 0, max_batch_words, depth, pe);
}

// max_batch_words != 0: a pipeline for packed batches only - its device buffers hold 4 bytes per 16 bases instead of 16
static int stream_create(bk_stream **out, bk_ctx *ctx, uint32_t max_batch_reads, uint64_t max_batch_bases, uint64_t max_batch_words, int depth, const bk_pe_params *pe)
{
    const bool packed_only = max_batch_words != 0;
    if (packed_only) max_batch_bases = 16 * max_batch_words;       // (what submit_packed's bound is derived from; no 1 byte/base batch is accepted)
    if (!out || !ctx || !max_batch_reads || !max_batch_bases || depth < 1 || depth > 8) return BK_ERR_PARAMS;
    if (pe && (pe->pe_mode < 1 || pe->pe_mode > 4 || pe->pair_min_len < 1 || pe->pair_max_len < pe->pair_min_len)) return BK_ERR_PARAMS;
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    bk_stream *s = new bk_stream();
    s->ctx = ctx;
    s->depth = depth;
    s->max_reads = max_batch_reads;
    s->max_bases = max_batch_bases;
    s->max_words = max_batch_words;
    s->list_modes = ctx->params.max_ml > 1 || ctx->params.micro_indel_len > 0 || ctx->params.splice_junct_len > 0 || ctx->params.min_chimeric_len > 0;
    if (pe) { s->has_pe = true; s->pe = *pe; }
    s->slots.resize((size_t)depth);
    hipError_t e = hipStreamCreateWithFlags(&s->s_up, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->s_al, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->s_dn, hipStreamNonBlocking);
    if (e == hipSuccess) e = bk::make_wait_event(&s->ev_w_al);
    if (e == hipSuccess) e = bk::make_wait_event(&s->ev_w_dn);
    auto alloc_slots = [&]() {
        hipError_t e2 = hipSuccess;
        for (Slot &sl : s->slots) {
            // (either form of a batch: max_batch_bases bytes, or one word per 16 bases and at most one more per read)
            //  + the words the read preparation may load behind the last read of a packed batch, bk::kPackedPadWords)
            const uint64_t bytes = packed_only ? 4ULL * max_batch_words : std::max<uint64_t>(max_batch_bases, max_batch_bases / 4 + 4ULL * max_batch_reads);
            if (e2 == hipSuccess) e2 = bk::dev_malloc(&sl.d_bases, bytes + 64 + 4ULL * bk::kPackedPadWords);
            if (e2 == hipSuccess) e2 = bk::dev_malloc(&sl.d_lens16, (size_t)max_batch_reads * 2);
            if (e2 == hipSuccess) e2 = bk::dev_malloc(&sl.d_offs, (size_t)max_batch_reads * 8);
            if (e2 == hipSuccess) e2 = bk::dev_malloc(&sl.d_lens, (size_t)max_batch_reads * 4);
            if (e2 == hipSuccess) e2 = bk::dev_malloc(&sl.d_out, (size_t)max_batch_reads * sizeof(bk_hit));
        }
        return e2;
    };
    if (e == hipSuccess) {
        e = alloc_slots();
        if (e == hipErrorOutOfMemory && ctx->d_swin) {
            // the context's window array (half of the HBM) goes before the pipeline is refused its buffers
            (void)hipGetLastError();
            for (Slot &sl : s->slots) {
                for (void **pp : {(void **)&sl.d_bases, (void **)&sl.d_lens16, (void **)&sl.d_offs, (void **)&sl.d_lens, (void **)&sl.d_out})
                    if (*pp) { (void)hipFree(*pp); *pp = nullptr; }
            }
            bk::release_swin(ctx);
            e = alloc_slots();
        }
    }
    for (Slot &sl : s->slots) {
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_up, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_al, hipEventDisableTiming);
    }
    if (e == hipSuccess) {
        rocprim::transform_iterator<const uint32_t *, CastU64, unsigned long long> in(nullptr, CastU64());
        e = bk::prim::exclusive_sum(nullptr, s->scan_tmp_bytes, in, (unsigned long long *)nullptr, (size_t)max_batch_reads, s->s_al);
    }
    if (e == hipSuccess) e = bk::dev_malloc(&s->d_scan_tmp, s->scan_tmp_bytes ? s->scan_tmp_bytes : 16);
    if (e == hipSuccess) e = bk::dev_malloc(&s->d_ext, 16);
    if (e == hipSuccess) e = hipHostMalloc(&s->h_ext, 16, hipHostMallocDefault);
    if (e != hipSuccess) {
        fprintf(stderr, "biokanga_amd: bk_stream_create: %s\n", hipGetErrorString(e));
        s->next_ticket = 1;
        bk_stream_destroy(s);
        return e == hipErrorOutOfMemory ? BK_ERR_MEM : BK_ERR_INTERNAL;
    }
    s->t_up = std::thread([s] { s->run_upload(); });
    s->t_al = std::thread([s] { s->run_align(); });
    s->t_dn = std::thread([s] { s->run_download(); });
    *out = s;
    return BK_OK;
}

int bk_stream_submit(bk_stream *s, const uint8_t *bases, uint64_t nbases, const uint64_t *offs, const uint32_t *lens, uint32_t nreads,
                     bk_hit *out, uint64_t *ticket)
{
    if (!s || !ticket || (nreads && (!bases || !lens || !out))) return BK_ERR_PARAMS;
    if (nreads > s->max_reads || nbases > s->max_bases || s->max_words || (s->has_pe && (nreads & 1))) return BK_ERR_PARAMS;
    Job *j = new Job();
    j->bases = bases; j->nbases = nbases; j->offs = offs; j->lens = lens; j->n = nreads; j->out = out;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->next_ticket - 1 - s->n_done < (uint64_t)s->depth; });     // a free set of device buffers
        j->ticket = s->next_ticket++;
        j->slot = (int)((j->ticket - 1) % (uint64_t)s->depth);
        j->t_submit = now_s();
        if (!s->span_open) { s->t_first_submit = j->t_submit; s->span_open = true; }      // T_align: first batch submitted -> last result back
        s->jobs[j->ticket] = j;
        s->q_up.push_back(j);
    }
    s->cv.notify_all();
    *ticket = j->ticket;
    return BK_OK;
}

int bk_stream_submit_device(bk_stream *s, const void *d_bases, const void *d_offs, const void *d_lens, uint32_t nreads, void *d_hits,
                            void *producer_stream, uint64_t *ticket)
{
    if (!s || !ticket || (nreads && (!d_bases || !d_offs || !d_lens || !d_hits))) return BK_ERR_PARAMS;
    if (s->has_pe && (nreads & 1)) return BK_ERR_PARAMS;
    Job *j = new Job();
    j->dev = true;
    j->d_bases = (const uint8_t *)d_bases; j->d_offs = (const uint64_t *)d_offs; j->d_lens = (const uint32_t *)d_lens; j->d_out = (bk_hit *)d_hits;
    j->n = nreads;
    hipError_t e = hipSetDevice(s->ctx->device);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&j->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(j->ev_in, (hipStream_t)producer_stream);
    if (e != hipSuccess) { delete j; return BK_ERR_INTERNAL; }
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->next_ticket - 1 - s->n_done < (uint64_t)s->depth; });     // at most `depth` batches in flight
        j->ticket = s->next_ticket++;
        j->slot = (int)((j->ticket - 1) % (uint64_t)s->depth);
        j->t_submit = now_s();
        if (!s->span_open) { s->t_first_submit = j->t_submit; s->span_open = true; }
        s->jobs[j->ticket] = j;
        s->q_up.push_back(j);
    }
    s->cv.notify_all();
    *ticket = j->ticket;
    return BK_OK;
}

int bk_stream_submit_packed(bk_stream *s, const uint32_t *words, uint64_t n_words, const uint16_t *lens, uint32_t nreads,
                            const bk_nbase *exc, uint64_t n_exc, bk_hit *out, uint64_t *ticket)
{
    if (!s || !ticket || (nreads && (!lens || !out)) || (n_words && !words) || (n_exc && !exc)) return BK_ERR_PARAMS;
    // (a batch's words are bounded like its bases: 16 of them per word, a last partial word per read)
    if (nreads > s->max_reads || n_words > (s->max_words ? s->max_words : s->max_bases / 16 + nreads) || (s->has_pe && (nreads & 1))) return BK_ERR_PARAMS;
    Job *j = new Job();
    j->words = words; j->n_words = n_words; j->lens16 = lens; j->exc = exc; j->n_exc = n_exc; j->n = nreads; j->out = out;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->next_ticket - 1 - s->n_done < (uint64_t)s->depth; });     // a free set of device buffers
        j->ticket = s->next_ticket++;
        j->slot = (int)((j->ticket - 1) % (uint64_t)s->depth);
        j->t_submit = now_s();
        if (!s->span_open) { s->t_first_submit = j->t_submit; s->span_open = true; }
        s->jobs[j->ticket] = j;
        s->q_up.push_back(j);
    }
    s->cv.notify_all();
    *ticket = j->ticket;
    return BK_OK;
}

int bk_stream_wait(bk_stream *s, uint64_t ticket)
{
    if (!s) return BK_ERR_PARAMS;
    std::unique_lock<std::mutex> lk(s->mu);
    auto it = s->jobs.find(ticket);
    if (it == s->jobs.end()) return BK_ERR_PARAMS;
    Job *j = it->second;
    s->cv.wait(lk, [&] { return j->done; });
    const int rc = j->rc;
    if (!s->list_modes || rc != BK_OK) { s->jobs.erase(it); delete j; }       // nothing more to fetch for this batch
    return rc;
}

int bk_stream_batch_loci(bk_stream *s, uint64_t ticket, const uint64_t **offs, const bk_loci **loci, uint64_t *n_loci)
{
    if (!s || !offs || !loci || !n_loci) return BK_ERR_PARAMS;
    std::lock_guard<std::mutex> lk(s->mu);
    auto it = s->jobs.find(ticket);
    if (it == s->jobs.end() || !it->second->done) return BK_ERR_PARAMS;
    Job *j = it->second;
    if (j->loci_offs.empty()) { *offs = nullptr; *loci = nullptr; *n_loci = 0; return BK_OK; }
    *offs = j->loci_offs.data();
    *loci = j->loci.data();
    *n_loci = j->loci.size();
    return BK_OK;
}

int bk_stream_batch_seg2(bk_stream *s, uint64_t ticket, const bk_seg2 **seg2, uint64_t *n)
{
    if (!s || !seg2 || !n) return BK_ERR_PARAMS;
    std::lock_guard<std::mutex> lk(s->mu);
    auto it = s->jobs.find(ticket);
    if (it == s->jobs.end() || !it->second->done) return BK_ERR_PARAMS;
    Job *j = it->second;
    *seg2 = j->seg2.empty() ? nullptr : j->seg2.data();
    *n = j->seg2.size();
    return BK_OK;
}

int bk_stream_batch_loci_trims(bk_stream *s, uint64_t ticket, const bk_loci_trims **trims, uint64_t *n_loci)
{
    if (!s || !trims || !n_loci) return BK_ERR_PARAMS;
    std::lock_guard<std::mutex> lk(s->mu);
    auto it = s->jobs.find(ticket);
    if (it == s->jobs.end() || !it->second->done) return BK_ERR_PARAMS;
    Job *j = it->second;
    *trims = j->loci_trims.empty() ? nullptr : j->loci_trims.data();
    *n_loci = j->loci_trims.size();
    return BK_OK;
}

int bk_stream_release(bk_stream *s, uint64_t ticket)
{
    if (!s) return BK_ERR_PARAMS;
    std::lock_guard<std::mutex> lk(s->mu);
    auto it = s->jobs.find(ticket);
    if (it == s->jobs.end() || !it->second->done) return BK_ERR_PARAMS;
    delete it->second;
    s->jobs.erase(it);
    return BK_OK;
}

int bk_stream_drain(bk_stream *s)
{
    if (!s) return BK_ERR_PARAMS;
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv.wait(lk, [&] { return s->n_done + 1 == s->next_ticket; });
    int rc = BK_OK;
    for (auto &kv : s->jobs)
        if (kv.second->rc != BK_OK && rc == BK_OK) rc = kv.second->rc;
    return rc;
}

int bk_stream_get_stats(bk_stream *s, bk_stream_stats *out, int reset)
{
    if (!s || !out) return BK_ERR_PARAMS;
    std::lock_guard<std::mutex> lk(s->mu);
    *out = s->stats;
    if (reset) { s->stats = bk_stream_stats{}; s->span_open = false; }
    return BK_OK;
}

}  // extern "C"
