// bk_exchange.cpp - the path's one exchange step (include/biokanga_amd.h: bk_seq_counts, bk_seq_counts_allreduce): per-sequence accepted-read
// counts (CAligner's -O statistics, Aligner.cpp:5475-5537) summed over the contexts of a process - on the device for contexts that share
// one, through RCCL (bound on first use) between devices.
#include "bk_engine_int.h"

using namespace bk;

extern "C" {

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


}  // extern "C"
