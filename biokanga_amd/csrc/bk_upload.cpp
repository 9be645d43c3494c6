// bk_upload.cpp - bk::upload_host (see bk_ctx_int.h): multi-threaded staged host -> device copies.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "bk_ctx_int.h"

namespace {

constexpr size_t kSlice = 16u << 20;        // bytes per staged slice
constexpr int kMaxThreads = 8;

// pinned staging buffers, kept for the life of the process (allocating page-locked memory is slow)
struct Pool {
    std::mutex mu;
    std::vector<void *> free_;
    void *get()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!free_.empty()) { void *p = free_.back(); free_.pop_back(); return p; }
        }
        void *p = nullptr;
        if (hipHostMalloc(&p, kSlice, hipHostMallocPortable) != hipSuccess) return nullptr;
        return p;
    }
    void put(void *p)
    {
        std::lock_guard<std::mutex> lk(mu);
        free_.push_back(p);
    }
};
Pool g_pool;

}  // namespace

bool bk::host_is_pinned(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

int bk::upload_host(void *d_dst, const void *h_src, size_t bytes, int device)
{
    if (!bytes) return BK_OK;
    HIP_TRY(hipSetDevice(device));
    if (bytes < (4u << 20) || host_is_pinned(h_src)) {
        HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
        return BK_OK;
    }
    const size_t n_slices = (bytes + kSlice - 1) / kSlice;
    const int nt = (int)std::min<size_t>((size_t)kMaxThreads, std::min<size_t>(n_slices, std::max(1u, std::thread::hardware_concurrency() / 2)));
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&]() {
        if (hipSetDevice(device) != hipSuccess) { failed = 1; return; }
        hipStream_t st = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        void *buf[2] = {g_pool.get(), g_pool.get()};
        bool ok = buf[0] && buf[1] && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
        bool used[2] = {false, false};
        for (int k = 0; ok && !failed; k ^= 1) {
            const size_t s = next.fetch_add(1);
            if (s >= n_slices) break;
            const size_t off = s * kSlice, n = std::min(kSlice, bytes - off);
            if (used[k] && hipEventSynchronize(ev[k]) != hipSuccess) { ok = false; break; }
            memcpy(buf[k], (const uint8_t *)h_src + off, n);
            ok = hipMemcpyAsync((uint8_t *)d_dst + off, buf[k], n, hipMemcpyHostToDevice, st) == hipSuccess && hipEventRecord(ev[k], st) == hipSuccess;
            used[k] = true;
        }
        if (st && hipStreamSynchronize(st) != hipSuccess) ok = false;
        if (!ok) failed = 1;
        for (int k = 0; k < 2; k++) { if (ev[k]) (void)hipEventDestroy(ev[k]); if (buf[k]) g_pool.put(buf[k]); }
        if (st) (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (failed) { fprintf(stderr, "biokanga_amd: staged host -> device copy failed\n"); return BK_ERR_INTERNAL; }
    return BK_OK;
}
