// bk_upload.cpp - bk::upload_host (see bk_ctx_int.h): multi-threaded staged host -> device copies.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "bk_ctx_int.h"
#include "bk_cpus.h"
#include "bk_wait.h"

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

namespace {
// slices of [0, bytes) filled into page-locked staging buffers by `fill(dst, offset, n)` and sent on, two in flight per thread
template <class Fill>
int upload_staged(void *d_dst, size_t bytes, int device, int max_threads, Fill fill);
}

int bk::upload_host(void *d_dst, const void *h_src, size_t bytes, int device)
{
    if (!bytes) return BK_OK;
    HIP_TRY(hipSetDevice(device));
    if (bytes < (4u << 20) || host_is_pinned(h_src)) {
        HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
        return BK_OK;
    }
    return upload_staged(d_dst, bytes, device, kMaxThreads, [h_src](void *dst, size_t off, size_t n) { memcpy(dst, (const uint8_t *)h_src + off, n); return true; });
}

// a range of a file: read() straight into the staging buffers - no mapping whose pages would be faulted in one by one and handed back at exit
// (15.5 GB of index image: 50 GB/s against 28 GB/s through a fresh mapping)
int bk::upload_file(void *d_dst, int fd, uint64_t file_ofs, size_t bytes, int device)
{
    if (!bytes) return BK_OK;
    HIP_TRY(hipSetDevice(device));
    // (four threads reach the link's 50 GB/s at half the CPU seconds of eight: profiles/r04_c_upload_methods.txt)
    return upload_staged(d_dst, bytes, device, 4, [fd, file_ofs](void *dst, size_t off, size_t n) {
        size_t got = 0;
        while (got < n) {
            const ssize_t r = pread(fd, (uint8_t *)dst + got, n - got, (off_t)(file_ofs + off + got));
            if (r <= 0) return false;
            got += (size_t)r;
        }
        return true;
    });
}

namespace {
template <class Fill>
int upload_staged(void *d_dst, size_t bytes, int device, int max_threads, Fill fill)
{
    const size_t n_slices = (bytes + kSlice - 1) / kSlice;
    // (threads by the CPUs this process may really use - affinity mask and cgroup quota -, not by the host's hardware threads)
    const int nt = (int)std::min<size_t>((size_t)max_threads, std::min<size_t>(n_slices, (size_t)std::max(1, bk::effective_cpus() / 2)));
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&]() {
        if (hipSetDevice(device) != hipSuccess) { failed = 1; return; }
        hipStream_t st = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        void *buf[2] = {g_pool.get(), g_pool.get()};
        bool ok = buf[0] && buf[1] && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
                  bk::make_wait_event(&ev[0]) == hipSuccess && bk::make_wait_event(&ev[1]) == hipSuccess;      // (waited for asleep)
        bool used[2] = {false, false};
        for (int k = 0; ok && !failed; k ^= 1) {
            const size_t s = next.fetch_add(1);
            if (s >= n_slices) break;
            const size_t off = s * kSlice, n = std::min(kSlice, bytes - off);
            if (used[k] && bk::wait_event(ev[k]) != hipSuccess) { ok = false; break; }
            if (!fill(buf[k], off, n)) { ok = false; break; }
            ok = hipMemcpyAsync((uint8_t *)d_dst + off, buf[k], n, hipMemcpyHostToDevice, st) == hipSuccess && hipEventRecord(ev[k], st) == hipSuccess;
            used[k] = true;
        }
        if (st && bk::wait_stream(st, ev[0]) != hipSuccess) ok = false;
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
}  // namespace
