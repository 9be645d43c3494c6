// alloc_bench.cpp - what a large HBM allocation costs on this box, by method (round 5: the window array's set-up time is its allocation).
//   hipcc -O2 -o tools/alloc_bench tools/alloc_bench.cpp -lpthread ;  tools/alloc_bench [GB = 32]
// Methods: one hipMalloc; the same bytes as T hipMallocs from T threads; virtual range + physical chunks (hipMemCreate / hipMemMap) from
// one and from T threads; hipMallocAsync out of a pool that keeps what it is given back.  Every allocation is then written once
// (hipMemsetAsync) to see whether the first touch costs more than the later ones.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(e_)); return; } } while (0)

static void touch(void *p, size_t bytes, const char *what)
{
    for (int rep = 0; rep < 2; rep++) {
        const double t0 = now();
        (void)hipMemsetAsync(p, rep, bytes, 0);
        (void)hipDeviceSynchronize();
        const double t = now() - t0;
        printf("    %s: write %d of %.0f GB in %.3f s = %.0f GB/s\n", what, rep, bytes / 1e9, t, bytes / 1e9 / t);
    }
}

static void plain(size_t bytes)
{
    void *p = nullptr;
    double t0 = now();
    CK(hipMalloc(&p, bytes));
    double t1 = now();
    printf("  hipMalloc %.0f GB: %.3f s (%.1f ms per GB)\n", bytes / 1e9, t1 - t0, (t1 - t0) * 1e3 / (bytes / 1e9));
    touch(p, bytes, "hipMalloc");
    t0 = now();
    CK(hipFree(p));
    printf("  hipFree: %.3f s\n", now() - t0);
}

static void threaded(size_t bytes, int T)
{
    std::vector<void *> p(T, nullptr);
    std::vector<std::thread> th;
    double t0 = now();
    for (int i = 0; i < T; i++) th.emplace_back([&, i]() { (void)hipSetDevice(0); if (hipMalloc(&p[i], bytes / T) != hipSuccess) p[i] = nullptr; });
    for (auto &t : th) t.join();
    double t1 = now();
    printf("  %d threads x hipMalloc %.1f GB: %.3f s\n", T, bytes / 1e9 / T, t1 - t0);
    t0 = now();
    th.clear();
    for (int i = 0; i < T; i++) th.emplace_back([&, i]() { (void)hipSetDevice(0); if (p[i]) (void)hipFree(p[i]); });
    for (auto &t : th) t.join();
    printf("  %d threads x hipFree: %.3f s\n", T, now() - t0);
}

static void vmm(size_t bytes, int T, size_t chunk)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t nchunk = (bytes + chunk - 1) / chunk;
    void *base = nullptr;
    double t0 = now();
    CK(hipMemAddressReserve(&base, nchunk * chunk, 0, nullptr, 0));
    const double t_res = now() - t0;
    std::vector<hipMemGenericAllocationHandle_t> h(nchunk);
    std::vector<int> ok(nchunk, 0);
    t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            (void)hipSetDevice(0);
            for (size_t i = t; i < nchunk; i += T) {
                if (hipMemCreate(&h[i], chunk, &prop, 0) != hipSuccess) continue;
                if (hipMemMap((char *)base + i * chunk, chunk, 0, h[i], 0) != hipSuccess) continue;
                hipMemAccessDesc ad = {};
                ad.location = prop.location;
                ad.flags = hipMemAccessFlagsProtReadWrite;
                if (hipMemSetAccess((char *)base + i * chunk, chunk, &ad, 1) != hipSuccess) continue;
                ok[i] = 1;
            }
        });
    for (auto &t : th) t.join();
    const double t_map = now() - t0;
    size_t good = 0;
    for (int v : ok) good += v;
    printf("  virtual range (granularity %zu MB) reserve %.3f s; %zu chunks of %.2f GB created + mapped by %d thread(s): %.3f s (%zu ok)\n", gran >> 20, t_res, nchunk,
           chunk / 1e9, T, t_map, good);
    if (good == nchunk) touch(base, nchunk * chunk, "mapped range");
    t0 = now();
    for (size_t i = 0; i < nchunk; i++)
        if (ok[i]) { (void)hipMemUnmap((char *)base + i * chunk, chunk); (void)hipMemRelease(h[i]); }
    (void)hipMemAddressFree(base, nchunk * chunk);
    printf("  unmap + release: %.3f s\n", now() - t0);
}

static void pool(size_t bytes)
{
    hipMemPool_t mp;
    CK(hipDeviceGetDefaultMemPool(&mp, 0));
    uint64_t keep = ~0ULL;
    CK(hipMemPoolSetAttribute(mp, hipMemPoolAttrReleaseThreshold, &keep));
    for (int rep = 0; rep < 2; rep++) {
        void *p = nullptr;
        double t0 = now();
        CK(hipMallocAsync(&p, bytes, 0));
        CK(hipStreamSynchronize(0));
        double t1 = now();
        printf("  hipMallocAsync %.0f GB, time %d: %.3f s\n", bytes / 1e9, rep, t1 - t0);
        t0 = now();
        CK(hipFreeAsync(p, 0));
        CK(hipStreamSynchronize(0));
        printf("  hipFreeAsync: %.3f s\n", now() - t0);
    }
    CK(hipMemPoolTrimTo(mp, 0));
}

int main(int argc, char **argv)
{
    const size_t gb = argc > 1 ? (size_t)atoll(argv[1]) : 32;
    const size_t bytes = gb << 30;
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    size_t fr = 0, tot = 0;
    (void)hipMemGetInfo(&fr, &tot);
    printf("alloc_bench: %zu GB; device has %.0f of %.0f GB free\n", gb, fr / 1e9, tot / 1e9);
    printf("one call\n");
    plain(bytes);
    plain(bytes);
    printf("threads\n");
    threaded(bytes, 4);
    threaded(bytes, 8);
    printf("virtual memory management\n");
    vmm(bytes, 1, (size_t)2 << 30);
    vmm(bytes, 4, (size_t)2 << 30);
    vmm(bytes, 8, (size_t)1 << 30);
    printf("stream-ordered pool\n");
    pool(bytes);
    return 0;
}
