// upload_bench - how a file of the page cache (a .sfx in /dev/shm) reaches HBM, by method, in wall-clock AND in CPU seconds (the GPU boxes
// run under a CPU quota: what a method burns is taken from the parser's threads).  Every method starts from a mapping no page of which has
// been touched, as bk_ctx_create finds it.  Run on the GPU box:
//   tools/upload_bench <file> [GiB to use]
//   staged      host threads memcpy 16 MB slices out of the mapping into pinned buffers, one HIP stream each
//   registered  the mapping is page-locked in place (hipHostRegister) in slices of 256 MB and DMA'd from there: no CPU copy of the bytes
//   pread       threads pread() into pinned slices (page cache -> pinned by the kernel's copy, no mapping) and DMA (what bk::upload_file does)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double cpu_now()
{
    rusage u;
    getrusage(RUSAGE_SELF, &u);
    return (double)u.ru_utime.tv_sec + 1e-6 * (double)u.ru_utime.tv_usec + (double)u.ru_stime.tv_sec + 1e-6 * (double)u.ru_stime.tv_usec;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: upload_bench <file> [GiB]\n"); return 1; }
    int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { perror("open"); return 1; }
    struct stat st;
    fstat(fd, &st);
    size_t bytes = (size_t)st.st_size;
    if (argc > 2) bytes = std::min(bytes, (size_t)(atof(argv[2]) * (1ULL << 30)));
    bytes &= ~((size_t)(2u << 20) - 1);
    uint8_t *d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    printf("file %s: %.2f GiB\n", argv[1], (double)bytes / (1ULL << 30));
    auto run = [&](const char *name, int nt, size_t slice, bool pinned_bufs, std::function<bool(const uint8_t *map, size_t off, size_t n, void *buf, hipStream_t s)> put) {
        const uint8_t *map = (const uint8_t *)mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
        if (map == MAP_FAILED) { perror("mmap"); return; }
        const size_t ns = (bytes + slice - 1) / slice;
        std::atomic<size_t> next{0};
        std::atomic<int> failed{0};
        std::vector<void *> bufs(2 * nt, nullptr);
        if (pinned_bufs) for (auto &b : bufs) hipHostMalloc(&b, slice, hipHostMallocDefault);
        const double t0 = now(), c0 = cpu_now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t]() {
            hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            hipEvent_t ev[2]; hipEventCreateWithFlags(&ev[0], hipEventDisableTiming); hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
            bool used[2] = {false, false};
            for (int k = 0;; k ^= 1) {
                const size_t i = next.fetch_add(1);
                if (i >= ns) break;
                const size_t off = i * slice, n = std::min(slice, bytes - off);
                if (used[k]) hipEventSynchronize(ev[k]);
                if (!put(map, off, n, bufs[2 * t + k], s)) { failed = 1; break; }
                hipEventRecord(ev[k], s);
                used[k] = true;
            }
            hipStreamSynchronize(s);
            hipStreamDestroy(s);
        });
        for (auto &x : th) x.join();
        const double dt = now() - t0, dc = cpu_now() - c0;
        const double u0 = now();
        munmap((void *)map, bytes);
        printf("%-11s %2d threads: %6.2f s  %6.2f GB/s  %6.2f CPU-seconds  (unmapping: %.3f s)%s\n", name, nt, dt, bytes / dt / 1e9, dc, now() - u0, failed ? "  FAILED" : "");
        for (auto &b : bufs) if (b) hipHostFree(b);
    };
    for (int nt : {4, 8}) {
        run("staged", nt, 16u << 20, true, [&](const uint8_t *map, size_t off, size_t n, void *buf, hipStream_t s) {
            memcpy(buf, map + off, n);
            return hipMemcpyAsync(d + off, buf, n, hipMemcpyHostToDevice, s) == hipSuccess;
        });
        run("registered", nt, 256u << 20, false, [&](const uint8_t *map, size_t off, size_t n, void *, hipStream_t s) {
            if (hipHostRegister((void *)(map + off), n, hipHostRegisterReadOnly) != hipSuccess &&
                hipHostRegister((void *)(map + off), n, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
            const bool ok = hipMemcpyAsync(d + off, map + off, n, hipMemcpyHostToDevice, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
            hipHostUnregister((void *)(map + off));
            return ok;
        });
        run("pread", nt, 16u << 20, true, [&](const uint8_t *, size_t off, size_t n, void *buf, hipStream_t s) {
            size_t got = 0;
            while (got < n) { ssize_t r = pread(fd, (uint8_t *)buf + got, n - got, (off_t)(off + got)); if (r <= 0) return false; got += (size_t)r; }
            return hipMemcpyAsync(d + off, buf, n, hipMemcpyHostToDevice, s) == hipSuccess;
        });
        run("pread 64 MB", nt, 64u << 20, true, [&](const uint8_t *, size_t off, size_t n, void *buf, hipStream_t s) {
            size_t got = 0;
            while (got < n) { ssize_t r = pread(fd, (uint8_t *)buf + got, n - got, (off_t)(off + got)); if (r <= 0) return false; got += (size_t)r; }
            return hipMemcpyAsync(d + off, buf, n, hipMemcpyHostToDevice, s) == hipSuccess;
        });
    }
    // one big DMA from pinned memory, for the PCIe ceiling
    {
        const size_t n = std::min(bytes, (size_t)4 << 30);
        void *p; hipHostMalloc(&p, n, hipHostMallocDefault);
        size_t got = 0;
        while (got < n) { ssize_t r = pread(fd, (uint8_t *)p + got, n - got, (off_t)got); if (r <= 0) break; got += (size_t)r; }
        const double t0 = now();
        hipMemcpy(d, p, n, hipMemcpyHostToDevice);
        const double dt = now() - t0;
        printf("pinned DMA  (ceiling): %6.2f GB/s\n", n / dt / 1e9);
        hipHostFree(p);
    }
    return 0;
}
