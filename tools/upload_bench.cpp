// upload_bench - how fast a file of the page cache (a .sfx in /dev/shm or on disk) reaches HBM, by method (run on the GPU box):
//   tools/upload_bench <file> [GiB to use]
//   staged      host threads copy 16 MB slices into pinned buffers, one HIP stream each (bk::upload_host, what bk_ctx_create does)
//   registered  the mapping is page-locked in place (hipHostRegister) in slices of 256 MB by a few threads and DMA'd from there:
//               no CPU copy of the bytes
//   pread       threads pread() into pinned slices (page cache -> pinned by the kernel's copy) and DMA
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
    const uint8_t *map = (const uint8_t *)mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    if (map == MAP_FAILED) { perror("mmap"); return 1; }
    uint8_t *d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    { volatile uint64_t acc = 0; for (size_t i = 0; i < bytes; i += 4096) acc += map[i]; }      // the file is in the page cache and mapped
    printf("file %s: %.2f GiB\n", argv[1], (double)bytes / (1ULL << 30));
    for (int nt : {4, 8, 16}) {
        // ---- staged
        {
            const size_t slice = 16u << 20;
            const size_t ns = (bytes + slice - 1) / slice;
            std::atomic<size_t> next{0};
            std::vector<void *> bufs(2 * nt);
            for (auto &b : bufs) hipHostMalloc(&b, slice, hipHostMallocDefault);
            const double t0 = now();
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
                    memcpy(bufs[2 * t + k], map + off, n);
                    hipMemcpyAsync(d + off, bufs[2 * t + k], n, hipMemcpyHostToDevice, s);
                    hipEventRecord(ev[k], s);
                    used[k] = true;
                }
                hipStreamSynchronize(s);
                hipStreamDestroy(s);
            });
            for (auto &x : th) x.join();
            const double dt = now() - t0;
            printf("staged      %2d threads: %6.2f s  %6.2f GB/s\n", nt, dt, bytes / dt / 1e9);
            for (auto &b : bufs) hipHostFree(b);
        }
        // ---- registered in place
        {
            const size_t slice = 256u << 20;
            const size_t ns = (bytes + slice - 1) / slice;
            std::atomic<size_t> next{0};
            std::atomic<int> failed{0};
            const double t0 = now();
            double t_reg = 0;
            std::vector<std::thread> th;
            std::vector<double> regs(nt, 0.0);
            for (int t = 0; t < nt; t++) th.emplace_back([&, t]() {
                hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= ns) break;
                    const size_t off = i * slice, n = std::min(slice, bytes - off);
                    const double r0 = now();
                    if (hipHostRegister((void *)(map + off), n, hipHostRegisterReadOnly) != hipSuccess &&
                        hipHostRegister((void *)(map + off), n, hipHostRegisterDefault) != hipSuccess) { failed = 1; (void)hipGetLastError(); break; }
                    regs[t] += now() - r0;
                    if (hipMemcpyAsync(d + off, map + off, n, hipMemcpyHostToDevice, s) != hipSuccess) failed = 1;
                    hipStreamSynchronize(s);
                    hipHostUnregister((void *)(map + off));
                }
                hipStreamDestroy(s);
            });
            for (auto &x : th) x.join();
            const double dt = now() - t0;
            for (double r : regs) t_reg += r;
            printf("registered  %2d threads: %6.2f s  %6.2f GB/s  (registering: %.2f thread-seconds)%s\n", nt, dt, bytes / dt / 1e9, t_reg, failed ? "  FAILED" : "");
        }
        // ---- pread into pinned
        {
            const size_t slice = 64u << 20;
            const size_t ns = (bytes + slice - 1) / slice;
            std::atomic<size_t> next{0};
            std::vector<void *> bufs(2 * nt);
            for (auto &b : bufs) hipHostMalloc(&b, slice, hipHostMallocDefault);
            const double t0 = now();
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
                    size_t got = 0;
                    while (got < n) { ssize_t r = pread(fd, (uint8_t *)bufs[2 * t + k] + got, n - got, (off_t)(off + got)); if (r <= 0) break; got += (size_t)r; }
                    hipMemcpyAsync(d + off, bufs[2 * t + k], n, hipMemcpyHostToDevice, s);
                    hipEventRecord(ev[k], s);
                    used[k] = true;
                }
                hipStreamSynchronize(s);
                hipStreamDestroy(s);
            });
            for (auto &x : th) x.join();
            const double dt = now() - t0;
            printf("pread       %2d threads: %6.2f s  %6.2f GB/s\n", nt, dt, bytes / dt / 1e9);
            for (auto &b : bufs) hipHostFree(b);
        }
    }
    // one big DMA from pinned memory, for the PCIe ceiling
    {
        const size_t n = std::min(bytes, (size_t)4 << 30);
        void *p; hipHostMalloc(&p, n, hipHostMallocDefault);
        memcpy(p, map, n);
        const double t0 = now();
        hipMemcpy(d, p, n, hipMemcpyHostToDevice);
        const double dt = now() - t0;
        printf("pinned DMA  (ceiling): %6.2f GB/s\n", n / dt / 1e9);
        hipHostFree(p);
    }
    return 0;
}
