// Which engine moves a device-to-host hipMemcpyAsync into pinned memory, and what it costs the kernels that run beside it: a 1 GB copy on
// one stream, a kernel that fills the chip on another, each timed alone and together.  A copy made by the runtime's shader kernel
// (__amd_rocclr_copyBuffer in a kernel trace) takes wave slots for as long as PCIe needs; one made by an SDMA engine takes none.
//   hipcc -O2 --offload-arch=gfx950 tools/d2h_engine_probe.cpp -o build/d2h_engine_probe
//   GPU_BLIT_ENGINE_TYPE=.. HSA_.. build/d2h_engine_probe [MB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_busy(const uint32_t *__restrict__ tab, uint32_t mask, uint32_t *__restrict__ out, int rounds)
{
    // dependent random loads, eight waves per SIMD: the shape of the search kernels
    uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        i = tab[(i >> 4) & mask] + i * 1664525u + 1013904223u;
        acc ^= i;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 1000;
    const size_t bytes = mb << 20;
    void *d = nullptr, *h = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CK(hipMemset(d, 1, bytes));
    memset(h, 0, bytes);
    const uint32_t tab_words = 1u << 28;                      // 1 GB table
    uint32_t *tab = nullptr, *out = nullptr;
    CK(hipMalloc(&tab, (size_t)tab_words * 4));
    CK(hipMemset(tab, 7, (size_t)tab_words * 4));
    const int blocks = 256 * 8 * 4;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipStream_t s_k, s_c;
    CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_c, hipStreamNonBlocking));
    hipEvent_t k0, k1, c0, c1;
    CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
    const int rounds = 1500;
    auto kernel = [&]() { hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, s_k, tab, tab_words - 1, out, rounds); };
    // warm-up
    kernel();
    CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s_c));
    CK(hipDeviceSynchronize());
    float ms_k = 0, ms_c = 0, ms_k2 = 0, ms_c2 = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(k0, s_k)); kernel(); CK(hipEventRecord(k1, s_k));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms_k, k0, k1));
        CK(hipEventRecord(c0, s_c)); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s_c)); CK(hipEventRecord(c1, s_c));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms_c, c0, c1));
        const double t0 = now();
        CK(hipEventRecord(k0, s_k)); kernel(); CK(hipEventRecord(k1, s_k));
        CK(hipEventRecord(c0, s_c)); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s_c)); CK(hipEventRecord(c1, s_c));
        CK(hipDeviceSynchronize());
        const double both = (now() - t0) * 1e3;
        CK(hipEventElapsedTime(&ms_k2, k0, k1));
        CK(hipEventElapsedTime(&ms_c2, c0, c1));
        printf("rep %d: kernel alone %.2f ms, copy of %zu MB alone %.2f ms (%.1f GB/s); together: kernel %.2f ms, copy %.2f ms, both done after %.2f ms\n",
               rep, ms_k, mb, ms_c, bytes / ms_c / 1e6, ms_k2, ms_c2, both);
    }
    const char *names[] = {"GPU_BLIT_ENGINE_TYPE", "HSA_ENABLE_SDMA", "HSA_FORCE_SDMA_SIZE", "HSA_ENABLE_SDMA_COPY_SIZE_OVERRIDE", "DEBUG_CLR_LIMIT_BLIT_WG", "GPU_FORCE_BLIT_COPY_SIZE", "HSA_REV_COPY_DIR"};
    for (const char *n : names) if (getenv(n)) printf("  %s=%s\n", n, getenv(n));
    return 0;
}
