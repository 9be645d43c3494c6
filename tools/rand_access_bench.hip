// microbenchmark: rate of random 8-byte loads from tables of different sizes on MI355X, as a function
// of independent loads in flight per lane.  Informs the bound of the k_search / k_wave access pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int ILP>
__global__ void k_rand(const uint64_t *__restrict__ tab, uint64_t mask, uint64_t *out, int iters, int dependent)
{
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x[ILP];
    for (int k = 0; k < ILP; k++) x[k] = (tid * 0x9E3779B97F4A7C15ULL + k * 0xD1B54A32D192ED03ULL) ;
    uint64_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint64_t v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            x[k] = x[k] * 6364136223846793005ULL + 1442695040888963407ULL;
            v[k] = tab[(x[k] >> 20) & mask];
        }
#pragma unroll
        for (int k = 0; k < ILP; k++) { acc += v[k]; if (dependent) x[k] ^= v[k]; }
    }
    if (acc == 0x1234567) out[0] = acc;
}
int main()
{
    uint64_t maxn = (16ULL << 30) / 8;
    uint64_t *tab, *out;
    hipMalloc(&tab, maxn * 8);
    hipMalloc(&out, 8);
    hipMemset(tab, 1, maxn * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint64_t bytes : {64ULL << 20, 1ULL << 30, 16ULL << 30}) {
        uint64_t mask = bytes / 8 - 1;
        for (int dep = 0; dep < 2; dep++)
        for (int occ : {1024, 2048, 4096, 8192}) {          // blocks of 256 threads
            auto run = [&](int ilp, auto kern) {
                int iters = 64;
                hipLaunchKernelGGL(kern, dim3(occ), dim3(256), 0, 0, tab, mask, out, 4, dep);
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(occ), dim3(256), 0, 0, tab, mask, out, iters, dep);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                double loads = (double)occ * 256 * iters * ilp;
                printf("table %6llu MB dep %d blocks %5d ilp %d: %7.2f G loads/s (%.0f GB/s of 64B lines)\n", (unsigned long long)(bytes >> 20), dep, occ, ilp, loads / ms / 1e6, loads / ms / 1e6 * 64);
            };
            run(1, k_rand<1>); run(4, k_rand<4>); run(8, k_rand<8>);
        }
    }
    return 0;
}
