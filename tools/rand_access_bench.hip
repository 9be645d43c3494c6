// microbenchmark: rate of random 8-byte loads from tables of different sizes on MI355X, as a function
// of independent loads in flight per lane.  Informs the bound of the k_search / k_wave access pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
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
// calibration kernels for the FETCH_SIZE counter in OUR access patterns (MI355X_MICROARCH.md: the counter is
// only calibrated for wide streaming reads): a known number of random 8-byte loads, and of random
// 80-byte windows read as five 16-byte loads at a 16-byte aligned offset (eval_window's pattern for a
// 100-base read).  Run under `rocprofv3 --pmc FETCH_SIZE` with the argument `calib`.
__global__ void k_calib8(const uint64_t *__restrict__ tab, uint64_t mask, uint64_t *out, int iters)
{
    uint64_t x = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL, acc = 0;
    for (int it = 0; it < iters; it++) {
        x = x * 6364136223846793005ULL + 1442695040888963407ULL;
        acc += tab[(x >> 20) & mask];
    }
    if (acc == 0x1234567) out[0] = acc;
}
__global__ void k_calib80(const uint4 *__restrict__ tab, uint64_t mask16, uint64_t *out, int iters)
{
    uint64_t x = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL, acc = 0;
    for (int it = 0; it < iters; it++) {
        x = x * 6364136223846793005ULL + 1442695040888963407ULL;
        uint64_t i = (x >> 20) & mask16;
#pragma unroll
        for (int q = 0; q < 5; q++) { uint4 v = tab[i + q]; acc += v.x + v.w; }
    }
    if (acc == 0x1234567) out[0] = acc;
}
static int calib()
{
    uint64_t bytes = 16ULL << 30;
    uint64_t *tab, *out;
    hipMalloc(&tab, bytes + 256);
    hipMalloc(&out, 8);
    hipMemset(tab, 1, bytes + 256);
    const int blocks = 8192, iters = 64;
    hipLaunchKernelGGL(k_calib8, dim3(blocks), dim3(256), 0, 0, tab, bytes / 8 - 1, out, iters);
    hipLaunchKernelGGL(k_calib80, dim3(blocks), dim3(256), 0, 0, (const uint4 *)tab, bytes / 16 - 1, out, iters);
    hipDeviceSynchronize();
    printf("calib: %llu accesses per kernel (k_calib8: 8 B each, k_calib80: 80 B each)\n", (unsigned long long)blocks * 256 * iters);
    return 0;
}
int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "calib") return calib();
    uint64_t maxn = (16ULL << 30) / 8;
    uint64_t *tab, *out;
    hipMalloc(&tab, maxn * 8);
    hipMalloc(&out, 8);
    hipMemset(tab, 1, maxn * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint64_t bytes : {16ULL << 10, 2ULL << 20, 24ULL << 20, 128ULL << 20, 16ULL << 30}) {
        uint64_t mask = bytes / 8 - 1;
        for (int dep = 0; dep < 2; dep++)
        for (int occ : {2048, 8192}) {          // blocks of 256 threads
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
