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
// the window array's pattern (k_wave with DevIndex::swin): a wave reads 64 CONSECUTIVE entries per round from a random place of the
// table.  The entries of a block of 32 are stored word by word (bk_device.h, sw_word_at: 32 first 16-byte words, 32 second ones, 32
// third ones), so every load instruction of the round reads contiguous memory - the "wide coalesced 16 B/lane streaming read" the guide
// says FETCH_SIZE may report at half its size.  runs of 1 round (what most k_wave rounds are) and of 16 rounds, all three words.
__device__ __forceinline__ const uint4 *calib_word_at(const uint4 *__restrict__ tab, uint64_t idx, int q) { return tab + ((idx >> 5) * 96 + (uint64_t)q * 32 + (idx & 31)); }
template <int ROUNDS>
__device__ __forceinline__ void calib_runs_body(const uint4 *__restrict__ tab, uint64_t n_entries, uint64_t *out, int runs_per_wave)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    uint64_t x = wave * 0x9E3779B97F4A7C15ULL + 12345, acc = 0;
    for (int r = 0; r < runs_per_wave; r++) {
        x = x * 6364136223846793005ULL + 1442695040888963407ULL;
        const uint64_t start = (x >> 16) % (n_entries - (uint64_t)ROUNDS * 64 - 64);
        for (int k = 0; k < ROUNDS; k++) {
            const uint64_t idx = start + (uint64_t)k * 64 + lane;
            const uint4 a = *calib_word_at(tab, idx, 0), b = *calib_word_at(tab, idx, 1), c = *calib_word_at(tab, idx, 2);
            acc += a.x ^ b.y ^ c.z;
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_calib_runs1(const uint4 *__restrict__ tab, uint64_t n_entries, uint64_t *out, int rpw) { calib_runs_body<1>(tab, n_entries, out, rpw); }
__global__ void __launch_bounds__(256) k_calib_runs16(const uint4 *__restrict__ tab, uint64_t n_entries, uint64_t *out, int rpw) { calib_runs_body<16>(tab, n_entries, out, rpw); }
// k_wave's round with the window array, whole: of the 64 consecutive entries the words the core's window reaches into - two of the
// three for the cores of a 100-base read at offsets 0, 25 and 75, all three at 50: 36 bytes on average - AND the 64 consecutive 4-byte
// suffix array elements that go with them (40 bytes per candidate)
__global__ void __launch_bounds__(256) k_calib_wave(const uint4 *__restrict__ tab, uint64_t n_entries, const uint32_t *__restrict__ sa, uint64_t *out, int rpw)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    uint64_t x = wave * 0x9E3779B97F4A7C15ULL + 12345, acc = 0;
    for (int r = 0; r < rpw; r++) {
        x = x * 6364136223846793005ULL + 1442695040888963407ULL;
        const uint64_t idx = (x >> 16) % (n_entries - 128) + lane;
        const int core = r & 3;                                  // offsets 0, 25, 50, 75: words {1,2} {1,2} {0,1,2} {0,1}
        uint4 a = make_uint4(0, 0, 0, 0), c = a;
        if (core >= 2) a = *calib_word_at(tab, idx, 0);
        const uint4 b = *calib_word_at(tab, idx, 1);
        if (core <= 2) c = *calib_word_at(tab, idx, 2);
        acc += a.x ^ b.y ^ c.z ^ sa[idx];
    }
    if (acc == 0x1234567) out[0] = acc;
}
// .. and the plain streaming read the guide's own calibration used: every lane 16 bytes, consecutive lanes consecutive addresses
__global__ void __launch_bounds__(256) k_calib_stream16(const uint4 *__restrict__ tab, uint64_t n16, uint64_t *out)
{
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = tab[i]; acc += v.x ^ v.w; }
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
    const uint64_t n_entries = bytes / 48;
    const int waves = 16384, rpw1 = 1024, rpw16 = 64;
    hipLaunchKernelGGL(k_calib_runs1, dim3(waves / 4), dim3(256), 0, 0, (const uint4 *)tab, n_entries, out, rpw1);
    hipLaunchKernelGGL(k_calib_runs16, dim3(waves / 4), dim3(256), 0, 0, (const uint4 *)tab, n_entries, out, rpw16);
    hipLaunchKernelGGL(k_calib_stream16, dim3(16384), dim3(256), 0, 0, (const uint4 *)tab, bytes / 16, out);
    uint32_t *sa;
    hipMalloc(&sa, n_entries * 4);
    hipMemset(sa, 1, n_entries * 4);
    hipLaunchKernelGGL(k_calib_wave, dim3(waves / 4), dim3(256), 0, 0, (const uint4 *)tab, n_entries, sa, out, rpw1);
    hipDeviceSynchronize();
    printf("calib: k_calib_wave: %llu entries of 40 B (36 bytes of a 48-byte window entry + 4-byte suffix array element) in runs of 64\n", (unsigned long long)waves * rpw1 * 64);
    printf("calib: k_calib_runs1: %llu entries of 48 B in runs of 64; k_calib_runs16: %llu entries of 48 B in runs of 1024; k_calib_stream16: %llu B streamed once\n",
           (unsigned long long)waves * rpw1 * 64, (unsigned long long)waves * rpw16 * 16 * 64, (unsigned long long)bytes);
    return 0;
}
// ---- `bin`: the k_wave restructuring the round-1 review asked to be tried (candidates binned by target region so that the window
// fetch hits the L2), reduced to its memory behaviour.  Three kernels, each moving what the real thing would move per candidate:
//   k_now     one wave per read: the read's 2-bit row sits in registers, every lane fetches ONE random 32-byte window of the 2-bit
//             target (1.56 GB: two copies of 0.78 GB) and compares                       = k_wave today
//   k_binned  one lane per (read, candidate) pair record (8 bytes, streamed): the window comes from a 2 MB region (L2 resident,
//             the bin), but the read's 48-byte row has to be fetched - from wherever in the 50 M-read batch that read lies
//   k_scatter the binning pass itself: 8-byte pair records streamed in, appended to one of 390 bins (block-aggregated appends)
// Prints G candidates/s of each; at 2.44 G candidates per step the binned form costs time(k_scatter) + time(k_binned) + a replay.
__global__ void __launch_bounds__(256) k_now(const uint4 *__restrict__ tgt, uint64_t mask16, const uint4 *__restrict__ rows, uint64_t nrows,
                                             uint64_t *out, int rounds)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t x = (wave * 64 + (threadIdx.x & 63)) * 0x9E3779B97F4A7C15ULL, acc = 0;
    const uint64_t r = (wave * 0x9E3779B97F4A7C15ULL) % nrows;
    const uint4 r0 = rows[r * 3], r1 = rows[r * 3 + 1];                    // wave-uniform: the read's row
    for (int it = 0; it < rounds; it++) {
        x = x * 6364136223846793005ULL + 1442695040888963407ULL;
        const uint64_t i = (x >> 20) & mask16;
        const uint4 a = tgt[i], b = tgt[i + 1];                            // 32 bytes inside one 64-byte line (i even)
        acc += __popcll(((uint64_t)(a.x ^ r0.x) << 32) | (a.y ^ r0.y)) + __popcll(((uint64_t)(b.z ^ r1.z) << 32) | (b.w ^ r1.w));
    }
    if (acc == 0x1234567) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_binned(const uint4 *__restrict__ tgt, uint64_t region16, const uint4 *__restrict__ rows, uint64_t nrows,
                                                const uint2 *__restrict__ pairs, uint64_t npairs, uint64_t *out)
{
    uint64_t acc = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npairs; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 pr = pairs[p];                                         // {read, position inside the bin's region}
        const uint64_t r = pr.x % nrows, i = (pr.y % region16) & ~1ULL;
        const uint4 r0 = rows[r * 3], r1 = rows[r * 3 + 1];                // random row: the line this form pays for
        const uint4 a = tgt[i], b = tgt[i + 1];                            // the bin's region: 2 MB, L2 resident
        acc += __popcll(((uint64_t)(a.x ^ r0.x) << 32) | (a.y ^ r0.y)) + __popcll(((uint64_t)(b.z ^ r1.z) << 32) | (b.w ^ r1.w));
    }
    if (acc == 0x1234567) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_scatter(const uint2 *__restrict__ pairs, uint64_t npairs, uint2 *__restrict__ binned, unsigned long long *__restrict__ cursors,
                                                 uint32_t nbins, uint64_t cap)
{
    __shared__ uint32_t s_cnt[512], s_base_lo[512];
    for (uint64_t base = (uint64_t)blockIdx.x * 4096; base < npairs; base += (uint64_t)gridDim.x * 4096) {
        for (uint32_t q = threadIdx.x; q < nbins; q += 256) s_cnt[q] = 0;
        __syncthreads();
        uint2 v[16];
        uint32_t off[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint64_t p = base + (uint64_t)k * 256 + threadIdx.x;
            v[k] = p < npairs ? pairs[p] : make_uint2(0, 0);
            off[k] = p < npairs ? atomicAdd(&s_cnt[v[k].y % nbins], 1u) : 0;
        }
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < nbins; q += 256)
            s_base_lo[q] = s_cnt[q] ? (uint32_t)atomicAdd(&cursors[q], (unsigned long long)s_cnt[q]) : 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint64_t p = base + (uint64_t)k * 256 + threadIdx.x;
            if (p < npairs) { const uint32_t bq = v[k].y % nbins; binned[(uint64_t)bq * cap + s_base_lo[bq] + off[k]] = v[k]; }
        }
        __syncthreads();
    }
}
static int bin_bench()
{
    const uint64_t tgt_bytes = 1ULL << 31, rows_n = 100000000ULL, npairs = 1ULL << 30;      // 2 GB of 2-bit target (two copies), 100 M rows of 48 B
    const uint32_t nbins = 390;
    const uint64_t cap = npairs / nbins * 5 / 4 + 4096;
    uint4 *tgt, *rows;
    uint2 *pairs, *binned;
    uint64_t *out;
    unsigned long long *cursors;
    hipMalloc(&tgt, tgt_bytes + 256); hipMalloc(&rows, rows_n * 48 + 256); hipMalloc(&pairs, npairs * 8); hipMalloc(&binned, (uint64_t)nbins * cap * 8);
    hipMalloc(&out, 8); hipMalloc(&cursors, nbins * 8);
    hipMemset(tgt, 1, tgt_bytes); hipMemset(rows, 2, rows_n * 48); hipMemset(cursors, 0, nbins * 8);
    // pair records: pseudo-random read and position
    {
        std::vector<uint2> h(1 << 20);
        uint64_t x = 88172645463325252ULL;
        for (auto &e : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; e.x = (uint32_t)(x >> 11); e.y = (uint32_t)(x >> 33); }
        for (uint64_t at = 0; at < npairs; at += h.size()) hipMemcpy(pairs + at, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    const int rounds = 64, blocks = 8192;             // 8192 x 4 waves, 64 candidates per lane
    hipLaunchKernelGGL(k_now, dim3(blocks), dim3(256), 0, 0, tgt, tgt_bytes / 16 - 2, rows, rows_n, out, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_now, dim3(blocks), dim3(256), 0, 0, tgt, tgt_bytes / 16 - 2, rows, rows_n, out, rounds);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double n_now = (double)blocks * 256 * rounds;
    printf("bin: k_now     (row in registers, window random over %.2f GB): %6.1f G candidates/s\n", tgt_bytes / 1e9, n_now / ms / 1e6);
    hipLaunchKernelGGL(k_binned, dim3(blocks), dim3(256), 0, 0, tgt, (uint64_t)(2 << 20) / 16, rows, rows_n, pairs, npairs / 64, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_binned, dim3(blocks), dim3(256), 0, 0, tgt, (uint64_t)(2 << 20) / 16, rows, rows_n, pairs, npairs, out);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("bin: k_binned  (window in a 2 MB region, row random over %.1f GB, pair records streamed): %6.1f G candidates/s\n", rows_n * 48 / 1e9, (double)npairs / ms / 1e6);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_scatter, dim3(4096), dim3(256), 0, 0, pairs, npairs, binned, cursors, nbins, cap);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("bin: k_scatter (8-byte pair records into %u bins): %6.1f G records/s (%.0f GB/s read + written)\n", nbins, (double)npairs / ms / 1e6, (double)npairs * 16 / ms / 1e6);
    return 0;
}
// What do the per-block global atomics of the list-appending kernels cost?  200 000 blocks of 256 threads (k_flat's grid for 50 M
// reads), each doing a little LDS work and then `nat` atomics: all blocks on the same words of one line (what the kernels did),
// spread over `stripes` lines, or none.  `ret` != 0: one lane waits for the returned value (a list append needs its base).
__global__ void __launch_bounds__(256) k_atomics(unsigned long long *__restrict__ ctr, int nat, int stripes, int ret, uint64_t *out)
{
    __shared__ uint32_t s_x[256];
    s_x[threadIdx.x] = threadIdx.x * 2654435761u;
    __syncthreads();
    uint32_t v = s_x[(threadIdx.x * 7 + 3) & 255];
    __syncthreads();
    unsigned long long got = 0;
    if (threadIdx.x < (unsigned)nat) {
        unsigned long long *p = ctr + (uint64_t)(blockIdx.x % (unsigned)stripes) * 8 + threadIdx.x;
        if (ret) got = atomicAdd(p, (unsigned long long)(v & 3)); else atomicAdd(p, (unsigned long long)(v & 3));
    }
    __syncthreads();
    if (got == 0x123456789ULL) out[0] = got;
}
static int atomic_bench()
{
    unsigned long long *ctr;
    uint64_t *out;
    hipMalloc(&ctr, 4096 * 64); hipMalloc(&out, 8);
    hipMemset(ctr, 0, 4096 * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 200000;
    for (int ret = 0; ret < 2; ret++)
        for (int nat : {0, 1, 3, 7})
            for (int stripes : {1, 64, 4096}) {
                if (nat == 0 && stripes > 1) continue;
                hipLaunchKernelGGL(k_atomics, dim3(blocks), dim3(256), 0, 0, ctr, nat, stripes, ret, out);
                hipEventRecord(e0);
                for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(k_atomics, dim3(blocks), dim3(256), 0, 0, ctr, nat, stripes, ret, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("atomic: %d blocks, %d atomics each on %4d line(s), value %s: %7.3f ms per launch\n", blocks, nat, stripes, ret ? "awaited" : "dropped", ms / 5);
            }
    return 0;
}
// "runs": what a suffix-ordered copy of the target windows would buy the wave kernel.  Today a candidate costs a coalesced 4-byte
// suffix array element plus ONE RANDOM 64-byte line (its 25-byte window in the 2-bit target).  With the windows of all suffixes
// stored in suffix array order (48 bytes each) the candidates of one core interval are a contiguous run: a wave reads 64 entries
// = 3 KB per round from a random place of a 140 GB array.  k_runs: one wave per run of `rounds` x 64 entries, each lane its
// entry's three 16-byte words; k_lines: the same number of candidates as random 64-byte lines out of a 1.5 GB table (+ the
// coalesced element).
__global__ void __launch_bounds__(256) k_runs(const uint4 *__restrict__ tab, uint64_t n_entries, uint64_t *out, int rounds, int runs_per_wave)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    uint64_t x = wave * 0x9E3779B97F4A7C15ULL + 12345, acc = 0;
    for (int r = 0; r < runs_per_wave; r++) {
        x = x * 6364136223846793005ULL + 1442695040888963407ULL;
        const uint64_t start = (x >> 16) % (n_entries - (uint64_t)rounds * 64);
        for (int k = 0; k < rounds; k++) {
            const uint4 *e = tab + (start + (uint64_t)k * 64 + lane) * 3;
            const uint4 a = e[0], b = e[1], c = e[2];
            acc += a.x ^ b.y ^ c.z;
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_lines(const uint4 *__restrict__ tab, uint64_t n_lines, const uint32_t *__restrict__ sa, uint64_t n_sa, uint64_t *out,
                                               int rounds, int runs_per_wave)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    uint64_t x = wave * 0x9E3779B97F4A7C15ULL + 12345, acc = 0;
    for (int r = 0; r < runs_per_wave; r++) {
        x = x * 6364136223846793005ULL + 1442695040888963407ULL;
        const uint64_t start = (x >> 16) % (n_sa - (uint64_t)rounds * 64);
        for (int k = 0; k < rounds; k++) {
            const uint64_t t = ((uint64_t)sa[start + (uint64_t)k * 64 + lane] * 0x9E3779B97F4A7C15ULL >> 20) % n_lines;
            const uint4 *e = tab + t * 4;
            const uint4 a = e[0], b = e[1];
            acc += a.x ^ b.y;
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}
static int runs_bench()
{
    const uint64_t n_entries = 2900000000ULL;                 // x 48 B = 139 GB
    uint4 *tab; uint64_t *out; uint32_t *sa;
    if (hipMalloc(&tab, n_entries * 48) != hipSuccess) { printf("runs: cannot allocate %llu GB\n", (unsigned long long)(n_entries * 48 >> 30)); return 1; }
    hipMalloc(&out, 8);
    const uint64_t n_sa = 1ULL << 30;
    hipMalloc(&sa, n_sa * 4);
    hipMemset(sa, 0x5a, n_sa * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rounds : {1, 2, 4, 16, 64}) {
        for (int waves : {8192, 16384}) {
            const int rpw = 4096 / rounds;
            float ms = 0;
            hipLaunchKernelGGL(k_runs, dim3(waves / 4), dim3(256), 0, 0, tab, n_entries, out, rounds, rpw);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_runs, dim3(waves / 4), dim3(256), 0, 0, tab, n_entries, out, rounds, rpw);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            const double cands = (double)waves * rpw * rounds * 64;
            printf("runs: %2d rounds per run, %5d waves: %6.1f G candidates/s (%5.2f TB/s of 48-byte entries)\n", rounds, waves, cands / ms / 1e6, cands * 48 / ms / 1e9);
            hipLaunchKernelGGL(k_lines, dim3(waves / 4), dim3(256), 0, 0, tab, (3ULL << 29) / 64, sa, n_sa, out, rounds, rpw);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_lines, dim3(waves / 4), dim3(256), 0, 0, tab, (3ULL << 29) / 64, sa, n_sa, out, rounds, rpw);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("lines:%2d rounds per run, %5d waves: %6.1f G candidates/s (random 64-byte line + coalesced element each)\n", rounds, waves, cands / ms / 1e6);
        }
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "runs") return runs_bench();
    if (argc > 1 && std::string(argv[1]) == "atomic") return atomic_bench();
    if (argc > 1 && std::string(argv[1]) == "calib") return calib();
    if (argc > 1 && std::string(argv[1]) == "bin") return bin_bench();
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
