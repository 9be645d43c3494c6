// bk_sa_build.hip - suffix array construction on the MI355X (replaces CSfxArrayV3::QSortSeq,
// libbiokanga/SfxArrayV2.cpp:9451-9542, which quick-sorts suffix offsets with a byte comparator).
//
// Order produced = the reference comparator's order (QSortSeqCmp32/40): lexicographic on the low
// nibble of each base (A0 C1 G2 T3 N4 < EOS7), comparison NOT stopped at EOS, bounded at
// 5*cMaxReadLen = 983 040 bases (ties beyond that are unordered in the reference; here: stable).
// Positions past the end of the concatenation compare as a sentinel below every base (the reference
// reads whatever memory follows - the suffix array being sorted; only suffixes already equal up to
// and including an EOS are affected, and the search comparator never looks past an EOS, so any
// tie-break is search-equivalent.  The sentinel reproduces what the reference wrote for the
// golden fixtures: the final EOS suffix sorts before the other EOS suffixes).
//
// Method: prefix doubling.  Round 0 sorts suffixes by their first 16 bases (one 64-bit nibble
// word); round r sorts by (rank[i], rank[i+h]), h = 16*2^(r-1), with rocPRIM's LSD radix sort
// (rocPRIM radix sort) - bandwidth-bound streaming passes over HBM.  Stops when all ranks are
// distinct or h >= 2^20.
#include <hip/hip_runtime.h>
#include "bk_prim.h"
#include "bk_env.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

namespace bk {

static inline unsigned grid_for(uint64_t n)
{
    uint64_t b = (n + 255) / 256;
    return (unsigned)(b > 262144 ? 262144 : (b ? b : 1));
}

#define SA_TRY(expr)                                                                               \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            fprintf(stderr, "biokanga_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            rc = e__ == hipErrorOutOfMemory ? -95 : -1;                                            \
            goto done;                                                                             \
        }                                                                                          \
    } while (0)

__global__ void k_sa_init(const uint8_t *__restrict__ seq, uint64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t v = 0;
        for (int k = 0; k < 16; k++) {
            uint64_t p = i + k;
            uint64_t nb = p < n ? (uint64_t)(seq[p] & 0x0f) + 1 : 0ULL;   // 0 = end sentinel, below every base
            v = (v << 4) | nb;
        }
        key[i] = v;
        val[i] = (uint32_t)i;
    }
}

// head[j] = j where a new group of equal keys starts, else 0
__global__ void k_sa_heads(const uint64_t *__restrict__ key, uint64_t n, uint32_t *__restrict__ head,
                           unsigned long long *__restrict__ n_groups)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long cnt = 0;
    for (; j < n; j += stride) {
        bool h = j == 0 || key[j] != key[j - 1];
        head[j] = h ? (uint32_t)j : 0u;
        cnt += h;
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(n_groups, cnt);
}

__global__ void k_sa_scatter_rank(const uint32_t *__restrict__ grp, const uint32_t *__restrict__ val, uint64_t n,
                                  uint32_t *__restrict__ rank)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; j < n; j += stride) rank[val[j]] = grp[j];
}

__global__ void k_sa_make_keys(const uint32_t *__restrict__ rank, const uint32_t *__restrict__ val, uint64_t n, uint64_t h,
                               uint64_t *__restrict__ key)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; j < n; j += stride) {
        uint64_t i = val[j];
        uint64_t r1 = rank[i];
        uint64_t r2 = i + h < n ? (uint64_t)rank[i + h] + 1 : 0ULL;         // past the end sorts first (sentinel)
        key[j] = (r1 << 32) | r2;
    }
}

__global__ void k_sa_write5(const uint32_t *__restrict__ val, uint64_t n, uint8_t *__restrict__ out)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; j < n; j += stride) {
        uint32_t v = val[j];
        uint8_t *p = out + j * 5;
        p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); p[4] = 0;
    }
}


// ------------------------------------------------------------------------------------------------
// Suffix arrays of 2^32 and more bases (5-byte elements; order of QSortSeqCmp40, SfxArrayV2.cpp:9517-9542 - the same comparison
// as QSortSeqCmp32 over 40-bit offsets).  Same prefix doubling, laid out so that a 17 Gbp genome fits one 288 GB device:
//   * suffix array and ranks are kept as 32 + 8 bit planes (5 bytes per suffix each); nothing of size n is ever held as 64-bit
//   * a round never sorts the whole array at once.  After round 0 the array is sorted by rank (= index of the group's first member),
//     so a round is a sort of (rank[i], rank[i+h]) WITHIN contiguous stretches of it: the array is walked in chunks of at most 2^g
//     suffix-array indexes cut at group boundaries, the key is (rank[i] - chunk start) << r | (rank[i+h] + 1) with r = bits of n and
//     g = 64 - r (one 64-bit radix sort of the chunk), and chunks holding no group of two or more are skipped
//   * ranks are refined in place chunk after chunk: a later chunk of the same round may already see the finer rank of i+h - that
//     only ever sorts by MORE than 2h bases, never against the true order (Larsson & Sadakane's observation)
//   * round 0 (16 bases per key, as below) runs bucket by bucket: the suffixes are counted by their first four bases, consecutive
//     buckets are batched up to the chunk size, each batch's positions are selected from the text in position order and sorted
// The stretch buffers (2 x (key + value) x 2^g... capped at kWideChunk elements) are the only scratch besides the planes.
namespace {

constexpr uint64_t kWideChunkDefault = 1ULL << 29;

__device__ __forceinline__ uint64_t get40(const uint32_t *lo, const uint8_t *hi, uint64_t i) { return (uint64_t)lo[i] | ((uint64_t)hi[i] << 32); }
__device__ __forceinline__ void put40(uint32_t *lo, uint8_t *hi, uint64_t i, uint64_t v) { lo[i] = (uint32_t)v; hi[i] = (uint8_t)(v >> 32); }

__device__ __forceinline__ uint32_t code4_at(const uint8_t *__restrict__ seq, uint64_t n, uint64_t i)
{
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t p = i + k;
        c = (c << 4) | (p < n ? (uint32_t)(seq[p] & 0x0f) + 1u : 0u);
    }
    return c;
}

__global__ void __launch_bounds__(256) k_w_hist(const uint8_t *__restrict__ seq, uint64_t n, unsigned long long *__restrict__ hist)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&hist[code4_at(seq, n, i)], 1ULL);
}

struct InBuckets {
    const uint8_t *seq;
    uint64_t n;
    uint32_t lo, hi;
    __host__ __device__ bool operator()(const unsigned long long &i) const
    {
        uint32_t c = 0;
        for (int k = 0; k < 4; k++) {
            const uint64_t p = i + k;
            c = (c << 4) | (p < n ? (uint32_t)(seq[p] & 0x0f) + 1u : 0u);
        }
        return c >= lo && c <= hi;
    }
};

__global__ void k_w_keys0(const uint8_t *__restrict__ seq, uint64_t n, const unsigned long long *__restrict__ pos, uint64_t m,
                          unsigned long long *__restrict__ key)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = pos[j];
        uint64_t v = 0;
        for (int k = 0; k < 16; k++) {
            const uint64_t p = i + k;
            v = (v << 4) | (p < n ? (uint64_t)(seq[p] & 0x0f) + 1 : 0ULL);
        }
        key[j] = v;
    }
}

// head[j] = chunk-local index j where a new group of equal keys starts, else 0 (element 0 always starts one)
__global__ void k_w_heads(const unsigned long long *__restrict__ key, uint64_t m, unsigned long long *__restrict__ head)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x)
        head[j] = (j == 0 || key[j] != key[j - 1]) ? j : 0ULL;
}

// the sorted stretch goes back into the planes: SA[j0 + j] = pos[j], rank[pos[j]] = j0 + (index of its group's first member);
// counts the members of groups of two or more (what a later round still has to look at)
__global__ void k_w_store(const unsigned long long *__restrict__ pos, const unsigned long long *__restrict__ grp, uint64_t m, uint64_t j0,
                          uint32_t *__restrict__ sa_lo, uint8_t *__restrict__ sa_hi, uint32_t *__restrict__ rk_lo, uint8_t *__restrict__ rk_hi,
                          unsigned long long *__restrict__ n_tied)
{
    unsigned long long tied = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = pos[j], g = grp[j];
        put40(sa_lo, sa_hi, j0 + j, p);
        put40(rk_lo, rk_hi, p, j0 + g);
        tied += (g != j) || (j + 1 < m && grp[j + 1] == g);
    }
    for (int off = 32; off > 0; off >>= 1) tied += __shfl_down(tied, off);
    if ((threadIdx.x & 63) == 0 && tied) atomicAdd(n_tied, tied);
}

// keys of a round over the stretch [j0, j0 + m): (group start - j0) << rbits | rank[i + h] + 1 (0 past the end); single != 0: the
// stretch is one group, the key is the second part alone.  Also counts the members of groups of two or more.
__global__ void k_w_keys(const uint32_t *__restrict__ sa_lo, const uint8_t *__restrict__ sa_hi, const uint32_t *__restrict__ rk_lo,
                         const uint8_t *__restrict__ rk_hi, uint64_t n, uint64_t j0, uint64_t m, uint64_t h, int rbits, int single,
                         unsigned long long *__restrict__ key, unsigned long long *__restrict__ val, unsigned long long *__restrict__ n_tied)
{
    unsigned long long tied = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = get40(sa_lo, sa_hi, j0 + j);
        const uint64_t g = get40(rk_lo, rk_hi, i);
        const uint64_t r2 = i + h < n ? get40(rk_lo, rk_hi, i + h) + 1 : 0ULL;
        key[j] = single ? r2 : (((g - j0) << rbits) | r2);
        val[j] = i;
        bool t = g != j0 + j;
        if (!t && j0 + j + 1 < n) t = get40(rk_lo, rk_hi, get40(sa_lo, sa_hi, j0 + j + 1)) == g;
        tied += t;
    }
    for (int off = 32; off > 0; off >>= 1) tied += __shfl_down(tied, off);
    if ((threadIdx.x & 63) == 0 && tied) atomicAdd(n_tied, tied);
}

// first index >= from whose suffix starts a group (rank[SA[j]] == j), searched 64 at a time by one wave; n when there is none
__global__ void k_w_next_head(const uint32_t *__restrict__ sa_lo, const uint8_t *__restrict__ sa_hi, const uint32_t *__restrict__ rk_lo,
                              const uint8_t *__restrict__ rk_hi, uint64_t n, uint64_t from, unsigned long long *__restrict__ out)
{
    for (uint64_t base = from; base < n; base += 64) {
        const uint64_t j = base + threadIdx.x;
        const bool head = j < n && get40(rk_lo, rk_hi, get40(sa_lo, sa_hi, j)) == j;
        const uint64_t m = __ballot(head);
        if (m) { if (threadIdx.x == 0) *out = base + (uint64_t)__builtin_ctzll(m); return; }
    }
    if (threadIdx.x == 0) *out = n;
}

__global__ void k_w_write5(const uint32_t *__restrict__ sa_lo, const uint8_t *__restrict__ sa_hi, uint64_t n, uint8_t *__restrict__ out)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = sa_lo[j];
        uint8_t *p = out + j * 5;
        p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); p[4] = sa_hi[j];
    }
}

}  // namespace

// chunk_cap: elements per stretch (0 = default 2^29); the test-suite forces small values on small inputs to run every branch
int build_sa_device_wide(const uint8_t *d_seq, uint64_t n, void *d_sa_out, int el_size, uint64_t chunk_cap, hipStream_t s)
{
    int rc = 0;
    if (n >= (1ULL << 40) || (el_size == 4 && n > 0xFFFFFFFFULL)) return -100;
    int rbits = 1;
    while ((1ULL << rbits) < n + 2) rbits++;
    const int gbits = 64 - rbits;
    uint64_t cap = chunk_cap ? chunk_cap : kWideChunkDefault;
    if (cap > (1ULL << (gbits - 1))) cap = 1ULL << (gbits - 1);         // nominal stretch; a stretch may grow to 2^gbits to end at a group boundary
    uint32_t *sa_lo = nullptr, *rk_lo = nullptr;
    uint8_t *sa_hi = nullptr, *rk_hi = nullptr, *sa_hi_own = nullptr;
    unsigned long long *key[2] = {nullptr, nullptr}, *val[2] = {nullptr, nullptr}, *d_hist = nullptr, *d_small = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    std::vector<unsigned long long> hist(65536);
    unsigned long long h_small[4] = {0, 0, 0, 0};
    uint64_t buf_cap = 0;
    auto ensure_bufs = [&](uint64_t m) -> hipError_t {                   // stretch buffers for m elements
        if (m <= buf_cap) return hipSuccess;
        for (int k = 0; k < 2; k++) { (void)hipFree(key[k]); (void)hipFree(val[k]); key[k] = val[k] = nullptr; }
        (void)hipFree(tmp);
        tmp = nullptr;
        buf_cap = 0;
        hipError_t e = hipSuccess;
        for (int k = 0; k < 2 && e == hipSuccess; k++) { e = hipMalloc(&key[k], m * 8); if (e == hipSuccess) e = hipMalloc(&val[k], m * 8); }
        size_t t1 = 0, t2 = 0, t3 = 0;
        if (e == hipSuccess) e = bk::prim::sort_pairs(nullptr, t1, key[0], key[1], val[0], val[1], (size_t)m, 0, 64, s);
        if (e == hipSuccess) e = bk::prim::inclusive_max(nullptr, t2, key[0], key[0], (size_t)m, s);
        if (e == hipSuccess) {
            rocprim::counting_iterator<unsigned long long> it(0ULL);
            e = bk::prim::select_if(nullptr, t3, it, val[0], d_small, (size_t)(1ULL << 30), InBuckets{d_seq, n, 0, 0}, s);
        }
        tmp_bytes = std::max(t1, std::max(t2, t3));
        if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes);
        if (e == hipSuccess) buf_cap = m;
        return e;
    };
    // sorts the m (key[0], val[0]) pairs of a stretch that starts at suffix array index j0, writes SA and ranks back
    auto sort_and_store = [&](uint64_t j0, uint64_t m, int end_bit) -> hipError_t {
        size_t tb = tmp_bytes;
        hipError_t e = bk::prim::sort_pairs(tmp, tb, key[0], key[1], val[0], val[1], (size_t)m, 0, end_bit, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_w_heads, dim3(grid_for(m)), dim3(256), 0, s, key[1], m, key[0]);
        tb = tmp_bytes;
        e = bk::prim::inclusive_max(tmp, tb, key[0], val[0], (size_t)m, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_w_store, dim3(grid_for(m)), dim3(256), 0, s, val[1], val[0], m, j0, sa_lo, sa_hi, rk_lo, rk_hi, d_small + 1);
        return hipGetLastError();
    };

    // the suffix array planes live inside the caller's output buffer while the rounds run (5-byte output: low words in its first 4n
    // bytes, high bytes in the last n); the element layout the .sfx wants is produced at the end through the rank planes
    sa_lo = (uint32_t *)d_sa_out;
    if (el_size == 5) sa_hi = (uint8_t *)d_sa_out + n * 4;
    else { SA_TRY(hipMalloc(&sa_hi_own, n)); sa_hi = sa_hi_own; }
    SA_TRY(hipMalloc(&rk_lo, n * 4));
    SA_TRY(hipMalloc(&rk_hi, n));
    SA_TRY(hipMalloc(&d_hist, 65536 * 8));
    SA_TRY(hipMalloc(&d_small, 4 * 8));
    SA_TRY(hipMemsetAsync(d_hist, 0, 65536 * 8, s));
    SA_TRY(hipMemsetAsync(d_small, 0, 4 * 8, s));
    hipLaunchKernelGGL(k_w_hist, dim3(grid_for(n)), dim3(256), 0, s, d_seq, n, d_hist);
    SA_TRY(hipMemcpyAsync(hist.data(), d_hist, 65536 * 8, hipMemcpyDeviceToHost, s));
    SA_TRY(hipStreamSynchronize(s));
    {   // ---- round 0, bucket batch by bucket batch
        uint64_t biggest = 0;
        for (unsigned long long c : hist) biggest = std::max<uint64_t>(biggest, c);
        SA_TRY(ensure_bufs(std::max(cap, biggest)));
        uint64_t j0 = 0;
        for (uint32_t b0 = 0; b0 < 65536;) {
            uint64_t m = hist[b0];
            uint32_t b1 = b0;
            while (b1 + 1 < 65536 && m + hist[b1 + 1] <= std::max(cap, (uint64_t)hist[b0])) m += hist[++b1];
            if (m) {
                // positions of the batch in text order: the selection runs over the text in pieces of 2^30 positions
                uint64_t got = 0;
                for (uint64_t at = 0; at < n; at += 1ULL << 30) {
                    const uint64_t cnt = std::min<uint64_t>(1ULL << 30, n - at);
                    rocprim::counting_iterator<unsigned long long> it((unsigned long long)at);
                    size_t tb = tmp_bytes;
                    SA_TRY(bk::prim::select_if(tmp, tb, it, val[0] + got, d_small, (size_t)cnt, InBuckets{d_seq, n, b0, b1}, s));
                    SA_TRY(hipMemcpyAsync(h_small, d_small, 8, hipMemcpyDeviceToHost, s));
                    SA_TRY(hipStreamSynchronize(s));
                    got += h_small[0];
                }
                if (got != m) { fprintf(stderr, "biokanga_amd: suffix sort: bucket batch holds %llu positions, %llu expected\n", (unsigned long long)got, (unsigned long long)m); rc = -1; goto done; }
                hipLaunchKernelGGL(k_w_keys0, dim3(grid_for(m)), dim3(256), 0, s, d_seq, n, val[0], m, key[0]);
                SA_TRY(sort_and_store(j0, m, 64));
                j0 += m;
            }
            b0 = b1 + 1;
        }
        if (j0 != n) { rc = -1; goto done; }
    }
    // ---- doubling rounds over stretches of the rank-sorted array
    for (uint64_t h = 16; h < (1ULL << 20); h <<= 1) {
        SA_TRY(hipMemcpyAsync(h_small, d_small, 16, hipMemcpyDeviceToHost, s));
        SA_TRY(hipStreamSynchronize(s));
        if (h_small[1] == 0) break;                                     // every group is a single suffix
        SA_TRY(hipMemsetAsync(d_small + 1, 0, 8, s));
        for (uint64_t j0 = 0; j0 < n;) {
            uint64_t j1 = n;
            if (n - j0 > cap) {
                hipLaunchKernelGGL(k_w_next_head, dim3(1), dim3(64), 0, s, sa_lo, sa_hi, rk_lo, rk_hi, n, j0 + cap, d_small + 2);
                SA_TRY(hipMemcpyAsync(h_small + 2, d_small + 2, 8, hipMemcpyDeviceToHost, s));
                SA_TRY(hipStreamSynchronize(s));
                j1 = h_small[2];
            }
            int single = 0;
            if (j1 - j0 > (1ULL << gbits)) {
                // no group boundary for 2^gbits indexes after j0 + cap: a giant group (a long N run, say) starts at or before j0 + cap.
                // What lies before it is one stretch; the group itself is the next one, keyed by the second part alone
                uint32_t lo32 = 0;
                uint8_t hi8 = 0;
                SA_TRY(hipMemcpyAsync(&lo32, sa_lo + j0 + cap, 4, hipMemcpyDeviceToHost, s));
                SA_TRY(hipMemcpyAsync(&hi8, sa_hi + j0 + cap, 1, hipMemcpyDeviceToHost, s));
                SA_TRY(hipStreamSynchronize(s));
                const uint64_t pos = (uint64_t)lo32 | ((uint64_t)hi8 << 32);
                SA_TRY(hipMemcpyAsync(&lo32, rk_lo + pos, 4, hipMemcpyDeviceToHost, s));
                SA_TRY(hipMemcpyAsync(&hi8, rk_hi + pos, 1, hipMemcpyDeviceToHost, s));
                SA_TRY(hipStreamSynchronize(s));
                const uint64_t jg = (uint64_t)lo32 | ((uint64_t)hi8 << 32);       // where the giant group starts
                if (jg > j0) j1 = jg;
                else single = 1;
            }
            const uint64_t m = j1 - j0;
            SA_TRY(ensure_bufs(m));
            SA_TRY(hipMemsetAsync(d_small + 3, 0, 8, s));
            hipLaunchKernelGGL(k_w_keys, dim3(grid_for(m)), dim3(256), 0, s, sa_lo, sa_hi, rk_lo, rk_hi, n, j0, m, h, rbits, single, key[0], val[0],
                               d_small + 3);
            SA_TRY(hipMemcpyAsync(h_small + 3, d_small + 3, 8, hipMemcpyDeviceToHost, s));
            SA_TRY(hipStreamSynchronize(s));
            if (h_small[3] != 0) SA_TRY(sort_and_store(j0, m, 64));     // else: all singletons, nothing to refine here
            j0 = j1;
        }
    }
    if (el_size == 5) {
        SA_TRY(hipMemcpyAsync(rk_lo, sa_lo, n * 4, hipMemcpyDeviceToDevice, s));
        SA_TRY(hipMemcpyAsync(rk_hi, sa_hi, n, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_w_write5, dim3(grid_for(n)), dim3(256), 0, s, rk_lo, rk_hi, n, (uint8_t *)d_sa_out);
    }                                                                   // 4-byte output: the low plane is the array
    SA_TRY(hipGetLastError());
    SA_TRY(hipStreamSynchronize(s));
done:
    for (int k = 0; k < 2; k++) { (void)hipFree(key[k]); (void)hipFree(val[k]); }
    (void)hipFree(tmp); (void)hipFree(sa_hi_own); (void)hipFree(rk_lo); (void)hipFree(rk_hi); (void)hipFree(d_hist); (void)hipFree(d_small);
    return rc;
}

int build_sa_device(const uint8_t *d_seq, uint64_t n, void *d_sa_out, int el_size, hipStream_t s)
{
    int rc = 0;
    {
        // BK_SA_WIDE_CHUNK=<elements>: force the chunked 40-bit path (tests run it on small inputs with small stretches)
        const unsigned long long force = bk::env::sa_wide_chunk();
        if (n >= 0xFFFFFFFFULL || force > 0) return build_sa_device_wide(d_seq, n, d_sa_out, el_size, (uint64_t)force, s);
    }
    uint64_t *key[2] = {nullptr, nullptr};
    uint32_t *val[2] = {nullptr, nullptr};
    uint32_t *rank = nullptr, *head = nullptr;
    unsigned long long *d_groups = nullptr, h_groups = 0;
    void *tmp = nullptr;
    size_t tmp_sort = 0, tmp_scan = 0, tmp_bytes = 0;
    int cur = 0;
    const unsigned g = grid_for(n);

    SA_TRY(hipMalloc(&key[0], n * 8));
    SA_TRY(hipMalloc(&key[1], n * 8));
    SA_TRY(hipMalloc(&val[0], n * 4));
    SA_TRY(hipMalloc(&val[1], n * 4));
    SA_TRY(hipMalloc(&rank, n * 4));
    SA_TRY(hipMalloc(&d_groups, 8));
    head = (uint32_t *)key[0];      // key[in] is dead once sorted: its storage is reused for heads/groups
    SA_TRY(bk::prim::sort_pairs(nullptr, tmp_sort, key[0], key[1], val[0], val[1], (size_t)n, 0, 64, s));
    SA_TRY(bk::prim::inclusive_max(nullptr, tmp_scan, head, head, (size_t)n, s));
    tmp_bytes = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
    SA_TRY(hipMalloc(&tmp, tmp_bytes));

    hipLaunchKernelGGL(k_sa_init, dim3(g), dim3(256), 0, s, d_seq, n, key[0], val[0]);
    for (uint64_t h = 16;; h <<= 1) {
        // sort (key[0], val[cur]) -> (key[1], val[cur^1])
        size_t tb = tmp_bytes;
        SA_TRY(bk::prim::sort_pairs(tmp, tb, key[0], key[1], val[cur], val[cur ^ 1], (size_t)n, 0, 64, s));
        cur ^= 1;
        SA_TRY(hipMemsetAsync(d_groups, 0, 8, s));
        head = (uint32_t *)key[0];
        uint32_t *grp = head + n;   // second half of the 8n-byte key[0] buffer
        hipLaunchKernelGGL(k_sa_heads, dim3(g), dim3(256), 0, s, key[1], n, head, d_groups);
        SA_TRY(hipMemcpyAsync(&h_groups, d_groups, 8, hipMemcpyDeviceToHost, s));
        SA_TRY(hipStreamSynchronize(s));
        if (h_groups == n || h >= (1ULL << 20)) break;
        tb = tmp_bytes;
        SA_TRY(bk::prim::inclusive_max(tmp, tb, head, grp, (size_t)n, s));
        hipLaunchKernelGGL(k_sa_scatter_rank, dim3(g), dim3(256), 0, s, grp, val[cur], n, rank);
        hipLaunchKernelGGL(k_sa_make_keys, dim3(g), dim3(256), 0, s, rank, val[cur], n, h, key[0]);
        SA_TRY(hipGetLastError());
    }
    if (el_size == 4)
        SA_TRY(hipMemcpyAsync(d_sa_out, val[cur], n * 4, hipMemcpyDeviceToDevice, s));
    else
        hipLaunchKernelGGL(k_sa_write5, dim3(g), dim3(256), 0, s, val[cur], n, (uint8_t *)d_sa_out);
    SA_TRY(hipGetLastError());
    SA_TRY(hipStreamSynchronize(s));
done:
    (void)hipFree(key[0]); (void)hipFree(key[1]); (void)hipFree(val[0]); (void)hipFree(val[1]);
    (void)hipFree(rank); (void)hipFree(d_groups); (void)hipFree(tmp);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Work-list sort used by the aligner: reorders a list of 32-bit items by 32-bit keys (rocPRIM radix sort).
// Called with tmp == nullptr it only reports the temporary storage needed.
int sort_list_by_key(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n,
                     void *tmp, size_t *tmp_bytes, hipStream_t s)
{
    size_t tb = *tmp_bytes;
    hipError_t e = bk::prim::sort_pairs(tmp, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, 32, s);
    if (tmp == nullptr) *tmp_bytes = tb;
    return e == hipSuccess ? 0 : -100;
}

// exclusive prefix sum over 64-bit counts (loci list offsets of the multi-loci modes); same tmp protocol
int scan_counts_u64(const unsigned long long *in, unsigned long long *out, uint32_t n, void *tmp, size_t *tmp_bytes, hipStream_t s)
{
    size_t tb = *tmp_bytes;
    hipError_t e = bk::prim::exclusive_sum(tmp, tb, in, out, (size_t)n, s);
    if (tmp == nullptr) *tmp_bytes = tb;
    return e == hipSuccess ? 0 : -100;
}

}  // namespace bk
