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
// (hipcub::DeviceRadixSort) - bandwidth-bound streaming passes over HBM.  Stops when all ranks are
// distinct or h >= 2^20.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>

namespace bk {

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

static inline unsigned grid_for(uint64_t n)
{
    uint64_t b = (n + 255) / 256;
    return (unsigned)(b > 262144 ? 262144 : (b ? b : 1));
}

int build_sa_device(const uint8_t *d_seq, uint64_t n, void *d_sa_out, int el_size, hipStream_t s)
{
    int rc = 0;
    if (n >= 0xFFFFFFFFULL) {
        fprintf(stderr, "biokanga_amd: device suffix sort of >= 2^32 bases is not built yet\n");
        return -100;
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
    SA_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, key[0], key[1], val[0], val[1], (size_t)n, 0, 64, s));
    SA_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, tmp_scan, head, head, hipcub::Max(), (size_t)n, s));
    tmp_bytes = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
    SA_TRY(hipMalloc(&tmp, tmp_bytes));

    hipLaunchKernelGGL(k_sa_init, dim3(g), dim3(256), 0, s, d_seq, n, key[0], val[0]);
    for (uint64_t h = 16;; h <<= 1) {
        // sort (key[0], val[cur]) -> (key[1], val[cur^1])
        size_t tb = tmp_bytes;
        SA_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, key[0], key[1], val[cur], val[cur ^ 1], (size_t)n, 0, 64, s));
        cur ^= 1;
        SA_TRY(hipMemsetAsync(d_groups, 0, 8, s));
        head = (uint32_t *)key[0];
        uint32_t *grp = head + n;   // second half of the 8n-byte key[0] buffer
        hipLaunchKernelGGL(k_sa_heads, dim3(g), dim3(256), 0, s, key[1], n, head, d_groups);
        SA_TRY(hipMemcpyAsync(&h_groups, d_groups, 8, hipMemcpyDeviceToHost, s));
        SA_TRY(hipStreamSynchronize(s));
        if (h_groups == n || h >= (1ULL << 20)) break;
        tb = tmp_bytes;
        SA_TRY(hipcub::DeviceScan::InclusiveScan(tmp, tb, head, grp, hipcub::Max(), (size_t)n, s));
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
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, 32, s);
    if (tmp == nullptr) *tmp_bytes = tb;
    return e == hipSuccess ? 0 : -100;
}

// exclusive prefix sum over 64-bit counts (loci list offsets of the multi-loci modes); same tmp protocol
int scan_counts_u64(const unsigned long long *in, unsigned long long *out, uint32_t n, void *tmp, size_t *tmp_bytes, hipStream_t s)
{
    size_t tb = *tmp_bytes;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, in, out, (size_t)n, s);
    if (tmp == nullptr) *tmp_bytes = tb;
    return e == hipSuccess ? 0 : -100;
}

}  // namespace bk
