// bk_index.hip - index set-up kernels (gfx950): the .sfx image as the path's kernels want it.
//   k_pack_target / k_pack_target2   1 B/base target -> 4 bit/base words; 2 bit/base copy + N/EOS region bitmap
//   k_split_sa5                      5-byte suffix elements -> lo32 + hi8 arrays
//   k_build_ktab / k_make_ktab2      k-mer table (+ entries that carry their bucket's first second-level key)
//   k_build_k2 / k_check_k2          second-level keys
//   k_build_isa                      inverse suffix array
//   k_build_swin                     suffix-ordered window array
#include "bk_dev_k2.h"
#include "bk_dev_window.h"

namespace bk {

// ------------------------------------------------------------------------------------------------
// index upload kernels

__global__ void k_pack_target(const uint8_t *__restrict__ seq, uint64_t n, uint64_t *__restrict__ tgt4, uint64_t nwords)
{
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; w < nwords; w += stride) {
        uint64_t base = w << 4;
        uint64_t v = 0;
        if (base + 16 <= n) {
            const uint4 q = *reinterpret_cast<const uint4 *>(seq + base);
            uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = 0; k < 4; k++) v = (v << 4) | ((d[j] >> (8 * k)) & 0x0f);
        } else {
            for (int k = 0; k < 16; k++) {
                uint64_t p = base + k;
                uint64_t nb = p < n ? (uint64_t)(seq[p] & 0x0f) : 7ULL;
                v = (v << 4) | nb;
            }
        }
        tgt4[w] = v;
    }
}

// 2 bit/base copy + "block holds N/EOS" bitmap, derived from the packed 4-bit target (padding included)
__global__ void k_pack_target2(const uint64_t *__restrict__ tgt4, uint64_t nwords4, uint64_t *__restrict__ tgt2,
                               unsigned int *__restrict__ nflag32, int flag_shift)
{
    // one thread per 64-base block = 4 nibble words -> 2 words of tgt2; flags per 2^flag_shift bases
    uint64_t nblocks = nwords4 / 4;
    for (uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; blk < nblocks; blk += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t a = tgt4[4 * blk], b = tgt4[4 * blk + 1], c = tgt4[4 * blk + 2], d = tgt4[4 * blk + 3];
        tgt2[2 * blk] = ((uint64_t)squeeze2(a) << 32) | squeeze2(b);
        tgt2[2 * blk + 1] = ((uint64_t)squeeze2(c) << 32) | squeeze2(d);
        if ((a | b | c | d) & 0x4444444444444444ULL) {
            uint64_t g = blk >> (flag_shift - 6);
            atomicOr(&nflag32[g >> 5], 1u << (g & 31));
        }
    }
}

__global__ void k_split_sa5(const uint8_t *__restrict__ sa5, uint64_t n, uint32_t *__restrict__ lo, uint8_t *__restrict__ hi)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint8_t *p = sa5 + i * 5;
        lo[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        hi[i] = p[4];
    }
}

// bucket of a suffix = 2-bit code of its first k bases; a suffix that meets N/EOS after j < k bases
// sorts after every real k-mer sharing those j bases, i.e. in the bucket "prefix padded with T"
__device__ __forceinline__ uint64_t suffix_bucket(const uint64_t *__restrict__ tgt, uint64_t pos, int k)
{
    uint64_t w = nib16(tgt, pos);
    uint64_t bad = w & 0x4444444444444444ULL;           // N(4) and EOS(7) have bit 2 set
    if (bad) {
        int j = __clzll(bad) >> 2;                       // first offending nibble
        if (j < 16) w |= (~0ULL >> (4 * j)) & 0x3333333333333333ULL;   // pad with T from there on
    }
    return (uint64_t)(squeeze2(w) >> (32 - 2 * k));
}

template <bool WIDE, typename TabT>
__global__ void k_build_ktab(DevIndex ix, TabT *__restrict__ tab, int k, uint64_t i0, uint64_t i1)
{
    // (suffix array indexes [i0, i1) of 0 .. n: a table can be made range by range, as the array arrives)
    uint64_t n = ix.n;
    uint64_t ncodes = 1ULL << (2 * k);
    uint64_t i = i0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < i1; i += stride) {
        // entries (prev, cur] receive i; prev = bucket(i-1) (or -1), cur = bucket(i) (or ncodes at i == n)
        uint64_t cur = i < n ? suffix_bucket(ix.tgt4, sa_get<WIDE>(ix, i), k) : ncodes;
        uint64_t from = i > 0 ? suffix_bucket(ix.tgt4, sa_get<WIDE>(ix, i - 1), k) + 1 : 0;
        for (uint64_t c = from; c <= cur; c++) tab[c] = (TabT)i;
    }
}

template <bool WIDE>
__global__ void k_build_k2(DevIndex ix, uint32_t *__restrict__ k2, uint64_t i0, uint64_t i1)
{
    for (uint64_t i = i0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < i1; i += (uint64_t)gridDim.x * blockDim.x)
        k2[i] = k2_make(ix.tgt4, sa_get<WIDE>(ix, i), ix.k);
}

// the bisection needs k2 non-decreasing inside every k-mer bucket; count the places where it is not
template <bool WIDE>
__global__ void k_check_k2(DevIndex ix, const uint32_t *__restrict__ k2, unsigned long long *__restrict__ bad, uint64_t i0, uint64_t i1)
{
    // (the pairs (i, i + 1) with i in [i0, i1))
    for (uint64_t i = i0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < i1 && i + 1 < ix.n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (k2[i] <= k2[i + 1]) continue;
        if (suffix_bucket(ix.tgt4, sa_get<WIDE>(ix, i), ix.k) == suffix_bucket(ix.tgt4, sa_get<WIDE>(ix, i + 1), ix.k))
            atomicAdd(bad, 1ULL);
    }
}

// entry i of the suffix-ordered window array: kSwBases bases of the 2-bit target from sa[i] - kSwPre on (bases before the target's
// start read as 0: no window that uses them passes the "candidate starts before the read does" test)
__global__ void __launch_bounds__(256) k_build_swin(DevIndex ix, uint4 *__restrict__ swin)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ix.n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t base0 = (int64_t)ix.sa_lo[i] - kSwPre;
        uint64_t wd[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int64_t pos = base0 + 32 * k;
            uint64_t v;
            if (pos >= 0) {
                const uint64_t wi = (uint64_t)pos >> 5;
                const unsigned sh = (unsigned)(pos & 31) << 1;
                const uint64_t a = ix.tgt2[wi], bq = ix.tgt2[wi + 1];
                v = (a << sh) | ((bq >> 1) >> (63 - sh));
            } else if (pos > -32)
                v = ix.tgt2[0] >> (unsigned)(2 * (-pos));
            else
                v = 0;
            wd[k] = v;
        }
#pragma unroll
        for (int q = 0; q < 3; q++)
            swin[i * 3 + q] = make_uint4((uint32_t)wd[2 * q], (uint32_t)(wd[2 * q] >> 32), (uint32_t)wd[2 * q + 1], (uint32_t)(wd[2 * q + 1] >> 32));
    }
}

void launch_build_swin(const DevIndex &ix, void *swin, hipStream_t s)
{
    hipLaunchKernelGGL(k_build_swin, dim3(65536), dim3(256), 0, s, ix, reinterpret_cast<uint4 *>(swin));
}

__global__ void k_build_isa(const uint32_t *__restrict__ sa, uint64_t i0, uint64_t i1, uint32_t *__restrict__ isa)
{
    uint64_t i = i0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < i1; i += stride) isa[sa[i]] = (uint32_t)i;
}

__global__ void k_fill_u64(unsigned long long *__restrict__ p, uint64_t n, unsigned long long v)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// launchers (called from bk_engine.cpp through plain function pointers-free C++ interface)

void launch_pack_target(const uint8_t *seq, uint64_t n, uint64_t *tgt4, uint64_t nwords, hipStream_t s)
{
    uint64_t blocks = (nwords + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_pack_target, dim3((unsigned)blocks), dim3(256), 0, s, seq, n, tgt4, nwords);
}

void launch_split_sa5(const uint8_t *sa5, uint64_t n, uint32_t *lo, uint8_t *hi, hipStream_t s)
{
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_split_sa5, dim3((unsigned)blocks), dim3(256), 0, s, sa5, n, lo, hi);
}

// (every table builder takes a range [i0, i1) of suffix array indexes; i1 = 0 stands for the whole array)
void launch_build_ktab(const DevIndex &ix, void *tab, int k, bool tab64, hipStream_t s, uint64_t i0, uint64_t i1)
{
    if (i1 == 0) i1 = ix.n + 1;                      // (the entry past the last bucket comes with index n)
    if (i1 <= i0) return;
    uint64_t blocks = (i1 - i0 + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    bool wide = ix.sa_hi != nullptr;
    if (wide) {
        if (tab64) hipLaunchKernelGGL((k_build_ktab<true, uint64_t>), dim3((unsigned)blocks), dim3(256), 0, s, ix, (uint64_t *)tab, k, i0, i1);
        else hipLaunchKernelGGL((k_build_ktab<true, uint32_t>), dim3((unsigned)blocks), dim3(256), 0, s, ix, (uint32_t *)tab, k, i0, i1);
    } else {
        if (tab64) hipLaunchKernelGGL((k_build_ktab<false, uint64_t>), dim3((unsigned)blocks), dim3(256), 0, s, ix, (uint64_t *)tab, k, i0, i1);
        else hipLaunchKernelGGL((k_build_ktab<false, uint32_t>), dim3((unsigned)blocks), dim3(256), 0, s, ix, (uint32_t *)tab, k, i0, i1);
    }
}

// hipMemsetAsync is not trusted with >= 4 GiB spans: clear with our own grid-stride kernel
void launch_fill_u64(unsigned long long *p, uint64_t n, unsigned long long v, hipStream_t s)
{
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (!blocks) return;
    hipLaunchKernelGGL(k_fill_u64, dim3((unsigned)blocks), dim3(256), 0, s, p, n, v);
}

void launch_pack_target2(const uint64_t *tgt4, uint64_t nwords4, uint64_t *tgt2, unsigned int *nflag32, int flag_shift, hipStream_t s)
{
    uint64_t blocks = (nwords4 / 4 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_pack_target2, dim3((unsigned)blocks), dim3(256), 0, s, tgt4, nwords4, tgt2, nflag32, flag_shift);
}

// k-mer table entries {bucket start, second-level key of the bucket's first suffix} (DevIndex::ktab2)
__global__ void __launch_bounds__(256) k_make_ktab2(const uint32_t *__restrict__ tab, const uint32_t *__restrict__ k2, uint64_t n_entries, uint64_t n,
                                                    uint2 *__restrict__ out)
{
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_entries; c += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t lo = tab[c];
        out[c] = make_uint2(lo, lo < n ? k2[lo] : 0u);
    }
}

void launch_make_ktab2(const uint32_t *tab, const uint32_t *k2, uint64_t n_entries, uint64_t n, void *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_make_ktab2, dim3(65536), dim3(256), 0, s, tab, k2, n_entries, n, reinterpret_cast<uint2 *>(out));
}

void launch_build_k2(const DevIndex &ix, uint32_t *k2, unsigned long long *bad, hipStream_t s, uint64_t i0, uint64_t i1)
{
    if (i1 == 0) i1 = ix.n;
    if (i1 <= i0) return;
    uint64_t blocks = (i1 - i0 + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    DevIndex t = ix;
    t.k2 = k2;
    // the order check looks at pairs (i, i + 1): a range checks the pair that straddles its start, and leaves the one at its end to the next
    const uint64_t c0 = i0 ? i0 - 1 : 0, c1 = i1 == ix.n ? i1 : i1 - 1;
    if (ix.sa_hi) {
        hipLaunchKernelGGL(k_build_k2<true>, dim3((unsigned)blocks), dim3(256), 0, s, t, k2, i0, i1);
        hipLaunchKernelGGL(k_check_k2<true>, dim3((unsigned)blocks), dim3(256), 0, s, t, k2, bad, c0, c1);
    } else {
        hipLaunchKernelGGL(k_build_k2<false>, dim3((unsigned)blocks), dim3(256), 0, s, t, k2, i0, i1);
        hipLaunchKernelGGL(k_check_k2<false>, dim3((unsigned)blocks), dim3(256), 0, s, t, k2, bad, c0, c1);
    }
}

void launch_build_isa(const uint32_t *sa, uint64_t n, uint32_t *isa, hipStream_t s, uint64_t i0, uint64_t i1)
{
    if (i1 == 0) i1 = n;
    if (i1 <= i0) return;
    uint64_t blocks = (i1 - i0 + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    hipLaunchKernelGGL(k_build_isa, dim3((unsigned)blocks), dim3(256), 0, s, sa, i0, i1, isa);
}

}  // namespace bk
