// bk_snp.hip - per-sequence accepted-read counts and the SNP pile-up / screening kernels (gfx950; Aligner.cpp:5475-5537, 6803-8071).
#include "bk_dev_util.h"

namespace bk {

// per-sequence counts of accepted reads (feeds the -O CSV and the cross-rank reduction): one pass over
// a finished chunk's hit records with a block-private LDS histogram, so that the alignment kernels
// carry no same-address global atomics (47 M of them per 50 M reads cost ~30 ms inside k_light)
constexpr uint32_t kHistLds = 4096;

__global__ void __launch_bounds__(256) k_count_seqs(const bk_hit *__restrict__ out, uint32_t n, const uint32_t *__restrict__ id2idx,
                                                    uint32_t n_ent, unsigned long long *__restrict__ counts)
{
    __shared__ uint32_t s_hist[kHistLds];
    const uint32_t nl = n_ent < kHistLds ? n_ent : kHistLds;
    for (uint32_t i = threadIdx.x; i < nl; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (out[i].nar != BK_NAR_ACCEPTED) continue;
        uint32_t e = id2idx[out[i].chrom_id];
        if (e < nl) atomicAdd(&s_hist[e], 1u);
        else atomicAdd(&counts[e], 1ULL);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nl; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&counts[i], (unsigned long long)s_hist[i]);
}

void launch_count_seqs(const bk_hit *out, uint32_t n, const uint32_t *id2idx, uint32_t n_ent, unsigned long long *counts, hipStream_t s)
{
    if (!n) return;
    uint32_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_count_seqs, dim3(blocks), dim3(256), 0, s, out, n, id2idx, n_ent, counts);
}

// ------------------------------------------------------------------------------------------------
// SNP pile-up and screening (CAligner::ProcessSNPs :7737-7960, OutputSNPs :6880-7110).
// Counts are six planes of uint32 over the concatenated target: plane 0 = NumRefBases, planes 1..5 = NonRefBaseCnts
// a,c,g,t,n; a wave walks one alignment, its lanes consecutive loci, so the adds of a plane coalesce.

__global__ void __launch_bounds__(256) k_snp_pileup(DevIndex ix, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offs,
                                                    const uint32_t *__restrict__ id2idx, const bk_snp_aln *__restrict__ alns, uint64_t n_alns,
                                                    uint32_t *__restrict__ planes)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t a = wave; a < n_alns; a += n_waves) {
        const bk_snp_aln al = alns[a];
        const uint32_t e = id2idx[al.chrom_id];
        const uint64_t g0 = ix.ent_start[e] + al.loci;
        uint32_t len = al.len;
        const uint64_t chrom_len = ix.ent_end[e] - ix.ent_start[e] + 1;
        if ((uint64_t)al.loci + len > chrom_len) {                       // :7826-7830
            if (al.loci + 10 > chrom_len) continue;
            len = (uint32_t)(chrom_len - al.loci);
        }
        const uint8_t *rd = bases + offs[al.read_idx] + al.read_ofs;
        const bool rev = al.strand == '-';
        for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
            uint32_t rb = rev ? rd[al.len - 1 - i] & 7u : rd[i] & 7u;
            if (rev && rb < 4) rb = 3 - rb;
            const uint64_t g = g0 + i;
            const uint32_t tb = (uint32_t)(ix.tgt4[g >> 4] >> (60 - 4 * (unsigned)(g & 15))) & 7u;
            if (tb >= 4 || rb > 4) continue;
            atomicAdd(&planes[(tb == rb ? 0 : (uint64_t)(1 + rb) * ix.n) + g], 1u);
        }
    }
}

__global__ void __launch_bounds__(256) k_snp_sites(DevIndex ix, const uint32_t *__restrict__ planes, uint64_t g0, uint32_t chrom_len, uint32_t min_reads,
                                                   double min_prop, bk_snp_site *__restrict__ sites, uint32_t cap, uint32_t *__restrict__ n_sites,
                                                   unsigned long long *__restrict__ totals /*[4]*/)
{
    constexpr uint32_t kFlank = 25, kWin = 2 * kFlank + 1;                 // cSNPBkgndRateWindow = 51
    __shared__ unsigned long long s_tot[4];
    if (threadIdx.x < 4) s_tot[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long t_m = 0, t_mm = 0, t_cov = 0, t_bases = 0;
    for (uint64_t l64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; l64 < chrom_len; l64 += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = (uint32_t)l64;
        const uint64_t g = g0 + l;
        const uint32_t ref = planes[g];
        uint32_t nr[5], nonref = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) { nr[k] = planes[(uint64_t)(1 + k) * ix.n + g]; nonref += nr[k]; }
        const uint32_t tot = ref + nonref;
        t_m += ref; t_mm += nonref;
        if (tot) { t_cov++; t_bases += tot; }
        if (tot < min_reads || nonref < 1u) continue;
        if ((double)nonref / (double)(int)tot < min_prop) continue;
        // the window OutputSNPs has slid to by the time it looks at this locus (:6885-6925)
        uint32_t w_lo = 0, w_n = chrom_len < kWin ? chrom_len : kWin;
        if (l > kFlank && chrom_len > kWin) {
            const uint32_t last = chrom_len - 1 - kFlank;
            w_lo = (l < last ? l : last) - kFlank;
        }
        uint32_t wm = 0, wmm = 0;
        for (uint32_t k = 0; k < w_n; k++) {
            const uint64_t q = g0 + w_lo + k;
            wm += planes[q];
#pragma unroll
            for (int b = 0; b < 5; b++) wmm += planes[(uint64_t)(1 + b) * ix.n + q];
        }
        const uint32_t slot = atomicAdd(n_sites, 1u);
        if (slot < cap) {
            bk_snp_site st;
            st.loci = l; st.num_ref = ref;
#pragma unroll
            for (int k = 0; k < 5; k++) st.non_ref[k] = nr[k];
            st.win_mismatches = wmm; st.win_matches = wm;
            st.ref_base = (uint32_t)(ix.tgt4[g >> 4] >> (60 - 4 * (unsigned)(g & 15))) & 7u;
            sites[slot] = st;
        }
    }
    atomicAdd(&s_tot[0], t_m); atomicAdd(&s_tot[1], t_mm); atomicAdd(&s_tot[2], t_cov); atomicAdd(&s_tot[3], t_bases);
    __syncthreads();
    if (threadIdx.x < 4 && s_tot[threadIdx.x]) atomicAdd(&totals[threadIdx.x], s_tot[threadIdx.x]);
}

__global__ void __launch_bounds__(256) k_snp_gather(DevIndex ix, const uint32_t *__restrict__ planes, uint64_t g0, uint32_t n, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t g = g0 + i;
#pragma unroll
    for (int k = 0; k < 6; k++) out[(uint64_t)i * 7 + k] = planes[(uint64_t)k * ix.n + g];
    out[(uint64_t)i * 7 + 6] = (uint32_t)(ix.tgt4[g >> 4] >> (60 - 4 * (unsigned)(g & 15))) & 7u;
}

void launch_snp_gather(const DevIndex &ix, const uint32_t *planes, uint64_t g0, uint32_t n, uint32_t *out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_snp_gather, dim3((n + 255) / 256), dim3(256), 0, s, ix, planes, g0, n, out);
}

__global__ void __launch_bounds__(256) k_snp_centroids(DevIndex ix, const uint32_t *__restrict__ planes, uint64_t g0, uint32_t chrom_len, uint32_t min_reads,
                                                       uint32_t *__restrict__ hist)
{
    __shared__ uint32_t s_hist[BK_SNP_CENTROIDS];                          // 64 KB of the CU's 160 KB LDS
    for (uint32_t i = threadIdx.x; i < BK_SNP_CENTROIDS; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    for (uint64_t l64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 3; l64 + 3 < chrom_len; l64 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t g = g0 + l64;
        uint32_t tot = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) tot += planes[(uint64_t)k * ix.n + g];
        if (tot < min_reads) continue;
        const uint64_t w = nib16(ix.tgt4, g - 3);                         // 7 target bases from the top nibble down
        uint32_t idx = 0;
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const uint32_t b = (uint32_t)(w >> (60 - 4 * j)) & 7u;
            ok = ok && b < 4;
            idx = (idx << 2) | (b & 3u);
        }
        if (ok) atomicAdd(&s_hist[idx], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < BK_SNP_CENTROIDS; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}

void launch_snp_centroids(const DevIndex &ix, const uint32_t *planes, uint64_t g0, uint32_t chrom_len, uint32_t min_reads, uint32_t *hist, hipStream_t s)
{
    if (chrom_len < 7) return;
    uint64_t blocks = ((uint64_t)chrom_len + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_snp_centroids, dim3((unsigned)blocks), dim3(256), 0, s, ix, planes, g0, chrom_len, min_reads, hist);
}

void launch_snp_pileup(const DevIndex &ix, const uint8_t *bases, const uint64_t *offs, const uint32_t *id2idx, const bk_snp_aln *alns, uint64_t n_alns,
                       uint32_t *planes, hipStream_t s)
{
    if (!n_alns) return;
    uint64_t blocks = (n_alns + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_snp_pileup, dim3((unsigned)blocks), dim3(256), 0, s, ix, bases, offs, id2idx, alns, n_alns, planes);
}

void launch_snp_sites(const DevIndex &ix, const uint32_t *planes, uint64_t g0, uint32_t chrom_len, uint32_t min_reads, double min_prop,
                      bk_snp_site *sites, uint32_t cap, uint32_t *n_sites, unsigned long long *totals, hipStream_t s)
{
    if (!chrom_len) return;
    uint64_t blocks = ((uint64_t)chrom_len + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_snp_sites, dim3((unsigned)blocks), dim3(256), 0, s, ix, planes, g0, chrom_len, min_reads, min_prop, sites, cap, n_sites, totals);
}

}  // namespace bk
