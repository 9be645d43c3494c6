// bk_prep.hip - read preparation (gfx950): CSeqTrans::ReverseComplement (SeqTrans.cpp:458-512) + the N policy of ProcCoredApprox
// (Aligner.cpp:9041-9063) + the first active list, from 1 byte/base reads or from packed 2 bit/base words; the checks of a packed
// batch; the compaction of striped work lists; 4-bit rows for the reads a general-family kernel is handed.
#include "bk_dev_util.h"

namespace bk {

// stripes -> dense lists, appended behind what each list already holds; k_finish_lists then adds the sizes to the lists' counts,
// folds the maximum into *max_out and clears the stripes' lines for the next launch

__global__ void __launch_bounds__(256) k_compact_lists(CompactJobs J)
{
    const int li = blockIdx.y;
    __shared__ uint32_t s_pre[kListStripes + 1];
    if (threadIdx.x < 64) {
        uint32_t run = 0;
        for (int s0 = 0; s0 < kListStripes; s0 += 64) {
            uint32_t v = J.set.cnt[(s0 + threadIdx.x) * 16 + li];
            for (int off = 1; off < 64; off <<= 1) { uint32_t u = __shfl_up(v, off); if ((int)threadIdx.x >= off) v += u; }
            s_pre[s0 + threadIdx.x + 1] = run + v;
            run += __shfl(v, 63);
        }
        if (threadIdx.x == 0) s_pre[0] = 0;
    }
    __syncthreads();
    const uint32_t tot = s_pre[kListStripes], old = *J.total[li];
    const uint32_t *__restrict__ stage = J.set.stage[li];
    uint32_t *__restrict__ dense = J.dense[li];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += gridDim.x * 256) {
        int lo = 0, hi = kListStripes - 1;                  // stripe s: s_pre[s] <= i < s_pre[s + 1]
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_pre[mid] <= i) lo = mid; else hi = mid - 1;
        }
        dense[old + i] = stage[(uint64_t)lo * J.set.cap + (i - s_pre[lo])];
    }
}

__global__ void k_finish_lists(CompactJobs J)
{
    uint32_t sum[3] = {0, 0, 0}, mx = 0;
    for (int s0 = threadIdx.x; s0 < kListStripes; s0 += 64) {
        uint32_t *line = J.set.cnt + s0 * 16;
#pragma unroll
        for (int li = 0; li < 3; li++) { sum[li] += line[li]; line[li] = 0; }
        uint32_t *mline = J.set.cnt + (kListStripes + s0) * 16;
        mx = mline[0] > mx ? mline[0] : mx;
        mline[0] = 0;
    }
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int li = 0; li < 3; li++) sum[li] += __shfl_down(sum[li], off);
        const uint32_t q = __shfl_down(mx, off);
        mx = q > mx ? q : mx;
    }
    if (threadIdx.x == 0) {
        for (int li = 0; li < J.n; li++) if (sum[li]) *J.total[li] += sum[li];
        if (J.max_out && mx > *J.max_out) *J.max_out = mx;
    }
}

// ------------------------------------------------------------------------------------------------
// read preparation: k_pack_reads (one lane per packed 16-base word of the read or of its reverse
// complement: 16 byte loads, one 8-byte store) then k_init_reads (one lane per read: N policy of
// Aligner.cpp:9041-9063 from the packed words, default result record, first active list).
// Nibble written for a base byte v: v & 7 (quality / mask bits dropped); values 5..7 mark a byte
// the reference would refuse ((*pSeq = (*pSeqVal & 0x07)) > eBaseN).

struct __attribute__((packed)) Bytes16 { uint64_t lo, hi; };      // unaligned 16-byte load (one global_load_dwordx4)

// 8 base bytes, first base in the MOST significant byte -> 8 nibbles (bits 0..2 of each byte kept)
__device__ __forceinline__ uint64_t pack8_msb(uint64_t y)
{
    y &= 0x0707070707070707ULL;
    y = (y | (y >> 4)) & 0x00FF00FF00FF00FFULL;
    y = (y | (y >> 8)) & 0x0000FFFF0000FFFFULL;
    y = (y | (y >> 16)) & 0x00000000FFFFFFFFULL;
    return y;
}

__device__ __forceinline__ uint64_t complement8(uint64_t x)       // A<->T, C<->G on 3-bit codes, others unchanged
{
    x &= 0x0707070707070707ULL;
    return x ^ (((~x >> 2) & 0x0101010101010101ULL) * 3);
}

// ---- packed batches (bk_align_batch_packed): 16 bases per 32-bit word at 2 bit/base, first base in the top bits ----------------

__device__ __forceinline__ uint32_t rev2_32(uint32_t x)            // the 16 2-bit fields in reverse order
{
    x = __brev(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
__device__ __forceinline__ uint64_t rev2_64(uint64_t x)            // the 32 2-bit fields in reverse order
{
    x = __brevll(x);
    return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
}

// bases 16w .. 16w + 15 of a packed read (rc: of its reverse complement) as one 2-bit word, zero beyond the read's end.  W = the
// read's words; the word behind its last one may be loaded (the buffers are followed by one more word), its bits are never used.
__device__ __forceinline__ uint32_t packed_word16(const uint32_t *__restrict__ W, int len, int w, bool rc)
{
    const int rem = len - 16 * w;                     // bases of the read in this word
    if (rem <= 0) return 0;
    const uint32_t keep = rem >= 16 ? 0xFFFFFFFFu : ~0u << (32 - 2 * rem);
    if (!rc) return W[w] & keep;
    // reverse complement: its bases 16w .. are the complement of the forward bases p + 15 .. p, p = len - 16w - 16
    const int p = len - 16 * w - 16;
    uint32_t x;
    if (p >= 0) {
        const int i = p >> 4;
        const unsigned s = (unsigned)(p & 15) << 1;
        const uint64_t c = ((uint64_t)W[i] << 32) | W[i + 1];
        x = (uint32_t)(c >> (32 - s));
    } else
        x = W[0] >> (unsigned)(2 * (-p));             // the read's first 16 + p bases, at the low end
    return ~rev2_32(x) & keep;
}

// reverse complement of a read held as W words of 32 bases (first base in the top bits, zero beyond its end).  Straight-line code:
// the word shift is a cascade of selects, one per bit of the shift count (a version that copied t[] into place under
// `if (shift == k)` inside an unrolled loop came out of hipcc 7.2 reading registers it had never written - rows wrong only in
// the blocks that did not start on a freshly zeroed register file; profiles/NOTES.md)
template <int W>
__device__ __forceinline__ void revcomp2(const uint64_t (&f)[W], int len, uint64_t (&r)[W])
{
    uint64_t v[W + 1];
#pragma unroll
    for (int i = 0; i < W; i++) v[i] = ~rev2_64(f[W - 1 - i]);      // the whole row reversed: the read now ends flush with the row's end
    v[W] = 0;
    const int sh = 2 * (32 * W - len), q = sh >> 6;                 // shift left by q words and bsh bits: 0 <= q <= W
    const unsigned bsh = (unsigned)(sh & 63);
#pragma unroll
    for (int step = 1; step <= W; step <<= 1) {
        const bool on = (q & step) != 0;
        uint64_t nv[W + 1];
#pragma unroll
        for (int i = 0; i <= W; i++) nv[i] = on ? (i + step <= W ? v[i + step <= W ? i + step : W] : 0ULL) : v[i];
#pragma unroll
        for (int i = 0; i <= W; i++) v[i] = nv[i];
    }
#pragma unroll
    for (int i = 0; i < W; i++) r[i] = bsh ? ((v[i] << bsh) | (v[i + 1] >> (64 - bsh))) : v[i];
}

// exceptions of a packed batch, one lane each.  k_mark_exc counts them into the reads' meta words BEFORE the read preparation runs
// (bits 16..30: N bases, bit 31: a code the reference refuses - what the N policy needs to know); k_exc_rows gives the reads that
// have one their 4-bit rows (lean batches; widened from the 2-bit rows), k_apply_exc then writes the codes into the rows of both strands
__global__ void __launch_bounds__(256) k_mark_exc(DevBatch b)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.pk_nexc) return;
    const bk_nbase e = b.pk_exc[i];
    const uint32_t rr = e.read - b.pk_read0;
    if (rr >= b.n_reads) return;
    if (e.code == 4) atomicAdd(&b.rmeta[rr], ((uint32_t)e.run + 1u) << 16);
    else atomicOr(&b.rmeta[rr], 1u << 31);
}

__global__ void __launch_bounds__(256) k_exc_rows(DevBatch b)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.pk_nexc) return;
    const uint32_t rd = b.pk_exc[i].read;
    const uint32_t rr = rd - b.pk_read0;
    if (rr >= b.n_reads || (i > 0 && b.pk_exc[i - 1].read == rd)) return;       // the first exception of a read does the read
    for (uint32_t st = 0; st < 2; st++)
        for (uint32_t w = 0; w < b.wpr; w++) {
            uint64_t v = 0;
            if (w < b.nw) {
                const uint64_t x = b.rd2[((uint64_t)rr * 2 + st) * (b.nw / 2) + (w >> 1)];
                v = spread2to4((w & 1) ? (uint32_t)x : (uint32_t)(x >> 32));
            }
            b.rd4[((uint64_t)rr * 2 + st) * b.wpr + w] = v;
        }
}

__global__ void __launch_bounds__(256) k_apply_exc(DevBatch b)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.pk_nexc) return;
    const bk_nbase e = b.pk_exc[i];
    const uint32_t rr = e.read - b.pk_read0;
    if (rr >= b.n_reads) return;
    const int len = (int)b.lens[rr];
    // the run's bases [lo, hi] of each strand (codes 4..7 are their own complement, SeqTrans.cpp:458-512), one 16-base word at a time
    const int lo2[2] = {(int)e.pos, len - 1 - (int)e.pos - (int)e.run}, hi2[2] = {(int)e.pos + (int)e.run, len - 1 - (int)e.pos};
    const unsigned long long code16 = 0x1111111111111111ULL * (unsigned long long)(e.code & 7);
    for (int st = 0; st < 2; st++) {
        for (int w = lo2[st] >> 4; w <= (hi2[st] >> 4); w++) {
            const int a = lo2[st] > 16 * w ? lo2[st] - 16 * w : 0, z = hi2[st] < 16 * w + 15 ? hi2[st] - 16 * w : 15;      // nibbles a..z of the word
            const unsigned long long m = (~0ULL >> (4 * a)) & (~0ULL << (60 - 4 * z));
            unsigned long long *p = reinterpret_cast<unsigned long long *>(b.rd4 + ((uint64_t)rr * 2 + st) * b.wpr + w);
            atomicAnd(p, ~m);
            atomicOr(p, code16 & m);
        }
    }
}

// one-time check of a packed batch's exception list (whole batch): codes 4..7, reads and positions in range, strictly ascending
__global__ void __launch_bounds__(256) k_check_exc(const bk_nbase *__restrict__ exc, uint64_t n_exc, const uint32_t *__restrict__ lens, uint32_t n_reads,
                                                   uint32_t *__restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_exc) return;
    const bk_nbase e = exc[i];
    bool ok = e.read < n_reads && e.code >= 4 && e.code <= 7;
    if (ok) ok = (uint32_t)e.pos + e.run < lens[e.read];
    if (ok && i > 0) {
        const bk_nbase q = exc[i - 1];
        ok = q.read < e.read || (q.read == e.read && (uint32_t)q.pos + q.run < e.pos);
    }
    if (!ok) atomicAdd(bad, 1u);
}

// lens16 -> lens32 and the words each read takes (the input of the offset scan)
__global__ void __launch_bounds__(256) k_widen_lens(const uint16_t *__restrict__ lens16, uint32_t n, uint32_t *__restrict__ lens32,
                                                    unsigned long long *__restrict__ nwords)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = lens16[i];
    lens32[i] = l;
    nwords[i] = (l + 15) >> 4;
}

// packed batch: [0] max over reads of (first word + words taken), [1] longest read
__global__ void __launch_bounds__(256) k_packed_extent(const uint64_t *__restrict__ offs, const uint32_t *__restrict__ lens, uint32_t n,
                                                       unsigned long long *__restrict__ out)
{
    unsigned long long e = 0, l = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long len = lens[i], end = offs[i] + ((len + 15) >> 4);
        e = end > e ? end : e;
        l = len > l ? len : l;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long e2 = __shfl_down(e, off), l2 = __shfl_down(l, off);
        e = e2 > e ? e2 : e;
        l = l2 > l ? l2 : l;
    }
    if ((threadIdx.x & 63) == 0) {
        if (e) atomicMax(out + 0, e);
        if (l) atomicMax(out + 1, l);
    }
}

void launch_packed_extent(const uint64_t *offs, const uint32_t *lens, uint32_t n, unsigned long long *out, hipStream_t s)
{
    unsigned blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (n) hipLaunchKernelGGL(k_packed_extent, dim3(blocks), dim3(256), 0, s, offs, lens, n, out);
}

void launch_widen_lens(const uint16_t *lens16, uint32_t n, uint32_t *lens32, unsigned long long *nwords, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_widen_lens, dim3((n + 255) / 256), dim3(256), 0, s, lens16, n, lens32, nwords);
}

void launch_check_exc(const bk_nbase *exc, uint64_t n_exc, const uint32_t *lens, uint32_t n_reads, uint32_t *bad, hipStream_t s)
{
    if (n_exc) hipLaunchKernelGGL(k_check_exc, dim3((unsigned)((n_exc + 255) / 256)), dim3(256), 0, s, exc, n_exc, lens, n_reads, bad);
}

// pairs != nullptr: only the two reads of each listed pair (the paired-end kernels look at the rows of the pairs they were handed, a
// few per cent of a batch)
__global__ void __launch_bounds__(256) k_pack_reads(DevBatch b, const uint32_t *__restrict__ pairs, uint32_t n_pairs)
{
    // a block packs 256 / (2 * wpr) whole reads: 32-bit index arithmetic only
    const uint32_t wpr = b.wpr;
    const uint32_t per_read = 2 * wpr;                      // <= 256 (kMaxReadLenAbs)
    const uint32_t rpb = 256 / per_read;
    const uint32_t lr = threadIdx.x / per_read;
    if (lr >= rpb) return;
    uint64_t r = (uint64_t)blockIdx.x * rpb + lr;
    if (pairs != nullptr) {
        if (r >= 2ULL * n_pairs) return;
        r = 2ULL * pairs[r >> 1] + (r & 1);
    }
    if (r >= b.n_reads) return;
    const uint32_t rem = threadIdx.x - lr * per_read;
    const uint32_t st = rem >= wpr ? 1 : 0, w = rem - st * wpr;
    int len = (int)b.lens[r];
    if (b.pk_words != nullptr) {
        // packed batch: the word, or the 16 bases of the forward read whose reverse complement it is, straight from the 2-bit words
        // (bases that are not a,c,g,t read as whatever their field holds; k_apply_exc writes their codes afterwards)
        b.rd4[r * per_read + rem] = spread2to4(packed_word16(b.pk_words + b.offs[r], len, (int)w, st != 0));
        return;
    }
    const uint8_t *s = b.bases + b.offs[r];
    uint64_t v = 0;
    int base0 = 16 * (int)w;
    if (base0 + 16 <= len) {
        // a full word: the 16 source bytes lie inside the read, fetch them with one load
        if (st == 0) {
            Bytes16 q = *reinterpret_cast<const Bytes16 *>(s + base0);
            v = (pack8_msb(__builtin_bswap64(q.lo)) << 32) | pack8_msb(__builtin_bswap64(q.hi));
        } else {
            // reverse complement (SeqTrans.cpp:458-512): output base k = complement of s[len-1-base0-k]
            Bytes16 q = *reinterpret_cast<const Bytes16 *>(s + (len - 16 - base0));
            v = (pack8_msb(complement8(q.hi)) << 32) | pack8_msb(complement8(q.lo));
        }
    } else if (base0 < len) {
        int cnt = len - base0;
        if (st == 0) {
            for (int k = 0; k < cnt; k++) v |= (uint64_t)(s[base0 + k] & 7) << (60 - 4 * k);
        } else {
            for (int k = 0; k < cnt; k++) {
                uint8_t x = s[len - 1 - base0 - k] & 7;
                x = x < 4 ? (uint8_t)(3 - x) : x;
                v |= (uint64_t)x << (60 - 4 * k);
            }
        }
    }
    b.rd4[r * per_read + rem] = v;
}

// k_pack_reads + k_init_reads in one pass for reads of <= 16*NW bases (the register-kernel path): one lane per read builds the
// rows of both strands in registers - from 16-byte loads of the read's bytes, or (PACKED) from its 2-bit words, of which the forward
// row is a copy - applies the N policy, initialises the result record and appends the read to the first active list.  Lean batches
// (b.rd2 set) get 2 bit/base rows, and 4 bit/base rows only for the reads that hold an N: 60 bytes written per 100-base read
// instead of the 248 of full rows in both forms.
#ifndef BK_PREP_TRANSPOSE
#define BK_PREP_TRANSPOSE 1
#endif
template <int NW, bool PACKED>
__global__ void __launch_bounds__(256) k_prep_fused(DevAlignCfg cfg, DevBatch b, StripeSet out)
{
    __shared__ uint32_t s_cnt, s_base, s_cmax;
    // TR: rows and result records leave through the wave's words of LDS (see the kernel's end): reads of up to 128 bases
    constexpr bool TR = NW == 8 && BK_PREP_TRANSPOSE;
    __shared__ uint4 s_tr[TR ? 4 : 1][TR ? 64 * (NW / 2) : 1];
    if (threadIdx.x == 0) { s_cnt = 0; s_cmax = 0; }
    __syncthreads();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool lean = b.rd2 != nullptr;
    const int wv = threadIdx.x >> 6;
    const bool wave_full = (r | 63u) < b.n_reads;                    // every lane of this wave has a read
    bool go = false, has_n = false;
    uint32_t my_cmax = 0;
    int len = 0;
    uint64_t fw[PACKED ? 1 : NW], rv[PACKED ? 1 : NW];              // 4 bit/base rows (1 byte/base input)
    uint64_t f2[NW / 2], r2[NW / 2];                                // 2 bit/base rows
    bk_hit h;
    if (r < b.n_reads) {
        len = (int)b.lens[r];
        int num_ns = 0;
        bool bad = false;
        if (PACKED) {
            // (all NW words are loaded whatever the read's length - straight-line loads, no exec-masked ones in an unrolled loop: see
            // DESIGN.md on hipcc 7.2; the words buffer is followed by NW more words, and what lies behind the read is masked off)
            const uint32_t *__restrict__ W = b.pk_words + b.offs[r];
            uint32_t wv[NW];
#pragma unroll
            for (int k = 0; k < NW; k++) wv[k] = W[k];
#pragma unroll
            for (int k = 0; k < NW / 2; k++) {
                uint64_t v = ((uint64_t)wv[2 * k] << 32) | wv[2 * k + 1];
                const int rem = len - 32 * k;                       // bases of the read in this word
                if (rem < 32) v = rem <= 0 ? 0ULL : (v & (~0ULL << (64 - 2 * rem)));
                f2[k] = v;
            }
            revcomp2<NW / 2>(f2, len, r2);
            const uint32_t pre = b.rmeta[r];                        // k_mark_exc: what the exception list holds for this read
            num_ns = (int)((pre >> 16) & 0x7FFFu);
            bad = (pre >> 31) != 0;
        } else {
            // forward rows from the bytes; the reverse complement rows come from them by bit work (revcomp2) unless the read has an N
            // or the batch keeps 4-bit rows for every read - then the bytes are walked a second time from the other end
            const uint8_t *s = b.bases + b.offs[r];
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const int base0 = 16 * w;
                uint64_t f = 0;
                if (base0 + 16 <= len) {
                    Bytes16 q = *reinterpret_cast<const Bytes16 *>(s + base0);
                    f = (pack8_msb(__builtin_bswap64(q.lo)) << 32) | pack8_msb(__builtin_bswap64(q.hi));
                } else if (base0 < len) {
                    const int cnt = len - base0;
                    for (int k = 0; k < cnt; k++) f |= (uint64_t)(s[base0 + k] & 7) << (60 - 4 * k);
                }
                fw[w] = f;
                rv[w] = 0;
            }
#pragma unroll
            for (int w = 0; w < NW; w++) {
                if (16 * w < len) {
                    uint64_t x = fw[w] & top_mask(len - 16 * w);
                    uint64_t hi = x & 0x4444444444444444ULL;
                    uint64_t lo = (x | (x >> 1)) & 0x1111111111111111ULL;
                    bad |= ((hi >> 2) & lo) != 0;
                    num_ns += __popcll(hi);
                }
            }
#pragma unroll
            for (int k = 0; k < NW / 2; k++) f2[k] = ((uint64_t)squeeze2(fw[2 * k]) << 32) | squeeze2(fw[2 * k + 1]);
            if (lean && !bad && num_ns == 0) revcomp2<NW / 2>(f2, len, r2);
            else {
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    const int base0 = 16 * w;
                    uint64_t v = 0;
                    if (base0 + 16 <= len) {
                        Bytes16 p = *reinterpret_cast<const Bytes16 *>(s + (len - 16 - base0));
                        v = (pack8_msb(complement8(p.hi)) << 32) | pack8_msb(complement8(p.lo));
                    } else if (base0 < len) {
                        const int cnt = len - base0;
                        for (int k = 0; k < cnt; k++) {
                            uint8_t x = s[len - 1 - base0 - k] & 7;
                            x = x < 4 ? (uint8_t)(3 - x) : x;
                            v |= (uint64_t)x << (60 - 4 * k);
                        }
                    }
                    rv[w] = v;
                }
#pragma unroll
                for (int k = 0; k < NW / 2; k++) r2[k] = ((uint64_t)squeeze2(rv[2 * k]) << 32) | squeeze2(rv[2 * k + 1]);
            }
        }
        // N policy and result record, as k_init_reads
        h.chrom_id = 0; h.match_loci = 0; h.match_len = 0; h.low_hit_instances = 0; h.rslt = 0;
        h.nar = BK_NAR_NOHIT; h.strand = '?'; h.low_mm = 0; h.nxt_low_mm = 0; h.num_hits = 0; h.mismatches = 0; h.flags = 0;
        int max_ns_seq = 0;
        if (cfg.max_ns) {
            max_ns_seq = (len * cfg.max_ns) / 100;
            if (max_ns_seq < cfg.max_ns) max_ns_seq = cfg.max_ns;
        }
        if (bad || num_ns > max_ns_seq) h.nar = BK_NAR_NS;
        has_n = bad || num_ns > 0;
        if (h.nar != BK_NAR_NS) {
            ReadPlan p = make_plan(len, cfg);
            if (p.n_phases > 0) {
                int mm, cl, cd, ofs[1];
                phase_params(p, cfg, 0, mm, cl, cd);
                int nc = core_offsets(len, cl, cd, p.max_slides, ofs, 0);
                if (nc <= kMaxCoresFast) my_cmax = (uint32_t)nc;
                go = true;
            }
        }
    }
    // the list append comes before the rows are written: its barriers wait for every store the wave has issued
    const int lane = threadIdx.x & 63;
    uint64_t m = __ballot(go);
    uint32_t my_off = 0;
    for (int off = 32; off > 0; off >>= 1) { uint32_t q = __shfl_down(my_cmax, off); my_cmax = q > my_cmax ? q : my_cmax; }
    if (m) {
        uint32_t w = 0;
        if (lane == 0) { w = atomicAdd(&s_cnt, (uint32_t)__popcll(m)); if (my_cmax) atomicMax(&s_cmax, my_cmax); }
        w = __builtin_amdgcn_readfirstlane(w);
        my_off = w + (uint32_t)__popcll(m & ((1ULL << lane) - 1));
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) s_base = stripe_reserve(out, 0, s_cnt);
    if (threadIdx.x == 64 && s_cmax) stripe_max(out, s_cmax);
    __syncthreads();
    if (go) stripe_put(out, 0, s_base + my_off, r);
    if (r < b.n_reads) {
        b.rmeta[r] = (uint32_t)len | (has_n ? kReadHasN : 0u);
        // 4 bit/base rows [read][strand][wpr] (zero padded, 16-byte aligned): every read of a batch without 2-bit rows; in lean
        // batches only the reads with an N (1 byte/base input: here; packed input: k_exc_rows + k_apply_exc after this kernel)
        if (PACKED ? !lean : (!lean || has_n)) {
            const uint32_t wpr = b.wpr;
            uint4 *row0 = reinterpret_cast<uint4 *>(b.rd4 + (uint64_t)r * 2 * wpr);
            uint4 *row1 = reinterpret_cast<uint4 *>(b.rd4 + ((uint64_t)r * 2 + 1) * wpr);
#pragma unroll
            for (int q = 0; q < NW / 2; q++) {
                if (2 * q < (int)wpr) {
                    uint64_t a0, a1, c0, c1;
                    if (PACKED) {
                        a0 = spread2to4((uint32_t)(f2[q] >> 32)); a1 = spread2to4((uint32_t)f2[q]);
                        c0 = spread2to4((uint32_t)(r2[q] >> 32)); c1 = spread2to4((uint32_t)r2[q]);
                    } else { a0 = fw[PACKED ? 0 : 2 * q]; a1 = fw[PACKED ? 0 : 2 * q + 1]; c0 = rv[PACKED ? 0 : 2 * q]; c1 = rv[PACKED ? 0 : 2 * q + 1]; }
                    row0[q] = make_uint4((uint32_t)a0, (uint32_t)(a0 >> 32), (uint32_t)a1, (uint32_t)(a1 >> 32));
                    row1[q] = make_uint4((uint32_t)c0, (uint32_t)(c0 >> 32), (uint32_t)c1, (uint32_t)(c1 >> 32));
                }
            }
            for (uint32_t q = NW / 2; 2 * q < wpr; q++) { row0[q] = make_uint4(0, 0, 0, 0); row1[q] = make_uint4(0, 0, 0, 0); }
        }
        if (lean && !(TR && wave_full)) {
            uint4 *t0 = reinterpret_cast<uint4 *>(b.rd2 + (uint64_t)r * 2 * (NW / 2));
            uint4 *t1 = reinterpret_cast<uint4 *>(b.rd2 + ((uint64_t)r * 2 + 1) * (NW / 2));
#pragma unroll
            for (int q = 0; q < NW / 4; q++) {
                t0[q] = make_uint4((uint32_t)f2[2 * q], (uint32_t)(f2[2 * q] >> 32), (uint32_t)f2[2 * q + 1], (uint32_t)(f2[2 * q + 1] >> 32));
                t1[q] = make_uint4((uint32_t)r2[2 * q], (uint32_t)(r2[2 * q] >> 32), (uint32_t)r2[2 * q + 1], (uint32_t)(r2[2 * q + 1] >> 32));
            }
        }
        if (!(TR && wave_full)) b.out[r] = h;
    }
    if (TR && wave_full) {
        // A wave's 64 reads are neighbours: their rows (64 bytes each) and result records (20 bytes each) are contiguous in memory.  Stored
        // lane by lane, every store instruction touched 64 lines for 16 (4) bytes of each; through the wave's words of LDS every
        // instruction writes 1 KB (256 bytes) of consecutive memory.
        constexpr int RW = NW / 2;                                   // 16-byte words of a read's two rows (NW = 8: four)
        uint4 *sr = s_tr[wv];
        if (lean) {
#pragma unroll
            for (int q = 0; q < NW / 4; q++) {
                sr[lane * RW + q] = make_uint4((uint32_t)f2[2 * q], (uint32_t)(f2[2 * q] >> 32), (uint32_t)f2[2 * q + 1], (uint32_t)(f2[2 * q + 1] >> 32));
                sr[lane * RW + NW / 4 + q] = make_uint4((uint32_t)r2[2 * q], (uint32_t)(r2[2 * q] >> 32), (uint32_t)r2[2 * q + 1], (uint32_t)(r2[2 * q + 1] >> 32));
            }
            __builtin_amdgcn_wave_barrier();
            uint4 *dst = reinterpret_cast<uint4 *>(b.rd2 + (uint64_t)(r - (uint32_t)lane) * 2 * (NW / 2));
#pragma unroll
            for (int k = 0; k < RW; k++) dst[k * 64 + lane] = sr[k * 64 + lane];
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t *sh = reinterpret_cast<uint32_t *>(sr);
        const uint32_t *hw = reinterpret_cast<const uint32_t *>(&h);
#pragma unroll
        for (int k = 0; k < 5; k++) sh[lane * 5 + k] = hw[k];
        __builtin_amdgcn_wave_barrier();
        uint32_t *od = reinterpret_cast<uint32_t *>(b.out + (r - (uint32_t)lane));
#pragma unroll
        for (int k = 0; k < 5; k++) od[k * 64 + lane] = sh[k * 64 + lane];
    }
}

__global__ void __launch_bounds__(1024) k_init_reads(DevAlignCfg cfg, DevBatch b, uint32_t *__restrict__ act,
                                                      uint32_t *__restrict__ act_cnt, uint32_t *__restrict__ cmax)
{
    __shared__ uint32_t s_cnt, s_base, s_cmax;
    if (threadIdx.x == 0) { s_cnt = 0; s_cmax = 0; }
    __syncthreads();
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    bool go = false;
    uint32_t my_cmax = 0;
    if (r < b.n_reads) {
        int len = (int)b.lens[r];
        bk_hit h;
        h.chrom_id = 0; h.match_loci = 0; h.match_len = 0; h.low_hit_instances = 0; h.rslt = 0;
        h.nar = BK_NAR_NOHIT; h.strand = '?'; h.low_mm = 0; h.nxt_low_mm = 0; h.num_hits = 0; h.mismatches = 0; h.flags = 0;
        int max_ns_seq = 0;
        if (cfg.max_ns) {
            max_ns_seq = (len * cfg.max_ns) / 100;
            if (max_ns_seq < cfg.max_ns) max_ns_seq = cfg.max_ns;
        }
        const uint64_t *fw = b.rd4 + (uint64_t)r * 2 * b.wpr;
        int num_ns = 0;
        bool bad = false;
        for (int w = 0; 16 * w < len; w++) {
            uint64_t x = fw[w] & top_mask(len - 16 * w);
            uint64_t hi = x & 0x4444444444444444ULL;                    // values 4..7
            uint64_t lo = (x | (x >> 1)) & 0x1111111111111111ULL;       // low two bits non-zero
            bad |= ((hi >> 2) & lo) != 0;                               // 5,6,7: not a base the reference accepts
            num_ns += __popcll(hi);
        }
        if (bad || num_ns > max_ns_seq) h.nar = BK_NAR_NS;
        b.out[r] = h;
        b.rmeta[r] = (uint32_t)len | ((bad || num_ns > 0) ? kReadHasN : 0u);
        if (h.nar != BK_NAR_NS) {
            ReadPlan p = make_plan(len, cfg);
            if (p.n_phases > 0) {
                int mm, cl, cd, ofs[1];
                phase_params(p, cfg, 0, mm, cl, cd);
                int nc = core_offsets(len, cl, cd, p.max_slides, ofs, 0);
                if (nc <= kMaxCoresFast) my_cmax = (uint32_t)nc;
                go = true;
            }
        }
    }
    // one global append per block (see k_light)
    const int lane = threadIdx.x & 63;
    uint64_t m = __ballot(go);
    uint32_t my_off = 0;
    for (int off = 32; off > 0; off >>= 1) { uint32_t q = __shfl_down(my_cmax, off); my_cmax = q > my_cmax ? q : my_cmax; }
    if (m) {
        uint32_t w = 0;
        if (lane == 0) { w = atomicAdd(&s_cnt, (uint32_t)__popcll(m)); if (my_cmax) atomicMax(&s_cmax, my_cmax); }
        w = __builtin_amdgcn_readfirstlane(w);
        my_off = w + (uint32_t)__popcll(m & ((1ULL << lane) - 1));
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) s_base = atomicAdd(act_cnt, s_cnt);
    if (threadIdx.x == 64 && s_cmax) atomicMax(cmax, s_cmax);
    __syncthreads();
    if (go) act[s_base + my_off] = r;
}

// reads AlignReads' phases left without any result (candidates for the -a pass)
// Lean batches keep 4 bit/base rows only for the reads with an N.  The kernels of the general family (k_heavy in all its forms,
// k_indel) read rd4 rows; the few reads they are handed get theirs here, widened from the 2-bit rows: one lane per 16-base word.
__global__ void __launch_bounds__(256) k_expand_rd4(DevBatch b, const uint32_t *__restrict__ list, uint32_t n_list)
{
    const uint32_t per_read = 2 * b.wpr;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t item = tid / per_read;
    if (item >= n_list) return;
    const uint32_t rem = (uint32_t)(tid - item * per_read);
    const uint32_t r = list ? list[item] : (uint32_t)item;
    if (b.rmeta[r] & kReadHasN) return;                          // its rows were written by the read preparation
    const uint32_t st = rem >= b.wpr ? 1 : 0, w = rem - st * b.wpr;
    uint64_t v = 0;
    if (w < b.nw) {
        const uint64_t x = b.rd2[((uint64_t)r * 2 + st) * (b.nw / 2) + (w >> 1)];
        v = spread2to4((w & 1) ? (uint32_t)x : (uint32_t)(x >> 32));
    }
    b.rd4[((uint64_t)r * 2 + st) * b.wpr + w] = v;
}

void expand_rd4(const DevBatch &b, const uint32_t *list, uint32_t n_list, hipStream_t s)
{
    if (b.rd2 == nullptr || !n_list) return;
    const uint64_t threads = (uint64_t)n_list * 2 * b.wpr;
    hipLaunchKernelGGL(k_expand_rd4, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, b, list, n_list);
}

__global__ void __launch_bounds__(256) k_max_len(const uint32_t *__restrict__ lens, uint32_t n, uint32_t *__restrict__ out)
{
    __shared__ uint32_t s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    uint32_t v = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w = lens[i];
        v = w > v ? w : v;
    }
    for (int off = 32; off > 0; off >>= 1) { uint32_t w = __shfl_down(v, off); v = w > v ? w : v; }
    if ((threadIdx.x & 63) == 0 && v) atomicMax(&s_max, v);
    __syncthreads();
    if (threadIdx.x == 0 && s_max) atomicMax(out, s_max);
}

void launch_max_len(const uint32_t *lens, uint32_t n, uint32_t *out, hipStream_t s)
{
    uint32_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (n) hipLaunchKernelGGL(k_max_len, dim3(blocks), dim3(256), 0, s, lens, n, out);
}

// the stripes of up to three lists -> their dense forms (appended behind total[i] entries; the counts and the maximum follow)
void launch_compact(const StripeSet &set, uint32_t *const *dense, uint32_t *const *total, int n, uint32_t *max_out, hipStream_t s)
{
    CompactJobs J;
    J.set = set;
    for (int i = 0; i < 3; i++) { J.dense[i] = dense[i < n ? i : 0]; J.total[i] = total[i < n ? i : 0]; }
    J.max_out = max_out;
    J.n = n;
    hipLaunchKernelGGL(k_compact_lists, dim3(1024, (unsigned)n), dim3(256), 0, s, J);
    hipLaunchKernelGGL(k_finish_lists, dim3(1), dim3(64), 0, s, J);
}

void launch_pack_rows(const DevBatch &b, hipStream_t s, const uint32_t *pairs, uint32_t n_pairs)
{
    const uint32_t rpb = 256 / (2 * b.wpr);
    const uint64_t n = pairs != nullptr ? 2ULL * n_pairs : b.n_reads;
    if (!n) return;
    hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), 0, s, b, pairs, n_pairs);
    if (b.pk_words != nullptr && b.pk_nexc) hipLaunchKernelGGL(k_apply_exc, dim3((unsigned)((b.pk_nexc + 255) / 256)), dim3(256), 0, s, b);
}

// stage: at least n_reads + (kListStripes + 2) * 1024 entries; stripe_cnt: kListStripes * 16 words, zero between launches
void launch_prep(const DevAlignCfg &cfg, const DevBatch &b, uint32_t *act, uint32_t *act_cnt, uint32_t *cmax, uint32_t *stage,
                 uint32_t *stripe_cnt, hipStream_t s)
{
    const bool packed = b.pk_words != nullptr;
    const unsigned eblocks = (unsigned)((b.pk_nexc + 255) / 256);
    if (b.nw == 8 || b.nw == 16 || b.nw == kNwLong || b.nw == kNwLongest) {          // register-kernel path: fused pack + init
        const unsigned blocks = (b.n_reads + 255) / 256;
        StripeSet out;
        out.cnt = stripe_cnt;
        out.stage[0] = out.stage[1] = out.stage[2] = stage;
        out.cap = stripe_cap(blocks, 256);
        if (packed) {
            // the exception list first tells every read how many N it holds (the N policy is decided in the fused kernel), and
            // afterwards writes the codes into 4-bit rows: existing ones, or - lean batches - rows made for just these reads
            launch_fill_u64(reinterpret_cast<unsigned long long *>(b.rmeta), ((uint64_t)b.n_reads + 1) / 2, 0ULL, s);      // (rmeta is allocated in whole 8-byte words)
            if (eblocks) hipLaunchKernelGGL(k_mark_exc, dim3(eblocks), dim3(256), 0, s, b);
            if (b.nw == 8) hipLaunchKernelGGL((k_prep_fused<8, true>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
            else if (b.nw == 16) hipLaunchKernelGGL((k_prep_fused<16, true>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
            else if (b.nw == kNwLong) hipLaunchKernelGGL((k_prep_fused<kNwLong, true>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
            else hipLaunchKernelGGL((k_prep_fused<kNwLongest, true>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
            if (eblocks && b.rd2 != nullptr) hipLaunchKernelGGL(k_exc_rows, dim3(eblocks), dim3(256), 0, s, b);
            if (eblocks) hipLaunchKernelGGL(k_apply_exc, dim3(eblocks), dim3(256), 0, s, b);
        } else if (b.nw == 8) hipLaunchKernelGGL((k_prep_fused<8, false>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
        else if (b.nw == 16) hipLaunchKernelGGL((k_prep_fused<16, false>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
        else if (b.nw == kNwLong) hipLaunchKernelGGL((k_prep_fused<kNwLong, false>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
        else hipLaunchKernelGGL((k_prep_fused<kNwLongest, false>), dim3(blocks), dim3(256), 0, s, cfg, b, out);
        launch_compact(out, &act, &act_cnt, 1, cmax, s);
        return;
    }
    launch_pack_rows(b, s);
    hipLaunchKernelGGL(k_init_reads, dim3((b.n_reads + 1023) / 1024), dim3(1024), 0, s, cfg, b, act, act_cnt, cmax);
}

}  // namespace bk
