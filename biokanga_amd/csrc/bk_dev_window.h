// bk_dev_window.h - the target window of a candidate against the read, in registers: 4 bit/base windows (Window), 2 bit/base windows
// with the laid-together mismatch map (IWindow), windows out of the suffix-ordered window array (eval_swin2i); shared by the
// extend kernels (bk_extend.hip), the wave kernel (bk_wave.hip) and the window array's builder (bk_index.hip).
#pragma once
#include "bk_dev_util.h"

namespace bk {

// ------------------------------------------------------------------------------------------------
// Register-resident candidate evaluation for reads of <= 16*NW bases (NW = 8 or 16 sixteen-base words).
// One pass over the target window gives everything the reference's loop derives per candidate:
//   * EOS inside the window  <=> the read would cross an entry boundary (MapChunkHit2Entry bounds test)
//   * a bit-per-base mismatch map: Hamming distance = popcount, "core c' matches exactly here" = its
//     bit range is clear (which is what the reference's dedupe set encodes, see k_wave)
// The nw+1 window words are independent loads, all in flight together.

template <int NW>
struct Window {
    uint64_t bm[NW / 4];        // mismatch bit map, bit b of bm[b/64] = base b differs
    int mm;
    bool eos;
};


template <int NW>
__device__ __forceinline__ void eval_window(const uint64_t (&rw)[NW], int len, const uint64_t *__restrict__ tgt, uint64_t t,
                                            Window<NW> &w)
{
    // window fetched with 16-byte loads: ceil((len/16 + 2) / 2) instructions instead of len/16 + 1
    const uint64_t i0 = t >> 4;
    const unsigned s = (unsigned)(t & 15) << 2;
    const bool odd = (i0 & 1) != 0;
    const uint4 *__restrict__ blk = reinterpret_cast<const uint4 *>(tgt) + (i0 >> 1);
    constexpr int NB = NW / 2 + 1;
    uint64_t r2[2 * NB];
    // words touched from i0 on: the window's bases plus the shift spill; 16-byte blocks from the even word
    // at or below i0.  Only what is needed is fetched (a 100-base window needs the 5th block in 3 of 32
    // alignments) - these kernels are bound by 32-byte sectors moved, see DESIGN.md
    const int nwords = ((int)(t & 15) + len + 15) >> 4;
    const int nblk = ((odd ? 1 : 0) + nwords + 1) >> 1;
#pragma unroll
    for (int q = 0; q < NB; q++) {
        // block q holds words 2q, 2q+1 counted from the even word at or below i0
        if (q < nblk) {
            uint4 v = blk[q];
            r2[2 * q] = ((uint64_t)v.y << 32) | v.x;
            r2[2 * q + 1] = ((uint64_t)v.w << 32) | v.z;
        } else {
            r2[2 * q] = 0;
            r2[2 * q + 1] = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < NW / 4; k++) w.bm[k] = 0;
    uint64_t eosacc = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        if (16 * k < len) {
            uint64_t a = odd ? r2[k + 1] : r2[k];
            uint64_t b = odd ? r2[k + 2] : r2[k + 1];
            uint64_t win = (a << s) | ((b >> 1) >> (63 - s));
            uint64_t m = top_mask(len - 16 * k);
            uint64_t x = (rw[k] ^ win) & m;
            uint64_t f = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ULL;
            eosacc |= win & (win >> 1) & (win >> 2) & m & 0x1111111111111111ULL;      // nibble 7 = EOS
            w.bm[k >> 2] |= (uint64_t)flags_to_bits16(f) << (16 * (k & 3));
        }
    }
    int mm = 0;
#pragma unroll
    for (int k = 0; k < NW / 4; k++) mm += __popcll(w.bm[k]);
    w.mm = mm;
    w.eos = eosacc != 0;
}

// eval_window for the rare window that the 2-bit compare cannot decide (N or a sequence end nearby): the same result, one 16-base
// word of read and target at a time from memory, so that the path costs the kernels that carry it a few registers instead of the
// NW + NW/2 + 2 words the all-at-once form holds
template <int NW>
__device__ __forceinline__ void eval_window_rare(const RdRow &rdrow, int len, const uint64_t *__restrict__ tgt, uint64_t t,
                                              Window<NW> &w)
{
#pragma unroll
    for (int k = 0; k < NW / 4; k++) w.bm[k] = 0;
    uint64_t eosacc = 0;
    const int nk = (len + 15) >> 4;
#pragma unroll 1
    for (int k = 0; k < nk; k++) {
        const uint64_t win = nib16(tgt, t + 16 * (uint64_t)k);
        const uint64_t m = top_mask(len - 16 * k);
        const uint64_t x = (rdrow.word16(k) ^ win) & m;
        const uint64_t f = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ULL;
        eosacc |= win & (win >> 1) & (win >> 2) & m & 0x1111111111111111ULL;
        const uint64_t bits = (uint64_t)flags_to_bits16(f) << (16 * (k & 3));
#pragma unroll
        for (int q = 0; q < NW / 4; q++) w.bm[q] |= (k >> 2) == q ? bits : 0ULL;
    }
    int mm = 0;
#pragma unroll
    for (int k = 0; k < NW / 4; k++) mm += __popcll(w.bm[k]);
    w.mm = mm;
    w.eos = eosacc != 0;
}

// true when bases [o, o+cl) of the read all match the window (cl >= 1)
template <int NW>
__device__ __forceinline__ bool core_clean(const Window<NW> &w, int o, int cl)
{
    const int hi = o + cl;
    bool dirty = false;
#pragma unroll
    for (int k = 0; k < NW / 4; k++) {
        int a = o > 64 * k ? o - 64 * k : 0;
        int b = hi < 64 * k + 64 ? hi - 64 * k : 64;
        if (a < b) {
            uint64_t m = (b >= 64 ? ~0ULL : ((1ULL << b) - 1)) & ~((1ULL << a) - 1);
            dirty |= (w.bm[k] & m) != 0;
        }
    }
    return !dirty;
}

template <int NW>
__device__ __forceinline__ void load_read_words(const uint64_t *__restrict__ rdw, int len, uint64_t (&rw)[NW])
{
    // rd4 rows are 16-byte aligned (wpr is even): 16-byte loads
    const uint4 *__restrict__ p = reinterpret_cast<const uint4 *>(rdw);
#pragma unroll
    for (int q = 0; q < NW / 2; q++) {
        if (32 * q < len) {
            uint4 v = p[q];
            rw[2 * q] = ((uint64_t)v.y << 32) | v.x;
            rw[2 * q + 1] = ((uint64_t)v.w << 32) | v.z;
        } else {
            rw[2 * q] = 0;
            rw[2 * q + 1] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2-bit window compare.  Read side: per read and strand NW/2 words at 2 bit/base (N held as A) plus NW/4
// words of "this read base is N" in the mismatch-map format; target side: DevIndex::tgt2.  Only for
// windows whose 64-base blocks hold no N/EOS (window_flagged), where a read N always is a mismatch and
// nothing else can differ from the 4-bit compare.

__device__ __forceinline__ bool window_flagged(const DevIndex &ix, uint64_t t, int len)
{
    const uint64_t g0 = t >> ix.flag_shift, g1 = (t + (uint64_t)len - 1) >> ix.flag_shift;     // a window spans <= 2 regions
    return (((ix.nflag[g0 >> 3] >> (g0 & 7)) | (ix.nflag[g1 >> 3] >> (g1 & 7))) & 1) != 0;
}

// The window's 16-byte blocks are requested by window2_load and consumed by window2_compare: a kernel that has several
// candidates per lane (k_flat) issues the loads of all of them before it touches the first result.
template <int NW>
__device__ __forceinline__ void window2_load(const uint64_t *__restrict__ tgt2, const uint64_t *__restrict__ tgt2s, uint64_t t, int len,
                                             uint4 (&v)[NW / 4 + 1])
{
    const uint64_t i0 = t >> 5;
    const bool odd = (i0 & 1) != 0;
    // tgt2s (optional) is the same data stored again 32 bytes later: a window that would straddle a 64-byte
    // line in one copy lies inside a line of the other (16-byte block index 2 or 3 within the line -> 0 or 1)
    const uint64_t blk0 = i0 >> 1;
    const uint4 *__restrict__ blk = (tgt2s != nullptr && (blk0 & 2)) ? reinterpret_cast<const uint4 *>(tgt2s) + (blk0 - 2)
                                                                      : reinterpret_cast<const uint4 *>(tgt2) + blk0;
    const int nwords = ((int)(t & 31) + len + 31) >> 5;
    const int nblk = ((odd ? 1 : 0) + nwords + 1) >> 1;
#pragma unroll
    for (int q = 0; q < NW / 4 + 1; q++) v[q] = q < nblk ? blk[q] : make_uint4(0, 0, 0, 0);
}

template <int NW>
__device__ __forceinline__ void window2_compare(const uint64_t (&r2w)[NW / 2], const uint64_t (&rnm)[NW / 4], int len, uint64_t t,
                                                const uint4 (&v)[NW / 4 + 1], Window<NW> &w)
{
    const unsigned s = (unsigned)(t & 31) << 1;
    const bool odd = ((t >> 5) & 1) != 0;
    constexpr int NB = NW / 4 + 1;
    uint64_t r[2 * NB];
#pragma unroll
    for (int q = 0; q < NB; q++) {
        r[2 * q] = ((uint64_t)v[q].y << 32) | v[q].x;
        r[2 * q + 1] = ((uint64_t)v[q].w << 32) | v[q].z;
    }
#pragma unroll
    for (int k = 0; k < NW / 4; k++) w.bm[k] = rnm[k];
#pragma unroll
    for (int k = 0; k < NW / 2; k++) {
        if (32 * k < len) {
            uint64_t a = odd ? r[k + 1] : r[k];
            uint64_t b = odd ? r[k + 2] : r[k + 1];
            uint64_t win = (a << s) | ((b >> 1) >> (63 - s));
            uint64_t x = r2w[k] ^ win;
            uint64_t y = (x | (x >> 1)) & 0x5555555555555555ULL;        // base j of the word: bit 62 - 2j
            const int rem = len - 32 * k;
            if (rem < 32) y &= ~0ULL << (64 - 2 * rem);
            uint64_t g = __brevll(y) >> 1;                               // base j: bit 2j
            g = (g | (g >> 1)) & 0x3333333333333333ULL;
            g = (g | (g >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
            g = (g | (g >> 4)) & 0x00FF00FF00FF00FFULL;
            g = (g | (g >> 8)) & 0x0000FFFF0000FFFFULL;
            g = (g | (g >> 16)) & 0x00000000FFFFFFFFULL;
            w.bm[k >> 1] |= g << (32 * (k & 1));
        }
    }
    int mm = 0;
#pragma unroll
    for (int k = 0; k < NW / 4; k++) mm += __popcll(w.bm[k]);
    w.mm = mm;
    w.eos = false;
}

template <int NW>
__device__ __forceinline__ void eval_window2(const uint64_t (&r2w)[NW / 2], const uint64_t (&rnm)[NW / 4], int len,
                                             const uint64_t *__restrict__ tgt2, const uint64_t *__restrict__ tgt2s, uint64_t t,
                                             Window<NW> &w)
{
    uint4 v[NW / 4 + 1];
    window2_load<NW>(tgt2, tgt2s, t, len, v);
    window2_compare<NW>(r2w, rnm, len, t, v, w);
}

template <int NW>
__device__ __forceinline__ void load_read_words2(const uint64_t *__restrict__ row, uint64_t (&r2w)[NW / 2])
{
    // rows are NW/2 words = a multiple of 16 bytes
    const uint4 *__restrict__ p = reinterpret_cast<const uint4 *>(row);
#pragma unroll
    for (int q = 0; q < NW / 4; q++) {
        uint4 u = p[q];
        r2w[2 * q] = ((uint64_t)u.y << 32) | u.x;
        r2w[2 * q + 1] = ((uint64_t)u.w << 32) | u.z;
    }
}

// ------------------------------------------------------------------------------------------------
// The wave kernels' form of the 2-bit compare.  k_wave spends two thirds of its issue slots on vector ALU work, and half of
// eval_window2 is the squeeze of the pair-per-base difference into one bit per base, which only exists so that core_clean can
// build its masks in base units.  Here the map stays where the compare leaves it: word i covers bases 64i .. 64i + 63, base
// 64i + j (j < 32) at bit 62 - 2j and base 64i + 32 + j at bit 63 - 2j (two 32-base compare words laid into each other, one
// shift-or).  The Hamming distance still is the popcount, and "core c matches exactly here" is an AND with a mask of the same
// layout that the wave computes once per read (the core geometry is the same for every candidate) and keeps in LDS.

template <int NW>
struct IWindow {
    uint64_t im[NW / 4];
    int mm;
    bool eos;
};

__device__ __forceinline__ uint64_t spread32(uint32_t v)       // bit p -> bit 2p
{
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x << 2)) & 0x3333333333333333ULL;
    x = (x | (x << 1)) & 0x5555555555555555ULL;
    return x;
}

// one bit per base (bit b = base b of the 64, Window::bm / the read rows' N words) -> the laid-together form
__device__ __forceinline__ uint64_t bits_to_imap(uint64_t bm)
{
    return spread32(__brev((uint32_t)bm)) | (spread32(__brev((uint32_t)(bm >> 32))) << 1);
}

__device__ __forceinline__ uint64_t pair_mask32(int a, int b)   // bases [a, b) of one 32-base compare word, clamped to it
{
    a = a < 0 ? 0 : a;
    b = b > 32 ? 32 : b;
    return a < b ? ((~0ULL >> (2 * a)) & (~0ULL << (64 - 2 * b)) & 0x5555555555555555ULL) : 0ULL;
}

// word i of the mask selecting bases [o, h) in an IWindow map
__device__ __forceinline__ uint64_t imask_word(int o, int h, int i)
{
    return pair_mask32(o - 64 * i, h - 64 * i) | (pair_mask32(o - 64 * i - 32, h - 64 * i - 32) << 1);
}

template <int NW>
__device__ __forceinline__ bool im_clean(const uint64_t (&im)[NW / 4], const uint64_t *__restrict__ mask)
{
    uint64_t a = 0;
#pragma unroll
    for (int i = 0; i < NW / 4; i++) a |= im[i] & mask[i];
    return a == 0;
}

template <int NW>
__device__ __forceinline__ void window_to_iwindow(const Window<NW> &w4, IWindow<NW> &w)
{
#pragma unroll
    for (int i = 0; i < NW / 4; i++) w.im[i] = bits_to_imap(w4.bm[i]);
    w.mm = w4.mm;
    w.eos = w4.eos;
}

template <bool WIDE>
__device__ __forceinline__ bool window_flagged_t(const DevIndex &ix, uint64_t t, int len)
{
    if (WIDE) return window_flagged(ix, t, len);
    // 4-byte indexes: the same test in 32-bit arithmetic
    const uint32_t t0 = (uint32_t)t;
    uint32_t t1 = t0 + (uint32_t)(len - 1);
    t1 = t1 < t0 ? 0xFFFFFFFFu : t1;
    const uint32_t g0 = t0 >> ix.flag_shift, g1 = t1 >> ix.flag_shift;
    return (((ix.nflag[g0 >> 3] >> (g0 & 7)) | (ix.nflag[g1 >> 3] >> (g1 & 7))) & 1) != 0;
}

template <int NW, bool WIDE>
__device__ __forceinline__ void eval_window2i(const uint64_t (&r2w)[NW / 2], const uint64_t (&rni)[NW / 4], int len,
                                              const uint64_t *__restrict__ tgt2, const uint64_t *__restrict__ tgt2s, uint64_t t,
                                              IWindow<NW> &w)
{
    // the loads of eval_window2 (4-byte indexes: block numbers fit 32 bits)
    const unsigned s = (unsigned)(t & 31) << 1;
    const bool odd = ((t >> 5) & 1) != 0;
    const uint4 *__restrict__ blk;
    if (WIDE) {
        const uint64_t blk0 = t >> 6;
        blk = (tgt2s != nullptr && (blk0 & 2)) ? reinterpret_cast<const uint4 *>(tgt2s) + (blk0 - 2) : reinterpret_cast<const uint4 *>(tgt2) + blk0;
    } else {
        const uint32_t blk0 = (uint32_t)t >> 6;
        blk = (tgt2s != nullptr && (blk0 & 2)) ? reinterpret_cast<const uint4 *>(tgt2s) + (blk0 - 2) : reinterpret_cast<const uint4 *>(tgt2) + blk0;
    }
    constexpr int NB = NW / 4 + 1;
    uint64_t r[2 * NB];
    const int nwords = ((int)(t & 31) + len + 31) >> 5;
    const int nblk = ((odd ? 1 : 0) + nwords + 1) >> 1;
#pragma unroll
    for (int q = 0; q < NB; q++) {
        if (q < nblk) {
            uint4 v = blk[q];
            r[2 * q] = ((uint64_t)v.y << 32) | v.x;
            r[2 * q + 1] = ((uint64_t)v.w << 32) | v.z;
        } else {
            r[2 * q] = 0;
            r[2 * q + 1] = 0;
        }
    }
    int mm = 0;
    uint64_t even = 0;
#pragma unroll
    for (int k = 0; k < NW / 2; k++) {
        uint64_t y = 0;
        if (32 * k < len) {
            uint64_t a = odd ? r[k + 1] : r[k];
            uint64_t b = odd ? r[k + 2] : r[k + 1];
            uint64_t win = (a << s) | ((b >> 1) >> (63 - s));
            uint64_t x = r2w[k] ^ win;
            y = (x | (x >> 1)) & 0x5555555555555555ULL;                  // base j of the word: bit 62 - 2j
            const int rem = len - 32 * k;
            if (rem < 32) y &= ~0ULL << (64 - 2 * rem);
        }
        if (k & 1) {
            // only even bits are set in y: the shift does not carry between the halves
            const uint32_t lo = ((uint32_t)y << 1) | (uint32_t)even, hi = ((uint32_t)(y >> 32) << 1) | (uint32_t)(even >> 32);
            const uint64_t m = (((uint64_t)hi << 32) | lo) | rni[k >> 1];
            w.im[k >> 1] = m;
            mm += __popcll(m);
        } else
            even = y;
    }
    w.mm = mm;
    w.eos = false;
}

// eval_window2i with the window taken from the candidate's entry of the suffix-ordered window array (DevIndex::swin): the three
// 16-byte words of entry `e`; the window starts bofs = kSwPre - (core offset) bases into it - the same for every lane of the wave.
// the compare of eval_swin2i once the five words the window starts in are known
template <int NW>
__device__ __forceinline__ void swin2i_compare(const uint64_t (&r2w)[NW / 2], const uint64_t (&rni)[NW / 4], int len, const uint64_t (&q)[5], unsigned s,
                                               IWindow<NW> &w)
{
    int mm = 0;
    uint64_t even = 0;
#pragma unroll
    for (int k = 0; k < NW / 2; k++) {
        uint64_t y = 0;
        if (k < 4 && 32 * k < len) {
            const uint64_t win = (q[k < 4 ? k : 0] << s) | ((q[k < 4 ? k + 1 : 0] >> 1) >> (63 - s));
            const uint64_t x = r2w[k] ^ win;
            y = (x | (x >> 1)) & 0x5555555555555555ULL;
            const int rem = len - 32 * k;
            if (rem < 32) y &= ~0ULL << (64 - 2 * rem);
        }
        if (k & 1) {
            const uint32_t lo = ((uint32_t)y << 1) | (uint32_t)even, hi = ((uint32_t)(y >> 32) << 1) | (uint32_t)(even >> 32);
            const uint64_t m = (((uint64_t)hi << 32) | lo) | rni[k >> 1];
            w.im[k >> 1] = m;
            mm += __popcll(m);
        } else
            even = y;
    }
    w.mm = mm;
    w.eos = false;
}

// The same compare in 16-base dwords for a window start that is the same in every lane: window dword w is one funnel shift
// (v_alignbit_b32) of entry dwords D2 + w and D2 + w + 1, whose registers are known at compile time once the scalar unit has branched
// on D2; five vector instructions per 16 bases and six per 64-base map word instead of the 64-bit shifts, selects and splits above.
template <int NW, int D2, int E>
__device__ __forceinline__ void swin2i_compare32(const uint64_t (&r2w)[NW / 2], const uint64_t (&rni)[NW / 4], int len, const uint32_t (&S)[4 * E + 1], unsigned sh,
                                                 IWindow<NW> &w)
{
    static_assert(D2 >= -1 && D2 <= SwGeo<E>::pre / 16, "the window starts inside the entry's lead");
    int mm = 0;
#pragma unroll
    for (int i = 0; i < NW / 4; i++) {
        uint32_t y[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int wd = 4 * i + d;
            y[d] = 0;
            // (dwords the entry does not hold are never asked for: the caller only comes here with the whole window inside the entry)
            if (D2 + wd + 1 <= 4 * E && 16 * wd < len) {
                const uint32_t hi = D2 + wd >= 0 ? S[D2 + wd >= 0 ? D2 + wd : 0] : 0u;
                const uint32_t lo = S[D2 + wd + 1 <= 4 * E ? D2 + wd + 1 : 4 * E];
                const uint32_t win = __builtin_amdgcn_alignbit(hi, lo, sh);
                const uint32_t rw = (wd & 1) ? (uint32_t)r2w[wd >> 1] : (uint32_t)(r2w[wd >> 1] >> 32);
                const uint32_t x = rw ^ win;
                uint32_t yy = (x | (x >> 1)) & 0x55555555u;                 // base j of the dword: bit 30 - 2j
                const int rem = len - 16 * wd;
                if (rem < 16) yy &= ~0u << (32 - 2 * rem);
                y[d] = yy;
            }
        }
        // the map word of these 64 bases: base j < 32 at bit 62 - 2j, base 32 + j at bit 63 - 2j
        const uint32_t mhi = y[0] | (y[2] << 1), mlo = y[1] | (y[3] << 1);
        const uint64_t m = (((uint64_t)mhi << 32) | mlo) | rni[i];
        w.im[i] = m;
        mm += __popcll(m);
    }
    w.mm = mm;
    w.eos = false;
}

// UNIFORM: bofs is the same for every lane of the wave (one core per round) - the choice of the starting word is then a branch the scalar
// unit takes instead of ten selects per lane.  E: 16-byte words of an entry (SwGeo); the per-lane form exists for E = 3 only
template <int NW, int E, int D2, int LAST>
__device__ __forceinline__ void swin2i_case(const uint64_t (&r2w)[NW / 2], const uint64_t (&rni)[NW / 4], int len, const uint32_t (&S)[4 * E + 1], unsigned sh, int d2,
                                            IWindow<NW> &w)
{
    // (a chain of scalar branches over the starting dwords D2 .. LAST)
    if constexpr (D2 > LAST) return;
    else {
        if (d2 == D2 || D2 == LAST) swin2i_compare32<NW, D2, E>(r2w, rni, len, S, sh, w);
        else swin2i_case<NW, E, D2 + 1, LAST>(r2w, rni, len, S, sh, d2, w);
    }
}

template <int NW, bool UNIFORM, int E = 3>
__device__ __forceinline__ void eval_swin2i(const uint64_t (&r2w)[NW / 2], const uint64_t (&rni)[NW / 4], int len, const uint4 (&e)[E],
                                            int bofs, IWindow<NW> &w)
{
    if (UNIFORM) {
        // the entry as 4 E + 1 16-base dwords in base order (the 64-bit words hold their first base in the top bits)
        uint32_t S[4 * E + 1];
#pragma unroll
        for (int i = 0; i < E; i++) { S[4 * i] = e[i].y; S[4 * i + 1] = e[i].x; S[4 * i + 2] = e[i].w; S[4 * i + 3] = e[i].z; }
        S[4 * E] = 0u;
        const int ub = __builtin_amdgcn_readfirstlane(bofs);
        const int rb = (ub & 15) << 1;
        const unsigned sh = (unsigned)(32 - rb) & 31u;
        const int d2 = (ub >> 4) - (rb ? 0 : 1);                  // -1 .. pre / 16 (bofs <= pre)
        if constexpr (E == 3) {
            switch (d2) {
            case -1: swin2i_compare32<NW, -1, 3>(r2w, rni, len, S, sh, w); break;
            case 0: swin2i_compare32<NW, 0, 3>(r2w, rni, len, S, sh, w); break;
            case 1: swin2i_compare32<NW, 1, 3>(r2w, rni, len, S, sh, w); break;
            case 2: swin2i_compare32<NW, 2, 3>(r2w, rni, len, S, sh, w); break;
            case 3: swin2i_compare32<NW, 3, 3>(r2w, rni, len, S, sh, w); break;
            case 4: swin2i_compare32<NW, 4, 3>(r2w, rni, len, S, sh, w); break;
            default: swin2i_compare32<NW, 5, 3>(r2w, rni, len, S, sh, w); break;
            }
        } else if (d2 < 4)
            swin2i_case<NW, E, -1, 3>(r2w, rni, len, S, sh, d2, w);
        else
            swin2i_case<NW, E, 4, SwGeo<E>::pre / 16>(r2w, rni, len, S, sh, d2, w);
        return;
    }
    if constexpr (E == 3) {
        uint64_t r[6], q[5];                                   // (reads of up to kSwLen <= 128 bases: four compare words at most)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const uint4 v = e[i];
            r[2 * i] = ((uint64_t)v.y << 32) | v.x;
            r[2 * i + 1] = ((uint64_t)v.w << 32) | v.z;
        }
        const int w0 = bofs >> 5;                                  // 0..2
        const unsigned s = (unsigned)(bofs & 31) << 1;
#pragma unroll
        for (int i = 0; i < 5; i++) q[i] = w0 == 0 ? r[i] : (w0 == 1 ? r[i + 1] : (i + 2 < 6 ? r[i + 2 < 6 ? i + 2 : 5] : 0ULL));
        swin2i_compare<NW>(r2w, rni, len, q, s, w);
    }
}

}  // namespace bk
