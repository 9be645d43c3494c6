// bk_dev_trim.h - CSfxArrayV3::AdaptiveTrim on the device, shared by the chimeric form of k_heavy (bk_heavy.hip) and the paired-end
// orphan recovery (bk_rescue.hip).
#pragma once
#include "bk_dev_util.h"

namespace bk {

// CSfxArrayV3::AdaptiveTrim (SfxArrayV2.cpp:5482-5682) for one candidate: the longest stretch of the read that starts and ends in a
// run of >= min_flank matching bases, is at least min_trim long and stays under (max_mm + 1) % mismatches - counting, as the
// reference does, every base of a mismatching run against the length from the stretch's start (first test) and against the
// stretch itself (second test), in double precision.  The reference keeps a table of match / mismatch runs; here the runs are
// read off a bitmap of the mismatching positions (bit i of word i / 64): ATW = 8 words for reads of up to 512 bases (registers), 32 for
// the longest reads the boundary takes (2000 bases; the map then lives in scratch memory - its own instantiation of the kernel).
template <int ATW>
__device__ __forceinline__ int at_run_end(const uint64_t (&bm)[ATW], int p, int n)    // first q > p with bit(q) != bit(p), or n
{
    constexpr int kATWords = ATW;
    const int bit = (int)((bm[p >> 6] >> (p & 63)) & 1);
    int w = p >> 6;
    uint64_t x = (bit ? ~bm[w] : bm[w]) >> (p & 63);
    if (x) { const int q = p + (__ffsll((unsigned long long)x) - 1); return q < n ? q : n; }
    for (w++; w < kATWords && (w << 6) < n; w++) {
        x = bit ? ~bm[w] : bm[w];
        if (x) { const int q = (w << 6) + (__ffsll((unsigned long long)x) - 1); return q < n ? q : n; }
    }
    return n;
}

// returns the trimmed length (0 = nothing acceptable); trim5 / trim3 = bases cut from the start / end of the read as given
template <int ATW>
__device__ int adaptive_trim_dev(const uint64_t *__restrict__ rdw, const uint64_t *__restrict__ tgt, uint64_t t, int len, int min_trim, int max_mm,
                                 int min_flank, int &trim_mm, int &trim5, int &trim3)
{
    constexpr int kATWords = ATW;
    trim_mm = 0; trim5 = 0; trim3 = 0;
    if (len < 25 || len > 64 * kATWords || min_trim < 15 || min_trim > len || max_mm > 15 || min_flank > 10) return 0;
    if (min_flank == 0) min_flank = 1;
    uint64_t bm[kATWords];
#pragma unroll
    for (int w = 0; w < kATWords; w++) bm[w] = 0;
    for (int i = 0; i < len; i += 16) {
        const int nv = len - i < 16 ? len - i : 16;
        const uint64_t x = (nib16(rdw, i) ^ nib16(tgt, t + i)) & top_mask(nv);
        const uint64_t f = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ULL;
        bm[i >> 6] |= (uint64_t)flags_to_bits16(f) << (i & 63);            // bit k of the 16 = base i + k mismatches
    }
    // pass 1: is there an exact run of >= 8; first / last run that may start a stretch, last run that may end one
    bool have8 = false;
    int first_start = -1, last_start = -1, last_end = -1, first_end = -1;
    for (int p = 0; p < len;) {
        const int q = at_run_end<ATW>(bm, p, len), rl = q - p;
        const bool mm = ((bm[p >> 6] >> (p & 63)) & 1) != 0;
        if (!mm) {
            if (rl >= 8) have8 = true;
            if (rl >= min_flank) {
                if (p <= len - min_trim) { last_start = p; if (first_start < 0) first_start = p; }
                if (p + rl >= min_trim) { last_end = p; if (first_end < 0) first_end = p; }
            }
        }
        p = q;
    }
    if (!have8 || first_start < 0 || first_end < 0) return 0;
    const double lim = (max_mm + 1.0) / 100.0;
    int best_len = 0, best_mm = 0, best_start = 0, best_end = 0;
    for (int sp = first_start; sp <= last_start;) {
        const int sq = at_run_end<ATW>(bm, sp, len);
        const bool smm = ((bm[sp >> 6] >> (sp & 63)) & 1) != 0;
        const bool can_start = !smm && (sq - sp) >= min_flank && sp <= len - min_trim;
        if (can_start) {
            int cur_len = 0, cur_mm = 0;
            for (int p = sp; p < len && p <= last_end;) {
                const int q = at_run_end<ATW>(bm, p, len), rl = q - p;
                const bool mm = ((bm[p >> 6] >> (p & 63)) & 1) != 0;
                const bool can_end = !mm && rl >= min_flank && p + rl >= min_trim;
                cur_len += rl;
                p = q;
                if (mm) {
                    if (max_mm == 0) break;
                    cur_mm += rl;
                    if (lim <= (double)cur_mm / (double)(len - sp)) break;
                } else if (best_len == 0) {
                    best_start = sp; best_end = len - (sp + cur_len); best_len = cur_len; best_mm = 0;
                    continue;
                }
                if (cur_len < min_trim || !can_end) continue;
                if (lim <= (double)cur_mm / (double)cur_len) continue;
                if (best_len < cur_len || (best_len == cur_len && (best_mm == 0 || cur_mm < best_mm))) {
                    best_start = sp; best_end = len - (sp + cur_len); best_len = cur_len; best_mm = cur_mm;
                }
            }
        }
        sp = sq;
    }
    if (best_len < min_trim) return 0;
    trim_mm = best_mm; trim5 = best_start; trim3 = best_end;
    return best_len;
}

}  // namespace bk
