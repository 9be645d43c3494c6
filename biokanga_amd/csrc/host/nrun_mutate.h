// nrun_mutate.h - what `biokanga index` does to one chunk of a sequence as it takes it in (kangax ProcessFastaFile, kangax.cpp:626-660):
// the soft-mask flag (0x08) comes off every base, and inside runs of more than 25 Ns - where four more Ns follow - every 13th N becomes a
// random base, drawn from the process-wide rand() (glibc_rand.h).  The counter starts again with every chunk and within five bases of
// a chunk's end.  The reference looks at a base with its flag off and at the four in front of it with theirs still on (a lowercase n
// there does not count as N); so does this, eight flag-free bases at a time where none of them is an N.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "glibc_rand.h"

namespace bk {

inline void mutate_n_runs(uint8_t *p, size_t chunk, GlibcRand &rnd)
{
    constexpr uint8_t N = 4;
    int seq_ns = 0;
    size_t k = 0;
    while (k < chunk) {
        if (k + 8 <= chunk) {
            uint64_t v;
            memcpy(&v, p + k, 8);
            const uint64_t t = (v & 0xF7F7F7F7F7F7F7F7ull) ^ 0x0404040404040404ull;       // a zero byte where an N or n stands
            if (!((t - 0x0101010101010101ull) & ~t & 0x8080808080808080ull)) {
                seq_ns = 0;
                k += 8;
                continue;
            }
        }
        if ((uint8_t)(p[k] & 0xF7) == N && k + 5 < chunk) {
            if (++seq_ns > 25 && p[k + 1] == N && p[k + 2] == N && p[k + 3] == N && p[k + 4] == N && !(seq_ns % 13)) p[k] = (uint8_t)(rnd.next() % 4);
        } else
            seq_ns = 0;
        k++;
    }
    for (size_t i = 0; i < chunk; i++) p[i] &= 0xF7;
}

}  // namespace bk
