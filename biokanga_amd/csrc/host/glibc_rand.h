// The value sequence of glibc's rand() for an unseeded process (srand(1); TYPE_3 additive feedback generator,
// degree 31, separation 3), kept as our own state: the reference draws its N-run mutations (kangax index,
// SfxArrayV2.cpp:1475-1488) and its `-r2` random locus picks (Aligner.cpp:9366) from the process-wide rand(), and in
// our process the HIP runtime consumes values of that shared generator before we get to it.
#pragma once
#include <cstdint>

namespace bk {

class GlibcRand {
public:
    explicit GlibcRand(uint32_t seed = 1)
    {
        int32_t r[344];
        r[0] = (int32_t)(seed ? seed : 1);
        for (int i = 1; i < 31; i++) {
            int64_t v = (16807LL * r[i - 1]) % 2147483647LL;
            if (v < 0) v += 2147483647LL;
            r[i] = (int32_t)v;
        }
        for (int i = 31; i < 34; i++) r[i] = r[i - 31];
        for (int i = 34; i < 344; i++) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
        for (int i = 0; i < 34; i++) st_[i] = (uint32_t)r[344 - 34 + i];
        pos_ = 0;
    }
    // next value of rand(): 0 .. RAND_MAX (2^31 - 1)
    int next()
    {
        // st_ holds the last 34 words, oldest first at pos_
        const uint32_t v = st_[(pos_ + 34 - 31) % 34] + st_[(pos_ + 34 - 3) % 34];
        st_[pos_] = v;
        pos_ = (pos_ + 1) % 34;
        return (int)(v >> 1);
    }

private:
    uint32_t st_[34];
    int pos_;
};

}  // namespace bk
