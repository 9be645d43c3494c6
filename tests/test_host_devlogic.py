"""Device helpers that are pure bit / index arithmetic, compiled for the host and checked against brute force (CPU only):
k2s_bounds (lower / upper bound of a masked key through the sample levels over the second-level keys, knob use_k2s) against a
linear scan; the layout helpers of k_wave's mismatch map (bits_to_imap, imask_word, im_clean) against per-base loops.
The function texts are taken from biokanga_amd/csrc/bk_kernels.hip as they stand (no copy kept here)."""
import os
import re
import subprocess

import helpers

SRC = os.path.join(helpers.ROOT, "biokanga_amd", "csrc", "bk_kernels.hip")


def _between(text, start, end):
    i = text.index(start)
    return text[i:text.index(end, i)]


def _hostify(code):
    code = code.replace("__device__ __forceinline__ ", "static inline ").replace("__restrict__", "")
    code = code.replace("__brev(", "brev32(").replace("#pragma unroll\n", "")
    return code


MAIN = r'''
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
static inline uint32_t brev32(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) if (v >> i & 1) r |= 1u << (31 - i); return r; }
constexpr int kK2Levels = 6;
struct DevIndex { const uint64_t *k2; const uint64_t *k2s; uint64_t k2s_off[6]; int k2s_levels; };
%s
static int test_k2s()
{
    std::mt19937_64 rng(1);
    for (int trial = 0; trial < 120; trial++) {
        const uint64_t n = 1 + rng() %% (trial < 80 ? 5000 : 1500000);
        const int nlev = 1 + rng() %% 6;
        std::vector<uint64_t> k2(n);
        std::vector<std::pair<uint64_t, uint64_t>> buckets;
        for (uint64_t at = 0; at < n;) {
            uint64_t sz = 1 + rng() %% (1 + (rng() %% 3 == 0 ? n : 300));
            if (at + sz > n) sz = n - at;
            const int bits = 1 + rng() %% 20;
            for (uint64_t i = 0; i < sz; i++) k2[at + i] = (rng() & ((1ULL << bits) - 1)) << 40;
            std::sort(k2.begin() + at, k2.begin() + at + sz);
            if (rng() %% 4 == 0) k2[at + sz - 1] = ~0ULL;
            buckets.push_back({at, at + sz});
            at += sz;
        }
        DevIndex ix; ix.k2 = k2.data(); ix.k2s_levels = nlev;
        std::vector<uint64_t> nl(7); nl[0] = n; uint64_t tot = 0;
        for (int L = 1; L <= 6; L++) { nl[L] = L <= nlev ? (nl[L - 1] + 7) / 8 : 0; ix.k2s_off[L - 1] = tot; tot += nl[L]; }
        std::vector<uint64_t> lev(tot);
        for (int L = 1; L <= nlev; L++) {
            const uint64_t *src = L == 1 ? k2.data() : lev.data() + ix.k2s_off[L - 2];
            for (uint64_t i = 0; i < nl[L]; i++) { uint64_t j = 8 * i + 7; lev[ix.k2s_off[L - 1] + i] = src[j < nl[L - 1] ? j : nl[L - 1] - 1]; }
        }
        ix.k2s = lev.data();
        for (auto &b : buckets)
            for (int q = 0; q < 6; q++) {
                const uint64_t m = ~0ULL << (rng() %% 50);
                const uint64_t q2 = (q & 1) ? (k2[b.first + rng() %% (b.second - b.first)] & m) : (((rng() & ((1ULL << 20) - 1)) << 40) & m);
                if (q2 == (~0ULL & m)) continue;
                uint64_t l1, l2, e1 = b.first, e2;
                k2s_bounds(ix, b.first, b.second, m, q2, l1, l2);
                while (e1 < b.second && k2_cmp(k2[e1], m, q2) < 0) e1++;
                for (e2 = e1; e2 < b.second && k2_cmp(k2[e2], m, q2) <= 0;) e2++;
                if (l1 != e1 || l2 != e2) { printf("k2s_bounds: trial %%d bucket [%%llu,%%llu) got %%llu %%llu want %%llu %%llu\n", trial, (unsigned long long)b.first, (unsigned long long)b.second, (unsigned long long)l1, (unsigned long long)l2, (unsigned long long)e1, (unsigned long long)e2); return 1; }
            }
    }
    return 0;
}
template <int NW> static int test_imap()
{
    std::mt19937_64 rng(7);
    for (int trial = 0; trial < 20000; trial++) {
        uint64_t bm[NW / 4], im[NW / 4];
        for (int i = 0; i < NW / 4; i++) { bm[i] = (trial & 1) ? (1ULL << (rng() %% 64)) | (1ULL << (rng() %% 64)) : (rng() & rng() & rng()); im[i] = bits_to_imap(bm[i]); }
        int pop = 0, pop2 = 0;
        for (int i = 0; i < NW / 4; i++) { pop += __builtin_popcountll(bm[i]); pop2 += __builtin_popcountll(im[i]); }
        if (pop != pop2) { printf("imap: popcount differs\n"); return 1; }
        const int o = (int)(rng() %% (16 * NW)), cl = 1 + (int)(rng() %% (16 * NW - o));
        uint64_t mask[NW / 4];
        for (int i = 0; i < NW / 4; i++) mask[i] = imask_word(o, o + cl, i);
        bool clean = true;
        for (int b = o; b < o + cl; b++) clean &= !((bm[b >> 6] >> (b & 63)) & 1);
        if (im_clean<NW>(im, mask) != clean) { printf("imap: NW %%d o %%d cl %%d: im_clean %%d, bit loop %%d\n", NW, o, cl, (int)!clean, (int)clean); return 1; }
    }
    return 0;
}
int main() { if (test_k2s() || test_imap<8>() || test_imap<16>()) return 1; printf("ok\n"); return 0; }
'''


def test_device_bit_helpers_against_brute_force(tmp_path):
    text = open(SRC).read()
    parts = [_between(text, "__device__ __forceinline__ int k2_cmp(", "// level L of the samples from level L - 1"),
             _between(text, "__device__ __forceinline__ uint64_t spread32(", "template <int NW>\n__device__ __forceinline__ void window_to_iwindow(")]
    code = MAIN % _hostify("\n".join(parts))
    assert "k2s_bounds" in code and "imask_word" in code and "im_clean" in code
    src = str(tmp_path / "devlogic.cpp")
    exe = str(tmp_path / "devlogic")
    open(src, "w").write(code)
    subprocess.check_call(helpers.cxx() + ["-o", exe, src])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


def _replay_in_order(recs, mm, init, max_hits):
    """the Low / NxtLow / instances state machine of LocateCoreMultiples as k_flat's in-order replay runs it (SfxArrayV2.cpp:6093-6205):
    recs = one result byte per candidate in walk order (255 = not a candidate)"""
    low_inst, low_mm, nxt, best, n_cand = 0, init, init, -1, 0
    for x, cm in enumerate(recs):
        if cm == 255:
            continue
        n_cand += 1
        if cm > mm or cm >= nxt:
            continue
        if cm < low_mm:
            low_inst, nxt, low_mm, best = 1, low_mm, cm, x
        elif cm == low_mm:
            low_inst += 1
        else:
            nxt = cm
        if low_inst > max_hits and low_mm == 0:
            break
    return low_inst, low_mm, nxt, best, n_cand


def _reduced(recs, mm, init, max_hits):
    """k_flat's reduction over the candidates' lanes: min of (mismatches << 16 | candidate number) over the acceptable candidates,
    how many share that count, the smallest count above it, how many were looked at; None = the early exit applies (replayed in order)"""
    k1, cnt_min, looked, nx = 0xFFFFFFFF, 0, 0, 0xFFFFFFFF
    for x, cm in enumerate(recs):
        if cm != 255:
            looked += 1
            if cm <= mm:
                k1 = min(k1, (cm << 16) | x)
    for cm in recs:
        if cm != 255 and cm <= mm:
            if cm == (k1 >> 16):
                cnt_min += 1
            else:
                nx = min(nx, cm)
    if k1 != 0xFFFFFFFF and (k1 >> 16) == 0 and cnt_min > max_hits:
        return None
    if k1 == 0xFFFFFFFF:
        return 0, init, init, -1, looked
    return cnt_min, k1 >> 16, min(nx, init), k1 & 0xFFFF, looked


def test_flat_reduction_equals_the_in_order_replay():
    """the claim behind k_flat's second half: up to the early exit after MaxHits + 1 exact matches the outcome of a read does not
    depend on the order its candidates are looked at"""
    import numpy as np
    rng = np.random.default_rng(11)
    exits = 0
    for _ in range(20000):
        n = int(rng.integers(0, 40))
        mm = int(rng.integers(0, 8))
        delta = int(rng.integers(1, 3))
        init = mm + delta + 1
        max_hits = int(rng.choice([1, 1, 2, 5]))
        hi = int(rng.choice([3, 6, 12]))
        recs = [255 if rng.integers(0, 5) == 0 else int(rng.integers(0, hi)) for _ in range(n)]
        want = _replay_in_order(recs, mm, init, max_hits)
        got = _reduced(recs, mm, init, max_hits)
        if got is None:
            exits += 1
            assert want[1] == 0 and want[0] == max_hits + 1           # the exit: exactly MaxHits + 1 exact instances counted
            continue
        assert got == want, (recs, mm, init, max_hits, got, want)
    assert exits > 200
