"""Device helpers that are pure bit / index arithmetic, compiled for the host and checked against brute force (CPU only):
the layout helpers of k_wave's mismatch map (bits_to_imap, imask_word, im_clean) against per-base loops.
The function texts are taken from the device sources under biokanga_amd/csrc/ (bk_dev_*.h, *.hip) as they stand (no copy kept here)."""
import os
import re
import subprocess

import helpers

CSRC = os.path.join(helpers.ROOT, "biokanga_amd", "csrc")


def _device_sources():
    """the text of every device source file, shared helper headers first"""
    names = sorted(f for f in os.listdir(CSRC) if f.startswith("bk_dev_") and f.endswith(".h")) + sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    return "\n".join(open(os.path.join(CSRC, f)).read() for f in names)


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
#include <random>
static inline uint32_t brev32(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) if (v >> i & 1) r |= 1u << (31 - i); return r; }
%s
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
int main() { if (test_imap<8>() || test_imap<16>()) return 1; printf("ok\n"); return 0; }
'''


def test_device_bit_helpers_against_brute_force(tmp_path):
    text = _device_sources()
    parts = [_between(text, "__device__ __forceinline__ uint64_t spread32(", "template <int NW>\n__device__ __forceinline__ void window_to_iwindow(")]
    code = MAIN % _hostify("\n".join(parts))
    assert "imask_word" in code and "im_clean" in code
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


PACKED_MAIN = r'''
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
static inline uint32_t brev32(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) if (v >> i & 1) r |= 1u << (31 - i); return r; }
static inline uint64_t __brevll(uint64_t v) { uint64_t r = 0; for (int i = 0; i < 64; i++) if (v >> i & 1) r |= 1ULL << (63 - i); return r; }
%s
template <int W> static int test()
{
    std::mt19937_64 rng(3);
    for (int trial = 0; trial < 20000; trial++) {
        const int len = 1 + rng() %% (32 * W);
        std::vector<int> b(len);
        for (auto &x : b) x = rng() & 3;
        uint64_t f[W] = {0}, r[W];
        for (int j = 0; j < len; j++) f[j / 32] |= (uint64_t)b[j] << (62 - 2 * (j %% 32));
        revcomp2<W>(f, len, r);
        for (int j = 0; j < 32 * W; j++) {
            const int got = (r[j / 32] >> (62 - 2 * (j %% 32))) & 3, want = j < len ? 3 - b[len - 1 - j] : 0;
            if (got != want) { printf("revcomp2<%%d> len %%d base %%d: %%d, want %%d\n", W, len, j, got, want); return 1; }
        }
        // the same read as packed words, with rubbish behind its last base and behind its last word
        std::vector<uint32_t> Wd((len + 15) / 16 + 2, 0xdeadbeefu);
        for (int w = 0; w < (len + 15) / 16; w++) {
            uint32_t v = 0;
            for (int k = 0; k < 16 && 16 * w + k < len; k++) v |= (uint32_t)b[16 * w + k] << (30 - 2 * k);
            if (16 * w + 16 > len && (len %% 16)) v |= (uint32_t)rng() & (0xFFFFFFFFu >> (2 * (len %% 16)));
            Wd[w] = v;
        }
        for (int w = 0; w < 2 * W; w++)
            for (int rc = 0; rc < 2; rc++) {
                const uint32_t got = packed_word16(Wd.data(), len, w, rc != 0);
                uint32_t want = 0;
                for (int k = 0; k < 16; k++) { const int j = 16 * w + k; if (j < len) want |= (uint32_t)(rc ? 3 - b[len - 1 - j] : b[j]) << (30 - 2 * k); }
                if (got != want) { printf("packed_word16 len %%d word %%d rc %%d: %%08x, want %%08x\n", len, w, rc, got, want); return 1; }
            }
    }
    return 0;
}
int main() { if (test<4>() || test<8>()) return 1; printf("ok\n"); return 0; }
'''


def test_packed_read_helpers_against_brute_force(tmp_path):
    """revcomp2 (reverse complement of a 2 bit/base row) and packed_word16 (a 16-base word of a packed read or of its reverse
    complement), as they stand in the device sources, against per-base loops"""
    text = _device_sources()
    code = _between(text, "__device__ __forceinline__ uint32_t rev2_32(", "// exceptions of a packed batch, one lane each.")
    src, exe = str(tmp_path / "packed.cpp"), str(tmp_path / "packed")
    open(src, "w").write(PACKED_MAIN % _hostify(code))
    subprocess.check_call(helpers.cxx() + ["-o", exe, src])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
