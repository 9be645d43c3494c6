"""The sampled levels behind the second-level keys (biokanga_amd/csrc/bk_device.h: k2s_start; bk_dev_k2.h: k2_count_range, k2_bounds) and the k-mer
table's map of a bucket's first five key bits (ktab2_absent; bk_index.hip: k_make_ktab2), restated line by line in Python and checked against
a plain search: every interval of a sorted key array, every mask, keys of the N kind and all-ones keys included.  CPU only - the kernels
themselves are checked against the oracle by the -m gpu tests."""
import numpy as np
import pytest

K2_LEVELS = 7
ABOVE = 0xFFFFFFFF


def k2s_pad(w):
    return (w + 15) & ~15


def k2s_count(n, j):
    return (n + (1 << (4 * j)) - 1) >> (4 * j)


def k2s_start(n, j):
    o = k2s_pad(n) + 16
    for i in range(1, j):
        o += k2s_pad(k2s_count(n, i)) + 16
    return o


def build_levels(k2):
    """k_build_k2_levels: the array with its levels and padding behind it"""
    n = len(k2)
    out = np.full(k2s_start(n, K2_LEVELS + 1), ABOVE, dtype=np.uint64)
    out[:n] = k2
    for j in range(1, K2_LEVELS + 1):
        st, cnt = k2s_start(n, j), k2s_count(n, j)
        for g in range(cnt):
            src = ((g + 1) << (4 * j)) - 1
            out[st + g] = k2[src if src < n else n - 1]
    return out


def count_range(L, a, b, m, q2, lines):
    n_lt = n_le = 0
    base = a & ~15
    while base < b:
        key = L[base:base + 16]
        assert len(key) == 16, "a line load runs off the allocation"
        lines[0] += 1
        lo = a - base if a > base else 0
        hi = b - base if b - base < 16 else 16
        for j in range(16):
            inside = lo <= j < hi
            km = ABOVE if int(key[j]) == ABOVE else int(key[j]) & m
            n_lt += 1 if inside and km < q2 else 0
            n_le += 1 if inside and km <= q2 else 0
        if n_le < base + hi - a:
            break
        base += 16
    return n_lt, n_le


def k2_bounds(arr, n, first, cnt, m, q2):
    lines = [0]
    l1, h1, l2, h2 = first, first + cnt, first, first + cnt
    ktop = (cnt.bit_length() - 1) >> 2 if cnt >= 16 else 0
    ktop = min(ktop, K2_LEVELS)
    for j in range(ktop, -1, -1):
        sh = 4 * j
        L = arr[k2s_start(n, j):] if j else arr
        a1, b1, a2, b2 = l1 >> sh, h1 >> sh, l2 >> sh, h2 >> sh
        le = 0
        if a1 < b1:
            lt, le = count_range(L, a1, b1, m, q2, lines)
            g = a1 + lt
            if lt:
                l1 = g << sh
            if g < b1:
                h1 = ((g + 1) << sh) - 1
        if a2 < b2:
            if a2 != a1 or b2 != b1:
                _, le = count_range(L, a2, b2, m, q2, lines)
            g = a2 + le
            if le:
                l2 = g << sh
            if g < b2:
                h2 = ((g + 1) << sh) - 1
    return l1, l2, lines[0]


def k2_mask(rem2):
    L = min(rem2, 15)
    return 0 if L <= 0 else (0xFFFFFFFF << (32 - 2 * L)) & 0xFFFFFFFF


def plain_bounds(k2, first, cnt, m, q2):
    lb = ub = 0
    for i in range(first, first + cnt):
        km = ABOVE if int(k2[i]) == ABOVE else int(k2[i]) & m
        lb += km < q2
        ub += km <= q2
    return first + lb, first + ub


def make_bucketed_keys(rng, n):
    """keys sorted inside buckets of random sizes, as the second-level keys are inside k-mer buckets: few distinct prefixes (long runs of equal
    masked keys), some keys of the N kind (low bits 1, filled with ones behind the N), all-ones keys at a bucket's end"""
    k2 = np.zeros(n, dtype=np.uint64)
    starts = [0]
    while starts[-1] < n:
        starts.append(min(n, starts[-1] + int(rng.choice([1, 2, 3, 5, 16, 17, 40, 300, 5000]))))
    for s, e in zip(starts[:-1], starts[1:]):
        keys = []
        for _ in range(e - s):
            r = rng.integers(0, 20)
            if r == 0:
                keys.append(ABOVE)
            else:
                code = int(rng.integers(0, 1 << 30))
                if rng.integers(0, 3):
                    code &= ~((1 << int(rng.choice([0, 8, 20, 26]))) - 1)          # few distinct values: runs of equal keys
                key = (code << 2) & 0xFFFFFFFC
                if r == 1:                                                            # N kind: ones behind the N's place
                    j = int(rng.integers(0, 15))
                    key = ((key | (0xFFFFFFFF >> (2 * j))) & 0xFFFFFFFC) | 1
                keys.append(key)
        keys.sort(key=lambda v: (v == ABOVE, v))
        k2[s:e] = keys
    return k2, starts


@pytest.mark.parametrize("n,seed", [(37, 1), (256, 2), (4097, 3), (70000, 4)])
def test_bounds_through_the_levels_equal_a_plain_count(n, seed):
    rng = np.random.default_rng(seed)
    k2, starts = make_bucketed_keys(rng, n)
    arr = build_levels(k2)
    assert len(arr) == k2s_start(n, K2_LEVELS + 1)
    worst = 0
    cases = 0
    for s, e in zip(starts[:-1], starts[1:]):
        if cases > 400:
            break
        for rem2 in (0, 1, 2, 3, 7, 15, 40):
            m = k2_mask(rem2)
            # probes: keys of the bucket (present), neighbours of keys, random ones
            probes = [int(k2[int(rng.integers(s, e))]) for _ in range(3)] + [int(rng.integers(0, 1 << 32)) for _ in range(2)]
            for p in probes:
                if p == ABOVE:
                    continue
                q2 = (p & 0xFFFFFFFC) & m
                exp = plain_bounds(k2, s, e - s, m, q2)
                lb, ub, lines = k2_bounds(arr, n, s, e - s, m, q2)
                assert (lb, ub) == exp, (n, s, e, rem2, hex(q2))
                worst = max(worst, lines)
                cases += 1
    # a line per level and bound, two where the first level's samples straddle a line: never the log2 of a halving search
    assert worst <= 2 * (2 + 5)


def ktab2_absent(bitmap, m, q2):
    m5 = m >> 27
    lo5 = (q2 >> 27) & m5
    hi5 = lo5 | (~m5 & 31)
    upto = 0xFFFFFFFF if hi5 == 31 else (1 << (hi5 + 1)) - 1
    return (bitmap & upto & ~((1 << lo5) - 1)) == 0


@pytest.mark.parametrize("seed", [11, 12])
def test_the_bucket_map_never_hides_a_match(seed):
    rng = np.random.default_rng(seed)
    k2, starts = make_bucketed_keys(rng, 6000)
    for s, e in zip(starts[:-1], starts[1:]):
        if e - s < 2 or e - s > 64:
            continue
        bitmap = 0
        for i in range(s, e):
            if int(k2[i]) != ABOVE:
                bitmap |= 1 << (int(k2[i]) >> 27)
        for rem2 in (0, 1, 2, 3, 9, 15):
            m = k2_mask(rem2)
            for p in [int(k2[int(rng.integers(s, e))]) for _ in range(4)] + [int(rng.integers(0, 1 << 32)) for _ in range(8)]:
                if p == ABOVE:
                    continue
                q2 = (p & 0xFFFFFFFC) & m
                lb, ub = plain_bounds(k2, s, e - s, m, q2)
                if ktab2_absent(bitmap, m, q2):
                    assert lb == ub, (s, e, rem2, hex(q2), hex(bitmap))
