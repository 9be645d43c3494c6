"""The suffix-ordered window array that holds PART of the suffix array (DevIndex::swmap): whatever it covers, the results are those of
the other index layouts and of the oracle.  The genomes here are made of repeat families whose runs of equal suffixes are shorter than,
within and far beyond what the coverage rule takes whole, so that core intervals lie inside covered stretches, straddle their ends,
reach past the covered head of a long run, or miss the array altogether; byte budgets cut the coverage off at arbitrary blocks."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import helpers  # noqa: E402

pytestmark = pytest.mark.gpu

FIELDS = ("chrom_id", "match_loci", "match_len", "low_hit_instances", "rslt", "nar", "strand", "low_mm", "nxt_low_mm", "num_hits", "mismatches")


def _bk():
    import biokanga_amd
    return biokanga_amd


def _family_genome(seed, n_genome, read_len, n_reads, max_e):
    """-> (seq with two sequences, entries, reads): families of 70 .. 2500 copies (runs the rule covers whole), one of 7000 copies
    (covered for its first entries only), copies with a few substitutions each so that longer cores select parts of the runs"""
    bk = _bk()
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, n_genome, dtype=np.uint8)
    spots = []
    for copies, flen, div in ((70, 220, 0.02), (90, 260, 0.01), (130, 220, 0.03), (200, 300, 0.02), (400, 220, 0.01), (900, 240, 0.02), (2500, 200, 0.01),
                              (66, 200, 0.0), (64, 200, 0.0), (7000, 130, 0.004)):
        fam = rng.integers(0, 4, flen, dtype=np.uint8)
        for p in rng.integers(0, n_genome - flen - 1, copies):
            cp = fam.copy()
            mut = rng.random(flen) < div
            cp[mut] = (cp[mut] + rng.integers(1, 4, int(mut.sum()))) % 4
            g[p:p + flen] = cp
            spots.append((int(p), flen))
    cut = n_genome // 2
    seq = np.concatenate([g[:cut], [7], g[cut:], [7]]).astype(np.uint8)
    ents = np.zeros(2, dtype=bk.ENTRY_DTYPE)
    ents[0] = (1, cut, 0, cut - 1, b"s1", b"")
    ents[1] = (2, n_genome - cut, cut + 1, n_genome, b"s2", b"")
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    reads = np.zeros((n_reads, read_len), dtype=np.uint8)
    for i in range(n_reads):
        if i % 3 and spots:                              # two reads in three start in or next to a family copy
            p, flen = spots[int(rng.integers(0, len(spots)))]
            st = int(np.clip(p + rng.integers(-read_len // 2, flen), 0, n_genome - read_len - 1))
        else:
            st = int(rng.integers(0, n_genome - read_len - 1))
        r = g[st:st + read_len].copy()
        for q in rng.choice(read_len, int(rng.integers(0, max_e + 1)), replace=False):
            r[q] = (r[q] + rng.integers(1, 4)) % 4
        if rng.integers(0, 2):
            r = comp[r[::-1]]
        reads[i] = r
    return seq, ents, reads


def _index(tmp_path, seq, ents, name):
    import torch
    bk = _bk()
    n = len(seq)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    sa = d_sa.cpu().numpy().view(np.uint32)
    path = str(tmp_path / (name + ".sfx"))
    helpers.write_sfx(path, name, [("s1", int(ents[0]["seq_len"])), ("s2", int(ents[1]["seq_len"]))], seq, sa)
    return path


def _same(got, exp, what):
    for f in FIELDS:
        bad = np.flatnonzero(got[f] != exp[f])
        assert bad.size == 0, f"{what}: field {f} differs for {bad.size} reads, first {bad[:5]}: {got[f][bad[:5]]} vs {exp[f][bad[:5]]}"


@pytest.mark.parametrize("read_len,max_subs", [(100, 3), (150, 5), (75, 2)])
def test_partial_window_array_changes_no_result(tmp_path, read_len, max_subs):
    bk = _bk()
    seq, ents, reads = _family_genome(500 + read_len, 700000, read_len, 24000, max_subs + 1)
    path = _index(tmp_path, seq, ents, f"fam{read_len}")
    nreads = len(reads)
    bases = reads.reshape(-1)
    offs = np.arange(nreads, dtype=np.uint64) * read_len
    lens = np.full(nreads, read_len, dtype=np.uint32)
    o = helpers.OracleSfx(path)
    exp, octr = o.align(bases, offs, lens, helpers.make_params(max_subs=max_subs), nthreads=8)
    o.close()
    covered = {}
    with bk.Aligner(path, bk.AlignParams(max_subs=max_subs)) as al:
        for name, knobs in (("none", [("use_swin", 0)]),
                            ("partial", [("use_swin", 2)]),
                            ("every suffix", [("use_swin", 3)]),
                            ("partial, 3 KB", [("use_swin", 0), ("swin_budget_kb", 3), ("use_swin", 2)]),
                            ("partial, 40 KB", [("use_swin", 0), ("swin_budget_kb", 40), ("use_swin", 2)]),
                            ("partial, 700 KB", [("use_swin", 0), ("swin_budget_kb", 700), ("use_swin", 2)]),
                            ("partial, 700 KB, every read through the wave kernel", [("heavy_thresh", 0)]),
                            ("partial, 5 MB, short intervals through the wave kernel", [("use_swin", 0), ("swin_budget_kb", 5000), ("use_swin", 2), ("heavy_thresh", 3)]),
                            ("partial again", [("use_swin", 0), ("swin_budget_kb", 0), ("use_swin", 2), ("heavy_thresh", 64)])):
            for kv in knobs:
                al.tune(*kv)
            al.counters(reset=True)
            got = al.align(bases, offs, lens)
            ctr = al.counters()
            _same(got, exp, name)
            assert (ctr["n_search"], ctr["n_cand"], ctr["n_lcm_calls"]) == (octr.n_search, octr.n_cand, octr.n_lcm_calls), name
            covered[name] = al.tune("swin_covered_ppm", 0) if al.tune("swin_resident", 0) else 0
    assert covered["none"] == 0 and covered["every suffix"] == 1_000_000
    # the rule covers the families' runs and little else; the budgets cut it down
    assert 0 < covered["partial, 3 KB"] < covered["partial, 40 KB"] < covered["partial, 700 KB"] <= covered["partial"] < 600_000
    assert covered["partial again"] == covered["partial"]
    assert np.count_nonzero(exp["nar"] == 1) > nreads // 4


@pytest.mark.parametrize("read_len,max_subs", [(100, 3), (150, 5)])
def test_partial_window_array_with_5_byte_elements(tmp_path, read_len, max_subs):
    """an index of 5-byte suffix array elements has no inverse suffix array: its wave kernel keeps the reference's set of seen keys
    (`SfxArrayV2.cpp:5932`) and, when the array is asked for ("use_swin" 2: the policy's 1 leaves such an index without), takes the windows
    of covered intervals from it - the same records as without, and as the oracle's"""
    bk = _bk()
    seq, ents, reads = _family_genome(900 + read_len, 600000, read_len, 16000, max_subs + 1)
    n = len(seq)
    import torch
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_sa = torch.empty(n * 5, dtype=torch.uint8, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 5, 0)
    path = str(tmp_path / "fam5.sfx")
    sa5 = d_sa.cpu().numpy().reshape(n, 5)
    sa = sa5[:, 0].astype(np.uint64) | (sa5[:, 1].astype(np.uint64) << 8) | (sa5[:, 2].astype(np.uint64) << 16) | (sa5[:, 3].astype(np.uint64) << 24) | (sa5[:, 4].astype(np.uint64) << 32)
    helpers.write_sfx(path, "fam5", [("s1", int(ents[0]["seq_len"])), ("s2", int(ents[1]["seq_len"]))], seq, sa, el_size=5)
    nreads = len(reads)
    bases = reads.reshape(-1)
    offs = np.arange(nreads, dtype=np.uint64) * read_len
    lens = np.full(nreads, read_len, dtype=np.uint32)
    o = helpers.OracleSfx(path)
    exp, octr = o.align(bases, offs, lens, helpers.make_params(max_subs=max_subs), nthreads=8)
    o.close()
    covered = {}
    with bk.Aligner(path, bk.AlignParams(max_subs=max_subs)) as al:
        assert al.lib.bk_sfx_el_size(al.h) == 5
        for name, knobs in (("policy", [("use_swin", 1)]),
                            ("none", [("use_swin", 0)]),
                            ("asked for", [("use_swin", 2)]),
                            ("asked for, every read through the wave kernel", [("heavy_thresh", 0)]),
                            ("asked for, 40 KB", [("use_swin", 0), ("swin_budget_kb", 40), ("use_swin", 2), ("heavy_thresh", 3)]),
                            ("asked for, unverified small buckets off", [("use_swin", 0), ("swin_budget_kb", 0), ("use_swin", 2), ("lazy_search", 0), ("heavy_thresh", 64)])):
            for kv in knobs:
                al.tune(*kv)
            al.counters(reset=True)
            got = al.align(bases, offs, lens)
            ctr = al.counters()
            _same(got, exp, name)
            assert (ctr["n_search"], ctr["n_cand"], ctr["n_lcm_calls"]) == (octr.n_search, octr.n_cand, octr.n_lcm_calls), name
            covered[name] = al.tune("swin_covered_ppm", 0) if al.tune("swin_resident", 0) else 0
    assert covered["policy"] == 0 and covered["none"] == 0
    assert 0 < covered["asked for, 40 KB"] < covered["asked for"] < 600_000
    assert np.count_nonzero(exp["nar"] == 1) > nreads // 4


def test_partial_window_array_follows_the_batch(tmp_path):
    """a batch whose reads are searched with other core lengths than the array was made for gets an array of its own - once: after that
    the array is kept whatever the next batch's read length (coverage never changes a result; batches of variable-length reads would
    otherwise rebuild 25 GB whenever their longest read crosses a core length), and every batch still gives the oracle's records"""
    bk = _bk()
    seq, ents, reads = _family_genome(77, 500000, 100, 12000, 4)
    path = _index(tmp_path, seq, ents, "fam_mixed")
    nreads = len(reads)
    o = helpers.OracleSfx(path)
    with bk.Aligner(path, bk.AlignParams(max_subs=3)) as al:
        seen = []
        for L in (100, 60, 100, 44):
            bases = np.ascontiguousarray(reads[:, :L]).reshape(-1)
            offs = np.arange(nreads, dtype=np.uint64) * L
            lens = np.full(nreads, L, dtype=np.uint32)
            exp, octr = o.align(bases, offs, lens, helpers.make_params(max_subs=3), nthreads=8)
            al.counters(reset=True)
            got = al.align(bases, offs, lens)
            ctr = al.counters()
            _same(got, exp, f"reads of {L} bases")
            assert (ctr["n_search"], ctr["n_cand"]) == (octr.n_search, octr.n_cand)
            assert al.tune("swin_resident", 0) == 1
            seen.append(al.tune("swin_core_lens", 0) & 0xff)
        assert seen == [25, 20, 20, 20]                  # (the last phase's core length of the first batch, then of the second: made again once)
    o.close()


@pytest.mark.parametrize("slices", [1, 3, 7])
def test_window_array_made_behind_the_suffix_arrays_upload(tmp_path, slices):
    """BK_CTX_WINDOW_ARRAY_EAGER: the partial array is made range by range behind the slices of the suffix array's upload (a run of equal
    suffixes that crosses a slice's end is cut there) - it is in place before the first batch, and the results are the oracle's"""
    bk = _bk()
    seq, ents, reads = _family_genome(900 + slices, 600000, 100, 16000, 4)
    path = _index(tmp_path, seq, ents, f"eager{slices}")
    nreads = len(reads)
    bases = reads.reshape(-1)
    offs = np.arange(nreads, dtype=np.uint64) * 100
    lens = np.full(nreads, 100, dtype=np.uint32)
    o = helpers.OracleSfx(path)
    exp, octr = o.align(bases, offs, lens, helpers.make_params(max_subs=3), nthreads=8)
    o.close()
    os.environ["BK_TABLE_SLICES"] = str(slices)
    try:
        with bk.Aligner(path, bk.AlignParams(max_subs=3), flags=1) as al:
            assert al.tune("swin_resident", 0) == 1 and al.tune("swin_core_lens", 0) & 0xff == 25
            covered = al.tune("swin_covered_ppm", 0)
            got = al.align(bases, offs, lens)
            ctr = al.counters()
            assert al.tune("swin_covered_ppm", 0) == covered           # (the batch found the array it needs)
            _same(got, exp, f"{slices} slices")
            assert (ctr["n_search"], ctr["n_cand"], ctr["n_lcm_calls"]) == (octr.n_search, octr.n_cand, octr.n_lcm_calls)
            # the same index without the flag: the array of one pass holds about as much
            with bk.Aligner(path, bk.AlignParams(max_subs=3)) as al2:
                got2 = al2.align(bases, offs, lens)
                _same(got2, exp, "made by the first batch")
                lazy = al2.tune("swin_covered_ppm", 0)
            # (.. or less: its memory was sized before its coverage was known, 10 bytes per suffix)
            assert 0 < covered <= lazy * 1.05 + 2000 and covered >= min(lazy * 0.8, 180_000)
    finally:
        del os.environ["BK_TABLE_SLICES"]
