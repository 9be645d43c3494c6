"""GPU tests of the suffix sort for 2^32 and more bases (bk_sa_build.hip, build_sa_device_wide: 40-bit planes, rank-sorted array
refined in stretches).  Small inputs are pushed through it with small stretch sizes and must give the array of the 32-bit
path element for element; a 4.4 Gbp genome is sorted for real and checked by properties and by alignment parity."""
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _bk():
    import biokanga_amd
    return biokanga_amd


def _build(seq_np, el_size, chunk=None):
    import torch
    bk = _bk()
    dev = torch.device("cuda", 0)
    n = len(seq_np)
    d_seq = torch.from_numpy(seq_np).to(dev)
    d_sa = torch.zeros(n * el_size, dtype=torch.uint8, device=dev)
    old = os.environ.pop("BK_SA_WIDE_CHUNK", None)
    try:
        if chunk:
            os.environ["BK_SA_WIDE_CHUNK"] = str(chunk)
        bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), el_size, 0)
    finally:
        os.environ.pop("BK_SA_WIDE_CHUNK", None)
        if old is not None:
            os.environ["BK_SA_WIDE_CHUNK"] = old
    raw = d_sa.cpu().numpy()
    if el_size == 4:
        return raw.view("<u4").astype(np.uint64)
    r = raw.reshape(n, 5)
    return r[:, :4].copy().view("<u4").reshape(n).astype(np.uint64) | (r[:, 4].astype(np.uint64) << np.uint64(32))


def _concat_seq(fixture_dir):
    """1 byte/base concatenation (EOS after every sequence) of a golden fixture, read from its reference-built .sfx"""
    import struct
    img = np.fromfile(os.path.join(fixture_dir, "genome.sfx"), dtype=np.uint8)
    blk = struct.unpack_from("<Q", img, 44)[0]
    n = struct.unpack_from("<Q", img, blk + 8)[0]
    return img[blk + 20: blk + 20 + n].copy()


def _tricky_genome():
    """long exact repeats, an N run longer than the doubling depth can tell apart in few rounds, tandem repeats, two sequences"""
    rng = np.random.default_rng(7)
    unit = rng.integers(0, 4, 5000, dtype=np.uint8)
    parts = [rng.integers(0, 4, 20000, dtype=np.uint8), unit, rng.integers(0, 4, 3000, dtype=np.uint8), unit, unit,
             np.full(40000, 4, dtype=np.uint8), rng.integers(0, 4, 1000, dtype=np.uint8), np.tile(np.array([0, 1], dtype=np.uint8), 6000),
             np.zeros(9000, dtype=np.uint8), np.array([7], dtype=np.uint8), unit[:3000], rng.integers(0, 4, 30000, dtype=np.uint8),
             np.full(100, 4, dtype=np.uint8), np.array([7], dtype=np.uint8)]
    return np.concatenate(parts)


@pytest.mark.parametrize("which", ["basic", "repeat", "tricky"])
@pytest.mark.parametrize("chunk", [64, 1000, 30000, 1 << 29])
def test_wide_path_gives_the_32_bit_paths_array(golden_tmp, which, chunk):
    seq = _tricky_genome() if which == "tricky" else _concat_seq(golden_tmp[which])
    ref = _build(seq, 4)
    got5 = _build(seq, 5, chunk)
    assert np.array_equal(got5, ref)
    if chunk == 1000:
        assert np.array_equal(_build(seq, 4, chunk), ref)
        assert np.array_equal(_build(seq, 5), ref)            # 5-byte elements from the 32-bit path


def test_suffix_sort_of_more_than_2_32_bases():
    """4.4 G bases: every position appears once; sampled neighbours are in nibble-lexicographic order (compared up to 4096 bases);
    reads aligned against the array give what the CPU oracle gives with the same array"""
    import torch
    bk = _bk()
    from biokanga_amd import synth
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(4_400_000_000, dev, seed=44)
    n = seq.numel()
    assert n > (1 << 32)
    # 100-mers copied to exactly 2^32 bases further on: the reference keys its set of seen targets by the target start truncated
    # to 32 bits (SfxArrayV2.cpp:5932), so reads from these places see ONE instance where there are two - to be reproduced
    planted = []
    for i in range(400):
        p0 = 1_000_000 + i * 250_003
        w1, w2 = seq[p0:p0 + 100], seq[p0 + (1 << 32):p0 + (1 << 32) + 100]
        if int(w1.max()) < 4 and int(w2.max()) < 4:
            seq[p0 + (1 << 32):p0 + (1 << 32) + 100] = w1
            planted.append(p0)
    assert len(planted) > 100
    # and 600-mers, for reads beyond the register kernels' 256 bases (the general search / extend / hash-set kernels)
    planted_long = []
    for i in range(200):
        p0 = 1_125_000 + i * 250_003
        w1, w2 = seq[p0:p0 + 600], seq[p0 + (1 << 32):p0 + (1 << 32) + 600]
        if int(w1.max()) < 4 and int(w2.max()) < 4:
            seq[p0 + (1 << 32):p0 + (1 << 32) + 600] = w1
            planted_long.append(p0)
    assert len(planted_long) > 50
    d_sa = torch.zeros(n * 5, dtype=torch.uint8, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, d_sa.data_ptr(), 5, 0)
    v = d_sa.view(n, 5)
    sa = torch.zeros(n, dtype=torch.int64, device=dev)
    for k in range(5):
        sa |= v[:, k].to(torch.int64) << (8 * k)
    assert int(sa.min()) == 0 and int(sa.max()) == n - 1
    seen = torch.zeros(n, dtype=torch.uint8, device=dev)
    seen[sa] = 1
    assert int(seen.sum(dtype=torch.int64)) == n                  # a permutation of 0..n-1
    del seen
    # neighbours in order: the reference's comparator looks at (base & 0x0f), past the end sorts first
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    j = torch.randint(0, n - 1, (1_000_000,), device=dev, generator=g)
    a, b = sa[j], sa[j + 1]
    undecided = torch.ones_like(a, dtype=torch.bool)
    ar = torch.arange(64, device=dev)
    for depth in range(0, 4096, 64):
        ia, ib = a[undecided, None] + depth + ar, b[undecided, None] + depth + ar
        ca = torch.where(ia < n, (seq[ia.clamp(max=n - 1)] & 15).to(torch.int16) + 1, torch.zeros((), dtype=torch.int16, device=dev))
        cb = torch.where(ib < n, (seq[ib.clamp(max=n - 1)] & 15).to(torch.int16) + 1, torch.zeros((), dtype=torch.int16, device=dev))
        diff = ca != cb
        anyd = diff.any(dim=1)
        first = diff.to(torch.int8).argmax(dim=1)
        rows = torch.nonzero(anyd).squeeze(1)
        assert bool((ca[rows, first[rows]] < cb[rows, first[rows]]).all())
        idx = torch.nonzero(undecided).squeeze(1)
        undecided[idx[anyd]] = False
        if not bool(undecided.any()):
            break
    assert int(undecided.sum()) < 20000                            # what is left agrees on 4096 bases (long repeats / N runs)
    del sa, j, a, b
    torch.cuda.empty_cache()
    # alignment parity on a sample, same array on both sides
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    nreads = 100_000
    rb, ro, rl, _ = synth.make_reads(seq, seq_lens, nreads, 100, dev, seed=9, max_subs=3)
    # + reads from the planted 100-mers with 0..3 substitutions (all AlignReads phases, k_flat and k_wave)
    rng = np.random.default_rng(3)
    extra = []
    for p0 in planted:
        w = seq[p0:p0 + 100].cpu().numpy().copy()
        for q in rng.choice(100, size=int(rng.integers(0, 4)), replace=False):
            w[q] = (w[q] + 1 + rng.integers(0, 3)) & 3
        extra.append(w if rng.integers(0, 2) else (3 - w[::-1]).astype(np.uint8))
    bases = np.concatenate([rb.cpu().numpy()] + extra)
    nreads += len(extra)
    offs = np.arange(nreads, dtype=np.uint64) * 100
    lens = np.full(nreads, 100, dtype=np.uint32)
    with bk.Aligner(None, bk.AlignParams(max_subs=3), d_seq=seq.data_ptr(), concat_len=n, d_sa=d_sa.data_ptr(), el_size=5, entries=ent) as al:
        got = al.align(bases, offs, lens)
        ctr = al.counters()
    ora = helpers.OracleSfx(seq=seq.cpu().numpy(), sa=d_sa.cpu().numpy(), el_size=5, entries=entries)
    exp, octr = ora.align(bases, offs, lens, helpers.make_params(max_subs=3), nthreads=16)
    ora.close()
    for f in ("chrom_id", "match_loci", "match_len", "low_hit_instances", "rslt", "nar", "strand", "low_mm", "nxt_low_mm", "num_hits", "mismatches"):
        bad = np.nonzero(got[f] != exp[f])[0]
        assert len(bad) == 0, (f, len(bad), [(int(i), int(i) - (nreads - len(extra)), got[int(i)], exp[int(i)]) for i in bad[:6]])
    assert (ctr["n_search"], ctr["n_cand"]) == (octr.n_search, octr.n_cand)
    assert int((got["nar"] == 1).sum()) > nreads // 2
    # the planted reads: two true instances, one seen - accepted as unique
    tail = got[-len(extra):]
    assert int((tail["nar"] == 1).sum()) > len(extra) * 3 // 4
    # the planted reads on their own, every read through the hash-set wave kernel (heavy_thresh 0)
    pl_bases = np.concatenate(extra)
    pl_offs = np.arange(len(extra), dtype=np.uint64) * 100
    pl_lens = np.full(len(extra), 100, dtype=np.uint32)
    # reads of 300 and 600 bases from the planted 600-mers (0..3 substitutions, either strand): beyond the register kernels
    long_reads = []
    for p0 in planted_long:
        for ln, o in ((300, 150), (600, 0)):
            w = seq[p0 + o:p0 + o + ln].cpu().numpy().copy()
            for q in rng.choice(ln, size=int(rng.integers(0, 4)), replace=False):
                w[q] = (w[q] + 1 + rng.integers(0, 3)) & 3
            long_reads.append(w if rng.integers(0, 2) else (3 - w[::-1]).astype(np.uint8))
    lg_lens = np.array([len(w) for w in long_reads], dtype=np.uint32)
    lg_offs = np.concatenate([[0], np.cumsum(lg_lens[:-1], dtype=np.uint64)]).astype(np.uint64)
    lg_bases = np.concatenate(long_reads)
    ora = helpers.OracleSfx(seq=seq.cpu().numpy(), sa=d_sa.cpu().numpy(), el_size=5, entries=entries)
    p3 = helpers.make_params(max_subs=3)
    lexp, lctr = ora.align(lg_bases, lg_offs, lg_lens, p3, nthreads=16)
    ora.close()
    fields = ("chrom_id", "match_loci", "match_len", "low_hit_instances", "rslt", "nar", "strand", "low_mm", "nxt_low_mm", "num_hits", "mismatches")
    with bk.Aligner(None, bk.AlignParams(max_subs=3), d_seq=seq.data_ptr(), concat_len=n, d_sa=d_sa.data_ptr(), el_size=5, entries=ent) as al:
        al.tune("heavy_thresh", 0)
        got2 = al.align(pl_bases, pl_offs, pl_lens)
        for f in fields:
            assert np.array_equal(got2[f], tail[f]), ("heavy_thresh=0", f)
        al.tune("heavy_thresh", 64)
        for knobs in ((), (("use_wave", 0),)):
            for kv in knobs:
                al.tune(*kv)
            al.counters(reset=True)
            lgot = al.align(lg_bases, lg_offs, lg_lens)
            c2 = al.counters()
            for f in fields:
                bad = np.nonzero(lgot[f] != lexp[f])[0]
                assert len(bad) == 0, (knobs, f, len(bad), [(int(i), lgot[int(i)], lexp[int(i)]) for i in bad[:4]])
            assert (c2["n_search"], c2["n_cand"]) == (lctr.n_search, lctr.n_cand)
        assert int((lgot["nar"] == 1).sum()) > len(long_reads) * 3 // 4           # two true instances, one seen
    # paired ends on the 5-byte index (config 5's shape: 2 x 150 bp, -s5 -U3 -d200 -D400), a sample against the oracle
    pb, po, pl = synth.make_pairs(seq, seq_lens, 20_000, 150, dev, seed=6)
    pbases, poffs, plens = pb.cpu().numpy(), po.cpu().numpy().astype(np.uint64), pl.cpu().numpy().astype(np.uint32)
    pe = bk.PEParams(3, 200, 400, False)
    with bk.Aligner(None, bk.AlignParams(max_subs=5), d_seq=seq.data_ptr(), concat_len=n, d_sa=d_sa.data_ptr(), el_size=5, entries=ent) as al:
        pgot = al.pair(pbases, poffs, plens, al.align(pbases, poffs, plens), pe)
    ora = helpers.OracleSfx(seq=seq.cpu().numpy(), sa=d_sa.cpu().numpy(), el_size=5, entries=entries)
    p5 = helpers.make_params(max_subs=5)
    pexp, _ = ora.align(pbases, poffs, plens, p5, nthreads=16)
    helpers.oracle_process_pe(ora, p5, 3, 200, 400, False, pbases, poffs, plens, pexp)
    ora.close()
    for f in ("chrom_id", "match_loci", "match_len", "low_hit_instances", "nar", "strand", "low_mm", "nxt_low_mm", "num_hits", "mismatches"):
        assert np.array_equal(pgot[f], pexp[f]), f
    assert np.array_equal(pgot["flags"] & 0x80, pexp["flags"] & 0x80)
