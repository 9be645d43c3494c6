"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
(a) the golden outputs of the real reference and (b) the CPU oracle, bit-exact on every bk_hit field."""
import os
import struct

import numpy as np
import pytest

import helpers
from test_oracle_golden import RUNS, MIN_LEN, MAX_LEN, check_hits_against_sam, chrom_names_from_hdr, expected_from_sam

pytestmark = pytest.mark.gpu

FIELDS = ["chrom_id", "match_loci", "match_len", "low_hit_instances", "rslt", "nar", "strand", "low_mm",
          "nxt_low_mm", "num_hits", "mismatches"]


def _bk():
    import biokanga_amd
    return biokanga_amd


def assert_hits_equal(got, exp, names=None):
    for f in FIELDS:
        if not np.array_equal(got[f], exp[f]):
            bad = np.nonzero(got[f] != exp[f])[0]
            i = int(bad[0])
            raise AssertionError(f"field {f}: {len(bad)} reads differ; first idx {i} "
                                 f"({names[i] if names else ''}): got {got[i]} exp {exp[i]}")


def load_fixture(golden_tmp, fixture, tag="s3"):
    d = golden_tmp[fixture]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    keep = helpers.filter_reads_by_len(names, bases, offs, lens, MIN_LEN.get(tag, 50), MAX_LEN.get(tag, 500))
    return d, names, bases, offs, lens, keep


@pytest.mark.parametrize("fixture,tag", [(f, t) for f in RUNS for t in RUNS[f]])
def test_hip_matches_reference_and_oracle(golden_tmp, fixture, tag):
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, fixture, tag)
    kw = RUNS[fixture][tag]
    with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(**kw)) as al:
        hits = al.align(bases, offs[keep], lens[keep])
        ctr = al.counters()
    hdr, recs = expected_from_sam(fixture, tag)
    check_hits_against_sam(names, lens, hits, recs, chrom_names_from_hdr(hdr), keep)
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    exp, octr = sfx.align(bases, offs[keep], lens[keep], helpers.make_params(**kw))
    assert_hits_equal(hits, exp, [names[i] for i in keep])
    # the counters behind the algorithmic-bytes figure are the reference algorithm's own counts
    assert ctr["n_search"] == octr.n_search
    assert ctr["n_cand"] == octr.n_cand
    assert ctr["n_lcm_calls"] == octr.n_lcm_calls
    sfx.close()


@pytest.mark.parametrize("fixture", ["basic", "repeat", "lengths"])
@pytest.mark.parametrize("knob", [("heavy_thresh", 3), ("heavy_thresh", 0), ("heavy_thresh", 7), ("heavy_thresh", 100), ("use_ktab", 0), ("kmer_bits", 4),
                                  ("kmer_bits", 12), ("chunk_reads", 333), ("use_wave", 0), ("lazy_search", 0), (("kmer_bits", 9), ("lazy_search", 0)), ("use_k2", 0),
                                  ("use_tgt2", 0), ("use_tgt2", 1), (("use_tgt2", 0), ("heavy_thresh", 0)), ("sort_lists", 0), ("sort_lists", 3), ("sort_lists", 1), ("sort_lists", 6),
                                  ("use_isa", 0), (("use_isa", 0), ("heavy_thresh", 0)), ("use_swin", 2), (("use_swin", 2), ("heavy_thresh", 0)), (("use_swin", 2), ("lazy_search", 0), ("heavy_thresh", 3)),
                                  ("use_swin", 3), (("use_swin", 3), ("heavy_thresh", 0)), ("use_swin", 0),
                                  (("kmer_bits", 6), ("lazy_search", 0)), (("kmer_bits", 9), ("use_wave", 0)),
                                  (("use_k2", 0), ("lazy_search", 0)), ("use_k3", 0), ("use_k3", 1), (("use_k3", 0), ("lazy_search", 0)), (("use_k3", 1), ("lazy_search", 0)), (("use_k3", 0), ("kmer_bits", 6)), (("kmer_bits", 6), ("heavy_thresh", 0)), ("use_ktab2", 0), ("use_ktab2", 2), (("use_ktab2", 2), ("lazy_search", 0)), (("use_ktab2", 2), ("heavy_thresh", 0)), (("use_ktab2", 2), ("use_iv32", 0)), (("use_ktab2", 0), ("kmer_bits", 9)), ("use_iv32", 0), (("use_iv32", 0), ("lazy_search", 0)),
                                  ("async_phases", 0), (("async_phases", 0), ("chunk_reads", 333)), (("async_phases", 0), ("sort_lists", 0)),
                                  ("ktab_wide", 1), ("ktab_wide", 2), (("ktab_wide", 1), ("lazy_search", 0)), (("ktab_wide", 1), ("kmer_bits", 9)), (("ktab_wide", 1), ("use_swin", 2))])
def test_paths_agree(golden_tmp, fixture, knob):
    """wave-per-read kernel == lane-per-read kernel; table-accelerated search == plain bisection;
    chunking does not matter."""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, fixture, "s3L" if fixture == "lengths" else "s3")
    with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(max_subs=3)) as al:
        ref = al.align(bases, offs[keep], lens[keep])
        c0 = al.counters(reset=True)
        for kv in (knob if isinstance(knob[0], tuple) else (knob,)):
            al.tune(*kv)
        got = al.align(bases, offs[keep], lens[keep])
        c1 = al.counters()
    assert_hits_equal(got, ref, [names[i] for i in keep])
    for k in ("n_search", "n_cand", "n_lcm_calls"):
        assert c0[k] == c1[k], (k, c0, c1)
    if knob == ("heavy_thresh", 0):
        # every call with at least one non-empty core interval went through the wave-per-read kernel
        assert c1["n_heavy"] > c0["n_heavy"] and c1["n_heavy"] >= np.count_nonzero(ref["rslt"] != 0)


def test_wide_kmer_table_is_packed(golden_tmp):
    """the k-mer table of an index beyond 2^32 suffixes (64-bit bucket starts) is kept as 32-bit offsets from a 64-bit start per 2^16
    codes; "ktab_wide" makes a small index build it: 1 packed, 2 plain 64-bit, 0 back to what the index needs"""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "repeat", "s3")
    with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(max_subs=3)) as al:
        ref = al.align(bases, offs[keep], lens[keep])
        assert al.tune("ktab_packed", 0) == 0
        for mode, packed in ((1, 1), (2, 0), (0, 0)):
            al.tune("ktab_wide", mode)
            assert al.tune("ktab_packed", 0) == packed
            assert_hits_equal(al.align(bases, offs[keep], lens[keep]), ref, [names[i] for i in keep])


@pytest.mark.parametrize("fixture", ["basic", "repeat", "lengths"])
@pytest.mark.parametrize("knob", [(), ("chunk_reads", 333), ("use_tgt2", 0), ("use_k2", 0), ("heavy_thresh", 0), ("use_wave", 0)])
def test_packed_batch_equals_byte_batch(golden_tmp, fixture, knob):
    """bk_align_batch_packed (2 bit/base words + 16-bit lengths + the list of bases that are not a,c,g,t) gives the records and the
    counters of bk_align_batch on the same reads: lean and full-row batches, the general path (reads > 256 bases), chunked batches
    (exception read numbers are batch-wide), reads with N, reads refused for their Ns."""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, fixture, "s3L" if fixture == "lengths" else "s3")
    words, lens16, exc = bk.pack_reads(bases, offs[keep], lens[keep])
    assert len(exc) > 0 or fixture != "basic"
    with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(max_subs=3)) as al:
        if knob:
            al.tune(*knob)
        ref = al.align(bases, offs[keep], lens[keep])
        c0 = al.counters(reset=True)
        got = al.align_packed(words, lens16, exc)
        c1 = al.counters()
    assert_hits_equal(got, ref, [names[i] for i in keep])
    for k in ("n_search", "n_cand", "n_lcm_calls"):
        assert c0[k] == c1[k], (k, c0, c1)


@pytest.mark.parametrize("el_size", [4, 5])
@pytest.mark.parametrize("top", [150, 192, 256, 300, 320, 321, 450, 512])
def test_reads_of_129_to_512_bases(tmp_path, top, el_size):
    """batches whose longest read has 129 .. 256 bases (the 16-word register-window kernels; 2 x 150 is the common case), 257 .. 320
    bases (the 20-word ones; 2 x 250 / 2 x 300) or 321 .. 512 bases (the 32-word ones): reads of that range and a few short ones
    against the oracle, every kernel family of that width, both index element sizes, bytes and packed input"""
    import torch
    bk = _bk()
    seq, ents, reads = _synth_case(1200 + top + el_size, 300000, 9000 if top <= 320 else 5000, top, 5, dup_len=top + 40)
    n = len(seq)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    sa = d_sa.cpu().numpy().view(np.uint32)
    path = str(tmp_path / "nw12.sfx")
    helpers.write_sfx(path, "nw12", [("s1", int(ents[0]["seq_len"])), ("s2", int(ents[1]["seq_len"]))], seq, sa.astype(np.uint64) if el_size == 5 else sa,
                      el_size=el_size)
    rng = np.random.default_rng(top)
    nreads = len(reads)
    lens = rng.integers(129 if top <= 256 else (257 if top <= 320 else 321), top + 1, nreads).astype(np.uint32)
    lens[rng.integers(0, nreads, 200)] = rng.integers(20, 129, 200)
    lens[0] = top
    offs = np.arange(nreads, dtype=np.uint64) * top
    bases = reads.reshape(-1)
    o = helpers.OracleSfx(path)
    exp, octr = o.align(bases, offs, lens, helpers.make_params(max_subs=4), nthreads=8)
    o.close()
    words, lens16, exc = bk.pack_reads(bases, offs, lens)
    with bk.Aligner(path, bk.AlignParams(max_subs=4)) as al:
        for knobs in ([], [("heavy_thresh", 0)], [("heavy_thresh", 100)], [("use_isa", 0)], [("use_tgt2", 0)], [("use_swin", 0)]):
            for k, v in knobs:
                al.tune(k, v)
            al.counters(reset=True)
            got = al.align(bases, offs, lens)
            ctr = al.counters(reset=True)
            assert_hits_equal(got, exp)
            assert (ctr["n_search"], ctr["n_cand"]) == (octr.n_search, octr.n_cand), knobs
            assert_hits_equal(al.align_packed(words, lens16, exc), exp)
            for k, v in knobs:
                al.tune(k, 1 if k.startswith("use_") else 64)


@pytest.mark.parametrize("read_len", [100, 150])
def test_packed_batch_of_many_reads(read_len):
    """more reads than one launch's first round of blocks (the rows of the later blocks once came out wrong: a code-generation
    hazard in the reverse complement of the 2-bit rows, profiles/NOTES.md), both row widths, paired ends through the stream"""
    import torch
    bk = _bk()
    from biokanga_amd import synth
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(8_000_000, dev, seed=23, n_seqs=3, repeat_frac=0.5)
    n = seq.numel()
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), 4, 0)
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    pb, po, pl = synth.make_pairs(seq, seq_lens, 100_000, read_len, dev, seed=6, max_subs=3)
    bases, offs, lens = pb.cpu().numpy(), po.cpu().numpy().astype(np.uint64), pl.cpu().numpy().astype(np.uint32)
    words, lens16, exc = bk.pack_reads(bases, None, lens)
    pe = bk.PEParams(3, 200, 400, False)
    with bk.Aligner(None, bk.AlignParams(max_subs=3), d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=4, entries=ent) as al:
        ref = al.align(bases, offs, lens)
        assert int((ref["nar"] == 1).sum()) > len(lens) // 2
        got = al.align_packed(words, lens16, exc)
        assert_hits_equal(got, ref)
        paired = al.pair(bases, offs, lens, ref.copy(), pe)
        out = np.zeros(len(lens), bk.HIT_DTYPE)
        with bk.Stream(al, len(lens), len(bases), depth=2, pe=pe) as st:
            st.wait(st.submit_packed(words, lens16, exc, out))
        for f in FIELDS + ["flags"]:
            assert np.array_equal(out[f], paired[f]), f


def test_packed_batch_is_checked(golden_tmp):
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "basic", "s3")
    words, lens16, exc = bk.pack_reads(bases, offs[keep], lens[keep])
    with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(max_subs=3)) as al:
        ref = al.align_packed(words, lens16, exc)
        for what in ("words", "read", "pos", "code", "order", "len"):
            w2, l2, e2 = words, lens16.copy(), exc.copy()
            if what == "words":
                w2 = words[:-1]
            elif what == "read":
                e2["read"][-1] = len(l2)
            elif what == "pos":
                e2["pos"][0] = l2[e2["read"][0]]
            elif what == "code":
                e2["code"][0] = 2
            elif what == "order":
                e2[[0, 1]] = e2[[1, 0]]
            elif what == "len":
                l2[3] = 2001
            with pytest.raises(bk.BkError) as e:
                al.align_packed(w2, l2, e2)
            assert e.value.rc == -100, what
        assert_hits_equal(al.align_packed(words, lens16, exc), ref)            # the context stays usable
        assert len(al.align_packed(np.zeros(0, np.uint32), np.zeros(0, np.uint16), np.zeros(0, bk.NBASE_DTYPE))) == 0


def test_seq_counts_and_empty(golden_tmp):
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "basic")
    with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(max_subs=3)) as al:
        hits = al.align(bases, offs[keep], lens[keep])
        counts = al.seq_counts(reset=True)
        ents = al.entries()
        assert [e["name"].decode() for e in ents] == ["chrA", "chrB"]
        for k, e in enumerate(ents):
            assert counts[k] == np.count_nonzero((hits["nar"] == 1) & (hits["chrom_id"] == e["entry_id"]))
        assert al.seq_counts().sum() == 0
        # empty batch is a no-op
        out = al.align(np.zeros(0, np.uint8), np.zeros(0, np.uint64), np.zeros(0, np.uint32))
        assert len(out) == 0
        # reads longer than the 2000 bp limit are refused, not truncated
        with pytest.raises(bk.BkError):
            al.align(np.zeros(2100, np.uint8), np.zeros(1, np.uint64), np.array([2100], np.uint32))


def test_device_resident_batch_and_sa_builder(golden_tmp):
    """bk_align_batch_device over torch-owned HBM buffers; bk_build_sa_device reproduces the suffix
    array the reference's qsort wrote into the golden .sfx."""
    import torch
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "basic")
    raw = open(os.path.join(d, "genome.sfx"), "rb").read()
    blk_ofs = struct.unpack_from("<Q", raw, 44)[0]
    n = struct.unpack_from("<Q", raw, blk_ofs + 8)[0]
    seq = np.frombuffer(raw, dtype=np.uint8, count=n, offset=blk_ofs + 20)
    sa = np.frombuffer(raw, dtype="<u4", count=n, offset=blk_ofs + 20 + n)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq.copy()).to(dev)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    torch.cuda.synchronize()
    got_sa = d_sa.cpu().numpy().view(np.uint32)
    # identical except among suffixes that are equal up to and including an EOS: the reference's
    # comparator then runs off the end of the sequence into the array being sorted, so their order is
    # arbitrary there (and irrelevant: no search ever compares past an EOS)
    for j in np.nonzero(got_sa != sa)[0]:
        a, b = int(got_sa[j]), int(sa[j])
        l = 0
        while a + l < n and b + l < n and seq[a + l] == seq[b + l]:
            l += 1
        assert 7 in seq[a:a + l], (j, a, b, l)
    assert np.array_equal(np.sort(got_sa), np.arange(n, dtype=np.uint32))

    with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(max_subs=3)) as al:
        ref = al.align(bases, offs[keep], lens[keep])
        ents = al.entries()
    # context from the device-resident image + device-resident reads
    with bk.Aligner(None, bk.AlignParams(max_subs=3), d_seq=d_seq.data_ptr(), concat_len=n, d_sa=d_sa.data_ptr(),
                    el_size=4, entries=ents) as al2:
        t_bases = torch.from_numpy(np.ascontiguousarray(bases)).to(dev)
        t_offs = torch.from_numpy(offs[keep].astype(np.int64)).to(dev)
        t_lens = torch.from_numpy(lens[keep].astype(np.int32)).to(dev)
        t_out = torch.zeros(len(keep) * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        al2.align_device(t_bases.data_ptr(), t_offs.data_ptr(), t_lens.data_ptr(), len(keep), t_out.data_ptr())
        got = t_out.cpu().numpy().view(bk.HIT_DTYPE)
    assert_hits_equal(got, ref)


def _synth_case(seed, n_genome, n_reads, read_len, max_e, dup_len=0):
    rng = np.random.default_rng(seed)
    # two sequences with a planted repeat family so multi-loci / truncation paths trigger
    g = rng.integers(0, 4, n_genome, dtype=np.uint8)
    fam = rng.integers(0, 4, 60, dtype=np.uint8)
    for p in rng.integers(0, n_genome - 100, 300):
        g[p:p + 60] = fam
    if dup_len:
        # segments longer than a read present in 2..9 places, half of the copies with one substitution:
        # reads from them align to several loci with the same (or a next-best) number of mismatches
        for _ in range(120):
            src = int(rng.integers(0, n_genome - dup_len))
            seg = g[src:src + dup_len].copy()
            for _c in range(int(rng.integers(1, 9))):
                dst = int(rng.integers(0, n_genome - dup_len))
                cp = seg.copy()
                if rng.integers(0, 2):
                    q = int(rng.integers(0, dup_len))
                    cp[q] = (cp[q] + 1) % 4
                g[dst:dst + dup_len] = cp
    cut = n_genome // 2
    seq = np.concatenate([g[:cut], [7], g[cut:], [7]]).astype(np.uint8)
    ents = np.zeros(2, dtype=_bk().ENTRY_DTYPE)
    ents[0] = (1, cut, 0, cut - 1, b"s1", b"")
    ents[1] = (2, n_genome - cut, cut + 1, n_genome, b"s2", b"")
    starts = rng.integers(0, n_genome - read_len, n_reads)
    reads = np.zeros((n_reads, read_len), dtype=np.uint8)
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    for i, st in enumerate(starts):
        r = g[st:st + read_len].copy()
        e = int(rng.integers(0, max_e + 1))
        for q in rng.choice(read_len, e, replace=False):
            r[q] = (r[q] + rng.integers(1, 4)) % 4
        if rng.integers(0, 2):
            r = comp[r[::-1]]
        if rng.integers(0, 40) == 0:
            r[int(rng.integers(0, read_len))] = 4
        reads[i] = r
    return seq, ents, reads


@pytest.mark.parametrize("read_len,max_subs", [(100, 3), (150, 5), (64, 10)])
def test_synthetic_parity_vs_oracle(tmp_path, read_len, max_subs):
    """index built on the GPU -> .sfx written in the reference's format -> oracle and HIP path agree"""
    import torch
    bk = _bk()
    seq, ents, reads = _synth_case(11 + read_len, 300000, 20000, read_len, 6)
    n = len(seq)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    sa = d_sa.cpu().numpy().view(np.uint32)
    path = str(tmp_path / "synth.sfx")
    helpers.write_sfx(path, "synth", [("s1", int(ents[0]["seq_len"])), ("s2", int(ents[1]["seq_len"]))], seq, sa)
    nreads = len(reads)
    bases = reads.reshape(-1)
    offs = (np.arange(nreads, dtype=np.uint64) * read_len)
    lens = np.full(nreads, read_len, dtype=np.uint32)
    with bk.Aligner(path, bk.AlignParams(max_subs=max_subs)) as al:
        got = al.align(bases, offs, lens)
        ctr = al.counters()
    o = helpers.OracleSfx(path)
    exp, octr = o.align(bases, offs, lens, helpers.make_params(max_subs=max_subs), nthreads=8)
    assert_hits_equal(got, exp)
    assert (ctr["n_search"], ctr["n_cand"]) == (octr.n_search, octr.n_cand)
    assert np.count_nonzero(got["nar"] == 1) > nreads // 3
    o.close()


def test_repeatable(tmp_path):
    """the same batch aligned repeatedly gives bit-identical results (guards against races between
    the search / extend / wave-per-read kernels; an earlier build showed ~1e-4 flaky interval counts)"""
    import torch
    bk = _bk()
    seq, ents, reads = _synth_case(75, 300000, 20000, 64, 6)
    n = len(seq)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    nreads = len(reads)
    bases = reads.reshape(-1)
    offs = (np.arange(nreads, dtype=np.uint64) * 64)
    lens = np.full(nreads, 64, dtype=np.uint32)
    with bk.Aligner(None, bk.AlignParams(max_subs=10), d_seq=d_seq.data_ptr(), concat_len=n, d_sa=d_sa.data_ptr(),
                    el_size=4, entries=ents) as al:
        ref = al.align(bases, offs, lens)
        for wave in (1, 0):
            al.tune("use_wave", wave)
            for thresh in (100, 8, 99, 10, 0):
                al.tune("heavy_thresh", thresh)
                for _ in range(3):
                    assert_hits_equal(al.align(bases, offs, lens), ref)


from test_oracle_pe import PE_RUNS, ALL_PE_RUNS, pe_cfg, pe_inputs, check_pe_hits_against_sam


@pytest.mark.parametrize("fixture,tag", ALL_PE_RUNS)
def test_pe_matches_reference_and_oracle(golden_tmp, tmp_path, fixture, tag):
    """K4: paired-end association + orphan recovery on the GPU vs the reference's PE SAM and the oracle (2 x 100 bp, and the
    2 x 150 bp geometry of C3: MaxTotMM 8, 16-mer cores, 9 cores per strand)"""
    bk = _bk()
    cfg = pe_cfg(fixture, tag)
    names, bases, offs, lens = pe_inputs(tmp_path, fixture)
    sfx_path = os.path.join(golden_tmp["basic"], "genome.sfx")
    with bk.Aligner(sfx_path, bk.AlignParams(max_subs=cfg["s"])) as al:
        hits = al.align(bases, offs, lens)
        hits = al.pair(bases, offs, lens, hits, bk.PEParams(cfg["pe"], cfg["d"], cfg["D"], cfg.get("E", False)))
    check_pe_hits_against_sam(names, hits, tag, ["chrA", "chrB"], fixture)
    o = helpers.OracleSfx(sfx_path)
    p = helpers.make_params(max_subs=cfg["s"])
    exp, _ = o.align(bases, offs, lens, p, nthreads=8)
    helpers.oracle_process_pe(o, p, cfg["pe"], cfg["d"], cfg["D"], cfg.get("E", False), bases, offs, lens, exp)
    o.close()
    assert_hits_equal(hits, exp, names)
    assert np.array_equal(hits["flags"] & 0x80, exp["flags"] & 0x80)


@pytest.mark.parametrize("slices", [1, 3, 7])
def test_tables_made_behind_the_suffix_arrays_slices(golden_tmp, slices):
    """bk_ctx_create sends a 4-byte suffix array in slices and makes k-mer table, second-level keys and inverse suffix array of slice i
    while slice i + 1 crosses PCIe (eight slices of a 3.1 Gbp index; BK_TABLE_SLICES forces several on a small one): the tables must be
    those of one pass - same hits, same counters, as the context made from the image already in HBM"""
    import torch
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "repeat", "s3")
    sfx_path = os.path.join(d, "genome.sfx")
    os.environ["BK_TABLE_SLICES"] = str(slices)
    try:
        with bk.Aligner(sfx_path, bk.AlignParams(max_subs=3)) as al:
            got = al.align(bases, offs[keep], lens[keep])
            ctr = al.counters()
    finally:
        del os.environ["BK_TABLE_SLICES"]
    sfx = helpers.OracleSfx(sfx_path)
    exp, octr = sfx.align(bases, offs[keep], lens[keep], helpers.make_params(max_subs=3))
    sfx.close()
    assert_hits_equal(got, exp, [names[i] for i in keep])
    assert ctr["n_search"] == octr.n_search and ctr["n_cand"] == octr.n_cand


from test_oracle_pe import PE_FILT_RUNS, filt_by_chroms


@pytest.mark.parametrize("tag", sorted(PE_FILT_RUNS))
def test_pe_with_chromosome_filters_matches_reference_and_oracle(golden_tmp, tmp_path, tag):
    """-Z / -z with -U: the pair rules ask AcceptThisChromID (bk_ctx_set_chrom_filter) - vs the oracle, record by record, and - after the
    host's FiltByChroms - vs the reference's SAM"""
    bk = _bk()
    cfg = PE_FILT_RUNS[tag]
    names, bases, offs, lens = pe_inputs(tmp_path, "pe")
    sfx_path = os.path.join(golden_tmp["basic"], "genome.sfx")
    chroms = ["chrA", "chrB"]
    accept = helpers.chrom_accept_table(chroms, exclude=cfg.get("Z", ()), include=cfg.get("z", ()))
    with bk.Aligner(sfx_path, bk.AlignParams(max_subs=5)) as al:
        se = al.align(bases, offs, lens)
        al.set_chrom_filter(accept)
        hits = al.pair(bases, offs, lens, se.copy(), bk.PEParams(cfg["pe"], 200, 400, False))
        al.set_chrom_filter(None)
        plain = al.pair(bases, offs, lens, se.copy(), bk.PEParams(cfg["pe"], 200, 400, False))
    o = helpers.OracleSfx(sfx_path)
    p = helpers.make_params(max_subs=5)
    exp, _ = o.align(bases, offs, lens, p, nthreads=8)
    exp_plain = exp.copy()
    helpers.oracle_process_pe(o, p, cfg["pe"], 200, 400, False, bases, offs, lens, exp, accept=accept)
    helpers.oracle_process_pe(o, p, cfg["pe"], 200, 400, False, bases, offs, lens, exp_plain)
    o.close()
    assert_hits_equal(hits, exp, names)
    assert np.array_equal(hits["flags"] & 0x80, exp["flags"] & 0x80)
    assert_hits_equal(plain, exp_plain, names)                   # the table is gone with set_chrom_filter(None)
    assert not np.array_equal(hits["nar"], plain["nar"])
    filt_by_chroms(hits, chroms, cfg.get("Z", ()), cfg.get("z", ()))
    check_pe_hits_against_sam(names, hits, tag, chroms, "pe")


from test_oracle_pe import PECHIM_RUNS, check_pechim_against_sam

SEG2_FIELDS = ("match_loci", "match_len", "read_ofs", "mismatches", "flags", "score")


def assert_seg2_equal(seg, eseg, names=None):
    for f in SEG2_FIELDS:
        if not np.array_equal(seg[f], eseg[f]):
            i = int(np.nonzero(seg[f] != eseg[f])[0][0])
            raise AssertionError(f"seg2 field {f} differs at read {i} ({names[i] if names is not None else ''}): got {seg[i]} exp {eseg[i]}")


@pytest.mark.parametrize("tag", sorted(PECHIM_RUNS))
def test_pe_with_chimeric_trimming_matches_reference_and_oracle(golden_tmp, tmp_path, tag):
    """-c together with -U: the pair rules on the trimmed loci and AlignPairedRead placing the partner end-trimmed, against the
    reference's SAM (flags, POS, CIGAR with its soft clips, PNEXT, TLEN, NAR) and, field by field, against the oracle; the
    device-resident entry point gives the same records"""
    import torch
    bk = _bk()
    cfg = PECHIM_RUNS[tag]
    names, bases, offs, lens = pe_inputs(tmp_path, "pechim")
    sfx_path = os.path.join(golden_tmp["chimeric"], "genome.sfx")
    pe = bk.PEParams(cfg["pe"], cfg["d"], cfg["D"], False)
    with bk.Aligner(sfx_path, bk.AlignParams(max_subs=cfg["s"], min_chimeric_len=cfg["c"])) as al:
        se = al.align(bases, offs, lens)
        se_seg = al.batch_seg2()
        if cfg["c"]:
            assert len(se_seg) == len(lens)
            with pytest.raises(bk.BkError):                 # the trims are part of the input
                al.pair(bases, offs, lens, se.copy(), pe)
            hits, seg = al.pair(bases, offs, lens, se.copy(), pe, seg2=se_seg.copy())
            dev = torch.device("cuda", 0)
            d_b = torch.from_numpy(np.ascontiguousarray(bases)).to(dev)
            d_o = torch.from_numpy(offs.astype(np.int64)).to(dev)
            d_l = torch.from_numpy(lens.astype(np.int32)).to(dev)
            d_h = torch.from_numpy(se.view(np.uint8).copy()).to(dev)
            d_s = torch.from_numpy(se_seg.view(np.uint8).copy()).to(dev)
            al.pair_device(d_b.data_ptr(), d_o.data_ptr(), d_l.data_ptr(), len(lens) // 2, d_h.data_ptr(), pe, d_seg2=d_s.data_ptr())
            assert np.array_equal(d_h.cpu().numpy().view(bk.HIT_DTYPE), hits)
            assert np.array_equal(d_s.cpu().numpy().view(bk.SEG2_DTYPE), seg)
        else:
            assert len(se_seg) == 0
            hits = al.pair(bases, offs, lens, se.copy(), pe)
            seg = np.zeros(len(lens), bk.SEG2_DTYPE)
    check_pechim_against_sam(names, hits, seg, tag)
    o = helpers.OracleSfx(sfx_path)
    p = helpers.make_params(max_subs=cfg["s"], min_chimeric_len=cfg["c"])
    exp, eseg = helpers.oracle_align_indel(o, bases, offs, lens, p, nthreads=8)
    helpers.oracle_process_pe(o, p, cfg["pe"], cfg["d"], cfg["D"], False, bases, offs, lens, exp, eseg)
    o.close()
    assert_hits_equal(hits, exp, names)
    assert np.array_equal(hits["flags"] & 0x80, exp["flags"] & 0x80)
    assert_seg2_equal(seg, eseg, names)


@pytest.mark.parametrize("read_len,d,D,subs,pct", [(100, 150, 600, 3, 50), (150, 200, 1600, 5, 60), (150, 200, 900, 3, 75), (600, 700, 1400, 2, 50),
                                                   (600, 650, 2600, 2, 65)])
def test_pe_chimeric_orphan_recovery_on_planted_pairs(tmp_path, read_len, d, D, subs, pct):
    """AlignPairedRead with MinChimericLen on pairs built for it: one mate unique, the other inside a duplicated block (so the SE pass
    leaves it with several loci) and with foreign bases at one or both ends in most pairs; window sizes either side of the 1000
    bases that switch between the scan of every offset and the core lookups (SfxArrayV2.cpp:8332), reads either side of the
    512 bases that switch the device's mismatch map - records, flags and trims against the oracle"""
    import torch
    bk = _bk()
    rng = np.random.default_rng(read_len * 7 + D)
    ng = 240000
    g = rng.integers(0, 4, ng, dtype=np.uint8)
    blk = 3 * read_len if read_len <= 150 else read_len + 150
    blocks = []
    for k in range(40):                                   # blocks present twice, well apart
        a = 2000 + k * 2500
        b = ng // 2 + 3000 + k * 2500
        g[b:b + blk] = g[a:a + blk]
        blocks.append(a)
    cut = ng // 2
    seq = np.concatenate([g[:cut], [7], g[cut:], [7]]).astype(np.uint8)
    n = len(seq)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    path = str(tmp_path / "planted.sfx")
    helpers.write_sfx(path, "planted", [("s1", cut), ("s2", ng - cut)], seq, d_sa.cpu().numpy().view(np.uint32))
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    reads = []
    for i in range(400):
        a = blocks[int(rng.integers(0, len(blocks)))]
        inb = a + int(rng.integers(0, blk - read_len))                    # mate inside the block
        ins = int(rng.integers(d - 40, min(D + 60, 1900)))
        fwd_first = bool(rng.integers(0, 2))
        p1 = inb - (ins - read_len) if fwd_first else inb + (ins - read_len)      # the unique mate, up- or downstream of the block
        if p1 < 0 or p1 + read_len >= cut:
            continue
        u = g[p1:p1 + read_len].copy()
        m = g[inb:inb + read_len].copy()
        for q in rng.choice(read_len, int(rng.integers(0, subs + 1)), replace=False):
            m[q] = (m[q] + 1 + rng.integers(0, 3)) % 4
        if rng.integers(0, 4):                                            # foreign ends
            k5 = int(rng.integers(5, read_len * (100 - pct) // 100 + 8)) if rng.integers(0, 3) else 0
            k3 = int(rng.integers(5, read_len * (100 - pct) // 100 + 8)) if (k5 == 0 or rng.integers(0, 3) == 0) else 0
            m[:k5] = rng.integers(0, 4, k5)
            if k3:
                m[read_len - k3:] = rng.integers(0, 4, k3)
        left, right = (u, m) if fwd_first else (m, u)
        right = comp[right[::-1]]
        pair = [left, right] if rng.integers(0, 2) else [right, left]
        reads += pair
    reads = np.array(reads, dtype=np.uint8)
    nreads = len(reads)
    bases = reads.reshape(-1)
    offs = np.arange(nreads, dtype=np.uint64) * read_len
    lens = np.full(nreads, read_len, dtype=np.uint32)
    o = helpers.OracleSfx(path)
    prm = helpers.make_params(max_subs=subs, min_chimeric_len=pct)
    exp, eseg = helpers.oracle_align_indel(o, bases, offs, lens, prm, nthreads=8)
    se_exp = exp.copy()
    helpers.oracle_process_pe(o, prm, 3, d, D, False, bases, offs, lens, exp, eseg)
    o.close()
    rec = np.nonzero((se_exp["match_loci"] != exp["match_loci"]) & (exp["nar"] == 1) & ((exp["flags"] & 0x80) != 0))[0]
    assert len(rec) > 30 and np.count_nonzero(eseg["flags"][rec] & 8) > 10, (len(rec), np.count_nonzero(eseg["flags"][rec] & 8))
    with bk.Aligner(path, bk.AlignParams(max_subs=subs, min_chimeric_len=pct)) as al:
        se = al.align(bases, offs, lens)
        hits, seg = al.pair(bases, offs, lens, se, bk.PEParams(3, d, D, False), seg2=al.batch_seg2())
    assert_hits_equal(hits, exp)
    assert np.array_equal(hits["flags"] & 0x80, exp["flags"] & 0x80)
    assert_seg2_equal(seg, eseg)


@pytest.mark.parametrize("read_len", [100, 300])
def test_five_byte_suffix_elements(tmp_path, read_len):
    """.sfx with 5-byte suffix elements (what the reference writes above 4 Gbp), forced onto a small
    genome: the WIDE kernels (sa_lo + sa_hi) agree with the oracle; 300 bp reads also exercise the
    long-read path (k_extend / k_heavy, no register window)."""
    import torch
    bk = _bk()
    seq, ents, reads = _synth_case(500 + read_len, 300000, 6000, read_len, 6)
    n = len(seq)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    sa = d_sa.cpu().numpy().view(np.uint32)
    path = str(tmp_path / "wide.sfx")
    helpers.write_sfx(path, "wide", [("s1", int(ents[0]["seq_len"])), ("s2", int(ents[1]["seq_len"]))], seq, sa.astype(np.uint64), el_size=5)
    nreads = len(reads)
    bases = reads.reshape(-1)
    offs = (np.arange(nreads, dtype=np.uint64) * read_len)
    lens = np.full(nreads, read_len, dtype=np.uint32)
    o = helpers.OracleSfx(path)
    exp, octr = o.align(bases, offs, lens, helpers.make_params(max_subs=5), nthreads=8)
    o.close()
    with bk.Aligner(path, bk.AlignParams(max_subs=5)) as al:
        assert al.lib.bk_sfx_el_size(al.h) == 5
        for thresh in (64, 0):
            al.tune("heavy_thresh", thresh)
            al.counters(reset=True)
            got = al.align(bases, offs, lens)
            ctr = al.counters()
            assert_hits_equal(got, exp)
            assert (ctr["n_search"], ctr["n_cand"]) == (octr.n_search, octr.n_cand)


def test_pe_device_resident_entry_point(golden_tmp, tmp_path):
    """bk_pair_batch_device on buffers in HBM == bk_pair_batch on host buffers"""
    import torch
    bk = _bk()
    cfg = PE_RUNS["U3"]
    names, bases, offs, lens = pe_inputs(tmp_path)
    sfx_path = os.path.join(golden_tmp["basic"], "genome.sfx")
    pe = bk.PEParams(cfg["pe"], cfg["d"], cfg["D"], cfg.get("E", False))
    with bk.Aligner(sfx_path, bk.AlignParams(max_subs=cfg["s"])) as al:
        hits = al.align(bases, offs, lens)
        exp = al.pair(bases, offs, lens, hits.copy(), pe)
        dev = torch.device("cuda", 0)
        d_b = torch.from_numpy(np.ascontiguousarray(bases)).to(dev)
        d_o = torch.from_numpy(offs.astype(np.int64)).to(dev)
        d_l = torch.from_numpy(lens.astype(np.int32)).to(dev)
        d_h = torch.zeros(len(lens) * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        al.align_device(d_b.data_ptr(), d_o.data_ptr(), d_l.data_ptr(), len(lens), d_h.data_ptr())
        al.pair_device(d_b.data_ptr(), d_o.data_ptr(), d_l.data_ptr(), len(lens) // 2, d_h.data_ptr(), pe)
        got = d_h.cpu().numpy().view(bk.HIT_DTYPE)
    assert np.array_equal(got, exp)


def test_edge_case_reads_match_oracle(golden_tmp):
    """degenerate reads through the C ABI: length 1..14 (shorter than any core), all-N, all one base, non-ACGT codes,
    overlapping / out-of-order offsets, a single read - every field as the oracle (= the reference algorithm) has it"""
    bk = _bk()
    d = golden_tmp["basic"]
    rng = np.random.default_rng(123)
    genome = helpers.read_fasta_reads(os.path.join(d, "genome.fa"))           # (names, bases, offs, lens) of the two sequences
    gbases, goffs, glens = genome[1], genome[2], genome[3]
    chunks, lens = [], []
    for L in list(range(1, 20)) + [24, 25, 26, 31, 32, 33, 47, 48, 49, 63, 64, 65, 127, 128, 129, 255, 256, 257]:
        for rep in range(4):
            p0 = int(rng.integers(0, int(glens[0]) - L))
            r = gbases[int(goffs[0]) + p0: int(goffs[0]) + p0 + L].copy() & 7
            if rep == 1 and L > 2:
                r[L // 2] = (r[L // 2] + 1) & 3
            if rep == 2:
                r[:] = 4                                                        # all N
            if rep == 3:
                r[:] = rng.integers(0, 4)                                       # homopolymer
            chunks.append(r); lens.append(L)
    chunks.append(np.array([5, 6, 7, 0, 1, 2, 3] * 10, dtype=np.uint8)); lens.append(70)      # codes the loader never produces
    bases = np.concatenate(chunks)
    lens = np.array(lens, dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.uint64)
    # same reads again through overlapping windows of one buffer, in reverse order
    offs2 = np.concatenate([offs, offs[::-1]]); lens2 = np.concatenate([lens, lens[::-1]])
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    for kw in (dict(max_subs=3), dict(max_subs=10), dict(max_subs=0)):
        exp, _ = sfx.align(bases, offs2, lens2, helpers.make_params(**kw))
        with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(**kw)) as al:
            # the longest read of a call picks the kernel family: <= 128, <= 256 bases (register kernels), longer
            for cap in (128, 256, 100000):
                sel = np.nonzero(lens2 <= cap)[0]
                got = al.align(bases, offs2[sel], lens2[sel])
                for f in FIELDS:
                    bad = np.nonzero(got[f] != exp[f][sel])[0]
                    assert len(bad) == 0, (kw, cap, f, bad[:10], lens2[sel][bad[:10]])
            one = al.align(bases, offs2[40:41], lens2[40:41])
            for f in FIELDS:
                assert one[f][0] == exp[f][40]
    sfx.close()


def test_many_sequences(golden_tmp):
    """300 short sequences (> 128: entry table outside LDS; > 64: no lane-per-entry lookup), some of them copies of
    each other, reads that hang over sequence ends: every field as the oracle has it, counts per sequence included"""
    import torch
    bk = _bk()
    rng = np.random.default_rng(314)
    n_seq, L = 300, 1000
    seqs = [rng.integers(0, 4, L).astype(np.uint8) for _ in range(n_seq)]
    for k in range(0, n_seq, 10):                                        # every tenth sequence repeats its neighbour
        seqs[k] = seqs[k + 1].copy()
        seqs[k][rng.integers(0, L, 5)] ^= 1
    concat = np.concatenate([np.concatenate([s, [7]]) for s in seqs]).astype(np.uint8)
    n = len(concat)
    entries = [(i + 1, L, i * (L + 1), i * (L + 1) + L - 1) for i in range(n_seq)]
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(concat).to(dev)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    sa = d_sa.cpu().numpy().view(np.uint32)
    reads, lens = [], []
    for i in range(6000):
        p = int(rng.integers(0, n - 120))                                # may span an EOS
        ln = int(rng.integers(50, 121))
        r = concat[p:p + ln].copy()
        r[r == 7] = rng.integers(0, 4)
        for _ in range(i % 4):
            q = int(rng.integers(0, ln)); r[q] = (r[q] + 1) & 3
        if i % 2:
            r = (3 - r)[::-1]
        reads.append(r); lens.append(ln)
    bases = np.concatenate(reads).astype(np.uint8)
    lens = np.array(lens, dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.uint64)
    ora = helpers.OracleSfx(seq=concat, sa=sa, el_size=4, entries=entries)
    exp, octr = ora.align(bases, offs, lens, helpers.make_params(max_subs=3))
    ora.close()
    ent = np.zeros(n_seq, dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"s{eid}".encode(), b"")
    with bk.Aligner(None, bk.AlignParams(max_subs=3), d_seq=d_seq.data_ptr(), concat_len=n, d_sa=d_sa.data_ptr(), el_size=4, entries=ent) as al:
        got = al.align(bases, offs, lens)
        counts = al.seq_counts()
        al.tune("use_wave", 0)                                           # general-path kernels too
        got2 = al.align(bases, offs, lens)
    for f in FIELDS:
        assert np.array_equal(got[f], exp[f]), f
        assert np.array_equal(got2[f], exp[f]), f
    assert (exp["nar"] == 1).sum() > 3000 and (exp["nar"] == 5).sum() > 100      # accepted and multi-loci (ML) both occur
    for k in range(n_seq):
        assert counts[k] == np.count_nonzero((exp["nar"] == 1) & (exp["chrom_id"] == k + 1))


def _assert_loci_equal(bk, al, nreads, got_hits, exp_hits, exp_offs, exp_loci):
    assert_hits_equal(got_hits, exp_hits)
    offs, loci = al.batch_loci(nreads)
    assert np.array_equal(offs, exp_offs)
    assert len(loci) == len(exp_loci)
    for f in ("chrom_id", "match_loci", "match_len", "strand", "mismatches"):
        if not np.array_equal(loci[f], exp_loci[f]):
            i = int(np.nonzero(loci[f] != exp_loci[f])[0][0])
            r = int(np.searchsorted(offs, i, side="right") - 1)
            raise AssertionError(f"loci field {f} differs at entry {i} (read {r}): got {loci[i]} exp {exp_loci[i]}")


@pytest.mark.parametrize("max_ml,clamp", [(2, 0), (2, 1), (5, 0), (5, 1), (64, 1), (500, 0)])
@pytest.mark.parametrize("fixture", ["basic", "repeat", "lengths", "multi"])
def test_multi_loci_lists_match_oracle(golden_tmp, fixture, max_ml, clamp):
    """MaxHits > 1 (the -R of the multi-loci modes): result records AND the pHits[] lists, in the reference's
    discovery order, against the oracle"""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, fixture, "s3L" if fixture == "lengths" else "s3")
    kw = dict(max_subs=3, max_ml=max_ml, clamp_ml=clamp)
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    exp, eo, el = helpers.oracle_align_multi(sfx, bases, offs[keep], lens[keep], helpers.make_params(**kw))
    sfx.close()
    for knobs in ([], [("use_wave", 0)], [("use_isa", 0)], [("heavy_thresh", 0)], [("chunk_reads", 257)]):
        with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(**kw)) as al:
            for k, v in knobs:
                al.tune(k, v)
            got = al.align(bases, offs[keep], lens[keep])
            _assert_loci_equal(bk, al, len(keep), got, exp, eo, el)


@pytest.mark.parametrize("read_len,max_subs,max_ml,clamp", [(100, 3, 5, 0), (100, 3, 3, 1), (150, 5, 20, 0), (64, 10, 3, 1), (300, 3, 8, 0)])
def test_multi_loci_synthetic(tmp_path, read_len, max_subs, max_ml, clamp):
    import torch
    bk = _bk()
    seq, ents, reads = _synth_case(23 + read_len, 300000, 12000, read_len, 4, dup_len=2 * read_len + 50)
    n = len(seq)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    sa = d_sa.cpu().numpy().view(np.uint32)
    path = str(tmp_path / "synth.sfx")
    helpers.write_sfx(path, "synth", [("s1", int(ents[0]["seq_len"])), ("s2", int(ents[1]["seq_len"]))], seq, sa)
    nreads = len(reads)
    bases = reads.reshape(-1)
    offs = (np.arange(nreads, dtype=np.uint64) * read_len)
    lens = np.full(nreads, read_len, dtype=np.uint32)
    kw = dict(max_subs=max_subs, max_ml=max_ml, clamp_ml=clamp)
    o = helpers.OracleSfx(path)
    exp, eo, el = helpers.oracle_align_multi(o, bases, offs, lens, helpers.make_params(**kw), nthreads=8)
    o.close()
    with bk.Aligner(path, bk.AlignParams(**kw)) as al:
        got = al.align(bases, offs, lens)
        _assert_loci_equal(bk, al, nreads, got, exp, eo, el)
    assert np.count_nonzero(np.diff(eo.astype(np.int64)) > 1) > 20


@pytest.mark.parametrize("max_ml,max_subs", [(2, 3), (5, 3), (5, 1), (64, 5), (500, 3)])
@pytest.mark.parametrize("fixture", ["basic", "repeat", "lengths", "multi"])
def test_best_matches_match_oracle(golden_tmp, fixture, max_ml, max_subs):
    """-N (LocateBestMatches): result records and the loci kept, ordered by mismatches then discovery, against the oracle's
    literal restatement of the reference's insertion list"""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, fixture, "s3L" if fixture == "lengths" else "s3")
    kw = dict(max_subs=max_subs, max_ml=max_ml, best_matches=1)
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    exp, eo, el = helpers.oracle_align_multi(sfx, bases, offs[keep], lens[keep], helpers.make_params(**kw))
    sfx.close()
    for knobs in ([], [("chunk_reads", 300)], [("kmer_bits", 8)]):
        with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(**kw)) as al:
            for k, v in knobs:
                al.tune(k, v)
            got = al.align(bases, offs[keep], lens[keep])
            ctr = al.counters()
            _assert_loci_equal(bk, al, len(keep), got, exp, eo, el)


def test_best_matches_synthetic(tmp_path):
    import torch
    bk = _bk()
    read_len = 100
    seq, ents, reads = _synth_case(77, 300000, 12000, read_len, 4, dup_len=250)
    n = len(seq)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    sa = d_sa.cpu().numpy().view(np.uint32)
    path = str(tmp_path / "synth.sfx")
    helpers.write_sfx(path, "synth", [("s1", int(ents[0]["seq_len"])), ("s2", int(ents[1]["seq_len"]))], seq, sa)
    nreads = len(reads)
    bases = reads.reshape(-1)
    offs = (np.arange(nreads, dtype=np.uint64) * read_len)
    lens = np.full(nreads, read_len, dtype=np.uint32)
    kw = dict(max_subs=4, max_ml=6, best_matches=1)
    o = helpers.OracleSfx(path)
    exp, eo, el = helpers.oracle_align_multi(o, bases, offs, lens, helpers.make_params(**kw), nthreads=8)
    octr = o.align(bases, offs, lens, helpers.make_params(**kw), nthreads=8)[1]
    o.close()
    with bk.Aligner(path, bk.AlignParams(**kw)) as al:
        got = al.align(bases, offs, lens)
        ctr = al.counters()
        _assert_loci_equal(bk, al, nreads, got, exp, eo, el)
    assert (ctr["n_search"], ctr["n_cand"]) == (octr.n_search, octr.n_cand)
    assert np.count_nonzero(np.diff(eo.astype(np.int64)) > 1) > 20


@pytest.mark.parametrize("kw", [dict(max_subs=3, micro_indel_len=10), dict(max_subs=5, micro_indel_len=3), dict(max_subs=3, micro_indel_len=20, align_strand=1),
                                dict(max_subs=3, micro_indel_len=20, align_strand=2), dict(max_subs=1, micro_indel_len=5), dict(max_subs=0, micro_indel_len=8),
                                dict(max_subs=3, splice_junct_len=5000), dict(max_subs=5, splice_junct_len=500), dict(max_subs=3, splice_junct_len=5000, micro_indel_len=5),
                                dict(max_subs=1, splice_junct_len=100000, align_strand=1), dict(max_subs=3, splice_junct_len=25, align_strand=2)])
@pytest.mark.parametrize("fixture", ["indel", "splice", "basic", "lengths"])
def test_micro_indels_match_oracle(golden_tmp, fixture, kw):
    """-a: result records and second segments (LocateInDels) against the oracle, which is pinned on the reference's -a output"""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, fixture, "s3L" if fixture == "lengths" else "s3")
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    exp, eseg = helpers.oracle_align_indel(sfx, bases, offs[keep], lens[keep], helpers.make_params(**kw))
    sfx.close()
    for knobs in ([], [("chunk_reads", 200)], [("use_wave", 0)]):
        with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(**kw)) as al:
            for k, v in knobs:
                al.tune(k, v)
            got = al.align(bases, offs[keep], lens[keep])
            seg = al.batch_seg2()
        assert_hits_equal(got, exp, [names[i] for i in keep])
        assert len(seg) == len(eseg)
        for f in ("match_loci", "match_len", "read_ofs", "mismatches", "flags", "score"):
            if not np.array_equal(seg[f], eseg[f]):
                i = int(np.nonzero(seg[f] != eseg[f])[0][0])
                raise AssertionError(f"seg2 field {f} differs at read {i} ({names[keep[i]]}): got {seg[i]} exp {eseg[i]} hit {got[i]}")
    if fixture == "indel" and kw.get("micro_indel_len"):
        assert np.count_nonzero(eseg["flags"] & 1) > 50 or kw["max_subs"] == 0
    if fixture == "splice" and kw.get("splice_junct_len", 0) >= 500 and kw["max_subs"] >= 3:
        assert np.count_nonzero(eseg["flags"] & 4) > 50


@pytest.mark.parametrize("kw", [dict(max_subs=3, min_chimeric_len=50), dict(max_subs=5, min_chimeric_len=70), dict(max_subs=3, min_chimeric_len=60, min_edit_dist=2),
                                dict(max_subs=3, min_chimeric_len=99, align_strand=1), dict(max_subs=1, min_chimeric_len=50, align_strand=2),
                                dict(max_subs=3, min_chimeric_len=55, micro_indel_len=6, splice_junct_len=3000),
                                dict(max_subs=3, min_chimeric_len=50, micro_indel_len=10), dict(max_subs=3, min_chimeric_len=60, splice_junct_len=5000),
                                dict(max_subs=5, min_chimeric_len=50, micro_indel_len=20, splice_junct_len=100000, min_edit_dist=2)])
@pytest.mark.parametrize("fixture", ["chimeric", "basic", "indel", "splice", "combined"])
def test_chimeric_placements_match_oracle(golden_tmp, fixture, kw):
    """-c: the chimeric LocateCoreMultiples call (AdaptiveTrim per candidate) - result records and the trims carried in bk_seg2 -
    against the oracle, which is pinned on the reference's -c output"""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, fixture, "s3")
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    exp, eseg = helpers.oracle_align_indel(sfx, bases, offs[keep], lens[keep], helpers.make_params(**kw))
    sfx.close()
    for knobs in ([], [("chunk_reads", 150)], [("use_wave", 0)]):
        with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(**kw)) as al:
            for k, v in knobs:
                al.tune(k, v)
            got = al.align(bases, offs[keep], lens[keep])
            seg = al.batch_seg2()
        assert_hits_equal(got, exp, [names[i] for i in keep])
        for f in ("match_loci", "match_len", "read_ofs", "mismatches", "flags", "score"):
            if not np.array_equal(seg[f], eseg[f]):
                i = int(np.nonzero(seg[f] != eseg[f])[0][0])
                raise AssertionError(f"seg2 field {f} differs at read {i} ({names[keep[i]]}): got {seg[i]} exp {eseg[i]} hit {got[i]}")
    if fixture == "chimeric" and kw["min_chimeric_len"] <= 70 and kw["max_subs"] >= 3:
        assert np.count_nonzero(eseg["flags"] & 8) > 100


@pytest.mark.parametrize("kw", [dict(max_subs=3, min_chimeric_len=50, max_ml=5), dict(max_subs=3, min_chimeric_len=55, max_ml=8, min_edit_dist=2),
                                dict(max_subs=3, min_chimeric_len=50, max_ml=3, clamp_ml=1), dict(max_subs=5, min_chimeric_len=70, max_ml=3, clamp_ml=1),
                                dict(max_subs=3, min_chimeric_len=60, max_ml=2, align_strand=1)])
@pytest.mark.parametrize("fixture", ["chimml", "chimeric", "multi"])
def test_chimeric_loci_lists_match_oracle(golden_tmp, fixture, kw):
    """-c together with the multi-loci modes: the chimeric call lists its loci (up to MaxHits of the same trimmed length and mismatches, in
    discovery order), each with its own end trims - result records, lists, trims and the per-read bk_seg2 records against the oracle,
    which is pinned on the reference's -r5 -c runs; one chunk and many"""
    bk = _bk()
    d = golden_tmp[fixture]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    o = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    exp, elo, eloci, etrims, eseg = helpers.oracle_align_multi_chimeric(o, bases, offs, lens, helpers.make_params(**kw), nthreads=8)
    o.close()
    for knobs in ([], [("chunk_reads", 97)], [("use_wave", 0)]):
        with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(**kw)) as al:
            for k, v in knobs:
                al.tune(k, v)
            got = al.align(bases, offs, lens)
            lo, loci = al.batch_loci(len(lens))
            trims = al.batch_loci_trims()
            seg = al.batch_seg2()
        assert_hits_equal(got, exp, names)
        assert np.array_equal(lo, elo)
        for f in ("chrom_id", "match_loci", "match_len", "strand", "mismatches"):
            assert np.array_equal(loci[f], eloci[f]), (f, knobs)
        assert len(trims) == len(loci)
        for f in ("left", "right", "chimeric"):
            if not np.array_equal(trims[f], etrims[f]):
                j = int(np.nonzero(trims[f] != etrims[f])[0][0])
                i = int(np.searchsorted(lo, j, side="right") - 1)
                raise AssertionError(f"trims field {f} differs at locus {j} of read {i} ({names[i]}): got {trims[j]} exp {etrims[j]} locus {loci[j]}")
        assert_seg2_equal(seg, eseg, names)
    if fixture == "chimml" and kw["min_chimeric_len"] <= 55:
        n_multi_chim = sum(1 for i in range(len(names)) if elo[i + 1] - elo[i] > 1 and etrims["chimeric"][int(elo[i])])
        assert n_multi_chim > 30


@pytest.mark.parametrize("kw", [dict(max_subs=3, min_chimeric_len=50, max_ml=5, micro_indel_len=8), dict(max_subs=3, min_chimeric_len=60, max_ml=5, micro_indel_len=5),
                                dict(max_subs=3, min_chimeric_len=55, max_ml=3, clamp_ml=1, micro_indel_len=10, splice_junct_len=200)])
def test_chimeric_loci_lists_with_indels_match_oracle(golden_tmp, kw):
    """-c with a multi-loci mode AND -a / -A: the microInDel / splice junction searches run between the substitution-only phases and the
    chimeric call (SfxArrayV2.cpp:7722-7757) - records, loci lists, trims and two-segment records against the oracle on the reads the
    reference's own runs of the combination are pinned on (tests/golden/chimmlindel, tests/test_gpu_cli.py)"""
    bk = _bk()
    d = golden_tmp["chimmlindel"]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    o = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    exp, elo, eloci, etrims, eseg = helpers.oracle_align_multi_chimeric(o, bases, offs, lens, helpers.make_params(**kw), nthreads=8)
    o.close()
    for knobs in ([], [("chunk_reads", 97)]):
        with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(**kw)) as al:
            for k, v in knobs:
                al.tune(k, v)
            got = al.align(bases, offs, lens)
            lo, loci = al.batch_loci(len(lens))
            trims = al.batch_loci_trims()
            seg = al.batch_seg2()
        assert_hits_equal(got, exp, names)
        assert np.array_equal(lo, elo)
        for f in ("chrom_id", "match_loci", "match_len", "strand", "mismatches"):
            assert np.array_equal(loci[f], eloci[f]), (f, knobs)
        for f in ("left", "right", "chimeric"):
            assert np.array_equal(trims[f], etrims[f]), (f, knobs)
        assert_seg2_equal(seg, eseg, names)
    assert np.count_nonzero(eseg["match_len"]) > 20           # (reads placed as two segments)


def test_chimeric_with_loci_lists_refuses_what_it_cannot_list(golden_tmp):
    bk = _bk()
    sfx = os.path.join(golden_tmp["chimml"], "genome.sfx")
    with pytest.raises(bk.BkError):                            # (the reference refuses it itself: kanga.cpp:712-716)
        bk.Aligner(sfx, bk.AlignParams(max_subs=3, min_chimeric_len=50, max_ml=5, best_matches=1))
    # with microInDels / splice junctions the combination is the reference's own (tests/test_gpu_cli.py, tests/golden/chimmlindel)
    for kw in (dict(min_chimeric_len=50, max_ml=5, micro_indel_len=5), dict(min_chimeric_len=50, max_ml=5, splice_junct_len=1000)):
        bk.Aligner(sfx, bk.AlignParams(max_subs=3, **kw)).close()


@pytest.mark.parametrize("kw", [dict(max_subs=1, min_chimeric_len=50), dict(max_subs=2, min_chimeric_len=70, align_strand=1), dict(max_subs=3, min_chimeric_len=50)])
def test_chimeric_reads_longer_than_512_bases(tmp_path, kw):
    """-c on reads of 520 .. 1900 bases (the device AdaptiveTrim keeps a 2048-base mismatch map for them): a genome piece with a few
    substitutions, flanked on one or both sides by foreign sequence - records and trims against the oracle.  (AdaptiveTrim refuses
    more than 15 allowed mismatches, SfxArrayV2.cpp:5523: at -s3 that is every read beyond 533 bases, at -s1 beyond 1599.)"""
    import torch
    bk = _bk()
    rng = np.random.default_rng(4242)
    seq, ents, _ = _synth_case(99, 400000, 10, 100, 0)
    n = len(seq)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    path = str(tmp_path / "long.sfx")
    helpers.write_sfx(path, "long", [("s1", int(ents[0]["seq_len"])), ("s2", int(ents[1]["seq_len"]))], seq, d_sa.cpu().numpy().view(np.uint32))
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    half = int(ents[0]["seq_len"])
    reads = []
    for i in range(600):
        L = int(rng.integers(520, 1901))
        keepf = float(rng.uniform(0.55, 1.0))                      # part of the read that comes from the genome
        core = max(60, int(L * keepf))
        st = int(rng.integers(0, half - core - 1))
        r = seq[st:st + core].copy()
        for q in rng.choice(core, int(rng.integers(0, 5)), replace=False):
            r[q] = (r[q] + rng.integers(1, 4)) % 4
        left = int(rng.integers(0, L - core + 1)) if rng.integers(0, 3) else 0
        r = np.concatenate([rng.integers(0, 4, left, dtype=np.uint8), r, rng.integers(0, 4, L - core - left, dtype=np.uint8)])
        if rng.integers(0, 2):
            r = comp[r[::-1]]
        reads.append(r.astype(np.uint8))
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
    bases = np.concatenate(reads)
    o = helpers.OracleSfx(path)
    exp, eseg = helpers.oracle_align_indel(o, bases, offs, lens, helpers.make_params(**kw))
    o.close()
    with bk.Aligner(path, bk.AlignParams(**kw)) as al:
        got = al.align(bases, offs, lens)
        seg = al.batch_seg2()
    assert_hits_equal(got, exp)
    for f in ("match_loci", "match_len", "read_ofs", "mismatches", "flags", "score"):
        assert np.array_equal(seg[f], eseg[f]), f
    if kw["max_subs"] < 3:
        assert np.count_nonzero(eseg["flags"] & 8) > (100 if kw["max_subs"] == 1 else 10)       # trimmed placements exist


def _assert_sites_equal(got, gtot, exp, etot, what):
    assert np.array_equal(gtot, etot), (what, gtot, etot)
    assert len(got) == len(exp), (what, len(got), len(exp))
    for f in ("loci", "num_ref", "non_ref", "win_mismatches", "win_matches", "ref_base"):
        if not np.array_equal(got[f], exp[f]):
            i = int(np.nonzero(np.atleast_2d((got[f] != exp[f]).T).any(axis=0))[0][0])
            raise AssertionError(f"{what}: site field {f} differs at {i}: got {got[i]} exp {exp[i]}")


def test_snp_pileup_and_sites_match_oracle_on_fixture(golden_tmp):
    """-p: reads of the snp fixture aligned on the device, piled up (in two calls) and screened per sequence: the site lists and the
    totals the oracle computes from the same alignments (oracle pinned on the reference's SNP rows in test_oracle_snp.py)"""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "snp", "s3")
    with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(max_subs=5)) as al:
        hits = al.align(bases, offs, lens)
        acc = np.nonzero(hits["nar"] == 1)[0]
        alns = np.zeros(len(acc), dtype=bk.SNP_ALN_DTYPE)
        alns["read_idx"] = acc; alns["chrom_id"] = hits["chrom_id"][acc]; alns["loci"] = hits["match_loci"][acc]
        alns["len"] = hits["match_len"][acc]; alns["strand"] = hits["strand"][acc]
        assert len(acc) > 9000
        al.snp_reset()
        h = len(alns) // 2
        al.snp_pileup(bases, offs, lens, alns[:h])
        al.snp_pileup(bases, offs, lens, alns[h:])
        sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
        n_sites = 0
        for min_reads, prop in ((1, 0.001), (5, 0.25), (12, 0.1)):
            for chrom in (1, 2, 3):
                got, gtot = al.snp_sites(chrom, min_reads, prop)
                exp, etot = helpers.oracle_snp_sites(sfx.h, bases, offs, alns, chrom, min_reads, prop)
                _assert_sites_equal(got, gtot, exp, etot, (min_reads, prop, chrom))
                n_sites += len(exp)
        assert n_sites > 3000
        al.snp_reset()                                                   # counts really are cleared
        got, gtot = al.snp_sites(1, 1, 0.001)
        assert len(got) == 0 and not gtot.any()
        import torch                                                     # device-resident entry point: same counts
        dev = torch.device("cuda:0")
        d_b, d_o = torch.from_numpy(bases).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev)
        d_a = torch.from_numpy(alns.view(np.uint8)).to(dev)
        al.snp_pileup_device(d_b.data_ptr(), d_o.data_ptr(), len(lens), d_a.data_ptr(), len(alns))
        for chrom in (1, 2, 3):
            got, gtot = al.snp_sites(chrom, 5, 0.25)
            exp, etot = helpers.oracle_snp_sites(sfx.h, bases, offs, alns, chrom, 5, 0.25)
            _assert_sites_equal(got, gtot, exp, etot, ("device", chrom))
        sfx.close()


def test_snp_pileup_synthetic_edges(tmp_path):
    """sequences shorter than, equal to and just longer than the 51 base background window, alignments on both strands with read
    offsets, N and other non-ACGT read codes, N in the target, alignments that run past the end of their sequence"""
    import torch
    bk = _bk()
    rng = np.random.default_rng(77)
    seq_lens = [30, 50, 51, 52, 53, 76, 77, 78, 100, 101, 102, 103, 400, 1000, 2500]
    seqs = [rng.integers(0, 4, L).astype(np.uint8) for L in seq_lens]
    seqs[12][100:110] = 4
    concat = np.concatenate([np.concatenate([s, [7]]) for s in seqs]).astype(np.uint8)
    n = len(concat)
    starts = np.concatenate([[0], np.cumsum(np.array(seq_lens) + 1)[:-1]])
    entries = [(i + 1, seq_lens[i], int(starts[i]), int(starts[i]) + seq_lens[i] - 1) for i in range(len(seqs))]
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(concat).to(dev)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(d_seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    sa = d_sa.cpu().numpy().view(np.uint32)
    n_reads = 4000
    lens = rng.integers(25, 130, n_reads).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.uint64)
    bases = rng.integers(0, 4, int(lens.sum())).astype(np.uint8)
    bases[rng.integers(0, len(bases), 600)] = 4
    bases[rng.integers(0, len(bases), 60)] = 5
    bases |= (rng.integers(0, 16, len(bases)).astype(np.uint8) << 4)      # quality nibbles must be ignored
    alns = np.zeros(9000, dtype=bk.SNP_ALN_DTYPE)
    for k in range(len(alns)):
        r = int(rng.integers(0, n_reads)); c = int(rng.integers(0, len(seqs))); L = seq_lens[c]
        ofs = int(rng.integers(0, 10)) if k % 3 == 0 else 0
        ln = max(int(min(int(lens[r]) - ofs, L)), 1)
        loci = int(rng.integers(0, L - ln + 1))
        if k % 50 == 0 and L >= 60:
            loci = L - ln + int(rng.integers(1, min(20, ln)))             # hangs over the end
        alns[k] = (r, c + 1, loci, ln, ofs, ord("+-"[k & 1]), 0)
    # and agreement with the target where a read really comes from it, so that matches dominate somewhere
    for k in range(0, len(alns), 2):
        a = alns[k]
        c, L = int(a["chrom_id"]) - 1, seq_lens[int(a["chrom_id"]) - 1]
        if a["loci"] + a["len"] > L or k % 50 == 0:
            continue
        lo, ln = int(a["loci"]), int(a["len"])
        seg = seqs[c][lo:lo + ln].copy()
        if chr(a["strand"]) == "-":
            seg = np.where(seg < 4, 3 - seg, seg)[::-1]
        o = int(offs[a["read_idx"]]) + int(a["read_ofs"])
        bases[o:o + ln] = seg
    ent = np.zeros(len(seqs), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"s{eid}".encode(), b"")
    ora = helpers.OracleSfx(seq=concat, sa=sa, el_size=4, entries=entries)
    with bk.Aligner(None, bk.AlignParams(max_subs=3), d_seq=d_seq.data_ptr(), concat_len=n, d_sa=d_sa.data_ptr(), el_size=4, entries=ent) as al:
        al.snp_reset()
        al.snp_pileup(bases, offs, lens, alns)
        tot_sites = 0
        for min_reads, prop in ((1, 0.0), (2, 0.2), (4, 0.5)):
            for c in range(len(seqs)):
                got, gtot = al.snp_sites(c + 1, min_reads, prop)
                exp, etot = helpers.oracle_snp_sites(ora.h, bases, offs, alns, c + 1, min_reads, prop)
                _assert_sites_equal(got, gtot, exp, etot, (min_reads, prop, c + 1))
                tot_sites += len(exp)
        assert tot_sites > 5000
        # bk_snp_counts (per-locus counts + target base) and the 7-mer centroid histogram built from them
        acc, exp_hist = None, np.zeros(16384, dtype=np.uint32)
        for c, L in enumerate(seq_lens):
            cnt = al.snp_counts(c + 1, 0, L)
            assert np.array_equal(cnt[:, 6], seqs[c] & 7)
            sites, _ = helpers.oracle_snp_sites(ora.h, bases, offs, alns, c + 1, 1, 0.0)
            for st in sites[:: max(1, len(sites) // 50)]:
                row = cnt[int(st["loci"])]
                assert row[0] == st["num_ref"] and np.array_equal(row[1:6], st["non_ref"])
            acc = al.snp_centroid_insts(c + 1, 2, acc)
            tot = cnt[:, :6].sum(axis=1)
            for l in range(3, L - 3):
                w = seqs[c][l - 3:l + 4]
                if tot[l] >= 2 and (w < 4).all():
                    exp_hist[int(sum(int(b) << (2 * (6 - j)) for j, b in enumerate(w)))] += 1
        assert np.array_equal(acc, exp_hist) and exp_hist.sum() > 3000
    ora.close()


@pytest.mark.parametrize("kw", [dict(max_subs=3, max_ml=5, micro_indel_len=10), dict(max_subs=3, max_ml=5, clamp_ml=1, micro_indel_len=10, splice_junct_len=5000),
                                dict(max_subs=5, max_ml=2, splice_junct_len=3000), dict(max_subs=3, max_ml=20, micro_indel_len=20, min_edit_dist=2)])
@pytest.mark.parametrize("fixture", ["indel", "splice", "multi", "combined", "repeat"])
def test_multi_loci_with_indel_and_splice_match_oracle(golden_tmp, fixture, kw):
    """-r1..-r4 together with -a / -A: AlignReads with MaxHits > 1 runs the same microInDel / splice branches for what its phases left
    unaligned; result records, loci lists and second segments against the oracle"""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, fixture, "s3")
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    exp, eo, el, eseg = helpers.oracle_align_multi_indel(sfx, bases, offs[keep], lens[keep], helpers.make_params(**kw))
    sfx.close()
    for knobs in ([], [("chunk_reads", 211)], [("use_wave", 0)]):
        with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(**kw)) as al:
            for k, v in knobs:
                al.tune(k, v)
            got = al.align(bases, offs[keep], lens[keep])
            seg = al.batch_seg2()
            _assert_loci_equal(bk, al, len(keep), got, exp, eo, el)
        for f in ("match_loci", "match_len", "read_ofs", "mismatches", "flags", "score"):
            if not np.array_equal(seg[f], eseg[f]):
                i = int(np.nonzero(seg[f] != eseg[f])[0][0])
                raise AssertionError(f"seg2 field {f} differs at read {i} ({names[keep[i]]}): got {seg[i]} exp {eseg[i]} hit {got[i]}")
    if (fixture in ("indel", "combined") and kw.get("micro_indel_len")) or (fixture in ("splice", "combined") and kw.get("splice_junct_len")):
        assert np.count_nonzero(eseg["flags"] & 5) > 20


def test_pe_with_sparse_entry_ids(golden_tmp, tmp_path):
    """bk_ctx_create_from_device accepts any EntryID values: with ids (7, 3) instead of (1, 2) every kernel that turns a
    ChromID back into a sequence - the orphan recovery of the PE pass included - must go through the id map.  Same placements
    as the file-built context, ChromIDs renamed."""
    import torch
    bk = _bk()
    cfg = PE_RUNS["U3"]
    names, bases, offs, lens = pe_inputs(tmp_path)
    sfx_path = os.path.join(golden_tmp["basic"], "genome.sfx")
    img = np.fromfile(sfx_path, dtype=np.uint8)
    blk = struct.unpack_from("<Q", img, 44)[0]
    n = struct.unpack_from("<Q", img, blk + 8)[0]
    seq = img[blk + 20: blk + 20 + n].copy()
    sa = img[blk + 20 + n: blk + 20 + 5 * n].view("<u4").copy()
    pe = bk.PEParams(cfg["pe"], cfg["d"], cfg["D"], False)
    with bk.Aligner(sfx_path, bk.AlignParams(max_subs=cfg["s"])) as al:
        ents = al.entries()
        exp = al.pair(bases, offs, lens, al.align(bases, offs, lens), pe)
    assert list(ents["entry_id"]) == [1, 2]
    ren = np.array([0, 7, 3], dtype=np.uint32)
    ents2 = ents.copy()
    ents2["entry_id"] = ren[ents["entry_id"]]
    dev = torch.device("cuda", 0)
    d_seq, d_sa = torch.from_numpy(seq).to(dev), torch.from_numpy(sa.view(np.int32)).to(dev)
    with bk.Aligner(None, bk.AlignParams(max_subs=cfg["s"]), d_seq=d_seq.data_ptr(), concat_len=int(n), d_sa=d_sa.data_ptr(), el_size=4,
                    entries=ents2) as al2:
        got = al2.pair(bases, offs, lens, al2.align(bases, offs, lens), pe)
        cnt = al2.seq_counts()
    want = exp.copy()
    want["chrom_id"] = ren[exp["chrom_id"]]
    assert_hits_equal(got, want, names)
    assert np.array_equal(got["flags"] & 0x80, want["flags"] & 0x80)
    assert cnt.sum() > 0


def _sfx_bytes(golden_tmp):
    return bytearray(open(os.path.join(golden_tmp["basic"], "genome.sfx"), "rb").read())


@pytest.mark.parametrize("what,rc", [("truncated", -85), ("version", -86), ("magic", -94), ("entry_outside", -85), ("concat_wraps", -85),
                                     ("blk_ofs_wraps", -85), ("missing", -90)])
def test_corrupt_sfx_files_are_refused_with_the_reference_codes(golden_tmp, tmp_path, what, rc):
    """.sfx files come from outside: every header field is checked against the mapping before it is used, and the failure comes
    back as the teBSFrsltCodes value the reference's loader would return (eBSFerrFileAccess -85, eBSFerrFileVer -86,
    eBSFerrNotBioseq -94, eBSFerrOpnFile -90) - never as a crash or a wild device copy"""
    bk = _bk()
    img = _sfx_bytes(golden_tmp)
    blk = struct.unpack_from("<Q", img, 44)[0]
    ent = struct.unpack_from("<Q", img, 20)[0]
    if what == "truncated":
        img = img[: len(img) // 2]
    elif what == "version":
        struct.pack_into("<i", img, 4, 2)
    elif what == "magic":
        img[0:4] = b"sfy5"
    elif what == "entry_outside":
        struct.pack_into("<Q", img, ent + 8 + 103, 1 << 40)             # EndOfs of the first entry
    elif what == "concat_wraps":
        struct.pack_into("<Q", img, blk + 8, (1 << 64) // 5 + 7)         # ConcatSeqLen * (1 + SfxElSize) wraps to a small number
    elif what == "blk_ofs_wraps":
        struct.pack_into("<Q", img, 44, (1 << 64) - 8)                   # SfxBlockOfs + header size wraps
    p = str(tmp_path / "bad.sfx")
    if what != "missing":
        open(p, "wb").write(img)
    with pytest.raises(bk.BkError) as e:
        bk.Aligner(p, bk.AlignParams(max_subs=3))
    assert e.value.rc == rc


def test_small_chunks_forced_by_memory_pressure(golden_tmp):
    """the chunk size follows the HBM that is free (align_device): with nearly all of it taken a batch runs in several chunks
    (65 536 reads is the floor) and must give the results of the one-chunk run"""
    import ctypes
    import torch
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "basic", "s3")
    offs, lens = offs[keep], lens[keep]
    reps = 200_000 // len(lens) + 1
    offs_r, lens_r = np.tile(offs, reps), np.tile(lens, reps)
    with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(max_subs=3)) as al:
        ref = al.align(bases, offs_r, lens_r)
        al.close()
    with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(max_subs=3)) as al:
        free, _total = torch.cuda.mem_get_info(0)
        hog = torch.empty(max(0, free - (220 << 20)), dtype=torch.uint8, device="cuda:0")      # leaves ~0.2 GB: half of it / 784 B per read < n
        free2, _ = torch.cuda.mem_get_info(0)
        try:
            got = al.align(bases, offs_r, lens_r)
        finally:
            del hog
            torch.cuda.empty_cache()                # or PyTorch's caching allocator keeps the HBM from the tests that follow
    assert free2 < (300 << 20)
    assert_hits_equal(got, ref)
    assert len(lens_r) * 784 > free2 // 2          # the batch could not have fitted one chunk


def test_clears_of_more_than_4_gib_reach_the_end_of_the_buffer():
    """bk_snp_reset clears 24 bytes per target base: 6 GB at 250 Mbp.  Every plane is dirtied at both ends of the target (the last
    plane's end lies 6 GB into the buffer), reset, and read back: a clear that stopped at 4 GiB would leave the far counts standing."""
    import torch
    bk = _bk()
    from biokanga_amd import synth
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(250_000_000, dev, seed=5, n_seqs=2, repeat_frac=0.0, n_gap_frac=0.0)
    n = seq.numel()
    assert n * 24 > (5 << 30)
    d_sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, d_sa.data_ptr(), 4, 0)
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"s{eid}".encode(), b"")
    last_id, last_len = entries[-1][0], entries[-1][1]
    # two reads of 100 bases that differ from the target everywhere (all five "non reference" planes and the reference plane get counts)
    places = [(1, 0), (last_id, last_len - 100)]
    bases = np.zeros(200, dtype=np.uint8)
    alns = np.zeros(2, dtype=bk.SNP_ALN_DTYPE)
    seq_h = {eid: seq[so:so + slen] for eid, slen, so, _ in entries}
    for k, (eid, lo) in enumerate(places):
        t = seq_h[eid][lo:lo + 100].cpu().numpy()
        r = t.copy()
        r[0::5] = (t[0::5] + 1) & 3; r[1::5] = (t[1::5] + 2) & 3; r[2::5] = (t[2::5] + 3) & 3; r[3::5] = 4       # a,c,g,t substitutions, N; every fifth base matches
        bases[100 * k:100 * k + 100] = r
        alns[k] = (k, eid, lo, 100, 0, ord("+"), (0, 0, 0))
    offs, lens = np.array([0, 100], dtype=np.uint64), np.array([100, 100], dtype=np.uint32)
    with bk.Aligner(None, bk.AlignParams(max_subs=3), d_seq=seq.data_ptr(), concat_len=n, d_sa=d_sa.data_ptr(), el_size=4, entries=ent) as al:
        al.snp_reset()
        al.snp_pileup(bases, offs, lens, alns)
        for eid, lo in places:
            c = al.snp_counts(eid, lo, 100)
            assert (c[:, :6].sum(axis=0) > 0).all(), c[:, :6].sum(axis=0)         # every plane holds counts there
        al.snp_reset()
        for eid, lo in places:
            assert int(al.snp_counts(eid, lo, 100)[:, :6].sum()) == 0


def _synth_index(bk, bp, seed, repeat_frac=0.5):
    import torch
    from biokanga_amd import synth
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(bp, dev, seed=seed, n_seqs=3, repeat_frac=repeat_frac)
    n = seq.numel()
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), 4, 0)
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    return dev, seq, seq_lens, n, sa, ent


def test_phase_loop_without_readbacks_and_the_enqueue_only_call():
    """The main path's phase loop keeps every count in device memory (PhaseCtl): launches are sized by bounds, the two work-list sorts
    by what the PREVIOUS chunk needed.  Batches that follow each other with very different needs (reads without substitutions out of
    unique sequence, then repeat-rich reads with three) must come out exactly as the loop that reads its counts back produces them,
    counters included; bk_align_batch_device_async enqueues the same batch on a caller's stream and a consumer kernel behind it on that
    stream sees the finished records without the host having waited."""
    import torch
    bk = _bk()
    from biokanga_amd import synth
    dev, seq, seq_lens, n, sa, ent = _synth_index(bk, 12_000_000, 29, repeat_frac=0.6)
    sets = []
    for seed, subs, cnt in ((11, 0, 150_000), (12, 3, 300_000), (13, 1, 80_000)):
        b, o, l, _ = synth.make_reads(seq, seq_lens, cnt, 100, dev, seed=seed, max_subs=subs)
        sets.append((b, o, l, cnt))
    with bk.Aligner(None, bk.AlignParams(max_subs=3), d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=4, entries=ent) as al:
        def run(which, out):
            b, o, l, cnt = sets[which]
            al.align_device(b.data_ptr(), o.data_ptr(), l.data_ptr(), cnt, out.data_ptr())
            return out.cpu().numpy().view(bk.HIT_DTYPE).copy()
        outs = [torch.zeros(s[3] * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev) for s in sets]
        al.tune("async_phases", 0)
        ref, ref_ctr = [], []
        for w in range(3):
            al.counters(reset=True)
            ref.append(run(w, outs[w]))
            ref_ctr.append(al.counters())
        assert int((ref[1]["nar"] == 1).sum()) > sets[1][3] // 2
        al.tune("async_phases", 1)
        for order in ((0, 1, 2), (2, 1, 0), (1, 1, 0)):          # the guesses of a phase's sort sizes come from whichever batch ran before
            for w in order:
                outs[w].zero_()
                al.counters(reset=True)
                got = run(w, outs[w])
                assert_hits_equal(got, ref[w])
                c = al.counters()
                for k in ("n_search", "n_cand", "n_lcm_calls"):
                    assert c[k] == ref_ctr[w][k], (order, w, k)
        # chunked batches: every chunk its own PhaseCtl lines, the history follows the last chunk
        al.tune("chunk_reads", 70_001)
        outs[1].zero_()
        assert_hits_equal(run(1, outs[1]), ref[1])
        al.tune("chunk_reads", 1 << 26)
        # ---- the call that only enqueues
        b, o, l, cnt = sets[1]
        with pytest.raises(bk.BkError):                          # scratch for 100-base reads only: a longer promise is refused before anything is launched
            al.align_device_async(b.data_ptr(), o.data_ptr(), l.data_ptr(), cnt, 400, outs[1].data_ptr())
        al.reserve(cnt, 100)
        st = torch.cuda.Stream(device=dev)
        outs[1].zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            al.align_device_async(b.data_ptr(), o.data_ptr(), l.data_ptr(), cnt, 100, outs[1].data_ptr(), stream=st.cuda_stream)
            # the consumer: a kernel on the same stream, enqueued while the phases have not run yet
            accepted = (outs[1].view(cnt, bk.HIT_DTYPE.itemsize)[:, bk.HIT_DTYPE.fields["nar"][1]] == 1).sum()
            snapshot = outs[1].clone()
        assert int(accepted.item()) == int((ref[1]["nar"] == 1).sum())
        assert_hits_equal(snapshot.cpu().numpy().view(bk.HIT_DTYPE), ref[1])
        # .. twice in a row on one stream (the second call's sorts are sized from the first one's counts when they have arrived), then a blocking call
        with torch.cuda.stream(st):
            for w in (0, 2):
                bb, oo, ll, cc = sets[w]
                outs[w].zero_()
                al.align_device_async(bb.data_ptr(), oo.data_ptr(), ll.data_ptr(), cc, 100, outs[w].data_ptr(), stream=st.cuda_stream)
        st.synchronize()
        for w in (0, 2):
            assert_hits_equal(outs[w].cpu().numpy().view(bk.HIT_DTYPE), ref[w])
        outs[1].zero_()
        assert_hits_equal(run(1, outs[1]), ref[1])
        # a read longer than promised is reported by the next call
        b2, o2, l2, _ = synth.make_reads(seq, seq_lens, 10_000, 120, dev, seed=14, max_subs=1)
        al.reserve(10_000, 120)
        out2 = torch.zeros(10_000 * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        al.align_device_async(b2.data_ptr(), o2.data_ptr(), l2.data_ptr(), 10_000, 100, out2.data_ptr())
        torch.cuda.synchronize()
        with pytest.raises(bk.BkError):
            for _ in range(50):                                   # (the counts travel behind the kernels; the next call that finds them there reports)
                al.align_device_async(b2.data_ptr(), o2.data_ptr(), l2.data_ptr(), 10_000, 120, out2.data_ptr())
                torch.cuda.synchronize()
