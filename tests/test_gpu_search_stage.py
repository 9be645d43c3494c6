"""Stage-level parity of the search: the suffix array interval the device search (pass A: k-mer table + small buckets out of the
second-level keys; pass B: sampled-level bisection of the second-, third- and fourth-level keys, suffix array + target behind them;
bk_search.hip, bk_dev_k2.h) leaves for every (read, strand, core) of a phase, against the reference's own two functions restated in the
oracle - LocateFirstExact (SfxArrayV2.cpp:7765-7876) and LocateLastExact (:7914-8027) - probe by probe.  The end-to-end tests only see
these intervals through what the extension makes of them; here a wrong interval names the probe that failed.

Cases: the `repeat` golden index (a 25-mer present thousands of times); a 200 Mbp genome with 45 % of repeat-derived bases at the
default k-mer table order (16: buckets of thousands of suffixes, bisected through several sampled levels, cores of 100 / 50 / 33 / 25 and
150 / 75 / .. bases reaching the fourth-level keys and the suffix array behind them), on the image with every table and on the lean one,
4- and 5-byte suffix array elements, with the unverified small buckets ("lazy_search") off and on."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import helpers  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
COMP = np.array([3, 2, 1, 0, 4, 5, 6, 7], dtype=np.uint8)
KIND_MASK = (1 << 30) - 1
LAZY = 1 << 31
ELEM = 1 << 30          # with LAZY and a count of 1: `first` is the one suffix's array element, not its index (kElemFlag)


def _bk():
    import biokanga_amd
    return biokanga_amd


# ProcCoredApprox's parameters and AlignReads' schedule (Aligner.cpp:9085-9095, SfxArrayV2.cpp:7695-7719), LocateCoreMultiples' core
# offsets (:5836-5847) - restated here on their own, so that the test does not take its core geometry from the code it checks
def _plan(length, max_subs, min_core_len, slides_per100=8, mm_delta=1):
    m = 0 if max_subs == 0 else max(1, int(0.5 + length * max_subs / 100.0))
    m = min(m, 63)
    core_len = max(min_core_len, length // (m + 1 if mm_delta == 1 else m + 2))
    max_slides = max(1, (slides_per100 * length + 99) // 100)
    core_delta = max(length // max_slides - 1, core_len)
    phases = []
    final = True
    if m > 0:
        final = False
        for a in range(m + 1):
            cl = length // (a + mm_delta)
            if cl <= core_len:
                final = True
                break
            phases.append((cl, cl))
    if final:
        phases.append((core_len, core_delta))
    return phases, max_slides


def _core_offsets(length, cl, cd, max_slides):
    out, cur, o = [], cd, 0
    while len(out) < max_slides and o <= length - cl and cur > cl // 3:
        if o + cl + cur > length:
            cur = length - (o + cl)
        out.append(o)
        o += cur
    return out


def _check_phase(al, osfx, bases, offs, lens, phase, max_subs, min_core_len, lazy, what):
    """every interval record of `phase` against LocateFirstExact / LocateLastExact; returns the number of probes compared"""
    act, first, count, ivc = al.search_intervals(bases, offs, lens, phase)
    if len(act) == 0:
        return 0
    # both strands of the reads on the list, 1 byte per base
    rows, row_ofs, at = [], {}, 0
    for r in act.tolist():
        b = bases[int(offs[r]):int(offs[r]) + int(lens[r])] & 7
        row_ofs[r] = at
        rows += [b, COMP[b[::-1]]]
        at += 2 * len(b)
    flat = np.ascontiguousarray(np.concatenate(rows))
    p_ofs, p_len, where = [], [], []
    for a, r in enumerate(act.tolist()):
        L = int(lens[r])
        phases, max_slides = _plan(L, max_subs, min_core_len)
        assert phase < len(phases), f"{what}: read {r} of {L} bases is on the list of phase {phase} but has {len(phases)} phases"
        cl, cd = phases[phase]
        cores = _core_offsets(L, cl, cd, max_slides)
        assert len(cores) <= ivc
        for st in (0, 1):
            for c in range(ivc):
                if c < len(cores):
                    p_ofs.append(row_ofs[r] + st * L + cores[c])
                    p_len.append(cl)
                    where.append((st * ivc + c, a, r, st, c, cores[c], cl))
                else:
                    assert (int(count[st * ivc + c, a]) & KIND_MASK) == 0 or True      # (slots of cores the read does not have are not cleared beyond cmax)
    n = len(p_ofs)
    p_ofs = np.array(p_ofs, dtype=np.uint64)
    p_len = np.array(p_len, dtype=np.int32)
    f = np.zeros(n, dtype=np.int64)
    l = np.zeros(n, dtype=np.int64)
    osfx.lib.ora_locate_cores(osfx.h, flat.ctypes.data, p_ofs.ctypes.data, p_len.ctypes.data, n, f.ctypes.data, l.ctypes.data, 8)
    bad = []
    n_lazy = 0
    for i, (plane, a, r, st, c, co, cl) in enumerate(where):
        got_first, raw = int(first[plane, a]), int(count[plane, a])
        got_n = raw & KIND_MASK
        assert (raw >> 30) in (0, 2, 3), f"{what}: read {r} strand {st} core {c}: a work item's kind left in the record ({raw:#x})"
        exp_first, exp_n = (int(f[i]) - 1, int(l[i]) - int(f[i]) + 1) if f[i] else (None, 0)
        if (raw >> 30) == 3:
            # a k-mer bucket of ONE suffix handed on as the suffix itself (its target position): where the core has a match at all it is
            # that suffix - the only one that begins with the core's first k bases
            n_lazy += 1
            assert lazy, f"{what}: element record with lazy_search off"
            ok = got_n == 1 and (exp_n == 0 or (exp_n == 1 and int(osfx.lib.ora_sa_element(osfx.h, exp_first)) == got_first))
        elif raw & LAZY:
            # an unverified k-mer bucket of at most four suffixes: the exact interval lies inside it
            n_lazy += 1
            assert lazy, f"{what}: lazy record with lazy_search off"
            ok = 1 <= got_n <= 4 and (exp_n == 0 or (got_first <= exp_first and exp_first + exp_n <= got_first + got_n))
        else:
            ok = got_n == exp_n and (exp_n == 0 or got_first == exp_first)
        if not ok:
            bad.append(f"read {r} strand {'+-'[st]} core {c} (offset {co}, {cl} bases): device [{got_first}, +{got_n}){' lazy' if raw & LAZY else ''}, "
                       f"LocateFirstExact / LocateLastExact [{exp_first}, +{exp_n})")
    assert not bad, f"{what}, phase {phase}: {len(bad)} of {n} probes differ:\n" + "\n".join(bad[:12])
    return n


def _reads_from(seq_h, rng, n_reads, length, max_e, n_with_n=0):
    """reads cut from the target, either strand, up to max_e substitutions; a few with an N"""
    out = np.zeros((n_reads, length), dtype=np.uint8)
    k = 0
    while k < n_reads:
        s = int(rng.integers(0, len(seq_h) - length))
        w = seq_h[s:s + length] & 7
        if (w > 3).any():
            continue
        w = w.copy()
        for p in rng.choice(length, size=int(rng.integers(0, max_e + 1)), replace=False):
            w[p] = (w[p] + 1 + rng.integers(0, 3)) & 3
        if rng.integers(0, 2):
            w = COMP[w[::-1]]
        if k < n_with_n:
            w[int(rng.integers(0, length))] = 4
        out[k] = w
        k += 1
    return out


def test_repeat_golden_intervals(tmp_path):
    bk = _bk()
    sfx = str(tmp_path / "genome.sfx")
    rd = str(tmp_path / "reads.fa")
    helpers.gunzip_to(os.path.join(GOLD, "repeat", "genome.sfx.gz"), sfx)
    helpers.gunzip_to(os.path.join(GOLD, "repeat", "reads.fa.gz"), rd)
    names, bases, offs, lens = helpers.read_fasta_reads(rd)
    keep = helpers.filter_reads_by_len(names, bases, offs, lens)
    offs, lens = offs[keep], lens[keep]
    o = helpers.OracleSfx(sfx)
    total = 0
    for lazy in (0, 1):
        for kbits in (0, 8):
            with bk.Aligner(sfx, bk.AlignParams(max_subs=3)) as al:
                al.tune("lazy_search", lazy)
                if kbits:
                    al.tune("kmer_bits", kbits)                 # (a smaller table: buckets of hundreds of suffixes on a 30 kbp genome)
                mcl = al.min_core_len
                for phase in range(4):
                    total += _check_phase(al, o, bases, offs, lens, phase, 3, mcl, lazy, f"repeat golden, lazy {lazy}, kmer_bits {kbits or 'default'}")
    o.close()
    assert total > 1000


@pytest.mark.parametrize("el_size", [4, 5], ids=["4_byte_elements", "5_byte_elements"])
def test_intervals_on_a_200_mbp_repeat_rich_index(el_size):
    import torch
    from biokanga_amd import synth
    bk = _bk()
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(200_000_000, dev, seed=61, n_seqs=6, repeat_frac=0.45)
    n = seq.numel()
    sa = torch.empty(n * 5, dtype=torch.uint8, device=dev) if el_size == 5 else torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), el_size, 0)
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    seq_h = seq.cpu().numpy()
    sa_h = sa.cpu().numpy()
    o = helpers.OracleSfx(seq=seq_h, sa=sa_h if el_size == 5 else sa_h.view(np.uint32), el_size=el_size, entries=entries)
    rng = np.random.default_rng(77 + el_size)
    total = 0
    cases = ((100, 3, 6000), (150, 5, 2500)) if el_size == 4 else ((100, 3, 3000), (150, 5, 1200))
    for length, subs, n_reads in cases:
        reads = _reads_from(seq_h, rng, n_reads, length, subs, n_with_n=40)
        bases = reads.reshape(-1)
        offs = np.arange(n_reads, dtype=np.uint64) * length
        lens = np.full(n_reads, length, dtype=np.uint32)
        with bk.Aligner(None, bk.AlignParams(max_subs=subs), device=0, d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=el_size, entries=ent) as al:
            assert al.lib.bk_sfx_el_size(al.h) == el_size
            mcl = al.min_core_len
            n_phases = len(_plan(length, subs, mcl)[0])
            images = [("every table", ())]
            if el_size == 4:
                assert al.tune("k3_resident", 0) == 2 and al.tune("ktab2_resident", 0) == 1
                assert al.tune("ktab2_elem", 0) == 0
                images.append(("k-mer table entries that carry the suffix of a bucket of one", (("use_ktab2", 2),)))
                images.append(("lean image", (("use_k3", 0), ("use_ktab2", 0))))
            for image, knobs in images:
                for kv in knobs:
                    al.tune(*kv)
                if image == "lean image":
                    assert al.tune("k3_resident", 0) == 0 and al.tune("ktab2_resident", 0) == 0
                elif knobs:
                    assert al.tune("ktab2_resident", 0) == 1 and al.tune("ktab2_elem", 0) == 1
                for lazy in (0, 1):
                    al.tune("lazy_search", lazy)
                    for phase in range(n_phases):
                        total += _check_phase(al, o, bases, offs, lens, phase, subs, mcl, lazy, f"{length}-base reads, {el_size}-byte elements, {image}, lazy {lazy}")
    o.close()
    assert total > 50000
