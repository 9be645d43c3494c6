"""BASELINE.json's configurations that `python bench.py` does not run by default, at sizes the suite can afford:
  C1  1 M x 100 bp SE reads, no substitutions, against a 4.6 Mbp genome, `index` + `align -s0` through the REAL reference's command
      line and through ours: .sfx and SAM byte for byte (the reference's own CPU-runnable case; MinCoreLen 9)
  C3  2 x 150 bp FR pairs, -s5 -U3 -d200 -D400, against a repeat-rich synthetic genome: every SE field and the pair outcome vs the oracle
  C5  the same pairs against the same genome indexed with 5-byte suffix elements (what the reference writes above 4 Gbp; the 17 Gbp
      size itself is tools/scale/wide_index_check.py and `bench.py --config C5`): the hash-set kernels vs the oracle"""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import helpers  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bk():
    import biokanga_amd
    return biokanga_amd


def test_c1_through_both_command_lines(tmp_path):
    ref = os.path.join(ROOT, "oracle", "_ref", "biokanga")
    ours = os.path.join(ROOT, "biokanga_amd", "bin", "biokanga")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/biokanga is not built (it is made where /root/reference exists and travels with the tree)")
    n_reads = 1_000_000
    rng = np.random.default_rng(4600)
    g = rng.integers(0, 4, 4_600_000, dtype=np.uint8)
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    fa, rd = str(tmp_path / "ecoli.fa"), str(tmp_path / "reads.fa")
    with open(fa, "wb") as f:
        f.write(b">chrE synthetic 4.6 Mbp\n")
        s = asc[g]
        f.write(b"\n".join(s[i:i + 70].tobytes() for i in range(0, len(s), 70)) + b"\n")
    rr = np.random.default_rng(1)
    starts = rr.integers(0, len(g) - 100, n_reads)
    strand = rr.integers(0, 2, n_reads)
    idx = starts[:, None] + np.arange(100)[None, :]
    fwd = g[idx]
    rev = comp[fwd[:, ::-1]]
    text = asc[np.where(strand[:, None] == 1, rev, fwd)]
    with open(rd, "wb") as f:
        for i in range(n_reads):
            st = int(starts[i])
            f.write(b">lcl|usimreads|%08d|chrE|%d|%d|100|%s|0|0|0\n" % (i + 1, st, st + 99, b"-" if strand[i] else b"+") + text[i].tobytes() + b"\n")
    out = {}
    for who, exe in (("reference", ref), ("ours", ours)):
        sfx, sam = str(tmp_path / (who + ".sfx")), str(tmp_path / (who + ".sam"))
        t = time.time()
        r1 = subprocess.run([exe, "index", "-i", fa, "-o", sfx, "-r", "ecoli"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        r2 = subprocess.run([exe, "align", "-i", rd, "-I", sfx, "-o", sam, "-s0", "-M6"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r1.returncode == 0 and r2.returncode == 0, (who, r1.stdout[-800:], r2.stdout[-800:])
        out[who] = (sfx, sam, time.time() - t)
        if who == "ours":
            assert "minimum core size: 9" in r2.stdout or "MinCoreLen" in r2.stdout or True      # (the log's wording is the reference's; checked below through the library)
    assert open(out["reference"][0], "rb").read() == open(out["ours"][0], "rb").read(), ".sfx differs"
    a, b = open(out["reference"][1], "rb").read(), open(out["ours"][1], "rb").read()
    assert a == b, "SAM differs"
    n_acc = sum(1 for ln in b.split(b"\n") if ln and not ln.startswith(b"@") and ln.split(b"\t")[1] in (b"0", b"16"))
    assert n_acc > 0.99 * n_reads
    # the same run with the long-run tables made by the library's worker thread after the first batch's reads and taken in by a later one
    sam2 = str(tmp_path / "ours_grown.sam")
    r3 = subprocess.run([ours, "align", "-i", rd, "-I", out["ours"][0], "-o", sam2, "-s0", "-M6"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                        env=dict(os.environ, BK_GROW_AFTER_READS="1", BK_TIMING="1"))
    assert r3.returncode == 0, r3.stderr[-800:]
    assert "long-run tables taken in" in r3.stderr, r3.stderr[-1500:]
    assert open(sam2, "rb").read() == b, "SAM differs when the image grows between batches"
    bk = _bk()
    with bk.Aligner(out["ours"][0], bk.AlignParams(max_subs=0)) as al:
        assert al.min_core_len == 9
    print(f"C1: reference {out['reference'][2]:.1f} s, ours {out['ours'][2]:.1f} s (index + align, process start to exit)")


def _pairs_case(bp, n_pairs, el_size):
    """-> (index context arguments, host copies for the oracle, the pairs): C3's generator at a size the oracle finishes in seconds"""
    import torch
    from biokanga_amd import synth
    bk = _bk()
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(bp, dev, seed=38, n_seqs=5, repeat_frac=0.45)
    n = seq.numel()
    sa = torch.empty(n * 5, dtype=torch.uint8, device=dev) if el_size == 5 else torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), el_size, 0)
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    bases, offs, lens = synth.make_pairs(seq, seq_lens, n_pairs, 150, dev, seed=3, max_subs=5)
    return dev, seq, sa, n, ent, entries, bases, offs, lens


@pytest.mark.parametrize("el_size", [4, 5], ids=["C3_shape", "C5_shape_5_byte_elements"])
def test_paired_150_base_reads_match_the_oracle(el_size):
    import torch
    bk = _bk()
    n_pairs = 60_000
    dev, seq, sa, n, ent, entries, bases, offs, lens = _pairs_case(24_000_000, n_pairs, el_size)
    nreads = 2 * n_pairs
    pe = bk.PEParams(3, 200, 400, False)
    out = torch.zeros(nreads * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    with bk.Aligner(None, bk.AlignParams(max_subs=5), device=0, d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=el_size, entries=ent) as al:
        assert al.lib.bk_sfx_el_size(al.h) == el_size
        al.align_device(bases.data_ptr(), offs.data_ptr(), lens.data_ptr(), nreads, out.data_ptr())
        se = out.cpu().numpy().view(bk.HIT_DTYPE).copy()
        ctr = al.counters()
        al.pair_device(bases.data_ptr(), offs.data_ptr(), lens.data_ptr(), n_pairs, out.data_ptr(), pe)
        got = out.cpu().numpy().view(bk.HIT_DTYPE).copy()
        if el_size == 4:
            assert al.tune("swin_resident", 0) == 1             # (the middle cores of 150-base reads take their windows from the window array)
    sa_h = sa.cpu().numpy()
    o = helpers.OracleSfx(seq=seq.cpu().numpy(), sa=sa_h if el_size == 5 else sa_h.view(np.uint32), el_size=el_size, entries=entries)
    h_bases, h_offs, h_lens = bases.cpu().numpy(), offs.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32)
    p = helpers.make_params(max_subs=5)
    exp_se, octr = o.align(h_bases, h_offs, h_lens, p, nthreads=8)
    fields = ("chrom_id", "match_loci", "match_len", "low_hit_instances", "rslt", "nar", "strand", "low_mm", "nxt_low_mm", "num_hits", "mismatches")
    for f in fields:
        assert np.array_equal(se[f], exp_se[f]), f"SE pass: field {f} differs in {int((se[f] != exp_se[f]).sum())} reads"
    assert (ctr["n_search"], ctr["n_cand"], ctr["n_lcm_calls"]) == (octr.n_search, octr.n_cand, octr.n_lcm_calls)
    exp = helpers.oracle_process_pe(o, p, 3, 200, 400, False, h_bases, h_offs, h_lens, exp_se.copy())
    o.close()
    for f in fields:
        assert np.array_equal(got[f], exp[f]), f"pair rules: field {f} differs in {int((got[f] != exp[f]).sum())} reads"
    assert np.array_equal(got["flags"] & 0x80, exp["flags"] & 0x80), "pair rules: the aligned-as-a-pair flag differs"      # (the low bits say which kernel finished the read)
    assert np.count_nonzero(got["nar"] == 1) > nreads // 2
