"""Multi-loci modes on CPU: (a) the oracle's pHits[] lists against what the real reference reported with -r5
(the first CSV column is the record's creation order, so the discovery ORDER of the loci is pinned, not just the
set); (b) the host-side policies above the C ABI - glibc rand() picks (-r2) and the clustering assignment
(-r3/-r4, multi_assign.h) - fed with the oracle's lists, against the reference's SAM files."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import helpers
from test_oracle_golden import check_hits_against_sam, chrom_names_from_hdr


@pytest.fixture(scope="module")
def multi_case(golden_tmp):
    d = golden_tmp["multi"]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    return d, names, bases, offs, lens


def _oracle(d, bases, offs, lens, max_ml, clamp=0, best=0, max_subs=3):
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    out = helpers.oracle_align_multi(sfx, bases, offs, lens, helpers.make_params(max_subs=max_subs, max_ml=max_ml, clamp_ml=clamp, best_matches=best),
                                     nthreads=8)
    sfx.close()
    return out


@pytest.mark.parametrize("tag,max_ml,best,max_subs", [("r5R5", 5, 0, 3), ("r5R5N", 5, 1, 3), ("r5R2Ns1", 2, 1, 1)])
def test_oracle_loci_order_matches_reference_r5(multi_case, tag, max_ml, best, max_subs):
    d, names, bases, offs, lens = multi_case
    hits, lo, loci = _oracle(d, bases, offs, lens, max_ml, 0, best, max_subs)
    rows = []
    for line in gzip.open(os.path.join(helpers.GOLDEN, "multi", f"{tag}.m0.csv.gz"), "rt"):
        f = line.rstrip("\n").split(",")
        rows.append((int(f[0]), f[3].strip('"'), int(f[4]), f[7].strip('"'), int(f[11]), f[13].strip('"')))
    rows.sort()
    assert [r[0] for r in rows] == list(range(1, len(rows) + 1)) and len(rows) == len(loci) > 3000
    chrom = {1: "mA", 2: "mB"}
    k = 0
    for i, nm in enumerate(names):
        for j in range(int(lo[i]), int(lo[i + 1])):
            L = loci[j]
            assert rows[k][1:] == (chrom[int(L["chrom_id"])], int(L["match_loci"]), chr(L["strand"]), int(L["mismatches"]), nm), (k, nm)
            k += 1
    assert np.count_nonzero(np.diff(lo.astype(np.int64)) > 1) > 500


@pytest.fixture(scope="module")
def multi_harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("mh") / "multi_harness")
    subprocess.check_call(helpers.cxx() + ["-pthread", "-o", exe, os.path.join(helpers.ROOT, "tests", "cpp", "multi_harness.cpp")])
    return exe


@pytest.mark.parametrize("tag,mode,max_ml,clamp,threads", [("r2R5", 2, 5, 0, 1), ("r3R5", 3, 5, 0, 4), ("r4R5", 4, 5, 0, 4), ("r4R3X", 4, 3, 1, 4),
                                                            ("r3R8T1", 3, 8, 0, 1), ("r3R5", 3, 5, 0, 64)])
def test_host_policies_match_reference(multi_case, multi_harness, tmp_path, tag, mode, max_ml, clamp, threads):
    d, names, bases, offs, lens = multi_case
    hits, lo, loci = _oracle(d, bases, offs, lens, max_ml, clamp)
    hp, op, lp, rp = (str(tmp_path / n) for n in ("hits.bin", "offs.bin", "loci.bin", "out.bin"))
    hits.tofile(hp); lo.tofile(op); loci.tofile(lp)
    subprocess.check_call([multi_harness, str(mode), str(threads), str(int(lens.max())), str(clamp), hp, op, lp, rp])
    got = np.fromfile(rp, dtype=helpers.HIT_DTYPE)
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, "multi", f"{tag}.m6.sam.gz"))
    check_hits_against_sam(names, lens, got, recs, chrom_names_from_hdr(hdr), list(range(len(names))))
    assert np.count_nonzero(got["nar"] == 1) > np.count_nonzero(hits["nar"] == 1)        # some multi-loci reads were placed


def test_r1_is_the_plain_result(multi_case):
    """-r1 only gathers statistics: the records are what AlignReads returned with MaxHits = -R"""
    d, names, bases, offs, lens = multi_case
    hits, lo, loci = _oracle(d, bases, offs, lens, 5)
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, "multi", "r1R5.m6.sam.gz"))
    check_hits_against_sam(names, lens, hits, recs, chrom_names_from_hdr(hdr), list(range(len(names))))


def test_parallel_sort_of_the_clustering(multi_harness):
    """par_sort (chunks sorted on all threads, merged pairwise) equals std::sort; sized so that the parallel path really runs"""
    assert subprocess.check_output([multi_harness, "sorttest", "3000000"]).strip() == b"ok"


@pytest.mark.parametrize("tag,mode,kw,threads", [("r2R5c50", 2, dict(max_subs=3, min_chimeric_len=50, max_ml=5), 1),
                                                 ("r3R5c50", 3, dict(max_subs=3, min_chimeric_len=50, max_ml=5), 4),
                                                 ("r4R5c60", 4, dict(max_subs=3, min_chimeric_len=60, max_ml=5), 4),
                                                 ("r4R3Xc70s5", 4, dict(max_subs=5, min_chimeric_len=70, max_ml=3, clamp_ml=1), 4)])
def test_host_policies_with_chimeric_trims_match_reference(golden_tmp, multi_harness, tmp_path, tag, mode, kw, threads):
    """-c with -r2 / -r3 / -r4: the loci the chimeric call lists carry their own end trims; the random pick takes them along and the
    clustering sorts and scores on the trimmed loci (AdjStartLoci / AdjEndLoci / AdjHitLen) - fed with the oracle's lists, against
    POS and CIGAR (soft clips) of the reference's SAM"""
    d = golden_tmp["chimml"]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    hits, lo, loci, trims, seg2 = helpers.oracle_align_multi_chimeric(sfx, bases, offs, lens, helpers.make_params(**kw), nthreads=8)
    sfx.close()
    hp, op, lp, rp, tp, otp = (str(tmp_path / n) for n in ("hits.bin", "offs.bin", "loci.bin", "out.bin", "trims.bin", "out_trims.bin"))
    hits.tofile(hp); lo.tofile(op); loci.tofile(lp); trims.tofile(tp)
    subprocess.check_call([multi_harness, str(mode), str(threads), str(int(lens.max())), str(kw.get("clamp_ml", 0)), hp, op, lp, rp, tp, otp])
    got = np.fromfile(rp, dtype=helpers.HIT_DTYPE)
    rt = np.fromfile(otp, dtype=helpers.TRIMS_DTYPE)
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, "chimml", f"{tag}.m6.sam.gz"))
    by = {r["qname"]: r for r in recs}
    chrom = chrom_names_from_hdr(hdr)
    n_trimmed = 0
    for i, nm in enumerate(names):
        h, r = got[i], by[nm]
        assert helpers.NAR_TAGS[h["nar"]] == r["nar"], (nm, h, r)
        if h["nar"] != 1:
            continue
        taken = not np.array_equal(got[i], hits[i])                       # the policy placed this read; else its own unique placement stands
        if taken:
            tl, tr = int(rt["left"][i]), int(rt["right"][i])
        else:
            tl, tr = (int(seg2["match_len"][i]), int(seg2["read_ofs"][i])) if seg2["flags"][i] & 8 else (0, 0)
        n_trimmed += 1 if (taken and (tl or tr)) else 0
        plus = chr(h["strand"]) == "+"
        c5, c3 = (tl, tr) if plus else (tr, tl)
        cig = (f"{c5}S" if c5 else "") + f"{int(h['match_len']) - tl - tr}M" + (f"{c3}S" if c3 else "")
        assert (chrom[h["chrom_id"] - 1], int(h["match_loci"]) + (tl if plus else tr) + 1, cig) == (r["rname"], r["pos"], r["cigar"]), (nm, h, rt[i], r)
    assert n_trimmed > 10, n_trimmed

@pytest.mark.parametrize("tag,kw", [("r1R5c50a8", dict(max_subs=3, min_chimeric_len=50, max_ml=5, micro_indel_len=8))])
def test_r1_with_chimeric_trims_and_indels_is_the_plain_result(golden_tmp, tag, kw):
    """-c with -r1 AND -a (the combination pinned in round 5, tests/golden/chimmlindel): what AlignReads returned with MaxHits = -R - the
    oracle's records against the reference's SAM: every read's outcome (a read placed as two segments around a microInDel is the
    reference's 'orphaned microInDel'), and locus + CIGAR with the chimeric call's soft clips of every read placed in one piece"""
    d = golden_tmp["chimmlindel"]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    hits, lo, loci, trims, seg2 = helpers.oracle_align_multi_chimeric(sfx, bases, offs, lens, helpers.make_params(**kw), nthreads=8)
    sfx.close()
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, "chimmlindel", f"{tag}.m6.sam.gz"))
    by = {r["qname"]: r for r in recs}
    chrom = chrom_names_from_hdr(hdr)
    n_two = n_clipped = 0
    for i, nm in enumerate(names):
        h, r = hits[i], by[nm]
        two = (int(seg2["flags"][i]) & 7) != 0                      # a microInDel / splice junction placement: two segments
        assert ("OM" if (two and h["nar"] == 1) else helpers.NAR_TAGS[h["nar"]]) == r["nar"], (nm, h, seg2[i], r)
        n_two += 1 if two else 0
        if h["nar"] != 1 or two:
            continue
        tl, tr = (int(seg2["match_len"][i]), int(seg2["read_ofs"][i])) if int(seg2["flags"][i]) & 8 else (0, 0)
        n_clipped += 1 if (tl or tr) else 0
        plus = chr(h["strand"]) == "+"
        c5, c3 = (tl, tr) if plus else (tr, tl)
        cig = (f"{c5}S" if c5 else "") + f"{int(h['match_len']) - tl - tr}M" + (f"{c3}S" if c3 else "")
        assert (chrom[h["chrom_id"] - 1], int(h["match_loci"]) + (tl if plus else tr) + 1, cig) == (r["rname"], r["pos"], r["cigar"]), (nm, h, seg2[i], r)
    assert n_two > 40 and n_clipped > 100, (n_two, n_clipped)

@pytest.mark.parametrize("tag,mode,kw,threads", [("r2R5c50a8", 2, dict(max_subs=3, min_chimeric_len=50, max_ml=5, micro_indel_len=8), 1),
                                                 ("r3R5c50a8", 3, dict(max_subs=3, min_chimeric_len=50, max_ml=5, micro_indel_len=8), 4),
                                                 ("r4R5c60a5", 4, dict(max_subs=3, min_chimeric_len=60, max_ml=5, micro_indel_len=5), 4),
                                                 ("r3R3Xc55a10A200", 3, dict(max_subs=3, min_chimeric_len=55, max_ml=3, clamp_ml=1, micro_indel_len=10, splice_junct_len=200), 4)])
def test_host_policies_with_chimeric_trims_and_indels_match_reference(golden_tmp, multi_harness, tmp_path, tag, mode, kw, threads):
    """-c with -r2 / -r3 / -r4 AND -a / -A: the random pick and the clustering over the chimeric call's lists, beside reads the microInDel /
    splice junction searches placed as two segments - fed with the oracle's lists, against the reference's SAM of the same options"""
    d = golden_tmp["chimmlindel"]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    hits, lo, loci, trims, seg2 = helpers.oracle_align_multi_chimeric(sfx, bases, offs, lens, helpers.make_params(**kw), nthreads=8)
    sfx.close()
    hp, op, lp, rp, tp, otp = (str(tmp_path / n) for n in ("hits.bin", "offs.bin", "loci.bin", "out.bin", "trims.bin", "out_trims.bin"))
    hits.tofile(hp); lo.tofile(op); loci.tofile(lp); trims.tofile(tp)
    subprocess.check_call([multi_harness, str(mode), str(threads), str(int(lens.max())), str(kw.get("clamp_ml", 0)), hp, op, lp, rp, tp, otp])
    got = np.fromfile(rp, dtype=helpers.HIT_DTYPE)
    rt = np.fromfile(otp, dtype=helpers.TRIMS_DTYPE)
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, "chimmlindel", f"{tag}.m6.sam.gz"))
    by = {r["qname"]: r for r in recs}
    chrom = chrom_names_from_hdr(hdr)
    n_taken = n_two = 0
    for i, nm in enumerate(names):
        h, r = got[i], by[nm]
        two = (int(seg2["flags"][i]) & 7) != 0 and np.array_equal(got[i], hits[i])
        if two and h["nar"] == 1:
            # (placed as two segments: kept, or dropped by the orphan filters that follow - Aligner.cpp:630-650, tests/test_host_filters.py - as
            # an orphaned microInDel or splice junction)
            assert r["nar"] in ("AA", "OM", "OJ"), (nm, h, seg2[i], r)
        else:
            assert helpers.NAR_TAGS[h["nar"]] == r["nar"], (nm, h, seg2[i], r)
        n_two += 1 if two else 0
        if h["nar"] != 1 or two:
            continue
        taken = not np.array_equal(got[i], hits[i])
        if taken:
            tl, tr = int(rt["left"][i]), int(rt["right"][i])
            n_taken += 1
        else:
            tl, tr = (int(seg2["match_len"][i]), int(seg2["read_ofs"][i])) if int(seg2["flags"][i]) & 8 else (0, 0)
        plus = chr(h["strand"]) == "+"
        c5, c3 = (tl, tr) if plus else (tr, tl)
        cig = (f"{c5}S" if c5 else "") + f"{int(h['match_len']) - tl - tr}M" + (f"{c3}S" if c3 else "")
        assert (chrom[h["chrom_id"] - 1], int(h["match_loci"]) + (tl if plus else tr) + 1, cig) == (r["rname"], r["pos"], r["cigar"]), (nm, h, rt[i], r)
    assert n_taken > 30 and n_two > 30, (n_taken, n_two)
