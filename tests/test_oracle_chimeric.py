"""Chimeric trimming (-c) on CPU: the oracle's chimeric form of LocateCoreMultiples (every candidate end-trimmed by AdaptiveTrim, the
longest then cleanest trimmed placement wins) against the real reference's -M6 SAM of tests/golden/chimeric: NAR tag of every read,
POS and CIGAR (soft clips) of every aligned one, and the trimmed mismatch count from the -M0 CSV."""
import gzip
import os

import numpy as np
import pytest

import helpers

CASES = {"c50": dict(max_subs=3, min_chimeric_len=50), "c70s5": dict(max_subs=5, min_chimeric_len=70), "c60e2": dict(max_subs=3, min_chimeric_len=60, min_edit_dist=2)}


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_chimeric_matches_reference(golden_tmp, tag):
    d = golden_tmp["chimeric"]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    hits, seg2 = helpers.oracle_align_indel(sfx, bases, offs, lens, helpers.make_params(**CASES[tag]))
    sfx.close()
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, "chimeric", f"{tag}.m6.sam.gz"))
    by = {r["qname"]: r for r in recs}
    chrom = [l.split("\t")[2][3:] for l in hdr if l.startswith("@SQ")]
    mm = {}
    for line in gzip.open(os.path.join(helpers.GOLDEN, "chimeric", f"{tag}.m0.csv.gz"), "rt"):
        f = line.rstrip("\n").split(",")
        mm[f[13].strip('"')] = int(f[11])
    n_chim = 0
    for i, nm in enumerate(names):
        h, r = hits[i], by[nm]
        assert helpers.NAR_TAGS[h["nar"]] == r["nar"], (nm, h, seg2[i], r)
        if h["nar"] != 1:
            continue
        tl, tr = (int(seg2["match_len"][i]), int(seg2["read_ofs"][i])) if seg2["flags"][i] & 8 else (0, 0)
        n_chim += 1 if seg2["flags"][i] & 8 else 0
        plus = chr(h["strand"]) == "+"
        start = int(h["match_loci"]) + (tl if plus else tr)
        c5, c3 = (tl, tr) if plus else (tr, tl)
        cig = (f"{c5}S" if c5 else "") + f"{int(h['match_len']) - tl - tr}M" + (f"{c3}S" if c3 else "")
        assert (chrom[h["chrom_id"] - 1], start + 1, cig) == (r["rname"], r["pos"], r["cigar"]), (nm, h, seg2[i], r)
        assert int(h["mismatches"]) == mm[nm], (nm, h, mm[nm])
    assert n_chim > 200


CHIMML = {"r5R5c50": dict(max_subs=3, min_chimeric_len=50, max_ml=5), "r5R8c55e2": dict(max_subs=3, min_chimeric_len=55, max_ml=8, min_edit_dist=2),
          "r5R3Xc50": dict(max_subs=3, min_chimeric_len=50, max_ml=3, clamp_ml=1)}


def chimml_rows(tag):
    """records of a reference -r5 run in creation order: (chrom, AdjStartLoci, AdjHitLen, strand, mismatches, read name)"""
    if os.path.exists(os.path.join(helpers.GOLDEN, "chimml", f"{tag}.m0.csv.gz")):
        rows = []
        for line in gzip.open(os.path.join(helpers.GOLDEN, "chimml", f"{tag}.m0.csv.gz"), "rt"):
            f = line.rstrip("\n").split(",")
            rows.append((int(f[0]), f[3].strip('"'), int(f[4]), int(f[6]), f[7].strip('"'), int(f[11]), f[13].strip('"')))
        rows.sort()
        assert [r[0] for r in rows] == list(range(1, len(rows) + 1))
        return [r[1:] for r in rows], True
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, "chimml", f"{tag}.m6.sam.gz"))
    rows = []
    for r in recs:
        if r["nar"] != "AA":
            continue
        import re
        m = sum(int(n) for n, op in re.findall(r"(\d+)([MS])", r["cigar"]) if op == "M")
        rows.append((r["rname"], r["pos"] - 1, m, "-" if r["flag"] & 16 else "+", None, r["qname"]))
    return rows, False


@pytest.mark.parametrize("tag", sorted(CHIMML))
def test_oracle_chimeric_loci_lists_match_reference_r5(golden_tmp, tag):
    """-c with the multi-loci modes: the chimeric call made with MaxHits = -R; every locus of a read's list, with its own end trims, against
    the records the reference wrote with -r5 (CSV: in creation order, so the discovery order is pinned; SAM of the -X run: as a set)"""
    d = golden_tmp["chimml"]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    hits, lo, loci, trims, seg2 = helpers.oracle_align_multi_chimeric(sfx, bases, offs, lens, helpers.make_params(**CHIMML[tag]), nthreads=8)
    sfx.close()
    rows, ordered = chimml_rows(tag)
    chrom = {1: "mA", 2: "mB"}
    exp = []
    for i, nm in enumerate(names):
        for j in range(int(lo[i]), int(lo[i + 1])):
            L, T = loci[j], trims[j]
            plus = chr(L["strand"]) == "+"
            tl, tr = int(T["left"]), int(T["right"])
            exp.append((chrom[int(L["chrom_id"])], int(L["match_loci"]) + (tl if plus else tr), int(L["match_len"]) - tl - tr, chr(L["strand"]),
                        int(L["mismatches"]) if ordered else None, nm))
    if ordered:
        assert exp == rows
    else:
        assert sorted(exp) == sorted(rows)
    multi_chim = sum(1 for i in range(len(names)) if lo[i + 1] - lo[i] > 1 and trims["chimeric"][int(lo[i])])
    assert multi_chim > 30, multi_chim
