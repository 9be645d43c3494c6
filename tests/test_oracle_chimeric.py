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
