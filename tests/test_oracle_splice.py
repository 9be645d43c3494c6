"""Splice junctions (-A) on CPU: the oracle's restatement of LocateSpliceJuncts / ExploreSpliceRight / ExploreSpliceLeft (+ the orphan
filters) against the real reference's output for tests/golden/splice: both segments of every spliced ("arj") or microInDel ("ari") read
from the -M0 CSV and the OJ / OM tags of the SAM.  Reads without a second segment are flank-trimmed by the reference in this mode (-A
switches -x on); that host step is covered by the command-line tests, here only their aligned / not aligned class is compared."""
import gzip
import os

import numpy as np
import pytest

import helpers

CASES = {"A5000": dict(max_subs=3, splice_junct_len=5000), "A500s5": dict(max_subs=5, splice_junct_len=500),
         "A5000a5": dict(max_subs=3, splice_junct_len=5000, micro_indel_len=5)}


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_splices_match_reference(golden_tmp, tag):
    d = golden_tmp["splice"]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    hits, seg2 = helpers.oracle_align_indel(sfx, bases, offs, lens, helpers.make_params(**CASES[tag]))
    sfx.close()
    helpers.remove_orphan_splices(hits, seg2)
    helpers.remove_orphan_indels(hits, seg2)
    exp = {}
    for line in gzip.open(os.path.join(helpers.GOLDEN, "splice", f"{tag}.m0.csv.gz"), "rt"):
        f = line.rstrip("\n").split(",")
        if f[1].strip('"') in ("arj", "ari"):
            exp.setdefault(f[13].strip('"'), []).append((f[1].strip('"'), f[3].strip('"'), int(f[4]), int(f[5]), int(f[6]), f[7].strip('"'), int(f[11])))
    chrom = {1: "sA", 2: "sB"}
    got = {}
    for i, nm in enumerate(names):
        h = hits[i]
        if h["nar"] != 1 or not (seg2["flags"][i] & 5):
            continue
        kind = "arj" if seg2["flags"][i] & 4 else "ari"
        st, ln, s1, l1 = int(h["match_loci"]), int(h["match_len"]), int(seg2["match_loci"][i]), int(seg2["match_len"][i])
        got[nm] = [(kind, chrom[int(h["chrom_id"])], st, st + ln - 1, ln, chr(h["strand"]), int(h["mismatches"])),
                   (kind, chrom[int(h["chrom_id"])], s1, s1 + l1 - 1, l1, chr(h["strand"]), int(seg2["mismatches"][i]))]
    assert set(got) == set(exp), (sorted(set(got) ^ set(exp))[:10])
    bad = [(k, got[k], exp[k]) for k in exp if got[k] != exp[k]]
    assert not bad, bad[:5]
    assert len(exp) > 60
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, "splice", f"{tag}.m6.sam.gz"))
    tags = {r["qname"]: r["nar"] for r in recs}
    for i, nm in enumerate(names):
        g, e = helpers.NAR_TAGS[hits["nar"][i]], tags[nm]
        assert g == e or (g == "AA" and e == "ET"), (nm, hits[i], e)
