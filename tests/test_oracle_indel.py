"""microInDels (-a) on CPU: the oracle's restatement of LocateInDels / ExploreInDelMatchRight / ExploreInDelMatchLeft (+ the orphan
filter) against what the real reference reported for tests/golden/indel: both segments of every read from the -M0 CSV (sequence,
start, end, length, strand, mismatches) and the NAR tag of every unaligned read from the -M6 SAM."""
import gzip
import os

import numpy as np
import pytest

import helpers

CASES = {"a10": dict(max_subs=3, micro_indel_len=10), "a3s5": dict(max_subs=5, micro_indel_len=3),
         "a20Q1": dict(max_subs=3, micro_indel_len=20, align_strand=1)}


def reference_rows(tag):
    rows = {}
    for line in gzip.open(os.path.join(helpers.GOLDEN, "indel", f"{tag}.m0.csv.gz"), "rt"):
        f = line.rstrip("\n").split(",")
        rows.setdefault(f[13].strip('"'), []).append((f[1].strip('"'), f[3].strip('"'), int(f[4]), int(f[5]), int(f[6]), f[7].strip('"'), int(f[11])))
    return rows


def oracle_rows(names, hits, seg2, chrom):
    out = {}
    for i, nm in enumerate(names):
        h = hits[i]
        if h["nar"] != 1:
            continue
        st, ln = int(h["match_loci"]), int(h["match_len"])
        kind = "ari" if seg2["flags"][i] & 1 else "ar"
        segs = [(kind, chrom[int(h["chrom_id"])], st, st + ln - 1, ln, chr(h["strand"]), int(h["mismatches"]))]
        if seg2["flags"][i] & 1:
            s1, l1 = int(seg2["match_loci"][i]), int(seg2["match_len"][i])
            segs.append((kind, chrom[int(h["chrom_id"])], s1, s1 + l1 - 1, l1, chr(h["strand"]), int(seg2["mismatches"][i])))
        out[nm] = segs
    return out


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_indels_match_reference(golden_tmp, tag):
    d = golden_tmp["indel"]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    hits, seg2 = helpers.oracle_align_indel(sfx, bases, offs, lens, helpers.make_params(**CASES[tag]))
    sfx.close()
    n_put = int(np.count_nonzero(seg2["flags"] & 1))
    helpers.remove_orphan_indels(hits, seg2)
    exp = reference_rows(tag)
    got = oracle_rows(names, hits, seg2, {1: "iA", 2: "iB"})
    assert set(got) == set(exp)
    bad = [(k, got[k], exp[k]) for k in exp if got[k] != exp[k]]
    assert not bad, bad[:5]
    assert sum(1 for v in exp.values() if len(v) == 2) > 50 and n_put > sum(1 for v in exp.values() if len(v) == 2)
    # NAR tags of everything else
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, "indel", f"{tag}.m6.sam.gz"))
    tags = {r["qname"]: r["nar"] for r in recs}
    for i, nm in enumerate(names):
        assert helpers.NAR_TAGS[hits["nar"][i]] == tags[nm], (nm, hits[i], tags[nm])
