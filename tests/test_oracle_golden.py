"""Pins the CPU oracle (oracle/bk_oracle.c) against outputs of the REAL reference executable
(tests/golden/*, produced by tests/golden/make_golden.py running oracle/_ref/biokanga).
Bit-exact: NAR class of every read, and chrom / 1-based POS / strand / mismatch count of every
accepted read, across -s/-e/-Q/-m/-n settings."""
import os

import numpy as np
import pytest

import helpers

RUNS = {
    "basic": {
        "s3": dict(max_subs=3), "s0": dict(max_subs=0), "s5": dict(max_subs=5), "dflt": dict(),
        "s3e2": dict(max_subs=3, min_edit_dist=2), "s3Q1": dict(max_subs=3, align_strand=1),
        "s3Q2": dict(max_subs=3, align_strand=2), "s3m1": dict(max_subs=3, pmode=1),
        "s3m2": dict(max_subs=3, pmode=2), "s3m3": dict(max_subs=3, pmode=3),
        "s3n0": dict(max_subs=3, max_ns=0), "s3n3": dict(max_subs=3, max_ns=3),
        "s2l30": dict(max_subs=2),
    },
    "repeat": {
        "s3": dict(max_subs=3), "s3m1": dict(max_subs=3, pmode=1), "s3m3": dict(max_subs=3, pmode=3),
        "s5": dict(max_subs=5),
    },
    # reads of 15..49 and 151..2000 bases on the basic genome, -l15 -L2000 (make_golden.make_lengths)
    "lengths": {
        "s3L": dict(max_subs=3), "s5L": dict(max_subs=5), "dfltL": dict(), "s0L": dict(max_subs=0),
    },
}
MIN_LEN = {"s2l30": 30, "s3L": 15, "s5L": 15, "dfltL": 15, "s0L": 15}
MAX_LEN = {"s3L": 2000, "s5L": 2000, "dfltL": 2000, "s0L": 2000}


def expected_from_sam(fixture, tag):
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, fixture, f"{tag}.m6.sam.gz"))
    return hdr, recs


def check_hits_against_sam(names, lens, hits, recs, chrom_names, keep):
    """recs: reference -M6 records (sorted order); match them to reads by QNAME multiset order."""
    by_name = {}
    for r in recs:
        by_name.setdefault(r["qname"], []).append(r)
    assert len(recs) == len(keep), (len(recs), len(keep))
    bad = []
    for j, i in enumerate(keep):
        h = hits[j]
        cands = by_name[names[i]]
        nar = helpers.NAR_TAGS[h["nar"]]
        # duplicates of a name carry identical sequences in the fixtures -> identical outcome
        r = cands[0]
        if nar != r["nar"]:
            bad.append((names[i], "nar", nar, r["nar"]))
            continue
        if nar == "AA":
            exp_strand = "-" if (r["flag"] & 16) else "+"
            got = (chrom_names[h["chrom_id"] - 1], int(h["match_loci"]) + 1, chr(h["strand"]), int(h["match_len"]))
            exp = (r["rname"], r["pos"], exp_strand, int(r["cigar"][:-1]))
            if got != exp:
                bad.append((names[i], "loc", got, exp))
        else:
            assert r["flag"] & 4 and r["rname"] == "*" and r["pos"] == 0
    assert not bad, bad[:10]


def chrom_names_from_hdr(hdr):
    return [l.split("\t")[2][3:] for l in hdr if l.startswith("@SQ")]


@pytest.mark.parametrize("fixture,tag", [(f, t) for f in RUNS for t in RUNS[f]])
def test_oracle_matches_reference(golden_tmp, fixture, tag):
    d = golden_tmp[fixture]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    keep = helpers.filter_reads_by_len(names, bases, offs, lens, MIN_LEN.get(tag, 50), MAX_LEN.get(tag, 500))
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    p = helpers.make_params(**RUNS[fixture][tag])
    hits, ctr = sfx.align(bases, offs[keep], lens[keep], p)
    hdr, recs = expected_from_sam(fixture, tag)
    check_hits_against_sam(names, lens, hits, recs, chrom_names_from_hdr(hdr), keep)
    # NAR histogram the reference logged
    exp_counts = {}
    with open(os.path.join(helpers.GOLDEN, fixture, f"{tag}.nar.txt")) as f:
        for line in f:
            t = line.split()
            exp_counts[t[1].strip("()")] = int(t[0])
    got = np.bincount(hits["nar"], minlength=20)
    for k, tagname in enumerate(helpers.NAR_TAGS):
        assert got[k] == exp_counts[tagname], (tagname, got[k], exp_counts[tagname])
    sfx.close()


@pytest.mark.parametrize("fixture", ["basic", "repeat"])
def test_oracle_mismatch_counts(golden_tmp, fixture):
    """-M0 CSV carries the per-read Hamming score (TrimMismatches) that SAM does not."""
    d = golden_tmp[fixture]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    keep = helpers.filter_reads_by_len(names, bases, offs, lens)
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    hits, _ = sfx.align(bases, offs[keep], lens[keep], helpers.make_params(max_subs=3))
    csv = helpers.parse_m0_csv(os.path.join(helpers.GOLDEN, fixture, "s3.m0.csv.gz"))
    n_acc = 0
    for j, i in enumerate(keep):
        h = hits[j]
        if h["nar"] != 1:
            continue
        n_acc += 1
        e = csv[names[i]]
        assert e["mismatches"] == h["mismatches"] == h["low_mm"], names[i]
        assert e["start"] == h["match_loci"] and e["end"] == h["match_loci"] + h["match_len"] - 1
        assert e["strand"] == chr(h["strand"])
        assert e["read_id"] == i + 1 if len(keep) == len(names) else True
    # every CSV row belongs to an accepted read (names may repeat for the dup reads)
    assert n_acc >= len(csv)
    sfx.close()


def test_oracle_search_primitives(golden_tmp):
    """LocateFirstExact/LocateLastExact restatements return the true lowest/highest SA index of a
    k-mer (brute force over the 200 kbp fixture)."""
    import ctypes
    d = golden_tmp["basic"]
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    raw = open(os.path.join(d, "genome.sfx"), "rb").read()
    import struct
    blk_ofs = struct.unpack_from("<Q", raw, 44)[0]
    n = struct.unpack_from("<Q", raw, blk_ofs + 8)[0]
    seq = np.frombuffer(raw, dtype=np.uint8, count=n, offset=blk_ofs + 20)
    sa = np.frombuffer(raw, dtype="<u4", count=n, offset=blk_ofs + 20 + n)
    rng = np.random.default_rng(5)
    lib = helpers.oracle_lib()
    for k in (4, 6, 9, 12):
        for _ in range(20):
            p0 = int(rng.integers(0, n - 200))
            probe = np.ascontiguousarray(seq[p0:p0 + k])
            if (probe > 4).any():
                continue
            # brute force: which suffixes start with probe
            m = np.ones(n - k, dtype=bool)
            for j in range(k):
                m &= seq[j:n - k + j] == probe[j]
            starts = set(np.nonzero(m)[0].tolist())
            idxs = [i for i in range(n) if int(sa[i]) in starts] if len(starts) < 2000 else None
            first = lib.ora_locate_first_exact(sfx.h, probe.ctypes.data, k, 0, n - 1, None)
            last = lib.ora_locate_last_exact(sfx.h, probe.ctypes.data, k, 0, n - 1, None)
            assert first >= 1 and last >= first
            assert last - first + 1 == len(starts)
            if idxs is not None:
                assert first - 1 == min(idxs) and last - 1 == max(idxs)
    sfx.close()
