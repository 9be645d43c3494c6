"""Host filters on CPU (host/post_filters.h): the flank trimmer (-x, forced by -A) and the orphan junction filters, fed with the
oracle's records, against what the real reference wrote: POS and CIGAR (soft clips, I/D/N) of every aligned read and the NAR tag
of every other read of the -M6 SAM files of tests/golden/{basic,indel,splice}."""
import os
import struct
import subprocess

import numpy as np
import pytest

import helpers
from test_oracle_golden import MIN_LEN, MAX_LEN


@pytest.fixture(scope="module")
def filters_harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fh") / "filters_harness")
    subprocess.check_call(helpers.cxx() + ["-pthread", "-o", exe, os.path.join(helpers.ROOT, "tests", "cpp", "filters_harness.cpp")])
    return exe


def sfx_target(path):
    """(concatenated bases uint8, [start offset per entry]) of a .sfx file"""
    raw = open(path, "rb").read()
    blk = struct.unpack_from("<Q", raw, 44)[0]
    n = struct.unpack_from("<Q", raw, blk + 8)[0]
    seq = np.frombuffer(raw, dtype=np.uint8, count=n, offset=blk + 20).copy()
    ent = struct.unpack_from("<Q", raw, 20)[0]
    n_ent = struct.unpack_from("<I", raw, ent)[0]
    starts = [struct.unpack_from("<Q", raw, ent + 8 + i * 111 + 95)[0] for i in range(n_ent)]
    return seq, np.array(starts, dtype=np.uint64)


def expected_cigar(h, s2, tl, tr, read_len):
    ln = int(h["match_len"]) - tl - tr
    c5, c3 = (tl, tr) if chr(h["strand"]) == "+" else (tr, tl)
    cig = (f"{c5}S" if c5 else "") + f"{ln}M" + (f"{c3}S" if c3 else "")
    if s2["flags"] & 5:
        gap = int(s2["match_loci"]) - (int(h["match_loci"]) + int(h["match_len"]))
        if s2["flags"] & 4:
            cig += f"{gap}N"
        elif s2["flags"] & 2:
            cig += f"{read_len - (int(h['match_len']) + int(s2['match_len']))}I"
        else:
            cig += f"{abs(gap)}D"
        cig += f"{int(s2['match_len'])}M"
    return cig


CASES = [("basic", "s3x5", dict(max_subs=3), 5), ("basic", "s10x6", dict(max_subs=10), 6), ("indel", "a10x4", dict(max_subs=3, micro_indel_len=10), 4),
         ("indel", "a10", dict(max_subs=3, micro_indel_len=10), 0), ("splice", "A5000", dict(max_subs=3, splice_junct_len=5000), 3),
         ("splice", "A5000a5", dict(max_subs=3, splice_junct_len=5000, micro_indel_len=5), 3), ("splice", "A500s5", dict(max_subs=5, splice_junct_len=500), 5),
         # -a / -A / -c in one run: each search hands its leftover LowHitInstances / LowMMCnt on to the next (with -c there is no forced trimming)
         ("combined", "a10c50", dict(max_subs=3, micro_indel_len=10, min_chimeric_len=50), 0),
         ("combined", "a10A5000c50", dict(max_subs=3, micro_indel_len=10, splice_junct_len=5000, min_chimeric_len=50), 0),
         ("combined", "A5000c60", dict(max_subs=3, splice_junct_len=5000, min_chimeric_len=60), 0),
         ("combined", "a10A5000", dict(max_subs=3, micro_indel_len=10, splice_junct_len=5000), 3)]


@pytest.mark.parametrize("fixture,tag,kw,min_flank", CASES)
def test_trims_and_orphan_filters_match_reference(golden_tmp, filters_harness, tmp_path, fixture, tag, kw, min_flank):
    d = golden_tmp[fixture]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    keep = helpers.filter_reads_by_len(names, bases, offs, lens, 50, 500)
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    hits, seg2 = helpers.oracle_align_indel(sfx, bases, offs[keep], lens[keep], helpers.make_params(**kw))
    sfx.close()
    seq, starts = sfx_target(os.path.join(d, "genome.sfx"))
    p = {k: str(tmp_path / (k + ".bin")) for k in ("hits", "seg2", "bases", "offs", "seq", "ent", "out", "trims")}
    hits.tofile(p["hits"]); seg2.tofile(p["seg2"]); np.ascontiguousarray(bases, dtype=np.uint8).tofile(p["bases"])
    np.ascontiguousarray(offs[keep], dtype=np.uint64).tofile(p["offs"]); seq.tofile(p["seq"]); starts.tofile(p["ent"])
    subprocess.check_call([filters_harness, str(min_flank), "0", "1" if kw.get("splice_junct_len") else "0", "1" if kw.get("micro_indel_len") else "0",
                           p["hits"], p["seg2"], p["bases"], p["offs"], p["seq"], p["ent"], p["out"], p["trims"]])
    got = np.fromfile(p["out"], dtype=helpers.HIT_DTYPE)
    trims = np.fromfile(p["trims"], dtype="<u2").reshape(-1, 3)
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, fixture, f"{tag}.m6.sam.gz"))
    by = {r["qname"]: r for r in recs}
    chrom = [l.split("\t")[2][3:] for l in hdr if l.startswith("@SQ")]
    n_clip = 0
    for j, i in enumerate(keep):
        r, h = by[names[i]], got[j]
        assert helpers.NAR_TAGS[h["nar"]] == r["nar"], (names[i], h, r)
        if h["nar"] != 1:
            continue
        tl, tr = int(trims[j][0]), int(trims[j][1])
        if seg2["flags"][j] & 8:                                    # chimeric placement: its own trims
            tl, tr = int(seg2["match_len"][j]), int(seg2["read_ofs"][j])
        start = int(h["match_loci"]) + (tl if chr(h["strand"]) == "+" else tr)
        assert (chrom[h["chrom_id"] - 1], start + 1) == (r["rname"], r["pos"]), (names[i], h, trims[j], r)
        assert expected_cigar(h, seg2[j], tl, tr, int(lens[i])) == r["cigar"], (names[i], h, seg2[j], trims[j], r)
        n_clip += (tl + tr) > 0
    if min_flank:
        assert n_clip > 20


@pytest.mark.parametrize("tag,win", [("k0", 0), ("k20", 20), ("k200", 200)])
def test_pcr_artefact_reduction_matches_reference(golden_tmp, filters_harness, tmp_path, tag, win):
    """-k: the oracle's records of the 30 000 stacked sortorder reads -> reference-order sort replica -> ReducePCRduplicates restatement;
    every read's NAR class (AA / DP / ...) against the reference's SAM"""
    d = golden_tmp["basic"]
    rd = str(tmp_path / "reads.fa")
    helpers.gunzip_to(os.path.join(helpers.GOLDEN, "sortorder", "reads.fa.gz"), rd)
    names, bases, offs, lens = helpers.read_fasta_reads(rd)
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    hits, _ = sfx.align(bases, offs, lens, helpers.make_params(max_subs=3), nthreads=8)
    sfx.close()
    hp, op = str(tmp_path / "h.bin"), str(tmp_path / "o.bin")
    hits.tofile(hp)
    subprocess.check_call([filters_harness, "pcr", str(win), hp, op])
    got = np.fromfile(op, dtype=helpers.HIT_DTYPE)
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, "sortorder", f"s3{tag}.m6.sam.gz"))
    tags = {r["qname"]: r["nar"] for r in recs}
    bad = [(nm, helpers.NAR_TAGS[got["nar"][i]], tags[nm]) for i, nm in enumerate(names) if helpers.NAR_TAGS[got["nar"][i]] != tags[nm]]
    assert not bad, (len(bad), bad[:5])
    assert np.count_nonzero(got["nar"] == 9) > 15000
