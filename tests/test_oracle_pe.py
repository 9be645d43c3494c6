"""Pins the oracle's paired-end restatement (ProcessPairedEnds, AlignPairedRead, AdaptiveTrim,
IterateExactsRange) against the real reference's PE outputs (tests/golden/pe/*, -U1..4, -E, default
and >= 1000 bp windows)."""
import os

import numpy as np
import pytest

import helpers

PE_RUNS = {
    "U3": dict(pe=3, d=200, D=400, s=5), "U1": dict(pe=1, d=200, D=400, s=5), "U2": dict(pe=2, d=200, D=400, s=5),
    "U4": dict(pe=4, d=200, D=400, s=5), "U3dflt": dict(pe=3, d=100, D=1000, s=3), "U3wide": dict(pe=3, d=150, D=1500, s=5),
    "U3E": dict(pe=3, d=200, D=400, s=5, E=True),
}


# C3 of SURVEY.md 8(d): 2 x 150 bp (MaxTotMM 8, 16-mer cores, 9 cores per strand) through AlignPairedRead's window scan
PE150_RUNS = {"U3": dict(pe=3, d=200, D=400, s=5), "U1": dict(pe=1, d=200, D=400, s=5), "U2": dict(pe=2, d=200, D=400, s=5),
              "U4": dict(pe=4, d=200, D=400, s=5)}
ALL_PE_RUNS = [("pe", t) for t in PE_RUNS] + [("pe150", t) for t in PE150_RUNS]


def pe_cfg(fixture, tag):
    return (PE_RUNS if fixture == "pe" else PE150_RUNS)[tag]


def pe_inputs(tmp_path, fixture="pe"):
    r1, r2 = str(tmp_path / "r1.fa"), str(tmp_path / "r2.fa")
    helpers.gunzip_to(os.path.join(helpers.GOLDEN, fixture, "reads_1.fa.gz"), r1)
    helpers.gunzip_to(os.path.join(helpers.GOLDEN, fixture, "reads_2.fa.gz"), r2)
    return helpers.interleave_pe(r1, r2)


def check_pe_hits_against_sam(names, hits, tag, chrom_names, fixture="pe"):
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, fixture, f"{tag}.m6.sam.gz"))
    by = {r["qname"]: r for r in recs}
    assert len(by) == len(recs) == len(names)
    bad = []
    for i, nm in enumerate(names):
        r = by[nm]
        flag, pos, rnext, pnext, tlen = helpers.expected_pe_sam_fields(hits, i)
        nar = helpers.NAR_TAGS[hits[i]["nar"]]
        got = (flag, pos, rnext, pnext, tlen, nar, chrom_names[hits[i]["chrom_id"] - 1] if nar == "AA" else "*")
        exp = (r["flag"], r["pos"], r["rnext"], r["pnext"], r["tlen"], r["nar"], r["rname"])
        if got != exp:
            bad.append((nm, got, exp))
    assert not bad, (len(bad), bad[:8])


@pytest.mark.parametrize("fixture,tag", ALL_PE_RUNS)
def test_oracle_pe_matches_reference(golden_tmp, tmp_path, fixture, tag):
    cfg = pe_cfg(fixture, tag)
    names, bases, offs, lens = pe_inputs(tmp_path, fixture)
    o = helpers.OracleSfx(os.path.join(golden_tmp["basic"], "genome.sfx"))
    p = helpers.make_params(max_subs=cfg["s"])
    hits, _ = o.align(bases, offs, lens, p, nthreads=8)
    helpers.oracle_process_pe(o, p, cfg["pe"], cfg["d"], cfg["D"], cfg.get("E", False), bases, offs, lens, hits)
    check_pe_hits_against_sam(names, hits, tag, ["chrA", "chrB"], fixture)
    exp = {}
    with open(os.path.join(helpers.GOLDEN, fixture, f"{tag}.nar.txt")) as f:
        for line in f:
            t = line.split()
            exp[t[1].strip("()")] = int(t[0])
    got = np.bincount(hits["nar"], minlength=20)
    for k, tg in enumerate(helpers.NAR_TAGS):
        assert got[k] == exp[tg], (tg, got[k], exp[tg])
    o.close()


# -Z / -z together with -U (round 4): the reference consults the filters INSIDE its pair rules (AcceptThisChromID, Aligner.cpp:2771-2786,
# 3224,3323,3445,3462), then runs FiltByChroms over what is left (:4019-4120) - four reference runs on the pe fixture
PE_FILT_RUNS = {"U3ZchrB": dict(pe=3, Z=["chrB"]), "U2zchrA": dict(pe=2, z=["chra"]), "U4ZchrA": dict(pe=4, Z=["chrA$"]), "U1ZchrB": dict(pe=1, Z=["chrB"])}


def filt_by_chroms(hits, names, exclude, include):
    """CAligner::FiltByChroms on the records the pair rules left: a sequence stays if an include expression matches, or - without include
    expressions - if no exclude expression does (include first: not AcceptThisChromID's order)"""
    import re
    keep = [True] * (len(names) + 1)
    for i, nm in enumerate(names):
        ok = any(re.search(e, nm, re.I) for e in include)
        if not ok and not include:
            ok = not any(re.search(e, nm, re.I) for e in exclude)
        keep[i + 1] = ok
    for h in hits:
        if h["nar"] == 1 and not keep[h["chrom_id"]]:
            h["nar"] = 11
            h["num_hits"] = 0
            h["low_hit_instances"] = 0


@pytest.mark.parametrize("tag", sorted(PE_FILT_RUNS))
def test_oracle_pe_with_chromosome_filters_matches_reference(golden_tmp, tmp_path, tag):
    cfg = PE_FILT_RUNS[tag]
    names, bases, offs, lens = pe_inputs(tmp_path, "pe")
    o = helpers.OracleSfx(os.path.join(golden_tmp["basic"], "genome.sfx"))
    p = helpers.make_params(max_subs=5)
    hits, _ = o.align(bases, offs, lens, p, nthreads=8)
    chroms = ["chrA", "chrB"]
    accept = helpers.chrom_accept_table(chroms, exclude=cfg.get("Z", ()), include=cfg.get("z", ()))
    helpers.oracle_process_pe(o, p, cfg["pe"], 200, 400, False, bases, offs, lens, hits, accept=accept)
    filt_by_chroms(hits, chroms, cfg.get("Z", ()), cfg.get("z", ()))
    check_pe_hits_against_sam(names, hits, tag, chroms, "pe")
    exp = {}
    with open(os.path.join(helpers.GOLDEN, "pe", f"{tag}.nar.txt")) as f:
        for line in f:
            t = line.split()
            exp[t[1].strip("()")] = int(t[0])
    got = np.bincount(hits["nar"], minlength=20)
    for k, tg in enumerate(helpers.NAR_TAGS):
        assert got[k] == exp[tg], (tg, got[k], exp[tg])
    o.close()


# -c together with -U (tests/golden/pechim: the chimeric fixture's genome, a third of the mates with foreign ends): the pair rules on
# end-trimmed loci, partners recovered end-trimmed
PECHIM_RUNS = {"U3c50": dict(pe=3, d=200, D=400, s=3, c=50), "U1c60": dict(pe=1, d=200, D=400, s=3, c=60), "U4c50": dict(pe=4, d=200, D=400, s=3, c=50),
               "U2c70s5": dict(pe=2, d=200, D=400, s=5, c=70), "U3c50wide": dict(pe=3, d=150, D=1500, s=3, c=50), "U3": dict(pe=3, d=200, D=400, s=3, c=0)}


def check_pechim_against_sam(names, hits, seg2, tag):
    hdr, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, "pechim", f"{tag}.m6.sam.gz"))
    by = {r["qname"]: r for r in recs}
    assert len(by) == len(recs) == len(names)
    chrom_names = ["cA", "cB"]
    bad = []
    for i, nm in enumerate(names):
        r = by[nm]
        flag, pos, rnext, pnext, tlen, cigar = helpers.expected_pe_sam_fields(hits, i, seg2)
        nar = helpers.NAR_TAGS[hits[i]["nar"]]
        got = (flag, pos, rnext, pnext, tlen, nar, chrom_names[hits[i]["chrom_id"] - 1] if nar == "AA" else "*", cigar if nar == "AA" else None)
        exp = (r["flag"], r["pos"], r["rnext"], r["pnext"], r["tlen"], r["nar"], r["rname"], r["cigar"] if r["nar"] == "AA" else None)
        if got != exp:
            bad.append((nm, got, exp))
    assert not bad, (len(bad), bad[:8])


@pytest.mark.parametrize("tag", sorted(PECHIM_RUNS))
def test_oracle_pe_with_chimeric_trimming_matches_reference(tmp_path, tag):
    cfg = PECHIM_RUNS[tag]
    names, bases, offs, lens = pe_inputs(tmp_path, "pechim")
    sfx = str(tmp_path / "genome.sfx")
    helpers.gunzip_to(os.path.join(helpers.GOLDEN, "chimeric", "genome.sfx.gz"), sfx)
    o = helpers.OracleSfx(sfx)
    p = helpers.make_params(max_subs=cfg["s"], min_chimeric_len=cfg["c"])
    hits, seg2 = helpers.oracle_align_indel(o, bases, offs, lens, p, nthreads=8)
    helpers.oracle_process_pe(o, p, cfg["pe"], cfg["d"], cfg["D"], False, bases, offs, lens, hits, seg2)
    check_pechim_against_sam(names, hits, seg2, tag)
    o.close()
