"""SNP pile-up / screening on CPU: the oracle's restatement of CAligner::ProcessSNPs + the screening loop of OutputSNPs against the real
reference's SNP CSV of tests/golden/snp (run `-p1 -P0.4 -1 0.1`, alignments taken from the reference's own -M0 CSV): for every SNP
row the reference reported - coverage, mismatches, reference base, per-base counts, the background window totals and rate - and the
covered-loci / coverage totals of its log."""
import gzip
import os
import re

import numpy as np

import helpers
from biokanga_amd.binding import SNP_ALN_DTYPE


def _alns_from_m0(names, path):
    idx = {n: i for i, n in enumerate(names)}
    rows = helpers.parse_m0_csv(path)
    chroms = sorted({r["chrom"] for r in rows.values()})
    alns = np.zeros(len(rows), dtype=SNP_ALN_DTYPE)
    for k, (nm, r) in enumerate(rows.items()):
        alns[k]["read_idx"] = idx[nm]
        alns[k]["chrom_id"] = chroms.index(r["chrom"]) + 1
        alns[k]["loci"] = r["start"]
        alns[k]["len"] = r["length"]
        alns[k]["strand"] = ord(r["strand"])
    return alns, chroms


def test_oracle_snp_sites_match_reference_rows(golden_tmp):
    d = golden_tmp["snp"]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    alns, chroms = _alns_from_m0(names, os.path.join(helpers.GOLDEN, "snp", "p1P40n1.csv.gz"))
    assert chroms == ["sA", "sB", "sC"]
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    per = {}
    covered = coverage = 0
    for c, nm in enumerate(chroms):
        sites, tot = helpers.oracle_snp_sites(sfx.h, bases, offs, alns, c + 1, 1, 0.001)
        per[nm] = ({int(s["loci"]): s for s in sites}, tot)
        covered += int(tot[2]); coverage += int(tot[3])
    sfx.close()
    log = open(os.path.join(helpers.GOLDEN, "snp", "p1P40n1.log.txt")).read()
    m = re.search(r"There are (\d+) aligned loci bases which are covered by (\d+) read bases", log)
    assert (covered, coverage) == (int(m.group(1)), int(m.group(2)))
    n = 0
    for line in gzip.open(os.path.join(helpers.GOLDEN, "snp", "p1P40n1.snp.gz"), "rt"):
        f = line.rstrip("\n").split(",")
        if f[0].startswith('"'):
            continue
        sites, tot = per[f[3].strip('"')]
        s = sites[int(f[4])]
        nonref = int(s["non_ref"].sum())
        assert int(f[10]) == nonref + int(s["num_ref"]) and int(f[11]) == nonref, line
        assert "ACGT"[int(s["ref_base"])] == f[12].strip('"'), line
        cnt = [int(x) for x in s["non_ref"]]
        cnt[int(s["ref_base"])] = int(s["num_ref"])
        assert [int(x) for x in f[13:18]] == cnt, line
        tmm = int(s["win_mismatches"]) - nonref if nonref <= int(s["win_mismatches"]) else 0
        tm = int(s["win_matches"]) - int(s["num_ref"]) if int(s["num_ref"]) < int(s["win_matches"]) else 0
        assert (int(f[19]), int(f[20])) == (tmm + tm, tmm), line
        glob = max(0.01, int(tot[1]) / (1 + int(tot[0]) + int(tot[1])))
        rate = glob if tmm + tm == 0 else max(glob, tmm / (tmm + tm))
        assert "%f" % rate == f[18], line
        n += 1
    assert n > 300
