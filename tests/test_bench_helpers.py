"""CPU checks of bench.py's helper code (file writers of the reference leg, profile lookup) - nothing here
touches a GPU or times anything."""
import gzip
import importlib.util
import os
import struct

import numpy as np

import helpers


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(helpers.ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_sfx_writer_matches_reference_layout(golden_tmp, tmp_path):
    """bench.write_sfx_file reproduces a reference-written .sfx from its parts"""
    b = _bench()
    ref = open(os.path.join(golden_tmp["basic"], "genome.sfx"), "rb").read()
    blk = struct.unpack_from("<Q", ref, 44)[0]
    n = struct.unpack_from("<Q", ref, blk + 8)[0]
    seq = np.frombuffer(ref, np.uint8, n, blk + 20)
    sa = np.frombuffer(ref, "<u4", n, blk + 20 + n)
    out = str(tmp_path / "w.sfx")
    b.write_sfx_file(out, seq, sa, [("chrA", 100000), ("chrB", 100000)])
    got = open(out, "rb").read()
    assert len(got) == len(ref)
    assert got[blk:] == ref[blk:]                      # block and entries byte-identical
    assert got[:52] == ref[:52]                        # magic, version, sizes, offsets (names follow)
    o = helpers.OracleSfx(out)                         # and the oracle's loader accepts it
    o.close()


def test_fasta_writer_and_profile_lookup(tmp_path):
    b = _bench()
    reads = np.arange(7 * 100, dtype=np.uint8) % 5
    p = str(tmp_path / "r.fa")
    b.write_fasta_file(p, reads, 7, 100, chunk=3)
    names, bases, offs, lens = helpers.read_fasta_reads(p)
    assert names == [f"r{i:09d}" for i in range(1, 8)] and list(lens) == [100] * 7
    assert np.array_equal(bases & 7, np.where(reads > 3, 4, reads))
    t, src = b.profiled_traffic("k_wave")
    assert t is None or (t[0] > 1e9 and t[1] >= 0 and src.endswith("_pmc_summary.csv"))


def test_fetch_size_correction_follows_the_committed_calibration():
    """random-line patterns are counted exactly; the window array's streaming runs are counted at about half (profiles/*_fetch_calibration.csv)"""
    b = _bench()
    assert b.fetch_correction("k_flat", True)[0] == 1.0 and b.fetch_correction("k_wave", False)[0] == 1.0
    f, src = b.fetch_correction("k_wave", True)
    assert 1.5 < f < 2.5, (f, src)
