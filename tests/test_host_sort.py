"""Host logic on CPU: the reference-order sort replica (mtqsort.h) reproduces the record order of
the real reference's SAM for 30 000 reads with many exact ties (>= 25 000 elements, i.e. the
reference's own quicksort path), and for the small fixtures (glibc stable merge sort path)."""
import os
import subprocess

import numpy as np
import pytest

import helpers


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("h") / "sort_harness")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(helpers.ROOT, "tests", "cpp", "sort_harness.cpp")])
    return exe


def sorted_names(harness, tmp_path, hits, names):
    hp, op = str(tmp_path / "hits.bin"), str(tmp_path / "order.bin")
    hits.tofile(hp)
    subprocess.check_call([harness, hp, op])
    order = np.fromfile(op, dtype=np.uint32)
    return [names[i] for i in order if hits["nar"][i] == 1]


def test_sort_replica_quicksort_path(harness, golden_tmp, tmp_path):
    d = golden_tmp["basic"]
    rd = str(tmp_path / "reads.fa")
    helpers.gunzip_to(os.path.join(helpers.GOLDEN, "sortorder", "reads.fa.gz"), rd)
    names, bases, offs, lens = helpers.read_fasta_reads(rd)
    assert len(names) == 30000
    o = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    hits, _ = o.align(bases, offs, lens, helpers.make_params(max_subs=3), nthreads=8)
    o.close()
    got = sorted_names(harness, tmp_path, hits, names)
    _, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, "sortorder", "s3.m5.sam.gz"))
    exp = [r["qname"] for r in recs]
    assert len(got) == len(exp) > 20000
    assert got == exp


@pytest.mark.parametrize("fixture", ["basic", "repeat"])
def test_sort_replica_small(harness, golden_tmp, tmp_path, fixture):
    d = golden_tmp[fixture]
    names, bases, offs, lens = helpers.read_fasta_reads(os.path.join(d, "reads.fa"))
    keep = helpers.filter_reads_by_len(names, bases, offs, lens)
    o = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    hits, _ = o.align(bases, offs[keep], lens[keep], helpers.make_params(max_subs=3))
    o.close()
    got = sorted_names(harness, tmp_path, hits, [names[i] for i in keep])
    _, recs = helpers.parse_sam(os.path.join(helpers.GOLDEN, fixture, "s3.m5.sam.gz"))
    assert got == [r["qname"] for r in recs]
