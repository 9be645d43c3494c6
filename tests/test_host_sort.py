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
    subprocess.check_call(helpers.cxx() + ["-pthread", "-o", exe, os.path.join(helpers.ROOT, "tests", "cpp", "sort_harness.cpp")])
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


def test_sort_threads_do_not_change_the_order(harness, tmp_path):
    """the multi-threaded form hands whole sub-partitions to other threads, like the reference's CMTqsort:
    the order of tied records must be the single-threaded one"""
    import biokanga_amd.binding as b
    rng = np.random.default_rng(11)
    n = 2_500_000                                            # (spans of 1 M elements and more are partitioned by all threads together)
    hits = np.zeros(n, dtype=b.HIT_DTYPE)
    hits["nar"] = rng.choice([1, 1, 1, 4, 6], n)
    hits["num_hits"] = (hits["nar"] == 1).astype(np.uint8)
    hits["chrom_id"] = rng.integers(1, 4, n)
    hits["match_loci"] = rng.integers(0, 20000, n)          # many exact ties
    hits["match_len"] = 100
    hits["strand"] = rng.choice([ord("+"), ord("-")], n)
    hits["low_mm"] = rng.integers(0, 3, n)
    hp = str(tmp_path / "hits.bin")
    hits.tofile(hp)
    orders = []
    for t in (1, 8, 3):
        op = str(tmp_path / f"order{t}.bin")
        subprocess.check_call([harness, hp, op, str(t)])
        orders.append(np.fromfile(op, dtype=np.uint32))
    assert np.array_equal(orders[0], orders[1]) and np.array_equal(orders[0], orders[2])
    srt = hits[orders[0]]
    assert np.all(np.diff(srt["nar"].astype(int)) >= 0)


def test_glibc_rand_replica(tmp_path):
    """glibc_rand.h yields the C library's unseeded rand() sequence (the reference's N-run mutations and -r2 picks)"""
    exe = str(tmp_path / "rand_harness")
    subprocess.check_call(helpers.cxx() + ["-o", exe, os.path.join(helpers.ROOT, "tests", "cpp", "rand_harness.cpp")])
    assert subprocess.check_output([exe, "200000"]).strip() == b"0"
