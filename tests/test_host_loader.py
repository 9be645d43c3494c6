"""The read store of `biokanga align` (host/read_loader.cpp; CAligner::LoadRawReads, biokanga/Aligner.cpp:10724-11427): what all
threads make of whole-file parses must be what the record-by-record loops make - reads, names, bases, order and log lines.  CPU only."""
import os
import re
import subprocess

import numpy as np
import pytest

import helpers
from test_host_fasta import write_bgzf


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("l") / "loader_harness")
    host = os.path.join(helpers.ROOT, "biokanga_amd", "csrc", "host")
    src = [os.path.join(helpers.ROOT, "tests", "cpp", "loader_harness.cpp"), os.path.join(host, "read_loader.cpp"), os.path.join(host, "fasta.cpp"),
           os.path.join(host, "fast_inflate.cpp")]
    subprocess.check_call(helpers.cxx() + ["-pthread", "-o", exe] + src + ["-lz"])
    return exe


def reads_file(path, n, seed, fastq=False, sim=False, lo=20, hi=300, lens=None):
    """records whose lengths straddle the acceptance window, names with blanks, names longer than the 79 / 127 characters that are kept"""
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)
    with open(path, "wb") as f:
        for i in range(n):
            L = int(lens[i]) if lens is not None else int(rng.integers(lo, hi))
            name = (b"lcl|usimreads|%d|chr1|%d|+|%d" % (i, i * 7, L)) if sim else b"r%d" % i
            if i % 17 == 3:
                name += b"x" * 90
            if i % 19 == 5:
                name += b" tail words " + b"y" * 140
            if i % 23 == 7:
                name += b"\tafter a tab"
            seq = letters[rng.integers(0, len(letters), L)].tobytes()
            if fastq:
                f.write(b"@" + name + b"\n" + seq + b"\n+\n" + rng.integers(33, 105, L, dtype=np.uint8).tobytes() + b"\n")
            else:
                f.write(b">" + name + b"\n" + b"\n".join(seq[o:o + 70] for o in range(0, L, 70)) + b"\n")


def both(harness, args, whole=None):
    """(result line, log lines without their time stamps) of the one-thread and the eight-thread load"""
    res = []
    for T in ("1", "8"):
        r = subprocess.run([harness, args[0], T] + [str(a) for a in args[1:]], capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
        if whole is not None:                              # files the all-thread path took: none with one thread, `whole` with eight
            assert r.stderr.decode().split("whole")[1].split()[0] == (str(whole) if T == "8" else "0"), (T, r.stderr)
        lines = r.stdout.decode(errors="replace").splitlines()
        res.append(("".join(l for l in lines if not l.startswith("[")), [re.sub(r"^\[[^\]]*\]", "", l) for l in lines if l.startswith("[")]))
    return res


@pytest.mark.parametrize("fastq,qmode,sim", [(False, 3, False), (True, 3, False), (True, 0, True), (False, 3, True)])
def test_single_end_store_by_all_threads_equals_the_serial_loaders(harness, tmp_path, fastq, qmode, sim):
    ext = "fq" if fastq else "fa"
    a, b = str(tmp_path / f"a.{ext}"), str(tmp_path / f"b.{ext}")
    reads_file(a, 14000, 1, fastq, sim)
    reads_file(b, 9000, 2, fastq, sim)
    assert os.path.getsize(a) > (1 << 20) and os.path.getsize(b) > (1 << 20)
    for trims in ((0, 0), (5, 3)):
        one, many = both(harness, ["se", trims[0], trims[1], 50, 250, qmode, 1, a, b], whole=2)       # (second file: appended to a store that holds reads)
        assert one[0].startswith("reads ") and one == many, (one, many)
        assert any("under length" in l for l in one[1]) and any("over length" in l for l in one[1])


def test_bgzip_and_sampled_inputs_give_the_plain_files_store(harness, tmp_path):
    a = str(tmp_path / "a.fq")
    reads_file(a, 14000, 3, fastq=True)
    write_bgzf(a + ".bgz", open(a, "rb").read())
    plain = both(harness, ["se", 0, 0, 50, 250, 1, 1, a], whole=1)
    packed = both(harness, ["se", 0, 0, 50, 250, 1, 1, a + ".bgz"], whole=1)
    assert plain[0][0] == plain[1][0] == packed[0][0] == packed[1][0], (plain, packed)
    one, many = both(harness, ["se", 0, 0, 50, 250, 1, 3, a], whole=0)                          # -# 3: every third read; the serial loop either way
    assert one == many and one[0] != plain[0][0]


@pytest.mark.parametrize("fastq,extra_b", [(False, 0), (True, 250)])
def test_paired_store_by_all_threads_equals_the_serial_loader(harness, tmp_path, fastq, extra_b):
    """mates of different lengths, either of which may fail either rule; a second file with surplus records"""
    ext = "fq" if fastq else "fa"
    files = []
    for k in range(2):
        a, b = str(tmp_path / f"a{k}.{ext}"), str(tmp_path / f"b{k}.{ext}")
        reads_file(a, 12000, 10 + k, fastq)
        reads_file(b, 12000 + extra_b, 20 + k, fastq)
        files += [a, b]
    for trims in ((0, 0), (4, 6)):
        one, many = both(harness, ["pe", trims[0], trims[1], 50, 250, 3 if not fastq else 2, 1] + files, whole=4)
        assert one[0].startswith("reads ") and int(one[0].split()[1]) % 2 == 0 and one == many, (one, many)


def test_a_short_second_mate_file_is_reported_as_the_serial_loader_does(harness, tmp_path):
    a, b = str(tmp_path / "a.fa"), str(tmp_path / "b.fa")
    reads_file(a, 12000, 31)
    reads_file(b, 11000, 32)
    one, many = both(harness, ["pe", 0, 0, 50, 250, 3, 1, a, b], whole=0)
    assert one == many and one[0].startswith("rc -63") and any("fewer reads" in l for l in one[1]), (one, many)


def test_gzip_mate_files_opened_side_by_side_give_the_plain_files_store(harness, tmp_path):
    """mates as one gzip member each, and bgzip'd: inflated whole, side by side, accepted by all threads - the plain files' store"""
    import gzip
    a, b = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    reads_file(a, 12000, 41, fastq=True)
    reads_file(b, 12000, 42, fastq=True)
    for f in (a, b):
        data = open(f, "rb").read()
        open(f + ".gz", "wb").write(gzip.compress(data, 4))
        write_bgzf(f + ".bgz", data)
    plain = both(harness, ["pe", 3, 2, 50, 250, 0, 1, a, b], whole=2)
    gz = both(harness, ["pe", 3, 2, 50, 250, 0, 1, a + ".gz", b + ".gz"], whole=2)
    bgz = both(harness, ["pe", 3, 2, 50, 250, 0, 1, a + ".bgz", b + ".gz"], whole=2)
    assert plain[0][0] == plain[1][0] == gz[0][0] == gz[1][0] == bgz[0][0] == bgz[1][0], (plain, gz, bgz)


def test_a_fifo_as_read_file_is_opened_once_and_read_to_its_end(harness, tmp_path):
    """-i names a FIFO (or a process substitution's pipe): it is not opened to be looked at first - its writer would lose its reader and
    die of SIGPIPE - and the record-by-record reader takes it: the plain file's store, the writer ends normally"""
    a, fifo = str(tmp_path / "a.fq"), str(tmp_path / "in.fifo")
    reads_file(a, 14000, 51, fastq=True)
    want = both(harness, ["se", 0, 0, 50, 250, 0, 1, a], whole=1)[0][0]
    os.mkfifo(fifo)
    for T in ("1", "8"):
        writer = subprocess.Popen(["sh", "-c", f"cat '{a}' > '{fifo}'"])
        r = subprocess.run([harness, "se", T, "0", "0", "50", "250", "0", "1", fifo], capture_output=True, timeout=300)
        assert writer.wait(timeout=60) == 0
        lines = [l for l in r.stdout.decode().splitlines() if not l.startswith("[")]
        assert r.returncode == 0 and "".join(lines) == want, (r.stdout[-500:], r.stderr[-500:])
