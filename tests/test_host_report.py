"""The CSV (-M0..3) and BED (-M4, with .ind / .jct) writers of `biokanga align` (host/report.cpp; CAligner::WriteReadHits,
Aligner.cpp:6336-6660): the lines are made by all threads and written in order.  A made-up result set (tests/cpp/report_harness.cpp)
must give, for any thread count, the files the one-thread loop gave before the threads came - their MD5 sums are kept here - and the
same text through a .gz name.  CPU only (the harness links the library but never touches a device)."""
import gzip
import hashlib
import os
import subprocess

import pytest

import helpers

# 300 000 records, seeds 1 and 2, from the record-by-record writer (commit fbaac2d)
WANT = {
    (0, 1): "fd1d4d573531daaa4e3ce6f12ad9a2a9", (0, 2): "fa4656685350d12ea8cdfeec61e5e78d",
    (1, 1): "316a025f2ada366e814ea80fe245f290", (1, 2): "f843dbd7ab6d66cf08e06a5f33b2d71a",
    (2, 1): "9e47879b64bc7ac6ffbbb3d29638c344", (2, 2): "2933c37b75e3a865aa772f6155d9e5bb",
    (3, 1): "36610765f93bce2737b8f4a673ab24bc", (3, 2): "9cbd69215e16e9d8f441b0448c256bfe",
    (4, 1): "3a8cc270c45ab77cf031f8f5b0999975", (4, 2): "31bf8e0919dfd19ee0ec3f23c2886ae1",
    (4, 1, "ind"): "61580de94d80683401e3e2efd01db637", (4, 2, "ind"): "4de8b2624f444731277a3a08fdaf5c28",
    (4, 1, "jct"): "438cdf4827c65af9d32eed5bff04ab33", (4, 2, "jct"): "a976c8b4b1e549417fef90d99cf0c010",
    # the host's SAM formatter (-M5, -M6) on the same records, as of the commit whose GPU suite checked it against the reference's files
    (5, 1): "8a4dca53038dc710eb2ddd266c68f8bf", (6, 1): "055b2a2a938232e4b3795b8f71639e19",
}


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    lib = os.path.join(helpers.ROOT, "biokanga_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libbiokanga_amd.so")):
        pytest.skip("library not built")
    exe = str(tmp_path_factory.mktemp("r") / "report_harness")
    c = os.path.join(helpers.ROOT, "biokanga_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-o", exe, os.path.join(helpers.ROOT, "tests", "cpp", "report_harness.cpp"),
                           os.path.join(c, "host", "report.cpp"), os.path.join(c, "host", "bam_writer.cpp"), os.path.join(c, "sfx_file.cpp"),
                           "-L" + lib, "-lbiokanga_amd", "-lz", "-lpthread", "-Wl,-rpath," + lib])
    return exe


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def bgzf_text_size(path):
    """the members' text lengths summed from header to header, as a reader that knows bgzip's framing finds them; the file must end with
    bgzip's empty member"""
    raw = open(path, "rb").read()
    o = total = n = 0
    while o < len(raw):
        assert raw[o:o + 4] == b"\x1f\x8b\x08\x04" and raw[o + 12:o + 14] == b"BC", o
        bsize = int.from_bytes(raw[o + 16:o + 18], "little") + 1
        isize = int.from_bytes(raw[o + bsize - 4:o + bsize], "little")
        assert isize <= 0xff00
        total += isize
        o += bsize
        n += 1
    assert o == len(raw) and isize == 0 and bsize == 28
    return total, n


@pytest.mark.parametrize("fmt,seed,threads", [(f, s, 8) for f in range(5) for s in (1, 2)] + [(3, 1, 1), (4, 2, 1), (0, 2, 3), (5, 1, 8), (6, 1, 8), (6, 1, 1)])
def test_lines_made_by_all_threads_are_the_one_thread_writers_file(harness, tmp_path, fmt, seed, threads):
    out = str(tmp_path / "o.txt")
    args = [harness, str(fmt), str(threads), "300000", str(seed), out] + ([str(tmp_path / "g.sfx")] if fmt in (1, 3) else [])
    r = subprocess.run(args, capture_output=True, timeout=600)
    assert r.returncode == 0 and b"rc 0" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    assert md5(out) == WANT[(fmt, seed)]
    if fmt == 4:
        assert md5(out + ".ind") == WANT[(4, seed, "ind")] and md5(out + ".jct") == WANT[(4, seed, "jct")]


def test_csv_through_a_gz_name_holds_the_same_text(harness, tmp_path):
    out = str(tmp_path / "o.csv.gz")
    r = subprocess.run([harness, "2", "8", "300000", "1", out], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = gzip.open(out, "rb").read()
    assert hashlib.md5(text).hexdigest() == WANT[(2, 1)]
    assert bgzf_text_size(out)[0] == len(text)                                   # (bgzip members: other readers' threads can share them)


def test_sam_through_a_gz_name_holds_the_plain_files_text(harness, tmp_path):
    """every formatting thread's stretch as gzip members of its own, in order"""
    out = str(tmp_path / "o.sam.gz")
    r = subprocess.run([harness, "6", "8", "300000", "1", out], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = gzip.open(out, "rb").read()
    assert hashlib.md5(text).hexdigest() == WANT[(6, 1)]
    assert bgzf_text_size(out) == (len(text), (len(text) + 0xff00 - 1) // 0xff00 + 1) or bgzf_text_size(out)[0] == len(text)


@pytest.mark.parametrize("threads", [1, 8])
def test_bam_and_its_index_do_not_depend_on_the_thread_count(harness, tmp_path, threads):
    """report_bam(): records formatted and BGZF blocks deflated by all threads; the files of the commit whose GPU suite compared them with
    the reference's (byte-identical BAM + BAI)"""
    out = str(tmp_path / "o.bam")
    r = subprocess.run([harness, "6", str(threads), "300000", "1", out], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert md5(out) == "bd47269729b2794ff0cd3d61fa50f3d6" and md5(out + ".bai") == "2f410f7332ecd6dfa287e8319050e018"
    assert gzip.open(out, "rb").read(4) == b"BAM\x01"


@pytest.mark.parametrize("fmt", [6, 0])
def test_a_fifo_gets_the_whole_text_in_order(harness, tmp_path, fmt):
    """-o names a FIFO (or the /dev/fd/N of a process substitution): no offsets to write at, so the threads' text goes out in order through
    write() - the regular file's bytes"""
    import threading
    fifo = str(tmp_path / "out.fifo")
    os.mkfifo(fifo)
    got = {}

    def reader():
        h = hashlib.md5()
        with open(fifo, "rb") as f:
            while True:
                b = f.read(1 << 20)
                if not b:
                    break
                h.update(b)
        got["md5"] = h.hexdigest()

    t = threading.Thread(target=reader)
    t.start()
    r = subprocess.run([harness, str(fmt), "8", "300000", "1", fifo], capture_output=True, timeout=600)
    t.join(timeout=60)
    assert r.returncode == 0 and not t.is_alive(), r.stderr[-2000:]
    assert got["md5"] == WANT[(fmt, 1)]
