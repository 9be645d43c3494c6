"""Host BAM/BAI writer (CPU only): feeding it the uncompressed record stream of a BAM the reference wrote must
reproduce that file and its .bai byte for byte - the BGZF block cuts (0xff00 bytes, plus the flush after the
last aligned record), zlib level 6 raw deflate, virtual offsets, chunk merging and the sparse linear index; for sequences of 512 Mbp and more the CSI's bins, loffsets and its own BGZF stream."""
import gzip
import os
import subprocess

import pytest

import helpers


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("b") / "bam_harness")
    src = [os.path.join(helpers.ROOT, "tests", "cpp", "bam_harness.cpp"),
           os.path.join(helpers.ROOT, "biokanga_amd", "csrc", "host", "bam_writer.cpp")]
    subprocess.check_call(helpers.cxx() + ["-pthread", "-o", exe] + src + ["-lz"])
    return exe


@pytest.mark.parametrize("fixture,name", [("basic", "s3.m6.bam"), ("basic", "s3.m5.bam"), ("pe", "U3.m6.bam"), ("csi", "s3.m6.bam")])
@pytest.mark.parametrize("threads", [1, 5])
def test_bam_writer_reproduces_reference_files(harness, tmp_path, fixture, name, threads):
    ref = os.path.join(helpers.GOLDEN, fixture, name)
    raw = str(tmp_path / "stream.bin")
    with open(raw, "wb") as f:
        f.write(gzip.open(ref, "rb").read())
    out = str(tmp_path / "out.bam")
    subprocess.check_call([harness, raw, out, str(threads)])
    assert open(out, "rb").read() == open(ref, "rb").read()
    idx = ".csi" if fixture == "csi" else ".bai"         # a 537 Mbp sequence in the header: BGZF-compressed CSI instead of the BAI
    assert open(out + idx, "rb").read() == open(ref + idx, "rb").read()
    assert not os.path.exists(out + (".bai" if fixture == "csi" else ".csi"))
