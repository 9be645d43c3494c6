"""GPU tests of the C++ front end (`biokanga_amd/bin/biokanga index|align`): files byte-identical to
what the real reference wrote for the same inputs (tests/golden/*)."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
BIN = os.path.join(helpers.ROOT, "biokanga_amd", "bin", "biokanga")


def run(args, cwd, env=None):
    r = subprocess.run([BIN] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900,     # (a hang is a failure)
                       env=None if env is None else dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def golden_bytes(fixture, name):
    return gzip.open(os.path.join(helpers.GOLDEN, fixture, name), "rb").read()


@pytest.mark.parametrize("fixture", ["basic", "repeat"])
def test_index_writes_the_reference_sfx(golden_tmp, tmp_path, fixture):
    """header, sequence bytes (incl. the rand()-mutated N run), entries identical; suffix array
    identical except among suffixes tied through an EOS (arbitrary in the reference)."""
    d = golden_tmp[fixture]
    out = str(tmp_path / "g.sfx")
    run(["index", "-i", os.path.join(d, "genome.fa"), "-o", out, "-r", fixture], str(tmp_path))
    got = open(out, "rb").read()
    exp = open(os.path.join(d, "genome.sfx"), "rb").read()
    assert len(got) == len(exp)
    assert got[:1224] == exp[:1224]
    blk = struct.unpack_from("<Q", exp, 44)[0]
    n = struct.unpack_from("<Q", exp, blk + 8)[0]
    assert got[blk:blk + 20 + n] == exp[blk:blk + 20 + n]              # block header + bases
    ent = struct.unpack_from("<Q", exp, 20)[0]
    assert got[ent:] == exp[ent:]                                        # entries block
    seq = np.frombuffer(exp, dtype=np.uint8, count=n, offset=blk + 20)
    sa_g = np.frombuffer(got, dtype="<u4", count=n, offset=blk + 20 + n)
    sa_e = np.frombuffer(exp, dtype="<u4", count=n, offset=blk + 20 + n)
    for j in np.nonzero(sa_g != sa_e)[0]:
        a, b = int(sa_g[j]), int(sa_e[j])
        l = 0
        while a + l < n and b + l < n and seq[a + l] == seq[b + l]:
            l += 1
        assert 7 in seq[a:a + l], (j, a, b)


def test_index_of_a_large_genome_file_by_all_threads_equals_one_thread(tmp_path):
    """a genome file of more than 1 MB is parsed by all threads in pieces cut at line starts (records of megabases, N runs, CRLF lines); -T1 keeps the record-by-record reader: the same .sfx, byte for byte - also from the gzip'd file"""
    from test_host_genome import genome_file
    fa = str(tmp_path / "g.fa")
    genome_file(fa, 5, [2_200_000, 40, 1_300_001])           # (three records: the fourth would be the nameless one, named after its file)
    open(fa + ".gz", "wb").write(gzip.compress(open(fa, "rb").read(), 4))
    outs = []
    for k, (T, f) in enumerate((("1", fa), ("8", fa), ("8", fa + ".gz"))):
        out = str(tmp_path / f"g{k}.sfx")
        run(["index", "-i", f, "-o", out, "-r", "big", f"-T{T}"], str(tmp_path))
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 5 * 3_400_000


@pytest.mark.parametrize("fixture,tag,flags", [
    ("basic", "s3", ["-s3"]), ("repeat", "s3", ["-s3"]), ("basic", "dflt", []), ("basic", "s3m2", ["-s3", "-m2"]),
    ("basic", "s3Q2", ["-s3", "-Q2"]), ("basic", "s2l30", ["-s2", "-l30"]), ("repeat", "s3m3", ["-s3", "-m3"])])
def test_align_sam_byte_identical(golden_tmp, tmp_path, fixture, tag, flags):
    d = golden_tmp[fixture]
    sfx, reads = os.path.join(d, "genome.sfx"), os.path.join(d, "reads.fa")
    out6 = str(tmp_path / "o6.sam")
    run(["align", "-i", reads, "-I", sfx, "-o", out6, "-M6"] + flags, str(tmp_path))
    assert open(out6, "rb").read() == golden_bytes(fixture, f"{tag}.m6.sam.gz")
    if tag == "s3":
        out5 = str(tmp_path / "o5.sam")
        log = run(["align", "-i", reads, "-I", sfx, "-o", out5] + flags, str(tmp_path))
        assert open(out5, "rb").read() == golden_bytes(fixture, "s3.m5.sam.gz")
        out0 = str(tmp_path / "o0.csv")
        run(["align", "-i", reads, "-I", sfx, "-o", out0, "-M0", "-O", str(tmp_path / "st.csv")] + flags, str(tmp_path))
        assert open(out0, "rb").read() == golden_bytes(fixture, "s3.m0.csv.gz")
        # the NAR histogram lines the reference logged
        exp = open(os.path.join(helpers.GOLDEN, fixture, "s3.nar.txt")).read().split("\n")
        for line in exp:
            if line.strip():
                assert line.strip() in log
        if fixture == "basic":
            assert open(tmp_path / "st.csv", "rb").read() == golden_bytes("basic", "s3.m5.stats.csv.gz")
        else:
            assert '"TargSeq","TargLen","NumHits"' in open(tmp_path / "st.csv").read()


def test_gz_reads_and_own_index_roundtrip(golden_tmp, tmp_path):
    """reads from a .gz file, index written by our own `index` -> same SAM"""
    d = golden_tmp["basic"]
    sfx = str(tmp_path / "own.sfx")
    run(["index", "-i", os.path.join(d, "genome.fa"), "-o", sfx, "-r", "basic"], str(tmp_path))
    gzr = os.path.join(helpers.GOLDEN, "basic", "reads.fa.gz")
    out = str(tmp_path / "o.sam")
    run(["align", "-i", gzr, "-I", sfx, "-o", out, "-M6", "-s3"], str(tmp_path))
    assert open(out, "rb").read() == golden_bytes("basic", "s3.m6.sam.gz")


def test_align_30000_reads_order_byte_identical(golden_tmp, tmp_path):
    """>= 25 000 reads: the reference orders its output with its own quicksort (tie order!)"""
    d = golden_tmp["basic"]
    out = str(tmp_path / "o.sam")
    run(["align", "-i", os.path.join(helpers.GOLDEN, "sortorder", "reads.fa.gz"), "-I", os.path.join(d, "genome.sfx"),
         "-o", out, "-s3"], str(tmp_path))
    assert open(out, "rb").read() == golden_bytes("sortorder", "s3.m5.sam.gz")


@pytest.mark.parametrize("tag,flags", [("U3", ["-U3", "-d200", "-D400", "-s5"]), ("U1", ["-U1", "-d200", "-D400", "-s5"]),
                                       ("U2", ["-U2", "-d200", "-D400", "-s5"]), ("U4", ["-U4", "-d200", "-D400", "-s5"]),
                                       ("U3dflt", ["-U3", "-s3"]), ("U3wide", ["-U3", "-d150", "-D1500", "-s5"]),
                                       ("U3E", ["-U3", "-d200", "-D400", "-s5", "-E"]),
                                       # chromosome filters inside the pair rules (AcceptThisChromID, Aligner.cpp:2771-2786,3224,3323,3445)
                                       ("U3ZchrB", ["-U3", "-d200", "-D400", "-s5", "-Z", "chrB"]), ("U2zchrA", ["-U2", "-d200", "-D400", "-s5", "-z", "chra"]),
                                       ("U4ZchrA", ["-U4", "-d200", "-D400", "-s5", "-Z", "chrA$"]), ("U1ZchrB", ["-U1", "-d200", "-D400", "-s5", "-Z", "chrB"])])
@pytest.mark.parametrize("fixture", ["pe", "pe150"])
def test_align_pe_sam_byte_identical(golden_tmp, tmp_path, tag, flags, fixture):
    if fixture == "pe150" and tag not in ("U1", "U2", "U3", "U4"):
        pytest.skip("the 2 x 150 bp fixture holds the four -U modes")
    d = golden_tmp["basic"]
    pe = os.path.join(helpers.GOLDEN, fixture)
    out = str(tmp_path / "pe.sam")
    run(["align", "-i", os.path.join(pe, "reads_1.fa.gz"), "-u", os.path.join(pe, "reads_2.fa.gz"), "-I", os.path.join(d, "genome.sfx"),
         "-o", out, "-M6"] + flags, str(tmp_path))
    assert open(out, "rb").read() == golden_bytes(fixture, f"{tag}.m6.sam.gz")
    if tag == "U3":
        run(["align", "-i", os.path.join(pe, "reads_1.fa.gz"), "-u", os.path.join(pe, "reads_2.fa.gz"), "-I", os.path.join(d, "genome.sfx"),
             "-o", out] + flags, str(tmp_path))
        assert open(out, "rb").read() == golden_bytes(fixture, "U3.m5.sam.gz")


@pytest.mark.parametrize("tag,flags", [("U3c50", ["-U3", "-c50", "-s3", "-d200", "-D400"]), ("U1c60", ["-U1", "-c60", "-s3", "-d200", "-D400"]),
                                       ("U4c50", ["-U4", "-c50", "-s3", "-d200", "-D400"]), ("U2c70s5", ["-U2", "-c70", "-s5", "-d200", "-D400"]),
                                       ("U3c50wide", ["-U3", "-c50", "-s3", "-d150", "-D1500"]), ("U3", ["-U3", "-s3", "-d200", "-D400"])])
def test_align_pe_with_chimeric_trimming_sam_byte_identical(golden_tmp, tmp_path, tag, flags):
    """`-c` together with `-U`: soft-clipped mates, inserts between trimmed ends, end-trimmed recovered partners - the reference's SAM"""
    d = golden_tmp["chimeric"]
    pe = os.path.join(helpers.GOLDEN, "pechim")
    out = str(tmp_path / "pechim.sam")
    run(["align", "-i", os.path.join(pe, "reads_1.fa.gz"), "-u", os.path.join(pe, "reads_2.fa.gz"), "-I", os.path.join(d, "genome.sfx"),
         "-o", out, "-M6"] + flags, str(tmp_path))
    assert open(out, "rb").read() == golden_bytes("pechim", f"{tag}.m6.sam.gz")
    if tag == "U3c50":
        out = str(tmp_path / "pechim.csv")
        run(["align", "-i", os.path.join(pe, "reads_1.fa.gz"), "-u", os.path.join(pe, "reads_2.fa.gz"), "-I", os.path.join(d, "genome.sfx"),
             "-o", out, "-M0"] + flags, str(tmp_path))
        assert open(out, "rb").read() == golden_bytes("pechim", "U3c50.m0.csv.gz")


@pytest.mark.parametrize("tag,flags,fmts", [
    ("r1R5c50", ["-r1", "-R5", "-c50", "-s3", "-T4"], ["m6.sam"]), ("r2R5c50", ["-r2", "-R5", "-c50", "-s3", "-T1"], ["m6.sam"]),
    ("r3R5c50", ["-r3", "-R5", "-c50", "-s3", "-T4"], ["m6.sam"]), ("r4R5c60", ["-r4", "-R5", "-c60", "-s3", "-T4"], ["m6.sam"]),
    ("r5R5c50", ["-r5", "-R5", "-c50", "-s3", "-T1"], ["m6.sam", "m0.csv", "m4.bed"]), ("r5R3Xc50", ["-r5", "-R3", "-X", "-c50", "-s3", "-T1"], ["m6.sam"]),
    ("r4R3Xc70s5", ["-r4", "-R3", "-X", "-c70", "-s5", "-T4"], ["m6.sam"]), ("r5R8c55e2", ["-r5", "-R8", "-c55", "-s3", "-e2", "-T1"], ["m0.csv"])])
def test_align_chimeric_with_multi_loci_modes_byte_identical(golden_tmp, tmp_path, tag, flags, fmts):
    """`-c` together with `-r1..5`: the chimeric call lists up to -R loci, each with its own soft clips; the random pick, the clustering
    (on trimmed loci) and the one-record-per-locus mode on top - the reference's files"""
    d = golden_tmp["chimml"]
    for ext in fmts:
        out = str(tmp_path / f"o.{ext}")
        run(["align", "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M" + ext[1]] + flags, str(tmp_path))
        assert open(out, "rb").read() == golden_bytes("chimml", f"{tag}.{ext}.gz"), ext


@pytest.mark.parametrize("tag,flags,fmts", [
    ("r1R5c50a8", ["-r1", "-R5", "-c50", "-a8", "-s3", "-T4"], ["m6.sam"]), ("r2R5c50a8", ["-r2", "-R5", "-c50", "-a8", "-s3", "-T1"], ["m6.sam"]),
    ("r3R5c50a8", ["-r3", "-R5", "-c50", "-a8", "-s3", "-T4"], ["m6.sam", "m0.csv"]), ("r4R5c60a5", ["-r4", "-R5", "-c60", "-a5", "-s3", "-T4"], ["m6.sam"]),
    ("r3R3Xc55a10A200", ["-r3", "-R3", "-X", "-c55", "-a10", "-A200", "-s3", "-T1"], ["m6.sam"])])
def test_align_chimeric_with_multi_loci_modes_and_indels_byte_identical(golden_tmp, tmp_path, tag, flags, fmts):
    """`-c` together with `-r1..4` AND `-a` / `-A` (the last combination earlier rounds refused): reads with small insertions and deletions
    from segments present in several places, foreign ends on half of them - the microInDel / splice junction searches and the chimeric call
    run on one set of counts and hits (SfxArrayV2.cpp:7722-7757) - the reference's files"""
    d = golden_tmp["chimmlindel"]
    for ext in fmts:
        out = str(tmp_path / f"o.{ext}")
        run(["align", "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M" + ext[1]] + flags, str(tmp_path))
        assert open(out, "rb").read() == golden_bytes("chimmlindel", f"{tag}.{ext}.gz"), ext


@pytest.mark.parametrize("fixture,name,flags", [
    ("basic", "s3.m6.bam", ["-M6", "-s3"]), ("basic", "s3.m5.bam", ["-M5", "-s3"]),
    ("pe", "U3.m6.bam", ["-M6", "-s5", "-U3", "-d200", "-D400"])])
def test_align_bam_and_bai_byte_identical(golden_tmp, tmp_path, fixture, name, flags):
    """output name ending in '.bam': BGZF blocks, BAM records and the BAI index are the reference's, byte for byte"""
    d = golden_tmp["basic"]
    out = str(tmp_path / "out_align.bam")
    if fixture == "pe":
        pe = os.path.join(helpers.GOLDEN, "pe")
        inputs = ["-i", os.path.join(pe, "reads_1.fa.gz"), "-u", os.path.join(pe, "reads_2.fa.gz")]
    else:
        inputs = ["-i", os.path.join(d, "reads.fa")]
    run(["align"] + inputs + ["-I", os.path.join(d, "genome.sfx"), "-o", out] + flags, str(tmp_path))
    exp = open(os.path.join(helpers.GOLDEN, fixture, name), "rb").read()
    got = open(out, "rb").read()
    assert gzip.decompress(got) == gzip.decompress(exp)           # records first: easier to read when it fails
    assert got == exp
    assert open(out + ".bai", "rb").read() == open(os.path.join(helpers.GOLDEN, fixture, name + ".bai"), "rb").read()


def test_align_gz_sam(golden_tmp, tmp_path):
    """output name ending in '.gz': the SAM text through zlib"""
    d = golden_tmp["basic"]
    out = str(tmp_path / "out.sam.gz")
    run(["align", "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M6", "-s3"], str(tmp_path))
    assert gzip.open(out, "rb").read() == golden_bytes("basic", "s3.m6.sam.gz")


def test_align_fastq_input(golden_tmp, tmp_path):
    """FASTQ ingest (CFasta::ParseFastQblockQ): quality lines starting with '@' / '>', optional repeated id, blank
    lines; SAM byte-identical to the reference's.  An IUPAC code inside a FASTQ sequence ends the run, as it does
    in the reference."""
    d = golden_tmp["basic"]
    out = str(tmp_path / "fq.sam")
    run(["align", "-i", os.path.join(helpers.GOLDEN, "basic", "reads.fq.gz"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M6", "-s3"],
        str(tmp_path))
    assert open(out, "rb").read() == golden_bytes("basic", "s3fq.m6.sam.gz")
    bad = str(tmp_path / "bad.fq")
    with open(bad, "w") as f:
        f.write("@ok\n" + "ACGT" * 15 + "\n+\n" + "I" * 60 + "\n@iupac\n" + "ACGR" * 15 + "\n+\n" + "I" * 60 + "\n")
    r = subprocess.run([BIN, "align", "-i", bad, "-I", os.path.join(d, "genome.sfx"), "-o", str(tmp_path / "bad.sam"), "-M6", "-s3"],
                       cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0


def test_stats_file(golden_tmp, tmp_path):
    """-O: substitution profile along the read + per-sequence counts (SE), preceded by the insert length table (PE);
    refused together with -M6 as in the reference"""
    d = golden_tmp["basic"]
    sfx = os.path.join(d, "genome.sfx")
    st = str(tmp_path / "se.csv")
    nj, mj = str(tmp_path / "none.fa"), str(tmp_path / "multi.fa")
    run(["align", "-i", os.path.join(d, "reads.fa"), "-I", sfx, "-o", str(tmp_path / "o.sam"), "-s3", "-M5", "-O", st, "-j", nj, "-J", mj],
        str(tmp_path))
    assert open(st, "rb").read() == golden_bytes("basic", "s3.m5.stats.csv.gz")
    assert open(nj, "rb").read() == golden_bytes("basic", "s3.none.fa.gz")          # -j: reads without any alignment
    assert open(mj, "rb").read() == golden_bytes("basic", "s3.multi.fa.gz")         # -J: multi-loci reads
    pe = os.path.join(helpers.GOLDEN, "pe")
    st = str(tmp_path / "pe.csv")
    run(["align", "-i", os.path.join(pe, "reads_1.fa.gz"), "-u", os.path.join(pe, "reads_2.fa.gz"), "-I", sfx, "-o", str(tmp_path / "p.sam"),
         "-s5", "-U3", "-d200", "-D400", "-M5", "-O", st, "-j", nj, "-J", mj], str(tmp_path))
    assert open(st, "rb").read() == golden_bytes("pe", "U3.m5.stats.csv.gz")
    assert open(nj, "rb").read() == golden_bytes("pe", "U3.none.fa.gz")
    assert open(mj, "rb").read() == golden_bytes("pe", "U3.multi.fa.gz")
    r = subprocess.run([BIN, "align", "-i", os.path.join(d, "reads.fa"), "-I", sfx, "-o", str(tmp_path / "x.sam"), "-s3", "-M6", "-O", st],
                       cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0


@pytest.mark.parametrize("m,name,extra", [(1, "s3.m1.csv", []), (2, "s3.m2.csv", []), (3, "s3.m3.csv", []), (4, "s3.m4.bed", []),
                                          (4, "s3.m4t.bed", ["-t", "my track"])])
def test_align_other_formats_byte_identical(golden_tmp, tmp_path, m, name, extra):
    """-M1..3: CSV with the matched target / the read / both; -M4: UCSC BED"""
    d = golden_tmp["basic"]
    out = str(tmp_path / name)
    run(["align", "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-s3", f"-M{m}"] + extra, str(tmp_path))
    assert open(out, "rb").read() == golden_bytes("basic", name + ".gz")


def test_align_end_trims(golden_tmp, tmp_path):
    """-y / -Y end trims with -l: the trimmed reads are what is aligned and reported"""
    d = golden_tmp["basic"]
    out = str(tmp_path / "trim.sam")
    run(["align", "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-s3", "-M6", "-y5", "-Y3", "-l60"], str(tmp_path))
    assert open(out, "rb").read() == golden_bytes("basic", "s3y5Y3l60.m6.sam.gz")


# multi-loci modes (-r1..-r5, -R, -X): files the reference wrote for tests/golden/multi (make_golden.py:make_multi);
# its -r2 / -r5 runs are single-threaded because their rand() sequence / record numbering follow thread timing
MULTI_CASES = [("r1R5", ["-r1", "-R5"], ["m6.sam"]), ("r2R5", ["-r2", "-R5"], ["m6.sam"]), ("r3R5", ["-r3", "-R5", "-T4"], ["m6.sam", "m5.sam"]),
               ("r4R5", ["-r4", "-R5", "-T4"], ["m6.sam"]), ("r4R3X", ["-r4", "-R3", "-X", "-T4"], ["m6.sam"]),
               ("r3R8T1", ["-r3", "-R8", "-T1"], ["m6.sam"]), ("r5R5", ["-r5", "-R5"], ["m6.sam", "m5.sam", "m0.csv", "m4.bed"]),
               ("r5R3X", ["-r5", "-R3", "-X"], ["m6.sam"]),
               # -N: LocateBestMatches instead of the AlignReads schedule
               ("r5R5N", ["-r5", "-R5", "-N"], ["m6.sam", "m0.csv"]), ("r5R2Ns1", ["-r5", "-R2", "-N", "-s1"], ["m0.csv"]),
               ("r3R4N", ["-r3", "-R4", "-N", "-T4"], ["m6.sam"]), ("r2R3N", ["-r2", "-R3", "-N"], ["m6.sam"]), ("r1R5N", ["-r1", "-R5", "-N"], ["m6.sam"]),
               # round 4: the filters the reference also runs over -r5's records (kanga.cpp:719-725,980-995 do not bar them): PCR artefact
               # reduction, flank trimming, chromosome filters
               ("r5R5k0", ["-r5", "-R5", "-k0"], ["m6.sam", "m0.csv"]), ("r5R5x4", ["-r5", "-R5", "-x4"], ["m6.sam", "m0.csv"]),
               ("r5R5ZmB", ["-r5", "-R5", "-Z", "mB"], ["m6.sam"]), ("r5R3XzmA", ["-r5", "-R3", "-X", "-z", "^ma$"], ["m5.sam"]),
               ("r5R5k20x3Z", ["-r5", "-R5", "-k20", "-x3", "-Z", "mB"], ["m4.bed"])]
FMT_FLAG = {"m6.sam": "-M6", "m5.sam": "-M5", "m0.csv": "-M0", "m4.bed": "-M4"}


@pytest.mark.parametrize("tag,flags,exts", MULTI_CASES)
def test_multi_loci_modes_byte_identical(golden_tmp, tmp_path, tag, flags, exts):
    d = golden_tmp["multi"]
    sfx, reads = os.path.join(d, "genome.sfx"), os.path.join(d, "reads.fa")
    for ext in exts:
        out = str(tmp_path / f"o.{ext}")
        run(["align", "-i", reads, "-I", sfx, "-o", out, FMT_FLAG[ext]] + ([] if "-s1" in flags else ["-s3"]) + flags, str(tmp_path))
        got, exp = open(out, "rb").read(), golden_bytes("multi", f"{tag}.{ext}.gz")
        if got != exp:
            g, e = got.split(b"\n"), exp.split(b"\n")
            k = next((i for i in range(min(len(g), len(e))) if g[i] != e[i]), min(len(g), len(e)))
            raise AssertionError(f"{tag}.{ext}: {len(g)} vs {len(e)} lines, first difference at line {k}:\n{g[k:k+1]}\n{e[k:k+1]}")


def test_multi_loci_option_checks(golden_tmp, tmp_path):
    d = golden_tmp["multi"]
    sfx, reads = os.path.join(d, "genome.sfx"), os.path.join(d, "reads.fa")
    for bad in (["-r6"], ["-r3", "-R1"], ["-r3", "-R501"], ["-r5", "-M2"]):
        r = subprocess.run([BIN, "align", "-i", reads, "-I", sfx, "-o", str(tmp_path / "x.sam")] + bad, cwd=str(tmp_path),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode != 0, bad


# microInDels (-a): files of tests/golden/indel (make_golden.py:make_indel)
INDEL_CASES = [("a10", ["-a10", "-s3"], ["m6.sam", "m5.sam", "m0.csv", "m3.csv", "m4.bed"]), ("a3s5", ["-a3", "-s5"], ["m6.sam", "m0.csv"]),
               ("a20Q1", ["-a20", "-s3", "-Q1"], ["m6.sam", "m0.csv"])]


@pytest.mark.parametrize("tag,flags,exts", INDEL_CASES)
def test_micro_indel_outputs_byte_identical(golden_tmp, tmp_path, tag, flags, exts):
    d = golden_tmp["indel"]
    sfx, reads = os.path.join(d, "genome.sfx"), os.path.join(d, "reads.fa")
    fmt = dict(FMT_FLAG, **{"m3.csv": "-M3"})
    for ext in exts:
        out = str(tmp_path / f"o.{ext}")
        run(["align", "-i", reads, "-I", sfx, "-o", out, fmt[ext]] + flags, str(tmp_path))
        pairs = [(out, f"{tag}.{ext}.gz")] + ([(out + ".ind", f"{tag}.{ext}.ind.gz")] if ext == "m4.bed" else [])
        for path, gold in pairs:
            got, exp = open(path, "rb").read(), golden_bytes("indel", gold)
            if got != exp:
                g, e = got.split(b"\n"), exp.split(b"\n")
                k = next((i for i in range(min(len(g), len(e))) if g[i] != e[i]), min(len(g), len(e)))
                raise AssertionError(f"{gold}: {len(g)} vs {len(e)} lines, first difference at line {k}:\n{g[k:k+1]}\n{e[k:k+1]}")


def test_micro_indel_bam_byte_identical(golden_tmp, tmp_path):
    d = golden_tmp["indel"]
    out = str(tmp_path / "o.bam")
    run(["align", "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M6", "-a10", "-s3"], str(tmp_path))
    gold = os.path.join(helpers.GOLDEN, "indel", "a10.m6.bam")
    assert open(out, "rb").read() == open(gold, "rb").read()
    assert open(out + ".bai", "rb").read() == open(gold + ".bai", "rb").read()


# -x (AutoTrimFlanks): soft clips in SAM / BAM, trimmed loci in CSV / BED, -O statistics over the kept part
def _cmp_bytes(path, fixture, gold):
    got, exp = open(path, "rb").read(), golden_bytes(fixture, gold)
    if got != exp:
        g, e = got.split(b"\n"), exp.split(b"\n")
        k = next((i for i in range(min(len(g), len(e))) if g[i] != e[i]), min(len(g), len(e)))
        raise AssertionError(f"{gold}: {len(g)} vs {len(e)} lines, first difference at line {k}:\n{g[k:k+1]}\n{e[k:k+1]}")


@pytest.mark.parametrize("tag,flags,exts", [("s3x5", ["-s3", "-x5"], ["m6.sam", "m5.sam", "m0.csv", "m3.csv", "m4.bed"]), ("s10x6", ["-s10", "-x6"], ["m6.sam", "m0.csv"])])
def test_flank_trim_outputs_byte_identical(golden_tmp, tmp_path, tag, flags, exts):
    d = golden_tmp["basic"]
    sfx, reads = os.path.join(d, "genome.sfx"), os.path.join(d, "reads.fa")
    fmt = dict(FMT_FLAG, **{"m3.csv": "-M3"})
    for ext in exts:
        out = str(tmp_path / f"o.{ext}")
        extra = ["-O", str(tmp_path / "st.csv")] if (tag == "s3x5" and ext == "m5.sam") else []
        run(["align", "-i", reads, "-I", sfx, "-o", out, fmt[ext]] + flags + extra, str(tmp_path))
        _cmp_bytes(out, "basic", f"{tag}.{ext}.gz")
        if extra:
            _cmp_bytes(extra[1], "basic", f"{tag}.m5.stats.csv.gz")
    if tag == "s3x5":
        out = str(tmp_path / "o.bam")
        run(["align", "-i", reads, "-I", sfx, "-o", out, "-M6"] + flags, str(tmp_path))
        gold = os.path.join(helpers.GOLDEN, "basic", "s3x5.m6.bam")
        assert open(out, "rb").read() == open(gold, "rb").read()
        assert open(out + ".bai", "rb").read() == open(gold + ".bai", "rb").read()


def test_flank_trim_pe_and_indel(golden_tmp, tmp_path):
    d = golden_tmp["basic"]
    r1, r2 = str(tmp_path / "r1.fa"), str(tmp_path / "r2.fa")
    helpers.gunzip_to(os.path.join(helpers.GOLDEN, "pe", "reads_1.fa.gz"), r1)
    helpers.gunzip_to(os.path.join(helpers.GOLDEN, "pe", "reads_2.fa.gz"), r2)
    out = str(tmp_path / "pe.sam")
    run(["align", "-i", r1, "-u", r2, "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M6", "-U3", "-d200", "-D400", "-s5", "-x4"], str(tmp_path))
    _cmp_bytes(out, "pe", "U3x4.m6.sam.gz")
    di = golden_tmp["indel"]
    for ext, m in (("m6.sam", "-M6"), ("m0.csv", "-M0")):
        out = str(tmp_path / f"i.{ext}")
        run(["align", "-i", os.path.join(di, "reads.fa"), "-I", os.path.join(di, "genome.sfx"), "-o", out, m, "-a10", "-s3", "-x4"], str(tmp_path))
        _cmp_bytes(out, "indel", f"a10x4.{ext}.gz")


# splice junctions (-A; switches -x on): files of tests/golden/splice (make_golden.py:make_splice)
@pytest.mark.parametrize("tag,flags,exts", [("A5000", ["-A5000", "-s3"], ["m6.sam", "m5.sam", "m0.csv", "m4.bed"]), ("A500s5", ["-A500", "-s5"], ["m6.sam", "m0.csv"]),
                                            ("A5000a5", ["-A5000", "-a5", "-s3"], ["m6.sam", "m0.csv"])])
def test_splice_outputs_byte_identical(golden_tmp, tmp_path, tag, flags, exts):
    d = golden_tmp["splice"]
    sfx, reads = os.path.join(d, "genome.sfx"), os.path.join(d, "reads.fa")
    for ext in exts:
        out = str(tmp_path / f"o.{ext}")
        run(["align", "-i", reads, "-I", sfx, "-o", out, FMT_FLAG[ext]] + flags, str(tmp_path))
        _cmp_bytes(out, "splice", f"{tag}.{ext}.gz")
        if ext != "m0.csv":
            _cmp_bytes(out + ".jct", "splice", f"{tag}.{ext}.jct.gz")


# chimeric trimming (-c): files of tests/golden/chimeric (make_golden.py:make_chimeric)
@pytest.mark.parametrize("tag,flags,exts", [("c50", ["-c50", "-s3"], ["m6.sam", "m5.sam", "m0.csv", "m3.csv", "m4.bed"]), ("c70s5", ["-c70", "-s5"], ["m6.sam", "m0.csv"]),
                                            ("c60e2", ["-c60", "-s3", "-e2"], ["m6.sam", "m0.csv"])])
def test_chimeric_outputs_byte_identical(golden_tmp, tmp_path, tag, flags, exts):
    d = golden_tmp["chimeric"]
    sfx, reads = os.path.join(d, "genome.sfx"), os.path.join(d, "reads.fa")
    fmt = dict(FMT_FLAG, **{"m3.csv": "-M3"})
    for ext in exts:
        out = str(tmp_path / f"o.{ext}")
        run(["align", "-i", reads, "-I", sfx, "-o", out, fmt[ext]] + flags, str(tmp_path))
        _cmp_bytes(out, "chimeric", f"{tag}.{ext}.gz")
    if tag == "c50":
        out = str(tmp_path / "o.bam")
        run(["align", "-i", reads, "-I", sfx, "-o", out, "-M6"] + flags, str(tmp_path))
        gold = os.path.join(helpers.GOLDEN, "chimeric", "c50.m6.bam")
        assert open(out, "rb").read() == open(gold, "rb").read()
        assert open(out + ".bai", "rb").read() == open(gold + ".bai", "rb").read()


# -a / -A / -c in one run (tests/golden/combined): each search of AlignReads hands its leftover state to the next one
@pytest.mark.parametrize("tag,flags", [("a10c50", ["-a10", "-c50", "-s3"]), ("a10A5000c50", ["-a10", "-A5000", "-c50", "-s3"]), ("A5000c60", ["-A5000", "-c60", "-s3"]),
                                       ("a10A5000", ["-a10", "-A5000", "-s3"])])
def test_combined_rescue_modes_byte_identical(golden_tmp, tmp_path, tag, flags):
    d = golden_tmp["combined"]
    sfx, reads = os.path.join(d, "genome.sfx"), os.path.join(d, "reads.fa")
    for ext in ("m6.sam", "m0.csv"):
        out = str(tmp_path / f"o.{ext}")
        run(["align", "-i", reads, "-I", sfx, "-o", out, FMT_FLAG[ext]] + flags, str(tmp_path))
        _cmp_bytes(out, "combined", f"{tag}.{ext}.gz")


# FASTQ quality modes (-g0..2): QUAL columns (SAM text, BAM), end trims applied to bases and scores alike, Phred bands of -O
@pytest.mark.parametrize("g", [0, 1, 2])
def test_fastq_quality_modes_byte_identical(golden_tmp, tmp_path, g):
    d = golden_tmp["basic"]
    sfx = os.path.join(d, "genome.sfx")
    fq = os.path.join(helpers.GOLDEN, "basic", "reads.fq.gz")
    out = str(tmp_path / "o.sam")
    run(["align", "-i", fq, "-I", sfx, "-o", out, "-M6", "-s3", f"-g{g}"], str(tmp_path))
    _cmp_bytes(out, "basic", f"s3fqg{g}.m6.sam.gz")
    if g == 0:
        out5, st = str(tmp_path / "o5.sam"), str(tmp_path / "st.csv")
        run(["align", "-i", fq, "-I", sfx, "-o", out5, "-M5", "-s3", "-g0", "-y3", "-Y5", "-O", st], str(tmp_path))
        _cmp_bytes(out5, "basic", "s3fqg0y3Y5.m5.sam.gz")
        _cmp_bytes(st, "basic", "s3fqg0y3Y5.m5.stats.csv.gz")
        bam = str(tmp_path / "o.bam")
        run(["align", "-i", fq, "-I", sfx, "-o", bam, "-M6", "-s3", "-g0"], str(tmp_path))
        gold = os.path.join(helpers.GOLDEN, "basic", "s3fqg0.m6.bam")
        assert open(bam, "rb").read() == open(gold, "rb").read()
        assert open(bam + ".bai", "rb").read() == open(gold + ".bai", "rb").read()


# -k (ReducePCRduplicates) on the 30 000 stacked reads of the sortorder fixture
@pytest.mark.parametrize("tag,flags", [("k0", ["-k0"]), ("k20", ["-k20"]), ("k200", ["-k200"]), ("k50x4", ["-k50", "-x4"])])
def test_pcr_artefact_reduction_byte_identical(golden_tmp, tmp_path, tag, flags):
    d = golden_tmp["basic"]
    rd = os.path.join(helpers.GOLDEN, "sortorder", "reads.fa.gz")
    out = str(tmp_path / "o.sam")
    run(["align", "-i", rd, "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M6", "-s3"] + flags, str(tmp_path))
    _cmp_bytes(out, "sortorder", f"s3{tag}.m6.sam.gz")


def test_sample_nth_read_byte_identical(golden_tmp, tmp_path):
    d = golden_tmp["basic"]
    sfx = os.path.join(d, "genome.sfx")
    out = str(tmp_path / "o.csv")
    run(["align", "-i", os.path.join(d, "reads.fa"), "-I", sfx, "-o", out, "-M0", "-s3", "-#3"], str(tmp_path))
    _cmp_bytes(out, "basic", "s3n3.m0.csv.gz")
    pe = os.path.join(helpers.GOLDEN, "pe")
    out = str(tmp_path / "pe.sam")
    run(["align", "-i", os.path.join(pe, "reads_1.fa.gz"), "-u", os.path.join(pe, "reads_2.fa.gz"), "-I", sfx, "-o", out, "-M6", "-U3", "-d200", "-D400", "-s5", "-#4"],
        str(tmp_path))
    _cmp_bytes(out, "pe", "U3n4.m6.sam.gz")


@pytest.mark.parametrize("tag,flags", [("ZchrB", ["-Z", "chrB"]), ("zchra", ["-z", "^chra$"]), ("zAZB", ["-z", "chr[AB]", "-Z", "chrA"])])
def test_chromosome_filters_byte_identical(golden_tmp, tmp_path, tag, flags):
    d = golden_tmp["basic"]
    out = str(tmp_path / "o.sam")
    run(["align", "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M6", "-s3"] + flags, str(tmp_path))
    _cmp_bytes(out, "basic", f"s3{tag}.m6.sam.gz")


SNP_RUNS = [("p5", ["-M5", "-p5"], "sam", None), ("p3P10n10", ["-M5", "-p3", "-P0.1", "-1", "10"], "sam", "vcf"), ("p5bed", ["-M4", "-p5"], "bed", None),
            ("p8x5", ["-M0", "-p8", "-x5"], "csv", None), ("p5c60", ["-M5", "-p5", "-c60"], "sam", None), ("p1P40n1", ["-M0", "-p1", "-P0.4", "-1", "0.1"], "csv", None)]


@pytest.mark.parametrize("tag,flags,ext,snpext", SNP_RUNS)
def test_snp_outputs_byte_identical(golden_tmp, tmp_path, tag, flags, ext, snpext):
    """-p / -P / -1 / -S: the SNP file (CSV, VCF or BED) and the DiSNP / TriSNP tables the reference wrote beside it.  In the VCF the
    reference prints stale stack contents as ALT / AF when every differing base of a locus is an N (nothing qualifies as ALT); those
    rows are compared without these two fields."""
    d = golden_tmp["snp"]
    out = str(tmp_path / f"{tag}.{ext}")
    cmd = ["align", "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-s5"] + flags
    snp = out + ".snp"
    if snpext:
        snp = str(tmp_path / f"{tag}.{snpext}")
        cmd += ["-S", snp]
    log = run(cmd, str(tmp_path))
    _cmp_bytes(out, "snp", f"{tag}.{ext}.gz")
    for extra in (".disnp.csv", ".trisnp.csv"):
        _cmp_bytes(snp + extra, "snp", f"{tag}{extra}.gz")
    if not snpext:
        _cmp_bytes(snp, "snp", f"{tag}.snp.gz")
    else:
        got, exp = open(snp, "rb").read().split(b"\n"), golden_bytes("snp", f"{tag}.{snpext}.gz").split(b"\n")
        assert len(got) == len(exp)
        stale = 0
        for g, e in zip(got, exp):
            if g.startswith(b"##reference="):
                continue
            if g != e:
                gf, ef = g.split(b"\t"), e.split(b"\t")
                assert gf[:4] == ef[:4] and gf[5:7] == ef[5:7] and gf[7].split(b";")[1] == ef[7].split(b";")[1], (g, e)
                stale += 1
        assert stale < len(exp) // 20
    want = [l for l in open(os.path.join(helpers.GOLDEN, "snp", f"{tag}.log.txt")).read().splitlines() if "putative SNPs" in l or "aligned loci bases" in l]
    for l in want:
        assert l.split(") ", 1)[-1] in log, l


def test_snp_option_checks(golden_tmp, tmp_path):
    d = golden_tmp["snp"]
    base = ["align", "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-o", str(tmp_path / "o.sam")]
    for bad in (["-p101"], ["-p5", "-P0.5"], ["-p5", "-1", "40"], ["-p5", "-r5"]):
        r = subprocess.run([BIN] + base + bad, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode != 0 and "Error" in r.stdout, bad
    # -M6 never opens a SNP file (Aligner.cpp:4488)
    run(base + ["-M6", "-p5", "-s5"], str(tmp_path))
    assert not os.path.exists(str(tmp_path / "o.sam.snp"))


SNP_COMBOS = [("x_pe", "pe", ["-U3", "-d200", "-D400", "-s5", "-M5", "-p1", "-P0.4", "-1", "5"], "sam"), ("x_indel", "indel", ["-a10", "-s3", "-M0", "-p1", "-P0.4", "-1", "2"], "csv"),
              ("x_multi", "multi", ["-r3", "-R5", "-s3", "-M0", "-p2", "-P0.4", "-1", "5"], "csv"),
              ("x_comb", "combined", ["-a8", "-A3000", "-c55", "-s3", "-M5", "-p1", "-P0.4", "-1", "2"], "sam"), ("x_splice", "splice", ["-A5000", "-s3", "-M4", "-p1", "-P0.4", "-1", "2"], "bed")]


@pytest.mark.parametrize("tag,fixture,flags,ext", SNP_COMBOS)
def test_snp_with_other_modes_byte_identical(golden_tmp, tmp_path, tag, fixture, flags, ext):
    """SNPs over what the other modes placed: paired ends, microInDel / spliced reads (left out of the pile-up), reads -r3 assigned,
    chimeric trims"""
    out = str(tmp_path / f"{tag}.{ext}")
    if fixture == "pe":
        r1, r2 = str(tmp_path / "r1.fa"), str(tmp_path / "r2.fa")
        helpers.gunzip_to(os.path.join(helpers.GOLDEN, "pe", "reads_1.fa.gz"), r1)
        helpers.gunzip_to(os.path.join(helpers.GOLDEN, "pe", "reads_2.fa.gz"), r2)
        src = ["-i", r1, "-u", r2, "-I", os.path.join(golden_tmp["basic"], "genome.sfx")]
    else:
        d = golden_tmp[fixture]
        src = ["-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx")]
    run(["align", "-o", out, "-T4"] + src + flags, str(tmp_path))
    _cmp_bytes(out, "snp", f"{tag}.{ext}.gz")
    for extra in (".snp", ".snp.disnp.csv", ".snp.trisnp.csv"):
        _cmp_bytes(out + extra, "snp", f"{tag}{extra}.gz")


MULTI_RESCUE = [("indel", "xi_r1R5a10", ["-r1", "-R5", "-a10", "-s3", "-M6"], "sam"), ("indel", "xi_r3R5a10", ["-r3", "-R5", "-a10", "-s3", "-M6"], "sam"),
                ("indel", "xi_r2R5a10", ["-r2", "-R5", "-a10", "-s3", "-M0"], "csv"), ("splice", "xs_r4R5A5000", ["-r4", "-R5", "-A5000", "-s3", "-M6"], "sam"),
                ("splice", "xs_r3R3XA5000", ["-r3", "-R3", "-X", "-A5000", "-s3", "-M5"], "sam"), ("combined", "xc_r3R3a8A3000", ["-r3", "-R3", "-a8", "-A3000", "-s3", "-M0"], "csv"),
                ("combined", "xc_r4R8a8A3000", ["-r4", "-R8", "-a8", "-A3000", "-s3", "-M4"], "bed"),
                # -a / -A together with -N: accepted, LocateBestMatches has no such branches, -A still trims flanks
                ("indel", "xn_r2R5Na10", ["-r2", "-R5", "-N", "-a10", "-s3", "-M6"], "sam"), ("splice", "xn_r3R4NA5000", ["-r3", "-R4", "-N", "-A5000", "-s3", "-M6"], "sam"),
                ("combined", "xn_r1R5Na8A3000", ["-r1", "-R5", "-N", "-a8", "-A3000", "-s3", "-M0"], "csv")]


@pytest.mark.parametrize("fixture,tag,flags,ext", MULTI_RESCUE)
def test_multi_loci_modes_with_indel_and_splice_byte_identical(golden_tmp, tmp_path, fixture, tag, flags, ext):
    """-r1..-r4 together with -a / -A: microInDel and spliced placements among reads handled by the multi-loci modes"""
    d = golden_tmp[fixture]
    out = str(tmp_path / f"{tag}.{ext}")
    run(["align", "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-T4"] + flags, str(tmp_path))
    _cmp_bytes(out, "multi", f"{tag}.{ext}.gz")


@pytest.mark.parametrize("tag,flags", [("k51", ["-M5", "-p5", "-K51"]), ("k25G10", ["-M0", "-p3", "-K25", "-G0.1", "-P0.2", "-1", "10"]), ("k120G45", ["-M5", "-p2", "-K120", "-G0.45"])])
def test_snp_marker_sequences_byte_identical(golden_tmp, tmp_path, tag, flags):
    """-K / -G: marker sequences assembled around the SNPs (`<snpfile>.markers`), and the SNP file restricted to the SNPs a marker could
    be built for, with their MarkerID / NumPolymorphicSites columns"""
    d = golden_tmp["snp"]
    out = str(tmp_path / f"{tag}.sam")
    log = run(["align", "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-T4", "-s5"] + flags, str(tmp_path))
    for extra in (".snp", ".snp.markers", ".snp.disnp.csv", ".snp.trisnp.csv"):
        _cmp_bytes(out + extra, "snp", f"{tag}{extra}.gz")
    for l in open(os.path.join(helpers.GOLDEN, "snp", f"{tag}.log.txt")).read().splitlines():
        assert l.split(" to file")[0] in log, l


@pytest.mark.parametrize("tag,flags", [("cent5", ["-M5", "-p5"]), ("cent2P40", ["-M0", "-p2", "-P0.4", "-1", "1"])])
def test_snp_centroids_byte_identical(golden_tmp, tmp_path, tag, flags):
    """-7: the SNP centroid table (per 7-mer context: covered loci counted on the device, SNPs and their base counts)"""
    d = golden_tmp["snp"]
    out, cent = str(tmp_path / f"{tag}.sam"), str(tmp_path / f"{tag}.centroids.csv")
    run(["align", "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-T4", "-s5", "-7", cent] + flags, str(tmp_path))
    _cmp_bytes(out + ".snp", "snp", f"{tag}.snp.gz")
    _cmp_bytes(cent, "snp", f"{tag}.centroids.csv.gz")


@pytest.mark.parametrize("fixture,tag,flags,pe", [("basic", "s3", ["-s3"], False), ("sortorder", "s3", ["-s3"], False),
                                                  ("pe", "U3", ["-U3", "-d200", "-D400", "-s5"], True)])
@pytest.mark.parametrize("devices", ["0,0", "0,0,0", "0,1", "0-1"])
def test_align_over_several_contexts_sam_byte_identical(golden_tmp, tmp_path, fixture, tag, flags, pe, devices):
    """`--devices a,b,..`: one context + pipeline per entry, batches dealt round-robin, per-sequence counts summed over the
    contexts by the exchange step - the files are those of the single-context run (= the reference's), byte for byte.  With
    one GPU visible the contexts share it; on a multi-GPU node the same code path reduces with RCCL."""
    if "1" in devices:
        import biokanga_amd
        if biokanga_amd.device_count() < 2:
            pytest.skip("needs two GPUs: index image cloned over xGMI, counts reduced with RCCL between distinct devices")
    d = golden_tmp["basic"]
    if pe:
        p = os.path.join(helpers.GOLDEN, "pe")
        inputs = ["-i", os.path.join(p, "reads_1.fa.gz"), "-u", os.path.join(p, "reads_2.fa.gz")]
    elif fixture == "sortorder":
        inputs = ["-i", os.path.join(helpers.GOLDEN, "sortorder", "reads.fa.gz")]
    else:
        inputs = ["-i", os.path.join(d, "reads.fa")]
    name = {"basic": "s3.m6.sam.gz", "sortorder": "s3.m5.sam.gz", "pe": "U3.m6.sam.gz"}[fixture]
    fmt = [] if fixture == "sortorder" else ["-M6"]
    out = str(tmp_path / "o.sam")
    log = run(["align"] + inputs + ["-I", os.path.join(d, "genome.sfx"), "-o", out, "--devices", devices] + fmt + flags, str(tmp_path))
    assert open(out, "rb").read() == golden_bytes(fixture, name)
    if not pe:
        assert f"reduced over {2 if devices == '0-1' else devices.count(',') + 1} devices" in log


def test_four_contexts_on_four_cpus(golden_tmp, tmp_path):
    """`--devices 0,0,0,0` with the process held to four CPUs (what a rank of an 8-GPU job gets under the box's CPU quota): four pipelines
    of three threads each wait for the device asleep (bk_wait.h) and size their thread pools by the CPUs they may use (bk_cpus.h) - the
    run finishes in about the time of the unconstrained one, and its file is the single-context run's, byte for byte"""
    import time
    d = golden_tmp["basic"]
    inputs = ["-i", os.path.join(helpers.GOLDEN, "sortorder", "reads.fa.gz")]
    cpus = sorted(os.sched_getaffinity(0))[:4]
    took = {}
    for name, pre in (("free", None), ("four", lambda: os.sched_setaffinity(0, set(cpus)))):
        out = str(tmp_path / f"{name}.sam")
        t = time.time()
        r = subprocess.run([BIN, "align"] + inputs + ["-I", os.path.join(d, "genome.sfx"), "-o", out, "--devices", "0,0,0,0", "-s3"], cwd=str(tmp_path),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, preexec_fn=pre)
        took[name] = time.time() - t
        assert r.returncode == 0, r.stdout[-3000:]
        assert open(out, "rb").read() == golden_bytes("sortorder", "s3.m5.sam.gz")
        assert "reduced over 4 devices" in r.stdout
    print(f"four contexts: {took['free']:.2f} s unconstrained, {took['four']:.2f} s on CPUs {cpus}")
    assert took["four"] < 3 * took["free"] + 5


@pytest.mark.parametrize("tag", ["sim", "mixed"])
def test_simreads_truth_check_line(golden_tmp, tmp_path, tag):
    """the reference's built-in correctness signal (CAligner::ReportAlignStats, Aligner.cpp:3581-3728): reads named by
    `biokanga simreads` are checked against the loci in their names; the two log lines must be the reference's, and a plainly
    named accepted read in the set drops the second one, as it does there"""
    d = golden_tmp["basic"]
    sim = os.path.join(helpers.GOLDEN, "simreads")
    out = str(tmp_path / "o.sam")
    log = run(["align", "-i", os.path.join(sim, f"{tag}.reads.fa.gz"), "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M6", "-s3"], str(tmp_path))
    assert open(out, "rb").read() == golden_bytes("simreads", f"{tag}.s3.m6.sam.gz")
    exp = [l for l in open(os.path.join(sim, f"{tag}.truthcheck.txt")).read().split("\n") if l.strip()]
    got = [l.split("](biokanga) ", 1)[-1].rstrip() for l in log.splitlines()
           if "accepted alignments" in l or "high confidence aligned simulated reads" in l]
    assert got == exp


def test_bam_gets_a_csi_index_when_a_sequence_reaches_512_mbp(tmp_path):
    """A BAI cannot address loci beyond 512 Mbp: the reference then writes a BGZF-compressed CSI (CSAMfile::StartAlignments,
    SAMfile.cpp:1602-1607).  The 537 Mbp genome is regenerated from its seed (tests/helpers.py write_big_genome), indexed by our
    own `index`, and `align -o x.bam` must give the reference's .bam and .bam.csi byte for byte (fixture: tests/golden/csi, made by
    the reference from the same genome)."""
    import shutil
    import tempfile
    gold = os.path.join(helpers.GOLDEN, "csi")
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (8 << 30) else str(tmp_path)
    work = tempfile.mkdtemp(prefix="bk_csi_", dir=base)
    try:
        fa, sfx, out = os.path.join(work, "big.fa"), os.path.join(work, "big.sfx"), os.path.join(work, "out_align.bam")
        helpers.write_big_genome(fa)
        run(["index", "-i", fa, "-o", sfx, "-r", "bigcsi"], work)
        os.remove(fa)
        log = run(["align", "-i", os.path.join(gold, "reads.fa.gz"), "-I", sfx, "-o", out, "-M6", "-s3"], work)
        assert not os.path.exists(out + ".bai")
        assert open(out, "rb").read() == open(os.path.join(gold, "s3.m6.bam"), "rb").read()
        assert open(out + ".csi", "rb").read() == open(os.path.join(gold, "s3.m6.bam.csi"), "rb").read()
        for line in open(os.path.join(gold, "s3.nar.txt")).read().split("\n"):
            if line.strip():
                assert line.strip() in log
    finally:
        shutil.rmtree(work, ignore_errors=True)


def test_options_from_a_parameter_file(golden_tmp, tmp_path):
    """`@file` arguments are replaced by the options in the file (CUtility::arg_parsefromfile, kanga.cpp:298): comment lines,
    several options per line, mixed with ordinary arguments"""
    d = golden_tmp["basic"]
    out = str(tmp_path / "o.sam")
    pf = tmp_path / "align.params"
    pf.write_text(f"# reads and index\n-i {os.path.join(d, 'reads.fa')}   -I {os.path.join(d, 'genome.sfx')}\n; output\n\n// format\n-M6\t-s3\n")
    run(["align", f"@{pf}", "-o", out], str(tmp_path))
    assert open(out, "rb").read() == golden_bytes("basic", "s3.m6.sam.gz")


@pytest.mark.parametrize("case", ["basic_m6", "basic_m6_early", "basic_m5", "repeat_m6", "sortorder", "pe_U3", "pe_U2", "pe150_U3", "fq_g0", "fq_g1", "two_contexts"])
def test_sam_formatted_on_the_device_byte_identical(golden_tmp, tmp_path, case):
    """bk_sam_format: the records of plain SAM runs formatted by the device (a lane per record) are the reference's lines byte for
    byte - single and paired ends, -M5 / -M6, FASTQ scores in QUAL, the 30 000-read order.  Small runs use the host threads by default;
    BK_SAM_DEVICE_MIN=1 sends these through the device, and the timing line says so."""
    d = golden_tmp["basic"]
    sfx = os.path.join(d, "genome.sfx")
    out = str(tmp_path / "o.sam")
    env = {"BK_SAM_DEVICE_MIN": "1", "BK_TIMING": "1"}
    if case == "basic_m6_early":                                  # the SAM file started from the input file's size, before a read is parsed
        env["BK_SAM_EARLY_MIN"] = "1"
    if case in ("basic_m6", "basic_m6_early", "basic_m5", "two_contexts"):
        args = ["-i", os.path.join(d, "reads.fa"), "-I", sfx, "-s3"] + (["-M6"] if case != "basic_m5" else []) + (["--devices", "0,0"] if case == "two_contexts" else [])
        gold = ("basic", "s3.m5.sam.gz" if case == "basic_m5" else "s3.m6.sam.gz")
    elif case == "repeat_m6":
        args = ["-i", os.path.join(golden_tmp["repeat"], "reads.fa"), "-I", os.path.join(golden_tmp["repeat"], "genome.sfx"), "-s3", "-M6"]
        gold = ("repeat", "s3.m6.sam.gz")
    elif case == "sortorder":
        args = ["-i", os.path.join(helpers.GOLDEN, "sortorder", "reads.fa.gz"), "-I", sfx, "-s3"]
        gold = ("sortorder", "s3.m5.sam.gz")
    elif case.startswith("pe"):
        fx, tag = case.split("_")
        p = os.path.join(helpers.GOLDEN, fx)
        args = ["-i", os.path.join(p, "reads_1.fa.gz"), "-u", os.path.join(p, "reads_2.fa.gz"), "-I", sfx, "-M6", f"-{tag}", "-d200", "-D400", "-s5"]
        gold = (fx, f"{tag}.m6.sam.gz")
    else:
        g = case[-1]
        args = ["-i", os.path.join(helpers.GOLDEN, "basic", "reads.fq.gz"), "-I", sfx, "-M6", "-s3", f"-g{g}"]
        gold = ("basic", f"s3fqg{g}.m6.sam.gz")
    log = run(["align", "-o", out] + args, str(tmp_path), env=env)
    assert "SAM formatted on the device" in log, log[-1500:]
    assert "head start taken" in log, log[-1500:]                 # bk_sam_prepare: the read store travelled while the host sorted
    assert open(out, "rb").read() == golden_bytes(*gold)


def test_paired_files_accepted_by_all_threads_like_the_serial_loader(golden_tmp, tmp_path):
    """plain-text mate files of more than 1 MB are parsed whole and their pairs accepted by all threads (accept_pairs); -T1 keeps the
    record-by-record loop: same SAM, same load line - with mates that fail the length rules on either side in the mix"""
    d = golden_tmp["basic"]
    pe = os.path.join(helpers.GOLDEN, "pe")
    recs = []
    for fn in ("reads_1.fa.gz", "reads_2.fa.gz"):
        txt = gzip.open(os.path.join(pe, fn), "rt").read().split(">")[1:]
        recs.append([(r.split("\n", 1)[0], "".join(r.split("\n")[1:])) for r in txt])
    files = [str(tmp_path / "big_1.fa"), str(tmp_path / "big_2.fa")]
    copies = 1 + (1 << 21) // (len(recs[0]) * 120)
    for e in (0, 1):
        with open(files[e], "w") as f:
            k = 0
            for c in range(copies):
                for name, seq in recs[e]:
                    k += 1
                    if k % 11 == 3 and e == 0: seq = seq[:30]                  # first mate under length
                    if k % 13 == 5 and e == 1: seq = seq[:20]                  # second mate under length
                    if k % 17 == 7 and e == 1: seq = seq * 7                   # second mate over length
                    f.write(f">{name}_{c}\n{seq}\n")
        assert os.path.getsize(files[e]) > (1 << 20)
    outs, loads = [], []
    for T in ("1", "8"):
        out = str(tmp_path / f"t{T}.sam")
        log = run(["align", "-i", files[0], "-u", files[1], "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M6", "-U3", "-d200", "-D400", "-s5", f"-T{T}"], str(tmp_path))
        outs.append(open(out, "rb").read())
        loads.append([l.split("](biokanga) ", 1)[1] for l in log.splitlines() if "pairs parsed" in l])
    assert loads[0] == loads[1] and loads[0], loads
    assert " 0 under length" not in loads[0][0] and " 0 over length" not in loads[0][0]
    assert outs[0] == outs[1] and len(outs[0]) > (1 << 20)


@pytest.mark.parametrize("g", ["0", "3"])
def test_plain_fastq_parsed_by_all_threads_like_the_serial_reader(golden_tmp, tmp_path, g):
    """a plain FASTQ file of more than 1 MB is parsed whole by all threads, scores packed as -g asks; -T1 keeps the serial reader: same SAM
    (QUAL included), same load line"""
    d = golden_tmp["basic"]
    txt = gzip.open(os.path.join(helpers.GOLDEN, "basic", "reads.fq.gz"), "rb").read()
    fq = str(tmp_path / "big.fq")
    with open(fq, "wb") as f:
        for c in range(1 + (3 << 20) // len(txt)):
            f.write(txt)
    assert os.path.getsize(fq) > (2 << 20)
    outs, loads = [], []
    for T in ("1", "8"):
        out = str(tmp_path / f"t{T}.sam")
        log = run(["align", "-i", fq, "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M6", "-s3", f"-g{g}", f"-T{T}"], str(tmp_path))
        outs.append(open(out, "rb").read())
        loads.append([l.split("](biokanga) ", 1)[1] for l in log.splitlines() if "reads parsed" in l])
    assert loads[0] == loads[1] and loads[0], loads
    assert outs[0] == outs[1] and len(outs[0]) > (1 << 20)


def test_bgzip_input_inflated_by_all_threads_like_the_plain_file(golden_tmp, tmp_path):
    """a bgzip'd FASTQ (members that name their size) is inflated and parsed by all threads: the plain file's SAM, QUAL included, and the
    plain file's load line; an ordinary gzip of the same text goes through gzread and gives the same again"""
    from test_host_fasta import write_bgzf
    d = golden_tmp["basic"]
    txt = gzip.open(os.path.join(helpers.GOLDEN, "basic", "reads.fq.gz"), "rb").read()
    data = txt * (1 + (3 << 20) // len(txt))
    names = {"plain": str(tmp_path / "big.fq"), "bgzf": str(tmp_path / "big.fq.bgz"), "gz": str(tmp_path / "big.fq.gz")}
    open(names["plain"], "wb").write(data)
    write_bgzf(names["bgzf"], data)
    open(names["gz"], "wb").write(gzip.compress(data, 1))
    outs, loads = {}, {}
    for k, path in names.items():
        out = str(tmp_path / f"{k}.sam")
        log = run(["align", "-i", path, "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M6", "-s3", "-g0", "-T8"], str(tmp_path))
        outs[k] = open(out, "rb").read()
        loads[k] = [l.split("](biokanga) ", 1)[1].replace(path, "F") for l in log.splitlines() if "reads parsed" in l]
    assert loads["plain"] and loads["plain"] == loads["bgzf"] == loads["gz"], loads
    assert outs["plain"] == outs["bgzf"] == outs["gz"] and len(outs["plain"]) > (1 << 20)


def test_large_gz_sam_made_by_all_threads_holds_the_plain_sam(golden_tmp, tmp_path):
    """-o out.sam.gz on some 400 000 reads: every formatting thread makes gzip members of its own stretch and they go out in order - the
    text of the plain file"""
    d = golden_tmp["basic"]
    txt = gzip.open(os.path.join(helpers.GOLDEN, "basic", "reads.fa.gz"), "rb").read()
    n1 = txt.count(b">")
    fa = str(tmp_path / "big.fa")
    with open(fa, "wb") as f:
        for c in range(1 + 400000 // n1):
            f.write(txt)
    plain, packed = str(tmp_path / "o.sam"), str(tmp_path / "o.sam.gz")
    for out in (plain, packed):
        run(["align", "-i", fa, "-I", os.path.join(d, "genome.sfx"), "-o", out, "-M6", "-s3", "-T8"], str(tmp_path))
    want = open(plain, "rb").read()
    assert want.count(b"\n") > 400000 and gzip.open(packed, "rb").read() == want
    assert open(packed, "rb").read().count(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC") > 400                # (bgzip members)


def test_device_declines_after_its_head_start(golden_tmp, tmp_path):
    """the packed reads are on the device, the read store's bases and the packed buffers have been given back - and then the device
    declines (forced): the host formatter loads the reads again and writes the same file"""
    d = golden_tmp["basic"]
    out = str(tmp_path / "o.sam")
    log = run(["align", "-o", out, "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-s3", "-M6"], str(tmp_path),
              env={"BK_SAM_DEVICE_MIN": "1", "BK_SAM_EARLY_MIN": "1", "BK_SAM_DEVICE_FAIL": "1", "BK_TIMING": "1"})
    assert "loading the reads again" in log, log[-2000:]
    assert "SAM formatted on the device" not in log
    assert open(out, "rb").read() == golden_bytes("basic", "s3.m6.sam.gz")


def test_sam_file_started_early_host_formatted(golden_tmp, tmp_path):
    """the file a large plain-text input starts early (pages allocated and mapped by background threads) also takes the host threads' text,
    and a run that fails after the start leaves it empty"""
    d = golden_tmp["basic"]
    out = str(tmp_path / "o.sam")
    run(["align", "-o", out, "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "genome.sfx"), "-s3", "-M6"], str(tmp_path), env={"BK_SAM_EARLY_MIN": "1"})
    assert open(out, "rb").read() == golden_bytes("basic", "s3.m6.sam.gz")
    p = subprocess.run([BIN, "align", "-o", out, "-i", os.path.join(d, "reads.fa"), "-I", os.path.join(d, "no_such.sfx"), "-s3", "-M6"], cwd=str(tmp_path),
                       env=dict(os.environ, BK_SAM_EARLY_MIN="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode != 0
    assert os.path.getsize(out) == 0
