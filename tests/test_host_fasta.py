"""Host ingest: the multi-threaded whole-file FASTA parse must give exactly the records of the serial
reader (CFasta semantics, libbiokanga/Fasta.cpp:907-1137) - CPU only."""
import os
import subprocess

import numpy as np
import pytest

import helpers


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("f") / "fasta_harness")
    src = [os.path.join(helpers.ROOT, "tests", "cpp", "fasta_harness.cpp"),
           os.path.join(helpers.ROOT, "biokanga_amd", "csrc", "host", "fasta.cpp"),
           os.path.join(helpers.ROOT, "biokanga_amd", "csrc", "host", "fast_inflate.cpp")]
    subprocess.check_call(helpers.cxx() + ["-pthread", "-o", exe] + src + ["-lz"])
    return exe


def nasty_fasta(path, n_records, seed):
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ACGTacgtNnRY-", dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"\n\n")
        for i in range(n_records):
            kind = i % 11
            name = f"read{i} extra words".encode()
            if kind == 3:
                name += b" with > inside the descriptor >twice"
            if kind == 5:
                name += b" caf\xc3\xa9"
            eol = b"\r\n" if kind == 7 else b"\n"
            f.write(b">" + name + eol)
            L = int(rng.integers(1, 400))
            seq = alphabet[rng.integers(0, len(alphabet), L)].tobytes()
            if kind == 2:
                seq = seq[: L // 2] + b" 12 \t" + seq[L // 2:]           # sloughed characters
            width = int(rng.integers(20, 90))
            for o in range(0, len(seq), width):
                f.write(seq[o:o + width] + eol)
            if kind == 9:
                f.write(eol)                                            # blank line between records
            if kind == 10:
                f.write(b"ACGT>midline descriptor" + eol + b"TTGA" + eol)   # '>' after bases starts a new record


@pytest.mark.parametrize("threads", [2, 7, 16])
def test_parallel_parse_equals_serial(harness, tmp_path, threads):
    p = str(tmp_path / "nasty.fa")
    nasty_fasta(p, 12000, seed=threads)
    assert os.path.getsize(p) > (1 << 20)
    out = subprocess.check_output([harness, p, str(threads)]).decode()
    assert out.startswith("OK"), out
    assert "parallel 1" in out and int(out.split("pieces")[1]) > 1, out


def test_small_files_take_the_serial_reader_and_gzip_files_are_inflated_whole(harness, tmp_path):
    import gzip
    p = str(tmp_path / "small.fa")
    nasty_fasta(p, 50, seed=1)
    assert "parallel 0" in subprocess.check_output([harness, p, "4"]).decode()
    big = str(tmp_path / "big.fa")
    nasty_fasta(big, 12000, seed=2)
    gz = str(tmp_path / "big.fa.gz")
    with open(big, "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    out = subprocess.check_output([harness, gz, "4"]).decode()
    assert out.startswith("OK") and "parallel 1" in out, out                    # (inflated into one buffer by fast_inflate, then parsed whole)
    assert out.split("parallel")[0] == subprocess.check_output([harness, big, "4"]).decode().split("parallel")[0]
    fq = str(tmp_path / "r.fq")
    with open(fq, "wb") as f:
        for i in range(30000):
            f.write(b"@q%d\nACGTNACGTTGCA\n+\nIIIIIIIIIIIII\n" % i)
    out = subprocess.check_output([harness, fq, "4"]).decode()
    assert out.startswith("OK records 30000"), out                              # (plain FASTQ of 1 MB and more is parsed whole as well)


@pytest.mark.parametrize("est", [1000, 4096, (128 << 20), (128 << 20) + 5, (300 << 20) + 12345])
def test_output_file_pages_made_in_the_background(tmp_path_factory, est):
    """SamPrealloc: every estimate - below, at and off the threads' 128 MB step - ends with the whole range ready; a kept file has the
    text's size and bytes, an abandoned one is empty"""
    d = tmp_path_factory.mktemp("pre")
    exe = str(d / "prealloc_harness")
    subprocess.check_call(helpers.cxx() + ["-pthread", "-o", exe, os.path.join(helpers.ROOT, "tests", "cpp", "prealloc_harness.cpp"), "-lz"])
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else str(d)
    f = os.path.join(shm, f"bk_prealloc_test_{os.getpid()}_{est}.bin")
    try:
        put = min(est, 777)
        subprocess.run([exe, f, str(est), str(put), "1"], check=True, timeout=120)
        assert open(f, "rb").read() == bytes(ord("a") + i % 23 for i in range(put))
        subprocess.run([exe, f, str(est), str(put), "0"], check=True, timeout=120)
        assert os.path.getsize(f) == 0
    finally:
        if os.path.exists(f):
            os.unlink(f)


def test_gzip_read_ahead_gives_the_plain_files_records(harness, tmp_path):
    """a gzip'd file larger than the reader's buffer is inflated one buffer ahead by a thread of the reader's own: same records, byte for
    byte, as the plain file (FASTA over several buffers, FASTQ with its four-line records across buffer ends)"""
    import gzip
    fa = str(tmp_path / "big.fa")
    nasty_fasta(fa, 60000, seed=5)
    assert os.path.getsize(fa) > (9 << 20)                       # more than two 4 MB buffers
    fq = str(tmp_path / "big.fq")
    rng = np.random.default_rng(3)
    with open(fq, "wb") as f:
        for i in range(120000):
            L = int(rng.integers(30, 151))
            f.write(b"@q%d extra\n" % i + np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, L)].tobytes() + b"\n+\n" + b"I" * L + b"\n")
    for plain in (fa, fq):
        gz = plain + ".gz"
        with open(plain, "rb") as f, gzip.open(gz, "wb", compresslevel=1) as g:
            g.write(f.read())
        a = subprocess.check_output([harness, plain, "1"]).decode()
        b = subprocess.check_output([harness, gz, "1"]).decode()
        assert a.startswith("OK") and b.startswith("OK"), (a, b)
        assert a.split("parallel")[0] == b.split("parallel")[0], (a, b)      # records, bases, checksum


def nasty_fastq(path, n_records, seed, broken_at=-1):
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"ACGTNacgtn", dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"\n \n")
        for i in range(n_records):
            kind = i % 13
            eol = b"\r\n" if kind == 4 else b"\n"
            L = int(rng.integers(20, 260))
            seq = letters[rng.integers(0, len(letters), L)].tobytes()
            qual = (rng.integers(33, 105, L, dtype=np.uint8)).tobytes()
            if kind == 7:
                qual = b"@" + qual[1:]                                  # a quality line that begins like an id line
            if kind == 9:
                qual = b"+" + qual[1:]
            if i == broken_at:
                qual = qual[:-1]                                        # one score short: the serial reader's error, and its alone
            f.write(b"@read%d some words" % i + eol)
            if kind == 5:
                f.write(eol)                                            # blank lines between the elements are sloughed
            f.write(seq + eol)
            f.write((b"+read%d" % i if kind == 2 else b"+") + eol)
            if kind == 6:
                f.write(eol + eol)
            f.write(qual + (b"" if i == n_records - 1 and kind != 4 else eol))


@pytest.mark.parametrize("threads,qmode", [(2, 3), (7, 0), (16, 1), (5, 2)])
def test_parallel_fastq_parse_equals_serial(harness, tmp_path, threads, qmode):
    """plain FASTQ files go through the whole-file parse too: pieces cut at records whose four lines check out, scores packed the way -g
    asks - against the serial reader, record by record"""
    p = str(tmp_path / "nasty.fq")
    nasty_fastq(p, 9000, seed=threads)
    assert os.path.getsize(p) > (2 << 20)
    out = subprocess.check_output([harness, p, str(threads), f"q{qmode}"]).decode()
    assert out.startswith("OK records 9000"), out
    assert "parallel 1" in out and int(out.split("pieces")[1]) > 1, out


def test_a_fastq_file_the_serial_reader_refuses_is_left_to_it(harness, tmp_path):
    p = str(tmp_path / "broken.fq")
    nasty_fastq(p, 9000, seed=3, broken_at=6001)
    r = subprocess.run([harness, p, "8", "q3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out = r.stdout.decode()
    # the parallel parse declines (parallel 0); both readers of the harness are then the serial one and stop at the same record with the same code
    assert "parallel 0" in out or r.returncode != 0, out


def write_bgzf(path, data, block=0xff00, level=6, eof_block=True):
    """the BGZF framing of SAMtools' bgzip: gzip members of <= 64 KB, each with its whole size (minus one) in a 'BC' extra field"""
    import struct, zlib
    with open(path, "wb") as f:
        pieces = [data[o:o + block] for o in range(0, len(data), block)] + ([b""] if eof_block else [])
        for piece in pieces:
            z = zlib.compressobj(level, zlib.DEFLATED, -15)
            c = z.compress(piece) + z.flush()
            f.write(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(c) + 25))
            f.write(c + struct.pack("<II", zlib.crc32(piece), len(piece)))


@pytest.mark.parametrize("threads", [2, 8])
def test_bgzip_files_are_inflated_and_parsed_by_all_threads(harness, tmp_path, threads):
    """a bgzip'd FASTA / FASTQ names its members' sizes: inflated member by member by every thread, parsed whole like a plain file, and
    the records are those gzread's stream gives the serial reader (the harness compares the two record by record)"""
    fa = str(tmp_path / "big.fa")
    nasty_fasta(fa, 30000, seed=11)
    fq = str(tmp_path / "big.fq")
    nasty_fastq(fq, 40000, seed=12)
    for plain in (fa, fq):
        data = open(plain, "rb").read()
        assert len(data) > (2 << 20)
        gz = plain + ".bgz"
        write_bgzf(gz, data)
        a = subprocess.check_output([harness, plain, str(threads)]).decode()
        b = subprocess.check_output([harness, gz, str(threads)]).decode()
        assert a.startswith("OK") and b.startswith("OK") and "parallel 1" in a and "parallel 1" in b, (a, b)
        assert a.split("parallel")[0] == b.split("parallel")[0], (a, b)
        write_bgzf(gz, data, block=777, level=1, eof_block=False)            # small members, no empty last one
        b = subprocess.check_output([harness, gz, str(threads)]).decode()
        assert b.startswith("OK") and "parallel 1" in b and a.split("parallel")[0] == b.split("parallel")[0], (a, b)


def test_other_gzip_layouts_are_inflated_by_one_thread_and_damaged_files_go_to_gzread(harness, tmp_path):
    """ordinary members (after bgzip ones, with a file name in the header) are inflated into one buffer and parsed whole; a wrong CRC-32, a
    wrong length, a cut-off file, bytes after the last member: the whole-file path declines and gzread decides"""
    import gzip, io
    fa = str(tmp_path / "big.fa")
    nasty_fasta(fa, 30000, seed=13)
    data = open(fa, "rb").read()
    want = subprocess.check_output([harness, fa, "4"]).decode().split("parallel")[0]
    gz = str(tmp_path / "mixed.gz")
    write_bgzf(gz, data[:1 << 20], eof_block=False)
    with open(gz, "ab") as f:
        f.write(gzip.compress(data[1 << 20:3 << 20]))
        buf = io.BytesIO()
        with gzip.GzipFile(filename="a name in the header", mode="wb", fileobj=buf, compresslevel=1) as g:
            g.write(data[3 << 20:])
        f.write(buf.getvalue())
    out = subprocess.check_output([harness, gz, "4"]).decode()
    assert out.startswith("OK") and "parallel 1" in out and out.split("parallel")[0] == want, out
    with open(gz, "ab") as f:
        f.write(b"bytes that are no gzip member")
    r = subprocess.run([harness, gz, "4"], capture_output=True)
    assert b"parallel 1" not in r.stdout, r.stdout
    good = str(tmp_path / "good.bgz")
    write_bgzf(good, data)
    plain = gzip.compress(data, 6)
    for kind, raw, first_end in (("bgz", bytearray(open(good, "rb").read()), None), ("gz", bytearray(plain), len(plain))):
        end = first_end or int.from_bytes(raw[16:18], "little") + 1                 # end of the first (or only) member
        for name, at in (("crc", end - 8), ("isize", end - 4)):
            bad = bytearray(raw)
            bad[at] ^= 0x55
            q = str(tmp_path / f"{name}.{kind}")
            open(q, "wb").write(bad)
            r = subprocess.run([harness, q, "4"], capture_output=True)
            assert b"parallel 1" not in r.stdout, (kind, name, r.stdout)                # (gzread reports the damage its own way)
        q = str(tmp_path / f"cut.{kind}")
        open(q, "wb").write(raw[: len(raw) // 2 + 5])
        r = subprocess.run([harness, q, "4"], capture_output=True)
        assert b"parallel 1" not in r.stdout, (kind, r.stdout)


def test_one_gzip_member_by_several_threads_gives_the_plain_files_records(harness, tmp_path):
    """a gzip member large enough for several threads (here: made small enough by BK_INFLATE_PIECE_MIN) is inflated in pieces that start
    at guessed block boundaries; FASTA and FASTQ, against the plain files; a file of two such members keeps to one thread per member
    after the first attempt"""
    import gzip
    fa, fq = str(tmp_path / "big.fa"), str(tmp_path / "big.fq")
    nasty_fasta(fa, 40000, seed=21)
    nasty_fastq(fq, 30000, seed=22)
    env = dict(os.environ, BK_INFLATE_PIECE_MIN="150000", BK_INFLATE_DEBUG="1")
    for plain in (fa, fq):
        data = open(plain, "rb").read()
        open(plain + ".gz", "wb").write(gzip.compress(data, 6))
        want = subprocess.check_output([harness, plain, "8"]).decode().split("parallel")[0]
        r = subprocess.run([harness, plain + ".gz", "8"], capture_output=True, env=env)
        assert r.stdout.decode().startswith("OK") and "parallel 1" in r.stdout.decode() and r.stdout.decode().split("parallel")[0] == want, r.stdout
        assert int(r.stderr.decode().split("inflate: ")[1].split()[0]) > 1, r.stderr                 # pieces
        open(plain + ".2.gz", "wb").write(gzip.compress(data[:len(data) // 2], 6) + gzip.compress(data[len(data) // 2:], 6))
        r = subprocess.run([harness, plain + ".2.gz", "8"], capture_output=True, env=env)
        assert r.stdout.decode().startswith("OK") and "parallel 1" in r.stdout.decode() and r.stdout.decode().split("parallel")[0] == want, r.stdout


def test_text_size_of_a_read_file_without_reading_it(harness, tmp_path):
    """plain: the file's size; bgzip'd: the members' lengths summed (exact); one gzip member: its length word, which is the size below 4 GB"""
    import gzip
    fa = str(tmp_path / "r.fa")
    nasty_fasta(fa, 9000, seed=31)
    data = open(fa, "rb").read()
    write_bgzf(fa + ".bgz", data)
    open(fa + ".gz", "wb").write(gzip.compress(data, 6))
    for f in (fa, fa + ".bgz", fa + ".gz"):
        assert subprocess.check_output([harness, f, "1", "est"]).decode().split() == ["text", str(len(data))], f
    assert subprocess.check_output([harness, str(tmp_path / "none"), "1", "est"]).decode().split() == ["text", "0"]


@pytest.mark.parametrize("file_gb,word_gb,want_gb", [(1.0, 0.2, 4.2), (1.0, 3.9, 3.9), (2.0, 0.3, 8.3), (0.5, 1.6, 1.6), (3.0, 2.5, 10.5)])
def test_text_size_of_a_large_gzip_member_from_its_length_word(harness, tmp_path, file_gb, word_gb, want_gb):
    """the length word counts modulo 4 GB: the text nearest to 3.5 times the file is taken (a sparse file stands in: only its first and last
    bytes are looked at)"""
    import struct
    p = str(tmp_path / "big.gz")
    size, word = int(file_gb * (1 << 30)), int(word_gb * (1 << 30))
    with open(p, "wb") as f:
        f.write(b"\x1f\x8b\x08\x00" + b"\0" * 6)
        f.truncate(size)
        f.seek(size - 8)
        f.write(struct.pack("<II", 0, word))
    got = int(subprocess.check_output([harness, p, "1", "est"]).decode().split()[1])
    assert got == word + int(round((want_gb - word_gb) / 4)) * (1 << 32), (got, word)


@pytest.mark.parametrize("lines", [0, 1, 260000])
def test_outputs_named_gz_are_gzip_members_in_order(harness, tmp_path, lines):
    """an output whose name ends in .gz: what is put and the members the writer's threads hand over come out as one gzip file - bgzip
    members, the text in order; nothing put: an empty text.  This package's loader takes such a file by all threads (its size estimate
    reads the members' lengths)"""
    import gzip
    exe = str(tmp_path / "outbuf_harness")
    subprocess.check_call(helpers.cxx() + ["-pthread", "-o", exe, os.path.join(helpers.ROOT, "tests", "cpp", "outbuf_harness.cpp"), "-lz"])
    out = str(tmp_path / "o.txt.gz")
    h, total, failed = subprocess.check_output([exe, out, str(lines)]).decode().split()
    text = gzip.open(out, "rb").read()
    assert failed == "0" and len(text) == int(total)
    raw = open(out, "rb").read()
    assert raw.startswith(b"\x1f\x8b\x08\x04") and raw[12:14] == b"BC" and raw.endswith(b"\x1b\0\x03\0" + b"\0" * 8)
    if lines:
        assert subprocess.check_output([harness, out, "1", "est"]).decode().split() == ["text", total]         # the loader's walk over the members
    f = 1469598103934665603
    if lines <= 1:
        for c in text:
            f = ((f ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        assert "%016x" % f == h
    else:
        assert text.startswith(b"read0\t0\tchr0\t0\t") and text.endswith(b"ACGTACGTTTGACCAGT%d\n" % ((lines - 1) % 1000)) and text.count(b"\n") == lines
        assert [l.split(b"\t")[0] for l in text.split(b"\n")[49990:50010]] == [b"read%d" % i for i in range(49990, 50010)]      # across the hand-over


def test_a_full_device_is_noticed_by_the_output_buffer(tmp_path):
    """writes that do not go through (here: /dev/full) leave the buffer marked as failed - the report ends with an error, not with a short file"""
    exe = str(tmp_path / "outbuf_harness")
    subprocess.check_call(helpers.cxx() + ["-pthread", "-o", exe, os.path.join(helpers.ROOT, "tests", "cpp", "outbuf_harness.cpp"), "-lz"])
    assert subprocess.check_output([exe, "/dev/full", "200000"]).decode().split()[2] == "1"
