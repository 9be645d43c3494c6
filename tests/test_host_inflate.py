"""host/fast_inflate.cpp, the DEFLATE decoder of the read loaders, against zlib: what it accepts must be zlib's bytes from zlib's
input length, what it declines is left to zlib - on whole streams of every block type, and on damaged copies.  The harness is built
with the address and undefined-behaviour sanitizers: the decoder reads its input eight bytes at a time and copies matches eight
bytes at a time, and must do neither outside the buffers it was given.  CPU only."""
import os
import subprocess
import zlib

import numpy as np
import pytest

import helpers


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("z") / "inflate_harness")
    src = [os.path.join(helpers.ROOT, "tests", "cpp", "inflate_harness.cpp"), os.path.join(helpers.ROOT, "biokanga_amd", "csrc", "host", "fast_inflate.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-o", exe] + src + ["-lz"])
    return exe


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
    z = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    out = b""
    if flush_every:
        for k, i in enumerate(range(0, len(data), flush_every)):
            out += z.compress(data[i:i + flush_every]) + z.flush(zlib.Z_FULL_FLUSH if k % 2 else zlib.Z_SYNC_FLUSH)     # empty stored blocks
    else:
        out = z.compress(data)
    return out + z.flush()


def streams():
    rng = np.random.default_rng(7)
    fq = b"".join(b"@r%d len\n" % i + bytes(rng.choice(list(b"ACGTN"), 100, p=[.24, .24, .24, .24, .04]).astype(np.uint8)) + b"\n+\n" +
                  bytes(np.minimum(73, 33 + np.abs(rng.normal(35, 5, 100))).astype(np.uint8)) + b"\n" for i in range(4000))
    text = open(os.path.join(helpers.ROOT, "DESIGN.md"), "rb").read() * 3
    return {
        "fastq level 6": (fq, 6, 0, 0), "fastq level 1": (fq, 1, 0, 0), "fastq level 9": (fq, 9, 0, 0),
        "fixed codes": (fq[:200000], 6, zlib.Z_FIXED, 0), "literals only": (fq[:300000], 6, zlib.Z_HUFFMAN_ONLY, 0),
        "flushed blocks": (fq[:400000], 6, 0, 7777), "incompressible": (bytes(rng.integers(0, 256, 300000, dtype=np.uint8)), 6, 0, 0),
        "stored": (fq[:100000], 0, 0, 0), "one long run": (b"\0" * 500000, 6, 0, 0),
        "short distances": (b"ab" * 100000 + b"abc" * 50000 + b"x" * 70000 + b"abcde" * 9000 + b"abcdefg" * 9000, 9, 0, 0),
        "empty": (b"", 6, 0, 0), "one byte": (b"A", 6, 0, 0), "text": (text, 9, 0, 0),
        "far matches": (bytes(rng.integers(0, 256, 32000, dtype=np.uint8)) * 5, 9, 0, 0),
    }


@pytest.mark.parametrize("name", list(streams()))
def test_whole_streams_inflate_to_zlibs_bytes(harness, tmp_path, name):
    data, level, strategy, flush = streams()[name]
    d, r = str(tmp_path / "s.deflate"), str(tmp_path / "s.raw")
    open(d, "wb").write(deflate(data, level, strategy, flush))
    open(r, "wb").write(data)
    out = subprocess.run([harness, "check", d, r], capture_output=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0 and out.stdout.decode().startswith(f"OK {len(data)}"), (out.stdout, out.stderr[-3000:])


@pytest.mark.parametrize("name", ["fastq level 6", "fixed codes", "flushed blocks", "stored", "short distances", "text"])
def test_damaged_streams_are_declined_or_inflate_as_zlib_does(harness, tmp_path, name):
    """flipped bits, cut-off ends, overwritten stretches: never a byte that zlib would not have given, never a step outside the buffers"""
    data, level, strategy, flush = streams()[name]
    d = str(tmp_path / "s.deflate")
    open(d, "wb").write(deflate(data, level, strategy, flush))
    out = subprocess.run([harness, "fuzz", d, "11", "400"], capture_output=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0 and out.stdout.decode().startswith("OK rounds 400"), (out.stdout, out.stderr[-3000:])
    f = out.stdout.decode().split()
    assert int(f[f.index("agreed") + 1]) > 50 and int(f[f.index("declined") + 1]) > 50, out.stdout      # (both outcomes were seen)


@pytest.mark.parametrize("name,several", [("fastq level 6", True), ("fastq level 1", True), ("fastq level 9", True), ("flushed blocks", True), ("text", False),
                                          ("literals only", True), ("incompressible", False), ("stored", False), ("one long run", False), ("short distances", False)])
def test_one_stream_by_several_threads_gives_the_one_thread_bytes(harness, tmp_path, name, several):
    """threads that start at guessed block starts, with the 32 KB in front of them unknown until the thread in front is done: the bytes
    of the one-thread decoder, for text in several pieces; streams with no block start to be found (binary, one block) fall back to it"""
    data, level, strategy, flush = streams()[name]
    d = str(tmp_path / "s.deflate")
    open(d, "wb").write(deflate(data, level, strategy, flush))
    for threads in (2, 5, 8):
        out = subprocess.run([harness, "par", d, str(threads)], capture_output=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", BK_INFLATE_PIECE_MIN="40000"))
        assert out.returncode == 0 and out.stdout.decode().startswith(f"OK bytes {len(data)}"), (out.stdout, out.stderr[-3000:])
        pieces = int(out.stdout.decode().split("pieces")[1].split()[0])
        assert (pieces > 1) if several else (pieces >= 1), out.stdout


@pytest.mark.parametrize("name", ["fastq level 6", "flushed blocks"])
def test_damaged_streams_through_several_threads(harness, tmp_path, name):
    """a damaged stream whose pieces do not meet is decoded again by one thread: declined, or zlib's bytes, as before"""
    data, level, strategy, flush = streams()[name]
    d = str(tmp_path / "s.deflate")
    open(d, "wb").write(deflate(data, level, strategy, flush))
    out = subprocess.run([harness, "fuzz", d, "23", "300", "4"], capture_output=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", BK_INFLATE_PIECE_MIN="40000"))
    assert out.returncode == 0 and out.stdout.decode().startswith("OK rounds 300"), (out.stdout, out.stderr[-3000:])
