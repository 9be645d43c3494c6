"""The .sfx image `biokanga index` writes (sfx_file.cpp, CSfxArrayV3::Finalise's layout): an image of 16 M bases and more goes out through
several threads that pwrite() their own slices - the same file as from one thread, and as the tests' own writer makes.  CPU only."""
import os
import subprocess

import numpy as np
import pytest

import helpers


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("s") / "sfx_harness")
    subprocess.check_call(helpers.cxx() + ["-pthread", "-o", out, os.path.join(helpers.ROOT, "tests", "cpp", "sfx_harness.cpp"),
                                           os.path.join(helpers.ROOT, "biokanga_amd", "csrc", "sfx_file.cpp")])
    return out


@pytest.mark.parametrize("n,el", [((16 << 20) + 5, 4), ((16 << 20) + 1, 5), (300_000, 4)])
def test_image_written_by_several_threads_is_the_one_thread_file(exe, tmp_path, n, el):
    rng = np.random.default_rng(n)
    seq = rng.integers(0, 5, n, dtype=np.uint8)
    cut = n // 3
    seq[cut] = 7
    seq[n - 1] = 7
    sa = rng.integers(0, 256, n * el, dtype=np.uint8)
    bp, sp = str(tmp_path / "bases.bin"), str(tmp_path / "sa.bin")
    seq.tofile(bp)
    sa.tofile(sp)
    ents = [("first", cut), ("second", n - cut - 2)]
    outs = []
    for T in (1, 8):
        out = str(tmp_path / f"t{T}.sfx")
        subprocess.check_call([exe, bp, sp, str(el), str(T), out] + [f"{a}:{b}" for a, b in ents])
        outs.append(out)
    a, b = (np.fromfile(o, dtype=np.uint8) for o in outs)
    assert a.size == b.size == 1224 + 20 + n + n * el + 8 + 111 * 2 and np.array_equal(a, b)
    if el == 4:
        ref = str(tmp_path / "py.sfx")
        helpers.write_sfx(ref, "ds", ents, seq, sa.view("<u4"))
        c = np.fromfile(ref, dtype=np.uint8)
        same = a == c
        same[52 + 80:52 + 80 + 1024 + 64] = True                      # (the test writer leaves description and title empty)
        assert c.size == a.size and same.all()
    for o in outs:
        os.unlink(o)
