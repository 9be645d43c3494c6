"""Option checks of `biokanga align` that end before anything touches a device: the messages and exit codes of kanga.cpp's own checks
(kanga.cpp:648-660,712-716) - CPU only.  (The one combination earlier rounds refused, -c with -r and -a / -A, is pinned on reference runs since
round 5: tests/test_gpu_cli.py, tests/golden/chimmlindel.)"""
import os
import subprocess

import pytest

import helpers

BIN = os.path.join(helpers.ROOT, "biokanga_amd", "bin", "biokanga")
pytestmark = pytest.mark.skipif(not os.path.exists(BIN), reason="the command line has not been built (python -c 'import __graft_entry__ as g; g.build()')")


def run(args):
    r = subprocess.run([BIN, "align", "-i", "none.fa", "-I", "none.sfx", "-o", "none.sam"] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
    return r.returncode, r.stdout


@pytest.mark.parametrize("args,text", [
    (["-c40"], "minimum chimeric length percentage '-c40' specified outside of range 50..99"),
    # the reference refuses this one itself, in these words (kanga.cpp:712-716)
    (["-c50", "-r3", "-R5", "-N"], "Error: Sorry, chimeric read processing not supported in this release if either SOLiD or locating multiple best matches also requested"),
    (["-r5", "-R5", "-A100"], "in report all multiloci mode '-r5', there is no splice junction processing"),
])
def test_refusals_before_any_device_work(args, text):
    rc, out = run(args)
    assert rc == 1, out
    assert text in out, out
    assert "Exit code: 1" in out
