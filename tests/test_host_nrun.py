"""`biokanga index` thins out long runs of N as it takes a sequence in (kangax.cpp:626-660): host/nrun_mutate.h, which skips eight bases at
a time where no N stands, against the reference's loop as written - same bases, same number of rand() values drawn.  CPU only."""
import os
import subprocess

import pytest

import helpers


@pytest.mark.parametrize("seed", [1, 7, 12345])
def test_n_run_mutation_equals_the_loop_as_written(tmp_path, seed):
    exe = str(tmp_path / "nrun_harness")
    subprocess.check_call(helpers.cxx() + ["-o", exe, os.path.join(helpers.ROOT, "tests", "cpp", "nrun_harness.cpp")])
    out = subprocess.check_output([exe, str(seed), "2000"]).decode()
    assert out.startswith("OK rounds 2000") and int(out.split()[-1]) > 10000, out
