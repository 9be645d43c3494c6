import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_tmp(tmp_path_factory):
    """Session-wide scratch dir holding the decompressed golden fixtures."""
    import helpers
    d = tmp_path_factory.mktemp("golden")
    out = {}
    for name in ("basic", "repeat", "multi", "indel", "splice", "chimeric", "chimml", "chimmlindel", "combined", "snp"):
        src = os.path.join(helpers.GOLDEN, name)
        dst = d / name
        dst.mkdir()
        helpers.gunzip_to(os.path.join(src, "genome.sfx.gz"), str(dst / "genome.sfx"))
        helpers.gunzip_to(os.path.join(src, "genome.fa.gz"), str(dst / "genome.fa"))
        helpers.gunzip_to(os.path.join(src, "reads.fa.gz"), str(dst / "reads.fa"))
        out[name] = str(dst)
    # fixtures that reuse the basic genome with their own reads
    for name in ("lengths",):
        dst = d / name
        dst.mkdir()
        os.symlink(os.path.join(out["basic"], "genome.sfx"), str(dst / "genome.sfx"))
        os.symlink(os.path.join(out["basic"], "genome.fa"), str(dst / "genome.fa"))
        helpers.gunzip_to(os.path.join(helpers.GOLDEN, name, "reads.fa.gz"), str(dst / "reads.fa"))
        out[name] = str(dst)
    return out
