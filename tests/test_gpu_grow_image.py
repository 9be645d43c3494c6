"""BK_CTX_GROW_IMAGE: a context that starts with the lean image and makes the long-run tables (key arrays behind the second-level keys, k-mer
table entries with their bucket's first key) on a thread of its own while batches run - every batch, whichever image it ran on, gives the
oracle's records and counts, and the grown image is the one a context built up front has."""
import os
import sys
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import helpers  # noqa: E402
from test_gpu_window_array import _family_genome, _index, _same  # noqa: E402

pytestmark = pytest.mark.gpu



def _bk():
    import biokanga_amd
    return biokanga_amd


def test_image_grows_between_batches(tmp_path):
    bk = _bk()
    seq, ents, reads = _family_genome(4242, 600000, 100, 20000, 4)
    path = _index(tmp_path, seq, ents, "grow")
    nreads = len(reads)
    bases = reads.reshape(-1)
    offs = np.arange(nreads, dtype=np.uint64) * 100
    lens = np.full(nreads, 100, dtype=np.uint32)
    o = helpers.OracleSfx(path)
    exp, octr = o.align(bases, offs, lens, helpers.make_params(max_subs=3), nthreads=8)
    o.close()

    def check(al, what):
        al.counters(reset=True)
        got = al.align(bases, offs, lens)
        ctr = al.counters()
        _same(got, exp, what)
        assert (ctr["n_search"], ctr["n_cand"], ctr["n_lcm_calls"]) == (octr.n_search, octr.n_cand, octr.n_lcm_calls), what

    with bk.Aligner(path, bk.AlignParams(max_subs=3), flags=_bk().CTX_GROW_IMAGE) as al:
        assert al.tune("grow_state", 0) == 0 and al.tune("k3_resident", 0) == 0 and al.tune("ktab2_resident", 0) == 0
        al.tune("grow_after_reads", 3 * nreads)
        check(al, "lean image, first batch")
        check(al, "lean image, second batch")
        assert al.tune("grow_state", 0) == 0
        check(al, "the batch that starts the worker")
        assert al.tune("grow_state", 0) in (1, 2)
        # batches while the worker runs, and the one that takes its tables in
        for i in range(200):
            check(al, f"batch {i} beside the worker")
            if al.tune("grow_state", 0) == 4:
                break
            time.sleep(0.01)
        assert al.tune("grow_state", 0) == 4
        assert al.tune("k3_resident", 0) == 2 and al.tune("ktab2_resident", 0) == 1
        check(al, "grown image")
        for kv in (("use_swin", 2), ("heavy_thresh", 0), ("lazy_search", 0)):
            al.tune(*kv)
        check(al, "grown image, window array, every read through the wave kernel")
    # asked for at once; a knob that rebuilds the tables while the worker runs
    with bk.Aligner(path, bk.AlignParams(max_subs=3), flags=_bk().CTX_GROW_IMAGE) as al:
        assert al.tune("image_wait", 0) == 2 + 4
        check(al, "image_wait")
    with bk.Aligner(path, bk.AlignParams(max_subs=3), flags=_bk().CTX_GROW_IMAGE) as al:
        al.tune("grow_after_reads", 1)
        check(al, "starts the worker")
        al.tune("kmer_bits", 9)                        # (the tables are made again: what the worker made is dropped)
        assert al.tune("k3_resident", 0) == 0
        check(al, "tables rebuilt under the worker")
    with bk.Aligner(path, bk.AlignParams(max_subs=3), flags=_bk().CTX_LEAN_IMAGE) as al:
        assert al.tune("image_wait", 0) == 0 and al.tune("grow_state", 0) == 5
        check(al, "lean image, nothing grows")
