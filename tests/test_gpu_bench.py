"""GPU test of bench.py's multi-rank path with the HIP library (not the CPU oracle): two ranks, started by bench.py itself because no
launcher set WORLD_SIZE, share the one visible GPU (--force-device 0) and talk over gloo; ONE read set dealt i mod 2 must reduce to
the per-sequence counts and the NAR histogram of a single-rank run of the whole set, and the weak / strong scaling lines must be there.
(On a multi-GPU node the same code runs with the RCCL backend, one rank per device.)"""
import json
import os
import subprocess
import sys

import pytest

import helpers

pytestmark = pytest.mark.gpu


def test_two_ranks_shard_one_read_set_like_a_single_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--force-device", "0", "--dist-backend", "gloo",
           "--genome-mbp", "30", "--reads", "400000", "--steps", "1", "--warmup", "1", "--cpu-baseline-secs", "0",
           "--stream-batch", "100000", "--shard-check-reads", "300000", "--no-live-traffic"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    sc = d["shard_check"]
    assert sc["per_sequence_counts_equal_1gpu_run"] and sc["nar_histogram_equal_1gpu_run"] and sc["accepted"] > 0
    assert d["strong_scaling"]["reads_total_per_step"] == 400000 and d["strong_scaling"]["value"] > 0
    assert d["t_align_host_resident"]["results_bit_identical_to_kernel_only_steps"]
    # `value` is the host-in / host-out clock (SURVEY 8d T_align), the kernel-only rate stands beside it
    assert d["value"] == d["t_align_host_resident"]["value"] and d["value_kernel_only"] > 0 and "host memory" in d["value_clock"]
    # (the layout `biokanga align` picks for C2's 50 M reads per device: the partial window array)
    assert d["config"]["window_array"].startswith("partial") and d["roofline"]["window_array"] == "partial"


def test_rank_count_mismatch_is_refused():
    """a launcher that started fewer ranks than --gpus asks for must not yield an N-GPU line"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--genome-mbp", "30", "--reads", "1000"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
