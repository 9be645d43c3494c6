"""world_size-2 test (gloo, CPU) of the N > 1 path: reads sharded i mod world with a replicated
index, per-sequence hit counts + NAR histogram sum-reduced, results reassembled in read order.
The per-rank aligner here is the CPU oracle (tests may use it); on the GPU box bench.py runs the
same plumbing over RCCL with the HIP aligner."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import helpers


def _worker(rank, world, port, sfx_path, reads_path, out_dir):
    sys.path.insert(0, os.path.join(helpers.ROOT, "tests"))
    sys.path.insert(0, helpers.ROOT)
    import torch.distributed as dist
    import dist_plumbing as bkdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names, bases, offs, lens = helpers.read_fasta_reads(reads_path)
    keep = np.array(helpers.filter_reads_by_len(names, bases, offs, lens))
    mine = keep[bkdist.shard_indices(len(keep), rank, world)]
    o = helpers.OracleSfx(sfx_path)
    hits, _ = o.align(bases, offs[mine], lens[mine], helpers.make_params(max_subs=3), nthreads=2)
    n_ent = 2
    sc = np.array([np.count_nonzero((hits["nar"] == 1) & (hits["chrom_id"] == e + 1)) for e in range(n_ent)])
    nh = np.bincount(hits["nar"], minlength=20)
    tot_sc, tot_nh = bkdist.reduce_stats(sc, nh)
    allhits = bkdist.gather_hits(hits, len(keep), rank, world)
    np.save(os.path.join(out_dir, f"sc{rank}.npy"), tot_sc)
    np.save(os.path.join(out_dir, f"nh{rank}.npy"), tot_nh)
    np.save(os.path.join(out_dir, f"hits{rank}.npy"), allhits)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_alignment_matches_single_process(golden_tmp, tmp_path):
    d = golden_tmp["basic"]
    sfx_path, reads_path = os.path.join(d, "genome.sfx"), os.path.join(d, "reads.fa")
    world = 2
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(world, port, sfx_path, reads_path, str(tmp_path)), nprocs=world, join=True)
    names, bases, offs, lens = helpers.read_fasta_reads(reads_path)
    keep = helpers.filter_reads_by_len(names, bases, offs, lens)
    o = helpers.OracleSfx(sfx_path)
    ref, _ = o.align(bases, offs[keep], lens[keep], helpers.make_params(max_subs=3))
    ref_sc = np.array([np.count_nonzero((ref["nar"] == 1) & (ref["chrom_id"] == e + 1)) for e in range(2)])
    ref_nh = np.bincount(ref["nar"], minlength=20)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"sc{r}.npy"), ref_sc)
        assert np.array_equal(np.load(tmp_path / f"nh{r}.npy"), ref_nh)
        got = np.load(tmp_path / f"hits{r}.npy")
        assert got.tobytes() == ref.tobytes()
    # golden log summary of the reference for the same run
    exp = {}
    with open(os.path.join(helpers.GOLDEN, "basic", "s3.nar.txt")) as f:
        for line in f:
            t = line.split()
            exp[t[1].strip("()")] = int(t[0])
    assert ref_nh[1] == exp["AA"] and ref_nh[3] == exp["NL"] and ref_nh[5] == exp["ML"]
