"""Multi-GPU plumbing of the hot path (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in CPU tests).  Test plumbing: bench.py shards on the device itself.

The path shards trivially: reads are independent given a replicated index.  Read i goes to rank
i mod world; the only exchange is the sum-reduction of the per-sequence accepted-read
counts (CAligner::ReportTargHitCnts, biokanga/Aligner.cpp:5475-5537) and of the NAR histogram
(Aligner.cpp:3744-3769) - a few hundred bytes, latency-bound."""
import numpy as np


def shard_indices(n_reads, rank, world):
    """indices of the reads rank `rank` aligns: i mod world == rank"""
    return np.arange(rank, n_reads, world, dtype=np.int64)


def reduce_stats(seq_counts, nar_hist, device=None):
    """sum-all-reduce of (per-sequence accepted counts, NAR histogram[20]); returns numpy arrays.
    No-op when torch.distributed is not initialised."""
    import torch
    import torch.distributed as dist
    sc = np.asarray(seq_counts, dtype=np.int64)
    nh = np.asarray(nar_hist, dtype=np.int64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sc.copy(), nh.copy()
    t = torch.from_numpy(np.concatenate([sc, nh]))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t.cpu().numpy()
    return t[:len(sc)].copy(), t[len(sc):].copy()


def gather_hits(local_hits, n_reads, rank, world):
    """reassembles the per-read results in global read order on every rank (host side; the
    reference keeps all tsReadHit records in one array, biokanga/Aligner.cpp:9943-9949)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return local_hits
    parts = [None] * world
    dist.all_gather_object(parts, local_hits.tobytes())
    out = np.zeros(n_reads, dtype=local_hits.dtype)
    for r in range(world):
        out[shard_indices(n_reads, r, world)] = np.frombuffer(parts[r], dtype=local_hits.dtype)
    return out
