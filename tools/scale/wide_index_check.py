#!/usr/bin/env python3
"""Suffix sort at wheat scale (BASELINE.json config 5: 17 Gbp, 5-byte elements): builds the suffix array of a synthetic
"wheat-like" genome (21 sequences, 85 % repeat-derived) on one MI355X and checks it by properties - every position once,
sampled neighbours in nibble-lexicographic order.  python tools/scale/wide_index_check.py [genome_mbp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import biokanga_amd as bk
from biokanga_amd import synth


def main():
    mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 17000.0
    dev = torch.device("cuda", 0)
    t = time.time()
    seq, seq_lens = synth.make_genome(int(mbp * 1e6), dev, seed=17, n_seqs=21, repeat_frac=0.85)
    n = seq.numel()
    torch.cuda.synchronize()
    print(f"genome: {n} bases in {len(seq_lens)} sequences, generated in {time.time() - t:.0f} s", flush=True)
    el = 5 if n >= 0xFFFFFFFF else 4
    d_sa = torch.empty(n * el, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t = time.time()
    bk.build_sa_device(seq.data_ptr(), n, d_sa.data_ptr(), el, 0)
    torch.cuda.synchronize()
    dt = time.time() - t
    free, total = torch.cuda.mem_get_info(0)
    print(f"suffix array of {n} suffixes ({el}-byte elements) built in {dt:.1f} s = {n / dt / 1e6:.0f} M suffixes/s; HBM in use afterwards {(total - free) / 1e9:.0f} GB", flush=True)
    g = torch.Generator(device=dev); g.manual_seed(5)
    v = d_sa.view(n, el)

    def sa_at(j):
        out = torch.zeros(j.shape, dtype=torch.int64, device=dev)
        for k in range(el):
            out |= v[j, k].to(torch.int64) << (8 * k)
        return out
    # permutation: mark every value, piecewise
    seen = torch.zeros(n, dtype=torch.uint8, device=dev)
    step = 1 << 30
    for lo in range(0, n, step):
        j = torch.arange(lo, min(n, lo + step), device=dev)
        seen[sa_at(j)] = 1
        del j
    ok_perm = int(seen.sum(dtype=torch.int64)) == n
    del seen
    print("every position appears exactly once:", ok_perm, flush=True)
    j = torch.randint(0, n - 1, (1_000_000,), device=dev, generator=g)
    a, b = sa_at(j), sa_at(j + 1)
    undecided = torch.ones_like(a, dtype=torch.bool)
    ar = torch.arange(64, device=dev)
    bad = 0
    for depth in range(0, 8192, 64):
        ia, ib = a[undecided, None] + depth + ar, b[undecided, None] + depth + ar
        zero = torch.zeros((), dtype=torch.int16, device=dev)
        ca = torch.where(ia < n, (seq[ia.clamp(max=n - 1)] & 15).to(torch.int16) + 1, zero)
        cb = torch.where(ib < n, (seq[ib.clamp(max=n - 1)] & 15).to(torch.int16) + 1, zero)
        diff = ca != cb
        anyd = diff.any(dim=1)
        first = diff.to(torch.int8).argmax(dim=1)
        rows = torch.nonzero(anyd).squeeze(1)
        bad += int((ca[rows, first[rows]] > cb[rows, first[rows]]).sum())
        idx = torch.nonzero(undecided).squeeze(1)
        undecided[idx[anyd]] = False
        if not bool(undecided.any()):
            break
    print(f"1 000 000 sampled neighbours: {bad} out of order, {int(undecided.sum())} still equal after 8192 bases", flush=True)
    sys.exit(0 if ok_perm and bad == 0 else 1)


if __name__ == "__main__":
    main()
