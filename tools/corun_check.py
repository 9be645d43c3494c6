#!/usr/bin/env python3
"""Experiment: do two halves of a step, aligned at the same time on two HIP streams (two contexts on one device, each with its own
scratch), finish sooner than one after the other?  k_wave is bound by instruction issue, the search passes by memory latency - if
the hardware interleaves them, a step split into two chunks in flight together gains what the two leave unused.
Usage: python tools/corun_check.py [--genome-mbp 3100] [--reads 50000000] [--swin 0|1]"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=3100.0)
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--max-subs", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import torch
    import biokanga_amd as bk
    from biokanga_amd import synth
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(int(args.genome_mbp * 1e6), dev, seed=38, n_seqs=24, repeat_frac=0.45)
    n = seq.numel()
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), 4, 0)
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    als = []
    for k in range(2):
        al = bk.Aligner(None, bk.AlignParams(max_subs=args.max_subs), device=0, d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=4, entries=ent)
        al.tune("use_swin", 0)
        als.append(al)
    half = args.reads // 2
    sets = []
    for k in range(2):
        b, o, l, _ = synth.make_reads(seq, seq_lens, half, args.read_len, dev, seed=77 + k, max_subs=args.max_subs)
        out = torch.zeros(half * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        sets.append((b, o, l, out))
    bw, ow, lw, _ = synth.make_reads(seq, seq_lens, args.reads, args.read_len, dev, seed=99, max_subs=args.max_subs)
    outw = torch.zeros(args.reads * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def run(k, s):
        b, o, l, out = sets[s]
        als[k].align_device(b.data_ptr(), o.data_ptr(), l.data_ptr(), half, out.data_ptr())

    for k in range(2):
        run(k, k)                                          # warm-up: scratch sized
    als[0].align_device(bw.data_ptr(), ow.data_ptr(), lw.data_ptr(), args.reads, outw.data_ptr())
    ref = [sets[0][3].clone(), sets[1][3].clone()]
    for rep in range(args.reps):
        t0 = time.time()
        als[0].align_device(bw.data_ptr(), ow.data_ptr(), lw.data_ptr(), args.reads, outw.data_ptr())
        t_whole = time.time() - t0
        t0 = time.time()
        run(0, 0)
        run(0, 1)
        t_seq = time.time() - t0
        t0 = time.time()
        th = [threading.Thread(target=run, args=(k, k)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        t_par = time.time() - t0
        same = all(bool(torch.equal(sets[k][3], ref[k])) for k in range(2))
        print(f"rep {rep}: one batch of {args.reads}: {1e3 * t_whole:.1f} ms; two halves one after the other: {1e3 * t_seq:.1f} ms; "
              f"two halves at the same time on two contexts: {1e3 * t_par:.1f} ms (results identical: {same})", flush=True)


if __name__ == "__main__":
    main()
