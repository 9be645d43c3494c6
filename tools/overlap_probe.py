#!/usr/bin/env python3
"""Do two batches of the hot path, enqueued on two streams, finish sooner than one after the other?  The wave kernel is bound by
instruction issue and the search / extend kernels by the rate of random lines, so their overlap could hide one behind the other - or
the two could just get in each other's way.  C2's workload: two contexts over one suffix array (each keeps an image and tables of its
own), every step = two half-batches; measured: a) one context, whole batches, b) one context, half-batches one after the other, c) two
contexts, the half-batches of a step enqueued together.

usage: overlap_probe.py [genome_mbp] [reads_per_step] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100.0
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    import torch
    import biokanga_amd as bk
    from biokanga_amd import synth
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(int(mbp * 1e6), dev, seed=38, n_seqs=24, repeat_frac=0.45)
    n = seq.numel()
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), 4, 0)
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    half = n_reads // 2
    als = []
    for i in range(2):
        al = bk.Aligner(None, bk.AlignParams(max_subs=3), device=0, d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=4, entries=ent)
        al.tune("use_swin", 1)
        al.reserve(n_reads if i == 0 else half, 100)
        als.append(al)
    del sa
    torch.cuda.empty_cache()
    b, o, l, _ = synth.make_reads(seq, seq_lens, n_reads, 100, dev, seed=1000, max_subs=3)
    halves = []
    for i in range(2):
        hb = b[i * half * 100:(i + 1) * half * 100].contiguous()
        ho = (o[i * half:(i + 1) * half] - o[i * half]).contiguous()
        hl = l[i * half:(i + 1) * half].contiguous()
        halves.append((hb, ho, hl))
    out = torch.zeros(n_reads * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    outs = [out[:half * bk.HIT_DTYPE.itemsize], out[half * bk.HIT_DTYPE.itemsize:2 * half * bk.HIT_DTYPE.itemsize]]
    ref = torch.zeros_like(out)

    def whole():
        als[0].align_device_async(b.data_ptr(), o.data_ptr(), l.data_ptr(), n_reads, 100, out.data_ptr())

    def serial_halves():
        for i in range(2):
            hb, ho, hl = halves[i]
            als[0].align_device_async(hb.data_ptr(), ho.data_ptr(), hl.data_ptr(), half, 100, outs[i].data_ptr())

    def together():
        for i in range(2):
            hb, ho, hl = halves[i]
            als[i].align_device_async(hb.data_ptr(), ho.data_ptr(), hl.data_ptr(), half, 100, outs[i].data_ptr())

    # blocking first calls: the window array is made by the first batch
    als[0].align_device(b.data_ptr(), o.data_ptr(), l.data_ptr(), n_reads, out.data_ptr())
    als[1].align_device(halves[1][0].data_ptr(), halves[1][1].data_ptr(), halves[1][2].data_ptr(), half, outs[1].data_ptr())
    print(f"window arrays: {[al.tune('swin_resident', 0) for al in als]}, free HBM {torch.cuda.mem_get_info()[0] / 1e9:.0f} GB", flush=True)
    res = {}
    for name, fn in (("one context, whole batches", whole), ("one context, half-batches in turn", serial_halves), ("two contexts, half-batches together", together),
                     ("one context, whole batches (again)", whole)):
        torch.cuda.synchronize()
        fn()                                  # warm-up (window array, scratch)
        torch.cuda.synchronize()
        if name.startswith("one context, whole") and "again" not in name:
            ref.copy_(out)
        else:
            same = bool(torch.equal(ref[:2 * half * bk.HIT_DTYPE.itemsize].view(-1, bk.HIT_DTYPE.itemsize)[:, :19],
                                    out[:2 * half * bk.HIT_DTYPE.itemsize].view(-1, bk.HIT_DTYPE.itemsize)[:, :19]))
            print(f"  {name}: first 19 bytes (all but the flags byte) of every result record equal to the whole batch's: {same}", flush=True)
        t0 = time.time()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / steps
        res[name] = dt
        print(f"{name}: {dt * 1e3:.1f} ms per step of {2 * half} reads = {2 * half / dt / 1e6:.1f} M reads/s", flush=True)
    for al in als:
        al.close()


if __name__ == "__main__":
    main()
