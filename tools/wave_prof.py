#!/usr/bin/env python3
"""Where the wave kernel's cycles go, from a build with section timers (tools/build_variant.sh wprof "-DBK_PROF=3" bk_wave.hip; run with
BK_LIB=biokanga_amd/lib/libbiokanga_amd_wprof.so): lane 0 of every wave adds the cycles between its marks - claiming an item and its read's
plan, a strand pass's set-up (read row, interval records, window array map), a core's set-up, the rounds, a read's end - over one C2 step.

usage: wave_prof.py [genome_mbp] [reads]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100.0
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
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
    al = bk.Aligner(None, bk.AlignParams(max_subs=3), device=0, d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=4, entries=ent)
    del sa
    torch.cuda.empty_cache()
    al.tune("use_swin", 1)
    b, o, l, _ = synth.make_reads(seq, seq_lens, n_reads, 100, dev, seed=1000, max_subs=3)
    out = torch.zeros(n_reads * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    al.align_device(b.data_ptr(), o.data_ptr(), l.data_ptr(), n_reads, out.data_ptr())        # (makes the window array)
    lib = bk.load_library()
    before = (ctypes.c_ulonglong * 8)()
    after = (ctypes.c_ulonglong * 8)()
    assert lib.bk_debug_prof_wave(before) == 0
    al.timing(reset=True)
    al.align_device(b.data_ptr(), o.data_ptr(), l.data_ptr(), n_reads, out.data_ptr())
    t = al.timing()
    assert lib.bk_debug_prof_wave(after) == 0
    d = [after[i] - before[i] for i in range(8)]
    tot = sum(d[:5])
    names = ["item claimed, read's plan", "strand pass set-up (row, interval records, window array map)", "core set-up and what follows a core", "rounds", "read's end (result, next phase's list)"]
    print(f"k_wave {t['ms_heavy']:.2f} ms per step under the timers; {d[5]:,} strand passes, {d[6]:,} rounds ({d[6] / max(1, d[5]):.2f} per pass)")
    for nme, v in zip(names, d[:5]):
        print(f"  {100.0 * v / tot:5.1f} %  {nme}")
    print(f"  cycles per strand pass {tot / max(1, d[5]):.0f}, per round {d[3] / max(1, d[6]):.0f}")
    al.close()


if __name__ == "__main__":
    main()
