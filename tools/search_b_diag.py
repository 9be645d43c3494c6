#!/usr/bin/env python3
"""What pass B of the search loads, counted by a build with -DBK_DIAG_B (tools/build_variant.sh diagb "-DBK_DIAG_B" bk_search.hip; run with
BK_LIB=biokanga_amd/lib/libbiokanga_amd_diagb.so): items and 64-byte lines of the key bisections (second- and third-level keys) and of the
bisection over suffix array + target that is left after them, for one C2 step - with both key arrays behind the second-level keys, one, none.

usage: search_b_diag.py [genome_mbp] [reads]"""
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
    b, o, l, _ = synth.make_reads(seq, seq_lens, n_reads, 100, dev, seed=1000, max_subs=3)
    out = torch.zeros(n_reads * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    for k3 in (2, 1, 0):
        al.tune("use_k3", k3)
        al.counters(reset=True)
        al.timing(reset=True)
        al.align_device(b.data_ptr(), o.data_ptr(), l.data_ptr(), n_reads, out.data_ptr())
        c, t = al.counters(), al.timing()
        keys, rest = c["reserved0"], c["reserved1"]
        print(f"{al.tune('k3_resident', 0)} key array(s) behind the second-level keys: key bisections {keys & 0xFFFFFFFF:,} items, {keys >> 32:,} lines; "
              f"suffix array + target after them {rest & 0xFFFFFFFF:,} items, {rest >> 32:,} lines; pass B {t['ms_search_b']:.2f} ms, pass A {t['ms_search_a']:.2f} ms "
              f"({c['n_search']:,} searches)", flush=True)
    al.close()


if __name__ == "__main__":
    main()
