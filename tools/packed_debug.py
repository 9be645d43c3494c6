"""debug: bk_align_batch_packed vs bk_align_batch on a synthetic genome, 4- and 5-byte elements"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biokanga_amd as bk
from biokanga_amd import synth

dev = torch.device("cuda", 0)
seq, seq_lens = synth.make_genome(60_000_000, dev, seed=17, n_seqs=5, repeat_frac=0.85)
n = seq.numel()
entries = synth.entry_table(seq_lens)
ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
for i, (eid, slen, so, eo) in enumerate(entries):
    ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
pb, po, pl = synth.make_pairs(seq, seq_lens, 100_000, 150, dev, seed=6, max_subs=5)
bases, offs, lens = pb.cpu().numpy(), po.cpu().numpy().astype(np.uint64), pl.cpu().numpy().astype(np.uint32)
print("offs contiguous:", bool(np.array_equal(offs, np.arange(len(lens), dtype=np.uint64) * 150)))


def diff(a, b):
    return np.unique(np.nonzero(a.view(np.uint8).reshape(-1, 20) != b.view(np.uint8).reshape(-1, 20))[0])


for E in (4,):
    sa = torch.empty(n * E, dtype=torch.uint8, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), E, 0)
    bb, ll, oo = bases[: 100000 * 150], lens[:100000], offs[:100000]
    words, l16, exc = bk.pack_reads(bb, None, ll)
    for knobs in ((), (("chunk_reads", 65536),), (("chunk_reads", 50000),), (("use_tgt2", 0),), (("use_flat", 0),), (("use_wave", 0),), (("use_k2", 0),)):
        with bk.Aligner(None, bk.AlignParams(max_subs=5), d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=E, entries=ent) as al:
            for kv in knobs:
                al.tune(*kv)
            b = al.align_packed(words, l16, exc)
            a = al.align(bb, oo, ll)
            b2 = al.align_packed(words, l16, exc)
            d, d2 = diff(a, b), diff(a, b2)
            print(f"knobs {knobs}: packed FIRST vs bytes {len(d)} differ (first {d[:4]}), packed after bytes {len(d2)} differ (first {d2[:4]}); strands {a['strand'][d[:8]]}")
