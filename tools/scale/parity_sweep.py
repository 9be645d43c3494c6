#!/usr/bin/env python3
"""Parity at full genome scale across option sets: for every set, a sample of the bench workload (and of a
mixed-length variant) is aligned on the GPU and by the CPU oracle on the same 3.1 Gbp index; every bk_hit field
and the n_search / n_cand / n_lcm counters must agree.  Prints one line per set.
  python tools/scale/parity_sweep.py [reads_per_set]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import biokanga_amd as bk
from biokanga_amd import synth
import helpers

FIELDS = ["chrom_id", "match_loci", "match_len", "low_hit_instances", "rslt", "nar", "strand", "low_mm", "nxt_low_mm", "num_hits", "mismatches"]
SETS = [dict(max_subs=3), dict(max_subs=0), dict(max_subs=5), dict(), dict(max_subs=3, min_edit_dist=2), dict(max_subs=3, align_strand=1),
        dict(max_subs=3, align_strand=2), dict(max_subs=3, pmode=1), dict(max_subs=3, pmode=2), dict(max_subs=3, pmode=3),
        dict(max_subs=3, max_ns=0), dict(max_subs=3, max_ns=3)]

def main():
    nr = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(3_100_000_000, dev, seed=38)
    n = seq.numel()
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), 4, 0)
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    # reads: 100 bp with 0..5 subs, plus a ragged set (50..150 bp) cut from 150 bp reads
    b100, _, _, _ = synth.make_reads(seq, seq_lens, nr, 100, dev, seed=77, max_subs=5)
    b150, _, _, _ = synth.make_reads(seq, seq_lens, nr // 4, 150, dev, seed=78, max_subs=6)
    b100 = b100.cpu().numpy(); b150 = b150.cpu().numpy().reshape(-1, 150)
    rng = np.random.default_rng(5)
    lens_r = rng.integers(50, 151, size=nr // 4).astype(np.uint32)
    ragged = np.concatenate([b150[i, :lens_r[i]] for i in range(nr // 4)])
    bases = np.concatenate([b100, ragged])
    lens = np.concatenate([np.full(nr, 100, np.uint32), lens_r])
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
    seq_h, sa_h = seq.cpu().numpy(), sa.cpu().numpy()
    ora = helpers.OracleSfx(seq=seq_h, sa=sa_h, el_size=4, entries=entries)
    al = bk.Aligner(None, bk.AlignParams(max_subs=3), device=0, d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=4, entries=ent)
    del seq, sa
    torch.cuda.empty_cache()
    bad_total = 0
    for kw in SETS:
        al.set_params(bk.AlignParams(**kw))
        al.counters(reset=True)
        t = time.time()
        got = al.align(bases, offs, lens)
        tg = time.time() - t
        c = al.counters()
        t = time.time()
        exp, octr = ora.align(bases, offs, lens, helpers.make_params(**kw), nthreads=os.cpu_count())
        to = time.time() - t
        bad = sum(int((got[f] != exp[f]).sum()) for f in FIELDS)
        cbad = int(c["n_search"] != octr.n_search) + int(c["n_cand"] != octr.n_cand) + int(c["n_lcm_calls"] != octr.n_lcm_calls)
        bad_total += bad + cbad
        nar, cnt = np.unique(got["nar"], return_counts=True)
        print(f"{kw}: {len(lens)} reads, mismatching fields {bad}, counters differing {cbad}, GPU {tg:.2f}s oracle {to:.1f}s, "
              f"NAR {dict(zip([bk.NAR_TAGS[int(k)] for k in nar], cnt.tolist()))}", flush=True)
    print("TOTAL mismatches", bad_total)

if __name__ == "__main__":
    main()
