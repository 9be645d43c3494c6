#!/usr/bin/env python3
"""Multi-loci modes at C2 scale: the bench workload (100 bp SE reads, 0-3 subs, -s3, synthetic GRCh38-like genome) aligned
with MaxHits = -R (default 5): time of the align call with and without the loci lists, share of reads with several loci,
and a sample checked against the CPU oracle (result records and the pHits[] lists, in order).
  python tools/scale/multi_bench.py [n_reads] [genome_mbp] [max_ml] [clamp] [best_matches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import biokanga_amd as bk
from biokanga_amd import synth
import helpers

def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    mbp = float(sys.argv[2]) if len(sys.argv) > 2 else 3100.0
    max_ml = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    clamp = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    best = int(sys.argv[5]) if len(sys.argv) > 5 else 0            # 1: -N (LocateBestMatches)
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(int(mbp * 1e6), dev, seed=38)
    n = seq.numel()
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), 4, 0)
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    bases, offs, lens = synth.make_reads(seq, seq_lens, n_reads, 100, dev, seed=2, max_subs=3)[:3]
    out = torch.zeros(n_reads * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    res = {}
    for ml, cl in ((1, 0), (max_ml, clamp)):
        al = bk.Aligner(None, bk.AlignParams(max_subs=3, max_ml=ml, clamp_ml=cl, best_matches=(best if ml > 1 else 0)), device=0, d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(),
                        el_size=4, entries=ent)
        for it in range(2):
            al.timing(reset=True)
            torch.cuda.synchronize(); t = time.time()
            al.align_device(bases.data_ptr(), offs.data_ptr(), lens.data_ptr(), n_reads, out.data_ptr())
            torch.cuda.synchronize(); dt = time.time() - t
        hits = out.cpu().numpy().view(bk.HIT_DTYPE).copy()
        lo, loci = al.batch_loci(n_reads)
        res[ml] = (dt, hits, lo, loci, al.timing())
        nar, cnt = np.unique(hits["nar"], return_counts=True)
        print(f"max_ml {ml} clamp {cl}: {n_reads} reads in {dt * 1e3:.1f} ms = {n_reads / dt / 1e6:.1f} M reads/s (align call incl. loci lists back on the host); "
              f"device ms {al.timing()['ms_total']:.1f}; NAR {({bk.NAR_TAGS[int(k)]: int(v) for k, v in zip(nar, cnt)})}")
        if ml > 1:
            c = np.diff(lo.astype(np.int64))
            print(f"  loci lists: {len(loci)} loci; reads with 1 locus {int((c == 1).sum())}, with 2..{ml} loci {int((c > 1).sum())} "
                  f"({100.0 * (c > 1).sum() / n_reads:.2f} % of reads)")
        al.close()
    dt, hits, lo, loci, _ = res[max_ml]
    ns = min(n_reads, 200_000)
    b_h, o_h, l_h = bases[: ns * 100].cpu().numpy(), offs[:ns].cpu().numpy().astype(np.uint64), lens[:ns].cpu().numpy().astype(np.uint32)
    ora = helpers.OracleSfx(seq=seq.cpu().numpy(), sa=sa.cpu().numpy(), el_size=4, entries=entries)
    exp, eo, el = helpers.oracle_align_multi(ora, b_h, o_h, l_h, helpers.make_params(max_subs=3, max_ml=max_ml, clamp_ml=clamp, best_matches=best), nthreads=os.cpu_count())
    fields = ["chrom_id", "match_loci", "match_len", "low_hit_instances", "rslt", "nar", "strand", "low_mm", "nxt_low_mm", "num_hits", "mismatches"]
    bad = sum(int((hits[:ns][f] != exp[f]).sum()) for f in fields)
    ok_offs = bool(np.array_equal(lo[: ns + 1], eo))
    nl = int(eo[-1])
    bad_l = sum(int((loci[:nl][f] != el[f]).sum()) for f in ("chrom_id", "match_loci", "match_len", "strand", "mismatches")) if ok_offs else -1
    print(f"oracle check on the first {ns} reads: mismatching result fields {bad}; list offsets equal {ok_offs}; mismatching loci fields {bad_l} of {nl} loci")

if __name__ == "__main__":
    main()
