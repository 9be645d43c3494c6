#!/usr/bin/env python3
"""SNP pile-up and screening at C2 scale: the bench workload (100 bp SE reads, 0-3 subs, -s3, synthetic GRCh38-like genome) aligned on
the device, every accepted read piled up over the 6 count planes in HBM (bk_snp_pileup, host-resident reads as the CLI hands them over),
every sequence screened (bk_snp_sites); wall time of each step, and the smallest sequences checked against the CPU oracle.
  python tools/scale/snp_bench.py [n_reads] [genome_mbp]"""
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
    al = bk.Aligner(None, bk.AlignParams(max_subs=3), device=0, d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=4, entries=ent)
    al.align_device(bases.data_ptr(), offs.data_ptr(), lens.data_ptr(), n_reads, out.data_ptr())
    torch.cuda.synchronize()
    hits = out.cpu().numpy().view(bk.HIT_DTYPE).copy()
    b_h, o_h, l_h = bases.cpu().numpy(), offs.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32)
    acc = np.nonzero(hits["nar"] == 1)[0]
    alns = np.zeros(len(acc), dtype=bk.SNP_ALN_DTYPE)
    alns["read_idx"] = acc; alns["chrom_id"] = hits["chrom_id"][acc]; alns["loci"] = hits["match_loci"][acc]
    alns["len"] = hits["match_len"][acc]; alns["strand"] = hits["strand"][acc]
    for it in range(2):
        t0 = time.time(); al.snp_reset(); t1 = time.time()
        al.snp_pileup(b_h, o_h, l_h, alns); t2 = time.time()
        n_sites, tot = 0, np.zeros(4, dtype=np.uint64)
        per = {}
        for eid, slen, so, eo in entries:
            s, t = al.snp_sites(eid, 5, 0.25)
            per[eid] = (s, t); n_sites += len(s); tot += t
        t3 = time.time()
        print(f"pass {it}: reset of {6 * 4 * n / 1e9:.1f} GB of counts {(t1 - t0) * 1e3:.1f} ms; pile-up of {len(alns)} alignments ({len(alns) * 100 / 1e9:.2f} G bases, reads over PCIe) "
              f"{(t2 - t1) * 1e3:.1f} ms; screening of {len(entries)} sequences ({n / 1e9:.2f} G loci) {(t3 - t2) * 1e3:.1f} ms -> {n_sites} putative loci at -p5 -1 25; "
              f"covered loci {int(tot[2])}, coverage {int(tot[3])} bases")
    d_alns = torch.from_numpy(alns.view(np.uint8)).to(dev)
    for it in range(2):
        al.snp_reset()
        torch.cuda.synchronize(); t0 = time.time()
        al.snp_pileup_device(bases.data_ptr(), offs.data_ptr(), n_reads, d_alns.data_ptr(), len(alns))
        t1 = time.time()
    s, t = al.snp_sites(entries[0][0], 5, 0.25)
    same = np.array_equal(t, per[entries[0][0]][1]) and all(np.array_equal(s[f], per[entries[0][0]][0][f]) for f in s.dtype.names)
    print(f"pile-up with reads and alignments resident in HBM: {(t1 - t0) * 1e3:.1f} ms = {len(alns) * 100 / (t1 - t0) / 1e9:.0f} G bases/s; sequence {entries[0][0]} "
          f"{'identical to' if same else 'DIFFERENT from'} the host-buffer run")
    small = sorted(entries, key=lambda e: e[1])[:2]
    ora = helpers.OracleSfx(seq=seq.cpu().numpy(), sa=np.zeros(1, dtype=np.uint32), el_size=4, entries=entries)
    bad = 0
    for eid, slen, so, eo in small:
        exp, etot = helpers.oracle_snp_sites(ora.h, b_h, o_h, alns, eid, 5, 0.25, max_sites=1 << 22)
        got, gtot = per[eid]
        same = len(exp) == len(got) and np.array_equal(etot, gtot) and all(np.array_equal(exp[f], got[f]) for f in exp.dtype.names)
        bad += 0 if same else 1
        print(f"oracle check of sequence {eid} ({slen} bases): {len(exp)} sites, totals {etot.tolist()} -> {'identical' if same else 'DIFFERENT'}")
    print("oracle check:", "no disagreement" if bad == 0 else f"{bad} sequences differ")
    al.close()

if __name__ == "__main__":
    main()
