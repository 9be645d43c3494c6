#!/usr/bin/env python3
"""C3-shaped sanity run: 2 x 150 bp FR pairs, insert ~N(300,50) in [200,400], 0..5 subs per
read, `-s5 -U3 -d200 -D400` against the synthetic GRCh38-like genome: SE pass (device-resident), then
bk_pair_batch; a sample is checked against the CPU oracle (SE fields and PE outcome).
  python tools/scale/pe_bench.py [n_pairs] [genome_mbp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import biokanga_amd as bk
from biokanga_amd import synth
import helpers

def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
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
    al = bk.Aligner(None, bk.AlignParams(max_subs=5), device=0, d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=4, entries=ent)
    bases, offs, lens = synth.make_pairs(seq, seq_lens, n_pairs, 150, dev, seed=3)
    nr = 2 * n_pairs
    out = torch.zeros(nr * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    prev = None
    for it in range(2):
        al.timing(reset=True)
        torch.cuda.synchronize(); t = time.time()
        al.align_device(bases.data_ptr(), offs.data_ptr(), lens.data_ptr(), nr, out.data_ptr())
        torch.cuda.synchronize(); dt = time.time() - t
        if prev is not None:
            print("SE pass repeatable (bitwise equal results of two runs):", bool(torch.equal(prev, out)))
        prev = out.clone()
    print(f"SE pass: {nr} reads of 150 bp, -s5: {dt * 1e3:.1f} ms = {nr / dt / 1e6:.1f} M reads/s; device {al.timing()}")
    print("counters", al.counters())
    hits = out.cpu().numpy().view(bk.HIT_DTYPE).copy()
    b_h, o_h, l_h = bases.cpu().numpy(), offs.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32)
    pe = bk.PEParams(3, 200, 400, False)
    t = time.time()
    paired = al.pair(b_h, o_h, l_h, hits.copy(), pe)
    dt = time.time() - t
    acc = int(((paired["flags"][0::2] & 0x80) != 0).sum())
    print(f"PE association (-U3 -d200 -D400, host buffers in/out): {dt * 1e3:.1f} ms = {n_pairs / dt / 1e6:.2f} M pairs/s; {acc} of {n_pairs} accepted as pairs")
    out2 = out.clone()
    torch.cuda.synchronize(); t = time.time()
    al.pair_device(bases.data_ptr(), offs.data_ptr(), lens.data_ptr(), n_pairs, out2.data_ptr(), pe)
    torch.cuda.synchronize(); dt2 = time.time() - t
    same = np.array_equal(out2.cpu().numpy().view(bk.HIT_DTYPE), paired)
    print(f"PE association, device-resident: {dt2 * 1e3:.1f} ms = {n_pairs / dt2 / 1e6:.1f} M pairs/s; identical to the host-buffer call: {same}")
    nar, cnt = np.unique(paired["nar"], return_counts=True)
    print("NAR histogram", {bk.NAR_TAGS[int(k)]: int(v) for k, v in zip(nar, cnt)})
    # oracle check on a sample
    ns = min(n_pairs, 100_000)
    seq_h, sa_h = seq.cpu().numpy(), sa.cpu().numpy()
    ora = helpers.OracleSfx(seq=seq_h, sa=sa_h, el_size=4, entries=entries)
    p = helpers.make_params(max_subs=5)
    sb, so, sl = b_h[: 2 * ns * 150], o_h[: 2 * ns], l_h[: 2 * ns]
    exp, _ = ora.align(sb, so, sl, p, nthreads=os.cpu_count())
    fields = ["chrom_id", "match_loci", "match_len", "low_hit_instances", "rslt", "nar", "strand", "low_mm", "nxt_low_mm", "num_hits", "mismatches"]
    bad_se = sum(int((hits[: 2 * ns][f] != exp[f]).sum()) for f in fields)
    exp_pe = helpers.oracle_process_pe(ora, p, 3, 200, 400, False, sb, so, sl, exp.copy())
    bad_pe = sum(int((paired[: 2 * ns][f] != exp_pe[f]).sum()) for f in fields if f != "rslt")
    bad_pe += int(((paired[: 2 * ns]["flags"] & 0x80) != (exp_pe["flags"] & 0x80)).sum())
    print(f"oracle check on the first {ns} pairs: SE mismatching fields {bad_se}, PE mismatching fields {bad_pe}")

if __name__ == "__main__":
    main()
