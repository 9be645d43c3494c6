#!/usr/bin/env python3
"""microInDels at C2 scale: the bench workload (100 bp SE reads, 0-3 subs, -s3, synthetic GRCh38-like genome) with a share of the
reads carrying a 1..8-base insertion or deletion, aligned with and without -a10: time of the extra pass, reads recovered,
and a sample checked against the CPU oracle (result records and second segments).
  python tools/scale/indel_bench.py [n_reads] [genome_mbp] [indel_frac]"""
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
    frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
    # 4th argument: extra alignment parameters of the second run, e.g. "micro_indel_len=10" (default), "splice_junct_len=5000",
    # "min_chimeric_len=50", or several separated by commas
    extra = dict(kv.split("=") for kv in (sys.argv[4] if len(sys.argv) > 4 else "micro_indel_len=10").split(","))
    extra = {k: int(v) for k, v in extra.items()}
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(int(mbp * 1e6), dev, seed=38)
    n = seq.numel()
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), 4, 0)
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    # 108-base reads; a share of them loses 1..8 bases at a random inner position (deletion from the read) or gets 1..8 bases
    # of another read spliced in (insertion); everything is cut back to 100 bases
    bases, offs, lens = synth.make_reads(seq, seq_lens, n_reads, 108, dev, seed=2, max_subs=3)[:3]
    g = torch.Generator(device=dev); g.manual_seed(7)
    rd = bases.view(n_reads, 108)
    sel = torch.rand(n_reads, generator=g, device=dev) < frac
    L = torch.randint(1, 9, (n_reads,), generator=g, device=dev)
    pos = torch.randint(20, 80, (n_reads,), generator=g, device=dev)
    is_ins = torch.rand(n_reads, generator=g, device=dev) < 0.5
    col = torch.arange(100, device=dev)[None, :]
    src_del = col + (col >= pos[:, None]) * L[:, None]                      # skip L read bases: deletion of genome bases? no: the read lacks them
    out_del = torch.gather(rd, 1, src_del.clamp(max=107))
    src_ins = col - ((col >= pos[:, None] + L[:, None]) * L[:, None])       # repeat nothing: bases pos..pos+L come from a shifted row
    out_ins = torch.gather(rd, 1, src_ins.clamp(min=0))
    foreign = torch.roll(rd, 1, 0)[:, :100]
    ins_zone = (col >= pos[:, None]) & (col < pos[:, None] + L[:, None])
    out_ins = torch.where(ins_zone, foreign, out_ins)
    plain = rd[:, :100]
    reads = torch.where(sel[:, None], torch.where(is_ins[:, None], out_ins, out_del), plain).contiguous()
    bases = reads.view(-1)
    offs = torch.arange(n_reads, device=dev, dtype=torch.int64) * 100
    lens = torch.full((n_reads,), 100, device=dev, dtype=torch.int32)
    out = torch.zeros(n_reads * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    res = {}
    for a in (0, 10):
        al = bk.Aligner(None, bk.AlignParams(max_subs=3, **(extra if a else {})), device=0, d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=4, entries=ent)
        for it in range(2):
            al.timing(reset=True)
            torch.cuda.synchronize(); t = time.time()
            al.align_device(bases.data_ptr(), offs.data_ptr(), lens.data_ptr(), n_reads, out.data_ptr())
            torch.cuda.synchronize(); dt = time.time() - t
        hits = out.cpu().numpy().view(bk.HIT_DTYPE).copy()
        seg = al.batch_seg2() if a else None
        res[a] = (hits, seg)
        nar, cnt = np.unique(hits["nar"], return_counts=True)
        print(f"{extra if a else 'plain'}: {n_reads} reads in {dt * 1e3:.1f} ms = {n_reads / dt / 1e6:.1f} M reads/s; NAR {({bk.NAR_TAGS[int(k)]: int(v) for k, v in zip(nar, cnt)})}"
              + (f"; placed with a microInDel {int((seg['flags'] & 1).sum())} (insertions {int(((seg['flags'] & 3) == 3).sum())}), with a splice junction "
                 f"{int(((seg['flags'] & 4) != 0).sum())}, end-trimmed {int(((seg['flags'] & 8) != 0).sum())}" if a else ""))
        al.close()
    hits, seg = res[10]
    ns = min(n_reads, 200_000)
    b_h, o_h, l_h = bases[: ns * 100].cpu().numpy(), offs[:ns].cpu().numpy().astype(np.uint64), lens[:ns].cpu().numpy().astype(np.uint32)
    ora = helpers.OracleSfx(seq=seq.cpu().numpy(), sa=sa.cpu().numpy(), el_size=4, entries=entries)
    exp, eseg = helpers.oracle_align_indel(ora, b_h, o_h, l_h, helpers.make_params(max_subs=3, **extra), nthreads=os.cpu_count())
    fields = ["chrom_id", "match_loci", "match_len", "low_hit_instances", "rslt", "nar", "strand", "low_mm", "nxt_low_mm", "num_hits", "mismatches"]
    bad = sum(int((hits[:ns][f] != exp[f]).sum()) for f in fields)
    bad2 = sum(int((seg[:ns][f] != eseg[f]).sum()) for f in ("match_loci", "match_len", "read_ofs", "mismatches", "flags", "score"))
    shown = 0
    for i in range(ns):
        if any(hits[i][f] != exp[i][f] for f in fields) or any(seg[i][f] != eseg[i][f] for f in ("match_loci", "match_len", "read_ofs", "mismatches", "flags", "score")):
            print("  read", i, "gpu", hits[i], seg[i], "oracle", exp[i], eseg[i])
            shown += 1
            if shown >= 6:
                break
    print(f"oracle check on the first {ns} reads: mismatching result fields {bad}, mismatching second-segment fields {bad2}; "
          f"two-segment / trimmed placements in the sample {int((eseg['flags'] != 0).sum())}")

if __name__ == "__main__":
    main()
