#!/usr/bin/env python3
"""One-off study on the GPU box's HOST cores: how the real reference (oracle/_ref/biokanga) and the C
restatement (oracle/bk_oracle.c) scale with threads on the bench workload, and how many reads the
reference survives (its loader hand-off breaks when loading takes > 3 s).  Not part of the product or
of bench.py; results are quoted in DESIGN.md."""
import os, sys, time, json, subprocess, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import biokanga_amd as bk
from biokanga_amd import synth
import bench, helpers

def main():
    dev = torch.device("cuda", 0)
    total_bp = 3_100_000_000
    seq, seq_lens = synth.make_genome(total_bp, dev, seed=38)
    n = seq.numel()
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), 4, 0)
    entries = synth.entry_table(seq_lens)
    NMAX = 4_000_000
    rd_bases, rd_offs, rd_lens, truth = synth.make_reads(seq, seq_lens, NMAX, 100, dev, seed=1000, max_subs=3)
    seq_h, sa_h, reads_h = seq.cpu().numpy(), sa.cpu().numpy(), rd_bases.cpu().numpy()
    del seq, sa; torch.cuda.empty_cache()
    named = [(f"chr{e[0]}", e[1]) for e in entries]
    out = {}
    # port scaling
    ora = helpers.OracleSfx(seq=seq_h, sa=sa_h, el_size=4, entries=entries)
    p = helpers.make_params(max_subs=3)
    for nt, nr in ((8, 200_000), (32, 500_000), (64, 1_000_000), (128, 2_000_000), (256, 2_000_000)):
        t = time.time()
        ora.align(reads_h[: nr * 100], np.arange(nr, dtype=np.uint64) * 100, np.full(nr, 100, np.uint32), p, nthreads=nt)
        dt = time.time() - t
        out[f"port_T{nt}"] = nr / dt
        print(f"port T{nt}: {nr / dt:.0f} reads/s", flush=True)
    ora.close()
    # reference: monkeypatch the ladder by calling reference_baseline's pieces through its public function
    hits = np.zeros(NMAX, dtype=bk.HIT_DTYPE)      # parity not checked here
    for nr in (1_000_000, 2_000_000, 3_000_000):
        for T in (0, 64, 32, 16, 8):
            bench.REF_LADDER = (T,)
            r = bench.reference_baseline(seq_h, sa_h, named, reads_h, 100, 3, hits, nr, None)
            key = f"ref_n{nr}_T{T}"
            out[key] = {k: r.get(k) for k in ("value", "t_align_s", "t_e2e_s", "sample")} if r else None
            print(key, out[key], flush=True)
            if r and "exited" in str(r.get("sample")):
                break
    print(json.dumps(out))

if __name__ == "__main__":
    main()
