#!/usr/bin/env python3
"""Paired-end end-to-end parity at genome scale: 2 x 150 bp FR pairs against the 3.1 Gbp synthetic index are run
through the real reference (oracle/_ref/biokanga align -U3 -d200 -D400 -s5 -M6) and through our command line;
the two SAM files must be byte-identical.  The pair count is limited by the reference's loader (2 M reads).
  python tools/scale/pe_e2e.py [n_pairs]"""
import os, sys, time, subprocess, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import biokanga_amd as bk
from biokanga_amd import synth
import bench

def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(3_100_000_000, dev, seed=38)
    n = seq.numel()
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), 4, 0)
    entries = synth.entry_table(seq_lens)
    bases, _, _ = synth.make_pairs(seq, seq_lens, n_pairs, 150, dev, seed=3)
    b = bases.cpu().numpy().reshape(n_pairs, 2, 150)
    seq_h, sa_h = seq.cpu().numpy(), sa.cpu().numpy()
    del seq, sa, bases
    torch.cuda.empty_cache()
    tmp = tempfile.mkdtemp(prefix="bk_pe_", dir="/dev/shm")
    try:
        sfx = os.path.join(tmp, "genome.sfx")
        bench.write_sfx_file(sfx, seq_h, sa_h, [(f"chr{e[0]}", e[1]) for e in entries])
        f1, f2 = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa")
        bench.write_fasta_file(f1, np.ascontiguousarray(b[:, 0, :]).reshape(-1), n_pairs, 150)
        bench.write_fasta_file(f2, np.ascontiguousarray(b[:, 1, :]).reshape(-1), n_pairs, 150)
        res = {}
        for tag, binary, extra in (("ref", os.path.join(ROOT, "oracle", "_ref", "biokanga"), ["-T0"]),
                                   ("ours", os.path.join(ROOT, "biokanga_amd", "bin", "biokanga"), [])):
            out = os.path.join(tmp, tag + ".sam")
            t = time.time()
            r = subprocess.run([binary, "align", "-i", f1, "-u", f2, "-I", sfx, "-o", out, "-s5", "-U3", "-d200", "-D400", "-M6",
                                "-F", os.path.join(tmp, tag + ".log")] + extra, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
            res[tag] = (r.returncode, time.time() - t, out)
            print(f"{tag}: rc {r.returncode}, {time.time() - t:.1f} s", flush=True)
        same = subprocess.run(["cmp", "-s", res["ref"][2], res["ours"][2]]).returncode == 0
        print(f"{n_pairs} pairs: SAM files byte-identical: {same} ({os.path.getsize(res['ref'][2]) / 1e6:.0f} MB)")
        for line in open(os.path.join(tmp, "ref.log"), errors="replace"):
            if "accepted as paired" in line or "Accepted" in line and "pair" in line.lower():
                print(line.rstrip()[:200])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

if __name__ == "__main__":
    main()
