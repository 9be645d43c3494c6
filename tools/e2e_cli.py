#!/usr/bin/env python3
"""End-to-end timing of our command line (`biokanga_amd/bin/biokanga align`) on the bench workload:
T_e2e (process start -> exit) and the phases from its time-stamped log.  Files live in /dev/shm.
  python tools/e2e_cli.py [n_reads]"""
import os, sys, time, subprocess, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import biokanga_amd as bk
from biokanga_amd import synth
import bench

def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(3_100_000_000, dev, seed=38)
    n = seq.numel()
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), 4, 0)
    entries = synth.entry_table(seq_lens)
    rd_bases, _, _, _ = synth.make_reads(seq, seq_lens, n_reads, 100, dev, seed=1000, max_subs=3)
    seq_h, sa_h, reads_h = seq.cpu().numpy(), sa.cpu().numpy(), rd_bases.cpu().numpy()
    del seq, sa, rd_bases
    torch.cuda.empty_cache()
    tmp = tempfile.mkdtemp(prefix="bk_e2e_", dir="/dev/shm")
    try:
        sfx, fa, sam, logf = (os.path.join(tmp, x) for x in ("genome.sfx", "reads.fa", "out.sam", "log.txt"))
        bench.write_sfx_file(sfx, seq_h, sa_h, [(f"chr{e[0]}", e[1]) for e in entries])
        bench.write_fasta_file(fa, reads_h, n_reads, 100)
        del seq_h, sa_h, reads_h
        t = time.time()
        rc = subprocess.run([os.path.join(ROOT, "biokanga_amd", "bin", "biokanga"), "align", "-i", fa, "-I", sfx, "-o", sam,
                             "-s3", "-M6", "-F", logf] + sys.argv[2:], stdout=subprocess.DEVNULL, stderr=open(logf + ".err", "w"), env=dict(os.environ, BK_TIMING="1")).returncode
        wall = time.time() - t
        print(f"rc {rc}; T_e2e {wall:.2f} s for {n_reads} reads = {n_reads / wall / 1e6:.2f} M reads/s; SAM {os.path.getsize(sam) / 1e9:.2f} GB")
        # process start -> first log line, last log line -> process gone (what the exit takes: the address space, the device memory)
        import datetime, re
        stamps = []
        for line in open(logf, errors="replace"):
            m = re.match(r"\[(\w+\s+\d+ \d+:\d+:\d+\.\d+ \d+)\]", line)
            if m:
                stamps.append(datetime.datetime.strptime(re.sub(r"\s+", " ", m.group(1)), "%b %d %H:%M:%S.%f %Y").timestamp())
        if stamps:
            print(f"start-up (spawn -> first log line) {stamps[0] - t:.2f} s; log span {stamps[-1] - stamps[0]:.2f} s; tear-down (last log line -> process gone) {t + wall - stamps[-1]:.2f} s")
        keys = ("Loading suffix", "suffix array loaded", "Loading reads", "Load:", "Now aligning", "Alignment of", "Sorting",
                "Header written", "Completed reporting", "Reporting of aligned result set completed", "phase:", "Device pipeline", "window array", "Exit code")
        print(open(logf + ".err").read())
        for line in open(logf, errors="replace"):
            if any(k in line for k in keys):
                print(line.rstrip())
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

if __name__ == "__main__":
    main()
