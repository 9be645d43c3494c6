#!/usr/bin/env python3
"""C1 of SURVEY.md §8d (the reference's own CPU-runnable case): one 4.6 Mbp sequence of uniform bases (seed 4600), 1 M x 100 bp
SE reads without substitutions named like simreads output (seed 1), `index` + `align -s0 -M6` through the REAL reference
(oracle/_ref/biokanga, all host threads) and through our command line; the two .sfx and the two SAM files are compared
byte for byte and the wall-clock times reported.
  python tools/scale/c1_check.py [n_reads]"""
import os, sys, time, subprocess, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def run(cmd):
    t = time.time()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return r.returncode, time.time() - t, r.stdout

def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ref = os.path.join(ROOT, "oracle", "_ref", "biokanga")
    ours = os.path.join(ROOT, "biokanga_amd", "bin", "biokanga")
    if not os.path.exists(ref):
        raise SystemExit("oracle/_ref/biokanga is missing (built by __graft_entry__.build() where /root/reference exists)")
    rng = np.random.default_rng(4600)
    g = rng.integers(0, 4, 4_600_000, dtype=np.uint8)
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    tmp = tempfile.mkdtemp(prefix="bk_c1_", dir="/dev/shm")
    try:
        fa, rd = os.path.join(tmp, "ecoli.fa"), os.path.join(tmp, "reads.fa")
        with open(fa, "wb") as f:
            f.write(b">chrE synthetic 4.6 Mbp\n")
            s = asc[g]
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70].tobytes() + b"\n")
        rr = np.random.default_rng(1)
        starts = rr.integers(0, len(g) - 100, n_reads)
        strand = rr.integers(0, 2, n_reads)
        with open(rd, "wb") as f:
            for i in range(n_reads):
                st = int(starts[i])
                b = g[st:st + 100]
                if strand[i]:
                    b = comp[b[::-1]]
                f.write(b">lcl|usimreads|%08d|chrE|%d|%d|100|%s|0|0|0\n" % (i + 1, st, st + 99, b"-" if strand[i] else b"+") + asc[b].tobytes() + b"\n")
        out = {}
        for who, exe in (("reference", ref), ("ours", ours)):
            sfx, sam = os.path.join(tmp, who + ".sfx"), os.path.join(tmp, who + ".sam")
            rc1, t1, l1 = run([exe, "index", "-i", fa, "-o", sfx, "-r", "ecoli"])
            rc2, t2, l2 = run([exe, "align", "-i", rd, "-I", sfx, "-o", sam, "-s0", "-M6"])
            if rc1 or rc2:
                print(l1[-1500:], l2[-1500:])
                raise SystemExit(f"{who}: index rc {rc1}, align rc {rc2}")
            out[who] = (sfx, sam, t1, t2)
            print(f"{who}: index {t1:.2f} s, align {t2:.2f} s ({n_reads / t2 / 1e6:.2f} M reads/s end to end)")
        same_sfx = open(out["reference"][0], "rb").read() == open(out["ours"][0], "rb").read()
        a, b = open(out["reference"][1], "rb").read(), open(out["ours"][1], "rb").read()
        n_acc = sum(1 for l in b.split(b"\n") if l and not l.startswith(b"@") and l.split(b"\t")[1] in (b"0", b"16"))
        print(f".sfx byte-identical: {same_sfx}; SAM byte-identical: {a == b} ({len(b)} bytes, {n_acc} of {n_reads} reads aligned)")
        # our align on the reference's index as well (drop-in for that path alone)
        sam2 = os.path.join(tmp, "ours_on_ref.sam")
        rc, t, _ = run([ours, "align", "-i", rd, "-I", out["reference"][0], "-o", sam2, "-s0", "-M6"])
        print(f"ours on the reference's .sfx: rc {rc}, {t:.2f} s, SAM byte-identical to the reference's: {open(sam2, 'rb').read() == a}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

if __name__ == "__main__":
    main()
