#!/usr/bin/env python3
"""T_e2e of `biokanga index` on a genome of BASELINE size: a 3.1 Gbp FASTA (24 sequences with bench.py's length ratios, 70 bases a line, N
runs) is written to /dev/shm and indexed by our command line with the stage clocks on (BK_TIMING=1); the .sfx is then read back by
`biokanga align` on a handful of reads as a smoke check.
  python tools/index_e2e.py [genome_mbp = 3100] [--repeat N]"""
import os, subprocess, sys, tempfile, time, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from biokanga_amd import synth


def main():
    mbp = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 3100.0
    repeat = int(sys.argv[sys.argv.index("--repeat") + 1]) if "--repeat" in sys.argv else 2
    rng = np.random.default_rng(38)
    ratios = np.array(synth.GRCH38_LENS, dtype=np.float64)
    lens = np.maximum(1000, (ratios / ratios.sum() * mbp * 1e6).astype(np.int64))
    tmp = tempfile.mkdtemp(prefix="bk_index_e2e_", dir="/dev/shm")
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    try:
        fa, sfx = os.path.join(tmp, "genome.fa"), os.path.join(tmp, "genome.sfx")
        t = time.time()
        with open(fa, "wb") as f:
            for si, G in enumerate(lens):
                G = int(G) // 70 * 70
                g = lut[rng.integers(0, 4, G, dtype=np.uint8)]
                for _ in range(3):
                    L = int(rng.integers(10_000, 2_000_000)); p = int(rng.integers(0, G - L)); g[p:p + L] = ord("N")
                f.write(f">chr{si + 1} synthetic\n".encode())
                body = np.empty((G // 70, 71), dtype=np.uint8)
                body[:, :70] = g.reshape(-1, 70)
                body[:, 70] = 10
                f.write(body.tobytes())
        print(f"genome: {os.path.getsize(fa) / 1e9:.2f} GB of FASTA written in {time.time() - t:.0f} s")
        exe = os.path.join(ROOT, "biokanga_amd", "bin", "biokanga")
        for rep in range(repeat):
            if os.path.exists(sfx):
                os.unlink(sfx)
            t = time.time()
            r = subprocess.run([exe, "index", "-i", fa, "-o", sfx, "-r", "synthetic"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, BK_TIMING="1"))
            wall = time.time() - t
            print(f"== run {rep}: rc {r.returncode}; T_e2e {wall:.2f} s; .sfx {os.path.getsize(sfx) / 1e9:.2f} GB")
            for line in r.stdout.splitlines():
                if "timing" in line or "Error" in line or "Fatal" in line:
                    print("  ", line[:200])
        reads = os.path.join(tmp, "r.fa")
        with open(fa, "rb") as f:
            f.readline()
            seq = f.read(71 * 40).replace(b"\n", b"")
        with open(reads, "wb") as f:
            for i in range(20):
                f.write(b">r%d\n" % i + seq[i * 100:i * 100 + 100] + b"\n")
        r = subprocess.run([exe, "align", "-i", reads, "-I", sfx, "-o", os.path.join(tmp, "o.sam"), "-M6", "-s3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        n_acc = sum(1 for ln in open(os.path.join(tmp, "o.sam")) if not ln.startswith("@") and ln.split("\t")[1] in ("0", "16")) if r.returncode == 0 else -1
        print(f"align on the new index: rc {r.returncode}, {n_acc} of 20 reads aligned")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
