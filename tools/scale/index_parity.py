#!/usr/bin/env python3
"""`biokanga index` parity at a size the CPU test-suite cannot afford: a multi-sequence genome with repeat
families, long N runs (so that the kangax N mutation and its libc rand() stream matter) and soft-masked
stretches is indexed by the real reference (oracle/_ref/biokanga) and by our front end; header, bases and entries
must be byte-identical and the suffix arrays may differ only among suffixes tied through an EOS.
  python tools/scale/index_parity.py [genome_mbp]"""
import os, struct, subprocess, sys, tempfile, time, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    rng = np.random.default_rng(99)
    tmp = tempfile.mkdtemp(prefix="bk_idx_", dir="/dev/shm")
    try:
        fa = os.path.join(tmp, "g.fa")
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        with open(fa, "wb") as f:
            for si, frac in enumerate((0.5, 0.3, 0.15, 0.05)):
                G = int(mbp * 1e6 * frac)
                g = lut[rng.integers(0, 4, G)]
                fam = lut[rng.integers(0, 4, 500)]
                for _ in range(G // 20000):                      # a repeat family, 3 % divergence
                    p = int(rng.integers(0, G - 500)); c = fam.copy(); m = rng.random(500) < 0.03
                    c[m] = lut[rng.integers(0, 4, int(m.sum()))]; g[p:p + 500] = c
                for _ in range(3):                               # N runs of 30 .. 50 000
                    L = int(rng.integers(30, 50000)); p = int(rng.integers(0, G - L)); g[p:p + L] = ord("N")
                p = int(rng.integers(0, G - 5000)); g[p:p + 5000] |= 0x20           # soft-masked stretch
                f.write(f">seq{si} synthetic sequence {si}\n".encode())
                body = g[: G // 70 * 70].reshape(-1, 70)
                f.write(b"\n".join(bytes(r) for r in body) + b"\n")
                if G % 70:
                    f.write(bytes(g[G // 70 * 70:]) + b"\n")
        outs = {}
        for tag, binary in (("ref", os.path.join(ROOT, "oracle", "_ref", "biokanga")), ("ours", os.path.join(ROOT, "biokanga_amd", "bin", "biokanga"))):
            out = os.path.join(tmp, tag + ".sfx")
            t = time.time()
            r = subprocess.run([binary, "index", "-i", fa, "-o", out, "-r", "parity", "-T0"], stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
            print(f"{tag}: rc {r.returncode}, {time.time() - t:.1f} s, {os.path.getsize(out) / 1e6:.1f} MB", flush=True)
            outs[tag] = out
        got = np.fromfile(outs["ours"], dtype=np.uint8); exp = np.fromfile(outs["ref"], dtype=np.uint8)
        assert len(got) == len(exp), (len(got), len(exp))
        assert bytes(got[:1224]) == bytes(exp[:1224]), "header differs"
        blk = struct.unpack_from("<Q", exp, 44)[0]
        n = struct.unpack_from("<Q", exp, blk + 8)[0]
        assert np.array_equal(got[blk:blk + 20 + n], exp[blk:blk + 20 + n]), "block header / bases differ"
        ent = struct.unpack_from("<Q", exp, 20)[0]
        assert np.array_equal(got[ent:], exp[ent:]), "entries differ"
        seq = exp[blk + 20: blk + 20 + n]
        sa_g = np.frombuffer(got, dtype="<u4", count=n, offset=blk + 20 + n)
        sa_e = np.frombuffer(exp, dtype="<u4", count=n, offset=blk + 20 + n)
        diff = np.nonzero(sa_g != sa_e)[0]
        bad = 0
        for j in diff[:200000]:
            a, b = int(sa_g[j]), int(sa_e[j]); l = 0
            while a + l < n and b + l < n and seq[a + l] == seq[b + l]:
                l += 1
            if 7 not in seq[a:a + l + 1]:
                bad += 1
        print(f"{n} suffixes; header, bases (incl. {int((seq == 4).sum())} N left after mutation), entries identical; "
              f"suffix array differs at {len(diff)} positions, {bad} of them NOT tied through an EOS")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

if __name__ == "__main__":
    main()
