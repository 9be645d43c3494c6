#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by RUNNING THE REAL REFERENCE executable
(oracle/_ref/biokanga, built by oracle/build_ref.sh from /root/reference) on small seeded synthetic
inputs.  Run in the build container only (the GPU box has no /root/reference and just consumes the
committed files).  Everything written here is data: inputs (FASTA), the reference-built .sfx and
the reference's outputs (SAM / CSV / log summaries), gzip-compressed.

    python tests/golden/make_golden.py            # regenerates every fixture

Fixtures
  basic/   2 x 100 kbp random sequences (+ a 100-mer shared by both, + an N run of 120 in chrB)
           reads: 0..4 substitutions, both strands, N-containing, lower case, entry-boundary
           spanning, exact duplicates, whitespace in names, lengths 50..150
  repeat/  unique background + four families of an exact 25-mer repeated K = 2000/3000/4000/6000
           times; reads whose only clean core is the repeat (pins the 100-candidate copy-count
           cut-off and MaxIter truncation, SfxArrayV2.cpp:5857-5875)
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "biokanga")
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "a": "t", "c": "g", "g": "c", "t": "a", "n": "n"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def mutate(rng, s, nsubs, positions=None):
    r = list(s)
    if positions is None:
        positions = rng.choice(len(r), nsubs, replace=False)
    for k in positions:
        r[k] = "ACGT"[("ACGT".index(r[k].upper()) + int(rng.integers(1, 4))) % 4]
    return "".join(r)


def write_fasta(path, recs, width=70):
    with open(path, "w") as f:
        for name, seq in recs:
            f.write(">" + name + "\n")
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + "\n")


def write_reads(path, reads):
    with open(path, "w") as f:
        for name, seq in reads:
            f.write(">" + name + "\n" + seq + "\n")


def run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout[-4000:])
        raise SystemExit("reference run failed: " + " ".join(cmd))
    return r.stdout


def gz_copy(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)


def nar_summary(log):
    out = []
    keep = False
    for line in log.splitlines():
        msg = line.split("](biokanga) ", 1)[-1]
        if "Read nonalignment reason summary" in msg:
            keep = True
            continue
        if keep:
            if msg.strip().split(" ")[0].isdigit():
                out.append(msg.strip())
            else:
                keep = False
    return "\n".join(out) + "\n"


def align_runs(tmp, outdir, sfx, reads, runs):
    for tag, flags in runs:
        for fmt, ext in (("-M6", "m6.sam"), ("-M5", "m5.sam"), ("-M0", "m0.csv")):
            if fmt != "-M6" and not tag.endswith("*"):
                continue
            t = tag.rstrip("*")
            out = os.path.join(tmp, f"{t}.{ext}")
            log = run([REF, "align", "-i", reads, "-I", sfx, "-o", out, fmt, "-T4"] + flags, tmp)
            gz_copy(out, os.path.join(outdir, f"{t}.{ext}.gz"))
            if fmt == "-M6":
                with open(os.path.join(outdir, f"{t}.nar.txt"), "w") as f:
                    f.write(nar_summary(log))
            print("  ran", t, fmt, flags)


def make_basic(tmp):
    rng = np.random.default_rng(20261002)
    outdir = os.path.join(HERE, "basic")
    os.makedirs(outdir, exist_ok=True)
    a = rand_seq(rng, 100000)
    b = rand_seq(rng, 100000)
    shared = rand_seq(rng, 100)
    a = a[:40000] + shared + a[40100:]
    b = b[:70000] + shared + b[70100:]
    b = b[:20000] + "N" * 120 + b[20120:]          # N run > 25 -> every 13th N mutated at index time
    a = a[:5000] + a[5000:5400].lower() + a[5400:]  # soft-masked stretch
    # near-duplicate segments (exercise NxtLowMMCnt / -e MMDelta): 1 sub per 100, 2 subs per 100,
    # and an exact 150-mer present three times
    def near(seg, every):
        seg = list(seg)
        for q in every:
            seg[q] = "ACGT"[("ACGT".index(seg[q]) + 1) % 4]
        return "".join(seg)
    b = b[:50000] + near(a[30000:30400], range(50, 400, 100)) + b[50400:]
    b = b[:52000] + near(a[32000:32400], [q for k in range(4) for q in (k * 100 + 30, k * 100 + 70)]) + b[52400:]
    b = b[:54000] + a[34000:34150] + b[54150:56000] + a[34000:34150] + b[56150:]
    genome = [("chrA basic test sequence", a), ("chrB", b)]
    fa = os.path.join(tmp, "basic.fa")
    write_fasta(fa, genome)
    seqs = {"chrA": a.upper(), "chrB": b.upper()}

    reads = []
    idx = 0

    def add(name, seq):
        nonlocal idx
        reads.append((name, seq))
        idx += 1

    for e in range(5):                      # substitution strata, 100 bp
        for _ in range(400):
            c = "chrA" if rng.integers(0, 2) == 0 else "chrB"
            p = int(rng.integers(0, 100000 - 100))
            s = seqs[c][p:p + 100]
            if "N" in s:
                continue
            r = mutate(rng, s, e)
            strand = "+"
            if rng.integers(0, 2):
                r = revcomp(r)
                strand = "-"
            add(f"s{idx}|{c}|{p}|{strand}|{e}", r)
    for L in (50, 60, 75, 90, 125, 150):     # other lengths, 0..3 subs
        for _ in range(60):
            c = "chrA" if rng.integers(0, 2) == 0 else "chrB"
            p = int(rng.integers(0, 100000 - L))
            s = seqs[c][p:p + L]
            if "N" in s:
                continue
            e = int(rng.integers(0, 4))
            r = mutate(rng, s, e)
            strand = "+"
            if rng.integers(0, 2):
                r = revcomp(r)
                strand = "-"
            add(f"l{idx}|{c}|{p}|{strand}|{e}|{L}", r)
    for base, L in ((30000, 400), (32000, 400), (34000, 150)):   # near-duplicate / triplicate segments
        for k in range(30):
            c = "chrA" if k % 3 else "chrB"
            start = base + (20000 if c == "chrB" else 0) + int(rng.integers(0, L - 100 + 1))
            e = k % 4 if k < 24 else 0
            r = mutate(rng, seqs[c][start:start + 100], e)
            if k % 2:
                r = revcomp(r)
            add(f"nd{idx}|{c}|{start}|{e}", r)
    for k in range(40):                      # reads with 1 N (costs a mismatch) and 2 Ns (EN)
        p = int(rng.integers(0, 19000))
        s = list(seqs["chrA"][p:p + 100])
        s[int(rng.integers(0, 100))] = "N"
        if k % 2:
            q = int(rng.integers(0, 100))
            s[q] = "N" if s[q] != "N" else s[q]
            s[(q + 7) % 100] = "N"
        r = "".join(s)
        if k % 3 == 0:
            r = mutate(rng, r.replace("N", "A"), 0)
            r = "".join("R" if i == 10 else ch for i, ch in enumerate(r))   # IUPAC -> N
        add(f"n{idx}|chrA|{p}", r)
    for k in range(10):                      # lower case reads
        p = int(rng.integers(0, 99000))
        add(f"lc{idx}|chrB|{p}", mutate(rng, seqs["chrB"][p:p + 100], k % 3).lower() if "N" not in seqs["chrB"][p:p + 100] else rand_seq(rng, 100))
    for k in range(6):                       # spanning the chrA/chrB boundary (EOS between entries)
        n_a = 30 + 8 * k
        add(f"span{idx}|{n_a}", seqs["chrA"][100000 - n_a:] + seqs["chrB"][:100 - n_a])
    add(f"shared{idx}", shared)               # occurs in both sequences -> ML
    add(f"sharedrc{idx}", revcomp(shared))
    add(f"shared1{idx}", mutate(rng, shared, 1))
    dup = seqs["chrA"][61234:61334]
    for k in range(4):                       # exact duplicates, consecutive and not
        add(f"dup{idx} extra words here", dup)
    add(f"dupx{idx}", seqs["chrB"][1234:1334])
    add(f"dup{idx}\ttabbed", dup)
    for k in range(5):                       # reads near N run of chrB (target Ns were mutated)
        p = 20000 - 60 + 10 * k
        add(f"nrun{idx}|{p}", seqs["chrB"][p:p + 100])
    for k in range(20):                      # random (unalignable) reads
        add(f"rnd{idx}", rand_seq(rng, 100))
    add("lcl|usimreads|00000001|chrA|300|399|100|+|0|0|0 kept whole", seqs["chrA"][300:400])
    # ends of sequences
    add(f"endA{idx}", seqs["chrA"][-100:])
    add(f"startA{idx}", seqs["chrA"][:100])
    add(f"endB{idx}", revcomp(seqs["chrB"][-100:]))
    add(f"startB{idx}", seqs["chrB"][:100])

    rd = os.path.join(tmp, "basic_reads.fa")
    write_reads(rd, reads)
    sfx = os.path.join(tmp, "basic.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "basic", "-T4"], tmp)
    gz_copy(fa, os.path.join(outdir, "genome.fa.gz"))
    gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
    gz_copy(sfx, os.path.join(outdir, "genome.sfx.gz"))
    align_runs(tmp, outdir, sfx, rd, [
        ("s3*", ["-s3"]),
        ("s0", ["-s0"]),
        ("s5", ["-s5"]),
        ("dflt", []),
        ("s3e2", ["-s3", "-e2"]),
        ("s3Q1", ["-s3", "-Q1"]),
        ("s3Q2", ["-s3", "-Q2"]),
        ("s3m1", ["-s3", "-m1"]),
        ("s3m2", ["-s3", "-m2"]),
        ("s3m3", ["-s3", "-m3"]),
        ("s3n0", ["-s3", "-n0"]),
        ("s3n3", ["-s3", "-n3"]),
        ("s2l30", ["-s2", "-l30"]),
    ])


def make_repeat(tmp):
    rng = np.random.default_rng(77)
    outdir = os.path.join(HERE, "repeat")
    os.makedirs(outdir, exist_ok=True)
    fams = [(2000, rand_seq(rng, 25)), (3000, rand_seq(rng, 25)), (4000, rand_seq(rng, 25)), (6000, rand_seq(rng, 25))]
    # chrU: unique background with one copy of each family's 25-mer embedded at known places
    u = list(rand_seq(rng, 60000))
    sites = []
    for fi, (K, R) in enumerate(fams):
        for j in range(12):
            p = 1000 + fi * 14000 + j * 1100
            u[p + 25:p + 50] = list(R)          # read [p, p+100) has core 1 == R
            sites.append((fi, p))
    u = "".join(u)
    # chrR: the repeat copies separated by short random spacers
    parts = []
    for K, R in fams:
        for _ in range(K):
            parts.append(R)
            parts.append(rand_seq(rng, int(rng.integers(4, 12))))
    r = "".join(parts)
    fa = os.path.join(tmp, "repeat.fa")
    write_fasta(fa, [("chrU", u), ("chrR", r)])
    reads = []
    for fi, p in sites:
        s = u[p:p + 100]
        # subs in cores 0, 2 and 3 only (final phase CoreLen 25 at offsets 0,25,50,75)
        for variant in range(3):
            pos = [int(rng.integers(0, 25)), int(rng.integers(50, 75)), int(rng.integers(75, 100))]
            rr = mutate(rng, s, 3, positions=pos)
            strand = "+"
            if variant == 2:
                rr = revcomp(rr)
                strand = "-"
            reads.append((f"rep|K{fams[fi][0]}|{p}|{strand}|v{variant}", rr))
        reads.append((f"rep0|K{fams[fi][0]}|{p}", s))
    # reads made only of repeat material (multi-loci / many instances)
    for fi, (K, R) in enumerate(fams):
        off = r.find(R)
        reads.append((f"inrep|K{K}", r[off:off + 100]))
        reads.append((f"inrep1|K{K}", mutate(rng, r[off + 7:off + 107], 1)))
    rd = os.path.join(tmp, "repeat_reads.fa")
    write_reads(rd, reads)
    sfx = os.path.join(tmp, "repeat.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "repeat", "-T4"], tmp)
    gz_copy(fa, os.path.join(outdir, "genome.fa.gz"))
    gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
    gz_copy(sfx, os.path.join(outdir, "genome.sfx.gz"))
    align_runs(tmp, outdir, sfx, rd, [
        ("s3*", ["-s3"]),
        ("s3m1", ["-s3", "-m1"]),
        ("s3m3", ["-s3", "-m3"]),
        ("s5", ["-s5"]),
    ])


def make_pe(tmp, L=100, name="pe"):
    """paired-end fixture on the basic genome: FR pairs with inserts in and out of the accepted range,
    same-strand pairs, different-chromosome pairs, one mate unalignable, one mate in a duplicated /
    triplicated segment (orphan recovery by the anchored window scan), mates with substitutions."""
    rng = np.random.default_rng(909)
    outdir = os.path.join(HERE, name)
    os.makedirs(outdir, exist_ok=True)
    basic = os.path.join(HERE, "basic")
    fa = os.path.join(tmp, "pe.fa")
    with gzip.open(os.path.join(basic, "genome.fa.gz"), "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    seqs = {}
    name = None
    for line in open(fa):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = []
        else:
            seqs[name].append(line.upper())
    seqs = {k: "".join(v) for k, v in seqs.items()}
    r1, r2 = [], []
    emax1, emax2 = (3, 4) if L == 100 else (7, 10)          # -s5 on 150 bp allows 8 substitutions: 0..6 / 0..9 per mate

    def pair(tag, c1, p1, c2, p2, flip1=False, flip2=True, e1=0, e2=0, rand2=False):
        a = seqs[c1][p1:p1 + L]
        b = seqs[c2][p2:p2 + L]
        if "N" in a or "N" in b or len(a) < L or len(b) < L:
            return
        a = mutate(rng, a, e1)
        b = mutate(rng, b, e2) if not rand2 else rand_seq(rng, L)
        if flip1:
            a = revcomp(a)
        if flip2:
            b = revcomp(b)
        i = len(r1)
        r1.append((f"p{i}_{tag}/1", a))
        r2.append((f"p{i}_{tag}/2", b))

    for k in range(900):                                   # proper FR pairs, inserts 150..650
        c = "chrA" if rng.integers(0, 2) == 0 else "chrB"
        ins = int(np.clip(rng.normal(300, 90), 150, 650))
        p = int(rng.integers(0, 100000 - ins))
        e1, e2 = int(rng.integers(0, emax1)), int(rng.integers(0, emax2))
        if rng.integers(0, 2):
            pair(f"fr{ins}", c, p, c, p + ins - L, False, True, e1, e2)
        else:                                               # the pair read from the other strand
            pair(f"rf{ins}", c, p + ins - L, c, p, True, False, e1, e2)
    for k in range(60):                                    # both mates on the same strand
        c = "chrA"
        p = int(rng.integers(0, 99000))
        pair("same", c, p, c, p + 200, False, False)
    for k in range(60):                                    # mates on different sequences
        pair("xchr", "chrA", int(rng.integers(0, 99000)), "chrB", int(rng.integers(0, 99000)), False, True)
    for k in range(60):                                    # mate 2 unalignable
        pair("r2rnd", "chrB", int(rng.integers(0, 99000)), "chrB", 0, False, True, 0, 0, True)
    # mate 2 inside the exact triplicate (chrA 34000.., chrB 54000.., chrB 56000..) / shared 100-mer, mate 1 unique nearby
    for k in range(40):
        ins = int(rng.integers(220, 390))
        q = 34000 + int(rng.integers(0, 50))
        pair("tripA", "chrA", q - (ins - L), "chrA", q, False, True, int(rng.integers(0, 2)), int(rng.integers(0, 3)))
        q = 54000 + int(rng.integers(0, 50))
        pair("tripB", "chrB", q - (ins - L), "chrB", q, False, True, 0, int(rng.integers(0, 2)))
        q = 56000 + int(rng.integers(0, 50))
        pair("tripC", "chrB", q + (ins - L), "chrB", q, True, False, 0, 0)     # anchor downstream, mate upstream
    for k in range(20):                                    # mate 1 in the near-duplicate (1 sub / 100) segment pair
        q = 30000 + int(rng.integers(0, 300))
        pair("nd", "chrA", q, "chrA", q + 250, False, True, 0, 1)
    for k in range(30):                                    # pairs near sequence ends
        pair("endA", "chrA", 100000 - 330 + k, "chrA", 100000 - L - int(rng.integers(0, 3)), False, True)
        pair("startB", "chrB", int(rng.integers(0, 5)), "chrB", 230 + k, False, True)
    f1 = os.path.join(tmp, f"{name}_1.fa")
    f2 = os.path.join(tmp, f"{name}_2.fa")
    write_reads(f1, r1)
    write_reads(f2, r2)
    sfx = os.path.join(tmp, "pe.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "basic", "-T4"], tmp)
    gz_copy(f1, os.path.join(outdir, "reads_1.fa.gz"))
    gz_copy(f2, os.path.join(outdir, "reads_2.fa.gz"))
    runs = [("U3", ["-U3", "-d200", "-D400", "-s5"]), ("U1", ["-U1", "-d200", "-D400", "-s5"]),
            ("U2", ["-U2", "-d200", "-D400", "-s5"]), ("U4", ["-U4", "-d200", "-D400", "-s5"]),
            ("U3dflt", ["-U3", "-s3"]), ("U3wide", ["-U3", "-d150", "-D1500", "-s5"]), ("U3E", ["-U3", "-d200", "-D400", "-s5", "-E"])]
    if L != 100:         # C3 of SURVEY.md 8(d): 2 x 150 bp, -U3 -d200 -D400 -s5 (+ the other -U modes)
        runs = runs[:4]
    for tag, flags in runs:
        out = os.path.join(tmp, f"{name}_{tag}.sam")
        log = run([REF, "align", "-i", f1, "-u", f2, "-I", sfx, "-o", out, "-M6", "-T4"] + flags, tmp)
        gz_copy(out, os.path.join(outdir, f"{tag}.m6.sam.gz"))
        with open(os.path.join(outdir, f"{tag}.nar.txt"), "w") as f:
            f.write(nar_summary(log))
        print("  ran PE", tag, flags)
    out = os.path.join(tmp, f"{name}_U3.m5.sam")
    run([REF, "align", "-i", f1, "-u", f2, "-I", sfx, "-o", out, "-M5", "-T4", "-U3", "-d200", "-D400", "-s5"], tmp)
    gz_copy(out, os.path.join(outdir, "U3.m5.sam.gz"))


def make_sortorder(tmp):
    """30 000 reads (>= cMinUseLibQsort 25 000, so the reference's own quicksort - not glibc's stable
    merge sort - orders the output) drawn from 2 500 loci, i.e. with many exact ties: pins the tie
    order the sort replica has to reproduce (MTqsort.cpp:313-479)."""
    rng = np.random.default_rng(4242)
    outdir = os.path.join(HERE, "sortorder")
    os.makedirs(outdir, exist_ok=True)
    basic = os.path.join(HERE, "basic")
    fa = os.path.join(tmp, "so.fa")
    with gzip.open(os.path.join(basic, "genome.fa.gz"), "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    seqs = {}
    name = None
    for line in open(fa):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = []
        else:
            seqs[name].append(line.upper())
    seqs = {k: "".join(v) for k, v in seqs.items()}
    loci = [("chrA" if rng.integers(0, 2) == 0 else "chrB", int(rng.integers(0, 99900))) for _ in range(2500)]
    reads = []
    for i in range(30000):
        c, p = loci[int(rng.integers(0, len(loci)))]
        s = seqs[c][p:p + 60]
        if "N" in s:
            s = seqs["chrA"][100:160]
        e = int(rng.integers(0, 3))
        r = mutate(rng, s, e)
        if rng.integers(0, 2):
            r = revcomp(r)
        reads.append((f"q{i}", r))
    rd = os.path.join(tmp, "so_reads.fa")
    write_reads(rd, reads)
    sfx = os.path.join(tmp, "so.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "basic", "-T4"], tmp)
    gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
    for threads in ("-T1", "-T8"):
        out = os.path.join(tmp, f"so{threads}.sam")
        run([REF, "align", "-i", rd, "-I", sfx, "-o", out, "-M5", "-s3", threads], tmp)
    a = open(os.path.join(tmp, "so-T1.sam"), "rb").read()
    b = open(os.path.join(tmp, "so-T8.sam"), "rb").read()
    assert a == b, "reference output order depends on the thread count"
    gz_copy(os.path.join(tmp, "so-T8.sam"), os.path.join(outdir, "s3.m5.sam.gz"))
    print("  sortorder fixture written")


def make_lengths(tmp):
    """reads of 15..49 and 151..2000 bases on the basic genome (-l15 -L2000): pins the short-read core
    geometry and the many-cores-per-strand paths (a 2000-base read at -s3 has MaxTotMM 60 and far more
    than 16 cores per strand) that the 50..150-base fixtures never reach."""
    rng = np.random.default_rng(777)
    outdir = os.path.join(HERE, "lengths")
    os.makedirs(outdir, exist_ok=True)
    basic = os.path.join(HERE, "basic")
    fa = os.path.join(tmp, "len.fa")
    with gzip.open(os.path.join(basic, "genome.fa.gz"), "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    seqs = {}
    name = None
    for line in open(fa):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = []
        else:
            seqs[name].append(line.upper())
    seqs = {k: "".join(v) for k, v in seqs.items()}
    reads = []
    k = 0
    for L in (15, 16, 17, 20, 24, 25, 26, 30, 33, 40, 49, 151, 200, 256, 257, 300, 400, 500, 501, 640, 800, 1000, 1500, 2000):
        for rep in range(20):
            c = "chrA" if rng.integers(0, 2) == 0 else "chrB"
            while True:
                p0 = int(rng.integers(0, len(seqs[c]) - L))
                s = seqs[c][p0:p0 + L]
                if "N" not in s:
                    break
            allowed = max(1, int(0.5 + L * 3 / 100.0))
            e = [0, 1, allowed - 1, allowed, allowed + 1, 2 * allowed + 3][rep % 6] if L >= 25 else rep % 3
            e = max(0, min(e, L // 2))
            r = mutate(rng, s, e)
            if rng.integers(0, 2):
                r = revcomp(r)
            reads.append((f"L{L}_{k}_e{e}", r))
            k += 1
    rd = os.path.join(tmp, "len_reads.fa")
    write_reads(rd, reads)
    sfx = os.path.join(tmp, "len.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "basic", "-T4"], tmp)
    gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
    align_runs(tmp, outdir, sfx, rd, [("s3L", ["-s3", "-l15", "-L2000"]), ("s5L", ["-s5", "-l15", "-L2000"]),
                                      ("dfltL", ["-l15", "-L2000"]), ("s0L", ["-s0", "-l15", "-L2000"])])
    print("  lengths fixture written")


def make_bam(tmp):
    """BAM + BAI as the reference writes them (output name ending in ".bam"): SE -M6 -s3 on the basic fixture and
    PE -U3.  Stored as they are (BGZF is already compressed)."""
    basic = os.path.join(HERE, "basic")
    pe = os.path.join(HERE, "pe")
    sfx = os.path.join(tmp, "bam.sfx")
    with gzip.open(os.path.join(basic, "genome.sfx.gz"), "rb") as f, open(sfx, "wb") as g:
        shutil.copyfileobj(f, g)
    def gunz(src, dst):
        with gzip.open(src, "rb") as f, open(dst, "wb") as g:
            shutil.copyfileobj(f, g)
    rd = os.path.join(tmp, "bam_reads.fa")
    gunz(os.path.join(basic, "reads.fa.gz"), rd)
    out = os.path.join(tmp, "out_s3m6.bam")
    run([REF, "align", "-i", rd, "-I", sfx, "-o", out, "-M6", "-s3", "-T4"], tmp)
    shutil.copy(out, os.path.join(basic, "s3.m6.bam"))
    shutil.copy(out + ".bai", os.path.join(basic, "s3.m6.bam.bai"))
    out = os.path.join(tmp, "out_s3m5.bam")
    run([REF, "align", "-i", rd, "-I", sfx, "-o", out, "-M5", "-s3", "-T4"], tmp)
    shutil.copy(out, os.path.join(basic, "s3.m5.bam"))
    shutil.copy(out + ".bai", os.path.join(basic, "s3.m5.bam.bai"))
    r1, r2 = os.path.join(tmp, "bam_r1.fa"), os.path.join(tmp, "bam_r2.fa")
    gunz(os.path.join(pe, "reads_1.fa.gz"), r1)
    gunz(os.path.join(pe, "reads_2.fa.gz"), r2)
    out = os.path.join(tmp, "out_U3m6.bam")
    run([REF, "align", "-i", r1, "-u", r2, "-I", sfx, "-o", out, "-M6", "-s5", "-U3", "-d200", "-D400", "-T4"], tmp)
    shutil.copy(out, os.path.join(pe, "U3.m6.bam"))
    shutil.copy(out + ".bai", os.path.join(pe, "U3.m6.bam.bai"))
    print("  bam fixtures written")


def make_fastq(tmp):
    """the basic reads as FASTQ (IUPAC codes replaced by N - the reference refuses them in FASTQ sequences;
    quality lines that start with '@' or '>', '+' lines with and without the repeated id, a blank line now and
    then): the reference's SAM for it, plus the proof that a FASTQ read with an IUPAC code ends the run."""
    import random
    import re
    rnd = random.Random(3)
    basic = os.path.join(HERE, "basic")
    sfx = os.path.join(tmp, "fq.sfx")
    with gzip.open(os.path.join(basic, "genome.sfx.gz"), "rb") as f, open(sfx, "wb") as g:
        shutil.copyfileobj(f, g)
    recs = []
    name = None
    for line in gzip.open(os.path.join(basic, "reads.fa.gz"), "rt"):
        line = line.rstrip("\n")
        if line.startswith(">"):
            name = line[1:]
        else:
            recs.append((name, re.sub(r"[^ACGTNacgtn]", "N", line)))
    qchars = "!\"#$%&'()*+,-./0123456789:;<=>?@ABCDEFGHIJ"
    fq = os.path.join(tmp, "reads.fq")
    with open(fq, "w") as f:
        for i, (n, s) in enumerate(recs):
            q = "".join(rnd.choice(qchars) for _ in s)
            if i % 7 == 0:
                q = "@" + q[1:]
            if i % 11 == 0:
                q = ">" + q[1:]
            f.write(f"@{n}\n{s}\n+{n if i % 3 == 0 else ''}\n{q}\n")
            if i % 13 == 0:
                f.write("\n")
    out = os.path.join(tmp, "fq.sam")
    run([REF, "align", "-i", fq, "-I", sfx, "-o", out, "-M6", "-s3", "-T4"], tmp)
    gz_copy(fq, os.path.join(basic, "reads.fq.gz"))
    gz_copy(out, os.path.join(basic, "s3fq.m6.sam.gz"))
    bad = os.path.join(tmp, "bad.fq")
    with open(bad, "w") as f:
        f.write("@ok\n" + "ACGT" * 15 + "\n+\n" + "I" * 60 + "\n@iupac\n" + "ACGR" * 15 + "\n+\n" + "I" * 60 + "\n")
    r = subprocess.run([REF, "align", "-i", bad, "-I", sfx, "-o", os.path.join(tmp, "bad.sam"), "-M6", "-s3", "-T4"], cwd=tmp,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0, "the reference was expected to refuse an IUPAC code inside a FASTQ sequence"
    print("  fastq fixture written")


def make_stats(tmp):
    """the -O statistics file: SE (-M5; the reference refuses -O with -M6) and PE (insert length table first)"""
    basic = os.path.join(HERE, "basic")
    pe = os.path.join(HERE, "pe")
    def gunz(src, dst):
        with gzip.open(src, "rb") as f, open(dst, "wb") as g:
            shutil.copyfileobj(f, g)
    sfx, rd, r1, r2 = (os.path.join(tmp, x) for x in ("st.sfx", "st_reads.fa", "st_r1.fa", "st_r2.fa"))
    gunz(os.path.join(basic, "genome.sfx.gz"), sfx)
    gunz(os.path.join(basic, "reads.fa.gz"), rd)
    gunz(os.path.join(pe, "reads_1.fa.gz"), r1)
    gunz(os.path.join(pe, "reads_2.fa.gz"), r2)
    st = os.path.join(tmp, "se_stats.csv")
    nj, mj = os.path.join(tmp, "se_none.fa"), os.path.join(tmp, "se_multi.fa")
    run([REF, "align", "-i", rd, "-I", sfx, "-o", os.path.join(tmp, "st.sam"), "-s3", "-M5", "-T4", "-O", st, "-j", nj, "-J", mj], tmp)
    gz_copy(st, os.path.join(basic, "s3.m5.stats.csv.gz"))
    gz_copy(nj, os.path.join(basic, "s3.none.fa.gz"))
    gz_copy(mj, os.path.join(basic, "s3.multi.fa.gz"))
    st = os.path.join(tmp, "pe_stats.csv")
    nj, mj = os.path.join(tmp, "pe_none.fa"), os.path.join(tmp, "pe_multi.fa")
    run([REF, "align", "-i", r1, "-u", r2, "-I", sfx, "-o", os.path.join(tmp, "stpe.sam"), "-s5", "-U3", "-d200", "-D400", "-M5", "-T4", "-O", st,
         "-j", nj, "-J", mj], tmp)
    gz_copy(st, os.path.join(pe, "U3.m5.stats.csv.gz"))
    gz_copy(nj, os.path.join(pe, "U3.none.fa.gz"))
    gz_copy(mj, os.path.join(pe, "U3.multi.fa.gz"))
    print("  stats fixtures written")


def make_formats(tmp):
    """-M1..3 (CSV with match / read / both sequences) and -M4 (UCSC BED, default and -t title) on the basic fixture"""
    basic = os.path.join(HERE, "basic")
    sfx, rd = os.path.join(tmp, "fm.sfx"), os.path.join(tmp, "fm_reads.fa")
    for src, dst in ((os.path.join(basic, "genome.sfx.gz"), sfx), (os.path.join(basic, "reads.fa.gz"), rd)):
        with gzip.open(src, "rb") as f, open(dst, "wb") as g:
            shutil.copyfileobj(f, g)
    for m, name, extra in ((1, "s3.m1.csv", []), (2, "s3.m2.csv", []), (3, "s3.m3.csv", []), (4, "s3.m4.bed", []), (4, "s3.m4t.bed", ["-t", "my track"])):
        out = os.path.join(tmp, name)
        run([REF, "align", "-i", rd, "-I", sfx, "-o", out, "-s3", f"-M{m}", "-T4"] + extra, tmp)
        gz_copy(out, os.path.join(basic, name + ".gz"))
    out = os.path.join(tmp, "s3y5Y3.sam")
    run([REF, "align", "-i", rd, "-I", sfx, "-o", out, "-s3", "-M6", "-y5", "-Y3", "-l60", "-T4"], tmp)      # end trims + length acceptance
    gz_copy(out, os.path.join(basic, "s3y5Y3l60.m6.sam.gz"))
    print("  format fixtures written")


MULTI_RUNS = [("r1R5", ["-r1", "-R5", "-T4"]), ("r2R5", ["-r2", "-R5", "-T1"]), ("r3R5", ["-r3", "-R5", "-T4"]),
              ("r4R5", ["-r4", "-R5", "-T4"]), ("r4R3X", ["-r4", "-R3", "-X", "-T4"]), ("r3R8T1", ["-r3", "-R8", "-T1"]),
              ("r5R5", ["-r5", "-R5", "-T1"]), ("r5R3X", ["-r5", "-R3", "-X", "-T1"])]


def make_multi(tmp):
    """multi-loci modes (-r1..-r5 with -R / -X): a genome with 300-base segments present in 2..9 places (some
    copies one substitution away), reads from them and from their unique surroundings so that the clustering
    modes have unique neighbours (-r3) and other multi-loci reads (-r4) to cluster with.  -r2 and -r5 runs use -T1:
    with more threads the reference's rand() sequence / record order depends on thread timing."""
    rng = np.random.default_rng(555)
    outdir = os.path.join(HERE, "multi")
    os.makedirs(outdir, exist_ok=True)
    g = [list(rand_seq(rng, 90000)), list(rand_seq(rng, 60000))]
    dups = []
    for _ in range(60):
        c = int(rng.integers(0, 2))
        src = int(rng.integers(0, len(g[c]) - 300))
        seg = g[c][src:src + 300]
        places = [(c, src)]
        for _k in range(int(rng.integers(1, 9))):
            c2 = int(rng.integers(0, 2))
            dst = int(rng.integers(0, len(g[c2]) - 300))
            cp = list(seg)
            if rng.integers(0, 3) == 0:
                q = int(rng.integers(0, 300))
                cp[q] = "ACGT"[("ACGT".index(cp[q]) + 1) % 4]
            g[c2][dst:dst + 300] = cp
            places.append((c2, dst))
        dups.append(places)
    seqs = ["".join(x) for x in g]
    fa = os.path.join(tmp, "multi.fa")
    write_fasta(fa, [("mA", seqs[0]), ("mB", seqs[1])])
    reads = []

    def add(c, p, tag):
        s = seqs[c][p:p + 100]
        s = mutate(rng, s, int(rng.integers(0, 3)))
        if rng.integers(0, 2):
            s = revcomp(s)
        reads.append((f"{tag}{len(reads)}", s))

    for places in dups:
        for (c, p) in places[:3]:
            for _r in range(int(rng.integers(2, 7))):            # inside the repeated segment: multi-loci
                add(c, p + int(rng.integers(0, 200)), "m")
            for _r in range(int(rng.integers(0, 6))):            # straddling its edge / next to it: unique neighbours
                q = p + int(rng.integers(-90, 290))
                add(c, max(0, min(q, len(seqs[c]) - 100)), "u")
    for _ in range(3000):
        c = int(rng.integers(0, 2))
        add(c, int(rng.integers(0, len(seqs[c]) - 100)), "b")
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    rd = os.path.join(tmp, "multi_reads.fa")
    write_reads(rd, reads)
    sfx = os.path.join(tmp, "multi.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "multi", "-T4"], tmp)
    gz_copy(fa, os.path.join(outdir, "genome.fa.gz"))
    gz_copy(sfx, os.path.join(outdir, "genome.sfx.gz"))
    gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
    for tag, flags in MULTI_RUNS:
        for fmt, ext in (("-M6", "m6.sam"), ("-M5", "m5.sam"), ("-M0", "m0.csv"), ("-M4", "m4.bed")):
            if fmt in ("-M0", "-M4") and not tag.startswith("r5R5"):
                continue
            if fmt == "-M5" and tag not in ("r5R5", "r3R5"):
                continue
            out = os.path.join(tmp, f"{tag}.{ext}")
            run([REF, "align", "-i", rd, "-I", sfx, "-o", out, fmt, "-s3"] + flags, tmp)
            gz_copy(out, os.path.join(outdir, f"{tag}.{ext}.gz"))
            print("  ran", tag, fmt, flags)
    # thread-count independence of the clustering modes (their block hand-out depends on -T only through
    # the shortcut for identical neighbours, Aligner.cpp:4978-4986)
    for mode in ("-r3", "-r4"):
        outs = []
        for th in ("-T1", "-T4", "-T8"):
            out = os.path.join(tmp, f"chk{mode}{th}.sam")
            run([REF, "align", "-i", rd, "-I", sfx, "-o", out, "-M6", "-s3", mode, "-R5", th], tmp)
            outs.append(open(out, "rb").read())
        print("  ", mode, "identical across -T1/-T4/-T8:", outs[0] == outs[1] == outs[2])
    print("  multi fixture written")


MULTI_BEST_RUNS = [("r5R5N", ["-r5", "-R5", "-N", "-T1"], ["-M6", "-M0"]), ("r5R2Ns1", ["-r5", "-R2", "-N", "-T1", "-s1"], ["-M0"]),
                   ("r3R4N", ["-r3", "-R4", "-N", "-T4"], ["-M6"]), ("r2R3N", ["-r2", "-R3", "-N", "-T1"], ["-M6"]),
                   ("r1R5N", ["-r1", "-R5", "-N", "-T4"], ["-M6"])]


def make_multi_best(tmp):
    """-N (CSfxArrayV3::LocateBestMatches: the best -R loci by mismatches instead of the AlignReads schedule) on the multi fixture"""
    outdir = os.path.join(HERE, "multi")
    sfx, rd = os.path.join(tmp, "mb.sfx"), os.path.join(tmp, "mb_reads.fa")
    for src, dst in ((os.path.join(outdir, "genome.sfx.gz"), sfx), (os.path.join(outdir, "reads.fa.gz"), rd)):
        with gzip.open(src, "rb") as f, open(dst, "wb") as g:
            shutil.copyfileobj(f, g)
    ext = {"-M6": "m6.sam", "-M0": "m0.csv"}
    for tag, flags, fmts in MULTI_BEST_RUNS:
        for fmt in fmts:
            out = os.path.join(tmp, f"{tag}.{ext[fmt]}")
            run([REF, "align", "-i", rd, "-I", sfx, "-o", out, fmt] + (["-s3"] if "-s1" not in flags else []) + flags, tmp)
            gz_copy(out, os.path.join(outdir, f"{tag}.{ext[fmt]}.gz"))
            print("  ran", tag, fmt, flags)


def make_indel(tmp):
    """microInDels (-a): reads carrying an insertion or a deletion of 1..8 bases (plus 0-1 substitutions), one to three reads
    per site so that both supported and orphan placements occur, among ordinary reads; -a10 and -a3 in SAM / CSV / BED."""
    rng = np.random.default_rng(99)
    outdir = os.path.join(HERE, "indel")
    os.makedirs(outdir, exist_ok=True)
    g = [rand_seq(rng, 40000), rand_seq(rng, 30000)]
    fa = os.path.join(tmp, "indel.fa")
    write_fasta(fa, [("iA", g[0]), ("iB", g[1])])
    reads = []
    for i in range(150):
        c = int(rng.integers(0, 2)); p = int(rng.integers(200, len(g[c]) - 400)); L = int(rng.integers(1, 9)); kind = int(rng.integers(0, 2))
        for r in range(int(rng.integers(1, 4))):
            st = p - int(rng.integers(12, 90))
            if kind == 0:
                s = g[c][st:p] + rand_seq(rng, L) + g[c][p:st + 100 - L]
            else:
                s = g[c][st:p] + g[c][p + L:st + 100 + L]
            s = mutate(rng, s[:100], int(rng.integers(0, 2)))
            if rng.integers(0, 2):
                s = revcomp(s)
            reads.append((f"{'ins' if kind == 0 else 'del'}{i}_{r}_L{L}", s))
    for i in range(400):
        c = int(rng.integers(0, 2)); p = int(rng.integers(0, len(g[c]) - 100))
        s = mutate(rng, g[c][p:p + 100], int(rng.integers(0, 5)))
        if rng.integers(0, 2):
            s = revcomp(s)
        reads.append((f"n{i}", s))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    rd = os.path.join(tmp, "indel_reads.fa")
    write_reads(rd, reads)
    sfx = os.path.join(tmp, "indel.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "indel", "-T4"], tmp)
    gz_copy(fa, os.path.join(outdir, "genome.fa.gz"))
    gz_copy(sfx, os.path.join(outdir, "genome.sfx.gz"))
    gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
    for tag, flags in (("a10", ["-a10", "-s3"]), ("a3s5", ["-a3", "-s5"]), ("a20Q1", ["-a20", "-s3", "-Q1"])):
        for fmt, ext in (("-M6", "m6.sam"), ("-M5", "m5.sam"), ("-M0", "m0.csv"), ("-M3", "m3.csv"), ("-M4", "m4.bed")):
            if tag != "a10" and fmt not in ("-M6", "-M0"):
                continue
            out = os.path.join(tmp, f"{tag}.{ext}")
            run([REF, "align", "-i", rd, "-I", sfx, "-o", out, fmt, "-T4"] + flags, tmp)
            gz_copy(out, os.path.join(outdir, f"{tag}.{ext}.gz"))
            if fmt == "-M4":
                gz_copy(out + ".ind", os.path.join(outdir, f"{tag}.{ext}.ind.gz"))
            print("  ran", tag, fmt, flags)
    bam = os.path.join(tmp, "a10.m6.bam")
    run([REF, "align", "-i", rd, "-I", sfx, "-o", bam, "-M6", "-T4", "-a10", "-s3"], tmp)
    shutil.copyfile(bam, os.path.join(outdir, "a10.m6.bam"))
    shutil.copyfile(bam + ".bai", os.path.join(outdir, "a10.m6.bam.bai"))


def make_trim(tmp):
    """-x (AutoTrimFlanks): the basic reads with -x5 -s3 and -x6 -s10 in every output kind, the pe fixture with -U3 -x4,
    the indel fixture with -a10 -x4 (segmented reads are left alone by the trimmer)"""
    basic, pe, indel = os.path.join(HERE, "basic"), os.path.join(HERE, "pe"), os.path.join(HERE, "indel")
    def unz(src, dst):
        with gzip.open(src, "rb") as f, open(dst, "wb") as g:
            shutil.copyfileobj(f, g)
    sfx, rd = os.path.join(tmp, "tr.sfx"), os.path.join(tmp, "tr_reads.fa")
    unz(os.path.join(basic, "genome.sfx.gz"), sfx)
    unz(os.path.join(basic, "reads.fa.gz"), rd)
    for tag, flags, fmts in (("s3x5", ["-s3", "-x5"], ["-M6", "-M5", "-M0", "-M3", "-M4"]), ("s10x6", ["-s10", "-x6"], ["-M6", "-M0"])):
        for fmt in fmts:
            ext = {"-M6": "m6.sam", "-M5": "m5.sam", "-M0": "m0.csv", "-M3": "m3.csv", "-M4": "m4.bed"}[fmt]
            out = os.path.join(tmp, f"{tag}.{ext}")
            extra = ["-O", os.path.join(tmp, "st.csv")] if (tag == "s3x5" and fmt == "-M5") else []
            run([REF, "align", "-i", rd, "-I", sfx, "-o", out, fmt, "-T4"] + flags + extra, tmp)
            gz_copy(out, os.path.join(basic, f"{tag}.{ext}.gz"))
            if extra:
                gz_copy(extra[1], os.path.join(basic, f"{tag}.m5.stats.csv.gz"))
    bam = os.path.join(tmp, "s3x5.m6.bam")
    run([REF, "align", "-i", rd, "-I", sfx, "-o", bam, "-M6", "-T4", "-s3", "-x5"], tmp)
    shutil.copyfile(bam, os.path.join(basic, "s3x5.m6.bam"))
    shutil.copyfile(bam + ".bai", os.path.join(basic, "s3x5.m6.bam.bai"))
    r1, r2 = os.path.join(tmp, "tr_r1.fa"), os.path.join(tmp, "tr_r2.fa")
    unz(os.path.join(pe, "reads_1.fa.gz"), r1)
    unz(os.path.join(pe, "reads_2.fa.gz"), r2)
    out = os.path.join(tmp, "U3x4.sam")
    run([REF, "align", "-i", r1, "-u", r2, "-I", sfx, "-o", out, "-M6", "-T4", "-U3", "-d200", "-D400", "-s5", "-x4"], tmp)
    gz_copy(out, os.path.join(pe, "U3x4.m6.sam.gz"))
    isfx, ird = os.path.join(tmp, "tri.sfx"), os.path.join(tmp, "tri_reads.fa")
    unz(os.path.join(indel, "genome.sfx.gz"), isfx)
    unz(os.path.join(indel, "reads.fa.gz"), ird)
    for fmt, ext in (("-M6", "m6.sam"), ("-M0", "m0.csv")):
        out = os.path.join(tmp, f"a10x4.{ext}")
        run([REF, "align", "-i", ird, "-I", isfx, "-o", out, fmt, "-T4", "-a10", "-s3", "-x4"], tmp)
        gz_copy(out, os.path.join(indel, f"a10x4.{ext}.gz"))
    print("  trim fixtures written")


def make_splice(tmp):
    """splice junctions (-A, which also switches flank trimming on: MinFlankExacts = -s): reads spanning an intron of 30..4000 bases
    (most with GT..AG or CT..AC ends, some without), one to three reads per junction so that supported and orphan junctions occur,
    plus ordinary reads; -A5000 and -A500 in SAM / CSV / BED (+ .jct)"""
    rng = np.random.default_rng(4711)
    outdir = os.path.join(HERE, "splice")
    os.makedirs(outdir, exist_ok=True)
    g = [list(rand_seq(rng, 60000)), list(rand_seq(rng, 40000))]
    juncs = []
    for i in range(140):
        c = int(rng.integers(0, 2)); a = int(rng.integers(300, len(g[c]) - 5000)); gap = int(rng.choice([30, 60, 120, 400, 900, 2500, 4000]))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            g[c][a:a + 2] = "GT"; g[c][a + gap - 2:a + gap] = "AG"
        elif kind == 1:
            g[c][a:a + 2] = "CT"; g[c][a + gap - 2:a + gap] = "AC"
        juncs.append((c, a, gap))
    seqs = ["".join(x) for x in g]
    fa = os.path.join(tmp, "splice.fa")
    write_fasta(fa, [("sA", seqs[0]), ("sB", seqs[1])])
    reads = []
    for i, (c, a, gap) in enumerate(juncs):
        for r in range(int(rng.integers(1, 4))):
            left = int(rng.integers(15, 86))
            s = seqs[c][a - left:a] + seqs[c][a + gap:a + gap + 100 - left]
            s = mutate(rng, s, int(rng.integers(0, 2)))
            if rng.integers(0, 2):
                s = revcomp(s)
            reads.append((f"j{i}_{r}_g{gap}", s))
    for i in range(400):
        c = int(rng.integers(0, 2)); p = int(rng.integers(0, len(seqs[c]) - 100))
        s = mutate(rng, seqs[c][p:p + 100], int(rng.integers(0, 5)))
        if rng.integers(0, 2):
            s = revcomp(s)
        reads.append((f"n{i}", s))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    rd = os.path.join(tmp, "splice_reads.fa")
    write_reads(rd, reads)
    sfx = os.path.join(tmp, "splice.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "splice", "-T4"], tmp)
    gz_copy(fa, os.path.join(outdir, "genome.fa.gz"))
    gz_copy(sfx, os.path.join(outdir, "genome.sfx.gz"))
    gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
    for tag, flags in (("A5000", ["-A5000", "-s3"]), ("A500s5", ["-A500", "-s5"]), ("A5000a5", ["-A5000", "-a5", "-s3"])):
        for fmt, ext in (("-M6", "m6.sam"), ("-M5", "m5.sam"), ("-M0", "m0.csv"), ("-M4", "m4.bed")):
            if tag != "A5000" and fmt not in ("-M6", "-M0"):
                continue
            out = os.path.join(tmp, f"{tag}.{ext}")
            run([REF, "align", "-i", rd, "-I", sfx, "-o", out, fmt, "-T4"] + flags, tmp)
            gz_copy(out, os.path.join(outdir, f"{tag}.{ext}.gz"))
            if fmt in ("-M4", "-M6", "-M5"):
                gz_copy(out + ".jct", os.path.join(outdir, f"{tag}.{ext}.jct.gz"))
            print("  ran", tag, fmt, flags)


def make_chimeric(tmp):
    """chimeric trimming (-c): reads whose 5' and / or 3' end (10..45 bases) comes from elsewhere (random bases or another place of
    the genome), with 0-2 substitutions in the genuine part, among ordinary reads; -c50, -c70 in SAM / BAM / CSV / BED"""
    rng = np.random.default_rng(2718)
    outdir = os.path.join(HERE, "chimeric")
    os.makedirs(outdir, exist_ok=True)
    g = [rand_seq(rng, 50000), rand_seq(rng, 30000)]
    # a duplicated region so that some chimeric placements are ambiguous
    g[1] = g[1][:5000] + g[0][7000:7400] + g[1][5400:]
    fa = os.path.join(tmp, "chim.fa")
    write_fasta(fa, [("cA", g[0]), ("cB", g[1])])
    reads = []
    for i in range(500):
        c = int(rng.integers(0, 2)); p = int(rng.integers(0, len(g[c]) - 100))
        if i < 40:
            c, p = 0, 7000 + int(rng.integers(0, 300))
        core = mutate(rng, g[c][p:p + 100], int(rng.integers(0, 3)))
        k5 = int(rng.integers(10, 46)) if rng.integers(0, 3) else 0
        k3 = int(rng.integers(10, 46)) if (rng.integers(0, 3) == 0 or k5 == 0) else 0
        if k5 + k3 > 55:
            k3 = 0
        def foreign(k):
            if rng.integers(0, 2):
                return rand_seq(rng, k)
            c2 = int(rng.integers(0, 2)); q = int(rng.integers(0, len(g[c2]) - k))
            return g[c2][q:q + k]
        s = foreign(k5) + core[k5:100 - k3] + foreign(k3)
        if rng.integers(0, 2):
            s = revcomp(s)
        reads.append((f"c{i}_{k5}_{k3}", s))
    for i in range(300):
        c = int(rng.integers(0, 2)); p = int(rng.integers(0, len(g[c]) - 100))
        s = mutate(rng, g[c][p:p + 100], int(rng.integers(0, 5)))
        if rng.integers(0, 2):
            s = revcomp(s)
        reads.append((f"n{i}", s))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    rd = os.path.join(tmp, "chim_reads.fa")
    write_reads(rd, reads)
    sfx = os.path.join(tmp, "chim.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "chim", "-T4"], tmp)
    gz_copy(fa, os.path.join(outdir, "genome.fa.gz"))
    gz_copy(sfx, os.path.join(outdir, "genome.sfx.gz"))
    gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
    for tag, flags in (("c50", ["-c50", "-s3"]), ("c70s5", ["-c70", "-s5"]), ("c60e2", ["-c60", "-s3", "-e2"])):
        for fmt, ext in (("-M6", "m6.sam"), ("-M5", "m5.sam"), ("-M0", "m0.csv"), ("-M3", "m3.csv"), ("-M4", "m4.bed")):
            if tag != "c50" and fmt not in ("-M6", "-M0"):
                continue
            out = os.path.join(tmp, f"{tag}.{ext}")
            run([REF, "align", "-i", rd, "-I", sfx, "-o", out, fmt, "-T4"] + flags, tmp)
            gz_copy(out, os.path.join(outdir, f"{tag}.{ext}.gz"))
            print("  ran", tag, fmt, flags)
    bam = os.path.join(tmp, "c50.m6.bam")
    run([REF, "align", "-i", rd, "-I", sfx, "-o", bam, "-M6", "-T4", "-c50", "-s3"], tmp)
    shutil.copyfile(bam, os.path.join(outdir, "c50.m6.bam"))
    shutil.copyfile(bam + ".bai", os.path.join(outdir, "c50.m6.bam.bai"))


CHIMML_RUNS = [("r1R5c50", ["-r1", "-R5", "-c50", "-s3", "-T4"], ["-M6"]), ("r2R5c50", ["-r2", "-R5", "-c50", "-s3", "-T1"], ["-M6"]),
               ("r3R5c50", ["-r3", "-R5", "-c50", "-s3", "-T4"], ["-M6"]), ("r4R5c60", ["-r4", "-R5", "-c60", "-s3", "-T4"], ["-M6"]),
               ("r5R5c50", ["-r5", "-R5", "-c50", "-s3", "-T1"], ["-M6", "-M0", "-M4"]), ("r5R3Xc50", ["-r5", "-R3", "-X", "-c50", "-s3", "-T1"], ["-M6"]),
               ("r4R3Xc70s5", ["-r4", "-R3", "-X", "-c70", "-s5", "-T4"], ["-M6"]), ("r5R8c55e2", ["-r5", "-R8", "-c55", "-s3", "-e2", "-T1"], ["-M0"])]


def make_chimml(tmp):
    """chimeric trimming together with the multi-loci modes (-c with -r1..-r5: the chimeric LocateCoreMultiples call is made with
    MaxHits = -R and every locus it returns keeps its own end trims, SfxArrayV2.cpp:5959-6080, Aligner.cpp:9222-9304): a genome
    with 260-base segments present in 2..6 places (some copies a substitution away), reads from them and from their unique
    surroundings, a 5' and / or 3' end (10..45 bases) foreign in half of the reads"""
    rng = np.random.default_rng(16180)
    outdir = os.path.join(HERE, "chimml")
    os.makedirs(outdir, exist_ok=True)
    g = [list(rand_seq(rng, 60000)), list(rand_seq(rng, 40000))]
    segs = []
    for k in range(40):
        c = int(rng.integers(0, 2)); p = int(rng.integers(0, len(g[c]) - 260))
        seg = g[c][p:p + 260]
        places = [(c, p)]
        for _ in range(int(rng.integers(1, 6))):
            c2 = int(rng.integers(0, 2)); q = int(rng.integers(0, len(g[c2]) - 260))
            cp = list(seg)
            if rng.integers(0, 3) == 0:
                j = int(rng.integers(0, 260)); cp[j] = "ACGT"[("ACGT".index(cp[j]) + 1) % 4]
            g[c2][q:q + 260] = cp
            places.append((c2, q))
        segs.append(places)
    g = ["".join(x) for x in g]
    fa = os.path.join(tmp, "chimml.fa")
    write_fasta(fa, [("mA", g[0]), ("mB", g[1])])

    def foreign(k):
        if rng.integers(0, 2):
            return rand_seq(rng, k)
        c2 = int(rng.integers(0, 2)); q = int(rng.integers(0, len(g[c2]) - k))
        return g[c2][q:q + k]

    reads = []
    for i in range(900):
        if i < 600:                                       # from a multi-copy segment (or straddling its edge)
            c, p = segs[int(rng.integers(0, len(segs)))][0]
            p = max(0, min(len(g[c]) - 100, p + int(rng.integers(-40, 200))))
        else:
            c = int(rng.integers(0, 2)); p = int(rng.integers(0, len(g[c]) - 100))
        core = mutate(rng, g[c][p:p + 100], int(rng.integers(0, 3)))
        k5 = k3 = 0
        if rng.integers(0, 2):
            k5 = int(rng.integers(10, 46)) if rng.integers(0, 3) else 0
            k3 = int(rng.integers(10, 46)) if (rng.integers(0, 3) == 0 or k5 == 0) else 0
            if k5 + k3 > 50:
                k3 = 0
        sq = foreign(k5) + core[k5:100 - k3] + foreign(k3)
        if rng.integers(0, 2):
            sq = revcomp(sq)
        reads.append((f"m{i}_{k5}_{k3}", sq))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    rd = os.path.join(tmp, "chimml_reads.fa")
    write_reads(rd, reads)
    sfx = os.path.join(tmp, "chimml.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "chimml", "-T4"], tmp)
    gz_copy(fa, os.path.join(outdir, "genome.fa.gz"))
    gz_copy(sfx, os.path.join(outdir, "genome.sfx.gz"))
    gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
    for tag, flags, fmts in CHIMML_RUNS:
        for fmt in fmts:
            ext = {"-M6": "m6.sam", "-M0": "m0.csv", "-M4": "m4.bed"}[fmt]
            out = os.path.join(tmp, f"{tag}.{ext}")
            log = run([REF, "align", "-i", rd, "-I", sfx, "-o", out, fmt] + flags, tmp)
            gz_copy(out, os.path.join(outdir, f"{tag}.{ext}.gz"))
            if fmt == fmts[0]:
                with open(os.path.join(outdir, f"{tag}.nar.txt"), "w") as f:
                    f.write(nar_summary(log))
            print("  ran", tag, fmt, flags)


CHIMMLINDEL_RUNS = [("r1R5c50a8", ["-r1", "-R5", "-c50", "-a8", "-s3", "-T4"], ["-M6"]), ("r2R5c50a8", ["-r2", "-R5", "-c50", "-a8", "-s3", "-T1"], ["-M6"]),
                    ("r3R5c50a8", ["-r3", "-R5", "-c50", "-a8", "-s3", "-T4"], ["-M6", "-M0"]), ("r4R5c60a5", ["-r4", "-R5", "-c60", "-a5", "-s3", "-T4"], ["-M6"]),
                    ("r3R3Xc55a10A200", ["-r3", "-R3", "-X", "-c55", "-a10", "-A200", "-s3", "-T1"], ["-M6"])]


def make_chimmlindel(tmp):
    """chimeric trimming together with a multi-loci mode AND microInDels / splice junctions (-c with -r1..-r4 and -a / -A: AlignReads runs
    LocateInDels / LocateSpliceJuncts with one hit between the substitution-only phases and the chimeric LocateCoreMultiples call, on
    the same in / out counts and hit buffer, SfxArrayV2.cpp:7722-7757): chimml's kind of genome (260-base segments in 2..6 places),
    reads from them and from unique places with a small insertion or deletion in a third of them - so that a microInDel search on a
    multi-copy segment is ambiguous -, a foreign end in half of them, both in some"""
    rng = np.random.default_rng(27182)
    outdir = os.path.join(HERE, "chimmlindel")
    os.makedirs(outdir, exist_ok=True)
    g = [list(rand_seq(rng, 60000)), list(rand_seq(rng, 40000))]
    segs = []
    for k in range(40):
        c = int(rng.integers(0, 2)); p = int(rng.integers(0, len(g[c]) - 260))
        seg = g[c][p:p + 260]
        places = [(c, p)]
        for _ in range(int(rng.integers(1, 6))):
            c2 = int(rng.integers(0, 2)); q = int(rng.integers(0, len(g[c2]) - 260))
            cp = list(seg)
            if rng.integers(0, 3) == 0:
                j = int(rng.integers(0, 260)); cp[j] = "ACGT"[("ACGT".index(cp[j]) + 1) % 4]
            g[c2][q:q + 260] = cp
            places.append((c2, q))
        segs.append(places)
    g = ["".join(x) for x in g]
    fa = os.path.join(tmp, "chimmlindel.fa")
    write_fasta(fa, [("mA", g[0]), ("mB", g[1])])

    def foreign(k):
        if rng.integers(0, 2):
            return rand_seq(rng, k)
        c2 = int(rng.integers(0, 2)); q = int(rng.integers(0, len(g[c2]) - k))
        return g[c2][q:q + k]

    reads = []
    for i in range(1200):
        if i < 800:                                       # from a multi-copy segment (or straddling its edge)
            c, p = segs[int(rng.integers(0, len(segs)))][0]
            p = max(0, min(len(g[c]) - 120, p + int(rng.integers(-40, 200))))
        else:
            c = int(rng.integers(0, 2)); p = int(rng.integers(0, len(g[c]) - 120))
        src = g[c][p:p + 120]
        kind = int(rng.integers(0, 3))
        d = 0
        if kind == 0:                                     # a deletion or an insertion of 1..9 bases somewhere in the middle
            at = int(rng.integers(25, 75)); d = int(rng.integers(1, 10))
            if rng.integers(0, 2):
                src = src[:at] + src[at + d:]
            else:
                src = src[:at] + rand_seq(rng, d) + src[at:]; d = -d
        core = mutate(rng, src[:100], int(rng.integers(0, 3)))
        k5 = k3 = 0
        if rng.integers(0, 2):
            k5 = int(rng.integers(10, 46)) if rng.integers(0, 3) else 0
            k3 = int(rng.integers(10, 46)) if (rng.integers(0, 3) == 0 or k5 == 0) else 0
            if k5 + k3 > 50:
                k3 = 0
        sq = foreign(k5) + core[k5:100 - k3] + foreign(k3)
        if rng.integers(0, 2):
            sq = revcomp(sq)
        reads.append((f"x{i}_{k5}_{k3}_{d}", sq))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    rd = os.path.join(tmp, "chimmlindel_reads.fa")
    write_reads(rd, reads)
    sfx = os.path.join(tmp, "chimmlindel.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "chimmlindel", "-T4"], tmp)
    gz_copy(fa, os.path.join(outdir, "genome.fa.gz"))
    gz_copy(sfx, os.path.join(outdir, "genome.sfx.gz"))
    gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
    for tag, flags, fmts in CHIMMLINDEL_RUNS:
        for fmt in fmts:
            ext = {"-M6": "m6.sam", "-M0": "m0.csv", "-M4": "m4.bed"}[fmt]
            out = os.path.join(tmp, f"{tag}.{ext}")
            log = run([REF, "align", "-i", rd, "-I", sfx, "-o", out, fmt] + flags, tmp)
            gz_copy(out, os.path.join(outdir, f"{tag}.{ext}.gz"))
            if fmt == fmts[0]:
                with open(os.path.join(outdir, f"{tag}.nar.txt"), "w") as f:
                    f.write(nar_summary(log))
            print("  ran", tag, fmt, flags)


def make_pechim(tmp):
    """chimeric trimming together with paired ends (-c with -U: the pair rules work on the trimmed loci, AdjStartLoci / AdjEndLoci,
    Aligner.cpp:2750-2769, and the orphan recovery may return an end-trimmed partner, AlignPairedRead with MinChimericLen,
    SfxArrayV2.cpp:8327): FR pairs on the chimeric fixture's genome, a mate's 5' and / or 3' end (10..45 bases) foreign in a third of
    the mates, mates inside the duplicated region (orphan recovery), inserts in and out of the accepted range"""
    rng = np.random.default_rng(31415)
    outdir = os.path.join(HERE, "pechim")
    os.makedirs(outdir, exist_ok=True)
    fa = os.path.join(tmp, "pechim.fa")
    with gzip.open(os.path.join(HERE, "chimeric", "genome.fa.gz"), "rb") as f, open(fa, "wb") as g_:
        shutil.copyfileobj(f, g_)
    seqs, name = {}, None
    for line in open(fa):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = []
        else:
            seqs[name].append(line.upper())
    g = {k: "".join(v) for k, v in seqs.items()}
    names = list(g)
    L = 100
    r1, r2 = [], []

    def foreign(k):
        if rng.integers(0, 2):
            return rand_seq(rng, k)
        c2 = names[int(rng.integers(0, 2))]; q = int(rng.integers(0, len(g[c2]) - k))
        return g[c2][q:q + k]

    def chim(s):
        """a read whose ends may come from elsewhere"""
        if rng.integers(0, 3):
            return s, 0, 0
        k5 = int(rng.integers(10, 46)) if rng.integers(0, 3) else 0
        k3 = int(rng.integers(10, 46)) if (rng.integers(0, 3) == 0 or k5 == 0) else 0
        if k5 + k3 > 45:
            k3 = 0
        return foreign(k5) + s[k5:L - k3] + foreign(k3), k5, k3

    for i in range(700):
        c = names[int(rng.integers(0, 2))]
        ins = int(np.clip(rng.normal(300, 70), 160, 560))
        p = int(rng.integers(0, len(g[c]) - ins))
        if i < 60:                                            # one mate inside the duplicated 400 bases (cA 7000.., cB 5000..)
            c, p = "cA", 7000 + int(rng.integers(0, 250)) - (ins - L if rng.integers(0, 2) else 0)
            p = max(0, p)
        a = mutate(rng, g[c][p:p + L], int(rng.integers(0, 3)))
        b = mutate(rng, g[c][p + ins - L:p + ins], int(rng.integers(0, 3)))
        if len(a) < L or len(b) < L:
            continue
        a, a5, a3 = chim(a)
        b, b5, b3 = chim(b)
        b = revcomp(b)
        if rng.integers(0, 2):                                # the pair read from the other strand
            a, b = b, a
        if i % 23 == 0:
            b = rand_seq(rng, L)                              # mate 2 unalignable
        r1.append((f"q{i}_{ins}_{a5}_{a3}/1", a))
        r2.append((f"q{i}_{ins}_{b5}_{b3}/2", b))
    f1, f2 = os.path.join(tmp, "pechim_1.fa"), os.path.join(tmp, "pechim_2.fa")
    write_reads(f1, r1)
    write_reads(f2, r2)
    sfx = os.path.join(tmp, "pechim.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "chim", "-T4"], tmp)
    gz_copy(f1, os.path.join(outdir, "reads_1.fa.gz"))
    gz_copy(f2, os.path.join(outdir, "reads_2.fa.gz"))
    for tag, flags in (("U3c50", ["-U3", "-c50", "-s3", "-d200", "-D400"]), ("U1c60", ["-U1", "-c60", "-s3", "-d200", "-D400"]),
                       ("U4c50", ["-U4", "-c50", "-s3", "-d200", "-D400"]), ("U2c70s5", ["-U2", "-c70", "-s5", "-d200", "-D400"]),
                       ("U3c50wide", ["-U3", "-c50", "-s3", "-d150", "-D1500"]), ("U3", ["-U3", "-s3", "-d200", "-D400"])):
        out = os.path.join(tmp, f"pechim_{tag}.sam")
        log = run([REF, "align", "-i", f1, "-u", f2, "-I", sfx, "-o", out, "-M6", "-T4"] + flags, tmp)
        gz_copy(out, os.path.join(outdir, f"{tag}.m6.sam.gz"))
        with open(os.path.join(outdir, f"{tag}.nar.txt"), "w") as f:
            f.write(nar_summary(log))
        print("  ran PE + chimeric", tag, flags)
    out = os.path.join(tmp, "pechim_U3c50.m0.csv")
    run([REF, "align", "-i", f1, "-u", f2, "-I", sfx, "-o", out, "-M0", "-T4", "-U3", "-c50", "-s3", "-d200", "-D400"], tmp)
    gz_copy(out, os.path.join(outdir, "U3c50.m0.csv.gz"))


def make_combined(tmp):
    """-a / -A / -c together (AlignReads tries them in that order and hands each one's leftover state to the next, :7722-7757):
    the indel, splice and chimeric reads in ONE run against a genome holding all three fixtures' sequences"""
    outdir = os.path.join(HERE, "combined")
    os.makedirs(outdir, exist_ok=True)
    recs, reads = [], []
    for fx in ("indel", "splice", "chimeric"):
        name, seq = None, []
        for line in gzip.open(os.path.join(HERE, fx, "genome.fa.gz"), "rt"):
            line = line.strip()
            if line.startswith(">"):
                if name:
                    recs.append((name, "".join(seq)))
                name, seq = line[1:].split()[0], []
            else:
                seq.append(line)
        recs.append((name, "".join(seq)))
        nm = None
        for line in gzip.open(os.path.join(HERE, fx, "reads.fa.gz"), "rt"):
            line = line.strip()
            if line.startswith(">"):
                nm = line[1:]
            else:
                reads.append((fx[0] + "_" + nm, line))
    # reads built to be ambiguous for the microInDel search: the same deletion read placed in two copies of a segment
    rng = np.random.default_rng(31337)
    seg = rand_seq(rng, 600)
    extra = rand_seq(rng, 3000) + seg + rand_seq(rng, 2000) + seg + rand_seq(rng, 3000)
    recs.append(("xD", extra))
    for i in range(60):
        p = 3000 + int(rng.integers(60, 400)); L = int(rng.integers(1, 7))
        s = extra[p - 50:p] + extra[p + L:p + L + 50]
        s = mutate(rng, s, int(rng.integers(0, 2)))
        k = int(rng.integers(0, 30))
        if k >= 12:
            s = rand_seq(rng, k) + s[k:]
        if rng.integers(0, 2):
            s = revcomp(s)
        reads.append((f"x_amb{i}_L{L}_{k}", s))
    fa, rd = os.path.join(tmp, "comb.fa"), os.path.join(tmp, "comb_reads.fa")
    write_fasta(fa, recs)
    order = rng.permutation(len(reads))
    write_reads(rd, [reads[i] for i in order])
    sfx = os.path.join(tmp, "comb.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "comb", "-T4"], tmp)
    gz_copy(fa, os.path.join(outdir, "genome.fa.gz"))
    gz_copy(sfx, os.path.join(outdir, "genome.sfx.gz"))
    gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
    for tag, flags in (("a10c50", ["-a10", "-c50", "-s3"]), ("a10A5000c50", ["-a10", "-A5000", "-c50", "-s3"]), ("A5000c60", ["-A5000", "-c60", "-s3"]),
                       ("a10A5000", ["-a10", "-A5000", "-s3"])):
        for fmt, ext in (("-M6", "m6.sam"), ("-M0", "m0.csv")):
            out = os.path.join(tmp, f"{tag}.{ext}")
            run([REF, "align", "-i", rd, "-I", sfx, "-o", out, fmt, "-T4"] + flags, tmp)
            gz_copy(out, os.path.join(outdir, f"{tag}.{ext}.gz"))
            print("  ran", tag, fmt, flags)


def make_quality(tmp):
    """FASTQ quality modes (-g0 Sanger, -g1 Illumina 1.3+, -g2 Solexa) on the basic FASTQ reads: QUAL columns of SAM, BAM and the
    Phred bands of the -O statistics"""
    basic = os.path.join(HERE, "basic")
    sfx, fq = os.path.join(tmp, "q.sfx"), os.path.join(tmp, "q_reads.fq")
    for src, dst in ((os.path.join(basic, "genome.sfx.gz"), sfx), (os.path.join(basic, "reads.fq.gz"), fq)):
        with gzip.open(src, "rb") as f, open(dst, "wb") as g:
            shutil.copyfileobj(f, g)
    for g_ in (0, 1, 2):
        out = os.path.join(tmp, f"g{g_}.sam")
        run([REF, "align", "-i", fq, "-I", sfx, "-o", out, "-M6", "-s3", f"-g{g_}", "-T4"], tmp)
        gz_copy(out, os.path.join(basic, f"s3fqg{g_}.m6.sam.gz"))
    out = os.path.join(tmp, "g0.m5.sam")
    st = os.path.join(tmp, "g0.stats.csv")
    run([REF, "align", "-i", fq, "-I", sfx, "-o", out, "-M5", "-s3", "-g0", "-y3", "-Y5", "-T4", "-O", st], tmp)
    gz_copy(out, os.path.join(basic, "s3fqg0y3Y5.m5.sam.gz"))
    gz_copy(st, os.path.join(basic, "s3fqg0y3Y5.m5.stats.csv.gz"))
    bam = os.path.join(tmp, "g0.m6.bam")
    run([REF, "align", "-i", fq, "-I", sfx, "-o", bam, "-M6", "-s3", "-g0", "-T4"], tmp)
    shutil.copyfile(bam, os.path.join(basic, "s3fqg0.m6.bam"))
    shutil.copyfile(bam + ".bai", os.path.join(basic, "s3fqg0.m6.bam.bai"))
    print("  quality fixtures written")


def make_pcr(tmp):
    """-k (ReducePCRduplicates) on the sortorder reads (30 000 reads stacked on 2 500 loci): window 0, 20 and 200, and with -c"""
    basic, so = os.path.join(HERE, "basic"), os.path.join(HERE, "sortorder")
    sfx, rd = os.path.join(tmp, "k.sfx"), os.path.join(tmp, "k_reads.fa")
    for src, dst in ((os.path.join(basic, "genome.sfx.gz"), sfx), (os.path.join(so, "reads.fa.gz"), rd)):
        with gzip.open(src, "rb") as f, open(dst, "wb") as g:
            shutil.copyfileobj(f, g)
    for tag, flags in (("k0", ["-k0"]), ("k20", ["-k20"]), ("k200", ["-k200"]), ("k50x4", ["-k50", "-x4"])):
        out = os.path.join(tmp, f"{tag}.sam")
        run([REF, "align", "-i", rd, "-I", sfx, "-o", out, "-M6", "-s3", "-T4"] + flags, tmp)
        gz_copy(out, os.path.join(so, f"s3{tag}.m6.sam.gz"))
    # -# (sample every Nth raw read / pair): SE CSV (shows the read numbering) and PE SAM
    rd2 = os.path.join(tmp, "k_basic.fa")
    with gzip.open(os.path.join(basic, "reads.fa.gz"), "rb") as f, open(rd2, "wb") as g:
        shutil.copyfileobj(f, g)
    out = os.path.join(tmp, "n3.csv")
    run([REF, "align", "-i", rd2, "-I", sfx, "-o", out, "-M0", "-s3", "-T4", "-#3"], tmp)
    gz_copy(out, os.path.join(basic, "s3n3.m0.csv.gz"))
    pe = os.path.join(HERE, "pe")
    r1, r2 = os.path.join(tmp, "k_r1.fa"), os.path.join(tmp, "k_r2.fa")
    for src, dst in ((os.path.join(pe, "reads_1.fa.gz"), r1), (os.path.join(pe, "reads_2.fa.gz"), r2)):
        with gzip.open(src, "rb") as f, open(dst, "wb") as g:
            shutil.copyfileobj(f, g)
    out = os.path.join(tmp, "U3n4.sam")
    run([REF, "align", "-i", r1, "-u", r2, "-I", sfx, "-o", out, "-M6", "-T4", "-U3", "-d200", "-D400", "-s5", "-#4"], tmp)
    gz_copy(out, os.path.join(pe, "U3n4.m6.sam.gz"))
    # -Z / -z chromosome filters
    for tag, flags in (("ZchrB", ["-Z", "chrB"]), ("zchra", ["-z", "^chra$"]), ("zAZB", ["-z", "chr[AB]", "-Z", "chrA"])):
        out = os.path.join(tmp, f"{tag}.sam")
        run([REF, "align", "-i", rd2, "-I", sfx, "-o", out, "-M6", "-s3", "-T4"] + flags, tmp)
        gz_copy(out, os.path.join(basic, f"s3{tag}.m6.sam.gz"))
    print("  pcr fixtures written")


def make_snp(tmp):
    """SNP calling (-p / -P / -1 / -S): reads drawn from a donor that differs from the indexed genome at a few hundred loci
    (homozygous, and heterozygous with about half of the reads carrying the other allele; some adjacent so that DiSNPs and
    TriSNPs occur), 13x coverage with 0.4% substitution errors, a few N bases; CSV, VCF and BED output, also with -x / -c."""
    rng = np.random.default_rng(4242)
    outdir = os.path.join(HERE, "snp")
    os.makedirs(outdir, exist_ok=True)
    g = [rand_seq(rng, 50000), rand_seq(rng, 25000), rand_seq(rng, 6000)]
    g[1] = g[1][:9000] + "N" * 40 + g[1][9040:]
    fa = os.path.join(tmp, "snp.fa")
    write_fasta(fa, [("sA", g[0]), ("sB", g[1]), ("sC", g[2])])
    donors = []
    for c, seq in enumerate(g):
        hom, het = list(seq), list(seq)
        pos = 150
        while pos < len(seq) - 150:
            if seq[pos] != "N":
                alt = "ACGT"[("ACGT".index(seq[pos]) + int(rng.integers(1, 4))) % 4]
                hom[pos] = alt
                if rng.integers(0, 3) != 0:
                    het[pos] = alt
            pos += int(rng.choice([1, 2, 7, 40, 180, 420, 700]))
        donors.append(("".join(hom), "".join(het)))
    reads = []
    for c, seq in enumerate(g):
        n = len(seq) * 13 // 100
        for i in range(n):
            p0 = int(rng.integers(0, len(seq) - 100))
            src = donors[c][int(rng.integers(0, 2))]
            s = src[p0:p0 + 100]
            if "N" in s:
                continue
            s = mutate(rng, s, int(rng.binomial(100, 0.004)))
            if rng.integers(0, 40) == 0:
                k = int(rng.integers(0, 100)); s = s[:k] + "N" + s[k + 1:]
            if rng.integers(0, 2):
                s = revcomp(s)
            reads.append((f"r{c}_{i}", s))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    rd = os.path.join(tmp, "snp_reads.fa")
    write_reads(rd, reads)
    sfx = os.path.join(tmp, "snp.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "snpsp", "-T4"], tmp)
    gz_copy(fa, os.path.join(outdir, "genome.fa.gz"))
    gz_copy(sfx, os.path.join(outdir, "genome.sfx.gz"))
    gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
    runs = (("p5", ["-M5", "-p5"], "sam", None), ("p3P10n10", ["-M5", "-p3", "-P0.1", "-1", "10"], "sam", "vcf"),
            ("p5bed", ["-M4", "-p5"], "bed", None), ("p8x5", ["-M0", "-p8", "-x5"], "csv", None), ("p5c60", ["-M5", "-p5", "-c60"], "sam", None),
            ("p1P40n1", ["-M0", "-p1", "-P0.4", "-1", "0.1"], "csv", None))
    for tag, flags, ext, snpext in runs:
        out = os.path.join(tmp, f"{tag}.{ext}")
        cmd = [REF, "align", "-i", rd, "-I", sfx, "-o", out, "-T4", "-s5"] + flags
        snp = out + ".snp"
        if snpext:
            snp = os.path.join(tmp, f"{tag}.{snpext}")
            cmd += ["-S", snp]
        log = run(cmd, tmp)
        gz_copy(out, os.path.join(outdir, f"{tag}.{ext}.gz"))
        gz_copy(snp, os.path.join(outdir, f"{tag}.snp.gz" if not snpext else f"{tag}.{snpext}.gz"))
        for extra in (".disnp.csv", ".trisnp.csv"):
            if os.path.exists(snp + extra):
                gz_copy(snp + extra, os.path.join(outdir, f"{tag}{extra}.gz"))
        with open(os.path.join(outdir, f"{tag}.log.txt"), "w") as f:
            f.write("".join(l.split(") ", 1)[-1] for l in log.splitlines(True) if "putative SNPs" in l or "aligned loci bases" in l))
        print("  ran", tag, flags)
    # marker sequences around the SNPs (-K / -G)
    for tag, flags in (("k51", ["-M5", "-p5", "-K51"]), ("k25G10", ["-M0", "-p3", "-K25", "-G0.1", "-P0.2", "-1", "10"]), ("k120G45", ["-M5", "-p2", "-K120", "-G0.45"])):
        out = os.path.join(tmp, f"{tag}.sam")
        log = run([REF, "align", "-i", rd, "-I", sfx, "-o", out, "-T4", "-s5"] + flags, tmp)
        for extra in (".snp", ".snp.markers", ".snp.disnp.csv", ".snp.trisnp.csv"):
            gz_copy(out + extra, os.path.join(outdir, f"{tag}{extra}.gz"))
        with open(os.path.join(outdir, f"{tag}.log.txt"), "w") as f:
            f.write("".join(l.split(") ", 1)[-1] for l in log.splitlines(True) if "putative SNPs" in l or "aligned loci bases" in l or "marker sequences writtten" in l))
        print("  ran", tag, flags)
    # SNP centroids (-7): per 7-mer context, how many well covered loci and how many SNPs
    for tag, flags in (("cent5", ["-M5", "-p5"]), ("cent2P40", ["-M0", "-p2", "-P0.4", "-1", "1"])):
        out, cent = os.path.join(tmp, f"{tag}.sam"), os.path.join(tmp, f"{tag}.centroids.csv")
        run([REF, "align", "-i", rd, "-I", sfx, "-o", out, "-T4", "-s5", "-7", cent] + flags, tmp)
        gz_copy(out + ".snp", os.path.join(outdir, f"{tag}.snp.gz"))
        gz_copy(cent, os.path.join(outdir, f"{tag}.centroids.csv.gz"))
        print("  ran", tag, flags)
    # SNPs over reads other options placed: paired ends, microInDel / spliced reads (left out of the pile-up), multi-loci reads assigned by -r3,
    # chimeric trims
    def unz(fix, name, dst):
        with gzip.open(os.path.join(HERE, fix, name), "rb") as f, open(dst, "wb") as g:
            shutil.copyfileobj(f, g)
        return dst
    bsfx = unz("basic", "genome.sfx.gz", os.path.join(tmp, "x_basic.sfx"))
    r1, r2 = unz("pe", "reads_1.fa.gz", os.path.join(tmp, "x_r1.fa")), unz("pe", "reads_2.fa.gz", os.path.join(tmp, "x_r2.fa"))
    combos = [("x_pe", ["-i", r1, "-u", r2, "-I", bsfx, "-U3", "-d200", "-D400", "-s5", "-M5", "-p1", "-P0.4", "-1", "5"], "sam")]
    for fix, tag, flags in (("indel", "x_indel", ["-a10", "-s3", "-M0", "-p1", "-P0.4", "-1", "2"]), ("multi", "x_multi", ["-r3", "-R5", "-s3", "-M0", "-p2", "-P0.4", "-1", "5"]),
                            ("combined", "x_comb", ["-a8", "-A3000", "-c55", "-s3", "-M5", "-p1", "-P0.4", "-1", "2"]), ("splice", "x_splice", ["-A5000", "-s3", "-M4", "-p1", "-P0.4", "-1", "2"])):
        fs, fr = unz(fix, "genome.sfx.gz", os.path.join(tmp, f"{tag}.sfx")), unz(fix, "reads.fa.gz", os.path.join(tmp, f"{tag}.fa"))
        combos.append((tag, ["-i", fr, "-I", fs] + flags, {"-M0": "csv", "-M5": "sam", "-M4": "bed"}[[f for f in flags if f.startswith("-M")][0]]))
    for tag, flags, ext in combos:
        out = os.path.join(tmp, f"{tag}.{ext}")
        run([REF, "align", "-o", out, "-T4"] + flags, tmp)
        gz_copy(out, os.path.join(outdir, f"{tag}.{ext}.gz"))
        for extra in (".snp", ".snp.disnp.csv", ".snp.trisnp.csv"):
            gz_copy(out + extra, os.path.join(outdir, f"{tag}{extra}.gz"))
        print("  ran", tag)


def make_multi_rescue(tmp, only=None):
    """-r1..-r4 together with -a / -A (the reference takes them; AlignReads with MaxHits > 1 runs its microInDel / splice branches): on
    the indel, splice and combined fixtures"""
    outdir = os.path.join(HERE, "multi")
    def unz(fix, name, dst):
        with gzip.open(os.path.join(HERE, fix, name), "rb") as f, open(dst, "wb") as g:
            shutil.copyfileobj(f, g)
        return dst
    for fix, tag, flags, fmt, ext in (("indel", "xi_r1R5a10", ["-r1", "-R5", "-a10", "-s3"], "-M6", "sam"), ("indel", "xi_r3R5a10", ["-r3", "-R5", "-a10", "-s3"], "-M6", "sam"),
                                     ("indel", "xi_r2R5a10", ["-r2", "-R5", "-a10", "-s3"], "-M0", "csv"), ("splice", "xs_r4R5A5000", ["-r4", "-R5", "-A5000", "-s3"], "-M6", "sam"),
                                     ("splice", "xs_r3R3XA5000", ["-r3", "-R3", "-X", "-A5000", "-s3"], "-M5", "sam"), ("combined", "xc_r3R3a8A3000", ["-r3", "-R3", "-a8", "-A3000", "-s3"], "-M0", "csv"),
                                     ("combined", "xc_r4R8a8A3000", ["-r4", "-R8", "-a8", "-A3000", "-s3"], "-M4", "bed"),
                                     # -N (LocateBestMatches) has no microInDel / splice branches: the options are accepted and only -A's flank trimming acts
                                     ("indel", "xn_r2R5Na10", ["-r2", "-R5", "-N", "-a10", "-s3"], "-M6", "sam"), ("splice", "xn_r3R4NA5000", ["-r3", "-R4", "-N", "-A5000", "-s3"], "-M6", "sam"),
                                     ("combined", "xn_r1R5Na8A3000", ["-r1", "-R5", "-N", "-a8", "-A3000", "-s3"], "-M0", "csv")):
        if only and not tag.startswith(only):
            continue
        fs, fr = unz(fix, "genome.sfx.gz", os.path.join(tmp, f"{tag}.sfx")), unz(fix, "reads.fa.gz", os.path.join(tmp, f"{tag}.fa"))
        out = os.path.join(tmp, f"{tag}.{ext}")
        run([REF, "align", "-i", fr, "-I", fs, "-o", out, fmt, "-T4"] + flags, tmp)
        gz_copy(out, os.path.join(outdir, f"{tag}.{ext}.gz"))
        print("  ran", tag)


def make_lifted(tmp):
    """Option combinations the reference takes and the MI355X command line refused until round 4 (kanga.cpp:648-660,712,719-725,980-995):
    -k / -x / -Z / -z with -r5 (every locus a record of its own: the filters see the records).  -T1: the record numbering follows thread
    timing otherwise.  -Z / -z with -U: the filters act inside the pair rules."""
    def unz(fix, name, dst):
        with gzip.open(os.path.join(HERE, fix, name), "rb") as f, open(dst, "wb") as g:
            shutil.copyfileobj(f, g)
        return dst
    ms, mr = unz("multi", "genome.sfx.gz", os.path.join(tmp, "lm.sfx")), unz("multi", "reads.fa.gz", os.path.join(tmp, "lm.fa"))
    for tag, flags, fmt, ext in (("r5R5k0", ["-r5", "-R5", "-s3", "-k0"], "-M6", "m6.sam"), ("r5R5k0", ["-r5", "-R5", "-s3", "-k0"], "-M0", "m0.csv"),
                                 ("r5R5x4", ["-r5", "-R5", "-s3", "-x4"], "-M6", "m6.sam"), ("r5R5x4", ["-r5", "-R5", "-s3", "-x4"], "-M0", "m0.csv"),
                                 ("r5R5ZmB", ["-r5", "-R5", "-s3", "-Z", "mB"], "-M6", "m6.sam"), ("r5R3XzmA", ["-r5", "-R3", "-X", "-s3", "-z", "^ma$"], "-M5", "m5.sam"),
                                 ("r5R5k20x3Z", ["-r5", "-R5", "-s3", "-k20", "-x3", "-Z", "mB"], "-M4", "m4.bed")):
        out = os.path.join(tmp, f"{tag}.{ext}")
        run([REF, "align", "-i", mr, "-I", ms, "-o", out, fmt, "-T1"] + flags, tmp)
        gz_copy(out, os.path.join(HERE, "multi", f"{tag}.{ext}.gz"))
        print("  ran multi", tag, fmt)
    # -Z / -z with -U: the reference consults the filters inside its pair rules (AcceptThisChromID, Aligner.cpp:2771-2786,3224,3323,3445)
    ps = unz("basic", "genome.sfx.gz", os.path.join(tmp, "lp.sfx"))
    p1, p2 = unz("pe", "reads_1.fa.gz", os.path.join(tmp, "lp_1.fa")), unz("pe", "reads_2.fa.gz", os.path.join(tmp, "lp_2.fa"))
    for tag, flags in (("U3ZchrB", ["-U3", "-d200", "-D400", "-s5", "-Z", "chrB"]), ("U2zchrA", ["-U2", "-d200", "-D400", "-s5", "-z", "chra"]),
                       ("U4ZchrA", ["-U4", "-d200", "-D400", "-s5", "-Z", "chrA$"]), ("U1ZchrB", ["-U1", "-d200", "-D400", "-s5", "-Z", "chrB"])):
        out = os.path.join(tmp, f"{tag}.m6.sam")
        log = run([REF, "align", "-i", p1, "-u", p2, "-I", ps, "-o", out, "-M6", "-T4"] + flags, tmp)
        gz_copy(out, os.path.join(HERE, "pe", f"{tag}.m6.sam.gz"))
        with open(os.path.join(HERE, "pe", f"{tag}.nar.txt"), "w") as f:
            f.write(nar_summary(log))
        print("  ran pe", tag)


def make_simreads(tmp):
    """Reads named the way `biokanga simreads` names them (lcl|usimreads|id|chrom|start|end|len|strand|...): the reference's only
    built-in correctness signal is the truth-check line of CAligner::ReportAlignStats (Aligner.cpp:3581-3728) - "There are N (a 2
    edge, b 1 edge) high confidence aligned simulated reads with m misaligned".  Claims: correct, end off (1 edge), both off,
    wrong sequence, the three-part sequence-name form.  `mixed`: one plainly named read among them (the check stops there and the
    line is not printed)."""
    rng = np.random.default_rng(4242)
    outdir = os.path.join(HERE, "simreads")
    os.makedirs(outdir, exist_ok=True)
    basic = os.path.join(HERE, "basic")
    fa = os.path.join(tmp, "sim.fa")
    with gzip.open(os.path.join(basic, "genome.fa.gz"), "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    seqs, name = {}, None
    for line in open(fa):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = []
        else:
            seqs[name].append(line.upper())
    seqs = {k: "".join(v) for k, v in seqs.items()}
    L = 100
    reads = []
    for k in range(600):
        c = "chrA" if rng.integers(0, 2) == 0 else "chrB"
        p = int(rng.integers(0, 100000 - L))
        s = seqs[c][p:p + L]
        if "N" in s:
            continue
        e = int(rng.integers(0, 5))                        # 4 substitutions: not aligned at -s3
        s = mutate(rng, s, e)
        strand = "+" if rng.integers(0, 2) == 0 else "-"
        if strand == "-":
            s = revcomp(s)
        kind = k % 10
        cc, st, en = c, p, p + L - 1
        if kind == 6:
            en += 5                                         # one edge agrees
        elif kind == 7:
            st += 3; en += 3                                # neither edge
        elif kind == 8:
            cc = "chrB" if c == "chrA" else "chrA"          # wrong sequence
        elif kind == 9:
            cc = "gnl|UG|" + c                              # three-part name form (never equals the target's name here)
        reads.append((f"lcl|usimreads|{len(reads) + 1:08d}|{cc}|{st}|{en}|{L}|{strand}|{e}|0|0", s))
    sfx = os.path.join(tmp, "sim.sfx")
    run([REF, "index", "-i", fa, "-o", sfx, "-r", "basic", "-T4"], tmp)
    for tag in ("sim", "mixed"):
        rd = list(reads)
        if tag == "mixed":                                  # an accepted read (0 substitutions) in the middle loses its simreads name
            j = next(i for i in range(300, len(rd)) if rd[i][0].split("|")[8] == "0")
            rd[j] = ("plainname extra words", rd[j][1])
        f = os.path.join(tmp, f"{tag}.fa")
        write_reads(f, rd)
        gz_copy(f, os.path.join(outdir, f"{tag}.reads.fa.gz"))
        out = os.path.join(tmp, f"{tag}.sam")
        log = run([REF, "align", "-i", f, "-I", sfx, "-o", out, "-M6", "-s3", "-T4"], tmp)
        gz_copy(out, os.path.join(outdir, f"{tag}.s3.m6.sam.gz"))
        keep = [line.split("](biokanga) ", 1)[-1].rstrip() for line in log.splitlines()
                if "accepted alignments" in line or "high confidence aligned simulated reads" in line]
        with open(os.path.join(outdir, f"{tag}.truthcheck.txt"), "w") as g:
            g.write("\n".join(keep) + "\n")
        print("  simreads", tag, keep)


def make_csi(tmp):
    """One sequence of 537 Mbp (> the 512 Mbp a BAI addresses): the reference writes a BGZF-compressed CSI index instead
    (SAMfile.cpp:1602-1607).  The genome is regenerated from its seed by the test (tests/helpers.py write_big_genome); committed
    are the reads and the reference's .bam + .bam.csi.  The reference's index build of 538 M suffixes takes a few minutes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    outdir = os.path.join(HERE, "csi")
    os.makedirs(outdir, exist_ok=True)
    big = "/dev/shm/bk_csi_golden"
    os.makedirs(big, exist_ok=True)
    try:
        fa = os.path.join(big, "big.fa")
        seqs = helpers.write_big_genome(fa)
        rng = np.random.default_rng(99)
        reads = []
        comp = np.zeros(256, dtype=np.uint8)
        for a, b in zip(b"ACGT", b"TGCA"):
            comp[a] = b
        for k in range(3000):
            name = "big" if k % 10 else "small"
            n = len(seqs[name])
            if name == "big" and k % 3:
                p = int(rng.integers(536_870_912 - 50, n - 100))          # around and beyond the BAI limit
            else:
                p = int(rng.integers(0, n - 100))
            s = seqs[name][p:p + 100].copy()
            for q in rng.choice(100, size=int(rng.integers(0, 3)), replace=False):
                s[q] = b"ACGT"[(b"ACGT".index(s[q]) + 1 + int(rng.integers(0, 3))) % 4]
            if rng.integers(0, 2):
                s = comp[s[::-1]]
            reads.append((f"r{k}_{name}_{p}", s.tobytes().decode()))
        rd = os.path.join(big, "reads.fa")
        write_reads(rd, reads)
        gz_copy(rd, os.path.join(outdir, "reads.fa.gz"))
        sfx = os.path.join(big, "big.sfx")
        run([REF, "index", "-i", fa, "-o", sfx, "-r", "bigcsi", "-T8"], big)
        out = os.path.join(big, "out_align.bam")
        log = run([REF, "align", "-i", rd, "-I", sfx, "-o", out, "-M6", "-s3", "-T8"], big)
        shutil.copyfile(out, os.path.join(outdir, "s3.m6.bam"))
        shutil.copyfile(out + ".csi", os.path.join(outdir, "s3.m6.bam.csi"))
        with open(os.path.join(outdir, "s3.nar.txt"), "w") as f:
            f.write(nar_summary(log))
        print("  csi fixture written;", [l for l in log.splitlines() if "CSI" in l])
    finally:
        shutil.rmtree(big, ignore_errors=True)


def main():
    if not os.path.exists(REF):
        raise SystemExit("build the reference first: oracle/build_ref.sh")
    with tempfile.TemporaryDirectory() as tmp:
        if "--only-multi-rescue" in sys.argv:
            make_multi_rescue(tmp)
            return
        if "--only-multi-rescue-n" in sys.argv:
            make_multi_rescue(tmp, only="xn_")
            return
        if "--only-lifted" in sys.argv:
            make_lifted(tmp)
            return
        if "--only-snp" in sys.argv:
            make_snp(tmp)
            return
        if "--only-pe" in sys.argv:
            make_pe(tmp)
            return
        if "--only-csi" in sys.argv:
            make_csi(tmp)
            return
        if "--only-simreads" in sys.argv:
            make_simreads(tmp)
            return
        if "--only-pe150" in sys.argv:
            make_pe(tmp, L=150, name="pe150")
            return
        if "--only-pcr" in sys.argv:
            make_pcr(tmp)
            return
        if "--only-quality" in sys.argv:
            make_quality(tmp)
            return
        if "--only-combined" in sys.argv:
            make_combined(tmp)
            return
        if "--only-chimeric" in sys.argv:
            make_chimeric(tmp)
        if "--only-pechim" in sys.argv:
            make_pechim(tmp)
            return
        if "--only-chimml" in sys.argv:
            make_chimml(tmp)
            return
        if "--only-chimmlindel" in sys.argv:
            make_chimmlindel(tmp)
            return
        if "--only-splice" in sys.argv:
            make_splice(tmp)
            return
        if "--only-trim" in sys.argv:
            make_trim(tmp)
            return
        if "--only-indel" in sys.argv:
            make_indel(tmp)
            return
        if "--only-multi-best" in sys.argv:
            make_multi_best(tmp)
            return
        if "--only-multi" in sys.argv:
            make_multi(tmp)
            return
        if "--only-formats" in sys.argv:
            make_formats(tmp)
            return
        if "--only-stats" in sys.argv:
            make_stats(tmp)
            return
        if "--only-fastq" in sys.argv:
            make_fastq(tmp)
            return
        if "--only-bam" in sys.argv:
            make_bam(tmp)
            return
        if "--only-lengths" in sys.argv:
            make_lengths(tmp)
            return
        if "--only-sortorder" in sys.argv:
            make_sortorder(tmp)
            return
        make_basic(tmp)
        make_repeat(tmp)
        make_sortorder(tmp)
        make_pe(tmp)
        make_pe(tmp, L=150, name="pe150")
        make_lengths(tmp)
        make_bam(tmp)
        make_fastq(tmp)
        make_stats(tmp)
        make_formats(tmp)
        make_multi(tmp)
        make_multi_best(tmp)
        make_indel(tmp)
        make_trim(tmp)
        make_splice(tmp)
        make_chimeric(tmp)
        make_combined(tmp)
        make_quality(tmp)
        make_pcr(tmp)
        make_simreads(tmp)
        make_lifted(tmp)
    print("done")


if __name__ == "__main__":
    sys.exit(main())
