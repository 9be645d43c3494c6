"""Seeded synthetic benchmark inputs, generated directly in HBM with torch (plumbing: the real
genomes - E. coli, GRCh38, wheat - are not available offline and `biokanga simreads` is seeded from
time(0)).

  make_genome()  "GRCh38-like" concatenation as a .sfx holds it: sequences with the length ratios of
                 chr1..22,X,Y, i.i.d. ACGT background, ~45 % of the bases overwritten with diverged
                 copies of a repeat-family library (300 bp .. 6 kbp consensi, 1-20 % divergence,
                 copy numbers from a power law), N gaps whose every 13th base is a random base (what
                 kangax.cpp:628-651 does to N runs > 25 at index time), eBaseEOS after every entry.
  make_reads()   `simreads`-like SE reads: uniform start and strand, substitution count uniform on
                 {0..max_subs} at uniform positions.
"""
import numpy as np
import torch

# GRCh38 primary assembly chromosome lengths (chr1..22, X, Y) - only their ratios are used
GRCH38_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
               138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
               83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]


def entry_table(seq_lens):
    """[(entry_id, seq_len, start_ofs, end_ofs)] for sequences concatenated with one EOS each."""
    out, ofs = [], 0
    for i, n in enumerate(seq_lens):
        out.append((i + 1, int(n), ofs, ofs + int(n) - 1))
        ofs += int(n) + 1
    return out


def make_genome(total_bp, device, seed=38, n_seqs=24, repeat_frac=0.45, n_gap_frac=0.02):
    """-> (seq uint8 tensor [concat_len] on device, seq_lens list).  Deterministic for a given
    (total_bp, seed, n_seqs)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    ratios = np.array(GRCH38_LENS[:n_seqs], dtype=np.float64)
    lens = np.maximum(1000, (ratios / ratios.sum() * total_bp).astype(np.int64))
    n_bases = int(lens.sum())
    bases = torch.randint(0, 4, (n_bases,), dtype=torch.uint8, device=device, generator=g)

    # repeat families
    target = int(n_bases * repeat_frac)
    placed = 0
    fam = 0
    while placed < target and fam < 4000:
        fam += 1
        L = int(rng.integers(300, 6001))
        div = float(rng.uniform(0.01, 0.20))
        # copy number: power law, bounded so one family never exceeds 10 % of the target
        copies = int(min(10 ** rng.uniform(1.0, 6.0), max(10, 0.10 * target / L), (target - placed) / L + 1))
        cons = torch.randint(0, 4, (L,), dtype=torch.uint8, device=device, generator=g)
        done = 0
        while done < copies:
            c = min(copies - done, max(1, (1 << 27) // L))        # <= 128 M elements per chunk
            pos = torch.randint(0, n_bases - L, (c,), dtype=torch.int64, device=device, generator=g)
            # copies of one chunk may overlap; an indexed store with duplicate indices is not deterministic on
            # the GPU, so the copies are laid down in position order and every base is written only by the
            # LAST copy covering it (= the part of a copy before the next copy's start)
            pos = torch.sort(pos).values
            nxt = torch.cat([pos[1:], torch.full((1,), n_bases, dtype=torch.int64, device=device)])
            ar = torch.arange(L, device=device)
            idx = (pos[:, None] + ar[None, :])
            keep = (idx < nxt[:, None]).reshape(-1)
            idx = idx.reshape(-1)
            mut = torch.rand((c * L,), device=device, generator=g) < div
            delta = torch.randint(1, 4, (c * L,), dtype=torch.uint8, device=device, generator=g)
            val = cons.repeat(c)
            val = torch.where(mut, (val + delta) & 3, val)
            bases[idx[keep]] = val[keep]
            done += c
            del pos, nxt, idx, keep, mut, delta, val
        placed += copies * L

    # N gaps: runs of 10 k .. 3 M (scaled down for small genomes)
    gap_budget = int(n_bases * n_gap_frac)
    max_gap = max(1000, min(3_000_000, n_bases // 200))
    while gap_budget > 0:
        glen = int(min(gap_budget, rng.integers(min(10_000, max_gap // 2), max_gap + 1)))
        start = int(rng.integers(0, n_bases - glen))
        run = torch.full((glen,), 4, dtype=torch.uint8, device=device)
        # kangax.cpp:628-651: inside N runs > 25 every 13th N becomes a random base
        k = torch.arange(25, glen, 13, device=device)
        if len(k):
            run[k] = torch.randint(0, 4, (len(k),), dtype=torch.uint8, device=device, generator=g)
        bases[start:start + glen] = run
        gap_budget -= glen

    # concatenate with EOS terminators
    concat_len = n_bases + len(lens)
    seq = torch.empty(concat_len, dtype=torch.uint8, device=device)
    src = 0
    dst = 0
    for n in lens:
        n = int(n)
        seq[dst:dst + n] = bases[src:src + n]
        seq[dst + n] = 7
        src += n
        dst += n + 1
    return seq, [int(x) for x in lens]


def make_reads(seq, seq_lens, n_reads, read_len, device, seed=2, max_subs=3):
    """-> (bases uint8 [n_reads*read_len], offs int64 [n_reads], lens int32 [n_reads], truth dict)"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ent = entry_table(seq_lens)
    lens_t = torch.tensor([e[1] for e in ent], dtype=torch.float64, device=device)
    starts_t = torch.tensor([e[2] for e in ent], dtype=torch.int64, device=device)
    chrom = torch.multinomial(lens_t / lens_t.sum(), n_reads, replacement=True, generator=g)
    span = (lens_t[chrom] - read_len + 1).clamp(min=1)
    local = (torch.rand(n_reads, dtype=torch.float64, device=device, generator=g) * span).to(torch.int64)
    start = starts_t[chrom] + local
    out = torch.empty((n_reads, read_len), dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, device=device)
    chunk = max(1, (1 << 26) // read_len)
    nsubs = torch.randint(0, max_subs + 1, (n_reads,), device=device, generator=g)
    strand = torch.randint(0, 2, (n_reads,), device=device, generator=g).bool()
    comp = torch.tensor([3, 2, 1, 0, 4, 5, 6, 7], dtype=torch.uint8, device=device)
    for lo in range(0, n_reads, chunk):
        hi = min(n_reads, lo + chunk)
        r = seq[(start[lo:hi, None] + ar[None, :])]
        # substitutions at distinct uniform positions: rank of a random key below the per-read count
        key = torch.rand((hi - lo, read_len), device=device, generator=g)
        rank = key.argsort(dim=1).argsort(dim=1)
        mut = (rank < nsubs[lo:hi, None]) & (r < 4)
        delta = torch.randint(1, 4, (hi - lo, read_len), dtype=torch.uint8, device=device, generator=g)
        r = torch.where(mut, (r + delta) & 3, r)
        rc = comp[r.long()].flip(1)
        out[lo:hi] = torch.where(strand[lo:hi, None], rc, r)
        del r, key, rank, mut, delta, rc
    offs = torch.arange(n_reads, dtype=torch.int64, device=device) * read_len
    lens = torch.full((n_reads,), read_len, dtype=torch.int32, device=device)
    truth = {"chrom": chrom, "local": local, "strand": strand, "nsubs": nsubs}
    return out.reshape(-1), offs, lens, truth


def make_pairs(seq, seq_lens, n_pairs, read_len, device, seed=3, max_subs=5, ins_mean=300.0, ins_sd=50.0, ins_lo=200, ins_hi=400):
    """FR paired ends (C3 of SURVEY.md §8d): insert ~ N(ins_mean, ins_sd) clipped to [ins_lo, ins_hi], substitutions
    per read uniform on 0..max_subs; half of the fragments come from the '-' strand.
    -> (bases uint8 [2*n_pairs*read_len] interleaved PE1,PE2, offs int64, lens int32)"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ent = entry_table(seq_lens)
    lens_t = torch.tensor([e[1] for e in ent], dtype=torch.float64, device=device)
    starts_t = torch.tensor([e[2] for e in ent], dtype=torch.int64, device=device)
    chrom = torch.multinomial(lens_t / lens_t.sum(), n_pairs, replacement=True, generator=g)
    ins = (torch.randn(n_pairs, device=device, generator=g) * ins_sd + ins_mean).round().clamp(ins_lo, ins_hi).to(torch.int64)
    ins = torch.maximum(ins, torch.full_like(ins, read_len))
    span = (lens_t[chrom] - ins.double() + 1).clamp(min=1)
    local = (torch.rand(n_pairs, dtype=torch.float64, device=device, generator=g) * span).to(torch.int64)
    start = starts_t[chrom] + local
    flip = torch.randint(0, 2, (n_pairs,), device=device, generator=g).bool()
    comp = torch.tensor([3, 2, 1, 0, 4, 5, 6, 7], dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, device=device)
    out = torch.empty((n_pairs, 2, read_len), dtype=torch.uint8, device=device)
    chunk = max(1, (1 << 25) // read_len)
    for lo in range(0, n_pairs, chunk):
        hi = min(n_pairs, lo + chunk)
        left = seq[(start[lo:hi, None] + ar[None, :])]
        right = seq[(start[lo:hi, None] + ins[lo:hi, None] - read_len + ar[None, :])]
        mates = []
        for r in (left, right):
            nsubs = torch.randint(0, max_subs + 1, (hi - lo,), device=device, generator=g)
            key = torch.rand((hi - lo, read_len), device=device, generator=g)
            rank = key.argsort(dim=1).argsort(dim=1)
            mut = (rank < nsubs[:, None]) & (r < 4)
            delta = torch.randint(1, 4, (hi - lo, read_len), dtype=torch.uint8, device=device, generator=g)
            mates.append(torch.where(mut, (r + delta) & 3, r))
        l, r = mates
        rc_r = comp[r.long()].flip(1)
        rc_l = comp[l.long()].flip(1)
        f = flip[lo:hi, None]
        out[lo:hi, 0] = torch.where(f, rc_r, l)        # '+' fragment: PE1 = left fwd, PE2 = right revcomp
        out[lo:hi, 1] = torch.where(f, l, rc_r)        # '-' fragment: PE1 = right revcomp ... seen from the other strand
        del left, right, l, r, rc_r, rc_l
    n = 2 * n_pairs
    offs = torch.arange(n, dtype=torch.int64, device=device) * read_len
    lens = torch.full((n,), read_len, dtype=torch.int32, device=device)
    return out.reshape(-1), offs, lens
