#!/usr/bin/env python3
"""Where in the suffix array do the wave kernel's candidates lie?  (round 5, task 1: the measurement in front of the partial window array)

Runs ONE step of a bench.py configuration (default C2: 50 M x 100 bp, -s3, 3.1 Gbp) against a build of the library whose k_wave counts
every window it fetches per block of 64 suffix array indexes and per length of the core interval it came from
(tools/build_variant.sh hist "-DBK_CAND_HIST" bk_wave.hip), then prices coverage rules for a suffix-ordered window array that holds
only part of the suffix array:
  * by measurement: the blocks sorted by their count - the best any rule could do;
  * by structure: block b is covered when its first and last suffix share at least W bases (the whole block then lies inside ONE
    interval of a W-base core) - a rule the index set-up can apply without having seen a read.
Writes a CSV (stdout or --out).  Usage (GPU box):
  BK_LIB=biokanga_amd/lib/libbiokanga_amd_hist.so python3 tools/cand_hist.py --out gpurun_out/r05_a/cand_hist.csv
"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=3100.0)
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--max-subs", type=int, default=3)
    ap.add_argument("--pairs", action="store_true", help="2 x read-len FR pairs (C3's reads) instead of single ends")
    ap.add_argument("--window-array", type=int, default=0, help="bk_ctx_tune use_swin of the measured step (0 none, 1 partial, 3 every suffix): the per-phase "
                                                                    "section then says which windows the array served")
    ap.add_argument("--no-rules", action="store_true", help="skip the pricing of coverage rules")
    ap.add_argument("--wheat", action="store_true", help="bench.py's C5: the wheat-like genome (21 sequences, 85 %% repeat-derived, seed 17; give --genome-mbp 17000) with 5-byte "
                                                         "suffix elements, 2 x 150 bp pairs, -s5; only the measured 'best possible' coverage is priced (entries of 80 bytes)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    import biokanga_amd as bk
    from biokanga_amd import synth
    lib = bk.load_library()
    if not hasattr(lib, "bk_debug_cand_hist"):
        raise SystemExit("this library was not built with -DBK_CAND_HIST (tools/build_variant.sh hist \"-DBK_CAND_HIST\" bk_wave.hip; BK_LIB=...)")
    lib.bk_debug_cand_hist.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_ulonglong]
    dev = torch.device("cuda", 0)
    t0 = time.time()
    if args.wheat:
        args.pairs, args.read_len, args.max_subs = True, 150, 5
        seq, seq_lens = synth.make_genome(int(args.genome_mbp * 1e6), dev, seed=17, n_seqs=21, repeat_frac=0.85)
    else:
        seq, seq_lens = synth.make_genome(int(args.genome_mbp * 1e6), dev)
    n = seq.numel()
    E = 5 if (args.wheat or n >= 0xFFFFFFFF) else 4
    sa = torch.empty(n * 5, dtype=torch.uint8, device=dev) if E == 5 else torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), E, 0)
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    al = bk.Aligner(None, bk.AlignParams(max_subs=args.max_subs), device=0, d_seq=seq.data_ptr(), concat_len=n, d_sa=sa.data_ptr(), el_size=E, entries=ent)
    al.tune("use_swin", args.window_array)
    if E == 5:
        # (as bench.py does: the context holds its own image; the tool's copies leave the HBM before the second-level keys are made)
        del sa
        seq_keep = seq
        torch.cuda.empty_cache()
        al.tune("use_k2", 1)
    if args.pairs:
        bases, offs, lens = synth.make_pairs(seq, seq_lens, args.reads // 2, args.read_len, dev, seed=1000, max_subs=args.max_subs)
    else:
        bases, offs, lens, _ = synth.make_reads(seq, seq_lens, args.reads, args.read_len, dev, seed=1000, max_subs=args.max_subs)
    out = torch.zeros(args.reads * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    print(f"set-up {time.time() - t0:.1f}s: {n} bases, MinCoreLen {al.min_core_len}", file=sys.stderr)
    if lib.bk_debug_cand_hist(0, None, n):
        raise SystemExit("histogram allocation failed")
    al.counters(reset=True)
    al.align_device(bases.data_ptr(), offs.data_ptr(), lens.data_ptr(), args.reads, out.data_ptr())
    torch.cuda.synchronize()
    ctr = al.counters()
    nblk = (n >> 6) + 1
    blk = np.zeros(nblk, dtype=np.uint32)
    if lib.bk_debug_cand_hist(1, blk.ctypes.data, nblk):
        raise SystemExit("histogram read-back failed")
    hl = np.zeros((3, 40), dtype=np.uint64)
    lib.bk_debug_cand_hist(2, hl.ctypes.data, 0)
    hp = np.zeros((8, 40, 2), dtype=np.uint64)
    lib.bk_debug_cand_hist(4, hp.ctypes.data, 0)
    lib.bk_debug_cand_hist(3, None, 0)
    total = int(blk.sum(dtype=np.uint64))
    lines = []
    w = lines.append
    w(f"# tools/cand_hist.py: {args.reads} reads x {args.read_len} bp{' (pairs)' if args.pairs else ''}, -s{args.max_subs}, genome {n} concatenated bases; one step, window array off")
    w(f"# counters of the step: n_search {ctr['n_search']} n_cand {ctr['n_cand']} n_cand_heavy (k_wave) {ctr['n_cand_heavy']}; windows k_wave fetched {total}")
    w("section,interval_length_from,interval_length_to,intervals,windows_fetched,candidates_processed,share_of_windows")
    for b in range(40):
        if hl[0, b] or hl[1, b]:
            w(f"by_interval_length,{1 << b},{(2 << b) - 1},{int(hl[0, b])},{int(hl[1, b])},{int(hl[2, b])},{int(hl[1, b]) / max(1, total):.4f}")
    w("section,phase,interval_length_from,interval_length_to,windows_fetched,from_the_window_array,share_of_all_windows,share_served")
    for ph in range(8):
        for b in range(40):
            if hp[ph, b, 0]:
                w(f"by_phase,{ph},{1 << b},{(2 << b) - 1},{int(hp[ph, b, 0])},{int(hp[ph, b, 1])},{int(hp[ph, b, 0]) / max(1, total):.4f},{int(hp[ph, b, 1]) / int(hp[ph, b, 0]):.3f}")
        if hp[ph, :, 0].sum():
            w(f"by_phase_total,{ph},,,{int(hp[ph, :, 0].sum())},{int(hp[ph, :, 1].sum())},{int(hp[ph, :, 0].sum()) / max(1, total):.4f},{int(hp[ph, :, 1].sum()) / int(hp[ph, :, 0].sum()):.3f}")
    if args.no_rules:
        text = "\n".join(lines) + "\n"
        if args.out:
            os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
            open(args.out, "w").write(text)
        sys.stdout.write(text)
        al.close()
        return
    # the best any rule could do: blocks by count
    order = np.sort(blk)[::-1].astype(np.uint64)
    cum = np.cumsum(order)
    nz = int((blk > 0).sum())
    eb = 80 if args.read_len > 128 else 48                 # bytes of a window array entry of the kernel family these reads use
    w(f"section,blocks_of_64_covered,share_of_suffix_array,GB_at_{eb}B_per_suffix,share_of_windows_served")
    for frac in (0.001, 0.002, 0.005, 0.01, 0.02, 0.03, 0.05, 0.075, 0.10, 0.15, 0.20, 0.30, 0.40, 0.50, 1.0):
        k = min(nblk, max(1, int(frac * nblk)))
        w(f"best_possible,{k},{k / nblk:.4f},{k * 64 * eb / 1e9:.2f},{int(cum[k - 1]) / max(1, total):.4f}")
    w(f"# blocks with at least one window: {nz} = {nz / nblk:.4f} of the suffix array")
    if E == 5:
        text = "\n".join(lines) + "\n"
        if args.out:
            os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
            open(args.out, "w").write(text)
        sys.stdout.write(text)
        al.close()
        return
    # the structural rule: first and last suffix of the block share >= W bases
    sa64 = sa.to(torch.int64)
    first = sa64[0::64][: nblk - 1]
    last = sa64[63::64][: nblk - 1]
    m = min(first.numel(), last.numel())
    first, last = first[:m], last[:m]
    lcp = torch.zeros(m, dtype=torch.int32, device=dev)
    alive = torch.ones(m, dtype=torch.bool, device=dev)
    for k in range(40):
        a = seq[(first + k).clamp(max=n - 1)]
        b2 = seq[(last + k).clamp(max=n - 1)]
        alive &= (a == b2) & (a < 4)
        lcp += alive.to(torch.int32)
    lcp = lcp.cpu().numpy()
    bl = blk[:m].astype(np.uint64)
    w("section,W_bases_shared_by_first_and_last_suffix,blocks_of_64_covered,share_of_suffix_array,GB_at_48B_per_suffix,share_of_windows_served,share_of_windows_in_rounds_whose_neighbour_block_is_covered_too")
    for W in (14, 16, 18, 20, 22, 24, 25, 26, 28, 30, 32, 33, 36, 40):
        cov = lcp >= W
        k = int(cov.sum())
        served = int(bl[cov].sum())
        both = cov.copy()
        both[:-1] &= cov[1:]
        both[1:] &= cov[:-1]
        served_both = int(bl[both].sum())
        w(f"shared_prefix_rule,{W},{k},{k / nblk:.4f},{k * 64 * 48 / 1e9:.2f},{served / max(1, total):.4f},{served_both / max(1, total):.4f}")
    text = "\n".join(lines) + "\n"
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            f.write(text)
    sys.stdout.write(text)
    al.close()


if __name__ == "__main__":
    main()
