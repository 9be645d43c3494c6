#!/usr/bin/env python3
"""Randomised differential test of the device path against the CPU oracle over EVERY mode of the boundary: random option sets
(substitutions, Hamming delta, strand, sensitivity, N policy; paired-end association in every -U mode; multi-loci with / without clamp, with -a / -A; SNP pile-up and screening of the accepted reads;
best matches; microInDels, splice
junctions, chimeric trimming and their combinations) on reads built to exercise them (substitutions, insertions / deletions,
introns, foreign ends, Ns, repeats, ragged lengths) against a repeat-rich synthetic genome.  Compares every bk_hit field, the loci
lists and the second-segment records; stops at the first disagreement and prints the case.
  python tools/fuzz_parity.py [rounds] [reads_per_round] [genome_mbp] [seed] [suffix element bytes 4|5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import biokanga_amd as bk
from biokanga_amd import synth
import helpers

FIELDS = ["chrom_id", "match_loci", "match_len", "low_hit_instances", "rslt", "nar", "strand", "low_mm", "nxt_low_mm", "num_hits", "mismatches"]
SEG = ["match_loci", "match_len", "read_ofs", "mismatches", "flags", "score"]
LOCI = ["chrom_id", "match_loci", "match_len", "strand", "mismatches"]


def make_reads(rng, g, n, max_len):
    """reads from concatenated genome bytes g (values 0..4, 7 = EOS between sequences)"""
    comp = np.array([3, 2, 1, 0, 4, 5, 6, 7], dtype=np.uint8)
    out, lens = [], []
    N = len(g)
    for _ in range(n):
        L = int(rng.integers(30, max_len + 1)) if rng.integers(0, 4) == 0 else min(100, max_len)
        st = int(rng.integers(0, N - 2 * L - 6000))
        kind = int(rng.integers(0, 10))
        if kind == 0:                                    # deletion from the read
            k = int(rng.integers(1, 12)); p = int(rng.integers(8, L - 8))
            r = np.concatenate([g[st:st + p], g[st + p + k:st + L + k]])
        elif kind == 1:                                  # insertion into the read
            k = int(rng.integers(1, 12)); p = int(rng.integers(8, L - 8 - k)) if L - 16 - k > 0 else 8
            r = np.concatenate([g[st:st + p], rng.integers(0, 4, k, dtype=np.uint8), g[st + p:st + L - k]])
        elif kind == 2:                                  # intron
            gap = int(rng.choice([25, 40, 90, 300, 1500, 5000])); p = int(rng.integers(12, L - 12))
            r = np.concatenate([g[st:st + p], g[st + p + gap:st + gap + L]])
        elif kind == 3:                                  # foreign end(s)
            k5 = int(rng.integers(0, L // 3)); k3 = int(rng.integers(0, L // 3)) if rng.integers(0, 2) else 0
            r = g[st:st + L].copy()
            r[:k5] = rng.integers(0, 4, k5, dtype=np.uint8)
            if k3:
                r[L - k3:] = rng.integers(0, 4, k3, dtype=np.uint8)
        else:
            r = g[st:st + L].copy()
        r = r[:L].copy()
        if len(r) < 20 or (r > 4).any():
            r = g[1000:1000 + 60].copy()
        for q in rng.choice(len(r), int(rng.integers(0, 5)), replace=False):
            r[q] = (r[q] + rng.integers(1, 4)) % 4 if r[q] < 4 else r[q]
        if rng.integers(0, 30) == 0:
            r[int(rng.integers(0, len(r)))] = 4
        if rng.integers(0, 2):
            r = comp[r[::-1]]
        out.append(r.astype(np.uint8)); lens.append(len(r))
    lens = np.array(lens, dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
    return np.concatenate(out), offs, lens


def make_pairs(rng, g, n_pairs, L):
    """FR pairs (insert 120..900, some far apart / same strand / one mate mutated beyond recognition), interleaved PE1, PE2"""
    comp = np.array([3, 2, 1, 0, 4, 5, 6, 7], dtype=np.uint8)
    N = len(g)
    out = []
    for _ in range(n_pairs):
        ins = int(rng.integers(120, 900)) if rng.integers(0, 8) else int(rng.integers(2000, 20000))
        st = int(rng.integers(0, N - ins - 2 * L - 100))
        a = g[st:st + L].copy()
        b = comp[g[st + max(ins, L) - L:st + max(ins, L)][::-1]].copy()
        if rng.integers(0, 12) == 0:
            b = comp[b[::-1]].copy()                     # same strand
        for r in (a, b):
            if (r > 4).any():
                r[:] = g[1000:1000 + L]
            k = int(rng.integers(0, 5)) if rng.integers(0, 10) else int(rng.integers(10, 25))
            for q in rng.choice(L, k, replace=False):
                r[q] = (r[q] + rng.integers(1, 4)) % 4 if r[q] < 4 else r[q]
        if rng.integers(0, 2):
            a, b = b, a
        out += [a, b]
    lens = np.full(2 * n_pairs, L, dtype=np.uint32)
    offs = (np.arange(2 * n_pairs, dtype=np.uint64) * L)
    return np.concatenate(out), offs, lens


def random_params(rng):
    kw = dict(max_subs=int(rng.choice([0, 1, 2, 3, 5, 8, 10])), min_edit_dist=int(rng.integers(1, 3)), align_strand=int(rng.choice([0, 0, 1, 2])),
              pmode=int(rng.integers(0, 4)), max_ns=int(rng.choice([0, 1, 1, 3])))
    mode = int(rng.integers(0, 4))
    if mode == 1:
        kw.update(max_ml=int(rng.choice([2, 3, 5, 20, 500])), clamp_ml=int(rng.integers(0, 2)))
        pick = int(rng.integers(0, 4))
        if pick == 0:                                    # -r1..-r4 take -a / -A too
            if rng.integers(0, 2):
                kw["micro_indel_len"] = int(rng.integers(1, 21))
            else:
                kw["splice_junct_len"] = int(rng.choice([25, 100, 2000, 6000]))
        elif pick == 1:                                  # .. and -c: the chimeric call lists its loci, each with its trims
            kw["min_chimeric_len"] = int(rng.integers(50, 100))
    elif mode == 2:
        kw.update(max_ml=int(rng.choice([2, 5, 50])), best_matches=1)
    elif mode == 3:
        if rng.integers(0, 2):
            kw["micro_indel_len"] = int(rng.integers(1, 21))
        if rng.integers(0, 2):
            kw["splice_junct_len"] = int(rng.choice([25, 100, 2000, 6000]))
        if rng.integers(0, 2):
            kw["min_chimeric_len"] = int(rng.integers(50, 100))
    return kw


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
    mbp = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    el = int(sys.argv[5]) if len(sys.argv) > 5 else 4             # 5: the same index as 5-byte suffix elements (the > 4 Gbp code paths)
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    seq, seq_lens = synth.make_genome(int(mbp * 1e6), dev, seed=100 + seed, n_seqs=7)
    n = seq.numel()
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), 4, 0)
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    seq_h, sa_h = seq.cpu().numpy(), sa.cpu().numpy()
    if el == 5:
        sa5 = np.zeros((n, 5), dtype=np.uint8)
        sa5[:, :4] = sa_h.view(np.uint32).astype("<u4").view(np.uint8).reshape(n, 4)
        sa_h = sa5.reshape(-1)
        d_sa5 = torch.from_numpy(sa_h).to(dev)
        sa_ptr = d_sa5.data_ptr()
    else:
        sa_ptr = sa.data_ptr()
    ora = helpers.OracleSfx(seq=seq_h, sa=sa_h, el_size=el, entries=entries)
    al = bk.Aligner(None, bk.AlignParams(max_subs=3), device=0, d_seq=seq.data_ptr(), concat_len=n, d_sa=sa_ptr, el_size=el, entries=ent)
    t0 = time.time()
    for it in range(rounds):
        if rng.integers(0, 5) == 0:                      # a paired-end round: SE pass, then the association on both sides
            kw = dict(max_subs=int(rng.choice([1, 3, 5, 8])), min_edit_dist=int(rng.integers(1, 3)), pmode=int(rng.integers(0, 4)))
            if rng.integers(0, 3) == 0:
                kw["min_chimeric_len"] = int(rng.integers(50, 100))       # -c with -U: trims into and out of the association
            L = int(rng.choice([50, 100, 150, 250, 300, 450]))
            bases, offs, lens = make_pairs(rng, seq_h, nreads // 2, L)
            pe = dict(pe_mode=int(rng.integers(1, 5)), pair_min_len=int(rng.choice([100, 200])), pair_max_len=int(rng.choice([400, 1000, 5000])),
                      pair_strand=int(rng.integers(0, 6) == 0))
            al.set_params(bk.AlignParams(**kw))
            al.tune("chunk_reads", 64 << 20); al.tune("use_wave", 1)
            if pe["pair_min_len"] < L:
                pe["pair_min_len"] = L
            if pe["pair_max_len"] < pe["pair_min_len"] + 100:
                pe["pair_max_len"] = pe["pair_min_len"] + 300
            if kw.get("min_chimeric_len"):                   # a third of the mates get a foreign end
                for i in rng.choice(len(lens), len(lens) // 3, replace=False):
                    k = int(rng.integers(5, L // 2))
                    o = int(offs[i])
                    if rng.integers(0, 2):
                        bases[o:o + k] = rng.integers(0, 4, k, dtype=np.uint8)
                    else:
                        bases[o + L - k:o + L] = rng.integers(0, 4, k, dtype=np.uint8)
            got = al.align(bases, offs, lens)
            p = helpers.make_params(**kw)
            exp, eseg = helpers.oracle_align_indel(ora, bases, offs, lens, p, nthreads=os.cpu_count())
            bad = None
            for f in FIELDS:
                if not np.array_equal(got[f], exp[f]):
                    bad = "SE pass " + f
            pep = bk.PEParams(pe["pe_mode"], pe["pair_min_len"], pe["pair_max_len"], bool(pe["pair_strand"]))
            # one round in three: chromosome filters inside the pair rules (-Z / -z with -U): a random accept table by sequence id
            accept = None
            if rng.integers(0, 3) == 0:
                accept = np.ones(int(max(e[0] for e in entries)) + 1, dtype=np.uint8)
                accept[rng.integers(1, len(accept), max(1, len(accept) // 3))] = 0
                pe["filtered_ids"] = [int(i) for i in np.nonzero(accept == 0)[0]]
            al.set_chrom_filter(accept)
            if kw.get("min_chimeric_len"):
                gp, gseg = al.pair(bases, offs, lens, got.copy(), pep, seg2=al.batch_seg2())
                ep = helpers.oracle_process_pe(ora, p, pe["pe_mode"], pe["pair_min_len"], pe["pair_max_len"], bool(pe["pair_strand"]), bases, offs, lens, exp.copy(), eseg, accept=accept)
                for f in SEG:
                    if not bad and not np.array_equal(gseg[f], eseg[f]):
                        i = int(np.nonzero(gseg[f] != eseg[f])[0][0])
                        bad = f"PE seg2.{f} at read {i}: gpu {gseg[i]} {gp[i]} oracle {eseg[i]} {ep[i]}"
            else:
                gp = al.pair(bases, offs, lens, got.copy(), pep)
                ep = helpers.oracle_process_pe(ora, p, pe["pe_mode"], pe["pair_min_len"], pe["pair_max_len"], bool(pe["pair_strand"]), bases, offs, lens, exp.copy(), accept=accept)
            al.set_chrom_filter(None)
            for f in [x for x in FIELDS if x != "rslt"]:
                if not np.array_equal(gp[f], ep[f]):
                    i = int(np.nonzero(gp[f] != ep[f])[0][0])
                    bad = f"PE {f} at read {i}: gpu {gp[i]} oracle {ep[i]}"
                    break
            if not bad and not np.array_equal(gp["flags"] & 0x80, ep["flags"] & 0x80):
                bad = "PE paired flag"
            nar, cnt = np.unique(gp["nar"], return_counts=True)
            print(f"round {it}: PE {kw} {pe} L {L}: {'OK' if not bad else 'MISMATCH ' + bad}; NAR {({bk.NAR_TAGS[int(k)]: int(v) for k, v in zip(nar, cnt)})}", flush=True)
            if bad:
                sys.exit(1)
            continue
        kw = random_params(rng)
        max_len = int(rng.choice([300, 500, 600])) if kw.get("min_chimeric_len") else int(rng.choice([100, 150, 256, 320, 400, 512, 700]))
        bases, offs, lens = make_reads(rng, seq_h, nreads, max_len)
        p = helpers.make_params(**kw)
        al.set_params(bk.AlignParams(**kw))
        # (the window array: none / the partial one / one cut off by a byte budget / every suffix - dropped first so that the next batch makes its own)
        for knob, val in (("chunk_reads", int(rng.choice([64 << 20, 7001]))), ("use_wave", int(rng.integers(0, 5) != 0)),
                          ("use_swin", 0), ("swin_budget_kb", int(rng.choice([0, 0, 3, 30, 300]))), ("use_swin", int(rng.choice([0, 1, 2, 2, 3])))):
            al.tune(knob, val)
        if int(rng.integers(0, 6)) == 0:                 # (the key arrays behind the second-level keys: none, one, both - the tables are rebuilt)
            al.tune("use_k3", int(rng.integers(0, 3)))
        if int(rng.integers(0, 8)) == 0:                 # (the k-mer table as an index beyond 2^32 suffixes has it: packed, plain 64-bit, back - rebuilt)
            al.tune("ktab_wide", int(rng.integers(0, 3)))
        packed = bool(rng.integers(0, 2))                # the same reads across the boundary at 2 bit/base
        got = al.align_packed(*bk.pack_reads(bases, offs, lens)) if packed else al.align(bases, offs, lens)
        kw = dict(kw, packed=packed)
        bad = None
        if kw.get("max_ml", 1) > 1:
            exp, eo, el, etr, eseg = helpers.oracle_align_multi_chimeric(ora, bases, offs, lens, p, nthreads=os.cpu_count())
            lo, loci = al.batch_loci(len(lens))
            if not np.array_equal(lo, eo) or any(not np.array_equal(loci[f], el[f]) for f in LOCI):
                bad = "loci lists"
            if kw.get("min_chimeric_len"):
                tr = al.batch_loci_trims()
                if len(tr) != len(etr) or any(not np.array_equal(tr[f], etr[f]) for f in ("left", "right", "chimeric")):
                    bad = "loci trims"
                seg = al.batch_seg2()
                for f in SEG:
                    if not np.array_equal(seg[f], eseg[f]):
                        bad = f"seg2.{f} (multi-loci run with -c)"
            if kw.get("micro_indel_len") or kw.get("splice_junct_len"):
                seg = al.batch_seg2()
                for f in SEG:
                    if not np.array_equal(seg[f], eseg[f]):
                        bad = f"seg2.{f} (multi-loci run)"
        else:
            exp, eseg = helpers.oracle_align_indel(ora, bases, offs, lens, p, nthreads=os.cpu_count())
            if any(kw.get(k) for k in ("micro_indel_len", "splice_junct_len", "min_chimeric_len")):
                seg = al.batch_seg2()
                for f in SEG:
                    if not np.array_equal(seg[f], eseg[f]):
                        i = int(np.nonzero(seg[f] != eseg[f])[0][0])
                        bad = f"seg2.{f} at read {i}: gpu {seg[i]} {got[i]} oracle {eseg[i]} {exp[i]}"
                        break
        for f in FIELDS:
            if not np.array_equal(got[f], exp[f]):
                i = int(np.nonzero(got[f] != exp[f])[0][0])
                bad = f"{f} at read {i} (len {lens[i]}): gpu {got[i]} oracle {exp[i]}"
                break
        if not bad and it % 3 == 0:                      # SNP pile-up of what was accepted, one random sequence screened against the oracle
            acc = np.nonzero((got["nar"] == 1) & (got["strand"] != ord("?")))[0]
            if kw.get("micro_indel_len") or kw.get("splice_junct_len") or kw.get("min_chimeric_len"):
                acc = acc[(al.batch_seg2()["flags"][acc] & 13) == 0]
            alns = np.zeros(len(acc), dtype=bk.SNP_ALN_DTYPE)
            alns["read_idx"] = acc; alns["chrom_id"] = got["chrom_id"][acc]; alns["loci"] = got["match_loci"][acc]
            alns["len"] = got["match_len"][acc]; alns["strand"] = got["strand"][acc]
            if len(alns):
                al.snp_reset(); al.snp_pileup(bases, offs, lens, alns)
                chrom = int(rng.choice(np.unique(alns["chrom_id"])))
                mr, pr = int(rng.choice([1, 2, 5])), float(rng.choice([0.001, 0.1, 0.25]))
                gs, gt = al.snp_sites(chrom, mr, pr)
                es, et = helpers.oracle_snp_sites(ora.h, bases, offs, alns, chrom, mr, pr, max_sites=1 << 22)
                if len(gs) != len(es) or not np.array_equal(gt, et) or any(not np.array_equal(gs[f], es[f]) for f in es.dtype.names):
                    bad = f"SNP sites of sequence {chrom} (-p{mr} -1 {pr * 100}): gpu {len(gs)} {gt.tolist()} oracle {len(es)} {et.tolist()}"
        nar, cnt = np.unique(got["nar"], return_counts=True)
        print(f"round {it}: {kw} max_len {max_len}: {'OK' if not bad else 'MISMATCH ' + bad}; NAR {({bk.NAR_TAGS[int(k)]: int(v) for k, v in zip(nar, cnt)})}", flush=True)
        if bad:
            np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{seed}_{it}_bases.npy"), bases)
            np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{seed}_{it}_lens.npy"), lens)
            sys.exit(1)
    print(f"{rounds} rounds x {nreads} reads: no disagreement ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
