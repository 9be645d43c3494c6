#!/usr/bin/env python3
"""bench.py - headline benchmark of the `biokanga align` hot path on MI355X.

Metric (BASELINE.json): aligned reads/s, results identical to the reference, 100 bp SE reads vs a
GRCh38-scale index.  One "step" = one pass of the hot path (bk_align_batch_device: pack, SA-interval
search, candidate walk + Hamming extension, classification - every AlignReads phase) over one batch
of synthetic reads that is already resident in HBM, followed by the path's only exchange step, the
sum-reduction of the per-sequence hit counts (RCCL all-reduce when N > 1).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload at N = 1: BASELINE.json configs[1] restated over a synthetic genome (SURVEY.md §8d "C2"):
50 M x 100 bp SE reads, 0-3 substitutions, vs a 24-sequence 3.1 Gbp GRCh38-like genome, `-s3`.
Reads are sharded over ranks with the index replicated per GPU (weak scaling: per-GPU batch fixed).
The genome, its suffix array (built on the GPU by bk_build_sa_device) and the reads are generated
in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baseline(seq_t, sa_t, entries, reads_t, read_len, params_kw, gpu_hits, budget_s, ref_reads=0, cli_device=0):
    """Times the CPU oracle (restatement of the reference path, pthreads on every host core) on a
    bounded sample of the same workload; also checks the GPU results of that sample against it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import helpers
    cores = os.cpu_count() or 1
    t0 = time.time()
    seq = seq_t.cpu().numpy()
    sa = sa_t.cpu().numpy()
    log(f"cpu_baseline: index image copied to host in {time.time() - t0:.1f}s ({(seq.nbytes + sa.nbytes) / 1e9:.1f} GB)")
    ora = helpers.OracleSfx(seq=seq, sa=sa, el_size=4, entries=entries)
    p = helpers.make_params(**params_kw)
    n_avail = reads_t.numel() // read_len
    # probe run to size the sample for ~budget_s seconds of CPU work
    n0 = min(n_avail, 20000 * max(1, cores // 8))
    bases = reads_t[: n0 * read_len].cpu().numpy()
    offs = (np.arange(n0, dtype=np.uint64) * read_len)
    lens = np.full(n0, read_len, dtype=np.uint32)
    t0 = time.time()
    ora.align(bases, offs, lens, p, nthreads=cores)
    rate0 = n0 / max(1e-6, time.time() - t0)
    n1 = int(min(n_avail, max(n0, rate0 * budget_s)))
    bases = reads_t[: n1 * read_len].cpu().numpy()
    offs = (np.arange(n1, dtype=np.uint64) * read_len)
    lens = np.full(n1, read_len, dtype=np.uint32)
    t0 = time.time()
    exp, octr = ora.align(bases, offs, lens, p, nthreads=cores)
    dt = time.time() - t0
    got = gpu_hits[:n1]
    nbad = 0
    for f in ("chrom_id", "match_loci", "match_len", "low_hit_instances", "rslt", "nar", "strand", "low_mm",
              "nxt_low_mm", "num_hits", "mismatches"):
        nbad += int((got[f] != exp[f]).sum())
    ora.close()
    port = {"value": n1 / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"first {n1} reads of the rank-0 batch, oracle/bk_oracle.c ora_align_batch with {cores} pthreads, "
                      f"{dt:.1f} s wall; index already in host RAM",
            "parity_mismatching_fields_vs_gpu": nbad,
            "n_search_per_read": octr.n_search / n1, "n_cand_per_read": octr.n_cand / n1}
    if ref_reads <= 0:
        return port
    # the real reference on the same box, when its binary travelled with the repo
    ref = None
    named = [(f"chr{eid}", slen) for (eid, slen, _so, _eo) in entries]
    for n_ref in dict.fromkeys((int(min(n_avail, ref_reads)), int(min(n_avail, ref_reads, 1_000_000)))):
        try:
            ref = reference_baseline(seq, sa, named, reads_t[: n_ref * read_len].cpu().numpy(), read_len,
                                     params_kw.get("max_subs", 3), gpu_hits, n_ref, cli_device)
        except Exception as e:
            log(f"cpu_baseline(reference) failed: {e!r}")
            ref = None
        if ref and "exited with" not in str(ref.get("sample")):
            break
    if not ref or ref.get("value") is None:
        port["reference_leg"] = ref
        return port
    ref["port"] = port
    return ref


def write_sfx_file(path, seq, sa, entries):
    """`.sfx` as the reference writes it: header 1224 B pack(4), block header 20 B + bases + 4-byte suffix
    array, entries 8 + 111 B each (SfxArrayV2.h:79-104,174-187).  entries: [(name, seq_len)]."""
    import struct
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    n = len(seq)
    hdr = bytearray(1224)
    hdr[0:4] = b"sfx5"
    struct.pack_into("<iI", hdr, 4, 5, 0)
    blk_size = 20 + n + n * 4
    ent_size = 8 + 111 * len(entries)
    struct.pack_into("<QQIIQQ", hdr, 12, 1224 + blk_size + ent_size, 1224 + blk_size, ent_size, 1, blk_size, 1224)
    hdr[52:57] = b"synth"
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(struct.pack("<IIQI", 1, len(entries), n, 4))
        f.write(memoryview(np.ascontiguousarray(seq, dtype=np.uint8)))
        f.write(memoryview(np.ascontiguousarray(sa, dtype="<u4")))
        f.write(struct.pack("<II", len(entries), len(entries)))
        ofs = 0
        for i, (name, slen) in enumerate(entries):
            nm = name.encode()[:80]
            f.write(struct.pack("<II", i + 1, 1) + nm + b"\0" * (81 - len(nm)) +
                    struct.pack("<HIQQ", helpers.gen_hash16(name), slen, ofs, ofs + slen - 1))
            ofs += slen + 1


def write_fasta_file(path, reads_np, n_reads, read_len, chunk=4_000_000):
    """>r000000001\n<bases>\n per read (fixed-width names so the whole file is one array)."""
    import numpy as np
    lut = np.frombuffer(b"ACGTNNNN", dtype=np.uint8)
    with open(path, "wb") as f:
        for lo in range(0, n_reads, chunk):
            m = min(chunk, n_reads - lo)
            rec = np.empty((m, 12 + read_len + 1), dtype=np.uint8)
            rec[:, 0] = ord(">"); rec[:, 1] = ord("r"); rec[:, 11] = 10; rec[:, -1] = 10
            idx = np.arange(lo + 1, lo + m + 1, dtype=np.int64)
            for d in range(9):
                rec[:, 10 - d] = 48 + (idx // 10 ** d) % 10
            rec[:, 12:12 + read_len] = lut[reads_np[lo * read_len: (lo + m) * read_len].reshape(m, read_len) & 7]
            f.write(memoryview(rec))


REF_LADDER = (0, 32, 8)        # -T values tried in turn by the reference leg


def reference_baseline(seq, sa, entries, reads_np, read_len, max_subs, gpu_hits, n_sample, cli_device):
    """Times the REAL reference (oracle/_ref/biokanga, built from /root/reference by oracle/build_ref.sh and
    shipped as a binary) on the box's host cores on the first `n_sample` reads of the rank-0 batch against
    the same index, written out as a `.sfx`; checks its SAM records against the GPU results of the timed
    steps; then runs our own command line on the same two files and compares the SAM files byte for byte.
    Returns None when the reference binary is not present."""
    import datetime
    import re
    import shutil
    import struct
    import subprocess
    import tempfile
    import numpy as np
    import helpers
    import biokanga_amd as bk
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "biokanga")
    our_bin = os.path.join(ROOT, "biokanga_amd", "bin", "biokanga")
    if not os.path.exists(ref_bin):
        return None
    n = len(seq)
    need = 5 * n + 200 * n_sample + (1 << 28)
    base = "/dev/shm" if shutil.disk_usage("/dev/shm").free > 2 * need else None
    tmp = tempfile.mkdtemp(prefix="bk_ref_", dir=base)
    try:
        t0 = time.time()
        sfx, fa = os.path.join(tmp, "genome.sfx"), os.path.join(tmp, "reads.fa")
        write_sfx_file(sfx, seq, sa, entries)
        write_fasta_file(fa, reads_np, n_sample, read_len)
        t_files = time.time() - t0

        def run(binary, out, logf, extra):
            t = time.time()
            r = subprocess.run([binary, "align", "-i", fa, "-I", sfx, "-o", out, f"-s{max_subs}", "-M6", "-F", logf] + extra,
                               stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT, timeout=1500)
            return r.returncode, time.time() - t

        ref_sam, ref_log = os.path.join(tmp, "ref.sam"), os.path.join(tmp, "ref.log")

        def t_align_of(logf):
            # "Now aligning with minimum core size" -> "Alignment of N from N loaded completed".  The worker
            # threads start right after the first line and the main thread then sleeps a fixed 5 s
            # (Aligner.cpp:8794-8799) before joining them, so the span is the alignment time only when it
            # exceeds that sleep.
            stamp = {}
            for line in open(logf, errors="replace"):
                m = re.match(r"\[(\w+\s+\d+ \d+:\d+:\d+\.\d+ \d+)\]", line)
                if not m:
                    continue
                ts = datetime.datetime.strptime(re.sub(r"\s+", " ", m.group(1)), "%b %d %H:%M:%S.%f %Y").timestamp()
                if "Now aligning with minimum core size" in line:
                    stamp["start"] = ts
                elif "Alignment of" in line and "completed" in line:
                    stamp["end"] = ts
            return stamp["end"] - stamp["start"]

        # -T0 = every core up to cMaxWorkerThreads 128 (Aligner.h:15).  The sample cannot grow beyond ~1 M reads
        # (the reference's loader hand-off breaks when loading takes > 3 s, Aligner.cpp:4822), so when all
        # cores finish it inside the 5 s sleep the run is repeated with fewer threads until it is measurable.
        tried = []
        t_align = ref_wall = None
        threads = None
        for T in REF_LADDER:
            rc, wall = run(ref_bin, ref_sam, ref_log, [f"-T{T}"])
            if rc != 0:
                return {"value": None, "unit": "reads/s", "kind": "reference", "sample": f"reference exited with {rc} at -T{T}"}
            ta = t_align_of(ref_log)
            nthr = min(os.cpu_count() or 1, 128) if T == 0 else T
            tried.append(f"-T{T} ({nthr} threads): {ta:.2f} s")
            t_align, ref_wall, threads = ta, wall, nthr
            if ta > 6.0:
                break
        # reference SAM records vs the GPU hits of the same reads
        names = {i + 1: nm for i, (nm, _) in enumerate(entries)}
        got = gpu_hits[:n_sample]
        bad = seen = 0
        for line in open(ref_sam):
            if line[0] == "@":
                continue
            fld = line.rstrip("\n").split("\t")
            i = int(fld[0][1:]) - 1
            h = got[i]
            seen += 1
            if h["nar"] == 1:
                exp = (16 if h["strand"] == ord("-") else 0, names[int(h["chrom_id"])], int(h["match_loci"]) + 1)
                bad += (int(fld[1]), fld[2], int(fld[3])) != exp
            else:
                bad += not (fld[1] == "4" and fld[2] == "*" and fld[-1] == "YU:Z:" + bk.NAR_TAGS[int(h["nar"])])
        bad += abs(seen - n_sample)
        res = {"value": n_sample / t_align if t_align > 6.0 else None, "unit": "reads/s", "cores": threads, "kind": "reference",
               "sample": f"oracle/_ref/biokanga align -s{max_subs} -M6 on the first {n_sample} reads of the rank-0 batch vs the same "
                         f"{n / 1e9:.2f} Gbp index written as .sfx; T_align = log 'Now aligning' -> 'Alignment of .. completed' "
                         f"(valid above the reference's fixed 5 s start-up sleep); runs: {'; '.join(tried)}; whole process {ref_wall:.1f} s",
               "t_align_s": t_align, "t_e2e_s": ref_wall, "sam_records_differing_from_gpu": int(bad)}
        if os.path.exists(our_bin) and cli_device is not None:
            our_sam, our_log = os.path.join(tmp, "our.sam"), os.path.join(tmp, "our.log")
            rc, our_wall = run(our_bin, our_sam, our_log, ["--device", str(cli_device)])
            same = rc == 0 and subprocess.run(["cmp", "-s", ref_sam, our_sam]).returncode == 0
            try:
                our_t_align = t_align_of(our_log)      # same two log lines; no start-up sleep on our side
            except Exception:
                our_t_align = None
            res["our_cli"] = {"t_e2e_s": our_wall, "t_align_s": our_t_align, "rc": rc, "sam_byte_identical_to_reference": bool(same)}
        log(f"cpu_baseline(reference): files {t_files:.1f}s, reference {ref_wall:.1f}s (T_align {t_align:.2f}s)")
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def profiled_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same command
    (profiles/*_pmc_summary.csv: FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes, KiB units,
    calibrated in this path's access patterns - see profiles/README.md).  bench.py cannot collect
    counters itself, so this is the last profiled value for the default workload, or None."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_summary.csv")))
    if not files:
        return None, None
    members = {"k_search": ("k_search_a", "k_search_b"), "k_wave": ("k_wave",), "k_flat": ("k_flat",)}[kernel]
    tot = 0.0
    seen = set()
    for r in csv.DictReader(open(files[-1])):
        if r["counter"] in ("FETCH_SIZE", "WRITE_SIZE") and r["kernel"].split("<")[0] in members:
            tot += float(r["mean_per_dispatch"]) * 1024.0
            seen.add((r["kernel"].split("<")[0], r["counter"]))
    if len(seen) != 2 * len(members):
        return None, None
    return tot, os.path.basename(files[-1])


def host_resident_leg(al, rd_bases, rd_lens, d_out, args, barrier, all_reduce, dev):
    """The metric's real T_align (first batch submitted -> last result back) with reads and results in HOST
    memory: the same reads leave pinned host buffers in batches through bk_stream_submit, cross PCIe while the previous
    batch runs through the AlignReads phases, and every bk_hit record is back in host memory when the clock stops.
    Checked bit for bit against the records the kernel-only steps left in HBM."""
    import numpy as np
    import torch
    import biokanga_amd as bk
    n, L = args.reads, args.read_len
    t0 = time.time()
    h_bases = bk.host_array(n * L, np.uint8)
    h_lens = bk.host_array(n, np.uint32)
    torch.from_numpy(h_bases).copy_(rd_bases[: n * L])
    torch.from_numpy(h_lens.view(np.int32)).copy_(rd_lens[:n])
    h_out = [bk.host_array(n, bk.HIT_DTYPE) for _ in range(2)]
    expect = d_out.cpu().numpy().view(bk.HIT_DTYPE)
    B = max(1, min(args.stream_batch, n))
    cuts = list(range(0, n, B)) + [n]
    log(f"host-resident leg: pinned buffers ready in {time.time() - t0:.1f}s; {len(cuts) - 1} batches of <= {B} reads per step")
    result = {}
    with bk.Stream(al, B, B * L, depth=3) as st:
        def one_step(k):
            out = h_out[k & 1]
            return [st.submit(h_bases[lo * L: hi * L], None, h_lens[lo:hi], out[lo:hi]) for lo, hi in zip(cuts[:-1], cuts[1:])]
        for t in one_step(0):            # warm-up (buffers touched, scratch sized)
            st.wait(t)
        st.stats(reset=True)
        al.timing(reset=True)
        barrier()
        t_start = time.time()
        tickets = []
        for k in range(args.stream_steps):
            tickets += one_step(k)
            # keep at most one step of tickets un-waited so that the two result buffers are never overwritten early
            while len(tickets) > len(cuts) - 1:
                st.wait(tickets.pop(0))
        for t in tickets:
            st.wait(t)
        barrier()
        elapsed = time.time() - t_start
        stats = st.stats()
        tim = al.timing()
    if all_reduce is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        import torch.distributed as dist
        all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    same = all(bool(np.array_equal(h.view(np.uint8), expect.view(np.uint8))) for h in h_out[: min(2, args.stream_steps)])
    total = n * args.stream_steps * world
    result.update({"value": total / elapsed, "unit": "reads/s", "steps": args.stream_steps, "reads_per_step_per_gpu": n,
                   "batch_reads": B, "depth": 3, "seconds": elapsed,
                   "t_align_first_submit_to_last_result_s": stats["seconds_first_submit_to_last_result"],
                   "pcie_bytes_per_read": {"h2d": stats["bytes_h2d"] / max(1, stats["reads"]), "d2h": stats["bytes_d2h"] / max(1, stats["reads"])},
                   "device_ms_per_step": tim["ms_total"] / max(1, args.stream_steps),
                   "results_bit_identical_to_kernel_only_steps": same,
                   "note": "host pinned buffers in -> host bk_hit out through bk_stream_* (3 HIP streams); never the headline `value`"})
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=3100.0, help="synthetic genome size (Mbp); 3100 = GRCh38 scale")
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per step per GPU")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--max-subs", type=int, default=3, help="`-s` of biokanga align")
    ap.add_argument("--cpu-baseline-secs", type=float, default=15.0, help="0 disables the cpu_baseline leg")
    ap.add_argument("--reference-reads", type=int, default=2_000_000,
                    help="reads given to the real reference binary (oracle/_ref/biokanga) in the cpu_baseline leg; 0 = port only. "
                         "Kept at 2 M: the reference's loader hand-off breaks when loading takes > 3 s (Aligner.cpp:4822) - "
                         "3 M reads crash it on the MI355X host about every other run (tools/ref_scaling.py)")
    ap.add_argument("--stream-steps", type=int, default=5, help="steps of the host-resident leg (bk_stream_*: host buffers in -> host "
                                                                "bk_hit out, PCIe overlapped with the kernels); 0 disables it")
    ap.add_argument("--stream-batch", type=int, default=12_500_000, help="reads per submitted batch of the host-resident leg")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend; 'gloo' only for dry runs of the multi-rank path")
    ap.add_argument("--force-device", type=int, default=-1, help="dry runs: put every rank on this GPU instead of LOCAL_RANK")
    ap.add_argument("--kmer-bits", type=int, default=0, help="override the k of the k-mer interval table")
    ap.add_argument("--tune", action="append", default=[], help="name=value passed to bk_ctx_tune (repeatable)")
    ap.add_argument("--sweep", default="", help="name=v1,v2,..: re-time the steps for each value and log device ms")
    args = ap.parse_args()

    import numpy as np
    import torch
    import biokanga_amd as bk
    from biokanga_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.force_device >= 0:
        local_rank = args.force_device
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    def all_reduce(t, op=None):
        """in place; through host memory when the backend is not RCCL"""
        kw = {} if op is None else {"op": op}
        if args.dist_backend == "nccl":
            dist.all_reduce(t, **kw)
        else:
            h = t.cpu()
            dist.all_reduce(h, **kw)
            t.copy_(h)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- workload (untimed set-up)
    t0 = time.time()
    total_bp = int(args.genome_mbp * 1e6)
    seq, seq_lens = synth.make_genome(total_bp, dev, seed=38)
    n = seq.numel()
    torch.cuda.synchronize()
    log(f"genome: {n} concatenated bases, {len(seq_lens)} sequences, generated in {time.time() - t0:.1f}s")
    t0 = time.time()
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), 4, local_rank)
    torch.cuda.synchronize()
    log(f"suffix array built on the GPU in {time.time() - t0:.1f}s")
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    params_kw = dict(max_subs=args.max_subs)
    t0 = time.time()
    al = bk.Aligner(None, bk.AlignParams(**params_kw), device=local_rank, d_seq=seq.data_ptr(), concat_len=n,
                    d_sa=sa.data_ptr(), el_size=4, entries=ent)
    if args.kmer_bits:
        al.tune("kmer_bits", args.kmer_bits)
    for kv in args.tune:
        k, v = kv.split("=")
        al.tune(k, int(v))
    log(f"context (packed target + k-mer table) ready in {time.time() - t0:.1f}s; MinCoreLen {al.min_core_len}")
    t0 = time.time()
    rd_bases, rd_offs, rd_lens, truth = synth.make_reads(seq, seq_lens, args.reads, args.read_len, dev,
                                                         seed=1000 + rank, max_subs=args.max_subs)
    out = torch.zeros(args.reads * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    counts_dev = torch.zeros(len(entries), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    log(f"reads: {args.reads} x {args.read_len} bp generated in {time.time() - t0:.1f}s")
    if args.cpu_baseline_secs <= 0 or rank != 0 or world != 1:
        del seq, sa            # the context holds its own packed copy
        seq = sa = None
        torch.cuda.empty_cache()

    def step():
        al.align_device(rd_bases.data_ptr(), rd_offs.data_ptr(), rd_lens.data_ptr(), args.reads, out.data_ptr())
        # the path's one exchange step: per-sequence hit counts summed over ranks
        c = al.seq_counts(reset=True)
        counts_dev.copy_(torch.from_numpy(c.astype(np.int64)))
        if dist is not None:
            all_reduce(counts_dev)

    for _ in range(args.warmup):
        step()
    first_out = out.clone() if args.warmup > 0 else None       # results of an untimed step, to check repeatability
    if args.sweep:
        name, vals = args.sweep.split("=")
        for v in vals.split(","):
            al.tune(name, int(v))
            step()
            al.timing(reset=True)
            torch.cuda.synchronize()
            t1 = time.time()
            step()
            torch.cuda.synchronize()
            log(f"sweep {name}={v}: {1e3 * (time.time() - t1):.1f} ms/step wall; device {al.timing(reset=True)}")
    al.counters(reset=True)
    al.timing(reset=True)
    barrier()
    t_start = time.time()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.time() - t_start
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ctr = al.counters()
    tim = al.timing()
    host_leg = None
    if args.stream_steps > 0:
        try:
            host_leg = host_resident_leg(al, rd_bases, rd_lens, out, args, barrier, all_reduce if dist is not None else None, dev)
        except Exception as e:       # reporting only - never lose the measured line
            host_leg = {"value": None, "error": repr(e)}
    repeatable = bool(torch.equal(first_out, out)) if first_out is not None else None
    del first_out
    hits = out.cpu().numpy().view(bk.HIT_DTYPE)
    accepted = int((hits["nar"] == 1).sum())
    total_reads = args.reads * world * args.steps
    value = total_reads / elapsed

    # roofline: algorithmic bytes over the kernel's summed launch time, measured with HIP
    # events on the stream the kernels run on.  Per search: ceil(log2 N) * (E + 8); per candidate
    # Hamming-extended: E + ceil(L/2).  The dominant kernel (largest share of device time) is quoted.
    E = 4
    log2n = math.ceil(math.log2(n))
    per_search = log2n * (E + 8)
    per_cand = E + (args.read_len + 1) // 2
    io_bytes = args.reads * args.steps * ((args.read_len + 3) // 4 + 16)
    # (k_search = passes k_search_a + k_search_b of one phase, timed together; k_flat = the block-cooperative
    # extend kernel; k_wave = the wave-per-call kernel for repeat reads)
    kern = {
        "k_search": dict(bytes=ctr["n_search"] * per_search, ms=tim["ms_search"], launches=tim["n_search_launches"]),
        "k_wave": dict(bytes=ctr["n_cand_heavy"] * per_cand, ms=tim["ms_heavy"], launches=tim["n_heavy_launches"]),
        "k_flat": dict(bytes=(ctr["n_cand"] - ctr["n_cand_heavy"]) * per_cand, ms=tim["ms_extend"], launches=tim["n_extend_launches"]),
    }
    for k in kern.values():
        k["GBs"] = k["bytes"] / max(1e-9, k["ms"] * 1e-3) / 1e9
    dom = max(kern, key=lambda k: kern[k]["ms"])
    ach = kern[dom]["GBs"]
    whole = (sum(k["bytes"] for k in kern.values()) + io_bytes) / max(1e-9, tim["ms_total"] * 1e-3) / 1e9
    default_workload = (args.reads == 50_000_000 and args.read_len == 100 and args.max_subs == 3 and total_bp == 3_100_000_000)
    traffic, traffic_src = profiled_traffic(dom) if default_workload else (None, None)
    roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": kern[dom]["bytes"] / max(1, kern[dom]["launches"]),
                "avg_launch_ms": kern[dom]["ms"] / max(1, kern[dom]["launches"]),
                "per_kernel": {k: {"algorithmic_GBs": round(v["GBs"], 1), "ms": round(v["ms"], 2), "launches": v["launches"]}
                               for k, v in kern.items()},
                "whole_step_algorithmic_GBs": whole, "whole_step_frac": whole / HBM_PEAK_GBS,
                "device_ms": {k: tim[k] for k in ("ms_total", "ms_search", "ms_extend", "ms_heavy", "ms_other")}}

    result = {
        "metric": "aligned reads/s (SAM-identical) on 100 bp SE vs GRCh38, 1->8 MI355X",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"C2: {args.reads} x {args.read_len} bp SE reads per GPU per step (0-{args.max_subs} subs, "
                               f"simreads-like) vs synthetic GRCh38-like genome of {total_bp} bp in {len(seq_lens)} "
                               f"sequences (45% repeat-derived, N gaps), biokanga align -s{args.max_subs}",
                   "reads_per_gpu_per_step": args.reads, "read_len": args.read_len, "genome_bp": total_bp,
                   "concat_len": n, "sfx_el_size": E, "index": "replicated per GPU, built on device",
                   "parallelism": f"reads sharded over {world} GPU(s)", "accepted_frac_rank0": accepted / args.reads,
                   "n_search_per_read": ctr["n_search"] / (args.reads * args.steps),
                   "n_cand_per_read": ctr["n_cand"] / (args.reads * args.steps),
                   "heavy_calls_frac": ctr["n_heavy"] / max(1, ctr["n_lcm_calls"]),
                   "results_bitwise_equal_across_steps": repeatable},
        "roofline": roofline,
        "t_align_host_resident": host_leg,
    }
    if rank == 0 and world == 1 and args.cpu_baseline_secs > 0:
        try:
            result["cpu_baseline"] = cpu_baseline(seq, sa, entries, rd_bases, args.read_len, params_kw, hits,
                                                  args.cpu_baseline_secs, args.reference_reads, local_rank)
        except Exception as e:      # the baseline is reporting only - never lose the measured line
            result["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": os.cpu_count(), "kind": "port",
                                      "sample": f"failed: {e!r}"}
    al.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
