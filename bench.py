#!/usr/bin/env python3
"""bench.py - headline benchmark of the `biokanga align` hot path on MI355X.

Metric (BASELINE.json): aligned reads/s, results identical to the reference, 100 bp SE reads vs a GRCh38-scale index.
One "step" = one pass of the hot path over one batch of synthetic reads: pack, SA-interval search, candidate walk + Hamming
extension, classification - every AlignReads phase - followed by the path's only exchange step, the sum-reduction of the
per-sequence hit counts (RCCL all-reduce when N > 1).

`value` is the metric's own clock (T_align: first batch submitted -> last result back): the reads of every step leave
pinned HOST buffers (2 bit/base, bk_stream_submit_packed), cross PCIe while the batch before them runs through the phases, and every
bk_hit record is back in HOST memory when the clock stops.  `value_kernel_only` is the same steps with reads and results resident in
HBM (bk_align_batch_device).  Both are measured in the index layout `biokanga align` itself would pick for the configuration's read
count (the 149 GB suffix-ordered window array only from 600 M reads per device on, host/biokanga_main.cpp) and, beside it, in the
other layout (`layouts`).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N            (no WORLD_SIZE in the environment: starts the N ranks itself, same launcher)
    python bench.py --config C3         (2 x 150 bp paired ends, -s5 -U3 -d200 -D400: SE pass + PE association per step)

Workload at N = 1: BASELINE.json configs[1] restated over a synthetic genome (SURVEY.md 8d "C2"):
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
# The index image and layout of the headline follow the library's one rule, bk_image_policy(reads a device aligns in the configuration) - the
# function `biokanga align` calls (host/biokanga_main.cpp, cmd_align): from BK_POLICY_MIN_READS reads per device on every table (k-mer table
# entries of two words, third- and fourth-level search keys) and the partial suffix-ordered window array come with the index, made behind
# the suffix array's upload; below, the lean image without the array.  --index-image / --window-array force either half.


def effective_cpus():
    """CPUs this process can keep busy: hardware threads it may run on, cut down to the cgroup CPU quota (the GPU boxes of
    this pool expose 256 hardware threads under a 16-CPU quota - more runnable threads than that only buy throttling)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except Exception:
            pass
    return n


_REAL_STDOUT = None       # the process's own stdout, kept aside by main() (see there)


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this process has not
    touched a GPU yet) through the same torch.distributed.run command line the driver uses, and hand their exit code back."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] no WORLD_SIZE in the environment: launching", " ".join(cmd), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baseline(seq_t, sa_t, entries, reads_t, read_len, params_kw, gpu_hits, budget_s, ref_reads=0, cli_device=0, pe=None,
                 full_cli=True):
    """Times the CPU oracle (restatement of the reference path, pthreads on every host core) on a
    bounded sample of the same workload; also checks the GPU results of that sample against it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import helpers
    hw_threads = os.cpu_count() or 1
    cores = effective_cpus()
    t0 = time.time()
    seq = seq_t.cpu().numpy()
    sa = sa_t.cpu().numpy()
    log(f"cpu_baseline: index image copied to host in {time.time() - t0:.1f}s ({(seq.nbytes + sa.nbytes) / 1e9:.1f} GB)")
    ora = helpers.OracleSfx(seq=seq, sa=sa, el_size=5 if sa.dtype == np.uint8 else 4, entries=entries)
    p = helpers.make_params(**params_kw)
    n_avail = reads_t.numel() // read_len
    # probe run to size the sample for ~budget_s seconds of CPU work
    n0 = min(n_avail, 20000 * max(1, cores // 8))
    bases = reads_t[: n0 * read_len].cpu().numpy()
    offs = (np.arange(n0, dtype=np.uint64) * read_len)
    lens = np.full(n0, read_len, dtype=np.uint32)
    t0 = time.time()
    ora.align(bases, offs, lens, p, nthreads=hw_threads)
    rate0 = n0 / max(1e-6, time.time() - t0)
    n1 = int(min(n_avail, max(n0, rate0 * budget_s)))
    n1 -= n1 & 1
    bases = reads_t[: n1 * read_len].cpu().numpy()
    offs = (np.arange(n1, dtype=np.uint64) * read_len)
    lens = np.full(n1, read_len, dtype=np.uint32)
    t0 = time.time()
    exp, octr = ora.align(bases, offs, lens, p, nthreads=hw_threads)
    if pe:          # CAligner::ProcessPairedEnds of the restatement (single-threaded there, as in the reference)
        helpers.oracle_process_pe(ora, p, pe["pe_mode"], pe["pair_min_len"], pe["pair_max_len"], False, bases, offs, lens, exp)
    dt = time.time() - t0
    got = gpu_hits[:n1]
    nbad = 0
    for f in ("chrom_id", "match_loci", "match_len", "low_hit_instances", "nar", "strand", "low_mm",
              "nxt_low_mm", "num_hits", "mismatches") + (() if pe else ("rslt",)):
        nbad += int((got[f] != exp[f]).sum())
    if pe:
        nbad += int(((got["flags"] & 0x80) != (exp["flags"] & 0x80)).sum())
    ora.close()
    port = {"value": n1 / dt, "unit": "reads/s", "cores": cores, "hardware_threads": hw_threads, "kind": "port",
            "sample": f"first {n1} reads of the rank-0 batch, oracle/bk_oracle.c ora_align_batch with {hw_threads} pthreads under a "
                      f"cgroup quota of {cores} CPUs, {dt:.1f} s wall; index already in host RAM",
            "parity_mismatching_fields_vs_gpu": nbad,
            "n_search_per_read": octr.n_search / n1, "n_cand_per_read": octr.n_cand / n1}
    if ref_reads <= 0:
        return port
    # the real reference on the same box, when its binary travelled with the repo
    ref = None
    named = [(f"chr{eid}", slen) for (eid, slen, _so, _eo) in entries]
    all_np = reads_t.cpu().numpy() if full_cli else None
    for n_ref in dict.fromkeys((int(min(n_avail, ref_reads)), int(min(n_avail, ref_reads, 1_000_000)))):
        try:
            ref = reference_baseline(seq, sa, named, reads_t[: n_ref * read_len].cpu().numpy(), read_len,
                                     params_kw.get("max_subs", 3), gpu_hits, n_ref, cli_device, pe=pe,
                                     full_reads=n_avail if full_cli else 0, all_reads_np=all_np)
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


def write_fasta_file(path, reads_np, n_reads, read_len, chunk=4_000_000, start=0, step=1):
    """>r000000001\n<bases>\n per read (fixed-width names = 1 + the read's index in `reads_np`, so the whole file is one array);
    start / step pick every step-th read from `start` on (the two files of a paired-end set)."""
    import numpy as np
    lut = np.frombuffer(b"ACGTNNNN", dtype=np.uint8)
    rows = reads_np[: n_reads * read_len].reshape(n_reads, read_len)[start::step]
    total = len(rows)
    with open(path, "wb") as f:
        for lo in range(0, total, chunk):
            m = min(chunk, total - lo)
            rec = np.empty((m, 12 + read_len + 1), dtype=np.uint8)
            rec[:, 0] = ord(">"); rec[:, 1] = ord("r"); rec[:, 11] = 10; rec[:, -1] = 10
            idx = (np.arange(lo, lo + m, dtype=np.int64) * step) + start + 1
            for d in range(9):
                rec[:, 10 - d] = 48 + (idx // 10 ** d) % 10
            rec[:, 12:12 + read_len] = lut[rows[lo:lo + m] & 7]
            f.write(memoryview(rec))


REF_LADDER = (0, 32, 8)        # -T values tried in turn by the reference leg


def reference_baseline(seq, sa, entries, reads_np, read_len, max_subs, gpu_hits, n_sample, cli_device, pe=None, full_reads=0,
                       all_reads_np=None):
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
        sfx, fa, fa2 = os.path.join(tmp, "genome.sfx"), os.path.join(tmp, "reads.fa"), os.path.join(tmp, "reads_2.fa")
        write_sfx_file(sfx, seq, sa, entries)
        if pe:          # interleaved PE1, PE2 -> two files; names stay 1 + the read's index
            write_fasta_file(fa, reads_np, n_sample, read_len, start=0, step=2)
            write_fasta_file(fa2, reads_np, n_sample, read_len, start=1, step=2)
            inputs = ["-i", fa, "-u", fa2, f"-U{pe['pe_mode']}", f"-d{pe['pair_min_len']}", f"-D{pe['pair_max_len']}"]
        else:
            write_fasta_file(fa, reads_np, n_sample, read_len)
            inputs = ["-i", fa]
        t_files = time.time() - t0

        def run(binary, out, logf, extra, inp=None):
            t = time.time()
            r = subprocess.run([binary, "align"] + (inp or inputs) + ["-I", sfx, "-o", out, f"-s{max_subs}", "-M6", "-F", logf] + extra,
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
            nthr = min(os.cpu_count() or 1, 128) if T == 0 else T          # its worker threads; the box's CPU quota is stated beside them
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
            if pe:
                flag, pos, rnext, pnext, tlen = helpers.expected_pe_sam_fields(got, i)
                rname = names[int(h["chrom_id"])] if h["nar"] == 1 else "*"
                bad += (int(fld[1]), fld[2], int(fld[3]), fld[6], int(fld[7]), int(fld[8])) != (flag, rname, pos, rnext, pnext, tlen)
                bad += h["nar"] != 1 and fld[-1] != "YU:Z:" + bk.NAR_TAGS[int(h["nar"])]
            elif h["nar"] == 1:
                exp = (16 if h["strand"] == ord("-") else 0, names[int(h["chrom_id"])], int(h["match_loci"]) + 1)
                bad += (int(fld[1]), fld[2], int(fld[3])) != exp
            else:
                bad += not (fld[1] == "4" and fld[2] == "*" and fld[-1] == "YU:Z:" + bk.NAR_TAGS[int(h["nar"])])
        bad += abs(seen - n_sample)
        res = {"value": n_sample / t_align if t_align > 6.0 else None, "unit": "reads/s", "cores": min(threads, effective_cpus()),
               "threads": threads, "kind": "reference", "value_is_a_lower_bound": True,
               "lower_bound_because": "T_align is read off the reference's log and sits only a few seconds above its fixed 5 s start-up "
                                      "sleep; the sample cannot grow (its loader hand-off breaks beyond ~3 s of loading, Aligner.cpp:4822)",
               "sample": f"oracle/_ref/biokanga align {' '.join(inputs[2:]) + ' ' if pe else ''}-s{max_subs} -M6 on the first {n_sample} reads of the rank-0 batch vs the same "
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
            if full_reads > n_sample and all_reads_np is not None:
                # T_e2e of our command line on the WHOLE step's reads (process start -> exit, files in the same tmp dir)
                t0 = time.time()
                fa_full, fa_full2 = os.path.join(tmp, "all.fa"), os.path.join(tmp, "all_2.fa")
                if pe:
                    write_fasta_file(fa_full, all_reads_np, full_reads, read_len, start=0, step=2)
                    write_fasta_file(fa_full2, all_reads_np, full_reads, read_len, start=1, step=2)
                    inp = ["-i", fa_full, "-u", fa_full2] + inputs[4:]
                else:
                    write_fasta_file(fa_full, all_reads_np, full_reads, read_len)
                    inp = ["-i", fa_full]
                t_w = time.time() - t0
                full_sam, full_log = os.path.join(tmp, "full.sam"), os.path.join(tmp, "full.log")
                rc, wall = run(our_bin, full_sam, full_log, ["--device", str(cli_device)], inp)
                try:
                    ta = t_align_of(full_log)
                except Exception:
                    ta = None
                res["our_cli_full"] = {"reads": full_reads, "t_e2e_s": wall, "reads_per_s_e2e": full_reads / wall if rc == 0 else None,
                                       "t_align_s": ta, "rc": rc, "sam_bytes": os.path.getsize(full_sam) if rc == 0 else None,
                                       "note": f"biokanga align -M6 on every read of one step, reads + index + SAM in {os.path.dirname(tmp) or '/tmp'}; "
                                               f"FASTA written in {t_w:.0f} s (untimed)"}
        log(f"cpu_baseline(reference): files {t_files:.1f}s, reference {ref_wall:.1f}s (T_align {t_align:.2f}s)")
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


SWIN_TUNE = {"off": 0, "partial": 1, "forced": 2, "full": 3}       # bk_ctx_tune("use_swin", ..) per index layout


def fetch_correction(kernel, window_array):
    """FETCH_SIZE calibration for `kernel`'s access pattern (MI355X_MICROARCH.md: the counter reports HALF of a wide coalesced
    16 B/lane streaming read).  Our patterns were measured with tools/rand_access_bench under `rocprofv3 --pmc FETCH_SIZE`
    (profiles/*_fetch_calibration.csv): random 8-byte loads count 64.0 B each and random 80-byte windows 96 B (no correction) - what
    the search passes, k_flat and k_wave WITHOUT the window array do; runs of 64 consecutive 48-byte window-array entries plus their
    coalesced suffix array elements - k_wave WITH the window array - count about half.  Returns (factor on FETCH_SIZE, source)."""
    if not (kernel == "k_wave" and window_array in ("partial", "full", True)):
        return 1.0, "random-line pattern: FETCH_SIZE counts 64.0 B per random 8-byte load (calibrated, no correction)"
    import csv
    import glob
    import re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_fetch_calibration.csv"))):
        rows, counts, sizes = {}, {}, {}
        for line in open(f):
            if line.startswith("calib:"):
                for name, n in re.findall(r"(k_calib_\w+): (\d+) entries", line):
                    counts[name] = int(n)
                for name, b in re.findall(r"(k_calib_\w+): \d+ entries of (\d+) B", line):
                    sizes[name] = int(b)
            elif line.startswith("k_calib"):
                r = next(csv.reader([line]))
                rows[r[0]] = float(r[3]) * 1024.0                        # FETCH_SIZE is in KiB
        for name, per_entry in (("k_calib_wave", 40), ("k_calib_runs1", 48)):
            per_entry = sizes.get(name, per_entry)                       # (what the file itself says an entry moved: the layout changed in round 5)
            if name in rows and name in counts:
                best = (counts[name] * per_entry / rows[name], f"{os.path.basename(f)}: {name}, {counts[name]} entries of {per_entry} B "
                        f"read as {rows[name] / counts[name]:.1f} B each")
                break
    return best if best else (2.0, "MI355X_MICROARCH.md: wide coalesced 16 B/lane reads count half (no calibration file found)")


def live_traffic(kernel, extra_args, window_array, budget_s=240):
    """HBM bytes of `kernel` in ONE step of this workload, OBSERVED in this run: two child processes repeat one step under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, counters only - MI355X_MICROARCH.md's
    recipe; KiB units).  Children, not an exec: this process has initialised the GPU.
    Returns ((fetch bytes as counted, write bytes, launches of the kernel in that step), description) or (None, reason)."""
    import csv
    import re
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled (no nested profiler)"
    got = {}
    launches = None
    tmp = tempfile.mkdtemp(prefix="bk_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "run", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--window-array", window_array] + extra_args
            t0 = time.time()
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=budget_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} child exited with {r.returncode}"
            files = [os.path.join(dp, f) for dp, _dn, fn in os.walk(d) for f in fn if f.endswith("_counter_collection.csv")]
            if not files:
                return None, "no counter_collection.csv written"
            tot, n = 0.0, 0
            seen = set()
            for f in files:
                for row in csv.DictReader(open(f)):
                    m = re.search(r"bk::(k_\w+)", row["Kernel_Name"])
                    key = (row["Process_Id"], row["Dispatch_Id"])
                    if m and m.group(1) == kernel and row["Counter_Name"] == counter and key not in seen:
                        seen.add(key)
                        tot += float(row["Counter_Value"])
                        n += 1
            log(f"live traffic: {counter}: {kernel}: {n} launches, {tot * 1024 / 1e9:.1f} GB as counted ({time.time() - t0:.0f}s)")
            if n == 0:
                return None, f"kernel {kernel} not in the {counter} pass"
            got[counter] = tot * 1024.0
            launches = n
        return (got["FETCH_SIZE"], got["WRITE_SIZE"], launches), \
            "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two child runs of ONE step of this workload; bytes of that step divided by this run's launches per step"
    except Exception as e:
        return None, f"live counter passes failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def profiled_traffic(kernel):
    """(fetch bytes as counted, write bytes) per launch of `kernel` from the last committed rocprofv3 PMC summary of this command
    (profiles/*_pmc_summary.csv: FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes, KiB units), or None."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.csv")))
    if not files:
        return None, None
    got = {}
    for r in csv.DictReader(open(files[-1])):
        if r["counter"] in ("FETCH_SIZE", "WRITE_SIZE") and r["kernel"].split("<")[0] == kernel:
            got[r["counter"]] = got.get(r["counter"], 0.0) + float(r["mean_per_dispatch"]) * 1024.0
    if len(got) != 2:
        return None, None
    return (got["FETCH_SIZE"], got["WRITE_SIZE"]), os.path.basename(files[-1])


class HostLeg:
    """The metric's own clock (SURVEY.md 8d T_align: first batch submitted -> last result back) with reads and results in HOST
    memory: the same reads leave pinned host buffers in batches through bk_stream_submit[_packed], cross PCIe while the previous
    batch runs through the AlignReads phases, and every bk_hit record is back in host memory when the clock stops."""

    def __init__(self, rd_bases, rd_lens, args):
        import numpy as np
        import torch
        import biokanga_amd as bk
        self.args = args
        n, L = args.reads, args.read_len
        t0 = time.time()
        self.h_lens = bk.host_array(n, np.uint32)
        torch.from_numpy(self.h_lens.view(np.int32)).copy_(rd_lens[:n])
        # (three sets of result records: a step's batches go in while the step before is aligned and the one before that comes back - with
        # two, a step's upload could only start once the step two back had come home, and a 2 x 150 step's 1.8 GB did not make it in time)
        self.n_sets = max(2, int(args.host_sets))
        self.h_out = [bk.host_array(n, bk.HIT_DTYPE) for _ in range(self.n_sets)]
        self.packed = args.stream_form == "packed"
        if self.packed:
            # the loader's side of the boundary: reads leave host memory at 2 bit/base (bk_pack_reads: host threads, untimed set-up here
            # as parsing is), 16-bit lengths, and the list of the bases that are not a,c,g,t
            h_bases = rd_bases[: n * L].cpu().numpy()
            self.p_words, self.p_lens16, self.p_exc = bk.pack_reads(h_bases, None, self.h_lens, pinned=True)
            del h_bases
            self.wpr = (L + 15) // 16
            log(f"host leg: {n} reads packed to {self.p_words.nbytes / n:.1f} + 2 B/read (+ {len(self.p_exc)} non-acgt bases)")
        else:
            self.h_bases = bk.host_array(n * L, np.uint8)
            torch.from_numpy(self.h_bases).copy_(rd_bases[: n * L])
        B = max(2, min(args.stream_batch or n, n))
        B -= B & 1
        self.B = B
        cuts = list(range(0, n, B)) + [n]

        # The first step grows its batches (6 %, 28 %, 66 % of a batch): the pipeline has nothing to overlap the very first upload with,
        # so the smaller it is the sooner the kernels start, and a batch crosses PCIe about 2.7 times faster than it is aligned (0.62 ns
        # against 1.65 ns per read, plus about 4 ms a batch), so each upload still hides behind the batch before it.  Small batches cost
        # more per read (every phase's wave-per-read launch lasts at least as long as its heaviest read), hence no finer ramp than that.
        def even(v):
            return max(2, int(v) & ~1)
        first = min(B, n)
        ramp = ([0] + [even(float(x) * first) for x in args.host_ramp.split(",") if x]) if first >= 1000 else [0]
        self.cuts = cuts
        self.cuts0 = ramp + [c for c in cuts if c > ramp[-1]]
        # .. and the last step ends on a tenth of a batch: the download of the very last batch has nothing to hide behind either
        down = [n - even(0.1 * first)] if first >= 1000 else []
        self.cuts9 = ([c for c in cuts if c < down[0]] + down + [n]) if down else cuts
        self.exc_of = {}
        if self.packed:
            er = self.p_exc["read"]
            for lo, hi in set(zip(self.cuts0[:-1], self.cuts0[1:])) | set(zip(cuts[:-1], cuts[1:])) | set(zip(self.cuts9[:-1], self.cuts9[1:])):
                a, z = np.searchsorted(er, lo), np.searchsorted(er, hi)
                e = bk.host_array(max(1, z - a), bk.NBASE_DTYPE)[: z - a]
                e[:] = self.p_exc[a:z]
                e["read"] -= lo                                      # exception read numbers are batch-relative
                self.exc_of[(lo, hi)] = e
        log(f"host leg: pinned buffers ready in {time.time() - t0:.1f}s; {len(cuts) - 1} batch(es) of <= {B} reads per step "
            f"(first step: batches of {[b - a for a, b in zip(self.cuts0[:-1], self.cuts0[1:])]}; last step: {[b - a for a, b in zip(self.cuts9[:-1], self.cuts9[1:])]})")

    def run(self, al, warmup, steps, barrier, all_reduce, dev, pe_params=None):
        """`warmup` untimed steps, then exactly `steps` steps between two barriers; returns the leg's record"""
        import torch
        import biokanga_amd as bk
        n, L = self.args.reads, self.args.read_len
        with bk.Stream(al, self.B, self.B * L, depth=3, pe=pe_params) as st:
            def one_step(k, first=False, last=False):
                out = self.h_out[k % self.n_sets]
                cc = self.cuts0 if first else (self.cuts9 if last else self.cuts)
                if self.packed:
                    return [st.submit_packed(self.p_words[lo * self.wpr: hi * self.wpr], self.p_lens16[lo:hi], self.exc_of[(lo, hi)], out[lo:hi])
                            for lo, hi in zip(cc[:-1], cc[1:])]
                return [st.submit(self.h_bases[lo * L: hi * L], None, self.h_lens[lo:hi], out[lo:hi]) for lo, hi in zip(cc[:-1], cc[1:])]
            for w in range(warmup):          # (buffers touched, scratch sized)
                for t in one_step(w):
                    st.wait(t)
            st.stats(reset=True)
            al.timing(reset=True)
            barrier()
            t_start = time.time()
            tickets = []
            for k in range(steps):
                tickets += one_step(k, first=(k == 0), last=(k == steps - 1 and k > 0))
                # keep at most two steps of tickets un-waited so that a set of result records is never overwritten early
                while len(tickets) > (self.n_sets - 1) * (len(self.cuts) - 1):
                    st.wait(tickets.pop(0))
            for t in tickets:
                st.wait(t)
            barrier()
            elapsed = time.time() - t_start
            stats = st.stats()
            tim = al.timing(reset=True)
        if all_reduce is not None:
            import torch.distributed as dist
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        world = int(os.environ.get("WORLD_SIZE", "1"))
        return {"value": n * steps * world / elapsed, "unit": "reads/s", "steps": steps, "warmup": warmup, "reads_per_step_per_gpu": n,
                "batch_reads": self.B, "depth": 3, "seconds": elapsed, "ms_per_step": elapsed / steps * 1e3,
                "form": "packed: 2 bit/base words + 16-bit lengths + non-acgt list (bk_stream_submit_packed)" if self.packed else "1 byte/base (bk_stream_submit)",
                "t_align_first_submit_to_last_result_s": stats["seconds_first_submit_to_last_result"],
                "pcie_bytes_per_read": {"h2d": stats["bytes_h2d"] / max(1, stats["reads"]), "d2h": stats["bytes_d2h"] / max(1, stats["reads"])},
                "device_ms_per_step": tim["ms_total"] / max(1, steps),
                "device_ms_per_step_by_stage": {k: round(v / max(1, steps), 2) for k, v in tim.items() if k.startswith("ms_") and k != "ms_total"},
                "note": "host pinned buffers in -> host bk_hit out through bk_stream_* (3 HIP streams)"}

    def same_as(self, expect, steps):
        import numpy as np
        return all(bool(np.array_equal(h.view(np.uint8), expect.view(np.uint8))) for h in self.h_out[: min(self.n_sets, steps)])


def metric_text(args, cfg, E):
    """BASELINE.json's metric, worded for the configuration that was actually run (C2 is the headline's own)"""
    shape = f"2x{args.read_len} bp PE" if cfg["pe"] else f"{args.read_len} bp SE"
    genome = "GRCh38" if E == 4 else "17 Gbp wheat-like genome (5-byte .sfx)"
    return f"aligned reads/s (SAM-identical) on {shape} vs {genome}, 1->8 MI355X"


CONFIGS = {
    # the synthetic restatements of BASELINE.json's configs that fit one GPU
    # job_reads_per_gpu: the reads ONE GPU aligns in the BASELINE.json configuration this restates - what `biokanga align`'s window-array
    # policy looks at (a step of the bench is a slice of that job)
    "C2": dict(read_len=100, max_subs=3, reads=50_000_000, pe=None, job_reads_per_gpu=50_000_000,
               text="{reads} x {read_len} bp SE reads per GPU per step (0-{max_subs} subs, simreads-like)", cli="biokanga align -s{max_subs}"),
    "C3": dict(read_len=150, max_subs=5, reads=40_000_000, pe=dict(pe_mode=3, pair_min_len=200, pair_max_len=400), job_reads_per_gpu=400_000_000,
               text="{pairs} x 2x{read_len} bp FR pairs per GPU per step = {reads} reads (insert ~N(300,50) in [200,400], 0-5 subs per read)",
               cli="biokanga align -s{max_subs} -U3 -d200 -D400"),
    # BASELINE.json configuration 5 restated for ONE GPU: the 17 Gbp index (5-byte suffix elements, built on the device) fits a 288 GB
    # MI355X, so the index is replicated and not partitioned; reads per step scaled to what is left of the HBM
    "C5": dict(read_len=150, max_subs=5, reads=20_000_000, pe=dict(pe_mode=3, pair_min_len=200, pair_max_len=400), job_reads_per_gpu=125_000_000,
               genome_mbp=17000.0, n_seqs=21, repeat_frac=0.85, seed=17, no_reference=True,
               text="{pairs} x 2x{read_len} bp FR pairs per GPU per step = {reads} reads (insert ~N(300,50) in [200,400], 0-5 subs per read), wheat-like",
               cli="biokanga align -s{max_subs} -U3 -d200 -D400"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS), help="C2 = BASELINE.json's headline workload (default); C3 = 2x150 bp paired ends; "
                                                                             "C5 = the same pairs against a 17 Gbp index with 5-byte suffix elements")
    ap.add_argument("--genome-mbp", type=float, default=0.0, help="synthetic genome size (Mbp); 0 = the config's own (3100 = GRCh38 scale; C5: 17000)")
    ap.add_argument("--reads", type=int, default=0, help="reads per step per GPU (0 = the config's own: 50 M for C2, 40 M = 20 M pairs for C3)")
    ap.add_argument("--read-len", type=int, default=0)
    ap.add_argument("--max-subs", type=int, default=-1, help="`-s` of biokanga align")
    ap.add_argument("--cpu-baseline-secs", type=float, default=15.0, help="0 disables the cpu_baseline leg")
    ap.add_argument("--reference-reads", type=int, default=2_000_000,
                    help="reads given to the real reference binary (oracle/_ref/biokanga) in the cpu_baseline leg; 0 = port only. "
                         "Kept at 2 M: the reference's loader hand-off breaks when loading takes > 3 s (Aligner.cpp:4822) - "
                         "3 M reads crash it on the MI355X host about every other run (profiles/r01_host_cpu_scaling.md)")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) one step only, nothing reported: what the live counter passes profile")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed profile instead of two live rocprofv3 --pmc passes")
    ap.add_argument("--no-full-cli", action="store_true", help="skip the T_e2e run of our command line on a whole step's reads")
    ap.add_argument("--window-array", default="policy", choices=["policy", "partial", "on", "off", "full", "forced"],
                    help="index layout of the headline: 'policy' = what `biokanga align` picks for this many reads per device; 'partial' (= 'on') = the "
                         "suffix-ordered window array for the part of the suffix array the wave kernel's long walks visit; 'full' = for every suffix "
                         "(149 GB at 3.1 Gbp); the other layouts are measured beside it")
    ap.add_argument("--index-image", default="policy", choices=["policy", "full", "lean"],
                    help="index image of the headline: 'policy' = what bk_image_policy gives this many reads per device; 'full' = every table; "
                         "'lean' = no second words in the k-mer table, no third- / fourth-level search keys; the other image is measured beside it")
    ap.add_argument("--no-other-layout", action="store_true", help="measure the headline's index layout and image only")
    ap.add_argument("--other-layouts", default="off,partial,full", help="which of off / partial / full to measure beside the headline's")
    ap.add_argument("--no-host-leg", action="store_true", help="kernel-only steps only (profiling runs): `value` is then the kernel-only rate and says so")
    ap.add_argument("--stream-form", default="packed", choices=["packed", "bytes"], help="form in which the reads cross PCIe: "
                    "2 bit/base (bk_stream_submit_packed) or 1 byte/base (bk_stream_submit)")
    ap.add_argument("--host-ramp", default="0.06,0.34", help="the first host-in / host-out step's batches end at these fractions of a step (then the rest)")
    ap.add_argument("--host-sets", type=int, default=3, help="sets of host result records of the host-in / host-out steps (steps in flight + 1)")
    ap.add_argument("--stream-batch", type=int, default=0, help="reads per submitted batch of the host-in / host-out steps (0 = a whole step; the first step ramps up to it)")
    ap.add_argument("--shard-check-reads", type=int, default=8_000_000, help="N > 1: size of the ONE read set that is sharded i mod N and "
                                                                              "whose reduced counts are compared with a 1-GPU run of all of it")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend; 'gloo' only for dry runs of the multi-rank path")
    ap.add_argument("--no-rccl-world1", action="store_true", help="N = 1: no torch.distributed process group of one rank (default: RCCL is initialised and the "
                                                                   "per-step all-reduce of the sequence counts runs through it, as at N > 1)")
    ap.add_argument("--force-device", type=int, default=-1, help="dry runs: put every rank on this GPU instead of LOCAL_RANK")
    ap.add_argument("--force-el5", action="store_true", help="experiments: 5-byte suffix elements (the > 4 Gbp kernels) on a smaller genome")
    ap.add_argument("--kmer-bits", type=int, default=0, help="override the k of the k-mer interval table")
    ap.add_argument("--tune", action="append", default=[], help="name=value passed to bk_ctx_tune (repeatable)")
    ap.add_argument("--sweep", default="", help="name=v1,v2,..: re-time the kernel-only steps for each value and log device ms")
    args = ap.parse_args()
    if args.pmc_child:
        args.steps, args.warmup, args.cpu_baseline_secs, args.no_live_traffic, args.no_host_leg, args.no_other_layout = 1, 0, 0.0, True, True, True
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))          # before anything here touches a GPU
    # ONE JSON line on stdout is the contract; libraries underneath (RCCL prints its version banner there when a communicator comes up) write
    # to file descriptor 1 behind Python's back - so descriptor 1 is pointed at stderr for the whole run and the line goes out through a
    # copy of the real one at the end
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    cfg = CONFIGS[args.config]
    args.reads = args.reads or cfg["reads"]
    args.genome_mbp = args.genome_mbp or cfg.get("genome_mbp", 3100.0)
    if cfg.get("no_reference"):
        args.reference_reads = 0           # a 102 GB .sfx for the reference binary is not written; the C restatement is the baseline
    args.read_len = args.read_len or cfg["read_len"]
    args.max_subs = cfg["max_subs"] if args.max_subs < 0 else args.max_subs
    pe = cfg["pe"]
    if pe:
        args.reads -= args.reads & 1

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
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s) - refusing to report a "
                         f"{args.gpus}-GPU line measured on {world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    rccl_info = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)
    elif args.dist_backend == "nccl" and not args.pmc_child and not args.no_rccl_world1:
        # N = 1 goes through RCCL too: a process group of ONE rank (no launcher, so the rendezvous is a TCP store of our own on 127.0.0.1),
        # the same per-step all-reduce of the per-sequence counts, the same barriers - the exchange step's code path runs on hardware
        # wherever this bench does.  Reporting only: a set-up that fails is noted and the run goes on without it.
        try:
            import datetime
            import socket
            import torch.distributed as dist_mod
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist_mod.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev,
                                        timeout=datetime.timedelta(seconds=120))
            one = torch.ones(1, dtype=torch.int64, device=dev)
            dist_mod.all_reduce(one)
            torch.cuda.synchronize()
            rccl_info = {"backend": "nccl (RCCL)", "world_size": 1, "rccl_ranks_seen": int(one.item()), "per_step_allreduce_of_sequence_counts": True}
            dist = dist_mod
        except Exception as e:
            rccl_info = {"backend": "nccl (RCCL)", "world_size": 1, "error": repr(e)}
            dist = None

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
    seq, seq_lens = synth.make_genome(total_bp, dev, seed=cfg.get("seed", 38), n_seqs=cfg.get("n_seqs", 24), repeat_frac=cfg.get("repeat_frac", 0.45))
    n = seq.numel()
    torch.cuda.synchronize()
    log(f"genome: {n} concatenated bases, {len(seq_lens)} sequences, generated in {time.time() - t0:.1f}s")
    t0 = time.time()
    E = 5 if (n >= 0xFFFFFFFF or args.force_el5) else 4            # SfxElSize, as `biokanga index` picks it
    sa = torch.empty(n * (5 if E == 5 else 4), dtype=torch.uint8, device=dev) if E == 5 else torch.empty(n, dtype=torch.int32, device=dev)
    bk.build_sa_device(seq.data_ptr(), n, sa.data_ptr(), E, local_rank)
    torch.cuda.synchronize()
    log(f"suffix array built on the GPU in {time.time() - t0:.1f}s")
    entries = synth.entry_table(seq_lens)
    ent = np.zeros(len(entries), dtype=bk.ENTRY_DTYPE)
    for i, (eid, slen, so, eo) in enumerate(entries):
        ent[i] = (eid, slen, so, eo, f"chr{eid}".encode(), b"")
    params_kw = dict(max_subs=args.max_subs)
    pe_params = bk.PEParams(pe["pe_mode"], pe["pair_min_len"], pe["pair_max_len"], False) if pe else None
    t0 = time.time()
    al = bk.Aligner(None, bk.AlignParams(**params_kw), device=local_rank, d_seq=seq.data_ptr(), concat_len=n,
                    d_sa=sa.data_ptr(), el_size=E, entries=ent)
    if args.kmer_bits:
        al.tune("kmer_bits", args.kmer_bits)
    for kv in args.tune:
        k, v = kv.split("=")
        al.tune(k, int(v))
    log(f"context (packed target + k-mer table) ready in {time.time() - t0:.1f}s; MinCoreLen {al.min_core_len}")

    def make_set(n_reads, seed):
        """one synthetic read set of this config: (bases, offs, lens) in HBM"""
        if pe:
            return synth.make_pairs(seq, seq_lens, n_reads // 2, args.read_len, dev, seed=seed, max_subs=args.max_subs)
        b, o, l, _ = synth.make_reads(seq, seq_lens, n_reads, args.read_len, dev, seed=seed, max_subs=args.max_subs)
        return b, o, l

    t0 = time.time()
    # weak scaling: every rank owns a batch of its own.  Read g of the job's global set = read g // N of rank g % N's batch,
    # i.e. the global set is dealt i mod N; the strong-scaling and shard-check legs below deal ONE common set
    rd_bases, rd_offs, rd_lens = make_set(args.reads, 1000 + rank)
    out = torch.zeros(args.reads * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    counts_dev = torch.zeros(len(entries), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    log(f"reads: {args.reads} x {args.read_len} bp generated in {time.time() - t0:.1f}s")
    keep_index_for_baseline = args.cpu_baseline_secs > 0 and rank == 0 and world == 1
    # the context holds its own image; the bench's copy of the suffix array (12 GB at 3.1 Gbp, 85 GB at 17 Gbp; with 5-byte elements
    # the sequence too) leaves the HBM before the batch scratch - and the context's window array - are sized
    t0 = time.time()
    sa = sa.cpu() if keep_index_for_baseline else None
    if E == 5:
        seq = seq.cpu() if keep_index_for_baseline else None
    torch.cuda.empty_cache()
    log(f"index copies moved off the device in {time.time() - t0:.0f}s")
    if E == 5:
        # with the bench's own 102 GB out of the way the second-level keys (68 GB at 17 Gbp) fit: built now (a context made from an
        # .sfx file has the room from the start)
        t0 = time.time()
        al.tune("use_k2", 1)
        log(f"second-level keys (re)built in {time.time() - t0:.1f}s")

    def run_step(bases, offs, lens, nreads, dst):
        al.align_device(bases.data_ptr(), offs.data_ptr(), lens.data_ptr(), nreads, dst.data_ptr())
        if pe:
            al.pair_device(bases.data_ptr(), offs.data_ptr(), lens.data_ptr(), nreads // 2, dst.data_ptr(), pe_params)
        # the path's one exchange step: per-sequence hit counts summed over ranks
        c = al.seq_counts(reset=True)
        counts_dev.copy_(torch.from_numpy(c.astype(np.int64)))
        if dist is not None:
            all_reduce(counts_dev)

    def step():
        run_step(rd_bases, rd_offs, rd_lens, args.reads, out)

    # ---------------------------------------------------------------- the index layout of the headline, and the other one
    job_reads = max(args.reads, cfg.get("job_reads_per_gpu", args.reads))
    policy_flags = bk.image_policy(job_reads)                       # (the function `biokanga align` calls)
    policy_layout = "partial" if policy_flags & bk.CTX_WINDOW_ARRAY_EAGER else "off"
    policy_image = "lean" if policy_flags & (bk.CTX_LEAN_IMAGE | bk.CTX_GROW_IMAGE) else "full"
    headline_image = policy_image if args.index_image == "policy" else args.index_image
    if E != 4:
        headline_image = "full"                                       # (5-byte indexes: whatever of the tables fits is made; there is no second image to switch to)

    def set_image(image):
        """the context was made with every table the HBM has room for; 'lean' drops the k-mer table's second words and the key arrays
        behind the second-level keys (the tables are made again: untimed set-up)"""
        if E != 4:
            return
        al.tune("use_ktab2", 1 if image == "full" else 0)
        al.tune("use_k3", 2 if image == "full" else 0)

    if headline_image == "lean":
        set_image("lean")
    image_resident = {"k_mer_table_second_words": al.tune("ktab2_resident", 0) == 1, "key_arrays_behind_the_second_level_keys": al.tune("k3_resident", 0)}
    headline = {"policy": policy_layout, "on": "partial"}.get(args.window_array, args.window_array)
    host = None
    if not args.no_host_leg:
        try:
            host = HostLeg(rd_bases, rd_lens, args)
        except Exception as e:       # reporting only - never lose the measured line
            log(f"host leg unavailable: {e!r}")

    def measure(layout):
        """both clocks in one index layout: `warmup` untimed + exactly `steps` timed steps each, a barrier + device synchronisation on
        both sides of the timed steps, the maximum over ranks"""
        res = {"layout": layout}
        torch.cuda.empty_cache()                           # (what the library may give the window array is what hipMemGetInfo calls free)
        t_set = time.time()
        al.tune("use_swin", SWIN_TUNE[layout])
        step()                                             # (builds the window array when it is asked for, fits and serves these reads)
        if E != 4 and layout == "forced":
            step()                                         # (an index of 5-byte elements: made when a second batch arrives, from what the first one's phases say)
        torch.cuda.synchronize()
        t_first = time.time() - t_set
        res["window_array_resident"] = al.tune("swin_resident", 0) == 1
        if res["window_array_resident"]:
            # the library's own clock around allocation + construction, what array and map occupy, the share of the suffix array held
            res["window_array"] = {"setup_s": al.tune("swin_setup_us", 0) / 1e6, "gb": al.tune("swin_mbytes", 0) * 1048576 / 1e9,
                                   "share_of_suffix_array": al.tune("swin_covered_ppm", 0) / 1e6}
        for _ in range(max(0, args.warmup - 1)):
            step()
        first_out = out.clone()                            # results of an untimed step, to check repeatability
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
        res.update(kernel_only_elapsed=elapsed, ctr=al.counters(), tim=al.timing(reset=True),
                   repeatable=bool(torch.equal(first_out, out)), first_step_s=t_first)
        del first_out
        res["hits"] = out.cpu().numpy().view(bk.HIT_DTYPE).copy()
        if host is not None:
            try:
                res["host"] = host.run(al, args.warmup, args.steps, barrier, all_reduce if dist is not None else None, dev, pe_params)
                res["host"]["results_bit_identical_to_kernel_only_steps"] = host.same_as(res["hits"], args.steps)
            except Exception as e:
                res["host"] = {"value": None, "error": repr(e)}
        return res

    if args.sweep:
        al.tune("use_swin", SWIN_TUNE[headline])
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
    if args.pmc_child:
        # what the counter passes profile: ONE step in the headline's layout, nothing reported
        al.tune("use_swin", SWIN_TUNE[headline])
        step()
        torch.cuda.synchronize()
        al.close()
        return
    main_leg = measure(headline)
    if headline != "off" and not main_leg["window_array_resident"]:
        headline = "off"                 # (asked for, but this index / these reads cannot have it: 5-byte elements, long reads, no room)
    other_legs = {}
    if not args.no_other_layout:
        # (an index of 5-byte elements: the partial array made when asked for, "use_swin" 2 - the policy leaves such an index without)
        for lay in ([x for x in args.other_layouts.split(",") if x in SWIN_TUNE and x != headline] if E == 4 else [x for x in ("forced",) if x != headline]):
            leg = measure(lay)
            if lay == "off" or leg["window_array_resident"]:      # (else the library did not build it: there is no such layout to report)
                other_legs[lay] = leg
        al.tune("use_swin", SWIN_TUNE[headline])
    # the other index image, beside the headline's: lean (what bk_image_policy gives a job below BK_POLICY_MIN_READS reads per device: no
    # second words in the k-mer table, no key arrays behind the second-level keys) when the headline's holds every table, and the other way round
    other_image_name = "lean" if headline_image == "full" else "full"
    other_image = None
    if not args.no_other_layout and E == 4 and world == 1:
        try:
            al.tune("use_swin", SWIN_TUNE[headline])
            set_image(other_image_name)
            step()
            torch.cuda.synchronize()
            al.timing(reset=True)
            t1 = time.time()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt = time.time() - t1
            tl = al.timing(reset=True)
            other_image = {"image": other_image_name, "value_kernel_only": args.reads * args.steps / dt, "ms_per_step_kernel_only": 1e3 * dt / args.steps,
                           "device_ms_per_step": {k: round(tl[k] / args.steps, 2) for k in ("ms_total", "ms_search", "ms_search_a", "ms_search_b", "ms_extend", "ms_heavy")},
                           "results_bit_identical_to_the_headline_image": bool(np.array_equal(main_leg["hits"].view(np.uint8), out.cpu().numpy()))}
            if host is not None:
                try:
                    hl = host.run(al, args.warmup, args.steps, barrier, None, dev, pe_params)
                    other_image["value_host_in_host_out"] = hl.get("value")
                    other_image["host_results_bit_identical"] = host.same_as(main_leg["hits"], args.steps)
                except Exception as e:
                    other_image["value_host_in_host_out"] = None
                    other_image["host_error"] = repr(e)
        except Exception as e:       # reporting only
            other_image = {"image": other_image_name, "value_kernel_only": None, "error": repr(e)}
        set_image(headline_image)
        step()
        torch.cuda.synchronize()
    # the library's own exchange step (`biokanga align --devices`: bk_seq_counts_allreduce) through its RCCL binding on this one device
    lib_rccl = None
    if world == 1 and not args.no_rccl_world1:
        try:
            step()
            torch.cuda.synchronize()
            al.align_device(rd_bases.data_ptr(), rd_offs.data_ptr(), rd_lens.data_ptr(), min(args.reads, 1_000_000), out.data_ptr())
            plain = al.seq_counts(reset=False)
            al.tune("force_rccl", 1)
            through = bk.seq_counts_allreduce([al], reset=True)
            lib_rccl = {"counts_equal_plain_path": bool(np.array_equal(plain, through)), "allreduces_through_rccl": al.tune("rccl_allreduces", 0),
                        "communicator_ranks": al.tune("rccl_ranks", 0)}
            al.tune("force_rccl", 0)
        except Exception as e:
            lib_rccl = {"error": repr(e)}
    headline_on = headline != "off"
    if os.environ.get("BK_DIAG"):
        print("diag counters:", main_leg["ctr"], file=sys.stderr)
        import ctypes
        arr = (ctypes.c_ulonglong * 16)()
        if bk.load_library().bk_debug_prof(arr) == 0:
            print("diag prof:", list(arr), file=sys.stderr)
        if hasattr(bk.load_library(), "bk_debug_prof_wave") and bk.load_library().bk_debug_prof_wave(arr) == 0:
            print("diag prof wave:", list(arr)[:8], file=sys.stderr)
    ctr, tim, hits = main_leg["ctr"], main_leg["tim"], main_leg["hits"]
    accepted = int((hits["nar"] == 1).sum())
    total_reads = args.reads * world * args.steps
    kernel_only_value = total_reads / main_leg["kernel_only_elapsed"]
    host_leg = main_leg.get("host")
    host_ok = bool(host_leg and host_leg.get("value"))
    value = host_leg["value"] if host_ok else kernel_only_value
    ms_per_step = host_leg["ms_per_step"] if host_ok else main_leg["kernel_only_elapsed"] / args.steps * 1e3

    multi = None
    if world > 1 and E == 4:
        multi = multi_gpu_legs(al, make_set, run_step, counts_dev, args, rank, world, dev, barrier, all_reduce, dist)

    # roofline: algorithmic bytes over the kernel's summed launch time, measured with HIP events on the stream the
    # kernels run on, in the headline's layout.  Per search: ceil(log2 N) * (E + 8) - the formula prices a search whichever of the two
    # passes settles it, so the search passes are timed separately (bk_timing) but priced as one stage; per candidate
    # Hamming-extended: E + ceil(L/2).  The dominant kernel = the one with the largest summed launch time.
    log2n = math.ceil(math.log2(n))
    per_search = log2n * (E + 8)
    per_cand = E + (args.read_len + 1) // 2
    io_bytes = args.reads * args.steps * ((args.read_len + 3) // 4 + 16)

    def kernels_of(c, t):
        stage_bytes = c["n_search"] * per_search
        k = {
            "k_search_a": dict(bytes=stage_bytes, ms=t["ms_search_a"], launches=t["n_search_launches"], priced_with="k_search stage"),
            "k_search_b": dict(bytes=stage_bytes, ms=t["ms_search_b"], launches=t["n_search_b_launches"], priced_with="k_search stage"),
            "k_wave": dict(bytes=c["n_cand_heavy"] * per_cand, ms=t["ms_heavy"], launches=t["n_heavy_launches"]),
            "k_flat": dict(bytes=(c["n_cand"] - c["n_cand_heavy"]) * per_cand, ms=t["ms_extend"], launches=t["n_extend_launches"]),
        }
        stage = dict(bytes=stage_bytes, ms=t["ms_search"], launches=t["n_search_launches"])
        for q in list(k.values()) + [stage]:
            q["GBs"] = q["bytes"] / max(1e-9, q["ms"] * 1e-3) / 1e9
        for nm in ("k_search_a", "k_search_b"):
            k[nm]["GBs"] = stage["GBs"]              # (a pass alone has no algorithmic-bytes figure of its own)
        return k, stage

    kern, search_stage = kernels_of(ctr, tim)
    dom = max(kern, key=lambda q: kern[q]["ms"])
    dom_ms, dom_launches = kern[dom]["ms"], kern[dom]["launches"]
    ach = kern[dom]["GBs"]
    whole = (search_stage["bytes"] + kern["k_wave"]["bytes"] + kern["k_flat"]["bytes"] + io_bytes) / max(1e-9, tim["ms_total"] * 1e-3) / 1e9
    default_workload = (args.config == "C2" and args.reads == 50_000_000 and args.read_len == 100 and args.max_subs == 3 and total_bp == 3_100_000_000)
    # everything below runs in child processes (counter passes under rocprofv3, the command lines): this process hands its HBM back first
    if keep_index_for_baseline:
        seq = seq.cpu() if seq is not None else None
        rd_bases = rd_bases.cpu()
    al.close()
    host = None
    del out, rd_offs, rd_lens, counts_dev
    if not keep_index_for_baseline:
        del rd_bases, seq
    torch.cuda.empty_cache()
    traffic = traffic_raw = None
    traffic_src = None
    corr, corr_src = fetch_correction(dom, headline)
    if rank == 0 and world == 1 and not args.no_live_traffic and E == 4:       # (two more 17 Gbp set-ups would take minutes)
        child_args = ["--config", args.config, "--reads", str(args.reads), "--genome-mbp", str(args.genome_mbp)] + \
                     [x for kv in args.tune for x in ("--tune", kv)]
        got, traffic_src = live_traffic(dom, child_args, headline)
        if got is None:
            log(f"live traffic unavailable: {traffic_src}")
        else:
            per = max(1.0, dom_launches / max(1, args.steps))                   # this run's launches per step
            traffic_raw = (got[0] + got[1]) / per
            traffic = (got[0] * corr + got[1]) / per
    if traffic is None and default_workload and not headline_on:
        got, src = profiled_traffic(dom)
        if got:
            traffic_raw, traffic = got[0] + got[1], got[0] * corr + got[1]
            traffic_src = f"committed profile {src} (not observed in this run)"
    roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_as_counted": traffic_raw, "traffic_source": traffic_src,
                "traffic_correction": {"factor_on_FETCH_SIZE": corr, "source": corr_src},
                "window_array": headline,
                "algorithmic_bytes_per_launch": kern[dom]["bytes"] / max(1, dom_launches),
                "avg_launch_ms": dom_ms / max(1, dom_launches),
                "per_kernel": {k: {"algorithmic_GBs": round(v["GBs"], 1), "ms": round(v["ms"], 2), "launches": v["launches"],
                                   **({"priced_with": v["priced_with"]} if "priced_with" in v else {})} for k, v in kern.items()},
                "k_search_stage": {"algorithmic_GBs": round(search_stage["GBs"], 1), "ms": round(search_stage["ms"], 2),
                                   "ms_pass_a": round(tim["ms_search_a"], 2), "ms_grouping": round(tim["ms_search_sort"], 2), "ms_pass_b": round(tim["ms_search_b"], 2)},
                "whole_step_algorithmic_GBs": whole, "whole_step_frac": whole / HBM_PEAK_GBS,
                "device_ms": {k: tim[k] for k in ("ms_total", "ms_search", "ms_extend", "ms_heavy", "ms_other", "ms_prep")}}

    def layout_record(leg):
        k2, st2 = kernels_of(leg["ctr"], leg["tim"])
        d = max(k2, key=lambda q: k2[q]["ms"])
        return {"window_array": leg["layout"], **({"window_array_built": leg["window_array"]} if "window_array" in leg else {}),
                "value_host_in_host_out": (leg.get("host") or {}).get("value"),
                "value_kernel_only": total_reads / leg["kernel_only_elapsed"],
                "ms_per_step_kernel_only": leg["kernel_only_elapsed"] / args.steps * 1e3,
                "device_ms_per_step": {k: round(leg["tim"][k] / args.steps, 2) for k in ("ms_total", "ms_search", "ms_search_a", "ms_search_b", "ms_extend", "ms_heavy", "ms_other")},
                "dominant_kernel": d, "dominant_kernel_frac_of_hbm_peak": k2[d]["GBs"] / HBM_PEAK_GBS,
                "results_bitwise_equal_across_steps": leg["repeatable"],
                "host_results_bit_identical": (leg.get("host") or {}).get("results_bit_identical_to_kernel_only_steps")}

    layouts = {"window_array_" + headline: layout_record(main_leg)}
    for lay, leg in other_legs.items():
        layouts["window_array_" + lay] = layout_record(leg)
    if other_legs:
        layouts["results_bit_identical_between_layouts"] = bool(all(np.array_equal(main_leg["hits"].view(np.uint8), leg["hits"].view(np.uint8))
                                                                    for leg in other_legs.values()))
    part_rec, full_rec, off_rec = layouts.get("window_array_partial") or layouts.get("window_array_forced"), layouts.get("window_array_full"), layouts.get("window_array_off")
    built = ((layouts.get("window_array_" + headline) or {}).get("window_array_built") or (part_rec or {}).get("window_array_built") or {})
    setup_s = built.get("setup_s")

    fmt = dict(reads=args.reads, pairs=args.reads // 2, read_len=args.read_len, max_subs=args.max_subs)
    result = {
        "metric": metric_text(args, cfg, E),
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "value_clock": ("T_align of SURVEY 8d: reads leave pinned host memory (2 bit/base), results are back in host memory; "
                        f"{args.steps} steps between two barriers") if host_ok else
                       "kernel-only: reads and results resident in HBM (the host-in / host-out steps were switched off or failed)",
        "value_kernel_only": kernel_only_value,
        "value_no_window_array": (off_rec or {}).get("value_host_in_host_out") or (off_rec or {}).get("value_kernel_only"),
        "value_window_array": (part_rec or {}).get("value_host_in_host_out") or (part_rec or {}).get("value_kernel_only"),
        "value_window_array_full": (full_rec or {}).get("value_host_in_host_out") or (full_rec or {}).get("value_kernel_only"),
        "window_array_setup_s": setup_s, "window_array_gb": built.get("gb"), "window_array_share_of_suffix_array": built.get("share_of_suffix_array"),
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{args.config}: {cfg['text'].format(**fmt)} vs synthetic {'wheat' if E == 5 else 'GRCh38'}-like genome of {total_bp} bp in "
                               f"{len(seq_lens)} sequences ({int(100 * cfg.get('repeat_frac', 0.45))}% repeat-derived, N gaps), {cfg['cli'].format(**fmt)}",
                   "window_array": headline + (f" (what `biokanga align` picks for the {job_reads} reads a device aligns in this configuration: the partial suffix-ordered "
                                    f"window array from {bk.POLICY_MIN_READS} reads on, or --window-array)" if args.window_array == "policy" else " (forced with --window-array)"),
                   "index_image": headline_image + (f" (what bk_image_policy - the function `biokanga align` calls - gives the {job_reads} reads a device aligns in this "
                                                    f"configuration: every table from {bk.POLICY_MIN_READS} reads on)" if args.index_image == "policy" else " (forced with --index-image)"),
                   "reads_per_gpu_per_step": args.reads, "read_len": args.read_len, "genome_bp": total_bp,
                   "concat_len": n, "sfx_el_size": E, "index": "replicated per GPU, built on device",
                   "parallelism": f"reads sharded over {world} GPU(s): read g of the job's set = read g // {world} of rank g % {world}",
                   "accepted_frac_rank0": accepted / args.reads,
                   "n_search_per_read": ctr["n_search"] / (args.reads * args.steps),
                   "n_cand_per_read": ctr["n_cand"] / (args.reads * args.steps),
                   "heavy_calls_frac": ctr["n_heavy"] / max(1, ctr["n_lcm_calls"]),
                   "results_bitwise_equal_across_steps": main_leg["repeatable"]},
        "roofline": roofline,
        "layouts": layouts,
        "index_image": {"headline": headline_image, "policy": policy_image, "policy_flags_of_bk_image_policy": policy_flags, "job_reads_per_gpu": job_reads,
                        "resident": image_resident,
                        "full_is": "k-mer table entries of two words (a bucket's only key, or the map of its keys' first five bits), third- and fourth-level search keys: 43 GB "
                                   "more at 3.1 Gbp, made behind the suffix array's upload with the other tables",
                        "other": other_image},
        "t_align_host_resident": host_leg,
        "rccl": {"torch_distributed": rccl_info, "library_binding_one_device": lib_rccl} if world == 1 else {"torch_distributed": {"backend": args.dist_backend, "world_size": world}},
    }
    if multi is not None:
        result.update(multi)
    if keep_index_for_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(seq, sa, entries, rd_bases, args.read_len, params_kw, hits,
                                                  args.cpu_baseline_secs, args.reference_reads, local_rank, pe=pe,
                                                  full_cli=not args.no_full_cli)
        except Exception as e:      # the baseline is reporting only - never lose the measured line
            result["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": effective_cpus(), "kind": "port",
                                      "sample": f"failed: {e!r}"}
    al.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), file=_REAL_STDOUT or sys.stdout, flush=True)


def multi_gpu_legs(al, make_set, run_step, counts_dev, args, rank, world, dev, barrier, all_reduce, dist):
    """N > 1 only.  (1) shard check: ONE read set, the same on every rank, is dealt i mod N; the all-reduced per-sequence counts
    and NAR histogram must equal those of rank 0 aligning the whole set alone.  (2) strong scaling: ONE set of the config's
    per-step size dealt i mod N, timed like the weak-scaling steps."""
    import numpy as np
    import torch
    import biokanga_amd as bk
    res = {}
    L = args.read_len
    unit = 2 if CONFIGS[args.config]["pe"] else 1            # pairs stay together

    def shard_of(bases, lens, n_total):
        """rows r, r + N, .. (whole pairs) of a set -> contiguous (bases, offs, lens, n)"""
        rows = bases.view(n_total // unit, unit * L)[rank::world].contiguous()
        m = rows.shape[0] * unit
        return rows.view(-1), torch.arange(m, dtype=torch.int64, device=dev) * L, lens[:m].contiguous(), m

    # ---- (1) shard check
    G = max(unit * world, args.shard_check_reads - args.shard_check_reads % (unit * world))
    gb, go, gl = make_set(G, 777)
    sb, so, sl, m = shard_of(gb, gl, G)
    d_hits = torch.zeros(m * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    al.seq_counts(reset=True)
    run_step(sb, so, sl, m, d_hits)                      # leaves the all-reduced counts in counts_dev
    reduced_counts = counts_dev.cpu().numpy().copy()
    nar = torch.from_numpy(np.bincount(d_hits.cpu().numpy().view(bk.HIT_DTYPE)["nar"], minlength=20).astype(np.int64)).to(dev)
    all_reduce(nar)
    check = None
    if rank == 0:
        whole = torch.zeros(G * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        al.align_device(gb.data_ptr(), go.data_ptr(), gl.data_ptr(), G, whole.data_ptr())
        if CONFIGS[args.config]["pe"]:
            pp = CONFIGS[args.config]["pe"]
            al.pair_device(gb.data_ptr(), go.data_ptr(), gl.data_ptr(), G // 2, whole.data_ptr(),
                           bk.PEParams(pp["pe_mode"], pp["pair_min_len"], pp["pair_max_len"], False))
        one_counts = al.seq_counts(reset=True).astype(np.int64)
        one_nar = np.bincount(whole.cpu().numpy().view(bk.HIT_DTYPE)["nar"], minlength=20).astype(np.int64)
        check = {"reads": G, "dealt": f"read i -> rank i mod {world}" + (" (pairs kept together)" if unit == 2 else ""),
                 "per_sequence_counts_equal_1gpu_run": bool(np.array_equal(reduced_counts, one_counts)),
                 "nar_histogram_equal_1gpu_run": bool(np.array_equal(nar.cpu().numpy(), one_nar)),
                 "accepted": int(one_counts.sum())}
        del whole
    al.seq_counts(reset=True)
    res["shard_check"] = check
    del gb, go, gl, sb, so, sl, d_hits
    # ---- (2) strong scaling: the config's per-step read count in total
    T = args.reads - args.reads % (unit * world)
    gb, go, gl = make_set(T, 2000)
    sb, so, sl, m = shard_of(gb, gl, T)
    del gb, go
    d_hits = torch.zeros(m * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    run_step(sb, so, sl, m, d_hits)                      # warm-up
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        run_step(sb, so, sl, m, d_hits)
    barrier()
    el = time.time() - t0
    t = torch.tensor([el], dtype=torch.float64, device=dev)
    all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    res["strong_scaling"] = {"reads_total_per_step": T, "reads_per_gpu_per_step": m, "value": T * args.steps / el, "unit": "reads/s",
                             "ms_per_step": el / args.steps * 1e3, "steps": args.steps}
    return res


if __name__ == "__main__":
    main()
