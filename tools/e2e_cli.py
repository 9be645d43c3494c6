#!/usr/bin/env python3
"""End-to-end timing of our command line (`biokanga_amd/bin/biokanga align`) on the bench workload:
T_e2e (process start -> exit) and the phases from its time-stamped log.  Files live in /dev/shm.
  python tools/e2e_cli.py [n_reads] [--variants "name:ENV=V,ENV2=V;other:ENV=W"] [--repeat N] [--quiet] [-- extra options of biokanga align]
Every variant is the same command with its own environment (the first run, "default", has none); the files are written once."""
import os, sys, time, subprocess, shutil, tempfile, datetime, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import biokanga_amd as bk
from biokanga_amd import synth
import bench

KEYS = ("Loading suffix", "suffix array loaded", "Loading reads", "Load:", "Now aligning", "Alignment of", "Sorting",
        "Header written", "Completed reporting", "Reporting of aligned result set completed", "phase:", "Device pipeline", "window array", "Exit code")


def stamps_of(logf):
    out = []
    for line in open(logf, errors="replace"):
        m = re.match(r"\[(\w+\s+\d+ \d+:\d+:\d+\.\d+ \d+)\]", line)
        if m:
            out.append((datetime.datetime.strptime(re.sub(r"\s+", " ", m.group(1)), "%b %d %H:%M:%S.%f %Y").timestamp(), line))
    return out


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        extra = argv[argv.index("--") + 1:]
        argv = argv[:argv.index("--")]
    n_reads = int(argv[0]) if argv and argv[0].isdigit() else 20_000_000
    variants = [("default", {})]
    if "--variants" in argv:
        for v in argv[argv.index("--variants") + 1].split(";"):
            name, _, envs = v.partition(":")
            variants.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
    repeat = int(argv[argv.index("--repeat") + 1]) if "--repeat" in argv else 1
    quiet = "--quiet" in argv
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        quota = "?"
    def cg(name):
        try:
            return open("/sys/fs/cgroup/" + name).read().strip().replace("\n", " ")
        except OSError:
            return "?"
    print("kernel", os.uname().release)
    print("cgroup memory: max", cg("memory.max"), "high", cg("memory.high"), "current", cg("memory.current"), "swap.max", cg("memory.swap.max"))
    print(f"host: {os.cpu_count()} cpus, {len(os.sched_getaffinity(0))} in the affinity mask, cgroup cpu.max '{quota}'")
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
        first_size = None
        for name, env in variants:
            for rep in range(repeat):
                for f in (sam, logf, logf + ".err"):
                    if os.path.exists(f):
                        os.unlink(f)
                t = time.time()
                rc = subprocess.run([os.path.join(ROOT, env.get("BK_E2E_BIN", os.path.join("biokanga_amd", "bin", "biokanga"))), "align", "-i", fa, "-I", sfx, "-o", sam, "-s3", "-M6", "-F", logf] + extra,
                                    stdout=subprocess.DEVNULL, stderr=open(logf + ".err", "w"), env=dict(os.environ, BK_TIMING="1", **env), timeout=300).returncode
                wall = time.time() - t
                print("   cgroup memory.current", cg("memory.current"), "events:", cg("memory.events"), "peak", cg("memory.peak"))
                size = os.path.getsize(sam) if os.path.exists(sam) else 0
                first_size = first_size or size
                st = stamps_of(logf)
                at = lambda key: next((ts for ts, line in st if key in line), None)
                span = lambda a, b: (at(b) - at(a)) if at(a) and at(b) else float("nan")
                print(f"== {name} {env if env else ''} run {rep}: rc {rc}; T_e2e {wall:.2f} s = {n_reads / wall / 1e6:.2f} M reads/s; SAM {size / 1e9:.2f} GB{'' if size == first_size else '  SIZE DIFFERS'}")
                if st:
                    print(f"   load {span('Loading suffix', 'suffix array loaded'):.2f} s (reads parsed after {span('Loading suffix', 'Loading reads'):.2f}, accepted after {span('Loading suffix', 'Load:'):.2f}); "
                          f"align {span('Now aligning', 'Alignment of'):.2f}; to sort {span('Alignment of', 'Sorting'):.2f}; sort {span('Sorting', 'Header written'):.2f}; "
                          f"SAM {span('Header written', 'Completed reporting'):.2f}; log span {st[-1][0] - st[0][0]:.2f}; tear-down {t + wall - st[-1][0]:.2f}")
                if not quiet:
                    print(open(logf + ".err").read())
                    for ts, line in st:
                        if any(k in line for k in KEYS):
                            print(line.rstrip())
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
