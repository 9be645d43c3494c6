#!/usr/bin/env python3
"""End-to-end timing of our command line (`biokanga_amd/bin/biokanga align`) on the bench workload:
T_e2e (process start -> exit) and the phases from its time-stamped log.  Files live in /dev/shm.
  python tools/e2e_cli.py [n_reads] [--variants "name:ENV=V,ENV2=V;other:ENV=W"] [--repeat N] [--pause S] [--quiet] [--gz] [-- extra options of biokanga align]
Every variant is the same command with its own environment (the first run, "default", has none; BK_E2E_ARGS=<options> adds options of
biokanga align to a variant's command); the files are written once.
--gz: the reads also as reads.fa.gz (one member, level 1) and reads.fa.bgz (bgzip members); every variant then runs on the three inputs
(the record-by-record gzread reader these were measured against in round 4 - BK_GZ_SERIAL - is still what takes the files the whole-file
decoder declines, but no longer something to switch to)."""
import os, sys, time, subprocess, shutil, tempfile, datetime, re, struct, zlib
from concurrent.futures import ThreadPoolExecutor
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


def bgzf_blocks(data):
    out = []
    for o in range(0, len(data), 0xff00):
        piece = data[o:o + 0xff00]
        z = zlib.compressobj(1, zlib.DEFLATED, -15)
        c = z.compress(piece) + z.flush()
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(c) + 25) + c + struct.pack("<II", zlib.crc32(piece), len(piece)))
    return b"".join(out)


def write_gzip_copies(fa):
    """reads.fa -> reads.fa.gz (zlib level 1, one member) and reads.fa.bgz (bgzip members made by every core)"""
    t = time.time()
    z = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(fa, "rb") as f, open(fa + ".gz", "wb") as g:
        while True:
            buf = f.read(64 << 20)
            if not buf:
                break
            g.write(z.compress(buf))
        g.write(z.flush())
    t1 = time.time()
    with open(fa, "rb") as f, open(fa + ".bgz", "wb") as g, ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:    # (zlib works outside the interpreter's lock)
        step = 0xff00 * 256
        chunks = iter(lambda: f.read(step), b"")
        for out in ex.map(bgzf_blocks, chunks):
            g.write(out)
        g.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0")
    print(f"gzip copies written: .gz {os.path.getsize(fa + '.gz') / 1e9:.2f} GB in {t1 - t:.1f} s, .bgz {os.path.getsize(fa + '.bgz') / 1e9:.2f} GB in {time.time() - t1:.1f} s")


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
    pause = float(argv[argv.index("--pause") + 1]) if "--pause" in argv else 0.0      # seconds of rest before every run: the driver wipes what the process before gave back at about 40 GB/s, and a run that starts meanwhile waits for it
    with_gz = "--gz" in argv
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
        sfx, fa, sam, logf = (os.path.join(tmp, x) for x in ("genome.sfx", "reads.fa", os.environ.get("BK_E2E_OUT", "out.sam"), "log.txt"))     # (BK_E2E_OUT=out.sam.gz / out.bam: the other writers)
        bench.write_sfx_file(sfx, seq_h, sa_h, [(f"chr{e[0]}", e[1]) for e in entries])
        bench.write_fasta_file(fa, reads_h, n_reads, 100)
        del seq_h, sa_h, reads_h
        inputs = [("", fa)]
        if with_gz:
            write_gzip_copies(fa)
            inputs += [(" .gz", fa + ".gz"), (" .bgz", fa + ".bgz")]
            variants = [(n + tag, dict(e, BK_E2E_INPUT=path)) for n, e in variants for tag, path in inputs]
        first_size = None
        for name, env in variants:
            env = dict(env)
            fa_in = env.pop("BK_E2E_INPUT", fa)
            pause_v = float(env.pop("BK_E2E_PAUSE", pause))        # (a variant's own rest before its runs)
            more = env.pop("BK_E2E_ARGS", "").split()              # (a variant's own options of biokanga align: "name:BK_E2E_ARGS=--no-window-array")
            for rep in range(repeat):
                for f in (sam, logf, logf + ".err"):
                    if os.path.exists(f):
                        os.unlink(f)
                if pause_v:
                    time.sleep(pause_v)
                t = time.time()
                rc = subprocess.run([os.path.join(ROOT, env.get("BK_E2E_BIN", os.path.join("biokanga_amd", "bin", "biokanga"))), "align", "-i", fa_in, "-I", sfx, "-o", sam, "-s3", "-M6", "-F", logf] + extra + more,
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
