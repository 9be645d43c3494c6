#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace csv: per kernel name the time spent, and the idle gaps between consecutive kernels (all streams
merged) over the last `--tail-ms` of the run, or between the a-th and b-th launch (counted from the end, `--anchor NAME a b`) of a kernel.
Usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [--tail-ms 400] [--anchor k_prep_fused 4 1]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    tail_ms = 400.0
    if "--tail-ms" in sys.argv:
        tail_ms = float(sys.argv[sys.argv.index("--tail-ms") + 1])
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70], r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(tail_ms * 1e6)
    if "--anchor" in sys.argv:
        i = sys.argv.index("--anchor")
        name, a, b = sys.argv[i + 1], int(sys.argv[i + 2]), int(sys.argv[i + 3])
        occ = [r[0] for r in rows if name in r[2]]
        t0, t_end = occ[-a], occ[-b]
        print(f"{len(occ)} launches of {name}; window = launch -{a} .. launch -{b}")
    rows = [r for r in rows if t0 <= r[0] < t_end]
    busy_until = rows[0][0]
    idle = 0
    per = {}
    gaps = []
    for s, e, name, q in rows:
        if s > busy_until:
            idle += s - busy_until
            gaps.append((s - busy_until, name, q))
        busy_until = max(busy_until, e)
        k = per.setdefault(name, [0, 0, set()])
        k[0] += e - s
        k[1] += 1
        k[2].add(q)
    span = rows[-1][1] - rows[0][0]
    print(f"window {span / 1e6:.1f} ms, {len(rows)} kernels, device idle between kernels {idle / 1e6:.2f} ms")
    for name, (t, n, qs) in sorted(per.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"  {t / 1e6:9.2f} ms {n:6d} x  {name}  queues {sorted(qs)}")
    # overlap: time during which kernels of more than one queue were running
    ev = []
    for s, e, name, q in rows:
        ev.append((s, 1, q))
        ev.append((e, -1, q))
    ev.sort()
    active = {}
    last = ev[0][0]
    multi = 0
    for t, d, q in ev:
        if sum(1 for v in active.values() if v > 0) > 1:
            multi += t - last
        last = t
        active[q] = active.get(q, 0) + d
    print(f"time with kernels of two or more queues running at once: {multi / 1e6:.2f} ms")
    print("largest gaps (ms, kernel that followed, queue):")
    for g, name, q in sorted(gaps, reverse=True)[:25]:
        print(f"  {g / 1e6:7.3f}  {name}  [{q}]")


if __name__ == "__main__":
    main()
