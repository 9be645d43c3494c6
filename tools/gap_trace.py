#!/usr/bin/env python3
"""Where the GPU idles during the host-in / host-out steps: from a rocprofv3 --kernel-trace csv of a bench run, the kernels of the LAST
`span_s` seconds of the trace (the host leg runs last when the other legs are switched off), their busy time, and every gap between one
kernel's end and the next one's start beyond `min_gap_us`, with the kernels on either side.

usage: gap_trace.py <dir with *_kernel_trace.csv> [span_s] [min_gap_us]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    span = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]))
    rows.sort()
    if not rows:
        print("no kernels")
        return
    t_end = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= t_end - span * 1e9]
    t0 = rows[0][0]
    busy = 0
    cur_end = rows[0][0]
    gaps = []
    prev = None
    for s, e, k in rows:
        if s > cur_end:
            if (s - cur_end) / 1e3 >= min_gap and prev:
                gaps.append(((s - cur_end) / 1e3, (cur_end - t0) / 1e6, prev, k))
            busy += e - s
            cur_end = e
            prev = k
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
                prev = k
    total = (cur_end - t0) / 1e6
    print(f"last {total:.1f} ms of the trace: {len(rows)} kernels, device busy {busy / 1e6:.1f} ms ({100 * busy / 1e6 / total:.1f} %), "
          f"{len(gaps)} gaps of >= {min_gap:.0f} us totalling {sum(g[0] for g in gaps) / 1e3:.1f} ms")
    small = total - busy / 1e6 - sum(g[0] for g in gaps) / 1e3
    print(f"gaps below {min_gap:.0f} us: {small:.1f} ms in all")
    by = {}
    for s_, e_, k in rows:
        c = by.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += e_ - s_
    print("kernels of that span by total duration (overlapping ones counted in full):")
    for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"  {k:50s} {n:5d} x, {t / 1e6:9.2f} ms, {t / n / 1e3:9.1f} us each")
    print("copy kernels of the runtime in that span (start ms, duration us):")
    print("  " + ", ".join(f"{(s_ - t0) / 1e6:.1f}: {(e_ - s_) / 1e3:.0f}" for s_, e_, k in rows if "copyBuffer" in k))
    for g, at, a, b in gaps:
        print(f"  at {at:9.2f} ms: {g:9.1f} us idle between {a} and {b}")


if __name__ == "__main__":
    main()
