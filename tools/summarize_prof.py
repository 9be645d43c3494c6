#!/usr/bin/env python3
"""Condense rocprofv3 output into the small summaries kept under profiles/.

  summarize_prof.py stats  <dir-with-*_kernel_stats.csv>          -> CSV of the bk:: kernels (name, calls, total/avg ns, %)
  summarize_prof.py pmc    <dir-with-*_counter_collection.csv>... -> per kernel, per counter: sum over dispatches and
                                                                    mean per dispatch (bk:: kernels + rand_access bench)

Kernel names are shortened to the bare function name (template arguments kept).  Nothing here touches
the GPU; it only reads CSV files that `rocprofv3 --kernel-trace --stats` / `--pmc` wrote.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"bk::(k_\w+(?:<[^>]*>)?)", name) or re.match(r"(k_calib\w+)", name)
    return m.group(1) if m else None


def find(d, pat):
    out = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not out:
        sys.exit(f"no {pat} under {d}")
    return out


def stats(d):
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct_of_all_gpu_time", "min_ns", "max_ns"])
    for f in find(d, "*_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k:
                w.writerow([k, r["Calls"], r["TotalDurationNs"], f'{float(r["AverageNs"]):.0f}', r["Percentage"], r["MinNs"], r["MaxNs"]])


def pmc(dirs):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    regs = {}
    for d in dirs:
        for f in find(d, "*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k:
                    continue
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
                regs[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "counter", "dispatches", "sum", "mean_per_dispatch", "vgpr", "agpr", "sgpr", "lds_bytes", "scratch_bytes"])
    for k in sorted(acc):
        for c in sorted(acc[k]):
            w.writerow([k, c, cnt[k][c], f"{acc[k][c]:.0f}", f"{acc[k][c] / cnt[k][c]:.1f}", *regs[k]])


if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in ("stats", "pmc"):
        sys.exit(__doc__)
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
