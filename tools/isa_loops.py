#!/usr/bin/env python3
"""Loops of one kernel in an AMDGPU assembly listing (hipcc --save-temps: *-gfx950.s), with what they are made of:
   tools/isa_loops.py <listing.s> <substring of the kernel's mangled name> [--dump N]
For every backward branch: the label range, instruction count, and the number of v_readlane / v_writelane (scalar registers spilled to
vector lanes), s_load, LDS, global/buffer memory and scalar / vector ALU instructions inside.  --dump N prints loop N's instructions."""
import re
import sys
from collections import Counter


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else None
    lines = open(path).read().split("\n")
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w*:", l) and key in l:
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    body = lines[start:end]
    ins = []            # (index in body, text)
    label_at = {}
    for i, l in enumerate(body):
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            label_at[m.group(1)] = len(ins)
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        ins.append(s.split(";")[0].strip())
    print(f"{lines[start][:-1]}: {len(ins)} instructions")
    def kind(t):
        op = t.split()[0]
        if op.startswith("v_readlane") or op.startswith("v_readfirstlane"): return op.split("_b32")[0]
        if op.startswith("v_writelane"): return "v_writelane"
        if op.startswith("s_load") or op.startswith("s_buffer_load"): return "s_load"
        if op.startswith("ds_"): return "lds"
        if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
        if op.startswith("scratch_"): return "scratch"
        if op.startswith("s_waitcnt"): return "s_waitcnt"
        if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
        if op.startswith("s_"): return "salu"
        if op.startswith("v_"): return "valu"
        return "other"
    tot = Counter(kind(t) for t in ins)
    print("whole kernel:", dict(tot))
    loops = []
    for i, t in enumerate(ins):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            loops.append((label_at[m.group(1)], i, m.group(1)))
    loops.sort(key=lambda x: (x[0], -x[1]))
    for n, (a, b, lab) in enumerate(loops):
        c = Counter(kind(t) for t in ins[a:b + 1])
        depth = sum(1 for (a2, b2, _) in loops if a2 <= a and b2 >= b) - 1
        print(f"loop {n:2d} depth {depth} {lab:12s} [{a:6d},{b:6d}] {b - a + 1:5d} instr: " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
        if dump == n:
            for t in ins[a:b + 1]:
                print("      ", t)


if __name__ == "__main__":
    main()
