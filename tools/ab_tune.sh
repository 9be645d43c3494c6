#!/bin/bash
# main library, with and without a tune knob
O=$1; shift
mkdir -p $O
for t in "$@"; do
  name=$(echo "$t" | tr '= ' '__')
  args=""
  [ "$t" != none ] && args="--tune $t"
  timeout 600 python3 bench.py --no-host-leg --cpu-baseline-secs 0 --no-live-traffic --no-other-layout --no-rccl-world1 $args > $O/abt_$name.json 2> $O/abt_$name.err
  python3 - "$O/abt_$name.json" "$t" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().split("\n")[0])
    r = d["roofline"]
    k = {n: round(q["ms"] / d["steps"], 2) for n, q in r["per_kernel"].items()}
    print(f"{sys.argv[2]:>16}: kernel-only {d['value_kernel_only'] / 1e6:7.1f} M reads/s; ms per step {k}; search stage {round(r['k_search_stage']['ms'] / d['steps'], 2)}")
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
done
