#!/bin/bash
# A/B of library variants on the GPU box (tools/build_variant.sh makes them): the kernel-only C2 steps of bench.py, once per variant, device
# milliseconds per kernel family side by side.   tools/ab_variants.sh <out dir> <variant> [<variant> ..]     ("main" = the regular library)
set -u
O=$1; shift
mkdir -p $O
for v in "$@"; do
  lib=biokanga_amd/lib/libbiokanga_amd_$v.so
  [ "$v" = main ] && lib=biokanga_amd/lib/libbiokanga_amd.so
  BK_LIB=$PWD/$lib timeout 600 python3 bench.py --no-host-leg --cpu-baseline-secs 0 --no-live-traffic --no-other-layout --no-rccl-world1 ${AB_ARGS:-} > $O/ab_$v.json 2> $O/ab_$v.err
  python3 - "$O/ab_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    k = {n: round(q["ms"] / d["steps"], 2) for n, q in r["per_kernel"].items()}
    print(f"{sys.argv[2]:>12}: kernel-only {d['value_kernel_only'] / 1e6:7.1f} M reads/s; ms per step {k}; search stage {round(r['k_search_stage']['ms'] / d['steps'], 2)}; read preparation {round(r['device_ms']['ms_prep'] / d['steps'], 2)}; dominant {r['kernel']} frac {r['frac']:.3f}; image {d['index_image']['headline']}")
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
done
