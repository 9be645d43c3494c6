#!/bin/bash
# A/B of library builds (tools/build_variant.sh) on the C2 kernel-only step, window array on:
#   tools/ab.sh <out dir> <variant name, or "-" for the regular build> ...
O=$1; shift
mkdir -p $O
for v in "$@"; do
  if [ "$v" = "-" ]; then unset BK_LIB; name=base; else export BK_LIB=$PWD/biokanga_amd/lib/libbiokanga_amd_$v.so; name=$v; fi
  python3 bench.py --no-host-leg --cpu-baseline-secs 0 --no-live-traffic --no-other-layout --window-array on --steps 3 > $O/ab_$name.json 2> $O/ab_$name.err
  python3 - "$O/ab_$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["layouts"].items():
    if isinstance(v, dict):
        print(sys.argv[2], k, "kernel-only %.1f M reads/s, %.2f ms/step" % (v["value_kernel_only"] / 1e6, v["ms_per_step_kernel_only"]), v["device_ms_per_step"])
PY
done
