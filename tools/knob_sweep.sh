#!/bin/bash
# Kernel-only rate of the default workload under single tuning knobs (run through gpurun from the repo root):
#   tools/knob_sweep.sh "search_ilp=4" "use_iv32=0" ...
for kv in "" "$@"; do
  args=""
  [ -n "$kv" ] && args="--tune $kv"
  line=$(python3 bench.py --steps 3 --warmup 1 --cpu-baseline-secs 0 --no-host-leg --no-live-traffic $args 2>/dev/null | grep '^{' | tail -1)
  echo "knob [$kv]: $(echo "$line" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms/step')")"
done
