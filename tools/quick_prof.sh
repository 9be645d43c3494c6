#!/bin/bash
# Kernel times of one short bench run under rocprofv3 (run through gpurun from the repo root):
#   tools/quick_prof.sh <tag> [extra bench.py arguments]
# Writes gpurun_out/quick_<tag>/kernel_stats.csv and the bench line.
set -u
tag=${1:-x}
shift
raw=gpurun_out/quick_$tag
mkdir -p $raw
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/stats -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-baseline-secs 0 --no-host-leg --no-live-traffic "$@" > $raw/stats.log 2>&1
grep '^{' $raw/stats.log | tail -1 > $raw/bench_line.json
python3 tools/summarize_prof.py stats $raw/stats > $raw/kernel_stats.csv
rm -rf $raw/stats
head -14 $raw/kernel_stats.csv
