#!/bin/bash
# profile session on the GPU box (through gpurun): kernel stats + counters of the default bench (both index layouts in the trace),
# ALU busy counters, host -> device upload methods.   tools/gpu_prof.sh <tag>
set -u
tag=${1:-x}
export TMPDIR=/tmp
tools/profile_round.sh $tag > gpurun_out/prof_$tag.log 2>&1; tail -25 gpurun_out/prof_$tag.log
tools/pmc_busy.sh $tag > gpurun_out/busy_$tag.log 2>&1
python3 - <<'PY'
import numpy as np, os
p='/dev/shm/upload_bench.bin'
if not os.path.exists(p):
    a=np.random.default_rng(1).integers(0,255,size=6<<30,dtype=np.uint8); a.tofile(p)
PY
tools/upload_bench /dev/shm/upload_bench.bin 6 > gpurun_out/upload_bench_$tag.txt 2>&1; cat gpurun_out/upload_bench_$tag.txt; rm -f /dev/shm/upload_bench.bin
