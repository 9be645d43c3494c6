#!/bin/bash
# How busy the vector / scalar ALUs are per kernel (run through gpurun from the repo root):
#   tools/pmc_busy.sh <tag> [extra bench.py arguments]
set -u
tag=${1:-x}
shift
raw=/tmp/busy_$tag
out=gpurun_out/busy_$tag
mkdir -p $raw $out
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 2 --warmup 1 --cpu-baseline-secs 0 --no-host-leg --no-live-traffic --no-other-layout --no-rccl-world1"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $raw/p1 -o run -- $BENCH "$@" > $raw/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $raw/p2 -o run -- $BENCH "$@" > $raw/p2.log 2>&1
python3 tools/summarize_prof.py pmc $raw/p1 $raw/p2 | grep -v 'k_sa_\|wrapper\|scan\|k_build\|k_check\|k_w_' > $out/busy.csv
rm -rf $raw/p1 $raw/p2
grep "k_wave\|k_flat\|k_search_a_ilp\|k_search_b" $out/busy.csv
