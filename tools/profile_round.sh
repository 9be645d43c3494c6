#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01_c
# Writes raw output under /tmp/prof_<tag>/ and the condensed summaries that get committed
# under gpurun_out/profiles_<tag>/ (copy those into profiles/).
# Counters are collected in their own runs (no trace domains besides --kernel-trace), FETCH_SIZE and
# WRITE_SIZE in separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
tag=${1:-rXX}
root=$(pwd)
raw=/tmp/prof_$tag                 # (the raw rocprofv3 output stays on the box: gpurun brings back 64 MiB at most)
out=gpurun_out/profiles_$tag
mkdir -p $raw $out
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 2 --warmup 1 --cpu-baseline-secs 0 --no-host-leg --no-live-traffic --no-other-layout --no-rccl-world1"      # (the layout the policy picks, nothing else in the trace)

rocprofv3 --kernel-trace --stats --output-format csv -d $raw/stats -o run -- $BENCH > $raw/stats.log 2>&1
grep '^{' $raw/stats.log | tail -1 > $out/${tag}_bench_line_under_rocprof.json
python3 tools/summarize_prof.py stats $raw/stats > $out/${tag}_kernel_stats.csv

rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $raw/fetch -o run -- $BENCH > $raw/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $raw/write -o run -- $BENCH > $raw/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU \
    --output-format csv -d $raw/sq -o run -- $BENCH > $raw/sq.log 2>&1
python3 tools/summarize_prof.py pmc $raw/fetch $raw/write $raw/sq | grep -v 'k_sa_\|wrapper\|scan' > $out/${tag}_pmc_summary.csv

# FETCH_SIZE calibration in our own access patterns
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $raw/calib -o run -- $root/tools/rand_access_bench calib > $raw/calib.log 2>&1
python3 tools/summarize_prof.py pmc $raw/calib > $out/${tag}_fetch_calibration.csv
grep calib: $raw/calib.log >> $out/${tag}_fetch_calibration.csv

# a plain run for the bench line without profiler overhead, with the CPU baseline
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_line.json 2> $raw/bench.log      # (the driver's command)
tail -2 $raw/bench.log
cat $out/${tag}_kernel_stats.csv
