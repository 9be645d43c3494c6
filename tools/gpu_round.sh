#!/bin/bash
# One GPU-box session of a development round (run through gpurun from the repo root): the GPU test suite, the calibration of the
# FETCH_SIZE counter in our access patterns, and the default bench line.  Output under gpurun_out/<tag>/.
#   tools/gpu_round.sh <tag> [tests|notests]
set -u
tag=${1:-x}
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
if [ "${2:-tests}" = tests ]; then
    python3 -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -3 $O/gputests.log
fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -o run -- $PWD/tools/rand_access_bench calib > $O/calib.log 2>&1
python3 tools/summarize_prof.py pmc $O/calib > $O/fetch_calibration.csv; grep calib: $O/calib.log >> $O/fetch_calibration.csv; rm -rf $O/calib
cat $O/fetch_calibration.csv
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -5 $O/bench.err; head -c 3000 $O/bench.json; echo
