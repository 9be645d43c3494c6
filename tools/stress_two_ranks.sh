#!/bin/bash
# the two-rank bench of tests/test_gpu_bench.py, N times: how often it fails, and the stderr of the failures
n=${1:-10}
mkdir -p gpurun_out/stress
fails=0
for i in $(seq 1 $n); do
  timeout 300 python3 bench.py --gpus 2 --force-device 0 --dist-backend gloo --genome-mbp 30 --reads 400000 --steps 1 --warmup 1 --cpu-baseline-secs 0 \
      --stream-batch 100000 --shard-check-reads 300000 --no-live-traffic > gpurun_out/stress/out_$i.json 2> gpurun_out/stress/err_$i.log
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "run $i: rc $rc"; grep -a "error\|Error\|abort\|exception" gpurun_out/stress/err_$i.log | head -5; else rm -f gpurun_out/stress/out_$i.json gpurun_out/stress/err_$i.log; fi
done
echo "$fails of $n runs failed"
