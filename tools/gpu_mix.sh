#!/bin/bash
# mixed session: CLI tests, e2e timeline, quick kernel-only benches (3 M and 50 M reads)
set -u
tag=${1:-x}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_cli.py -x -q > gpurun_out/$tag/cli_tests.log 2>&1; tail -12 gpurun_out/$tag/cli_tests.log
python3 tools/e2e_cli.py 50000000 > gpurun_out/$tag/e2e.log 2>&1; grep -v "^\[" gpurun_out/$tag/e2e.log | cut -c1-200; grep "^\[" gpurun_out/$tag/e2e.log | cut -c1-200
python3 bench.py --reads 3000000 --no-host-leg --cpu-baseline-secs 0 --no-live-traffic > gpurun_out/$tag/b3m.json 2> gpurun_out/$tag/b3m.err
python3 bench.py --no-host-leg --cpu-baseline-secs 0 --no-live-traffic > gpurun_out/$tag/b50m.json 2> gpurun_out/$tag/b50m.err
python3 - <<PY
import json
for f in ('b3m','b50m'):
    d=json.load(open('gpurun_out/$tag/%s.json'%f))
    for k,v in d['layouts'].items():
        if isinstance(v,dict): print(f,k,'kernel-only %.1f M reads/s, %.2f ms/step'%(v['value_kernel_only']/1e6,v['ms_per_step_kernel_only']),v['device_ms_per_step'])
PY
