#!/bin/bash
# command-line session on the GPU box: the CLI golden tests, then T_e2e of a whole C2 step with the stage clocks on
set -u
tag=${1:-x}
mkdir -p gpurun_out/$tag
python3 -m pytest tests/test_gpu_cli.py -x -q > gpurun_out/$tag/cli_tests.log 2>&1; tail -15 gpurun_out/$tag/cli_tests.log
python3 tools/e2e_cli.py 50000000 > gpurun_out/$tag/e2e.log 2>&1; cat gpurun_out/$tag/e2e.log | cut -c1-260
