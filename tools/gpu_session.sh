#!/bin/bash
# One session on the GPU box (run through gpurun from the repo root; everything lands under gpurun_out/<tag>/):
#   tools/gpu_session.sh <tag> <what> [<what> ..]
#     tests     the whole GPU suite (python -m pytest tests -m gpu)
#     cli       the command-line golden tests only
#     bench     python bench.py (the driver's command) -> bench.json / bench.err
#     quick     kernel-only bench lines of 3 M and 50 M reads per step, both index layouts
#     calib     FETCH_SIZE in our access patterns (tools/rand_access_bench calib under rocprofv3 --pmc)
#     prof      tools/profile_round.sh + tools/pmc_busy.sh (kernel stats, counters, ALU busy) -> gpurun_out/profiles_<tag>/
#     e2e       T_e2e of `biokanga align` on a whole C2 step with the stage clocks on (tools/e2e_cli.py)
#     e2e_csv   the same with -M0 (the reference's default CSV) and with -o ending in .sam.gz: the writers that all threads share since round 4
#     e2e_gz    the same on 20 M reads, from the plain file, its .gz and its .bgz, and the gzip'd ones through gzread as well
#     inflate   the loaders' DEFLATE decoder on the box's host: one thread, 2 .. 16 threads, zlib (tools/inflate_bench.sh; no GPU work)
#     upload    host -> device upload methods, wall-clock and CPU seconds (tools/upload_bench)
#     index_e2e `biokanga index` end to end on a 3.1 Gbp FASTA in /dev/shm with the stage clocks (tools/index_e2e.py)
#     quota     two bench ranks / four command-line contexts on the one GPU, unconstrained and under `taskset -c 0-3`
set -u
tag=${1:-x}
shift
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    tests) timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -4 $O/gputests.log ;;
    cli)   timeout 600 python3 -m pytest tests/test_gpu_cli.py -x -q > $O/cli_tests.log 2>&1; tail -12 $O/cli_tests.log ;;
    bench) timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -4 $O/bench.err; head -c 2500 $O/bench.json; echo ;;
    quick)
      python3 bench.py --reads 3000000 --no-host-leg --cpu-baseline-secs 0 --no-live-traffic > $O/b3m.json 2> $O/b3m.err
      python3 bench.py --no-host-leg --cpu-baseline-secs 0 --no-live-traffic > $O/b50m.json 2> $O/b50m.err
      python3 - "$O" <<'PY'
import json, sys
for f in ("b3m", "b50m"):
    d = json.load(open(f"{sys.argv[1]}/{f}.json"))
    for k, v in d["layouts"].items():
        if isinstance(v, dict):
            print(f, k, "kernel-only %.1f M reads/s, %.2f ms/step" % (v["value_kernel_only"] / 1e6, v["ms_per_step_kernel_only"]), v["device_ms_per_step"], v.get("window_array_built", ""))
PY
      ;;
    calib)
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -o run -- $PWD/tools/rand_access_bench calib > $O/calib.log 2>&1
      python3 tools/summarize_prof.py pmc $O/calib > $O/fetch_calibration.csv; grep calib: $O/calib.log >> $O/fetch_calibration.csv; rm -rf $O/calib
      cat $O/fetch_calibration.csv ;;
    prof)  tools/profile_round.sh $tag > $O/profile_round.log 2>&1; tail -25 $O/profile_round.log; tools/pmc_busy.sh $tag > $O/busy.log 2>&1; tail -8 $O/busy.log ;;
    e2e)   timeout 600 python3 tools/e2e_cli.py 50000000 > $O/e2e.log 2>&1; grep -v "^\[" $O/e2e.log | cut -c1-200; grep "^\[" $O/e2e.log | cut -c1-200 ;;
    e2e_csv) timeout 560 python3 tools/e2e_cli.py 20000000 --quiet -- -M0 > $O/e2e_csv.log 2>&1; grep -a "T_e2e\|load " $O/e2e_csv.log | cut -c1-220
             BK_E2E_OUT=out.sam.gz timeout 560 python3 tools/e2e_cli.py 20000000 --quiet > $O/e2e_samgz.log 2>&1; grep -a "T_e2e\|load " $O/e2e_samgz.log | cut -c1-220 ;;
    e2e_gz) timeout 560 python3 tools/e2e_cli.py 20000000 --gz --quiet > $O/e2e_gz.log 2>&1; grep -a "gzip copies\|T_e2e\|load " $O/e2e_gz.log | cut -c1-220 ;;
    inflate) BK_INFLATE_DEBUG=1 timeout 300 bash tools/inflate_bench.sh 600 > $O/inflate_bench.txt 2>&1; grep -v "piece at" $O/inflate_bench.txt | tail -14 ;;
    upload)
      python3 -c "import numpy as np; np.random.default_rng(1).integers(0, 255, size=6 << 30, dtype=np.uint8).tofile('/dev/shm/upload_bench.bin')"
      tools/upload_bench /dev/shm/upload_bench.bin 6 > $O/upload_methods.txt 2>&1; cat $O/upload_methods.txt; rm -f /dev/shm/upload_bench.bin ;;
    index_e2e) timeout 1200 python3 tools/index_e2e.py 3100 --repeat 2 > $O/index_e2e.txt 2>&1; grep -v "^   .*host: at exit" $O/index_e2e.txt | cut -c1-200 | tail -60 ;;
    quota)
      # two ranks on the one GPU and four contexts of the command line, unconstrained and held to four CPUs (what a rank of an 8-GPU job
      # gets of the box's 16-CPU quota): the host side must not need more
      B="python3 bench.py --gpus 2 --force-device 0 --dist-backend gloo --genome-mbp 800 --reads 12000000 --cpu-baseline-secs 0 --no-live-traffic --no-other-layout --steps 3"
      timeout 300 $B > $O/quota_2ranks_free.json 2> $O/quota_2ranks_free.err
      timeout 300 taskset -c 0-3 $B > $O/quota_2ranks_4cpus.json 2> $O/quota_2ranks_4cpus.err
      python3 - "$O" <<'PY'
import json, sys
for f in ("quota_2ranks_free", "quota_2ranks_4cpus"):
    try:
        d = json.load(open(f"{sys.argv[1]}/{f}.json"))
        print(f, "host-in/host-out %.1f M reads/s, kernel-only %.1f M reads/s over %d ranks" % (d["value"] / 1e6, d["value_kernel_only"] / 1e6, d["n_gpus"]))
    except Exception as e:
        print(f, "failed:", e)
PY
      timeout 600 python3 tools/e2e_cli.py 20000000 --quiet --repeat 2 -- --devices 0,0,0,0 > $O/quota_cli_free.log 2>&1; grep -a "T_e2e\|load " $O/quota_cli_free.log | cut -c1-220
      timeout 600 taskset -c 0-3 python3 tools/e2e_cli.py 20000000 --quiet --repeat 2 -- --devices 0,0,0,0 > $O/quota_cli_4cpus.log 2>&1; grep -a "T_e2e\|load " $O/quota_cli_4cpus.log | cut -c1-220 ;;
    *) echo "gpu_session.sh: unknown step '$what'" ;;
  esac
done
