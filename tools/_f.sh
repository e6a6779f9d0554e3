#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/r03_lanes4.txt
run() { echo "== $1" >> gpurun_out/r03_lanes4.txt; python3 bench.py $1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value']), d['ms_per_step'], c.get('solved_per_step'))" >> gpurun_out/r03_lanes4.txt; }
run "--inflight 8"
run "--inflight 10"
run "--inflight 12"
run "--inflight 16"
run "--inflight 16 --steps 20 --warmup 5"
run "--inflight 12 --steps 20 --warmup 5"
run "--inflight 16 --config C3"
run "--inflight 8 --config C3"
run "--inflight 16 --config C4"
run "--inflight 8 --config C4"
run "--inflight 16 --config C5"
cat gpurun_out/r03_lanes4.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "lanes or handles or multi" 2>&1 | tail -3
