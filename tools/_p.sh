#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for c in C2 C3 C4; do
  timeout -k 10 330 bash tools/profile_round.sh r03_$c --config $c > gpurun_out/prof_r03_$c.log 2>&1
  echo "profile $c done"; grep -E "WARNING|^FP64|total:" gpurun_out/prof_r03_$c/pmc.txt
done
python3 bench.py --config C5 > gpurun_out/r03_C5_bench.json 2> gpurun_out/r03_C5_bench.err
rm -f gpurun_out/r03_variants.txt
for v in "--warm" "--integrator rk4" "--start-steer 0" "--second-start 0" "--second-start 0 --start-steer 0" "--second-start 2" "--inflight 1" "--inflight 3" "--inflight 6" "--inflight 8" "--no-restoration" "--config C5 --second-start 0" "--config C5 --second-start 1" "--steps 20 --warmup 5"; do
  echo "== $v" >> gpurun_out/r03_variants.txt
  python3 bench.py $v --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value']), d['ms_per_step'], c.get('solved_per_step'), c.get('iters_mean_solved'), c.get('failure_rate_steps'), c.get('scenes_all_steps_solved'), c.get('scenes_with_collision'))" >> gpurun_out/r03_variants.txt
done
cat gpurun_out/r03_variants.txt
