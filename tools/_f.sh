#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_fused_pytest.log 2>&1; tail -3 gpurun_out/r3_fused_pytest.log
rm -f gpurun_out/r03_fused.txt
for v in "" "" "--inflight 1" "--config C3" "--second-start 1" "--second-start 2" "--warm"; do
  echo "== $v" >> gpurun_out/r03_fused.txt
  python3 bench.py $v --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value']), d['ms_per_step'], c.get('solved_per_step'), c.get('iters_mean_solved'), d['roofline']['frac_over_wall_clock'])" >> gpurun_out/r03_fused.txt
done
timeout -k 10 300 python tools/probe_variants.py 256 > gpurun_out/r03_fused_variants.txt 2>&1 || true
tail -5 gpurun_out/r03_fused_variants.txt
cat gpurun_out/r03_fused.txt
