# GPU regression + throughput probe used during kernel work:  bash tools/gpu_check.sh [configs...]
set -e
timeout -k 10 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for c in "${@:-C2}"; do
  timeout -k 10 300 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', round(d['value']), round(d['ms_per_step'],3), d['config'].get('solved_per_step'))"
done
timeout -k 10 120 python tools/probe_batch.py 30 32768 3
