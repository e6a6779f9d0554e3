mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_parity_evidence.py 2>&1 | tail -40 > gpurun_out/r3_pytest2.log; tail -30 gpurun_out/r3_pytest2.log
: > gpurun_out/r3_bench2.log
for c in C2 C3 C4; do
  timeout -k 10 300 python bench.py --config $c --steps 60 --warmup 5 --no-cpu-baseline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$c', round(d['value']), round(d['ms_per_step'],3), 'solved/step', c['solved_per_step'], 'hist', c['status_histogram_rank0'], 'iters mean solved', c['iters_mean_solved'], 'max', c['iters_max'], 'unsolved share', round(c['iters_share_of_unsolved'],3), 'e2e', d['end_to_end'] and round(d['end_to_end']['value']))" >> gpurun_out/r3_bench2.log
done
timeout -k 10 300 python bench.py --config C2 --steps 60 --warmup 5 --no-cpu-baseline --inflight 1 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 inflight 1', round(d['value']), round(d['ms_per_step'],3))" >> gpurun_out/r3_bench2.log
cat gpurun_out/r3_bench2.log
