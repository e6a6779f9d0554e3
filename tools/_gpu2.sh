mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_parity_evidence.py > gpurun_out/r3_pytest4.log 2>&1; tail -12 gpurun_out/r3_pytest4.log
: > gpurun_out/r3_bench2.log
for c in "--config C2" "--config C3" "--config C4" "--config C2 --integrator rk4" "--config C2 --inflight 1"; do
  timeout -k 10 300 python bench.py $c --steps 60 --warmup 5 --no-cpu-baseline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$c', round(d['value']), round(d['ms_per_step'],3), 'solved/step', c['solved_per_step'], 'hist', c['status_histogram_rank0'], 'iters mean solved', round(c['iters_mean_solved'],2), 'max', c['iters_max'], 'unsolved share', round(c['iters_share_of_unsolved'],3), 'e2e', d['end_to_end'] and round(d['end_to_end']['value']))" >> gpurun_out/r3_bench2.log
done
cat gpurun_out/r3_bench2.log
