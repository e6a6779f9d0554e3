#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -q > gpurun_out/r3_tight.log 2>&1
tail -8 gpurun_out/r3_tight.log
for s in 31 32 33; do timeout -k 10 250 python tools/fuzz_gpu_vs_oracle.py 100 $s > gpurun_out/r03_fuzz_$s.txt 2>&1; tail -1 gpurun_out/r03_fuzz_$s.txt; grep -o "one side only [0-9]*" gpurun_out/r03_fuzz_$s.txt | sort | uniq -c; done
