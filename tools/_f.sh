#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3_steer_pytest.log 2>&1
tail -30 gpurun_out/r3_steer_pytest.log
timeout -k 10 300 python tools/probe_variants.py 256 > gpurun_out/r03_steer_variants.txt 2>&1
grep -c "status equal 1.0000" gpurun_out/r03_steer_variants.txt; grep -v "status equal 1.0000" gpurun_out/r03_steer_variants.txt | head
timeout -k 10 300 python tools/fuzz_gpu_vs_oracle.py 100 31 > gpurun_out/r03_steer_fuzz.txt 2>&1; tail -3 gpurun_out/r03_steer_fuzz.txt
