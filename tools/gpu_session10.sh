#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --steps 10 --workload two_galaxies --n 524288 --source-bits 16 > gpurun_out/bench_cfg5_1gpu.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --steps 10 --workload two_galaxies --n 524288 > gpurun_out/bench_cfg5_1gpu_fp32.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --steps 20 --n 65536 > gpurun_out/bench_cfg2.log 2>&1
