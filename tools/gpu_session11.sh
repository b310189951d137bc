#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_brute.py -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
rm -f gpurun_out/sweep4.log
for v in 1 5; do for b in 2 4; do for s in 4 8 16 32 64; do
  echo "variant=$v" >> gpurun_out/sweep4.log
  timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --variant $v --bpt $b --jsplit $s >> gpurun_out/sweep4.log 2>&1
done; done; done
