#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bh.py tests/test_gpu_bh_device_tree.py tests/test_gpu_full_size_configs.py -m gpu -q > gpurun_out/pytest_bh.log 2>&1; echo rc=$? >> gpurun_out/pytest_bh.log
for k in 1 2 3; do
BH_NO_CPU=1 timeout 600 python tools/bench_bh.py > gpurun_out/bench_bh_$k.json 2> gpurun_out/bench_bh.err
done
NBX_TIMING=1 timeout 300 python tools/bh_steps.py host 6 > gpurun_out/bh_timing.log 2>&1
