#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bh_device_tree.py tests/test_gpu_bh.py -m gpu -q -x > gpurun_out/pytest_tree.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_tree.log
for k in 1 2 3; do
BH_NO_CPU=1 timeout 600 python tools/bench_bh.py > gpurun_out/bench_bh_$k.json 2> gpurun_out/bench_bh.err
done
