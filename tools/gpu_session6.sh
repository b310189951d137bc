#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_bh.py -m gpu -q > gpurun_out/pytest_bh.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_bh.log
timeout 900 python tools/bench_bh.py > gpurun_out/bench_bh.log 2>&1
