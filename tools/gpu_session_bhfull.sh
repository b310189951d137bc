#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python tools/bench_bh.py > gpurun_out/bench_bh.json 2> gpurun_out/bench_bh.err
timeout 600 python tools/bh_small.py > gpurun_out/bh_small.json 2> gpurun_out/bh_small.err
