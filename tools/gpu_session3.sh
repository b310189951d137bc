#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/sweep_shapes.py > gpurun_out/shapes.log 2>&1
timeout 900 python tools/bench_bh.py > gpurun_out/bench_bh.log 2>&1
