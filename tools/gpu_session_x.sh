#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for k in 1 2; do
timeout 600 python tools/bh_small.py > gpurun_out/bh_small_$k.json 2> gpurun_out/bh_small.err
BH_NO_CPU=1 timeout 600 python tools/bench_bh.py > gpurun_out/bench_bh_$k.json 2> gpurun_out/bench_bh.err
timeout 600 python tools/frame_loop.py > gpurun_out/frame_loop_$k.log 2>&1
done
