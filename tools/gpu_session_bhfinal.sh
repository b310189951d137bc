#!/bin/bash
# Barnes-Hut evidence for profiles/: bench, kernel stats, PMC summary of the walk, small-N timings, level-1 frame loop
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
timeout 900 python tools/bench_bh.py > gpurun_out/bench_bh.json 2> gpurun_out/bench_bh.err
timeout 600 python tools/bh_small.py > gpurun_out/bh_small.json 2> gpurun_out/bh_small.err
timeout 600 python tools/frame_loop.py > gpurun_out/frame_loop.log 2>&1
export TMPDIR=/tmp
cd /tmp
BH_NO_CPU=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bh -o bh -- python $R/tools/bench_bh.py > $R/gpurun_out/rocprof_bh.log 2>&1
cd $R
find gpurun_out/prof_bh -name "*kernel_stats.csv" -exec cp {} gpurun_out/bh_kernel_stats.csv \;
rm -rf gpurun_out/prof_bh
bash tools/gpu_session_bhpmc.sh
