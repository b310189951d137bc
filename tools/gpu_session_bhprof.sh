#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
BH_NO_CPU=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bh -o bh -- python $R/tools/bench_bh.py > $R/gpurun_out/rocprof_bh.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_strict -o strict -- python $R/bench.py --no-cpu-baseline --mode strict --n 65536 --steps 5 --warmup 1 > $R/gpurun_out/rocprof_strict.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_half -o half -- python $R/bench.py --no-cpu-baseline --workload two_galaxies --n 524288 --source-bits 16 --steps 5 --warmup 1 > $R/gpurun_out/rocprof_half.log 2>&1
