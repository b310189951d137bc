#!/bin/bash
# bit-exact mode: parity tests, throughput at a few sizes, rocprofv3 kernel stats at N = 65536
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
: > gpurun_out/strict_bench.jsonl
for n in 1024 10000 16384 65536 262144; do
timeout 600 python bench.py --no-cpu-baseline --mode strict --n $n --steps 5 --warmup 1 >> gpurun_out/strict_bench.jsonl 2>> gpurun_out/strict_bench.err
done
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_strict -o strict -- python $R/bench.py --no-cpu-baseline --mode strict --n 65536 --steps 5 --warmup 1 > $R/gpurun_out/rocprof_strict.log 2>&1
cd $R
find gpurun_out/prof_strict -name "*kernel_stats.csv" -exec cp {} gpurun_out/strict_kernel_stats.csv \;
rm -rf gpurun_out/prof_strict
