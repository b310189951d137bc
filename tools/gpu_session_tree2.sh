#!/bin/bash
# device tree rebuild check: tests, BH bench, kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bh_device_tree.py tests/test_gpu_bh.py -m gpu -q -x > gpurun_out/pytest_tree.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_tree.log
timeout 600 python tools/bench_bh.py > gpurun_out/bench_bh.json 2> gpurun_out/bench_bh.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bh -- python $GRAFT_REPO_ROOT/tools/bench_bh.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_bh -name "*kernel_stats.csv" -exec cp {} gpurun_out/bh_kernel_stats.csv \;
rm -rf gpurun_out/prof_bh
