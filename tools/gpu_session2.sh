#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
rm -f gpurun_out/sweep2.log
for v in 1 3; do for b in 2 4; do for s in 2 4 8 16 32; do
  echo "variant=$v" >> gpurun_out/sweep2.log
  timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --variant $v --bpt $b --jsplit $s >> gpurun_out/sweep2.log 2>&1
done; done; done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_r01" -o bench -- python "$OLDPWD/bench.py" --no-cpu-baseline --variant 1 --bpt 2 --jsplit 8 > "$OLDPWD/gpurun_out/rocprof.log" 2>&1 )
ls -R gpurun_out/prof_r01 | head -30 >> gpurun_out/rocprof.log
