#!/bin/bash
# first GPU session: smoke, parity tests, VALU microbench, launch-shape sweep, bench, rocprof
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit|Max Clock" | head -8 > gpurun_out/rocminfo.log
nproc > gpurun_out/nproc.log; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/nproc.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 tools/ubench_valu > gpurun_out/ubench.log 2>&1
for v in 0 2; do for b in 1 2 4; do for s in 0 1 2 4 8 16; do
  timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --variant $v --bpt $b --jsplit $s >> gpurun_out/sweep.log 2>&1
done; done; done
timeout 600 python bench.py > gpurun_out/bench.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_r01" -o bench -- python "$OLDPWD/bench.py" --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1 )
ls -R gpurun_out/prof_r01 | head -30 >> gpurun_out/rocprof.log
