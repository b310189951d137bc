#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
VARIANTS=1,5 timeout 1200 python tools/sweep_shapes.py > gpurun_out/shapes_v15.log 2>&1
cd /tmp
for v in 1 5; do
B="python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 --variant $v --bpt 4 --jsplit 32"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmcw${v}_sq -o p -- $B > $R/gpurun_out/pmcw${v}_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcw${v}_grbm -o p -- $B > $R/gpurun_out/pmcw${v}_grbm.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcw${v}_fetch -o p -- $B > $R/gpurun_out/pmcw${v}_fetch.log 2>&1
done
