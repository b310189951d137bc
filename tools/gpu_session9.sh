#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for v in 1 4; do
B="python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 --variant $v --bpt 4 --jsplit 32"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmcv${v}_sq -o p -- $B > $R/gpurun_out/pmcv${v}_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcv${v}_grbm -o p -- $B > $R/gpurun_out/pmcv${v}_grbm.log 2>&1
done
