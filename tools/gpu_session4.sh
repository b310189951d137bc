#!/bin/bash
# official bench line + rocprof kernel stats + PMC passes (each in its own run, kernel-trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/rocprof_stats.log 2>&1
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
B="python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- $B > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- $B > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq1 -o p -- $B > $R/gpurun_out/pmc_sq1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $R/gpurun_out/pmc_sq2 -o p -- $B > $R/gpurun_out/pmc_sq2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $R/gpurun_out/pmc_grbm -o p -- $B > $R/gpurun_out/pmc_grbm.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $R/gpurun_out/pmc_tcc -o p -- $B > $R/gpurun_out/pmc_tcc.log 2>&1
cd $R; ls -R gpurun_out/pmc_* gpurun_out/prof_stats | head -60 > gpurun_out/pmc_ls.log
