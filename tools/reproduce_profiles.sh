#!/bin/bash
# Regenerates everything under profiles/ for one round on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 3000 -- 'bash tools/reproduce_profiles.sh'   ;   then, locally:  python tools/pmc_summary.py r01
# and copy the files listed at the end from gpurun_out/ into profiles/ (what each file is: profiles/README.md).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
set -x
bash tools/gpu_session4.sh            # gpu tests, bench.py (N=1), rocprofv3 --kernel-trace --stats, six PMC passes
bash tools/gpu_session_bhprof.sh      # rocprofv3 kernel stats: Barnes-Hut, strict, fp16-source paths
bash tools/gpu_session_bhfinal.sh     # Barnes-Hut: bench_bh.json, bh_small.json, frame loop, kernel stats, PMC summary of the walk
NB_BH_TREE=device NB_DRAW=device python tools/frame_loop.py > gpurun_out/frame_loop_dev.log 2>&1
bash tools/gpu_session_strict.sh      # bit-exact mode: bench.py --mode strict at several N, kernel stats
python tools/bh_strict_probe.py > gpurun_out/bh_strict.json 2> gpurun_out/bh_strict.err
python tools/sweep_shapes.py > gpurun_out/shapes.log 2>&1
VARIANTS=1,5 python tools/sweep_shapes.py > gpurun_out/shapes_v15.log 2>&1
python tools/pcie_inclusive.py > gpurun_out/pcie.log 2>&1
python tools/launch_gap.py > gpurun_out/launch_gap.log 2>&1
tools/ubench_valu > gpurun_out/ubench.log 2>&1
tools/ubench_banks > gpurun_out/ubench_banks.log 2>&1
