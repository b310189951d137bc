#!/bin/bash
# Regenerates what profiles/ holds for one round on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 3000 -- 'TAG=r02 bash tools/reproduce_profiles.sh'   ;   then, locally:  python tools/pmc_summary.py r02
# and copy the gpurun_out/<TAG>_* files named in profiles/README.md into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_session.sh smoke tests bench bh prof pmc power shapes ubench
python tools/bench_bh.py > gpurun_out/${TAG:-r02}_bench_bh_tool.json 2> gpurun_out/bench_bh.err
python tools/frame_loop.py > gpurun_out/${TAG:-r02}_frame_loop_level1.txt 2>&1
python tools/pcie_inclusive.py > gpurun_out/${TAG:-r02}_pcie_inclusive.json 2>&1
